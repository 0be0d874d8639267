"""Random draws of the sampling path: reproducible per sample, safe beside hipGraph captures on other host threads.

* ``with rng.seeded(seed, device):`` gives the CURRENT HOST THREAD a ``torch.Generator`` of its own: every draw the package makes on
  that thread inside the block (start latent, the eta = 1 per-step DDIM noise, the first-stage posterior sample, the random
  conditioning-frame index) comes from it.  The generation harness opens one per test batch, seeded from ``--seed`` and the
  batch's dataset index, so a sample gets the same noise whichever lane, rank or run produces it (the reference seeds once per
  process, ``seed_everything(seed + global_rank)``, main/trainer.py:62, and draws in data order).
* Without one the draws come from the device's default generator.  While a ``torch.cuda.graph`` capture is in progress that
  generator is in capture mode for EVERY thread: a ``torch.randn`` on another (non-capturing) stream then fails with "Offset
  increment outside graph capture encountered unexpectedly".  With several clips in flight (one host thread + HIP stream each) a
  lane may be capturing its graphs while another draws, so captures (``sampler._GraphedClip``) and default-generator draws take
  this one lock.  Draws with an explicit generator do not need it.
"""
import threading

import torch

LOCK = threading.RLock()
_TLS = threading.local()


def generator():
    """The generator installed for this thread by ``seeded`` (None: the device's default generator)."""
    return getattr(_TLS, "gen", None)


class seeded:
    """Context manager: draws of this host thread come from a fresh generator on ``device`` seeded with ``seed``."""

    def __init__(self, seed, device):
        self.gen = torch.Generator(device=device)
        self.gen.manual_seed(int(seed) & 0x7FFFFFFFFFFFFFFF)
        # a host-side draw inside the block stays per-sample reproducible too: a CPU generator seeded from the same value
        self.host = self.gen if torch.device(device).type == "cpu" else torch.Generator(device="cpu")
        if self.host is not self.gen:
            self.host.manual_seed(int(seed) & 0x7FFFFFFFFFFFFFFF)

    def __enter__(self):
        self.prev = (generator(), getattr(_TLS, "host", None))
        _TLS.gen, _TLS.host = self.gen, self.host
        return self.gen

    def __exit__(self, *exc):
        _TLS.gen, _TLS.host = self.prev
        return False


def _gen_for(device):
    g = generator()
    if g is None:
        return None
    kind = torch.device(device).type if device is not None else "cpu"
    if kind == g.device.type:
        return g
    if kind == "cpu":
        return getattr(_TLS, "host", None)          # the block's host generator (same seed): still one stream of draws per sample
    raise RuntimeError(f"rng: a draw on {kind!r} inside a seeded block whose generator lives on {g.device.type!r}")


def randn(*size, **kw):
    g = _gen_for(kw.get("device"))
    if g is not None:
        return torch.randn(*size, generator=g, **kw)
    with LOCK:
        return torch.randn(*size, **kw)


def randn_like(t, **kw):
    g = _gen_for(t.device)
    if g is not None:
        return torch.randn(t.shape, generator=g, dtype=kw.pop("dtype", t.dtype), device=t.device, **kw)
    with LOCK:
        return torch.randn_like(t, **kw)


def normal_(t):
    g = _gen_for(t.device)
    if g is not None:
        return t.normal_(generator=g)
    with LOCK:
        return t.normal_()


def randint(*a, **kw):
    g = _gen_for(kw.get("device"))
    if g is not None:
        return torch.randint(*a, generator=g, **kw)
    with LOCK:
        return torch.randint(*a, **kw)
