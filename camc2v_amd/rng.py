"""Draws from the default CUDA generator, safe beside hipGraph captures on other host threads.

While a ``torch.cuda.graph`` capture is in progress the device's default generator is in capture mode for EVERY thread: a
``torch.randn`` on another (non-capturing) stream then fails with "Offset increment outside graph capture encountered
unexpectedly".  With several clips in flight (one host thread + HIP stream each, ``bench.py --lanes`` / ``generate.py --lanes``)
a lane may be capturing its graphs while another draws its start latent or a step's noise, so captures (``sampler._GraphedClip``)
and the package's own default-generator draws take this one lock.  Draws with an explicit ``generator=`` do not need it.
"""
import threading

import torch

LOCK = threading.RLock()


def randn(*size, **kw):
    with LOCK:
        return torch.randn(*size, **kw)


def randn_like(t, **kw):
    with LOCK:
        return torch.randn_like(t, **kw)


def normal_(t):
    with LOCK:
        return t.normal_()


def randint(*a, **kw):
    with LOCK:
        return torch.randint(*a, **kw)
