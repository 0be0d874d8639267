"""Frame-sharded single-clip mode (BASELINE.json configs[2], SURVEY.md section 8e "single-clip latency modes"): the frames of
every clip are split evenly over the ranks of a ``torch.distributed`` group (RCCL over xGMI on a node: backend "nccl"; the CPU-side
tests use gloo); every rank runs the UNet on its frames and the layers that mix frames exchange what they need:

  * ``TemporalConvBlock`` / ``TemporalTransformer`` GroupNorm over (t, h, w): one all_reduce of the [b, 32 groups, 2] sums;
  * ``TemporalConvBlock`` convolutions over t (kernel 3): the neighbours' edge frames (one all_gather of two frames per rank);
  * temporal self-attention (per pixel over t) and the epipolar attention (over all t*h*w tokens): one all_gather of K | V;
  * the UNet output: one all_gather of the predicted noise, so that every rank runs the (tiny) DDIM update on whole clips.

This is ~250 collectives per forward: a latency mode for ONE clip on several GPUs, not a throughput mode -- independent clips
shard over GPUs with no collective at all (bench.py).  The reference has no counterpart (its only parallelism is Lightning's
data-parallel test loop, 02_generate_videos.py:173,318).
"""
import threading

import torch
import torch.distributed as dist

from .lib import CcvError

_CUR = threading.local()


def current():
    """The FrameCtx of the forward running on this thread, or None."""
    return getattr(_CUR, "ctx", None)


def _first_collective(group):
    """One tiny device collective right away: the backend sets up its device-side resources (communicator, streams, pinned
    staging buffers) here instead of in the middle of the first sampling step -- with gloo that first call hung when it came
    right after a hipGraph replay of the same process (seen in the one-GPU rehearsal of `bench.py --cfg-split`)."""
    if torch.cuda.is_available():
        t = torch.zeros(8, device=torch.device("cuda", torch.cuda.current_device()))
        dist.all_reduce(t, group=group)
        torch.cuda.current_stream().synchronize()


def _host_staged(group):
    """gloo stages device tensors through the host on streams of its own: work still in flight on the current stream (a hipGraph
    replay) must be drained before such a collective is issued (seen in the one-GPU rehearsal of `bench.py --cfg-split`: both ranks
    stuck in all_gather; drained first it runs).  RCCL ("nccl") enqueues on the device, in stream order: no drain."""
    return dist.get_backend(group) != "nccl"


def broadcast_from_first(t, group=None):
    """t of the group's first rank on every rank (in place; returns t)."""
    if _host_staged(group) and t.is_cuda:
        torch.cuda.current_stream(t.device).synchronize()
    dist.broadcast(t, src=dist.get_global_rank(group, 0) if group is not None else 0, group=group)
    return t


class FrameShard:
    def __init__(self, group=None):
        if not dist.is_initialized():
            raise CcvError("FrameShard needs an initialised torch.distributed process group")
        self.group = group
        self.rank, self.world = dist.get_rank(group), dist.get_world_size(group)
        _first_collective(group)

    def all_reduce_sum(self, t):
        dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.group)
        return t

    def all_gather(self, x):
        """x (same shape on every rank) -> list of the ranks' tensors."""
        x = x.contiguous()
        parts = [torch.empty_like(x) for _ in range(self.world)]
        dist.all_gather(parts, x, group=self.group)
        return parts


class CfgSplit:
    """Classifier-free guidance over TWO ranks (SURVEY.md section 8e "CFG split"): rank 0 runs the conditional forward of every step,
    rank 1 the unconditional one; one all_gather of the 256 KB noise prediction per step, after which both ranks run the guidance
    + DDIM update on identical data.  ``model.cfg_split = CfgSplit(group)``; with ``use_graph=True`` each rank's forward is a
    hipGraph and only the exchange and the two-launch update stay outside it."""

    def __init__(self, group=None):
        if not dist.is_initialized():
            raise CcvError("CfgSplit needs an initialised torch.distributed process group")
        self.group = group
        self.rank, self.world = dist.get_rank(group), dist.get_world_size(group)
        if self.world != 2:
            raise CcvError(f"the CFG split is for exactly 2 ranks (conditional / unconditional), the group has {self.world}")
        _first_collective(group)

    def exchange(self, mine):
        """this rank's noise prediction -> (e_cond, e_uncond)"""
        mine = mine.contiguous()
        # Drain the stream first.  gloo stages device tensors through the host on streams of its own, and ordered behind a hipGraph
        # replay still in flight on the current stream its copy never started (one-GPU rehearsal of `bench.py --cfg-split`: both
        # ranks stuck in all_gather; drained first it runs).  RCCL enqueues on the device and should not need it, but that path has not
        # run anywhere yet; the wait costs the host's run-ahead over one ~17 ms step (the next replay needs the exchanged result anyway).
        torch.cuda.current_stream().synchronize()
        parts = [torch.empty_like(mine), torch.empty_like(mine)]
        dist.all_gather(parts, mine, group=self.group)
        return parts[0], parts[1]


class FrameCtx:
    """One sharded forward: T frames in all, this rank holds [f0, f0 + t_loc)."""

    def __init__(self, shard, T):
        if T % shard.world:
            raise CcvError(f"{T} frames do not split evenly over {shard.world} ranks")
        self.shard, self.T = shard, T
        self.t_loc = T // shard.world
        self.f0 = shard.rank * self.t_loc

    def __enter__(self):
        self._prev = current()
        _CUR.ctx = self
        return self

    def __exit__(self, *exc):
        _CUR.ctx = self._prev

    # ---- exchanges (rows are token-major [(b t hw), C]) ------------------------------------------------------------------
    def gather_frames(self, rows, b, hw):
        """this rank's rows [(b t_loc hw), C] -> all frames [(b T hw), C]."""
        C = rows.shape[-1]
        parts = self.shard.all_gather(rows.reshape(b, self.t_loc, hw, C))
        return torch.stack(parts, 1).reshape(b * self.T * hw, C)

    def with_halo(self, rows, b, hw):
        """[(b t_loc hw), C] -> [(b (t_loc + 2) hw), C]: the previous rank's last frame in front and the next rank's first frame
        behind every clip's local frames (zeros at the clip's ends: the convolution's padding)."""
        C = rows.shape[-1]
        z = rows.reshape(b, self.t_loc, hw, C)
        parts = self.shard.all_gather(torch.stack([z[:, 0], z[:, -1]], 0))          # [2, b, hw, C] per rank
        r, w = self.shard.rank, self.shard.world
        prev = parts[r - 1][1] if r > 0 else torch.zeros_like(z[:, 0])
        nxt = parts[r + 1][0] if r < w - 1 else torch.zeros_like(z[:, 0])
        return torch.cat([prev[:, None], z, nxt[:, None]], 1).reshape(b * (self.t_loc + 2) * hw, C)

    def inner(self, rows_ext, b, hw):
        C = rows_ext.shape[-1]
        return rows_ext.reshape(b, self.t_loc + 2, hw, C)[:, 1:-1].reshape(b * self.t_loc * hw, C).contiguous()

    def local_frames(self, rows, nb, hw):
        """rows of all T frames [(nb T hw), C] -> this rank's [(nb t_loc hw), C]."""
        C = rows.shape[-1]
        return rows.reshape(nb, self.T, hw, C)[:, self.f0:self.f0 + self.t_loc].reshape(nb * self.t_loc * hw, C).contiguous()


__all__ = ["FrameShard", "FrameCtx", "CfgSplit", "current"]
