"""Frame-sharded single-clip mode (BASELINE.json configs[2], SURVEY.md section 8e "single-clip latency modes"): the frames of
every clip are split evenly over the ranks of a ``torch.distributed`` group (RCCL over xGMI on a node: backend "nccl"; the CPU-side
tests use gloo); every rank runs the UNet on its frames and the layers that mix frames exchange what they need:

  * ``TemporalConvBlock``: per convolution ONE collective -- an uneven all_to_all in which every rank sends its partial GroupNorm
    sums [b, 32 groups, 2] to everybody and its first / last (un-normalised) frame ONLY to the rank that holds the neighbouring frames
    (round 4; rounds 2-3 all-gathered every rank's two edge frames: world x the bytes a rank uses); every rank normalises the two
    halo frames it receives with the same clip-wide statistics;
  * ``TemporalTransformer`` GroupNorm over (t, h, w): one all_reduce of the sums;
  * temporal self-attention (per pixel over t) and the epipolar attention (over all t*h*w tokens): the K | V of a camera block's two
    attentions travel in one all_gather, the second temporal attention's in another;
  * the UNet output: one all_gather of the predicted noise, so that every rank runs the (tiny) DDIM update on whole clips.
All gathers are ``all_gather_into_tensor`` into one [world, ...] buffer, issued on the compute stream; which form of a collective a
backend gets is decided ONCE from its name (never by catching a failed collective: ranks that disagree would hang).

This is 140 collectives per forward (22 x 4 temporal convolutions + 17 norms + 16 + 17 + 1 K|V gathers + 1; round 2: ~250): the
frame-mixing layers are sequentially dependent, so it stays a latency mode for ONE clip on several GPUs -- independent clips
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


def _tensor_collectives(group):
    """True when the backend has the single-buffer device collectives (all_gather_into_tensor, all_to_all_single on device tensors):
    RCCL ("nccl").  Decided from the backend's name, once per group -- NOT by catching a failing collective: if a collective failed on
    some ranks only (an asynchronous RCCL error, a timeout), the ranks would diverge onto different collectives and hang, and the
    real error would be swallowed."""
    return dist.get_backend(group) == "nccl"


def _gather_into(x, world, group, state=None):
    """x (same shape on every rank) -> [world, *x.shape]: ONE all_gather_into_tensor into one buffer (no per-rank list, no stack
    copy) on RCCL; the list form on backends without the tensor form (gloo: the tests)."""
    x = x.contiguous()
    if x.dim() == 0:
        x = x.reshape(1)
    flat = torch.empty((world * x.shape[0],) + tuple(x.shape[1:]), dtype=x.dtype, device=x.device)   # the ranks' tensors back to back
    out = flat.view((world,) + tuple(x.shape))
    if _tensor_collectives(group):
        dist.all_gather_into_tensor(flat, x, group=group)
    else:
        dist.all_gather(list(out.unbind(0)), x, group=group)
    return out


class FrameShard:
    def __init__(self, group=None):
        if not dist.is_initialized():
            raise CcvError("FrameShard needs an initialised torch.distributed process group")
        self.group = group
        self.rank, self.world = dist.get_rank(group), dist.get_world_size(group)
        self.collectives = 0          # issued so far (tests assert the per-forward count)
        self._state = {}
        _first_collective(group)

    def all_reduce_sum(self, t):
        self.collectives += 1
        dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.group)
        return t

    def all_gather(self, x):
        """x (same shape on every rank) -> [world, *x.shape] (index r = rank r's tensor)."""
        self.collectives += 1
        return _gather_into(x, self.world, self.group, self._state)

    def neighbour_exchange(self, first, last, shared):
        """ONE collective: `shared` (a small tensor, e.g. partial GroupNorm sums) goes to every rank, `first` to rank - 1 and `last` to
        rank + 1 only.  Returns (every rank's `shared` [world, *shape], rank - 1's `last` or None, rank + 1's `first` or None).
        An all_to_all with per-peer sizes: under RCCL a group of point-to-point sends / receives over the xGMI links to the two
        neighbours -- a rank receives 2 frames instead of the 2 x world of an all-gather."""
        self.collectives += 1
        r, w = self.rank, self.world
        sh = shared.contiguous().view(torch.uint8).reshape(-1)
        sh = torch.cat([sh, sh.new_zeros((-sh.numel()) % 16)])                 # keep the frames 16-byte aligned behind it
        fb, lb = first.contiguous().view(torch.uint8).reshape(-1), last.contiguous().view(torch.uint8).reshape(-1)
        ns, nf = sh.numel(), fb.numel()
        send, in_sizes, out_sizes = [], [], []
        for d in range(w):
            parts = [sh]
            if d == r - 1:
                parts.append(fb)          # my first frame is the frame after rank - 1's last one
            if d == r + 1:
                parts.append(lb)
            send.extend(parts)
            in_sizes.append(sum(p.numel() for p in parts))
            out_sizes.append(ns + (nf if d in (r - 1, r + 1) else 0))
        inp = torch.cat(send)
        out = torch.empty(sum(out_sizes), dtype=torch.uint8, device=inp.device)
        if _tensor_collectives(self.group):
            dist.all_to_all_single(out, inp, out_sizes, in_sizes, group=self.group)
        else:
            # gloo (the tests): all_to_all on host tensors; the current stream is drained by the copy
            out_h = torch.empty(out.shape, dtype=torch.uint8)
            dist.all_to_all_single(out_h, inp.cpu(), out_sizes, in_sizes, group=self.group)
            out.copy_(out_h)
        got_shared, prev, nxt, off = [], None, None, 0
        for s_ in range(w):
            got_shared.append(out[off:off + shared.numel() * shared.element_size()])
            if s_ == r - 1:
                prev = out[off + ns:off + ns + nf].view(last.dtype).reshape(last.shape)
            if s_ == r + 1:
                nxt = out[off + ns:off + ns + nf].view(first.dtype).reshape(first.shape)
            off += out_sizes[s_]
        all_shared = torch.stack(got_shared, 0).view(shared.dtype).reshape((w,) + tuple(shared.shape))
        return all_shared, prev, nxt

    def all_gather_packed(self, tensors):
        """Several tensors of any dtypes in ONE collective: they travel as one byte buffer.  Returns, per input tensor, the gathered
        [world, *shape] tensor."""
        flat = [t.contiguous().view(torch.uint8).reshape(-1) for t in tensors]
        sizes = [f.numel() for f in flat]
        pad = [(-n) % 16 for n in sizes]                      # keep every part 16-byte aligned inside the buffer
        buf = torch.cat([torch.cat([f, f.new_zeros(p)]) if p else f for f, p in zip(flat, pad)])
        got = self.all_gather(buf)                            # [world, bytes]
        outs, off = [], 0
        for t, n, p in zip(tensors, sizes, pad):
            outs.append(got[:, off:off + n].contiguous().view(t.dtype).reshape((self.world,) + tuple(t.shape)))
            off += n + p
        return outs


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
        self._state = {}
        _first_collective(group)

    def exchange(self, mine):
        """this rank's noise prediction -> (e_cond, e_uncond)"""
        mine = mine.contiguous()
        # gloo stages device tensors through the host on streams of its own: drain the current stream first (_host_staged).  RCCL
        # enqueues the collective on the device in stream order: no drain, the host keeps running ahead (and the step can be captured).
        if _host_staged(self.group):
            torch.cuda.current_stream().synchronize()
        both = _gather_into(mine, 2, self.group, self._state)
        return both[0], both[1]


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
    def _frames_first(self, parts, b, hw, C):
        """[world, b, t_loc, hw, C] -> [(b T hw), C]; a view when b == 1 (one clip: the latency mode's case)."""
        if b == 1:
            return parts.reshape(self.T * hw, C)
        return parts.permute(1, 0, 2, 3, 4).reshape(b * self.T * hw, C)

    def gather_frames(self, rows, b, hw):
        """this rank's rows [(b t_loc hw), C] -> all frames [(b T hw), C]."""
        C = rows.shape[-1]
        return self._frames_first(self.shard.all_gather(rows.reshape(b, self.t_loc, hw, C)), b, hw, C)

    def gather_frames_multi(self, rows_list, b, hw):
        """Several row tensors (e.g. the K|V of a block's temporal attention and of its epipolar attention) in ONE collective."""
        shaped = [r.reshape(b, self.t_loc, hw, r.shape[-1]) for r in rows_list]
        return [self._frames_first(g, b, hw, g.shape[-1]) for g in self.shard.all_gather_packed(shaped)]

    def halo_frames(self, parts_edges):
        """parts_edges [world, 2, b, hw, C] (every rank's first and last local frame) -> (previous rank's last frame, next rank's
        first frame), each [b, hw, C]; zeros at the clip's ends (the convolution's padding)."""
        r, w = self.shard.rank, self.shard.world
        prev = parts_edges[r - 1][1] if r > 0 else torch.zeros_like(parts_edges[0][0])
        nxt = parts_edges[r + 1][0] if r < w - 1 else torch.zeros_like(parts_edges[0][0])
        return prev, nxt

    def with_halo(self, rows, b, hw):
        """[(b t_loc hw), C] -> [(b (t_loc + 2) hw), C]: the previous rank's last frame in front and the next rank's first frame
        behind every clip's local frames (zeros at the clip's ends: the convolution's padding)."""
        C = rows.shape[-1]
        z = rows.reshape(b, self.t_loc, hw, C)
        _, prev, nxt = self.shard.neighbour_exchange(z[:, 0], z[:, -1], rows.new_zeros(4))
        prev = torch.zeros_like(z[:, 0]) if prev is None else prev
        nxt = torch.zeros_like(z[:, 0]) if nxt is None else nxt
        return torch.cat([prev[:, None], z, nxt[:, None]], 1).reshape(b * (self.t_loc + 2) * hw, C)

    def edges_and_sums(self, rows, sums, b, hw):
        """ONE collective for a temporal convolution's GroupNorm + halo (FrameShard.neighbour_exchange): the partial GroupNorm sums
        [b, 64] of every rank, and of the UN-normalised rows only the two frames this rank needs -- rank - 1's last and rank + 1's first.
        Returns (previous frame, next frame, summed statistics, is_first, is_last): the caller normalises its own rows and the two halo
        frames with the same clip-wide statistics (zeros stand in at the clip's ends and are overwritten by the caller)."""
        C = rows.shape[-1]
        z = rows.reshape(b, self.t_loc, hw, C)
        all_sums, prev, nxt = self.shard.neighbour_exchange(z[:, 0], z[:, -1], sums)
        prev = torch.zeros_like(z[:, 0]) if prev is None else prev
        nxt = torch.zeros_like(z[:, 0]) if nxt is None else nxt
        return prev, nxt, all_sums.sum(0), self.shard.rank == 0, self.shard.rank == self.shard.world - 1

    def inner(self, rows_ext, b, hw):
        C = rows_ext.shape[-1]
        return rows_ext.reshape(b, self.t_loc + 2, hw, C)[:, 1:-1].reshape(b * self.t_loc * hw, C).contiguous()

    def local_frames(self, rows, nb, hw):
        """rows of all T frames [(nb T hw), C] -> this rank's [(nb t_loc hw), C]."""
        C = rows.shape[-1]
        return rows.reshape(nb, self.T, hw, C)[:, self.f0:self.f0 + self.t_loc].reshape(nb * self.t_loc * hw, C).contiguous()


__all__ = ["FrameShard", "FrameCtx", "CfgSplit", "current"]
