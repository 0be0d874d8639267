"""Host logic of the frame-shard exchanges (camc2v_amd/parallel.py) on CPU tensors over gloo with EIGHT ranks -- the 8 x MI355X
layout of BASELINE.json configs[2]: 16 frames, 2 per rank, i.e. every local frame is an edge frame.  The GPU box of the test pool
admits at most 6 processes on its one card, so the 8-rank case of the HIP path itself cannot run there; what 8 ranks add over the
2- and 4-rank GPU tests (tests/test_frame_shard_gpu.py) is exactly this byte plumbing: who receives which frame, the packed
per-peer sizes of the uneven all_to_all, the clip-wide GroupNorm sums."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, T, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from camc2v_amd import parallel
    b, hw, C = 2, 6, 8
    g = torch.Generator().manual_seed(5)
    full = torch.randn(b, T, hw, C, generator=g).to(torch.float16)            # the whole clip, identical on every rank
    shard = parallel.FrameShard()
    with parallel.FrameCtx(shard, T) as fc:
        assert fc.t_loc == T // world and fc.f0 == rank * fc.t_loc
        mine = full[:, fc.f0:fc.f0 + fc.t_loc].reshape(b * fc.t_loc * hw, C).contiguous()
        sums = torch.arange(b * 64, dtype=torch.float32).reshape(b, 64) * (rank + 1)
        c0 = shard.collectives
        prev, nxt, tot, first, last = fc.edges_and_sums(mine, sums, b, hw)
        assert shard.collectives - c0 == 1, "one collective per temporal convolution"
        assert first == (rank == 0) and last == (rank == world - 1)
        want_prev = full[:, fc.f0 - 1] if rank > 0 else torch.zeros_like(full[:, 0])
        want_next = full[:, fc.f0 + fc.t_loc] if rank < world - 1 else torch.zeros_like(full[:, 0])
        assert torch.equal(prev, want_prev) and torch.equal(nxt, want_next)
        assert torch.equal(tot, torch.arange(b * 64, dtype=torch.float32).reshape(b, 64) * sum(range(1, world + 1)))
        ext = fc.with_halo(mine, b, hw).reshape(b, fc.t_loc + 2, hw, C)
        assert torch.equal(ext[:, 0], want_prev) and torch.equal(ext[:, -1], want_next) and torch.equal(ext[:, 1:-1], full[:, fc.f0:fc.f0 + fc.t_loc])
        assert torch.equal(fc.inner(ext.reshape(-1, C), b, hw), mine)
        # gathers: every rank ends with all frames in clip order; several tensors in one collective
        allf = fc.gather_frames(mine, b, hw)
        assert torch.equal(allf, full.reshape(b * T * hw, C))
        k2 = (mine.float() * 2).to(torch.bfloat16)
        a, bb = fc.gather_frames_multi([mine, k2], b, hw)
        assert torch.equal(a, full.reshape(b * T * hw, C)) and torch.equal(bb, (full.float() * 2).to(torch.bfloat16).reshape(b * T * hw, C))
        assert torch.equal(fc.local_frames(allf, b, hw), mine)
        red = shard.all_reduce_sum(torch.full((3,), float(rank + 1)))
        assert torch.equal(red, torch.full((3,), float(sum(range(1, world + 1)))))
    open(os.path.join(out_dir, f"ok{rank}"), "w").write("ok")
    dist.destroy_process_group()


@pytest.mark.parametrize("world,T", [(8, 16), (2, 16), (4, 8)])
def test_frame_shard_exchanges_over_gloo(world, T, tmp_path):
    mp.spawn(_worker, args=(world, _free_port(), T, str(tmp_path)), nprocs=world, join=True)
    assert all(os.path.exists(os.path.join(str(tmp_path), f"ok{r}")) for r in range(world))
