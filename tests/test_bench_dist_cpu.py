"""The multi-process side of bench.py on CPU (gloo, world size 2): W untimed + K timed clips between barriers,
MAX over ranks, whole-job frames/s.  The clip itself is a stand-in (the HIP path needs a GPU); what is checked
is the launcher contract the driver relies on (clips shard across ranks, no data-path
collective)."""
import json
import os
import socket
import sys
import time

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import bench
    calls = {"n": 0}

    def fake_clip():                       # rank 1 is the slow rank: the job time must be ITS time
        calls["n"] += 1
        time.sleep(0.05 * (1 + rank))
        return torch.zeros(1)

    elapsed, mine, _ = bench.timed_clips(fake_clip, steps=3, warmup=2, dist=dist)
    line = bench.result_line(elapsed, 3, 2, world, use_graph=True) if rank == 0 else None
    with open(os.path.join(out_dir, f"rank{rank}.json"), "w") as f:
        json.dump({"calls": calls["n"], "elapsed": elapsed, "mine": mine, "line": line}, f)
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_timing_contract(tmp_path):
    world, port = 2, _free_port()
    mp.spawn(_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    r = [json.load(open(tmp_path / f"rank{i}.json")) for i in range(world)]
    assert r[0]["calls"] == r[1]["calls"] == 5                      # 2 warm-up + exactly 3 timed
    assert r[0]["elapsed"] == pytest.approx(r[1]["elapsed"])        # MAX over ranks is shared
    assert r[0]["elapsed"] >= r[1]["mine"] - 1e-6 and r[1]["mine"] > r[0]["mine"] * 0.9
    assert r[0]["elapsed"] >= 3 * 0.1 - 0.02
    line = r[0]["line"]
    assert line["n_gpus"] == 2 and line["steps"] == 3 and line["warmup"] == 2 and line["scaling"] == "weak"
    assert line["value"] == pytest.approx(16.0 * 3 * 2 / r[0]["elapsed"])    # whole-job frames/s
    assert line["ms_per_step"] == pytest.approx(1e3 * r[0]["elapsed"] / 3)
    assert line["roofline"]["bound"] == "mfma" and line["vs_baseline"] is None and line["dtype"] == "bf16"
    assert set(line) >= {"metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better",
                         "scaling", "vs_baseline", "dtype", "data", "config", "roofline"}


def test_bench_refuses_to_run_without_gpu():
    import subprocess
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "1", "--warmup", "0"],
                       capture_output=True, text=True, timeout=120)
    assert p.returncode != 0 and "no CPU fallback" in (p.stderr + p.stdout)
