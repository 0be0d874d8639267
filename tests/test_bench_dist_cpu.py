"""The multi-process side of bench.py on CPU (gloo, world size 2): `bench.main()` itself is driven with a stand-in
clip (the HIP path needs a GPU) -- W untimed + K timed clips between barriers, per-clip conditioning sets, the final
all_gather of the latents, MAX over ranks, whole-job frames/s, one JSON line from rank 0 -- plus the launcher logic
of `--gpus N` without a launcher (child process, never an exec).  What is checked is the contract the driver relies
on (clips shard across ranks, no data-path collective inside the sampling loop)."""
import contextlib
import io
import json
import os
import socket
import subprocess
import sys
import time

import pytest
import torch
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, out_dir, lanes=1, steps=3, extra=()):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world))
    import bench
    calls = {"clips": [], "sets": 0}

    def make_inputs(model, device, rank=0, clip=0):
        calls["sets"] += 1
        return (rank, clip)

    def fake_clip(model, r, clip, use_graph):       # rank 1 is the slow rank: the job time must be ITS time
        calls["clips"].append(clip)
        time.sleep(0.05 * (1 + r))
        return torch.full((1, 4, 2, 2, 2), float(1 + r + 10 * clip))

    buf = io.StringIO()
    with contextlib.redirect_stdout(buf):
        rc = bench.main(["--gpus", str(world), "--steps", str(steps), "--warmup", "2", "--lanes", str(lanes), *extra],
                        hooks=dict(device=torch.device("cpu"), backend="gloo", build_model=lambda dev: None,
                                   synthetic_inputs=make_inputs, sample_clip=fake_clip))
    with open(os.path.join(out_dir, f"rank{rank}.json"), "w") as f:
        json.dump({"rc": rc, "clips": calls["clips"], "sets": calls["sets"], "stdout": buf.getvalue()}, f)


def test_two_rank_bench_main(tmp_path):
    world, port = 2, _free_port()
    mp.spawn(_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    r = [json.load(open(tmp_path / f"rank{i}.json")) for i in range(world)]
    assert r[0]["rc"] == 0 and r[1]["rc"] == 0
    assert r[0]["clips"] == r[1]["clips"] == [0, 1, 2, 3, 4]        # 2 warm-up + exactly 3 timed, each its own input set
    assert r[0]["sets"] == r[1]["sets"] == 5
    assert r[1]["stdout"].strip() == ""                             # only rank 0 prints
    lines = [ln for ln in r[0]["stdout"].splitlines() if ln.strip()]
    assert len(lines) == 1
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["steps"] == 3 and line["warmup"] == 2 and line["scaling"] == "weak"
    # whole-job frames/s over the SLOWEST rank's time: rank 1 sleeps 0.1 s per clip
    assert line["ms_per_step"] >= 100.0 - 5.0
    assert line["value"] == pytest.approx(16.0 * 3 * 2 / (line["ms_per_step"] * 3 / 1e3))
    assert line["config"]["ranks_in_final_all_gather"] == 2         # both ranks' latents arrived, and they differ
    assert line["roofline"]["bound"] == "mfma" and line["vs_baseline"] is None and line["dtype"] == "bf16"
    assert set(line) >= {"metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better",
                         "scaling", "vs_baseline", "dtype", "data", "config", "roofline"}


def test_eight_rank_bench_main(tmp_path):
    """The driver's N = 8 launch shape on CPU: eight gloo ranks, one clip stream each (no collective inside the loop), the final
    all_gather sees eight distinct latents, the job time is the slowest rank's, and the line lists every rank's own time."""
    world, port = 8, _free_port()
    mp.spawn(_worker, args=(world, port, str(tmp_path), 1, 2), nprocs=world, join=True)
    r = [json.load(open(tmp_path / f"rank{i}.json")) for i in range(world)]
    assert all(x["rc"] == 0 for x in r) and all(x["clips"] == [0, 1, 2, 3] for x in r)
    assert all(x["stdout"].strip() == "" for x in r[1:])
    line = json.loads([ln for ln in r[0]["stdout"].splitlines() if ln.strip()][0])
    assert line["n_gpus"] == 8 and line["config"]["ranks_in_final_all_gather"] == 8 and line["config"]["parallelism"] == "clip-dp8"
    per = line["config"]["per_rank_ms_per_step"]
    assert len(per) == 8 and max(per) <= line["ms_per_step"] + 1e-6        # (the job time adds the closing all_gather + barriers of 8 gloo processes)
    assert per[7] >= per[0] + 200.0                    # rank r sleeps 0.05 (1 + r) s per clip: the slow rank sets the job time
    assert line["value"] == pytest.approx(16.0 * 2 * 8 / (line["ms_per_step"] * 2 / 1e3))


def test_two_rank_bench_main_with_two_clips_in_flight(tmp_path):
    """--lanes 2: every lane runs the W warm-up clips itself (its graphs are captured there), then the lanes share exactly K
    timed clips (each sampled once, by whichever lane is free) and overlap them in time."""
    world, port = 2, _free_port()
    mp.spawn(_worker, args=(world, port, str(tmp_path), 2, 4), nprocs=world, join=True)
    r = [json.load(open(tmp_path / f"rank{i}.json")) for i in range(world)]
    for i in range(world):
        assert r[i]["rc"] == 0
        assert r[i]["clips"][:4] == [0, 1, 0, 1]                      # warm-up: lane 0, then lane 1
        assert sorted(r[i]["clips"][4:]) == [2, 3, 4, 5]              # exactly K = 4 timed clips, each once
        assert r[i]["sets"] == 6
    line = json.loads([ln for ln in r[0]["stdout"].splitlines() if ln.strip()][0])
    assert line["steps"] == 4 and line["config"]["clips_in_flight_per_gpu"] == 2
    # rank 1 sleeps 0.1 s per clip: 4 clips on 2 lanes take ~0.2 s, not 0.4 s
    assert 50.0 - 5.0 <= line["ms_per_step"] <= 85.0
    assert line["value"] == pytest.approx(16.0 * 4 * 2 / (line["ms_per_step"] * 4 / 1e3))
    assert line["config"]["ranks_in_final_all_gather"] == 2


def test_two_rank_bench_main_frame_sharded(tmp_path):
    """--frame-shard: the ranks work on the SAME clips (rank 0's input sets), so the job's frames are those of K clips, not K x N
    (strong scaling)."""
    world, port = 2, _free_port()
    mp.spawn(_worker, args=(world, port, str(tmp_path), 1, 3, ("--frame-shard",)), nprocs=world, join=True)
    r = [json.load(open(tmp_path / f"rank{i}.json")) for i in range(world)]
    assert r[0]["rc"] == 0 and r[1]["rc"] == 0 and r[0]["clips"] == r[1]["clips"] == [0, 1, 2, 3, 4]
    line = json.loads([ln for ln in r[0]["stdout"].splitlines() if ln.strip()][0])
    assert line["scaling"] == "strong" and line["config"]["parallelism"] == "frame-shard2" and line["config"]["launch"] == "eager"
    assert line["value"] == pytest.approx(16.0 * 3 / (line["ms_per_step"] * 3 / 1e3))


def test_gpus_flag_must_match_the_launcher(monkeypatch):
    sys.path.insert(0, ROOT)
    import bench
    monkeypatch.setenv("WORLD_SIZE", "4")
    with pytest.raises(SystemExit) as e:
        bench.main(["--gpus", "2"], hooks=dict(device=torch.device("cpu")))
    assert "WORLD_SIZE=4" in str(e.value)


def test_gpus_flag_without_launcher_spawns_a_child_job(monkeypatch):
    """`python bench.py --gpus 2` with no WORLD_SIZE: the N-rank torch.distributed.run job is started as a child process
    (subprocess.run, exit code relayed) before anything touches the GPU."""
    sys.path.insert(0, ROOT)
    import bench
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    seen = {}

    def fake_run(cmd, env=None, **kw):
        seen["cmd"], seen["env"] = cmd, env

        class R:
            returncode = 7
        return R()

    monkeypatch.setattr(bench.subprocess, "run", fake_run)
    monkeypatch.setattr(torch.cuda, "is_available", lambda: (_ for _ in ()).throw(AssertionError("GPU touched before the spawn")))
    assert bench.main(["--gpus", "2", "--steps", "4", "--warmup", "1"]) == 7
    cmd = seen["cmd"]
    assert cmd[:3] == [sys.executable, "-m", "torch.distributed.run"]
    assert "--nproc-per-node=2" in cmd and "--nnodes=1" in cmd and cmd[cmd.index("--master-addr") + 1] == "127.0.0.1"
    assert cmd[-6:] == ["--gpus", "2", "--steps", "4", "--warmup", "1"] and cmd[-7].endswith("bench.py")
    assert seen["env"]["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"


def test_bench_refuses_to_run_without_gpu():
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "1", "--warmup", "0"],
                       capture_output=True, text=True, timeout=120)
    assert p.returncode != 0 and "no CPU fallback" in (p.stderr + p.stdout)


def test_timed_clips_lanes_share_exactly_k_clips_and_reraise_errors():
    """Device-independent lane logic of bench.timed_clips: any K (odd too) is sampled exactly once each by the lanes, the warm-up runs
    per lane, and an exception inside a lane thread comes back on the caller's thread."""
    sys.path.insert(0, ROOT)
    import bench
    seen = []

    def clip(i):
        seen.append(i)
        time.sleep(0.01)
        return torch.tensor([float(i)])
    for lanes, steps, warmup in ((2, 5, 1), (3, 7, 2), (2, 1, 0)):
        seen.clear()
        _, _, out, extra = bench.timed_clips(clip, steps, warmup, after=lambda o: float(o), lanes=lanes)
        assert seen[:lanes * warmup] == list(range(warmup)) * lanes
        assert sorted(seen[lanes * warmup:]) == list(range(warmup, warmup + steps))
        assert float(out) == warmup + steps - 1 and extra == warmup + steps - 1     # `after` sees the LAST clip's output

    def bad(i):
        if i == 3:
            raise ValueError("clip 3 failed")
        return torch.zeros(1)
    with pytest.raises(ValueError, match="clip 3 failed"):
        bench.timed_clips(bad, 6, 0, lanes=2)
