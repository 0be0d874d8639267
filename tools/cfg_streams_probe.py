#!/usr/bin/env python
"""One CFG step of the benchmark clip, three ways, as hipGraphs on one GPU:
  pair      the batched cond + uncond forward on one stream (what the sampler runs),
  streams   the conditional forward on one stream and the unconditional forward on another, joined by events
            (single-clip latency: each forward is half the batch, the two fill each other's idle CUs),
  serial    the same two half forwards one after the other on one stream.
python tools/cfg_streams_probe.py [reps]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402

dev = torch.device("cuda:0")
torch.set_grad_enabled(False)
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 10
model = bench.build_model(dev)
cond, uncond, fs, x_T, _ = bench.synthetic_inputs(model, dev)
t = torch.full((1,), 439, dtype=torch.long, device=dev)
uc = dict(uncond, camera_condition=dict(cond["camera_condition"], is_uc=True))
kw = dict(fs=fs, enable_camera_condition=True)

pair = lambda: model.apply_model_pair(x_T, t, cond, uc, **kw)
half_c = lambda: model.apply_model(x_T, t, cond, **kw)
half_u = lambda: model.apply_model(x_T, t, uc, **kw)


def capture(fn, stream):
    with torch.cuda.stream(stream):
        fn()
        fn()
    stream.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.stream(stream):
        with torch.cuda.graph(g, stream=stream):
            out = fn()
    stream.synchronize()
    return g, out


A, B = torch.cuda.Stream(), torch.cuda.Stream()
g_pair, o_pair = capture(pair, A)
g_c, o_c = capture(half_c, A)
g_u, o_u = capture(half_u, B)


def run_pair():
    with torch.cuda.stream(A):
        g_pair.replay()


def run_serial():
    with torch.cuda.stream(A):
        g_c.replay()
    with torch.cuda.stream(A):
        g_u.replay()


ev_a, ev_b = torch.cuda.Event(), torch.cuda.Event()


def run_streams():
    ev_a.record(A)
    B.wait_event(ev_a)
    with torch.cuda.stream(B):
        g_u.replay()
        ev_b.record(B)
    with torch.cuda.stream(A):
        g_c.replay()
    A.wait_event(ev_b)


def timeit(fn):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(A)
    for _ in range(reps):
        fn()
    e1.record(A)
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


for _ in range(2):
    for name, fn in (("pair", run_pair), ("serial", run_serial), ("streams", run_streams)):
        ms = timeit(fn)
        print(f"{name:8s} {ms:7.3f} ms per CFG step  -> {25 * ms:7.1f} ms per 25-step clip, {16 / (25e-3 * ms):6.2f} frames/s one clip at a time", flush=True)
run_pair()
run_streams()
torch.cuda.synchronize()
e_c, e_uc = o_pair
d = lambda a, b: ((a.float() - b.float()).norm() / b.float().norm()).item()
print(f"half forwards against the pair: rel-L2 cond {d(o_c, e_c):.3e}, uncond {d(o_u, e_uc):.3e}")
