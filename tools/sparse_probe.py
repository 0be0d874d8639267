#!/usr/bin/env python
"""The epipolar attention launches of one CFG step (b = 2) on the benchmark trajectory's masks, without the model:
5 x (L=16384, H=5) + 5 x (L=4096, H=10), timed with HIP events (hipGraph of the 10 launches).  Small enough to run
under `rocprofv3 --pmc FETCH_SIZE` / `--kernel-trace --stats` for the per-launch traffic and duration of
attn_sparse_kernel.   python tools/sparse_probe.py [reps]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from camc2v_amd import camera, ops  # noqa: E402

dev = torch.device("cuda:0")
torch.set_grad_enabled(False)
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
T, px = 16, 256
K = torch.tensor([[px / 2, 0, px / 2], [0, px / 2, px / 2], [0, 0, 1.0]], device=dev).repeat(1, T, 1, 1)
motion = os.environ.get("SPARSE_PROBE_CAMERA", "benchmark")     # benchmark (yaw + x / z translation) | vertical | forward | orbit
if motion == "benchmark":
    w2c = camera.synthetic_trajectory(1, T, dev)
else:
    fr = torch.arange(T, dtype=torch.float32)
    c2w = torch.eye(4).repeat(T, 1, 1)
    if motion == "vertical":
        c2w[:, 1, 3] = 0.05 * fr
    elif motion == "forward":
        c2w[:, 2, 3] = 0.08 * fr
    elif motion == "orbit":      # yaw + pitch + translation along all three axes
        import math
        for i in range(T):
            a, b = 0.03 * i, 0.02 * i
            ry = torch.tensor([[math.cos(a), 0, math.sin(a)], [0, 1, 0], [-math.sin(a), 0, math.cos(a)]])
            rx = torch.tensor([[1, 0, 0], [0, math.cos(b), -math.sin(b)], [0, math.sin(b), math.cos(b)]])
            c2w[i, :3, :3] = ry @ rx
        c2w[:, 0, 3], c2w[:, 1, 3], c2w[:, 2, 3] = 0.04 * fr, 0.03 * fr, 0.03 * fr
    else:
        raise SystemExit(f"SPARSE_PROBE_CAMERA={motion!r}")
    w2c = torch.linalg.inv(c2w).unsqueeze(0).to(dev)
print(f"# camera: {motion}", flush=True)
rel = camera.relative_c2w(w2c, torch.zeros(1, dtype=torch.long, device=dev))
F = camera.pairwise_fundamental(K, rel, generator=torch.Generator(device=dev).manual_seed(3))
packed = camera.epipolar_masks_packed(F, T, px, px)
g = torch.Generator(device=dev).manual_seed(5)
calls = []
for d, hl, H in ((8, 32, 5), (16, 16, 10)):
    bits, flags, perm, wbits, order = packed[d]
    L = T * hl * hl
    qkv = torch.randn(2 * L, 3 * H * 64, device=dev, generator=g).to(torch.bfloat16)
    kreg = torch.randn(4, H * 64, device=dev, generator=g).to(torch.bfloat16)
    s = (L * 3 * H * 64, 0, 3 * H * 64)
    out = torch.empty(2 * L, H * 64, device=dev, dtype=torch.bfloat16)
    kw = dict(B=2, inner=1, H=H, Lq=L, Lk=L, q_str=s, k_str=s, v_str=s, mask_bits=bits, tile_flags=flags, mask_nb=1, wave_bits=wbits,
              group_order=order, perm=perm, kreg=kreg, vreg=kreg, out=out, o_str=(L * H * 64, 0, H * 64))
    calls.append((qkv, H, kw, L))


def timeit(fn, n):
    fn()
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        fn()
    torch.cuda.current_stream().wait_stream(side)
    with torch.cuda.graph(graph):
        fn()
    graph.replay()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        graph.replay()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n


variants = [int(v) for v in os.environ.get("SPARSE_PROBE_VARIANTS", "6,4,5").split(",")]   # 6 per-wave, 4 / 5 workgroup-shared (8 / 4 waves)
for qkv, H, kw, L in calls:
    ref = None
    for variant in variants:
        fn = lambda: [ops.attention(qkv, qkv[:, H * 64:], qkv[:, 2 * H * 64:], variant=variant, **kw) for _ in range(5)]
        us = timeit(fn, reps) / 5
        out = kw["out"].clone()
        same = "" if ref is None else f"   bit-identical to variant {variants[0]}: {torch.equal(out, ref)}"
        ref = out if ref is None else ref
        alg = 2 * L * H * 64 * 2 * 4 + kw["mask_bits"].numel() * 4
        print(f"sparse L={L} H={H} b=2 variant {variant}: {us:8.1f} us/launch   dense-equivalent {4.0 * L * L * 64 * H * 2 / us / 1e6:7.1f} TF/s   "
              f"algorithmic bytes {alg / 1e6:.1f} MB (q,k,v,o + mask bits){same}", flush=True)
