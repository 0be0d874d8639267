#!/usr/bin/env python
"""Two independent clips in flight on one GPU (two host threads, two HIP streams, two model instances = two sets of
hipGraphs), against one clip at a time: does the second clip fill the CUs that the 8x8 / 4x4-latent layers leave idle?
    python tools/lanes_probe.py [clips_per_lane]"""
import os
import sys
import threading
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402

dev = torch.device("cuda:0")
torch.set_grad_enabled(False)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 3
models = [bench.build_model(dev) for _ in range(2)]
inputs = [[bench.synthetic_inputs(models[l], dev, clip=10 * l + i) for i in range(n + 1)] for l in range(2)]


def lane(l, clips, stream, out):
    with torch.cuda.stream(stream):
        for i in clips:
            out.append(bench.sample_clip(models[l], *inputs[l][i], True))
        stream.synchronize()


streams = [torch.cuda.Stream(), torch.cuda.Stream()]
for l in range(2):   # warm-up: pack, capture
    lane(l, [0], streams[l], [])
torch.cuda.synchronize()
t0 = time.perf_counter()
lane(0, range(1, n + 1), streams[0], [])
torch.cuda.synchronize()
t_one = (time.perf_counter() - t0) / n
print(f"one lane : {1e3 * t_one:7.1f} ms per clip  {16 / t_one:6.2f} frames/s", flush=True)
outs = [[], []]
th = [threading.Thread(target=lane, args=(l, range(1, n + 1), streams[l], outs[l])) for l in range(2)]
t0 = time.perf_counter()
for t in th:
    t.start()
for t in th:
    t.join()
torch.cuda.synchronize()
t_two = (time.perf_counter() - t0) / (2 * n)
print(f"two lanes: {1e3 * t_two:7.1f} ms per clip  {16 / t_two:6.2f} frames/s  ({2 * n} clips)", flush=True)
ok = all(torch.isfinite(o).all().item() for lst in outs for o in lst)
print("finite:", ok)
