#!/usr/bin/env python
"""bf16 vs e4m3 sparse epipolar attention on the benchmark masks (b = 2): us per call (incl. the fp8 quantisation pass) and
the difference between the two.   python tools/fp8_probe.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from camc2v_amd import camera, ops  # noqa: E402
from tools.bench_kernels import timeit  # noqa: E402

dev = torch.device("cuda:0")
torch.set_grad_enabled(False)
T, px = 16, 256
K = torch.tensor([[px / 2, 0, px / 2], [0, px / 2, px / 2], [0, 0, 1.0]], device=dev).repeat(1, T, 1, 1)
w2c = camera.synthetic_trajectory(1, T, dev)
F = camera.pairwise_fundamental(K, camera.relative_c2w(w2c, torch.zeros(1, dtype=torch.long, device=dev)), generator=torch.Generator(device=dev).manual_seed(3))
packed = camera.epipolar_masks_packed(F, T, px, px)
g = torch.Generator(device=dev).manual_seed(5)
for d, hl, H in ((8, 32, 5), (16, 16, 10)):
    bits, flags, perm, wbits, order = packed[d]
    L, C = T * hl * hl, H * 64
    qkv = torch.randn(2 * L, 3 * C, device=dev, generator=g).to(torch.bfloat16)
    reg = torch.randn(4, C, device=dev, generator=g).to(torch.bfloat16)
    s = (L * 3 * C, 0, 3 * C)
    o16 = torch.empty(2 * L, C, device=dev, dtype=torch.bfloat16)
    o8 = torch.empty_like(o16)
    f16 = lambda: ops.attention(qkv, qkv[:, C:], qkv[:, 2 * C:], B=2, inner=1, H=H, Lq=L, Lk=L, q_str=s, k_str=s, v_str=s, mask_bits=bits,
                                tile_flags=flags, mask_nb=1, wave_bits=wbits, group_order=order, perm=perm, kreg=reg, vreg=reg, out=o16, o_str=(L * C, 0, C))
    f8 = lambda: ops.attention_sparse_fp8(qkv, qkv[:, C:], qkv[:, 2 * C:], B=2, H=H, L=L, q_str=s, k_str=s, v_str=s, mask_bits=bits, wave_bits=wbits,
                                          mask_nb=1, group_order=order, kreg=reg, vreg=reg, perm=perm, out=o8)
    t16, t8 = timeit(f16), timeit(f8)
    l2 = ((o8.float() - o16.float()).norm() / o16.float().norm()).item()
    print(f"L={L} H={H} b=2: bf16 {t16:7.1f} us | e4m3 (quantise + attend) {t8:7.1f} us | rel-L2 between them {l2:.3e}", flush=True)
