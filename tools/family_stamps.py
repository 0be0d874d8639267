#!/usr/bin/env python
"""Phase breakdown of the family GEMM kernel's two-stage loop (gemm_dma_kernel, ST = 2) from s_memtime stamps.

Needs the DIAGNOSTIC build of the library (never the shipped one): ccv_gemm.hip compiled with -DCCV_FAMILY_STAMPS and linked with the
other objects, handed over as CCV_HIP_LIB:
    hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -DCCV_FAMILY_STAMPS -c camc2v_amd/csrc/ccv_gemm.hip -o /tmp/gemm_stamps.o
    hipcc --offload-arch=gfx950 -shared -fPIC -o camc2v_amd/libccv_hip_stamps.so /tmp/gemm_stamps.o camc2v_amd/build/ccv_{attn,fused,norm,misc,pose}.o
    CCV_HIP_LIB=$PWD/camc2v_amd/libccv_hip_stamps.so CCV_GEMM_TUNE=1 python tools/family_stamps.py
Lane 0 of every wave stamps, per 64-deep slab: (0) top, (1) next slab's DMA issued, (2) fragment reads done + all MFMAs issued,
(3) s_waitcnt vmcnt(0) passed, (4) barrier passed.  Printed: median cycles per phase over all waves and slabs, per shape.
"""
import ctypes as C
import os
import sys

import numpy as np

os.environ["CCV_GEMM_TUNE"] = "1"
import torch  # noqa: E402

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from camc2v_amd import ops, pack  # noqa: E402
from camc2v_amd.lib import lib  # noqa: E402

dev = torch.device("cuda:0")
torch.set_grad_enabled(False)
ops.TRACK_GEMM_PLAN = True
SLABS, POINTS = 48, 5
WORDS = SLABS * POINTS + 4

SHAPES = [  # kind, M, N, K, forced family tile (10 MT + NT), split
    ("conv", 32768, 320, 320, 45, 1), ("conv", 8192, 640, 640, 45, 2), ("tconv", 32768, 320, 320, 45, 1),
    ("lin", 32768, 320, 1280, 45, 1), ("lin", 8192, 1920, 640, 45, 1), ("lin", 8192, 640, 640, 42, 1), ("lin", 2048, 1280, 1280, 42, 1),
    ("lin", 8192, 1920, 640, 44, 1), ("linres", 8192, 640, 640, 42, 1), ("linres", 2048, 1280, 1280, 42, 1), ("linres", 32768, 320, 1280, 45, 1),
]      # linres: the fp16 residual stream read-modify-written in the epilogue


def main():
    setter = lib().ccv_debug_family_stamps
    setter.argtypes, setter.restype = [C.c_void_p], C.c_int
    for kind, M, N, K, ft, sp in SHAPES:
        taps = {"conv": 9, "tconv": 3}.get(kind, 1)
        kw = {}
        if kind == "conv":
            side = int(round((M // 32) ** 0.5))
            W = pack.pack_conv3x3(torch.randn(N, K, 3, 3, device=dev) * 0.02)
            kw = dict(k=K, taps=9, gather=ops.GATHER_CONV3X3, conv=(side, side, side, side, 1, 0))
        elif kind == "tconv":
            W = pack.pack_tconv3(torch.randn(N, K, 3, 1, 1, device=dev) * 0.02)
            kw = dict(k=K, taps=3, gather=ops.GATHER_TCONV3, tconv=(16, M // 32))
        else:
            W = (torch.randn(N, K, device=dev) * 0.03).to(torch.bfloat16)
        a = torch.randn(M, K, device=dev).to(torch.bfloat16)
        if kind == "linres":
            stream = torch.randn(M, N, device=dev).to(torch.float16)
            kw = dict(residual=stream, out_dtype=torch.float16, out=torch.empty_like(stream), bias=torch.zeros(N, device=dev))
        os.environ.update(CCV_GEMM_RING="-1", CCV_GEMM_FAMTILE=str(ft), CCV_GEMM_ST="2", CCV_GEMM_SPLIT=str(sp))
        bm, bn = 32 * (ft // 10), 32 * (ft % 10)
        tiles = ((M + bm - 1) // bm) * ((N + bn - 1) // bn) * sp
        buf = torch.zeros(tiles * 4 * WORDS, dtype=torch.int32, device=dev)
        assert setter(C.c_void_p(0)) == 0
        for _ in range(3):
            ops.gemm(a, W, **kw)            # warm, unstamped
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        ops.gemm(a, W, **kw)
        e1.record()
        torch.cuda.synchronize()
        t_plain = e0.elapsed_time(e1) * 1e3
        assert setter(C.c_void_p(buf.data_ptr())) == 0
        e0.record()
        ops.gemm(a, W, **kw)
        e1.record()
        torch.cuda.synchronize()
        t_stamped = e0.elapsed_time(e1) * 1e3
        assert setter(C.c_void_p(0)) == 0
        plan = ops.LAST_GEMM_PLAN
        t = buf.cpu().numpy().view(np.uint32).reshape(tiles, 4, WORDS).astype(np.int64)
        nslab = int(t[0, 0, SLABS * POINTS + 2])
        n = min(nslab, SLABS)
        st = t[:, :, :SLABS * POINTS].reshape(tiles, 4, SLABS, POINTS)[:, :, :n]
        d = lambda x: (x + (1 << 32)) % (1 << 32)            # 32-bit wrap
        issue = d(st[..., 1] - st[..., 0])
        mult = d(st[..., 2] - st[..., 1])
        wait = d(st[..., 3] - st[..., 2])
        barrier = d(st[..., 4] - st[..., 3])
        slab = d(st[:, :, 1:, 0] - st[:, :, :-1, 0]) if n > 1 else issue
        first = d(st[:, :, 0, 0] - t[:, :, SLABS * POINTS])           # kernel start -> first slab ready (prologue: gather setup + first DMA round trip)
        total = d(t[:, :, SLABS * POINTS + 3] - t[:, :, SLABS * POINTS])
        epi = d(t[:, :, SLABS * POINTS + 3] - t[:, :, SLABS * POINTS + 1])
        mfma_cyc = (ft // 10) * (ft % 10) * 2 * 16          # this wave's MFMAs per slab x 16 cycles
        med = lambda x: float(np.median(x))
        print(f"{kind:5s} M={M:6d} N={N:5d} K={K:5d} tile {bm}x{bn} split {sp} plan {plan}: {tiles} workgroups, {nslab} slabs each; {t_plain:.1f} us plain, {t_stamped:.1f} us stamped\n"
              f"    per slab (median cycles over waves x slabs): DMA issue {med(issue):.0f} | fragment reads + MFMA issue {med(mult):.0f} "
              f"(the wave's own MFMAs: {mfma_cyc}) | vmcnt(0) {med(wait):.0f} | barrier {med(barrier):.0f} | slab to slab {med(slab):.0f}\n"
              f"    per tile: prologue (gather setup + first DMA round trip) {med(first):.0f} | main loop {med(slab) * n:.0f} | epilogue (to the issue of its last store) {med(epi):.0f} | whole wave {med(total):.0f}", flush=True)


if __name__ == "__main__":
    main()
