#!/usr/bin/env python
"""Per-kernel microbenchmarks on the shapes of the 256x256 model at the CFG-pair batch (b=2, t=16).
Prints one line per shape: average device time (HIP events on the launch stream) and TFLOP/s or GB/s.
    python tools/bench_kernels.py [gemm] [conv] [tconv] [attn] [epi] [norm]
"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from camc2v_amd import camera, ops, pack  # noqa: E402

dev = torch.device("cuda:0")
torch.set_grad_enabled(False)


def timeit(fn, iters=20, warm=3):
    """Average device time of fn() in microseconds.  The calls are captured into a hipGraph and replayed, so that the
    ~13 us of Python/ctypes launch cost per call does not hide kernels shorter than that (CCV_BENCH_EAGER=1: eager)."""
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    if os.environ.get("CCV_BENCH_EAGER") == "1":
        e0.record()
        for _ in range(iters):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / iters * 1e3
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        fn()
    torch.cuda.current_stream().wait_stream(side)
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        for _ in range(iters):
            fn()
    graph.replay()
    torch.cuda.synchronize()
    e0.record()
    graph.replay()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3  # us


def rnd(*s, dtype=torch.bfloat16, scale=1.0):
    return (torch.randn(*s, device=dev) * scale).to(dtype)


def bench_gemm():
    print("== linear GEMM (M, N, K)  [bf16 A, bf16 out]")
    shapes = []
    for M, C in ((32768, 320), (8192, 640), (2048, 1280), (512, 1280)):
        shapes += [(M, 3 * C, C, ""), (M, C, C, "res"), (M, 8 * C, C, "geglu"), (M, C, 4 * C, "res")]
    shapes += [(32768, 1536, 512, ""), (32768, 4096, 512, "geglu"), (32768, 512, 2048, "res"), (2, 21120, 1280, "f32")]
    for M, N, K, kind in shapes:
        a, w = rnd(M, K), rnd(N, K, scale=0.05)
        bias = torch.zeros(N, device=dev)
        if kind == "geglu":
            fn = lambda: ops.gemm(a, w, bias=bias, geglu=True)
        elif kind == "res":
            stream = torch.zeros(M, N, device=dev)
            fn = lambda: ops.gemm(a, w, bias=bias, residual=stream, out_f32=True, out=stream)
        elif kind == "f32":
            fn = lambda: ops.gemm(a, w, bias=bias, out_f32=True)
        else:
            fn = lambda: ops.gemm(a, w)
        us = timeit(fn)
        print(f"  {M:6d} {N:6d} {K:5d} {kind:6s} {us:8.1f} us  {2 * M * N * K / us / 1e6:7.1f} TF/s")


def bench_conv():
    print("== conv3x3 (frames=32, h, Cin -> Cout)  [bf16 A, fp32 out + residual]")
    for h, cin, cout in ((32, 320, 320), (32, 640, 320), (32, 960, 320), (16, 320, 640), (16, 640, 640), (16, 1280, 640),
                         (16, 960, 640), (8, 640, 1280), (8, 1280, 1280), (8, 2560, 1280), (8, 1920, 1280),
                         (4, 1280, 1280), (4, 2560, 1280)):
        M = 32 * h * h
        a = rnd(M, cin)
        w = pack.pack_conv3x3(torch.randn(cout, cin, 3, 3, device=dev) * 0.02)
        bias = torch.zeros(cout, device=dev)
        res = torch.zeros(M, cout, device=dev)
        fn = lambda: ops.gemm(a, w, k=cin, taps=9, bias=bias, residual=res, out_f32=True, gather=ops.GATHER_CONV3X3,
                              conv=(h, h, h, h, 1, 0))
        us = timeit(fn)
        print(f"  h={h:2d} {cin:5d}->{cout:5d}  {us:8.1f} us  {2 * M * cout * 9 * cin / us / 1e6:7.1f} TF/s")


def bench_tconv():
    print("== temporal conv (3,1,1) (clips=2, t=16, h, C)")
    for h, c in ((32, 320), (16, 640), (8, 1280), (4, 1280)):
        M = 2 * 16 * h * h
        a = rnd(M, c)
        w = pack.pack_tconv3(torch.randn(c, c, 3, 1, 1, device=dev) * 0.02)
        bias = torch.zeros(c, device=dev)
        fn = lambda: ops.gemm(a, w, k=c, taps=3, bias=bias, gather=ops.GATHER_TCONV3, tconv=(16, h * h))
        us = timeit(fn)
        print(f"  h={h:2d} C={c:5d}  {us:8.1f} us  {2 * M * c * 3 * c / us / 1e6:7.1f} TF/s")


def bench_attn():
    print("== attention (dense flops 4*B*H*Lq*Lk*64)")
    for h, H in ((32, 5), (16, 10), (8, 20), (4, 20)):
        C, hw, B = H * 64, h * h, 32
        qkv = rnd(B * hw, 3 * C)
        ld = 3 * C
        s = (hw * ld, 0, ld)
        us = timeit(lambda: ops.attention(qkv, qkv[:, C:], qkv[:, 2 * C:], B=B, inner=1, H=H, Lq=hw, Lk=hw, q_str=s, k_str=s, v_str=s))
        print(f"  spatial self  h={h:2d} H={H:2d}: {us:8.1f} us  {4 * B * H * hw * hw * 64 / us / 1e6:7.1f} TF/s")
        q = rnd(B * hw, C)
        kv_t, kv_i = rnd(2 * 77, 2 * C), rnd(2 * 768, 2 * C)
        st, si = (77 * 2 * C, 0, 2 * C), (768 * 2 * C, 0, 2 * C)
        us = timeit(lambda: ops.attention(q, kv_t, kv_t[:, C:], B=B, inner=16, H=H, Lq=hw, Lk=77, q_str=(16 * hw * C, hw * C, C),
                                          k_str=st, v_str=st, k2=kv_i, v2=kv_i[:, C:], k2_str=si, v2_str=si, Lk2=768, gate2=1.0))
        print(f"  cross 77+768  h={h:2d} H={H:2d}: {us:8.1f} us  {4 * B * H * hw * 845 * 64 / us / 1e6:7.1f} TF/s")
        T = 16
        st_ = (T * hw * ld, ld, hw * ld)
        o = torch.empty(B * hw, C, dtype=torch.bfloat16, device=dev)
        us = timeit(lambda: ops.attention(qkv, qkv[:, C:], qkv[:, 2 * C:], B=2 * hw, inner=hw, H=H, Lq=T, Lk=T, q_str=st_, k_str=st_,
                                          v_str=st_, out=o, o_str=(T * hw * C, C, hw * C)))
        gb = (qkv.numel() + o.numel()) * 2 / 1e9
        print(f"  temporal      h={h:2d} H={H:2d}: {us:8.1f} us  {gb / us * 1e6:7.1f} GB/s")


def bench_epi():
    print("== epipolar attention (b=2 sharing one mask; synthetic trajectory of the benchmark)")
    b, T, px = 1, 16, 256
    K = torch.tensor([[128.0, 0, 128], [0, 128, 128], [0, 0, 1]], device=dev).repeat(b, T, 1, 1)
    w2c = camera.synthetic_trajectory(b, T, dev)
    cam = camera.camera_condition(K, w2c, torch.zeros(b, dtype=torch.long, device=dev), px, px)
    Fm = cam["fundamental"]
    for d, H in ((8, 5), (16, 10), (32, 20), (64, 20)):
      for patch in (False, True):
        hh = px // d
        if patch and not ops.patch_order_ok(hh, hh):
            continue
        mp = ops.epipolar_mask_bits(Fm, T, hh, hh, d, patch_order=patch)
        bits, flags = mp
        perm = (hh * hh, hh) if patch else None
        L = bits.shape[1]
        C = H * 64
        pop = sum(bin(x & 0xffffffff).count("1") for x in bits[0, ::max(1, L // 64)].flatten().tolist())
        dens = pop / (len(range(0, L, max(1, L // 64))) * L)
        qkv = rnd(2 * L, 3 * C)
        kreg, vreg = rnd(4, C), rnd(4, C)
        ld = 3 * C
        s = (L * ld, 0, ld)
        us = timeit(lambda: ops.attention(qkv, qkv[:, C:], qkv[:, 2 * C:], B=2, inner=1, H=H, Lq=L, Lk=L, q_str=s, k_str=s, v_str=s,
                                          mask_bits=bits, mask_nb=1, tile_flags=flags, wave_bits=mp.wave_bits, kreg=kreg, vreg=vreg, perm=perm), iters=10)
        us_l = timeit(lambda: ops.attention(qkv, qkv[:, C:], qkv[:, 2 * C:], B=2, inner=1, H=H, Lq=L, Lk=L, q_str=s, k_str=s, v_str=s,
                                            mask_bits=bits, mask_nb=1, tile_flags=flags, wave_bits=mp.wave_bits, group_order=mp.group_order,
                                            kreg=kreg, vreg=vreg, perm=perm), iters=10)
        us_t = timeit(lambda: ops.attention(qkv, qkv[:, C:], qkv[:, 2 * C:], B=2, inner=1, H=H, Lq=L, Lk=L, q_str=s, k_str=s, v_str=s,
                                            mask_bits=bits, mask_nb=1, tile_flags=flags, kreg=kreg, vreg=vreg, perm=perm), iters=10)
        us_d = timeit(lambda: ops.attention(qkv, qkv[:, C:], qkv[:, 2 * C:], B=2, inner=1, H=H, Lq=L, Lk=L, q_str=s, k_str=s, v_str=s),
                      iters=5)
        print(f"  L={L:6d} H={H:2d} {'patch ' if patch else 'raster'}: sparse {us:8.1f} us, longest-first {us_l:8.1f} us, tiled {us_t:8.1f} us (dense-equivalent {4 * 2 * H * L * L * 64 / us_l / 1e6:7.1f} TF/s), "
              f"unmasked {us_d:8.1f} us ({4 * 2 * H * L * L * 64 / us_d / 1e6:7.1f} TF/s); "
              f"element density {dens:.3f}, 128x64 tile density {flags.float().mean().item():.3f}")


def bench_norm():
    print("== norms")
    for rows, C, inst in ((32768, 320, 32), (32768, 320, 2), (32768, 960, 32), (8192, 640, 32), (8192, 640, 2),
                          (2048, 1280, 32), (2048, 1280, 2), (512, 1280, 32), (512, 2560, 32)):
        for f32 in (True, False):
            x = rnd(rows, C, dtype=torch.float32 if f32 else torch.bfloat16)
            g, b_ = torch.ones(C, device=dev), torch.zeros(C, device=dev)
            us = timeit(lambda: ops.groupnorm(x, g, b_, instances=inst, eps=1e-5, silu=True))
            gb = x.numel() * (x.element_size() * 2 + 2) / 1e9
            print(f"  groupnorm rows={rows:6d} C={C:5d} inst={inst:3d} {'f32' if f32 else 'bf16'}: {us:7.1f} us  {gb / us * 1e6:7.1f} GB/s")
    for rows, C in ((32768, 320), (8192, 640), (2048, 1280), (32768, 512)):
        x = rnd(rows, C, dtype=torch.float32)
        g, b_ = torch.ones(C, device=dev), torch.zeros(C, device=dev)
        us = timeit(lambda: ops.layernorm(x, g, b_))
        print(f"  layernorm rows={rows:6d} C={C:5d}: {us:7.1f} us  {x.numel() * 6 / 1e9 / us * 1e6:7.1f} GB/s")


if __name__ == "__main__":
    which = sys.argv[1:] or ["gemm", "conv", "tconv", "attn", "epi", "norm"]
    for w in which:
        globals()["bench_" + w]()
