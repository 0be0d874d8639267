"""Per-kernel parity tests of the HIP library (through the C ABI) on a real MI355X.

Checker: plain torch fp32 math on the same (bf16-representable) inputs; for the DDIM
step and the epipolar mask the CPU oracle.  Tolerances are stated per test: outputs that
are stored as bf16 carry a 2^-9 relative rounding, fp32 outputs only the accumulation order.
"""
import math
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

torch.set_grad_enabled(False)


@pytest.fixture(scope="module")
def ops():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from camc2v_amd import ops as o
    return o


def dev():
    return torch.device("cuda:0")


def rnd(*shape, seed=0, scale=1.0, dtype=torch.bfloat16):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(*shape, generator=g) * scale).to(dtype).to(dev())


def ccv_patch_row_py(idx, hw, w):
    """camc2v_amd/csrc/ccv_common.h: ccv_patch_row (token index in 4x8-patch order -> stored raster row)."""
    f, rem = divmod(idx, hw)
    patch, within = rem >> 5, rem & 31
    ppr = w >> 3
    if ppr % 2 == 0 and (hw // w) % 8 == 0:      # 2x2 quads of patches, quads row-major
        quad, sub = patch >> 2, patch & 3
        qy, qx = divmod(quad, ppr >> 1)
        py, px = 2 * qy + (sub >> 1), 2 * qx + (sub & 1)
    else:
        py, px = divmod(patch, ppr)
    return f * hw + (py * 4 + (within >> 3)) * w + px * 8 + (within & 7)


L2_REPORT = os.environ.get("CCV_TEST_L2_REPORT")     # a file: every comparison's (what, tol, rel-L2) is appended (bound-setting aid)


def assert_close(got, ref, tol, what="", l2=None):
    """Two bounds.  (1) max |err| <= tol * max |ref| (range-relative: catches a wrong element anywhere).  (2) relative L2
    ||got - ref|| / ||ref|| <= l2, default 0.4 * tol -- 4e-3 for the bf16-output cases (tol 1e-2), 8e-4 for the fp32-output cases on
    bf16-representable inputs (tol 2e-3): the range metric alone would pass a systematic ~1 % error on the many small elements
    (round-3 review); a rounding-level error has rel-L2 ~ 2^-9 / sqrt(3) ~ 1.1e-3 for one bf16 rounding of the output."""
    got, ref = got.float().cpu(), ref.float().cpu()
    err = (got - ref).abs().max().item()
    scale = ref.abs().max().item()
    assert math.isfinite(err), f"{what}: non-finite output"
    assert err <= tol * max(scale, 1e-6), f"{what}: max abs err {err:.4e} vs absmax {scale:.4e} (tol {tol})"
    rl2 = ((got - ref).norm() / ref.norm().clamp_min(1e-30)).item()
    if L2_REPORT:
        with open(L2_REPORT, "a") as f:
            f.write(f"{what}\t{tol}\t{rl2:.3e}\n")
        return
    bound = 0.4 * tol if l2 is None else l2
    assert rl2 <= bound, f"{what}: rel-L2 {rl2:.3e} > {bound:.1e} (max-norm bound {tol} passed)"


# ------------------------------------------------------------------------------------------
# GEMM family
# ------------------------------------------------------------------------------------------
@pytest.mark.parametrize("M,N,K", [(300, 320, 320), (4096, 1280, 640), (2, 1280, 320), (16384, 320, 320),
                                   (1000, 64, 128), (512, 2560, 1280)])
def test_gemm_linear_bf16(ops, M, N, K):
    a, w = rnd(M, K, seed=1), rnd(N, K, seed=2, scale=0.05)
    bias = rnd(N, seed=3, dtype=torch.float32)
    out = ops.gemm(a, w, bias=bias)
    ref = a.float() @ w.float().t() + bias
    assert_close(out, ref, 1e-2, f"linear {M}x{N}x{K}")
    out32 = ops.gemm(a, w, bias=bias, out_f32=True)
    assert_close(out32, ref, 2e-3, "linear fp32 out")


def test_gemm_fp32_a_residual_act(ops):
    M, N, K = 1024, 640, 320
    a = rnd(M, K, seed=4, dtype=torch.float32)
    w = rnd(N, K, seed=5, scale=0.05)
    res = rnd(M, N, seed=6, dtype=torch.float32)
    out = ops.gemm(a, w, residual=res, out_f32=True)
    ref = a.to(torch.bfloat16).float() @ w.float().t() + res
    assert_close(out, ref, 2e-3, "fp32 A + residual")
    # in-place accumulation into the stream (C aliases residual)
    stream = res.clone()
    ops.gemm(a, w, residual=stream, out_f32=True, out=stream)
    assert_close(stream, ref, 2e-3, "in-place residual")
    for act, fn in ((ops.ACT_SILU, F.silu), (ops.ACT_GELU, F.gelu)):
        out = ops.gemm(a, w, act=act)
        assert_close(out, fn(a.to(torch.bfloat16).float() @ w.float().t()), 1e-2, f"act {act}")


@pytest.mark.parametrize("M", [1, 2, 3, 4])
@pytest.mark.parametrize("a_f32", [False, True])
def test_gemm_row_vector_kernel(ops, M, a_f32, monkeypatch):
    """M <= 4 rows (the timestep / frame-stride MLPs, the fused ResBlock embedding projection) take the row-vector kernel: same
    epilogues as the tiled kernels, checked against torch fp32 and against the tiled kernel itself (CCV_GEMM_SKINNY=0 is read once
    per process, so the tiled result comes from M = 5 rows with the first M compared)."""
    ops.TRACK_GEMM_PLAN = True
    try:
        for (N, K) in ((1280, 320), (1280, 1280), (21120, 1280), (64, 64)):
            a = rnd(M, K, seed=40 + M, dtype=torch.float32 if a_f32 else torch.bfloat16)
            w, bias = rnd(N, K, seed=41, scale=0.05), rnd(N, seed=42, dtype=torch.float32)
            ref = a.to(torch.bfloat16).float() @ w.float().t() + bias
            out = ops.gemm(a, w, bias=bias, out_f32=True)
            assert ops.LAST_GEMM_PLAN == (-5, 1), ops.LAST_GEMM_PLAN
            assert_close(out, ref, 2e-3, f"row vector {M}x{N}x{K} fp32 out")
            assert_close(ops.gemm(a, w, bias=bias, act=ops.ACT_SILU), F.silu(ref), 1e-2, "row vector + SiLU, bf16 out")
            res = rnd(M, N, seed=43, dtype=torch.float32)
            assert_close(ops.gemm(a, w, bias=bias, residual=res, out_f32=True), ref + res, 2e-3, "row vector + residual")
            a5 = a.repeat(5, 1)[:5].contiguous()
            tiled = ops.gemm(a5, w, bias=bias, out_f32=True)
            assert ops.LAST_GEMM_PLAN[0] != -5
            assert_close(out, tiled[:M], 1e-3, "row vector vs tiled kernel")
    finally:
        ops.TRACK_GEMM_PLAN = False


def test_gemm_bias2_strided(ops):
    M, N, K, nb = 2048, 320, 320, 2
    a, w = rnd(M, K, seed=7), rnd(N, K, seed=8, scale=0.05)
    table = rnd(nb, 1000, seed=9, dtype=torch.float32)
    off = 200
    out = ops.gemm(a, w, bias2=table[:, off:], ldb2=1000, rows_per_batch=M // nb)
    ref = a.float() @ w.float().t() + table[:, off:off + N].repeat_interleave(M // nb, 0)
    assert_close(out, ref, 1e-2, "bias2")


def test_gemm_geglu(ops):
    M, C = 1024, 320
    a = rnd(M, C, seed=10)
    w = rnd(8 * C, C, seed=11, scale=0.05)          # rows [0,4C) value, [4C,8C) gate
    bias = rnd(8 * C, seed=12, dtype=torch.float32)
    from camc2v_amd.pack import interleave_geglu
    wp, bp = interleave_geglu(w, bias)
    out = ops.gemm(a, wp, bias=bp, geglu=True)
    full = a.float() @ w.float().t() + bias
    val, gate = full.chunk(2, dim=-1)
    assert out.shape == (M, 4 * C)
    assert_close(out, val * F.gelu(gate), 1.5e-2, "geglu")


@pytest.mark.parametrize("stride,upsample,a_f32", [(1, 0, False), (2, 0, True), (1, 1, True), (1, 0, True)])
def test_gemm_conv3x3(ops, stride, upsample, a_f32):
    from camc2v_amd.pack import pack_conv3x3
    n, cin, cout, hs, ws = 3, 128, 192, 12, 10
    x = rnd(n, cin, hs, ws, seed=13, dtype=torch.float32)
    x = x.to(torch.bfloat16).float()  # bf16 representable
    wt = rnd(cout, cin, 3, 3, seed=14, scale=0.05, dtype=torch.float32).to(torch.bfloat16).float()
    bias = rnd(cout, seed=15, dtype=torch.float32)
    src = F.interpolate(x, scale_factor=2, mode="nearest") if upsample else x
    ref = F.conv2d(src, wt, bias, stride=stride, padding=1)
    oh, ow = ref.shape[-2:]
    rows = x.permute(0, 2, 3, 1).reshape(-1, cin).contiguous()
    a = rows if a_f32 else rows.to(torch.bfloat16)
    out = ops.gemm(a, pack_conv3x3(wt), k=cin, taps=9, m=n * oh * ow, bias=bias, gather=ops.GATHER_CONV3X3,
                   conv=(oh, ow, hs, ws, stride, upsample), out_f32=True)
    assert_close(out.reshape(n, oh, ow, cout).permute(0, 3, 1, 2), ref, 2e-3, "conv3x3")


def test_gemm_tconv3(ops):
    from camc2v_amd.pack import pack_tconv3
    b, c, t, hw, cout = 2, 128, 16, 20, 128
    x = rnd(b, c, t, hw, 1, seed=16, dtype=torch.float32).to(torch.bfloat16).float()
    wt = rnd(cout, c, 3, 1, 1, seed=17, scale=0.05, dtype=torch.float32).to(torch.bfloat16).float()
    bias = rnd(cout, seed=18, dtype=torch.float32)
    ref = F.conv3d(x, wt, bias, padding=(1, 0, 0))  # [b, cout, t, hw, 1]
    rows = x[..., 0].permute(0, 2, 3, 1).reshape(-1, c).to(torch.bfloat16).contiguous()
    out = ops.gemm(rows, pack_tconv3(wt), k=c, taps=3, bias=bias, gather=ops.GATHER_TCONV3, tconv=(t, hw), out_f32=True)
    assert_close(out.reshape(b, t, hw, cout).permute(0, 3, 1, 2), ref[..., 0], 2e-3, "tconv3")


def test_gemm_split_k_paths(ops):
    """Few output tiles + long K -> the library splits K over extra workgroups (workspace provided by ops.gemm);
    every epilogue flavour must survive the two-pass form."""
    from camc2v_amd.pack import interleave_geglu, pack_conv3x3
    M, N, K = 512, 1280, 5120
    a, w = rnd(M, K, seed=70), rnd(N, K, seed=71, scale=0.02)
    bias = rnd(N, seed=72, dtype=torch.float32)
    res = rnd(M, N, seed=73, dtype=torch.float32)
    ref = a.float() @ w.float().t() + bias
    assert_close(ops.gemm(a, w, bias=bias), ref, 1e-2, "split-K bf16 out")
    stream = res.clone()
    ops.gemm(a, w, bias=bias, residual=stream, out_f32=True, out=stream)
    assert_close(stream, ref + res, 2e-3, "split-K in-place residual")
    assert_close(ops.gemm(a, w, bias=bias, act=ops.ACT_SILU), F.silu(ref), 1e-2, "split-K silu")
    wg = rnd(2 * 640, K, seed=74, scale=0.02)
    bg = rnd(2 * 640, seed=75, dtype=torch.float32)
    wp, bp = interleave_geglu(wg, bg)
    full = a.float() @ wg.float().t() + bg
    val, gate = full.chunk(2, dim=-1)
    assert_close(ops.gemm(a, wp, bias=bp, geglu=True), val * F.gelu(gate), 1.5e-2, "split-K geglu")
    # 4x4-latent conv: 32 frames x 16 pixels, 1280 -> 1280
    n, cin, cout, hs = 32, 1280, 1280, 4
    x = rnd(n, cin, hs, hs, seed=76, dtype=torch.float32).to(torch.bfloat16).float()
    wt = rnd(cout, cin, 3, 3, seed=77, scale=0.02, dtype=torch.float32).to(torch.bfloat16).float()
    ref = F.conv2d(x, wt, None, padding=1)
    rows = x.permute(0, 2, 3, 1).reshape(-1, cin).to(torch.bfloat16).contiguous()
    out = ops.gemm(rows, pack_conv3x3(wt), k=cin, taps=9, gather=ops.GATHER_CONV3X3, conv=(hs, hs, hs, hs, 1, 0), out_f32=True)
    assert_close(out.reshape(n, hs, hs, cout).permute(0, 3, 1, 2), ref, 2e-3, "split-K conv")


def test_gemm_wide_tile_paths(ops):
    """Shapes that take the 128x320 4-stage LDS-DMA kernel (N % 320 == 0, >= 192 tiles): linear with every
    epilogue, conv3x3 (padding rows fetch the zero line), temporal conv, GEGLU, ragged M."""
    from camc2v_amd.pack import interleave_geglu, pack_conv3x3, pack_tconv3
    M, N, K = 8192 + 40, 960, 2048           # ragged M: last tile partially out of range
    a, w = rnd(M, K, seed=80), rnd(N, K, seed=81, scale=0.02)
    bias = rnd(N, seed=82, dtype=torch.float32)
    ref = a.float() @ w.float().t() + bias
    assert_close(ops.gemm(a, w, bias=bias), ref, 1e-2, "wide linear")
    res = rnd(M, N, seed=83, dtype=torch.float32)
    stream = res.clone()
    ops.gemm(a, w, bias=bias, residual=stream, out_f32=True, out=stream)
    assert_close(stream, ref + res, 2e-3, "wide in-place residual")
    M2 = 16384
    a2 = rnd(M2, 2048, seed=84)
    wg, bg = rnd(640, 2048, seed=85, scale=0.02), rnd(640, seed=86, dtype=torch.float32)
    wp, bp = interleave_geglu(wg, bg)
    val, gate = (a2.float() @ wg.float().t() + bg).chunk(2, dim=-1)
    assert_close(ops.gemm(a2, wp, bias=bp, geglu=True), val * F.gelu(gate), 1.5e-2, "wide geglu")
    # conv3x3: 32 frames of 32x32, 64 -> 320 channels, with the per-clip embedding bias
    n, cin, cout, hs = 32, 256, 320, 32
    x = rnd(n, cin, hs, hs, seed=87, dtype=torch.float32).to(torch.bfloat16).float()
    wt = rnd(cout, cin, 3, 3, seed=88, scale=0.05, dtype=torch.float32).to(torch.bfloat16).float()
    cb = rnd(cout, seed=89, dtype=torch.float32)
    emb = rnd(2, cout, seed=90, dtype=torch.float32)
    refc = F.conv2d(x, wt, cb, padding=1) + emb.repeat_interleave(16, 0)[:, :, None, None]
    rows = x.permute(0, 2, 3, 1).reshape(-1, cin).to(torch.bfloat16).contiguous()
    out = ops.gemm(rows, pack_conv3x3(wt), k=cin, taps=9, bias=cb, bias2=emb, ldb2=cout, rows_per_batch=16 * hs * hs,
                   gather=ops.GATHER_CONV3X3, conv=(hs, hs, hs, hs, 1, 0), out_f32=True)
    assert_close(out.reshape(n, hs, hs, cout).permute(0, 3, 1, 2), refc, 2e-3, "wide conv3x3")
    # temporal conv: 2 clips x 16 frames x 1024 pixels
    b, c, t, hw = 2, 704, 16, 1024
    xt = rnd(b, c, t, hw, 1, seed=91, dtype=torch.float32).to(torch.bfloat16).float()
    wtt = rnd(320, c, 3, 1, 1, seed=92, scale=0.05, dtype=torch.float32).to(torch.bfloat16).float()
    reft = F.conv3d(xt, wtt, None, padding=(1, 0, 0))
    rows = xt[..., 0].permute(0, 2, 3, 1).reshape(-1, c).to(torch.bfloat16).contiguous()
    out = ops.gemm(rows, pack_tconv3(wtt), k=c, taps=3, gather=ops.GATHER_TCONV3, tconv=(t, hw), out_f32=True)
    assert_close(out.reshape(b, t, hw, 320).permute(0, 3, 1, 2), reft[..., 0], 2e-3, "wide tconv3")


@pytest.mark.parametrize("ring,split", [(0, 1), (0, 2), (1, 1), (1, 3), (2, 1), (2, 4), (3, 1), (3, 2), (4, 1), (4, 2), (5, 1), (5, 2)])
def test_gemm_ring_configs(ops, ring, split, monkeypatch):
    """Every LDS-ring tile configuration (128x320, 64x320, 128x160, 64x160 with a 4- and 8-deep ring), unsplit and
    split-K, forced through the tuning variables: ragged-M linear with residual, GEGLU, conv3x3 with padding,
    temporal conv; each against the fp32 torch reference of the same op."""
    from camc2v_amd.pack import interleave_geglu, pack_conv3x3, pack_tconv3
    monkeypatch.setenv("CCV_GEMM_RING", str(ring))
    monkeypatch.setenv("CCV_GEMM_SPLIT", str(split))
    monkeypatch.setattr(ops, "TRACK_GEMM_PLAN", True)
    M, N, K = 1024 + 40, 640, 2048
    a, w = rnd(M, K, seed=180), rnd(N, K, seed=181, scale=0.02)
    bias, res = rnd(N, seed=182, dtype=torch.float32), rnd(M, N, seed=183, dtype=torch.float32)
    ref = a.float() @ w.float().t() + bias
    assert_close(ops.gemm(a, w, bias=bias, residual=res, out_f32=True), ref + res, 2e-3, "ring linear + residual")
    assert ops.LAST_GEMM_PLAN == (ring, split)
    assert_close(ops.gemm(a, w, bias=bias, act=ops.ACT_SILU), F.silu(ref), 1e-2, "ring linear + SiLU, bf16 out")
    if ring in (0, 1, 5):   # GEGLU pairs 16-column groups inside a wave tile: 320-wide tiles only
        wg, bg = rnd(640, K, seed=185, scale=0.02), rnd(640, seed=186, dtype=torch.float32)
        wp, bp = interleave_geglu(wg, bg)
        val, gate = (a.float() @ wg.float().t() + bg).chunk(2, dim=-1)
        assert_close(ops.gemm(a, wp, bias=bp, geglu=True), val * F.gelu(gate), 1.5e-2, "ring geglu")
        assert ops.LAST_GEMM_PLAN == (ring, split)
    n, cin, cout, hs = 6, 256, 320, 12
    x = rnd(n, cin, hs, hs, seed=187, dtype=torch.float32).to(torch.bfloat16).float()
    wt = rnd(cout, cin, 3, 3, seed=188, scale=0.05, dtype=torch.float32).to(torch.bfloat16).float()
    cb = rnd(cout, seed=189, dtype=torch.float32)
    rows = x.permute(0, 2, 3, 1).reshape(-1, cin).to(torch.bfloat16).contiguous()
    out = ops.gemm(rows, pack_conv3x3(wt), k=cin, taps=9, bias=cb, gather=ops.GATHER_CONV3X3, conv=(hs, hs, hs, hs, 1, 0), out_f32=True)
    assert ops.LAST_GEMM_PLAN == (ring, split)
    assert_close(out.reshape(n, hs, hs, cout).permute(0, 3, 1, 2), F.conv2d(x, wt, cb, padding=1), 2e-3, "ring conv3x3")
    b, c, t, hw = 2, 512, 8, 36
    xt = rnd(b, c, t, hw, 1, seed=191, dtype=torch.float32).to(torch.bfloat16).float()
    wtt = rnd(320, c, 3, 1, 1, seed=192, scale=0.05, dtype=torch.float32).to(torch.bfloat16).float()
    rows = xt[..., 0].permute(0, 2, 3, 1).reshape(-1, c).to(torch.bfloat16).contiguous()
    out = ops.gemm(rows, pack_tconv3(wtt), k=c, taps=3, gather=ops.GATHER_TCONV3, tconv=(t, hw), out_f32=True)
    assert ops.LAST_GEMM_PLAN == (ring, min(split, 3))   # 48 slabs of 32: at most 3 splits of >= 16
    assert_close(out.reshape(b, t, hw, 320).permute(0, 3, 1, 2), F.conv3d(xt, wtt, None, padding=(1, 0, 0))[..., 0], 2e-3, "ring tconv3")


@pytest.mark.parametrize("tile,split", [(45, 1), (45, 4), (25, 1), (25, 2)])
def test_gemm_family_160_column_tiles(ops, tile, split, monkeypatch):
    """The family kernel (two stages of 64-deep slabs) on its 128x160 / 64x160 tiles, forced through the tuning variables,
    unsplit and split-K: ragged-M linear with residual, conv3x3 with padding (stride 1 and 2), temporal conv; each against
    the fp32 torch reference of the same op.  (The planner picks the 128x160 tile for the 32x32-latent convolutions and the
    long-K layers at 8x8 latents.)"""
    from camc2v_amd.pack import pack_conv3x3, pack_tconv3
    monkeypatch.setenv("CCV_GEMM_RING", "-1")
    monkeypatch.setenv("CCV_GEMM_FAMTILE", str(tile))
    monkeypatch.setenv("CCV_GEMM_SPLIT", str(split))
    monkeypatch.setattr(ops, "TRACK_GEMM_PLAN", True)
    M, N, K = 1024 + 40, 640, 2048
    a, w = rnd(M, K, seed=280), rnd(N, K, seed=281, scale=0.02)
    bias, res = rnd(N, seed=282, dtype=torch.float32), rnd(M, N, seed=283, dtype=torch.float32)
    ref = a.float() @ w.float().t() + bias
    assert_close(ops.gemm(a, w, bias=bias, residual=res, out_f32=True), ref + res, 2e-3, "160-column linear + residual")
    assert ops.LAST_GEMM_PLAN == (-1, split)
    assert_close(ops.gemm(a, w, bias=bias, act=ops.ACT_SILU), F.silu(ref), 1e-2, "160-column linear + SiLU, bf16 out")
    n, cin, cout, hs = 6, 256, 320, 12
    x = rnd(n, cin, hs, hs, seed=287, dtype=torch.float32).to(torch.bfloat16).float()
    wt = rnd(cout, cin, 3, 3, seed=288, scale=0.05, dtype=torch.float32).to(torch.bfloat16).float()
    cb = rnd(cout, seed=289, dtype=torch.float32)
    rows = x.permute(0, 2, 3, 1).reshape(-1, cin).to(torch.bfloat16).contiguous()
    out = ops.gemm(rows, pack_conv3x3(wt), k=cin, taps=9, bias=cb, gather=ops.GATHER_CONV3X3, conv=(hs, hs, hs, hs, 1, 0), out_f32=True)
    assert_close(out.reshape(n, hs, hs, cout).permute(0, 3, 1, 2), F.conv2d(x, wt, cb, padding=1), 2e-3, "160-column conv3x3")
    out = ops.gemm(rows, pack_conv3x3(wt), k=cin, taps=9, m=n * (hs // 2) ** 2, bias=cb, gather=ops.GATHER_CONV3X3,
                   conv=(hs // 2, hs // 2, hs, hs, 2, 0), out_f32=True)
    assert_close(out.reshape(n, hs // 2, hs // 2, cout).permute(0, 3, 1, 2), F.conv2d(x, wt, cb, padding=1, stride=2), 2e-3,
                 "160-column conv3x3 stride 2")
    from camc2v_amd.lib import CcvError
    with pytest.raises(CcvError):   # M taken from the source rows: the wrapper refuses instead of letting the gather run past A
        ops.gemm(rows, pack_conv3x3(wt), k=cin, taps=9, bias=cb, gather=ops.GATHER_CONV3X3, conv=(hs // 2, hs // 2, hs, hs, 2, 0), out_f32=True)
    b, c, t, hw = 2, 512, 8, 36
    xt = rnd(b, c, t, hw, 1, seed=291, dtype=torch.float32).to(torch.bfloat16).float()
    wtt = rnd(320, c, 3, 1, 1, seed=292, scale=0.05, dtype=torch.float32).to(torch.bfloat16).float()
    rows = xt[..., 0].permute(0, 2, 3, 1).reshape(-1, c).to(torch.bfloat16).contiguous()
    out = ops.gemm(rows, pack_tconv3(wtt), k=c, taps=3, gather=ops.GATHER_TCONV3, tconv=(t, hw), out_f32=True)
    assert_close(out.reshape(b, t, hw, 320).permute(0, 3, 1, 2), F.conv3d(xt, wtt, None, padding=(1, 0, 0))[..., 0], 2e-3, "160-column tconv3")


def test_gemm_rejects_bad_shapes(ops):
    from camc2v_amd.lib import CcvError
    with pytest.raises(CcvError):
        ops.gemm(rnd(64, 100), rnd(64, 100))          # K not a multiple of 64
    with pytest.raises(CcvError):
        ops.gemm(torch.zeros(64, 64, dtype=torch.bfloat16), torch.zeros(64, 64, dtype=torch.bfloat16))  # CPU tensors


# ------------------------------------------------------------------------------------------
# attention
# ------------------------------------------------------------------------------------------
def ref_attn(q, k, v, mask=None):
    """q [B,Lq,H,64] k,v [B,Lk,H,64] fp32; mask bool [B,Lq,Lk]."""
    sim = torch.einsum("bihd,bjhd->bhij", q, k) / 8.0
    if mask is not None:
        sim = sim.masked_fill(~mask[:, None], float("-inf"))
    return torch.einsum("bhij,bjhd->bihd", sim.softmax(-1), v)


@pytest.mark.parametrize("variant", [0, 1, 2])
@pytest.mark.parametrize("B,H,Lq,Lk", [(3, 5, 1024, 1024), (2, 2, 200, 77), (4, 1, 16, 16), (1, 3, 130, 333)])
def test_attention_plain(ops, variant, B, H, Lq, Lk):
    q, k, v = rnd(B, Lq, H, 64, seed=20), rnd(B, Lk, H, 64, seed=21), rnd(B, Lk, H, 64, seed=22)
    C = H * 64
    out = ops.attention(q, k, v, B=B, inner=1, H=H, Lq=Lq, Lk=Lk, q_str=(Lq * C, 0, C), k_str=(Lk * C, 0, C),
                        v_str=(Lk * C, 0, C), variant=variant)
    ref = ref_attn(q.float(), k.float(), v.float())
    assert_close(out.reshape(B, Lq, H, 64), ref, 1.5e-2, f"attention v{variant} {B},{H},{Lq},{Lk}")


@pytest.mark.parametrize("variant", [0, 1, 2])
def test_attention_fused_qkv_and_dual_context(ops, variant):
    """q/k/v as column slices of one fused [rows, 3C] projection; text + gated image context with
    K/V shared by all frames of a clip (inner batch stride 0)."""
    clips, frames, hw, H = 2, 4, 96, 2
    C = H * 64
    B = clips * frames
    qkv = rnd(B * hw, 3 * C, seed=23)
    out = ops.attention(qkv, qkv[:, C:], qkv[:, 2 * C:], B=B, inner=1, H=H, Lq=hw, Lk=hw,
                        q_str=(hw * 3 * C, 0, 3 * C), k_str=(hw * 3 * C, 0, 3 * C), v_str=(hw * 3 * C, 0, 3 * C),
                        variant=variant)
    x = qkv.float().reshape(B, hw, 3, H, 64)
    assert_close(out.reshape(B, hw, H, 64), ref_attn(x[:, :, 0], x[:, :, 1], x[:, :, 2]), 1.5e-2, "fused qkv")

    q = rnd(B * hw, C, seed=24)
    kv_t = rnd(clips * 77, 2 * C, seed=25)     # text: per clip
    kv_i = rnd(clips * frames * 16, 2 * C, seed=26)  # image: 16 tokens per frame
    gate = 1.37
    out = ops.attention(q, kv_t, kv_t[:, C:], B=B, inner=frames, H=H, Lq=hw, Lk=77,
                        q_str=(frames * hw * C, hw * C, C), k_str=(77 * 2 * C, 0, 2 * C), v_str=(77 * 2 * C, 0, 2 * C),
                        k2=kv_i, v2=kv_i[:, C:], k2_str=(frames * 16 * 2 * C, 16 * 2 * C, 2 * C),
                        v2_str=(frames * 16 * 2 * C, 16 * 2 * C, 2 * C), Lk2=16, gate2=gate, variant=variant)
    qf = q.float().reshape(B, hw, H, 64)
    kt = kv_t.float().reshape(clips, 77, 2, H, 64).repeat_interleave(frames, 0)
    ki = kv_i.float().reshape(B, 16, 2, H, 64)
    ref = ref_attn(qf, kt[:, :, 0], kt[:, :, 1]) + gate * ref_attn(qf, ki[:, :, 0], ki[:, :, 1])
    assert_close(out.reshape(B, hw, H, 64), ref, 1.5e-2, "dual context")


@pytest.mark.parametrize("Lq,Lk2,per_frame", [(1024, 768, False), (256, 256, False), (1024, 16, True), (200, 100, False), (64, 333, False)])
def test_attention_dual_context_shapes(ops, Lq, Lk2, per_frame):
    """Two-context cross-attention (77 text tokens + gated image tokens, separate softmaxes) in the LDS-DMA kernel at the
    model's shapes: image tokens shared by the frames of a clip or 16 per frame, ragged lengths, several clips."""
    clips, frames, H = 2, 4, 5
    C = H * 64
    B = clips * frames
    q = rnd(B * Lq, C, seed=124)
    kv_t = rnd(clips * 77, 2 * C, seed=125)
    n_img = clips * (frames if per_frame else 1) * Lk2
    kv_i = rnd(n_img, 2 * C, seed=126)
    gate = 0.83
    st_i = (frames * Lk2 * 2 * C, Lk2 * 2 * C, 2 * C) if per_frame else (Lk2 * 2 * C, 0, 2 * C)
    out = ops.attention(q, kv_t, kv_t[:, C:], B=B, inner=frames, H=H, Lq=Lq, Lk=77,
                        q_str=(frames * Lq * C, Lq * C, C), k_str=(77 * 2 * C, 0, 2 * C), v_str=(77 * 2 * C, 0, 2 * C),
                        k2=kv_i, v2=kv_i[:, C:], k2_str=st_i, v2_str=st_i, Lk2=Lk2, gate2=gate)
    qf = q.float().reshape(B, Lq, H, 64)
    kt = kv_t.float().reshape(clips, 77, 2, H, 64).repeat_interleave(frames, 0)
    ki = kv_i.float().reshape(-1, Lk2, 2, H, 64)
    if not per_frame:
        ki = ki.repeat_interleave(frames, 0)
    ref = ref_attn(qf, kt[:, :, 0], kt[:, :, 1]) + gate * ref_attn(qf, ki[:, :, 0], ki[:, :, 1])
    assert_close(out.reshape(B, Lq, H, 64), ref, 1.5e-2, f"dual context Lq={Lq} Lk2={Lk2} per_frame={per_frame}")
    # 32 queries per wave (attn2_kernel<.., .., 1>, variant 7) and 64 (variant 8): the same arithmetic per query -> bit for bit
    o78 = [ops.attention(q, kv_t, kv_t[:, C:], B=B, inner=frames, H=H, Lq=Lq, Lk=77,
                         q_str=(frames * Lq * C, Lq * C, C), k_str=(77 * 2 * C, 0, 2 * C), v_str=(77 * 2 * C, 0, 2 * C),
                         k2=kv_i, v2=kv_i[:, C:], k2_str=st_i, v2_str=st_i, Lk2=Lk2, gate2=gate, variant=vv) for vv in (7, 8)]
    assert torch.equal(o78[0], o78[1]) and torch.equal(out, o78[0])


@pytest.mark.parametrize("B,H,Lq,Lk", [(3, 5, 1024, 1024), (2, 2, 200, 77), (1, 3, 130, 333)])
def test_attention_32_queries_per_wave_equals_64(ops, B, H, Lq, Lk):
    """attn2_kernel<.., .., 1> (round 4: 32 queries per wave, four workgroups per CU; variant 7) against the 64-query form (variant 8),
    unmasked, ragged lengths: bit-identical, and both within the usual distance of torch fp32.  (The masked tiled kernel has no 32-query
    instance.)"""
    q, k, v = rnd(B, Lq, H, 64, seed=220), rnd(B, Lk, H, 64, seed=221), rnd(B, Lk, H, 64, seed=222)
    C = H * 64
    kw = dict(B=B, inner=1, H=H, Lq=Lq, Lk=Lk, q_str=(Lq * C, 0, C), k_str=(Lk * C, 0, C), v_str=(Lk * C, 0, C))
    a7, a8 = ops.attention(q, k, v, variant=7, **kw), ops.attention(q, k, v, variant=8, **kw)
    assert torch.equal(a7, a8)
    assert_close(a7.reshape(B, Lq, H, 64), ref_attn(q.float(), k.float(), v.float()), 1.5e-2, f"32 queries per wave {B},{H},{Lq},{Lk}")


@pytest.mark.parametrize("variant", [0, 1, 2])
@pytest.mark.parametrize("T", [16, 9])
def test_attention_temporal_strided(ops, variant, T):
    """batch = (clip, pixel), tokens = frames with stride hw*C (token-major activations untouched).
    variant 0 takes the one-wave-per-(pixel, head) kernel for T <= 16."""
    clips, hw, H = 2, 25, 3
    C = H * 64
    qkv = rnd(clips * T * hw, 3 * C, seed=27)
    ld = 3 * C
    out = torch.empty(clips * T * hw, C, dtype=torch.bfloat16, device=dev())
    ops.attention(qkv, qkv[:, C:], qkv[:, 2 * C:], B=clips * hw, inner=hw, H=H, Lq=T, Lk=T,
                  q_str=(T * hw * ld, ld, hw * ld), k_str=(T * hw * ld, ld, hw * ld), v_str=(T * hw * ld, ld, hw * ld),
                  out=out, o_str=(T * hw * C, C, hw * C), variant=variant)
    x = qkv.float().reshape(clips, T, hw, 3, H, 64).permute(0, 2, 1, 3, 4, 5).reshape(clips * hw, T, 3, H, 64)
    ref = ref_attn(x[:, :, 0], x[:, :, 1], x[:, :, 2])
    got = out.reshape(clips, T, hw, H, 64).permute(0, 2, 1, 3, 4).reshape(clips * hw, T, H, 64)
    assert_close(got, ref, 1.5e-2, "temporal")


@pytest.mark.parametrize("variant", [0, 1, 2])
@pytest.mark.parametrize("L,density", [(1024, 0.04), (320, 0.3), (48, 0.5)])
def test_attention_masked_register_tokens(ops, variant, L, density):
    """Epipolar form: bit mask + tile flags + always-visible register tokens; mask shared by the
    two halves of a CFG batch (mask_nb=1 while B=2).  Includes fully empty tiles and rows."""
    B, H, nreg = 2, 2, 4
    C = H * 64
    g = torch.Generator().manual_seed(30)
    mask = torch.rand(1, L, L, generator=g) < density
    mask[:, :, : L // 3] &= (torch.rand(1, L, 1, generator=g) < 0.5)   # some rows see nothing in a band
    if L >= 256:
        mask[:, :128, 64:192] = False                                   # an empty 128x128 region
        mask[:, 5] = False                                              # a row that only sees the registers
    mask = mask.to(dev())
    mp = ops.pack_mask(mask)
    bits, flags = mp
    qkv = rnd(B * L, 3 * C, seed=31)
    kreg, vreg = rnd(nreg, C, seed=32), rnd(nreg, C, seed=33)
    ld = 3 * C
    out = ops.attention(qkv, qkv[:, C:], qkv[:, 2 * C:], B=B, inner=1, H=H, Lq=L, Lk=L, q_str=(L * ld, 0, ld),
                        k_str=(L * ld, 0, ld), v_str=(L * ld, 0, ld), mask_bits=bits, mask_nb=1, tile_flags=flags,
                        kreg=kreg, vreg=vreg, variant=variant)
    x = qkv.float().reshape(B, L, 3, H, 64)
    k = torch.cat([kreg.float().reshape(1, nreg, H, 64).expand(B, -1, -1, -1), x[:, :, 1]], 1)
    v = torch.cat([vreg.float().reshape(1, nreg, H, 64).expand(B, -1, -1, -1), x[:, :, 2]], 1)
    m = F.pad(mask.expand(B, -1, -1), (nreg, 0), value=True)
    assert_close(out.reshape(B, L, H, 64), ref_attn(x[:, :, 0], k, v, m), 1.5e-2, f"masked L={L}")
    # without flags (only the in-kernel word test) the result is identical
    out2 = ops.attention(qkv, qkv[:, C:], qkv[:, 2 * C:], B=B, inner=1, H=H, Lq=L, Lk=L, q_str=(L * ld, 0, ld),
                         k_str=(L * ld, 0, ld), v_str=(L * ld, 0, ld), mask_bits=bits, mask_nb=1,
                         kreg=kreg, vreg=vreg, variant=variant)
    assert torch.equal(out, out2)
    # per-wave sparse kernel (block bitmap from the packer): same math, different schedule
    out3 = ops.attention(qkv, qkv[:, C:], qkv[:, 2 * C:], B=B, inner=1, H=H, Lq=L, Lk=L, q_str=(L * ld, 0, ld),
                         k_str=(L * ld, 0, ld), v_str=(L * ld, 0, ld), mask_bits=bits, mask_nb=1, tile_flags=flags,
                         wave_bits=mp.wave_bits, kreg=kreg, vreg=vreg, variant=3 if variant == 0 else variant)
    assert_close(out3.reshape(B, L, H, 64), ref_attn(x[:, :, 0], k, v, m), 1.5e-2, f"sparse kernel L={L}")
    wb = mp.wave_bits.cpu().numpy().view(np.uint32)
    need = mask[0].reshape(-1, L).cpu()
    nq, nk = (L + 63) // 64, (L + 31) // 32
    ref_need = torch.zeros(nq * 64, nk * 32, dtype=torch.bool)
    ref_need[:L, :L] = need
    ref_need = ref_need.reshape(nq, 64, nk, 32).any(3).any(1)
    got_need = np.unpackbits(wb.view(np.uint8), axis=-1, bitorder="little")[0, :, :nk].astype(bool)
    assert np.array_equal(got_need, ref_need.numpy())
    # longest-first schedule: groups by decreasing number of needed key blocks, ties by index; same result with it
    cnt = ref_need.sum(1).numpy()
    want = np.array(sorted(range(nq), key=lambda i: (-cnt[i], i)), dtype=np.int32)
    assert np.array_equal(mp.group_order.cpu().numpy()[0, :nq], want)
    # behind it: the workgroup-shared kernel's items (ops.WG_MERGE consecutive groups, union of their blocks), longest first
    mg = ops.WG_MERGE
    ni = (nq + mg - 1) // mg
    pad = np.zeros((ni * mg, ref_need.shape[1]), dtype=bool)
    pad[:nq] = ref_need.numpy()
    ucnt = pad.reshape(ni, mg, -1).any(1).sum(1)
    want_wg = np.array(sorted(range(ni), key=lambda i: (-ucnt[i], i)), dtype=np.int32)
    assert np.array_equal(mp.group_order.cpu().numpy()[0, nq:], want_wg)
    out4 = ops.attention(qkv, qkv[:, C:], qkv[:, 2 * C:], B=B, inner=1, H=H, Lq=L, Lk=L, q_str=(L * ld, 0, ld),
                         k_str=(L * ld, 0, ld), v_str=(L * ld, 0, ld), mask_bits=bits, mask_nb=1, tile_flags=flags,
                         wave_bits=mp.wave_bits, group_order=mp.group_order, kreg=kreg, vreg=vreg, variant=3 if variant == 0 else variant)
    assert torch.equal(out3, out4)   # each group is still computed by one wave in the same block order


@pytest.mark.parametrize("L,H,density,perm", [(1024, 2, 0.04, None), (320, 2, 0.3, None), (48, 2, 0.5, None), (6 * 128, 3, 0.06, (128, 16)),
                                              (5 * 288, 2, 0.05, (288, 24)), (16 * 256, 5, 0.02, (256, 16))])
def test_attention_shared_kv_kernel(ops, L, H, density, perm):
    """Round 4: the workgroup-shared sparse kernel (variant 4: 8 waves x 32 queries, variant 5: 4 waves; K / V blocks of the union of
    the workgroup's bitmaps staged once into a 4-deep ring) against torch fp32 and, BIT FOR BIT, against the per-wave sparse kernel
    (variant 6): every query meets the same key blocks in the same order in all three; ragged last groups, rows that only see
    the register tokens, empty bands, both patch-grid arithmetic paths, with and without the longest-first item order."""
    B, nreg = 2, 4
    C = H * 64
    g = torch.Generator().manual_seed(130 + L)
    mask = torch.rand(1, L, L, generator=g) < density
    mask[:, :, : L // 3] &= (torch.rand(1, L, 1, generator=g) < 0.5)
    if L >= 256:
        mask[:, :128, 64:192] = False
        mask[:, 5] = False
        mask[:, 200:240] = False              # a whole wave's queries (and more) see nothing but the registers
    mask = mask.to(dev())
    mp = ops.pack_mask(mask, perm)
    bits, flags = mp
    qkv = rnd(B * L, 3 * C, seed=131)
    kreg, vreg = rnd(nreg, C, seed=132), rnd(nreg, C, seed=133)
    ld = 3 * C
    st = (L * ld, 0, ld)
    kw = dict(B=B, inner=1, H=H, Lq=L, Lk=L, q_str=st, k_str=st, v_str=st, mask_bits=bits, mask_nb=1, tile_flags=flags, wave_bits=mp.wave_bits,
              kreg=kreg, vreg=vreg, perm=perm)
    kw["split"] = False          # key-split items have a test of their own below
    per_wave = ops.attention(qkv, qkv[:, C:], qkv[:, 2 * C:], variant=6, group_order=mp.group_order, **kw)
    x = qkv.float().reshape(B, L, 3, H, 64)
    k = torch.cat([kreg.float().reshape(1, nreg, H, 64).expand(B, -1, -1, -1), x[:, :, 1]], 1)
    v = torch.cat([vreg.float().reshape(1, nreg, H, 64).expand(B, -1, -1, -1), x[:, :, 2]], 1)
    m = F.pad(mask.expand(B, -1, -1), (nreg, 0), value=True)
    ref = ref_attn(x[:, :, 0], k, v, m)
    assert_close(per_wave.reshape(B, L, H, 64), ref, 1.5e-2, f"per-wave kernel L={L}")
    for variant in (4, 5):
        for order in (mp.group_order, None):
            out = ops.attention(qkv, qkv[:, C:], qkv[:, 2 * C:], variant=variant, group_order=order, **kw)
            assert_close(out.reshape(B, L, H, 64), ref, 1.5e-2, f"shared kernel v{variant} L={L}")
            assert torch.equal(out, per_wave), f"variant {variant} (order given: {order is not None}) differs from the per-wave kernel"
    # no register tokens
    kw2 = dict(kw, kreg=None, vreg=None)
    a = ops.attention(qkv, qkv[:, C:], qkv[:, 2 * C:], variant=6, group_order=mp.group_order, **kw2)
    b4 = ops.attention(qkv, qkv[:, C:], qkv[:, 2 * C:], variant=4, group_order=mp.group_order, **kw2)
    assert torch.equal(a, b4)


@pytest.mark.parametrize("T,hl,H,B,split_all", [(16, 16, 5, 2, 4),      # 640 items over 1024 workgroup slots: on request every item in 3 parts
                                                (16, 32, 5, 2, 0),      # 1280 items: the 260 shortest in 2 parts (the default rule)
                                                (32, 32, 5, 2, 0),      # a 32-frame clip: 2560 items of up to 1025 steps, 1540 of them in 2 parts
                                                (12, 16, 3, 1, 4)])     # 72 items: on request 4 parts
def test_attention_shared_kernel_key_split_items(ops, T, hl, H, B, split_all):
    """Key-split items of the workgroup-shared sparse kernel (the queue's tail in 2-4 parts, merged by the part that finishes last):
    against the unsplit kernel within ONE bf16 rounding realisation (a part rounds its P = exp2(s - m) against its own running maximum,
    so the products are another rounding of the same numbers: rel-L2 ~ 2e-3, not bit for bit; against torch fp32 the split and the
    whole result are equally close), bitwise reproducible from launch to launch although the merging workgroup is whoever arrives
    last, the workspace's counters back at zero, and unchanged results when the same workspace is reused by calls of other sizes."""
    from camc2v_amd import camera
    dev_ = dev()
    px = 8 * hl
    K = torch.tensor([[px / 2, 0, px / 2], [0, px / 2, px / 2], [0, 0, 1.0]], device=dev_).repeat(1, T, 1, 1)
    w2c = camera.synthetic_trajectory(1, T, dev_)
    Fm = camera.pairwise_fundamental(K, camera.relative_c2w(w2c, torch.zeros(1, dtype=torch.long, device=dev_)),
                                     generator=torch.Generator(device=dev_).manual_seed(9))
    mp = ops.epipolar_mask_bits(Fm, T, hl, hl, 8, patch_order=True)
    L, C = T * hl * hl, H * 64
    qkv = rnd(B * L, 3 * C, seed=140)
    kreg = rnd(4, C, seed=141)
    st = (L * 3 * C, 0, 3 * C)
    kw = dict(B=B, inner=1, H=H, Lq=L, Lk=L, q_str=st, k_str=st, v_str=st, mask_bits=mp[0], mask_nb=1, tile_flags=mp[1], wave_bits=mp.wave_bits,
              group_order=mp.group_order, kreg=kreg, vreg=kreg, perm=(hl * hl, hl), variant=5, split_all=split_all)
    whole = ops.attention(qkv, qkv[:, C:], qkv[:, 2 * C:], split=False, **kw)
    key = (dev_, torch.cuda.current_stream(dev_).cuda_stream)
    ops._SPLIT_WS.pop(key, None)
    outs = [ops.attention(qkv, qkv[:, C:], qkv[:, 2 * C:], **kw).clone() for _ in range(3)]
    assert key in ops._SPLIT_WS, "this problem is expected to split (ccv_attn_split_ws_bytes > 0)"
    assert torch.equal(outs[0], outs[1]) and torch.equal(outs[0], outs[2]), "key-split result is not reproducible"
    assert_close(outs[0], whole, 1e-2, f"key-split vs whole items T={T} {hl}x{hl}", l2=4e-3)
    if L <= 4096:      # both against fp32 attention: the same distance
        x = qkv.float().reshape(B, L, 3, H, 64)
        kk = torch.cat([kreg.float().reshape(1, 4, H, 64).expand(B, -1, -1, -1), x[:, :, 1]], 1)
        vv = torch.cat([kreg.float().reshape(1, 4, H, 64).expand(B, -1, -1, -1), x[:, :, 2]], 1)
        shifts = torch.arange(32, device=dev_, dtype=torch.int32)
        dense = ((mp[0].unsqueeze(-1) >> shifts) & 1).bool().reshape(1, L, -1)[:, :, :L]      # rows / columns in patch order
        order = torch.tensor([ccv_patch_row_py(i, hl * hl, hl) for i in range(L)], device=dev_)
        raster = torch.zeros((L, L), dtype=torch.bool, device=dev_)
        raster[order[:, None], order[None, :]] = dense[0]          # entry (i, j) of the patch-ordered mask belongs to stored rows order[i], order[j]
        m = F.pad(raster[None].expand(B, -1, -1), (4, 0), value=True)
        ref = ref_attn(x[:, :, 0], kk, vv, m)
        for o, what in ((outs[0], "key-split"), (whole, "whole items")):
            assert_close(o.reshape(B, L, H, 64), ref, 1.5e-2, f"{what} vs torch fp32, T={T} {hl}x{hl}")
    torch.cuda.synchronize()
    ws = ops._SPLIT_WS[key]
    assert int(ws[:256].view(torch.int32).abs().sum()) == 0, "the merging parts must leave the counters at zero"      # (>= 64 counters in every case here)
    # calls of other sizes (fewer / more split items: as the layers of one forward alternate) share the workspace; then this one again
    for other in (dict(kw, B=1), dict(kw, H=H - 1), dict(kw, B=1, H=2, split_all=3)):
        ops.attention(qkv, qkv[:, C:], qkv[:, 2 * C:], **other)
        again = ops.attention(qkv, qkv[:, C:], qkv[:, 2 * C:], **kw)
        assert torch.equal(again, outs[0]), f"result changed after a call with {other.get('B')}, {other.get('H')} sharing the workspace"


@pytest.mark.parametrize("fh,fw", [(8, 16), (12, 24)])
def test_attention_patch_order(ops, fh, fw):
    """4x8-patch token order: q/k/v/o rows stay in raster order in memory, the kernel walks them patch by patch and
    the mask is packed in the same order; the result must equal raster-order masked attention.  8x16 frames: the sparse
    kernel's shift arithmetic (power-of-two patch grid); 12x24: its general division path (288 tokens, 3 patches per row)."""
    B, H, T, nreg = 2, 2, 6, 4
    hw = fh * fw
    L = T * hw
    C = H * 64
    g = torch.Generator().manual_seed(95)
    mask = (torch.rand(1, L, L, generator=g) < 0.08).to(dev())
    qkv = rnd(B * L, 3 * C, seed=96)
    kreg, vreg = rnd(nreg, C, seed=97), rnd(nreg, C, seed=98)
    ld = 3 * C
    st = (L * ld, 0, ld)
    outs = []
    for perm in (None, (hw, fw)):
        mp = ops.pack_mask(mask, perm)
        bits, flags = mp
        outs.append(ops.attention(qkv, qkv[:, C:], qkv[:, 2 * C:], B=B, inner=1, H=H, Lq=L, Lk=L, q_str=st, k_str=st, v_str=st,
                                  mask_bits=bits, mask_nb=1, tile_flags=flags, kreg=kreg, vreg=vreg, perm=perm))
        o_sparse = ops.attention(qkv, qkv[:, C:], qkv[:, 2 * C:], B=B, inner=1, H=H, Lq=L, Lk=L, q_str=st, k_str=st, v_str=st,
                                 mask_bits=bits, mask_nb=1, wave_bits=mp.wave_bits, kreg=kreg, vreg=vreg, perm=perm, variant=3)
        assert_close(o_sparse, outs[-1], 1e-2, f"sparse vs tiled, perm={perm}")
    x = qkv.float().reshape(B, L, 3, H, 64)
    k = torch.cat([kreg.float().reshape(1, nreg, H, 64).expand(B, -1, -1, -1), x[:, :, 1]], 1)
    v = torch.cat([vreg.float().reshape(1, nreg, H, 64).expand(B, -1, -1, -1), x[:, :, 2]], 1)
    m = F.pad(mask.expand(B, -1, -1), (nreg, 0), value=True)
    ref = ref_attn(x[:, :, 0], k, v, m)
    assert_close(outs[0].reshape(B, L, H, 64), ref, 1.5e-2, "raster order")
    assert_close(outs[1].reshape(B, L, H, 64), ref, 1.5e-2, "patch order")
    # unmasked self attention is order independent too
    o_r = ops.attention(qkv, qkv[:, C:], qkv[:, 2 * C:], B=B, inner=1, H=H, Lq=L, Lk=L, q_str=st, k_str=st, v_str=st)
    o_p = ops.attention(qkv, qkv[:, C:], qkv[:, 2 * C:], B=B, inner=1, H=H, Lq=L, Lk=L, q_str=st, k_str=st, v_str=st, perm=(hw, fw))
    assert_close(o_p, o_r, 1e-2, "dense patch order")


def test_epipolar_mask_bits_patch_order(ops, golden_dir):
    """Native patch-order mask == pack_mask(perm) of the same mask in raster order (both built on the GPU)."""
    fx = np.load(os.path.join(golden_dir, "geometry.npz"))
    Fm = torch.from_numpy(fx["F64"]).to(dev())
    H = W = 8                                    # 64 px / d=8
    bits_r, _ = ops.epipolar_mask_bits(Fm, 16, H, W, 8)
    L = 16 * H * W
    raster = torch.from_numpy(np.unpackbits(bits_r.cpu().numpy().view(np.uint8), axis=-1, bitorder="little")[..., :L].astype(bool)).to(dev())
    bits_p, flags_p = ops.epipolar_mask_bits(Fm, 16, H, W, 8, patch_order=True)
    ref_bits, ref_flags = ops.pack_mask(raster, (H * W, W))
    assert torch.equal(bits_p, ref_bits) and torch.equal(flags_p, ref_flags)


def test_attention_softmax_rescale_spike(ops):
    """Force the online-softmax rescale: one key tile carries a huge score late in the sequence."""
    B, H, Lq, Lk = 1, 1, 64, 512
    q, k, v = rnd(B, Lq, H, 64, seed=34), rnd(B, Lk, H, 64, seed=35), rnd(B, Lk, H, 64, seed=36)
    k[0, 300, 0] = q[0, 7, 0] * 4.0     # spike for query 7 in tile 4
    k[0, 450, 0] = q[0, 40, 0] * 6.0
    out = ops.attention(q, k, v, B=B, inner=1, H=H, Lq=Lq, Lk=Lk, q_str=(Lq * 64, 0, 64), k_str=(Lk * 64, 0, 64),
                        v_str=(Lk * 64, 0, 64))
    assert_close(out.reshape(B, Lq, H, 64), ref_attn(q.float(), k.float(), v.float()), 1.5e-2, "spike")


# ------------------------------------------------------------------------------------------
# norms
# ------------------------------------------------------------------------------------------
@pytest.mark.parametrize("C,inst,rpi,f32,silu", [(320, 6, 1024, True, True), (2560, 4, 16, True, True),
                                                 (64, 2, 4096, False, False), (1280, 2, 1024, False, True),
                                                 (960, 3, 100, True, False),
                                                 # single-launch path (small per-group slices, >= 128 workgroups)
                                                 (320, 32, 16, True, True), (320, 32, 20, False, True),
                                                 (1280, 32, 64, False, True), (640, 16, 37, True, False)])
def test_groupnorm(ops, C, inst, rpi, f32, silu):
    x = rnd(inst * rpi, C, seed=40, dtype=torch.float32 if f32 else torch.bfloat16) * 2.0 + 0.5
    gamma = 1.0 + rnd(C, seed=41, dtype=torch.float32) * 0.1
    beta = rnd(C, seed=42, dtype=torch.float32) * 0.1
    y = ops.groupnorm(x, gamma, beta, instances=inst, eps=1e-5, silu=silu)
    xr = x.float().reshape(inst, rpi, C).permute(0, 2, 1)     # [inst, C, rpi]
    ref = F.group_norm(xr, 32, gamma, beta, 1e-5)
    if silu:
        ref = F.silu(ref)
    assert_close(y.reshape(inst, rpi, C), ref.permute(0, 2, 1), 1e-2, "groupnorm")


@pytest.mark.parametrize("frames,side,cin,cout", [(32, 32, 320, 320), (32, 16, 640, 640), (8, 16, 960, 640), (32, 8, 1280, 1280), (4, 32, 640, 320)])
def test_groupnorm_statistics_from_the_conv_epilogue(ops, frames, side, cin, cout):
    """ResBlock order conv3x3 (+bias, +per-frame embedding) -> GroupNorm+SiLU: the convolution's epilogue hands the norm its
    statistics (gn_rows = pixels per frame); the normalised result must match the norm that makes its own statistics pass over
    the same bf16 tensor (same values summed, in another order) and the fp32 reference."""
    from camc2v_amd import pack
    rows = frames * side * side
    x = rnd(rows, cin, seed=60)
    w = pack.pack_conv3x3((torch.randn(cout, cin, 3, 3, generator=torch.Generator().manual_seed(61)) * 0.02).to(dev()))
    bias = rnd(cout, seed=62, dtype=torch.float32) * 0.1
    emb = rnd(2, cout, seed=63, dtype=torch.float32) * 0.2
    gamma = 1.0 + rnd(cout, seed=64, dtype=torch.float32) * 0.1
    beta = rnd(cout, seed=65, dtype=torch.float32) * 0.1
    kw = dict(k=cin, taps=9, bias=bias, bias2=emb, ldb2=emb.stride(0), rows_per_batch=rows // 2, gather=ops.GATHER_CONV3X3,
              conv=(side, side, side, side, 1, 0))
    ops.TRACK_GEMM_PLAN = True
    try:
        h, st = ops.gemm(x, w, gn_rows=side * side, **kw)
        plan = ops.LAST_GEMM_PLAN
    finally:
        ops.TRACK_GEMM_PLAN = False
    h0 = ops.gemm(x, w, **kw)
    assert torch.equal(h, h0)
    y0 = ops.groupnorm(h0, gamma, beta, instances=frames, eps=1e-5, silu=True)
    if st is None:       # only where the norm itself is a single launch (small slices): statistics from the producer would save nothing
        from camc2v_amd.lib import lib
        assert lib().ccv_groupnorm_single_launch(frames, side * side, cout, 0) == 1, f"no epilogue statistics (plan {plan})"
        return
    part, gn_rows = st
    assert gn_rows == side * side and part.shape[0] == frames and part.shape[2] == 64
    # the slots add up to the plain sums of the stored values
    hv = h.float().reshape(frames, side * side, 32, cout // 32)
    want = torch.stack([hv.sum((1, 3)), (hv * hv).sum((1, 3))], -1).reshape(frames, 64)
    got = part.sum(1)
    assert_close(got, want, 2e-4, "epilogue statistics")
    y = ops.groupnorm(h, gamma, beta, instances=frames, eps=1e-5, silu=True, stats=st)
    d = (y.float() - y0.float()).abs()
    assert d.max().item() <= 2.0 ** -6 * max(1.0, y0.float().abs().max().item()), d.max().item()   # at most a bf16 rounding step apart
    assert (d > 0).float().mean().item() < 0.02
    ref = F.silu(F.group_norm(h.float().reshape(frames, side * side, cout).permute(0, 2, 1), 32, gamma, beta, 1e-5)).permute(0, 2, 1)
    assert_close(y.reshape(frames, side * side, cout), ref, 1e-2, "groupnorm on epilogue statistics")
    with pytest.raises(Exception):
        ops.groupnorm(h, gamma, beta, instances=frames // 2, eps=1e-5, silu=True, stats=st)


@pytest.mark.parametrize("C", [64, 320, 512, 640, 1280, 2048])
def test_layernorm(ops, C):
    rows = 1000
    x = rnd(rows, C, seed=43, dtype=torch.float32) * 3 + 1
    gamma = 1.0 + rnd(C, seed=44, dtype=torch.float32) * 0.1
    beta = rnd(C, seed=45, dtype=torch.float32) * 0.1
    ref = F.layer_norm(x, (C,), gamma, beta, 1e-5)
    assert_close(ops.layernorm(x, gamma, beta), ref, 8e-3, "layernorm")
    add = rnd(rows // 2, C, seed=46)
    y, y2 = ops.layernorm(x, gamma, beta, addend=add)
    assert_close(y, ref, 8e-3, "layernorm y")
    assert_close(y2, ref + add.float().repeat(2, 1), 8e-3, "layernorm y2")


# ------------------------------------------------------------------------------------------
# layout / elementwise / DDIM / masks
# ------------------------------------------------------------------------------------------
def test_layout_roundtrip(ops):
    b, t, h, w = 2, 16, 8, 8
    x = rnd(b, 4, t, h, w, seed=50, dtype=torch.float32)
    c = rnd(b, 4, t, h, w, seed=51, dtype=torch.float32)
    rows = ops.pack_nchw_to_rows(x, c, ldo=64)
    ref = torch.cat([x, c], 1).permute(0, 2, 3, 4, 1).reshape(-1, 8)
    assert torch.equal(rows[:, :8], ref) and float(rows[:, 8:].abs().max()) == 0.0
    back = ops.unpack_rows_to_nchw(rows, 8, b, t, h, w)
    assert torch.equal(back, torch.cat([x, c], 1))
    a, bb = rnd(100, 64, seed=52, dtype=torch.float32), rnd(100, 128, seed=53, dtype=torch.float32)
    assert torch.equal(ops.concat_rows(a, bb), torch.cat([a, bb], 1))
    cat32, cat16 = ops.concat_rows(a, bb, with_bf16=True)
    assert torch.equal(cat32, torch.cat([a, bb], 1)) and torch.equal(cat16, torch.cat([a, bb], 1).to(torch.bfloat16))
    f = rnd(2, 320, 16, 4, 4, seed=54, dtype=torch.float32)
    assert torch.equal(ops.nchw_to_rows_bf16(f), f.permute(0, 2, 3, 4, 1).reshape(-1, 320).to(torch.bfloat16))
    assert torch.equal(ops.cast_bf16(a), a.to(torch.bfloat16))


def test_timestep_embedding_and_add_silu(ops):
    from oracle.unet_oracle import timestep_embedding
    t = torch.tensor([999, 439, 39, 0, 8], dtype=torch.long)
    got = ops.timestep_embedding(t.to(dev()), 320)
    assert_close(got, timestep_embedding(t, 320), 6e-3, "timestep embedding")
    a, b = rnd(7, 1280, seed=55, dtype=torch.float32), rnd(7, 1280, seed=56, dtype=torch.float32)
    assert_close(ops.add_silu_bf16(a, b), F.silu(a + b), 6e-3, "add_silu")


def test_ddim_cfg_step_vs_oracle_and_golden(ops, golden_dir):
    from oracle import ddim_oracle
    fx = np.load(os.path.join(golden_dir, "ddim.npz"))
    tab = ddim_oracle.ddim_tables(25, 1.0)
    x, e_c, e_uc = (torch.from_numpy(fx[k]).to(dev()) for k in ("step_x", "step_e_c", "step_e_uc"))
    for index in (24, 7, 0):
        z = torch.from_numpy(fx[f"step{index}_noise"]).to(dev())
        coef = torch.tensor([tab["alphas"][index], tab["alphas_prev"][index], tab["sigmas"][index],
                             tab["sqrt_one_minus_alphas"][index]], dtype=torch.float32, device=dev())
        x_prev, x0 = ops.ddim_cfg_step(x, e_c, e_uc, z, coef, 7.5, 0.7)
        assert_close(x_prev, torch.from_numpy(fx[f"step{index}_x_prev"]), 2e-5, "x_prev vs reference fixture")
        assert_close(x0, torch.from_numpy(fx[f"step{index}_pred_x0"]), 2e-5, "pred_x0 vs reference fixture")
    z = torch.from_numpy(fx["noguid_noise"]).to(dev())
    coef = torch.tensor([tab["alphas"][3], tab["alphas_prev"][3], tab["sigmas"][3], tab["sqrt_one_minus_alphas"][3]],
                        dtype=torch.float32, device=dev())
    x_prev, _ = ops.ddim_cfg_step(x, e_c, None, z, coef, 1.0, 0.0)
    assert_close(x_prev, torch.from_numpy(fx["noguid_x_prev"]), 2e-5, "no guidance")


def test_pack_mask_matches_numpy(ops):
    g = torch.Generator().manual_seed(60)
    for (B, Lq, Lk) in ((2, 256, 256), (1, 130, 80), (1, 64, 16)):
        m = torch.rand(B, Lq, Lk, generator=g) < 0.2
        bits, flags = ops.pack_mask(m.to(dev()))
        ref = np.packbits(m.numpy().astype(np.uint8), axis=-1, bitorder="little")
        pad = (-ref.shape[-1]) % 4
        ref = np.pad(ref, ((0, 0), (0, 0), (0, pad))).view(np.uint32).astype(np.int64)
        assert np.array_equal(bits.cpu().numpy().view(np.uint32).astype(np.int64), ref)
        kt, qt = (Lk + 63) // 64, (Lq + 127) // 128
        mp = torch.zeros(B, qt * 128, kt * 64, dtype=torch.bool)
        mp[:, :Lq, :Lk] = m
        assert torch.equal(flags.cpu().bool(), mp.reshape(B, qt, 128, kt, 64).any(4).any(2))


def test_epipolar_mask_bits_vs_oracle(ops, golden_dir):
    """Native mask build from F: bit-exact (the kernel evaluates the two 3-term dot products as the reference's matrix
    products do: product, fma, add -- csrc/ccv_misc.hip dot3_chain), flags included."""
    from oracle import geometry_oracle
    fx = np.load(os.path.join(golden_dir, "geometry.npz"))
    for px, key in ((64, "F64"), (256, "F256")):
        Fm = torch.from_numpy(fx[key])
        for d in (8, 16, 32, 64):
            H = px // d
            if H * H * 16 > 4096:      # keep the CPU oracle side small
                continue
            ref = geometry_oracle.epipolar_mask(Fm, H, H, d)
            bits, flags = ops.epipolar_mask_bits(Fm.to(dev()), 16, H, H, d)
            refbits, refflags = ops.pack_mask(ref.to(dev()))
            diff = (bits ^ refbits).cpu().numpy().view(np.uint32)
            nflip = int(np.unpackbits(diff.view(np.uint8)).sum())
            assert nflip == 0, f"px={px} d={d}: {nflip} flipped mask bits"
            assert torch.equal(flags, refflags)


def test_sampler_camera_guidance_vs_reference_fixture(ops, golden_dir):
    """DDIMSampler.p_sample_ddim with camera_cfg != 1 (third forward without the camera, constant / cosine weight) against
    the reference sampler's outputs (oracle/gen_golden_camcfg.py); the model is a duck-typed object returning the fixture's
    predictions, so this pins the branch logic + ccv_camera_cfg_fold + ccv_ddim_cfg_step."""
    from camc2v_amd.sampler import DDIMSampler, make_beta_schedule
    fx = np.load(os.path.join(golden_dir, "ddim_camera_cfg.npz"))
    betas = make_beta_schedule("linear", 1000, 0.00085, 0.012)
    ac = np.cumprod(1.0 - betas)
    x, e_c, e_uc, e_nc = (torch.from_numpy(fx[k]).to(dev()) for k in ("x", "e_c", "e_uc", "e_nc"))

    class Duck:
        num_timesteps = 1000
        device = dev()
        parameterization = "eps"
        betas = torch.tensor(make_beta_schedule("linear", 1000, 0.00085, 0.012), dtype=torch.float32, device=dev())
        alphas_cumprod = torch.tensor(ac, dtype=torch.float32, device=dev())
        alphas_cumprod_prev = torch.tensor(np.append(1.0, ac[:-1]), dtype=torch.float32, device=dev())
        calls = []

        def apply_model(self, x_, t_, c_, **kw):
            cam = c_.get("camera_condition")
            which = "nc" if cam is None else ("uc" if cam.get("is_uc") else "c")
            self.calls.append(which)
            return {"c": e_c, "uc": e_uc, "nc": e_nc}[which][:x_.shape[0]]

    duck = Duck()
    s = DDIMSampler(duck)
    s.make_schedule(25, "uniform_trailing", 1.0, verbose=False)
    for scheduler, camera_cfg, index in (("constant", 2.0, 20), ("cosine", 1.5, 20), ("cosine", 3.0, 2)):
        tag = f"{scheduler}_{camera_cfg:g}_{index}"
        z, t = torch.from_numpy(fx[f"{tag}_noise"]).to(dev()), torch.from_numpy(fx[f"{tag}_t"]).to(dev())
        nb = z.shape[0]
        duck.calls.clear()
        cond = {"tag": "c", "camera_condition": {"cond_frame_index": torch.zeros(nb, dtype=torch.long)}}
        x_prev, x0 = s.p_sample_ddim(x[:nb].contiguous(), cond, t, index=index, unconditional_guidance_scale=7.5,
                                     unconditional_conditioning={"tag": "uc"}, guidance_rescale=0.7, noise=z,
                                     enable_camera_condition=True, camera_cfg=camera_cfg, camera_cfg_scheduler=scheduler)
        assert duck.calls == ["c", "uc", "nc"]
        assert_close(x_prev, torch.from_numpy(fx[f"{tag}_x_prev"]), 2e-5, f"camera guidance {tag}: x_prev")
        assert_close(x0, torch.from_numpy(fx[f"{tag}_pred_x0"]), 2e-5, f"camera guidance {tag}: pred_x0")
    # camera_cfg without enable_camera_condition: two forwards, plain guidance
    duck.calls.clear()
    t = torch.full((2,), int(s.ddim_timesteps[2]), dtype=torch.long, device=dev())
    x_prev, _ = s.p_sample_ddim(x, {"tag": "c", "camera_condition": {}}, t, index=2, unconditional_guidance_scale=7.5,
                                unconditional_conditioning={"tag": "uc", "camera_condition": {"is_uc": True}}, guidance_rescale=0.7,
                                noise=torch.from_numpy(fx["disabled_noise"]).to(dev()), camera_cfg=3.0)
    assert duck.calls == ["c", "uc"]
    assert_close(x_prev, torch.from_numpy(fx["disabled_x_prev"]), 2e-5, "camera_cfg ignored without enable_camera_condition")
    with pytest.raises(NotImplementedError):
        s.p_sample_ddim(x, {"tag": "c", "camera_condition": {}}, t, index=2, unconditional_guidance_scale=7.5,
                        unconditional_conditioning={"tag": "uc"}, enable_camera_condition=True, camera_cfg=2.0,
                        camera_cfg_scheduler="linear")


@pytest.mark.parametrize("cfg", ["auto", "ring2", "ring5s2", "fam45", "fam22"])
def test_gemm_stacked_segments(ops, cfg, monkeypatch):
    """gather 3: `taps` stacked operands [taps][seg_rows][K] against W = [W_0 | W_1 | W_2] -- three linear maps into one
    output as one GEMM (the fused stream update of a camera-conditioned temporal block) -- on every kernel family:
    planner's choice, LDS-ring tiles (unsplit / split-K), family tiles; ragged M, padded segments."""
    env = {"auto": {}, "ring2": {"CCV_GEMM_RING": "2"}, "ring5s2": {"CCV_GEMM_RING": "5", "CCV_GEMM_SPLIT": "2"},
           "fam45": {"CCV_GEMM_RING": "-1", "CCV_GEMM_FAMTILE": "45"}, "fam22": {"CCV_GEMM_RING": "-1", "CCV_GEMM_FAMTILE": "22"}}[cfg]
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    M, seg, N, K, taps = 1024 + 40, 1100, 320, 640, 3
    a = rnd(taps * seg, K, seed=300)
    w = rnd(N, taps * K, seed=301, scale=0.03)
    bias, res = rnd(N, seed=302, dtype=torch.float32), rnd(M, N, seed=303, dtype=torch.float32)
    ref = sum(a[t * seg:t * seg + M].float() @ w[:, t * K:(t + 1) * K].float().t() for t in range(taps)) + bias
    out = ops.gemm(a, w, k=K, taps=taps, m=M, gather=ops.GATHER_SEGMENTS, seg_rows=seg, bias=bias, residual=res, out_f32=True)
    assert_close(out, ref + res, 2e-3, f"segments {cfg}")
    stream = res.clone()       # in place, as the model uses it
    ops.gemm(a, w, k=K, taps=taps, m=M, gather=ops.GATHER_SEGMENTS, seg_rows=seg, bias=bias, residual=stream, out_f32=True, out=stream)
    assert torch.equal(stream, out)
    from camc2v_amd.lib import CcvError
    with pytest.raises(CcvError):
        ops.gemm(a, w, k=K, taps=taps, m=M, gather=ops.GATHER_SEGMENTS, seg_rows=M - 8, out_f32=True)   # segments would overlap
    with pytest.raises(CcvError):
        ops.gemm(a[:2 * seg], w, k=K, taps=taps, m=M, gather=ops.GATHER_SEGMENTS, seg_rows=seg, out_f32=True)   # third segment missing


@pytest.mark.parametrize("kind", ["qkv", "out_res", "proj_f32", "geglu", "bf16_bias", "narrow_ldc"])
def test_gemm_a_stationary_short_k(ops, kind, monkeypatch):
    """The A-stationary kernel (one workgroup per CU keeps its 128 activation rows in registers, weight strips stream
    through a 3-deep LDS ring) on the short-K linear layers of the 32x32-latent blocks: every epilogue flavour against fp32
    torch, the plan reports tile code -4, and CCV_GEMM_ASTAT-independent shapes (K != 320, small M) still take the tiled kernels."""
    monkeypatch.setattr(ops, "TRACK_GEMM_PLAN", True)
    M, K = 24576, 320
    a = rnd(M, K, seed=400)
    if kind == "qkv":                       # fused QKV projection: bf16 out, no bias
        w = rnd(960, K, seed=401, scale=0.05)
        out = ops.gemm(a, w)
        assert ops.LAST_GEMM_PLAN == (-4, 1)
        assert_close(out, a.float() @ w.float().t(), 1e-2, kind)
    elif kind == "out_res":                 # attention output projection accumulating into the fp32 stream, in place
        w, bias = rnd(320, K, seed=402, scale=0.05), rnd(320, seed=403, dtype=torch.float32)
        res = rnd(M, 320, seed=404, dtype=torch.float32)
        ref = a.float() @ w.float().t() + bias + res
        stream = res.clone()
        ops.gemm(a, w, bias=bias, residual=stream, out_f32=True, out=stream)
        assert ops.LAST_GEMM_PLAN == (-4, 1)
        assert_close(stream, ref, 2e-3, kind)
        assert_close(ops.gemm(a, w, residual=res, out_f32=True, alpha=0.5), 0.5 * (a.float() @ w.float().t()) + res, 2e-3, kind + " alpha, no bias")
    elif kind == "proj_f32":                # proj_in: fp32 out, bias, no residual
        w, bias = rnd(320, K, seed=405, scale=0.05), rnd(320, seed=406, dtype=torch.float32)
        assert_close(ops.gemm(a, w, bias=bias, out_f32=True), a.float() @ w.float().t() + bias, 2e-3, kind)
        assert ops.LAST_GEMM_PLAN == (-4, 1)
    elif kind == "geglu":
        from camc2v_amd.pack import interleave_geglu
        wg, bg = rnd(2560, K, seed=407, scale=0.05), rnd(2560, seed=408, dtype=torch.float32)
        wp, bp = interleave_geglu(wg, bg)
        val, gate = (a.float() @ wg.float().t() + bg).chunk(2, dim=-1)
        out = ops.gemm(a, wp, bias=bp, geglu=True)
        assert ops.LAST_GEMM_PLAN == (-4, 1) and out.shape == (M, 1280)
        assert_close(out, val * F.gelu(gate), 1.5e-2, kind)
    elif kind == "bf16_bias":
        w, bias = rnd(320, K, seed=409, scale=0.05), rnd(320, seed=410, dtype=torch.float32)
        assert_close(ops.gemm(a, w, bias=bias), a.float() @ w.float().t() + bias, 1e-2, kind)
        assert ops.LAST_GEMM_PLAN == (-4, 1)
    else:                                   # output rows that are not 16-byte aligned: the 8-byte store variant
        w = rnd(320, K, seed=411, scale=0.05)
        buf = torch.zeros(M, 324, dtype=torch.bfloat16, device=dev())
        out = ops.gemm(a, w, out=buf[:, :320])
        assert ops.LAST_GEMM_PLAN == (-4, 1) and buf[:, 320:].abs().max().item() == 0
        assert_close(out, a.float() @ w.float().t(), 1e-2, kind)
    # outside its range the planner still answers with the tiled kernels
    ops.gemm(rnd(4096, K, seed=412), rnd(320, K, seed=413))
    assert ops.LAST_GEMM_PLAN[0] != -4
    ops.gemm(rnd(M, 640, seed=414), rnd(320, 640, seed=415))
    assert ops.LAST_GEMM_PLAN[0] != -4


# ------------------------------------------------------------------------------------------
# fp16 residual stream (round 3): every kernel family's epilogue, the norms and the layout helpers on fp16 rows
# ------------------------------------------------------------------------------------------
def _f16(t):
    return t.to(torch.float16)


@pytest.mark.parametrize("M,N,K,taps", [(32768, 320, 320, 1),      # A-stationary kernel (K = 320, one 128-row workgroup per CU)
                                        (4096, 1280, 640, 1),      # family tile, two-stage loop
                                        (512, 1280, 1280, 1),      # family tile, 3-stage ring (M <= 1024)
                                        (512, 1280, 5120, 1),      # split-K: the reduce kernel owns the epilogue
                                        (300, 320, 320, 1),        # ragged rows
                                        (2048, 640, 640, 3)])      # stacked segments on a ring / family tile
def test_gemm_fp16_stream_epilogues(ops, M, N, K, taps):
    """out = A W^T + bias + residual with the residual and the output in fp16 (fp32 arithmetic): against torch fp32, tolerance =
    fp16's 2^-11 relative rounding of the stored sum + the accumulation order; in place (C aliases the residual) as the UNet runs it."""
    a = rnd(taps * M, K, seed=21)
    w = rnd(N, taps * K, seed=22, scale=0.05)
    bias = rnd(N, seed=23, dtype=torch.float32)
    res = _f16(rnd(M, N, seed=24, dtype=torch.float32))
    kw = dict(k=K, taps=taps, m=M, gather=ops.GATHER_SEGMENTS, seg_rows=M) if taps > 1 else {}
    ref = sum(a[t * M:(t + 1) * M].float() @ w[:, t * K:(t + 1) * K].float().t() for t in range(taps)) + bias + res.float()
    out = ops.gemm(a, w, bias=bias, residual=res, out_dtype=torch.float16, **kw)
    assert out.dtype == torch.float16
    assert_close(out, ref, 1.5e-3, f"fp16 stream {M}x{N}x{K} taps {taps}")
    stream = res.clone()
    ops.gemm(a, w, bias=bias, residual=stream, out_dtype=torch.float16, out=stream, **kw)
    assert torch.equal(stream, out), "in-place update differs from the out-of-place one"
    # fp16 residual into a bf16 output (last feed-forward of a transformer) and a fresh fp16 output without residual (proj_in)
    outb = ops.gemm(a, w, bias=bias, residual=res, **kw)
    assert outb.dtype == torch.bfloat16
    assert_close(outb, ref, 1e-2, "fp16 residual, bf16 out")
    outn = ops.gemm(a, w, bias=bias, out_dtype=torch.float16, **kw)
    assert_close(outn, ref - res.float(), 1.5e-3, "fp16 out, no residual")


@pytest.mark.parametrize("M,N,K", [(1000, 320, 640), (300, 64, 320), (8192, 640, 640), (2048, 1280, 1280)])
def test_gemm_row_epilogue_edges_match_the_direct_epilogue(ops, M, N, K):
    """The family kernel's epilogue through LDS rows (two-byte outputs, 16-byte aligned rows) against the direct epilogue it replaces:
    a column slice of a wider matrix whose row pitch is NOT a multiple of 8 elements takes the direct path, and both must give the same
    bits -- ragged M (rows past M in the last tile), N smaller than a tile, in-place fp16 stream update, bf16 output."""
    a, w = rnd(M, K, seed=61), rnd(N, K, seed=62, scale=0.05)
    bias = rnd(N, seed=63, dtype=torch.float32)
    res = _f16(rnd(M, N, seed=64, dtype=torch.float32))
    ref = a.float() @ w.float().t() + bias + res.float()
    rows = ops.gemm(a, w, bias=bias, residual=res, out_dtype=torch.float16)                       # aligned: rows epilogue
    assert_close(rows, ref, 1.5e-3, f"rows epilogue {M}x{N}x{K}")
    wide_res = torch.zeros(M, N + 4, dtype=torch.float16, device=res.device)                       # pitch N + 4: rows only 8-byte aligned
    wide_res[:, :N] = res
    wide_out = torch.zeros(M, N + 4, dtype=torch.float16, device=res.device)
    direct = ops.gemm(a, w, bias=bias, residual=wide_res[:, :N], out_dtype=torch.float16, out=wide_out[:, :N])
    assert torch.equal(direct, rows), "direct and row epilogues differ"
    assert torch.count_nonzero(wide_out[:, N:]) == 0, "the epilogue wrote past column N"
    stream = res.clone()
    ops.gemm(a, w, bias=bias, residual=stream, out_dtype=torch.float16, out=stream)
    assert torch.equal(stream, rows), "in-place stream update differs"
    b16 = ops.gemm(a, w, bias=bias)
    wide_b = torch.zeros(M, N + 4, dtype=torch.bfloat16, device=res.device)
    assert torch.equal(ops.gemm(a, w, bias=bias, out=wide_b[:, :N]), b16), "bf16: direct and row epilogues differ"


def test_gemm_fp16_stream_convs(ops):
    """conv3x3 (+ fp16 residual) and the temporal convolution writing the fp16 stream."""
    from camc2v_amd.pack import pack_conv3x3, pack_tconv3
    n, cin, cout, hs = 16, 320, 320, 16
    x = rnd(n, cin, hs, hs, seed=25, dtype=torch.float32).to(torch.bfloat16).float()
    wt = rnd(cout, cin, 3, 3, seed=26, scale=0.03, dtype=torch.float32).to(torch.bfloat16).float()
    bias = rnd(cout, seed=27, dtype=torch.float32)
    res = _f16(rnd(n * hs * hs, cout, seed=28, dtype=torch.float32))
    ref = F.conv2d(x, wt, bias, padding=1).permute(0, 2, 3, 1).reshape(-1, cout) + res.float()
    rows = x.permute(0, 2, 3, 1).reshape(-1, cin).to(torch.bfloat16).contiguous()
    out = ops.gemm(rows, pack_conv3x3(wt), k=cin, taps=9, bias=bias, residual=res, gather=ops.GATHER_CONV3X3, conv=(hs, hs, hs, hs, 1, 0),
                   out_dtype=torch.float16)
    assert_close(out, ref, 1.5e-3, "conv3x3 fp16 stream")
    b, c, t, hw = 2, 320, 16, 64
    xt = rnd(b, c, t, hw, 1, seed=29, dtype=torch.float32).to(torch.bfloat16).float()
    w3 = rnd(c, c, 3, 1, 1, seed=30, scale=0.03, dtype=torch.float32).to(torch.bfloat16).float()
    rest = _f16(rnd(b * t * hw, c, seed=31, dtype=torch.float32))
    reft = F.conv3d(xt, w3, None, padding=(1, 0, 0))[..., 0].permute(0, 2, 3, 1).reshape(-1, c) + rest.float()
    rowst = xt[..., 0].permute(0, 2, 3, 1).reshape(-1, c).to(torch.bfloat16).contiguous()
    outt = ops.gemm(rowst, pack_tconv3(w3), k=c, taps=3, gather=ops.GATHER_TCONV3, tconv=(t, hw), residual=rest, out_dtype=torch.float16)
    assert_close(outt, reft, 1.5e-3, "tconv3 fp16 stream")


@pytest.mark.parametrize("rows,C,instances", [(32768, 320, 32), (32768, 320, 2), (2048, 1280, 32), (512, 1280, 2), (8192, 640, 32)])
def test_groupnorm_fp16_rows(ops, rows, C, instances):
    """GroupNorm(32)+SiLU on fp16 rows (chunked two-launch path and the single-launch one) against torch fp32 on the same values."""
    x = _f16(rnd(rows, C, seed=32, dtype=torch.float32) * 1.7 + 0.3)
    gamma, beta = rnd(C, seed=33, dtype=torch.float32), rnd(C, seed=34, dtype=torch.float32)
    y = ops.groupnorm(x, gamma, beta, instances=instances, eps=1e-5, silu=True)
    xr = x.float().reshape(instances, rows // instances, C).permute(0, 2, 1)
    ref = F.silu(F.group_norm(xr, 32, gamma, beta, 1e-5)).permute(0, 2, 1).reshape(rows, C)
    assert_close(y, ref, 1e-2, "groupnorm fp16 rows")
    # the fp32 path on the same values (the single-launch / chunked choice depends on the element size, i.e. the summation order may differ)
    assert_close(y, ops.groupnorm(x.float(), gamma, beta, instances=instances, eps=1e-5, silu=True), 8e-3, "fp16 rows vs the same values as fp32 rows")


@pytest.mark.parametrize("kind", ["f16", "bf16"])
def test_groupnorm_rows_off_a_16_byte_boundary(ops, kind):
    """The two-byte GroupNorm apply moves 16 bytes per lane (gn_apply8); rows that start 8 bytes off a 16-byte boundary (a view into a
    larger buffer) must take the 8-byte kernel and give the very same bits."""
    rows, C, inst = 4096, 320, 4          # chunked two-launch path
    gamma, beta = rnd(C, seed=44, dtype=torch.float32), rnd(C, seed=45, dtype=torch.float32)
    vals = rnd(rows, C, seed=43, dtype=torch.float32) * 1.5 + 0.2
    x = _f16(vals) if kind == "f16" else vals.to(ops.BF16)
    buf = torch.empty(rows * C + 8, dtype=x.dtype, device=x.device)
    off = buf[4:4 + rows * C].view(rows, C)      # 8 bytes past the allocation's (256-byte aligned) start
    off.copy_(x)
    assert off.data_ptr() % 16 == 8 and x.data_ptr() % 16 == 0
    y = ops.groupnorm(x, gamma, beta, instances=inst, eps=1e-5, silu=True)
    y_off = ops.groupnorm(off, gamma, beta, instances=inst, eps=1e-5, silu=True)
    assert torch.equal(y, y_off)
    xr = x.float().reshape(inst, rows // inst, C).permute(0, 2, 1)
    ref = F.silu(F.group_norm(xr, 32, gamma, beta, 1e-5)).permute(0, 2, 1).reshape(rows, C)
    assert_close(y, ref, 1e-2, "groupnorm, 16-byte apply")


@pytest.mark.parametrize("rows,C", [(32768, 320), (8192, 640), (2048, 1280), (100, 320)])
def test_layernorm_fp16_rows(ops, rows, C):
    x = _f16(rnd(rows, C, seed=35, dtype=torch.float32) * 2.0 - 0.5)
    gamma, beta = rnd(C, seed=36, dtype=torch.float32), rnd(C, seed=37, dtype=torch.float32)
    add = rnd(rows // 2 if rows % 2 == 0 else rows, C, seed=38)
    y, y2 = ops.layernorm(x, gamma, beta, addend=add)
    ref = F.layer_norm(x.float(), (C,), gamma, beta, 1e-5)
    assert_close(y, ref, 1e-2, "layernorm fp16 rows")
    assert_close(y2, ref + add.float().repeat(rows // add.shape[0], 1), 1e-2, "layernorm fp16 rows + addend")
    assert torch.equal(y, ops.layernorm(x.float(), gamma, beta))


def test_concat_and_cast_fp16_rows(ops):
    a, b = _f16(rnd(4096, 640, seed=39, dtype=torch.float32)), _f16(rnd(4096, 320, seed=40, dtype=torch.float32))
    out, out16 = ops.concat_rows(a, b, with_bf16=True)
    ref = torch.cat([a, b], 1)
    assert out.dtype == torch.float16 and torch.equal(out, ref)
    assert torch.equal(out16, ref.float().to(torch.bfloat16))
    assert torch.equal(ops.cast_bf16(a), a.float().to(torch.bfloat16))


@pytest.mark.parametrize("case", ["conv_res_clip_splitk", "tconv_res_frame", "linear_res_clip", "linear_splitk_clip", "conv_bf16_splitk_frame"])
def test_groupnorm_statistics_from_stream_epilogues(ops, case):
    """Round 3: statistics from every producer flavour the UNet has -- residual added, fp16 (stream) or bf16 output, linear /
    3x3 / temporal gathers, the tile epilogue and the split-K reduce pass: the output must equal the plain call's bit for bit, the
    slots must add up to the sums of the stored values, and the norm on them must match the norm that makes its own pass."""
    from camc2v_amd import pack
    g = torch.Generator().manual_seed(70)
    if case == "conv_res_clip_splitk":          # ResBlock conv2 + skip at 16x16 latents -> temporal block's first norm (clip-wise)
        b, t, side, cin, cout = 2, 16, 16, 640, 640
        rows, inst_rows = b * t * side * side, t * side * side
        x = rnd(rows, cin, seed=71)
        w = pack.pack_conv3x3((torch.randn(cout, cin, 3, 3, generator=g) * 0.02).to(dev()))
        kw = dict(k=cin, taps=9, gather=ops.GATHER_CONV3X3, conv=(side, side, side, side, 1, 0), out_dtype=torch.float16,
                  residual=rnd(rows, cout, seed=72, dtype=torch.float32).to(torch.float16))
    elif case == "tconv_res_frame":             # last temporal convolution + block input at 32x32 latents -> SpatialTransformer norm (frame-wise)
        b, t, side, cin, cout = 2, 16, 32, 320, 320
        rows, inst_rows = b * t * side * side, side * side
        x = rnd(rows, cin, seed=71)
        w = pack.pack_tconv3((torch.randn(cout, cin, 3, 1, 1, generator=g) * 0.03).to(dev()))
        kw = dict(k=cin, taps=3, gather=ops.GATHER_TCONV3, tconv=(t, side * side), out_dtype=torch.float16,
                  residual=rnd(rows, cout, seed=72, dtype=torch.float32).to(torch.float16))
    elif case == "linear_res_clip":             # SpatialTransformer proj_out + input at 16x16 latents -> TemporalTransformer norm (clip-wise)
        rows, inst_rows, cin, cout = 8192, 4096, 640, 640
        x = rnd(rows, cin, seed=71)
        w = rnd(cout, cin, seed=73, scale=0.04)
        kw = dict(out_dtype=torch.float16, residual=rnd(rows, cout, seed=72, dtype=torch.float32).to(torch.float16))
    elif case == "linear_splitk_clip":          # a long-K projection whose plan splits K, 8x8 latents, clip-wise consumer
        rows, inst_rows, cin, cout = 2048, 1024, 5120, 1280
        x = rnd(rows, cin, seed=71)
        w = rnd(cout, cin, seed=73, scale=0.02)
        kw = dict(out_dtype=torch.float16, residual=rnd(rows, cout, seed=72, dtype=torch.float32).to(torch.float16))
    else:                                       # ResBlock conv1 (+ per-clip embedding) at 16x16 latents, split-K -> second norm (frame-wise), bf16
        b, t, side, cin, cout = 2, 16, 16, 1280, 640
        rows, inst_rows = b * t * side * side, side * side
        x = rnd(rows, cin, seed=71)
        w = pack.pack_conv3x3((torch.randn(cout, cin, 3, 3, generator=g) * 0.02).to(dev()))
        emb = rnd(2, cout, seed=74, dtype=torch.float32) * 0.2
        kw = dict(k=cin, taps=9, gather=ops.GATHER_CONV3X3, conv=(side, side, side, side, 1, 0), bias2=emb, ldb2=emb.stride(0), rows_per_batch=rows // 2)
    bias = rnd(cout, seed=75, dtype=torch.float32) * 0.1
    gamma, beta = 1.0 + rnd(cout, seed=76, dtype=torch.float32) * 0.1, rnd(cout, seed=77, dtype=torch.float32) * 0.1
    inst = rows // inst_rows
    ops.TRACK_GEMM_PLAN = True
    try:
        h, st = ops.gemm(x, w, bias=bias, gn_rows=inst_rows, **kw)
        plan = ops.LAST_GEMM_PLAN
    finally:
        ops.TRACK_GEMM_PLAN = False
    assert st is not None, f"{case}: no epilogue statistics (plan {plan})"
    if "splitk" in case:
        assert plan[1] > 1, f"{case}: expected a split-K plan, got {plan}"
    h0 = ops.gemm(x, w, bias=bias, **kw)
    assert torch.equal(h, h0), f"{case}: the statistics-emitting epilogue stores other values"
    part, gn_rows = st
    hv = h.float().reshape(inst, inst_rows, 32, cout // 32)
    want = torch.stack([hv.sum((1, 3)), (hv * hv).sum((1, 3))], -1).reshape(inst, 64)
    assert_close(part.sum(1), want, 3e-4, f"{case}: epilogue statistics")
    y0 = ops.groupnorm(h0, gamma, beta, instances=inst, eps=1e-5, silu=True)
    y = ops.groupnorm(ops.tag_stats(h, st), gamma, beta, instances=inst, eps=1e-5, silu=True)     # picked up from the tag
    d = (y.float() - y0.float()).abs()
    assert d.max().item() <= 2.0 ** -6 * max(1.0, y0.float().abs().max().item()), d.max().item()
    assert (d > 0).float().mean().item() < 0.02


@pytest.mark.parametrize("N,geglu", [(960, False), (320, False), (2560, True)])
def test_gemm_with_layernorm_prologue(ops, N, geglu):
    """LayerNorm folded into the A-stationary kernel's prologue (K = 320, the fp16 stream as the operand): against LayerNorm -> bf16 ->
    GEMM through the separate kernels (same arithmetic up to the order of the row reductions) and against torch fp32."""
    from camc2v_amd.pack import interleave_geglu
    M, K = 32768, 320
    x = (rnd(M, K, seed=80, dtype=torch.float32) * 1.5 + 0.4).to(torch.float16)
    gamma, beta = 1.0 + rnd(K, seed=81, dtype=torch.float32) * 0.2, rnd(K, seed=82, dtype=torch.float32) * 0.1
    w = rnd(N, K, seed=83, scale=0.05)
    bias = rnd(N, seed=84, dtype=torch.float32) if geglu else None
    kw = {}
    if geglu:
        w, bias = interleave_geglu(w, bias)
        kw = dict(bias=bias, geglu=True)
    ops.TRACK_GEMM_PLAN = True
    try:
        got = ops.gemm(ops.LazyLN(x, gamma, beta, 1e-5), w, **kw)
        plan = ops.LAST_GEMM_PLAN
    finally:
        ops.TRACK_GEMM_PLAN = False
    assert plan[0] == -4, f"expected the A-stationary kernel, plan {plan}"
    n = ops.layernorm(x, gamma, beta, eps=1e-5)
    sep = ops.gemm(n, w, **kw)
    d = (got.float() - sep.float()).abs()
    assert d.max().item() <= 2.0 ** -5 * max(1.0, sep.float().abs().max().item()), d.max().item()     # bf16 rounding steps of n and of the output
    assert (d > 0).float().mean().item() < 0.2
    nf = F.layer_norm(x.float(), (K,), gamma, beta, 1e-5).to(torch.bfloat16).float()
    # reference with the same operand rounding (bf16 LayerNorm output), fp32 accumulate
    wf = w.float()
    full = nf @ wf.t() + (bias if bias is not None else 0.0)
    if geglu:      # interleaved 16-row value / gate blocks
        blocks = full.reshape(M, N // 32, 2, 16)
        full = (blocks[:, :, 0] * F.gelu(blocks[:, :, 1])).reshape(M, N // 2)
    assert_close(got, full, 1.5e-2, "LayerNorm prologue vs torch")
    # a shape without the prologue falls back to LayerNorm + GEMM transparently
    x2 = x[:4096]
    fb = ops.gemm(ops.LazyLN(x2, gamma, beta, 1e-5), w, **kw)
    assert torch.equal(fb, ops.gemm(ops.layernorm(x2, gamma, beta, eps=1e-5), w, **kw))
