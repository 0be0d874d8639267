"""Parity of the one-launch transformer sub-blocks (csrc/ccv_fused.hip) on a real MI355X, through the C ABI.

Checkers: (1) plain torch fp32 math in the REFERENCE's formulation (lvdm/modules/attention.py:248-253, 431-458: nn.LayerNorm ->
GEGLU -> Linear -> + x) on bf16-representable weights, with the two roundings the numerical contract of the HIP path states
(bf16 MFMA operands: the normalised rows and the gated hidden units); (2) the three-launch path (ccv_layernorm + two ccv_gemm) the
kernel replaces.  Tolerances: fp16 / bf16 outputs carry a 2^-11 / 2^-9 relative rounding of values of the stream's magnitude; the
fp32 accumulation order differs between all three computations.
"""
import math

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


def rnd(*shape, seed=0, scale=1.0, dtype=torch.float32):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(*shape, generator=g) * scale).to(dtype).to(dev())


def rel_l2(got, ref):
    got, ref = got.double().cpu(), ref.double().cpu()
    return ((got - ref).norm() / ref.norm()).item()


def ff_case(M, C=320, seed=0):
    x = rnd(M, C, seed=seed, scale=2.0).to(torch.float16)
    gamma = 1.0 + 0.2 * rnd(C, seed=seed + 1)
    beta = 0.1 * rnd(C, seed=seed + 2)
    w1 = rnd(8 * C, C, seed=seed + 3, scale=0.05).to(torch.bfloat16).float()      # rows [0, 4C) value, [4C, 8C) gate
    b1 = 0.3 * rnd(8 * C, seed=seed + 4)
    w2 = rnd(C, 4 * C, seed=seed + 5, scale=0.03).to(torch.bfloat16).float()
    b2 = 0.2 * rnd(C, seed=seed + 6)
    return x, gamma, beta, w1, b1, w2, b2


def ff_reference(x, gamma, beta, w1, b1, w2, b2, eps=1e-5):
    """attention.py:253 with the HIP path's operand roundings (n and the gated hidden units go through bf16)."""
    n = F.layer_norm(x.float(), (x.shape[1],), gamma, beta, eps).to(torch.bfloat16).float()
    val, gate = (n @ w1.t() + b1).chunk(2, dim=-1)
    hid = (val * F.gelu(gate)).to(torch.bfloat16).float()
    return x.float() + hid @ w2.t() + b2


@pytest.mark.parametrize("M", [128, 384, 32768])
@pytest.mark.parametrize("out_dtype", [torch.float16, torch.bfloat16])
def test_ff_fused_vs_reference_formulation(ops, M, out_dtype):
    from camc2v_amd import pack
    x, gamma, beta, w1, b1, w2, b2 = ff_case(M, seed=100 + M % 7)
    w1p, b1p = pack.interleave_geglu(w1, b1)
    w2p = pack.permute_k16_for_acc_operand(w2)
    assert ops.ff_fusable(x, w1p, w2p)
    got = ops.ff_fused(x, gamma, beta, 1e-5, w1p, b1p, w2p, b2, out_dtype=out_dtype)
    ref = ff_reference(x, gamma, beta, w1, b1, w2, b2)
    assert got.dtype == out_dtype and tuple(got.shape) == (M, 320)
    err = (got.float().cpu() - ref.cpu()).abs().max().item()
    scale = ref.abs().max().item()
    r = rel_l2(got, ref)
    # bf16 hidden units: a unit that sits on a rounding boundary may flip between the two fp32 summation orders (2^-9 of its value,
    # times |w2| <= 0.15): far below the output's own rounding
    tol_r = 6e-4 if out_dtype == torch.float16 else 2.5e-3
    assert math.isfinite(err) and r <= tol_r and err <= 8 * tol_r * scale, f"M={M} {out_dtype}: rel-L2 {r:.3e}, max err {err:.3e} of {scale:.3e}"


def test_ff_fused_in_place_equals_three_launches(ops):
    """The stream update in place (out = x) against ccv_layernorm + GEGLU GEMM + down GEMM with the residual epilogue."""
    from camc2v_amd import pack
    M = 4096
    x, gamma, beta, w1, b1, w2, b2 = ff_case(M, seed=7)
    w1p, b1p = pack.interleave_geglu(w1, b1)
    w2p, w2l = pack.permute_k16_for_acc_operand(w2), pack.pack_linear(w2)
    n = ops.layernorm(x, gamma, beta, eps=1e-5)
    hidden = ops.gemm(n, w1p, bias=b1p, geglu=True)
    three = ops.gemm(hidden, w2l, bias=b2, residual=x, out_dtype=torch.float16)
    stream = x.clone()
    out = ops.ff_fused(stream, gamma, beta, 1e-5, w1p, b1p, w2p, b2, out=stream)
    assert out is stream
    r = rel_l2(stream, three)
    err = (stream.float() - three.float()).abs().max().item()
    # same roundings, different fp32 summation orders (and LayerNorm statistics order): fp16 ulps of the stream
    assert r <= 6e-4 and err <= 2e-2 * three.float().abs().max().item(), f"rel-L2 {r:.3e}, max {err:.3e}"
    # rows are independent: a tile computed alone equals the same rows of the big launch, bit for bit
    part = ops.ff_fused(x[1024:1152].contiguous(), gamma, beta, 1e-5, w1p, b1p, w2p, b2)
    assert torch.equal(part, stream[1024:1152])


def test_ff_fused_refuses_what_it_was_not_built_for(ops):
    from camc2v_amd import pack
    from camc2v_amd.lib import CcvError
    x, gamma, beta, w1, b1, w2, b2 = ff_case(256)
    w1p, b1p = pack.interleave_geglu(w1, b1)
    w2p = pack.permute_k16_for_acc_operand(w2)
    assert not ops.ff_fusable(x[:200].contiguous(), w1p, w2p)            # rows not a multiple of 128
    assert not ops.ff_fusable(x.float(), w1p, w2p)                       # fp32 stream
    with pytest.raises(CcvError):
        ops.ff_fused(x[:200].contiguous(), gamma, beta, 1e-5, w1p, b1p, w2p, b2)
    x640 = rnd(256, 640).to(torch.float16)
    assert not ops.ff_fusable(x640, rnd(5120, 640).to(torch.bfloat16), rnd(640, 2560).to(torch.bfloat16))


def test_feedforward_module_takes_the_fused_path_and_matches(ops, monkeypatch):
    """unet.FeedForward.run on the fp16 stream of a C = 320 block: the one-launch form against the three-launch form (CCV_FUSE_FF)."""
    from camc2v_amd import unet
    ff = unet.FeedForward(320, glu=True).to(dev())
    for i, prm in enumerate(ff.parameters()):
        prm.data.copy_(rnd(*prm.shape, seed=50 + i, scale=0.05))
    gamma, beta = 1.0 + 0.1 * rnd(320, seed=60), 0.1 * rnd(320, seed=61)
    x = rnd(2048, 320, seed=62, scale=1.5).to(torch.float16)
    calls = []
    real = ops.ff_fused
    monkeypatch.setattr(ops, "ff_fused", lambda *a, **k: (calls.append(1), real(*a, **k))[1])
    s1 = x.clone()
    ff.run(ops.LazyLN(s1, gamma, beta, 1e-5), s1)
    fin = ff.run(ops.LazyLN(x, gamma, beta, 1e-5), x, final=True)
    assert len(calls) == 2 and fin.dtype == torch.bfloat16
    monkeypatch.setattr(unet, "FUSE_FF", False)
    s2 = x.clone()
    ff.run(ops.LazyLN(s2, gamma, beta, 1e-5), s2)
    fin2 = ff.run(ops.LazyLN(x, gamma, beta, 1e-5), x, final=True)
    assert len(calls) == 2
    assert rel_l2(s1, s2) <= 6e-4 and rel_l2(fin, fin2) <= 2.5e-3
