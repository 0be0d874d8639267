"""Checkpoint tensor -> kernel operand layouts (run once per load, not on the step path).

All functions take fp32 tensors in the reference's checkpoint layout and return bf16
tensors in the layout ccv_gemm expects: W [N, taps*K], k index = tap*K + channel.
"""
import torch

from .lib import OPERANDS

BF16 = torch.bfloat16 if OPERANDS == "bf16" else torch.float16      # the MFMA operand type (lib.OPERANDS)


def _pad_dim(t, dim, mult):
    n = t.shape[dim]
    pad = (-n) % mult
    if pad == 0:
        return t
    shape = list(t.shape)
    shape[dim] = pad
    return torch.cat([t, t.new_zeros(shape)], dim=dim)


def pack_linear(w, k_mult=64, n_mult=16):
    """nn.Linear / 1x1 conv weight [N, K(,1,1)] -> bf16 [N', K'] zero padded."""
    w = w.reshape(w.shape[0], -1)
    return _pad_dim(_pad_dim(w, 1, k_mult), 0, n_mult).to(BF16).contiguous()


def pack_conv3x3(w, k_mult=64, n_mult=16):
    """nn.Conv2d weight [Cout, Cin, 3, 3] -> bf16 [Cout', 9*Cin'] with tap = ky*3+kx major."""
    w = _pad_dim(_pad_dim(w, 1, k_mult), 0, n_mult)
    co, ci = w.shape[:2]
    return w.permute(0, 2, 3, 1).reshape(co, 9 * ci).to(BF16).contiguous()


def pack_tconv3(w, k_mult=64, n_mult=16):
    """nn.Conv3d weight [Cout, Cin, 3, 1, 1] -> bf16 [Cout', 3*Cin'] with tap = kt major."""
    w = _pad_dim(_pad_dim(w[..., 0, 0], 1, k_mult), 0, n_mult)  # [Cout, Cin, 3]
    co, ci = w.shape[:2]
    return w.permute(0, 2, 1).reshape(co, 3 * ci).to(BF16).contiguous()


def pad_bias(b, n_mult=16):
    return _pad_dim(b.float(), 0, n_mult).contiguous()


def interleave_geglu(w, b):
    """GEGLU proj [2*inner, K] (rows [0,inner) = value, [inner,2*inner) = gate, lvdm/modules/attention.py:436-438)
    -> rows interleaved in 16-row blocks (value block, gate block, ...) so that ccv_gemm's geglu epilogue finds
    value and gate of one output column in the same lane.  Bias likewise."""
    inner = w.shape[0] // 2
    assert inner % 16 == 0
    wv, wg = w[:inner].reshape(inner // 16, 16, -1), w[inner:].reshape(inner // 16, 16, -1)
    wp = torch.stack([wv, wg], dim=1).reshape(2 * inner, -1)
    bv, bg = b[:inner].reshape(inner // 16, 16), b[inner:].reshape(inner // 16, 16)
    bp = torch.stack([bv, bg], dim=1).reshape(2 * inner)
    return wp.to(BF16).contiguous(), bp.float().contiguous()


def permute_k16_for_acc_operand(w):
    """[N, K] (K % 16 == 0) -> bf16 [N, K] with the columns of every 16-column group in the order 0-3, 8-11, 4-7, 12-15: the k
    order in which a 32x32 MFMA accumulator hands its rows to the next MFMA as an operand (csrc/ccv_fused.hip): the weight of the
    product that consumes an in-register intermediate is stored this way, so the intermediate needs no lane movement."""
    n, k = w.shape
    assert k % 16 == 0
    return w.reshape(n, k // 16, 4, 4)[:, :, [0, 2, 1, 3], :].reshape(n, k).to(BF16).contiguous()
