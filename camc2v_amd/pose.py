"""Once-per-clip camera feeders on the HIP kernels: ``ray_condition`` and
``CameraPoseEncoder`` -- relative poses -> Pluecker embedding -> the four pose feature maps the UNet's temporal blocks
add to their LayerNorm output (``camera_condition["pluker_embedding_features"]``).

Same class, attribute and ``state_dict`` names as the reference (model/base.py:112-174,
model/modules/camera_pose_encoder.py:15-376).  The reference builds the temporal blocks from the ``diffusers`` package
(``Attention`` with AttnProcessor2_0, ``FeedForward(activation_fn="geglu")``), which is absent from the build
container: the parameter containers below reproduce the state_dict layout those classes are documented to have
(``to_q/to_k/to_v`` without bias, ``to_out.0`` with bias, ``ff.net.0.proj``, ``ff.net.2``), and the arithmetic follows
oracle/pose_oracle.py; **parity of the encoder is unpinned** (see that file's header), ``ray_condition`` is pinned.

Only the shipped block configuration is built: ksize 1, sk True, use_conv False (average-pool downsampling), one
"Temporal_Self" attention per block, sinusoidal frame encoding.
"""
import math

import torch
import torch.nn as nn

from . import ops, pack
from .lib import CcvError
from .unet import _Prepared, _dev_f32


def ray_condition(K, c2w, H, W, device=None, flip_flag=None, camera_embedding="plucker"):
    """K [B,V,3,3], c2w [B,V,4,4] (relative to the conditioning frame) -> fp32 [B, 6, V, H, W] (model/base.py:112-174).
    The cross product is taken over the xyz axis (the reference's ``torch.cross`` without ``dim`` picks the first axis of
    size 3, which is the same axis unless B or V equals 3)."""
    if flip_flag is not None and bool(torch.as_tensor(flip_flag).any()):
        raise NotImplementedError("flip augmentation is a training-side option")
    return ops.ray_condition(K, c2w, H, W, plucker=(camera_embedding == "plucker"))


class Downsample(nn.Module):
    def __init__(self, channels, use_conv, dims=2, out_channels=None, padding=1):
        super().__init__()
        if use_conv or dims != 2:
            raise NotImplementedError("the shipped pose encoder downsamples with AvgPool2d (use_conv False)")
        self.channels, self.out_channels, self.use_conv, self.dims = channels, out_channels or channels, use_conv, dims
        self.op = nn.AvgPool2d(kernel_size=2, stride=2)


class ResnetBlock(nn.Module, _Prepared):
    """[avg-pool] -> [1x1 in_conv] -> conv3x3 -> ReLU -> 1x1 conv -> + input (camera_pose_encoder.py:257-290)."""

    def __init__(self, in_c, out_c, down, ksize=3, sk=False, use_conv=True):
        super().__init__()
        if ksize != 1 or not sk:
            raise NotImplementedError("the shipped pose encoder uses ksize 1 and sk True")
        self.in_c, self.out_c = in_c, out_c
        self.in_conv = nn.Conv2d(in_c, out_c, ksize, 1, 0) if in_c != out_c else None
        self.block1 = nn.Conv2d(out_c, out_c, 3, 1, 1)
        self.act = nn.ReLU()
        self.block2 = nn.Conv2d(out_c, out_c, ksize, 1, 0)
        self.skep = None
        self.down = down
        if down:
            self.down_opt = Downsample(in_c, use_conv=use_conv)

    def _pack(self):
        pk = dict(w1=pack.pack_conv3x3(self.block1.weight), b1=_dev_f32(self.block1.bias),
                  w2=pack.pack_linear(self.block2.weight), b2=_dev_f32(self.block2.bias))
        if self.in_conv is not None:
            pk["wi"], pk["bi"] = pack.pack_linear(self.in_conv.weight), _dev_f32(self.in_conv.bias)
        return pk

    def forward_rows(self, x, n, h, w):
        """x fp32 [(n h w), in_c] -> (fp32 [(n h' w'), out_c], h', w')."""
        pk = self._pk()
        if self.down:
            x = ops.avgpool2_rows(x, n, h, w)
            h, w = h // 2, w // 2
        if "wi" in pk:
            x = ops.gemm(x, pk["wi"], bias=pk["bi"], out_f32=True)
        t = ops.gemm(ops.cast_bf16(x), pk["w1"], k=self.out_c, taps=9, bias=pk["b1"], act=ops.ACT_RELU, gather=ops.GATHER_CONV3X3,
                     conv=(h, w, h, w, 1, 0))
        return ops.gemm(t, pk["w2"], bias=pk["b2"], residual=x, out_f32=True), h, w


class PositionalEncoding(nn.Module):
    def __init__(self, d_model, dropout=0.0, max_len=32):
        super().__init__()
        self.dropout = nn.Dropout(p=dropout)
        position = torch.arange(max_len).unsqueeze(1)
        div_term = torch.exp(torch.arange(0, d_model, 2) * (-math.log(10000.0) / d_model))
        pe = torch.zeros(1, max_len, d_model)
        pe[0, :, 0::2] = torch.sin(position * div_term)
        pe[0, :, 1::2] = torch.cos(position * div_term)
        self.register_buffer("pe", pe)


class TemporalSelfAttention(nn.Module):
    """Parameter container with the layout of diffusers' ``Attention`` (query_dim = inner_dim, no q/k/v bias, output
    projection with bias) + the frame encoding (camera_pose_encoder.py:103-158)."""

    def __init__(self, query_dim, heads, dim_head, temporal_position_encoding=False, temporal_position_encoding_max_len=32,
                 rescale_output_factor=1.0):
        super().__init__()
        if rescale_output_factor != 1.0:
            raise NotImplementedError("rescale_output_factor != 1 is not used by the shipped config")
        inner = heads * dim_head
        self.heads, self.dim_head = heads, dim_head
        self.to_q = nn.Linear(query_dim, inner, bias=False)
        self.to_k = nn.Linear(query_dim, inner, bias=False)
        self.to_v = nn.Linear(query_dim, inner, bias=False)
        self.to_out = nn.ModuleList([nn.Linear(inner, query_dim, bias=True), nn.Dropout(0.0)])
        self.pos_encoder = PositionalEncoding(query_dim, max_len=temporal_position_encoding_max_len) if temporal_position_encoding else None


class _GEGLU(nn.Module):
    def __init__(self, dim_in, dim_out):
        super().__init__()
        self.proj = nn.Linear(dim_in, dim_out * 2)


class _DiffusersFeedForward(nn.Module):
    """Parameter container with the layout of diffusers' ``FeedForward(dim, activation_fn="geglu")``: net.0 = GEGLU
    (proj: dim -> 8 dim), net.1 = Dropout, net.2 = Linear(4 dim -> dim)."""

    def __init__(self, dim, mult=4):
        super().__init__()
        self.net = nn.ModuleList([_GEGLU(dim, dim * mult), nn.Dropout(0.0), nn.Linear(dim * mult, dim)])


class TemporalTransformerBlock(nn.Module, _Prepared):
    """LayerNorm (+ frame encoding) -> self-attention over the frames of a pixel -> + x -> LayerNorm -> GEGLU feed-forward
    -> + x (camera_pose_encoder.py:15-78), on frame-major token rows [(b f h w), C]."""

    def __init__(self, dim, num_attention_heads, attention_head_dim, attention_block_types=("Temporal_Self",), dropout=0.0,
                 cross_attention_dim=None, temporal_position_encoding=False, temporal_position_encoding_max_len=32,
                 rescale_output_factor=1.0, **ignored):
        super().__init__()
        if tuple(attention_block_types) != ("Temporal_Self",):
            raise NotImplementedError("the shipped pose encoder has one Temporal_Self attention per block")
        self.dim, self.heads, self.dim_head = dim, num_attention_heads, attention_head_dim
        self.attention_block_types = tuple(attention_block_types)
        self.attention_blocks = nn.ModuleList([TemporalSelfAttention(dim, num_attention_heads, attention_head_dim,
                                                                     temporal_position_encoding, temporal_position_encoding_max_len,
                                                                     rescale_output_factor)])
        self.norms = nn.ModuleList([nn.LayerNorm(dim)])
        self.ff = _DiffusersFeedForward(dim)
        self.ff_norm = nn.LayerNorm(dim)

    def _pack(self):
        a = self.attention_blocks[0]
        if self.dim_head % 8 or self.dim_head > 256 or self.dim % 64:
            raise CcvError("pose encoder: head width must be a multiple of 8 (<= 256) and the block width a multiple of 64")
        w1, b1 = pack.interleave_geglu(self.ff.net[0].proj.weight, self.ff.net[0].proj.bias)
        pk = dict(g1=_dev_f32(self.norms[0].weight), b1n=_dev_f32(self.norms[0].bias),
                  w_qkv=pack.pack_linear(torch.cat([a.to_q.weight, a.to_k.weight, a.to_v.weight], 0)),
                  w_o=pack.pack_linear(a.to_out[0].weight), b_o=_dev_f32(a.to_out[0].bias),
                  g2=_dev_f32(self.ff_norm.weight), b2n=_dev_f32(self.ff_norm.bias),
                  w1=w1, b1=b1, w2=pack.pack_linear(self.ff.net[2].weight), b2=_dev_f32(self.ff.net[2].bias))
        pk["pe"] = a.pos_encoder.pe[0].detach().float() if a.pos_encoder is not None else None
        pk["pe_rows"] = {}
        return pk

    def forward_rows(self, x, b, f, hw):
        """x fp32 [(b f hw), C], updated in place and returned."""
        pk = self._pk()
        C, H, D = self.dim, self.heads, self.dim_head
        if pk["pe"] is not None:
            if f > pk["pe"].shape[0]:
                raise CcvError(f"pose encoder: {f} frames exceed temporal_position_encoding_max_len {pk['pe'].shape[0]}")
            tab = pk["pe_rows"].get((f, hw))
            if tab is None:      # frame encoding as an addend table over one clip's rows: row (frame, pixel) -> pe[frame]
                tab = pk["pe"][:f].to(ops.BF16).repeat_interleave(hw, 0).contiguous()
                pk["pe_rows"] = {(f, hw): tab}
            _, n = ops.layernorm(x, pk["g1"], pk["b1n"], addend=tab)
        else:
            n = ops.layernorm(x, pk["g1"], pk["b1n"])
        qkv = ops.gemm(n, pk["w_qkv"])
        ld = 3 * C
        st = (f * hw * ld, ld, hw * ld)                       # (clip, pixel, frame) strides
        o = torch.empty((b * f * hw, C), dtype=ops.BF16, device=x.device)
        ops.attention_small(qkv, qkv[:, C:], qkv[:, 2 * C:], B=b * hw, inner=hw, H=H, T=f, head_dim=D, q_str=st, k_str=st, v_str=st,
                            out=o, o_str=(f * hw * C, C, hw * C))
        ops.gemm(o, pk["w_o"], bias=pk["b_o"], residual=x, out_f32=True, out=x)
        h = ops.gemm(ops.layernorm(x, pk["g2"], pk["b2n"]), pk["w1"], bias=pk["b1"], geglu=True)
        ops.gemm(h, pk["w2"], bias=pk["b2"], residual=x, out_f32=True, out=x)
        return x


class CameraPoseEncoder(nn.Module, _Prepared):
    def __init__(self, downscale_factor, channels=(320, 640, 1280, 1280), nums_rb=3, cin=64, ksize=3, sk=False, use_conv=True,
                 compression_factor=1, temporal_attention_nhead=8, attention_block_types=("Temporal_Self",),
                 temporal_position_encoding=False, temporal_position_encoding_max_len=16, rescale_output_factor=1.0):
        super().__init__()
        if compression_factor != 1:
            raise NotImplementedError("compression_factor != 1 is not used by the shipped config")
        self.downscale_factor, self.channels, self.nums_rb, self.cin = downscale_factor, list(channels), nums_rb, cin
        self.unshuffle = nn.PixelUnshuffle(downscale_factor)
        self.encoder_down_conv_blocks = nn.ModuleList()
        self.encoder_down_attention_blocks = nn.ModuleList()
        for i, c in enumerate(self.channels):
            convs, attns = nn.ModuleList(), nn.ModuleList()
            for j in range(nums_rb):
                in_dim = self.channels[i - 1] if (j == 0 and i != 0) else c
                convs.append(ResnetBlock(in_dim, c, down=(j == 0 and i != 0), ksize=ksize, sk=sk, use_conv=use_conv))
                attns.append(TemporalTransformerBlock(dim=c, num_attention_heads=temporal_attention_nhead,
                                                      attention_head_dim=c // temporal_attention_nhead,
                                                      attention_block_types=attention_block_types,
                                                      temporal_position_encoding=temporal_position_encoding,
                                                      temporal_position_encoding_max_len=temporal_position_encoding_max_len,
                                                      rescale_output_factor=rescale_output_factor))
            self.encoder_down_conv_blocks.append(convs)
            self.encoder_down_attention_blocks.append(attns)
        self.encoder_conv_in = nn.Conv2d(cin, self.channels[0], 3, 1, 1)
        self._register_load_state_dict_pre_hook(lambda *a, **k: self.invalidate_all())

    def invalidate_all(self):
        for m in self.modules():
            if isinstance(m, _Prepared):
                m.invalidate()

    def _apply(self, fn, *args, **kwargs):
        out = super()._apply(fn, *args, **kwargs)
        self.invalidate_all()
        return out

    def _pack(self):
        if self.cin % 64:
            raise CcvError("pose encoder: cin must be a multiple of 64")
        return dict(w_in=pack.pack_conv3x3(self.encoder_conv_in.weight), b_in=_dev_f32(self.encoder_conv_in.bias))

    @torch.no_grad()
    def forward(self, x):
        """x [b, 6, f, H, W] ray / Pluecker embedding -> list of 4 fp32 feature maps [(b f), C_i, H/8/2^i, W/8/2^i]."""
        if not x.is_cuda:
            raise CcvError("CameraPoseEncoder.forward: the product path runs on the GPU only (see oracle/pose_oracle.py)")
        pk = self._pk()
        b, c, f, H, W = x.shape
        r = self.downscale_factor
        if c * r * r != self.cin:
            raise CcvError(f"pose encoder: {c} channels x {r}^2 != cin {self.cin}")
        n, h, w = b * f, H // r, W // r
        rows = ops.pixel_unshuffle_rows(x.float().permute(0, 2, 1, 3, 4).reshape(n, c, H, W), r)
        y = ops.gemm(rows, pk["w_in"], k=self.cin, taps=9, bias=pk["b_in"], out_f32=True, gather=ops.GATHER_CONV3X3, conv=(h, w, h, w, 1, 0))
        feats = []
        for convs, attns, ch in zip(self.encoder_down_conv_blocks, self.encoder_down_attention_blocks, self.channels):
            for conv, attn in zip(convs, attns):
                y, h, w = conv.forward_rows(y, n, h, w)
                y = attn.forward_rows(y, b, f, h * w)
            feats.append(ops.unpack_rows_to_nchw(y, ch, n, 1, h, w).reshape(n, ch, h, w))
        return feats


__all__ = ["CameraPoseEncoder", "TemporalTransformerBlock", "TemporalSelfAttention", "PositionalEncoding", "ResnetBlock",
           "Downsample", "ray_condition", "CcvError"]
