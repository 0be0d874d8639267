"""First-stage decoder on the HIP kernels: ``AutoencoderKL.decode`` (SURVEY.md section 8, row f2 -- the step right
after the DDIM path, turning the sampler's latents into frames).

Same class, attribute and ``state_dict`` names as the reference (lvdm/models/autoencoder.py:13-199,
lvdm/modules/networks/ae_modules.py:150-212, 24-82, 117-132, 364-583), so a ``first_stage_model.*`` checkpoint slice
loads strictly.  ``decode`` follows the path (latents -> frames); ``encode`` runs once per clip on the conditioning and
context frames in front of it (frames -> posterior parameters -> latents).

Execution: activations stay token-major ``[(n h w), C]`` like in the UNet; GroupNorm(32)+swish is ``ccv_groupnorm``;
every 3x3 conv (incl. the nearest-2x upsample convs) is the implicit-GEMM ``ccv_gemm`` with bias / residual fused; the
single-head attention of the middle block has head dim = C (512), which the d=64 attention kernels do not cover, so it
runs as GEMM (Q K^T, alpha = C^-1/2) -> ``ccv_softmax_rows`` -> GEMM (P V) per frame, with V^T produced directly by a
GEMM whose "activation" operand is the value weight.
"""
import torch
import torch.nn as nn

from . import ops, pack, rng
from .lib import CcvError
from .unet import _Prepared, _dev_f32


def Normalize(in_channels, num_groups=32):
    return nn.GroupNorm(num_groups=num_groups, num_channels=in_channels, eps=1e-6, affine=True)


class _Geo:
    """n images of h x w pixels stored as rows [(n h w), C]."""

    def __init__(self, n, h, w):
        self.n, self.h, self.w = n, h, w

    @property
    def conv(self):
        return (self.h, self.w, self.h, self.w, 1, 0)


class ResnetBlock(nn.Module, _Prepared):
    """GN+swish+conv3x3 -> GN+swish+conv3x3 -> + (1x1 conv of) input (reference ae_modules.py:150-212; temb unused)."""

    def __init__(self, *, in_channels, out_channels=None, conv_shortcut=False, dropout=0.0, temb_channels=0):
        super().__init__()
        if conv_shortcut or temb_channels > 0:
            raise NotImplementedError("conv_shortcut / timestep embedding are not used by the first-stage model")
        out_channels = in_channels if out_channels is None else out_channels
        self.in_channels, self.out_channels, self.use_conv_shortcut = in_channels, out_channels, False
        self.norm1 = Normalize(in_channels)
        self.conv1 = nn.Conv2d(in_channels, out_channels, 3, 1, 1)
        self.norm2 = Normalize(out_channels)
        self.dropout = nn.Dropout(dropout)
        self.conv2 = nn.Conv2d(out_channels, out_channels, 3, 1, 1)
        if in_channels != out_channels:
            self.nin_shortcut = nn.Conv2d(in_channels, out_channels, 1, 1, 0)

    def _pack(self):
        pk = dict(g1=_dev_f32(self.norm1.weight), b1=_dev_f32(self.norm1.bias), w1=pack.pack_conv3x3(self.conv1.weight),
                  cb1=_dev_f32(self.conv1.bias), g2=_dev_f32(self.norm2.weight), b2=_dev_f32(self.norm2.bias),
                  w2=pack.pack_conv3x3(self.conv2.weight), cb2=_dev_f32(self.conv2.bias))
        if self.in_channels != self.out_channels:
            pk["ws"], pk["bs"] = pack.pack_linear(self.nin_shortcut.weight), _dev_f32(self.nin_shortcut.bias)
        return pk

    def forward_rows(self, x, g):
        pk = self._pk()
        h = ops.groupnorm(x, pk["g1"], pk["b1"], instances=g.n, eps=1e-6, silu=True)
        h = ops.gemm(h, pk["w1"], k=self.in_channels, taps=9, bias=pk["cb1"], gather=ops.GATHER_CONV3X3, conv=g.conv)
        h = ops.groupnorm(h, pk["g2"], pk["b2"], instances=g.n, eps=1e-6, silu=True)
        skip = ops.gemm(x, pk["ws"], bias=pk["bs"], out_f32=True) if "ws" in pk else x
        return ops.gemm(h, pk["w2"], k=self.out_channels, taps=9, bias=pk["cb2"], residual=skip, out_f32=True,
                        gather=ops.GATHER_CONV3X3, conv=g.conv)


class AttnBlock(nn.Module, _Prepared):
    """Single-head self-attention over the pixels of one image, head dim = C (reference ae_modules.py:24-82)."""

    def __init__(self, in_channels):
        super().__init__()
        self.in_channels = in_channels
        self.norm = Normalize(in_channels)
        self.q = nn.Conv2d(in_channels, in_channels, 1)
        self.k = nn.Conv2d(in_channels, in_channels, 1)
        self.v = nn.Conv2d(in_channels, in_channels, 1)
        self.proj_out = nn.Conv2d(in_channels, in_channels, 1)

    def _pack(self):
        return dict(g=_dev_f32(self.norm.weight), b=_dev_f32(self.norm.bias),
                    w_q=pack.pack_linear(self.q.weight), b_q=_dev_f32(self.q.bias),
                    w_k=pack.pack_linear(self.k.weight), b_k=_dev_f32(self.k.bias),
                    w_v=pack.pack_linear(self.v.weight), b_v=_dev_f32(self.v.bias),
                    w_o=pack.pack_linear(self.proj_out.weight), b_o=_dev_f32(self.proj_out.bias))

    def forward_rows(self, x, g):
        pk = self._pk()
        C, L = self.in_channels, g.h * g.w
        if L % 64 or C % 64:
            raise CcvError(f"AttnBlock: {g.h}x{g.w} pixels x {C} channels; the GEMM path needs multiples of 64")
        n = ops.groupnorm(x, pk["g"], pk["b"], instances=g.n, eps=1e-6, silu=False)
        q = ops.gemm(n, pk["w_q"], bias=pk["b_q"])                          # [(n L), C] bf16
        k = ops.gemm(n, pk["w_k"], bias=pk["b_k"])
        o = torch.empty((g.n * L, C), dtype=ops.BF16, device=x.device)
        for f in range(g.n):
            rows = slice(f * L, (f + 1) * L)
            # V^T [C, L] = W_v [C, C] . n_f^T: the weight is the "activation" operand, the frame's rows are the "weight"
            # operand; the value bias is added after the attention (softmax rows sum to 1)
            v_t = ops.gemm(pk["w_v"], n[rows])
            s = ops.gemm(q[rows], k[rows], out_f32=True, alpha=float(C) ** -0.5)   # [L, L] logits: K rows are the "weight"
            p = ops.softmax_rows(s)
            ops.gemm(p, v_t, bias=pk["b_v"], out=o[rows])
        return ops.gemm(o, pk["w_o"], bias=pk["b_o"], residual=x, out_f32=True)


class Upsample(nn.Module, _Prepared):
    """nearest 2x + conv3x3, fused into the conv's gather (reference ae_modules.py:117-132)."""

    def __init__(self, in_channels, with_conv):
        super().__init__()
        if not with_conv:
            raise NotImplementedError("resamp_with_conv=False is not used by the first-stage model")
        self.with_conv, self.in_channels = with_conv, in_channels
        self.conv = nn.Conv2d(in_channels, in_channels, 3, 1, 1)

    def _pack(self):
        return dict(w=pack.pack_conv3x3(self.conv.weight), b=_dev_f32(self.conv.bias))

    def forward_rows(self, x, g):
        pk = self._pk()
        g2 = _Geo(g.n, 2 * g.h, 2 * g.w)
        y = ops.gemm(ops.cast_bf16(x), pk["w"], k=self.in_channels, taps=9, m=g.n * g2.h * g2.w, bias=pk["b"], out_f32=True,
                     gather=ops.GATHER_CONV3X3, conv=(g2.h, g2.w, g.h, g.w, 1, 1))
        return y, g2


class Downsample(nn.Module, _Prepared):
    """F.pad(x, (0,1,0,1)) + stride-2 conv3x3 (reference ae_modules.py:95-115): the conv gather pads only after the
    last row / column (``no_lead_pad``)."""

    def __init__(self, in_channels, with_conv):
        super().__init__()
        if not with_conv:
            raise NotImplementedError("resamp_with_conv=False is not used by the first-stage model")
        self.with_conv, self.in_channels = with_conv, in_channels
        self.conv = nn.Conv2d(in_channels, in_channels, 3, 2, 0)

    def _pack(self):
        return dict(w=pack.pack_conv3x3(self.conv.weight), b=_dev_f32(self.conv.bias))

    def forward_rows(self, x, g):
        pk = self._pk()
        if g.h % 2 or g.w % 2:
            raise CcvError("Downsample: odd feature-map sizes are not produced by the shipped configuration")
        g2 = _Geo(g.n, g.h // 2, g.w // 2)
        y = ops.gemm(ops.cast_bf16(x), pk["w"], k=self.in_channels, taps=9, m=g.n * g2.h * g2.w, bias=pk["b"], out_f32=True,
                     gather=ops.GATHER_CONV3X3, conv=(g2.h, g2.w, g.h, g.w, 2, 0, 1))
        return y, g2


def _mid(block_in):
    mid = nn.Module()
    mid.block_1 = ResnetBlock(in_channels=block_in, out_channels=block_in)
    mid.attn_1 = AttnBlock(block_in)
    mid.block_2 = ResnetBlock(in_channels=block_in, out_channels=block_in)
    return mid


class Encoder(nn.Module, _Prepared):
    """conv_in -> levels of 2 ResnetBlocks + Downsample -> mid -> GN+swish+conv_out (reference ae_modules.py:364-468)."""

    def __init__(self, *, ch, out_ch, ch_mult=(1, 2, 4, 8), num_res_blocks, attn_resolutions, dropout=0.0,
                 resamp_with_conv=True, in_channels, resolution, z_channels, double_z=True, **ignored):
        super().__init__()
        if list(attn_resolutions):
            raise NotImplementedError("attn_resolutions is empty in the shipped first-stage config")
        self.ch, self.num_resolutions, self.num_res_blocks = ch, len(ch_mult), num_res_blocks
        self.in_channels = in_channels
        self.conv_in = nn.Conv2d(in_channels, ch, 3, 1, 1)
        in_ch_mult = (1,) + tuple(ch_mult)
        self.down = nn.ModuleList()
        block_in = ch
        for i_level in range(self.num_resolutions):
            block_in, block_out = ch * in_ch_mult[i_level], ch * ch_mult[i_level]
            down = nn.Module()
            down.block = nn.ModuleList()
            down.attn = nn.ModuleList()
            for _ in range(num_res_blocks):
                down.block.append(ResnetBlock(in_channels=block_in, out_channels=block_out))
                block_in = block_out
            if i_level != self.num_resolutions - 1:
                down.downsample = Downsample(block_in, resamp_with_conv)
            self.down.append(down)
        self.mid = _mid(block_in)
        self.norm_out = Normalize(block_in)
        self.conv_out = nn.Conv2d(block_in, 2 * z_channels if double_z else z_channels, 3, 1, 1)
        self.mid_channels, self.out_channels = block_in, self.conv_out.out_channels

    def _pack(self):
        return dict(in_pad=(self.in_channels + 63) // 64 * 64,
                    w_in=pack.pack_conv3x3(self.conv_in.weight), b_in=_dev_f32(self.conv_in.bias),
                    g=_dev_f32(self.norm_out.weight), b=_dev_f32(self.norm_out.bias),
                    w_out=pack.pack_conv3x3(self.conv_out.weight), b_out=pack.pad_bias(self.conv_out.bias))

    def forward_rows(self, x_rows, g, out=None):
        """x_rows fp32 [(n H W), in_pad] zero padded -> fp32 rows [(n H/8 W/8), 16k] (first out_channels columns valid);
        ``out`` may be a (strided) view to write into."""
        pk = self._pk()
        h = ops.gemm(x_rows, pk["w_in"], k=pk["in_pad"], taps=9, bias=pk["b_in"], out_f32=True, gather=ops.GATHER_CONV3X3, conv=g.conv)
        for i_level in range(self.num_resolutions):
            for blk in self.down[i_level].block:
                h = blk.forward_rows(h, g)
            if i_level != self.num_resolutions - 1:
                h, g = self.down[i_level].downsample.forward_rows(h, g)
        h = self.mid.block_1.forward_rows(h, g)
        h = self.mid.attn_1.forward_rows(h, g)
        h = self.mid.block_2.forward_rows(h, g)
        y = ops.groupnorm(h, pk["g"], pk["b"], instances=g.n, eps=1e-6, silu=True)
        y = ops.gemm(y, pk["w_out"], k=self.mid_channels, taps=9, bias=pk["b_out"], out_f32=True, out=out,
                     gather=ops.GATHER_CONV3X3, conv=g.conv)
        return y, g

    def forward(self, x):
        """x [n, 3, H, W] -> [n, 2*z_channels, H/8, W/8] (fp32)."""
        if not x.is_cuda:
            raise CcvError("Encoder.forward: the product path runs on the GPU only (see oracle/vae_oracle.py)")
        n, c, hh, ww = x.shape
        rows = ops.pack_nchw_to_rows(x.float().reshape(n, c, 1, hh, ww), None, ldo=self._pk()["in_pad"])
        y, g = self.forward_rows(rows, _Geo(n, hh, ww))
        return ops.unpack_rows_to_nchw(y, self.out_channels, n, 1, g.h, g.w).reshape(n, self.out_channels, g.h, g.w)


class Decoder(nn.Module, _Prepared):
    """conv_in -> mid (ResnetBlock, AttnBlock, ResnetBlock) -> levels of 3 ResnetBlocks + Upsample -> GN+swish+conv_out
    (reference ae_modules.py:471-583)."""

    def __init__(self, *, ch, out_ch, ch_mult=(1, 2, 4, 8), num_res_blocks, attn_resolutions, dropout=0.0,
                 resamp_with_conv=True, in_channels, resolution, z_channels, give_pre_end=False, tanh_out=False,
                 use_linear_attn=False, attn_type="vanilla", **ignored):
        super().__init__()
        if list(attn_resolutions) or give_pre_end or tanh_out or use_linear_attn or attn_type != "vanilla":
            raise NotImplementedError("only the shipped first-stage decoder configuration is built")
        self.ch, self.out_ch, self.z_channels = ch, out_ch, z_channels
        self.num_resolutions, self.num_res_blocks, self.resolution = len(ch_mult), num_res_blocks, resolution
        block_in = ch * ch_mult[self.num_resolutions - 1]
        curr_res = resolution // 2 ** (self.num_resolutions - 1)
        self.z_shape = (1, z_channels, curr_res, curr_res)
        self.conv_in = nn.Conv2d(z_channels, block_in, 3, 1, 1)
        self.mid = _mid(block_in)
        self.up = nn.ModuleList()
        for i_level in reversed(range(self.num_resolutions)):
            block_out = ch * ch_mult[i_level]
            up = nn.Module()
            up.block = nn.ModuleList()
            up.attn = nn.ModuleList()
            for _ in range(num_res_blocks + 1):
                up.block.append(ResnetBlock(in_channels=block_in, out_channels=block_out))
                block_in = block_out
            if i_level != 0:
                up.upsample = Upsample(block_in, resamp_with_conv)
            self.up.insert(0, up)
        self.norm_out = Normalize(block_in)
        self.conv_out = nn.Conv2d(block_in, out_ch, 3, 1, 1)

    def _pack(self):
        return dict(zc_pad=(self.z_channels + 63) // 64 * 64,
                    w_in=pack.pack_conv3x3(self.conv_in.weight), b_in=_dev_f32(self.conv_in.bias),
                    g=_dev_f32(self.norm_out.weight), b=_dev_f32(self.norm_out.bias),
                    w_out=pack.pack_conv3x3(self.conv_out.weight), b_out=pack.pad_bias(self.conv_out.bias))

    def forward_rows(self, z_rows, g):
        """z_rows fp32 [(n h w), zc_pad] zero padded -> fp32 rows [(n 8h 8w), 16] (first out_ch columns valid)."""
        pk = self._pk()
        h = ops.gemm(z_rows, pk["w_in"], k=pk["zc_pad"], taps=9, bias=pk["b_in"], out_f32=True, gather=ops.GATHER_CONV3X3,
                     conv=g.conv)
        h = self.mid.block_1.forward_rows(h, g)
        h = self.mid.attn_1.forward_rows(h, g)
        h = self.mid.block_2.forward_rows(h, g)
        for i_level in reversed(range(self.num_resolutions)):
            for blk in self.up[i_level].block:
                h = blk.forward_rows(h, g)
            if i_level != 0:
                h, g = self.up[i_level].upsample.forward_rows(h, g)
        y = ops.groupnorm(h, pk["g"], pk["b"], instances=g.n, eps=1e-6, silu=True)
        y = ops.gemm(y, pk["w_out"], k=self.up[0].block[-1].out_channels, taps=9, bias=pk["b_out"], out_f32=True,
                     gather=ops.GATHER_CONV3X3, conv=g.conv)
        return y, g

    def forward(self, z):
        """z [n, z_channels, h, w] -> [n, out_ch, 8h, 8w] (fp32)."""
        if not z.is_cuda:
            raise CcvError("Decoder.forward: the product path runs on the GPU only (see oracle/vae_oracle.py)")
        n, c, hh, ww = z.shape
        rows = ops.pack_nchw_to_rows(z.float().reshape(n, c, 1, hh, ww), None, ldo=self._pk()["zc_pad"])
        y, g = self.forward_rows(rows, _Geo(n, hh, ww))
        return ops.unpack_rows_to_nchw(y, self.out_ch, n, 1, g.h, g.w).reshape(n, self.out_ch, g.h, g.w)


class DiagonalGaussianDistribution:
    """Posterior of the first-stage encoder (reference lvdm/distributions.py:24-40): parameters = (mean | logvar)."""

    def __init__(self, parameters, deterministic=False):
        self.parameters = parameters
        self.mean, self.logvar = torch.chunk(parameters, 2, dim=1)
        self.logvar = torch.clamp(self.logvar, -30.0, 20.0)
        self.deterministic = deterministic
        self.std = torch.exp(0.5 * self.logvar)
        self.var = torch.exp(self.logvar)
        if deterministic:
            self.var = self.std = torch.zeros_like(self.mean)

    def sample(self, noise=None):
        if noise is None:
            noise = rng.randn_like(self.mean)
        return self.mean + self.std * noise.to(self.mean.device)

    def mode(self):
        return self.mean


class AutoencoderKL(nn.Module, _Prepared):
    """Reference lvdm/models/autoencoder.py:13-199, decode side."""

    def __init__(self, ddconfig, lossconfig=None, embed_dim=4, ckpt_path=None, ignore_keys=(), image_key="image",
                 colorize_nlabels=None, monitor=None, test=False, logdir=None, input_dim=4, test_args=None):
        super().__init__()
        if ckpt_path is not None or test:
            raise NotImplementedError("checkpoints are loaded by the caller (load_state_dict); test mode is not built")
        assert ddconfig["double_z"]
        self.image_key, self.embed_dim, self.input_dim = image_key, embed_dim, input_dim
        self.encoder = Encoder(**ddconfig)
        self.decoder = Decoder(**ddconfig)
        self.quant_conv = nn.Conv2d(2 * ddconfig["z_channels"], 2 * embed_dim, 1)
        self.post_quant_conv = nn.Conv2d(embed_dim, ddconfig["z_channels"], 1)
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
        return dict(w_pq=pack.pack_linear(self.post_quant_conv.weight), b_pq=pack.pad_bias(self.post_quant_conv.bias),
                    pad=(self.embed_dim + 63) // 64 * 64,
                    w_q=pack.pack_linear(self.quant_conv.weight), b_q=pack.pad_bias(self.quant_conv.bias),
                    q_pad=(self.quant_conv.in_channels + 63) // 64 * 64)

    @torch.no_grad()
    def encode(self, x, **kwargs):
        """x [n, 3, H, W] -> DiagonalGaussianDistribution over [n, embed_dim, H/8, W/8] (reference autoencoder.py:97-101)."""
        if not x.is_cuda:
            raise CcvError("AutoencoderKL.encode: the product path runs on the GPU only (see oracle/vae_oracle.py)")
        pk = self._pk()
        enc = self.encoder
        n, c, hh, ww = x.shape
        rows = ops.pack_nchw_to_rows(x.float().reshape(n, c, 1, hh, ww), None, ldo=enc._pk()["in_pad"])
        # conv_out writes its 2*z_channels columns into a zeroed 64-wide row buffer: the quant_conv (1x1) operand
        n_out = enc._pk()["w_out"].shape[0]
        hq = torch.zeros((n * (hh // 8) * (ww // 8), pk["q_pad"]), dtype=torch.float32, device=x.device)
        _, g = enc.forward_rows(rows, _Geo(n, hh, ww), out=hq[:, :n_out])
        mom = ops.gemm(hq, pk["w_q"], bias=pk["b_q"], out_f32=True)
        moments = ops.unpack_rows_to_nchw(mom, 2 * self.embed_dim, n, 1, g.h, g.w).reshape(n, 2 * self.embed_dim, g.h, g.w)
        return DiagonalGaussianDistribution(moments)

    @torch.no_grad()
    def decode(self, z, **kwargs):
        """z [n, embed_dim, h, w] -> [n, 3, 8h, 8w] fp32 (reference autoencoder.py:103-106)."""
        if not z.is_cuda:
            raise CcvError("AutoencoderKL.decode: the product path runs on the GPU only (see oracle/vae_oracle.py)")
        pk = self._pk()
        dec = self.decoder
        n, c, hh, ww = z.shape
        rows = ops.pack_nchw_to_rows(z.float().reshape(n, c, 1, hh, ww), None, ldo=pk["pad"])
        # post_quant_conv (1x1) writes its z_channels columns into a zeroed, 64-wide row buffer: the conv_in operand
        zc_pad = dec._pk()["zc_pad"]
        zq = torch.zeros((rows.shape[0], zc_pad), dtype=torch.float32, device=z.device)
        ops.gemm(rows, pk["w_pq"], bias=pk["b_pq"], out_f32=True, out=zq[:, :pk["w_pq"].shape[0]])
        y, g = dec.forward_rows(zq, _Geo(n, hh, ww))
        return ops.unpack_rows_to_nchw(y, dec.out_ch, n, 1, g.h, g.w).reshape(n, dec.out_ch, g.h, g.w)

    def forward(self, input, sample_posterior=True):
        raise NotImplementedError("encode + decode round trips are a training-side operation")


__all__ = ["AutoencoderKL", "DiagonalGaussianDistribution", "Decoder", "Encoder", "ResnetBlock", "AttnBlock", "Upsample", "Downsample", "CcvError"]
