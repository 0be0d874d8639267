"""MI355X-native lvdm 3D-UNet with CamContextI2V camera conditioning.

Same constructor signature, module/attribute names and ``state_dict`` keys as the reference
(``lvdm/modules/networks/openaimodel3d.py``, ``lvdm/modules/attention.py``,
``model/modules/epipolar.py``; the camera additions of ``model/camcontexti2v.py:111-170`` are
built natively by :meth:`UNetModel.enable_camera_conditioning` instead of monkey patching), so
the reference's yaml configs and checkpoints load unchanged.  The torch modules only hold the
fp32 parameters; every forward runs hand-written HIP kernels through ``camc2v_amd.ops``:

  * activations stay token-major ``[(b t h w), C]`` for the whole network (no NCHW<->token
    transposes: spatial, temporal and epipolar attention only differ in the strides handed to
    the attention kernel); the residual stream is fp16 (fp32 arithmetic in every epilogue and norm), GEMM operands bf16;
  * GroupNorm+SiLU -> bf16, then implicit-GEMM conv3x3 / temporal conv with bias, time-embedding
    add and residual fused into the epilogue; LayerNorm (+Pluecker add) -> fused QKV GEMM ->
    attention -> out-projection accumulating into the stream; GEGLU fused into its GEMM;
  * step-invariant work is hoisted: context K/V projections, packed epipolar masks, Pluecker
    feature rows and register-token K/V are cached per input tensor.

There is no CPU fallback: calling a forward with CPU tensors raises (the CPU restatement lives
in ``oracle/`` and is test infrastructure only).
"""
import math
import os
import threading
from collections import OrderedDict, namedtuple

import torch
import torch.nn as nn

from . import ops, pack, parallel
from .lib import CcvError

Geom = namedtuple("Geom", "b t h w")
# A/B aid: CCV_FUSE_CAM=0 runs the three stream updates of a camera-conditioned temporal block as three GEMMs
FUSE_CAMERA_PROJECTIONS = __import__("os").environ.get("CCV_FUSE_CAM", "1") != "0"
# A/B aid: CCV_FUSE_FF=0 runs every feed-forward as LayerNorm + two GEMMs (the one-launch form exists for C = 320: csrc/ccv_fused.hip)
FUSE_FF = __import__("os").environ.get("CCV_FUSE_FF", "1") != "0"
TEXT_LEN = 77  # CrossAttention.text_context_len (reference attention.py:49)
# The residual stream between layers: fp16 -- what the reference itself carries between blocks under its fp16 autocast
# (main/trainer.py:193: conv / linear outputs are fp16, only the norms compute in fp32) -- with fp32 arithmetic inside every
# epilogue and norm.  Half the bytes of every GroupNorm / LayerNorm read and residual epilogue of the fp32 stream of rounds 1-2.
# CCV_STREAM=f32 (A/B aid) restores the fp32 stream; every kernel takes both.
STREAM = torch.float32 if os.environ.get("CCV_STREAM", "f16") == "f32" else torch.float16


def _zero(module):
    for p in module.parameters():
        nn.init.zeros_(p)
    return module


def _dev_f32(p):
    return p.detach().float().contiguous()


_PACK_LOCK = threading.RLock()


# GroupNorm statistics from the producer's epilogue: every GEMM whose output feeds a GroupNorm(32) that would otherwise take the
# two-launch path is asked for them (ops.gemm(gn_rows=): conv / temporal-conv tiles, transformer output projections, the split-K
# reduce pass; residual added, values as stored) and its output carries them to ``ops.groupnorm`` (ops.tag_stats), which then runs
# its normalise half only.  Producers whose kernel cannot emit them (the A-stationary kernel of the K = 320 projections, concat)
# leave the norm to compute its own.  CCV_GN_EPILOGUE=0 (A/B aid) switches the hand-over off.
GN_STATS_FROM_EPILOGUE = os.environ.get("CCV_GN_EPILOGUE", "1") != "0"


def _gn_rows(kind, g):
    """Rows per GroupNorm instance of the layer that consumes an output: 'frame' (ResBlock / SpatialTransformer norms), 'clip'
    (TemporalConvBlock / TemporalTransformer norms), None (no norm follows, or frames are sharded over ranks)."""
    if not GN_STATS_FROM_EPILOGUE or kind is None or parallel.current() is not None:
        return None
    return g.h * g.w if kind == "frame" else g.t * g.h * g.w


def _gemm_gn(gn_rows, *a, **kw):
    """ops.gemm whose output carries its GroupNorm statistics when `gn_rows` is given and the kernel can emit them."""
    if gn_rows is None:
        return ops.gemm(*a, **kw)
    out, st = ops.gemm(*a, gn_rows=gn_rows, **kw)
    return ops.tag_stats(out, st)


def _clip_groupnorm(x, gamma, beta, g, eps, silu):
    """GroupNorm whose statistics run over a whole clip (t, h, w): one instance per sample -- over the frames of EVERY rank when
    the clip's frames are sharded (parallel.FrameCtx)."""
    fc = parallel.current()
    if fc is None:
        return ops.groupnorm(x, gamma, beta, instances=g.b, eps=eps, silu=silu)
    return ops.groupnorm_sharded(x, gamma, beta, instances=g.b, eps=eps, silu=silu, reduce_sums=fc.shard.all_reduce_sum,
                                 total_rows_per_instance=fc.T * g.h * g.w)


class _Prepared:
    """Mixin: lazily packed device operands, dropped when parameters are (re)loaded or moved."""

    def _pk(self):
        dev = next(self.parameters()).device
        cache = self.__dict__.get("_pk_cache")
        if cache is None or cache["dev"] != dev:
            if dev.type != "cuda":
                raise CcvError(f"{type(self).__name__}: parameters live on {dev}; the product path is GPU only")
            with _PACK_LOCK:      # several host threads may drive the same module (one per clip in flight)
                cache = self.__dict__.get("_pk_cache")
                if cache is None or cache["dev"] != dev:
                    with torch.no_grad():
                        cache = self._pack()
                    if not torch.cuda.is_current_stream_capturing():
                        torch.cuda.current_stream(dev).synchronize()   # packed on this thread's stream, read from any
                    cache["dev"] = dev
                    self.__dict__["_pk_cache"] = cache
        return cache

    def invalidate(self):
        self.__dict__.pop("_pk_cache", None)


# =============================================================================================
# attention building blocks (reference: lvdm/modules/attention.py)
# =============================================================================================
class CrossAttention(nn.Module, _Prepared):
    """Multi-head attention with optional gated image-token branch (reference attention.py:44-211)."""

    def __init__(self, query_dim, context_dim=None, heads=8, dim_head=64, dropout=0.0, relative_position=False,
                 temporal_length=None, video_length=None, image_cross_attention=False,
                 image_cross_attention_scale=1.0, image_cross_attention_scale_learnable=False, text_context_len=77):
        super().__init__()
        if relative_position:
            raise NotImplementedError("relative_position attention is not used by any shipped config")
        inner = dim_head * heads
        self.context_dim = context_dim
        self.query_dim = query_dim
        cdim = context_dim if context_dim is not None else query_dim
        self.scale = dim_head ** -0.5
        self.heads, self.dim_head = heads, dim_head
        self.to_q = nn.Linear(query_dim, inner, bias=False)
        self.to_k = nn.Linear(cdim, inner, bias=False)
        self.to_v = nn.Linear(cdim, inner, bias=False)
        self.to_out = nn.Sequential(nn.Linear(inner, query_dim), nn.Dropout(dropout))
        self.relative_position = False
        self.video_length = video_length
        self.image_cross_attention = image_cross_attention
        self.image_cross_attention_scale = image_cross_attention_scale
        self.text_context_len = text_context_len
        self.image_cross_attention_scale_learnable = image_cross_attention_scale_learnable
        if image_cross_attention:
            self.to_k_ip = nn.Linear(cdim, inner, bias=False)
            self.to_v_ip = nn.Linear(cdim, inner, bias=False)
            if image_cross_attention_scale_learnable:
                self.register_parameter("alpha", nn.Parameter(torch.tensor(0.0)))

    def _pack(self):
        if self.dim_head != 64:
            raise CcvError("the HIP attention kernel is specialised for head dim 64")
        pk = {}
        if self.context_dim is None:
            pk["w_qkv"] = pack.pack_linear(torch.cat([self.to_q.weight, self.to_k.weight, self.to_v.weight], 0))
        else:
            pk["w_q"] = pack.pack_linear(self.to_q.weight)
            pk["w_kv"] = pack.pack_linear(torch.cat([self.to_k.weight, self.to_v.weight], 0))
            if self.image_cross_attention:
                pk["w_kv_ip"] = pack.pack_linear(torch.cat([self.to_k_ip.weight, self.to_v_ip.weight], 0))
                gate = self.image_cross_attention_scale
                if self.image_cross_attention_scale_learnable:
                    gate = gate * (math.tanh(float(self.alpha)) + 1.0)
                pk["gate"] = float(gate)
        pk["w_o"] = pack.pack_linear(self.to_out[0].weight)
        pk["b_o"] = _dev_f32(self.to_out[0].bias)
        return pk

    # ---- self attention over the pixels of each frame: n [(b t hw), C] -------------------------
    def self_attn_spatial(self, n, stream, g):
        pk = self._pk()
        C, H, hw = self.query_dim, self.heads, g.h * g.w
        qkv = ops.gemm(n, pk["w_qkv"])
        ld = 3 * C
        s = (hw * ld, 0, ld)
        o = ops.attention(qkv, qkv[:, C:], qkv[:, 2 * C:], B=g.b * g.t, inner=1, H=H, Lq=hw, Lk=hw,
                          q_str=s, k_str=s, v_str=s, scale=self.scale)
        ops.gemm(o, pk["w_o"], bias=pk["b_o"], residual=stream, out_dtype=stream.dtype, out=stream)

    # ---- self attention over the frames of each pixel; activations stay token-major -------------
    def project_qkv(self, n):
        return ops.gemm(n, self._pk()["w_qkv"])

    def temporal_attend(self, n, g, out=None, qkv=None, kv_all=None):
        """softmax(q k^T) v over the frames of each pixel, before the output projection: bf16 [(b t hw), C] (into `out`).
        qkv / kv_all (frame-sharded mode): this rank's projection and the K | V rows of all ranks, when the caller gathered them
        together with another attention's (one collective for both)."""
        pk = self._pk()
        C, H, hw = self.query_dim, self.heads, g.h * g.w
        if qkv is None:
            qkv = ops.gemm(n, pk["w_qkv"])
        ld = 3 * C
        s = (g.t * hw * ld, ld, hw * ld)
        o = torch.empty((n.shape[0], C), dtype=ops.BF16, device=n.device) if out is None else out
        fc = parallel.current()
        if fc is not None:     # frames sharded over ranks: this rank's queries against every rank's keys / values
            kv = kv_all if kv_all is not None else fc.gather_frames(qkv[:, C:].contiguous(), g.b, hw)      # [(b T hw), 2C]
            sk = (fc.T * hw * 2 * C, 2 * C, hw * 2 * C)
            ops.attention(qkv, kv, kv[:, C:], B=g.b * hw, inner=hw, H=H, Lq=g.t, Lk=fc.T, q_str=s, k_str=sk, v_str=sk, out=o,
                          o_str=(g.t * hw * C, C, hw * C), scale=self.scale)
            return o
        ops.attention(qkv, qkv[:, C:], qkv[:, 2 * C:], B=g.b * hw, inner=hw, H=H, Lq=g.t, Lk=g.t,
                      q_str=s, k_str=s, v_str=s, out=o, o_str=(g.t * hw * C, C, hw * C), scale=self.scale)
        return o

    def self_attn_temporal(self, n, stream, g):
        pk = self._pk()
        ops.gemm(self.temporal_attend(n, g), pk["w_o"], bias=pk["b_o"], residual=stream, out_dtype=stream.dtype, out=stream)

    # ---- cross attention against cached context projections --------------------------------------
    def project_context(self, text_rows, img_rows):
        """text_rows [nb*77, Dctx] bf16, img_rows [nb*Li, Dctx] bf16 or None -> (kv_text, kv_img)."""
        pk = self._pk()
        kv_t = ops.gemm(text_rows, pk["w_kv"])
        kv_i = ops.gemm(img_rows, pk["w_kv_ip"]) if (img_rows is not None and self.image_cross_attention) else None
        return kv_t, kv_i

    def cross_attn(self, n, stream, g, groups):
        """groups: list of (clip_start, n_clips, kv_text, kv_img, img_tokens, per_frame)."""
        pk = self._pk()
        C, H, hw = self.query_dim, self.heads, g.h * g.w
        q = ops.gemm(n, pk["w_q"])
        o = torch.empty_like(q)
        rows_per_clip = g.t * hw
        for (c0, nc, kv_t, kv_i, li, per_frame) in groups:
            r0 = c0 * rows_per_clip
            qs, os_ = q[r0:], o[r0:]
            kw = {}
            if kv_i is not None:
                if per_frame:   # 16 tokens of its own for every frame (reference openaimodel3d.py:575-579)
                    fc = parallel.current()     # frame-sharded: this rank's frames start at f0 of the clip's T
                    t_all, f0 = (fc.T, fc.f0) if fc is not None else (g.t, 0)
                    st = (t_all * li * 2 * C, li * 2 * C, 2 * C)
                    kv_i = kv_i[f0 * li:] if f0 else kv_i
                else:           # the same image tokens for every frame (:580-581)
                    st = (li * 2 * C, 0, 2 * C)
                kw = dict(k2=kv_i, v2=kv_i[:, C:], k2_str=st, v2_str=st, Lk2=li, gate2=pk["gate"])
            st_t = (TEXT_LEN * 2 * C, 0, 2 * C)
            ops.attention(qs, kv_t, kv_t[:, C:], B=nc * g.t, inner=g.t, H=H, Lq=hw, Lk=TEXT_LEN,
                          q_str=(rows_per_clip * C, hw * C, C), k_str=st_t, v_str=st_t,
                          out=os_, o_str=(rows_per_clip * C, hw * C, C), scale=self.scale, **kw)
        ops.gemm(o, pk["w_o"], bias=pk["b_o"], residual=stream, out_dtype=stream.dtype, out=stream)

    # ---- reference-shaped entry point (module level parity tests) ---------------------------------
    def forward(self, x, context=None, mask=None):
        """x [b, n, C]; context None (self attention) or [b, L, Dctx] (first 77 tokens text, rest image).
        Returns fp32 [b, n, C] = to_out(attention), like the reference forward (attention.py:85-146)."""
        if mask is not None:
            raise NotImplementedError("attention masks are only used by the epipolar module")
        b, n, C = x.shape
        xr = x.reshape(b * n, C).float().contiguous()
        xb = ops.cast_bf16(xr)
        stream = torch.zeros_like(xr)
        g = Geom(b, 1, 1, n)
        if context is None:
            self.self_attn_spatial(xb, stream, g)
        else:
            ct = ops.cast_bf16(context[:, :self.text_context_len].float().contiguous().reshape(-1, context.shape[-1]))
            ci = None
            li = context.shape[1] - self.text_context_len
            if self.image_cross_attention and li > 0:
                ci = ops.cast_bf16(context[:, self.text_context_len:].float().contiguous().reshape(-1, context.shape[-1]))
            kv_t, kv_i = self.project_context(ct, ci)
            self.cross_attn(xb, stream, g, [(0, b, kv_t, kv_i, li, False)])
        return stream.reshape(b, n, C)


class GEGLU(nn.Module):
    def __init__(self, dim_in, dim_out):
        super().__init__()
        self.proj = nn.Linear(dim_in, dim_out * 2)


class FeedForward(nn.Module, _Prepared):
    """Linear(C, 8C) -> value * gelu(gate) -> Linear(4C, C) (reference attention.py:431-458)."""

    def __init__(self, dim, dim_out=None, mult=4, glu=False, dropout=0.0):
        super().__init__()
        if not glu:
            raise NotImplementedError("only the gated (GEGLU) feed-forward is used by the UNet")
        inner = int(dim * mult)
        self.net = nn.Sequential(GEGLU(dim, inner), nn.Dropout(dropout), nn.Linear(inner, dim_out or dim))

    def _pack(self):
        w, b = pack.interleave_geglu(self.net[0].proj.weight, self.net[0].proj.bias)
        pk = dict(w1=w, b1=b, w2=pack.pack_linear(self.net[2].weight), b2=_dev_f32(self.net[2].bias))
        if self.net[2].in_features == 4 * self.net[2].out_features == 4 * self.net[0].proj.in_features and self.net[2].out_features == 320:
            pk["w2p"] = pack.permute_k16_for_acc_operand(self.net[2].weight)     # operand of the one-launch form (ops.ff_fused)
        return pk

    def run(self, n, stream, final=False):
        """n: LayerNorm of the stream (an ops.LazyLN, or the bf16 rows).  final: the stream is only read as a GEMM operand afterwards
        (proj_out), so the sum is handed back rounded to bf16 (the rounding the operand load would apply anyway) instead of updating
        the stream."""
        pk = self._pk()
        if FUSE_FF and isinstance(n, ops.LazyLN) and n._t is None and n.x is stream and ops.ff_fusable(stream, pk["w1"], pk.get("w2p")):
            # LayerNorm -> GEGLU up -> down -> + x as ONE launch: the [M, 4C] hidden activation never reaches memory
            if final:
                return ops.ff_fused(stream, n.gamma, n.beta, n.eps, pk["w1"], pk["b1"], pk["w2p"], pk["b2"], out_dtype=ops.BF16)
            return ops.ff_fused(stream, n.gamma, n.beta, n.eps, pk["w1"], pk["b1"], pk["w2p"], pk["b2"], out=stream)
        hidden = ops.gemm(n, pk["w1"], bias=pk["b1"], geglu=True)
        if final:
            return ops.gemm(hidden, pk["w2"], bias=pk["b2"], residual=stream)
        ops.gemm(hidden, pk["w2"], bias=pk["b2"], residual=stream, out_dtype=stream.dtype, out=stream)
        return stream

    def forward(self, x):
        shp = x.shape
        xr = x.reshape(-1, shp[-1]).float().contiguous()
        stream = torch.zeros_like(xr)
        self.run(ops.cast_bf16(xr), stream)
        return stream.reshape(shp)


class EpipolarCrossAttention(nn.Module):
    """Parameter container (reference model/modules/epipolar.py:43-102)."""

    def __init__(self, query_dim, context_dim=None, out_dim=None, heads=8, dim_head=64, dropout=0.0, num_register_tokens=0):
        super().__init__()
        inner = dim_head * heads
        self.context_dim = context_dim
        cdim = context_dim if context_dim is not None else query_dim
        self.scale, self.heads, self.dim_head = dim_head ** -0.5, heads, dim_head
        self.to_q = nn.Linear(query_dim, inner, bias=False)
        self.to_k = nn.Linear(cdim, inner, bias=False)
        self.to_v = nn.Linear(cdim, inner, bias=False)
        self.to_out = nn.Sequential(nn.Linear(inner, out_dim if out_dim is not None else query_dim), nn.Dropout(dropout))
        self.num_register_tokens = num_register_tokens
        if num_register_tokens > 0:
            self.register_tokens = nn.Parameter(torch.randn((1, num_register_tokens, cdim)))


class Epipolar(nn.Module, _Prepared):
    """Masked attention over all T*H*W tokens of a clip with always-visible register tokens
    (reference model/modules/epipolar.py:105-157)."""

    def __init__(self, query_dim, context_dim, heads, origin_h=256, origin_w=256, is_3d_full_attn=False,
                 num_register_tokens=0, compression_factor=1, attention_resolution=(8, 4, 2, 1),
                 only_on_cond_frame=False, **kwargs):
        super().__init__()
        if only_on_cond_frame:
            raise NotImplementedError("only_on_cond_frame is not used by any shipped config")
        self.attention_resolution = list(attention_resolution)
        self.origin_h, self.origin_w = origin_h, origin_w
        self.num_heads = heads
        self.is_3d_full_attn = is_3d_full_attn
        self.only_on_cond_frame = False
        self.compression_factor = compression_factor
        self.query_dim = query_dim
        self.epipolar_attn = EpipolarCrossAttention(
            query_dim=query_dim, context_dim=context_dim, heads=heads,
            dim_head=int(query_dim // heads // compression_factor), num_register_tokens=num_register_tokens)
        _zero(self.epipolar_attn.to_out[0])

    def _pack(self):
        a = self.epipolar_attn
        if a.dim_head != 64:
            raise CcvError("the HIP attention kernel is specialised for head dim 64")
        pk = dict(w_qkv=pack.pack_linear(torch.cat([a.to_q.weight, a.to_k.weight, a.to_v.weight], 0)),
                  w_o=pack.pack_linear(a.to_out[0].weight), b_o=_dev_f32(a.to_out[0].bias))
        if a.num_register_tokens > 0:
            # register K/V do not depend on the input: project them once (epipolar.py:86-90)
            reg = a.register_tokens[0].float()
            pk["kreg"] = (reg @ a.to_k.weight.float().t()).to(ops.BF16).contiguous()
            pk["vreg"] = (reg @ a.to_v.weight.float().t()).to(ops.BF16).contiguous()
        return pk

    def run(self, src, stream, g, packed_mask):
        """src [(b t hw), C] bf16 (= LN(x) + Pluecker rows); packed_mask (bits, flags, nb) or None."""
        pk = self._pk()
        ops.gemm(self.attend(src, g, packed_mask), pk["w_o"], bias=pk["b_o"], residual=stream, out_dtype=stream.dtype, out=stream)

    def project_qkv(self, src):
        return ops.gemm(src, self._pk()["w_qkv"])

    def attend(self, src, g, packed_mask, out=None, qkv=None, kv_all=None):
        """The masked attention over all T*H*W tokens, before the output projection: bf16 [(b t hw), C] (into `out`).
        qkv / kv_all: see CrossAttention.temporal_attend."""
        pk = self._pk()
        C, H = self.query_dim, self.num_heads
        L = g.t * g.h * g.w
        if qkv is None:
            qkv = ops.gemm(src, pk["w_qkv"])
        ld = 3 * C
        s = (L * ld, 0, ld)
        kw = {}
        if packed_mask is not None and not self.is_3d_full_attn:
            bits, flags, nb, perm, wbits, order = packed_mask
            if bits.shape[1] != L:
                raise CcvError(f"epipolar mask has {bits.shape[1]} query rows, feature map has {L} tokens")
            if perm is not None and perm != (g.h * g.w, g.w):
                raise CcvError(f"epipolar mask was packed for frames {perm}, feature map is {g.h}x{g.w}")
            kw = dict(mask_bits=bits, tile_flags=flags, mask_nb=nb, perm=perm, wave_bits=wbits, group_order=order)
        fc = parallel.current()
        if fc is not None:     # frames sharded over ranks: local queries (and their mask rows, prepared in UNetModel.forward)
            kv = kv_all if kv_all is not None else fc.gather_frames(qkv[:, C:].contiguous(), g.b, g.h * g.w)     # against all T*h*w keys / values
            La = fc.T * g.h * g.w
            if out is not None:
                kw.update(out=out, o_str=(L * C, 0, C))
            return ops.attention(qkv, kv, kv[:, C:], B=g.b, inner=1, H=H, Lq=L, Lk=La, q_str=s, k_str=(La * 2 * C, 0, 2 * C),
                                 v_str=(La * 2 * C, 0, 2 * C), kreg=pk.get("kreg"), vreg=pk.get("vreg"), scale=self.epipolar_attn.scale, **kw)
        if out is not None:
            kw.update(out=out, o_str=(L * C, 0, C))
        return ops.attention(qkv, qkv[:, C:], qkv[:, 2 * C:], B=g.b, inner=1, H=H, Lq=L, Lk=L, q_str=s, k_str=s, v_str=s,
                             kreg=pk.get("kreg"), vreg=pk.get("vreg"), scale=self.epipolar_attn.scale, **kw)


class BasicTransformerBlock(nn.Module, _Prepared):
    """LN -> attn1 -> LN -> attn2 -> LN -> GEGLU FF, each residual (reference attention.py:214-253;
    temporal blocks additionally carry the camera branch of modified_forwards.py:505-536)."""

    def __init__(self, dim, n_heads, d_head, dropout=0.0, context_dim=None, gated_ff=True, checkpoint=True,
                 disable_self_attn=False, attention_cls=None, video_length=None, image_cross_attention=False,
                 image_cross_attention_scale=1.0, image_cross_attention_scale_learnable=False, text_context_len=77,
                 is_output_block=False, ds=1):
        super().__init__()
        if disable_self_attn:
            raise NotImplementedError("disable_self_attn is not used by any shipped config")
        self.ds, self.dim = ds, dim
        self.context_dim = context_dim
        self.disable_self_attn = False
        self.attn1 = CrossAttention(query_dim=dim, heads=n_heads, dim_head=d_head, dropout=dropout, context_dim=None)
        self.ff = FeedForward(dim, dropout=dropout, glu=gated_ff)
        self.attn2 = CrossAttention(query_dim=dim, context_dim=context_dim, heads=n_heads, dim_head=d_head,
                                    dropout=dropout, video_length=video_length,
                                    image_cross_attention=image_cross_attention,
                                    image_cross_attention_scale=image_cross_attention_scale,
                                    image_cross_attention_scale_learnable=image_cross_attention_scale_learnable,
                                    text_context_len=text_context_len)
        self.image_cross_attention = image_cross_attention
        self.norm1, self.norm2, self.norm3 = nn.LayerNorm(dim), nn.LayerNorm(dim), nn.LayerNorm(dim)
        self.checkpoint = checkpoint
        self.is_output_block = is_output_block

    def _pack(self):
        pk = {}
        for i, ln in enumerate((self.norm1, self.norm2, self.norm3), 1):
            pk[f"g{i}"], pk[f"b{i}"] = _dev_f32(ln.weight), _dev_f32(ln.bias)
        if hasattr(self, "pluker_projection"):
            pk["w_pl"] = pack.pack_linear(self.pluker_projection.weight)
            pk["b_pl"] = _dev_f32(self.pluker_projection.bias)
            if hasattr(self, "epipolar") and self.epipolar.epipolar_attn.dim_head == 64:
                # 'add_to_main_branch': x += attn1.to_out(o1) + pluker_projection(n + P) + epipolar.to_out(o2) -- three linear maps
                # into the stream: ONE GEMM over the stacked operands [o1; n + P; o2] against [W_o1 | W_pl | W_oe]
                # (K = 3C, one read-modify-write of the fp32 stream instead of three)
                wo1, woe = self.attn1.to_out[0], self.epipolar.epipolar_attn.to_out[0]
                pk["w_cam"] = pack.pack_linear(torch.cat([wo1.weight, self.pluker_projection.weight, woe.weight], 1))
                pk["b_cam"] = _dev_f32(wo1.bias.float() + self.pluker_projection.bias.float() + woe.bias.float())
        return pk

    def _ln(self, i, stream, addend=None, out2=None):
        """LayerNorm i of the stream: with an addend (the Pluecker rows) the two bf16 tensors, else a lazy operand -- the projection
        that consumes it runs the norm in its own prologue where its kernel can (ops.LazyLN)."""
        pk = self._pk()
        if addend is None and out2 is None:
            return ops.LazyLN(stream, pk[f"g{i}"], pk[f"b{i}"], getattr(self, f"norm{i}").eps)
        return ops.layernorm(stream, pk[f"g{i}"], pk[f"b{i}"], eps=getattr(self, f"norm{i}").eps, addend=addend, out2=out2)

    def run_spatial(self, stream, g, ctx_groups, final=False):
        self.attn1.self_attn_spatial(self._ln(1, stream), stream, g)
        self.attn2.cross_attn(self._ln(2, stream), stream, g, ctx_groups)
        return self.ff.run(self._ln(3, stream), stream, final)

    def run_temporal(self, stream, g, cam, final=False):
        """cam: None or dict(rows=bf16 Pluecker rows or None, mask=(bits, flags, nb) or None, add_type=str)."""
        patched = hasattr(self, "pluker_projection") or hasattr(self, "epipolar")
        if cam is None or not patched:
            self.attn1.self_attn_temporal(self._ln(1, stream), stream, g)
        else:
            pk = self._pk()
            prow = cam.get("rows")
            main = cam.get("add_type") == "add_to_main_branch"
            if main and prow is not None and "w_cam" in pk and FUSE_CAMERA_PROJECTIONS:
                rows, C = stream.shape
                stack = torch.empty((3, rows, C), dtype=ops.BF16, device=stream.device)   # [o1 ; n + P ; o2]
                n, _ = self._ln(1, stream, addend=prow, out2=stack[1])
                fc = parallel.current()
                if fc is not None:     # frames sharded over ranks: the K | V of both attentions in ONE all_gather
                    Cq = self.attn1.query_dim
                    qkv1, qkv2 = self.attn1.project_qkv(n), self.epipolar.project_qkv(stack[1])
                    kv1, kv2 = fc.gather_frames_multi([qkv1[:, Cq:].contiguous(), qkv2[:, Cq:].contiguous()], g.b, g.h * g.w)
                    self.attn1.temporal_attend(n, g, out=stack[0], qkv=qkv1, kv_all=kv1)
                    self.epipolar.attend(stack[1], g, cam.get("mask"), out=stack[2], qkv=qkv2, kv_all=kv2)
                else:
                    self.attn1.temporal_attend(n, g, out=stack[0])
                    self.epipolar.attend(stack[1], g, cam.get("mask"), out=stack[2])
                ops.gemm(stack.view(3 * rows, C), pk["w_cam"], k=C, taps=3, m=rows, gather=ops.GATHER_SEGMENTS, seg_rows=rows,
                         bias=pk["b_cam"], residual=stream, out_dtype=stream.dtype, out=stream)
                self.attn2.self_attn_temporal(self._ln(2, stream), stream, g)
                return self.ff.run(self._ln(3, stream), stream, final)
            if prow is not None:
                n, src = self._ln(1, stream, addend=prow)
            else:
                n = src = self._ln(1, stream).tensor()     # several consumers, the stream changes between them: materialise
            target = stream if main else torch.zeros_like(stream)
            if main:
                self.attn1.self_attn_temporal(n, stream, g)
            if prow is not None and "w_pl" in pk:
                ops.gemm(src, pk["w_pl"], bias=pk["b_pl"], residual=target, out_dtype=target.dtype, out=target)
            if hasattr(self, "epipolar"):
                self.epipolar.run(src, target, g, cam.get("mask"))
            if not main:
                # x = attn1(LN(x) + z) + x   (modified_forwards.py:531)
                _, nz = self._ln(1, stream, addend=ops.cast_bf16(target))
                self.attn1.self_attn_temporal(nz, stream, g)
        self.attn2.self_attn_temporal(self._ln(2, stream), stream, g)
        return self.ff.run(self._ln(3, stream), stream, final)


class SpatialTransformer(nn.Module, _Prepared):
    """GroupNorm -> proj_in -> blocks -> proj_out -> + input (reference attention.py:256-320)."""

    def __init__(self, in_channels, n_heads, d_head, depth=1, dropout=0.0, context_dim=None, use_checkpoint=True,
                 disable_self_attn=False, use_linear=False, video_length=None, image_cross_attention=False,
                 image_cross_attention_scale_learnable=False, is_output_block=False, ds=1):
        super().__init__()
        self.ds, self.in_channels = ds, in_channels
        inner = n_heads * d_head
        self.norm = nn.GroupNorm(32, in_channels, eps=1e-6, affine=True)
        self.proj_in = nn.Linear(in_channels, inner) if use_linear else nn.Conv2d(in_channels, inner, 1)
        self.is_output_block = is_output_block
        self.transformer_blocks = nn.ModuleList([
            BasicTransformerBlock(inner, n_heads, d_head, dropout=dropout, context_dim=context_dim,
                                  disable_self_attn=disable_self_attn, checkpoint=use_checkpoint,
                                  video_length=video_length, image_cross_attention=image_cross_attention,
                                  image_cross_attention_scale_learnable=image_cross_attention_scale_learnable,
                                  is_output_block=is_output_block, ds=ds) for _ in range(depth)])
        self.proj_out = _zero(nn.Linear(inner, in_channels) if use_linear else nn.Conv2d(inner, in_channels, 1))
        self.use_linear = use_linear

    def _pack(self):
        return dict(gn_g=_dev_f32(self.norm.weight), gn_b=_dev_f32(self.norm.bias),
                    w_in=pack.pack_linear(self.proj_in.weight), b_in=_dev_f32(self.proj_in.bias),
                    w_out=pack.pack_linear(self.proj_out.weight), b_out=_dev_f32(self.proj_out.bias))

    def forward_rows(self, x, g, ctx_groups_per_block, next_norm=None):
        """next_norm: 'frame' / 'clip' / None -- the GroupNorm that consumes the output (its statistics come out of proj_out's epilogue)."""
        pk = self._pk()
        n = ops.groupnorm(x, pk["gn_g"], pk["gn_b"], instances=g.b * g.t, eps=self.norm.eps, silu=False)
        s = ops.gemm(n, pk["w_in"], bias=pk["b_in"], out_dtype=x.dtype)
        last = len(self.transformer_blocks) - 1
        for i, (blk, groups) in enumerate(zip(self.transformer_blocks, ctx_groups_per_block)):
            s = blk.run_spatial(s, g, groups, final=(i == last))
        return _gemm_gn(_gn_rows(next_norm, g), s, pk["w_out"], bias=pk["b_out"], residual=x, out_dtype=x.dtype)


class TemporalTransformer(nn.Module, _Prepared):
    """GroupNorm over the clip -> proj_in -> blocks over the frame axis -> proj_out -> + input
    (reference attention.py:323-428 / modified_forwards.py:401-450)."""

    def __init__(self, in_channels, n_heads, d_head, depth=1, dropout=0.0, context_dim=None, use_checkpoint=True,
                 use_linear=False, only_self_att=True, causal_attention=False, causal_block_size=1,
                 relative_position=False, temporal_length=None, is_output_block=False, ds=1):
        super().__init__()
        if not only_self_att or causal_attention or relative_position:
            raise NotImplementedError("temporal cross / causal / relative-position attention is not used by any shipped config")
        self.ds = ds
        self.only_self_att, self.relative_position, self.causal_attention = True, False, False
        self.causal_block_size = causal_block_size
        self.in_channels = in_channels
        inner = n_heads * d_head
        self.norm = nn.GroupNorm(32, in_channels, eps=1e-6, affine=True)
        self.proj_in = nn.Linear(in_channels, inner) if use_linear else nn.Conv1d(in_channels, inner, 1)
        self.is_output_block = is_output_block
        self.transformer_blocks = nn.ModuleList([
            BasicTransformerBlock(inner, n_heads, d_head, dropout=dropout, context_dim=None,
                                  checkpoint=use_checkpoint, is_output_block=is_output_block, ds=ds)
            for _ in range(depth)])
        self.proj_out = _zero(nn.Linear(inner, in_channels) if use_linear else nn.Conv1d(inner, in_channels, 1))
        self.use_linear = use_linear

    _pack = SpatialTransformer._pack

    def forward_rows(self, x, g, cam, next_norm=None):
        pk = self._pk()
        n = _clip_groupnorm(x, pk["gn_g"], pk["gn_b"], g, self.norm.eps, False)
        s = ops.gemm(n, pk["w_in"], bias=pk["b_in"], out_dtype=x.dtype)
        last = len(self.transformer_blocks) - 1
        for i, blk in enumerate(self.transformer_blocks):
            s = blk.run_temporal(s, g, cam, final=(i == last))
        return _gemm_gn(_gn_rows(next_norm, g), s, pk["w_out"], bias=pk["b_out"], residual=x, out_dtype=x.dtype)


# =============================================================================================
# convolutional blocks (reference: lvdm/modules/networks/openaimodel3d.py)
# =============================================================================================
class TimestepBlock(nn.Module):
    pass


class TimestepEmbedSequential(nn.Sequential, TimestepBlock):
    """Container; dispatch happens in UNetModel._run_block (reference openaimodel3d.py:30-48)."""


class Downsample(nn.Module, _Prepared):
    def __init__(self, channels, use_conv, dims=2, out_channels=None, padding=1):
        super().__init__()
        if not use_conv or dims != 2:
            raise NotImplementedError("only conv_resample=True, dims=2 is used")
        self.channels, self.out_channels, self.use_conv, self.dims = channels, out_channels or channels, True, dims
        self.op = nn.Conv2d(channels, self.out_channels, 3, stride=2, padding=padding)

    def _pack(self):
        return dict(w=pack.pack_conv3x3(self.op.weight), b=_dev_f32(self.op.bias))

    def forward_rows(self, x, g, next_norm=None):
        pk = self._pk()
        oh, ow = (g.h + 1) // 2, (g.w + 1) // 2
        sd = x.dtype
        x = ops.cast_bf16(x)   # bf16 operand: the conv then runs on the LDS-DMA kernel (a stream-typed A needs register staging)
        go = Geom(g.b, g.t, oh, ow)
        y = _gemm_gn(_gn_rows(next_norm, go), x, pk["w"], k=self.channels, taps=9, m=g.b * g.t * oh * ow, bias=pk["b"], out_dtype=sd,
                     gather=ops.GATHER_CONV3X3, conv=(oh, ow, g.h, g.w, 2, 0))
        return y, go


class Upsample(nn.Module, _Prepared):
    def __init__(self, channels, use_conv, dims=2, out_channels=None, padding=1):
        super().__init__()
        if not use_conv or dims != 2:
            raise NotImplementedError("only conv_resample=True, dims=2 is used")
        self.channels, self.out_channels, self.use_conv, self.dims = channels, out_channels or channels, True, dims
        self.conv = nn.Conv2d(channels, self.out_channels, 3, padding=padding)

    def _pack(self):
        return dict(w=pack.pack_conv3x3(self.conv.weight), b=_dev_f32(self.conv.bias))

    def forward_rows(self, x, g, next_norm=None):
        pk = self._pk()
        oh, ow = 2 * g.h, 2 * g.w   # nearest 2x is folded into the conv's gather
        sd = x.dtype
        x = ops.cast_bf16(x)
        y = ops.gemm(x, pk["w"], k=self.channels, taps=9, m=g.b * g.t * oh * ow, bias=pk["b"], out_dtype=sd,
                     gather=ops.GATHER_CONV3X3, conv=(oh, ow, g.h, g.w, 1, 1))
        return y, Geom(g.b, g.t, oh, ow)


class TemporalConvBlock(nn.Module, _Prepared):
    """4 x [GroupNorm(clip) + SiLU + Conv3d(3,1,1)] + identity (reference openaimodel3d.py:239-279)."""

    def __init__(self, in_channels, out_channels=None, dropout=0.0, spatial_aware=False):
        super().__init__()
        if spatial_aware:
            raise NotImplementedError("tempspatial_aware is not used by any shipped config")
        out_channels = out_channels or in_channels
        self.in_channels, self.out_channels = in_channels, out_channels
        k, p = (3, 1, 1), (1, 0, 0)
        self.conv1 = nn.Sequential(nn.GroupNorm(32, in_channels), nn.SiLU(), nn.Conv3d(in_channels, out_channels, k, padding=p))
        self.conv2 = nn.Sequential(nn.GroupNorm(32, out_channels), nn.SiLU(), nn.Dropout(dropout), nn.Conv3d(out_channels, in_channels, k, padding=p))
        self.conv3 = nn.Sequential(nn.GroupNorm(32, out_channels), nn.SiLU(), nn.Dropout(dropout), nn.Conv3d(out_channels, in_channels, k, padding=p))
        self.conv4 = nn.Sequential(nn.GroupNorm(32, out_channels), nn.SiLU(), nn.Dropout(dropout), nn.Conv3d(out_channels, in_channels, k, padding=p))
        _zero(self.conv4[-1])

    def _pack(self):
        pk = {}
        for i, seq in enumerate((self.conv1, self.conv2, self.conv3, self.conv4)):
            pk[f"g{i}"], pk[f"b{i}"] = _dev_f32(seq[0].weight), _dev_f32(seq[0].bias)
            pk[f"w{i}"], pk[f"cb{i}"] = pack.pack_tconv3(seq[-1].weight), _dev_f32(seq[-1].bias)
        return pk

    def forward_rows(self, x, g, next_norm=None):
        pk = self._pk()
        C = self.in_channels
        h = x
        fc = parallel.current()
        for i in range(4):
            if fc is None:    # (h carries the statistics its producing convolution emitted: whole clip = one instance)
                z = ops.groupnorm(h, pk[f"g{i}"], pk[f"b{i}"], instances=g.b, eps=1e-5, silu=True)
            else:
                # frames sharded over ranks: ONE collective per convolution -- every rank's partial GroupNorm sums travel with its
                # two (un-normalised) edge frames; the clip-wide statistics then normalise the local rows and the two halo frames alike
                hw = g.h * g.w
                total = fc.T * hw
                prev, nxt, sums, first, lastr = fc.edges_and_sums(h, ops.groupnorm_partial_sums(h, g.b), g.b, hw)
                norm = lambda rows: ops.groupnorm_apply_sums(rows, pk[f"g{i}"], pk[f"b{i}"], sums, instances=g.b,
                                                             total_rows_per_instance=total, eps=1e-5, silu=True)
                z = norm(h)
                halo = norm(torch.stack([prev, nxt], 1).reshape(g.b * 2 * hw, C)).reshape(g.b, 2, hw, C)
                if first:
                    halo[:, 0].zero_()        # the clip's ends: the convolution's zero padding (not normalised zeros)
                if lastr:
                    halo[:, 1].zero_()
                ze = torch.cat([halo[:, :1], z.reshape(g.b, g.t, hw, C), halo[:, 1:]], 1).reshape(g.b * (g.t + 2) * hw, C)
            last = i == 3
            if fc is not None:     # frames sharded over ranks: the neighbours' edge frames in front of / behind the local ones
                hw = g.h * g.w
                he = ops.gemm(ze, pk[f"w{i}"], k=C, taps=3, bias=pk[f"cb{i}"], gather=ops.GATHER_TCONV3, tconv=(g.t + 2, hw),
                              out_dtype=torch.float32 if last else ops.BF16)
                h = fc.inner(he, g.b, hw)
                if last:
                    h = (h + x.float()).to(x.dtype)
                continue
            if last:
                h = _gemm_gn(_gn_rows(next_norm, g), z, pk[f"w{i}"], k=C, taps=3, bias=pk[f"cb{i}"], gather=ops.GATHER_TCONV3,
                             tconv=(g.t, g.h * g.w), residual=x, out_dtype=x.dtype)
            else:
                h = _gemm_gn(_gn_rows("clip", g), z, pk[f"w{i}"], k=C, taps=3, bias=pk[f"cb{i}"], gather=ops.GATHER_TCONV3,
                             tconv=(g.t, g.h * g.w))
        return h


class ResBlock(TimestepBlock, _Prepared):
    """GN+SiLU+conv3x3 -> +emb -> GN+SiLU+conv3x3 -> +skip -> temporal conv (reference openaimodel3d.py:109-236)."""

    def __init__(self, channels, emb_channels, dropout, out_channels=None, use_scale_shift_norm=False, dims=2,
                 use_checkpoint=False, use_conv=False, up=False, down=False, use_temporal_conv=False,
                 tempspatial_aware=False):
        super().__init__()
        if use_scale_shift_norm or up or down or dims != 2:
            raise NotImplementedError("scale-shift norm / resblock_updown are not used by any shipped config")
        self.channels, self.emb_channels, self.dropout = channels, emb_channels, dropout
        self.out_channels = out_channels or channels
        self.use_conv, self.use_checkpoint = use_conv, use_checkpoint
        self.use_scale_shift_norm, self.use_temporal_conv, self.updown = False, use_temporal_conv, False
        self.in_layers = nn.Sequential(nn.GroupNorm(32, channels), nn.SiLU(),
                                       nn.Conv2d(channels, self.out_channels, 3, padding=1))
        self.h_upd = self.x_upd = nn.Identity()
        self.emb_layers = nn.Sequential(nn.SiLU(), nn.Linear(emb_channels, self.out_channels))
        self.out_layers = nn.Sequential(nn.GroupNorm(32, self.out_channels), nn.SiLU(), nn.Dropout(p=dropout),
                                        _zero(nn.Conv2d(self.out_channels, self.out_channels, 3, padding=1)))
        if self.out_channels == channels:
            self.skip_connection = nn.Identity()
        elif use_conv:
            self.skip_connection = nn.Conv2d(channels, self.out_channels, 3, padding=1)
        else:
            self.skip_connection = nn.Conv2d(channels, self.out_channels, 1)
        if use_temporal_conv:
            # sic: the attribute name (and checkpoint key) is misspelt in the reference (openaimodel3d.py:190)
            self.temopral_conv = TemporalConvBlock(self.out_channels, self.out_channels, dropout=0.1,
                                                   spatial_aware=tempspatial_aware)
        self.emb_slice = None  # (offset, width) into the UNet-wide fused emb projection

    def _pack(self):
        pk = dict(g1=_dev_f32(self.in_layers[0].weight), b1=_dev_f32(self.in_layers[0].bias),
                  w1=pack.pack_conv3x3(self.in_layers[2].weight), cb1=_dev_f32(self.in_layers[2].bias),
                  g2=_dev_f32(self.out_layers[0].weight), b2=_dev_f32(self.out_layers[0].bias),
                  w2=pack.pack_conv3x3(self.out_layers[3].weight), cb2=_dev_f32(self.out_layers[3].bias))
        sc = self.skip_connection
        if isinstance(sc, nn.Conv2d):
            pk["ws"] = pack.pack_conv3x3(sc.weight) if sc.kernel_size[0] == 3 else pack.pack_linear(sc.weight)
            pk["bs"] = _dev_f32(sc.bias)
            pk["skip_taps"] = 9 if sc.kernel_size[0] == 3 else 1
        return pk

    def forward_rows(self, x, emb_all, g, x_bf16=None, next_norm=None):
        """x [(b t h w), Cin] in the stream dtype; emb_all fp32 [b, sum Cout] (all ResBlock emb projections in one GEMM);
        x_bf16: optional bf16 rounding of x (what the skip convolution's operand load would produce anyway)."""
        pk = self._pk()
        cin, cout = self.channels, self.out_channels
        conv = (g.h, g.w, g.h, g.w, 1, 0)
        off, width = self.emb_slice
        assert width == cout
        h = ops.groupnorm(x, pk["g1"], pk["b1"], instances=g.b * g.t, eps=1e-5, silu=True)
        # the second norm's statistics come out of the first convolution's epilogue (or its split-K reduce pass)
        h = _gemm_gn(_gn_rows("frame", g), h, pk["w1"], k=cin, taps=9, bias=pk["cb1"], bias2=emb_all[:, off:], ldb2=emb_all.stride(0),
                     rows_per_batch=g.t * g.h * g.w, gather=ops.GATHER_CONV3X3, conv=conv)
        h = ops.groupnorm(h, pk["g2"], pk["b2"], instances=g.b * g.t, eps=1e-5, silu=True)
        skip = x
        if "ws" in pk:
            # operand of the skip convolution: the bf16 rounding the caller already made, else the fp32 stream itself (register-staged
            # kernel, converts on load) or a bf16 copy of the fp16 stream
            xs = x_bf16 if x_bf16 is not None else (x if x.dtype == torch.float32 else ops.cast_bf16(x))
            if pk["skip_taps"] == 9:
                skip = ops.gemm(xs, pk["ws"], k=cin, taps=9, bias=pk["bs"], out_dtype=x.dtype, gather=ops.GATHER_CONV3X3, conv=conv)
            else:
                skip = ops.gemm(xs, pk["ws"], bias=pk["bs"], out_dtype=x.dtype)
        out = _gemm_gn(_gn_rows("clip" if self.use_temporal_conv else next_norm, g), h, pk["w2"], k=cout, taps=9, bias=pk["cb2"],
                       residual=skip, out_dtype=skip.dtype, gather=ops.GATHER_CONV3X3, conv=conv)
        if self.use_temporal_conv:
            out = self.temopral_conv.forward_rows(out, g, next_norm)
        return out


# =============================================================================================
# the UNet
# =============================================================================================
class _InputCache:
    """Small identity-keyed cache for step-invariant derived tensors (context K/V, packed masks, ...)."""

    def __init__(self, capacity=24):
        self.capacity = capacity
        self.items = OrderedDict()
        self.lock = threading.RLock()      # several host threads may run forwards of one module (clips in flight)

    @staticmethod
    def key(tag, t):
        # a tensor marked ``_ccv_static`` (the sampler's static conditioning copies) is refreshed in place together with
        # everything derived from it (the captured prologue graph recomputes those): its version counter is not part of the key
        return (tag, id(t), t.data_ptr(), tuple(t.shape), None if getattr(t, "_ccv_static", False) else t._version)

    def get(self, tag, t, make):
        """The derived value is produced on the calling thread's current stream; a hit from ANOTHER stream (two lanes sharing one
        conditioning tensor in eager mode) waits for the event recorded behind its producer kernels.  Inside a graph capture no event
        is recorded or waited for: a capture's entries belong to tensors of its own static tree (one set per stream)."""
        k = self.key(tag, t)
        with self.lock:
            hit = self.items.get(k)
            capturing = torch.cuda.is_current_stream_capturing()
            if hit is not None and hit[0] is t:
                self.items.move_to_end(k)
                if hit[2] is not None and not capturing and hit[3] != torch.cuda.current_stream(t.device).cuda_stream:
                    torch.cuda.current_stream(t.device).wait_event(hit[2])
                return hit[1]
            val = make()
            ev = None
            if t.is_cuda and not capturing:
                ev = torch.cuda.Event()
                ev.record(torch.cuda.current_stream(t.device))
            self.items[k] = (t, val, ev, torch.cuda.current_stream(t.device).cuda_stream if t.is_cuda else 0)  # keeping `t` alive pins id()/data_ptr()
            while len(self.items) > self.capacity:
                self.items.popitem(last=False)
            return val

    def forget(self, tensors):
        ids = {id(t) for t in tensors}
        with self.lock:
            for k in [k for k, v in self.items.items() if id(v[0]) in ids]:
                del self.items[k]

    def clear(self):
        with self.lock:
            self.items.clear()


class UNetModel(nn.Module, _Prepared):
    """lvdm 3D UNet (reference lvdm/modules/networks/openaimodel3d.py:281-624), same signature."""

    def __init__(self, in_channels, model_channels, out_channels, num_res_blocks, attention_resolutions,
                 dropout=0.0, channel_mult=(1, 2, 4, 8), conv_resample=True, dims=2, context_dim=None,
                 use_scale_shift_norm=False, resblock_updown=False, num_heads=-1, num_head_channels=-1,
                 transformer_depth=1, use_linear=False, use_checkpoint=False, temporal_conv=False,
                 tempspatial_aware=False, temporal_attention=True, use_relative_position=True,
                 use_causal_attention=False, temporal_length=None, use_fp16=False, addition_attention=False,
                 temporal_selfatt_only=True, image_cross_attention=False,
                 image_cross_attention_scale_learnable=False, default_fs=4, fs_condition=False):
        super().__init__()
        if num_heads == -1 and num_head_channels == -1:
            raise ValueError("Either num_heads or num_head_channels has to be set")
        if resblock_updown:
            raise NotImplementedError("resblock_updown is not used by any shipped config")
        self.in_channels, self.model_channels, self.out_channels = in_channels, model_channels, out_channels
        self.num_res_blocks = num_res_blocks
        self.attention_resolutions = attention_resolutions
        self.dropout, self.channel_mult, self.conv_resample = dropout, channel_mult, conv_resample
        self.temporal_attention = temporal_attention
        self.use_checkpoint = use_checkpoint
        self.dtype = torch.float16 if use_fp16 else torch.float32
        self.addition_attention = addition_attention
        self.temporal_length = temporal_length
        self.image_cross_attention = image_cross_attention
        self.image_cross_attention_scale_learnable = image_cross_attention_scale_learnable
        self.default_fs, self.fs_condition = default_fs, fs_condition
        self.context_dim = context_dim
        ted = model_channels * 4

        def mlp():
            return nn.Sequential(nn.Linear(model_channels, ted), nn.SiLU(), nn.Linear(ted, ted))

        self.time_embed = mlp()
        if fs_condition:
            self.fps_embedding = mlp()
            _zero(self.fps_embedding[-1])

        def heads_of(ch):
            return (ch // num_heads if num_head_channels == -1 else num_head_channels,
                    num_heads if num_head_channels == -1 else ch // num_head_channels)

        def res(cin, cout):
            return ResBlock(cin, ted, dropout, out_channels=cout, dims=dims, use_checkpoint=use_checkpoint,
                            use_scale_shift_norm=use_scale_shift_norm, tempspatial_aware=tempspatial_aware,
                            use_temporal_conv=temporal_conv)

        def attention_pair(ch, ds, is_out):
            d_head, n_heads = heads_of(ch)
            layers = [SpatialTransformer(ch, n_heads, d_head, depth=transformer_depth, context_dim=context_dim,
                                         use_linear=use_linear, use_checkpoint=use_checkpoint, video_length=temporal_length,
                                         image_cross_attention=image_cross_attention,
                                         image_cross_attention_scale_learnable=image_cross_attention_scale_learnable,
                                         is_output_block=is_out, ds=ds)]
            if temporal_attention:
                layers.append(TemporalTransformer(ch, n_heads, d_head, depth=transformer_depth, context_dim=context_dim,
                                                  use_linear=use_linear, use_checkpoint=use_checkpoint,
                                                  only_self_att=True, causal_attention=use_causal_attention,
                                                  relative_position=use_relative_position,
                                                  temporal_length=temporal_length, is_output_block=is_out, ds=ds))
            return layers

        self.input_blocks = nn.ModuleList([TimestepEmbedSequential(nn.Conv2d(in_channels, model_channels, 3, padding=1))])
        if addition_attention:
            self.init_attn = TimestepEmbedSequential(TemporalTransformer(
                model_channels, n_heads=8, d_head=num_head_channels, depth=transformer_depth, context_dim=context_dim,
                use_checkpoint=use_checkpoint, only_self_att=temporal_selfatt_only, causal_attention=False,
                relative_position=use_relative_position, temporal_length=temporal_length, ds=1))
        chans = [model_channels]
        ch, ds = model_channels, 1
        self.input_ds, self.output_ds = [ds], []
        for level, mult in enumerate(channel_mult):
            for _ in range(num_res_blocks):
                layers = [res(ch, mult * model_channels)]
                ch = mult * model_channels
                if ds in attention_resolutions:
                    layers += attention_pair(ch, ds, False)
                self.input_blocks.append(TimestepEmbedSequential(*layers))
                self.input_ds.append(ds)
                chans.append(ch)
            if level != len(channel_mult) - 1:
                self.input_blocks.append(TimestepEmbedSequential(Downsample(ch, conv_resample, dims=dims, out_channels=ch)))
                self.input_ds.append(ds)
                chans.append(ch)
                ds *= 2
        mid = [res(ch, ch)] + attention_pair(ch, ds, False) + [res(ch, ch)]
        self.middle_block = TimestepEmbedSequential(*mid)
        self.output_blocks = nn.ModuleList([])
        for level, mult in list(enumerate(channel_mult))[::-1]:
            for i in range(num_res_blocks + 1):
                layers = [res(ch + chans.pop(), mult * model_channels)]
                ch = model_channels * mult
                if ds in attention_resolutions:
                    layers += attention_pair(ch, ds, True)
                self.output_ds.append(ds)
                if level and i == num_res_blocks:
                    layers.append(Upsample(ch, conv_resample, dims=dims, out_channels=ch))
                    ds //= 2
                self.output_blocks.append(TimestepEmbedSequential(*layers))
        self.out = nn.Sequential(nn.GroupNorm(32, ch), nn.SiLU(),
                                 _zero(nn.Conv2d(model_channels, out_channels, 3, padding=1)))
        self.__dict__["out_norm_marker"] = object()     # stands for `self.out` (GroupNorm over a frame) in the forward's lookahead
        self._inputs = _InputCache()
        self._register_load_state_dict_pre_hook(lambda *a, **k: self.invalidate_all())

    # ---- camera conditioning (reference model/camcontexti2v.py:141-170, built natively) -----------
    def camera_blocks(self):
        """Temporal BasicTransformerBlocks that the reference patches: context_dim None and an inner
        width different from init_attn's (model/camcontexti2v.py:143)."""
        init_inner = self.init_attn[0].proj_in.out_channels if self.addition_attention else -1
        for name, m in self.named_modules():
            if isinstance(m, BasicTransformerBlock) and m.context_dim is None and m.dim != init_inner:
                yield name, m

    def enable_camera_conditioning(self, epipolar_config=None, pluker=True):
        """Add ``pluker_projection`` (zero-init Linear) and ``epipolar`` (Epipolar) to every temporal block."""
        for _, m in self.camera_blocks():
            dim = m.dim
            if pluker and not hasattr(m, "pluker_projection"):
                m.add_module("pluker_projection", _zero(nn.Linear(dim, dim)))
            if epipolar_config is not None and not hasattr(m, "epipolar"):
                m.add_module("epipolar", Epipolar(query_dim=dim, context_dim=dim, heads=m.attn1.heads,
                                                  **dict(epipolar_config)))
        ref = next(self.parameters())
        self.to(ref.device)
        self.invalidate_all()
        return self

    # ---- packing ------------------------------------------------------------------------------------
    def invalidate_all(self):
        """Drop every packed operand and derived input.  Runs on load_state_dict, .to() and enable_camera_conditioning;
        IN-PLACE parameter edits (``p.data.copy_``, an EMA swap) are invisible to the module and must be followed by a
        call to this method.  ``weights_generation`` lets holders of captured hipGraphs notice (camc2v_amd/sampler.py)."""
        for m in self.modules():
            if isinstance(m, _Prepared):
                m.invalidate()
        self._inputs.clear()
        self.__dict__["weights_generation"] = self.__dict__.get("weights_generation", 0) + 1

    def forget_inputs(self, tensors):
        """Drop the derived inputs (context K/V, Pluecker rows, packed masks) cached for the given input tensors."""
        self._inputs.forget(tensors)

    def inputs_only(self):
        """Context manager: forwards compute the step-invariant inputs of their arguments (filling the input cache) and
        return zeros.  The sampler captures this as the once-per-clip prologue graph."""
        tls = self.__dict__.setdefault("_inputs_only_tls", threading.local())     # per host thread: another lane's eager forward
                                                                                   # must not see this thread's capture mode
        class _Ctx:
            def __enter__(self_):
                tls.on = True

            def __exit__(self_, *exc):
                tls.on = False
        return _Ctx()

    def _apply(self, fn, *args, **kwargs):  # .to()/.cuda() move parameters: packed copies become stale
        out = super()._apply(fn, *args, **kwargs)
        self.invalidate_all()
        return out

    def _res_blocks(self):
        return [m for m in self.modules() if isinstance(m, ResBlock)]

    def _pack(self):
        pk = {}
        for name in ("time_embed",) + (("fps_embedding",) if self.fs_condition else ()):
            seq = getattr(self, name)
            pk[name] = (pack.pack_linear(seq[0].weight), _dev_f32(seq[0].bias),
                        pack.pack_linear(seq[2].weight), _dev_f32(seq[2].bias))
        # every ResBlock's emb_layers Linear fused into one [sum Cout, 4*mc] projection
        ws, bs, off = [], [], 0
        for rb in self._res_blocks():
            lin = rb.emb_layers[1]
            rb.emb_slice = (off, lin.out_features)
            off += lin.out_features
            ws.append(lin.weight)
            bs.append(lin.bias)
        pk["w_emb"] = pack.pack_linear(torch.cat(ws, 0))
        pk["b_emb"] = _dev_f32(torch.cat(bs, 0))
        conv_in = self.input_blocks[0][0]
        pk["cin_pad"] = (conv_in.in_channels + 63) // 64 * 64
        pk["w_in"], pk["b_in"] = pack.pack_conv3x3(conv_in.weight), _dev_f32(conv_in.bias)
        pk["gn_g"], pk["gn_b"] = _dev_f32(self.out[0].weight), _dev_f32(self.out[0].bias)
        pk["w_out"], pk["b_out"] = pack.pack_conv3x3(self.out[2].weight), pack.pad_bias(self.out[2].bias)
        return pk

    def prepare(self):
        """Pack every module's operands now (otherwise done lazily on the first forward)."""
        for m in self.modules():
            if isinstance(m, _Prepared):
                m._pk()
        return self

    # ---- step-invariant inputs ------------------------------------------------------------------------
    def _context_groups(self, context, t):
        """context: tensor [B, L, D] or list of tensors [b_i, L_i, D] (CFG halves with different lengths).
        Returns, per spatial transformer block, the list of (clip0, n_clips, kv_text, kv_img, img_tokens, per_frame)."""
        ctxs = context if isinstance(context, (list, tuple)) else [context]
        blocks = [blk for m in self.modules() if isinstance(m, SpatialTransformer) for blk in m.transformer_blocks]
        per_block = [[] for _ in blocks]
        c0 = 0
        for ctx in ctxs:
            def make(ctx=ctx):
                nb, L, D = ctx.shape
                cf = ctx.float()
                text = ops.cast_bf16(cf[:, :TEXT_LEN].contiguous().reshape(nb * TEXT_LEN, D))
                li = L - TEXT_LEN
                img = ops.cast_bf16(cf[:, TEXT_LEN:].contiguous().reshape(nb * li, D)) if li > 0 else None
                return [blk.attn2.project_context(text, img) for blk in blocks]
            kvs = self._inputs.get("ctx", ctx, make)
            nb, L = ctx.shape[0], ctx.shape[1]
            per_frame = (L == TEXT_LEN + t * 16)   # reference openaimodel3d.py:575 ("HARD CODE here")
            li = 16 if per_frame else L - TEXT_LEN
            for i, (kv_t, kv_i) in enumerate(kvs):
                per_block[i].append((c0, nb, kv_t, kv_i, li, per_frame))
            c0 += nb
        return c0, iter(per_block)

    def _camera_inputs(self, camera_condition, b, t, H, W):
        if camera_condition is None:
            return None
        feats = camera_condition.get("pluker_embedding_features")
        rows = None
        if feats is not None:
            rows = [self._inputs.get("pluker", f, lambda f=f: ops.nchw_to_rows_bf16(f)) for f in feats]
            for f in feats:
                if b % f.shape[0]:
                    raise CcvError("batch is not a multiple of the camera-feature batch")
        masks = {}
        packed = camera_condition.get("sample_locs_packed")
        if packed is not None:
            masks = {k: (v[0], v[1], v[0].shape[0], v[2] if len(v) > 2 else None, v[3] if len(v) > 3 else getattr(v, "wave_bits", None),
                         v[4] if len(v) > 4 else getattr(v, "group_order", None))
                     for k, v in packed.items()}
        elif camera_condition.get("sample_locs_dict") is not None:
            origin_h = getattr(self, "epipolar_origin_h", 8 * H)
            for k, m in camera_condition["sample_locs_dict"].items():
                # key k = origin_h // feature height (model/modules/epipolar.py:138): recover the feature-map shape
                hh = origin_h // k
                ww = (W * hh) // H if H else 0
                perm = (hh * ww, ww) if (hh > 0 and ww > 0 and ops.patch_order_ok(hh, ww) and t * hh * ww == m.shape[1]) else None
                mp = self._inputs.get(("mask", perm), m, lambda m=m, perm=perm: ops.pack_mask(m, perm))
                masks[k] = (mp[0], mp[1], m.shape[0], perm, mp.wave_bits, mp.group_order)
        return dict(rows=rows, masks=masks, add_type=camera_condition.get("add_type"))

    def _local_camera_inputs(self, cam, fc, H, W):
        """Frame-sharded forward: this rank's Pluecker rows and the mask rows of its queries (against all T*h*w keys)."""
        rows = None
        if cam["rows"] is not None:
            rows = []
            for lvl, r in enumerate(cam["rows"]):
                hw = max(H >> lvl, 1) * max(W >> lvl, 1)
                nb = r.shape[0] // (fc.T * hw)
                rows.append(self._inputs.get(("pluker_local", fc.f0, fc.t_loc), r, lambda r=r, nb=nb, hw=hw: fc.local_frames(r, nb, hw)))
        masks = {}
        for k, (bits, flags, nb, perm, wbits, order) in cam["masks"].items():
            L = bits.shape[1]
            hw = L // fc.T

            def make(bits=bits, L=L, hw=hw):
                loc = bits[:, fc.f0 * hw:(fc.f0 + fc.t_loc) * hw]
                shifts = torch.arange(32, device=bits.device, dtype=torch.int32)
                dense = ((loc.unsqueeze(-1) >> shifts) & 1).bool().reshape(loc.shape[0], loc.shape[1], -1)[:, :, :L]
                return ops.pack_mask(dense.contiguous())      # rows / bit columns stay in the order they were packed in (`perm`)
            mp = self._inputs.get(("mask_local", fc.f0, fc.t_loc), bits, make)
            masks[k] = (mp[0], mp[1], nb, perm, mp.wave_bits, mp.group_order)
        return dict(rows=rows, masks=masks, add_type=cam["add_type"])

    def enable_frame_sharding(self, group=None):
        """Shard the frames of every clip over the ranks of ``group`` (default: the world); see parallel.py.  The exchanges are
        torch.distributed calls on the compute stream: under RCCL ("nccl") the sampling step can be captured into a hipGraph
        (bench.py --frame-shard --shard-graph), under gloo (tests) it runs eagerly."""
        self.__dict__["frame_shard"] = parallel.FrameShard(group)
        return self

    def disable_frame_sharding(self):
        self.__dict__.pop("frame_shard", None)
        return self

    # ---- forward ----------------------------------------------------------------------------------------
    def forward(self, x, timesteps, context=None, features_adapter=None, fs=None, camera_condition=None,
                cfg_shared_input=False, **kwargs):
        """x [b, in_channels, t, h, w]; timesteps [b]; context [b, L, context_dim] (or a list of such tensors
        covering consecutive batch slices); fs [b]; camera_condition as built by
        model/camcontexti2v.py:565-570.  Unknown kwargs are ignored like the reference does.  Returns fp32
        [b, out_channels, t, h, w].

        cfg_shared_input: x / timesteps / fs hold ONE copy of the b0 samples while ``context`` (a list of two
        tensors) and the camera condition describe 2*b0 samples -- the conditional and unconditional halves of a
        classifier-free-guidance step.  Both halves see identical activations until the first cross-attention
        (input conv, init_attn, the first ResBlock: none of them reads the context or the camera), so that prefix
        runs once and is duplicated; the output then has 2*b0 samples."""
        if features_adapter is not None:
            raise NotImplementedError("features_adapter is not used on the generation path")
        if not x.is_cuda:
            raise CcvError("UNetModel.forward: the product path runs on the GPU only (see oracle/ for the CPU restatement)")
        shard = self.__dict__.get("frame_shard")
        if shard is not None and parallel.current() is None:
            # frames of every clip sharded over the ranks of a process group (parallel.py): run this rank's frames, gather the output
            with parallel.FrameCtx(shard, x.shape[2]) as fc:
                xl = x[:, :, fc.f0:fc.f0 + fc.t_loc].contiguous()
                yl = self.forward(xl, timesteps, context, None, fs, camera_condition, cfg_shared_input, **kwargs)
                return torch.cat(list(shard.all_gather(yl).unbind(0)), 2)
        if not ops.in_queue_counter_arena():   # one zeroed buffer for the queue counters of this forward's sparse attentions (per thread)
            with ops.queue_counter_arena(x.device):
                return self.forward(x, timesteps, context, None, fs, camera_condition, cfg_shared_input, **kwargs)
        pk = self._pk()
        b0, _, t, H, W = x.shape
        shared = bool(cfg_shared_input)
        b = 2 * b0 if shared else b0
        g = Geom(b0, t, H, W)          # geometry of the (possibly shared) prefix
        mc = self.model_channels
        fc = parallel.current()
        t_all = fc.T if fc is not None else t     # frames of the whole clip (t = this rank's share when sharded)

        nclips, ctx_iter = self._context_groups(context, t_all)
        if nclips != b:
            raise CcvError(f"context covers {nclips} samples, x has {b}")
        cam = self._camera_inputs(camera_condition, b, t_all, H, W)
        if fc is not None and cam is not None:
            cam = self._local_camera_inputs(cam, fc, H, W)
        if getattr(self.__dict__.get("_inputs_only_tls"), "on", False):
            return torch.zeros((b, self.out_channels, t, H, W), dtype=torch.float32, device=x.device)

        # -- timestep / frame-stride embedding -> one fused projection for all ResBlocks ------------------
        def mlp(name, values):
            w0, b0, w2, b2 = pk[name]
            hid = ops.gemm(ops.timestep_embedding(values, mc), w0, bias=b0, act=ops.ACT_SILU)
            return ops.gemm(hid, w2, bias=b2, out_f32=True)

        emb = mlp("time_embed", timesteps)
        emb_f = None
        if self.fs_condition:
            nt = timesteps.shape[0]
            if fs is None:
                fs = torch.full((nt,), self.default_fs, dtype=torch.long, device=x.device)
            elif fs.numel() == 1 and nt > 1:
                fs = fs.reshape(1).expand(nt)          # the reference broadcasts a single frame stride over the batch
            elif fs.numel() != nt:
                raise CcvError(f"fs has {fs.numel()} entries, timesteps has {nt}")
            emb_f = mlp("fps_embedding", fs)
        emb_all = ops.gemm(ops.add_silu_bf16(emb, emb_f), pk["w_emb"], bias=pk["b_emb"], out_f32=True)
        emb_state = [emb_all]          # run() reads the current one; replaced after the shared prefix

        origin_h = None
        if cam is not None:
            origin_h = getattr(self, "epipolar_origin_h", 8 * H)

        def cam_for(level, hh):
            if cam is None:
                return None
            rows = cam["rows"][level] if cam["rows"] is not None else None
            mask = cam["masks"].get(origin_h // hh) if cam["masks"] else None
            return dict(rows=rows, mask=mask, add_type=cam["add_type"])

        def norm_of(layer):     # the GroupNorm a layer starts with: over a frame, over a clip, or none
            if isinstance(layer, (ResBlock, SpatialTransformer)) or layer is self.out_norm_marker:
                return "frame"
            return "clip" if isinstance(layer, TemporalTransformer) else None

        def run(block, h, g, level, h16=None, after=None):
            """after: the layer that consumes the block's output directly (None: a concat or nothing)."""
            layers = list(block)
            for idx, layer in enumerate(layers):
                nn_ = norm_of(layers[idx + 1] if idx + 1 < len(layers) else after)
                if isinstance(layer, ResBlock):
                    h = layer.forward_rows(h, emb_state[0], g, h16, nn_)
                    h16 = None
                elif isinstance(layer, SpatialTransformer):
                    h = layer.forward_rows(h, g, [next(ctx_iter) for _ in layer.transformer_blocks], nn_)
                elif isinstance(layer, TemporalTransformer):
                    h = layer.forward_rows(h, g, cam_for(level, g.h), nn_)
                elif isinstance(layer, (Downsample, Upsample)):
                    h, g = layer.forward_rows(h, g, nn_)
                else:
                    raise CcvError(f"unexpected layer {type(layer).__name__} in a UNet block")
            return h, g

        # -- input conv on the zero-padded 64-channel fp32 rows -------------------------------------------
        rows = ops.pack_nchw_to_rows(x, None, ldo=pk["cin_pad"])
        h = ops.gemm(rows, pk["w_in"], k=pk["cin_pad"], taps=9, bias=pk["b_in"], out_dtype=STREAM,
                     gather=ops.GATHER_CONV3X3, conv=(H, W, H, W, 1, 0))
        if self.addition_attention:
            # never camera conditioned (modified_forwards.py:80-81); its output feeds the first ResBlock's norm
            h = self.init_attn[0].forward_rows(h, g, None, None if shared else "frame")
        def widen(rows_b0):   # [b0 rows] -> [cond rows | uncond rows]
            return torch.cat([rows_b0, rows_b0], 0)

        hs = [(h, g)]
        for i, block in enumerate(self.input_blocks):
            if i == 0:
                continue
            layers = list(block)
            if shared and g.b == b0:
                # the leading context-free layers of this block still run on the single copy
                n_free = 0
                while n_free < len(layers) and isinstance(layers[n_free], (ResBlock, Downsample)):
                    n_free += 1
                if n_free < len(layers):   # a transformer follows: duplicate here
                    h, g = run(layers[:n_free], h, g, int(math.log2(self.input_ds[i])))
                    h, g = widen(h), Geom(b, g.t, g.h, g.w)
                    hs = [(widen(hh), Geom(b, gg.t, gg.h, gg.w)) for hh, gg in hs]
                    emb_state[0] = widen(emb_all)
                    layers = layers[n_free:]
            nxt_block = self.input_blocks[i + 1] if i + 1 < len(self.input_blocks) else self.middle_block
            h, g = run(layers, h, g, int(math.log2(self.input_ds[i])), after=None if (shared and g.b == b0) else nxt_block[0])
            hs.append((h, g))
        if shared and g.b == b0:           # a UNet without any transformer in its encoder
            h, g = widen(h), Geom(b, g.t, g.h, g.w)
            hs = [(widen(hh), Geom(b, gg.t, gg.h, gg.w)) for hh, gg in hs]
            emb_state[0] = widen(emb_all)
        h, g = run(self.middle_block, h, g, -1)
        for i, block in enumerate(self.output_blocks):
            skip, _ = hs.pop()
            h, h16 = ops.concat_rows(h, skip, with_bf16=True)   # the bf16 copy feeds the ResBlock's 1x1 skip convolution
            h, g = run(block, h, g, int(math.log2(self.output_ds[i])), h16, after=self.out_norm_marker if i + 1 == len(self.output_blocks) else None)
        y = ops.groupnorm(h, pk["gn_g"], pk["gn_b"], instances=g.b * g.t, eps=1e-5, silu=True)
        y = ops.gemm(y, pk["w_out"], k=mc, taps=9, bias=pk["b_out"], out_f32=True, gather=ops.GATHER_CONV3X3,
                     conv=(g.h, g.w, g.h, g.w, 1, 0))
        return ops.unpack_rows_to_nchw(y, self.out_channels, b, t, H, W)
