"""Context-frame adaptor on the HIP kernels: ``MultiLatentEpipolarAdaptor`` (SURVEY.md section 8, row f1 -- runs once
per clip and produces the ``c_concat`` latents the DDIM path consumes).

Same class, attribute and ``state_dict`` names as the reference (model/modules/adaptors.py:36-182, with
``EpipolarCrossAttention`` model/modules/epipolar.py:43-102 and the resampler ``FeedForward``
lvdm/modules/encoders/resampler.py:31-38), for the shipped configuration (configs/models/camcontexti2v_256.yaml:140-151:
no Pluecker input, no context positional encoding, 'sinusoidal_embedded' frame embedding, no upscaler); other
constructor options raise.

Execution (token-major rows, fp32 latent stream, bf16 GEMM operands): per layer one fused K|V projection of the context
tokens, the Q projection of the 16 x 1024 learnable latents, masked attention with the layer's register tokens through
``ccv_attn_fwd`` (bit-packed mask, per-wave sparse kernel), output projection and the LayerNorm -> Linear -> GELU ->
Linear feed-forward with the residual adds in the GEMM epilogues.  The per-frame sinusoidal embedding goes through the
output projection once at pack time and rides into the last GEMM as a per-frame bias; the 4-channel output LayerNorm is
``ccv_layernorm_small``.
"""
import torch
import torch.nn as nn

from . import ops, pack
from .lib import CcvError
from .unet import EpipolarCrossAttention, _Prepared, _dev_f32


def FeedForward(dim, mult=4):
    """Parameter container with the reference's Sequential layout (resampler.py:31-38)."""
    inner = int(dim * mult)
    return nn.Sequential(nn.LayerNorm(dim), nn.Linear(dim, inner, bias=False), nn.GELU(), nn.Linear(inner, dim, bias=False))


def _sinusoid(n, dim, device, max_period=10000.0):
    """lvdm/models/utils_diffusion.py:8-28 for the integer frame indices 0..n-1 (cos first, then sin)."""
    half = dim // 2
    freqs = torch.exp(-torch.log(torch.tensor(max_period)) * torch.arange(half, dtype=torch.float32) / half).to(device)
    args = torch.arange(n, dtype=torch.float32, device=device)[:, None] * freqs[None]
    emb = torch.cat([torch.cos(args), torch.sin(args)], -1)
    return torch.cat([emb, torch.zeros_like(emb[:, :1])], -1) if dim % 2 else emb


class CrossNormalization(nn.Module):
    """``model.modules.utils.CrossNormalization`` (reference model/modules/utils.py:5-45) on ``ccv_cross_norm``: x is moved to
    the mean and (unbiased) standard deviation x_ref has over ``dims``.  Built for dims = (-3, -2, -1), the only value the
    reference constructs (model/camcontexti2v.py:80); the leading dimensions of x_ref must equal those of x or be 1 from
    some position on (the broadcasts of model/camcontexti2v.py:357-360)."""

    def __init__(self, dims, eps=1e-5):
        super().__init__()
        if tuple(dims) != (-3, -2, -1):
            raise NotImplementedError("CrossNormalization: only dims = (-3, -2, -1) is built")
        self.dims, self.eps = tuple(dims), eps
        self._enabled = True

    def enable(self):
        self._enabled = True

    def disable(self):
        self._enabled = False

    @torch.no_grad()
    def forward(self, x, x_ref=None):
        if not self._enabled:
            return x
        if not x.is_cuda:
            raise CcvError("CrossNormalization.forward: the product path runs on the GPU only (see oracle/adaptor_oracle.py)")
        x_ref = x if x_ref is None else x_ref
        if x.dim() < 3 or x_ref.dim() < 3:
            raise CcvError("CrossNormalization: inputs need at least three dimensions")
        lead, lead_ref = list(x.shape[:-3]), list(x_ref.shape[:-3])
        lead_ref = [1] * (len(lead) - len(lead_ref)) + lead_ref
        if len(lead_ref) != len(lead):
            raise CcvError(f"CrossNormalization: reference {tuple(x_ref.shape)} has more leading dimensions than x {tuple(x.shape)}")
        j = 0
        while j < len(lead) and lead_ref[j] == lead[j]:
            j += 1
        if any(d != 1 for d in lead_ref[j:]):
            raise CcvError(f"CrossNormalization: cannot broadcast reference {tuple(x_ref.shape)} over x {tuple(x.shape)}")
        n, per_ref = 1, 1
        for d in lead:
            n *= d
        for d in lead[j:]:
            per_ref *= d
        # the reference adds the literal 1e-5 to std_x, not self.eps (utils.py:43)
        return ops.cross_norm(x, x_ref, max(n, 1), per_ref, eps=1e-5).reshape(x.shape)


class MultiLatentEpipolarAdaptor(nn.Module, _Prepared):
    def __init__(self, query_dim=512, depth=8, dim_head=64, heads=16, num_queries=1024, output_queries=None,
                 embedding_dim=768, output_dim=1024, ff_mult=4, num_register_tokens=2, use_mask=True, checkpoint=False,
                 video_length=None, use_plucker_embedding=False, allow_plucker_embedding_param=False,
                 context_positional_encoding=False, context_positional_encoding_dim=None,
                 timestep_embedding_type="none", timestep_embedding_dim=32, plucker_embedding_dim=320,
                 plucker_input_strategy="add"):
        super().__init__()
        if (use_plucker_embedding or allow_plucker_embedding_param or context_positional_encoding
                or (output_queries is not None and output_queries != num_queries) or plucker_input_strategy != "add"):
            raise NotImplementedError("only the shipped adaptor configuration is built (no Pluecker input, no context "
                                      "positional encoding, no upscaler)")
        if timestep_embedding_type not in ("none", "sinusoidal_embedded"):
            raise NotImplementedError(f"timestep_embedding_type {timestep_embedding_type!r} is not used by the shipped config")
        self.num_queries = num_queries
        self.video_length = video_length if video_length is not None else 16
        self.use_mask, self.checkpoint, self.use_plucker_embedding = use_mask, checkpoint, False
        self.timestep_embedding_type = timestep_embedding_type
        self.timestep_embedding_dim_in = timestep_embedding_dim
        self.query_dim, self.embedding_dim, self.output_dim = query_dim, embedding_dim, output_dim
        self.timestep_embedding_func = None
        if timestep_embedding_type == "sinusoidal_embedded":
            self.timestep_embedding_func = nn.Sequential(nn.Linear(timestep_embedding_dim, query_dim), nn.SiLU(),
                                                         nn.Linear(query_dim, query_dim))
        n_lat = num_queries * video_length if video_length is not None else num_queries
        self.latents = nn.Parameter(torch.randn(1, n_lat, query_dim) / query_dim ** 0.5)
        self.proj_in = nn.Linear(embedding_dim, query_dim)
        self.proj_out = nn.Linear(query_dim, output_dim)
        self.norm_out = nn.LayerNorm(output_dim)
        self.plucker_in = None
        self.layers = nn.ModuleList([nn.ModuleList([
            EpipolarCrossAttention(query_dim=query_dim, context_dim=query_dim, out_dim=query_dim, num_register_tokens=num_register_tokens),
            FeedForward(dim=query_dim, mult=ff_mult)]) for _ in range(depth)])
        self._register_load_state_dict_pre_hook(lambda *a, **k: self.invalidate())

    def _apply(self, fn, *args, **kwargs):
        out = super()._apply(fn, *args, **kwargs)
        self.invalidate()
        return out

    def _pack(self):
        dev = self.latents.device
        pk = dict(in_pad=(self.embedding_dim + 63) // 64 * 64,
                  w_in=pack.pack_linear(self.proj_in.weight), b_in=_dev_f32(self.proj_in.bias),
                  w_out=pack.pack_linear(self.proj_out.weight), b_out=pack.pad_bias(self.proj_out.bias),
                  g_out=_dev_f32(self.norm_out.weight), bt_out=_dev_f32(self.norm_out.bias), layers=[])
        for attn, ff in self.layers:
            if attn.dim_head != 64:
                raise CcvError("the HIP attention kernel is specialised for head dim 64")
            lp = dict(w_q=pack.pack_linear(attn.to_q.weight), w_kv=pack.pack_linear(torch.cat([attn.to_k.weight, attn.to_v.weight], 0)),
                      w_o=pack.pack_linear(attn.to_out[0].weight), b_o=_dev_f32(attn.to_out[0].bias),
                      g=_dev_f32(ff[0].weight), b=_dev_f32(ff[0].bias), w1=pack.pack_linear(ff[1].weight), w2=pack.pack_linear(ff[3].weight))
            if attn.num_register_tokens > 0:   # register K/V do not depend on the input (epipolar.py:86-90)
                reg = attn.register_tokens[0].float()
                lp["kreg"] = (reg @ attn.to_k.weight.float().t()).to(ops.BF16).contiguous()
                lp["vreg"] = (reg @ attn.to_v.weight.float().t()).to(ops.BF16).contiguous()
            pk["layers"].append(lp)
        # per-frame embedding: parameters only -> through the output projection once (adaptors.py:170-178; proj_out is
        # linear, so proj_out(latents + e_f) = proj_out(latents) + W e_f)
        n_out = pk["w_out"].shape[0]
        te = torch.zeros((self.video_length, n_out), dtype=torch.float32, device=dev)
        if self.timestep_embedding_func is not None:
            f0, f2 = self.timestep_embedding_func[0], self.timestep_embedding_func[2]
            e = _sinusoid(self.video_length, self.timestep_embedding_dim_in, dev)
            e = torch.nn.functional.silu(e @ f0.weight.float().t() + f0.bias.float()) @ f2.weight.float().t() + f2.bias.float()
            te[:, :self.output_dim] = e @ self.proj_out.weight.float().t()
        pk["te"] = te.contiguous()
        return pk

    @torch.no_grad()
    def forward(self, x, mask=None, plucker_embedding_features=None):
        """x [B, N*num_queries, embedding_dim] (latents of the conditioning + context frames), mask bool
        [B, T*num_queries, N*num_queries] (True = visible) -> fp32 [B, T*num_queries, output_dim]."""
        if plucker_embedding_features is not None:
            raise NotImplementedError("use_plucker_embedding is off in the shipped adaptor configuration")
        if not x.is_cuda:
            raise CcvError("MultiLatentEpipolarAdaptor.forward: the product path runs on the GPU only (see oracle/adaptor_oracle.py)")
        pk = self._pk()
        B, Lk, _ = x.shape
        Lq, C, H = self.latents.shape[1], self.query_dim, 8
        inner = H * 64
        xp = torch.zeros((B * Lk, pk["in_pad"]), dtype=torch.float32, device=x.device)
        xp[:, :self.embedding_dim] = x.reshape(B * Lk, -1).float()
        ctx = ops.gemm(xp, pk["w_in"], bias=pk["b_in"])                                   # [B Lk, C] bf16
        lat = self.latents.detach().float().expand(B, Lq, C).reshape(B * Lq, C).contiguous()
        mk = {}
        if mask is not None and self.use_mask:
            if isinstance(mask, ops.MaskPack):        # already bit-packed (ops.epipolar_mask_bits / ops.pack_mask)
                mp = mask
                if mp[0].shape[0] != B or mp[0].shape[1] != Lq or mp[0].shape[2] * 32 < Lk:
                    raise CcvError(f"adaptor: packed mask {tuple(mp[0].shape)} does not cover [{B}, {Lq}, {Lk}]")
            else:
                if mask.dtype != torch.bool or tuple(mask.shape) != (B, Lq, Lk):
                    raise CcvError(f"adaptor mask must be bool [{B}, {Lq}, {Lk}]")
                mp = ops.pack_mask(mask)
            mk = dict(mask_bits=mp[0], tile_flags=mp[1], mask_nb=B, wave_bits=mp.wave_bits, group_order=mp.group_order)
        for lp in pk["layers"]:
            q = ops.gemm(ops.cast_bf16(lat), lp["w_q"])
            kv = ops.gemm(ctx, lp["w_kv"])
            o = ops.attention(q, kv, kv[:, inner:], B=B, inner=1, H=H, Lq=Lq, Lk=Lk, q_str=(Lq * inner, 0, inner),
                              k_str=(Lk * 2 * inner, 0, 2 * inner), v_str=(Lk * 2 * inner, 0, 2 * inner),
                              kreg=lp.get("kreg"), vreg=lp.get("vreg"), **mk)
            ops.gemm(o, lp["w_o"], bias=lp["b_o"], residual=lat, out_f32=True, out=lat)
            h = ops.gemm(ops.layernorm(lat, lp["g"], lp["b"]), lp["w1"], act=ops.ACT_GELU)
            ops.gemm(h, lp["w2"], residual=lat, out_f32=True, out=lat)
        T = self.video_length
        y = ops.gemm(ops.cast_bf16(lat), pk["w_out"], bias=pk["b_out"], bias2=pk["te"].repeat(B, 1), ldb2=pk["te"].shape[1],
                     rows_per_batch=Lq // T, out_f32=True)
        return ops.layernorm_small(y, pk["g_out"], pk["bt_out"], eps=self.norm_out.eps).reshape(B, Lq, self.output_dim)


__all__ = ["MultiLatentEpipolarAdaptor", "CrossNormalization", "FeedForward", "EpipolarCrossAttention", "CcvError"]
