"""Image-token Resampler on the HIP kernels (the part without third-party weights): turns
the image encoder's tokens into the 16 x 16 image-context tokens of ``c_crossattn``, once per clip.

Same class, attribute and ``state_dict`` names as the reference (lvdm/modules/encoders/resampler.py:31-165).  Execution:
fp32 latent stream, bf16 GEMM operands; per layer LayerNorm of the image tokens and of the latents, one fused K|V
projection over [image tokens ; latents], ``ccv_attn_fwd`` (head dim 64), bias-free output projection and the
LayerNorm -> Linear -> GELU -> Linear feed-forward with the residual adds in the GEMM epilogues; the per-frame sinusoidal
embedding goes through ``proj_out`` once at pack time (it depends on parameters only) and enters the last GEMM as a
per-frame bias; the output LayerNorm returns fp32 (``ccv_layernorm_small``).
"""
import torch
import torch.nn as nn

from . import ops, pack
from .adaptor import FeedForward, _sinusoid
from .lib import CcvError
from .unet import _Prepared, _dev_f32


class PerceiverAttention(nn.Module):
    """Parameter container (reference resampler.py:52-67)."""

    def __init__(self, *, dim, dim_head=64, heads=8):
        super().__init__()
        self.scale, self.dim_head, self.heads = dim_head ** -0.5, dim_head, heads
        inner = dim_head * heads
        self.norm1, self.norm2 = nn.LayerNorm(dim), nn.LayerNorm(dim)
        self.to_q = nn.Linear(dim, inner, bias=False)
        self.to_kv = nn.Linear(dim, inner * 2, bias=False)
        self.to_out = nn.Linear(inner, dim, bias=False)


class Resampler(nn.Module, _Prepared):
    def __init__(self, dim=1024, depth=8, dim_head=64, heads=16, num_queries=8, embedding_dim=768, output_dim=1024, ff_mult=4,
                 video_length=None, use_timestep_emb=False):
        super().__init__()
        self.num_queries, self.video_length, self.use_timestep_emb, self.dim = num_queries, video_length, use_timestep_emb, dim
        self.heads, self.dim_head, self.embedding_dim, self.output_dim = heads, dim_head, embedding_dim, output_dim
        n_lat = num_queries * video_length if video_length is not None else num_queries
        self.latents = nn.Parameter(torch.randn(1, n_lat, dim) / dim ** 0.5)
        self.proj_in = nn.Linear(embedding_dim, dim)
        self.proj_out = nn.Linear(dim, output_dim)
        self.norm_out = nn.LayerNorm(output_dim)
        self.layers = nn.ModuleList([nn.ModuleList([PerceiverAttention(dim=dim, dim_head=dim_head, heads=heads),
                                                    FeedForward(dim=dim, mult=ff_mult)]) for _ in range(depth)])
        if use_timestep_emb:
            if video_length is None:
                raise NotImplementedError("use_timestep_emb needs video_length (as in the shipped configuration)")
            self.timestep_embedding_func = nn.Sequential(nn.Linear(dim, dim), nn.SiLU(), nn.Linear(dim, dim))
        self._register_load_state_dict_pre_hook(lambda *a, **k: self.invalidate())

    def _apply(self, fn, *args, **kwargs):
        out = super()._apply(fn, *args, **kwargs)
        self.invalidate()
        return out

    def _pack(self):
        if self.dim_head != 64:
            raise CcvError("the HIP attention kernel is specialised for head dim 64")
        if self.embedding_dim % 64 or self.dim % 64:
            raise CcvError("Resampler: embedding_dim and dim must be multiples of 64")
        dev = self.latents.device
        pk = dict(w_in=pack.pack_linear(self.proj_in.weight), b_in=_dev_f32(self.proj_in.bias),
                  w_out=pack.pack_linear(self.proj_out.weight), b_out=pack.pad_bias(self.proj_out.bias),
                  g_out=_dev_f32(self.norm_out.weight), bt_out=_dev_f32(self.norm_out.bias), layers=[])
        for attn, ff in self.layers:
            pk["layers"].append(dict(
                g1=_dev_f32(attn.norm1.weight), b1=_dev_f32(attn.norm1.bias), g2=_dev_f32(attn.norm2.weight), b2=_dev_f32(attn.norm2.bias),
                w_q=pack.pack_linear(attn.to_q.weight), w_kv=pack.pack_linear(attn.to_kv.weight), w_o=pack.pack_linear(attn.to_out.weight),
                g=_dev_f32(ff[0].weight), b=_dev_f32(ff[0].bias), w1=pack.pack_linear(ff[1].weight), w2=pack.pack_linear(ff[3].weight)))
        T = self.video_length if self.video_length is not None else 1
        te = torch.zeros((T, pk["w_out"].shape[0]), dtype=torch.float32, device=dev)
        if self.use_timestep_emb:   # parameters only: through proj_out once (resampler.py:152-160; proj_out is linear)
            f0, f2 = self.timestep_embedding_func[0], self.timestep_embedding_func[2]
            e = _sinusoid(T, self.dim, dev)
            e = torch.nn.functional.silu(e @ f0.weight.float().t() + f0.bias.float()) @ f2.weight.float().t() + f2.bias.float()
            te[:, :self.output_dim] = e @ self.proj_out.weight.float().t()
        pk["te"] = te.contiguous()
        return pk

    @torch.no_grad()
    def forward(self, x):
        """x [B, n, embedding_dim] image tokens -> fp32 [B, video_length*num_queries, output_dim]."""
        if not x.is_cuda:
            raise CcvError("Resampler.forward: the product path runs on the GPU only (see oracle/resampler_oracle.py)")
        pk = self._pk()
        B, n, _ = x.shape
        L, C, H = self.latents.shape[1], self.dim, self.heads
        inner = H * 64
        xs = ops.gemm(x.reshape(B * n, -1).float().contiguous(), pk["w_in"], bias=pk["b_in"], out_f32=True)     # [B n, C] fp32
        lat = self.latents.detach().float().expand(B, L, C).reshape(B * L, C).contiguous()
        Lk = n + L
        for lp in pk["layers"]:
            xn = ops.layernorm(xs, lp["g1"], lp["b1"])
            ln = ops.layernorm(lat, lp["g2"], lp["b2"])
            q = ops.gemm(ln, lp["w_q"])
            # keys / values over [image tokens ; latents] of each sample: the two row blocks are projected into one buffer
            kv = torch.empty((B * Lk, 2 * inner), dtype=ops.BF16, device=x.device).view(B, Lk, 2 * inner)
            kv[:, :n] = ops.gemm(xn, lp["w_kv"]).view(B, n, 2 * inner)
            kv[:, n:] = ops.gemm(ln, lp["w_kv"]).view(B, L, 2 * inner)
            kv = kv.view(B * Lk, 2 * inner)
            o = ops.attention(q, kv, kv[:, inner:], B=B, inner=1, H=H, Lq=L, Lk=Lk, q_str=(L * inner, 0, inner),
                              k_str=(Lk * 2 * inner, 0, 2 * inner), v_str=(Lk * 2 * inner, 0, 2 * inner))
            ops.gemm(o, lp["w_o"], residual=lat, out_f32=True, out=lat)
            h = ops.gemm(ops.layernorm(lat, lp["g"], lp["b"]), lp["w1"], act=ops.ACT_GELU)
            ops.gemm(h, lp["w2"], residual=lat, out_f32=True, out=lat)
        T = pk["te"].shape[0]
        y = ops.gemm(ops.cast_bf16(lat), pk["w_out"], bias=pk["b_out"], bias2=pk["te"].repeat(B, 1), ldb2=pk["te"].shape[1],
                     rows_per_batch=L // T, out_f32=True)
        return ops.layernorm_small(y, pk["g_out"], pk["bt_out"], eps=self.norm_out.eps).reshape(B, L, self.output_dim)


__all__ = ["Resampler", "PerceiverAttention", "FeedForward", "CcvError"]
