"""fp32 CPU restatement of the image-token Resampler that produces the image part of ``c_crossattn`` (SURVEY.md
section 8, row f4, the part that needs no third-party weights): flat reference-layout state_dict.
TEST INFRASTRUCTURE (see oracle/__init__.py).

Reference anchors (relative to /root/reference/CamContextI2V):
  Resampler.forward            lvdm/modules/encoders/resampler.py:145-165
  PerceiverAttention.forward   lvdm/modules/encoders/resampler.py:69-98
  FeedForward                  lvdm/modules/encoders/resampler.py:31-38
  timestep_embedding           lvdm/models/utils_diffusion.py:8-28
Parity is pinned by tests/golden/resampler_small.npz (oracle/gen_golden_resampler.py ran the reference's module).
"""
import torch
import torch.nn.functional as F

try:
    from .unet_oracle import _r, timestep_embedding
except ImportError:  # loaded by file path from the golden generator
    import importlib.util
    import os
    _spec = importlib.util.spec_from_file_location("_ccv_oracle_unet_oracle", os.path.join(os.path.dirname(os.path.abspath(__file__)), "unet_oracle.py"))
    _uo = importlib.util.module_from_spec(_spec)
    _spec.loader.exec_module(_uo)
    _r, timestep_embedding = _uo._r, _uo.timestep_embedding

# configs/models/camcontexti2v_256.yaml:109-121
FULL_CFG = dict(dim=1024, depth=4, dim_head=64, heads=12, num_queries=16, embedding_dim=1280, output_dim=1024, ff_mult=4,
                video_length=16, use_timestep_emb=True)
SMALL_CFG = dict(FULL_CFG, dim=128, depth=2, heads=2, num_queries=4, embedding_dim=64, output_dim=128, video_length=4)


def _lin(sd, p, x):
    return F.linear(_r(x), _r(sd[p + ".weight"]), sd.get(p + ".bias"))


def _ln(sd, p, x):
    return F.layer_norm(x, (x.shape[-1],), sd[p + ".weight"], sd[p + ".bias"], 1e-5)


def perceiver_attention(sd, p, x, latents, heads, dim_head):
    x, latents = _ln(sd, p + ".norm1", x), _ln(sd, p + ".norm2", latents)
    b, l, _ = latents.shape
    q = _lin(sd, p + ".to_q", latents)
    k, v = _lin(sd, p + ".to_kv", torch.cat([x, latents], -2)).chunk(2, dim=-1)
    split = lambda t: t.reshape(b, t.shape[1], heads, dim_head).transpose(1, 2)
    q, k, v = split(q), split(k), split(v)
    w = torch.softmax((_r(q) @ _r(k).transpose(-2, -1)) * dim_head ** -0.5, dim=-1)
    out = (_r(w) @ _r(v)).permute(0, 2, 1, 3).reshape(b, l, -1)
    return _lin(sd, p + ".to_out", out)


def feed_forward(sd, p, x):
    return _lin(sd, p + ".3", F.gelu(_lin(sd, p + ".1", _ln(sd, p + ".0", x))))


def resampler_forward(sd, cfg, x):
    """x [B, n, embedding_dim] image tokens -> [B, video_length*num_queries, output_dim]."""
    B = x.shape[0]
    latents = sd["latents"].repeat(B, 1, 1)
    x = _lin(sd, "proj_in", x)
    for i in range(cfg["depth"]):
        latents = perceiver_attention(sd, f"layers.{i}.0", x, latents, cfg["heads"], cfg["dim_head"]) + latents
        latents = feed_forward(sd, f"layers.{i}.1", latents) + latents
    if cfg.get("use_timestep_emb"):
        T = cfg["video_length"]
        t_emb = timestep_embedding(torch.arange(T), cfg["dim"])
        t_emb = _lin(sd, "timestep_embedding_func.2", F.silu(_lin(sd, "timestep_embedding_func.0", t_emb)))
        per = latents.shape[1] // T
        latents = latents + t_emb[None, :, None, :].expand(B, T, per, -1).reshape(B, T * per, -1)
    return _ln(sd, "norm_out", _lin(sd, "proj_out", latents))
