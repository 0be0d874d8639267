"""fp32 CPU restatement of the context-frame adaptor that produces the hot path's ``c_concat`` (SURVEY.md section 8,
row f1): ``MultiLatentEpipolarAdaptor`` over a flat reference-layout state_dict.
TEST INFRASTRUCTURE (see oracle/__init__.py).

Reference anchors (relative to /root/reference/CamContextI2V):
  MultiLatentEpipolarAdaptor._forward        model/modules/adaptors.py:138-182
  EpipolarCrossAttention.efficient_forward   model/modules/epipolar.py:75-102
  FeedForward (LayerNorm, Linear, GELU, Linear; no biases)   lvdm/modules/encoders/resampler.py:31-38
  timestep_embedding                          lvdm/models/utils_diffusion.py:8-28
  CrossNormalization.forward                  model/modules/utils.py:30-45 (its two call forms: model/camcontexti2v.py:354-364)
Configuration covered: the shipped one (configs/models/camcontexti2v_256.yaml:140-151): no Pluecker input, no context
positional encoding, timestep_embedding_type 'sinusoidal_embedded', no upscaler.

Parity is pinned by tests/golden/adaptor_small.npz (oracle/gen_golden_adaptor.py ran the reference's module) and
tests/golden/crossnorm_small.npz (oracle/gen_golden_crossnorm.py ran the reference's CrossNormalization).
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

# configs/models/camcontexti2v_256.yaml:140-151 (heads / dim_head / registers: constructor defaults, adaptors.py:38-60, epipolar.py:45)
FULL_CFG = dict(query_dim=512, num_queries=1024, video_length=16, embedding_dim=4, output_dim=4, depth=12,
                timestep_embedding_type="sinusoidal_embedded", use_plucker_embedding=False)
SMALL_CFG = dict(FULL_CFG, query_dim=128, num_queries=16, video_length=4, depth=2)
HEADS, DIM_HEAD, TIMESTEP_DIM = 8, 64, 32


def _lin(sd, p, x):
    return F.linear(_r(x), _r(sd[p + ".weight"]), sd.get(p + ".bias"))


def epipolar_cross_attention(sd, p, x, context, mask):
    """x [B, L1, C], context [B, L2, C'], mask bool [B, L1, L2] or None -> [B, L1, out_dim]."""
    B = x.shape[0]
    q = _lin(sd, p + ".to_q", x)
    reg = sd.get(p + ".register_tokens")
    if reg is not None:
        context = torch.cat([reg.repeat(B, 1, 1), context], 1)
        if mask is not None:
            mask = F.pad(mask, (reg.shape[1], 0), value=True)
    k, v = _lin(sd, p + ".to_k", context), _lin(sd, p + ".to_v", context)
    split = lambda t: t.reshape(B, t.shape[1], HEADS, DIM_HEAD).permute(0, 2, 1, 3)
    q, k, v = split(q), split(k), split(v)
    s = torch.einsum("bhid,bhjd->bhij", _r(q), _r(k)) * DIM_HEAD ** -0.5
    if mask is not None:
        s = s.masked_fill(~mask[:, None], float("-inf"))
    o = torch.einsum("bhij,bhjd->bhid", _r(torch.softmax(s, -1)), _r(v))
    o = o.permute(0, 2, 1, 3).reshape(B, -1, HEADS * DIM_HEAD)
    return _lin(sd, p + ".to_out.0", o)


def feed_forward(sd, p, x):
    h = F.layer_norm(x, (x.shape[-1],), sd[p + ".0.weight"], sd[p + ".0.bias"], 1e-5)
    return _lin(sd, p + ".3", F.gelu(_lin(sd, p + ".1", h)))


def adaptor_forward(sd, cfg, x, mask=None):
    """x [B, N*num_queries, embedding_dim] latents of the context frames, mask bool [B, T*num_queries, N*num_queries]
    -> [B, T*num_queries, output_dim]."""
    B = x.shape[0]
    T = cfg["video_length"]
    latents = sd["latents"].repeat(B, 1, 1)
    ctx = _lin(sd, "proj_in", x)
    for i in range(cfg["depth"]):
        latents = epipolar_cross_attention(sd, f"layers.{i}.0", latents, ctx, mask) + latents
        latents = feed_forward(sd, f"layers.{i}.1", latents) + latents
    t_emb = timestep_embedding(torch.arange(T), TIMESTEP_DIM)
    t_emb = _lin(sd, "timestep_embedding_func.2", F.silu(_lin(sd, "timestep_embedding_func.0", t_emb)))   # [T, C]
    per_frame = latents.shape[1] // T
    latents = latents + t_emb[None, :, None, :].expand(B, T, per_frame, -1).reshape(B, T * per_frame, -1)
    out = _lin(sd, "proj_out", latents)
    return F.layer_norm(out, (out.shape[-1],), sd["norm_out.weight"], sd["norm_out.bias"], 1e-5)


def cross_normalization(x, x_ref=None, dims=(-3, -2, -1)):
    """CrossNormalization.forward (model/modules/utils.py:30-45): x moved to x_ref's mean / unbiased std over ``dims``;
    the epsilon added to std_x is the literal 1e-5 of the reference."""
    x_ref = x if x_ref is None else x_ref
    mean_ref, std_ref = torch.mean(x_ref, dim=dims, keepdim=True), torch.std(x_ref, dim=dims, keepdim=True)
    mean_x, std_x = torch.mean(x, dim=dims, keepdim=True), torch.std(x, dim=dims, keepdim=True)
    return (x - mean_x) * (std_ref / (std_x + 1e-5)) + mean_ref


def cross_normalize_adaptor_output(lat, z_cond, T, H, W, mode="spatio_temporal"):
    """The two call forms after the adaptor (model/camcontexti2v.py:354-364): lat [B, (T H W), D], z_cond [B, D, H, W]
    -> [B, T, D, H, W]."""
    B, _, D = lat.shape
    if mode == "spatio_temporal":
        x = lat.reshape(B, T, H, W, D).permute(0, 1, 4, 2, 3)
        return cross_normalization(x, z_cond[:, None])
    x = cross_normalization(lat[:, None], z_cond)
    if x.dim() == 4:
        x = x.squeeze(1)
    return x.reshape(B, T, H, W, D).permute(0, 1, 4, 2, 3)
