"""fp32 CPU restatement of the once-per-clip camera feeders in front of the hot path:
``ray_condition`` (Pluecker / ray embedding of the relative poses) and ``CameraPoseEncoder`` (the multi-scale pose
features the UNet's temporal blocks add to their LayerNorm output).  TEST INFRASTRUCTURE (see oracle/__init__.py).

Reference anchors (relative to /root/reference/CamContextI2V):
  ray_condition                    model/base.py:112-174
  CameraPoseEncoder.forward        model/modules/camera_pose_encoder.py:361-376
  ResnetBlock / Downsample         model/modules/camera_pose_encoder.py:219-290
  TemporalTransformerBlock         model/modules/camera_pose_encoder.py:15-78
  TemporalSelfAttention / PositionalEncoding   model/modules/camera_pose_encoder.py:81-158

PINNING.  ``ray_condition`` is pinned by tests/golden/pose_small.npz (oracle/gen_golden_pose.py ran the reference's
method).  ``CameraPoseEncoder`` is **parity unpinned**: the reference builds its attention and feed-forward from the
``diffusers`` package (``diffusers.models.attention_processor.Attention`` with the default ``AttnProcessor2_0``, and
``diffusers.models.attention.FeedForward(activation_fn="geglu")``; unpinned in requirements.txt:24), which is not
installed in the build container and has no copy under /root/reference, so the reference's module cannot be imported or
run here.  Their published arithmetic is restated below: Attention = to_q / to_k / to_v without bias, softmax(q k^T
d^-1/2) v per head, to_out[0] with bias (rescale_output_factor 1, no residual inside); FeedForward = GEGLU
(proj: dim -> 8 dim, value * gelu(gate), erf GELU) -> Linear(4 dim -> dim).
"""
import math

import torch
import torch.nn.functional as F

try:
    from .unet_oracle import _r
except ImportError:  # loaded by file path from the golden generator
    def _r(x):
        return x

# configs/models/camcontexti2v_256.yaml:124-138
FULL_CFG = dict(downscale_factor=8, channels=[320, 640, 1280, 1280], nums_rb=2, cin=384, ksize=1, sk=True, use_conv=False,
                compression_factor=1, temporal_attention_nhead=8, attention_block_types=["Temporal_Self"],
                temporal_position_encoding=True, temporal_position_encoding_max_len=16)
SMALL_CFG = dict(FULL_CFG, channels=[64, 128, 128, 128])


def ray_condition(K, c2w, H, W, plucker=True):
    """K [B,V,3,3] pixel intrinsics, c2w [B,V,4,4] -> [B, 6, V, H, W]: (o x d | d) Pluecker coordinates of the ray through
    every pixel centre, or (o | d) for camera_embedding == 'ray'."""
    B, V = K.shape[:2]
    j, i = torch.meshgrid(torch.linspace(0, H - 1, H, dtype=c2w.dtype), torch.linspace(0, W - 1, W, dtype=c2w.dtype), indexing="ij")
    i = i.reshape(1, 1, H * W).expand(B, V, H * W) + 0.5
    j = j.reshape(1, 1, H * W).expand(B, V, H * W) + 0.5
    fx, fy, cx, cy = (K[..., 0, 0].unsqueeze(-1), K[..., 1, 1].unsqueeze(-1), K[..., 0, 2].unsqueeze(-1), K[..., 1, 2].unsqueeze(-1))
    zs = torch.ones_like(i)
    d = torch.stack(((i - cx) / fx * zs, (j - cy) / fy * zs, zs), dim=-1)
    d = d / d.norm(dim=-1, keepdim=True)
    rays_d = d @ c2w[..., :3, :3].transpose(-1, -2)
    rays_o = c2w[..., :3, 3][:, :, None].expand_as(rays_d)
    first = torch.cross(rays_o, rays_d, dim=-1) if plucker else rays_o
    enc = torch.cat([first, rays_d], dim=-1).reshape(B, V, H, W, 6)
    return enc.permute(0, 4, 1, 2, 3)


def _conv(sd, p, x, padding=0):
    return F.conv2d(_r(x), _r(sd[p + ".weight"]), sd[p + ".bias"], padding=padding)


def _lin(sd, p, x):
    return F.linear(_r(x), _r(sd[p + ".weight"]), sd.get(p + ".bias"))


def resnet_block(sd, p, x, down, ksize):
    if down:
        x = F.avg_pool2d(x, 2, 2)                        # use_conv False
    if (p + ".in_conv.weight") in sd:
        x = _conv(sd, p + ".in_conv", x, ksize // 2)
    h = _conv(sd, p + ".block2", F.relu(_conv(sd, p + ".block1", x, 1)), ksize // 2)
    return h + (_conv(sd, p + ".skep", x, ksize // 2) if (p + ".skep.weight") in sd else x)


def positional_encoding(d_model, max_len):
    position = torch.arange(max_len).unsqueeze(1)
    div = torch.exp(torch.arange(0, d_model, 2) * (-math.log(10000.0) / d_model))
    pe = torch.zeros(max_len, d_model)
    pe[:, 0::2] = torch.sin(position * div)
    pe[:, 1::2] = torch.cos(position * div)
    return pe


def temporal_block(sd, p, x, heads, pe):
    """x [(b h w), f, c]: LayerNorm -> (+ positional encoding) self-attention over the frames -> + x ->
    LayerNorm -> GEGLU feed-forward -> + x."""
    n, f, c = x.shape
    h = F.layer_norm(x, (c,), sd[p + ".norms.0.weight"], sd[p + ".norms.0.bias"], 1e-5)
    if pe is not None:
        h = h + pe[None, :f]
    a = p + ".attention_blocks.0"
    d = c // heads
    split = lambda t: t.reshape(n, f, heads, d).transpose(1, 2)
    q, k, v = split(_lin(sd, a + ".to_q", h)), split(_lin(sd, a + ".to_k", h)), split(_lin(sd, a + ".to_v", h))
    w = torch.softmax((_r(q) @ _r(k).transpose(-2, -1)) * d ** -0.5, dim=-1)
    o = (_r(w) @ _r(v)).transpose(1, 2).reshape(n, f, c)
    x = _lin(sd, a + ".to_out.0", o) + x
    h = F.layer_norm(x, (c,), sd[p + ".ff_norm.weight"], sd[p + ".ff_norm.bias"], 1e-5)
    val, gate = _lin(sd, p + ".ff.net.0.proj", h).chunk(2, dim=-1)
    return _lin(sd, p + ".ff.net.2", val * F.gelu(gate)) + x


def pose_encoder_forward(sd, cfg, x):
    """x [b, 6, f, H, W] Pluecker embedding -> list of 4 feature maps [(b f), C_i, H/8/2^i, W/8/2^i]."""
    b, _, f, H, W = x.shape
    x = F.pixel_unshuffle(x.permute(0, 2, 1, 3, 4).reshape(b * f, -1, H, W), cfg["downscale_factor"])
    x = _conv(sd, "encoder_conv_in", x, 1)
    feats = []
    for i, c in enumerate(cfg["channels"]):
        pe = positional_encoding(c, cfg["temporal_position_encoding_max_len"]) if cfg["temporal_position_encoding"] else None
        for j in range(cfg["nums_rb"]):
            x = resnet_block(sd, f"encoder_down_conv_blocks.{i}.{j}", x, down=(j == 0 and i != 0), ksize=cfg["ksize"])
            hh, ww = x.shape[-2:]
            t = x.reshape(b, f, c, hh * ww).permute(0, 3, 1, 2).reshape(b * hh * ww, f, c)
            t = temporal_block(sd, f"encoder_down_attention_blocks.{i}.{j}", t, cfg["temporal_attention_nhead"], pe)
            x = t.reshape(b, hh * ww, f, c).permute(0, 2, 3, 1).reshape(b * f, c, hh, ww)
        feats.append(x)
    return feats


def pose_encoder_manifest(cfg):
    """key -> shape of the reference module's state_dict for ``cfg`` (ksize 1, sk True, use_conv False), derived from the
    constructor (camera_pose_encoder.py:295-352) and the diffusers sub-module layouts named in the header."""
    man = {"encoder_conv_in.weight": [cfg["channels"][0], cfg["cin"], 3, 3], "encoder_conv_in.bias": [cfg["channels"][0]]}
    ks = cfg["ksize"]
    for i, c in enumerate(cfg["channels"]):
        for j in range(cfg["nums_rb"]):
            cin = cfg["channels"][i - 1] if (j == 0 and i != 0) else c
            p = f"encoder_down_conv_blocks.{i}.{j}"
            if cin != c or not cfg["sk"]:
                man[p + ".in_conv.weight"], man[p + ".in_conv.bias"] = [c, cin, ks, ks], [c]
            man[p + ".block1.weight"], man[p + ".block1.bias"] = [c, c, 3, 3], [c]
            man[p + ".block2.weight"], man[p + ".block2.bias"] = [c, c, ks, ks], [c]
            if not cfg["sk"]:
                man[p + ".skep.weight"], man[p + ".skep.bias"] = [c, cin, ks, ks], [c]
            a = f"encoder_down_attention_blocks.{i}.{j}"
            for nm in ("to_q", "to_k", "to_v"):
                man[f"{a}.attention_blocks.0.{nm}.weight"] = [c, c]
            man[f"{a}.attention_blocks.0.to_out.0.weight"], man[f"{a}.attention_blocks.0.to_out.0.bias"] = [c, c], [c]
            if cfg["temporal_position_encoding"]:
                man[f"{a}.attention_blocks.0.pos_encoder.pe"] = [1, cfg["temporal_position_encoding_max_len"], c]
            for nm in ("norms.0", "ff_norm"):
                man[f"{a}.{nm}.weight"], man[f"{a}.{nm}.bias"] = [c], [c]
            man[f"{a}.ff.net.0.proj.weight"], man[f"{a}.ff.net.0.proj.bias"] = [8 * c, c], [8 * c]
            man[f"{a}.ff.net.2.weight"], man[f"{a}.ff.net.2.bias"] = [c, 4 * c], [c]
    return man
