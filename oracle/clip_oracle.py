"""CPU restatement (test infrastructure) of the OpenCLIP ViT-H/14 towers as the reference's two wrappers call them:
``FrozenOpenCLIPEmbedder.encode_with_transformer`` (lvdm/modules/encoders/condition.py:212-230) and
``FrozenOpenCLIPImageEmbedderV2.encode_with_vision_transformer`` (:344-372).

PARITY UNPINNED.  The towers are third-party code, ``open_clip_torch==2.22.0`` (reference requirements.txt:5), absent from this
image and from /root/reference, and the reference holds no fixtures for them; ``kornia`` (the resize in ``preprocess``) is absent
too.  What is restated is open_clip's published ``ViT-H-14`` definition: ``ResidualAttentionBlock`` = ``x + attn(ln_1(x))``,
``x + mlp(ln_2(x))`` with ``nn.MultiheadAttention(width, heads)`` (packed ``in_proj``, scale 1/sqrt(head width), additive causal
mask in the text tower) and ``mlp = c_fc -> GELU(erf) -> c_proj``; text: token embedding + positional embedding, ``ln_final``;
vision: ``conv1`` (patch, stride = patch, no bias), class token, positional embedding, ``ln_pre``.  tests/test_clip_cpu.py checks
this restatement against a tower assembled from torch's own ``nn.MultiheadAttention`` / ``nn.LayerNorm`` modules (the modules
open_clip is built from); the GPU tests hold the HIP path to this oracle.

Functions take a flat state dict with open_clip's names (the reference checkpoint's ``cond_stage_model.model.`` /
``embedder.model.`` slices with the prefix removed).  fp32, plain torch ops; only tests/ may import this file.
"""
import math

import torch
import torch.nn.functional as F


def _ln(x, sd, name, eps=1e-5):
    return F.layer_norm(x, (x.shape[-1],), sd[name + ".weight"].float(), sd[name + ".bias"].float(), eps)


def _block(x, sd, pre, heads, attn_mask=None):
    """x [B, L, C]."""
    B, L, C = x.shape
    d = C // heads
    n = _ln(x, sd, pre + "ln_1")
    qkv = n @ sd[pre + "attn.in_proj_weight"].float().t() + sd[pre + "attn.in_proj_bias"].float()
    q, k, v = (t.reshape(B, L, heads, d).permute(0, 2, 1, 3) for t in qkv.chunk(3, dim=-1))
    s = (q @ k.transpose(-1, -2)) / math.sqrt(d)
    if attn_mask is not None:
        s = s + attn_mask
    o = (s.softmax(-1) @ v).permute(0, 2, 1, 3).reshape(B, L, C)
    x = x + o @ sd[pre + "attn.out_proj.weight"].float().t() + sd[pre + "attn.out_proj.bias"].float()
    n = _ln(x, sd, pre + "ln_2")
    h = F.gelu(n @ sd[pre + "mlp.c_fc.weight"].float().t() + sd[pre + "mlp.c_fc.bias"].float())
    return x + h @ sd[pre + "mlp.c_proj.weight"].float().t() + sd[pre + "mlp.c_proj.bias"].float()


def _n_blocks(sd, pre):
    return 1 + max(int(k[len(pre):].split(".")[0]) for k in sd if k.startswith(pre))


def text_tokens(sd, tokens, heads, layer_idx=1):
    """token ids [B, L] -> [B, L, width]: all but the last ``layer_idx`` blocks (``layer: penultimate`` = 1), then ln_final."""
    x = sd["token_embedding.weight"].float()[tokens] + sd["positional_embedding"].float()[:tokens.shape[1]]
    L = tokens.shape[1]
    mask = torch.full((L, L), float("-inf")).triu_(1)
    pre = "transformer.resblocks."
    for i in range(_n_blocks(sd, pre) - layer_idx):
        x = _block(x, sd, f"{pre}{i}.", heads, mask)
    return _ln(x, sd, "ln_final")


def vision_tokens(sd, img, heads, patch):
    """img [B, 3, H, W] already resized and normalised -> [B, (H/patch)(W/patch) + 1, width] (no ln_post / proj)."""
    x = F.conv2d(img.float(), sd["visual.conv1.weight"].float(), stride=patch)
    B, C = x.shape[:2]
    x = x.reshape(B, C, -1).permute(0, 2, 1)
    x = torch.cat([sd["visual.class_embedding"].float().expand(B, 1, C), x], 1) + sd["visual.positional_embedding"].float()
    x = _ln(x, sd, "visual.ln_pre")
    pre = "visual.transformer.resblocks."
    for i in range(_n_blocks(sd, pre)):
        x = _block(x, sd, f"{pre}{i}.", heads)
    return x


def preprocess(x, size=224, antialias=True):
    """condition.py:327-335 with kornia's documented resize (Gaussian blur in front of a bicubic, align_corners=True
    interpolation when downscaling): an independent fp64 statement of camc2v_amd.clip.clip_preprocess."""
    x = x.double()
    H, W = x.shape[-2:]
    fy, fx = H / size, W / size
    if antialias and max(fy, fx) > 1:
        for axis, f in ((2, fy), (3, fx)):
            s = max((f - 1) / 2, 0.001)
            k = int(max(4 * s, 3))
            k += 1 - k % 2
            t = torch.arange(k, dtype=torch.float64) - k // 2
            g = torch.exp(-t * t / (2 * s * s))
            g = g / g.sum()
            idx = torch.arange(-(k // 2), x.shape[axis] + k // 2).abs()
            n = x.shape[axis]
            idx = torch.where(idx >= n, 2 * (n - 1) - idx, idx)                         # reflect
            xp = x.index_select(axis, idx)
            x = sum(g[j] * xp.narrow(axis, j, n) for j in range(k))
    x = F.interpolate(x, size=(size, size), mode="bicubic", align_corners=True)
    x = (x + 1.0) / 2.0
    mean = torch.tensor([0.48145466, 0.4578275, 0.40821073], dtype=torch.float64).view(1, 3, 1, 1)
    std = torch.tensor([0.26862954, 0.26130258, 0.27577711], dtype=torch.float64).view(1, 3, 1, 1)
    return ((x - mean) / std).float()
