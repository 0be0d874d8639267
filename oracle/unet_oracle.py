"""fp32 CPU restatement of the lvdm 3D-UNet forward with CamContextI2V camera
conditioning.  TEST INFRASTRUCTURE (see oracle/__init__.py).

Functional style: the network is a flat ``state_dict`` (reference checkpoint key
names, without the ``model.diffusion_model.`` prefix) plus the ``unet_config``
params of ``configs/models/camcontexti2v_256.yaml:40-72``.  No nn.Module tree is
built; the block topology is re-derived from the config exactly as the
reference constructor derives it.

Reference anchors (relative to /root/reference/CamContextI2V):
  UNetModel.__init__ / forward      lvdm/modules/networks/openaimodel3d.py:311-624
  camera-conditioned forward        model/modules/modified_forwards.py:29-131
  ResBlock / TemporalConvBlock      lvdm/modules/networks/openaimodel3d.py:109-279
  Spatial/TemporalTransformer       lvdm/modules/attention.py:256-428
  CrossAttention (einsum path)      lvdm/modules/attention.py:85-146
  temporal block with camera        model/modules/modified_forwards.py:505-536
  Epipolar / EpipolarCrossAttention model/modules/epipolar.py:43-157
  timestep_embedding                lvdm/models/utils_diffusion.py:8-28
"""
import math

import torch
import torch.nn.functional as F

TEXT_LEN = 77  # CrossAttention.text_context_len, lvdm/modules/attention.py:49

# Optional operand rounding.  None = exact fp32 restatement (the oracle proper).  With
# torch.bfloat16 every matmul / conv / attention operand is rounded to bf16 first (fp32
# accumulation), which is the arithmetic contract of the HIP path: tests use it to separate
# "bf16 operands" error from implementation error.
OPERAND_DTYPE = None


def _r(x):
    return x if OPERAND_DTYPE is None else x.to(OPERAND_DTYPE).float()


class operand_rounding:
    """with operand_rounding(torch.bfloat16): ...  (restores the previous setting on exit)"""

    def __init__(self, dtype):
        self.dtype = dtype

    def __enter__(self):
        global OPERAND_DTYPE
        self.prev, OPERAND_DTYPE = OPERAND_DTYPE, self.dtype

    def __exit__(self, *exc):
        global OPERAND_DTYPE
        OPERAND_DTYPE = self.prev


# --------------------------------------------------------------------------
# topology
# --------------------------------------------------------------------------
def unet_topology(cfg):
    """Re-derive the block list of UNetModel.__init__ (openaimodel3d.py:404-559).

    Returns dict(input=[...], middle=[...], output=[...], input_ds, output_ds)
    where every block is a list of layer tuples:
      ("conv_in", cin, cout) | ("res", cin, cout) | ("spatial", ch, heads, dhead)
      | ("temporal", ch, heads, dhead) | ("down", ch) | ("up", ch)
    """
    mc = cfg["model_channels"]
    mult = list(cfg.get("channel_mult", (1, 2, 4, 8)))
    nrb = cfg["num_res_blocks"]
    att = set(cfg["attention_resolutions"])
    nhc = cfg.get("num_head_channels", -1)
    nh = cfg.get("num_heads", -1)
    temporal = cfg.get("temporal_attention", True)

    def heads_of(ch):
        if nhc == -1:
            return nh, ch // nh
        return ch // nhc, nhc

    inp = [[("conv_in", cfg["in_channels"], mc)]]
    chans = [mc]
    ch, ds = mc, 1
    input_ds, output_ds = [ds], []
    for level, m in enumerate(mult):
        for _ in range(nrb):
            layers = [("res", ch, m * mc)]
            ch = m * mc
            if ds in att:
                h, d = heads_of(ch)
                layers.append(("spatial", ch, h, d))
                if temporal:
                    layers.append(("temporal", ch, h, d))
            inp.append(layers)
            input_ds.append(ds)
            chans.append(ch)
        if level != len(mult) - 1:
            inp.append([("down", ch)])
            input_ds.append(ds)
            chans.append(ch)
            ds *= 2
    h, d = heads_of(ch)
    mid = [("res", ch, ch), ("spatial", ch, h, d)]
    if temporal:
        mid.append(("temporal", ch, h, d))
    mid.append(("res", ch, ch))
    out = []
    for level, m in list(enumerate(mult))[::-1]:
        for i in range(nrb + 1):
            ich = chans.pop()
            layers = [("res", ch + ich, m * mc)]
            ch = m * mc
            if ds in att:
                h, d = heads_of(ch)
                layers.append(("spatial", ch, h, d))
                if temporal:
                    layers.append(("temporal", ch, h, d))
            output_ds.append(ds)
            if level and i == nrb:
                layers.append(("up", ch))
                ds //= 2
            out.append(layers)
    return dict(input=inp, middle=mid, output=out, input_ds=input_ds, output_ds=output_ds)


# --------------------------------------------------------------------------
# leaf ops
# --------------------------------------------------------------------------
def timestep_embedding(t, dim, max_period=10000):
    """utils_diffusion.py:8-28 (cos first, then sin)."""
    half = dim // 2
    freqs = torch.exp(-math.log(max_period) * torch.arange(half, dtype=torch.float32) / half)
    args = t[:, None].float() * freqs[None]
    emb = torch.cat([torch.cos(args), torch.sin(args)], dim=-1)
    if dim % 2:
        emb = torch.cat([emb, torch.zeros_like(emb[:, :1])], dim=-1)
    return emb


def _lin(sd, p, x, bias=True):
    return F.linear(_r(x), _r(sd[p + ".weight"]), sd[p + ".bias"] if bias and (p + ".bias") in sd else None)


def _gn(sd, p, x, eps):
    return F.group_norm(x.float(), 32, sd[p + ".weight"], sd[p + ".bias"], eps)


def _ln(sd, p, x):
    return F.layer_norm(x, (x.shape[-1],), sd[p + ".weight"], sd[p + ".bias"], 1e-5)


def _heads(t, h):
    b, n, c = t.shape
    return t.reshape(b, n, h, c // h).permute(0, 2, 1, 3)  # b h n d


def _attend(q, k, v, heads, mask=None):
    """softmax(q k^T / sqrt(d)) v, fp32.  q:[b,n,C] k,v:[b,m,C]; mask bool [b,n,m] (True = keep)."""
    qh, kh, vh = _heads(_r(q), heads), _heads(_r(k), heads), _heads(_r(v), heads)
    n, m = qh.shape[2], kh.shape[2]
    # softmax is per query row, so the scores may be formed a slab of queries at a time: the full-size epipolar
    # attention (16384 x 16388 scores x 5 heads) would otherwise hold three 5.4 GB temporaries at once
    rows = n if n * m * qh.shape[0] * heads <= (1 << 28) else max(64, (1 << 28) // (m * qh.shape[0] * heads))
    outs = []
    for i0 in range(0, n, rows):
        sim = torch.einsum("bhid,bhjd->bhij", qh[:, :, i0:i0 + rows], kh) * (qh.shape[-1] ** -0.5)
        if mask is not None:
            sim = sim.masked_fill(~mask[:, None, i0:i0 + rows], float("-inf"))
        outs.append(torch.einsum("bhij,bhjd->bhid", _r(sim.softmax(-1)), vh))
    out = outs[0] if len(outs) == 1 else torch.cat(outs, 2)
    b, h, n, d = out.shape
    return out.permute(0, 2, 1, 3).reshape(b, n, h * d)


def cross_attention(sd, p, x, context, heads, image_branch):
    """CrossAttention.forward, attention.py:85-146 (== efficient_forward :148-211).

    context None -> self attention.  With ``image_branch`` the context is split at
    token 77 into text and image tokens, attended separately and summed with the
    learnable gate tanh(alpha)+1 (image_cross_attention_scale == 1.0).
    """
    q = _lin(sd, p + ".to_q", x, bias=False)
    if context is None:
        out = _attend(q, _lin(sd, p + ".to_k", x, False), _lin(sd, p + ".to_v", x, False), heads)
    else:
        ctx_t = context[:, :TEXT_LEN]
        out = _attend(q, _lin(sd, p + ".to_k", ctx_t, False), _lin(sd, p + ".to_v", ctx_t, False), heads)
        if image_branch:
            ctx_i = context[:, TEXT_LEN:]
            out_ip = _attend(q, _lin(sd, p + ".to_k_ip", ctx_i, False), _lin(sd, p + ".to_v_ip", ctx_i, False), heads)
            gate = 1.0
            if (p + ".alpha") in sd:
                gate = torch.tanh(sd[p + ".alpha"]) + 1
            out = out + out_ip * gate
    return _lin(sd, p + ".to_out.0", out)


def feed_forward(sd, p, x):
    """FeedForward with GEGLU, attention.py:431-458 (erf GELU)."""
    a, g = _lin(sd, p + ".net.0.proj", x).chunk(2, dim=-1)
    return _lin(sd, p + ".net.2", a * F.gelu(g))


def epipolar_attention(sd, p, feats, mask, heads):
    """Epipolar.forward + EpipolarCrossAttention.efficient_forward, epipolar.py:75-157.

    feats [B, T*H*W, C] (token order t,h,w); mask bool [B, L, L] or None.  The
    learnable register tokens are prepended to the keys/values and are always
    visible (mask padded with True on the left, epipolar.py:94).
    """
    pa = p + ".epipolar_attn"
    q = _lin(sd, pa + ".to_q", feats, False)
    ctx = feats
    nreg = 0
    if (pa + ".register_tokens") in sd:
        reg = sd[pa + ".register_tokens"]
        nreg = reg.shape[1]
        ctx = torch.cat([reg.expand(feats.shape[0], -1, -1), feats], dim=1)
    k = _lin(sd, pa + ".to_k", ctx, False)
    v = _lin(sd, pa + ".to_v", ctx, False)
    if mask is not None and nreg:
        mask = F.pad(mask, (nreg, 0), value=True)
    return _lin(sd, pa + ".to_out.0", _attend(q, k, v, heads, mask))


# --------------------------------------------------------------------------
# blocks
# --------------------------------------------------------------------------
def temporal_conv_block(sd, p, x5):
    """TemporalConvBlock.forward, openaimodel3d.py:272-279.  x5: [b,c,t,h,w].
    nn.GroupNorm on the 5-D tensor => statistics over (c/32, t, h, w)."""
    h = x5
    for i in (1, 2, 3, 4):
        q = f"{p}.conv{i}"
        last = "2" if i == 1 else "3"  # conv1 has no Dropout slot (openaimodel3d.py:255-266)
        h = F.silu(_gn(sd, q + ".0", h, 1e-5))
        h = F.conv3d(_r(h), _r(sd[f"{q}.{last}.weight"]), sd[f"{q}.{last}.bias"], padding=(1, 0, 0))
    return x5 + h


def res_block(sd, p, x, emb, b, temporal_conv):
    """ResBlock._forward, openaimodel3d.py:210-236.  x: [(b t), c, h, w]."""
    h = F.silu(_gn(sd, p + ".in_layers.0", x, 1e-5))
    h = F.conv2d(_r(h), _r(sd[p + ".in_layers.2.weight"]), sd[p + ".in_layers.2.bias"], padding=1)
    e = _lin(sd, p + ".emb_layers.1", F.silu(emb))
    h = _r(h + e[:, :, None, None])
    h = F.silu(_gn(sd, p + ".out_layers.0", h, 1e-5))
    h = F.conv2d(_r(h), _r(sd[p + ".out_layers.3.weight"]), sd[p + ".out_layers.3.bias"], padding=1)
    if (p + ".skip_connection.weight") in sd:
        w = sd[p + ".skip_connection.weight"]
        x = F.conv2d(_r(x), _r(w), sd[p + ".skip_connection.bias"], padding=w.shape[-1] // 2)
    h = x + h
    if temporal_conv and (p + ".temopral_conv.conv1.0.weight") in sd:
        bt, c, hh, ww = h.shape
        h5 = h.reshape(b, bt // b, c, hh, ww).permute(0, 2, 1, 3, 4)
        h5 = temporal_conv_block(sd, p + ".temopral_conv", h5)
        h = h5.permute(0, 2, 1, 3, 4).reshape(bt, c, hh, ww)
    return h


def spatial_transformer(sd, p, x, context, heads, depth, image_branch):
    """SpatialTransformer.forward (use_linear), attention.py:304-320."""
    bt, c, hh, ww = x.shape
    s = _gn(sd, p + ".norm", x, 1e-6).permute(0, 2, 3, 1).reshape(bt, hh * ww, c)
    s = _lin(sd, p + ".proj_in", s)
    for i in range(depth):
        q = f"{p}.transformer_blocks.{i}"
        s = cross_attention(sd, q + ".attn1", _ln(sd, q + ".norm1", s), None, heads, False) + s
        s = cross_attention(sd, q + ".attn2", _ln(sd, q + ".norm2", s), context, heads, image_branch) + s
        s = feed_forward(sd, q + ".ff", _ln(sd, q + ".norm3", s)) + s
    s = _lin(sd, p + ".proj_out", s)
    return s.reshape(bt, hh, ww, c).permute(0, 3, 1, 2) + x


def temporal_transformer(sd, p, x5, heads, depth, cam):
    """TemporalTransformer.forward (only_self_att, no relative position),
    attention.py:381-428 / modified_forwards.py:401-450, with the camera-patched
    block body modified_forwards.py:505-536.

    x5: [b,c,t,h,w].  cam: None or dict(feat=[b,c,t,h,w] or None, mask=bool[b,L,L] or None,
    add_type=str).  Conv1d(k=1) proj_in/out (init_attn, use_linear False) is the same
    linear map as nn.Linear with the trailing kernel axis squeezed.
    """
    b, c, t, hh, ww = x5.shape
    s = _gn(sd, p + ".norm", x5, 1e-6)
    s = s.permute(0, 3, 4, 2, 1).reshape(b * hh * ww, t, c)  # (b h w) t c

    def proj(name, z):
        w = sd[f"{p}.{name}.weight"]
        return F.linear(_r(z), _r(w.reshape(w.shape[0], w.shape[1])), sd[f"{p}.{name}.bias"])

    s = proj("proj_in", s)
    inner = s.shape[-1]
    for i in range(depth):
        q = f"{p}.transformer_blocks.{i}"
        n = _ln(sd, q + ".norm1", s)
        a1 = cross_attention(sd, q + ".attn1", n, None, heads, False)
        patched = (q + ".pluker_projection.weight") in sd or (q + ".epipolar.epipolar_attn.to_q.weight") in sd
        if cam is not None and patched:
            z = torch.zeros_like(n)
            npf = n
            if cam.get("feat") is not None:
                pf = cam["feat"].permute(0, 3, 4, 2, 1).reshape(b * hh * ww, t, inner)
                npf = n + pf
                if (q + ".pluker_projection.weight") in sd:
                    z = z + _lin(sd, q + ".pluker_projection", npf)
            if (q + ".epipolar.epipolar_attn.to_q.weight") in sd:
                # '(b h w) f c -> b (f h w) c'
                feats = npf.reshape(b, hh * ww, t, inner).permute(0, 2, 1, 3).reshape(b, t * hh * ww, inner)
                eo = epipolar_attention(sd, q + ".epipolar", feats, cam.get("mask"), heads)
                z = z + eo.reshape(b, t, hh * ww, inner).permute(0, 2, 1, 3).reshape(b * hh * ww, t, inner)
            if cam.get("add_type") == "add_to_main_branch":
                s = z + a1 + s
            else:
                s = cross_attention(sd, q + ".attn1", n + z, None, heads, False) + s
        else:
            s = a1 + s
        s = cross_attention(sd, q + ".attn2", _ln(sd, q + ".norm2", s), None, heads, False) + s
        s = feed_forward(sd, q + ".ff", _ln(sd, q + ".norm3", s)) + s
    s = proj("proj_out", s)
    s = s.reshape(b, hh, ww, t, c).permute(0, 4, 3, 1, 2)
    return s + x5


# --------------------------------------------------------------------------
# whole network
# --------------------------------------------------------------------------
def unet_forward(sd, cfg, x, timesteps, context, fs=None, camera_condition=None, origin_h=None):
    """UNetModel.forward with the camera patch (modified_forwards.py:29-131).

    x [b, in_channels, t, h, w]; timesteps [b]; context [b, L, context_dim]; fs [b] long.
    camera_condition: None or dict with
        "pluker_embedding_features": list of tensors [b, C_i, t, h_i, w_i] (index log2(ds)) or None
        "sample_locs_dict": {origin_h // h_i: bool [b, t*h_i*w_i, t*h_i*w_i]} or None
        "add_type": str
    ``origin_h`` = Epipolar.origin_h (pixel height the mask keys refer to; defaults to 8*h).
    Returns eps [b, out_channels, t, h, w] (fp32).
    """
    x = x.float()
    b, _, t, H, W = x.shape
    mc = cfg["model_channels"]
    topo = unet_topology(cfg)
    depth = cfg.get("transformer_depth", 1)
    image_branch = bool(cfg.get("image_cross_attention", False))
    tconv = bool(cfg.get("temporal_conv", False))
    if origin_h is None:
        origin_h = 8 * H

    emb = _lin(sd, "time_embed.2", F.silu(_lin(sd, "time_embed.0", timestep_embedding(timesteps, mc))))
    # context: per-frame image tokens iff L == 77 + 16 t (modified_forwards.py:37-44)
    if context.shape[1] == TEXT_LEN + t * 16:
        ct = context[:, :TEXT_LEN].repeat_interleave(t, dim=0)
        ci = context[:, TEXT_LEN:].reshape(b * t, 16, context.shape[-1])
        ctx = torch.cat([ct, ci], dim=1)
    else:
        ctx = context.repeat_interleave(t, dim=0)
    emb = emb.repeat_interleave(t, dim=0)
    if cfg.get("fs_condition", False):
        if fs is None:
            fs = torch.full((b,), cfg.get("default_fs", 4), dtype=torch.long)
        fe = _lin(sd, "fps_embedding.2", F.silu(_lin(sd, "fps_embedding.0", timestep_embedding(fs, mc))))
        emb = emb + fe.repeat_interleave(t, dim=0)

    h = x.permute(0, 2, 1, 3, 4).reshape(b * t, x.shape[1], H, W)

    def cam_for(ds, hh, feature_id=None):
        if camera_condition is None:
            return None
        feats = camera_condition.get("pluker_embedding_features")
        feat = None
        if feats is not None:
            fid = int(math.log2(ds)) if feature_id is None else feature_id
            feat = feats[fid].float()
        masks = camera_condition.get("sample_locs_dict")
        mask = masks.get(origin_h // hh) if masks is not None else None
        return dict(feat=feat, mask=mask, add_type=camera_condition.get("add_type"))

    def run(layers, prefix, h, ds, feature_id=None):
        for j, layer in enumerate(layers):
            p = f"{prefix}.{j}"
            kind = layer[0]
            if kind == "conv_in":
                h = F.conv2d(_r(h), _r(sd[p + ".weight"]), sd[p + ".bias"], padding=1)
            elif kind == "res":
                h = res_block(sd, p, h, emb, b, tconv)
            elif kind == "spatial":
                h = spatial_transformer(sd, p, h, ctx, layer[2], depth, image_branch)
            elif kind == "temporal":
                bt, c, hh, ww = h.shape
                h5 = h.reshape(b, t, c, hh, ww).permute(0, 2, 1, 3, 4)
                h5 = temporal_transformer(sd, p, h5, layer[2], depth, cam_for(ds, hh, feature_id))
                h = h5.permute(0, 2, 1, 3, 4).reshape(bt, c, hh, ww)
            elif kind == "down":
                h = F.conv2d(_r(h), _r(sd[p + ".op.weight"]), sd[p + ".op.bias"], stride=2, padding=1)
            elif kind == "up":
                h = F.interpolate(h, scale_factor=2, mode="nearest")
                h = F.conv2d(_r(h), _r(sd[p + ".conv.weight"]), sd[p + ".conv.bias"], padding=1)
        return h

    hs = []
    for i, layers in enumerate(topo["input"]):
        h = run(layers, f"input_blocks.{i}", h, topo["input_ds"][i])
        if i == 0 and cfg.get("addition_attention", False):
            bt, c, hh, ww = h.shape
            h5 = h.reshape(b, t, c, hh, ww).permute(0, 2, 1, 3, 4)
            # init_attn: n_heads = 8 (openaimodel3d.py:389-402), never camera conditioned
            h5 = temporal_transformer(sd, "init_attn.0", h5, 8, depth, None)
            h = h5.permute(0, 2, 1, 3, 4).reshape(bt, c, hh, ww)
        hs.append(h)
    mid_ds = 2 ** (len(cfg.get("channel_mult", (1, 2, 4, 8))) - 1)
    h = run(topo["middle"], "middle_block", h, mid_ds, feature_id=-1)
    for i, layers in enumerate(topo["output"]):
        h = torch.cat([h, hs.pop()], dim=1)
        h = run(layers, f"output_blocks.{i}", h, topo["output_ds"][i])
    y = F.silu(_gn(sd, "out.0", h, 1e-5))
    y = F.conv2d(_r(y), _r(sd["out.2.weight"]), sd["out.2.bias"], padding=1)
    return y.reshape(b, t, y.shape[1], H, W).permute(0, 2, 1, 3, 4)


# --------------------------------------------------------------------------
# deterministic weights shared by the golden generator, the tests and bench
# --------------------------------------------------------------------------
def seeded_state_dict(manifest, seed=20230211, std=0.02):
    """key -> shape manifest  =>  fp32 tensors ~ N(0, std) (norm weights 1 + N(0, std)).

    Every tensor gets its own generator seeded by crc32(key) ^ seed, so the
    result does not depend on iteration order and zero-initialised reference
    tensors receive noise too.
    """
    import zlib

    out = {}
    for key, shape in manifest.items():
        g = torch.Generator().manual_seed((zlib.crc32(key.encode()) ^ seed) & 0x7FFFFFFF)
        w = torch.randn(tuple(shape), generator=g, dtype=torch.float32) * std
        is_norm_w = key.endswith(".weight") and len(shape) == 1
        if is_norm_w:
            w = w + 1.0
        if key.endswith(".alpha"):
            w = w * 10.0  # make the tanh gate visibly != 1
        out[key] = w
    return out
