"""Seeded inputs shared by oracle/gen_golden.py (which feeds them to the reference) and
the tests (which feed them to the oracle and to the HIP path).  TEST INFRASTRUCTURE.

The large random tensors (cross-attention contexts) are regenerated from the seed
instead of being committed; the fixture stores a float64 checksum of each so a
torch RNG change would be caught instead of silently comparing different inputs.
"""
import torch

SEED = 20230211

SMALL_CFG = dict(
    in_channels=8, out_channels=4, model_channels=64, attention_resolutions=[4, 2, 1],
    num_res_blocks=2, channel_mult=[1, 2, 4, 4], dropout=0.1, num_head_channels=64,
    transformer_depth=1, context_dim=1024, use_linear=True, use_checkpoint=False,
    temporal_conv=True, temporal_attention=True, temporal_selfatt_only=True,
    use_relative_position=False, use_causal_attention=False, temporal_length=16,
    addition_attention=True, image_cross_attention=True,
    image_cross_attention_scale_learnable=True, default_fs=3, fs_condition=True,
)
# configs/models/camcontexti2v_256.yaml:40-72
FULL_CFG = dict(SMALL_CFG, model_channels=320)
# Better conditioned than SMALL_CFG (whose 1x1 / 2x2 feature maps with 2 channels per GroupNorm group turn
# the normalisation into a sign function): used for the tight numerical parity checks.
MEDIUM_CFG = dict(SMALL_CFG, model_channels=128)


def medium_inputs():
    return small_inputs(b=1, T=16, hl=16, seed=SEED + 3, chans=(128, 256, 512, 512))


def small_inputs(b=2, T=16, hl=8, seed=SEED + 2, context_dim=1024, chans=(64, 128, 256, 256)):
    """Inputs of the reduced-width UNet fixtures (draw order is part of the contract)."""
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(b, 8, T, hl, hl, generator=g)
    t = torch.tensor([999, 439, 39, 679][:b], dtype=torch.long)
    fs = torch.tensor([8, 3, 5, 1][:b], dtype=torch.long)
    ctx_pf = torch.randn(b, 77 + 16 * T, context_dim, generator=g)    # per-frame image tokens (uncond style)
    ctx_rep = torch.randn(b, 77 + 256 * 3, context_dim, generator=g)   # (1+N)=3 frames x 256 tokens (cond style)
    feats = [torch.randn(b, chans[i], T, max(hl >> i, 1), max(hl >> i, 1), generator=g) * 0.1 for i in range(4)]
    c_concat = torch.randn(b, 4, T, hl, hl, generator=g)
    x_T = torch.randn(b, 4, T, hl, hl, generator=g)
    return dict(x=x, t=t, fs=fs, ctx_pf=ctx_pf, ctx_rep=ctx_rep, feats=feats, c_concat=c_concat, x_T=x_T)


def checksum(t):
    return float(t.double().sum().item())


# ---- per-op fixtures: seeded inputs of the reference's own modules inside the MEDIUM_CFG network -----------
OPS_NAMES = dict(
    gn="input_blocks.1.0.in_layers.0",                         # a15 GroupNorm32 (fp32 statistics)
    ln="input_blocks.1.1.transformer_blocks.0.norm1",          # a15 LayerNorm
    res="input_blocks.4.0",                                    # a6 ResBlock 128 -> 256 with 1x1 skip + its TemporalConvBlock
    tconv="input_blocks.1.0.temopral_conv",                    # a7 TemporalConvBlock alone (sic)
    down="input_blocks.3.0",                                   # a8 Downsample (3x3 stride 2)
    up="output_blocks.8.3",                                    # a8 Upsample (nearest 2x + 3x3)
    st="input_blocks.1.1",                                     # a9 SpatialTransformer
    xattn="input_blocks.1.1.transformer_blocks.0.attn2",       # a10 CrossAttention (text + gated image tokens)
    sattn="input_blocks.1.1.transformer_blocks.0.attn1",       # a10 self attention
    ff="input_blocks.1.1.transformer_blocks.0.ff",             # a14 GEGLU feed-forward
    tt="input_blocks.4.2",                                     # a11 / a12 TemporalTransformer with the camera block (C = 256, L = 256)
    epi1024="input_blocks.1.2.transformer_blocks.0.epipolar",  # a13 Epipolar, Lq = 16 * 8 * 8
    epi256="input_blocks.4.2.transformer_blocks.0.epipolar",   # a13 Epipolar, Lq = 16 * 4 * 4
)
OPS_PX = 64          # origin_h / origin_w of the epipolar modules: 8x8 latents, mask keys 8 (L = 1024) and 16 (L = 256)


def ops_inputs():
    """Seeded inputs of the per-op fixtures (draw order is part of the contract); b = 1 clip of T = 16 frames, 8x8 latents."""
    g = torch.Generator().manual_seed(SEED + 11)
    r = lambda *s: torch.randn(*s, generator=g)
    T = 16
    Tf = 4            # frames of the per-frame ops (norms, resampling, spatial transformer, feed-forward)
    return dict(
        T=T, Tf=Tf,
        gn_x=r(Tf, 128, 8, 8) * 1.5 + 0.3, ln_x=r(Tf, 64, 128),
        res_x=r(T, 128, 4, 4), res_emb=r(1, 512),
        tconv_x=r(1, 128, T, 4, 4),
        down_x=r(Tf, 128, 8, 8), up_x=r(Tf, 256, 4, 4),
        st_x=r(Tf, 128, 8, 8), st_ctx=r(Tf, 77 + 16, 1024),
        xattn_x=r(4, 64, 128), xattn_ctx77=r(4, 77, 1024), xattn_ctx93=r(4, 77 + 16, 1024), xattn_ctx845=r(4, 77 + 768, 1024),
        ff_x=r(Tf, 64, 128),
        tt_x=r(1, 256, T, 4, 4), tt_p=r(1, 256, T, 4, 4) * 0.1,
        epi1024_x=r(1, T, 128, 8, 8), epi256_x=r(1, T, 256, 4, 4),
    )
