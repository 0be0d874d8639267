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
