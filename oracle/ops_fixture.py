"""Per-op fixtures (tests/golden/ops_medium.npz, written by oracle/gen_golden_ops.py from the reference's own modules): loader +
the oracle's restatement of every op on the same seeded inputs.  TEST INFRASTRUCTURE.

Reference anchors (relative to /root/reference/CamContextI2V): lvdm/basics.py:78-91 (GroupNormSpecific), lvdm/modules/networks/
openaimodel3d.py:51-106 (Down / Upsample), :109-236 (ResBlock), :239-279 (TemporalConvBlock), lvdm/modules/attention.py:85-253
(CrossAttention, BasicTransformerBlock), :256-320 (SpatialTransformer), :431-458 (FeedForward), model/modules/modified_forwards.py:
401-450, 505-536 (camera-patched TemporalTransformer), model/modules/epipolar.py:43-157 (Epipolar)."""
import json
import os

import numpy as np
import torch
import torch.nn.functional as F

from . import geometry_oracle, unet_oracle as uo
from .golden_inputs import MEDIUM_CFG, OPS_NAMES, OPS_PX, SEED, checksum, ops_inputs


def load(golden_dir):
    """-> (fixture arrays, seeded state dict, seeded inputs, {8, 16: bool mask} from the fixture's F through the geometry oracle)."""
    fx = dict(np.load(os.path.join(golden_dir, "ops_medium.npz")))
    man = json.load(open(os.path.join(golden_dir, "unet_medium_manifest.json")))
    sd = uo.seeded_state_dict(man, SEED)
    inp = ops_inputs()
    for k, v in inp.items():
        if torch.is_tensor(v):
            assert abs(checksum(v) - float(fx["checksum_" + k])) <= 1e-6 * max(1.0, abs(float(fx["checksum_" + k]))), f"seeded input {k} changed"
    Fm = torch.from_numpy(fx["F64"])
    masks = {d: geometry_oracle.epipolar_mask(Fm, OPS_PX // d, OPS_PX // d, d) for d in (8, 16)}
    for d in (8, 16):
        assert int(masks[d].sum()) == int(fx[f"mask_d{d}_popcount"]), f"epipolar mask d={d} differs from the reference's"
    return fx, sd, inp, masks


def oracle_outputs(sd, inp, masks):
    """The oracle's per-op functions on the fixture inputs; keys as in ops_medium.npz."""
    N, T = OPS_NAMES, inp["T"]
    heads = lambda c: c // MEDIUM_CFG["num_head_channels"]
    out = {}
    out["gn_y"] = uo._gn(sd, N["gn"], inp["gn_x"], 1e-5)
    out["ln_y"] = uo._ln(sd, N["ln"], inp["ln_x"])
    out["res_y"] = uo.res_block(sd, N["res"], inp["res_x"], inp["res_emb"].repeat_interleave(T, 0), 1, True)
    out["tconv_y"] = uo.temporal_conv_block(sd, N["tconv"], inp["tconv_x"])
    out["down_y"] = F.conv2d(inp["down_x"], sd[N["down"] + ".op.weight"], sd[N["down"] + ".op.bias"], stride=2, padding=1)
    up = F.interpolate(inp["up_x"], scale_factor=2, mode="nearest")
    out["up_y"] = F.conv2d(up, sd[N["up"] + ".conv.weight"], sd[N["up"] + ".conv.bias"], padding=1)
    out["st_y"] = uo.spatial_transformer(sd, N["st"], inp["st_x"], inp["st_ctx"], heads(128), 1, True)
    out["sattn_y"] = uo.cross_attention(sd, N["sattn"], inp["xattn_x"], None, heads(128), False)
    for L in (77, 93, 845):
        out[f"xattn_y{L}"] = uo.cross_attention(sd, N["xattn"], inp["xattn_x"], inp[f"xattn_ctx{L}"], heads(128), True)
    out["ff_y"] = uo.feed_forward(sd, N["ff"], inp["ff_x"])
    for add_type in ("add_to_main_branch", "add_into_temporal_attn"):
        cam = dict(feat=inp["tt_p"], mask=masks[16], add_type=add_type)
        out[f"tt_y_{add_type}"] = uo.temporal_transformer(sd, N["tt"], inp["tt_x"], heads(256), 1, cam)

    def epi(name, x, mask):   # features B T C H W -> tokens (t h w) -> '(B H W) T C'
        B, T1, C, H, W = x.shape
        tok = x.permute(0, 1, 3, 4, 2).reshape(B, T1 * H * W, C)
        o = uo.epipolar_attention(sd, N[name], tok, mask, heads(C))
        return o.reshape(B, T1, H * W, C).permute(0, 2, 1, 3).reshape(B * H * W, T1, C)
    out["epi1024_y"] = epi("epi1024", inp["epi1024_x"], masks[8])
    out["epi256_y"] = epi("epi256", inp["epi256_x"], masks[16])
    out["epi256_y_nomask"] = epi("epi256", inp["epi256_x"], None)
    return out
