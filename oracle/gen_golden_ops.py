#!/usr/bin/env python
"""Per-op fixtures by RUNNING THE REFERENCE'S OWN MODULES (build container only).

TEST INFRASTRUCTURE.  Builds the reference UNetModel at MEDIUM_CFG with the camera patch of model/camcontexti2v.py:111-170 replayed
(oracle/gen_golden.py: build_reference_unet), loads the seeded weights every test regenerates (unet_oracle.seeded_state_dict), and
calls single modules of that network -- ResBlock, TemporalConvBlock, Down / Upsample, SpatialTransformer, CrossAttention (self, text
only, text + 16 and text + 768 image tokens), FeedForward, the camera-patched TemporalTransformer (both add_types, with Pluecker
features and the reference's own epipolar mask), Epipolar at Lq = 1024 and 256 (register tokens), GroupNormSpecific, LayerNorm -- on
the seeded inputs of golden_inputs.ops_inputs().  Only the OUTPUTS (fp32) and input checksums are stored: tests/golden/ops_medium.npz.

Usage:  python oracle/gen_golden_ops.py [--out tests/golden]      (never on the GPU box: it has no /root/reference)
"""
import argparse
import importlib.util
import json
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))


def _load(name):
    spec = importlib.util.spec_from_file_location(f"_ccv_oracle_{name}", os.path.join(HERE, name + ".py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=os.path.join(os.path.dirname(HERE), "tests", "golden"))
    args = ap.parse_args()
    gg = _load("gen_golden")
    gi = _load("golden_inputs")
    uo = _load("unet_oracle")
    geo = _load("geometry_oracle")
    gg._install_shims()
    torch.set_grad_enabled(False)
    repo_root = os.path.dirname(HERE)
    assert all(os.path.abspath(p or os.getcwd()) != repo_root for p in sys.path), \
        "run as `python oracle/gen_golden_ops.py`: the repo root on sys.path would shadow the reference's packages"

    unet = gg.build_reference_unet(gi.MEDIUM_CFG, camera=True, origin=gi.OPS_PX)
    man = json.load(open(os.path.join(args.out, "unet_medium_manifest.json")))
    assert gg.manifest_of(unet) == man, "medium manifest changed"
    unet.load_state_dict(uo.seeded_state_dict(man, gi.SEED), strict=True)
    mods = dict(unet.named_modules())
    m = {k: mods[v] for k, v in gi.OPS_NAMES.items()}
    for k, v in gi.OPS_NAMES.items():
        print(f"{k:8s} {v:60s} {type(mods[v]).__name__}")
    inp = gi.ops_inputs()
    T = inp["T"]

    # the reference's own epipolar masks of the synthetic trajectory at 64x64 px (keys 8 and 16 are read)
    w2c = geo.synthetic_trajectory(1, T)
    K = torch.tensor([[gi.OPS_PX / 2, 0, gi.OPS_PX / 2], [0, gi.OPS_PX / 2, gi.OPS_PX / 2], [0, 0, 1.0]]).repeat(1, T, 1, 1)
    pn = torch.randn(1, T, T, 3, 1, generator=torch.Generator().manual_seed(gi.SEED))
    _, F, masks = gg.geometry_via_reference(K, w2c, torch.zeros(1, dtype=torch.long), gi.OPS_PX, gi.OPS_PX, pn)

    out = {}
    out["gn_y"] = m["gn"](inp["gn_x"])
    out["ln_y"] = m["ln"](inp["ln_x"])
    out["res_y"] = m["res"](inp["res_x"], inp["res_emb"].repeat_interleave(T, 0), batch_size=1)
    out["tconv_y"] = m["tconv"](inp["tconv_x"])
    out["down_y"] = m["down"](inp["down_x"])
    out["up_y"] = m["up"](inp["up_x"])
    out["st_y"] = m["st"](inp["st_x"], inp["st_ctx"])
    out["sattn_y"] = m["sattn"](inp["xattn_x"])
    for L in (77, 93, 845):
        out[f"xattn_y{L}"] = m["xattn"](inp["xattn_x"], context=inp[f"xattn_ctx{L}"])
    out["ff_y"] = m["ff"](inp["ff_x"])
    for add_type in ("add_to_main_branch", "add_into_temporal_attn"):
        cam = dict(pluker_embedding_features=inp["tt_p"], sample_locs_dict=masks, cond_frame_index=torch.zeros(1, dtype=torch.long),
                   add_type=add_type, h=4, w=4)
        out[f"tt_y_{add_type}"] = m["tt"](inp["tt_x"], None, camera_condition=cam)
    out["epi1024_y"] = m["epi1024"](inp["epi1024_x"], sample_locs_dict=masks)
    out["epi256_y"] = m["epi256"](inp["epi256_x"], sample_locs_dict=masks)
    out["epi256_y_nomask"] = m["epi256"](inp["epi256_x"], sample_locs_dict=None)

    arrays = {k: v.float().numpy() for k, v in out.items()}
    arrays["F64"] = F.numpy()
    arrays["perturb_noise"] = pn.numpy()
    for k, v in inp.items():
        if torch.is_tensor(v):
            arrays["checksum_" + k] = np.array(gi.checksum(v))
    for d in (8, 16):
        arrays[f"mask_d{d}_popcount"] = np.array(int(masks[d].sum()))
    path = os.path.join(args.out, "ops_medium.npz")
    np.savez_compressed(path, **arrays)
    print("wrote", path, f"{os.path.getsize(path) / 1e6:.2f} MB;", {k: tuple(v.shape) for k, v in out.items()})


if __name__ == "__main__":
    main()
