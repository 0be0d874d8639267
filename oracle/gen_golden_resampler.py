#!/usr/bin/env python
"""Generate tests/golden/resampler_small.npz + manifests by RUNNING THE REFERENCE's Resampler (build
container only).  TEST INFRASTRUCTURE, companion of oracle/gen_golden.py (same stand-ins for cv2, pytorch_lightning,
torchvision.utils.make_grid).  Fixtures hold tensors and scalars only.

Usage:  python oracle/gen_golden_resampler.py [--out tests/golden]
"""
import argparse
import importlib.util
import json
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference/CamContextI2V"


def _load(name):
    spec = importlib.util.spec_from_file_location(f"_ccv_oracle_{name}", os.path.join(HERE, name + ".py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=os.path.join(os.path.dirname(HERE), "tests", "golden"))
    args = ap.parse_args()
    repo_root = os.path.dirname(HERE)
    assert all(os.path.abspath(p or os.getcwd()) != repo_root for p in sys.path), \
        "run as `python oracle/gen_golden_resampler.py`: the repo root on sys.path would shadow the reference's packages"
    gg = _load("gen_golden")
    gg._install_shims()
    ro = _load("resampler_oracle")
    seeded_state_dict = _load("unet_oracle").seeded_state_dict
    torch.set_grad_enabled(False)

    from lvdm.modules.encoders.resampler import Resampler
    assert sys.modules["lvdm.modules.encoders.resampler"].__file__.startswith(REF)

    full = Resampler(**ro.FULL_CFG)
    with open(os.path.join(args.out, "resampler_full_manifest.json"), "w") as f:
        json.dump(gg.manifest_of(full), f, indent=0, sort_keys=True)
    del full

    small = Resampler(**ro.SMALL_CFG).eval()
    man = gg.manifest_of(small)
    with open(os.path.join(args.out, "resampler_small_manifest.json"), "w") as f:
        json.dump(man, f, indent=0, sort_keys=True)
    sd = seeded_state_dict(man, gg.SEED + 31, std=0.05)
    small.load_state_dict(sd, strict=True)

    g = torch.Generator().manual_seed(gg.SEED + 32)
    x = torch.randn(2, 9, ro.SMALL_CFG["embedding_dim"], generator=g)      # 2 images x (1 class + 8 patch) tokens
    y = small(x)
    err = (ro.resampler_forward(sd, ro.SMALL_CFG, x) - y).abs().max().item()
    assert err < 2e-4 * y.abs().max().item(), err
    np.savez_compressed(os.path.join(args.out, "resampler_small.npz"), x=x.numpy(), y=y.numpy(), seed=np.int64(gg.SEED + 31),
                        std=np.float32(0.05))
    print(f"resampler_small: y absmax {y.abs().max().item():.3f}, oracle max abs err {err:.2e}; {len(man)} tensors")


if __name__ == "__main__":
    main()
