#!/usr/bin/env python
"""Generate tests/golden/adaptor_small.npz + manifests by RUNNING THE REFERENCE's MultiLatentEpipolarAdaptor (build
container only).  TEST INFRASTRUCTURE, companion of oracle/gen_golden.py (same stand-ins for cv2, pytorch_lightning,
torchvision.utils.make_grid).  Fixtures hold tensors and scalars only.

Usage:  python oracle/gen_golden_adaptor.py [--out tests/golden]
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
        "run as `python oracle/gen_golden_adaptor.py`: the repo root on sys.path would shadow the reference's packages"
    gg = _load("gen_golden")
    gg._install_shims()
    ao = _load("adaptor_oracle")
    seeded_state_dict = _load("unet_oracle").seeded_state_dict
    torch.set_grad_enabled(False)

    from model.modules.adaptors import MultiLatentEpipolarAdaptor
    assert sys.modules["model.modules.adaptors"].__file__.startswith(REF)

    full = MultiLatentEpipolarAdaptor(**ao.FULL_CFG)
    with open(os.path.join(args.out, "adaptor_full_manifest.json"), "w") as f:
        json.dump(gg.manifest_of(full), f, indent=0, sort_keys=True)
    del full

    small = MultiLatentEpipolarAdaptor(**ao.SMALL_CFG).eval()
    man = gg.manifest_of(small)
    with open(os.path.join(args.out, "adaptor_small_manifest.json"), "w") as f:
        json.dump(man, f, indent=0, sort_keys=True)
    sd = seeded_state_dict(man, gg.SEED + 21, std=0.05)
    small.load_state_dict(sd, strict=True)

    g = torch.Generator().manual_seed(gg.SEED + 22)
    B, N = 2, 2                                              # 2 clips, (cond frame + 1 extra context frame)
    Lq = ao.SMALL_CFG["video_length"] * ao.SMALL_CFG["num_queries"]
    Lk = N * ao.SMALL_CFG["num_queries"]
    x = torch.randn(B, Lk, 4, generator=g)
    mask = torch.rand(B, Lq, Lk, generator=g) < 0.3
    mask[:, 5] = False                                      # a query that only sees the register tokens
    y = small(x, mask)
    y_nomask = small(x, None)
    err = (ao.adaptor_forward(sd, ao.SMALL_CFG, x, mask) - y).abs().max().item()
    err2 = (ao.adaptor_forward(sd, ao.SMALL_CFG, x, None) - y_nomask).abs().max().item()
    assert err < 2e-4 * y.abs().max().item() and err2 < 2e-4 * y_nomask.abs().max().item(), (err, err2)
    np.savez_compressed(os.path.join(args.out, "adaptor_small.npz"), x=x.numpy(), mask=np.packbits(mask.numpy(), axis=-1, bitorder="little"),
                        y=y.numpy(), y_nomask=y_nomask.numpy(), seed=np.int64(gg.SEED + 21), std=np.float32(0.05))
    print(f"adaptor_small: y absmax {y.abs().max().item():.3f}, oracle max abs err {err:.2e} / {err2:.2e}; {len(man)} tensors")


if __name__ == "__main__":
    main()
