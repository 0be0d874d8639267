#!/usr/bin/env python
"""Generate tests/golden/pose_small.npz by RUNNING THE REFERENCE's ``ray_condition`` (model/base.py:112-174, called
unbound on a stub ``self``) (build container only).  TEST INFRASTRUCTURE, companion of oracle/gen_golden.py.  The pose
encoder itself cannot be run here (needs the absent diffusers package): see oracle/pose_oracle.py.

Usage:  python oracle/gen_golden_pose.py [--out tests/golden]
"""
import argparse
import importlib.util
import os
import sys
import types

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
        "run as `python oracle/gen_golden_pose.py`: the repo root on sys.path would shadow the reference's packages"
    gg = _load("gen_golden")
    gg._install_shims()
    po = _load("pose_oracle")
    geo = _load("geometry_oracle")
    torch.set_grad_enabled(False)
    from model.base import CameraControlLVDM
    assert sys.modules["model.base"].__file__.startswith(REF)

    B, V, H, W = 2, 5, 24, 40
    g = torch.Generator().manual_seed(gg.SEED + 41)
    K = torch.tensor([[30.0, 0, 20.0], [0, 28.0, 12.0], [0, 0, 1.0]]).repeat(B, V, 1, 1) + 0.5 * torch.rand(B, V, 3, 3, generator=g) * torch.tensor([[1.0, 0, 1], [0, 1, 1], [0, 0, 0]])
    w2c = geo.synthetic_trajectory(B, V)
    w2c[1, :, :3, 3] += 0.3 * torch.randn(V, 3, generator=g)
    c2w = geo.relative_c2w(w2c, torch.tensor([0, 2]))
    out = {}
    for mode in ("plucker", "ray"):
        stub = types.SimpleNamespace(camera_embedding=mode)
        # the method is decorated (no_grad, autocast(cuda, enabled=False)); both are no-ops for a CPU call
        y = CameraControlLVDM.ray_condition(stub, K, c2w, H, W, "cpu")
        err = (po.ray_condition(K, c2w, H, W, plucker=(mode == "plucker")) - y).abs().max().item()
        assert err < 1e-5, (mode, err)
        out[mode] = y.numpy()
    np.savez_compressed(os.path.join(args.out, "pose_small.npz"), K=K.numpy(), c2w=c2w.numpy(), plucker=out["plucker"], ray=out["ray"])
    print("pose_small: ray_condition", out["plucker"].shape, "oracle agrees")


if __name__ == "__main__":
    main()
