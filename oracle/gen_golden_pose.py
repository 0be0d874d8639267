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
    # target-frame x context-frame epipolar mask: the sub-functions compute_conditional_epipolar_mask
    # (model/camcontexti2v.py:493-521) calls, composed in its order (the method itself reads a data batch through
    # super().get_input and cannot be called on a stub)
    import model.camcontexti2v as cc
    from einops import rearrange, repeat
    stub = types.SimpleNamespace(epipolar_config=types.SimpleNamespace(
        apply_epipolar_soft_mask=False, epipolar_hybrid_attention=False, epipolar_hybrid_attention_v2=False,
        only_self_pixel_on_current_frame=False, current_frame_as_register_token=False))
    Tt, px = 4, 64
    Kc = torch.tensor([[40.0, 0, 32.0], [0, 44.0, 30.0], [0, 0, 1.0]]).repeat(1, Tt, 1, 1)
    w2c_t = geo.synthetic_trajectory(1, Tt)
    w2c_ctx = geo.synthetic_trajectory(1, 8)[:, [5, 7]]
    w2c_ctx[0, :, :3, 3] += 0.2 * torch.randn(2, 3, generator=g)
    cond_idx = torch.tensor([1])
    c2w_t, c2w_c = w2c_t.inverse(), w2c_ctx.inverse()
    c2w_c = torch.cat((c2w_t[torch.arange(1), cond_idx].unsqueeze(1), c2w_c), dim=1)
    rel = CameraControlLVDM.get_pairwise_relative_pose(stub, c2w_c, c2w_t)
    rel = rearrange(rel, 'B T C H W -> B C T H W')
    R, t = rel[..., :3, :3], rel[..., :3, 3:4]
    Cn = R.shape[2]
    Kr = repeat(Kc, 'B T H W -> B (T C) H W', C=Cn)
    Fc = cc.CamContextI2V.get_fundamental_matrix(stub, Kr, rearrange(R, 'B T C H W -> B (T C) H W'), rearrange(t, 'B T C H W -> B (T C) H W'))
    Fc = rearrange(Fc, 'B (T C) H W -> B T C H W', C=Cn)
    cmask = cc.CamContextI2V.get_epipolar_mask(stub, Fc, Tt, px // 8, px // 8, 8, True)
    Fo = geo.conditional_fundamental(Kc, w2c_t, w2c_ctx, cond_idx)
    assert (Fo - Fc).abs().max().item() < 1e-4 * Fc.abs().max().item()
    assert torch.equal(geo.epipolar_mask(Fc, px // 8, px // 8, 8), cmask)
    np.savez_compressed(os.path.join(args.out, "pose_small.npz"), K=K.numpy(), c2w=c2w.numpy(), plucker=out["plucker"], ray=out["ray"],
                        cond_K=Kc.numpy(), cond_w2c=w2c_t.numpy(), cond_w2c_ctx=w2c_ctx.numpy(), cond_index=cond_idx.numpy(),
                        cond_F=Fc.numpy(), cond_mask=np.packbits(cmask.numpy(), axis=-1, bitorder="little"),
                        cond_mask_shape=np.array(cmask.shape))
    print("conditional mask", tuple(cmask.shape), "density %.3f" % cmask.float().mean().item())
    print("pose_small: ray_condition", out["plucker"].shape, "oracle agrees")


if __name__ == "__main__":
    main()
