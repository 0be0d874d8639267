#!/usr/bin/env python
"""Generate tests/golden/crossnorm_small.npz by RUNNING THE REFERENCE's ``CrossNormalization`` (model/modules/utils.py;
build container only).  TEST INFRASTRUCTURE, companion of oracle/gen_golden.py.  Fixtures hold tensors only.

What is run: the module with dims (-3, -2, -1) as model/camcontexti2v.py:80 constructs it, in the two call forms that
follow the context-frame adaptor (model/camcontexti2v.py:357-360) and in the self-referenced form.

Usage:  cd oracle && python gen_golden_crossnorm.py [--out ../tests/golden]
"""
import argparse
import importlib.util
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
        "run from oracle/: the repo root on sys.path would shadow the reference's packages"
    gg = _load("gen_golden")
    gg._install_shims()
    ao = _load("adaptor_oracle")
    torch.set_grad_enabled(False)

    from model.modules.utils import CrossNormalization
    assert sys.modules["model.modules.utils"].__file__.startswith(REF)
    cn = CrossNormalization((-3, -2, -1))

    g = torch.Generator().manual_seed(gg.SEED + 31)
    B, T, D, H, W = 2, 4, 4, 8, 8
    lat = torch.randn(B, T * H * W, D, generator=g) * 1.7 + 0.3            # adaptor output, token-major
    z_cond = torch.randn(B, D, H, W, generator=g) * 0.6 - 0.2               # conditioning latent
    x_st = lat.reshape(B, T, H, W, D).permute(0, 1, 4, 2, 3).contiguous()    # 'B (T H W) D -> B T D H W'
    y_st = cn(x_st, z_cond.unsqueeze(1))                                   # camcontexti2v.py:357-358
    y_tok = cn(lat.unsqueeze(1), z_cond)                                   # camcontexti2v.py:360
    y_self = cn(x_st)
    for got, ref in ((ao.cross_normalization(x_st, z_cond[:, None]), y_st), (ao.cross_normalization(lat[:, None], z_cond), y_tok),
                     (ao.cross_normalization(x_st), y_self),
                     (ao.cross_normalize_adaptor_output(lat, z_cond, T, H, W, "spatio_temporal"), y_st)):
        assert (got - ref).abs().max().item() < 1e-6, (got - ref).abs().max().item()
    np.savez_compressed(os.path.join(args.out, "crossnorm_small.npz"), lat=lat.numpy(), z_cond=z_cond.numpy(), y_st=y_st.numpy(),
                        y_tok=y_tok.numpy(), y_self=y_self.numpy())
    print(f"crossnorm_small: y_st absmax {y_st.abs().max().item():.3f}, y_tok shape {tuple(y_tok.shape)}")


if __name__ == "__main__":
    main()
