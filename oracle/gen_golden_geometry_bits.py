#!/usr/bin/env python
"""Generate tests/golden/geometry_bits.npz by RUNNING THE REFERENCE (build container only).

TEST INFRASTRUCTURE.  Positions, not only per-row counts, of the reference's epipolar masks at 256 x 256 px
(`CamContextI2V.get_epipolar_mask`, model/camcontexti2v.py:200-271, on the F256 of tests/golden/geometry.npz, i.e. the
same 16-frame trajectory and the same perturbation draw):

  * mask256_d{16,32,64}_bits : the boolean masks packed little-endian along the key axis (uint8 [1, L, L/8]);
  * mask256_d{8,16,32,64}_sha256 : SHA-256 of those packed bytes (the 16384 x 16384 mask of the 32x32-latent level is 32 MiB
    packed: its digest pins every position without shipping it).

Usage:  python oracle/gen_golden_geometry_bits.py [--out tests/golden]     (the GPU box never runs this)
"""
import argparse
import hashlib
import importlib.util
import os

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
    gg._install_shims()
    torch.set_grad_enabled(False)
    fx = np.load(os.path.join(args.out, "geometry.npz"))
    import model.camcontexti2v as cc
    assert cc.__file__.startswith(gg.REF), cc.__file__
    import types
    stub = types.SimpleNamespace()
    stub.epipolar_config = types.SimpleNamespace(
        apply_epipolar_soft_mask=False, epipolar_hybrid_attention=False, epipolar_hybrid_attention_v2=False,
        only_self_pixel_on_current_frame=False, current_frame_as_register_token=False)
    F = torch.from_numpy(fx["F256"])
    out = {}
    for d in (8, 16, 32, 64):
        m = cc.CamContextI2V.get_epipolar_mask(stub, F, F.shape[1], 256 // d, 256 // d, d)
        assert np.array_equal(m.sum(-1).to(torch.int32).numpy(), fx[f"mask256_d{d}_popcount_rows"]), d
        bits = np.packbits(m.numpy().astype(np.uint8), axis=-1, bitorder="little")
        out[f"mask256_d{d}_sha256"] = np.frombuffer(hashlib.sha256(bits.tobytes()).digest(), dtype=np.uint8)
        if d >= 16:
            out[f"mask256_d{d}_bits"] = bits
        print(f"d={d}: mask {tuple(m.shape)}, {int(m.sum())} visible pairs, sha256 {hashlib.sha256(bits.tobytes()).hexdigest()[:16]}...")
    np.savez_compressed(os.path.join(args.out, "geometry_bits.npz"), **out)


if __name__ == "__main__":
    main()
