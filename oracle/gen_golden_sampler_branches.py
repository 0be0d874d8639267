#!/usr/bin/env python
"""tests/golden/ddim_branches.npz: the side branches of the REFERENCE's DDIMSampler (lvdm/models/samplers/ddim.py) run on the analytic
noise model of oracle/sampler_cases.py (build container only; TEST INFRASTRUCTURE).  Stores the final latents of every case, the
zero-terminal-SNR betas of the reference's rescale_zero_terminal_snr and input checksums.

Usage:  python oracle/gen_golden_sampler_branches.py [--out tests/golden]"""
import argparse
import importlib.util
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
    sc = _load("sampler_cases")
    gg._install_shims()
    torch.set_grad_enabled(False)
    from lvdm.models.samplers.ddim import DDIMSampler
    from lvdm.models.utils_diffusion import rescale_zero_terminal_snr
    assert sys.modules["lvdm.models.samplers.ddim"].__file__.startswith(gg.REF)

    class CpuSampler(DDIMSampler):
        def register_buffer(self, name, attr):      # the reference hard-codes cuda (ddim.py:18-22)
            setattr(self, name, attr)

    out = {}
    for name in sc.CASES:
        y = sc.run_case(name, CpuSampler)
        assert torch.isfinite(y).all(), name
        out["y_" + name] = y.float().numpy()
        print(f"{name:22s} |y| max {y.abs().max().item():.4f}")
    t = sc.tensors()
    for k, v in t.items():
        if torch.is_tensor(v):
            out["checksum_" + k] = np.array(float(v.double().sum()))
    out["betas_zero_snr"] = rescale_zero_terminal_snr(sc.schedule()).astype(np.float64)
    path = os.path.join(args.out, "ddim_branches.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, f"{os.path.getsize(path) / 1e3:.1f} kB")


if __name__ == "__main__":
    main()
