"""The DDIM sampler's side branches against the REFERENCE's DDIMSampler (tests/golden/ddim_branches.npz, written by
oracle/gen_golden_sampler_branches.py from lvdm/models/samplers/ddim.py on the analytic noise model of oracle/sampler_cases.py):
mask / x0 blending with and without re-noising (ddim.py:174-181), pasted overlap frames of the autoregressive loop (:183-189, :228-231,
:321-324), scene-constrained noise shaping (:191-201), the conditioning frame pasted into the predicted x0 (:316-320), v
parameterisation (:285-286, :310-311), dynamic rescale (:31-33, :313-317) -- 5 steps, CFG 3.0, guidance_rescale 0.5, eta 0.  The plain
case runs the fused HIP guidance + update kernel, the others the sampler's general fp32 step."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("name", ["plain", "mask_noised", "mask_clean", "paste_overlap", "noise_shaping", "noise_shaping_scene",
                                  "paste_cond_frame", "v_param", "dynamic_rescale"])
def test_sampler_branch_vs_reference_fixture(name, golden_dir):
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from camc2v_amd.sampler import DDIMSampler
    from oracle import sampler_cases as sc
    fx = np.load(os.path.join(golden_dir, "ddim_branches.npz"))
    t = sc.tensors()
    for k, v in t.items():
        if torch.is_tensor(v):
            assert abs(float(v.double().sum()) - float(fx["checksum_" + k])) < 1e-6, f"seeded input {k} changed"
    got = sc.run_case(name, DDIMSampler, device="cuda:0").float().cpu()
    ref = torch.from_numpy(fx["y_" + name])
    err = (got - ref).abs().max().item()
    print(f"[parity] sampler branch {name}: max abs err {err:.3e} (|ref| max {ref.abs().max().item():.2f})")
    assert torch.isfinite(got).all() and err <= 1e-4 * ref.abs().max().item(), (name, err)
    if name != "plain":      # the branch did something
        assert not np.allclose(fx["y_" + name], fx["y_plain"])
