"""Everything built on the HIP kernels chained at the shipped sizes (seeded weights, synthetic inputs): first-stage
encode -> adaptor, Resampler, poses -> embedding -> pose encoder, masks, CFG DDIM steps, first-stage decode."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_generate_demo_two_steps():
    import torch
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "generate_demo.py"), "--steps", "2"], capture_output=True,
                         text=True, timeout=600, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-2000:]
    line = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    assert line["frames"] == [1, 3, 16, 256, 256]
    for key in ("first_stage_encode_3_images_ms", "pose_encoder_ms", "context_concat_adaptor_ms", "resampler_ms", "ddim_2_cfg_steps_ms",
                "first_stage_decode_16_frames_ms"):
        assert line[key] > 0


def test_full_size_clip_is_reproducible():
    """Size-independent property at BASELINE.json's full workload (1 x 16 x 256 x 256, camera + 2 context frames, 25 CFG steps,
    hipGraph replay): the same inputs and noise draws give bit-identical latents clip after clip, although the sparse
    attention hands its work out through a device counter and the GEMMs split K (fixed-order reductions everywhere)."""
    import torch
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "determinism_check.py"), "3"], capture_output=True, text=True,
                         timeout=600, cwd=ROOT)
    assert out.returncode == 0, (out.stdout[-1000:], out.stderr[-2000:])
    assert out.stdout.count("identical to clip 0: True") == 3
