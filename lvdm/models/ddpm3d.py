"""``lvdm.models.ddpm3d`` import path (sampling-side slice only).  Implementation: camc2v_amd.diffusion."""
from camc2v_amd.diffusion import DiffusionWrapper, LatentDiffusionCore  # noqa: F401

LatentDiffusion = LatentDiffusionCore
