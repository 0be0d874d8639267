"""``lvdm.models.samplers.ddim`` import path.  Implementation: camc2v_amd.sampler."""
from camc2v_amd.sampler import DDIMSampler  # noqa: F401
