"""``lvdm.models.utils_diffusion`` import path.  Implementation: camc2v_amd.sampler."""
from camc2v_amd.sampler import (make_beta_schedule, make_ddim_sampling_parameters,  # noqa: F401
                                make_ddim_timesteps)
