"""``lvdm.models.autoencoder`` import path (yaml ``first_stage_config.target``).  Implementation: camc2v_amd.vae."""
from camc2v_amd.vae import AutoencoderKL  # noqa: F401
