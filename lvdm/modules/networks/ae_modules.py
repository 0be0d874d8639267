"""``lvdm.modules.networks.ae_modules`` import path.  Implementation: camc2v_amd.vae (decode side)."""
from camc2v_amd.vae import AttnBlock, Decoder, Downsample, Encoder, Normalize, ResnetBlock, Upsample  # noqa: F401
