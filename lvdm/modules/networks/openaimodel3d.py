"""``lvdm.modules.networks.openaimodel3d`` -- dotted path used by ``unet_config.target`` in the
reference's yaml (configs/models/camcontexti2v_256.yaml:41).  Implementation: camc2v_amd.unet."""
from camc2v_amd.unet import (Downsample, ResBlock, TemporalConvBlock, TimestepBlock,  # noqa: F401
                             TimestepEmbedSequential, UNetModel, Upsample)
