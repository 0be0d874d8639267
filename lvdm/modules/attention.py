"""``lvdm.modules.attention`` import path.  Implementation: camc2v_amd.unet."""
from camc2v_amd.unet import (BasicTransformerBlock, CrossAttention, FeedForward, GEGLU,  # noqa: F401
                             SpatialTransformer, TemporalTransformer)
