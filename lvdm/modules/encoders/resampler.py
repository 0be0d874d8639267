"""``lvdm.modules.encoders.resampler`` import path (yaml ``image_proj_stage_config.target``).  Implementation:
camc2v_amd.resampler."""
from camc2v_amd.resampler import FeedForward, PerceiverAttention, Resampler  # noqa: F401
