"""``model.modules.camera_pose_encoder`` import path (yaml ``pose_encoder_config.target``).  Implementation:
camc2v_amd.pose."""
from camc2v_amd.pose import CameraPoseEncoder, PositionalEncoding, ResnetBlock, TemporalSelfAttention, TemporalTransformerBlock  # noqa: F401
