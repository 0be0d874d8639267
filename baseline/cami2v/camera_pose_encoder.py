"""``baseline.cami2v.camera_pose_encoder`` import path (configs/baseline/cami2v_256.yaml ``pose_encoder_config.target``; the
reference's file is identical to model/modules/camera_pose_encoder.py).  Implementation: camc2v_amd.pose."""
from camc2v_amd.pose import CameraPoseEncoder, PositionalEncoding, ResnetBlock, TemporalSelfAttention, TemporalTransformerBlock  # noqa: F401
