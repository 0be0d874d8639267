"""``lvdm.modules.encoders.condition`` import path (yaml ``cond_stage_config`` / ``img_cond_stage_config`` targets,
reference lvdm/modules/encoders/condition.py:174-372): the OpenCLIP ViT-H/14 text and image embedders on the HIP kernels."""
from camc2v_amd.clip import AbstractEncoder, FrozenOpenCLIPEmbedder, FrozenOpenCLIPImageEmbedderV2  # noqa: F401

__all__ = ["AbstractEncoder", "FrozenOpenCLIPEmbedder", "FrozenOpenCLIPImageEmbedderV2"]
