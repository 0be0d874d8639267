"""``lvdm.modules.encoders.condition`` import path (yaml ``cond_stage_config`` / ``img_cond_stage_config`` targets).

The OpenCLIP ViT-H text and image encoders are third-party code (``open_clip``) with third-party weights
(reference lvdm/modules/encoders/condition.py:174-372): they are not part of the DDIM
hot path and are not rebuilt here.  The classes exist so that a reference yaml resolves and fails with a message that
says what the sampling path expects instead: the encoders' OUTPUT tensors -- text tokens [b, 77, 1024] and image tokens
[b, 257, 1280] (the latter go through ``lvdm.modules.encoders.resampler.Resampler``, which is built).
"""
import torch.nn as nn


class AbstractEncoder(nn.Module):
    def encode(self, *args, **kwargs):
        raise NotImplementedError


class _NeedsOpenClip(AbstractEncoder):
    _what = "encoder"

    def __init__(self, *args, **kwargs):
        super().__init__()
        raise NotImplementedError(
            f"{type(self).__name__}: the OpenCLIP {self._what} is third-party code and weights and is outside the MI355X hot path; "
            "run the reference's encoder (or any OpenCLIP ViT-H/14 build) and hand its output tensor to the model as "
            "c_crossattn / image tokens (see DESIGN.md section 7, tools/generate_demo.py)")


class FrozenOpenCLIPEmbedder(_NeedsOpenClip):
    """Text encoder: tokens [b, 77, 1024] (reference condition.py:174-235)."""
    _what = "text encoder"


class FrozenOpenCLIPImageEmbedderV2(_NeedsOpenClip):
    """Image encoder: tokens [b, 257, 1280] before the Resampler (reference condition.py:295-372)."""
    _what = "image encoder"


__all__ = ["AbstractEncoder", "FrozenOpenCLIPEmbedder", "FrozenOpenCLIPImageEmbedderV2"]
