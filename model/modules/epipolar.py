"""``model.modules.epipolar`` import path.  Implementation: camc2v_amd.unet."""
from camc2v_amd.unet import Epipolar, EpipolarCrossAttention  # noqa: F401


def pix2coord(x, downsample):
    """pixel index -> pixel-centre coordinate (reference model/modules/epipolar.py:32-34)."""
    return x * downsample + downsample / 2.0 - 0.5
