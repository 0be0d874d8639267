"""``baseline.cami2v.epipolar`` import path (identical to model/modules/epipolar.py in the reference).  Implementation:
camc2v_amd.unet."""
from camc2v_amd.unet import Epipolar, EpipolarCrossAttention  # noqa: F401
