"""``baseline.cami2v.cami2v`` import path.  Implementation: camc2v_amd.models."""
from camc2v_amd.models import CamI2V  # noqa: F401
