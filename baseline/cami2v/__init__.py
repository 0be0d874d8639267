"""``baseline.cami2v`` import path (configs/baseline/cami2v_256.yaml names ``baseline.cami2v.CamI2V``)."""
from camc2v_amd.models import CamI2V  # noqa: F401
