"""``model.camcontexti2v`` import path (configs/models/camcontexti2v_256.yaml:5).  Implementation: camc2v_amd.models."""
from camc2v_amd.models import CamContextI2V  # noqa: F401
