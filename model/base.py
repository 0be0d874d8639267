"""``model.base`` import path.  Implementation: camc2v_amd.models."""
from camc2v_amd.models import CameraControlLVDM  # noqa: F401
