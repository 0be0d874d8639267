"""``model.dynamicrafter`` import path.  Implementation: camc2v_amd.models."""
from camc2v_amd.models import DynamiCrafter  # noqa: F401
