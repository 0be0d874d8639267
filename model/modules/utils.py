"""``model.modules.utils`` import path.  Implementation: camc2v_amd.adaptor."""
from camc2v_amd.adaptor import CrossNormalization  # noqa: F401
