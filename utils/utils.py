"""``utils.utils`` import path: the reference's plugin mechanism.  Implementation: camc2v_amd.config."""
from camc2v_amd.config import get_obj_from_str, instantiate_from_config  # noqa: F401
