"""Import-path shim: the reference's ``main.utils_train.load_checkpoints`` (main/utils_train.py:165-214)."""
from camc2v_amd.checkpoint import load_checkpoints  # noqa: F401
