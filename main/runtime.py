"""``main.runtime`` import path: the autoregressive chunk loop of the reference's ``Image2Video.get_image``
(main/runtime.py:260-326).  Implementation: camc2v_amd.runtime."""
from camc2v_amd.runtime import extend_trajectory, generate_autoregressive  # noqa: F401
