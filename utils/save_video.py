"""``utils.save_video`` import path: the per-sample writers of the generation harness (reference utils/save_video.py:65-157,
234-251).  Implementation: camc2v_amd.video_io."""
from camc2v_amd.video_io import log_evaluation, prepare_to_log, write_png, write_video  # noqa: F401
