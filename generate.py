#!/usr/bin/env python
"""`python generate.py <eval_config.yaml>`: the generation step of the reference (02_generate_videos.py -> trainer.test ->
ImageLogger -> log_images) on the MI355X path.  Implementation: camc2v_amd/harness.py."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from camc2v_amd.harness import main  # noqa: E402

if __name__ == "__main__":
    sys.exit(main())
