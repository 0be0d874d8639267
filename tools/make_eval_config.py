"""Write the eval_config.yaml that 02_generate_videos.py would leave for the shipped CamContextI2V-256 config (no checkpoint: use
`generate.py --random-init`), for timing the generation harness.   python tools/make_eval_config.py <repo root> <output dir>"""
import copy, json, os, sys, yaml
root = sys.argv[1]; out = sys.argv[2]
sys.path.insert(0, root)
from tests.test_harness_gpu import _eval_config
ms = copy.deepcopy(json.load(open(os.path.join(root, "tests/golden/model_configs.json")))["camcontexti2v_256"]["model"])
ms["pretrained_checkpoint"] = "/nonexistent.ckpt"
yaml.safe_dump(_eval_config(ms, out, 8, 256), open(os.path.join(out, "eval_config.yaml"), "w"))
