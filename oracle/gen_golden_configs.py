#!/usr/bin/env python
"""Write tests/golden/model_configs.json: the `model:` sections of the reference's shipped 256x256 configs
(configs/models/camcontexti2v_256.yaml, configs/baseline/cami2v_256.yaml, configs/baseline/dynamicrafter_256.yaml) plus
the eval-time `log_images_kwargs` 02_generate_videos.py writes (CamContextI2V/02_generate_videos.py:318-327), as data.
TEST INFRASTRUCTURE: the tests instantiate these through the plugin mechanism where /root/reference is absent.

Usage:  python oracle/gen_golden_configs.py [--out tests/golden]"""
import argparse
import json
import os

import yaml

REF = "/root/reference/configs"
HERE = os.path.dirname(os.path.abspath(__file__))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=os.path.join(os.path.dirname(HERE), "tests", "golden"))
    args = ap.parse_args()
    out = {}
    for name, rel in (("camcontexti2v_256", "models/camcontexti2v_256.yaml"), ("cami2v_256", "baseline/cami2v_256.yaml"),
                      ("dynamicrafter_256", "baseline/dynamicrafter_256.yaml")):
        cfg = yaml.safe_load(open(os.path.join(REF, rel)))
        entry = {"model": cfg["model"]}
        kw = cfg.get("lightning", {}).get("callbacks", {}).get("batch_logger", {}).get("params", {}).get("log_images_kwargs")
        if kw:
            entry["log_images_kwargs"] = kw
        test = cfg.get("data", {}).get("params", {}).get("validation", {}).get("params", {})
        entry["test_data_params"] = {k: test[k] for k in ("video_length", "frame_stride", "resolution", "num_additional_cond_frames") if k in test}
        out[name] = entry
    path = os.path.join(args.out, "model_configs.json")
    with open(path, "w") as f:
        json.dump(out, f, indent=1, sort_keys=True)
    print("wrote", path, os.path.getsize(path), {k: v["model"]["target"] for k, v in out.items()})


if __name__ == "__main__":
    main()
