"""Checkpoint loading with the reference's container handling (reference main/utils_train.py:165-214; the same
logic sits in main/runtime.py:104-116 and lvdm/models/ddpm3d.py:205-221).

A CamContextI2V / DynamiCrafter checkpoint is one flat ``state_dict`` stored in one of three containers:

    {"state_dict": {...}, ...}      Lightning / DDP        (the released 256 models)
    {"module": {...}, ...}          DeepSpeed ZeRO stage 1  (``checkpoint/mp_rank_00_model_states.pt``)
    {...}                           a bare state_dict

Old DynamiCrafter-256 files call the frame-stride MLP ``framestride_embed``; the module attribute is
``fps_embedding`` (renamed on load, as the reference does).  The load is tried strictly first and falls back to
non-strict like the reference's -- but instead of swallowing what did not match, the fallback returns a report: this
package does not instantiate the OpenCLIP encoders, losses or EMA copies, so ``cond_stage_model.*``, ``embedder.*``,
``logvar`` ... of a real checkpoint are listed as ignored prefixes, while anything MISSING for the modules that do
exist is listed by name.
"""
import logging
import os
from collections import OrderedDict

import torch

log = logging.getLogger("mainlogger")


class LoadReport(dict):
    """{"container", "strict", "renamed", "missing", "unexpected", "ignored_prefixes", "loaded"}"""

    def __str__(self):
        ig = ", ".join(f"{k} ({v})" for k, v in self["ignored_prefixes"].items()) or "-"
        return (f"checkpoint container '{self['container']}': {self['loaded']} tensors loaded, strict={self['strict']}, "
                f"{len(self['renamed'])} renamed, {len(self['missing'])} missing, ignored prefixes: {ig}")


def _cfg_get(cfg, key):
    if cfg is None:
        return None
    if isinstance(cfg, dict):
        return cfg.get(key)
    return getattr(cfg, key, None)


def extract_state_dict(obj):
    """-> (container name, flat state dict) for the three layouts the reference accepts."""
    if isinstance(obj, dict) and "state_dict" in obj and isinstance(obj["state_dict"], dict):
        return "state_dict", obj["state_dict"]
    if isinstance(obj, dict) and "module" in obj and isinstance(obj["module"], dict):
        return "module", obj["module"]
    if isinstance(obj, dict):
        return "bare", obj
    raise TypeError(f"unsupported checkpoint object of type {type(obj).__name__}")


def rename_legacy_keys(sd):
    """framestride_embed -> fps_embedding (main/utils_train.py:182-186).  Returns (new dict, list of renamed keys)."""
    out, renamed = OrderedDict(), []
    for k, v in sd.items():
        if "framestride_embed" in k:
            renamed.append(k)
            k = k.replace("framestride_embed", "fps_embedding")
        out[k] = v
    return out, renamed


def load_state_dict_with_report(model, sd, container="bare"):
    sd, renamed = rename_legacy_keys(sd)
    own = set(model.state_dict().keys())
    missing = sorted(own - set(sd))
    unexpected = sorted(set(sd) - own)
    strict = not missing and not unexpected
    if strict:
        model.load_state_dict(sd, strict=True)
    else:
        # shapes must still agree wherever a key matches: nn.Module.load_state_dict raises on a size mismatch
        model.load_state_dict(sd, strict=False)
    ignored = OrderedDict()
    for k in unexpected:
        p = k.split(".", 1)[0]
        ignored[p] = ignored.get(p, 0) + 1
    return LoadReport(container=container, strict=strict, renamed=renamed, missing=missing, unexpected=unexpected,
                      ignored_prefixes=ignored, loaded=len(own & set(sd)))


def load_checkpoints(model, model_cfg, report=None):
    """Same call as the reference's ``load_checkpoints(model, config.model)``: reads ``model_cfg.pretrained_checkpoint``
    (attribute or key), loads it into ``model`` and returns the model.  ``report`` (optional list) receives the LoadReport."""
    ckpt = _cfg_get(model_cfg, "pretrained_checkpoint")
    if not ckpt:
        log.info(">>> Start from randomly initialised weights (no pretrained_checkpoint)")
        return model
    assert os.path.exists(ckpt), "Error: Pre-trained checkpoint NOT found at:%s" % ckpt
    log.info(">>> Load weights from pretrained checkpoint")
    try:
        obj = torch.load(ckpt, map_location="cpu", weights_only=True)
    except Exception as e:   # Lightning / DeepSpeed files carry pickled hyper-parameter objects next to the tensors
        # the reference loads with full unpickling (main/utils_train.py:170: torch.load(ckpt, map_location="cpu")), which runs
        # whatever code the file names: do that only for files the caller trusts, and say so
        if os.environ.get("CCV_CHECKPOINT_WEIGHTS_ONLY", "0") == "1":
            raise RuntimeError(f"{ckpt}: not loadable with weights_only=True ({type(e).__name__}: {e}) and CCV_CHECKPOINT_WEIGHTS_ONLY=1 "
                               "forbids full unpickling") from e
        log.warning("%s is not a tensors-only file (%s: %s): loading it with full unpickling, as the reference does -- only do this "
                    "with checkpoints you trust (CCV_CHECKPOINT_WEIGHTS_ONLY=1 refuses instead)", ckpt, type(e).__name__, str(e)[:200])
        obj = torch.load(ckpt, map_location="cpu", weights_only=False)
    container, sd = extract_state_dict(obj)
    rep = load_state_dict_with_report(model, sd, container)
    if report is not None:
        report.append(rep)
    log.info(">>> Loaded weights from pretrained checkpoint: %s (%s)", ckpt, rep)
    if rep["missing"]:
        log.warning("checkpoint lacks %d tensors of the model, e.g. %s", len(rep["missing"]), rep["missing"][:4])
    del obj
    return model


__all__ = ["load_checkpoints", "load_state_dict_with_report", "extract_state_dict", "rename_legacy_keys", "LoadReport"]
