"""The reference's plugin mechanism: {"target": "pkg.mod.Class", "params": {...}}
(reference utils/utils.py:44-58)."""
import importlib


def get_obj_from_str(string, reload=False):
    module, cls = string.rsplit(".", 1)
    mod = importlib.import_module(module)
    if reload:
        mod = importlib.reload(mod)
    return getattr(mod, cls)


def instantiate_from_config(config):
    if "target" not in config:
        if config in ("__is_first_stage__", "__is_unconditional__"):
            return None
        raise KeyError("Expected key `target` to instantiate.")
    params = config.get("params", dict())
    return get_obj_from_str(config["target"])(**(dict(params) if params is not None else {}))
