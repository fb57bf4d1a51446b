"""Config factory of the drop-in boundary (mirrors the reference's utils/util.py:45-63).

``instantiate_from_config`` accepts plain dicts, PyYAML output or OmegaConf nodes
(omegaconf is not installed in this image, so nothing here imports it).  Reference
yaml targets ``modules.mage_model.*`` / ``modules.vqvae_model.*`` resolve to the
MI355X-native classes in ``mage_amd.modules`` (the repo-root ``modules`` package is a
re-export shim, so either spelling works).
"""
from __future__ import annotations

import importlib
from inspect import isfunction
from typing import Any, Mapping, Optional

_ALIASES = {
    "modules.mage_model": "mage_amd.modules.mage_model",
    "modules.vqvae_model": "mage_amd.modules.vqvae_model",
}


def exists(x: Any) -> bool:
    return x is not None


def default(val: Any, d: Any) -> Any:
    if exists(val):
        return val
    return d() if isfunction(d) else d


def zero_module(module):
    for p in module.parameters():
        p.detach().zero_()
    return module


def get_obj_from_str(string: str, reload: bool = False):
    module, cls = string.rsplit(".", 1)
    module = _ALIASES.get(module, module)
    mod = importlib.import_module(module)
    if reload:
        mod = importlib.reload(mod)
    return getattr(mod, cls)


def _plain(cfg: Any) -> Any:
    if isinstance(cfg, Mapping) or (hasattr(cfg, "keys") and hasattr(cfg, "__getitem__")):
        return {k: _plain(cfg[k]) for k in cfg.keys()}
    if isinstance(cfg, (list, tuple)) or type(cfg).__name__ == "ListConfig":
        return [_plain(v) for v in cfg]
    return cfg


def instantiate_from_config(config: Any, merge: Optional[Mapping] = None):
    """``cls(**params)`` for ``{"target": "pkg.mod.Class", "params": {...}}``.  ``merge`` overrides
    params (the reference merges through a set literal, utils/util.py:53, so precedence is formally
    unordered there; key sets are disjoint in every shipped yaml, and overrides win here)."""
    if "target" not in config:
        if config in ("__is_first_stage__", "__is_unconditional__"):
            return None
        raise KeyError("Expected key `target` to instantiate.")
    params = dict(_plain(config.get("params", {}) or {}))
    if merge is not None:
        params.update(_plain(merge))
    return get_obj_from_str(config["target"])(**params)
