"""Harness-side import of the read-only reference (build container only).

Registers three stub modules in ``sys.modules`` so that ``modules.mage_model``
imports without touching /root/reference:
``pytorch_transformers`` (only BertTextualHead uses it), ``omegaconf``
(``OmegaConf.merge/load`` + a hashable ``DictConfig``; utils/util.py:53 puts
two configs in a set) and ``ldm.models.autoencoder.DiagonalGaussianDistribution``
(only an isinstance check, mage_model.py:543).  Never shipped to the GPU box's
test path: only tools/gen_golden.py uses it.
"""
import sys
import types

REF = "/root/reference"


class DictConfig(dict):
    def __hash__(self):
        return id(self)


def to_cfg(d):
    if isinstance(d, dict):
        return DictConfig({k: to_cfg(v) for k, v in d.items()})
    return d


def import_reference():
    # our repo root also has `modules/` + `utils/` drop-in shims (regular packages, which beat the reference's
    # namespace packages whatever the path order): hide the repo root while the reference is imported
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    saved = list(sys.path)
    sys.path[:] = [REF] + [p for p in saved if os.path.abspath(p or os.getcwd()) != root and p != REF]
    try:
        return _import_reference_inner()
    finally:
        sys.path[:] = saved


def _import_reference_inner():
    for name in list(sys.modules):
        if name == "modules" or name.startswith("modules.") or name == "utils" or name.startswith("utils."):
            del sys.modules[name]
    sys.modules.setdefault("pytorch_transformers", types.ModuleType("pytorch_transformers"))

    class OmegaConf:
        @staticmethod
        def merge(*cfgs):
            out = DictConfig()
            for c in cfgs:
                out.update(c)
            return out

        @staticmethod
        def load(path):
            import yaml
            return to_cfg(yaml.safe_load(open(path)))

    om = types.ModuleType("omegaconf")
    om.OmegaConf, om.DictConfig = OmegaConf, DictConfig
    sys.modules["omegaconf"] = om
    for n in ("ldm", "ldm.models", "ldm.models.autoencoder"):
        sys.modules.setdefault(n, types.ModuleType(n))

    class DiagonalGaussianDistribution:  # noqa: D401 - stand-in for an isinstance() check only
        pass

    sys.modules["ldm.models.autoencoder"].DiagonalGaussianDistribution = DiagonalGaussianDistribution
    import modules.mage_model as ref_mage
    import modules.vqvae_model as ref_vq
    assert ref_mage.__file__.startswith(REF), ref_mage.__file__
    return ref_mage, ref_vq
