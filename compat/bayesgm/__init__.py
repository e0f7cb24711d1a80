"""`bayesgm` import name for callers written against the reference package (the R wrapper's reticulate::import("bayesgm.models") /
("bayesgm.datasets"), r-package/bayesgm/R/python-config.R:86,106; scripts that `from bayesgm.models import CausalBGM`).

This is an ALIAS: every public name is the object of the same name in bayesgm_amd (bayesgm.models.CausalBGM is
bayesgm_amd.models.CausalBGM); nothing is implemented here.  It is opt-in: installed next to bayesgm_amd by `pip install .`
(pyproject.toml maps it from compat/), or put compat/ on PYTHONPATH / BAYESGM_PYTHONPATH for an in-tree checkout.  Names of the reference
outside the hot path this build covers (FullMCMCCausalBGM, MNISTBGM, the ACIC / Twins loaders ...) raise AttributeError naming the gap.
Submodules resolve lazily, as the reference's do (src/bayesgm/__init__.py:50-56): importing `bayesgm` loads neither torch nor the HIP library.
"""
from importlib import import_module

import bayesgm_amd as _impl

__version__ = "1.0.2"            # the reference release whose surface this mirrors
__backend__ = "bayesgm_amd " + _impl.__version__

_SYMBOL_TO_MODULE = {
    "CausalBGM": "bayesgm.models", "IdentifiableCausalBGM": "bayesgm.models", "BGM": "bayesgm.models",
    "Base_sampler": "bayesgm.datasets", "Sim_Hirano_Imbens_sampler": "bayesgm.datasets", "Sim_Sun_sampler": "bayesgm.datasets",
    "Sim_Colangelo_sampler": "bayesgm.datasets",
}
_MODULES = ("models", "datasets", "utils", "cli")
__all__ = sorted(_SYMBOL_TO_MODULE)


def __getattr__(name):
    if name in _SYMBOL_TO_MODULE:
        return getattr(import_module(_SYMBOL_TO_MODULE[name]), name)
    if name in _MODULES:
        return import_module("bayesgm." + name)
    raise AttributeError("module 'bayesgm' (alias of bayesgm_amd) has no attribute %r" % (name,))


def __dir__():
    return sorted(set(globals()) | set(__all__) | set(_MODULES))
