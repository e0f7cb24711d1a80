"""bayesgm_amd -- MI355X-native hot path of bayesgm behind the reference's Python surface.

``import bayesgm_amd as bayesgm`` gives ``bayesgm.models`` (CausalBGM, BGM), ``bayesgm.datasets``, ``bayesgm.utils`` and
``bayesgm.cli`` on first use, as the reference package resolves its submodules lazily (src/bayesgm/__init__.py:50-56).
Nothing here touches the GPU or loads libbgm_hip.so until a model is constructed.
"""
import importlib

__version__ = "0.1.0"
_SUBMODULES = ("models", "datasets", "utils", "cli", "main", "parallel")


def __getattr__(name):
    if name in _SUBMODULES:
        return importlib.import_module("." + name, __name__)
    raise AttributeError("module %r has no attribute %r" % (__name__, name))


def __dir__():
    return sorted(list(globals()) + list(_SUBMODULES))
