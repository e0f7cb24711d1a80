"""bayesgm.models -> bayesgm_amd.models (the same class objects)."""
from bayesgm_amd.models import BGM, CausalBGM, IdentifiableCausalBGM

__all__ = ["CausalBGM", "IdentifiableCausalBGM", "BGM"]


def __getattr__(name):
    raise AttributeError("bayesgm.models.%s is outside the hot path bayesgm_amd implements (available: %s)" % (name, ", ".join(__all__)))
