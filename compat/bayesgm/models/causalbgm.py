"""bayesgm.models.causalbgm -> bayesgm_amd.models"""
from bayesgm_amd.models import CausalBGM, IdentifiableCausalBGM

__all__ = ["CausalBGM", "IdentifiableCausalBGM"]
