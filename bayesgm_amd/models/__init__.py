"""Drop-in mirror of ``bayesgm.models`` for the hot path."""
from .causalbgm import CausalBGM
from .bgm import BGM
from .identifiable import IdentifiableCausalBGM

__all__ = ["CausalBGM", "BGM", "IdentifiableCausalBGM"]
