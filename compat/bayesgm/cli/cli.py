"""bayesgm.cli.cli -> bayesgm_amd.cli (`bayesgm = bayesgm.cli.cli:main`, `causalBGM = bayesgm.cli.cli:main_causalbgm` in the
reference's setup.py:29-33 resolve here when a caller pins those paths)."""
from bayesgm_amd.cli import main, main_causalbgm

__all__ = ["main", "main_causalbgm"]
