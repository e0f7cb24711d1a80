"""bayesgm.models.bgm -> bayesgm_amd.models"""
from bayesgm_amd.models import BGM

__all__ = ["BGM"]
