"""bayesgm.utils -> bayesgm_amd.utils (the same objects)."""
from bayesgm_amd.utils import get_ADRF, parse_file, parse_file_triplet, save_data

__all__ = ["save_data", "parse_file", "parse_file_triplet", "get_ADRF"]
