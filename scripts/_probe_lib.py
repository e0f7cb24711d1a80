"""ctypes loader of the measurement-aid library (bayesgm_amd/csrc/probes/libbgm_probe.so; `python -m bayesgm_amd.csrc.build --probes`).
Not part of the product ABI (include/bgm_hip.h)."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))


def load():
    from bayesgm_amd.csrc.build import build_probes
    lib = C.CDLL(build_probes(verbose=False))
    lib.bgm_probe_group.argtypes = [C.c_int, C.c_int32, C.c_int32, C.c_int32, C.POINTER(C.c_double)]
    lib.bgm_probe_clock.argtypes = [C.c_int, C.c_int32, C.POINTER(C.c_double), C.POINTER(C.c_double)]
    lib.bgm_probe_bf16x3.argtypes = [C.c_int, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(C.c_double)]
    return lib
