"""MFMA throughput of the scheduled tile-group block alone vs waves per CU (debug)."""
import sys, os, ctypes as C
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from bayesgm_amd.engine import CausalEngine
eng = CausalEngine(200, [1, 1, 1, 7])
for mode in (0, 1):
    for w in (4, 8, 12, 16):
        tf = C.c_double()
        rc = eng.lib.bgm_debug_group_probe(eng.h, mode, w, 20000, C.byref(tf))
        print("mode", mode, "waves/CU", w, "TFLOP/s %.1f" % tf.value, "(%.1f %% of 157.3)" % (100 * tf.value / 157.3))
