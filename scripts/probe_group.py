"""MFMA throughput of the scheduled tile-group block alone vs waves per CU (measurement aid, probes/probe_kernels.hip)."""
import ctypes as C
from _probe_lib import load
lib = load()
for mode in (0, 1):
    for w in (4, 8, 12, 16):
        tf = C.c_double()
        rc = lib.bgm_probe_group(0, mode, w, 20000, C.byref(tf))
        print("mode", mode, "waves/CU", w, "rc", rc, "TFLOP/s %.1f" % tf.value, "(%.1f %% of 157.3)" % (100 * tf.value / 157.3))
mhz, tf = C.c_double(), C.c_double()
lib.bgm_probe_clock(0, 200000, C.byref(mhz), C.byref(tf))
print("sustained shader clock under a pure fp32-MFMA load: %.0f MHz, %.1f TFLOP/s" % (mhz.value, tf.value))
