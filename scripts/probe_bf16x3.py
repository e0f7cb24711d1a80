"""fp32 MFMA vs split-precision bf16 x 3 for one hidden 64 -> 64 layer of the MH kernel (measurement for the next step):
time per layer and error against float64.  usage: python scripts/probe_bf16x3.py"""
import ctypes as C, json, sys
import numpy as np
sys.path.insert(0, "scripts")
from _probe_lib import load
lib = load()


def check(rc, what):
    assert rc == 0, (what, rc)


rs = np.random.RandomState(0)
lim = np.sqrt(6.0 / 128)
W = rs.uniform(-lim, lim, size=(64, 64)).astype(np.float32) * 1.6          # glorot-uniform, scaled to keep |h| O(1) over many layers
x = rs.standard_normal((16, 64)).astype(np.float32)
res = {}
for n_layers in (1, 4, 16):
    ref = x.astype(np.float64)
    for _ in range(n_layers):
        ref = ref @ W.astype(np.float64).T
        ref = np.maximum(ref, 0.2 * ref)
    for mode, name in ((0, "fp32"), (1, "bf16x3")):
        out = np.empty((16, 64), np.float32)
        ns = C.c_double()
        check(lib.bgm_probe_bf16x3(0, mode, n_layers, W.ctypes.data_as(C.c_void_p), x.ctypes.data_as(C.c_void_p),
                                              out.ctypes.data_as(C.c_void_p), C.byref(ns)), "probe")
        res["%s_rel_err_%d_layers" % (name, n_layers)] = float(np.abs(out - ref).max() / np.abs(ref).max())
for mode, name in ((0, "fp32"), (1, "bf16x3")):
    out = np.empty((16, 64), np.float32)
    ns = C.c_double()
    check(lib.bgm_probe_bf16x3(0, mode, 200000, W.ctypes.data_as(C.c_void_p), x.ctypes.data_as(C.c_void_p),
                                          out.ctypes.data_as(C.c_void_p), C.byref(ns)), "probe")
    res[name + "_ns_per_layer_8_waves_per_cu"] = ns.value
    res[name + "_equiv_fp32_tflops"] = 256 * 8 * 2.0 * 64 * 64 * 16 / (ns.value * 1e-9) / 1e12
res["speedup"] = res["fp32_ns_per_layer_8_waves_per_cu"] / res["bf16x3_ns_per_layer_8_waves_per_cu"]
print(json.dumps(res))
