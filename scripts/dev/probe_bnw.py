"""Cost of the any-width Bayesian sampling path (bnw_kernels.h): python scripts/dev/probe_bnw.py [N]"""
import sys, time, json
import numpy as np, torch
sys.path.insert(0, ".")
from bayesgm_amd.bnn_engine import BnnEngine
from oracle import bnn as OB
N = int(float(sys.argv[1])) if len(sys.argv) > 1 else 100000
p, zd, bs = 200, [1, 1, 1, 7], 10000
for name, units in (("[128,128]", dict(g_units=(128, 128), e_units=(128, 128), f_units=(128, 128), h_units=(128, 128))),
                    ("[256]x3", dict(g_units=(256,) * 3, e_units=(256,) * 3, f_units=(256,) * 3, h_units=(256,) * 3))):
    m = OB.init_model(0, zd, p, False, **units)
    for k in ("g", "e", "f", "h"):
        m[k]["norm"] = "fixed"
    eng = BnnEngine(p, zd, False, norm_mode=1, **units)
    eng.begin(m)
    dev = eng.device
    x = torch.rand(N, device=dev); y = torch.randn(N, device=dev); v = torch.randn(N, p, device=dev)
    state = torch.zeros(N, 10, device=dev)
    eng.mh_run(x, y, v, state, bs, 0, 2, 10 ** 9, 0.3, 5, init=True)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    iters = 5
    eng.mh_run(x, y, v, state, bs, 2, iters, 10 ** 9, 0.3, 5)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / iters
    macs = sum(a * b for net in ("g", "h", "f") for a, b in zip(OB.net_dims(m[net])[:-1], OB.net_dims(m[net])[1:]))
    print(json.dumps(dict(shape=name, N=N, ms_per_iteration=1e3 * dt, transitions_per_s=N / dt, tflops=2 * 2 * 2 * macs * N / dt / 1e12)))
    eng.close()
