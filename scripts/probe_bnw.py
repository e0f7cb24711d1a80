"""Timing of the wide Bayesian-network sampling path (csrc/bnw_kernels.h: use_bnn=True with hidden widths > 64, inference-mode input
normalisation): MH iterations and kept iterations (20 fresh-noise doses) on a panel of N rows, bs = 10000.
usage: python scripts/probe_bnw.py [N=1e5] [iters=10]      (shapes: [128, 128] and [256] x 3; p = 200, z_dims [1,1,1,7])"""
import sys, time
import numpy as np
import torch
sys.path.insert(0, ".")
from oracle import bnn as OB
from bayesgm_amd.bnn_engine import BnnEngine

N = int(float(sys.argv[1])) if len(sys.argv) > 1 else 100000
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 10
p, z_dims, bs = 200, [1, 1, 1, 7], 10000
q = sum(z_dims)
for name, u in (("[128, 128]", dict(g_units=(128, 128), e_units=(128, 128), f_units=(128, 128), h_units=(128, 128))),
                ("[256] x 3", dict(g_units=(256,) * 3, e_units=(256,) * 3, f_units=(256,) * 3, h_units=(256,) * 3))):
    m = OB.init_model(0, z_dims, p, False, **u)
    eng = BnnEngine(p, z_dims, False, max_batch=64, norm_mode=1, **{k: list(v) for k, v in u.items()})
    eng.begin(m)
    dev = eng.device
    g = torch.Generator(device=dev); g.manual_seed(0)
    v = torch.randn(N, p, device=dev, generator=g); x = torch.rand(N, device=dev, generator=g); y = torch.randn(N, device=dev, generator=g)
    state = torch.empty(N, q, device=dev)
    eng.mh_run(x, y, v, state, bs, 0, 2, 0, 1.0, 1, init=True)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    eng.mh_run(x, y, v, state, bs, 2, iters, 0, 1.0, 1)
    torch.cuda.synchronize(); t = (time.perf_counter() - t0) / iters
    macs = sum(a * b for net in ("g", "f", "h") for a, b in zip(OB.net_dims(m[net])[:-1], OB.net_dims(m[net])[1:]))
    macs_f = sum(a * b for a, b in zip(OB.net_dims(m["f"])[:-1], OB.net_dims(m["f"])[1:]))
    fl = 2 * 2 * 2 * macs * N
    print("bnw %s N=%d: MH iteration %.2f ms = %.1f TFLOP/s = %.3f of the fp32-MFMA peak (Flipout: 4 x %d MAC per row-transition)"
          % (name, N, 1e3 * t, fl / t / 1e12, fl / t / 1e12 / 157.3, macs), flush=True)
    xs = torch.linspace(0, 3, 20, device=dev)
    adrf = torch.zeros(20, iters, device=dev, dtype=torch.float64)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    eng.mh_run(x, y, v, state, bs, 100, iters, 100, 1.0, 1, n_keep=iters, effect=1, x_values=xs, adrf_sum=adrf)
    torch.cuda.synchronize(); t2 = (time.perf_counter() - t0) / iters
    fe = 20 * 2 * 2 * macs_f * N
    print("bnw %s N=%d: kept iteration (20 doses) %.2f ms; outcome-net part %.2f ms = %.1f TFLOP/s = %.3f of peak"
          % (name, N, 1e3 * t2, 1e3 * (t2 - t), fe / max(t2 - t, 1e-9) / 1e12, fe / max(t2 - t, 1e-9) / 1e12 / 157.3), flush=True)
    eng.close()
