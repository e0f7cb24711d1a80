"""Timing of the BGM HMC kernel at BASELINE config C4's shape (p=500, q=10) on one GPU (random-init weights).
   python scripts/probe_bgm_wide.py [n_rows] [n_iters]"""
import sys, time, os, json
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import torch
from bayesgm_amd.engine import BgmEngine

def glorot(rs, a, b):
    l = np.sqrt(6.0 / (a + b)); return rs.uniform(-l, l, (a, b)).astype(np.float32)

def main():
    n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 200000
    iters = int(sys.argv[2]) if len(sys.argv) > 2 else 20
    res = {}
    for p in (100, 500):
        q = 10
        rs = np.random.RandomState(0)
        g = {"bn": dict(gamma=np.ones(q, np.float32), beta=np.zeros(q, np.float32), mean=np.zeros(q, np.float32), var=np.ones(q, np.float32)),
             "trunk": [(glorot(rs, q if i == 0 else 64, 64), np.zeros(64, np.float32)) for i in range(5)],
             "mean": (glorot(rs, 64, p), np.zeros(p, np.float32)), "var": (glorot(rs, 64, p), np.zeros(p, np.float32))}
        eng = BgmEngine(p, q, g_units=[64] * 5)
        eng.set_weights(g)
        if os.environ.get("BGM_PROBE_PRECISION"):      # "f16x3": split-precision heads (csrc/bgm_kernels.h)
            eng.set_precision(os.environ["BGM_PROBE_PRECISION"])
        x = torch.randn(n, p, device="cuda")
        x[torch.rand(n, p, device="cuda") < 0.1] = float("nan")
        L = 10
        # raw kernel timing through hmc_run (no adaptation launches)
        state = torch.zeros(n, q, device="cuda"); logp = torch.zeros(n, device="cuda"); grad = torch.zeros(n, q, device="cuda")
        step = torch.full((1,), 0.01, device="cuda")
        eng.hmc_run(x, state, logp, grad, step, 0, 1, 2**30, L, 1, init=True)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        eng.hmc_run(x, state, logp, grad, step, 1, iters, 2**30, L, 1)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        macs = q * 64 + 4 * 4096 + 2 * 64 * p
        flop = (L) * 4 * macs * n * iters     # L gradient evaluations (fwd+bwd) per transition (initial gradient cached)
        res[f"p{p}"] = dict(n=n, iters=iters, s=dt, transitions_per_s=n * iters / dt, tflops=flop / dt / 1e12)
    print(json.dumps(res))

main()
