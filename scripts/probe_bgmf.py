"""Timing of the frozen-noise HMC kernel of BGM with the Bayesian generator at BASELINE config C4's shape (p=500, q=10, 5 x 64) on one
GPU (random-init posterior), fp32 (bgmf_hmc_kernel) and split precision (bgmfx_hmc_kernel).
   python scripts/probe_bgmf.py [n_rows] [n_iters]"""
import sys, time, os, json
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "tests"))
import torch
from bayesgm_amd.bvn_engine import BvnEngine
from oracle import bgm_bnn as OV      # (initial parameters only)


def main():
    n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 200000
    iters = int(sys.argv[2]) if len(sys.argv) > 2 else 10
    q, p, units, L = 10, 500, (64,) * 5, 10
    net = OV.init_vnet(np.random.RandomState(0), q, list(units), p)
    res = {}
    x = torch.randn(n, p, device="cuda")
    x[torch.rand(n, p, device="cuda") < 0.1] = float("nan")
    for mode in os.environ.get("BGM_PROBE_MODES", "fp32,f16x3").split(","):
        eng = BvnEngine(p, q, g_units=units, hmc_frozen_noise=True)
        eng.begin(net)
        eng.set_precision(mode)
        state = torch.zeros(n, q, device="cuda"); logp = torch.zeros(n, device="cuda"); grad = torch.zeros(n, q, device="cuda")
        step = torch.full((1,), 0.01, device="cuda")
        acc = torch.zeros(iters + 1, device="cuda", dtype=torch.int32)
        eng.hmc_run(x, state, logp, grad, step, 0, 1, 2 ** 30, L, 1, init=True, acc_count=acc)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        eng.hmc_run(x, state, logp, grad, step, 1, iters, 2 ** 30, L, 1, acc_count=acc)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        res[mode] = dict(n=n, iters=iters, s=dt, ms_per_transition=1e3 * dt / iters, transitions_per_s=n * iters / dt,
                         accept=float(acc[1:].sum().item()) / (n * iters))
        eng.close()
    print(json.dumps(res))


main()
