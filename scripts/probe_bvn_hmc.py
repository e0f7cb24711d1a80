"""BGM(use_bnn=True) HMC kernel alone at BASELINE config C4's shape (p = 500, q = 10, L = 10), frozen noise (the shipped default):
register-chained row tiles (bgmf_kernels.h; default) vs the LDS-tile engine (gx_flipout.h; BGM_BVN_NO_CHAINS=1) vs the workspace kernel
(BGM_BVN_NO_TILES=1).  python scripts/probe_bvn_hmc.py [N] [iters] [frozen|fresh]
fresh = params['bnn_mcmc_noise'] = 'fresh' (the reference as written: a new perturbation per gradient evaluation): row-tile chains with a linear
stream (default) vs the workspace kernel (BGM_BVN_NO_CHAINS=1)."""
import json, os, sys, time
import numpy as np, torch
sys.path.insert(0, ".")
from bayesgm_amd.bvn_engine import BvnEngine
from oracle import bgm_bnn as OV
N = int(float(sys.argv[1])) if len(sys.argv) > 1 else 400000
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 6
fresh = len(sys.argv) > 3 and sys.argv[3] == "fresh"
p, q, L = 500, 10, 10
net = OV.init_vnet(np.random.RandomState(0), q, (64,) * 5, p)
eng = BvnEngine(p, q, g_units=[64] * 5, hmc_frozen_noise=not fresh)
eng.begin(net)
dev = eng.device
x = torch.randn(N, p, device=dev); x[torch.rand(N, p, device=dev) < 0.1] = float("nan")
state = torch.empty((N, q), device=dev); logp = torch.empty(N, device=dev); grad = torch.empty((N, q), device=dev)
step = torch.full((1,), 0.01, device=dev)
eng.hmc_run(x, state, logp, grad, step, 0, 1, 2 ** 30, L, 1, init=True)
torch.cuda.synchronize(); t0 = time.perf_counter()
eng.hmc_run(x, state, logp, grad, step, 1, iters, 2 ** 30, L, 1)
torch.cuda.synchronize(); dt = time.perf_counter() - t0
macs = q * 64 + 4 * 4096 + 2 * 64 * p
flop = L * 4 * macs * 2 * N * iters          # L gradient evaluations (forward + backward to the input) of TWO products per Flipout layer
print(json.dumps(dict(noise="fresh" if fresh else "frozen", kernel="workspace (bgmb_hmc_kernel)" if (os.environ.get("BGM_BVN_NO_TILES") or (fresh and os.environ.get("BGM_BVN_NO_CHAINS"))) else "LDS tiles (gxf_bgm_hmc_kernel)" if os.environ.get("BGM_BVN_NO_CHAINS") else "row-tile chains (bgmf_hmc_kernel)", N=N, iters=iters,
                      ms_per_transition=1e3 * dt / iters, transitions_per_s=N * iters / dt, tflops=flop / dt / 1e12, frac_of_157_3=flop / dt / 1e12 / 157.3)))
