"""Kernel split of BGM(use_bnn=True).predict at p = 500 (dev aid): python scripts/dev/probe_bvn_predict.py [N] [n_mcmc] [burn_in]"""
import sys, time, json
import numpy as np, torch
sys.path.insert(0, ".")
from bayesgm_amd.models import BGM
N = int(float(sys.argv[1])) if len(sys.argv) > 1 else 100000
n_mcmc = int(sys.argv[2]) if len(sys.argv) > 2 else 200
burn = int(sys.argv[3]) if len(sys.argv) > 3 else 200
p = 500
bp = dict(dataset="t", output_dir="gpurun_out/sec", save_res=False, save_model=False, use_bnn=True, z_dim=10, x_dim=p,
          lr_theta=5e-3, lr_z=5e-3, g_units=[64] * 5, e_units=[64] * 5, dz_units=[64, 32, 8], dx_units=[64, 32, 8],
          kl_weight=5e-5, lr=1e-3, g_d_freq=1, use_z_rec=True, alpha=0.0, gamma=0.0)
bm = BGM(bp, random_seed=0)
rs = np.random.RandomState(0)
data = rs.randn(N, p).astype(np.float32)
data[rs.rand(N, p) < 0.1] = np.nan
torch.cuda.synchronize(); t0 = time.time()
imp, interval = bm.predict(data, n_mcmc=n_mcmc, burn_in=burn)
torch.cuda.synchronize(); dt = time.time() - t0
print(json.dumps(dict(N=N, n_mcmc=n_mcmc, burn_in=burn, predict_s=dt, acceptance=bm.last_acceptance_rate)))
