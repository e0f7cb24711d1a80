"""BGM.predict at a C4-like shape for profiling:  python scripts/profile_bgm_predict.py N p burn n_mcmc"""
import sys, time, json
import numpy as np, torch
import os; sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from bayesgm_amd.models import BGM
N, p, burn, n_mcmc = int(float(sys.argv[1])), int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
bp = dict(dataset="t", output_dir="gpurun_out/sec", save_res=False, save_model=False, use_bnn=False, z_dim=10, x_dim=p,
          lr_theta=5e-3, lr_z=5e-3, g_units=[64] * 5, e_units=[64] * 5, dz_units=[64, 32, 8], dx_units=[64, 32, 8],
          kl_weight=5e-5, lr=1e-3, g_d_freq=1, use_z_rec=True, alpha=0.0, gamma=0.0)
bm = BGM(bp, random_seed=0)
rs = np.random.RandomState(0)
data = rs.randn(N, p).astype(np.float32)
data[rs.rand(N, p) < 0.1] = np.nan
bm.predict(data[:2000], n_mcmc=10, burn_in=10)
torch.cuda.synchronize(); t0 = time.time()
imp, interval = bm.predict(data, n_mcmc=n_mcmc, burn_in=burn)
torch.cuda.synchronize(); dt = time.time() - t0
print(json.dumps(dict(N=N, p=p, burn=burn, n_mcmc=n_mcmc, predict_s=dt, timing=getattr(bm, "last_predict_timing", None))))
