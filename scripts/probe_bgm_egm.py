"""Latency of the native BGM EGM steps:  python scripts/probe_bgm_egm.py [p] [iters]"""
import sys, os, time, json
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from bayesgm_amd.models import BGM
p = int(sys.argv[1]) if len(sys.argv) > 1 else 20
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 500
bp = dict(dataset="t", output_dir="gpurun_out/begm", save_res=False, save_model=False, use_bnn=False, z_dim=10, x_dim=p,
          lr_theta=5e-3, lr_z=5e-3, g_units=[64] * 5, e_units=[64] * 5, dz_units=[64, 32, 8], dx_units=[64, 32, 8],
          kl_weight=5e-5, lr=1e-3, g_d_freq=1, use_z_rec=True, alpha=0.0, gamma=0.0)
m = BGM(bp, random_seed=0)
data = np.random.RandomState(0).randn(5000, p).astype(np.float32)
t0 = time.time()
m.egm_init(data, egm_n_iter=iters, egm_batches_per_eval=iters, verbose=0)
torch.cuda.synchronize()
dt = time.time() - t0
print(json.dumps(dict(p=p, iters=iters, s=dt, ms_per_iteration=1e3 * dt / iters, est_20000_iters_s=20000 * dt / iters)))
