"""Seconds of BGM.fit per minibatch and of egm_init per iteration at C4's shape (x_dim 500, z_dim 10, g_units [64] x 5), deterministic and
Bayesian generator.   usage: python scripts/probe_bgm_fit.py [N=20000]"""
import sys, time
import numpy as np
import torch
sys.path.insert(0, ".")
from bayesgm_amd.models import BGM

n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 20000
p, q = (int(sys.argv[2]) if len(sys.argv) > 2 else 500), (int(sys.argv[3]) if len(sys.argv) > 3 else 10)
rs = np.random.RandomState(0)
x = rs.randn(n, p).astype(np.float32)
for use_bnn in (False, True):
    bp = dict(dataset="probe", output_dir="/tmp", save_res=False, save_model=False, use_bnn=use_bnn, z_dim=q, x_dim=p, g_units=[64] * 5, e_units=[64] * 5,
              dz_units=[64, 32, 8], dx_units=[64, 32, 8], lr_theta=1e-4, lr_z=1e-4, lr=2e-4, g_d_freq=5, use_z_rec=True, kl_weight=1e-4, alpha=0.0, gamma=0.0)
    m = BGM(bp, timestamp="probe", random_seed=0)
    def timed(f):
        torch.cuda.synchronize(); t0 = time.perf_counter(); f(); torch.cuda.synchronize(); return time.perf_counter() - t0
    # slopes (the first call of each opens sessions and allocates): 300 -> 900 iterations, 1 -> 3 epochs
    m.egm_init(x, egm_n_iter=50, batch_size=32, egm_batches_per_eval=1000, verbose=0)
    t_e1 = timed(lambda: m.egm_init(x, egm_n_iter=300, batch_size=32, egm_batches_per_eval=1000, verbose=0))
    t_e2 = timed(lambda: m.egm_init(x, egm_n_iter=900, batch_size=32, egm_batches_per_eval=1000, verbose=0))
    t_f1 = timed(lambda: m.fit(x, batch_size=32, epochs=1, epochs_per_eval=10, use_egm_init=False, verbose=0))
    t_f2 = timed(lambda: m.fit(x, batch_size=32, epochs=3, epochs_per_eval=10, use_egm_init=False, verbose=0))
    print("BGM(use_bnn=%s) x_dim=%d N=%d: egm_init %.2f ms per iteration; fit %.1f us per minibatch (slopes)"
          % (use_bnn, p, n, 1e3 * (t_e2 - t_e1) / 600, 1e6 * (t_f2 - t_f1) / (2 * (n // 32))), flush=True)
