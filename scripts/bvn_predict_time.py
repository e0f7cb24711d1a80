"""End-to-end time of BGM(use_bnn=True): short fit, then predict (imputation with 10 % missing cells) at N = 20 000, p = 100.
usage: python scripts/bvn_predict_time.py"""
import sys, time, json
import numpy as np, torch
sys.path.insert(0, ".")
from bayesgm_amd.models import BGM
N, p, q = 20000, 100, 10
rs = np.random.RandomState(0)
data = (rs.standard_normal((N, q)) @ rs.standard_normal((q, p)) * 0.3 + 0.1 * rs.standard_normal((N, p))).astype(np.float32)
params = dict(dataset="t", output_dir="gpurun_out/bvnp", save_res=False, save_model=False, use_bnn=True, z_dim=q, x_dim=p, g_units=[64] * 5,
              e_units=[64] * 5, dz_units=[64, 32, 8], dx_units=[64, 32, 8], lr=1e-3, lr_theta=5e-3, lr_z=5e-3, g_d_freq=1, kl_weight=5e-5,
              gamma=0.0, alpha=0.0, bnn_mcmc_noise="frozen")
m = BGM(params, random_seed=1)
t0 = time.time(); m.fit(data, epochs=5, epochs_per_eval=5, use_egm_init=True, egm_n_iter=1000, egm_batches_per_eval=1000, verbose=0); torch.cuda.synchronize(); t_fit = time.time() - t0
miss = data.copy(); miss[rs.uniform(size=miss.shape) < 0.1] = np.nan
t0 = time.time(); imp, itv = m.predict(miss, n_mcmc=1000, burn_in=1000); torch.cuda.synchronize(); t_pred = time.time() - t0
print(json.dumps(dict(N=N, p=p, fit_s=t_fit, fit_steps=6 * (N // 32) + 2001, predict_s=t_pred, transitions_per_s=N * 2000 / t_pred, acceptance=m.last_acceptance_rate)))
