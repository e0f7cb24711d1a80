"""End-to-end accuracy: CausalBGM fit + predict on Sim_Hirano_Imbens (the reference tutorial's setting,
docs/source/causalbgm/tutorial_py.ipynb: N=20000, p=200, 20 doses on [0,3]; reported ADRF RMSE 0.0188 / MAPE
0.0103 with EGM warm start + 100 epochs, use_bnn=True, package v1.0.1).
usage: python scripts/accuracy_e2e.py [N] [epochs] [batch] [egm_iters] [use_bnn 0|1] [bnn_norm batch|fixed]"""
import json, sys, time
import numpy as np
sys.path.insert(0, ".")
from bayesgm_amd.models import CausalBGM
from bayesgm_amd.datasets import Sim_Hirano_Imbens_sampler
from bayesgm_amd.utils import get_ADRF

N = int(float(sys.argv[1])) if len(sys.argv) > 1 else 20000
epochs = int(sys.argv[2]) if len(sys.argv) > 2 else 100
batch = int(sys.argv[3]) if len(sys.argv) > 3 else 32
egm_iters = int(sys.argv[4]) if len(sys.argv) > 4 else 0
use_bnn = bool(int(sys.argv[5])) if len(sys.argv) > 5 else False
bnn_norm = sys.argv[6] if len(sys.argv) > 6 else "batch"
x, y, v = Sim_Hirano_Imbens_sampler(N=N, v_dim=200, seed=0).load_all()
params = dict(dataset="Sim_Hirano_Imbens", output_dir="gpurun_out/e2e", save_res=False, save_model=False,
              binary_treatment=False, use_bnn=use_bnn, z_dims=[1, 1, 1, 7], v_dim=200, lr_theta=1e-4, lr_z=1e-4,
              g_units=[64] * 5, f_units=[64, 32, 8], h_units=[64, 32, 8], e_units=[64] * 5, dz_units=[64, 32, 8],
              kl_weight=1e-4, lr=2e-4, g_d_freq=5, use_z_rec=True, bnn_norm=bnn_norm)
model = CausalBGM(params, random_seed=123)
t0 = time.time()
model.fit((x, y, v), epochs=epochs, epochs_per_eval=max(1, epochs // 5), batch_size=batch, use_egm_init=egm_iters > 0,
          egm_n_iter=egm_iters, egm_batches_per_eval=max(1, egm_iters // 6), verbose=1)
t_fit = time.time() - t0
xs = np.linspace(0, 3, 20)
t0 = time.time()
adrf, interval = model.predict((x, y, v), alpha=0.01, n_mcmc=3000, burn_in=5000, x_values=xs, q_sd=1.0, verbose=1,
                               **(dict(bs=20000) if use_bnn else {}))          # the tutorial passes bs=20000
t_pred = time.time() - t0
truth = get_ADRF(x_values=list(xs), dataset="Imbens")
rmse = float(np.sqrt(np.mean((adrf - truth) ** 2)))
mape = float(np.mean(np.abs((adrf - truth) / truth)))
cover = float(np.mean((interval[:, 0] <= truth) & (truth <= interval[:, 1])))
if use_bnn:
    print(json.dumps(dict(N=N, epochs=epochs, batch=batch, egm_iters=egm_iters, use_bnn=True, bnn_norm=bnn_norm, fit_s=t_fit, fit_obs_per_s=N * (epochs + 1) / t_fit,
                          predict_s=t_pred, predict_transitions_per_s=N * 8000 / t_pred, adrf_rmse=rmse, adrf_mape=mape,
                          interval_coverage=cover, acceptance=model.last_acceptance_rate, adrf=[float(a) for a in adrf],
                          truth=[float(t) for t in truth])))
    sys.exit(0)
# oracle predict with the SAME trained weights / Philox streams on a row subset
from oracle import causal as OC
m = dict(g=model.nets["g"], f=model.nets["f"], h=model.nets["h"], e=model.nets["e"], z_dims=[1, 1, 1, 7], v_dim=200,
         binary_treatment=False)
sub = slice(0, 256)
model._seed_counter -= 0
seed = model._next_seed()
out = model.engine.mh_sample(x[sub].ravel(), y[sub].ravel(), v[sub], 300, 100, 1.0, seed, effect=1, x_values=xs)
ref_post = OC.mh_sampler(m, (x[sub], y[sub], v[sub]), 300, 100, 1.0, seed)
ref_eff = OC.infer_from_latent_posterior(m, ref_post, xs, True, seed, burn_in=300)
d_oracle = float(np.abs(out["adrf"].cpu().numpy().mean(axis=1) - ref_eff.mean(axis=1)).max())
print(json.dumps(dict(N=N, epochs=epochs, batch=batch, egm_iters=egm_iters, fit_s=t_fit, fit_obs_per_s=N * (epochs + 1) / t_fit, predict_s=t_pred,
                      predict_transitions_per_s=N * 8000 / t_pred, adrf_rmse=rmse, adrf_mape=mape, interval_coverage=cover,
                      acceptance=model.last_acceptance_rate, adrf_max_abs_diff_vs_oracle_256rows=d_oracle,
                      adrf=[float(a) for a in adrf], truth=[float(t) for t in truth])))
