"""Accuracy experiments on the tutorial setting (docs/source/causalbgm/tutorial_py.ipynb of the reference: Hirano-Imbens
N = 20000, p = 200, 30000 EGM iterations + 100 epochs, predict 5000 + 3000 at 20 doses, bs = 20000).  Every run prints the
same log lines the reference prints (EGM every 500 iterations, MSE_x / MSE_y / MSE_v every 10 epochs), so a run can be laid
next to the published trace (tests/golden/tutorial_trace.json), and ends with one JSON line.

usage: python scripts/accuracy_runs.py NAME [key=value ...]
  keys: use_bnn (0|1), bnn_norm, seed, data_seed, N, epochs, egm, q_sd, lr_theta, lr_z, kl_weight, batch, bs, n_mcmc, burn_in"""
import json
import sys
import time

import numpy as np

sys.path.insert(0, ".")
from bayesgm_amd.models import CausalBGM
from bayesgm_amd.datasets import Sim_Hirano_Imbens_sampler
from bayesgm_amd.utils import get_ADRF

name = sys.argv[1]
kv = dict(a.split("=", 1) for a in sys.argv[2:])
N = int(float(kv.get("N", 20000)))
use_bnn = bool(int(kv.get("use_bnn", 1)))
seed = int(kv.get("seed", 123))
x, y, v = Sim_Hirano_Imbens_sampler(N=N, v_dim=200, seed=int(kv.get("data_seed", 0))).load_all()
params = dict(dataset="Sim_Hirano_Imbens", output_dir="gpurun_out/acc", save_res=False, save_model=False,
              binary_treatment=False, use_bnn=use_bnn, z_dims=[1, 1, 1, 7], v_dim=200, lr_theta=float(kv.get("lr_theta", 1e-4)),
              lr_z=float(kv.get("lr_z", 1e-4)), g_units=[64] * 5, f_units=[64, 32, 8], h_units=[64, 32, 8], e_units=[64] * 5,
              dz_units=[64, 32, 8], kl_weight=float(kv.get("kl_weight", 1e-4)), lr=2e-4, g_d_freq=5, use_z_rec=True)
for k in ("bnn_norm", "disc_norm"):
    if k in kv:
        params[k] = kv[k]
model = CausalBGM(params, random_seed=seed)
t0 = time.time()
egm = int(kv.get("egm", 30000))
model.fit((x, y, v), epochs=int(kv.get("epochs", 100)), epochs_per_eval=10, batch_size=int(kv.get("batch", 32)),
          use_egm_init=egm > 0, egm_n_iter=egm, egm_batches_per_eval=500, verbose=1)
t_fit = time.time() - t0
xs = np.linspace(0, 3, 20)
truth = get_ADRF(x_values=list(xs), dataset="Imbens")
variants = [("q_sd=%g" % float(kv.get("q_sd", 1.0)), float(kv.get("q_sd", 1.0)), "fp32")]
if int(kv.get("extra", 1)):
    variants.append(("q_sd adaptive", -1.0, "fp32"))
    if not use_bnn:
        variants.append(("bf16x3 q_sd=%g" % float(kv.get("q_sd", 1.0)), float(kv.get("q_sd", 1.0)), "bf16x3"))
preds = []
for label, q_sd, prec in variants:
    if not use_bnn:
        model.engine.set_precision(prec)
    t0 = time.time()
    adrf, interval = model.predict((x, y, v), alpha=0.01, n_mcmc=int(kv.get("n_mcmc", 3000)), burn_in=int(kv.get("burn_in", 5000)),
                                   x_values=xs, q_sd=q_sd, bs=int(kv.get("bs", 20000)))
    t_pred = time.time() - t0
    preds.append(dict(variant=label, predict_s=t_pred, adrf_rmse=float(np.sqrt(np.mean((adrf - truth) ** 2))),
                      adrf_mape=float(np.mean(np.abs((adrf - truth) / truth))),
                      interval_coverage=float(np.mean((interval[:, 0] <= truth) & (truth <= interval[:, 1]))),
                      acceptance=float(model.last_acceptance_rate), adrf=[float(a) for a in adrf]))
    print("PREDICT " + json.dumps(preds[-1]))
if not use_bnn:
    model.engine.set_precision("fp32")
first = preds[0]
print("RESULT " + json.dumps(dict(name=name, args=kv, fit_s=t_fit, predict_s=first["predict_s"], adrf_rmse=first["adrf_rmse"],
                                   adrf_mape=first["adrf_mape"], interval_coverage=first["interval_coverage"],
                                   acceptance=first["acceptance"], adrf=first["adrf"], truth=[float(t) for t in truth],
                                   variants=preds)))
