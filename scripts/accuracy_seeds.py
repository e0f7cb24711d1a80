"""Accuracy half of the metric on the reference's SHIPPED default (use_bnn=True) at the headline size, more than one seed
(VERDICT r5 item 8): the whole job -- egm_init (30000 iterations) + fit(epochs) + predict(5000 + 3000, 20 doses) -- on the
Hirano-Imbens panel N = 1e6, p = 200 for each seed; ADRF RMSE and average-effect error against the analytic curve, whether
diagnostics.SecondOptimumWarning fired, seconds per phase.  One JSON object on stdout.
usage: python scripts/accuracy_seeds.py [epochs=100] [seed ...]      (about 5 minutes per seed on one MI355X)"""
import json, os, sys, types
import numpy as np
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
import torch
import bench

epochs = int(sys.argv[1]) if len(sys.argv) > 1 else 100
seeds = [int(a) for a in sys.argv[2:]] or [123, 7, 2026]
n, p, z_dims = int(float(os.environ.get("BGM_ACC_ROWS", "1e6"))), 200, [1, 1, 1, 7]
dev = torch.device("cuda", 0)
x, y, v = bench.make_panel(n, p, seed=0, device=dev)
params = dict(dataset="Sim_Hirano_Imbens", output_dir=".", save_res=False, save_model=False, binary_treatment=False, use_bnn=True,
              z_dims=z_dims, v_dim=p, lr_theta=1e-4, lr_z=1e-4, g_units=[64] * 5, f_units=[64, 32, 8], h_units=[64, 32, 8], kl_weight=1e-4,
              lr=2e-4, g_d_freq=5, use_z_rec=True, e_units=[64] * 5, dz_units=[64, 32, 8])
args = types.SimpleNamespace(n_mcmc=3000, burn_in=5000, p=p)
xs = np.linspace(0, 3, 20)
runs = []
for s in seeds:
    out, m = bench.end_to_end_leg(params, x, y, v, xs, n, args, use_bnn=True, epochs=epochs, tag="seed%d" % s, seed=s)
    runs.append(out)
    print(json.dumps({k: out[k] for k in ("random_seed", "adrf_rmse", "average_effect_abs_error", "second_optimum_warning", "seconds")}), file=sys.stderr, flush=True)
    del m
    torch.cuda.empty_cache()
ae = [r["average_effect_abs_error"] for r in runs]
rm = [r["adrf_rmse"] for r in runs]
print(json.dumps({"model": "CausalBGM(use_bnn=True), product defaults", "rows": n, "p": p, "epochs": epochs, "seeds": seeds,
                  "average_effect_abs_error": {"values": ae, "median": float(np.median(ae)), "min": min(ae), "max": max(ae)},
                  "adrf_rmse": {"values": rm, "median": float(np.median(rm)), "min": min(rm), "max": max(rm)},
                  "second_optimum_warnings": [r["second_optimum_warning"] for r in runs], "runs": runs}))
