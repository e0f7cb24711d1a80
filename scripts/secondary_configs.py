"""Throughput of the secondary BASELINE.json configurations on one MI355X (random-init weights).
C1: CausalBGM binary treatment N=1e5 p=100 z_dims [3,3,6,6] (cli defaults): predict (MH + ITE + per-row quantiles)
C0/BGM: BGM imputation, 10 % cells missing: N=2000 p=20 and N=1e5 p=100, HMC L=10.
C4 (one GPU's share of N=5e6 over 8 GPUs): BGM imputation N=625000 p=500, 1000 burn-in + 1000 retained draws.
FLOP accounting: L=10 gradient evaluations (forward+backward = 4 MACs(g)) per transition are EXECUTED (the
gradient at the current state is cached, as TFP does); SURVEY 8(d)'s figure of 44 MACs counts one more."""
import json, sys, time
import numpy as np, torch
sys.path.insert(0, ".")
from bayesgm_amd.models import CausalBGM, BGM
from bayesgm_amd.datasets import Sim_Hirano_Imbens_sampler, binarize_treatment

out = {}
N, p = 100000, 100
x, y, v = Sim_Hirano_Imbens_sampler(N=N, v_dim=p, seed=0).load_all()
xb = binarize_treatment(x)
params = dict(dataset="t", output_dir="gpurun_out/sec", save_res=False, save_model=False, binary_treatment=True, use_bnn=False,
              z_dims=[3, 3, 6, 6], v_dim=p, lr_theta=1e-4, lr_z=1e-4, g_units=[64] * 5, f_units=[64, 32, 8], h_units=[64, 32, 8],
              e_units=[64] * 5, dz_units=[64, 32, 8], kl_weight=1e-4, lr=2e-4, g_d_freq=5, use_z_rec=True)
m = CausalBGM(params, random_seed=0)
for rep in range(2):
    torch.cuda.synchronize(); t0 = time.time()
    ite, interval = m.predict((xb, y, v), alpha=0.01, n_mcmc=3000, burn_in=5000, verbose=0)
    torch.cuda.synchronize(); dt = time.time() - t0
served, total = m.engine.outcome_cache_stats(reset=True)
m.engine.set_outcome_cache(False)       # the same call with the outcome net evaluated at every retained draw (the reference's work)
m._seed_counter -= 1
torch.cuda.synchronize(); t0 = time.time()
ite_full, _ = m.predict((xb, y, v), alpha=0.01, n_mcmc=3000, burn_in=5000, verbose=0)
torch.cuda.synchronize(); dt_full = time.time() - t0
m.engine.set_outcome_cache(True)
out["C1_causal_binary_N1e5_p100"] = dict(predict_s=dt, transitions_per_s=N * 8000 / dt, acceptance=m.last_acceptance_rate,
                                         ate=float(ite.mean()), shapes=[list(ite.shape), list(interval.shape)],
                                         outcome_cache_served_fraction=served / max(1, total), predict_s_cache_off=dt_full,
                                         ite_identical_with_cache_off=bool(np.array_equal(ite, ite_full)))
cases = [(2000, 20, 5000, 5000, False), (100000, 100, 1000, 1000, False)]
if "--only-c4" in sys.argv:
    cases = []
if "--c4" in sys.argv:
    cases.append((625000, 500, 1000, 1000, False))
if "--c4-bnn" in sys.argv:          # the same share with the Bayesian generator (use_bnn=True, frozen noise: bgmf_hmc_kernel), two products per layer
    cases.append((625000, 500, 1000, 1000, True))
for (N, p, n_mcmc, burn, use_bnn) in cases:
    bp = dict(dataset="t", output_dir="gpurun_out/sec", save_res=False, save_model=False, use_bnn=use_bnn, z_dim=10, x_dim=p,
              lr_theta=5e-3, lr_z=5e-3, g_units=[64] * 5, e_units=[64] * 5, dz_units=[64, 32, 8], dx_units=[64, 32, 8],
              kl_weight=5e-5, lr=1e-3, g_d_freq=1, use_z_rec=True, alpha=0.0, gamma=0.0)
    bm = BGM(bp, random_seed=0)
    rs = np.random.RandomState(0)
    data = rs.randn(N, p).astype(np.float32)
    data[rs.rand(N, p) < 0.1] = np.nan
    for rep in range(1 if N > 200000 else 2):
        torch.cuda.synchronize(); t0 = time.time()
        imp, interval = bm.predict(data, n_mcmc=n_mcmc, burn_in=burn)
        torch.cuda.synchronize(); dt = time.time() - t0
    macs = 10 * 64 + 4 * 4096 + 2 * 64 * p
    out[f"BGM_impute_N{N}_p{p}" + ("_use_bnn" if use_bnn else "")] = dict(predict_s=dt, hmc_transitions_per_s=N * (n_mcmc + burn) / dt,
                                        tflops_executed=N * (n_mcmc + burn) * 40 * macs * (2 if use_bnn else 1) / dt / 1e12,
                                        acceptance=bm.last_acceptance_rate)
print(json.dumps(out))
