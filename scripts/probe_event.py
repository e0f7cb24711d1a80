"""Retained phase of CausalBGM.predict at the bench shape in the three outcome-cache modes (off / per wave / per chain = event form):
seconds per predict, keep-phase interval (HIP events inside the library), served fractions, ADRF equality.
    python scripts/probe_event.py [rows] [budget_mb] [q_sd]"""
import json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from bayesgm_amd.models import CausalBGM
from bayesgm_amd.datasets import Sim_Hirano_Imbens_sampler
n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 1000000
budget = int(sys.argv[2]) if len(sys.argv) > 2 else 0
q_sd = float(sys.argv[3]) if len(sys.argv) > 3 else 1.0
burn, keep = int(os.environ.get("BURN", 5000)), int(os.environ.get("KEEP", 3000))
p, z_dims = 200, [1, 1, 1, 7]
params = dict(dataset="Sim_Hirano_Imbens", output_dir=".", save_res=False, save_model=False, binary_treatment=False, use_bnn=False,
              z_dims=z_dims, v_dim=p, lr_theta=1e-4, lr_z=1e-4, g_units=[64] * 5, f_units=[64, 32, 8], h_units=[64, 32, 8], kl_weight=1e-4,
              lr=2e-4, g_d_freq=5, use_z_rec=True, e_units=[64] * 5, dz_units=[64, 32, 8])
m = CausalBGM(params, timestamp="probe", random_seed=0)
eng = m.engine
x, y, v = Sim_Hirano_Imbens_sampler(N=n, v_dim=p, seed=0).load_all()
data = tuple(torch.from_numpy(a).cuda() for a in (x, y, v))
xs = np.linspace(0, 3, 20)
if budget:
    eng.set_event_budget(budget << 20)
eng.set_precision(os.environ.get("PREC", "fp32"))       # PREC=bf16x3 | f16x3: split-precision transitions
res = {}
for mode in (False, "wave", True, True):
    eng.set_outcome_cache(mode)
    m._seed_counter = 7
    eng.outcome_cache_stats(reset=True)
    eng.timing_enable(True); eng.timing_read(kind=-1, reset=True)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    adrf, _ = m.predict(data, alpha=0.01, n_mcmc=keep, burn_in=burn, x_values=xs, q_sd=q_sd, sample_y=True, verbose=0)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    nb, msb = eng.timing_read(kind=0, reset=False); nk, msk = eng.timing_read(kind=1, reset=True); eng.timing_enable(False)
    served, total = eng.outcome_cache_stats()
    res[str(mode)] = dict(seconds=dt, burn_ms=msb, keep_ms=msk, keep_intervals=nk, served=served, total=total, acc=m.last_acceptance_rate, adrf=np.asarray(adrf))
    print(mode, "predict %.3f s  burn-in %.1f ms  keep %.1f ms (%d interval)  served %.4f  acceptance %.4f" % (dt, msb, msk, nk, served / max(1, total), m.last_acceptance_rate), flush=True)
print("max |ADRF(event) - ADRF(off)| =", float(np.abs(res["True"]["adrf"] - res["False"]["adrf"]).max()),
      " max |ADRF(wave) - ADRF(off)| =", float(np.abs(res["wave"]["adrf"] - res["False"]["adrf"]).max()))
