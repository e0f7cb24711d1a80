"""CausalBGM(use_bnn=True).predict at the bench panel size, a tenth of the bench's iterations (what bench.py's bayesian leg runs)."""
import sys, time
import numpy as np, torch
sys.path.insert(0, ".")
from bayesgm_amd.models import CausalBGM
from bayesgm_amd.datasets import Sim_Hirano_Imbens_sampler
N = int(float(sys.argv[1])) if len(sys.argv) > 1 else 1000000
burn, keep = (int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (500, 300)
p = 200
params = dict(dataset="bench", output_dir="/tmp/bgm_bench", save_model=False, save_res=False, binary_treatment=False, use_bnn=True,
              z_dims=[1, 1, 1, 7], v_dim=p, lr_theta=1e-4, lr_z=1e-4, lr=2e-4, g_d_freq=5, use_z_rec=True, kl_weight=1e-4,
              g_units=[64] * 5, e_units=[64] * 5, f_units=[64, 32, 8], h_units=[64, 32, 8], dz_units=[64, 32, 8])
x, y, v = Sim_Hirano_Imbens_sampler(N=N, v_dim=p, seed=0).load_all()
m = CausalBGM(params, timestamp="chk", random_seed=0)
xv = np.linspace(0, 3, 20)
m.predict((x, y, v), alpha=0.01, n_mcmc=2, burn_in=2, x_values=xv, q_sd=1.0, bs=10000, verbose=0)
torch.cuda.synchronize(); t = time.perf_counter()
adrf, iv = m.predict((x, y, v), alpha=0.01, n_mcmc=keep, burn_in=burn, x_values=xv, q_sd=1.0, bs=10000, verbose=0)
torch.cuda.synchronize(); dt = time.perf_counter() - t
print("predict %.2f s -> %.3e transitions/s; acceptance %.4f; adrf[:3] %s" % (dt, N * (burn + keep) / dt, m.last_acceptance_rate, adrf[:3]))
