import sys, time, numpy as np, torch
sys.path.insert(0, ".")
from bayesgm_amd.models import CausalBGM
from bayesgm_amd.datasets import Sim_Hirano_Imbens_sampler, binarize_treatment
N, p = 100000, 100
x, y, v = Sim_Hirano_Imbens_sampler(N=N, v_dim=p, seed=0).load_all()
x = binarize_treatment(x)
params = dict(dataset="c1", output_dir="gpurun_out/c1", save_res=False, save_model=False, binary_treatment=True, use_bnn=True,
              z_dims=[3, 3, 6, 6], v_dim=p, lr_theta=1e-4, lr_z=1e-4, g_units=[64] * 5, f_units=[64, 32, 8], h_units=[64, 32, 8],
              e_units=[64] * 5, dz_units=[64, 32, 8], kl_weight=1e-4, lr=2e-4, g_d_freq=5, use_z_rec=True)
import warnings; warnings.simplefilter("ignore")
m = CausalBGM(params, random_seed=1)
m.predict((x, y, v), alpha=0.01, n_mcmc=10, burn_in=10, q_sd=1.0, bs=1000, verbose=0)
torch.cuda.synchronize(); t = time.time()
ite, itv = m.predict((x, y, v), alpha=0.01, n_mcmc=3000, burn_in=5000, q_sd=1.0, bs=1000, verbose=1)
torch.cuda.synchronize(); dt = time.time() - t
print("C1 binary use_bnn predict N=%d p=%d bs=1000: %.1f s, %.3e transitions/s, ATE(untrained)=%.4f" % (N, p, dt, N * 8000 / dt, ite.mean()))
