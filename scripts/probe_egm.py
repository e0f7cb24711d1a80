import sys, time; sys.path.insert(0, ".")
import numpy as np
from bayesgm_amd.models import CausalBGM
from bayesgm_amd.datasets import Sim_Hirano_Imbens_sampler
x, y, v = Sim_Hirano_Imbens_sampler(N=5000, v_dim=200, seed=0).load_all()
params = dict(dataset="t", output_dir="gpurun_out/egm", save_res=False, save_model=False, binary_treatment=False, use_bnn=False,
              z_dims=[1, 1, 1, 7], v_dim=200, lr_theta=1e-4, lr_z=1e-4, g_units=[64] * 5, f_units=[64, 32, 8], h_units=[64, 32, 8],
              e_units=[64] * 5, dz_units=[64, 32, 8], kl_weight=1e-4, lr=2e-4, g_d_freq=5, use_z_rec=True)
m = CausalBGM(params, random_seed=1)
t0 = time.time(); m.egm_init((x, y, v), egm_n_iter=1000, egm_batches_per_eval=250); print("egm 1000 iters: %.1f s" % (time.time() - t0))
