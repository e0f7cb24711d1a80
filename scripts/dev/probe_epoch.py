"""Per-minibatch cost of CausalBGM.fit with the loop in Python vs inside the library (development probe)."""
import sys, time, tempfile
import numpy as np, torch
sys.path.insert(0, ".")
from bayesgm_amd.models import CausalBGM
from bayesgm_amd.datasets import Sim_Hirano_Imbens_sampler
n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 20000
x, y, v = Sim_Hirano_Imbens_sampler(N=n, v_dim=200, seed=0).load_all()
for bnn in (False, True):
    for host_loop in (True, False):
        params = dict(dataset="t", output_dir=tempfile.mkdtemp(), save_res=False, save_model=False, binary_treatment=False, use_bnn=bnn,
                      z_dims=[1, 1, 1, 7], v_dim=200, lr_theta=1e-4, lr_z=1e-4, lr=2e-4, g_d_freq=5, use_z_rec=True, kl_weight=1e-4,
                      g_units=[64] * 5, e_units=[64] * 5, f_units=[64, 32, 8], h_units=[64, 32, 8], dz_units=[64, 32, 8])
        m = CausalBGM(params, timestamp="t", random_seed=1)
        m.fit((x, y, v), epochs=0, epochs_per_eval=1000, use_egm_init=False, verbose=0, host_loop=host_loop)
        torch.cuda.synchronize(); t0 = time.time()
        ep = 3
        m.fit((x, y, v), epochs=ep - 1, epochs_per_eval=1000, use_egm_init=False, verbose=0, host_loop=host_loop)
        torch.cuda.synchronize(); dt = time.time() - t0
        nb = ep * ((n + 31) // 32)
        print("use_bnn=%s host_loop=%s: %.1f us per minibatch (%d minibatches, %.2f s incl. one evaluation)" % (bnn, host_loop, dt / nb * 1e6, nb, dt), flush=True)
