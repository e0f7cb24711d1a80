"""Model / data of scripts/dp_egm_smoke.py, shared with the single-process side of tests/test_gpu_egm.py."""
import numpy as np
from bayesgm_amd.models import CausalBGM
from bayesgm_amd.datasets import Sim_Hirano_Imbens_sampler

N_ITER, BATCH, PER_EVAL = 12, 32, 6
DATA = Sim_Hirano_Imbens_sampler(N=1505, v_dim=50, seed=1).load_all()


def build(use_bnn, dev):
    params = dict(dataset="dp_egm", output_dir="gpurun_out/dp", save_res=False, save_model=False, binary_treatment=False, use_bnn=use_bnn,
                  z_dims=[1, 1, 1, 7], v_dim=50, lr_theta=1e-3, lr_z=1e-3, g_units=[64] * 5, f_units=[64, 32, 8], h_units=[64, 32, 8],
                  e_units=[64] * 5, dz_units=[64, 32, 8], kl_weight=1e-4, lr=2e-4, g_d_freq=5, use_z_rec=True)
    return CausalBGM(params, random_seed=2, device=dev)


def flat_weights(m):
    """Every parameter of g, e, f, h as one float32 vector."""
    if isinstance(m.nets["g"], dict):                     # Bayesian networks: the session's flat parameter vector
        return np.asarray(m.engine.read(0), np.float32).copy()
    return np.concatenate([np.concatenate([np.asarray(w, np.float32).ravel(), np.asarray(b, np.float32).ravel()])
                           for k in ("g", "e", "f", "h") for (w, b) in m.nets[k]])
