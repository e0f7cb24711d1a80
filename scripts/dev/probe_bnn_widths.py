"""Which hidden widths do the Bayesian-network classes accept?  (development probe)"""
import sys, traceback, tempfile
import numpy as np
sys.path.insert(0, ".")
from bayesgm_amd.models import CausalBGM, BGM
from bayesgm_amd.datasets import simulate_z_hetero

def causal(units, dz, binary):
    rs = np.random.RandomState(0)
    n, p = 96, 12
    v = rs.randn(n, p).astype(np.float32)
    x = ((rs.rand(n, 1) > 0.5) if binary else rs.exponential(size=(n, 1))).astype(np.float32)
    y = (x + 0.3 * v[:, :1] + rs.randn(n, 1)).astype(np.float32)
    params = dict(dataset="t", output_dir=tempfile.mkdtemp(), save_res=False, save_model=False, binary_treatment=binary, use_bnn=True,
                  z_dims=[1, 1, 1, 3], v_dim=p, lr_theta=1e-4, lr_z=1e-4, lr=2e-4, g_d_freq=5, use_z_rec=True, kl_weight=1e-4,
                  g_units=units["g"], e_units=units["e"], f_units=units["f"], h_units=units["h"], dz_units=dz)
    m = CausalBGM(params, timestamp="t", random_seed=1)
    m.fit((x, y, v), epochs=2, epochs_per_eval=1, batch_size=32, use_egm_init=True, egm_n_iter=10, egm_batches_per_eval=5, verbose=0)
    kw = {} if binary else dict(x_values=[0.0, 1.0, 2.0])
    eff, iv = m.predict((x, y, v), n_mcmc=5, burn_in=10, q_sd=0.5, **kw)
    return np.isfinite(eff).all()

def bgm(units, e_units, d):
    X, Y = simulate_z_hetero(n=128, k=3, d=7, seed=42)
    data = np.c_[X, Y].astype(np.float32)
    params = dict(dataset="t", output_dir=tempfile.mkdtemp(), save_res=False, save_model=False, use_bnn=True, z_dim=3, x_dim=8,
                  lr_theta=5e-3, lr_z=5e-3, g_units=units, e_units=e_units, dz_units=d, dx_units=d, kl_weight=5e-5, lr=1e-3,
                  g_d_freq=1, use_z_rec=True, alpha=0.0, gamma=0.0)
    model = BGM(params, random_seed=1)
    model.fit(data, batch_size=32, epochs=2, epochs_per_eval=1, use_egm_init=True, egm_n_iter=10, egm_batches_per_eval=5, verbose=0)
    test = data[:32].copy(); test[:, -1] = np.nan
    imp, iv = model.predict(test, alpha=0.05, bs=16, n_mcmc=5, burn_in=10, step_size=0.01, num_leapfrog_steps=3, seed=42)
    return np.isfinite(imp).all()

for name, u, dz in (("r_test", dict(g=[8, 8], e=[8, 8], f=[8, 4], h=[8, 4]), [8, 4]),
                    ("w64x2", dict(g=[64, 64], e=[64, 64], f=[64, 64], h=[64, 64]), [64, 32]),
                    ("w128", dict(g=[128, 128], e=[128, 128], f=[128, 128], h=[128, 128]), [128, 128]),
                    ("w256", dict(g=[256] * 3, e=[256] * 3, f=[256] * 3, h=[256] * 3), [256] * 3)):
    for binary in (True, False):
        try:
            print("causal", name, binary, causal(u, dz, binary), flush=True)
        except Exception as e:
            print("causal", name, binary, "FAILED:", type(e).__name__, str(e)[:300], flush=True)
    try:
        print("bgm", name, bgm(u["g"], u["e"], dz), flush=True)
    except Exception as e:
        print("bgm", name, "FAILED:", type(e).__name__, str(e)[:300], flush=True)
