"""Helper of test_gpu_bnn.py::test_general_steps_over_the_chip_equal_the_one_launch_form: a short fit (EGM warm start + two passes) of
CausalBGM(use_bnn=True) at hidden widths outside the default shapes and of BGM(use_bnn=True), parameters and latents saved to argv[1].
Run twice by the test, with and without BGM_BNN_STEP_ONE_LAUNCH=1 (the switch is read once per process)."""
import sys
import numpy as np

sys.path.insert(0, ".")
from bayesgm_amd.models import BGM, CausalBGM

out = sys.argv[1]
rs = np.random.RandomState(0)
n, p = 200, 60
v = rs.randn(n, p).astype(np.float32)
x = rs.exponential(size=(n, 1)).astype(np.float32)
y = (x + 0.3 * v[:, :1] + rs.randn(n, 1)).astype(np.float32)
params = dict(dataset="t", output_dir="/tmp", save_res=False, save_model=False, binary_treatment=False, use_bnn=True, z_dims=[1, 1, 1, 7], v_dim=p,
              lr_theta=1e-3, lr_z=1e-3, lr=2e-4, g_d_freq=2, use_z_rec=True, kl_weight=1e-4, g_units=[128, 96], e_units=[100], f_units=[80, 40],
              h_units=[72], dz_units=[64, 32, 8])
m = CausalBGM(params, timestamp="t", random_seed=3)
m.fit((x, y, v), epochs=1, epochs_per_eval=5, batch_size=32, use_egm_init=True, egm_n_iter=6, egm_batches_per_eval=100, verbose=0)
res = {"causal_z": m.data_z.cpu().numpy(), "causal_theta": m.engine.read(0)}
xb = rs.randn(n, 40).astype(np.float32)
bp = dict(dataset="t", output_dir="/tmp", save_res=False, save_model=False, use_bnn=True, z_dim=5, x_dim=40, g_units=[64] * 3, e_units=[64] * 3,
          dz_units=[64, 32, 8], dx_units=[64, 32, 8], lr_theta=1e-3, lr_z=1e-3, lr=2e-4, g_d_freq=1, use_z_rec=True, kl_weight=1e-4, alpha=0.0, gamma=0.0)
b = BGM(bp, timestamp="t", random_seed=3)
b.fit(xb, batch_size=32, epochs=2, epochs_per_eval=5, use_egm_init=False, verbose=0)
res["bgm_z"] = b.data_z.cpu().numpy() if hasattr(b.data_z, "cpu") else np.asarray(b.data_z)
res["bgm_theta"] = b.engine.read(0)
np.savez(out, **res)
