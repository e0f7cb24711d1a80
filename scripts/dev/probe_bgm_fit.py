"""Timing of BGM.fit's minibatch loop (bgm/base.py:399-413) per minibatch.  usage: probe_bgm_fit.py [N] [p] [q]"""
import sys, time, io, contextlib, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import numpy as np, torch
from bayesgm_amd.models import BGM
N = int(float(sys.argv[1])) if len(sys.argv) > 1 else 20000
p = int(sys.argv[2]) if len(sys.argv) > 2 else 100
q = int(sys.argv[3]) if len(sys.argv) > 3 else 10
params = dict(dataset="t", output_dir="/tmp/bgmfit", save_res=False, save_model=False, use_bnn=False, z_dim=q, x_dim=p, lr_theta=1e-4,
              lr_z=1e-4, g_units=[64] * 5, e_units=[64] * 5, dz_units=[64, 32, 8], dx_units=[64, 32, 8], alpha=0.0, beta=0.0, gamma=10.0, g_d_freq=1, kl_weight=1e-4,
              lr=1e-4, use_z_rec=True)
rs = np.random.RandomState(0)
x = rs.randn(N, p).astype(np.float32)
m = BGM(params, random_seed=1)
for ep in (0, 0, 4):      # the first call warms up (allocations, module load)
    torch.cuda.synchronize(); t0 = time.time()
    with contextlib.redirect_stdout(io.StringIO()):
        m.fit(x, batch_size=32, epochs=ep, epochs_per_eval=1000, use_egm_init=False, verbose=0)
    torch.cuda.synchronize(); dt = time.time() - t0
    print(f"epochs={ep + 1}: {dt:.3f} s")
    if ep: print(f"N={N} p={p}: {1e6 * (dt - d0) / (ep * (N // 32)):.1f} us per minibatch (incl. evaluation at the start of the call)")
    d0 = dt
