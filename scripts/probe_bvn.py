"""Timing probe of BGM with the Bayesian generator (use_bnn=True): minibatch steps, EGM iteration, HMC transition, decode.
usage: python scripts/probe_bvn.py [N] [p] [q]"""
import json, sys, time
import numpy as np
import torch
sys.path.insert(0, ".")
from bayesgm_amd.models import BGM

N = int(float(sys.argv[1])) if len(sys.argv) > 1 else 2000
p = int(sys.argv[2]) if len(sys.argv) > 2 else 20
q = int(sys.argv[3]) if len(sys.argv) > 3 else 10
rs = np.random.RandomState(0)
data = (rs.standard_normal((N, q)) @ rs.standard_normal((q, p)) + 0.1 * rs.standard_normal((N, p))).astype(np.float32)
params = dict(dataset="probe", output_dir="gpurun_out/probe_bvn", save_res=False, save_model=False, use_bnn=True, z_dim=q, x_dim=p,
              g_units=[64] * 5, e_units=[64] * 5, dz_units=[64, 32, 8], dx_units=[64, 32, 8], lr=1e-3, lr_theta=5e-3, lr_z=5e-3,
              g_d_freq=1, kl_weight=5e-5, gamma=0.0, alpha=0.0)
m = BGM(params, random_seed=1)
eng = m.engine
dev = eng.device
sync = lambda: torch.cuda.synchronize(dev)
out = {}
t0 = time.perf_counter(); m.egm_init(data, egm_n_iter=400, batch_size=32, egm_batches_per_eval=400, verbose=0); sync()
out["egm_ms_per_iteration_incl_2_evals"] = (time.perf_counter() - t0) * 1e3 / 401
x = torch.from_numpy(data).to(dev)
z = torch.randn(N, q, device=dev)
idx = torch.arange(32, dtype=torch.int32, device=dev)
for name, fn in (("theta_step_ms", lambda t: eng.theta_step(x, z, idx, 5e-3, 1, 2 * t)), ("z_step_ms", lambda t: eng.z_step(x, z, idx, 5e-3, 1, 2 * t + 1))):
    for t in range(20): fn(t)
    sync(); t0 = time.perf_counter()
    for t in range(200): fn(t)
    sync(); out[name] = (time.perf_counter() - t0) * 1e3 / 200
xm = x.clone(); xm[torch.rand_like(xm) < 0.2] = float("nan")
state = torch.empty((N, q), device=dev); logp = torch.empty(N, device=dev); grad = torch.empty((N, q), device=dev)
step = torch.full((1,), 0.01, device=dev)
eng.hmc_run(xm, state, logp, grad, step, 0, 2, 1000, 10, 42, init=True)
sync(); t0 = time.perf_counter()
eng.hmc_run(xm, state, logp, grad, step, 2, 50, 1000, 10, 42)
sync(); dt = time.perf_counter() - t0
out["hmc_ms_per_transition_L10"] = dt * 1e3 / 50
out["hmc_transitions_per_s"] = N * 50 / dt
macs = q * 64 + 4 * 64 * 64 + 2 * 64 * p
out["hmc_tflops_algorithmic"] = N * 50 * 10 * macs * 2 * 2 * 3 / dt / 1e12      # 2 Flipout GEMMs, forward + 2 backward products... (input-gradient only: x2)
draws = torch.randn(200, min(N, 100), q, device=dev)
eng.decode(draws, 1, 7, want_full=True); sync(); t0 = time.perf_counter()
for _ in range(5): eng.decode(draws, 1, 7, want_full=True)
sync(); out["decode_ms_200x100"] = (time.perf_counter() - t0) * 1e3 / 5
out.update(N=N, p=p, q=q)
print(json.dumps(out))
