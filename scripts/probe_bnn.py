"""Timing of the Bayesian-network (use_bnn=True) kernels on one MI355X: MH iterations at the north-star panel size,
minibatch steps, EGM steps.  python scripts/probe_bnn.py [N] [p] [iters] [z_dims, e.g. 3,6,3,6]"""
import sys, time
import numpy as np
import torch
sys.path.insert(0, ".")
from oracle import bnn as OB
from bayesgm_amd.bnn_engine import BnnEngine

N = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000
p = int(sys.argv[2]) if len(sys.argv) > 2 else 200
iters = int(sys.argv[3]) if len(sys.argv) > 3 else 20
z_dims = [int(t) for t in sys.argv[4].split(",")] if len(sys.argv) > 4 else [1, 1, 1, 7]
q = sum(z_dims)
m = OB.init_model(0, z_dims, p, False)
eng = BnnEngine(p, z_dims, False, max_batch=64, norm_mode=1)      # inference-mode input normalisation, the models' default
eng.begin(m)
if __import__("os").environ.get("BNN_PRECISION"):      # "f16x3": the split-precision sampling kernels (csrc/bnx_kernels.h)
    eng.set_precision(__import__("os").environ["BNN_PRECISION"])
dev = eng.device
g = torch.Generator(device=dev); g.manual_seed(0)
v = torch.randn(N, p, device=dev, generator=g)
x = torch.rand(N, device=dev, generator=g)
y = torch.randn(N, device=dev, generator=g)
state = torch.empty(N, q, device=dev)
bs = 10000


def timed(fn, reps=1):
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / reps


eng.mh_run(x, y, v, state, bs, 0, 2, 0, 1.0, 1, init=True)      # warm-up, packs the kernels
t = timed(lambda: eng.mh_run(x, y, v, state, bs, 2, iters, 0, 1.0, 1))
macs = sum(a * b for net in ("g", "f", "h") for a, b in zip(OB.net_dims(m[net])[:-1], OB.net_dims(m[net])[1:]))
flops = 2 * 2 * 2 * macs * N * iters          # two states x two GEMMs per layer
if __import__("os").environ.get("BNN_PROBE_MH_ONLY"):
    print("MH only: %.2f ms / iteration" % (1e3 * t / iters)); sys.exit(0)
print("MH (burn-in) N=%d p=%d bs=%d: %.2f ms / iteration, %.3e transitions/s, %.1f TFLOP/s (Flipout: 4 x %d MAC per row)"
      % (N, p, bs, 1e3 * t / iters, N * iters / t, flops / t / 1e12, macs))
xs = torch.linspace(0, 3, 20, device=dev)
adrf = torch.zeros(20, iters, device=dev, dtype=torch.float64)
t2 = timed(lambda: eng.mh_run(x, y, v, state, bs, 100, iters, 100, 1.0, 1, n_keep=iters, effect=1, x_values=xs, adrf_sum=adrf))
print("MH (keep, 20 doses): %.2f ms / iteration" % (1e3 * t2 / iters))
print("predict(burn_in=5000, n_mcmc=3000) estimate: %.1f s" % (5000 * t / iters + 3000 * t2 / iters))
if __import__("os").environ.get("BNN_PROBE_SAMPLING_ONLY"):
    sys.exit(0)
# minibatch steps
n = 20000
idx = torch.randperm(n, device=dev)[:32].int()
z = torch.randn(n, q, device=dev)
zm, zv = torch.zeros_like(z), torch.zeros_like(z)
eng.theta_step(z, idx, x, y, v, 1e-4, 1, 0)
tt = timed(lambda: eng.theta_step(z, idx, x, y, v, 1e-4, 1, 0), 50)
tz = timed(lambda: eng.z_step(x, y, v, z, zm, zv, idx, 1e-4, 1, 1), 50)
print("fit minibatch (B=32): theta step %.0f us, latent step %.0f us -> %.1f s per epoch of N=20000" % (1e6 * tt, 1e6 * tz, (tt + tz) * 625))
_rs = np.random.RandomState(0)
_dd = [q, 64, 32, 8, 1]
dz = {"W": [_rs.uniform(-1, 1, (_dd[i], _dd[i + 1])).astype(np.float32) * np.float32(np.sqrt(6.0 / (_dd[i] + _dd[i + 1]))) for i in range(4)],
      "b": [np.zeros(d, np.float32) for d in _dd[1:]], "gamma": [np.ones(d, np.float32) for d in _dd[1:-1]],
      "beta": [np.zeros(d, np.float32) for d in _dd[1:-1]]}
eng.set_disc_norm("fixed")      # the models' default (DESIGN_HISTORY.md section 2b)
eng.egm_begin(dz, 32, 2e-4, 1)
zp = torch.randn(32, q, device=dev)
eng.egm_disc_step(zp, idx, v, 0.3, 1, 0); eng.egm_gen_step(zp, idx, v, x, y, 1, 1)
td = timed(lambda: eng.egm_disc_step(zp, idx, v, 0.3, 1, 0), 50)
tg = timed(lambda: eng.egm_gen_step(zp, idx, v, x, y, 1, 1), 50)
print("EGM (B=32): disc step %.0f us, gen step %.0f us -> %.2f ms per iteration (5 + 1), %.0f s per 30000" % (1e6 * td, 1e6 * tg, 1e3 * (5 * td + tg), 3e4 * (5 * td + tg)))
te = timed(lambda: eng.evaluate(x, y, v, None, x_values=np.linspace(0, 3, 200), seed=1, stream_id=0))
print("evaluate (N=%d, 200 doses, incl. encoder pass): %.1f ms" % (N, 1e3 * te))
