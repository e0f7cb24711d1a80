"""Fit-step throughput probe (dev tool): python scripts/probe_fit.py N B [dense|lazy|replay] [steps]"""
import sys, time
import numpy as np, torch
import os; sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from bayesgm_amd.engine import CausalEngine
from oracle import causal as OC
N = int(float(sys.argv[1])); B = int(float(sys.argv[2])); lazy = {"dense": 0, "lazy": 1, "replay": 2}[sys.argv[3]] if len(sys.argv) > 3 else 0
steps = int(sys.argv[4]) if len(sys.argv) > 4 else 200
z_dims, p = [1, 1, 1, 7], 200
U = {}
if os.environ.get("FIT_UNITS"):      # e.g. FIT_UNITS=128,128: the same hidden widths for g, e, f, h (general-width engine)
    u = tuple(int(t) for t in os.environ["FIT_UNITS"].split(","))
    U = dict(g_units=u, e_units=u, f_units=u, h_units=u)
m = OC.init_model(0, z_dims, p, **U)
eng = CausalEngine(p, z_dims, **{k: list(w) for k, w in U.items()}); eng.set_model(g=m["g"], f=m["f"], h=m["h"], e=m["e"])
g = torch.Generator(device="cuda").manual_seed(0)
v = torch.randn(N, p, device="cuda", generator=g); x = torch.rand(N, device="cuda", generator=g); y = torch.randn(N, device="cuda", generator=g)
z = torch.randn(N, 10, device="cuda", generator=g); zm = torch.zeros_like(z); zv = torch.zeros_like(z)
npar = eng.fit_begin(N, B); grad = torch.empty(npar, device="cuda")
perm = torch.randperm(N, device="cuda", generator=g).to(torch.int32)
def run(k):
    for s in range(k):
        i = (s * B) % max(1, N - B)
        idx = perm[i:i + B]
        if lazy == 2:
            eng.fit_z_sync(z, zm, zv, idx, 1e-4)
        eng.fit_theta_grad(x, y, v, z, idx, B, grad); eng.fit_theta_apply(grad, 1e-4)
        eng.fit_z_step(x, y, v, z, zm, zv, idx, B, 1e-4, lazy=lazy)
run(5); torch.cuda.synchronize(); t0 = time.time(); run(steps); torch.cuda.synchronize(); dt = time.time() - t0
if lazy != 0 or True:      # the same minibatches from ONE library call (bgm_causal_fit_epoch)
    k = min(steps * B, (N // B) * B)
    eng.fit_epoch(x, y, v, z, zm, zv, perm[:5 * B], B, 1e-4, 1e-4, lazy); torch.cuda.synchronize()
    t0 = time.time(); eng.fit_epoch(x, y, v, z, zm, zv, perm[:k], B, 1e-4, 1e-4, lazy); dh = time.time() - t0; torch.cuda.synchronize(); de = time.time() - t0
    print(f"  library epoch loop: {de / (k // B) * 1e6:.1f} us/step over {k // B} minibatches (host issue {dh / (k // B) * 1e6:.1f} us/step)")
flop = 348480.0 * B * steps
print(f"N={N} B={B} lazy={lazy}: {dt/steps*1e6:.1f} us/step, {B*steps/dt:.3e} obs/s, {flop/dt/1e12:.2f} TFLOP/s algorithmic")
eng.fit_end()
