"""Latency of the native EGM steps (one launch each):  python scripts/probe_egm_native.py [p] [iters] [fixed|batch] [z_dims, e.g. 3,3,6,6]
(fixed = the shipped discriminator normalisation; BGM_EGM_NO_CHAIN=1 forces the phase-machine kernels)"""
import sys, os, time, json
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from bayesgm_amd.engine import CausalEngine

def _glorot(rs, a, b):
    l = np.sqrt(6.0 / (a + b)); return rs.uniform(-l, l, (a, b)).astype(np.float32)
def _mlp(rs, dims):
    return [(_glorot(rs, dims[i], dims[i + 1]), np.zeros(dims[i + 1], np.float32)) for i in range(len(dims) - 1)]
def _disc(rs, dims):
    return {"W": [_glorot(rs, dims[i], dims[i + 1]) for i in range(len(dims) - 1)], "b": [np.zeros(d, np.float32) for d in dims[1:]],
            "gamma": [np.ones(d, np.float32) for d in dims[1:-1]], "beta": [np.zeros(d, np.float32) for d in dims[1:-1]]}

p = int(sys.argv[1]) if len(sys.argv) > 1 else 200
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 200
z_dims = [int(t) for t in sys.argv[4].split(",")] if len(sys.argv) > 4 else [1, 1, 1, 7]
q = sum(z_dims); B = 32; n = 20000
rs = np.random.RandomState(0)
nets = {"g": _mlp(rs, [q] + [64] * 5 + [p + 1]), "e": _mlp(rs, [p] + [64] * 5 + [q]),
        "f": _mlp(rs, [z_dims[0] + z_dims[1] + 1, 64, 32, 8, 2]), "h": _mlp(rs, [z_dims[0] + z_dims[2], 64, 32, 8, 2])}
dz = _disc(rs, [q, 64, 32, 8, 1])
eng = CausalEngine(p, z_dims)
eng.set_model(g=nets["g"], f=nets["f"], h=nets["h"], e=nets["e"])
norm = sys.argv[3] if len(sys.argv) > 3 else "fixed"
eng.set_disc_norm(norm)
eng.egm_begin(B, [64, 32, 8], 2e-4, True, dz)
v = torch.randn(n, p, device="cuda"); x = torch.rand(n, device="cuda"); y = torch.randn(n, device="cuda")
zs = torch.randn(iters * 6, B, q, device="cuda")
idx = torch.randint(0, n, (iters * 6, B), device="cuda", dtype=torch.int32)
def run(k):
    for it in range(k):
        for j in range(5):
            eng.egm_disc_step(zs[it * 6 + j], idx[it * 6 + j], v, 0.5)
        eng.egm_gen_step(zs[it * 6 + 5], idx[it * 6 + 5], v, x, y)
run(10); torch.cuda.synchronize()
t0 = time.perf_counter(); run(iters); torch.cuda.synchronize(); dt = time.perf_counter() - t0
t1 = time.perf_counter()
for it in range(iters): eng.egm_disc_step(zs[it], idx[it], v, 0.5)
torch.cuda.synchronize(); td = (time.perf_counter() - t1) / iters
t1 = time.perf_counter()
for it in range(iters): eng.egm_gen_step(zs[it], idx[it], v, x, y)
torch.cuda.synchronize(); tg = (time.perf_counter() - t1) / iters
print(json.dumps(dict(p=p, z_dims=z_dims, disc_norm=norm, ms_per_iteration=1e3 * dt / iters, disc_step_us=1e6 * td, gen_step_us=1e6 * tg,
                      est_30000_iters_s=30000 * dt / iters)))
