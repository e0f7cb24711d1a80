"""The retained-iteration cache of the outcome net (causal_effects_cached): ADRF sums with and without the skip must be bit-identical."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, ".")
from bayesgm_amd.engine import CausalEngine
from bayesgm_amd import _lib
from oracle import causal as OC
N = int(float(sys.argv[1])) if len(sys.argv) > 1 else 200000
q_sd = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
z_dims, p = [1, 1, 1, 7], 200
m = OC.init_model(0, z_dims, p)
eng = CausalEngine(p, z_dims)
eng.set_model(g=m["g"], f=m["f"], h=m["h"], e=m["e"])
g = torch.Generator(device="cuda").manual_seed(0)
v = torch.randn(N, p, device="cuda", generator=g); x = torch.rand(N, device="cuda", generator=g); y = torch.randn(N, device="cuda", generator=g)
xs = np.linspace(0, 3, 20)
res = {}
for mode in ("skip", "full", "skip", "full"):
    eng.set_outcome_cache(mode == "skip")
    eng.timing_enable(True); eng.timing_read(-1, True)
    torch.cuda.synchronize(); t0 = time.time()
    out = eng.mh_sample(x, y, v, 60, 40, q_sd, 1, effect=_lib.EFFECT_ADRF, x_values=xs)
    torch.cuda.synchronize(); dt = time.time() - t0
    nb, msb = eng.timing_read(0, False); nk, msk = eng.timing_read(1, True)
    acc = out["acc_count"].sum().item() / (100 * N)
    res[mode] = out["adrf"].cpu().numpy()
    sk, tot = eng.outcome_cache_stats()
    print(f"{mode}: burn-in {msb:.2f} ms, keep {msk:.2f} ms ({msk/40:.3f} ms / kept iteration), acceptance {acc:.4f}, served from cache {sk} / {tot}")
d = np.abs(res["skip"] - res["full"]).max()
print("ADRF max |skip - full| =", d, "bit-identical" if np.array_equal(res["skip"], res["full"]) else "DIFFERENT")
