"""Cost of the general-width engine (csrc/gx_api.hip): the bench shape forced through it (BGM_FORCE_GX=1) next to the resident kernels,
and hidden widths only it holds.  python scripts/probe_gx.py [N]   (run once as is and once with BGM_FORCE_GX=1)"""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, ".")
from bayesgm_amd.engine import CausalEngine
from bayesgm_amd import _lib
from oracle import causal as OC

forced = bool(os.environ.get("BGM_FORCE_GX"))
N = int(float(sys.argv[1])) if len(sys.argv) > 1 else 1000000
D = dict(g_units=(64,) * 5, e_units=(64,) * 5, f_units=(64, 32, 8), h_units=(64, 32, 8))
cases = [("default widths", D, N)]
if forced:
    cases += [("g/f/h [8, 8] / [8, 4] (R test)", dict(g_units=(8, 8), e_units=(8, 8), f_units=(8, 4), h_units=(8, 4)), N),
              ("[128, 128]", dict(g_units=(128, 128), e_units=(128, 128), f_units=(128, 128), h_units=(128, 128)), N // 2),
              ("[256, 256, 256]", dict(g_units=(256,) * 3, e_units=(256,) * 3, f_units=(256,) * 3, h_units=(256,) * 3), N // 4)]
only = os.environ.get("GX_ONLY")
if only == "w256":
    cases = [c for c in cases if c[0].startswith("[256")]
elif only == "default":
    cases = cases[:1]
elif only == "w128":
    cases = [c for c in cases if c[0].startswith("[128")]
z_dims, p = [1, 1, 1, 7], 200
for name, u, n in cases:
    m = OC.init_model(0, z_dims, p, **u)
    eng = CausalEngine(p, z_dims, **{k: list(v) for k, v in u.items()})
    eng.set_model(g=m["g"], f=m["f"], h=m["h"], e=m["e"])
    eng.set_precision(os.environ.get("GX_PRECISION", "fp32"))      # "f16x3": split precision on the row-tile-per-wave kernels (hidden widths <= 128)
    g = torch.Generator(device="cuda").manual_seed(0)
    v = torch.randn(n, p, device="cuda", generator=g); x = torch.rand(n, device="cuda", generator=g); y = torch.randn(n, device="cuda", generator=g)
    xs = np.linspace(0, 3, 20)
    macs = {k: sum(w.shape[0] * w.shape[1] for w, _ in m[k]) for k in "gfhe"}
    for burn, keep in ((20, 0), (0, 10)):
        eng.mh_sample(x, y, v, 2, 2, 1.0, 1, effect=_lib.EFFECT_ADRF, x_values=xs)
        torch.cuda.synchronize(); t0 = time.time()
        eng.mh_sample(x, y, v, burn, keep, 1.0, 1, effect=_lib.EFFECT_ADRF if keep else 0, x_values=xs)
        torch.cuda.synchronize(); dt = time.time() - t0
        it = burn + keep
        fl = 2 * (macs["g"] + macs["f"] + macs["h"]) + (2 * 20 * macs["f"] if keep else 0)
        # (kept rows: the product-default outcome cache is ON, the FLOP of cached doses are not executed -- a work-equivalent rate, not a utilisation)
        print("%s %s N %d: %s %.3f ms / iteration, %.3e transitions/s, %.1f TFLOP/s %s"
              % ("gx      " if forced else "resident", name, n, "burn-in" if burn else "kept (20 doses)", 1e3 * dt / it, n * it / dt, fl * n * it / dt / 1e12,
                 "algorithmic" if burn else "WORK-EQUIVALENT (outcome cache on)"), flush=True)
    eng.encode(v[:1024]); torch.cuda.synchronize(); t0 = time.time()
    for _ in range(5):
        eng.encode(v)
    torch.cuda.synchronize(); dt = (time.time() - t0) / 5
    print("%s %s N %d: encoder %.3f ms per pass, %.1f TFLOP/s algorithmic" % ("gx      " if forced else "resident", name, n, 1e3 * dt, 2 * macs["e"] * n / dt / 1e12), flush=True)
    eng.close()
