"""Cost of the general (streamed-fragment) sampling path of the deterministic CausalBGM next to the LDS-resident kernels, and at shapes only it
holds.  python scripts/probe_general.py   (run once as is and once with BGM_FORCE_GENERAL=1)"""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, ".")
from bayesgm_amd.engine import CausalEngine
from bayesgm_amd import _lib
from oracle import causal as OC

forced = bool(os.environ.get("BGM_FORCE_GENERAL"))
cases = [([1, 1, 1, 7], 200, 1000000)] if not forced else [([1, 1, 1, 7], 200, 1000000), ([3, 6, 3, 6], 177, 1000000), ([5, 5, 5, 5], 100, 1000000),
                                                            ([1, 1, 1, 7], 500, 500000)]
for z_dims, p, N in cases:
    m = OC.init_model(0, z_dims, p)
    eng = CausalEngine(p, z_dims)
    eng.set_model(g=m["g"], f=m["f"], h=m["h"], e=m["e"])
    g = torch.Generator(device="cuda").manual_seed(0)
    v = torch.randn(N, p, device="cuda", generator=g); x = torch.rand(N, device="cuda", generator=g); y = torch.randn(N, device="cuda", generator=g)
    xs = np.linspace(0, 3, 20)
    macs = sum(a * b for net in ("g", "f", "h") for a, b in zip([w.shape[0] for w, _ in m[net]], [w.shape[1] for w, _ in m[net]]))
    for burn, keep in ((40, 0), (0, 20)):
        eng.mh_sample(x, y, v, 2, 2, 1.0, 1, effect=_lib.EFFECT_ADRF, x_values=xs)
        torch.cuda.synchronize(); t0 = time.time()
        eng.mh_sample(x, y, v, burn, keep, 1.0, 1, effect=_lib.EFFECT_ADRF if keep else 0, x_values=xs)
        torch.cuda.synchronize(); dt = time.time() - t0
        it = burn + keep
        print("%s z_dims %s p %d N %d: %s %.3f ms / iteration, %.3e transitions/s, %.1f TFLOP/s algorithmic (transition part)"
              % ("general " if forced else "resident", z_dims, p, N, "burn-in" if burn else "kept (20 doses)", 1e3 * dt / it, N * it / dt,
                 2 * macs * N * it / dt / 1e12))
    eng.close()
