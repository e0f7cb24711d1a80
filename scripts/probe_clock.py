import sys; sys.path.insert(0, ".")
import numpy as np, torch
from bayesgm_amd.engine import CausalEngine
from oracle import causal as OC
N, iters = int(float(sys.argv[1])) if len(sys.argv) > 1 else int(1e6), 100
m = OC.init_model(0, [1,1,1,7], 200)
eng = CausalEngine(200, [1,1,1,7]); eng.set_model(g=m["g"], f=m["f"], h=m["h"], e=m["e"])
if len(sys.argv) > 2: eng.set_precision(sys.argv[2])      # "bf16x3" | "f16x3": the split-precision kernel's in-kernel clock
g = torch.Generator(device="cuda").manual_seed(0)
v = torch.randn(N, 200, device="cuda", generator=g); x = torch.rand(N, device="cuda", generator=g); y = torch.randn(N, device="cuda", generator=g)
state = torch.empty(N, 10, device="cuda"); logp = torch.empty(N, device="cuda")
ns = eng.mh_slots(N)
clk = torch.zeros(ns * 4, dtype=torch.int64, device="cuda")
for rep in range(2):
    eng.mh_run(x, y, v, state, logp, 0, iters, iters, 1.0, 1, init=True, clock=clk)
    torch.cuda.synchronize()
c = clk.cpu().numpy().reshape(ns, 4)
dur = c[:, 1] / 1e5; start = (c[:, 2] - c[:, 2].min()) / 1e5; xcc = c[:, 3]
print("slots", ns, "dur ms: min %.1f med %.1f max %.1f" % (dur.min(), np.median(dur), dur.max()), " start ms max %.2f" % start.max(),
      " end ms max %.1f" % (start + dur).max())
print("clock MHz: min %.0f med %.0f max %.0f" % tuple(np.percentile(c[:, 0] / (c[:, 1] / 100.0), [0, 50, 100])))
blk = dur.reshape(-1, 8)
print("per-block mean dur: min %.1f med %.1f max %.1f" % (blk.mean(1).min(), np.median(blk.mean(1)), blk.mean(1).max()))
print("per-wave-in-block mean dur:", np.round(blk.mean(0), 1))
for k in range(8):
    sel = xcc == k
    if sel.any(): print("xcc", k, "waves", sel.sum(), "dur med %.1f max %.1f" % (np.median(dur[sel]), dur[sel].max()))
h, e = np.histogram(dur, bins=10); print("hist", h, np.round(e, 1))
