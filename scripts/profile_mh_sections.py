"""Per-section cycle profile of one wave of the MH kernel (needs a library built with -D BGM_PROF, e.g.
   python bayesgm_amd/csrc/build.py -D BGM_PROF -o bayesgm_amd/libbgm_prof.so;  BGM_HIP_LIB=.../libbgm_prof.so python scripts/profile_mh_sections.py)"""
import sys, os, json
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np, torch
from bayesgm_amd.engine import CausalEngine
from bayesgm_amd.datasets import Sim_Hirano_Imbens_sampler
n, iters, p = 1000000, 100, 200
x, y, v = Sim_Hirano_Imbens_sampler(N=n, v_dim=p, seed=0).load_all()
rs = np.random.RandomState(0)
def glorot(a, b):
    l = np.sqrt(6.0 / (a + b)); return rs.uniform(-l, l, (a, b)).astype(np.float32)
def mlp(d): return [(glorot(d[i], d[i + 1]), np.zeros(d[i + 1], np.float32)) for i in range(len(d) - 1)]
eng = CausalEngine(p, [1, 1, 1, 7])
eng.set_model(g=mlp([10] + [64] * 5 + [p + 1]), f=mlp([3, 64, 32, 8, 2]), h=mlp([2, 64, 32, 8, 2]), e=mlp([p] + [64] * 5 + [10]))
xd, yd, vd = (torch.from_numpy(a).cuda() for a in (x.reshape(-1), y.reshape(-1), v))
state = torch.zeros(n, 10, device="cuda"); logp = torch.zeros(n, device="cuda")
n_slots = eng.mh_slots(n)
clk = torch.zeros(n_slots * 4 + 64, dtype=torch.int64, device="cuda")
eng.mh_run(xd, yd, vd, state, logp, 0, 5, 10**6, 1.0, 1, init=True)
torch.cuda.synchronize()
eng.mh_run(xd, yd, vd, state, logp, 5, iters, 10**6, 1.0, 1, clock=clk)
torch.cuda.synchronize()
c = clk.cpu().numpy()
t = c[4 * n_slots:4 * n_slots + 8].astype(np.float64)
names = ["proposal RNG", "g layer 1", "g hidden x4", "g last+NLL", "f,h nets", "logp assemble", "accept", "loop/prio"]
tiles = c[0 + 0]  # total cycles of slot 0
per_it = t / (iters * max(1, round(n / 16 / n_slots)))
print(json.dumps({k: round(v_, 0) for k, v_ in zip(names, per_it)}), "sum", round(per_it.sum()), "tiles/wave", n / 16 / n_slots)
