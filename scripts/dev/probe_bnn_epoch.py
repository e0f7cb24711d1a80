"""dev aid: one bgm_bnn_fit_epoch call (Bayesian nets, B = 32) for a kernel trace: python scripts/dev/probe_bnn_epoch.py [N] [minibatches]"""
import sys, time
import numpy as np, torch
sys.path.insert(0, ".")
from oracle import bnn as OB
from bayesgm_amd.bnn_engine import BnnEngine
N = int(float(sys.argv[1])) if len(sys.argv) > 1 else 100000
K = int(sys.argv[2]) if len(sys.argv) > 2 else 400
p, z_dims, B = 200, [1, 1, 1, 7], 32
m = OB.init_model(0, z_dims, p, False)
eng = BnnEngine(p, z_dims, False, max_batch=64, norm_mode=1)
eng.begin(m)
dev = eng.device
g = torch.Generator(device=dev); g.manual_seed(0)
v = torch.randn(N, p, device=dev, generator=g); x = torch.rand(N, device=dev, generator=g); y = torch.randn(N, device=dev, generator=g)
z = torch.randn(N, 10, device=dev, generator=g); zm, zv = torch.zeros_like(z), torch.zeros_like(z)
perm = torch.randperm(N, device=dev, generator=g).to(torch.int32)
eng.fit_epoch(x, y, v, z, zm, zv, perm[:8 * B], B, 1e-4, 1e-4, 2, 1, 0); torch.cuda.synchronize()
t0 = time.time(); n = eng.fit_epoch(x, y, v, z, zm, zv, perm[:K * B], B, 1e-4, 1e-4, 2, 1, 100); torch.cuda.synchronize()
print("bnn fit epoch: %.1f us per minibatch over %d minibatches" % (1e6 * (time.time() - t0) / n, n))
