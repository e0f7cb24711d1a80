"""dev aid: time of one log-posterior evaluation (bgm_bnn_logpost: perturbations + sign words + the MODE 0 sampler kernel) at the bench shape,
fp32 and split precision.  python scripts/dev/probe_bnx_logpost.py [N]"""
import sys, time
import numpy as np
import torch
sys.path.insert(0, ".")
from oracle import bnn as OB
from bayesgm_amd.bnn_engine import BnnEngine
N = int(float(sys.argv[1])) if len(sys.argv) > 1 else 1000000
p, z_dims = 200, [1, 1, 1, 7]
m = OB.init_model(0, z_dims, p, False)
eng = BnnEngine(p, z_dims, False, max_batch=64, norm_mode=1)
eng.begin(m)
dev = eng.device
g = torch.Generator(device=dev); g.manual_seed(0)
v = torch.randn(N, p, device=dev, generator=g); x = torch.rand(N, device=dev, generator=g); y = torch.randn(N, device=dev, generator=g)
z = torch.randn(N, 10, device=dev, generator=g)
for mode in ("fp32", "f16x3"):
    eng.set_precision(mode)
    eng.logpost(x, y, v, z, 10000, 1, 0)
    torch.cuda.synchronize(); t = time.perf_counter()
    for i in range(10):
        eng.logpost(x, y, v, z, 10000, 1, i)
    torch.cuda.synchronize()
    print("%s: %.3f ms per log-posterior evaluation of %d rows" % (mode, 1e2 * (time.perf_counter() - t), N))
