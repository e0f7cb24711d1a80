import sys, os, ctypes as C
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import numpy as np, torch
from bayesgm_amd import _lib
from tests.test_gpu_causal import _model, _engine
from oracle import identifiable as OI
rs = np.random.RandomState(11)
q, k, units, B, n = 10, 10, (64,), 32, 90
eng = _engine(_model(5, [1, 1, 1, 7], 20, False)); dev = eng.device
pn32 = OI.init_prior_net(rs, k, q, units)
dims = [k] + list(units) + [q + 1]
cfg = _lib.PriorConfig(len(dims) - 1, (C.c_int32 * 5)(*(dims + [0] * (5 - len(dims)))))
cnt = C.c_int64(); print("n_params rc", eng.lib.bgm_prior_n_params(C.byref(cfg), C.byref(cnt)), cnt.value, flush=True)
flat = np.concatenate([np.concatenate([W.ravel(), b.ravel()]) for W, b in pn32]); print(flat.size, flush=True)
theta = torch.from_numpy(flat).to(dev); m_, v_ = torch.zeros_like(theta), torch.zeros_like(theta)
tab = torch.empty((k, q + 2), device=dev)
print("table rc", eng.lib.bgm_prior_table(eng.h, C.byref(cfg), theta.data_ptr(), tab.data_ptr(), None), flush=True)
torch.cuda.synchronize(); print(tab[0], flush=True)
zd = torch.randn(n, q, device=dev); segd = torch.randint(0, k, (n,), device=dev, dtype=torch.int32)
idx = torch.randperm(n, device=dev)[:B].to(torch.int32); dz = torch.randn(B, q, device=dev); out = torch.zeros(2, device=dev)
print("step rc", eng.lib.bgm_prior_step(eng.h, C.byref(cfg), theta.data_ptr(), m_.data_ptr(), v_.data_ptr(), segd.data_ptr(), zd.data_ptr(), idx.data_ptr(), B, dz.data_ptr(), 1e-3, 1e-3, 3, 5, out.data_ptr(), None), flush=True)
torch.cuda.synchronize(); print(out, flush=True)
