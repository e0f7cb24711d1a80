"""Timing of IdentifiableCausalBGM(use_bnn=True) sampling under batch statistics (params['bnn_norm'] = 'batch', the reference as
written) at the default widths: MH iterations of the Bayesian sampler with the conditional latent prior on, blocks of bs rows.
usage: python scripts/probe_ident_batch.py [N=1e5] [iters=10] [bs=10000]"""
import ctypes as C
import sys, time
import numpy as np
import torch
sys.path.insert(0, ".")
from oracle import bnn as OB
from bayesgm_amd import _lib
from bayesgm_amd.bnn_engine import BnnEngine, flatten_bnn

N = int(float(sys.argv[1])) if len(sys.argv) > 1 else 100000
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 10
bs = int(sys.argv[3]) if len(sys.argv) > 3 else 10000
p, z_dims, k = 200, [1, 1, 1, 7], 6
q = sum(z_dims)
m = OB.init_model(0, z_dims, p, False)
rs = np.random.RandomState(0)

for prior_on in (False, True):
    eng = BnnEngine(p, z_dims, False, max_batch=64, norm_mode=0)
    eng.begin(m)
    dev = eng.device
    g = torch.Generator(device=dev); g.manual_seed(0)
    v = torch.randn(N, p, device=dev, generator=g); x = torch.rand(N, device=dev, generator=g); y = torch.randn(N, device=dev, generator=g)
    if prior_on:
        from tests.test_gpu_identifiable_bnn import _prior, _cfg
        pn32 = _prior(rs, k, q, (64,), "batch")
        cfg = _cfg([k, 64, q + 1])
        theta = torch.from_numpy(flatten_bnn(pn32)).to(dev)
        seg = torch.from_numpy(rs.randint(0, k, N).astype(np.int32)).to(dev)
        _lib.check(eng.lib.bgm_bnn_set_prior(eng.h, C.byref(cfg), theta.data_ptr(), seg.data_ptr()), "bgm_bnn_set_prior")
    state = torch.empty(N, q, device=dev)
    eng.mh_run(x, y, v, state, bs, 0, 2, 0, 1.0, 1, init=True)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    eng.mh_run(x, y, v, state, bs, 2, iters, 0, 1.0, 1)
    torch.cuda.synchronize(); t = (time.perf_counter() - t0) / iters
    macs = sum(a * b for net in ("g", "f", "h") for a, b in zip(OB.net_dims(m[net])[:-1], OB.net_dims(m[net])[1:]))
    fl = 2 * 2 * 2 * macs * N
    print("batch statistics, default widths, conditional prior %s, N=%d bs=%d: MH iteration %.3f ms = %.2e row-transitions/s = %.1f TFLOP/s = %.3f of the fp32-MFMA peak"
          % ("on" if prior_on else "off", N, bs, 1e3 * t, N / t, fl / t / 1e12, fl / t / 1e12 / 157.3), flush=True)
    eng.close()
