"""Encoder pass z = e(V) (causalbgm/base.py:479, the Z initialisation of fit): throughput of causal_encode_kernel at the bench
shape.  usage: python scripts/probe_encode.py [N] [p] [reps]   (north_star: >= 40 % of the MFMA roofline on this kernel)"""
import json, sys, time
import numpy as np, torch
sys.path.insert(0, ".")
from bayesgm_amd.engine import CausalEngine
from oracle import causal as OC
N = int(float(sys.argv[1])) if len(sys.argv) > 1 else 1000000
p = int(sys.argv[2]) if len(sys.argv) > 2 else 200
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 20
z_dims = [1, 1, 1, 7]
m = OC.init_model(0, z_dims, p)
eng = CausalEngine(p, z_dims)
eng.set_model(g=m["g"], f=m["f"], h=m["h"], e=m["e"])
g = torch.Generator(device="cuda").manual_seed(0)
v = torch.randn(N, p, device="cuda", generator=g)
z = eng.encode(v)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(reps):
    z = eng.encode(v)
e1.record()
torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / reps
macs = p * 64 + 4 * 64 * 64 + 64 * sum(z_dims)
flop = 2.0 * macs * N
from oracle.nets import mlp_forward
ref = mlp_forward(OC.cast_model(m, np.float64)["e"], v[:512].cpu().numpy().astype(np.float64))
print(json.dumps(dict(N=N, p=p, ms_per_pass=ms, rows_per_s=N / (ms * 1e-3), algorithmic_tflops=flop / (ms * 1e-3) / 1e12,
                      frac_of_fp32_mfma_peak=flop / (ms * 1e-3) / 1e12 / 157.3, hbm_read_GBps=N * p * 4 / (ms * 1e-3) / 1e9,
                      frac_of_hbm_peak=N * p * 4 / (ms * 1e-3) / 8e12, flop_per_row=2.0 * macs, bytes_per_row=4 * p,
                      max_abs_err_vs_oracle_512rows=(float(np.abs(z[:512].cpu().numpy() - ref).max()) if ref is not None else None))))
