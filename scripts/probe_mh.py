"""Quick MH-kernel throughput probe (dev tool): python scripts/probe_mh.py N iters [keep [fp32|bf16x3|f16x3]]"""
import sys, time
import numpy as np, torch
sys.path.insert(0, ".")
from bayesgm_amd.engine import CausalEngine
from bayesgm_amd import _lib
from oracle import causal as OC

N = int(float(sys.argv[1])) if len(sys.argv) > 1 else 100000
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 200
keep = int(sys.argv[3]) if len(sys.argv) > 3 else 0
prec = sys.argv[4] if len(sys.argv) > 4 else "fp32"
z_dims, p = [1, 1, 1, 7], 200
m = OC.init_model(0, z_dims, p)
eng = CausalEngine(p, z_dims)
eng.set_model(g=m["g"], f=m["f"], h=m["h"], e=m["e"])
eng.set_precision(prec)
print("precision", prec)
g = torch.Generator(device="cuda").manual_seed(0)
v = torch.randn(N, p, device="cuda", generator=g)
x = torch.rand(N, device="cuda", generator=g)
y = torch.randn(N, device="cuda", generator=g)
info = eng.mh_info(N)
print("grid", info.grid_blocks, "waves/block", info.waves_per_block, "rows/wave", info.rows_per_wave,
      "mfma/transition/wave", info.mfma_per_transition_per_wave, "flop/row", info.flop_per_row_transition)
xs = np.linspace(0, 3, 20)
for rep in range(3):
    eng.timing_enable(True)
    torch.cuda.synchronize(); t0 = time.time()
    out = eng.mh_sample(x, y, v, iters - keep, keep, 1.0, 1, effect=_lib.EFFECT_ADRF if keep else 0, x_values=xs)
    torch.cuda.synchronize(); t1 = time.time()
    nl, ms = eng.timing_read(-1, True)
    acc = out["acc_count"].sum().item() / (iters * N)
    tr = N * iters / (ms * 1e-3)
    print(f"rep{rep}: wall {t1-t0:.3f}s kernel {ms:.1f} ms ({nl} launches)  {tr/1e9:.3f} G row-transitions/s  "
          f"{tr*info.flop_per_row_transition/1e12:.1f} TFLOP/s algorithmic  acc={acc:.3f}")
