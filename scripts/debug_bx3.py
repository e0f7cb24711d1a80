import sys, numpy as np, torch
sys.path.insert(0, ".")
from bayesgm_amd.engine import CausalEngine
from bayesgm_amd import _lib
from oracle import causal as OC
z_dims, p = [1, 1, 1, 7], 200
m = OC.init_model(0, z_dims, p)
eng = CausalEngine(p, z_dims); eng.set_model(g=m["g"], f=m["f"], h=m["h"], e=m["e"])
from bayesgm_amd.datasets import Sim_Hirano_Imbens_sampler
xs = np.linspace(0, 3, 20)
def run(N, burn, keep, label):
    x, y, v = Sim_Hirano_Imbens_sampler(N=N, v_dim=p, seed=0).load_all()
    res = {}
    for mode in ("fp32", "bf16x3"):
        eng.set_precision(mode)
        out = eng.mh_sample(x, y, v, burn, keep, 1.0, 5, effect=_lib.EFFECT_ADRF, x_values=xs)
        acc = out["acc_count"].double().sum().item() / ((burn + keep) * N)
        st = out["state"].cpu().numpy()
        res[mode] = (acc, out["adrf"].mean(dim=1).cpu().numpy()[:3], st, out["logp"].cpu().numpy())
    eng.set_precision("fp32")
    lp32 = eng.logpost(x.ravel(), y.ravel(), v, res["bf16x3"][2]).cpu().numpy()
    eng.set_precision("bf16x3")
    lpbx = eng.logpost(x.ravel(), y.ravel(), v, res["bf16x3"][2]).cpu().numpy()
    eng.set_precision("fp32")
    print(label, "N", N, "acc fp32 %.4f bx3 %.4f" % (res["fp32"][0], res["bf16x3"][0]), "adrf fp32", res["fp32"][1], "bx3", res["bf16x3"][1])
    print("   on bx3 final states: |lp_bx3_kernel_cached - lp_fp32| max %.3e, |lp_bx3_logpost - lp_fp32| max %.3e, |lp| mean %.1f" % (
        np.abs(res["bf16x3"][3] - lp32).max(), np.abs(lpbx - lp32).max(), np.abs(lp32).mean()))
    print("   state |z| max fp32 %.2f bx3 %.2f" % (np.abs(res["fp32"][2]).max(), np.abs(res["bf16x3"][2]).max()))
run(2048, 300, 200, "short/small")
run(2048, 3000, 1000, "long/small")
run(100000, 300, 200, "short/multi-tile")
