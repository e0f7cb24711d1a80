import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import numpy as np
from oracle import causal as OC
from tests.test_gpu_causal import _model, _data, _engine
for case in [dict(z_dims=[1, 1, 1, 7], p=20, binary=False, n=100), dict(z_dims=[1, 1, 1, 7], p=50, binary=False, n=33),
             dict(z_dims=[2, 2, 2, 6], p=150, binary=True, n=64), dict(z_dims=[3, 3, 6, 6], p=17, binary=False, n=16)]:
    m = _model(3, case["z_dims"], case["p"], case["binary"])
    x, y, v = _data(case["n"], case["p"], 4, case["binary"])
    z = np.random.RandomState(5).randn(case["n"], sum(case["z_dims"])).astype(np.float32)
    eng = _engine(m)
    ref = OC.log_posterior(OC.cast_model(m, np.float64), x.astype(np.float64), y.astype(np.float64), v.astype(np.float64), z.astype(np.float64))
    res = {}
    for mode in ("fp32", "bf16x3", "f16x3"):
        eng.set_precision(mode)
        res[mode] = eng.logpost(x.ravel(), y.ravel(), v, z).cpu().numpy()
    eng.set_precision("fp32")
    e = {k: np.abs(r - ref) for k, r in res.items()}
    i = int(np.argmax(e["f16x3"]))
    print(case, "max err", {k: float(a.max()) for k, a in e.items()}, "row", i, "ref", ref[i], {k: float(a[i]) for k, a in e.items()},
          "median", {k: float(np.median(a)) for k, a in e.items()})
    print("   |v| max", np.abs(v).max(), "|x|", np.abs(x).max(), "|y|", np.abs(y).max(), "z max", np.abs(z).max())
