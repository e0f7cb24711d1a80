import sys; sys.path.insert(0, ".")
import numpy as np
from tests.test_gpu_causal import _model, _data, _engine, _as64, CASES
from oracle import causal as OC
for case in CASES[:3]:
    m = _model(1, case["z_dims"], case["p"], case["binary"])
    x, y, v = _data(case["n"], case["p"], 2, case["binary"])
    z = np.random.RandomState(3).randn(case["n"], sum(case["z_dims"])).astype(np.float32)
    got = _engine(m).logpost(x.ravel(), y.ravel(), v, z).cpu().numpy()
    m64, (x64, y64, v64, z64) = _as64(m, x, y, v, z)
    ref = OC.log_posterior(m64, x64, y64, v64, z64)
    err = np.abs(got - ref); i = err.argmax()
    print(case, "max err", err.max(), "at", i, got[i], ref[i], "mean err", err.mean(), "nan", np.isnan(got).sum())
