"""Arbitrary hidden widths / depths of the deterministic CausalBGM (the general-width engine, csrc/gx_api.hip) against the oracle
through the C ABI.  The reference forwards ANY `nb_units` list to BaseFullyConnectedNet (models/networks/base.py:7-28, default
[256, 256, 256]; models/causalbgm/base.py:64-81) and its own integration tests construct g_units (8, 8), f / h / dz_units (8, 4)
(r-package/bayesgm/tests/testthat/test-causalbgm.R:28-34).  Shapes here: those, [128, 128], [256, 256, 256], mixed depths.
Same tolerances as test_gpu_causal.py / test_gpu_general.py.  BGM_FORCE_GX=1 sends a default-width model through the same engine."""
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from oracle import causal as OC  # noqa: E402
from oracle import fit as OF      # noqa: E402

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

SHAPES = {
    "r_test": dict(g_units=(8, 8), e_units=(8, 8), f_units=(8, 4), h_units=(8, 4)),                  # test-causalbgm.R:28-34
    "w128": dict(g_units=(128, 128), e_units=(128, 128), f_units=(128, 128), h_units=(128, 128)),
    "w256": dict(g_units=(256, 256, 256), e_units=(256, 256, 256), f_units=(256, 256, 256), h_units=(256, 256, 256)),   # networks/base.py:7
    "mixed": dict(g_units=(48, 100, 17), e_units=(33,), f_units=(20, 7, 5, 3), h_units=(10,)),
}


def _engine(m, units):
    from bayesgm_amd.engine import CausalEngine
    eng = CausalEngine(m["v_dim"], m["z_dims"], binary_treatment=m["binary_treatment"],
                       sigma_v=m.get("sigma_v"), sigma_x=m.get("sigma_x"), sigma_y=m.get("sigma_y"),
                       **{k: list(v) for k, v in units.items()})
    eng.set_model(g=m["g"], f=m["f"], h=m["h"], e=m["e"])
    return eng


def _data(n, p, seed, binary=False):
    rs = np.random.RandomState(seed)
    v = rs.randn(n, p).astype(np.float32)
    x = rs.exponential(size=(n, 1)).astype(np.float32)
    if binary:
        x = (x > np.median(x)).astype(np.float32)
    y = (x + rs.randn(n, 1)).astype(np.float32)
    return x, y, v


def _model(seed, z_dims, p, binary=False, **kw):
    m = OC.init_model(seed, z_dims, p, binary_treatment=binary, **kw)
    rs = np.random.RandomState(seed + 99)
    for k in ("g", "f", "h", "e"):
        m[k] = [(W.astype(np.float32), (0.1 * rs.randn(*b.shape)).astype(np.float32)) for W, b in m[k]]
    return m


def _as64(m, *arrs):
    return OC.cast_model(m, np.float64), [a.astype(np.float64) for a in arrs]


CASES = [
    dict(shape="r_test", z_dims=[1, 1, 1, 1], p=4, binary=True, n=64),          # the R integration test, binary treatment
    dict(shape="r_test", z_dims=[1, 1, 1, 1], p=4, binary=False, n=64),         # ... continuous treatment
    dict(shape="w128", z_dims=[1, 1, 1, 7], p=200, binary=False, n=150),
    dict(shape="w256", z_dims=[3, 3, 6, 6], p=100, binary=True, n=101),
    dict(shape="mixed", z_dims=[2, 3, 4, 5], p=77, binary=False, n=45),
    dict(shape="w128", z_dims=[10, 10, 10, 10], p=500, binary=False, n=70),     # sum(z_dims) = 40, wide data
]


@pytest.mark.parametrize("shape,z_dims,p,binary", [("mixed", [2, 3, 4, 5], 77, False), ("w128", [1, 1, 1, 7], 200, True)])
def test_conditional_prior_on_the_general_width_engine(shape, z_dims, p, binary):
    """IdentifiableCausalBGM's per-row latent prior N(mu(u), sigma^2(u) I) (identifiable.py:541-551; bgm_causal_set_prior) in the
    general-width engine's log posterior and chains, against oracle.causal(prior=...) -- the checks of test_gpu_identifiable.py."""
    import torch
    from oracle import identifiable as OI
    rs = np.random.RandomState(3)
    units = SHAPES[shape]
    m = _model(5, z_dims, p, binary, **{k: list(v) for k, v in units.items()})
    q, k, n = sum(z_dims), 7, 120
    x, y, v = _data(n, p, 6, binary)
    z = rs.randn(n, q).astype(np.float32)
    seg = rs.randint(0, k, n)
    pn = [(W, (0.3 * rs.randn(*b.shape)).astype(np.float32)) for W, b in OI.init_prior_net(rs, k, q)]
    tab = OI.prior_table(pn, q)
    mu, s2, _ = OI.prior_params([(W.astype(np.float64), b.astype(np.float64)) for W, b in pn], seg)
    eng = _engine(m, units)
    eng.set_prior(torch.from_numpy(seg.astype(np.int32)).cuda(), torch.from_numpy(tab).cuda())
    m64, (x64, y64, v64, z64) = _as64(m, x, y, v, z)
    lp = eng.logpost(x.ravel(), y.ravel(), v, z).cpu().numpy()
    ref = OC.log_posterior(m64, x64, y64, v64, z64, prior=(mu, s2))
    assert np.all(np.abs(lp - ref) <= 2e-6 * np.abs(ref) + 5e-4), np.abs(lp - ref).max()
    eng.set_precision("f16x3")                                # the split-precision kernels of the engine carry the prior too (fp32 assembly)
    lpx = eng.logpost(x.ravel(), y.ravel(), v, z).cpu().numpy()
    eng.set_precision("fp32")
    assert np.all(np.abs(lpx - ref) <= 1e-5 * np.abs(ref) + 1e-3), np.abs(lpx - ref).max()
    std = OC.log_posterior(m64, x64, y64, v64, z64)
    assert np.abs(ref - std).max() > 0.1                      # the prior matters in this test
    out = eng.mh_sample(x, y, v, 30, 10, 0.4, 77, want_draws=True)
    ref_draws = OC.mh_sampler(m, (x, y, v), 30, 10, 0.4, 77, prior=(mu.astype(np.float32), s2.astype(np.float32)))
    same = np.all(np.abs(out["draws"].cpu().numpy()[-1] - ref_draws[-1]) <= 1e-4, axis=1).mean()
    assert same >= 0.97, same
    eng.set_prior(None, None)
    lp0 = eng.logpost(x.ravel(), y.ravel(), v, z).cpu().numpy()
    assert np.all(np.abs(lp0 - std) <= 2e-6 * np.abs(std) + 2e-4)          # cleared: back to N(0, I)


@pytest.mark.parametrize("case", CASES)
def test_logpost_matches_oracle(case):
    u = SHAPES[case["shape"]]
    m = _model(1, case["z_dims"], case["p"], case["binary"], **u)
    x, y, v = _data(case["n"], case["p"], 2, case["binary"])
    z = np.random.RandomState(3).randn(case["n"], sum(case["z_dims"])).astype(np.float32)
    eng = _engine(m, u)
    got = eng.logpost(x.ravel(), y.ravel(), v, z).cpu().numpy()
    m64, (x64, y64, v64, z64) = _as64(m, x, y, v, z)
    ref = OC.log_posterior(m64, x64, y64, v64, z64)
    err = np.abs(got - ref)
    assert np.all(err <= 1e-5 * np.abs(ref) + 1e-3), (err.max(), np.abs(ref).max())


X3_CASES = [dict(shape="r_test", z_dims=[1, 1, 1, 1], p=4, binary=True, n=64), dict(shape="w128", z_dims=[1, 1, 1, 7], p=200, binary=False, n=300),
            dict(shape="mixed", z_dims=[2, 3, 2, 3], p=37, binary=False, n=129), dict(shape="mixed", z_dims=[3, 3, 6, 6], p=100, binary=True, n=77)]


@pytest.mark.parametrize("case", X3_CASES)
def test_split_precision_on_the_general_width_engine(case):
    """bgm_causal_set_precision('f16x3') outside the default shapes (hidden widths up to 128: the row-tile-per-wave kernels,
    gx_dense_x3): every contraction as three fp16 products of hi / lo splits, fp32 accumulation -- the log posterior within the fp32
    kernel's own bound of the float64 oracle, chains draw-for-draw equal to the fp32 engine's except where an accept decision lies within
    the arithmetic's error, fused effects within 2e-3 of the fp32 run's (test_gpu_bx3.py states the same for the default shapes)."""
    from bayesgm_amd import _lib
    u = SHAPES[case["shape"]]
    m = _model(1, case["z_dims"], case["p"], case["binary"], **u)
    x, y, v = _data(case["n"], case["p"], 2, case["binary"])
    z = np.random.RandomState(3).randn(case["n"], sum(case["z_dims"])).astype(np.float32)
    eng = _engine(m, u)
    m64, (x64, y64, v64, z64) = _as64(m, x, y, v, z)
    ref = OC.log_posterior(m64, x64, y64, v64, z64)
    lp32 = eng.logpost(x.ravel(), y.ravel(), v, z).cpu().numpy()
    eng.set_precision("f16x3")
    lpx = eng.logpost(x.ravel(), y.ravel(), v, z).cpu().numpy()
    err32, errx = np.abs(lp32 - ref), np.abs(lpx - ref)
    print("logpost err: fp32 max %.2e, f16x3 max %.2e (|lp| ~ %.0f)" % (err32.max(), errx.max(), np.abs(ref).mean()))
    assert np.all(errx <= 1e-5 * np.abs(ref) + 1e-3), errx.max()
    assert np.abs(lpx - lp32).max() > 0.0                                   # (another arithmetic did run)
    xs = np.linspace(0, 3, 11)
    kw = dict(effect=_lib.EFFECT_ITE) if case["binary"] else dict(effect=_lib.EFFECT_ADRF, x_values=xs)
    outs = {}
    for mode in ("fp32", "f16x3"):
        eng.set_precision(mode)
        out = eng.mh_sample(x, y, v, 30, 20, 0.3, 987654321, want_draws=True, sample_y=True, **kw)
        outs[mode] = (out["draws"].cpu().numpy(), (out["ite"] if case["binary"] else out["adrf"]).cpu().numpy(), out["acc_count"].cpu().numpy().sum())
    eng.set_precision("fp32")
    (d0, e0, a0), (d1, e1, a1) = outs["fp32"], outs["f16x3"]
    same = np.all(np.abs(d0[-1] - d1[-1]) <= 1e-4, axis=1)
    assert same.mean() >= 0.97, same.mean()
    assert abs(int(a0) - int(a1)) <= max(2, 3 * int((~same).sum()))
    if case["binary"]:
        assert np.abs(e0 - e1)[same].max() <= 2e-3
    else:
        assert np.abs(e0 - e1).max() <= 2e-3 + 3.0 * float((~same).sum()) / case["n"]


def test_split_precision_says_where_it_does_not_exist():
    u = SHAPES["w256"]
    m = _model(1, [1, 1, 1, 7], 50, False, **u)
    x, y, v = _data(40, 50, 2)
    z = np.random.RandomState(3).randn(40, 10).astype(np.float32)
    eng = _engine(m, u)
    eng.set_precision("f16x3")
    with pytest.raises(Exception, match="hidden widths up to 128"):
        eng.logpost(x.ravel(), y.ravel(), v, z)
    eng.set_precision("fp32")
    eng.logpost(x.ravel(), y.ravel(), v, z)


def test_logpost_fixed_sigmas():
    u = SHAPES["mixed"]
    m = _model(5, [2, 3, 4, 5], 77, False, sigma_v=0.8, sigma_x=1.3, sigma_y=0.5, **u)
    x, y, v = _data(100, 77, 6)
    z = np.random.RandomState(7).randn(100, 14).astype(np.float32)
    got = _engine(m, u).logpost(x.ravel(), y.ravel(), v, z).cpu().numpy()
    m64, (x64, y64, v64, z64) = _as64(m, x, y, v, z)
    ref = OC.log_posterior(m64, x64, y64, v64, z64)
    assert np.all(np.abs(got - ref) <= 1e-5 * np.abs(ref) + 1e-3)


@pytest.mark.parametrize("case", [CASES[0], CASES[1], CASES[2], CASES[3], CASES[4]])
def test_encoder_matches_oracle(case):
    from oracle.nets import mlp_forward
    u = SHAPES[case["shape"]]
    m = _model(41, case["z_dims"], case["p"], case["binary"], **u)
    _, _, v = _data(case["n"], case["p"], 42, case["binary"])
    got = _engine(m, u).encode(v).cpu().numpy()
    ref = mlp_forward(OC.cast_model(m, np.float64)["e"], v.astype(np.float64))
    assert np.abs(got - ref).max() <= 1e-5 * np.abs(ref).max() + 1e-5


def test_encoder_wide_data_default_widths():
    """e_units = [64] x 5 at v_dim = 500: no LDS-resident encoder shape holds it, the general-width engine streams the V rows."""
    from oracle.nets import mlp_forward
    m = _model(43, [1, 1, 1, 7], 500)
    _, _, v = _data(77, 500, 44)
    from bayesgm_amd.engine import CausalEngine
    eng = CausalEngine(500, [1, 1, 1, 7]); eng.set_model(g=m["g"], f=m["f"], h=m["h"], e=m["e"])
    got = eng.encode(v).cpu().numpy()
    ref = mlp_forward(OC.cast_model(m, np.float64)["e"], v.astype(np.float64))
    assert np.abs(got - ref).max() <= 1e-5 * np.abs(ref).max() + 1e-5


@pytest.mark.parametrize("case", [CASES[0], CASES[1], CASES[2], CASES[3], CASES[4]])
def test_mh_chain_and_effects_match_oracle(case):
    from bayesgm_amd import _lib
    burn, keep, q_sd, seed = 20, 15, 0.3, 1234567890123
    u = SHAPES[case["shape"]]
    m = _model(21, case["z_dims"], case["p"], case["binary"], **u)
    x, y, v = _data(case["n"], case["p"], 22, case["binary"])
    eng = _engine(m, u)
    xs = np.linspace(0, 3, 21)
    kw = dict(effect=_lib.EFFECT_ITE) if case["binary"] else dict(effect=_lib.EFFECT_ADRF, x_values=xs)
    out = eng.mh_sample(x, y, v, burn, keep, q_sd, seed, want_draws=True, chunk=11, sample_y=True, **kw)
    draws = out["draws"].cpu().numpy()
    acc = out["acc_count"].cpu().numpy()
    ref, ref_acc, _ = OC.mh_sampler(m, (x, y, v), burn, keep, q_sd, seed, return_acc=True)
    assert draws.shape == ref.shape
    row_ok = np.all(np.abs(draws[-1] - ref[-1]) <= 1e-4, axis=1)
    assert row_ok.mean() >= 0.97, row_ok.mean()
    assert np.abs(acc.astype(np.int64) - ref_acc).max() <= max(2, int((~row_ok).sum()))
    ref_eff = OC.infer_from_latent_posterior(OC.cast_model(m, np.float64), draws.astype(np.float64), None if case["binary"] else xs, True,
                                             seed, burn_in=burn)
    if case["binary"]:
        assert np.abs(out["ite"].cpu().numpy().T - ref_eff).max() <= 5e-4
    else:
        assert np.abs(out["adrf"].cpu().numpy() - ref_eff).max() <= 2e-4
    alone = eng.effects(x, out["draws"], burn, seed, x_values=None if case["binary"] else xs, sample_y=True).cpu().numpy()
    assert np.abs(alone - ref_eff).max() <= 5e-4


@pytest.mark.parametrize("shape,binary", [("r_test", False), ("w128", False), ("mixed", True), ("w256", False), ("w256", True)])
def test_outcome_cache_on_the_general_width_engine_is_bit_identical(shape, binary):
    """bgm_causal_set_outcome_cache on the general-width engine: only the (chain, dose) pairs of the chains that moved since their cached
    (mean, sd) were formed go through the outcome net, packed densely into row passes (gw_kernels.h: per wave of 16 chains, hidden widths
    up to 128; gx_causal_kernels.h: per workgroup of 32 chains, w256): effects, chains and acceptance counts equal to the last bit with
    the cache on and off."""
    from bayesgm_amd import _lib
    u = SHAPES[shape]
    m = _model(31, [2, 2, 2, 6], 60, binary, **u)
    x, y, v = _data(700, 60, 32, binary)
    eng = _engine(m, u)
    xs = np.linspace(0, 3, 9)
    kw = dict(effect=_lib.EFFECT_ITE) if binary else dict(effect=_lib.EFFECT_ADRF, x_values=xs)
    res = {}
    for on in (True, False):
        eng.set_outcome_cache(on)
        eng.outcome_cache_stats(reset=True)
        out = eng.mh_sample(x, y, v, 20, 40, 1.5, 77, want_draws=True, sample_y=True, **kw)
        eff = (out["ite"] if binary else out["adrf"]).cpu().numpy()
        res[on] = (eff, out["draws"].cpu().numpy(), out["acc_count"].cpu().numpy(), eng.outcome_cache_stats())
    eng.set_outcome_cache(True)
    print("served from cache: %d of %d retained tile-iterations" % res[True][3])
    assert res[True][3][0] > 0 and res[False][3][0] == 0 and res[True][3][1] == 44 * 40
    for a, b in zip(res[True][:3], res[False][:3]):
        assert np.array_equal(a, b)


@pytest.mark.parametrize("case", [CASES[0], CASES[2], CASES[3]])
def test_evaluate_matches_oracle(case):
    import torch
    from oracle.nets import mlp_forward
    u = SHAPES[case["shape"]]
    m = _model(31, case["z_dims"], case["p"], case["binary"], **u)
    x, y, v = _data(case["n"], case["p"], 32, case["binary"])
    z = np.random.RandomState(33).randn(case["n"], sum(case["z_dims"])).astype(np.float32)
    eng = _engine(m, u)
    xs = np.linspace(0.1, 2.9, 200)
    T = lambda a_: torch.from_numpy(np.ascontiguousarray(a_)).to(eng.device)
    sums, causal = eng.evaluate(T(x.ravel()), T(y.ravel()), T(v), T(z), x_values=None if case["binary"] else xs)
    sums = sums.cpu().numpy()
    n = case["n"]
    gv, gx, gy = sums[0] / (n * case["p"]), sums[1] / n, sums[2] / n
    causal = causal.cpu().numpy() if case["binary"] else causal.cpu().numpy() / n
    m64 = OC.cast_model(m, np.float64)
    z64 = z.astype(np.float64)
    z0d, z1d, z2d, _ = case["z_dims"]
    g_out = mlp_forward(m64["g"], z64)
    mv = ((v - g_out[:, :case["p"]]) ** 2).mean()
    h_out = mlp_forward(m64["h"], np.concatenate([z64[:, :z0d], z64[:, z0d + z1d:z0d + z1d + z2d]], axis=1))[:, 0]
    xp = 1.0 / (1.0 + np.exp(-h_out)) if case["binary"] else h_out
    mx = ((x[:, 0] - xp) ** 2).mean()
    fy = lambda xv: mlp_forward(m64["f"], np.concatenate([z64[:, :z0d + z1d], xv], axis=1))[:, 0]
    my = ((y[:, 0] - fy(x.astype(np.float64))) ** 2).mean()
    assert abs(gv - mv) <= 1e-4 * mv and abs(gx - mx) <= 1e-4 * max(mx, 1e-3) and abs(gy - my) <= 1e-4 * my
    if case["binary"]:
        ref = fy(np.ones((n, 1))) - fy(np.zeros((n, 1)))
    else:
        ref = np.array([fy(np.full((n, 1), t)).mean() for t in xs])
    assert np.abs(np.asarray(causal) - ref).max() <= 2e-4


def _flat(grads):
    return np.concatenate([np.concatenate([dW.ravel(), db.ravel()]) for dW, db in grads])


@pytest.mark.parametrize("case,B", [(CASES[0], 32), (CASES[1], 20), (CASES[2], 32), (CASES[3], 7), (CASES[4], 32), (CASES[4], 100), (CASES[5], 45)])
def test_fit_gradients_match_oracle(case, B):
    """theta gradients (g, f, h), the reported losses and the latent gradient of one minibatch -- including minibatches that are not
    a multiple of the 32-row tile and one larger than it."""
    import torch
    u = SHAPES[case["shape"]]
    n = max(case["n"], B + 5)
    m = _model(7, case["z_dims"], case["p"], case["binary"], **u)
    x, y, v = _data(n, case["p"], 8, case["binary"])
    z = np.random.RandomState(9).randn(n, sum(case["z_dims"])).astype(np.float32)
    eng = _engine(m, u)
    dev = eng.device
    xd, yd, vd, zd = (torch.from_numpy(a).to(dev) for a in (x.ravel(), y.ravel(), v, z))
    idx_np = np.random.RandomState(3).choice(n, B, replace=False).astype(np.int32)
    idx = torch.from_numpy(idx_np).to(dev)
    npar = eng.fit_begin(n, B)
    assert "gx_causal_fit_kernel" in eng.describe(B)
    grad = torch.empty(npar, device=dev)
    loss = torch.zeros(8, device=dev, dtype=torch.float64)
    eng.fit_theta_grad(xd, yd, vd, zd, idx, B, grad, loss)
    m64 = OC.cast_model(m, np.float64)
    bz, bx, by, bv = (a[idx_np].astype(np.float64) for a in (z, x, y, v))
    lv, _, gg, _ = OF.g_loss_and_grads(m64, bz, bv)
    lx, _, gh, _ = OF.h_loss_and_grads(m64, bz, bx)
    ly, _, gf, _ = OF.f_loss_and_grads(m64, bz, bx, by)
    got = grad.cpu().numpy()
    o = 0
    for part in (_flat(gg), _flat(gf), _flat(gh)):
        g_ = got[o:o + part.size]
        assert np.abs(g_ - part).max() <= 5e-5 * np.abs(part).max() + 1e-7, (np.abs(g_ - part).max(), np.abs(part).max())
        o += part.size
    l = loss.cpu().numpy()
    assert np.allclose([l[0] / B, l[2] / B, l[4] / B], [lv, lx, ly], rtol=5e-5)
    zm = torch.zeros_like(zd); zv = torch.zeros_like(zd)
    z_before = zd.clone()
    loss.zero_()
    eng.fit_z_step(xd, yd, vd, zd, zm, zv, idx, B, 1e-3, lazy=True, loss=loss)
    lz_ref, dz_ref = OF.z_loss_and_grad(m64, bz, bx, by, bv)
    assert np.isclose(loss.cpu().numpy()[6] / B, lz_ref, rtol=5e-5)
    gm = zm.cpu().numpy()[idx_np] / 0.1
    assert np.abs(gm - dz_ref).max() <= 5e-5 * np.abs(dz_ref).max() + 1e-8
    untouched = np.setdiff1d(np.arange(n), idx_np)
    assert torch.equal(zd[untouched], z_before[untouched])
    eng.fit_end()


@pytest.mark.parametrize("shape", ["r_test", "w256"])
def test_fit_steps_then_sampling_with_the_trained_parameters(shape):
    """Four Adam iterations (the last minibatch short) track the oracle; the log-posterior DURING the fit session reads the padded
    copies the Adam kernel keeps current, and after fit_end the copies rebuilt from the host parameters give the same values."""
    import torch
    u = SHAPES[shape]
    z_dims, p, n = [3, 6, 3, 6], 50, 96
    m = _model(11, z_dims, p, True, **u)
    x, y, v = _data(n, p, 12, True)
    z = np.random.RandomState(13).randn(n, sum(z_dims)).astype(np.float32)
    eng = _engine(m, u)
    dev = eng.device
    xd, yd, vd, zd = (torch.from_numpy(a).to(dev) for a in (x.ravel(), y.ravel(), v, z.copy()))
    zm = torch.zeros_like(zd); zv = torch.zeros_like(zd)
    B, lr = 32, 1e-3
    npar = eng.fit_begin(n, B)
    grad = torch.empty(npar, device=dev)
    st = OF.FitState(OC.cast_model(m, np.float64), z.astype(np.float64), lr, lr)
    x64, y64, v64 = (a.astype(np.float64) for a in (x, y, v))
    rs = np.random.RandomState(5)
    for step in range(4):
        idx_np = rs.choice(n, B if step < 3 else 17, replace=False).astype(np.int32)
        idx = torch.from_numpy(idx_np).to(dev)
        eng.fit_theta_grad(xd, yd, vd, zd, idx, len(idx_np), grad)
        eng.fit_theta_apply(grad, lr)
        eng.fit_z_step(xd, yd, vd, zd, zm, zv, idx, len(idx_np), lr, lazy=False)
        OF.fit_step(st, x64, y64, v64, idx_np, lazy_z=False)
    assert np.abs(zd.cpu().numpy() - st.data_z).max() <= 3e-4
    lp = eng.logpost(xd, yd, vd, zd).cpu().numpy()
    ref = OC.log_posterior(dict(st.m), x64, y64, v64, st.data_z)
    assert np.abs(lp - ref).max() <= 5e-4          # measured 7.5e-6 on |log p| <= 75
    eng.fit_end()
    lp2 = eng.logpost(xd, yd, vd, zd).cpu().numpy()
    assert np.abs(lp - lp2).max() <= 1e-5 * np.abs(lp).max()


def test_default_widths_fit_beyond_the_chain_envelope():
    """Default hidden widths at v_dim = 500 (no row-tile chain, no LDS blob): fit on the general-width engine, a p = 500 CausalBGM
    trains and its four-step Adam trace tracks oracle.fit; and a 100-row minibatch of a chain-only shape."""
    import torch
    from bayesgm_amd.engine import CausalEngine
    z_dims, p, n = [1, 1, 1, 7], 500, 128
    m = _model(51, z_dims, p, False)
    x, y, v = _data(n, p, 52)
    z = np.random.RandomState(53).randn(n, sum(z_dims)).astype(np.float32)
    eng = CausalEngine(p, z_dims); eng.set_model(g=m["g"], f=m["f"], h=m["h"], e=m["e"])
    dev = eng.device
    xd, yd, vd, zd = (torch.from_numpy(a).to(dev) for a in (x.ravel(), y.ravel(), v, z.copy()))
    zm = torch.zeros_like(zd); zv = torch.zeros_like(zd)
    B, lr = 32, 1e-3
    npar = eng.fit_begin(n, B)
    assert "gx_causal_fit_kernel" in eng.describe(B)
    grad = torch.empty(npar, device=dev)
    st = OF.FitState(OC.cast_model(m, np.float64), z.astype(np.float64), lr, lr)
    x64, y64, v64 = (a.astype(np.float64) for a in (x, y, v))
    rs = np.random.RandomState(5)
    for step in range(4):
        idx_np = rs.choice(n, B if step < 3 else 17, replace=False).astype(np.int32)
        idx = torch.from_numpy(idx_np).to(dev)
        eng.fit_theta_grad(xd, yd, vd, zd, idx, len(idx_np), grad)
        eng.fit_theta_apply(grad, lr)
        eng.fit_z_step(xd, yd, vd, zd, zm, zv, idx, len(idx_np), lr, lazy=False)
        OF.fit_step(st, x64, y64, v64, idx_np, lazy_z=False)
    assert np.abs(zd.cpu().numpy() - st.data_z).max() <= 3e-4
    lp = eng.logpost(xd, yd, vd, zd).cpu().numpy()          # (streamed-fragment sampling path, repacked from the device parameters)
    ref = OC.log_posterior(dict(st.m), x64, y64, v64, st.data_z)
    assert np.abs(lp - ref).max() <= 1e-3
    eng.fit_end()


def test_forced_general_width_engine_equals_resident_kernels():
    """The bench shape through both kernel families: chains, effects and log-posteriors agree to rounding."""
    code = r'''
import numpy as np, sys
sys.path.insert(0, %r)
from oracle import causal as OC
from bayesgm_amd.engine import CausalEngine
from bayesgm_amd import _lib
m = OC.init_model(3, [1, 1, 1, 7], 200)
rs = np.random.RandomState(4)
n = 400
v = rs.randn(n, 200).astype(np.float32); x = rs.exponential(size=(n, 1)).astype(np.float32); y = (x + rs.randn(n, 1)).astype(np.float32)
eng = CausalEngine(200, [1, 1, 1, 7]); eng.set_model(g=m["g"], f=m["f"], h=m["h"], e=m["e"])
z = rs.randn(n, 10).astype(np.float32)
lp = eng.logpost(x.ravel(), y.ravel(), v, z).cpu().numpy()
enc = eng.encode(v).cpu().numpy()
out = eng.mh_sample(x, y, v, 30, 10, 0.4, 77, want_draws=True, effect=_lib.EFFECT_ADRF, x_values=np.linspace(0, 3, 20))
np.savez(sys.argv[1], lp=lp, enc=enc, draws=out["draws"].cpu().numpy(), adrf=out["adrf"].cpu().numpy(), path=eng.describe())
''' % ROOT
    import tempfile
    with tempfile.TemporaryDirectory() as d:
        res = {}
        for tag, env in (("resident", {}), ("gx", {"BGM_FORCE_GX": "1"})):
            path = os.path.join(d, tag + ".npz")
            subprocess.run([sys.executable, "-c", code, path], check=True, env=dict(os.environ, **env), timeout=600)
            res[tag] = np.load(path)
        a, b = res["resident"], res["gx"]
        assert "gx_causal_mh_kernel" in str(b["path"]) and "gx_causal_mh_kernel" not in str(a["path"])
        assert np.abs(a["lp"] - b["lp"]).max() <= 1e-5 * np.abs(a["lp"]).max() + 1e-3
        assert np.abs(a["enc"] - b["enc"]).max() <= 1e-5
        same = np.all(np.abs(a["draws"][-1] - b["draws"][-1]) <= 1e-4, axis=1).mean()
        assert same >= 0.97, same
        assert np.abs(a["adrf"] - b["adrf"]).max() <= 5e-3


# ---------------------------------------------------------------------------------------------------------------------------
# the class surface on the parameter dicts of the reference's own integration tests
# ---------------------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("binary", [True, False])
def test_r_integration_test_parameters_through_the_class(tmp_path, binary):
    """r-package/bayesgm/tests/testthat/test-causalbgm.R:17-60 / :62-120: n = 64, v_dim = 4, z_dims (1,1,1,1), g / e_units (8, 8),
    f / h / dz_units (8, 4), fit(epochs = 0, use_egm_init = False), predict(n_mcmc = 5, burn_in = 10, q_sd = 0.5) -- and the same
    model through egm_init, fit(epochs = 5), evaluate."""
    from bayesgm_amd.models import CausalBGM
    rs = np.random.RandomState(123)
    n = 64
    v = rs.randn(n, 4).astype(np.float32)
    if binary:
        x = (rs.rand(n, 1) < 0.5).astype(np.float32)
        y = (1.5 * x[:, :1] + 0.7 * v[:, :1] - 0.3 * v[:, 1:2] + 0.25 * rs.randn(n, 1)).astype(np.float32)
    else:
        x = rs.exponential(size=(n, 1)).astype(np.float32)
        y = (x + 0.5 * v[:, :1] + 0.3 * rs.randn(n, 1)).astype(np.float32)
    params = dict(dataset="RBinary" if binary else "RContinuous", output_dir=str(tmp_path), save_res=False, save_model=False,
                  binary_treatment=binary, use_bnn=False, z_dims=[1, 1, 1, 1], v_dim=4, lr_theta=1e-4, lr_z=1e-4, lr=2e-4, g_d_freq=5,
                  use_z_rec=True, kl_weight=1e-4, g_units=[8, 8], e_units=[8, 8], f_units=[8, 4], h_units=[8, 4], dz_units=[8, 4])
    m = CausalBGM(dict(params), timestamp="t", random_seed=1)
    m.fit((x, y, v), epochs=0, epochs_per_eval=1, batch_size=32, use_egm_init=False, egm_n_iter=0, egm_batches_per_eval=1, verbose=0)
    kw = {} if binary else dict(x_values=[0.0, 1.0, 2.0])
    eff, iv = m.predict((x, y, v), n_mcmc=5, burn_in=10, q_sd=0.5, **kw)
    if binary:
        assert eff.shape == (n,) and iv.shape == (n, 2)
    else:
        assert eff.shape == (3,) and iv.shape == (3, 2)
    assert np.isfinite(eff).all() and np.isfinite(iv).all()
    m2 = CausalBGM(dict(params), timestamp="t2", random_seed=2)
    m2.egm_init((x, y, v), egm_n_iter=40, batch_size=32, egm_batches_per_eval=20, verbose=0)
    m2.fit((x, y, v), epochs=5, epochs_per_eval=1, batch_size=32, use_egm_init=False, verbose=0)
    causal_pre, mse_x, mse_y, mse_v = m2.evaluate((x, y, v))
    assert np.isfinite(np.asarray(causal_pre)).all() and np.isfinite([mse_x, mse_y, mse_v]).all()
    eff, iv = m2.predict((x, y, v), n_mcmc=5, burn_in=10, q_sd=0.5, **kw)
    assert np.isfinite(eff).all()


def test_reference_default_nb_units_through_the_class(tmp_path):
    """nb_units = [256, 256, 256] (the default of BaseFullyConnectedNet, networks/base.py:7) for every network: warm start, five
    epochs (short last minibatch), evaluate, predict."""
    from bayesgm_amd.models import CausalBGM
    rs = np.random.RandomState(0)
    n, p = 150, 30
    v = rs.randn(n, p).astype(np.float32)
    x = rs.exponential(size=(n, 1)).astype(np.float32)
    y = (x + 0.3 * v[:, :1] + rs.randn(n, 1)).astype(np.float32)
    params = dict(dataset="w256", output_dir=str(tmp_path), save_res=False, save_model=False, binary_treatment=False, use_bnn=False,
                  z_dims=[1, 1, 1, 7], v_dim=p, lr_theta=1e-4, lr_z=1e-4, lr=2e-4, g_d_freq=5, use_z_rec=True, kl_weight=1e-4,
                  g_units=[256] * 3, e_units=[256] * 3, f_units=[256] * 3, h_units=[256] * 3, dz_units=[256] * 3)
    m = CausalBGM(params, timestamp="t", random_seed=1)
    m.fit((x, y, v), epochs=5, epochs_per_eval=5, batch_size=32, use_egm_init=True, egm_n_iter=20, egm_batches_per_eval=10, verbose=0)
    assert m.data_z.shape == (n, 10)
    causal_pre, mse_x, mse_y, mse_v = m.evaluate((x, y, v))
    assert np.isfinite(np.asarray(causal_pre)).all() and np.isfinite([mse_x, mse_y, mse_v]).all()
    eff, iv = m.predict((x, y, v), alpha=0.05, n_mcmc=5, burn_in=10, x_values=np.linspace(0, 2, 4), q_sd=0.5)
    assert eff.shape == (4,) and iv.shape == (4, 2) and np.isfinite(eff).all()


# ===========================================================================================================================
# BGM: generators g_net = BaseVariationalNet with arbitrary trunk widths / depth / latent width (csrc/gx_bgm_api.hip)
# ===========================================================================================================================
from oracle import bgm as OB  # noqa: E402


def _bgm_model(seed, q, p, units):
    m = OB.init_model(seed, q, p, g_units=tuple(units))
    rs = np.random.RandomState(seed + 7)
    g = m["g"]
    g["bn"].update(gamma=(1 + 0.1 * rs.randn(q)).astype(np.float32), beta=(0.1 * rs.randn(q)).astype(np.float32),
                   mean=(0.2 * rs.randn(q)).astype(np.float32), var=(0.5 + rs.rand(q)).astype(np.float32))
    g["trunk"] = [(W, (0.1 * rs.randn(*b.shape)).astype(np.float32)) for W, b in g["trunk"]]
    g["mean"] = (g["mean"][0], (0.1 * rs.randn(p)).astype(np.float32))
    g["var"] = (g["var"][0], (0.1 * rs.randn(p)).astype(np.float32))
    return m


def _bgm_data(n, p, seed, miss=0.2):
    rs = np.random.RandomState(seed)
    x = rs.randn(n, p).astype(np.float32)
    x[rs.rand(n, p) < miss] = np.nan
    x[0, :] = np.nan
    if n > 1:
        x[1, :] = rs.randn(p)
    return x


def _bgm_engine(m):
    from bayesgm_amd.engine import BgmEngine
    eng = BgmEngine(m["x_dim"], m["z_dim"], g_units=[W.shape[1] for W, _ in m["g"]["trunk"]])
    eng.set_weights(m["g"])
    return eng


BGM_CASES = [
    dict(q=3, p=8, n=96, units=(8, 8)),                  # r-package/bayesgm/tests/testthat/test-bgm.R:17-50
    dict(q=10, p=100, n=333, units=(128, 128)),
    dict(q=10, p=50, n=70, units=(256, 256, 256)),       # networks/base.py:56 default nb_units
    dict(q=20, p=500, n=130, units=(64, 64)),            # z_dim > 16, two hidden layers, wide data
    dict(q=5, p=37, n=33, units=(40, 24, 100, 9)),
]


@pytest.mark.parametrize("case", BGM_CASES)
def test_bgm_logpost_and_gradient_match_oracle(case):
    m = _bgm_model(1, case["q"], case["p"], case["units"])
    x = _bgm_data(case["n"], case["p"], 2)
    z = np.random.RandomState(3).randn(case["n"], case["q"]).astype(np.float32)
    eng = _bgm_engine(m)
    lp, gr = eng.logpost(z, x, want_grad=True)
    lp0 = eng.logpost(z, x)
    obs, clean = OB.obs_mask_of(x)
    m64 = OB.cast_model(m, np.float64)
    ref_lp, ref_gr = OB.log_posterior_and_grad(m64, z.astype(np.float64), clean.astype(np.float64), obs.astype(np.float64))
    lp, gr, lp0 = lp.cpu().numpy(), gr.cpu().numpy(), lp0.cpu().numpy()
    assert np.abs(lp - lp0).max() <= 1e-5 * np.abs(lp).max()
    assert np.all(np.abs(lp - ref_lp) <= 2e-6 * np.abs(ref_lp) + 2e-4), np.abs(lp - ref_lp).max()
    assert np.abs(gr - ref_gr).max() <= 2e-5 * np.abs(ref_gr).max() + 2e-5, np.abs(gr - ref_gr).max()
    assert abs(lp[0] + 0.5 * (z[0] ** 2).sum()) < 1e-5 and np.allclose(gr[0], -z[0], atol=1e-6)


@pytest.mark.parametrize("case", BGM_CASES[:4])
def test_bgm_hmc_chain_and_step_adaptation_match_oracle(case):
    import torch
    m = _bgm_model(11, case["q"], case["p"], case["units"])
    x = _bgm_data(case["n"], case["p"], 12)
    burn, keep, L, seed = 20, 10, 4, 77
    eng = _bgm_engine(m)
    out = eng.hmc_sample(x, keep, burn, step_size=0.02, n_leapfrog=L, seed=seed)
    obs, clean = OB.obs_mask_of(x)
    ref, info = OB.hmc_sampler(m, clean, obs.astype(np.float32), keep, burn, 0.02, L, seed, return_info=True)
    draws = out["draws"].cpu().numpy()
    assert draws.shape == ref.shape
    ok = np.all(np.abs(draws[-1] - ref[-1]) <= 2e-3, axis=1)
    assert ok.mean() >= 0.97, ok.mean()
    assert abs(float(out["step"].item()) / info["step"] - 1) < 1e-5
    acc = out["acc_count"].cpu().numpy()[burn:].sum() / (keep * case["n"])
    assert abs(acc - info["accept_rate"]) < 0.03 and acc > 0.5
    out2 = eng.hmc_sample(x, keep, burn, step_size=0.02, n_leapfrog=L, seed=seed)
    assert torch.equal(out2["draws"], out["draws"])


@pytest.mark.parametrize("case", [BGM_CASES[0], BGM_CASES[2], BGM_CASES[3]])
def test_bgm_predictive_draws_match_oracle_on_same_latents(case):
    import torch
    q, p = case["q"], case["p"]
    m = _bgm_model(31, q, p, case["units"])
    rs = np.random.RandomState(32)
    draws = rs.randn(6, 40, q).astype(np.float32)
    eng = _bgm_engine(m)
    ref = OB.predict_on_posteriors(OB.cast_model(m, np.float64), draws.astype(np.float64), seed=9, burn_in=13)
    _, full = eng.predict_draws(torch.from_numpy(draws).cuda(), 13, 9, want_full=True)
    assert np.abs(full.cpu().numpy() - ref).max() <= 2e-4
    miss = rs.rand(40, p) < 0.2
    slot = np.full((40, p), -1, np.int32)
    k = 0
    for i in range(40):
        c = np.where(miss[i])[0]
        slot[i, c] = np.arange(len(c))
        k = max(k, len(c))
    cells, _ = eng.predict_draws(torch.from_numpy(draws).cuda(), 13, 9, slot=torch.from_numpy(slot).cuda(), k_slots=k)
    cells = cells.cpu().numpy().reshape(40, k, 6)
    for i in range(40):
        c = np.where(miss[i])[0]
        if len(c):
            assert np.abs(cells[i, :len(c)] - ref[:, i, c].T).max() <= 2e-4


@pytest.mark.parametrize("case,B", [(BGM_CASES[0], 32), (BGM_CASES[1], 20), (BGM_CASES[2], 32), (BGM_CASES[3], 77)])
def test_bgm_fit_steps_match_oracle(case, B):
    """BGM fit step functions (training-mode BatchNorm, per-dimension variance head, fresh-slot Adam on Z) on the general engine."""
    import torch
    q, p, lr = case["q"], case["p"], 2e-3
    n = max(96, B + 40)
    m = _bgm_model(61, q, p, case["units"])
    rs = np.random.RandomState(62)
    x = rs.randn(n, p).astype(np.float32)
    z = rs.randn(n, q).astype(np.float32)
    eng = _bgm_engine(m)
    xd, zd = torch.from_numpy(x).cuda(), torch.from_numpy(z.copy()).cuda()
    npar = eng.fit_begin(n, B)
    grad = torch.empty(npar, device="cuda")
    loss = torch.zeros(4, device="cuda", dtype=torch.float64)
    m64 = OB.cast_model(m, np.float64)
    st = OB.BgmFitState(m64, z.astype(np.float64), lr, lr)
    x64 = x.astype(np.float64)
    idx_np = rs.choice(n, B, replace=False).astype(np.int32)
    idx = torch.from_numpy(idx_np).cuda()
    eng.fit_theta_grad(xd, zd, idx, grad, loss)
    l_ref, mse_ref, gr, _, _ = OB.g_loss_and_grads(m64, st.data_z[idx_np], x64[idx_np])
    ref = np.concatenate([a.ravel() for a in OB._flat_bgm_grads(gr)])
    got = grad.cpu().numpy()
    got_t = np.concatenate([got[:2 * q], got[4 * q:]])
    assert np.all(got[2 * q:4 * q] == 0)
    assert np.abs(got_t - ref).max() <= 2e-5 * np.abs(ref).max() + 1e-7, np.abs(got_t - ref).max()
    assert np.isclose(loss.cpu().numpy()[0] / B, l_ref, rtol=2e-5)
    eng.fit_end()
    eng.set_weights(m["g"])
    eng.fit_begin(n, B)
    for step in range(3):
        idx_np = rs.choice(n, B, replace=False).astype(np.int32)
        idx = torch.from_numpy(idx_np).cuda()
        eng.fit_theta_grad(xd, zd, idx, grad)
        eng.fit_theta_apply(grad, lr)
        eng.fit_z_step(xd, zd, idx, lr)
        OB.fit_step(st, x64, idx_np)
    assert np.abs(zd.cpu().numpy() - st.data_z).max() <= 5e-4
    # the log posterior DURING the session reads the packs the Adam kernel keeps current and the device BatchNorm statistics
    lp_in = eng.logpost(zd, xd).cpu().numpy()
    g_tr = eng.get_weights()
    for k in ("gamma", "beta", "mean", "var"):
        assert np.abs(g_tr["bn"][k] - m64["g"]["bn"][k]).max() <= 5e-4, k
    for (W, b), (Wr, br) in zip(g_tr["trunk"], m64["g"]["trunk"]):
        assert np.abs(W - Wr).max() <= 5e-4 and np.abs(b - br).max() <= 5e-4
    for k in ("mean", "var"):
        assert np.abs(g_tr[k][0] - m64["g"][k][0]).max() <= 5e-4
    eng.fit_end()
    lp = eng.logpost(zd, xd).cpu().numpy()
    ref_lp = OB.log_posterior(m64, st.data_z, x64)
    assert np.abs(lp - ref_lp).max() <= 0.1
    assert np.abs(lp - lp_in).max() <= 1e-5 * np.abs(lp).max() + 1e-4


def test_bgm_r_integration_test_parameters_through_the_class(tmp_path):
    """r-package/bayesgm/tests/testthat/test-bgm.R:17-66: x_dim 8, z_dim 3, g / e_units (8, 8), dz / dx_units (8, 4); fit(epochs = 0,
    use_egm_init = False); predict(bs = 16, n_mcmc = 5, burn_in = 10, step_size 0.01, num_leapfrog_steps = 3) on rows whose last column
    is missing -- and the same model through egm_init, fit(epochs = 5), evaluate, generate."""
    from bayesgm_amd.models import BGM
    from bayesgm_amd.datasets import simulate_z_hetero
    X, Y = simulate_z_hetero(n=128, k=3, d=7, seed=42)
    data = np.c_[X, Y].astype(np.float32)
    train, test = data[:96], data[96:].copy()
    test[:, -1] = np.nan
    params = dict(dataset="RSimHetero", output_dir=str(tmp_path), save_res=False, save_model=False, use_bnn=False, z_dim=3, x_dim=8,
                  lr_theta=5e-3, lr_z=5e-3, g_units=[8, 8], e_units=[8, 8], dz_units=[8, 4], dx_units=[8, 4], kl_weight=5e-5, lr=1e-3,
                  g_d_freq=1, use_z_rec=True, alpha=0.0, gamma=0.0)
    model = BGM(dict(params), random_seed=1)
    model.fit(train, batch_size=32, epochs=0, epochs_per_eval=1, use_egm_init=False, egm_n_iter=0, egm_batches_per_eval=1, verbose=0)
    imputed, interval = model.predict(test, alpha=0.05, bs=16, n_mcmc=5, burn_in=10, step_size=0.01, num_leapfrog_steps=3, seed=42)
    assert imputed.shape == test.shape and np.asarray(interval).shape == (len(test), 1, 2)
    assert not np.isnan(imputed[:, -1]).any()
    m2 = BGM(dict(params), random_seed=2)
    m2.fit(train, batch_size=32, epochs=5, epochs_per_eval=5, use_egm_init=True, egm_n_iter=40, egm_batches_per_eval=20, verbose=0)
    res = m2.evaluate(train)
    assert np.isfinite(np.asarray(res[0] if isinstance(res, (tuple, list)) else res)).all()
    gen = m2.generate(nb_samples=10)
    assert np.asarray(gen).shape[-1] == 8 and np.isfinite(np.asarray(gen)).all()
    imputed, interval = m2.predict(test, alpha=0.05, bs=16, n_mcmc=5, burn_in=10, step_size=0.01, num_leapfrog_steps=3, seed=42)
    assert np.isfinite(imputed).all()


def test_bgm_default_nb_units_through_the_class(tmp_path):
    """g_units = e_units = [256, 256, 256] (networks/base.py:56): warm start, five epochs, predict."""
    from bayesgm_amd.models import BGM
    from bayesgm_amd.datasets import simulate_z_hetero
    X, Y = simulate_z_hetero(n=200, k=3, d=19, seed=1)
    data = np.c_[X, Y].astype(np.float32)
    params = dict(dataset="w256", output_dir=str(tmp_path), save_res=False, save_model=False, use_bnn=False, z_dim=10, x_dim=20,
                  lr_theta=2e-3, lr_z=2e-3, g_units=[256] * 3, e_units=[256] * 3, dz_units=[256] * 3, dx_units=[256] * 3, kl_weight=5e-5,
                  lr=1e-3, g_d_freq=1, use_z_rec=True, alpha=0.0, gamma=0.0)
    model = BGM(params, random_seed=3)
    model.fit(data, batch_size=32, epochs=5, epochs_per_eval=5, use_egm_init=True, egm_n_iter=20, egm_batches_per_eval=10, verbose=0)
    test = data[:40].copy()
    test[:, -1] = np.nan
    imputed, interval = model.predict(test, alpha=0.05, bs=20, n_mcmc=5, burn_in=10, step_size=0.01, num_leapfrog_steps=3, seed=4)
    assert imputed.shape == test.shape and np.isfinite(imputed).all() and np.asarray(interval).shape == (40, 1, 2)
