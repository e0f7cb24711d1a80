"""Split-precision sampling kernels (csrc/causal_bx3_kernels.h, opt-in through bgm_causal_set_precision /
params['mh_precision'] = 'bf16x3' | 'f16x3') against the float64 oracle and against the fp32 kernels.  The unit is compiled for two
16-bit operand formats: bf16 (8 + 8 mantissa bits per hi / lo pair, fp32 range) and fp16 (11 + 11 bits, fp16 range: |activation| < 65504).

The arithmetic differs from the reference's fp32 (three bf16 products per contraction, fp32 accumulation, ~6e-6 relative per
layer), so parity is stated as tolerances, written here:
  log-posterior   |bx3 - float64 oracle| <= 2e-5 * |lp| + 2e-3      (fp32 kernel: 2e-6 * |lp| + 2e-4; observed 3e-5 ... 9e-4)
  chains          same Philox streams as the fp32 kernel; accept / reject decisions flip where |log u - dlogp| is below the
                  arithmetic's error, so chains agree statistically: acceptance rate within 0.01, per-row posterior means
                  of z within 0.15 posterior sd on average, ADRF within 0.02 of the fp32 kernel's on the same draws' law
  effects         on GIVEN draws (standalone path is fp32) not applicable; the fused ADRF is compared through the chain test.
f16x3 carries 22 mantissa bits through every contraction: its log-posterior bound is the fp32 kernel's own."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from oracle import causal as OC  # noqa: E402
from tests.test_gpu_causal import _model, _data, _engine  # noqa: E402


@pytest.mark.parametrize("case", [dict(z_dims=[1, 1, 1, 7], p=200, binary=False, n=517),
                                  dict(z_dims=[3, 3, 6, 6], p=100, binary=True, n=300),
                                  dict(z_dims=[1, 1, 1, 7], p=20, binary=False, n=100),
                                  dict(z_dims=[1, 1, 1, 7], p=50, binary=False, n=33),
                                  dict(z_dims=[2, 2, 2, 6], p=150, binary=True, n=64),
                                  dict(z_dims=[3, 3, 6, 6], p=17, binary=False, n=16)])
@pytest.mark.parametrize("mode", ["bf16x3", "f16x3"])
def test_bx3_log_posterior_matches_oracle(case, mode):
    m = _model(3, case["z_dims"], case["p"], case["binary"])
    x, y, v = _data(case["n"], case["p"], 4, case["binary"])
    z = np.random.RandomState(5).randn(case["n"], sum(case["z_dims"])).astype(np.float32)
    eng = _engine(m)
    ref = OC.log_posterior(OC.cast_model(m, np.float64), x.astype(np.float64), y.astype(np.float64), v.astype(np.float64),
                           z.astype(np.float64))
    lp32 = eng.logpost(x.ravel(), y.ravel(), v, z).cpu().numpy()
    eng.set_precision(mode)
    lpbx = eng.logpost(x.ravel(), y.ravel(), v, z).cpu().numpy()
    eng.set_precision("fp32")
    err32, errbx = np.abs(lp32 - ref), np.abs(lpbx - ref)
    print("logpost err: fp32 max %.2e, %s max %.2e (|lp| ~ %.0f)" % (err32.max(), mode, errbx.max(), np.abs(ref).mean()))
    rel, ab = (2e-5, 2e-3) if mode == "bf16x3" else (2e-6, 2e-4)          # f16x3: the fp32 kernel's own bound (observed <= 3.5e-5)
    assert np.all(errbx <= rel * np.abs(ref) + ab), errbx.max()


@pytest.mark.parametrize("mode", ["bf16x3", "f16x3"])
def test_bx3_chains_agree_statistically_with_fp32(mode):
    from bayesgm_amd import _lib
    m = _model(7, [1, 1, 1, 7], 200)
    x, y, v = _data(2048, 200, 8)
    xs = np.linspace(0, 3, 20)
    eng = _engine(m)
    burn, keep, seed = 300, 200, 31
    outs = {}
    for mode_ in ("fp32", mode):
        eng.set_precision(mode_)
        out = eng.mh_sample(x, y, v, burn, keep, 0.3, seed, want_draws=True, effect=_lib.EFFECT_ADRF, x_values=xs)
        outs[mode_] = dict(draws=out["draws"].cpu().numpy(), adrf=out["adrf"].cpu().numpy(),
                          acc=out["acc_count"].cpu().numpy().sum() / ((burn + keep) * len(x)))
    eng.set_precision("fp32")
    a, b = outs["fp32"], outs[mode]
    print("acceptance fp32 %.4f %s %.4f" % (a["acc"], mode, b["acc"]))
    assert abs(a["acc"] - b["acc"]) <= 0.01
    # identical streams: most chains are still draw-for-draw identical after 500 transitions
    same = np.all(np.abs(a["draws"][-1] - b["draws"][-1]) <= 1e-3, axis=1).mean()
    print("rows whose last draw coincides: %.3f" % same)
    assert same >= 0.95                                                    # observed 0.995 (bf16x3), 0.997 (f16x3)
    ma, mb = a["draws"].mean(axis=0), b["draws"].mean(axis=0)
    sd = a["draws"].std(axis=0) + 1e-3
    assert np.mean(np.abs(ma - mb) / sd) <= 0.15
    d = np.abs(a["adrf"].mean(axis=1) - b["adrf"].mean(axis=1)).max()
    print("ADRF (mean over draws) max diff %.2e" % d)
    assert d <= 0.02


@pytest.mark.parametrize("mode", ["bf16x3", "f16x3"])
def test_bx3_ite_and_class_predict(mode):
    """params['mh_precision'] = 'bf16x3' through the class: binary-treatment ITEs with intervals, against the fp32 class."""
    from bayesgm_amd.models import CausalBGM
    from tests.test_gpu_class_level import _params, GOLD
    from bayesgm_amd.datasets import binarize_treatment
    g = np.load(GOLD)
    x, y, v = binarize_treatment(g["x"][:600]), g["y"][:600], g["v"][:600]
    res = {}
    for mode_ in ("fp32", mode):
        model = CausalBGM(dict(_params(binary=True, z_dims=(3, 3, 6, 6)), mh_precision=mode_), random_seed=12)
        res[mode_] = model.predict((x, y, v), alpha=0.05, n_mcmc=100, burn_in=150, q_sd=0.5, sample_y=True, verbose=0)
    ite32, int32 = res["fp32"]
    itebx, intbx = res[mode]
    assert itebx.shape == (600,) and intbx.shape == (600, 2) and np.all(np.isfinite(itebx)) and np.all(intbx[:, 0] <= intbx[:, 1])
    same = (np.abs(ite32 - itebx) <= 1e-3).mean()
    print("ITE rows agreeing with fp32: %.3f, ATE fp32 %.4f %s %.4f" % (same, ite32.mean(), mode, itebx.mean()))
    assert same >= 0.8 and abs(float(ite32.mean()) - float(itebx.mean())) <= 0.01


@pytest.mark.parametrize("mode", ["bf16x3", "f16x3"])
def test_bx3_with_the_conditional_prior(mode):
    """IdentifiableCausalBGM's prior p(z | u) inside the split-precision kernels (the row's table entry in registers, the difference to the
    standard-normal term added in fp32 as the PRIOR = 1 fp32 instantiations do): log posterior against the float64 oracle, chains against
    the fp32 kernel's, and the event form of the retained phase bit-identical to the fused split-precision kernel."""
    import torch
    from bayesgm_amd import _lib
    from oracle import identifiable as OI
    from tests.test_gpu_identifiable import _prior
    rs = np.random.RandomState(3)
    z_dims, p, n, k = [1, 1, 1, 7], 200, 700, 7
    q = sum(z_dims)
    m = _model(5, z_dims, p, False)
    x, y, v = _data(n, p, 6, False)
    z = rs.randn(n, q).astype(np.float32)
    seg = rs.randint(0, k, n)
    pn = _prior(rs, k, q)
    tab = OI.prior_table(pn, q)
    mu, s2, _ = OI.prior_params([(W.astype(np.float64), b.astype(np.float64)) for W, b in pn], seg)
    m64 = OC.cast_model(m, np.float64)
    f64 = lambda a: a.astype(np.float64)
    ref = OC.log_posterior(m64, f64(x), f64(y), f64(v), f64(z), prior=(mu, s2))
    std = OC.log_posterior(m64, f64(x), f64(y), f64(v), f64(z))
    eng = _engine(m)
    eng.set_prior(torch.from_numpy(seg.astype(np.int32)).cuda(), torch.from_numpy(tab).cuda())
    lp32 = eng.logpost(x.ravel(), y.ravel(), v, z).cpu().numpy()
    eng.set_precision(mode)
    lp = eng.logpost(x.ravel(), y.ravel(), v, z).cpu().numpy()
    tol = (2e-4 if mode == "bf16x3" else 2e-6) * np.abs(ref) + (2e-2 if mode == "bf16x3" else 5e-4)
    assert np.all(np.abs(lp - ref) <= tol), np.abs(lp - ref).max()
    assert np.abs(ref - std).max() > 0.1 and np.abs(lp - lp32).max() > 0.0       # the prior matters; the split kernels ran
    xs = np.linspace(0, 3, 20)
    outs = {}
    for cache in (False, True):
        eng.set_outcome_cache(cache)
        eng.outcome_cache_stats(reset=True)
        outs[cache] = eng.mh_sample(x, y, v, 20, 25, 0.5, 9, want_draws=True, effect=_lib.EFFECT_ADRF, x_values=xs)
        outs[cache]["stats"] = eng.outcome_cache_stats()
    assert outs[True]["stats"][1] == n * 25                                      # chain-iterations: the event form ran
    for kk in ("draws", "acc_count", "state"):
        assert np.array_equal(outs[True][kk].cpu().numpy(), outs[False][kk].cpu().numpy()), kk
    eng.set_precision("fp32")
    eng.set_outcome_cache(False)
    o32 = eng.mh_sample(x, y, v, 20, 25, 0.5, 9, want_draws=True, effect=_lib.EFFECT_ADRF, x_values=xs)
    same = np.all(np.abs(o32["draws"].cpu().numpy()[-1] - outs[False]["draws"].cpu().numpy()[-1]) <= 1e-4, axis=1).mean()
    assert same >= (0.80 if mode == "bf16x3" else 0.97), same
    eng.set_prior(None, None)
    eng.close()


@pytest.mark.parametrize("mode", ["bf16x3", "f16x3"])
def test_bx3_binary_treatment_in_the_event_form(mode):
    """Split-precision transitions with the retained phase of a binary-treatment predict in its event form (round 6): the chains are
    the fused split-precision kernel's draw for draw (the transitions are the same kernel plus the event append); the events' outcome
    net runs in fp32, so the ITE draws differ from the fused run's by its split-precision outcome-net error only."""
    from bayesgm_amd import _lib
    m = _model(51, [3, 3, 6, 6], 100, True)
    x, y, v = _data(1500, 100, 52, True)
    eng = _engine(m)
    eng.set_precision(mode)
    outs = {}
    for cache in (False, True):
        eng.set_outcome_cache(cache)
        eng.outcome_cache_stats(reset=True)
        outs[cache] = eng.mh_sample(x, y, v, 20, 40, 1.0, 9, want_draws=True, effect=_lib.EFFECT_ITE)
        outs[cache]["stats"] = eng.outcome_cache_stats()
    assert outs[True]["stats"][1] == 1500 * 40                                   # chain-iterations: the event form ran
    for kk in ("draws", "acc_count", "state"):
        assert np.array_equal(outs[True][kk].cpu().numpy(), outs[False][kk].cpu().numpy()), kk
    d = np.abs(outs[True]["ite"].cpu().numpy() - outs[False]["ite"].cpu().numpy()).max()
    print("%s: max |ITE(event form, fp32 outcome net) - ITE(fused split precision)| = %.2e" % (mode, d))
    assert d <= (2e-2 if mode == "bf16x3" else 2e-3)
    eng.set_precision("fp32")
    eng.set_outcome_cache(True)
    eng.close()
