"""GPU parity tests of the BGM posterior path (masked log-posterior + gradient, HMC with step-size
adaptation, posterior-predictive draws) vs the NumPy oracle.

Tolerances: log-posterior <= 2e-6*|ref| + 2e-4 and gradient <= 2e-5*max|ref| + 2e-5 vs the float64 oracle;
HMC chains share the Philox stream with the oracle, so after a short run >= 97 % of the rows must agree
to 2e-3 (leapfrog integration amplifies fp32 rounding more than a random-walk proposal does) and the
adapted step size must follow the same multiply/divide schedule.
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from oracle import bgm as OB  # noqa: E402


def _model(seed, q, p, n_hidden=5):
    m = OB.init_model(seed, q, p, g_units=(64,) * n_hidden)
    rs = np.random.RandomState(seed + 7)
    g = m["g"]
    g["bn"].update(gamma=(1 + 0.1 * rs.randn(q)).astype(np.float32), beta=(0.1 * rs.randn(q)).astype(np.float32),
                   mean=(0.2 * rs.randn(q)).astype(np.float32), var=(0.5 + rs.rand(q)).astype(np.float32))
    g["trunk"] = [(W, (0.1 * rs.randn(*b.shape)).astype(np.float32)) for W, b in g["trunk"]]
    g["mean"] = (g["mean"][0], (0.1 * rs.randn(p)).astype(np.float32))
    g["var"] = (g["var"][0], (0.1 * rs.randn(p)).astype(np.float32))
    return m


def _data(n, p, seed, miss=0.2):
    rs = np.random.RandomState(seed)
    x = rs.randn(n, p).astype(np.float32)
    x[rs.rand(n, p) < miss] = np.nan
    x[0, :] = np.nan          # a row with nothing observed (prior only)
    if n > 1:
        x[1, :] = rs.randn(p)  # a fully observed row
    return x


def _engine(m):
    from bayesgm_amd.engine import BgmEngine
    eng = BgmEngine(m["x_dim"], m["z_dim"], g_units=[W.shape[1] for W, _ in m["g"]["trunk"]])
    eng.set_weights(m["g"])
    return eng


@pytest.mark.parametrize("case", [dict(q=10, p=100, n=333, nh=5), dict(q=10, p=20, n=50, nh=5),
                                  dict(q=3, p=20, n=17, nh=3), dict(q=10, p=97, n=1, nh=3)])
def test_bgm_logpost_and_gradient_match_oracle(case):
    m = _model(1, case["q"], case["p"], case["nh"])
    x = _data(case["n"], case["p"], 2)
    z = np.random.RandomState(3).randn(case["n"], case["q"]).astype(np.float32)
    eng = _engine(m)
    lp, gr = eng.logpost(z, x, want_grad=True)
    lp0 = eng.logpost(z, x)
    obs, clean = OB.obs_mask_of(x)
    m64 = OB.cast_model(m, np.float64)
    ref_lp, ref_gr = OB.log_posterior_and_grad(m64, z.astype(np.float64), clean.astype(np.float64), obs.astype(np.float64))
    lp, gr, lp0 = lp.cpu().numpy(), gr.cpu().numpy(), lp0.cpu().numpy()
    assert np.array_equal(lp, lp0)
    assert np.all(np.abs(lp - ref_lp) <= 2e-6 * np.abs(ref_lp) + 2e-4), np.abs(lp - ref_lp).max()
    assert np.abs(gr - ref_gr).max() <= 2e-5 * np.abs(ref_gr).max() + 2e-5, np.abs(gr - ref_gr).max()
    # row 0 has no observed feature: posterior = prior
    assert abs(lp[0] + 0.5 * (z[0] ** 2).sum()) < 1e-5 and np.allclose(gr[0], -z[0], atol=1e-6)


@pytest.mark.parametrize("case", [dict(q=10, p=100, n=150), dict(q=10, p=20, n=64)])
def test_hmc_chain_and_step_adaptation_match_oracle(case):
    import torch
    m = _model(11, case["q"], case["p"])
    x = _data(case["n"], case["p"], 12)
    burn, keep, L, seed = 20, 10, 4, 77
    eng = _engine(m)
    out = eng.hmc_sample(x, keep, burn, step_size=0.02, n_leapfrog=L, seed=seed)
    obs, clean = OB.obs_mask_of(x)
    ref, info = OB.hmc_sampler(m, clean, obs.astype(np.float32), keep, burn, 0.02, L, seed, return_info=True)
    draws = out["draws"].cpu().numpy()
    assert draws.shape == ref.shape
    ok = np.all(np.abs(draws[-1] - ref[-1]) <= 2e-3, axis=1)
    assert ok.mean() >= 0.97, ok.mean()
    assert abs(float(out["step"].item()) / info["step"] - 1) < 1e-5      # same *1.01 / /1.01 schedule
    acc = out["acc_count"].cpu().numpy()[burn:].sum() / (keep * case["n"])
    assert abs(acc - info["accept_rate"]) < 0.03 and acc > 0.5
    # determinism
    out2 = eng.hmc_sample(x, keep, burn, step_size=0.02, n_leapfrog=L, seed=seed)
    assert torch.equal(out2["draws"], out["draws"])


def test_hmc_samples_the_prior_when_nothing_is_observed():
    """Known-answer property: all-missing rows -> posterior N(0, I)."""
    m = _model(21, 10, 20)
    x = np.full((512, 20), np.nan, np.float32)
    eng = _engine(m)
    out = eng.hmc_sample(x, 200, 100, step_size=0.1, n_leapfrog=5, seed=5)
    d = out["draws"].cpu().numpy().reshape(-1, 10)
    assert np.abs(d.mean(0)).max() < 0.03 and np.abs(d.var(0) - 1).max() < 0.06
    assert float(out["step"].item()) > 0.1


def test_predictive_draws_match_oracle_on_same_latents():
    import torch
    m = _model(31, 10, 100)
    rs = np.random.RandomState(32)
    draws = rs.randn(6, 40, 10).astype(np.float32)
    eng = _engine(m)
    ref = OB.predict_on_posteriors(OB.cast_model(m, np.float64), draws.astype(np.float64), seed=9, burn_in=13)
    # full samples
    _, full = eng.predict_draws(torch.from_numpy(draws).cuda(), 13, 9, want_full=True)
    assert np.abs(full.cpu().numpy() - ref).max() <= 2e-4
    # compact cells for a ragged missing pattern
    miss = rs.rand(40, 100) < 0.1
    slot = np.full((40, 100), -1, np.int32)
    k = 0
    for i in range(40):
        c = np.where(miss[i])[0]
        slot[i, c] = np.arange(len(c))
        k = max(k, len(c))
    cells, _ = eng.predict_draws(torch.from_numpy(draws).cuda(), 13, 9, slot=torch.from_numpy(slot).cuda(), k_slots=k)
    cells = cells.cpu().numpy().reshape(40, k, 6)
    for i in range(40):
        c = np.where(miss[i])[0]
        assert np.abs(cells[i, :len(c)] - ref[:, i, c].T).max() <= 2e-4
