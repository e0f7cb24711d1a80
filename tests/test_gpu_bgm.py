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


# p = 20 / 100 / 97 run the LDS-resident variants; every other width (incl. BASELINE config C4's p = 500 and
# ragged widths that are no multiple of 4 or 16) runs the wide variant that streams the head weights.
@pytest.mark.parametrize("case", [dict(q=10, p=100, n=333, nh=5), dict(q=10, p=20, n=50, nh=5),
                                  dict(q=3, p=20, n=17, nh=3), dict(q=10, p=97, n=1, nh=3),
                                  dict(q=10, p=500, n=300, nh=5), dict(q=10, p=131, n=129, nh=5),
                                  dict(q=4, p=50, n=2100, nh=3), dict(q=10, p=7, n=33, nh=5)])
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


@pytest.mark.parametrize("case", [dict(q=10, p=100, n=150), dict(q=10, p=20, n=64), dict(q=10, p=500, n=150),
                                  dict(q=10, p=61, n=2100)])
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


@pytest.mark.parametrize("p", [100, 500, 37])
def test_predictive_draws_match_oracle_on_same_latents(p):
    import torch
    m = _model(31, 10, p)
    rs = np.random.RandomState(32)
    draws = rs.randn(6, 40, 10).astype(np.float32)
    eng = _engine(m)
    ref = OB.predict_on_posteriors(OB.cast_model(m, np.float64), draws.astype(np.float64), seed=9, burn_in=13)
    # full samples
    _, full = eng.predict_draws(torch.from_numpy(draws).cuda(), 13, 9, want_full=True)
    assert np.abs(full.cpu().numpy() - ref).max() <= 2e-4
    # compact cells for a ragged missing pattern
    miss = rs.rand(40, p) < 0.1
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
        assert np.abs(cells[i, :len(c)] - ref[:, i, c].T).max() <= 2e-4


def _bgm_params(tmp_path, p, q=10):
    return dict(dataset="t", output_dir=str(tmp_path), save_res=False, save_model=False, use_bnn=False, z_dim=q,
                x_dim=p, lr_theta=5e-3, lr_z=5e-3, g_units=[64] * 5, e_units=[64] * 5, dz_units=[64, 32, 8],
                dx_units=[64, 32, 8], kl_weight=5e-5, lr=1e-3, g_d_freq=1, use_z_rec=True, alpha=0.0, gamma=0.0)


def test_bgm_class_predict_matches_oracle_and_reference_shapes(tmp_path):
    """BGM.predict (bgm/base.py:527-663): shared missing pattern -> interval [n, n_miss, 2]; observed cells are
    returned untouched; imputations / intervals agree with the oracle run on the same Philox streams."""
    from bayesgm_amd.models import BGM
    p, n = 20, 48
    m = _model(41, 10, p)
    rs = np.random.RandomState(42)
    x = rs.randn(n, p).astype(np.float32)
    x[:, [3, 17]] = np.nan                       # same pattern for all rows
    model = BGM(_bgm_params(tmp_path, p), random_seed=0)
    model.set_weights(m["g"])
    imp, interval = model.predict(x, alpha=0.1, n_mcmc=60, burn_in=30, step_size=0.05, num_leapfrog_steps=4, seed=5)
    assert imp.shape == (n, p) and interval.shape == (n, 2, 2)
    obs = ~np.isnan(x)
    assert np.array_equal(imp[obs], x[obs]) and not np.isnan(imp).any()
    ref_imp, ref_int = OB.predict(m, x, alpha=0.1, n_mcmc=60, burn_in=30, step_size=0.05, n_leapfrog=4, seed=5)
    d = np.abs(imp[:, [3, 17]] - ref_imp[:, [3, 17]]).max(axis=1)
    # measured: 48 of 48 rows within 1.8e-7; one chain may flip an accept decision that lies within fp32 rounding of its uniform
    assert (d < 1e-4).sum() >= n - 1, d
    assert np.abs(interval - ref_int).max(axis=(1, 2))[d < 1e-4].max() < 2e-3
    assert np.all(interval[..., 0] <= interval[..., 1])
    # return_samples
    smp, _ = model.predict(x, alpha=0.1, return_samples=True, n_mcmc=12, burn_in=10, step_size=0.05,
                           num_leapfrog_steps=4, seed=5)
    assert smp.shape == (12, n, p)


def test_bgm_class_predict_ragged_pattern_and_edge_cases(tmp_path):
    from bayesgm_amd.models import BGM
    p, n = 20, 33
    m = _model(51, 10, p)
    x = _data(n, p, 52, miss=0.3)          # row 0 all missing, row 1 fully observed
    model = BGM(_bgm_params(tmp_path, p), random_seed=0)
    model.set_weights(m["g"])
    imp, interval = model.predict(x, n_mcmc=25, burn_in=15, step_size=0.05, num_leapfrog_steps=3, max_draw_bytes=1 << 14)
    assert isinstance(interval, list) and len(interval) == n
    miss = np.isnan(x)
    for i in range(n):
        assert interval[i].shape == (int(miss[i].sum()), 2)
    assert interval[1].shape == (0, 2) and interval[0].shape == (p, 2)
    assert not np.isnan(imp).any() and np.array_equal(imp[~miss], x[~miss])
    # no missing value at all: nothing to impute, empty interval array (bgm/base.py:631-633)
    full = np.random.RandomState(1).randn(16, p).astype(np.float32)
    imp2, int2 = model.predict(full, n_mcmc=5, burn_in=5)
    assert np.array_equal(imp2, full) and int2.shape == (16, 0, 2)
    with pytest.raises(AssertionError):
        model.predict(full, alpha=0.0)
    # generate / evaluate / log-posterior wrappers
    gen, var = model.generate(nb_samples=50)
    assert gen.shape == (50, p) and var.shape == (50, p) and np.all(var > 0)
    z = np.random.RandomState(2).randn(16, 10).astype(np.float32)
    mse = model.evaluate(full, data_z=z, use_x_sd=False)
    mu, _ = OB.varnet_forward(m["g"], z, training=False)
    assert isinstance(mse, np.float32) and abs(mse - np.mean((full - mu) ** 2)) < 1e-4
    lp = model.get_log_posterior(z, full)
    assert np.allclose(lp, OB.log_posterior(m, z, full), rtol=1e-5, atol=1e-3)
    lp_idx = model.get_log_posterior(z, full, ind_x1=np.array([0, 5, 7]))
    mk = np.zeros_like(full); mk[:, [0, 5, 7]] = 1
    assert np.allclose(lp_idx, OB.log_posterior(m, z, full, mk), rtol=1e-5, atol=1e-3)


@pytest.mark.parametrize("p,B", [(20, 32), (500, 32), (45, 200), (100, 32), (500, 20), (131, 9)])
def test_bgm_fit_steps_match_oracle(tmp_path, p, B):
    """BGM fit step functions (training-mode BatchNorm, per-dimension variance head, fresh-slot Adam on Z)."""
    import torch
    q, n, lr = 10, max(96, B + 40), 2e-3
    m = _model(61, q, p)
    rs = np.random.RandomState(62)
    x = rs.randn(n, p).astype(np.float32)
    z = rs.randn(n, q).astype(np.float32)
    eng = _engine(m)
    xd, zd = torch.from_numpy(x).cuda(), torch.from_numpy(z.copy()).cuda()
    npar = eng.fit_begin(n, B)
    grad = torch.empty(npar, device="cuda")
    loss = torch.zeros(4, device="cuda", dtype=torch.float64)
    m64 = OB.cast_model(m, np.float64)
    st = OB.BgmFitState(m64, z.astype(np.float64), lr, lr)
    x64 = x.astype(np.float64)
    # gradient parity on the first minibatch
    idx_np = rs.choice(n, B, replace=False).astype(np.int32)
    idx = torch.from_numpy(idx_np).cuda()
    eng.fit_theta_grad(xd, zd, idx, grad, loss)
    l_ref, mse_ref, gr, _, _ = OB.g_loss_and_grads(m64, st.data_z[idx_np], x64[idx_np])
    ref = np.concatenate([a.ravel() for a in OB._flat_bgm_grads(gr)])
    got = grad.cpu().numpy()
    got_t = np.concatenate([got[:2 * q], got[4 * q:]])          # moving statistics carry no gradient
    assert np.all(got[2 * q:4 * q] == 0)
    assert np.abs(got_t - ref).max() <= 2e-5 * np.abs(ref).max() + 1e-7, np.abs(got_t - ref).max()
    assert np.isclose(loss.cpu().numpy()[0] / B, l_ref, rtol=2e-5)
    # the gradient call already moved the BN moving statistics once; mirror that, then run full steps
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
    g_tr = eng.get_weights()
    for k in ("gamma", "beta", "mean", "var"):
        assert np.abs(g_tr["bn"][k] - m64["g"]["bn"][k]).max() <= 5e-4, k
    for (W, b), (Wr, br) in zip(g_tr["trunk"], m64["g"]["trunk"]):
        assert np.abs(W - Wr).max() <= 5e-4 and np.abs(b - br).max() <= 5e-4
    for k in ("mean", "var"):
        assert np.abs(g_tr[k][0] - m64["g"][k][0]).max() <= 5e-4
    eng.fit_end()
    # after fit_end the inference blob is rebuilt with the new moving statistics folded in
    lp = eng.logpost(zd, xd).cpu().numpy()
    ref_lp = OB.log_posterior(m64, st.data_z, x64)
    assert np.abs(lp - ref_lp).max() <= 0.1


@pytest.mark.parametrize("p", [20, 500])
def test_bgm_epoch_loop_inside_the_library_equals_the_host_loop(tmp_path, p):
    """BGM.fit(host_loop=False) -- one bgm_bgm_fit_epoch call per epoch -- gives the generator, the latents and the per-epoch losses of
    the per-minibatch calls from Python bit for bit (batch 32: the small-minibatch passes with the head tiles dealt over the waves)."""
    from bayesgm_amd.models import BGM
    rs = np.random.RandomState(11)
    data = rs.randn(700, p).astype(np.float32)
    res = []
    for host_loop in (True, False):
        np.random.seed(5)
        model = BGM(_bgm_params(tmp_path, p), random_seed=3)
        model.fit(data, epochs=2, epochs_per_eval=100, use_egm_init=False, verbose=0, host_loop=host_loop)
        g = model.engine.get_weights()
        flat = np.concatenate([g["bn"][k] for k in ("gamma", "beta", "mean", "var")] + [a.ravel() for W, b in g["trunk"] for a in (W, b)]
                              + [a.ravel() for k in ("mean", "var") for a in g[k]])
        res.append((model.data_z.cpu().numpy().copy(), flat, [dict(h) for h in model.fit_history]))
    (za, wa, ha), (zb, wb, hb) = res
    assert np.array_equal(za, zb)
    assert np.array_equal(wa, wb)
    assert len(ha) == len(hb) == 3
    for a, b in zip(ha, hb):
        assert a == b




def test_bgm_class_fit_reduces_reconstruction_error(tmp_path):
    from bayesgm_amd.models import BGM
    from bayesgm_amd.datasets import simulate_z_hetero
    X, Y = simulate_z_hetero(n=2000, k=3, d=19, seed=42)
    data = np.c_[X, Y].astype(np.float32)
    params = _bgm_params(tmp_path, 20)
    params.update(save_res=True, lr_theta=2e-3, lr_z=2e-3)
    model = BGM(params, random_seed=3)
    with pytest.raises(RuntimeError):
        model.evaluate(data)                          # no encoder before egm_init: loud
    model.fit(data, epochs=8, epochs_per_eval=4, use_egm_init=False, verbose=0)
    assert len(model.history_loss) == 3 and model.history_loss[-1] < model.history_loss[0]
    import os
    assert os.path.exists(os.path.join(model.save_dir, "data_gen_at_4.npz"))
    miss = data[:64].copy()
    miss[:, -1] = np.nan
    imp, interval = model.predict(miss, n_mcmc=30, burn_in=30)
    assert imp.shape == (64, 20) and interval.shape == (64, 1, 2)


def test_wide_variant_agrees_with_resident_variant(monkeypatch):
    """The streamed-head (wide) kernels and the LDS-resident kernels evaluate the same arithmetic per row (the
    compiler may contract the elementwise epilogue differently): log-posteriors / gradients agree to fp32
    rounding and short HMC chains stay together on a width both support."""
    import torch
    m = _model(71, 10, 100)
    x = _data(400, 100, 72)
    z = np.random.RandomState(73).randn(400, 10).astype(np.float32)
    eng = _engine(m)
    lp_r, gr_r = eng.logpost(z, x, want_grad=True)
    out_r = eng.hmc_sample(x, 5, 10, step_size=0.02, n_leapfrog=3, seed=5)
    monkeypatch.setenv("BGM_FORCE_WIDE", "1")
    eng_w = _engine(m)
    lp_w, gr_w = eng_w.logpost(z, x, want_grad=True)
    out_w = eng_w.hmc_sample(x, 5, 10, step_size=0.02, n_leapfrog=3, seed=5)
    assert (lp_r - lp_w).abs().max().item() <= 1e-4 and (gr_r - gr_w).abs().max().item() <= 1e-4
    close = ((out_r["draws"][-1] - out_w["draws"][-1]).abs() <= 1e-3).all(dim=1).float().mean().item()
    assert close >= 0.99, close


def test_bgm_default_fit_with_egm_warm_start(tmp_path):
    """BGM.fit with its defaults (use_egm_init=True): the native EGM warm start (bgm_egm_kernels.h; step parity in
    tests/test_gpu_egm.py) trains g / e, initialises Z = e(X) and hands the generator -- including the BatchNorm
    moving statistics moved by every training-mode call -- to the posterior kernels."""
    from bayesgm_amd.models import BGM
    from bayesgm_amd.datasets import simulate_z_hetero
    X, Y = simulate_z_hetero(n=2000, k=3, d=19, seed=42)
    data = np.c_[X, Y].astype(np.float32)
    params = _bgm_params(tmp_path, 20)
    params.update(save_res=True, lr_theta=2e-3, lr_z=2e-3, gamma=1.0)     # gamma > 0: gradient-penalty path
    model = BGM(params, random_seed=3)
    bn0 = {k: v.copy() for k, v in model.g["bn"].items()}
    model.fit(data, epochs=4, epochs_per_eval=4, egm_n_iter=300, egm_batches_per_eval=150, verbose=0)
    assert model.data_z.shape == (2000, 10) and len(model.history_loss) == 2
    import os
    assert os.path.exists(os.path.join(model.save_dir, "init_data_gen_at_150.npz"))
    # three training-mode generator calls per EGM iteration (g_d_freq = 1) + the evaluation passes moved the statistics
    assert not np.allclose(model.g["bn"]["mean"], bn0["mean"]) and not np.allclose(model.g["bn"]["var"], bn0["var"])
    mse_enc = model.evaluate(data, use_x_sd=False)            # encoder path of evaluate
    assert np.isfinite(mse_enc) and mse_enc < float(np.mean(data ** 2)) * 1.5
    # the warm start reduced the reconstruction error of e -> g on the data compared with the untrained pair
    fresh = BGM(params, random_seed=3)
    fresh.egm_init(data, egm_n_iter=0, egm_batches_per_eval=1000, verbose=0)
    assert mse_enc < fresh.evaluate(data, use_x_sd=False)


def test_bgm_wide_panel_properties():
    """BASELINE config C4's row width (p = 500, wide variant) on a 120 001-row panel, through size-independent
    properties: determinism, invariance to row blocking (Philox keyed by the global row index), agreement of sampled
    rows with the oracle chain, acceptance statistics consistent between the full run and its blocks."""
    import torch
    n, p, q, burn, keep, L = 120_001, 500, 10, 6, 3, 5
    m = _model(81, q, p)
    rs = np.random.RandomState(82)
    x = rs.standard_normal((n, p)).astype(np.float32)
    x[rs.rand(n, p) < 0.1] = np.nan
    eng = _engine(m)
    xd = torch.from_numpy(x).cuda()

    def run(lo, hi):
        n_ = hi - lo
        state = torch.empty((n_, q), device="cuda"); logp = torch.empty(n_, device="cuda"); grad = torch.empty((n_, q), device="cuda")
        step = torch.full((1,), 0.02, device="cuda")
        acc = torch.zeros(burn + keep, device="cuda", dtype=torch.int32)
        draws = torch.empty((keep, n_, q), device="cuda")
        eng.hmc_run(xd[lo:hi], state, logp, grad, step, 0, burn + keep, burn, L, 9, init=True, row_base=lo, acc_count=acc, draws=draws)
        return state, logp, acc, draws
    s1, l1, a1, d1 = run(0, n)
    s2, l2, a2, d2 = run(0, n)
    assert torch.equal(s1, s2) and torch.equal(l1, l2) and torch.equal(a1, a2) and torch.equal(d1, d2)
    lo, hi = 40_016, 40_016 + 33_333
    s3, l3, a3, d3 = run(lo, hi)
    assert torch.equal(s1[lo:hi], s3) and torch.equal(l1[lo:hi], l3) and torch.equal(d1[:, lo:hi], d3)
    s4, l4, a4, d4 = run(0, lo)
    s5, l5, a5, d5 = run(hi, n)
    assert torch.equal(a1, a3 + a4 + a5)                           # per-iteration acceptance counts add up over blocks
    idx = np.sort(rs.choice(n, 24, replace=False))
    obs, clean = OB.obs_mask_of(x[idx])
    ref, fragile = [], np.zeros(len(idx), bool)
    m64 = OB.cast_model(m, np.float64)
    for k, i in enumerate(idx):                       # fixed step size (hmc_run does not adapt): oracle transitions directly
        xk, mk = clean[k:k + 1].astype(np.float64), obs[k:k + 1].astype(np.float64)
        z = OB.hmc_init_state(1, q, 9, int(i)).astype(np.float64)
        lp, gr = OB.log_posterior_and_grad(m64, z, xk, mk)
        for it in range(burn + keep):
            u = OB.R.uniforms(np.array([int(i)]), it, OB.R.TAG_HACC, 9)
            z, lp, gr, lr, _ = OB.hmc_transition(m64, z, xk, mk, 0.02, L, it, 9, int(i), lp, gr)
            fragile[k] |= bool(abs(np.log(u[0]) - lr[0]) < 5e-3)    # accept decision within fp32 rounding of the energy difference
        ref.append(z[0])
    ref = np.stack(ref)
    got = s1.cpu().numpy()[idx]
    # a chain is a deterministic function of its Philox streams except at the accept decisions: every sampled row whose decisions
    # all had a margin matches the float64 oracle to fp32 rounding through 9 transitions x 5 leapfrog steps of p = 500 residuals
    ok = ~fragile
    assert ok.sum() >= 20, fragile
    assert np.abs(got[ok] - ref[ok]).max() <= 5e-4, np.abs(got[ok] - ref[ok]).max(axis=1)


def test_bgm_fit_global_batch_scaling_and_two_rank_run(tmp_path):
    """Data-parallel fit: (i) with batch_global = 2 B every gradient entry is exactly half of the local-batch gradient
    (the loss is a mean over the global batch; the per-rank BatchNorm statistics are unaffected), so the all-reduced
    SUM over ranks is the global-mean gradient; (ii) a two-rank run (both ranks on this GPU, gloo) ends with bit-identical
    parameters on both ranks and a falling reconstruction error."""
    import os, subprocess, sys, torch
    p, q, n, B = 20, 10, 200, 32
    m = _model(91, q, p)
    rs = np.random.RandomState(92)
    x = torch.from_numpy(rs.randn(n, p).astype(np.float32)).cuda()
    z = torch.from_numpy(rs.randn(n, q).astype(np.float32)).cuda()
    idx = torch.from_numpy(rs.choice(n, B, replace=False).astype(np.int32)).cuda()
    eng = _engine(m)
    npar = eng.fit_begin(n, B)
    g1, g2 = torch.empty(npar, device="cuda"), torch.empty(npar, device="cuda")
    eng.fit_theta_grad(x, z, idx, g1)
    eng.fit_set_global_batch(2 * B)
    eng.fit_theta_grad(x, z, idx, g2)
    assert torch.equal(g2 * 2.0, g1)
    eng.fit_end()
    from conftest import run_two_ranks
    r = run_two_ranks("dp_bgm_fit_smoke.py")
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert r.stdout.count('"param_spread": 0.0') == 2, r.stdout[-2000:]


@pytest.mark.gpu
@pytest.mark.parametrize("mode", ["egm", "egm_noseed"])
def test_two_rank_fit_with_replicated_egm_keeps_replicas_identical(mode):
    """The EGM warm start is replicated under torch.distributed: its reparameterisation noise comes from a device generator
    keyed by the rank-shared seed (also with random_seed=None, where rank 0's seed is broadcast), so both ranks enter the
    data-parallel fit with the same generator / encoder and end with bit-identical parameters."""
    from conftest import run_two_ranks
    r = run_two_ranks("dp_bgm_fit_smoke.py", extra_args=(mode,))
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert r.stdout.count('"param_spread": 0.0') == 2, r.stdout[-2000:]


def test_bgm_c4_one_gpu_share_through_the_class(tmp_path):
    """One GPU's share of BASELINE configs[4] (BGM imputation, N = 5e6 over 8 GPUs, p = 500) through BGM.predict itself -- the wide
    variant of the HMC kernel, row-blocked draw buffers, per-cell intervals -- at the full 625 000 x 500 panel with short chains,
    checked through size-independent properties (the oracle covers small panels in the tests above)."""
    from bayesgm_amd.models import BGM
    n, p, q = 625_000, 500, 10
    m = _model(91, q, p)
    rs = np.random.RandomState(92)
    x = rs.standard_normal((n, p)).astype(np.float32)
    x[:, rs.choice(p, 50, replace=False)] = np.nan            # 10 % of the cells, one pattern (C4's 3.1e7 missing cells)
    model = BGM(_bgm_params(tmp_path, p), random_seed=0)
    model.set_weights(m["g"])
    kw = dict(alpha=0.1, n_mcmc=12, burn_in=12, step_size=0.02, num_leapfrog_steps=5, seed=7)
    imp, interval = model.predict(x, **kw)
    miss = np.isnan(x)
    assert imp.shape == (n, p) and interval.shape == (n, 50, 2)
    assert np.array_equal(imp[~miss], x[~miss]) and np.isfinite(imp).all() and np.isfinite(interval).all()
    assert np.all(interval[..., 0] <= interval[..., 1])
    cells = imp[miss].reshape(n, 50)
    assert np.all((cells >= interval[..., 0] - 1e-4) & (cells <= interval[..., 1] + 1e-4))     # a posterior mean lies inside its interval
    t = model.last_predict_timing
    assert t and all(v >= 0.0 for v in t.values())
    # deterministic: the same call reproduces every imputed cell and interval bit for bit
    imp2, int2 = model.predict(x, **kw)
    assert np.array_equal(imp2, imp) and np.array_equal(int2, interval)
