"""BGM with the Bayesian generator (use_bnn=True): kernels against oracle/bgm_bnn.py through the C ABI."""
import numpy as np
import pytest
import torch

from oracle import bgm_bnn as OV
from oracle import bnn as OB

pytestmark = pytest.mark.gpu


def _net(q, units, p, seed=0):
    rs = np.random.RandomState(seed)
    net = OV.init_vnet(rs, q, list(units), p)
    net["gamma"] = (1.0 + 0.2 * rs.standard_normal(q)).astype(np.float32)
    net["beta"] = (0.1 * rs.standard_normal(q)).astype(np.float32)
    net["mean_mv"] = (0.2 * rs.standard_normal(q)).astype(np.float32)
    net["var_mv"] = (0.7 + 0.5 * rs.uniform(size=q)).astype(np.float32)
    return net


def _engine(net, q, units, p, **kw):          # kw: kl_weight, max_batch, hmc_frozen_noise
    from bayesgm_amd.bvn_engine import BvnEngine
    eng = BvnEngine(p, q, g_units=units, **kw)
    eng.begin(net)
    return eng


def _rel(a, b):
    return float(np.abs(np.asarray(a, np.float64) - np.asarray(b, np.float64)).max() / (np.abs(b).max() + 1e-30))


def _flat(parts):
    return np.concatenate([np.asarray(a, np.float64).ravel() for a in parts])


@pytest.mark.parametrize("q,units,p,B", [(10, (64,) * 5, 20, 32), (5, (24, 40), 37, 19), (3, (16,), 9, 64),
                                         (10, (64,) * 5, 20, 160), (4, (32, 32), 12, 200),       # minibatches beyond 64 rows (any batch_size, bgm/base.py:343)
                                         (10, (64,) * 5, 20, 300)])                              # ... and beyond 256 (params['max_batch'])
def test_theta_step_gradient_matches_oracle(q, units, p, B):
    net = _net(q, units, p)
    rs = np.random.RandomState(1)
    N = max(200, B + 20)
    z = rs.standard_normal((N, q)).astype(np.float32)
    x = rs.standard_normal((N, p)).astype(np.float32)
    idx = rs.choice(N, B, replace=False).astype(np.int32)
    eng = _engine(net, q, units, p, kl_weight=0.01, max_batch=max(64, B))
    dev = eng.device
    T = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    seed, stream = (3 << 32) | 17, 6
    out = torch.zeros(2, device=dev)
    eng.theta_step(T(x), T(z), T(idx), 1e-3, seed, stream, apply=False, out=out)
    got = eng.read(1).astype(np.float64)
    n64 = OV.cast_vnet(net, np.float64)
    noise = OV.draw(n64, B, seed, stream, dtype=np.float64)
    loss, mse, g, _, c = OV.loss_and_grads(n64, z[idx].astype(np.float64), x[idx].astype(np.float64), noise)
    klv, gk = OV.vkl(n64)
    ref = _flat(OV.flat_grads(OB.add_grads(g, gk, 0.01)))
    assert _rel(got, ref) < 2e-4
    o = out.cpu().numpy()
    assert abs(o[0] - (loss + 0.01 * klv)) < 1e-4 * abs(loss + 0.01 * klv) + 1e-4
    assert abs(o[1] - mse) < 1e-4 * mse
    # the training-mode call moved the BatchNorm statistics
    th = eng.read(0)
    OV.move_stats(n64, c)
    np.testing.assert_allclose(th[2 * q:3 * q], n64["mean_mv"], rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(th[3 * q:4 * q], n64["var_mv"], rtol=1e-5, atol=1e-6)
    eng.close()


def test_fit_steps_follow_oracle():
    q, units, p, B, N = 4, (32, 32), 12, 32, 160
    net = _net(q, units, p, seed=2)
    rs = np.random.RandomState(3)
    z0 = rs.standard_normal((N, q)).astype(np.float32)
    x = rs.standard_normal((N, p)).astype(np.float32)
    eng = _engine(net, q, units, p, kl_weight=5e-5)
    dev = eng.device
    xd, zd = torch.from_numpy(x).to(dev), torch.from_numpy(z0.copy()).to(dev)
    seed = 91
    onet = OV.cast_vnet(net, np.float64)
    st = OV.FitState(onet, z0.astype(np.float64), 5e-3, 5e-3, 5e-5, seed)
    out_t, out_z = torch.zeros(2, device=dev), torch.zeros(1, device=dev)
    for t in range(4):
        idx = rs.choice(N, B, replace=False).astype(np.int32)
        idd = torch.from_numpy(idx).to(dev)
        eng.theta_step(xd, zd, idd, 5e-3, seed, 2 * t, out=out_t)
        eng.z_step(xd, zd, idd, 5e-3, seed, 2 * t + 1, out=out_z)
        lx, mse, lp = st.step(x.astype(np.float64), idx)
        assert abs(out_t.cpu().numpy()[0] - lx) < 2e-3 * abs(lx) + 1e-3
        assert abs(out_z.cpu().numpy()[0] - lp) < 2e-3 * abs(lp) + 1e-3
    got = eng.read(0).astype(np.float64)
    ref = _flat(OV.flat_params(onet))
    assert np.abs(got - ref).max() < 2e-3          # Adam steps of 5e-3: sign-like updates amplify rounding
    assert np.abs(zd.cpu().numpy() - st.data_z).max() < 2e-3
    eng.close()


@pytest.mark.parametrize("n,row_base", [(150, 0), (64, 1000)])
def test_logpost_and_gradient_match_oracle(n, row_base):
    q, units, p = 10, (64,) * 5, 20
    net = _net(q, units, p, seed=4)
    rs = np.random.RandomState(5)
    z = rs.standard_normal((n, q)).astype(np.float32)
    x = rs.standard_normal((n, p)).astype(np.float32)
    x[rs.uniform(size=x.shape) < 0.3] = np.nan
    x[3] = np.nan                                     # a row without observed cells
    eng = _engine(net, q, units, p)
    seed, stream = 1234567, 9
    lp, gr = eng.logpost(z, x, seed, stream, row_base=row_base, want_grad=True)
    n64 = OV.cast_vnet(net, np.float64)
    mask = (~np.isnan(x)).astype(np.float64)
    xc = np.where(np.isnan(x), 0.0, x).astype(np.float64)
    ref_lp, ref_gr = OV.log_posterior_and_grad(n64, z.astype(np.float64), xc, mask, OV.draw(n64, n, seed, stream, row_base, np.float64))
    assert _rel(lp.cpu().numpy(), ref_lp) < 1e-5
    assert _rel(gr.cpu().numpy(), ref_gr) < 1e-4
    lp_only = eng.logpost(z, x, seed, stream, row_base=row_base)
    assert torch.equal(lp_only, lp)
    eng.close()


@pytest.mark.parametrize("frozen", [False, True])
def test_hmc_follows_oracle_over_a_few_transitions(frozen):
    q, units, p, n = 3, (16, 16), 8, 100
    net = _net(q, units, p, seed=6)
    rs = np.random.RandomState(7)
    x = rs.standard_normal((n, p)).astype(np.float32)
    x[rs.uniform(size=x.shape) < 0.25] = np.nan
    eng = _engine(net, q, units, p, hmc_frozen_noise=frozen)
    seed = 42
    out = eng.hmc_sample(x, n_mcmc=3, burn_in=5, step_size=0.05, n_leapfrog=4, seed=seed, row_base=7)
    mask = (~np.isnan(x)).astype(np.float32)
    xc = np.where(np.isnan(x), 0.0, x).astype(np.float32)
    ref, info = OV.hmc_sampler(OV.cast_vnet(net, np.float64), xc.astype(np.float64), mask.astype(np.float64), 3, 5, 0.05, 4, seed, row0=7,
                               return_info=True, frozen=frozen)
    got = out["draws"].cpu().numpy()
    assert got.shape == ref.shape
    assert abs(float(out["step"].item()) - info["step"]) < 1e-6
    # accept/reject decisions near the threshold may flip in float32: almost all chains must agree closely
    close = np.abs(got - ref).max(axis=(0, 2)) < 1e-3
    assert close.mean() > 0.95
    eng.close()


def test_hmc_segment_split_over_several_launches_is_identical(monkeypatch):
    """A segment of transitions is cut into launches by the size of the perturbation buffer; the cut must not change a bit."""
    q, units, p, n = 3, (16, 16), 8, 90
    net = _net(q, units, p, seed=6)
    x = np.random.RandomState(7).standard_normal((n, p)).astype(np.float32)
    x[np.random.RandomState(8).uniform(size=x.shape) < 0.25] = np.nan
    outs = []
    for budget in (None, "1"):                      # default: one launch; 1 byte: one transition per launch
        if budget is None:
            monkeypatch.delenv("BGM_BVN_NOISE_BYTES", raising=False)
        else:
            monkeypatch.setenv("BGM_BVN_NOISE_BYTES", budget)
        eng = _engine(net, q, units, p)
        dev = eng.device
        xd = torch.from_numpy(x).to(dev)
        state, logp, grad = torch.empty((n, q), device=dev), torch.empty(n, device=dev), torch.empty((n, q), device=dev)
        step = torch.full((1,), 0.05, device=dev)
        draws = torch.empty((5, n, q), device=dev)
        acc = torch.zeros(7, device=dev, dtype=torch.int32)
        eng.hmc_run(xd, state, logp, grad, step, 0, 7, 2, 4, 42, init=True, row_base=3, acc_count=acc, draws=draws)
        outs.append((draws.cpu().numpy(), state.cpu().numpy(), logp.cpu().numpy(), acc.cpu().numpy()))
        eng.close()
    for a, b in zip(outs[0], outs[1]):
        np.testing.assert_array_equal(a, b)


def test_hmc_all_missing_rows_sample_the_prior():
    q, units, p, n = 2, (8,), 5, 4096
    net = _net(q, units, p, seed=8)
    x = np.full((n, p), np.nan, np.float32)
    eng = _engine(net, q, units, p)
    out = eng.hmc_sample(x, n_mcmc=20, burn_in=40, step_size=0.3, n_leapfrog=5, seed=3)
    d = out["draws"].cpu().numpy()
    assert abs(d.mean()) < 0.02 and abs(d.var() - 1.0) < 0.05
    acc = out["acc_count"].cpu().numpy()[40:].sum() / (20.0 * n)
    assert acc > 0.6
    eng.close()


def test_decode_matches_oracle_predict_on_posteriors():
    q, units, p, n, nd = 6, (32, 16), 21, 37, 5
    net = _net(q, units, p, seed=9)
    rs = np.random.RandomState(10)
    post = rs.standard_normal((nd, n, q)).astype(np.float32)
    eng = _engine(net, q, units, p)
    seed, block, burn = 77, 2, 11
    from bayesgm_amd.bvn_engine import STREAM_PREDICT
    slot = -np.ones((n, p), np.int32)
    slot[:, 4], slot[:, 9] = 0, 1
    sl = torch.from_numpy(slot).to(eng.device)
    cells, full, var = eng.decode(post, seed, STREAM_PREDICT + block, burn_in=burn, row_base=300, slot=sl, k_slots=2, want_full=True,
                                  want_var=True)
    ref = OV.predict_on_posteriors(OV.cast_vnet(net, np.float64), post.astype(np.float64), seed, block=block, row0=300, burn_in=burn)
    assert _rel(full.cpu().numpy(), ref) < 2e-5
    c = cells.cpu().numpy()
    np.testing.assert_array_equal(c[:, 0, :], full.cpu().numpy()[:, :, 4].T)
    np.testing.assert_array_equal(c[:, 1, :], full.cpu().numpy()[:, :, 9].T)
    _, mean_only = eng.decode(post, seed, STREAM_PREDICT + block, burn_in=burn, row_base=300, want_full=True, add_noise=False)
    m_ref, s2_ref = OV.decode(OV.cast_vnet(net, np.float64), post.reshape(nd * n, q).astype(np.float64), seed, STREAM_PREDICT + block)
    assert _rel(mean_only.cpu().numpy().reshape(nd * n, p), m_ref) < 2e-5
    assert _rel(var.cpu().numpy().reshape(nd * n, p), s2_ref) < 2e-5
    eng.close()


# ------------------------------------------------------------------------------------------------ EGM warm start
def _egm_setup(p, q, B, units=(64,) * 5, gamma=0.0, alpha=0.0, seed=0, n=128):
    from oracle import egm as OE
    from oracle import nets as N
    net = _net(q, units, p, seed=seed)
    rs = np.random.RandomState(seed + 100)
    e = [(W, (0.1 * rs.randn(*b.shape)).astype(np.float32)) for W, b in N.init_mlp(rs, [p] + [64] * 5 + [q])]
    ds = []
    for in_dim in (q, p):
        d = OE.init_disc(rs, in_dim, [64, 32, 8])
        d["b"] = [(0.1 * rs.randn(*b.shape)).astype(np.float32) for b in d["b"]]
        d["gamma"] = [(1 + 0.2 * rs.randn(*b.shape)).astype(np.float32) for b in d["gamma"]]
        d["beta"] = [(0.1 * rs.randn(*b.shape)).astype(np.float32) for b in d["beta"]]
        ds.append(d)
    eng = _engine(net, q, units, p)
    eng.egm_begin(B, [64] * 5, [64, 32, 8], [64, 32, 8], 1e-3, gamma, alpha, e, ds[0], ds[1])
    x = rs.randn(n, p).astype(np.float32)
    c64 = (OV.cast_vnet(net, np.float64), N.cast_net(e, np.float64), OE.cast_disc(ds[0], np.float64), OE.cast_disc(ds[1], np.float64))
    return eng, c64, x, rs


@pytest.mark.parametrize("case", [dict(p=20, B=32, gamma=0.0, alpha=0.0), dict(p=37, B=19, gamma=0.5, alpha=0.2)])
def test_egm_step_gradients_match_oracle(case):
    from oracle import egm as OE
    p, B, q = case["p"], case["B"], 10
    eng, (g64, e64, dz64, dx64), x, rs = _egm_setup(p, q, B, gamma=case["gamma"], alpha=case["alpha"])
    z = rs.randn(B, q).astype(np.float32)
    xb = x[rs.choice(len(x), B, replace=False)]
    n1, n2 = rs.randn(B, p).astype(np.float32), rs.randn(B, p).astype(np.float32)
    d = lambda a: torch.from_numpy(a).cuda()
    seed = 555
    out_d = torch.zeros(3, device="cuda")
    eng.egm_disc_step(d(z), d(xb), d(n1), 0.31, 0.64, seed, 4, apply=False, out=out_d)
    losses, gr, caches = OV.egm_disc_step_grads(g64, e64, dz64, dx64, z.astype(np.float64), xb.astype(np.float64), n1.astype(np.float64),
                                                0.31, 0.64, case["gamma"], OV.draw(g64, B, seed, 4, dtype=np.float64))
    ref = np.concatenate([a.ravel() for a in OE.disc_param_list(gr["dz"]) + OE.disc_param_list(gr["dx"])])
    got = eng.egm_read(3)
    assert np.all(np.abs(out_d.cpu().numpy() - losses) <= 3e-5 * np.abs(losses) + 1e-6), (out_d.cpu().numpy(), losses)
    n_dz = sum(a.size for a in OE.disc_param_list(gr["dz"]))
    for lo, hi in ((0, n_dz), (n_dz, ref.size)):
        assert _rel(got[lo:hi], ref[lo:hi]) <= 1e-4
    OV.move_stats(g64, caches[0])
    th = eng.egm_read(0)
    assert np.abs(th[2 * q:3 * q] - g64["mean_mv"]).max() <= 1e-6 and np.abs(th[3 * q:4 * q] - g64["var_mv"]).max() <= 1e-6
    out_g = torch.zeros(6, device="cuda")
    eng.egm_gen_step(d(z), d(xb), d(n1), d(n2), seed, 6, apply=False, out=out_g)
    losses, gr, caches = OV.egm_gen_step_grads(g64, e64, dz64, dx64, z.astype(np.float64), xb.astype(np.float64), n1.astype(np.float64),
                                               n2.astype(np.float64), case["alpha"], OV.draw(g64, B, seed, 6, dtype=np.float64),
                                               OV.draw(g64, B, seed, 7, dtype=np.float64))
    got = eng.egm_read(2)
    assert np.all(np.abs(out_g.cpu().numpy() - losses) <= 3e-5 * np.abs(losses) + 1e-6), (out_g.cpu().numpy(), losses)
    ref_g = _flat(OV.flat_grads(gr["g"]))
    ref_e = np.concatenate([a.ravel() for Wb in gr["e"] for a in Wb])
    assert _rel(got[:ref_g.size], ref_g) <= 1e-4
    assert _rel(got[ref_g.size:], ref_e) <= 1e-4
    eng.egm_end()
    eng.close()


def test_egm_alternating_adam_steps_track_oracle_and_sync():
    from oracle import egm as OE
    from oracle import nets as N
    p, q, B = 20, 10, 32
    eng, (g64, e64, dz64, dx64), x, rs = _egm_setup(p, q, B, gamma=0.5, alpha=0.2, seed=3)
    seed = 99
    st = OV.EgmState(g64, e64, dz64, dx64, dict(lr=1e-3, gamma=0.5, alpha=0.2), seed)
    d = lambda a: torch.from_numpy(a).cuda()
    s = 0
    for it in range(4):
        z = rs.randn(B, q).astype(np.float32); xb = x[rs.choice(len(x), B, replace=False)]
        n1 = rs.randn(B, p).astype(np.float32); ez, ex = float(rs.rand()), float(rs.rand())
        eng.egm_disc_step(d(z), d(xb), d(n1), ez, ex, seed, 2 * s)
        st.disc_step(z.astype(np.float64), xb.astype(np.float64), n1.astype(np.float64), ez, ex)
        s += 1
        z = rs.randn(B, q).astype(np.float32); xb = x[rs.choice(len(x), B, replace=False)]
        n1, n2 = rs.randn(B, p).astype(np.float32), rs.randn(B, p).astype(np.float32)
        eng.egm_gen_step(d(z), d(xb), d(n1), d(n2), seed, 2 * s)
        st.gen_step(z.astype(np.float64), xb.astype(np.float64), n1.astype(np.float64), n2.astype(np.float64))
        s += 1
    got_g = eng.egm_read(0)
    ref_g = np.concatenate([_flat(OV.flat_params(st.g))] + [a.ravel() for Wb in st.e for a in Wb])
    assert np.abs(got_g - ref_g).max() <= 1e-4, np.abs(got_g - ref_g).max()
    z_enc = eng.egm_encode(x).cpu().numpy()
    assert np.abs(z_enc - N.mlp_forward(st.e, x.astype(np.float64))).max() <= 1e-4
    eng.egm_end()
    th = eng.read(0)
    assert np.abs(th - _flat(OV.flat_params(st.g))).max() <= 1e-4
    eng.close()


# ------------------------------------------------------------------------------------------------ model class
def _params(tmp_path, p, q=4, **kw):
    d = dict(dataset="t", output_dir=str(tmp_path), save_res=False, save_model=False, use_bnn=True, z_dim=q, x_dim=p,
             g_units=[32, 32], e_units=[32, 32], dz_units=[16, 8], dx_units=[16, 8], lr=1e-3, lr_theta=5e-3, lr_z=5e-3,
             g_d_freq=1, kl_weight=5e-5, gamma=1.0, alpha=0.01)
    d.update(kw)
    return d


def _linear_panel(n, p, q, seed=0):
    rs = np.random.RandomState(seed)
    z = rs.standard_normal((n, q)).astype(np.float32)
    return (z @ rs.standard_normal((q, p)) + 0.1 * rs.standard_normal((n, p))).astype(np.float32)


def test_model_fit_evaluate_generate_predict(tmp_path):
    from bayesgm_amd.models import BGM
    n, p, q = 640, 12, 4
    data = _linear_panel(n, p, q)
    model = BGM(_params(tmp_path, p, q, bnn_mcmc_noise="frozen"), random_seed=7)
    assert type(model).__name__ == "BGMBayes"
    mse0 = float(model.evaluate(data, data_z=np.zeros((n, q), np.float32), use_x_sd=False))
    model.fit(data, batch_size=32, epochs=40, epochs_per_eval=10, use_egm_init=True, egm_n_iter=60, egm_batches_per_eval=30, verbose=1)
    assert model.history_loss[-1] < 0.75 * mse0 and model.history_loss[-1] < model.history_loss[0]
    assert float(model.evaluate(data, use_x_sd=False)) < 10 * model.history_loss[-1] + 1.0      # encoder of the warm start
    gen, var = model.generate(nb_samples=500)
    assert gen.shape == (500, p) and np.isfinite(gen).all() and (var > 0).all()
    miss = data[:150].copy()
    rs = np.random.RandomState(5)
    miss[rs.uniform(size=miss.shape) < 0.2] = np.nan
    miss[0, :] = data[0, :]
    miss[0, 3] = np.nan
    imputed, intervals = model.predict(miss, alpha=0.1, bs=64, n_mcmc=100, burn_in=300, step_size=0.05, num_leapfrog_steps=5, seed=11)
    assert imputed.shape == miss.shape and np.isfinite(imputed).all()
    obs = ~np.isnan(miss)
    np.testing.assert_array_equal(imputed[obs], miss[obs])
    assert len(intervals) == 150 and intervals[0].shape == (1, 2)
    truth, est = data[:150][~obs], imputed[~obs]
    print('imputation mse', np.mean((truth - est) ** 2), 'baseline', np.mean((truth - truth.mean()) ** 2))
    assert np.mean((truth - est) ** 2) < np.mean((truth - truth.mean()) ** 2)          # better than the column-free mean
    cover = np.mean([(iv[:, 0] <= data[i][np.isnan(miss[i])]).mean() for i, iv in enumerate(intervals) if len(iv)])
    assert cover > 0.5
    samples, _ = model.predict(miss[:70], alpha=0.1, return_samples=True, bs=64, n_mcmc=10, burn_in=10, step_size=0.02,
                               num_leapfrog_steps=3, seed=11)
    assert samples.shape == (10, 70, p)


def test_model_predict_is_block_consistent_with_the_oracle(tmp_path):
    """One predictive call per bs-block with signs keyed inside the block: compare the engine's decode of a block part
    (sign_off > 0) with oracle.predict_on_posteriors."""
    q, units, p, n, nd = 4, (16, 16), 9, 23, 4
    net = _net(q, units, p, seed=12)
    rs = np.random.RandomState(13)
    post = rs.standard_normal((nd, n, q)).astype(np.float32)
    eng = _engine(net, q, units, p)
    from bayesgm_amd.bvn_engine import STREAM_PREDICT
    _, full = eng.decode(post, 5, STREAM_PREDICT + 3, burn_in=2, row_base=340, want_full=True, sign_stride=100, sign_off=40)
    ref = OV.predict_on_posteriors(OV.cast_vnet(net, np.float64), post.astype(np.float64), 5, block=3, row0=340, burn_in=2, bs=100, off=40)
    assert _rel(full.cpu().numpy(), ref) < 2e-5
    eng.close()


def test_mcmc_noise_default_is_one_weight_draw_and_the_as_written_mode_warns(tmp_path):
    """params['bnn_mcmc_noise']: the default samples each HMC run on one weight draw (a deterministic target); 'fresh' -- the
    reference as executed, chains freeze -- is an explicit choice and says so (models/bgm_bnn.py, DESIGN_HISTORY.md section 7b)."""
    import warnings
    from bayesgm_amd.models import BGM
    p, q = 6, 2
    x = np.random.RandomState(0).standard_normal((20, p)).astype(np.float32)
    x[:, 1] = np.nan
    model = BGM(_params(tmp_path, p, q), random_seed=3)
    assert model._mcmc_noise == "frozen" and model.engine.cfg.hmc_frozen_noise == 1
    with warnings.catch_warnings():
        warnings.simplefilter("error")
        model.predict(x, n_mcmc=3, burn_in=3, num_leapfrog_steps=2)
    model = BGM(_params(tmp_path, p, q, bnn_mcmc_noise="fresh"), random_seed=3)
    assert model._mcmc_noise == "fresh" and model.engine.cfg.hmc_frozen_noise == 0
    with pytest.warns(UserWarning, match="bnn_mcmc_noise"):
        model.predict(x, n_mcmc=3, burn_in=3, num_leapfrog_steps=2)
    with pytest.raises(ValueError):
        BGM(_params(tmp_path, p, q, bnn_mcmc_noise="other"), random_seed=3)


def test_model_checkpoint_roundtrip(tmp_path):
    from bayesgm_amd.models import BGM
    p, q = 7, 3
    model = BGM(_params(tmp_path, p, q, save_model=True), random_seed=1)
    path = model.save_checkpoint(0)
    other = BGM(_params(tmp_path, p, q), random_seed=2)
    other.load_checkpoint(path)
    z = np.random.RandomState(0).standard_normal((10, q)).astype(np.float32)
    a, _ = model._decode(z, False, seed=3)
    b, _ = other._decode(z, False, seed=3)
    np.testing.assert_array_equal(a, b)


def test_two_rank_fit_and_predict_agree(tmp_path):
    """Two ranks on one GPU over gloo (scripts/dp_bgm_bnn_smoke.py; shards differ by a row, bs-blocks split over the ranks):
    bit-identical generators, imputations and intervals on both ranks."""
    from conftest import run_two_ranks
    r = run_two_ranks("dp_bgm_bnn_smoke.py")
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert r.stdout.count('"param_spread": 0.0') == 2, r.stdout[-2000:]


def test_predict_does_not_depend_on_row_chunking(tmp_path):
    from bayesgm_amd.models import BGM
    p, q, n = 9, 3, 230
    model = BGM(_params(tmp_path, p, q, bnn_mcmc_noise="frozen"), random_seed=4)
    data = _linear_panel(n, p, q, seed=3)
    data[np.random.RandomState(1).uniform(size=data.shape) < 0.2] = np.nan
    a, ia = model.predict(data, bs=64, n_mcmc=8, burn_in=8, step_size=0.05, num_leapfrog_steps=3, seed=5)
    b, ib = model.predict(data, bs=64, n_mcmc=8, burn_in=8, step_size=0.05, num_leapfrog_steps=3, seed=5, max_draw_bytes=1)   # one bs-block per chunk
    np.testing.assert_array_equal(a, b)
    for u, v in zip(ia, ib):
        np.testing.assert_array_equal(u, v)


def test_index_and_nan_forms_of_missingness_agree(tmp_path):
    """The reference passes observed features as index lists + obs_mask (bgm/base.py:578-592, 689-700); the build marks missing
    cells with NaN.  Both forms must give the same log posterior and the same chains."""
    from bayesgm_amd.models import BGM
    p, q, n = 9, 3, 40
    model = BGM(_params(tmp_path, p, q), random_seed=6)
    rs = np.random.RandomState(2)
    data = rs.standard_normal((n, p)).astype(np.float32)
    z = rs.standard_normal((n, q)).astype(np.float32)
    keep = [sorted(rs.choice(p, rs.randint(1, p), replace=False).tolist()) for _ in range(n)]
    nan_form = np.full_like(data, np.nan)
    for i, r in enumerate(keep):
        nan_form[i, r] = data[i, r]
    kmax = max(len(r) for r in keep)
    ind = np.zeros((n, kmax), np.int64)
    msk = np.zeros((n, kmax), np.float32)
    for i, r in enumerate(keep):
        ind[i, :len(r)] = r
        msk[i, :len(r)] = 1.0
    a = model.get_log_posterior(z, nan_form, seed=9)
    b = model.get_log_posterior(z, data, ind_x1=ind, obs_mask=msk, seed=9)
    np.testing.assert_array_equal(a, b)
    d1 = model.tfp_mcmc_sampler(nan_form, n_mcmc=4, burn_in=4, step_size=0.05, num_leapfrog_steps=3, seed=3)
    d2 = model.tfp_mcmc_sampler(data, ind_x1=keep, n_mcmc=4, burn_in=4, step_size=0.05, num_leapfrog_steps=3, seed=3)
    np.testing.assert_array_equal(d1, d2)
    assert d1.shape == (4, n, q)
    # shared pattern given as one index list
    d3 = model.tfp_mcmc_sampler(data, ind_x1=[0, 2, 5], n_mcmc=2, burn_in=2, step_size=0.05, num_leapfrog_steps=3, seed=3)
    shared = np.full_like(data, np.nan)
    shared[:, [0, 2, 5]] = data[:, [0, 2, 5]]
    np.testing.assert_array_equal(d3, model.tfp_mcmc_sampler(shared, n_mcmc=2, burn_in=2, step_size=0.05, num_leapfrog_steps=3, seed=3))


def test_empty_inputs_are_accepted(tmp_path):
    q, units, p = 3, (16,), 7
    net = _net(q, units, p, seed=1)
    eng = _engine(net, q, units, p)
    dev = eng.device
    z0, x0 = torch.empty((0, q), device=dev), torch.empty((0, p), device=dev)
    lp, gr = eng.logpost(z0, x0, 1, 0, want_grad=True)
    assert lp.shape == (0,) and gr.shape == (0, q)
    out = eng.hmc_sample(x0, n_mcmc=2, burn_in=2, seed=1)
    assert out["draws"].shape == (2, 0, q)
    _, full = eng.decode(torch.empty((3, 0, q), device=dev), 1, 7, want_full=True)
    assert full.shape == (3, 0, p)
    eng.close()


def test_session_errors_are_reported(tmp_path):
    from bayesgm_amd.bvn_engine import BvnEngine
    eng = BvnEngine(7, 3, g_units=(16,))
    with pytest.raises(RuntimeError, match="no session"):
        eng.read(0)
    with pytest.raises(RuntimeError, match="wrong parameter count"):
        eng.begin(np.zeros(5, np.float32))
    with pytest.raises(ValueError):
        BvnEngine(7, 3, g_units=(16,) * 7)
    eng.begin(_net(3, (16,), 7))
    x = torch.zeros((40, 7), device=eng.device)
    z = torch.zeros((40, 3), device=eng.device)
    with pytest.raises(RuntimeError, match="bad argument"):
        eng.theta_step(x, z, torch.zeros(1, dtype=torch.int32, device=eng.device), 1e-3, 1, 0)        # a batch of one row
    with pytest.raises(RuntimeError, match="bad argument"):
        eng.theta_step(x, z, torch.zeros(33, dtype=torch.int32, device=eng.device), 1e-3, 1, 0)       # above max_batch
    eng.close()


def test_logpost_and_hmc_when_workgroups_walk_several_tiles():
    """More row tiles than launched workgroups (8 per CU): a workgroup then walks tiles with one workspace slice."""
    q, units, p, n = 3, (16,), 8, 140000
    net = _net(q, units, p, seed=14)
    rs = np.random.RandomState(15)
    z = rs.standard_normal((n, q)).astype(np.float32)
    x = rs.standard_normal((n, p)).astype(np.float32)
    x[rs.uniform(size=x.shape) < 0.2] = np.nan
    eng = _engine(net, q, units, p)
    lp, gr = eng.logpost(z, x, 77, 5, row_base=11, want_grad=True)
    n64 = OV.cast_vnet(net, np.float64)
    mask = (~np.isnan(x)).astype(np.float64)
    xc = np.where(np.isnan(x), 0.0, x).astype(np.float64)
    ref_lp, ref_gr = OV.log_posterior_and_grad(n64, z.astype(np.float64), xc, mask, OV.draw(n64, n, 77, 5, 11, np.float64))
    assert _rel(lp.cpu().numpy(), ref_lp) < 1e-5 and _rel(gr.cpu().numpy(), ref_gr) < 1e-4
    # two transitions for all rows in one launch == the same rows run as two halves (chains are independent)
    dev = eng.device
    xd = torch.from_numpy(x).to(dev)
    step = torch.full((1,), 0.05, device=dev)
    def run(lo, hi):
        st, lg, gd = torch.empty((hi - lo, q), device=dev), torch.empty(hi - lo, device=dev), torch.empty((hi - lo, q), device=dev)
        eng.hmc_run(xd[lo:hi], st, lg, gd, step, 0, 2, 0, 3, 9, init=True, row_base=lo)
        return st.cpu().numpy()
    whole = run(0, n)
    np.testing.assert_array_equal(whole[:1000], run(0, 1000))
    np.testing.assert_array_equal(whole[n - 3000:], run(n - 3000, n))
    eng.close()


# ---------------------------------------------------------------------------------------------------------------------------
# frozen-noise HMC as register-chained row tiles (csrc/bgmf_kernels.h): the reference's generator shape, hidden layers of 64 units
# ---------------------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("frozen", [True, False])
@pytest.mark.parametrize("q,units,p,n", [(3, (64,) * 3, 40, 100), (10, (64,) * 5, 50, 150), (20, (64,) * 5, 23, 70)])
def test_frozen_hmc_row_tile_chains_follow_oracle(q, units, p, n, frozen):
    """Same criterion as test_hmc_follows_oracle_over_a_few_transitions on the shapes bgmf_hmc_kernel serves (3 / 5 hidden layers of 64
    units; q = 20: two latent tiles; p not a multiple of 16: masked head columns; a row count that is no multiple of the 16-chain tile).
    frozen = False: the reference as written -- a new perturbation and new signs at every gradient evaluation (the linear stream)."""
    net = _net(q, units, p, seed=16)
    rs = np.random.RandomState(17)
    x = rs.standard_normal((n, p)).astype(np.float32)
    x[rs.uniform(size=x.shape) < 0.25] = np.nan
    eng = _engine(net, q, units, p, hmc_frozen_noise=frozen)
    seed = 43
    out = eng.hmc_sample(x, n_mcmc=3, burn_in=5, step_size=0.03, n_leapfrog=4, seed=seed, row_base=9)
    mask = (~np.isnan(x)).astype(np.float32)
    xc = np.where(np.isnan(x), 0.0, x).astype(np.float32)
    ref, info = OV.hmc_sampler(OV.cast_vnet(net, np.float64), xc.astype(np.float64), mask.astype(np.float64), 3, 5, 0.03, 4, seed, row0=9,
                               return_info=True, frozen=frozen)
    got = out["draws"].cpu().numpy()
    assert got.shape == ref.shape
    assert abs(float(out["step"].item()) - info["step"]) < 1e-6
    close = np.abs(got - ref).max(axis=(0, 2)) < 1e-3
    print("MEASURED frozen chains close to the oracle: %d of %d" % (close.sum(), n))
    assert close.mean() > 0.95
    eng.close()


@pytest.mark.parametrize("budget", [None, "1"])
def test_fresh_hmc_row_tile_chains_equal_the_workspace_kernel(monkeypatch, budget):
    """Fresh noise on bgmf_hmc_kernel<FRESH> against bgmb_hmc_kernel (BGM_BVN_NO_CHAINS=1) at BASELINE C4's shape; budget "1": one
    transition per launch (the perturbation buffer's size cuts a segment into launches) -- the cut must not change the chains."""
    q, units, p, n = 10, (64,) * 5, 500, 2500
    net = _net(q, units, p, seed=18)
    rs = np.random.RandomState(19)
    x = rs.standard_normal((n, p)).astype(np.float32)
    x[rs.uniform(size=x.shape) < 0.1] = np.nan
    if budget is None:
        monkeypatch.delenv("BGM_BVN_NOISE_BYTES", raising=False)
    else:
        monkeypatch.setenv("BGM_BVN_NOISE_BYTES", budget)
    res = []
    for no_chains in (False, True):
        if no_chains:
            monkeypatch.setenv("BGM_BVN_NO_CHAINS", "1")
        else:
            monkeypatch.delenv("BGM_BVN_NO_CHAINS", raising=False)
        eng = _engine(net, q, units, p, hmc_frozen_noise=False)
        dev = eng.device
        xd = torch.from_numpy(x).to(dev)
        state, logp, grad = torch.empty((n, q), device=dev), torch.empty(n, device=dev), torch.empty((n, q), device=dev)
        step = torch.full((1,), 0.02, device=dev)
        acc = torch.zeros(5, device=dev, dtype=torch.int32)
        eng.hmc_run(xd, state, logp, grad, step, 0, 3, 2 ** 30, 4, 11, init=True, row_base=5, acc_count=acc)
        eng.hmc_run(xd, state, logp, grad, step, 3, 2, 2 ** 30, 4, 11, row_base=5, acc_count=acc)
        res.append((state.cpu().numpy(), logp.cpu().numpy(), acc.cpu().numpy()))
        eng.close()
    (s0, l0, a0), (s1, l1, a1) = res
    close = np.abs(s0 - s1).max(axis=1) < 1e-3
    print("MEASURED fresh chains equal to the workspace kernel: %d of %d, acceptance %s vs %s" % (close.sum(), n, a0.tolist(), a1.tolist()))
    assert close.mean() > 0.99 and np.abs(a0 - a1).max() <= 0.01 * n
    assert np.abs(l0 - l1)[close].max() < 2e-3 * max(1.0, np.abs(l1).max())


@pytest.mark.parametrize("q,units,p,n", [(10, (64,) * 5, 500, 3000), (5, (64,) * 3, 100, 40000)])
def test_frozen_hmc_row_tile_chains_equal_the_lds_tile_engine(monkeypatch, q, units, p, n):
    """bgmf_hmc_kernel against gxf_bgm_hmc_kernel (BGM_BVN_NO_CHAINS=1) at BASELINE C4's shape and on a panel with more row tiles than one
    pass of the grid covers: same target, same random numbers, different summation orders -> the chains agree except where an
    accept / reject decision sat on the threshold."""
    net = _net(q, units, p, seed=18)
    rs = np.random.RandomState(19)
    x = rs.standard_normal((n, p)).astype(np.float32)
    x[rs.uniform(size=x.shape) < 0.1] = np.nan
    res = []
    for no_chains in (False, True):
        if no_chains:
            monkeypatch.setenv("BGM_BVN_NO_CHAINS", "1")
        else:
            monkeypatch.delenv("BGM_BVN_NO_CHAINS", raising=False)
        eng = _engine(net, q, units, p, hmc_frozen_noise=True)
        dev = eng.device
        xd = torch.from_numpy(x).to(dev)
        state, logp, grad = torch.empty((n, q), device=dev), torch.empty(n, device=dev), torch.empty((n, q), device=dev)
        step = torch.full((1,), 0.02, device=dev)
        acc = torch.zeros(4, device=dev, dtype=torch.int32)
        eng.hmc_run(xd, state, logp, grad, step, 0, 2, 2 ** 30, 5, 11, init=True, row_base=5, acc_count=acc)
        eng.hmc_run(xd, state, logp, grad, step, 2, 2, 2 ** 30, 5, 11, row_base=5, acc_count=acc)          # continued from the stored state
        res.append((state.cpu().numpy(), logp.cpu().numpy(), grad.cpu().numpy(), acc.cpu().numpy()))
        eng.close()
    (s0, l0, g0, a0), (s1, l1, g1, a1) = res
    close = np.abs(s0 - s1).max(axis=1) < 1e-3
    print("MEASURED chains equal to the tile engine: %d of %d, acceptance %s vs %s" % (close.sum(), n, a0.tolist(), a1.tolist()))
    assert close.mean() > 0.99 and np.abs(a0 - a1).max() <= 0.01 * n
    assert np.abs(l0 - l1)[close].max() < 2e-3 * max(1.0, np.abs(l1).max())
    # dlogp/dz jumps where a LeakyReLU input changes sign, so two states 1e-4 apart may sit on different sides of a kink: per row
    g_close = np.abs(g0 - g1).max(axis=1) < 2e-3 * np.abs(g1).max()
    assert g_close[close].mean() > 0.99, g_close[close].mean()


def test_model_fit_with_minibatches_beyond_64_rows(tmp_path):
    """BGM(use_bnn=True).fit(batch_size=128): any batch size in the reference (bgm/base.py:343); up to 256 rows here."""
    from bayesgm_amd.models import BGM
    n, p, q = 640, 12, 4
    data = _linear_panel(n, p, q)
    model = BGM(_params(tmp_path, p, q, bnn_mcmc_noise="frozen"), random_seed=7)
    mse0 = float(model.evaluate(data, data_z=np.zeros((n, q), np.float32), use_x_sd=False))
    model.fit(data, batch_size=128, epochs=40, epochs_per_eval=20, use_egm_init=True, egm_n_iter=40, egm_batches_per_eval=20, verbose=0)
    assert model.history_loss[-1] < mse0 and np.isfinite(model.history_loss).all()
    with pytest.raises(ValueError):
        model.fit(data, batch_size=300, epochs=1, use_egm_init=False, verbose=0)


@pytest.mark.parametrize("n", [1, 17, 129])
@pytest.mark.parametrize("frozen", [True, False])
def test_row_tile_chains_on_tiny_panels(monkeypatch, n, frozen):
    """One row, one row more than a tile, one row more than a workgroup's eight tiles: the chain kernel against the kernels it replaced."""
    q, units, p = 4, (64,) * 3, 21
    net = _net(q, units, p, seed=20)
    rs = np.random.RandomState(21)
    x = rs.standard_normal((n, p)).astype(np.float32)
    x[rs.uniform(size=x.shape) < 0.3] = np.nan
    res = []
    for no_chains in (False, True):
        if no_chains:
            monkeypatch.setenv("BGM_BVN_NO_CHAINS", "1")
        else:
            monkeypatch.delenv("BGM_BVN_NO_CHAINS", raising=False)
        eng = _engine(net, q, units, p, hmc_frozen_noise=frozen)
        out = eng.hmc_sample(x, n_mcmc=2, burn_in=3, step_size=0.05, n_leapfrog=3, seed=5, row_base=2)
        res.append(out["draws"].cpu().numpy())
        eng.close()
    assert res[0].shape == (2, n, q) and np.isfinite(res[0]).all()
    assert (np.abs(res[0] - res[1]).max(axis=(0, 2)) < 1e-3).mean() >= (1.0 if n == 1 else 0.9)
