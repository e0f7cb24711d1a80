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


def _engine(net, q, units, p, **kw):
    from bayesgm_amd.bvn_engine import BvnEngine
    eng = BvnEngine(p, q, g_units=units, **kw)
    eng.begin(net)
    return eng


def _rel(a, b):
    return float(np.abs(np.asarray(a, np.float64) - np.asarray(b, np.float64)).max() / (np.abs(b).max() + 1e-30))


def _flat(parts):
    return np.concatenate([np.asarray(a, np.float64).ravel() for a in parts])


@pytest.mark.parametrize("q,units,p,B", [(10, (64,) * 5, 20, 32), (5, (24, 40), 37, 19), (3, (16,), 9, 64)])
def test_theta_step_gradient_matches_oracle(q, units, p, B):
    net = _net(q, units, p)
    rs = np.random.RandomState(1)
    N = 200
    z = rs.standard_normal((N, q)).astype(np.float32)
    x = rs.standard_normal((N, p)).astype(np.float32)
    idx = rs.choice(N, B, replace=False).astype(np.int32)
    eng = _engine(net, q, units, p, kl_weight=0.01, max_batch=64)
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


def test_hmc_follows_oracle_over_a_few_transitions():
    q, units, p, n = 3, (16, 16), 8, 100
    net = _net(q, units, p, seed=6)
    rs = np.random.RandomState(7)
    x = rs.standard_normal((n, p)).astype(np.float32)
    x[rs.uniform(size=x.shape) < 0.25] = np.nan
    eng = _engine(net, q, units, p)
    seed = 42
    out = eng.hmc_sample(x, n_mcmc=3, burn_in=5, step_size=0.05, n_leapfrog=4, seed=seed, row_base=7)
    mask = (~np.isnan(x)).astype(np.float32)
    xc = np.where(np.isnan(x), 0.0, x).astype(np.float32)
    ref, info = OV.hmc_sampler(OV.cast_vnet(net, np.float64), xc.astype(np.float64), mask.astype(np.float64), 3, 5, 0.05, 4, seed, row0=7,
                               return_info=True)
    got = out["draws"].cpu().numpy()
    assert got.shape == ref.shape
    assert abs(float(out["step"].item()) - info["step"]) < 1e-6
    # accept/reject decisions near the threshold may flip in float32: almost all chains must agree closely
    close = np.abs(got - ref).max(axis=(0, 2)) < 1e-3
    assert close.mean() > 0.95
    eng.close()


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
