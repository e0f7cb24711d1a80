"""IdentifiableCausalBGM (models/causalbgm/identifiable.py of the reference): the conditional latent prior in
the sampling kernels (through the C ABI: bgm_causal_set_prior) and the class's fit / predict against oracle/identifiable.py and
oracle/causal.py (prior=...).  Tolerances as for the standard-prior kernels: log-posterior 2e-6 |lp| + 5e-4; chains identical on
>= 97 % of the rows; fit traces 2e-5 relative (torch fp32 prior-network ops on the host side of the ABI)."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from oracle import causal as OC            # noqa: E402
from oracle import identifiable as OI      # noqa: E402
from tests.test_gpu_causal import _model, _data, _engine  # noqa: E402
from tests.test_gpu_class_level import _params, _oracle_model, _flat, GOLD  # noqa: E402


def _prior(rs, k, q):
    pn = OI.init_prior_net(rs, k, q)
    return [(W, (0.3 * rs.randn(*b.shape)).astype(np.float32)) for W, b in pn]


@pytest.mark.parametrize("case", [dict(z_dims=[1, 1, 1, 7], p=200, binary=False, n=333),
                                  dict(z_dims=[3, 3, 6, 6], p=100, binary=True, n=200),
                                  dict(z_dims=[1, 1, 1, 7], p=20, binary=False, n=50)])
def test_conditional_prior_log_posterior_and_chains(case):
    import torch
    rs = np.random.RandomState(3)
    m = _model(5, case["z_dims"], case["p"], case["binary"])
    q, k = sum(case["z_dims"]), 7
    x, y, v = _data(case["n"], case["p"], 6, case["binary"])
    z = rs.randn(case["n"], q).astype(np.float32)
    seg = rs.randint(0, k, case["n"])
    pn = _prior(rs, k, q)
    tab = OI.prior_table(pn, q)
    mu, s2, _ = OI.prior_params([(W.astype(np.float64), b.astype(np.float64)) for W, b in pn], seg)
    eng = _engine(m)
    eng.set_prior(torch.from_numpy(seg.astype(np.int32)).cuda(), torch.from_numpy(tab).cuda())
    lp = eng.logpost(x.ravel(), y.ravel(), v, z).cpu().numpy()
    ref = OC.log_posterior(OC.cast_model(m, np.float64), x.astype(np.float64), y.astype(np.float64), v.astype(np.float64),
                           z.astype(np.float64), prior=(mu, s2))
    assert np.all(np.abs(lp - ref) <= 2e-6 * np.abs(ref) + 5e-4), np.abs(lp - ref).max()
    std = OC.log_posterior(OC.cast_model(m, np.float64), x.astype(np.float64), y.astype(np.float64), v.astype(np.float64), z.astype(np.float64))
    assert np.abs(ref - std).max() > 0.1                      # the prior matters in this test
    out = eng.mh_sample(x, y, v, 30, 10, 0.4, 77, want_draws=True)
    ref_draws = OC.mh_sampler(m, (x, y, v), 30, 10, 0.4, 77, prior=(mu.astype(np.float32), s2.astype(np.float32)))
    same = np.all(np.abs(out["draws"].cpu().numpy()[-1] - ref_draws[-1]) <= 1e-4, axis=1).mean()
    assert same >= 0.97, same
    eng.set_prior(None, None)
    lp0 = eng.logpost(x.ravel(), y.ravel(), v, z).cpu().numpy()
    assert np.all(np.abs(lp0 - std) <= 2e-6 * np.abs(std) + 2e-4)          # cleared: back to N(0, I)


def test_identifiable_fit_trace_and_predict():
    from bayesgm_amd.models import IdentifiableCausalBGM
    g = np.load(GOLD)
    x, y, v = g["x"][:1000], g["y"][:1000], g["v"][:1000]
    n, q, lr = len(x), 10, 1e-3
    prm = dict(_params(), lr_theta=lr, lr_z=lr, n_segments=6)
    model = IdentifiableCausalBGM(prm, random_seed=8)
    m = _oracle_model(model, np.float64)
    pn = [(W.astype(np.float64), b.astype(np.float64)) for W, b in model.prior_parameters()]
    init_p = _flat(pn)
    host_state = np.random.get_state()
    model.fit((x, y, v), batch_size=32, epochs=1, epochs_per_eval=1, use_egm_init=False, verbose=0)
    np.random.set_state(host_state)
    seg = np.random.randint(0, 6, size=n)
    assert np.array_equal(seg, model.segments)
    z0 = np.random.normal(0, 1, size=(n, q)).astype('float32')
    st = OI.IdentState(m, pn, z0.astype(np.float64), lr, lr)
    x64, y64, v64 = x.astype(np.float64), y.astype(np.float64), v.astype(np.float64)
    for epoch in range(2):
        perm = np.random.choice(n, n, replace=False)
        hist = np.array([OI.fit_step(st, x64, y64, v64, perm[i:i + 32], seg) for i in range(0, n - 32 + 1, 32)])   # 31 batches, 8 rows skipped
        assert len(hist) == 31
        want = hist.mean(axis=0)
        got = model.fit_history[epoch]
        for key, w in zip(("loss_x", "loss_mse_x", "loss_y", "loss_mse_y", "loss_v", "loss_mse_v", "loss_postrior_z"), want):
            assert abs(got[key] - w) <= 2e-5 * abs(w) + 1e-6, (epoch, key, got[key], w)
        dose, mse_x, mse_y, mse_v = OC.evaluate(m, (x64, y64, v64), data_z=st.data_z)
        assert abs(got["mse_y"] - mse_y) <= 2e-5 * abs(mse_y) and abs(got["mse_v"] - mse_v) <= 2e-5 * abs(mse_v)
    moved = np.abs(_flat(st.pnet) - init_p).max()
    diff = np.abs(_flat(model.prior_parameters()) - _flat(st.pnet)).max()
    print("prior net moved %.3e, |class - oracle| %.3e" % (moved, diff))
    assert diff <= 0.02 * moved + 1e-7
    dz = np.abs(model.data_z.cpu().numpy() - st.data_z).max()
    print("latents moved %.3e, |class - oracle| %.3e" % (np.abs(st.data_z - z0).max(), dz))
    assert dz <= 0.02 * np.abs(st.data_z - z0).max() + 1e-7
    # predict: shapes, ordering of the interval, and the log-posterior / sampler surface with data_u
    xs = np.linspace(0, 3, 5)
    adrf, interval = model.predict((x, y, v), alpha=0.05, n_mcmc=40, x_values=xs, q_sd=0.5, burn_in=60, verbose=0)
    assert adrf.shape == (5,) and interval.shape == (5, 2) and np.all(interval[:, 0] <= adrf) and np.all(adrf <= interval[:, 1])
    u = np.eye(6, dtype=np.float32)[seg]
    lp = model.get_log_posterior(x, y, v, z0, u)
    mu, s2, _ = OI.prior_params(st.pnet, seg)
    ref = OC.log_posterior(m, x64, y64, v64, z0.astype(np.float64), prior=(mu, s2))
    assert np.all(np.abs(lp - ref) <= 5e-6 * np.abs(ref) + 2e-3), np.abs(lp - ref).max()
    samples, data_u = model.metropolis_hastings_sampler((x, y, v), q_sd=0.5, burn_in=10, n_keep=5)
    assert samples.shape == (5, n, q) and data_u.shape == (n, 6) and np.all(data_u.sum(axis=1) == 1)
    with pytest.raises(NotImplementedError):
        IdentifiableCausalBGM(dict(prm, use_bnn=True))


def test_checkpoint_round_trip_restores_the_prior_network(tmp_path):
    """tf.train.Checkpoint of the reference tracks prior_net and prior_optimizer (identifiable.py:112-128): a model re-created on the
    timestamp of a saved run must continue with the TRAINED prior network, its Adam slots and step counters, not a fresh one."""
    from bayesgm_amd.models import IdentifiableCausalBGM
    g = np.load(GOLD)
    x, y, v = g["x"][:256], g["y"][:256], g["v"][:256]
    prm = dict(_params(), n_segments=5, output_dir=str(tmp_path), save_model=True)
    a = IdentifiableCausalBGM(prm, timestamp="ck", random_seed=3)
    a.fit((x, y, v), batch_size=32, epochs=1, epochs_per_eval=1, use_egm_init=False, verbose=0)
    a.save_checkpoint("final")
    z = np.random.RandomState(0).randn(len(x), 10).astype(np.float32)
    u = np.eye(5, dtype=np.float32)[a.segments]
    want = a.get_log_posterior(x, y, v, z, u)
    b = IdentifiableCausalBGM(prm, timestamp="ck", random_seed=99)       # restores the latest checkpoint of the directory
    for (Wa, ba), (Wb, bb) in zip(a.prior_parameters(), b.prior_parameters()):
        assert np.array_equal(Wa, Wb) and np.array_equal(ba, bb)
    assert (b._prior_t, b._z_t) == (a._prior_t, a._z_t) and a._prior_t > 0
    for (ma, _), (mb, _) in zip(a._prior_m, b._prior_m):
        assert np.array_equal(ma.cpu().numpy(), mb.cpu().numpy())
    got = b.get_log_posterior(x, y, v, z, u)
    assert np.array_equal(want, got)
    c = IdentifiableCausalBGM(prm, timestamp="other", random_seed=99)    # a fresh directory: a fresh prior network
    assert not np.array_equal(c.prior_parameters()[0][0], a.prior_parameters()[0][0])
