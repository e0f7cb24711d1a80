"""IdentifiableCausalBGM (models/causalbgm/identifiable.py of the reference): the conditional latent prior in
the sampling kernels (through the C ABI: bgm_causal_set_prior) and the class's fit / predict against oracle/identifiable.py and
oracle/causal.py (prior=...).  Tolerances as for the standard-prior kernels: log-posterior 2e-6 |lp| + 5e-4; chains identical on
>= 97 % of the rows; fit traces 2e-5 relative; the prior-network kernels (bgm_prior_table / bgm_prior_step) 1e-5."""
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
                                  dict(z_dims=[1, 1, 1, 7], p=20, binary=False, n=50),
                                  # outside the LDS-resident shapes (sum(z_dims) = 20; v_dim = 300): served by the general-width engine
                                  dict(z_dims=[5, 5, 5, 5], p=100, binary=False, n=90), dict(z_dims=[1, 1, 1, 7], p=300, binary=True, n=64)])
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


@pytest.mark.parametrize("units,k,B", [((64,), 10, 32), ((24, 40), 7, 17), ((), 5, 8), ((16, 16, 16), 3, 64)])
def test_prior_network_kernels_match_oracle(units, k, B):
    """bgm_prior_table and two consecutive bgm_prior_step calls (so that the Adam slots matter) against the float64 restatement of
    identifiable.py:195-226 in oracle/identifiable.py: latent step with fresh slots, prior-net gradients through the one-hot first
    layer, Adam on every prior parameter.  Shapes: the default, two hidden layers with an odd batch, no hidden layer, three."""
    import ctypes as C
    import torch
    from bayesgm_amd import _lib
    from oracle.nets import mlp_backward, sigmoid
    from oracle.fit import AdamState, adam_lr_t, flat_params, flat_grads, B1, B2, ADAM_EPS
    rs = np.random.RandomState(11)
    z_dims, p, n = [1, 1, 1, 7], 20, 90
    q = sum(z_dims)
    eng = _engine(_model(5, z_dims, p, False))
    dev = eng.device
    pn32 = [(W, (0.3 * rs.randn(*b.shape)).astype(np.float32)) for W, b in OI.init_prior_net(rs, k, q, units)]
    pn = [(W.astype(np.float64), b.astype(np.float64)) for W, b in pn32]
    dims = [k] + list(units) + [q + 1]
    cfg = _lib.PriorConfig(len(dims) - 1, (C.c_int32 * 5)(*(dims + [0] * (5 - len(dims)))))
    cnt = C.c_int64()
    _lib.check(eng.lib.bgm_prior_n_params(C.byref(cfg), C.byref(cnt)))
    flat = np.concatenate([np.concatenate([W.ravel(), b.ravel()]) for W, b in pn32])
    assert cnt.value == flat.size
    theta = torch.from_numpy(flat).to(dev)
    m_, v_ = torch.zeros_like(theta), torch.zeros_like(theta)
    tab = torch.empty((k, q + 2), device=dev)
    _lib.check(eng.lib.bgm_prior_table(eng.h, C.byref(cfg), theta.data_ptr(), tab.data_ptr(), None), "bgm_prior_table")
    ref_tab = OI.prior_table(pn, q)
    assert np.abs(tab.cpu().numpy() - ref_tab).max() <= 1e-5 * np.abs(ref_tab).max()
    z = rs.randn(n, q)
    seg = rs.randint(0, k, n)
    zd = torch.from_numpy(z.astype(np.float32)).to(dev)
    segd = torch.from_numpy(seg.astype(np.int32)).to(dev)
    z64 = z.astype(np.float32).astype(np.float64)
    opt = AdamState(flat_params(pn))
    out = torch.zeros(2, device=dev)
    lr_z, lr_p = 3e-3, 2e-3
    used = []
    for step, (tz, tp) in enumerate(((3, 5), (4, 6))):
        idx = rs.choice(n, B, replace=False).astype(np.int32)
        used.append(idx)
        dz = rs.randn(B, q).astype(np.float32)
        idx_d, dz_d = torch.from_numpy(idx).to(dev), torch.from_numpy(dz).to(dev)          # (held: the call takes raw pointers)
        _lib.check(eng.lib.bgm_prior_step(eng.h, C.byref(cfg), theta.data_ptr(), m_.data_ptr(), v_.data_ptr(), segd.data_ptr(), zd.data_ptr(),
                                          idx_d.data_ptr(), B, dz_d.data_ptr(), lr_z, lr_p, tz, tp, out.data_ptr(), None), "bgm_prior_step")
        torch.cuda.synchronize()
        zb = z64[idx].copy()
        mu, s2, (o, cache) = OI.prior_params(pn, seg[idx])
        d = zb - mu
        ssq = (d ** 2).sum(axis=1)
        loss_prior = (ssq / (2 * s2) + q * np.log(s2) / 2).mean()
        g = dz.astype(np.float64) - zb / B + d / s2[:, None] / B
        dout = np.zeros_like(o)
        dout[:, :-1] = -d / s2[:, None] / B
        dout[:, -1] = (-ssq / (2 * s2 * s2) + q / (2 * s2)) / B * sigmoid(o[:, -1])
        pgrads, _ = mlp_backward(pn, cache, dout)
        if step == 0:
            opt.t = tp - 1                                   # the step counter the call was given
        z64[idx] = zb - adam_lr_t(lr_z, tz) * ((1 - B1) * g) / (np.sqrt((1 - B2) * g * g) + ADAM_EPS)
        opt.apply(flat_params(pn), flat_grads(pgrads), lr_p)
        got = out.cpu().numpy()
        assert abs(got[0] - loss_prior) <= 1e-5 * abs(loss_prior) + 1e-6 and abs(got[1] - 0.5 * (zb ** 2).sum(axis=1).mean()) <= 1e-5
    # Adam normalises a step to ~lr whatever the gradient's size: compare against the distance moved (as the fit tests do)
    ref_flat = np.concatenate([np.concatenate([W.ravel(), b.ravel()]) for W, b in pn])
    moved = np.abs(ref_flat - flat).max()
    assert moved > 1e-3 and np.abs(theta.cpu().numpy() - ref_flat).max() <= 2e-3 * moved
    zmoved = np.abs(z64 - z.astype(np.float32)).max()
    assert np.abs(zd.cpu().numpy() - z64).max() <= 2e-3 * zmoved
    untouched = ~np.isin(np.arange(n), np.concatenate(used))
    assert untouched.any() and np.array_equal(zd.cpu().numpy()[untouched], z.astype(np.float32)[untouched])


def test_prior_grad_and_apply_equal_the_fused_step():
    """Data-parallel form of the prior step: two "ranks" run bgm_prior_grad on the halves of a minibatch (batch means over the whole
    minibatch), their gradients are added (the all-reduce) and bgm_prior_apply takes the Adam step -> the parameters, slots and latents
    bgm_prior_step produces on the whole minibatch."""
    import ctypes as C
    import torch
    from bayesgm_amd import _lib
    rs = np.random.RandomState(12)
    z_dims, p, n, k, B = [1, 1, 1, 7], 20, 90, 6, 32
    q = sum(z_dims)
    eng = _engine(_model(5, z_dims, p, False))
    dev = eng.device
    pn32 = [(W, (0.3 * rs.randn(*b.shape)).astype(np.float32)) for W, b in OI.init_prior_net(rs, k, q, (64,))]
    dims = [k, 64, q + 1]
    cfg = _lib.PriorConfig(len(dims) - 1, (C.c_int32 * 5)(*(dims + [0] * (5 - len(dims)))))
    flat = np.concatenate([np.concatenate([W.ravel(), b.ravel()]) for W, b in pn32])
    z = rs.randn(n, q).astype(np.float32)
    segd = torch.from_numpy(rs.randint(0, k, n).astype(np.int32)).to(dev)
    idx = rs.choice(n, B, replace=False).astype(np.int32)
    dz = rs.randn(B, q).astype(np.float32)
    res = []
    for split in (False, True):
        theta = torch.from_numpy(flat.copy()).to(dev)
        m_, v_ = torch.full_like(theta, 0.01), torch.full_like(theta, 0.002)
        zd = torch.from_numpy(z.copy()).to(dev)
        out = torch.zeros(2, device=dev)
        idx_d, dz_d = torch.from_numpy(idx).to(dev), torch.from_numpy(dz).to(dev)
        if not split:
            _lib.check(eng.lib.bgm_prior_step(eng.h, C.byref(cfg), theta.data_ptr(), m_.data_ptr(), v_.data_ptr(), segd.data_ptr(), zd.data_ptr(),
                                              idx_d.data_ptr(), B, dz_d.data_ptr(), 3e-3, 2e-3, 3, 5, out.data_ptr(), None), "bgm_prior_step")
            tot = out.cpu().numpy()
        else:
            gsum, tot = torch.zeros_like(theta), np.zeros(2)
            for r in range(2):
                g = torch.zeros_like(theta)
                i_r, d_r = idx_d[16 * r:16 * r + 16].contiguous(), dz_d[16 * r:16 * r + 16].contiguous()
                _lib.check(eng.lib.bgm_prior_grad(eng.h, C.byref(cfg), theta.data_ptr(), segd.data_ptr(), zd.data_ptr(), i_r.data_ptr(), 16, B,
                                                  d_r.data_ptr(), 3e-3, 3, g.data_ptr(), out.data_ptr(), None), "bgm_prior_grad")
                gsum += g
                tot += out.cpu().numpy()
            _lib.check(eng.lib.bgm_prior_apply(eng.h, C.byref(cfg), theta.data_ptr(), m_.data_ptr(), v_.data_ptr(), gsum.data_ptr(), 2e-3, 5, None),
                       "bgm_prior_apply")
        torch.cuda.synchronize()
        res.append((theta.cpu().numpy(), m_.cpu().numpy(), v_.cpu().numpy(), zd.cpu().numpy(), tot))
    a, b = res
    assert np.abs(a[0] - b[0]).max() <= 1e-6 and np.abs(a[1] - b[1]).max() <= 1e-6 * np.abs(a[1]).max() + 1e-9
    assert np.abs(a[2] - b[2]).max() <= 1e-6 * np.abs(a[2]).max() + 1e-12 and np.abs(a[3] - b[3]).max() <= 1e-6
    assert np.abs(a[4] - b[4]).max() <= 1e-5 * np.abs(a[4]).max()
    assert np.abs(a[0] - flat).max() > 1e-4                   # the step moved the prior net


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
    # use_bnn=True constructs the Bayesian form (models/identifiable_bnn.py; tests/test_gpu_identifiable_bnn.py)
    assert type(IdentifiableCausalBGM(dict(prm, use_bnn=True))).__name__ == "IdentifiableCausalBGMBayes"


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
    assert np.array_equal(a._prior_m.cpu().numpy(), b._prior_m.cpu().numpy()) and np.array_equal(a._prior_v.cpu().numpy(), b._prior_v.cpu().numpy())
    assert float(a._prior_m.abs().max()) > 0
    got = b.get_log_posterior(x, y, v, z, u)
    assert np.array_equal(want, got)
    c = IdentifiableCausalBGM(prm, timestamp="other", random_seed=99)    # a fresh directory: a fresh prior network
    assert not np.array_equal(c.prior_parameters()[0][0], a.prior_parameters()[0][0])


def test_two_rank_predict_equals_the_single_process_predict():
    """IdentifiableCausalBGM.predict under torch.distributed (two ranks on this GPU over gloo): rank 0's segment draw is broadcast,
    rows are sharded, the ADRF draw sums / per-row ITE results reduced -- the result equals the single-process predict of the same
    seeded model (chains keyed by the global row; only the order of the partial sums differs)."""
    import json
    from conftest import run_two_ranks
    from bayesgm_amd.models import IdentifiableCausalBGM
    from bayesgm_amd.datasets import Sim_Hirano_Imbens_sampler
    r = run_two_ranks("dp_ident_smoke.py")
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    objs, pos, dec = [], 0, json.JSONDecoder()
    while True:
        pos = r.stdout.find('{"rank"', pos)
        if pos < 0:
            break
        o, end = dec.raw_decode(r.stdout[pos:])
        objs.append(o)
        pos += end
    assert len(objs) == 2 and objs[0]["adrf"] == objs[1]["adrf"] and objs[0]["ite_sum"] == objs[1]["ite_sum"]
    # data-parallel fit: identical networks and prior net on both ranks, finite and identical epoch statistics
    assert all(o["fit_spread"] == 0.0 and o["fit_finite"] for o in objs) and objs[0]["fit_loss"] == objs[1]["fit_loss"]
    assert np.all(np.isfinite(objs[0]["fit_loss"])) and len(objs[0]["fit_mse_v"]) == 2
    two = objs[0]
    x, y, v = Sim_Hirano_Imbens_sampler(N=1205, v_dim=50, seed=1).load_all()

    def params(binary):
        return dict(dataset="dpi", output_dir="gpurun_out/dpi", save_res=False, save_model=False, binary_treatment=binary, use_bnn=False,
                    z_dims=[1, 1, 1, 7], v_dim=50, lr_theta=1e-3, lr_z=1e-3, g_units=[64] * 5, f_units=[64, 32, 8], h_units=[64, 32, 8],
                    e_units=[64] * 5, dz_units=[64, 32, 8], kl_weight=1e-4, lr=2e-4, g_d_freq=5, use_z_rec=True, n_segments=6)
    m = IdentifiableCausalBGM(params(False), random_seed=2)
    np.random.seed(5)
    adrf, interval = m.predict((x, y, v), alpha=0.05, n_mcmc=40, burn_in=40, x_values=np.linspace(0, 3, 6), q_sd=0.5, verbose=0)
    assert np.abs(np.array(two["adrf"]) - adrf).max() <= 1e-5 and np.abs(np.array(two["interval"]) - interval.ravel()).max() <= 1e-5
    assert abs(two["acc"] - m.last_acceptance_rate) < 1e-12
    mb = IdentifiableCausalBGM(params(True), random_seed=3)
    np.random.seed(6)
    ite, iv = mb.predict(((x > np.median(x)).astype(np.float32), y, v), alpha=0.05, n_mcmc=40, burn_in=40, q_sd=0.5, verbose=0)
    assert np.abs(np.array(two["ite_head"]) - ite[:5]).max() <= 1e-6 and np.abs(np.array(two["ite_tail"]) - ite[-5:]).max() <= 1e-6
    assert abs(two["ite_sum"] - float(ite.sum())) <= 1e-3 and abs(two["iv_sum"] - float(iv.sum())) <= 1e-3
    # adaptive proposal scale under data parallelism (identifiable.py:585-606: one acceptance window over ALL rows): the window's
    # count is all-reduced, so the two-rank run walks the single-process schedule of scales and returns its curve
    assert objs[0]["adrf_adaptive"] == objs[1]["adrf_adaptive"]
    ma = IdentifiableCausalBGM(params(False), random_seed=2)
    np.random.seed(5)
    adrf_a, _ = ma.predict((x, y, v), alpha=0.05, n_mcmc=20, burn_in=160, x_values=np.linspace(0, 3, 6), q_sd=-1.0, verbose=0)
    assert np.abs(np.array(two["adrf_adaptive"]) - adrf_a).max() <= 1e-5
    assert abs(two["acc_adaptive"] - ma.last_acceptance_rate) < 1e-12


def test_general_width_pack_follows_a_fit_on_the_row_tile_chains():
    """default widths on a shape that is not LDS-resident, WITH a conditional prior, sample on the general-width
    engine's packed weights, while a minibatch fit of that shape steps on the row-tile chains and refreshes the host nets only.  The
    pack must be rebuilt after bgm_causal_fit_end: log posterior after the fit = log posterior of a fresh handle given the fitted nets
    (and differs from the pre-fit one)."""
    import torch
    rs = np.random.RandomState(8)
    z_dims, p, n, k = [5, 5, 5, 5], 100, 96, 4
    q = sum(z_dims)
    m = _model(5, z_dims, p, False)
    x, y, v = _data(n, p, 6, False)
    z = rs.randn(n, q).astype(np.float32)
    seg = torch.from_numpy(rs.randint(0, k, n).astype(np.int32)).cuda()
    tab = torch.from_numpy(OI.prior_table(_prior(rs, k, q), q)).cuda()
    eng = _engine(m)
    eng.set_prior(seg, tab)
    lp0 = eng.logpost(x.ravel(), y.ravel(), v, z).cpu().numpy()          # builds the general-width pack
    assert eng.describe().startswith("sampler=gx_causal_mh_kernel")
    dev = eng.device
    xd, yd, vd = (torch.from_numpy(a).to(dev) for a in (x.ravel(), y.ravel(), v))
    zd = torch.from_numpy(z).to(dev)
    zm, zv = torch.zeros_like(zd), torch.zeros_like(zd)
    eng.set_prior(None, None)            # as IdentifiableCausalBGM.fit does: the minibatch steps take the prior through bgm_prior_step
    npar = eng.fit_begin(n, 32)
    assert "general-width" not in eng.describe(32).split("fit=")[-1]     # the minibatch steps do NOT run on the general-width engine
    grad = torch.empty(npar, device=dev)
    for s_ in range(3):
        idx = torch.arange(32 * s_, 32 * s_ + 32, device=dev, dtype=torch.int32)
        eng.fit_theta_grad(xd, yd, vd, zd, idx, 32, grad)
        eng.fit_theta_apply(grad, 1e-2)
    eng.fit_end()
    eng.set_prior(seg, tab)
    nets = {name: eng.get_weights(i, [W.shape[0] for W, _ in m[name]] + [m[name][-1][0].shape[1]]) for i, name in enumerate(("g", "f", "h"))}
    assert np.abs(nets["g"][0][0] - m["g"][0][0]).max() > 1e-3           # the fit moved the weights
    lp1 = eng.logpost(x.ravel(), y.ravel(), v, z).cpu().numpy()
    fresh = _engine(dict(m, **nets))
    fresh.set_prior(seg, tab)
    ref = fresh.logpost(x.ravel(), y.ravel(), v, z).cpu().numpy()
    assert np.abs(lp1 - lp0).max() > 1e-2
    np.testing.assert_allclose(lp1, ref, rtol=1e-6, atol=1e-5)


def test_event_form_with_the_conditional_prior_is_bit_identical():
    """the event form of the retained phase (outcome cache mode 2, csrc/causal_event_kernels.h) on the PRIOR = 1 transition kernel: chains and
    per-slot ADRF sums equal the fused kernel's with every dose evaluated at every retained draw"""
    import torch
    from bayesgm_amd import _lib
    rs = np.random.RandomState(4)
    z_dims, p, n, k = [1, 1, 1, 7], 200, 1500, 5
    q = sum(z_dims)
    m = _model(5, z_dims, p, False)
    x, y, v = _data(n, p, 6, False)
    seg = torch.from_numpy(rs.randint(0, k, n).astype(np.int32)).cuda()
    tab = torch.from_numpy(OI.prior_table(_prior(rs, k, q), q)).cuda()
    eng = _engine(m)
    eng.set_prior(seg, tab)
    outs = {}
    for mode in (False, True):
        eng.set_outcome_cache(mode)
        eng.outcome_cache_stats(reset=True)
        outs[mode] = eng.mh_sample(x, y, v, 20, 45, 1.0, 9, want_draws=True, effect=_lib.EFFECT_ADRF, x_values=np.linspace(0, 3, 20))
        outs[mode]["stats"] = eng.outcome_cache_stats()
    assert outs[True]["stats"][1] == n * 45                               # chain-iterations: the event form ran
    for kk in ("adrf_partial", "draws", "acc_count", "state"):
        assert np.array_equal(outs[True][kk].cpu().numpy(), outs[False][kk].cpu().numpy()), kk
    eng.set_prior(None, None)
    plain = eng.mh_sample(x, y, v, 20, 45, 1.0, 9, effect=_lib.EFFECT_ADRF, x_values=np.linspace(0, 3, 20))
    assert not np.array_equal(plain["state"].cpu().numpy(), outs[True]["state"].cpu().numpy())      # the prior matters here
