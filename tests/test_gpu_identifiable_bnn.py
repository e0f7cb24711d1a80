"""IdentifiableCausalBGM(use_bnn=True) (reference: models/causalbgm/identifiable.py:56-67, 195-226, 497-614 with Bayesian nets; SURVEY.md 8f
row N4): the Bayesian conditional-prior network through the C ABI -- the joint latent / prior-net step (bgm_bprior_step), the prior inside
the log posterior and the Metropolis-Hastings sampler (bgm_bnn_set_prior) -- and the class, against oracle/identifiable.py (bnn_*
functions, float64, same Philox streams).  Tolerances: the step kernel 2e-5 relative on parameters after two steps (Adam slots in play),
log posterior as the other Bayesian kernels (2e-5 |lp|_max + 2e-3), chains identical except where an accept decision lies within fp32 noise."""
import ctypes as C

import numpy as np
import pytest
import torch

from oracle import bnn as OB
from oracle import identifiable as OI
from oracle.fit import AdamState

pytestmark = pytest.mark.gpu
f64 = lambda a: a.astype(np.float64)


def _model(binary, z_dims=(1, 1, 1, 7), p=50, seed=0):
    from tests.test_gpu_bnf import _model as mk
    return mk(binary, z_dims=z_dims, p=p, seed=seed)


def _prior(rs, k, q, units, norm):
    pn = OI.init_prior_bnn(rs, k, q, units)
    pn["gamma"] = (1.0 + 0.2 * rs.standard_normal(k)).astype(np.float32)
    pn["beta"] = (0.1 * rs.standard_normal(k)).astype(np.float32)
    if norm == "fixed":
        pn["norm"] = "fixed"
    return pn


def _cfg(dims):
    from bayesgm_amd import _lib
    return _lib.PriorConfig(len(dims) - 1, (C.c_int32 * 5)(*(list(dims) + [0] * (5 - len(dims)))))


@pytest.mark.parametrize("units,k,B,norm", [((64,), 10, 32, "fixed"), ((64,), 10, 32, "batch"), ((24, 40), 7, 17, "fixed"), ((), 5, 8, "batch"),
                                            ((16, 16, 16), 3, 64, "fixed")])
def test_bayesian_prior_step_matches_oracle(units, k, B, norm):
    """two consecutive bgm_bprior_step calls (so that the Adam slots matter): latents of the batch rows, every prior parameter, the three
    outputs, against oracle.identifiable.bnn_prior_step_given_dz with the same noise (net id 4, key = seed, call ids 11 and 12)"""
    from bayesgm_amd import _lib
    from bayesgm_amd.bnn_engine import flatten_bnn, unflatten_bnn
    from tests.test_gpu_bnf import _engine
    rs = np.random.RandomState(3)
    z_dims, n = [1, 1, 1, 7], 90
    q = sum(z_dims)
    eng = _engine(_model(False, z_dims=tuple(z_dims)))
    dev = eng.device
    dims = [k] + list(units) + [q + 1]
    pn32 = _prior(rs, k, q, units, norm)
    pn = OB.cast_bnn(pn32, np.float64)
    cfg = _cfg(dims)
    cnt = C.c_int64()
    _lib.check(eng.lib.bgm_bprior_n_params(C.byref(cfg), C.byref(cnt)))
    flat = flatten_bnn(pn32)
    assert cnt.value == flat.size
    theta = torch.from_numpy(flat).to(dev)
    m_, v_ = torch.zeros_like(theta), torch.zeros_like(theta)
    data_z32 = rs.standard_normal((n, q)).astype(np.float32)
    data_z = torch.from_numpy(data_z32.copy()).to(dev)
    zo = f64(data_z32)
    seg = rs.randint(0, k, n)
    seg_dev = torch.from_numpy(seg.astype(np.int32)).to(dev)
    popt = AdamState(OB.flat_params(pn))
    out = torch.zeros(3, device=dev)
    seed, klw, lr_z, lr_p = (5 << 32) | 99, 0.37, 1e-2, 3e-3
    for step in (1, 2):
        idx = rs.choice(n, B, replace=False).astype(np.int32)
        dz32 = (0.3 * rs.standard_normal((B, q))).astype(np.float32)
        idx_dev, dz_dev = torch.from_numpy(idx).to(dev), torch.from_numpy(dz32).to(dev)
        _lib.check(eng.lib.bgm_bprior_step(eng.h, C.byref(cfg), 1 if norm == "fixed" else 0, klw, theta.data_ptr(), m_.data_ptr(), v_.data_ptr(),
                                           seg_dev.data_ptr(), data_z.data_ptr(), idx_dev.data_ptr(), B, B, 0, dz_dev.data_ptr(), lr_z, lr_p, step,
                                           step, seed, 10 + step, None, 1, out.data_ptr(), eng._stream()), "bgm_bprior_step")
        noise = OI.bnn_prior_noise(pn, B, seed, 10 + step, dtype=np.float64)
        lp, lz, klv = OI.bnn_prior_step_given_dz(pn, popt, zo, idx, seg[idx], f64(dz32), noise, lr_z, step, lr_p, klw)
        got = out.cpu().numpy()
        assert abs(got[0] - lp) <= 2e-5 * abs(lp) + 1e-5 and abs(got[1] - lz) <= 2e-5 * abs(lz) + 1e-5 and abs(got[2] - klv) <= 2e-5 * abs(klv)
    np.testing.assert_allclose(data_z.cpu().numpy(), zo, rtol=2e-5, atol=2e-6)
    ref = np.concatenate([a.ravel() for a in OB.flat_params(pn)])
    np.testing.assert_allclose(theta.cpu().numpy(), ref, rtol=2e-5, atol=3e-6)
    assert np.abs(theta.cpu().numpy() - flat).max() > 1e-3            # the net moved
    eng.close()


@pytest.mark.parametrize("precision", ["fp32", "f16x3"])
@pytest.mark.parametrize("binary,p,z_dims,n,bs", [(False, 200, (1, 1, 1, 7), 700, 300), (True, 100, (3, 3, 6, 6), 520, 520), (False, 50, (4, 4, 4, 4), 75, 75)])
def test_conditional_prior_log_posterior_blocks(binary, p, z_dims, n, bs, precision):
    from bayesgm_amd import _lib
    from bayesgm_amd.bnn_engine import flatten_bnn
    from tests.test_gpu_bnf import _engine, _panel
    m = _model(binary, z_dims=z_dims, p=p)
    z, x, y, v = _panel(m, n)
    q, k = sum(z_dims), 6
    rs = np.random.RandomState(8)
    pn32 = _prior(rs, k, q, (64,), "fixed")
    seg = rs.randint(0, k, n)
    eng = _engine(m)
    eng.set_precision(precision)          # f16x3: the split-precision kernels (csrc/bnx_kernels.h) read the same per-row prior table
    T = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(eng.device)
    cfg = _cfg([k, 64, q + 1])
    theta, seg_dev = T(flatten_bnn(pn32)), T(seg.astype(np.int32))
    seed, stream = (3 << 32) | 1234, 77
    std = eng.logpost(T(x[:, 0]), T(y[:, 0]), T(v), T(z), bs, seed, stream, block0=2).cpu().numpy()
    _lib.check(eng.lib.bgm_bnn_set_prior(eng.h, C.byref(cfg), theta.data_ptr(), seg_dev.data_ptr()), "bgm_bnn_set_prior")
    got = eng.logpost(T(x[:, 0]), T(y[:, 0]), T(v), T(z), bs, seed, stream, block0=2).cpu().numpy()
    ref = OI.bnn_log_posterior_blocks(OB.cast_model(m, np.float64), OB.cast_bnn(pn32, np.float64), seg, f64(x), f64(y), f64(v), f64(z), bs, seed, stream, block0=2)
    assert np.abs(got - ref).max() < 2e-5 * np.abs(ref).max() + 2e-3, np.abs(got - ref).max()
    assert np.abs(got - std).max() > 0.1                                # the prior matters here
    _lib.check(eng.lib.bgm_bnn_set_prior(eng.h, C.byref(cfg), None, None), "bgm_bnn_set_prior")
    again = eng.logpost(T(x[:, 0]), T(y[:, 0]), T(v), T(z), bs, seed, stream, block0=2).cpu().numpy()
    assert np.array_equal(again, std)                                   # cleared: back to N(0, I)
    eng.close()


@pytest.mark.parametrize("precision", ["fp32", "f16x3"])
def test_conditional_prior_mh_iterations(precision):
    from bayesgm_amd import _lib
    from bayesgm_amd.bnn_engine import flatten_bnn
    from tests.test_gpu_bnf import _engine, _panel
    m = _model(False, p=50)
    n, bs, q, k = 600, 256, 10, 10
    z, x, y, v = _panel(m, n)
    rs = np.random.RandomState(9)
    pn32 = _prior(rs, k, q, (64,), "fixed")
    seg = rs.randint(0, k, n)
    eng = _engine(m)
    eng.set_precision(precision)          # f16x3: proposal / evaluation of (item, state) units / accept step as three launches
    T = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(eng.device)
    cfg = _cfg([k, 64, q + 1])
    theta, seg_dev = T(flatten_bnn(pn32)), T(seg.astype(np.int32))
    _lib.check(eng.lib.bgm_bnn_set_prior(eng.h, C.byref(cfg), theta.data_ptr(), seg_dev.data_ptr()), "bgm_bnn_set_prior")
    seed = (9 << 32) | 4321
    state = T(z)
    acc = torch.zeros(1, dtype=torch.int32, device=eng.device)
    eng.mh_run(T(x[:, 0]), T(y[:, 0]), T(v), state, bs, it_begin=5, n_iters=2, burn_in=0, q_sd=0.3, seed=seed, row_base=1000, acc_count=acc)
    zo, m64, p64 = f64(z), OB.cast_model(m, np.float64), OB.cast_bnn(pn32, np.float64)
    n_acc, fragile = 0, np.zeros(n, bool)
    for it in (5, 6):
        zo, a, lpp, lpc = OI.bnn_mh_iteration(m64, p64, seg, f64(x), f64(y), f64(v), zo, it, 0.3, seed, bs, row_base=1000)
        n_acc += int(a.sum())
        u = OB.R.uniforms(np.arange(1000, 1000 + n), it, OB.R.TAG_ACC, seed)
        fragile |= np.abs(u - np.exp(np.minimum(lpp - lpc, 0))) < 1e-3
    got = state.cpu().numpy()
    ok = ~fragile
    assert ok.sum() >= 0.98 * n
    assert np.abs(got[ok] - zo[ok]).max() < 1e-5
    assert abs(int(acc[0]) - n_acc) <= int(fragile.sum())
    eng.close()


@pytest.mark.parametrize("units", [dict(g_units=(128, 96), e_units=(100,), f_units=(80, 40), h_units=(72,)),
                                   dict(g_units=(24, 40), e_units=(16,), f_units=(20, 12), h_units=(9, 5))])
def test_conditional_prior_outside_the_default_shapes(units):
    """The conditional prior in the log posterior and the sampler for nets of other widths (wider than 64, or narrower than the default
    shapes): the any-width path (csrc/bnw_kernels.h) reads the same per-row prior tables (identifiable.py:541-551 with any nb_units)."""
    from bayesgm_amd import _lib
    from bayesgm_amd.bnn_engine import flatten_bnn
    from tests.test_gpu_bnn import _model as model_b, _panel as panel_b, _engine as engine_b
    m = model_b(False, p=50, fixed=True, **units)
    n, bs, q, k = 300, 128, 10, 6
    z, x, y, v = panel_b(m, n)
    rs = np.random.RandomState(8)
    pn32 = _prior(rs, k, q, (64,), "fixed")
    seg = rs.randint(0, k, n)
    eng = engine_b(m, norm_mode=1, **units)
    T = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(eng.device)
    cfg = _cfg([k, 64, q + 1])
    theta, seg_dev = T(flatten_bnn(pn32)), T(seg.astype(np.int32))
    m64, p64 = OB.cast_model(m, np.float64), OB.cast_bnn(pn32, np.float64)
    seed, stream = (3 << 32) | 1234, 77
    std = eng.logpost(T(x[:, 0]), T(y[:, 0]), T(v), T(z), bs, seed, stream, block0=2).cpu().numpy()
    _lib.check(eng.lib.bgm_bnn_set_prior(eng.h, C.byref(cfg), theta.data_ptr(), seg_dev.data_ptr()), "bgm_bnn_set_prior")
    got = eng.logpost(T(x[:, 0]), T(y[:, 0]), T(v), T(z), bs, seed, stream, block0=2).cpu().numpy()
    ref = OI.bnn_log_posterior_blocks(m64, p64, seg, f64(x), f64(y), f64(v), f64(z), bs, seed, stream, block0=2)
    assert np.abs(got - ref).max() < 2e-3 * np.abs(ref).max(), np.abs(got - ref).max()
    assert np.abs(got - std).max() > 0.1                                # the prior matters here
    state = T(z)
    acc = torch.zeros(1, dtype=torch.int32, device=eng.device)
    eng.mh_run(T(x[:, 0]), T(y[:, 0]), T(v), state, bs, it_begin=5, n_iters=2, burn_in=0, q_sd=0.3, seed=seed, row_base=1000, acc_count=acc)
    zo, n_acc, fragile = f64(z), 0, np.zeros(n, bool)
    for it in (5, 6):
        zo, a, lpp, lpc = OI.bnn_mh_iteration(m64, p64, seg, f64(x), f64(y), f64(v), zo, it, 0.3, seed, bs, row_base=1000)
        n_acc += int(a.sum())
        u = OB.R.uniforms(np.arange(1000, 1000 + n), it, OB.R.TAG_ACC, seed)
        fragile |= np.abs(u - np.exp(np.minimum(lpp - lpc, 0))) < 2e-3
    ok = ~fragile
    assert ok.sum() >= 0.97 * n and np.abs(state.cpu().numpy()[ok] - zo[ok]).max() < 1e-5
    assert abs(int(acc[0]) - n_acc) <= int(fragile.sum())
    eng.close()


@pytest.mark.parametrize("units,binary", [(dict(), False), (dict(), True), (dict(g_units=(128, 96), e_units=(100,), f_units=(80, 40), h_units=(72,)), False)])
def test_conditional_prior_with_batch_statistics(units, binary):
    """The reference as written (params['bnn_norm'] = 'batch'): every input BatchNormalization -- of g, h, f and of the prior net
    (bnn.py:26 on the one-hot segments, identifiable.py:541) -- normalises with the statistics of the block of rows of the call.  For
    the prior net these are the shares of the block's rows per segment (bprior_hist_kernel, once per run); the nets run on the batch-statistics
    kernels (csrc/bnn_sample_kernels.h at the default widths, csrc/bnw_kernels.h otherwise).  Log posterior and two sampler iterations against the float64 oracle, blocks of
    128 rows with a short last block."""
    from bayesgm_amd import _lib
    from bayesgm_amd.bnn_engine import flatten_bnn
    from tests.test_gpu_bnn import _model as model_b, _panel as panel_b, _engine as engine_b
    m = model_b(binary, p=50, fixed=False, **units)
    n, bs, q, k = 300, 128, 10, 6
    z, x, y, v = panel_b(m, n)
    rs = np.random.RandomState(8)
    pn32 = _prior(rs, k, q, (64,), "batch")
    seg = rs.randint(0, k, n)
    seg[256:] = rs.randint(0, 3, n - 256)            # the short last block sees only some of the segments (zero-variance columns)
    eng = engine_b(m, norm_mode=0, **units)
    T = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(eng.device)
    cfg = _cfg([k, 64, q + 1])
    theta, seg_dev = T(flatten_bnn(pn32)), T(seg.astype(np.int32))
    m64, p64 = OB.cast_model(m, np.float64), OB.cast_bnn(pn32, np.float64)
    seed, stream = (3 << 32) | 1234, 77
    std = eng.logpost(T(x[:, 0]), T(y[:, 0]), T(v), T(z), bs, seed, stream, block0=2).cpu().numpy()
    _lib.check(eng.lib.bgm_bnn_set_prior(eng.h, C.byref(cfg), theta.data_ptr(), seg_dev.data_ptr()), "bgm_bnn_set_prior")
    got = eng.logpost(T(x[:, 0]), T(y[:, 0]), T(v), T(z), bs, seed, stream, block0=2).cpu().numpy()
    ref = OI.bnn_log_posterior_blocks(m64, p64, seg, f64(x), f64(y), f64(v), f64(z), bs, seed, stream, block0=2)
    assert np.abs(got - ref).max() < 2e-3 * np.abs(ref).max(), np.abs(got - ref).max()
    assert np.abs(got - std).max() > 0.1                                # the prior matters here
    state = T(z)
    acc = torch.zeros(1, dtype=torch.int32, device=eng.device)
    eng.mh_run(T(x[:, 0]), T(y[:, 0]), T(v), state, bs, it_begin=5, n_iters=2, burn_in=0, q_sd=0.3, seed=seed, row_base=1000, acc_count=acc)
    zo, n_acc, fragile = f64(z), 0, np.zeros(n, bool)
    for it in (5, 6):
        zo, a, lpp, lpc = OI.bnn_mh_iteration(m64, p64, seg, f64(x), f64(y), f64(v), zo, it, 0.3, seed, bs, row_base=1000)
        n_acc += int(a.sum())
        u = OB.R.uniforms(np.arange(1000, 1000 + n), it, OB.R.TAG_ACC, seed)
        fragile |= np.abs(u - np.exp(np.minimum(lpp - lpc, 0))) < 2e-3
    ok = ~fragile
    assert ok.sum() >= 0.97 * n and np.abs(state.cpu().numpy()[ok] - zo[ok]).max() < 1e-5
    assert abs(int(acc[0]) - n_acc) <= int(fragile.sum())
    # a second panel of the same size on the same session: bgm_bnn_set_prior renews the segment counts
    seg2 = rs.randint(0, k, n)
    seg2_dev = T(seg2.astype(np.int32))
    _lib.check(eng.lib.bgm_bnn_set_prior(eng.h, C.byref(cfg), theta.data_ptr(), seg2_dev.data_ptr()), "bgm_bnn_set_prior")
    got2 = eng.logpost(T(x[:, 0]), T(y[:, 0]), T(v), T(z), bs, seed, stream, block0=2).cpu().numpy()
    ref2 = OI.bnn_log_posterior_blocks(m64, p64, seg2, f64(x), f64(y), f64(v), f64(z), bs, seed, stream, block0=2)
    assert np.abs(got2 - ref2).max() < 2e-3 * np.abs(ref2).max()
    eng.close()


def _params(tmp_path, binary, **kw):
    p = dict(dataset="ident_bnn", output_dir=str(tmp_path), save_res=False, save_model=False, binary_treatment=binary, use_bnn=True,
             z_dims=[1, 1, 1, 7], v_dim=20, lr_theta=1e-3, lr_z=1e-2, g_units=[64] * 5, f_units=[64, 32, 8], h_units=[64, 32, 8], e_units=[64] * 5,
             dz_units=[64, 32, 8], kl_weight=1e-4, lr=2e-4, g_d_freq=5, use_z_rec=True, n_segments=6)
    p.update(kw)
    return p


def test_class_fit_trace_matches_oracle(tmp_path):
    """IdentifiableCausalBGM(use_bnn=True).fit(use_egm_init=False, epochs=0) = one pass of N / 32 minibatches: theta steps of g, h, f
    (oracle.bnn.theta_step), then the joint latent / prior-net step (oracle.identifiable.bnn_z_and_prior_step), on the class's own host
    random stream (segments, initial latents, permutation) and noise streams (4 call ids per minibatch).  Latent table, every network
    parameter and the prior net against the float32 oracle after three minibatches (Adam steps of 1e-3 / 1e-2: 2e-3 / 5e-3 absolute, as
    tests/test_gpu_bnn.py::test_steps_apply_adam_like_oracle)."""
    from bayesgm_amd.models import IdentifiableCausalBGM
    from bayesgm_amd.models.identifiable_bnn import IdentifiableCausalBGMBayes
    from bayesgm_amd.datasets import Sim_Hirano_Imbens_sampler
    n, B, k, seed0 = 96, 32, 6, 5
    x, y, v = Sim_Hirano_Imbens_sampler(N=n, v_dim=20, seed=0).load_all()
    model = IdentifiableCausalBGM(_params(tmp_path, False), random_seed=seed0)
    assert isinstance(model, IdentifiableCausalBGMBayes)
    mo = {kk: {"gamma": vv["gamma"].copy(), "beta": vv["beta"].copy(), "layers": [tuple(a.copy() for a in L) for L in vv["layers"]], "norm": "fixed"}
          for kk, vv in model.nets.items()}
    mo.update(z_dims=[1, 1, 1, 7], v_dim=20, binary_treatment=False)
    pn = model.prior_parameters()
    pn["norm"] = "fixed"
    np.random.seed(77)            # the host stream fit draws from (segments, initial latents, permutation), replayed below
    model.fit((x, y, v), batch_size=B, epochs=0, epochs_per_eval=1, use_egm_init=False, verbose=0)
    np.random.seed(77)
    seg = np.random.randint(0, k, size=n)
    zo = np.random.normal(0, 1, size=(n, 10)).astype('float32')
    perm = np.random.choice(n, n, replace=False)
    assert np.array_equal(seg, model.segments)
    key = model._noise_seed(per_rank=True)
    opt = {kk: AdamState(OB.flat_params(mo[kk])) for kk in ("g", "h", "f")}
    popt = AdamState(OB.flat_params(pn))
    for it in range(n // B):
        idx = perm[it * B:(it + 1) * B]
        s0 = 4 * it
        for name in ("g", "h", "f"):
            noise = OB.draw_noise(OB.net_dims(mo[name]), B, key, s0, OB.NET_ID[name])
            _, _, g = OB.theta_step(mo, name, zo[idx], x[idx], y[idx], v[idx], noise, 1e-4)
            opt[name].apply(OB.flat_params(mo[name]), OB.flat_grads(g), 1e-3)
        noises = {kk: tuple(OB.draw_noise(OB.net_dims(mo[kk]), B, key, s0 + 1 + c, OB.NET_ID[kk]) for c in (0, 1)) for kk in ("g", "h", "f")}
        pnoise = OI.bnn_prior_noise(pn, B, model._noise_seed(per_rank=False), s0 + 3)
        OI.bnn_z_and_prior_step(mo, pn, popt, zo, idx, seg[idx], x[idx], y[idx], v[idx], noises, pnoise, 1e-2, it + 1, 1e-3, 1e-4)
    assert np.abs(model.data_z.cpu().numpy() - zo).max() < 5e-3
    got = model.nets
    for name in ("g", "h", "f"):
        for a, b in zip([got[name]["gamma"], got[name]["beta"]] + [t for L in got[name]["layers"] for t in L], OB.flat_params(mo[name])):
            assert np.abs(a - b).max() < 2e-3, (name, a.shape)
    gp = model.prior_parameters()
    for a, b in zip([gp["gamma"], gp["beta"]] + [t for L in gp["layers"] for t in L], OB.flat_params(pn)):
        assert np.abs(a - b).max() < 2e-3, a.shape
    assert np.abs(gp["layers"][0][0] - IdentifiableCausalBGM(_params(tmp_path, False), random_seed=seed0).prior_parameters()["layers"][0][0]).max() > 1e-3


@pytest.mark.parametrize("binary", [False, True])
def test_model_surface(tmp_path, binary):
    """construction through IdentifiableCausalBGM(params) with use_bnn=True, EGM warm start, fit with evaluations and a checkpoint,
    get_log_posterior / metropolis_hastings_sampler with data_u, predict; a second object restores the checkpoint incl. the prior net"""
    from bayesgm_amd.models import IdentifiableCausalBGM
    from bayesgm_amd.datasets import Sim_Hirano_Imbens_sampler, binarize_treatment
    x, y, v = Sim_Hirano_Imbens_sampler(N=400, v_dim=20, seed=0).load_all()
    if binary:
        x = binarize_treatment(x)
    prm = _params(tmp_path, binary, save_model=True)
    model = IdentifiableCausalBGM(prm, timestamp="t0", random_seed=3)
    _, _, _, mv0 = model.evaluate((x, y, v))
    model.fit((x, y, v), batch_size=32, epochs=2, epochs_per_eval=1, use_egm_init=True, egm_n_iter=20, egm_batches_per_eval=10, verbose=0)
    _, mx1, my1, mv1 = model.evaluate((x, y, v), data_z=model.data_z.cpu().numpy())
    assert np.isfinite([mx1, my1, mv1]).all() and mv1 < mv0
    assert len(model.fit_history) == 3 and all(np.isfinite(h["loss_postrior_z"]) and h["kl_prior"] > 0 for h in model.fit_history)
    u = np.eye(6, dtype=np.float32)[np.random.randint(0, 6, 400)]
    z = np.random.randn(400, 10).astype(np.float32)
    lp_u = model.get_log_posterior(x, y, v, z, u)
    assert lp_u.shape == (400,) and np.isfinite(lp_u).all()
    draws, du = model.metropolis_hastings_sampler((x, y, v), q_sd=0.5, burn_in=10, n_keep=5)
    assert draws.shape == (5, 400, 10) and du.shape == (400, 6) and np.isfinite(draws).all()
    eff, interval = model.predict((x, y, v), alpha=0.05, n_mcmc=20, burn_in=20, x_values=None if binary else np.linspace(0, 3, 4), q_sd=0.5, verbose=0)
    assert eff.shape == ((400,) if binary else (4,)) and np.isfinite(eff).all() and np.isfinite(interval).all()
    eff_a, _ = model.predict((x, y, v), alpha=0.05, n_mcmc=10, burn_in=60, x_values=None if binary else np.linspace(0, 3, 4), q_sd=-1, verbose=0)
    assert np.isfinite(eff_a).all()
    again = IdentifiableCausalBGM(prm, timestamp="t0", random_seed=4)          # restores the latest checkpoint of this timestamp
    a, b = model_ckpt_prior(model), again.prior_parameters()
    if a is not None:
        assert np.array_equal(a["layers"][0][0], b["layers"][0][0])


def model_ckpt_prior(model):
    """prior parameters as saved by the model's LAST checkpoint (best epoch), or None when none was written"""
    import glob
    import os
    files = sorted(glob.glob(os.path.join(model.checkpoint_path, "ckpt-*.npz")), key=os.path.getmtime)
    if not files:
        return None
    from bayesgm_amd.bnn_engine import unflatten_bnn
    return unflatten_bnn(np.load(files[-1])["prior_theta"], model._prior_dims)


def test_class_with_other_hidden_widths(tmp_path):
    """IdentifiableCausalBGM(use_bnn=True) with g_units = [128, 128]: fit and predict through the class (the sampler with its prior on the
    any-width path)."""
    from bayesgm_amd.models import IdentifiableCausalBGM
    from bayesgm_amd.datasets import Sim_Hirano_Imbens_sampler
    x, y, v = Sim_Hirano_Imbens_sampler(N=200, v_dim=20, seed=0).load_all()
    model = IdentifiableCausalBGM(_params(tmp_path, False, g_units=[128, 128], f_units=[96, 48]), random_seed=3)
    model.fit((x, y, v), epochs=2, epochs_per_eval=1, batch_size=32, use_egm_init=True, egm_n_iter=20, egm_batches_per_eval=10, verbose=0)
    eff, interval = model.predict((x, y, v), alpha=0.05, n_mcmc=8, burn_in=8, x_values=np.linspace(0, 3, 4), q_sd=0.5, verbose=0)
    assert eff.shape == (4,) and np.isfinite(eff).all() and np.isfinite(interval).all()


def test_class_with_batch_statistics(tmp_path):
    """params['bnn_norm'] = 'batch' through the class: fit (bgm_bprior_step with batch statistics) and predict (the panel is ONE block:
    the statistics of g, h, f and of the prior net are those of all rows).  The run is deterministic and differs from the 'fixed' run."""
    from bayesgm_amd.models import IdentifiableCausalBGM
    from bayesgm_amd.datasets import Sim_Hirano_Imbens_sampler
    x, y, v = Sim_Hirano_Imbens_sampler(N=300, v_dim=20, seed=0).load_all()
    import warnings
    out = {}
    for norm in ("batch", "batch", "fixed"):
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            model = IdentifiableCausalBGM(_params(tmp_path, False, bnn_norm=norm), random_seed=3)
            np.random.seed(5)
            model.fit((x, y, v), batch_size=32, epochs=1, epochs_per_eval=1, use_egm_init=False, verbose=0)
            np.random.seed(6)
            adrf, iv = model.predict((x, y, v), n_mcmc=20, burn_in=30, x_values=[0.5, 1.0], q_sd=0.5, verbose=0)
        assert adrf.shape == (2,) and iv.shape == (2, 2) and np.isfinite(adrf).all() and np.isfinite(iv).all()
        assert 0.0 < model.last_acceptance_rate < 1.0
        out.setdefault(norm, []).append(adrf)
    np.testing.assert_array_equal(out["batch"][0], out["batch"][1])
    assert np.abs(out["batch"][0] - out["fixed"][0]).max() > 1e-6


def test_two_rank_fit_and_predict():
    """data-parallel fit (theta and prior-net gradients all-reduced: identical networks on both ranks, each holding its half of the rows)
    and the replicated predict; two ranks on this GPU over gloo (RCCL form: tests/test_gpu_rccl.py)"""
    import json
    from conftest import run_two_ranks
    r = run_two_ranks("dp_ident_bnn_smoke.py", timeout=400)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    objs = [json.loads(l) for l in r.stdout.splitlines() if l.startswith('{"rank"')]
    assert len(objs) == 2 and all(o["spread"] == 0.0 and o["finite"] for o in objs)
    assert sorted(o["rows"] for o in objs) == [602, 603] and objs[0]["adrf"] == objs[1]["adrf"]
    assert all(k > 0 for k in objs[0]["kl_prior"]) and np.all(np.isfinite(objs[0]["loss"]))
    # round 6: predict shards the ONE block of the panel by rows (mh_run(block_row0)): the two-rank result is the single-process result,
    # with a fixed and with the adaptive proposal scale (the block's acceptance counts are summed over the ranks)
    assert all(o["predict_sharded"] for o in objs)
    from bayesgm_amd.models import IdentifiableCausalBGM
    from bayesgm_amd.datasets import Sim_Hirano_Imbens_sampler
    prm = dict(dataset="dpib", output_dir="gpurun_out/dpib1", save_res=False, save_model=False, binary_treatment=False, use_bnn=True,
               z_dims=[1, 1, 1, 7], v_dim=50, lr_theta=1e-3, lr_z=1e-3, g_units=[64] * 5, f_units=[64, 32, 8], h_units=[64, 32, 8],
               e_units=[64] * 5, dz_units=[64, 32, 8], kl_weight=1e-4, lr=2e-4, g_d_freq=5, use_z_rec=True, n_segments=6)
    x, y, v = Sim_Hirano_Imbens_sampler(N=1205, v_dim=50, seed=1).load_all()
    m = IdentifiableCausalBGM(prm, random_seed=4)
    np.random.seed(5)
    a1, i1 = m.predict((x, y, v), alpha=0.05, n_mcmc=20, burn_in=20, x_values=np.linspace(0, 3, 5), q_sd=0.5, verbose=0)
    a2, _ = m.predict((x, y, v), alpha=0.05, n_mcmc=20, burn_in=120, x_values=np.linspace(0, 3, 5), q_sd=-1.0, verbose=0)
    assert float(np.asarray(m.last_q_sd).ravel()[0]) != 1.0          # the scale did adapt
    np.testing.assert_allclose(objs[0]["adrf_untrained"], a1, rtol=0, atol=2e-6)
    np.testing.assert_allclose(objs[0]["interval_untrained"], i1.ravel(), rtol=0, atol=2e-6)
    np.testing.assert_allclose(objs[0]["q_sd_adapted"], np.asarray(m.last_q_sd).ravel(), rtol=1e-6)
    np.testing.assert_allclose(objs[0]["adrf_untrained_adaptive"], a2, rtol=0, atol=2e-6)
