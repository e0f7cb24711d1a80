"""GPU parity tests of the CausalBGM fit step functions and evaluate (HIP through the C ABI vs the
NumPy oracle, whose hand-derived gradients are themselves checked against PyTorch autograd).

Tolerances (fp32 MFMA vs float64 oracle):
  gradients     |hip - ref| <= 2e-5 * max|ref| + 1e-7   per tensor
  parameters / latents after k Adam steps: Adam normalises the step to ~lr, so a gradient entry that is
                tiny relative to its tensor may move by up to lr per step with either sign; the test uses
                lr = 1e-3, 4 steps and asserts <= 3e-4 abs on weights and latents (observed ~1e-6).
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from oracle import causal as OC  # noqa: E402
from oracle import fit as OF      # noqa: E402
from tests.test_gpu_causal import _model, _data, _engine  # noqa: E402


def _flat(grads):
    return np.concatenate([np.concatenate([dW.ravel(), db.ravel()]) for dW, db in grads])


def _setup(z_dims, p, binary, n, seed):
    m = _model(seed, z_dims, p, binary)
    x, y, v = _data(n, p, seed + 1, binary)
    z = np.random.RandomState(seed + 2).randn(n, sum(z_dims)).astype(np.float32)
    return m, x, y, v, z


@pytest.mark.parametrize("case", [dict(z_dims=[1, 1, 1, 7], p=200, binary=False, n=300, B=32),
                                  dict(z_dims=[3, 3, 6, 6], p=100, binary=True, n=500, B=77),
                                  dict(z_dims=[1, 1, 1, 7], p=20, binary=False, n=700, B=600),
                                  dict(z_dims=[1, 1, 1, 7], p=20, binary=False, n=40, B=1),
                                  dict(z_dims=[1, 1, 1, 7], p=50, binary=False, n=90, B=48),       # padded shapes
                                  dict(z_dims=[2, 2, 2, 6], p=150, binary=True, n=90, B=33),
                                  # the row-tile-chain kernels (fit_chain.h): B = 16 / 32 at p in (96, 112] or (192, 208]
                                  dict(z_dims=[2, 2, 2, 6], p=100, binary=True, n=300, B=16),
                                  dict(z_dims=[1, 2, 3, 4], p=105, binary=False, n=300, B=32),
                                  dict(z_dims=[1, 1, 1, 7], p=45, binary=False, n=300, B=32),      # 13-tile kernels, masked columns
                                  # two latent input tiles (16 < q <= 19, the widest latent of the deterministic engine's sampling kernels)
                                  dict(z_dims=[3, 3, 6, 6], p=150, binary=True, n=300, B=32),
                                  dict(z_dims=[5, 5, 5, 4], p=60, binary=False, n=300, B=32)])
def test_theta_gradients_and_z_gradient_match_oracle(case):
    import torch
    m, x, y, v, z = _setup(case["z_dims"], case["p"], case["binary"], case["n"], 7)
    eng = _engine(m)
    dev = eng.device
    xd, yd, vd, zd = (torch.from_numpy(a).to(dev) for a in (x.ravel(), y.ravel(), v, z))
    B = case["B"]
    idx_np = np.random.RandomState(3).choice(case["n"], B, replace=False).astype(np.int32)
    idx = torch.from_numpy(idx_np).to(dev)
    npar = eng.fit_begin(case["n"], B)
    grad = torch.empty(npar, device=dev)
    loss = torch.zeros(8, device=dev, dtype=torch.float64)
    eng.fit_theta_grad(xd, yd, vd, zd, idx, B, grad, loss)
    m64 = OC.cast_model(m, np.float64)
    bz, bx, by, bv = (a[idx_np].astype(np.float64) for a in (z, x, y, v))
    lv, mse_v, gg, dzg = OF.g_loss_and_grads(m64, bz, bv)
    lx, aux_x, gh, dzh = OF.h_loss_and_grads(m64, bz, bx)
    ly, mse_y, gf, dzf = OF.f_loss_and_grads(m64, bz, bx, by)
    ref = np.concatenate([_flat(gg), _flat(gf), _flat(gh)])
    got = grad.cpu().numpy()
    assert got.shape == ref.shape
    # per-net tolerance relative to the largest entry of that net's gradient
    o = 0
    for part in (_flat(gg), _flat(gf), _flat(gh)):
        g_ = got[o:o + part.size]
        assert np.abs(g_ - part).max() <= 2e-5 * np.abs(part).max() + 1e-7, (np.abs(g_ - part).max(), np.abs(part).max())
        o += part.size
    l = loss.cpu().numpy()
    assert np.allclose([l[0] / B, l[2] / B, l[4] / B], [lv, lx, ly], rtol=2e-5)
    assert np.isclose(l[1] / (B * case["p"]), mse_v, rtol=2e-5) and np.isclose(l[5] / B, mse_y, rtol=2e-5)
    # z phase with lr_z so that one Adam step from zero slots moves by ~lr*sign(g): check the gradient
    # through the update of a fresh Adam state: z_new = z - lr_t * (0.1 g)/(sqrt(0.01 g^2)+eps)
    zm = torch.zeros_like(zd); zv = torch.zeros_like(zd)
    z_before = zd.clone()
    loss.zero_()
    eng.fit_z_step(xd, yd, vd, zd, zm, zv, idx, B, 1e-3, lazy=True, loss=loss)
    lz_ref, dz_ref = OF.z_loss_and_grad(m64, bz, bx, by, bv)
    assert np.isclose(loss.cpu().numpy()[6] / B, lz_ref, rtol=2e-5)
    gm = zm.cpu().numpy()[idx_np] / 0.1   # m = (1-b1) g
    assert np.abs(gm - dz_ref).max() <= 2e-5 * np.abs(dz_ref).max() + 1e-8
    untouched = np.setdiff1d(np.arange(case["n"]), idx_np)
    assert torch.equal(zd[untouched], z_before[untouched])
    eng.fit_end()


@pytest.mark.parametrize("lazy", [False, True, 2])
def test_fit_steps_match_oracle(lazy):
    """lazy = 2 (replay, csrc/z_replay.h) is checked against the oracle's DENSE recursion: it is the same optimizer."""
    import torch
    m, x, y, v, z = _setup([1, 1, 1, 7], 200, False, 96, 11)
    eng = _engine(m)
    dev = eng.device
    xd, yd, vd, zd = (torch.from_numpy(a).to(dev) for a in (x.ravel(), y.ravel(), v, z.copy()))
    zm = torch.zeros_like(zd); zv = torch.zeros_like(zd)
    B, lr = 32, 1e-3
    npar = eng.fit_begin(96, B)
    grad = torch.empty(npar, device=dev)
    st = OF.FitState(OC.cast_model(m, np.float64), z.astype(np.float64), lr, lr)
    x64, y64, v64 = (a.astype(np.float64) for a in (x, y, v))
    rs = np.random.RandomState(5)
    for step in range(4):
        idx_np = rs.choice(96, B if step < 3 else 17, replace=False).astype(np.int32)   # last batch short
        idx = torch.from_numpy(idx_np).to(dev)
        if lazy == 2:
            eng.fit_z_sync(zd, zm, zv, idx, lr)
        eng.fit_theta_grad(xd, yd, vd, zd, idx, len(idx_np), grad)
        eng.fit_theta_apply(grad, lr)
        eng.fit_z_step(xd, yd, vd, zd, zm, zv, idx, len(idx_np), lr, lazy=lazy)
        OF.fit_step(st, x64, y64, v64, idx_np, lazy_z=(lazy is True))
    if lazy == 2:
        eng.fit_z_sync(zd, zm, zv, None, lr)
    assert np.abs(zd.cpu().numpy() - st.data_z).max() <= 3e-4
    from bayesgm_amd import _lib
    for nid, key in ((_lib.NET_G, "g"), (_lib.NET_F, "f"), (_lib.NET_H, "h")):
        dims = [st.m[key][0][0].shape[0]] + [W.shape[1] for W, _ in st.m[key]]
        got = eng.get_weights(nid, dims)
        for (W, b), (Wr, br) in zip(got, st.m[key]):
            assert np.abs(W - Wr).max() <= 3e-4 and np.abs(b - br).max() <= 3e-4
    # the packed weights used by the samplers were refreshed in place: log-posterior with trained nets
    lp = eng.logpost(xd, yd, vd, zd).cpu().numpy()
    m_tr = dict(st.m)
    ref = OC.log_posterior(m_tr, x64, y64, v64, st.data_z)
    assert np.abs(lp - ref).max() <= 5e-2   # parameters differ by <=3e-4 each
    eng.fit_end()
    lp2 = eng.logpost(xd, yd, vd, zd).cpu().numpy()
    assert np.array_equal(lp, lp2)


def test_replayed_latent_adam_equals_the_dense_sweep():
    """z_adam = "replay" against z_adam = "dense" on the same minibatch sequence (N = 4096 rows, 400 minibatches of 32: a row waits
    ~128 steps between two uses, some are never used), latent phase only (fixed networks: with the theta phase in the loop the two
    runs separate like any two fp32 Adam trajectories do -- a parameter whose gradient is rounding noise moves by +-lr -- which says
    nothing about the latent optimizer; the class-level traces cover the full loop).  The gradients of a row are then the same
    function of its z in both runs, and the tables may differ only by the fp32 rounding of the deferred steps: dense rounds z after
    every step, so it random-walks (~0.3 ulp sqrt(k)) AND stagnates: once a step's update is below half an ulp of z (after ~90
    steps at lr = 1e-3) it is dropped altogether, ~1e-6 (|z| in [1, 2)) to ~2.5e-6 (|z| in [2, 4)) per waiting period that the
    replay's series keeps.  Over 400 minibatches (a row is used ~3 times) the two tables end up to 1.1e-5 apart (measured), with
    replay the closer one to the exact recursion.  Bound: 2.5e-5 absolute on z (the drift itself is up to ~1e-2 at
    lr = 1e-3), 2e-5 relative on the Adam slots.  Also: the mode guards."""
    import torch
    n, B, lr, steps = 4096, 32, 1e-3, 400
    m, x, y, v, z = _setup([1, 1, 1, 7], 200, False, n, 21)
    rs = np.random.RandomState(6)
    order = [rs.choice(n, B if k % 7 else 19, replace=False).astype(np.int32) for k in range(steps)]     # some short minibatches
    out = {}
    for mode in (0, 2):
        eng = _engine(m)
        dev = eng.device
        xd, yd, vd, zd = (torch.from_numpy(a).to(dev) for a in (x.ravel(), y.ravel(), v, z.copy()))
        zm = torch.zeros_like(zd); zv = torch.zeros_like(zd)
        eng.fit_begin(n, B)
        for k, idx_np in enumerate(order):
            idx = torch.from_numpy(idx_np).to(dev)
            if mode == 2:
                eng.fit_z_sync(zd, zm, zv, idx, lr)
            eng.fit_z_step(xd, yd, vd, zd, zm, zv, idx, len(idx_np), lr, lazy=mode)
            if mode == 2 and k == 250:
                eng.fit_z_sync(zd, zm, zv, None, lr)                 # a flush in the middle (evaluate between epochs) changes nothing
        if mode == 2:
            with pytest.raises(RuntimeError, match="fit_z_sync"):
                eng.fit_z_step(xd, yd, vd, zd, zm, zv, idx, len(idx_np), lr, lazy=2)       # rows not brought up to date first
            with pytest.raises(RuntimeError, match="flush"):
                eng.fit_z_step(xd, yd, vd, zd, zm, zv, idx, len(idx_np), lr, lazy=0)       # pending steps: no silent mode change
            eng.fit_z_sync(zd, zm, zv, None, lr)
        out[mode] = (zd.cpu().numpy(), zm.cpu().numpy(), zv.cpu().numpy())
        eng.fit_end()
    (z0, m0, v0), (z2, m2, v2) = out[0], out[2]
    assert np.abs(z0 - z).max() > 1e-3                                # the table moved
    untouched = np.setdiff1d(np.arange(n), np.concatenate(order))
    assert len(untouched) > 0 and np.array_equal(z2[untouched], z[untouched]) and np.array_equal(z0[untouched], z[untouched])
    assert np.abs(z2 - z0).max() <= 2.5e-5, np.abs(z2 - z0).max()
    assert np.median(np.abs(z2 - z0)) <= 3e-7
    assert np.abs(m2 - m0).max() <= 2e-5 * np.abs(m0).max() and np.abs(v2 - v0).max() <= 2e-5 * np.abs(v0).max()


@pytest.mark.parametrize("binary", [False, True])
def test_evaluate_matches_oracle(binary):
    import torch
    z_dims, p = ([3, 3, 6, 6], 100) if binary else ([1, 1, 1, 7], 200)
    m, x, y, v, z = _setup(z_dims, p, binary, 450, 13)
    eng = _engine(m)
    dev = eng.device
    xd, yd, vd, zd = (torch.from_numpy(a).to(dev) for a in (x.ravel(), y.ravel(), v, z))
    m64 = OC.cast_model(m, np.float64)
    ref_c, mx, my, mv = OC.evaluate(m64, (x.astype(np.float64), y.astype(np.float64), v.astype(np.float64)),
                                    z.astype(np.float64), nb_intervals=30)
    if binary:
        sums, ite = eng.evaluate(xd, yd, vd, zd)
        causal = ite.cpu().numpy().reshape(-1, 1)
    else:
        xs = np.linspace(OC.percentile_nearest(x, 5.0), OC.percentile_nearest(x, 95.0), 30).astype(np.float32)
        sums, dose = eng.evaluate(xd, yd, vd, zd, xs)
        causal = dose.cpu().numpy() / 450
    s = sums.cpu().numpy()
    assert np.isclose(s[0] / (450 * p), mv, rtol=1e-5) and np.isclose(s[1] / 450, mx, rtol=1e-5)
    assert np.isclose(s[2] / 450, my, rtol=1e-5)
    assert np.abs(causal - ref_c).max() <= 2e-5 * max(1.0, np.abs(ref_c).max())


def test_causalbgm_fit_predict_end_to_end(tmp_path):
    """Drop-in surface: fit (no EGM) lowers the losses, evaluate/predict return the reference's shapes/types."""
    from bayesgm_amd.models import CausalBGM
    from bayesgm_amd.datasets import Sim_Hirano_Imbens_sampler
    x, y, v = Sim_Hirano_Imbens_sampler(N=2000, v_dim=200, seed=0).load_all()
    params = dict(dataset="t", output_dir=str(tmp_path), save_res=True, save_model=True, binary_treatment=False,
                  use_bnn=False, z_dims=[1, 1, 1, 7], v_dim=200, lr_theta=1e-3, lr_z=1e-3, g_units=[64] * 5,
                  f_units=[64, 32, 8], h_units=[64, 32, 8], e_units=[64] * 5, dz_units=[64, 32, 8], kl_weight=1e-4,
                  lr=2e-4, g_d_freq=5, use_z_rec=True)
    model = CausalBGM(params, random_seed=1)
    assert type(model).__name__ == "CausalBGM"
    assert type(CausalBGM(dict(params, use_bnn=True, save_res=False, save_model=False))).__name__ == "CausalBGMBayes"
    c0, mx0, my0, mv0 = model.evaluate((x, y, v))
    model.fit((x, y, v), epochs=6, epochs_per_eval=3, batch_size=32, use_egm_init=False, verbose=0)
    c1, mx1, my1, mv1 = model.evaluate((x, y, v), data_z=model.data_z.cpu().numpy())
    assert c1.shape == (200,) and isinstance(mx1, np.float32)
    assert my1 < my0 and mv1 < mv0
    assert model.best_epoch in (0, 3, 6)
    import os
    assert os.path.exists(os.path.join(model.save_dir, "causal_pre_at_3.txt"))
    assert os.path.exists(os.path.join(model.save_dir, "params.txt"))
    adrf, interval = model.predict((x[:300], y[:300], v[:300]), alpha=0.05, n_mcmc=40, burn_in=60,
                                   x_values=[0.5, 1.0, 2.0], verbose=0)
    assert adrf.shape == (3,) and interval.shape == (3, 2) and np.all(interval[:, 0] <= interval[:, 1])
    with pytest.raises(ValueError):
        model.predict((x[:10], y[:10], v[:10]), n_mcmc=5, burn_in=5)   # continuous needs x_values
    with pytest.raises(AssertionError):
        model.predict((x[:10], y[:10], v[:10]), alpha=1.5, x_values=[1.0])


def test_causalbgm_default_fit_with_egm_warm_start(tmp_path):
    """fit() with its default use_egm_init=True: the native EGM warm start (egm_kernels.h) trains g,e,f,h,
    Z is initialised as e(V) by the HIP encoder, and the iterative updates continue from there."""
    from bayesgm_amd.models import CausalBGM
    from bayesgm_amd.datasets import Sim_Hirano_Imbens_sampler
    x, y, v = Sim_Hirano_Imbens_sampler(N=1500, v_dim=200, seed=1).load_all()
    params = dict(dataset="t", output_dir=str(tmp_path), save_res=True, save_model=False, binary_treatment=False,
                  use_bnn=False, z_dims=[1, 1, 1, 7], v_dim=200, lr_theta=1e-4, lr_z=1e-4, g_units=[64] * 5,
                  f_units=[64, 32, 8], h_units=[64, 32, 8], e_units=[64] * 5, dz_units=[64, 32, 8], kl_weight=1e-4,
                  lr=2e-4, g_d_freq=5, use_z_rec=True)
    model = CausalBGM(params, random_seed=2)
    _, _, my0, _ = model.evaluate((x, y, v))
    model.fit((x, y, v), epochs=1, epochs_per_eval=1, egm_n_iter=400, egm_batches_per_eval=200, verbose=0)
    _, _, my1, _ = model.evaluate((x, y, v), data_z=model.data_z.cpu().numpy())
    assert my1 < my0
    import os
    assert os.path.exists(os.path.join(model.save_dir, "causal_pre_egm_init_iter-400.txt"))
    # Z was initialised by the encoder: data_z stays close to e(V) after one epoch at lr_z = 1e-4
    z_enc = model.engine.encode(model._dev(v)).cpu().numpy()
    assert np.abs(model.data_z.cpu().numpy() - z_enc).mean() < 0.5


def test_large_batch_gradient_is_the_sum_of_its_shards():
    """Size-independent property at a throughput-sized minibatch (B = 65 536 rows of a 200 000-row panel, p = 200):
    the theta gradient of a batch equals the sum of the gradients of its two halves when both are scaled by the
    GLOBAL batch size -- the identity the data-parallel fit relies on (each rank computes its local rows with
    1/B_global, one all-reduce adds them) -- and is reproducible bit for bit."""
    import torch
    n, p, B = 200_000, 200, 65_536
    m = _model(17, [1, 1, 1, 7], p)
    rs = np.random.RandomState(18)
    v = rs.standard_normal((n, p)).astype(np.float32)
    x = rs.exponential(size=n).astype(np.float32)
    y = (x + rs.standard_normal(n)).astype(np.float32)
    z = rs.standard_normal((n, 10)).astype(np.float32)
    eng = _engine(m)
    dev = eng.device
    xd, yd, vd, zd = (torch.from_numpy(a).to(dev) for a in (x, y, v, z))
    idx = torch.from_numpy(rs.choice(n, B, replace=False).astype(np.int32)).to(dev)
    npar = eng.fit_begin(n, B)
    g_all, g_again, g_a, g_b = (torch.empty(npar, device=dev) for _ in range(4))
    eng.fit_theta_grad(xd, yd, vd, zd, idx, B, g_all)
    eng.fit_theta_grad(xd, yd, vd, zd, idx, B, g_again)
    assert torch.equal(g_all, g_again)                                   # deterministic reduction order
    eng.fit_theta_grad(xd, yd, vd, zd, idx[:B // 2].contiguous(), B, g_a, batch=B // 2)
    eng.fit_theta_grad(xd, yd, vd, zd, idx[B // 2:].contiguous(), B, g_b, batch=B // 2)
    ref = g_all.double()
    err = (g_a.double() + g_b.double() - ref).abs().max().item()
    assert err <= 2e-5 * ref.abs().max().item(), err
    assert torch.isfinite(g_all).all() and g_all.abs().max().item() > 0
    eng.fit_end()


def test_two_rank_causal_fit_and_predict_run():
    """The data-parallel code paths of CausalBGM (row shards, fused gradient all-reduce, Z rows local, ADRF all-reduce)
    executed for real: two ranks on this GPU over gloo end with bit-identical networks, ADRF and intervals."""
    import os, subprocess, sys
    from conftest import run_two_ranks
    r = run_two_ranks("dp_causal_smoke.py")
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert r.stdout.count('"spread": 0.0') == 2, r.stdout[-2000:]
    # the sharded predict equals the single-process predict of the same (seeded, untrained) model: chains are keyed by the
    # global row index, only the order of the ADRF partial sums differs
    import json
    from bayesgm_amd.models import CausalBGM
    from bayesgm_amd.datasets import Sim_Hirano_Imbens_sampler
    # the two ranks print concurrently: their JSON objects may share a line
    two = json.JSONDecoder().raw_decode(r.stdout[r.stdout.index('{"rank": 0'):])[0]
    x, y, v = Sim_Hirano_Imbens_sampler(N=1505, v_dim=50, seed=1).load_all()
    params = dict(dataset="dp", output_dir="gpurun_out/dp", save_res=False, save_model=False, binary_treatment=False, use_bnn=False,
                  z_dims=[1, 1, 1, 7], v_dim=50, lr_theta=1e-3, lr_z=1e-3, g_units=[64] * 5, f_units=[64, 32, 8], h_units=[64, 32, 8],
                  e_units=[64] * 5, dz_units=[64, 32, 8], kl_weight=1e-4, lr=2e-4, g_d_freq=5, use_z_rec=True)
    m = CausalBGM(params, random_seed=2)
    adrf, interval = m.predict((x, y, v), alpha=0.05, n_mcmc=40, burn_in=40, x_values=np.linspace(0, 3, 6), q_sd=1.0, verbose=0)
    assert np.abs(np.array(two["adrf_untrained"]) - adrf).max() <= 1e-5
    assert np.abs(np.array(two["interval_untrained"]) - interval.ravel()).max() <= 1e-5
