"""Bayesian-network (use_bnn=True) kernels against oracle/bnn.py through the C ABI."""
import numpy as np
import pytest
import torch

from oracle import bnn as OB

pytestmark = pytest.mark.gpu


def _model(binary, z_dims=(1, 1, 1, 7), p=50, seed=0, fixed=False, **units):
    """fixed=True: inference-mode input normalisation (the models' default); at p = 100 / 200 with the default units the minibatch
    steps then run as row-tile chains (egm_chain_bnn.h)."""
    m = OB.init_model(seed, list(z_dims), p, binary, **units)
    if fixed:
        for k in ("g", "e", "f", "h"):
            m[k]["norm"] = "fixed"
    rs = np.random.RandomState(seed + 11)
    for k in ("g", "e", "f", "h"):
        m[k]["gamma"] = (1.0 + 0.2 * rs.standard_normal(m[k]["gamma"].shape)).astype(np.float32)
        m[k]["beta"] = (0.1 * rs.standard_normal(m[k]["beta"].shape)).astype(np.float32)
    return m


def _panel(m, n, seed=1):
    rs = np.random.RandomState(seed)
    q = sum(m["z_dims"])
    z = rs.standard_normal((n, q)).astype(np.float32)
    v = rs.standard_normal((n, m["v_dim"])).astype(np.float32)
    x = ((rs.rand(n, 1) > 0.5) if m["binary_treatment"] else rs.exponential(size=(n, 1))).astype(np.float32)
    y = (x + rs.standard_normal((n, 1))).astype(np.float32)
    return z, x, y, v


def _engine(m, max_batch=32, kl_weight=1e-4, **units):
    from bayesgm_amd.bnn_engine import BnnEngine
    eng = BnnEngine(m["v_dim"], m["z_dims"], m["binary_treatment"], kl_weight=kl_weight, max_batch=max_batch,
                    sigma_v=m.get("sigma_v"), sigma_x=m.get("sigma_x"), sigma_y=m.get("sigma_y"), **units)
    eng.begin(m)
    return eng


def _rel(a, b):
    return float(np.abs(a - b).max() / (np.abs(b).max() + 1e-30))


ZD = (1, 1, 1, 7)
ACIC = (3, 6, 3, 6)       # Semi_acic.yaml: q = 18, two latent input tiles on the chains


@pytest.mark.parametrize("binary,B,units,p,zd", [
    (False, 32, {}, 50, ZD),
    (True, 32, {}, 50, ZD),
    (False, 19, dict(g_units=(24, 40), e_units=(16,), f_units=(20, 12), h_units=(9, 5)), 37, ZD),
    (False, 32, {}, 200, ZD), (True, 32, {}, 100, ZD), (False, 16, {}, 100, ZD),       # row-tile chains
    (False, 32, {}, 45, ZD),                                                     # ... the 13-tile kernels on a narrower panel (masked columns)
    (True, 32, {}, 177, ACIC), (False, 32, {}, 100, (5, 5, 5, 5)),               # ... with two latent input tiles (Semi_acic, Sim_Colangelo)
    (False, 100, {}, 50, ZD), (True, 160, {}, 200, ZD),                          # minibatches beyond 64 rows (any batch_size, causalbgm/base.py:434): batch statistics / fixed normalisation
    (False, 300, {}, 50, ZD), (True, 520, {}, 100, ZD),                          # ... and beyond 256
])
def test_theta_step_gradients_match_oracle(binary, B, units, p, zd):
    chain = p >= 100 or p == 45
    m = _model(binary, z_dims=zd, p=p, fixed=chain, **units)
    z, x, y, v = _panel(m, max(200, B + 40))
    eng = _engine(m, max_batch=max(32, B), kl_weight=0.01, **(dict(norm_mode=1) if chain else {}), **units)
    dev = eng.device
    rs = np.random.RandomState(4)
    idx = rs.choice(max(200, B + 40), B, replace=False).astype(np.int32)
    seed, stream = (5 << 32) | 77, 12
    out = torch.zeros(8, device=dev)
    T = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    eng.theta_step(T(z), T(idx), T(x[:, 0]), T(y[:, 0]), T(v), 1e-3, seed, stream, apply=False, out=out)
    grad = eng.split(eng.read(1))
    out = out.cpu().numpy()
    m64 = OB.cast_model(m, np.float64)
    for w, name in enumerate(("g", "h", "f")):
        dims = OB.net_dims(m[name])
        noise = OB.draw_noise(dims, B, seed, stream, OB.NET_ID[name], dtype=np.float64)
        loss, aux, g = OB.theta_step(m64, name, z[idx].astype(np.float64), x[idx].astype(np.float64), y[idx].astype(np.float64),
                                     v[idx].astype(np.float64), noise, 0.01)
        assert abs(out[2 * w] - loss) < 2e-4 * max(1.0, abs(loss)), (name, out[2 * w], loss)
        assert abs(out[2 * w + 1] - aux) < 2e-4 * max(1.0, abs(aux)), name
        got = [grad[name]["gamma"], grad[name]["beta"]] + [a for L in grad[name]["layers"] for a in L]
        for a, b in zip(got, OB.flat_grads(g)):
            assert _rel(a, b) < 2e-3, (name, a.shape, _rel(a, b))
    eng.close()


@pytest.mark.parametrize("binary,p,B,zd", [(False, 50, 32, ZD), (True, 50, 32, ZD), (False, 200, 32, ZD), (True, 100, 32, ZD), (False, 100, 16, ZD),
                                           (True, 45, 32, ZD), (True, 177, 32, ACIC), (False, 100, 32, (5, 5, 5, 5)),
                                           (False, 50, 96, ZD), (True, 200, 100, ZD),           # minibatches beyond 64 rows
                                           (False, 50, 300, ZD), (True, 100, 520, ZD)])         # ... and beyond 256
def test_z_step_gradient_matches_oracle(binary, p, B, zd):
    chain = p >= 100 or p == 45
    m = _model(binary, z_dims=zd, p=p, fixed=chain)
    z, x, y, v = _panel(m, max(100, B))
    eng = _engine(m, max_batch=max(32, B), norm_mode=1) if chain else _engine(m, max_batch=max(32, B))
    dev = eng.device
    idx = np.random.RandomState(9).choice(max(100, B), B, replace=False).astype(np.int32)
    seed, stream = 123456789, 40
    T = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    out = torch.zeros(4, device=dev)
    dz = torch.zeros(B, sum(m["z_dims"]), device=dev)
    eng.z_step(T(x[:, 0]), T(y[:, 0]), T(v), T(z), None, None, T(idx), 1e-3, seed, stream, out=out, dz_out=dz)
    m64 = OB.cast_model(m, np.float64)
    noises = {k: tuple(OB.draw_noise(OB.net_dims(m[k]), B, seed, stream + c, OB.NET_ID[k], dtype=np.float64) for c in (0, 1))
              for k in ("g", "h", "f")}
    loss, ref = OB.z_step(m64, z[idx].astype(np.float64), x[idx].astype(np.float64), y[idx].astype(np.float64),
                          v[idx].astype(np.float64), noises)
    assert abs(float(out[0]) - loss) < 2e-4 * abs(loss)
    assert _rel(dz.cpu().numpy(), ref) < 2e-3
    eng.close()


@pytest.mark.parametrize("p", [50, 100])
def test_steps_apply_adam_like_oracle(p):
    """Three full minibatch iterations (theta step + latent step, dense-decay Adam on the table) track the oracle
    (p = 100 with inference-mode input normalisation: the row-tile-chain kernels)."""
    from oracle.fit import AdamState, adam_lr_t
    chain = p >= 100
    m = _model(False, p=p, fixed=chain)
    n, B = 64, 32
    z, x, y, v = _panel(m, n)
    eng = _engine(m, norm_mode=1) if chain else _engine(m)
    dev = eng.device
    T = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    dz_t, zm, zv = T(z), torch.zeros(n, z.shape[1], device=dev), torch.zeros(n, z.shape[1], device=dev)
    xs, ys, vs = T(x[:, 0]), T(y[:, 0]), T(v)
    seed = 99
    mo = OB.cast_model(m, np.float32)
    zo = z.copy()
    opt = {k: AdamState(OB.flat_params(mo[k])) for k in ("g", "h", "f")}
    zm_o, zv_o = np.zeros_like(zo), np.zeros_like(zo)
    rs = np.random.RandomState(3)
    for it in range(3):
        idx = rs.choice(n, B, replace=False).astype(np.int32)
        eng.theta_step(dz_t, T(idx), xs, ys, vs, 1e-3, seed, 4 * it, apply=True)
        eng.z_step(xs, ys, vs, dz_t, zm, zv, T(idx), 1e-2, seed, 4 * it + 1)
        for name in ("g", "h", "f"):
            noise = OB.draw_noise(OB.net_dims(mo[name]), B, seed, 4 * it, OB.NET_ID[name])
            _, _, g = OB.theta_step(mo, name, zo[idx], x[idx], y[idx], v[idx], noise, 1e-4)
            opt[name].apply(OB.flat_params(mo[name]), OB.flat_grads(g), 1e-3)
        noises = {k: tuple(OB.draw_noise(OB.net_dims(mo[k]), B, seed, 4 * it + 1 + c, OB.NET_ID[k]) for c in (0, 1))
                  for k in ("g", "h", "f")}
        _, dz = OB.z_step(mo, zo[idx], x[idx], y[idx], v[idx], noises)
        lr_t = np.float32(adam_lr_t(1e-2, it + 1))
        zm_o *= np.float32(0.9); zv_o *= np.float32(0.99)
        zm_o[idx] += np.float32(0.1) * dz; zv_o[idx] += np.float32(0.01) * dz * dz
        zo -= lr_t * zm_o / (np.sqrt(zv_o) + np.float32(1e-7))
    got = eng.split(eng.read(0))
    for name in ("g", "h", "f"):
        for a, b in zip([got[name]["gamma"], got[name]["beta"]] + [t for L in got[name]["layers"] for t in L], OB.flat_params(mo[name])):
            assert np.abs(a - b).max() < 2e-3, (name, a.shape, np.abs(a - b).max())   # 3 Adam steps of lr 1e-3
    assert np.abs(dz_t.cpu().numpy() - zo).max() < 5e-3
    eng.close()


def test_replayed_latent_adam_equals_the_dense_sweep():
    """bgm_bnn_z_sync + lazy = 2 against lazy = 0 over 120 minibatches of a 1024-row table with the same noise streams: the tables
    differ by the fp32 rounding of the deferred zero-gradient steps only (bound as in test_gpu_fit.py)."""
    m = _model(False, p=100, fixed=True)
    n, B, steps = 1024, 32, 120
    z, x, y, v = _panel(m, n)
    rs = np.random.RandomState(4)
    order = [rs.choice(n, B, replace=False).astype(np.int32) for _ in range(steps)]
    out = {}
    for mode in (0, 2):
        eng = _engine(m, norm_mode=1)
        dev = eng.device
        T = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
        zt, zm, zv = T(z), torch.zeros(n, z.shape[1], device=dev), torch.zeros(n, z.shape[1], device=dev)
        xs, ys, vs = T(x[:, 0]), T(y[:, 0]), T(v)
        for it, idx in enumerate(order):
            if mode == 2:
                eng.z_sync(zt, zm, zv, T(idx), 1e-3)
            eng.z_step(xs, ys, vs, zt, zm, zv, T(idx), 1e-3, 7, 4 * it + 1, lazy=mode)      # latent phase only, see test_gpu_fit.py
        if mode == 2:
            with pytest.raises(RuntimeError, match="z_sync"):
                eng.z_step(xs, ys, vs, zt, zm, zv, T(idx), 1e-3, 7, 1, lazy=2)
            eng.z_sync(zt, zm, zv, None, 1e-3)
        out[mode] = (zt.cpu().numpy(), zm.cpu().numpy(), zv.cpu().numpy())
        eng.close()
    (z0, m0, v0), (z2, m2, v2) = out[0], out[2]
    assert np.abs(z0 - z).max() > 1e-3
    assert np.abs(z2 - z0).max() <= 5e-6, np.abs(z2 - z0).max()
    assert np.abs(m2 - m0).max() <= 2e-5 * np.abs(m0).max() and np.abs(v2 - v0).max() <= 2e-5 * np.abs(v0).max()


@pytest.mark.parametrize("binary,p,z_dims,n,bs", [
    (False, 200, (1, 1, 1, 7), 700, 300),      # ragged last block, several workgroups per block
    (True, 100, (3, 3, 6, 6), 520, 520),
    (False, 37, (2, 1, 2, 3), 100, 64),
])
def test_logpost_blocks_match_oracle(binary, p, z_dims, n, bs):
    m = _model(binary, z_dims=z_dims, p=p)
    z, x, y, v = _panel(m, n)
    eng = _engine(m)
    T = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(eng.device)
    seed, stream = (3 << 32) | 1234, 77
    got = eng.logpost(T(x[:, 0]), T(y[:, 0]), T(v), T(z), bs, seed, stream, block0=2).cpu().numpy()
    m64 = OB.cast_model(m, np.float64)
    ref = OB.log_posterior_blocks(m64, x.astype(np.float64), y.astype(np.float64), v.astype(np.float64), z.astype(np.float64),
                                  bs, seed, stream, block0=2)
    assert np.abs(got - ref).max() < 2e-3 * np.abs(ref).max(), np.abs(got - ref).max()
    eng.close()


def test_mh_iterations_match_oracle():
    m = _model(False, p=50)
    n, bs = 600, 256
    z, x, y, v = _panel(m, n)
    eng = _engine(m)
    T = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(eng.device)
    seed = (9 << 32) | 4321
    state = T(z)
    acc = torch.zeros(1, dtype=torch.int32, device=eng.device)
    eng.mh_run(T(x[:, 0]), T(y[:, 0]), T(v), state, bs, it_begin=5, n_iters=2, burn_in=0, q_sd=0.3, seed=seed, row_base=1000,
               acc_count=acc)
    zo = z.astype(np.float64)
    m64 = OB.cast_model(m, np.float64)
    n_acc, fragile = 0, np.zeros(n, bool)
    for it in (5, 6):
        zo, a, lpp, lpc = OB.mh_iteration(m64, x.astype(np.float64), y.astype(np.float64), v.astype(np.float64), zo, it, 0.3,
                                          seed, bs, row_base=1000)
        n_acc += int(a.sum())
        u = OB.R.uniforms(np.arange(1000, 1000 + n), it, OB.R.TAG_ACC, seed)
        fragile |= np.abs(u - np.exp(np.minimum(lpp - lpc, 0))) < 1e-3     # accept decisions within fp32 noise
    got = state.cpu().numpy()
    ok = ~fragile
    assert ok.sum() >= 0.98 * n          # rows whose accept decision lies within fp32 noise of u: measured 4 of 600 (expected 2 x 2e-3 x n)
    assert np.abs(got[ok] - zo[ok]).max() < 1e-5
    assert abs(int(acc[0]) - n_acc) <= int(fragile.sum())
    eng.close()


@pytest.mark.parametrize("binary,units", [(False, {}), (True, {}),
                                          (False, dict(f_units=(20, 12))),          # pipelined dose loop, run-time layer extents
                                          (True, dict(f_units=(64, 64, 64)))])      # outcome net wider than one staged chunk: the generic routine
def test_mh_effects_match_oracle(binary, units):
    """Kept draws + causal effects of the sampler against the oracle evaluated on the kernel's own draws.  The default outcome net
    runs the effects kernel's compile-time-shaped dose loop; the two other shapes its run-time-shaped and its generic form."""
    m = _model(binary, p=50, **units)
    n, bs, burn, keep = 300, 128, 2, 3
    z, x, y, v = _panel(m, n)
    eng = _engine(m, **units)
    dev = eng.device
    T = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    seed = (2 << 32) | 555
    state = torch.zeros(n, z.shape[1], device=dev)
    draws = torch.zeros(keep, n, z.shape[1], device=dev)
    xs = np.array([0.0, 0.7, 1.5, 2.2, 3.0], np.float32)
    adrf = torch.zeros(len(xs), keep, device=dev, dtype=torch.float64)
    ite = torch.zeros(n, keep, device=dev)
    eng.mh_run(T(x[:, 0]), T(y[:, 0]), T(v), state, bs, 0, burn + keep, burn, 0.4, seed, init=True, row_base=40, block0=1,
               draws=draws, n_keep=keep, effect=2 if binary else 1, sample_y=True, x_values=None if binary else T(xs),
               adrf_sum=None if binary else adrf, ite=ite if binary else None)
    dr = draws.cpu().numpy()
    # the stand-alone form on the kept draws gives the same effects as the fused pass
    alone = eng.effects(draws, bs, seed, it0=burn, x_values=None if binary else xs, sample_y=True, row_base=40, block0=1).cpu().numpy()
    fused = ite.t().cpu().numpy() if binary else (adrf / n).float().cpu().numpy()
    assert np.abs(alone - fused).max() < 1e-5
    init = OB.R.normals(np.arange(40, 40 + n), 0, z.shape[1], OB.R.TAG_INIT, seed)
    assert np.abs(dr[-1] - state.cpu().numpy()).max() == 0.0
    assert np.abs(dr[0] - init).max() < 10.0 and np.abs(dr[0] - dr[-1]).max() > 0.0
    m64 = OB.cast_model(m, np.float64)
    for d in range(keep):
        ref = OB.effects_draw(m64, dr[d].astype(np.float64), [1.0, 0.0] if binary else xs.astype(np.float64), d, burn + d, True, seed,
                              bs, block0=1, row_base=40)
        if binary:
            assert np.abs(ite[:, d].cpu().numpy() - (ref[0] - ref[1])).max() < 2e-3
        else:
            assert np.abs(adrf[:, d].cpu().numpy() / n - ref.mean(axis=1)).max() < 5e-4
    eng.close()


@pytest.mark.parametrize("binary,p", [(False, 200), (True, 37)])
def test_evaluate_matches_oracle(binary, p):
    m = _model(binary, p=p)
    n = 700
    z, x, y, v = _panel(m, n)
    eng = _engine(m)
    T = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(eng.device)
    xs = np.linspace(0.1, 2.5, 7).astype(np.float32)
    seed, stream = 31337, 900
    zt, sums, causal = eng.evaluate(T(x[:, 0]), T(y[:, 0]), T(v), None, x_values=xs, seed=seed, stream_id=stream)
    m64 = OB.cast_model(m, np.float64)
    zr, cr, mse_x, mse_y, mse_v = OB.evaluate(m64, (x.astype(np.float64), y.astype(np.float64), v.astype(np.float64)), None, xs, seed, stream)
    assert np.abs(zt.cpu().numpy() - zr).max() < 2e-3 * max(1.0, np.abs(zr).max())
    s = sums.cpu().numpy()
    assert abs(s[0] / (n * p) - mse_v) < 1e-3 * mse_v and abs(s[1] / n - mse_x) < 2e-3 * mse_x and abs(s[2] / n - mse_y) < 2e-3 * mse_y
    if binary:
        assert np.abs(causal.cpu().numpy() - cr).max() < 2e-3
    else:
        assert np.abs(causal.cpu().numpy() / n - cr).max() < 5e-4
    # with the latents given
    _, sums2, _ = eng.evaluate(T(x[:, 0]), T(y[:, 0]), T(v), T(z), x_values=xs, seed=seed, stream_id=stream)
    _, _, mx2, my2, mv2 = OB.evaluate(m64, (x.astype(np.float64), y.astype(np.float64), v.astype(np.float64)), z.astype(np.float64), xs, seed, stream)
    s2 = sums2.cpu().numpy()
    assert abs(s2[0] / (n * p) - mv2) < 1e-3 * mv2 and abs(s2[2] / n - my2) < 2e-3 * my2
    eng.close()


@pytest.mark.parametrize("p,B", [(100, 32), (100, 16), (50, 4)])
def test_egm_split_step_equals_fused_step(p, B):
    """Data-parallel form of the warm start with Bayesian nets: step(apply = 0) + bgm_bnn_egm_grad + bgm_bnn_egm_apply takes the Adam
    steps of the fused step (chains at p = 100; B = 4: a rank's share of a 32-row minibatch at 8 GPUs, the phase-machine kernels)."""
    from oracle import egm as OE
    res = []
    for split in (False, True):
        m = _model(False, z_dims=ZD, p=p)
        n = 120
        _, x, y, v = _panel(m, n)
        q = sum(m["z_dims"])
        rs = np.random.RandomState(33)
        dz = OE.init_disc(rs, q, [64, 32, 8])
        dz["fixed_norm"] = True
        for k in ("g", "e", "f", "h"):
            m[k]["norm"] = "fixed"
        eng = _engine(m, norm_mode=1)
        dev = eng.device
        T = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
        eng.set_disc_norm("fixed")
        eng.egm_begin(dz, B, 2e-4, 1)
        n_gen, n_dz = eng.egm_sizes()
        bg, bd = torch.empty(n_gen, device=dev), torch.empty(n_dz, device=dev)
        seed, stream = (1 << 32) | 7, 500
        for it in range(3):
            for _ in range(2):
                z = rs.standard_normal((B, q)).astype(np.float32)
                idx = rs.choice(n, B, replace=False).astype(np.int32)
                eng.egm_disc_step(T(z), T(idx), T(v), float(rs.rand()), seed, stream, apply=not split)
                stream += 1
                if split:
                    eng.egm_grad(1, 1.0, bd)
                    eng.egm_apply(1, bd)
            z = rs.standard_normal((B, q)).astype(np.float32)
            idx = rs.choice(n, B, replace=False).astype(np.int32)
            eng.egm_gen_step(T(z), T(idx), T(v), T(x[:, 0]), T(y[:, 0]), seed, stream, apply=not split)
            stream += 9
            if split:
                eng.egm_grad(0, 1.0, bg)
                eng.egm_apply(0, bg)
        res.append((eng.read(0).copy(), eng.egm_read(1).copy()))
        eng.egm_end()
        eng.close()
    (t0, d0), (t1, d1) = res
    # (one ulp of a parameter of magnitude ~3: the two Adam expressions may be contracted differently)
    assert np.abs(t0 - t1).max() <= 5e-7 and np.abs(d0 - d1).max() <= 1e-7, (np.abs(t0 - t1).max(), np.abs(d0 - d1).max())


@pytest.mark.parametrize("binary,disc_norm,p,zd", [(False, "batch", 50, ZD), (True, "batch", 50, ZD), (False, "fixed", 50, ZD), (True, "fixed", 50, ZD),
                                                   (False, "fixed", 100, ZD), (True, "fixed", 200, ZD), (False, "fixed", 45, ZD),
                                                   (True, "fixed", 177, ACIC), (False, "fixed", 100, (5, 5, 5, 5))])
def test_egm_steps_match_oracle(binary, disc_norm, p, zd):
    """EGM warm-start steps with Bayesian nets: gradients of the discriminator step and of the nine-call generator step.
    disc_norm = "fixed" (the models' default): the discriminator passes of the step run as register-chained row tiles; at
    p = 100 / 200 (with the inference-mode input normalisation of `_model`, when it has it) so does the Flipout encoder call."""
    from oracle import egm as OE
    from bayesgm_amd.engine import CausalEngine
    m = _model(binary, z_dims=zd, p=p)
    n, B = 120, 32
    _, x, y, v = _panel(m, n)
    q = sum(m["z_dims"])
    rs = np.random.RandomState(21)
    dz = OE.init_disc(rs, q, [64, 32, 8])
    for l in range(3):
        dz["gamma"][l] = (1.0 + 0.2 * rs.standard_normal(dz["gamma"][l].shape)).astype(np.float32)
        dz["beta"][l] = (0.1 * rs.standard_normal(dz["beta"][l].shape)).astype(np.float32)
    if p >= 100 or p == 45:          # the models' default input normalisation as well (bnn_norm = "fixed")
        for k in ("g", "e", "f", "h"):
            m[k]["norm"] = "fixed"
    eng = _engine(m, norm_mode=1) if (p >= 100 or p == 45) else _engine(m)
    dev = eng.device
    T = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    if disc_norm == "fixed":
        eng.set_disc_norm("fixed")
        dz["fixed_norm"] = True
    eng.egm_begin(dz, B, 2e-4, 1)
    z = rs.standard_normal((B, q)).astype(np.float32)
    idx = rs.choice(n, B, replace=False).astype(np.int32)
    seed, stream, eps = (1 << 32) | 42, 1000, 0.37
    m64, dz64 = OB.cast_model(m, np.float64), OE.cast_disc(dz, np.float64)
    f64 = lambda a: a.astype(np.float64)
    # discriminator step
    out = torch.zeros(2, device=dev)
    eng.egm_disc_step(T(z), T(idx), T(v), eps, seed, stream, apply=False, out=out)
    noises = OB.egm_noises(m64, B, seed, stream, np.float64, disc_only=True)
    dl, dtot, gd = OB.egm_disc_step_grads(m64, dz64, f64(z), f64(v[idx]), eps, noises)
    o = out.cpu().numpy()
    assert abs(o[0] - dl) < 1e-4 * max(1.0, abs(dl)) and abs(o[1] - dtot) < 1e-3 * max(1.0, abs(dtot))
    ref = CausalEngine.flatten_disc(gd).astype(np.float64)
    got = eng.egm_read(3)
    # hidden-layer discriminator biases have zero true gradient (BatchNorm removes them): compare the rest
    assert np.abs(got - ref).max() < 2e-3 * np.abs(ref).max()
    # generator step
    out = torch.zeros(6, device=dev)
    eng.egm_gen_step(T(z), T(idx), T(v), T(x[:, 0]), T(y[:, 0]), seed, stream + 16, apply=False, out=out)
    noises = OB.egm_noises(m64, B, seed, stream + 16, np.float64)
    losses, gr = OB.egm_gen_step_grads(m64, dz64, 1, f64(z), f64(v[idx]), f64(x[idx]), f64(y[idx]), noises)
    assert np.all(np.abs(out.cpu().numpy() - losses) <= 2e-4 * np.abs(losses) + 1e-5), (out.cpu().numpy(), losses)
    got = eng.split(eng.read(1))
    for name in ("g", "e", "f", "h"):
        gg = [got[name]["gamma"], got[name]["beta"]] + [a for L in got[name]["layers"] for a in L]
        for a, b in zip(gg, OB.flat_grads(gr[name])):
            assert _rel(a, b) < 3e-3, (name, a.shape, _rel(a, b))
    # an applied step moves the parameters of all four nets
    before = eng.read(0)
    eng.egm_gen_step(T(z), T(idx), T(v), T(x[:, 0]), T(y[:, 0]), seed, stream + 32, apply=True)
    after = eng.read(0)
    for k in range(4):
        sl = slice(eng.offsets[k], eng.offsets[k + 1])
        assert np.abs(after[sl] - before[sl]).max() > 1e-5
    eng.egm_end()
    eng.close()


def _params(tmp_path, binary, p=20):
    return dict(dataset="t", output_dir=str(tmp_path), save_res=True, save_model=False, binary_treatment=binary, use_bnn=True,
                z_dims=[1, 1, 1, 7], v_dim=p, lr_theta=1e-3, lr_z=1e-3, g_units=[64] * 5, f_units=[64, 32, 8], h_units=[64, 32, 8],
                e_units=[64] * 5, dz_units=[64, 32, 8], kl_weight=1e-4, lr=2e-4, g_d_freq=5, use_z_rec=True)


@pytest.mark.parametrize("binary", [False, True])
def test_model_surface_with_bayesian_nets(tmp_path, binary):
    """CausalBGM(use_bnn=True): EGM warm start, iterative updates, evaluate and predict through the reference's class surface."""
    from bayesgm_amd.models import CausalBGM
    from bayesgm_amd.datasets import Sim_Hirano_Imbens_sampler
    x, y, v = Sim_Hirano_Imbens_sampler(N=600, v_dim=20, seed=0).load_all()
    if binary:
        x = (x > np.median(x)).astype(np.float32)
    model = CausalBGM(_params(tmp_path, binary), random_seed=3)
    assert type(model).__name__ == "CausalBGMBayes" and isinstance(model, CausalBGM)
    c0, mx0, my0, mv0 = model.evaluate((x, y, v))
    model.fit((x, y, v), epochs=4, epochs_per_eval=2, batch_size=32, use_egm_init=True, egm_n_iter=40, egm_batches_per_eval=20,
              verbose=0)
    assert model.data_z.shape == (600, 10) and torch.isfinite(model.data_z).all()
    c1, mx1, my1, mv1 = model.evaluate((x, y, v), data_z=model.data_z.cpu().numpy())
    assert c1.shape == ((600, 1) if binary else (200,)) and isinstance(my1, np.float32)
    assert np.isfinite([mx1, my1, mv1]).all() and mv1 < mv0
    if binary:
        eff, interval = model.predict((x, y, v), alpha=0.05, n_mcmc=40, burn_in=30, q_sd=0.5, bs=256, verbose=0)
        assert eff.shape == (600,) and interval.shape == (600, 2)
    else:
        xs = np.linspace(0, 3, 5)
        eff, interval = model.predict((x, y, v), alpha=0.05, n_mcmc=40, burn_in=30, x_values=xs, q_sd=0.5, bs=256, verbose=0)
        assert eff.shape == (5,) and interval.shape == (5, 2)
        with pytest.raises(ValueError):
            model.predict((x, y, v), n_mcmc=5, burn_in=5, verbose=0)
    assert np.isfinite(eff).all() and np.all(interval[:, 0] <= interval[:, 1])
    assert 0.0 < model.last_acceptance_rate <= 1.0
    draws = model.metropolis_hastings_sampler((x[:100], y[:100], v[:100]), q_sd=0.5, burn_in=10, n_keep=6)
    assert draws.shape == (6, 100, 10) and np.isfinite(draws).all()
    lp = model.get_log_posterior(x[:100], y[:100], v[:100], draws[-1])
    assert lp.shape == (100,) and np.isfinite(lp).all()
    eff_alone = model.infer_from_latent_posterior(draws, x_values=None if binary else [0.5, 1.5])
    assert eff_alone.shape == ((6, 100) if binary else (2, 6)) and np.isfinite(eff_alone).all()
    # adaptive proposal scale
    model.predict((x, y, v), alpha=0.05, n_mcmc=10, burn_in=120, x_values=None if binary else [1.0], q_sd=-1, bs=300, verbose=0)
    # adaptive proposal scale: one value per block (each block is its own sampler run), moved at counter = 50 and 100
    allowed = np.array([0.81, 0.9, 0.99, 1.0, 1.1, 1.21])
    assert model.last_q_sd.shape == (2,) and np.all(np.abs(model.last_q_sd[:, None] - allowed[None, :]).min(axis=1) < 1e-5)


def test_semi_acic_configuration_through_the_class(tmp_path):
    """The reference's Semi_acic.yaml shape (binary treatment, z_dims [3, 6, 3, 6], v_dim 177, use_bnn): warm start, iterative
    updates and predict through the class surface; with 16 < q <= 32 the training steps run the two-latent-tile row chains."""
    from bayesgm_amd.models import CausalBGM
    from bayesgm_amd.datasets import Sim_Hirano_Imbens_sampler
    x, y, v = Sim_Hirano_Imbens_sampler(N=320, v_dim=177, seed=1).load_all()
    x = (x > np.median(x)).astype(np.float32)
    params = _params(tmp_path, True, p=177)
    params["z_dims"] = [3, 6, 3, 6]
    model = CausalBGM(params, random_seed=5)
    _, mx0, my0, mv0 = model.evaluate((x, y, v))
    model.fit((x, y, v), epochs=3, epochs_per_eval=3, batch_size=32, use_egm_init=True, egm_n_iter=30, egm_batches_per_eval=30, verbose=0)
    assert model.data_z.shape == (320, 18) and torch.isfinite(model.data_z).all()
    _, mx1, my1, mv1 = model.evaluate((x, y, v), data_z=model.data_z.cpu().numpy())
    assert np.isfinite([mx1, my1, mv1]).all() and mv1 < mv0
    ite, interval = model.predict((x, y, v), alpha=0.05, n_mcmc=30, burn_in=30, q_sd=0.5, bs=160, verbose=0)
    assert ite.shape == (320,) and interval.shape == (320, 2) and np.isfinite(ite).all() and np.all(interval[:, 0] <= interval[:, 1])


def test_two_rank_fit_and_predict_with_bayesian_nets():
    """Data-parallel paths of the Bayesian model executed for real (two ranks on this GPU over gloo): identical networks on
    both ranks after EGM + fit; the block-sharded predict equals the single-process predict of the same seeded model."""
    import json, os, subprocess, sys
    from conftest import run_two_ranks
    r = run_two_ranks("dp_bnn_smoke.py")
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert r.stdout.count('"spread": 0.0') == 2, r.stdout[-2000:]
    from bayesgm_amd.models import CausalBGM
    from bayesgm_amd.datasets import Sim_Hirano_Imbens_sampler
    # the two ranks print concurrently: their JSON objects may share a line
    two = json.JSONDecoder().raw_decode(r.stdout[r.stdout.index('{"rank": 0'):])[0]
    x, y, v = Sim_Hirano_Imbens_sampler(N=1089, v_dim=30, seed=1).load_all()
    m = CausalBGM(dict(_params("gpurun_out/dp", False, p=30), save_res=False), random_seed=2)
    adrf, interval = m.predict((x, y, v), alpha=0.05, n_mcmc=30, burn_in=30, x_values=np.linspace(0, 3, 6), q_sd=0.5, bs=256, verbose=0)
    assert np.abs(np.array(two["adrf_untrained"]) - adrf).max() <= 1e-5
    assert np.abs(np.array(two["interval_untrained"]) - interval.ravel()).max() <= 1e-5


def test_fixed_statistics_mode_matches_oracle():
    """Build option bnn_norm="fixed" (mean 0 / variance 1 instead of the batch statistics): steps and log-posterior."""
    from bayesgm_amd.bnn_engine import BnnEngine
    m = _model(False, p=50)
    for k in ("g", "e", "f", "h"):
        m[k]["norm"] = "fixed"
    z, x, y, v = _panel(m, 300)
    eng = BnnEngine(m["v_dim"], m["z_dims"], False, kl_weight=0.01, max_batch=32, norm_mode=1)
    eng.begin(m)
    dev = eng.device
    T = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    B, seed, stream = 32, 777, 3
    idx = np.random.RandomState(0).choice(300, B, replace=False).astype(np.int32)
    m64 = OB.cast_model(m, np.float64)
    f64 = lambda a: a.astype(np.float64)
    eng.theta_step(T(z), T(idx), T(x[:, 0]), T(y[:, 0]), T(v), 1e-3, seed, stream, apply=False)
    grad = eng.split(eng.read(1))
    for name in ("g", "h", "f"):
        noise = OB.draw_noise(OB.net_dims(m[name]), B, seed, stream, OB.NET_ID[name], dtype=np.float64)
        _, _, g = OB.theta_step(m64, name, f64(z[idx]), f64(x[idx]), f64(y[idx]), f64(v[idx]), noise, 0.01)
        got = [grad[name]["gamma"], grad[name]["beta"]] + [a for L in grad[name]["layers"] for a in L]
        for a, b in zip(got, OB.flat_grads(g)):
            assert _rel(a, b) < 2e-3, (name, a.shape, _rel(a, b))
    dz = torch.zeros(B, 10, device=dev)
    eng.z_step(T(x[:, 0]), T(y[:, 0]), T(v), T(z), None, None, T(idx), 1e-3, seed, stream + 4, dz_out=dz)
    noises = {k: tuple(OB.draw_noise(OB.net_dims(m[k]), B, seed, stream + 4 + c, OB.NET_ID[k], dtype=np.float64) for c in (0, 1))
              for k in ("g", "h", "f")}
    _, ref = OB.z_step(m64, f64(z[idx]), f64(x[idx]), f64(y[idx]), f64(v[idx]), noises)
    assert _rel(dz.cpu().numpy(), ref) < 2e-3
    got = eng.logpost(T(x[:, 0]), T(y[:, 0]), T(v), T(z), 128, seed, 9).cpu().numpy()
    ref = OB.log_posterior_blocks(m64, f64(x), f64(y), f64(v), f64(z), 128, seed, 9)
    assert np.abs(got - ref).max() < 2e-3 * np.abs(ref).max()
    # a constant treatment column is no longer normalised away: the dose moves the outcome net's prediction
    xs = np.array([0.0, 3.0], np.float32)
    _, _, dose = eng.evaluate(T(x[:, 0]), T(y[:, 0]), T(v), T(z), x_values=xs, seed=seed, stream_id=20)
    eng.close()


def test_full_size_panel_blocks_are_independent_sampler_runs():
    """BASELINE-size panel (N = 10^6, p = 200, bs = 10^4): block b of the lock-step sampler equals a stand-alone run on that
    block's rows with the same block id and row offset (statistics, perturbations and chains never cross blocks), for a
    middle block and for the ragged last one."""
    from bayesgm_amd.bnn_engine import BnnEngine
    m = _model(False, p=200)
    n, bs, p = 1000000 - 3700, 10000, 200
    eng = BnnEngine(p, m["z_dims"], False, max_batch=32)
    eng.begin(m)
    dev = eng.device
    g = torch.Generator(device=dev); g.manual_seed(5)
    v = torch.randn(n, p, device=dev, generator=g)
    x = torch.rand(n, device=dev, generator=g)
    y = torch.randn(n, device=dev, generator=g)
    seed = (11 << 32) | 2024
    state = torch.empty(n, 10, device=dev)
    acc = torch.zeros(1, device=dev, dtype=torch.int32)
    eng.mh_run(x, y, v, state, bs, 0, 3, 0, 0.7, seed, init=True, acc_count=acc)
    assert torch.isfinite(state).all() and 0 < int(acc[0]) < 3 * n
    for blk in (37, n // bs):                      # the last block has 6300 rows
        lo, hi = blk * bs, min(n, (blk + 1) * bs)
        sub = torch.empty(hi - lo, 10, device=dev)
        eng.mh_run(x[lo:hi].contiguous(), y[lo:hi].contiguous(), v[lo:hi].contiguous(), sub, bs, 0, 3, 0, 0.7, seed, init=True,
                   row_base=lo, block0=blk)
        # fp64 atomics accumulate the block statistics in launch order: allow a last-bit difference to flip a rare accept decision
        differ = ((sub - state[lo:hi]).abs().amax(dim=1) > 1e-5).float().mean().item()
        assert differ < 2e-3, differ
    eng.close()


@pytest.mark.parametrize("z_adam,units,batch", [("replay", {}, 32), ("lazy", {}, 32),
                                                 ("replay", dict(g_units=[128, 96], e_units=[100], f_units=[80, 40], h_units=[72]), 32),
                                                 ("lazy", dict(g_units=[128, 96], e_units=[100], f_units=[80, 40], h_units=[72]), 32),
                                                 ("replay", {}, 40)])
def test_epoch_loop_inside_the_library_equals_the_host_loop(tmp_path, z_adam, units, batch):
    """CausalBGM(use_bnn=True).fit(host_loop=False) -- one bgm_bnn_fit_epoch call per epoch, the latent phase of a minibatch on a second
    stream beside the chains of the next -- gives the parameters and the latent table of the per-minibatch calls from Python bit for
    bit (n = 200 = 6 x 32 + 8: the short last minibatch runs on the phase machine).  Other widths / minibatch sizes: the general step
    kernels, whose latent step runs beside the reading part of the next theta step on its own workspace slices (HIP events)."""
    from bayesgm_amd.models import CausalBGM
    rs = np.random.RandomState(0)
    n, p = 200, 100
    v = rs.randn(n, p).astype(np.float32)
    x = rs.exponential(size=(n, 1)).astype(np.float32)
    y = (x + 0.3 * v[:, :1] + rs.randn(n, 1)).astype(np.float32)
    params = dict(dataset="t", output_dir=str(tmp_path), save_res=False, save_model=False, binary_treatment=False, use_bnn=True,
                  z_dims=[1, 1, 1, 7], v_dim=p, lr_theta=1e-3, lr_z=1e-3, lr=2e-4, g_d_freq=5, use_z_rec=True, kl_weight=1e-4,
                  g_units=[64] * 5, e_units=[64] * 5, f_units=[64, 32, 8], h_units=[64, 32, 8], dz_units=[64, 32, 8])
    params.update(units)
    res = []
    for host_loop in (True, False):
        m = CausalBGM(dict(params), timestamp="t%d" % host_loop, random_seed=3)
        m.fit((x, y, v), epochs=2, epochs_per_eval=1, batch_size=batch, use_egm_init=False, verbose=0, z_adam=z_adam, host_loop=host_loop)
        res.append((m.data_z.cpu().numpy().copy(), m.engine.read(0).copy(), m._stream))
    (za, ta, sa), (zb, tb, sb) = res
    assert sa == sb
    assert np.array_equal(za, zb) and np.array_equal(ta, tb)


def test_fit_with_minibatches_beyond_64_rows(tmp_path):
    """fit(batch_size=...) takes any size in the reference (causalbgm/base.py:434); here up to 4096 rows per rank: sizes other than 16 / 32
    run the one-workgroup-per-net step kernels, incl. the EGM warm start and the short last minibatch of an epoch (600 = 4 x 128 + 88)."""
    from bayesgm_amd.models import CausalBGM
    from bayesgm_amd.datasets import Sim_Hirano_Imbens_sampler
    x, y, v = Sim_Hirano_Imbens_sampler(N=600, v_dim=20, seed=0).load_all()
    model = CausalBGM(_params(tmp_path, False), random_seed=3)
    _, _, _, mv0 = model.evaluate((x, y, v))
    model.fit((x, y, v), epochs=3, epochs_per_eval=3, batch_size=128, use_egm_init=True, egm_n_iter=30, egm_batches_per_eval=30, verbose=0)
    _, mx1, my1, mv1 = model.evaluate((x, y, v), data_z=model.data_z.cpu().numpy())
    assert np.isfinite([mx1, my1, mv1]).all() and mv1 < mv0
    with pytest.raises(ValueError, match="max_batch"):      # the session has been sized for 256 rows and has taken steps
        model.fit((x, y, v), epochs=1, batch_size=300, use_egm_init=False, verbose=0)
    # beyond 256 rows per rank: the session is re-opened for the larger minibatch before its first step (or sized by params['max_batch'])
    for extra in ({}, {"max_batch": 512}):
        prm = _params(tmp_path, False)
        prm.update(extra)
        big = CausalBGM(prm, random_seed=3)
        _, _, _, mv0 = big.evaluate((x, y, v))
        big.fit((x, y, v), epochs=3, epochs_per_eval=3, batch_size=300, use_egm_init=True, egm_n_iter=30, egm_batches_per_eval=30, verbose=0)
        _, mx1, my1, mv1 = big.evaluate((x, y, v), data_z=big.data_z.cpu().numpy())
        assert np.isfinite([mx1, my1, mv1]).all() and mv1 < mv0
        assert big.engine.cfg.max_batch == (512 if extra else 300)


@pytest.mark.parametrize("fixed_norm,p,B", [(True, 100, 32), (True, 100, 16), (False, 50, 19), (True, 50, 40)])
def test_fixed_sigmas_with_bayesian_nets(fixed_norm, p, B):
    """params['sigma_v' | 'sigma_x' | 'sigma_y'] with use_bnn=True (causalbgm/base.py:161,195,224 theta steps, :257,268,283 latent step,
    :698 outcome noise, :765-817 log posterior): the likelihood variances are the given constants, the variance heads are neither read
    nor trained by the data terms.  Row-tile chains (p = 100, B = 16 / 32), the one-workgroup-per-net kernels, both sampling families."""
    m = _model(False, p=p, fixed=fixed_norm)
    m.update(sigma_v=0.8, sigma_x=1.3, sigma_y=0.6)
    n = 160
    z, x, y, v = _panel(m, n)
    eng = _engine(m, max_batch=max(32, B), kl_weight=0.01, **(dict(norm_mode=1) if fixed_norm else {}))
    dev = eng.device
    T = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    m64 = OB.cast_model(m, np.float64)
    f64 = lambda a: a.astype(np.float64)
    idx = np.random.RandomState(4).choice(n, B, replace=False).astype(np.int32)
    seed, stream = (5 << 32) | 77, 12
    # theta step
    out = torch.zeros(8, device=dev)
    eng.theta_step(T(z), T(idx), T(x[:, 0]), T(y[:, 0]), T(v), 1e-3, seed, stream, apply=False, out=out)
    grad = eng.split(eng.read(1))
    o = out.cpu().numpy()
    for w, name in enumerate(("g", "h", "f")):
        noise = OB.draw_noise(OB.net_dims(m[name]), B, seed, stream, OB.NET_ID[name], dtype=np.float64)
        loss, aux, g = OB.theta_step(m64, name, f64(z[idx]), f64(x[idx]), f64(y[idx]), f64(v[idx]), noise, 0.01)
        assert abs(o[2 * w] - loss) < 2e-4 * max(1.0, abs(loss)), (name, o[2 * w], loss)
        got = [grad[name]["gamma"], grad[name]["beta"]] + [a for L in grad[name]["layers"] for a in L]
        for a, b in zip(got, OB.flat_grads(g)):
            assert _rel(a, b) < 2e-3, (name, a.shape, _rel(a, b))
    # latent step
    out = torch.zeros(4, device=dev)
    dz = torch.zeros(B, sum(m["z_dims"]), device=dev)
    eng.z_step(T(x[:, 0]), T(y[:, 0]), T(v), T(z), None, None, T(idx), 1e-3, seed, stream + 20, out=out, dz_out=dz)
    noises = {k: tuple(OB.draw_noise(OB.net_dims(m[k]), B, seed, stream + 20 + c, OB.NET_ID[k], dtype=np.float64) for c in (0, 1))
              for k in ("g", "h", "f")}
    loss, ref = OB.z_step(m64, f64(z[idx]), f64(x[idx]), f64(y[idx]), f64(v[idx]), noises)
    assert abs(float(out[0]) - loss) < 2e-4 * abs(loss)
    assert _rel(dz.cpu().numpy(), ref) < 2e-3
    # log posterior on blocks, and its value changes with the fixed scale (the option is live)
    bs = 64
    got = eng.logpost(T(x[:, 0]), T(y[:, 0]), T(v), T(z), bs, seed, 77, block0=2).cpu().numpy()
    ref = OB.log_posterior_blocks(m64, f64(x), f64(y), f64(v), f64(z), bs, seed, 77, block0=2)
    assert np.abs(got - ref).max() < 2e-3 * np.abs(ref).max(), np.abs(got - ref).max()
    free = dict(m64); [free.pop(k) for k in ("sigma_v", "sigma_x", "sigma_y")]
    assert np.abs(OB.log_posterior_blocks(free, f64(x), f64(y), f64(v), f64(z), bs, seed, 77, block0=2) - ref).max() > 1.0
    # outcome draws of the effects pass use sigma_y
    keep = 2
    draws = T(np.stack([z, z[::-1].copy()]))
    xs = np.array([0.0, 1.1, 2.4], np.float32)
    alone = eng.effects(draws, bs, seed, it0=3, x_values=xs, sample_y=True, row_base=40, block0=1).cpu().numpy()
    for d in range(keep):
        refd = OB.effects_draw(m64, f64(draws[d].cpu().numpy()), f64(xs), d, 3 + d, True, seed, bs, block0=1, row_base=40)
        assert np.abs(alone[:, d] - refd.mean(axis=1)).max() < 5e-4, (d, alone[:, d], refd.mean(axis=1))
    eng.close()


def test_class_accepts_fixed_sigmas_with_bayesian_nets(tmp_path):
    from bayesgm_amd.models import CausalBGM
    from bayesgm_amd.datasets import Sim_Hirano_Imbens_sampler
    x, y, v = Sim_Hirano_Imbens_sampler(N=400, v_dim=20, seed=0).load_all()
    prm = dict(_params(tmp_path, False), sigma_v=1.0, sigma_y=0.5)
    model = CausalBGM(prm, random_seed=3)
    assert model.engine.cfg.sigma_v == 1.0 and model.engine.cfg.sigma_x == 0.0 and model.engine.cfg.sigma_y == 0.5
    model.fit((x, y, v), epochs=2, epochs_per_eval=2, batch_size=32, use_egm_init=True, egm_n_iter=20, egm_batches_per_eval=20, verbose=0)
    eff, interval = model.predict((x, y, v), alpha=0.05, n_mcmc=20, burn_in=20, x_values=np.linspace(0, 3, 4), q_sd=0.5, bs=128, verbose=0)
    assert eff.shape == (4,) and np.isfinite(eff).all() and np.isfinite(interval).all()
    with pytest.raises(ValueError):
        CausalBGM(dict(prm, sigma_x=-1.0), random_seed=3)


# ---------------------------------------------------------------------------------------------------------------------------
# hidden widths beyond 64 (the any-width sampling / evaluation path, csrc/bnw_kernels.h; inference-mode normalisation)
# ---------------------------------------------------------------------------------------------------------------------------
WIDE = dict(g_units=(128, 96), e_units=(100,), f_units=(80, 40), h_units=(72,))
WIDE256 = dict(g_units=(256, 256, 256), e_units=(256, 256, 256), f_units=(256, 256, 256), h_units=(256, 256, 256))      # networks/base.py:7 default nb_units


@pytest.mark.parametrize("binary,units,p,n,bs,fixed", [(False, WIDE, 50, 300, 128, True), (True, WIDE, 37, 150, 150, True), (False, WIDE256, 120, 200, 70, True),
                                                        (False, WIDE, 50, 300, 128, False), (True, WIDE, 37, 150, 150, False), (False, WIDE256, 120, 200, 70, False)])
def test_wide_bayesian_nets_sampling_and_evaluation_match_oracle(binary, units, p, n, bs, fixed):
    """use_bnn=True with hidden widths > 64: log posterior on blocks, two MH iterations, effects of kept draws and evaluate against
    oracle/bnn.py -- the checks of the narrow-shape tests above (ragged last block, several row tiles per block).  fixed = False: the
    input BatchNormalization on the statistics of the block / the panel a call sees (networks/bnn.py:25-27 as written,
    params['bnn_norm'] = 'batch'), any width."""
    m = _model(binary, p=p, fixed=fixed, **units)
    z, x, y, v = _panel(m, n)
    eng = _engine(m, norm_mode=1 if fixed else 0, **units)
    dev = eng.device
    T = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    m64 = OB.cast_model(m, np.float64)
    f64 = lambda a: a.astype(np.float64)
    seed, stream = (3 << 32) | 1234, 77
    got = eng.logpost(T(x[:, 0]), T(y[:, 0]), T(v), T(z), bs, seed, stream, block0=2).cpu().numpy()
    ref = OB.log_posterior_blocks(m64, f64(x), f64(y), f64(v), f64(z), bs, seed, stream, block0=2)
    assert np.abs(got - ref).max() < 2e-3 * np.abs(ref).max(), np.abs(got - ref).max()
    # two MH iterations from the given state
    state = T(z)
    acc = torch.zeros(1, dtype=torch.int32, device=dev)
    eng.mh_run(T(x[:, 0]), T(y[:, 0]), T(v), state, bs, it_begin=5, n_iters=2, burn_in=0, q_sd=0.3, seed=seed, row_base=1000, acc_count=acc)
    zo, n_acc, fragile = f64(z), 0, np.zeros(n, bool)
    for it in (5, 6):
        zo, a_, lpp, lpc = OB.mh_iteration(m64, f64(x), f64(y), f64(v), zo, it, 0.3, seed, bs, row_base=1000)
        n_acc += int(a_.sum())
        u = OB.R.uniforms(np.arange(1000, 1000 + n), it, OB.R.TAG_ACC, seed)
        fragile |= np.abs(u - np.exp(np.minimum(lpp - lpc, 0))) < 2e-3
    ok = ~fragile
    assert ok.sum() >= 0.97 * n and np.abs(state.cpu().numpy()[ok] - zo[ok]).max() < 1e-5
    assert abs(int(acc[0]) - n_acc) <= int(fragile.sum())
    # effects of two given draws (with outcome noise) and the fused pass of a short run
    draws = T(np.stack([z, z[::-1].copy()]))
    xs = np.array([0.0, 0.9, 2.1], np.float32)
    alone = eng.effects(draws, bs, seed, it0=3, x_values=None if binary else xs, sample_y=True, row_base=40, block0=1).cpu().numpy()
    for d in range(2):
        refd = OB.effects_draw(m64, f64(draws[d].cpu().numpy()), [1.0, 0.0] if binary else f64(xs), d, 3 + d, True, seed, bs, block0=1, row_base=40)
        if binary:
            assert np.abs(alone[d] - (refd[0] - refd[1])).max() < 2e-3
        else:
            assert np.abs(alone[:, d] - refd.mean(axis=1)).max() < 5e-4
    # evaluate: encoder, reconstruction errors, dose grid / ITE
    xg = np.linspace(0.1, 2.5, 5).astype(np.float32)
    zt, sums, causal = eng.evaluate(T(x[:, 0]), T(y[:, 0]), T(v), None, x_values=xg, seed=31337, stream_id=900)
    zr, cr, mse_x, mse_y, mse_v = OB.evaluate(m64, (f64(x), f64(y), f64(v)), None, xg, 31337, 900)
    assert np.abs(zt.cpu().numpy() - zr).max() < 2e-3 * max(1.0, np.abs(zr).max())
    s_ = sums.cpu().numpy()
    assert abs(s_[0] / (n * p) - mse_v) < 1e-3 * mse_v and abs(s_[1] / n - mse_x) < 2e-3 * mse_x and abs(s_[2] / n - mse_y) < 2e-3 * mse_y
    if binary:
        assert np.abs(causal.cpu().numpy() - cr).max() < 2e-3
    else:
        assert np.abs(causal.cpu().numpy() / n - cr).max() < 5e-4
    eng.close()


@pytest.mark.parametrize("bnn_norm", ["fixed", "batch"])
def test_class_with_wide_bayesian_nets(tmp_path, bnn_norm):
    """CausalBGM(use_bnn=True) with g_units = [128, 128] and nb_units-default-sized f / h: EGM warm start, fit (with its evaluations) and
    predict run through the class -- also with the input BatchNormalization on batch statistics (the reference as written)."""
    import warnings
    from bayesgm_amd.models import CausalBGM
    from bayesgm_amd.datasets import Sim_Hirano_Imbens_sampler
    x, y, v = Sim_Hirano_Imbens_sampler(N=400, v_dim=20, seed=0).load_all()
    prm = dict(_params(tmp_path, False), g_units=[128, 128], e_units=[128, 128], f_units=[256, 256, 256], h_units=[96, 48], bnn_norm=bnn_norm)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        model = CausalBGM(prm, random_seed=3)
    _, _, _, mv0 = model.evaluate((x, y, v))
    model.fit((x, y, v), epochs=2, epochs_per_eval=1, batch_size=32, use_egm_init=True, egm_n_iter=20, egm_batches_per_eval=10, verbose=0)
    _, mx1, my1, mv1 = model.evaluate((x, y, v), data_z=model.data_z.cpu().numpy())
    assert np.isfinite([mx1, my1, mv1]).all() and mv1 < mv0
    eff, interval = model.predict((x, y, v), alpha=0.05, n_mcmc=10, burn_in=10, x_values=np.linspace(0, 3, 4), q_sd=0.5, bs=128, verbose=0)
    assert eff.shape == (4,) and np.isfinite(eff).all() and np.isfinite(interval).all()


def test_wide_bayesian_nets_on_tiny_blocks():
    """Blocks smaller than a tile, a last block of one row (any-width path)."""
    m = _model(False, p=11, fixed=True, **WIDE)
    n, bs = 7, 3
    z, x, y, v = _panel(m, n)
    eng = _engine(m, norm_mode=1, **WIDE)
    T = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(eng.device)
    got = eng.logpost(T(x[:, 0]), T(y[:, 0]), T(v), T(z), bs, 99, 5).cpu().numpy()
    m64 = OB.cast_model(m, np.float64)
    ref = OB.log_posterior_blocks(m64, x.astype(np.float64), y.astype(np.float64), v.astype(np.float64), z.astype(np.float64), bs, 99, 5)
    assert np.abs(got - ref).max() < 2e-3 * np.abs(ref).max()
    eng.close()


@pytest.mark.parametrize("use_bnn", [True, False])
def test_as_written_semantics_stay_a_supported_mode(tmp_path, use_bnn):
    """The literal reading of the reference -- input BatchNormalization of the Bayesian nets and the discriminator's BatchNormalization on
    the statistics of the batch at hand (networks/bnn.py:24-27, base.py:364-379: call(..., training=True) is the call default), Keras'
    dense-decay Adam on the whole latent table at every minibatch (base.py:296-302) -- is not the build's default but
    stays selectable: params['bnn_norm'] = params['disc_norm'] = 'batch', fit(z_adam='dense').  The tutorial's workflow through the
    class in that mode: warm start, fit with evaluations, predict; finite results, falling reconstruction error, and the batch-statistics
    kernels' acceptance behaviour."""
    import warnings
    from bayesgm_amd.models import CausalBGM
    from bayesgm_amd.datasets import Sim_Hirano_Imbens_sampler
    x, y, v = Sim_Hirano_Imbens_sampler(N=640, v_dim=20, seed=0).load_all()
    prm = dict(_params(tmp_path, False), use_bnn=use_bnn, bnn_norm="batch", disc_norm="batch")
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        model = CausalBGM(prm, random_seed=3)
    _, _, _, mv0 = model.evaluate((x, y, v))
    model.fit((x, y, v), epochs=3, epochs_per_eval=1, batch_size=32, use_egm_init=True, egm_n_iter=40, egm_batches_per_eval=20, verbose=0,
              z_adam="dense")
    _, mx1, my1, mv1 = model.evaluate((x, y, v), data_z=model.data_z.cpu().numpy())
    assert np.isfinite([mx1, my1, mv1]).all() and mv1 < mv0
    xs = np.linspace(0, 3, 5)
    eff, interval = model.predict((x, y, v), alpha=0.05, n_mcmc=30, burn_in=30, x_values=xs, q_sd=0.5, bs=320, verbose=0)
    assert eff.shape == (5,) and np.isfinite(eff).all() and np.all(interval[:, 0] <= interval[:, 1])
    assert 0.0 < model.last_acceptance_rate <= 1.0
    if use_bnn:
        # batch statistics normalise a constant treatment column away (the reason the build's default differs): the dose-response
        # estimate does not depend on the dose in this mode
        assert np.ptp(eff) < 0.2 * (np.abs(eff).mean() + 1.0)


def test_general_steps_over_the_chip_equal_the_one_launch_form(tmp_path):
    """The general (any-width / any-minibatch) Bayesian training steps run their elementwise parts (the calls' eps / dW, KL terms, Adam) and
    their parameter-gradient tiles as launches over the chip around the one-workgroup step kernels, the latent step's two noise calls on
    workgroups of their own.  Same arithmetic per element and per tile, same order of every sum: parameters and latents of a
    short job -- CausalBGM(use_bnn=True) at [128, 96]-type nets with the EGM warm start, BGM(use_bnn=True) -- equal those of the
    one-launch form (BGM_BNN_STEP_ONE_LAUNCH=1, read once per process: two subprocesses) up to the last bits (hipcc is free to contract
    a multiply-add of the same source expression differently in two kernels: 3e-7 on parameters of order one after the job)."""
    import os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    outs = []
    for tag, extra in (("chip", {}), ("one", {"BGM_BNN_STEP_ONE_LAUNCH": "1"})):
        out = str(tmp_path / ("steps_%s.npz" % tag))
        env = dict(os.environ)
        env.pop("BGM_BNN_STEP_ONE_LAUNCH", None)
        env.update(extra)
        r = subprocess.run([sys.executable, os.path.join(root, "tests", "_wide_steps_helper.py"), out], cwd=root, env=env, capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr[-2000:]
        outs.append(np.load(out))
    a, b = outs
    for key in ("causal_theta", "causal_z", "bgm_theta", "bgm_z"):
        assert np.isfinite(a[key]).all()
        assert np.abs(a[key] - b[key]).max() < 5e-6, (key, float(np.abs(a[key] - b[key]).max()))
