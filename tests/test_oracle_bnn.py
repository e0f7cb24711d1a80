"""Hand-derived gradients of the Bayesian-net step functions (oracle/bnn.py) against PyTorch autograd (fp64),
and the counter-based noise layout."""
import numpy as np
import pytest
import torch

from oracle import bnn as OB


def _t(a):
    return torch.tensor(np.asarray(a), dtype=torch.float64)


def _tnet(net):
    return {"gamma": _t(net["gamma"]).requires_grad_(), "beta": _t(net["beta"]).requires_grad_(),
            "layers": [tuple(_t(a).requires_grad_() for a in L) for L in net["layers"]]}


def _tfwd(net, x, noise):
    mu = x.mean(0)
    var = x.var(0, unbiased=False)
    h = (x - mu) / torch.sqrt(var + 1e-3) * net["gamma"] + net["beta"]
    L = len(net["layers"])
    for l, (loc, rho, bias) in enumerate(net["layers"]):
        sg = OB.SCALE_EPS + torch.nn.functional.softplus(rho)
        pre = h @ loc + ((h * _t(noise["sin"][l])) @ (sg * _t(noise["eps"][l]))) * _t(noise["sout"][l]) + bias
        h = torch.nn.functional.leaky_relu(pre, 0.2) if l < L - 1 else pre
    return h


def _tkl(net):
    v = 0.0
    for loc, rho, _ in net["layers"]:
        sg = OB.SCALE_EPS + torch.nn.functional.softplus(rho)
        q = torch.distributions.Normal(loc, sg)
        v = v + torch.distributions.kl_divergence(q, torch.distributions.Normal(torch.zeros_like(loc), 1.0)).sum()
    return v


def _gauss(ssq, raw, dim):
    s2 = torch.nn.functional.softplus(raw) + 1e-6
    return (ssq / (2 * s2) + dim * torch.log(s2) / 2).mean()


def _model(binary, seed=0):
    m = OB.init_model(seed, [2, 1, 2, 3], 9, binary, g_units=(12, 10), e_units=(8,), f_units=(7, 5), h_units=(6, 4),
                      dtype=np.float64)
    rs = np.random.RandomState(seed + 1)
    for k in ("g", "f", "h"):        # non-trivial gamma / beta
        m[k]["gamma"] = 1.0 + 0.3 * rs.standard_normal(m[k]["gamma"].shape)
        m[k]["beta"] = 0.2 * rs.standard_normal(m[k]["beta"].shape)
    return m


def _data(m, B, seed=3):
    rs = np.random.RandomState(seed)
    q = sum(m["z_dims"])
    z = rs.standard_normal((B, q))
    v = rs.standard_normal((B, m["v_dim"]))
    x = (rs.rand(B, 1) > 0.5).astype(np.float64) if m["binary_treatment"] else rs.exponential(size=(B, 1))
    y = rs.standard_normal((B, 1))
    return z, x, y, v


@pytest.mark.parametrize("binary", [False, True])
@pytest.mark.parametrize("name", ["g", "h", "f"])
def test_theta_step_matches_autograd(binary, name):
    m = _model(binary)
    z, x, y, v = _data(m, 11)
    rs = np.random.RandomState(5)
    noise = OB.random_noise(rs, OB.net_dims(m[name]), len(z))
    loss, aux, g = OB.theta_step(m, name, z, x, y, v, noise, kl_weight=0.37)
    tn = _tnet(m[name])
    fin, hin = OB._inputs(m, z, x)
    inp = {"g": z, "f": fin, "h": hin}[name]
    out = _tfwd(tn, _t(inp), noise)
    if name == "g":
        p = m["v_dim"]
        tl = _gauss(((_t(v) - out[:, :p]) ** 2).sum(1), out[:, -1], p)
    elif name == "h" and binary:
        tl = torch.nn.functional.binary_cross_entropy_with_logits(out[:, 0], _t(x[:, 0]))
    else:
        tgt = x if name == "h" else y
        tl = _gauss((_t(tgt[:, 0]) - out[:, 0]) ** 2, out[:, -1], 1)
    tl = tl + 0.37 * _tkl(tn)
    tl.backward()
    assert abs(float(tl.detach()) - loss) < 1e-9 * max(1.0, abs(loss))
    ref = [tn["gamma"].grad, tn["beta"].grad] + [a.grad for L in tn["layers"] for a in L]
    for a, b in zip(OB.flat_grads(g), ref):
        np.testing.assert_allclose(a, b.numpy(), rtol=1e-8, atol=1e-11)


@pytest.mark.parametrize("binary", [False, True])
def test_z_step_matches_autograd(binary):
    m = _model(binary, seed=2)
    z, x, y, v = _data(m, 9)
    rs = np.random.RandomState(7)
    noises = {k: (OB.random_noise(rs, OB.net_dims(m[k]), len(z)), OB.random_noise(rs, OB.net_dims(m[k]), len(z)))
              for k in ("g", "h", "f")}
    loss, dz = OB.z_step(m, z, x, y, v, noises)
    tz = _t(z).requires_grad_()
    z0d, z1d, z2d, _ = m["z_dims"]
    fin = torch.cat([tz[:, :z0d + z1d], _t(x)], 1)
    hin = torch.cat([tz[:, :z0d], tz[:, z0d + z1d:z0d + z1d + z2d]], 1)
    tg, tf, th = (_tnet(m[k]) for k in ("g", "f", "h"))
    p = m["v_dim"]
    o1, o2 = _tfwd(tg, tz, noises["g"][0]), _tfwd(tg, tz, noises["g"][1])
    tl = _gauss(((_t(v) - o1[:, :p]) ** 2).sum(1), o2[:, -1], p)
    h1 = _tfwd(th, hin, noises["h"][0])
    if binary:
        tl = tl + torch.nn.functional.binary_cross_entropy_with_logits(h1[:, 0], _t(x[:, 0]))
    else:
        h2 = _tfwd(th, hin, noises["h"][1])
        tl = tl + _gauss((_t(x[:, 0]) - h1[:, 0]) ** 2, h2[:, -1], 1)
    f1, f2 = _tfwd(tf, fin, noises["f"][0]), _tfwd(tf, fin, noises["f"][1])
    tl = tl + _gauss((_t(y[:, 0]) - f1[:, 0]) ** 2, f2[:, -1], 1) + (tz ** 2).sum(1).mean() / 2
    tl.backward()
    assert abs(float(tl.detach()) - loss) < 1e-9 * max(1.0, abs(loss))
    np.testing.assert_allclose(dz, tz.grad.numpy(), rtol=1e-8, atol=1e-11)


def test_fixed_stats_forward_equals_batch_stats_and_logpost_finite():
    m = _model(False)
    z, x, y, v = _data(m, 40)
    dims = {k: OB.net_dims(m[k]) for k in ("g", "h", "f")}
    noises = {k: OB.draw_noise(dims[k], len(z), key=1234567, stream=5, net_id=OB.NET_ID[k], dtype=np.float64)
              for k in dims}
    fin, hin = OB._inputs(m, z, x)
    stats = {"g": OB.batch_stats(z), "h": OB.batch_stats(hin), "f": OB.batch_stats(fin)}
    a = OB.log_posterior(m, x, y, v, z, noises)
    b = OB.log_posterior(m, x, y, v, z, noises, stats)
    np.testing.assert_allclose(a, b, rtol=1e-12)
    assert np.isfinite(a).all()


def test_noise_layout_is_deterministic_and_balanced():
    dims = [10, 64, 64, 201]
    n1 = OB.draw_noise(dims, 300, key=(7 << 32) | 99, stream=3, net_id=0)
    n2 = OB.draw_noise(dims, 300, key=(7 << 32) | 99, stream=3, net_id=0)
    n3 = OB.draw_noise(dims, 300, key=(7 << 32) | 99, stream=4, net_id=0)
    for a, b in zip(n1["eps"] + n1["sin"] + n1["sout"], n2["eps"] + n2["sin"] + n2["sout"]):
        assert np.array_equal(a, b)
    assert not np.array_equal(n1["eps"][1], n3["eps"][1])
    assert n1["sout"][2].shape == (300, 201) and set(np.unique(n1["sin"][1])) == {-1.0, 1.0}
    assert abs(n1["sout"][2].mean()) < 0.02 and abs(n1["eps"][1].std() - 1.0) < 0.05
    # rows of a later slice of the batch see the same words as the full batch
    part = OB.draw_noise(dims, 100, key=(7 << 32) | 99, stream=3, net_id=0, row0=200)
    assert np.array_equal(part["sin"][1], n1["sin"][1][200:])


@pytest.mark.parametrize("binary", [False, True])
def test_egm_gen_step_matches_autograd(binary):
    from oracle import egm as OE
    m = _model(binary, seed=4)
    for k in ("e",):
        rs = np.random.RandomState(8)
        m[k]["gamma"] = 1.0 + 0.3 * rs.standard_normal(m[k]["gamma"].shape)
        m[k]["beta"] = 0.2 * rs.standard_normal(m[k]["beta"].shape)
    B = 10
    z, x, y, v = _data(m, B)
    q = z.shape[1]
    dz = OE.init_disc(np.random.RandomState(1), q, (7, 5), dtype=np.float64)
    rs = np.random.RandomState(12)
    noises = {c: OB.random_noise(rs, OB.net_dims(m[c[0]]), B) for c in OB.EGM_CALLS}
    losses, grads = OB.egm_gen_step_grads(m, dz, 1, z, v, x, y, noises)
    tn = {k: _tnet(m[k]) for k in ("g", "e", "f", "h")}
    p = m["v_dim"]
    z0d, z1d, z2d, _ = m["z_dims"]
    tz, tv, tx, ty = _t(z), _t(v), _t(x), _t(y)
    gz = _tfwd(tn["g"], tz, noises["g1"])
    gzs = _tfwd(tn["g"], tz, noises["g1s"])
    z_ = _tfwd(tn["e"], tv, noises["e1"])
    z__ = _tfwd(tn["e"], gz[:, :p], noises["e2"])
    gv = _tfwd(tn["g"], z_, noises["g2"])

    def tdisc(a):
        h = a
        for l in range(len(dz["W"]) - 1):
            u = h @ _t(dz["W"][l]) + _t(dz["b"][l])
            u = (u - u.mean(0)) / torch.sqrt(u.var(0, unbiased=False) + 1e-3) * _t(dz["gamma"][l]) + _t(dz["beta"][l])
            h = torch.tanh(u)
        return h @ _t(dz["W"][-1]) + _t(dz["b"][-1])
    d_ = tdisc(z_)
    fin = torch.cat([z_[:, :z0d + z1d], tx], 1)
    hin = torch.cat([z_[:, :z0d], z_[:, z0d + z1d:z0d + z1d + z2d]], 1)
    fo, fs = _tfwd(tn["f"], fin, noises["f"]), _tfwd(tn["f"], fin, noises["fs"])
    ho, hs = _tfwd(tn["h"], hin, noises["h"]), _tfwd(tn["h"], hin, noises["hs"])
    l2x = torch.nn.functional.binary_cross_entropy_with_logits(ho[:, :1], tx) if binary else ((ho[:, :1] - tx) ** 2).mean()
    total = -d_.mean() + ((tv - gv[:, :p]) ** 2).mean() + ((tz - z__) ** 2).mean() + l2x + ((fo[:, :1] - ty) ** 2).mean() \
        + 0.001 * ((gzs[:, -1] ** 2).mean() + (fs[:, -1] ** 2).mean() + (hs[:, -1] ** 2).mean())
    total.backward()
    assert abs(float(total.detach()) - losses[5]) < 1e-9 * max(1.0, abs(losses[5]))
    for k in ("g", "e", "f", "h"):
        ref = [tn[k]["gamma"].grad, tn[k]["beta"].grad] + [a.grad for L in tn[k]["layers"] for a in L]
        for a, b in zip(OB.flat_grads(grads[k]), ref):
            np.testing.assert_allclose(a, b.numpy(), rtol=1e-7, atol=1e-11)


@pytest.mark.parametrize("norm", ["batch", "fixed"])
def test_identifiable_bayesian_prior_step_matches_autograd(norm):
    """oracle.identifiable.bnn_prior_step_given_dz (identifiable.py:195-226 with use_bnn): the z gradient of the conditional-prior term and
    the prior net's gradient of  batch-mean prior term + kl_weight * KL  against autograd; the updates are then plain Adam (checked in
    test_oracle_autograd.py) -- verified here through the first step, where Adam's update is -lr_t * sign-like g / (|g| + eps)."""
    from oracle import identifiable as OI
    from oracle.fit import AdamState, adam_lr_t
    rs = np.random.RandomState(4)
    k, q, B, klw = 5, 6, 12, 0.3
    pn = OI.init_prior_bnn(rs, k, q, (7,), dtype=np.float64)
    pn["gamma"] = 1.0 + 0.3 * rs.standard_normal(k)
    pn["beta"] = 0.2 * rs.standard_normal(k)
    if norm == "fixed":
        pn["norm"] = "fixed"
    seg = rs.randint(0, k, B)
    seg[:k] = np.arange(k)                                   # every column has variance under batch statistics
    z = rs.standard_normal((B, q))
    dz_std = 0.1 * rs.standard_normal((B, q)) + z / B       # stands for d(NLL + |z|^2 / 2)/dz
    noise = OB.random_noise(rs, OB.net_dims(pn), B)
    # autograd
    tn = _tnet(pn)
    zt = _t(z).requires_grad_()
    u = _t(np.eye(k)[seg])
    if norm == "fixed":
        h = u / np.sqrt(1.0 + 1e-3) * tn["gamma"] + tn["beta"]
        L = len(tn["layers"])
        for l, (loc, rho, bias) in enumerate(tn["layers"]):
            sg = OB.SCALE_EPS + torch.nn.functional.softplus(rho)
            pre = h @ loc + ((h * _t(noise["sin"][l])) @ (sg * _t(noise["eps"][l]))) * _t(noise["sout"][l]) + bias
            h = torch.nn.functional.leaky_relu(pre, 0.2) if l < L - 1 else pre
        out = h
    else:
        out = _tfwd(tn, u, noise)
    s2 = torch.nn.functional.softplus(out[:, -1]) + 1e-6
    prior = (((zt - out[:, :-1]) ** 2).sum(1) / (2 * s2) + q * torch.log(s2) / 2).mean()
    loss = prior + klw * _tkl(tn)
    params = [tn["gamma"], tn["beta"]] + [a for Lr in tn["layers"] for a in Lr]
    grads = torch.autograd.grad(loss, [zt] + params)
    g_z = dz_std - z / B + grads[0].numpy()
    # oracle step with lr so small that parameters barely move; compare through the first Adam step: delta = -lr_t * m / (sqrt(v) + eps)
    data_z = z.copy()
    pn2 = OB.cast_bnn(pn, np.float64)
    before = [a.copy() for a in OB.flat_params(pn2)]
    popt = AdamState(OB.flat_params(pn2))
    lr_z, lr_p = 1e-3, 1e-4
    lp, lz, klv = OI.bnn_prior_step_given_dz(pn2, popt, data_z, np.arange(B), seg, dz_std, noise, lr_z, 1, lr_p, klw)
    assert abs(lp - float(prior)) < 1e-10 and abs(klv - float(_tkl(tn))) < 1e-9 and abs(lz - (z ** 2).sum(1).mean() / 2) < 1e-12
    exp_z = z - adam_lr_t(lr_z, 1) * (0.1 * g_z) / (np.sqrt(0.01 * g_z * g_z) + 1e-7)
    np.testing.assert_allclose(data_z, exp_z, rtol=0, atol=1e-9)
    for m_, g in zip(popt.m, grads[1:]):
        np.testing.assert_allclose(m_ / 0.1, g.numpy(), rtol=1e-8, atol=1e-11)      # first moment after one step = (1 - b1) g
    assert any(np.abs(a - b).max() > 0 for a, b in zip(before, OB.flat_params(pn2)))
