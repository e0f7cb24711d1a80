"""Independent second opinion on the oracle's hand-derived gradients and Adam: PyTorch-CPU autograd /
torch.optim.Adam (float64).  This is how the "parity unpinned" oracle is kept honest."""
import numpy as np
import torch

from oracle import causal as OC
from oracle import fit as OF
from oracle.nets import mlp_forward


def _t(a):
    return torch.tensor(np.asarray(a, np.float64), dtype=torch.float64)


def _torch_net(net):
    return [(_t(W).requires_grad_(), _t(b).requires_grad_()) for W, b in net]


def _fwd(net, x):
    h = x
    for i, (W, b) in enumerate(net):
        h = h @ W + b
        if i < len(net) - 1:
            h = torch.maximum(h, 0.2 * h)
    return h


def _sp(x):
    return torch.nn.functional.softplus(x)


def _losses(tm, m, z, x, y, v):
    p = m["v_dim"]
    z0d, z1d, z2d, _ = m["z_dims"]
    go = _fwd(tm["g"], z)
    s2v = _sp(go[:, -1]) + 1e-6
    lv = (((v - go[:, :p]) ** 2).sum(1) / (2 * s2v) + p * torch.log(s2v) / 2).mean()
    ho = _fwd(tm["h"], torch.cat([z[:, :z0d], z[:, z0d + z1d:z0d + z1d + z2d]], 1))
    if m["binary_treatment"]:
        lx = torch.nn.functional.binary_cross_entropy_with_logits(ho[:, :1], x)
    else:
        s2x = _sp(ho[:, -1]) + 1e-6
        lx = (((x - ho[:, :1]) ** 2).sum(1) / (2 * s2x) + torch.log(s2x) / 2).mean()
    fo = _fwd(tm["f"], torch.cat([z[:, :z0d + z1d], x], 1))
    s2y = _sp(fo[:, -1]) + 1e-6
    ly = (((y - fo[:, :1]) ** 2).sum(1) / (2 * s2y) + torch.log(s2y) / 2).mean()
    return lv, lx, ly


def _setup(binary, seed=0, n=24, p=17, z_dims=(2, 1, 3, 4)):
    m = OC.cast_model(OC.init_model(seed, list(z_dims), p, binary_treatment=binary), np.float64)
    rs = np.random.RandomState(seed + 1)
    for k in ("g", "f", "h"):
        m[k] = [(W, 0.1 * rs.randn(*b.shape)) for W, b in m[k]]
    z = rs.randn(n, sum(z_dims))
    v = rs.randn(n, p)
    x = (rs.rand(n, 1) > 0.5).astype(np.float64) if binary else rs.exponential(size=(n, 1))
    y = rs.randn(n, 1)
    return m, z, x, y, v


def test_theta_and_z_gradients_match_autograd():
    for binary in (False, True):
        m, z, x, y, v = _setup(binary)
        tm = {k: _torch_net(m[k]) for k in ("g", "f", "h")}
        tz = _t(z).requires_grad_()
        lv, lx, ly = _losses(tm, m, tz, _t(x), _t(y), _t(v))
        total = lv + lx + ly + ((tz ** 2).sum(1) / 2).mean()
        total.backward()
        l_v, _, gg, _ = OF.g_loss_and_grads(m, z, v)
        l_x, _, gh, _ = OF.h_loss_and_grads(m, z, x)
        l_y, _, gf, _ = OF.f_loss_and_grads(m, z, x, y)
        assert np.allclose([l_v, l_x, l_y], [lv.item(), lx.item(), ly.item()], rtol=1e-12)
        for key, grads in (("g", gg), ("h", gh), ("f", gf)):
            for (dW, db), (W, b) in zip(grads, tm[key]):
                assert np.allclose(dW, W.grad.numpy(), rtol=1e-9, atol=1e-12)
                assert np.allclose(db, b.grad.numpy(), rtol=1e-9, atol=1e-12)
        lz, dz = OF.z_loss_and_grad(m, z, x, y, v)
        assert np.isclose(lz, total.item(), rtol=1e-12)
        assert np.allclose(dz, tz.grad.numpy(), rtol=1e-9, atol=1e-12)


def test_fixed_sigma_gradients():
    m, z, x, y, v = _setup(False, seed=3)
    m.update(sigma_v=0.7, sigma_x=1.2, sigma_y=0.9)
    tz = _t(z).requires_grad_()
    tm = {k: _torch_net(m[k]) for k in ("g", "f", "h")}
    p = m["v_dim"]
    go = _fwd(tm["g"], tz)
    lv = (((_t(v) - go[:, :p]) ** 2).sum(1) / (2 * 0.49) + p * np.log(0.49) / 2).mean()
    lv.backward()
    l_v, _, gg, dz = OF.g_loss_and_grads(m, z, v)
    assert np.isclose(l_v, lv.item())
    assert np.allclose(dz, tz.grad.numpy(), rtol=1e-9, atol=1e-12)
    assert np.allclose(gg[-1][0][:, -1], 0)  # variance column receives no gradient


def test_adam_matches_torch_adam_with_keras_epsilon_form():
    """Keras: var -= lr_t*m/(sqrt(v)+eps) with lr_t folding both bias corrections; torch.optim.Adam:
    lr/(1-b1^t) * m / (sqrt(v)/sqrt(1-b2^t) + eps).  They agree when eps_torch = eps_keras/sqrt(1-b2^t);
    check the oracle against the closed form over several steps instead."""
    rs = np.random.RandomState(0)
    p = rs.randn(5, 3)
    ref = p.copy()
    st = OF.AdamState([p])
    m_ = np.zeros_like(p)
    v_ = np.zeros_like(p)
    for t in range(1, 8):
        g = rs.randn(5, 3)
        st.apply([p], [g], 1e-2)
        m_ = 0.9 * m_ + 0.1 * g
        v_ = 0.99 * v_ + 0.01 * g * g
        ref -= 1e-2 * np.sqrt(1 - 0.99 ** t) / (1 - 0.9 ** t) * m_ / (np.sqrt(v_) + 1e-7)
        assert np.allclose(p, ref, rtol=1e-13)


def test_sparse_adam_dense_decay_moves_untouched_rows():
    m, z, x, y, v = _setup(False, seed=5, n=12)
    st = OF.FitState(m, z.copy(), 1e-3, 1e-2)
    OF.fit_step(st, x, y, v, np.array([0, 1, 2, 3]))
    z1 = st.data_z.copy()
    OF.fit_step(st, x, y, v, np.array([4, 5, 6, 7]))
    # rows 0..3 keep drifting by their decaying momentum under the reference's dense-decay semantics
    assert np.all(np.abs(st.data_z[:4] - z1[:4]) > 0)
    assert np.all(st.data_z[8:] == z[8:])  # never-touched rows: m = v = 0 -> no movement
    st2 = OF.FitState(OC.cast_model(m, np.float64), z.copy(), 1e-3, 1e-2)
    OF.fit_step(st2, x, y, v, np.array([0, 1, 2, 3]), lazy_z=True)
    z2 = st2.data_z.copy()
    OF.fit_step(st2, x, y, v, np.array([4, 5, 6, 7]), lazy_z=True)
    assert np.all(st2.data_z[:4] == z2[:4])


def test_fit_epochs_reduces_loss():
    m, z, x, y, v = _setup(False, seed=7, n=64)
    st = OF.FitState(m, z.copy(), 1e-3, 1e-3)
    hist = OF.fit_epochs(st, (x, y, v), 20, 32, np.random.RandomState(0))
    assert hist.shape == (21 * 2, 7)
    assert hist[-4:, 4].mean() < hist[:4, 4].mean()  # loss_v decreases


# ----------------------------------------------------------------------------- BGM
def _bgm_setup(seed=0, n=20, p=13, q=5):
    from oracle import bgm as OB
    m = OB.cast_model(OB.init_model(seed, q, p, g_units=(16, 16, 16)), np.float64)
    rs = np.random.RandomState(seed + 1)
    g = m["g"]
    g["bn"].update(gamma=1 + 0.1 * rs.randn(q), beta=0.1 * rs.randn(q), mean=0.2 * rs.randn(q), var=0.5 + rs.rand(q))
    g["trunk"] = [(W, 0.1 * rs.randn(*b.shape)) for W, b in g["trunk"]]
    z = rs.randn(n, q)
    x = rs.randn(n, p)
    mask = (rs.rand(n, p) > 0.3).astype(np.float64)
    return OB, m, z, x, mask


def _torch_varnet(g, z, training):
    bn = g["bn"]
    if training:
        mu = z.mean(0)
        var = z.var(0, unbiased=False)
        zn = (z - mu) / torch.sqrt(var + 1e-3) * bn["gamma"] + bn["beta"]
    else:
        zn = (z - bn["mean"]) / torch.sqrt(bn["var"] + 1e-3) * bn["gamma"] + bn["beta"]
    h = zn
    for W, b in g["trunk"]:
        h = h @ W + b
        h = torch.maximum(h, 0.2 * h)
    return h @ g["mean"][0] + g["mean"][1], torch.nn.functional.softplus(h @ g["var"][0] + g["var"][1]) + 1e-6


def _tg(g, req):
    f = (lambda a: _t(a).requires_grad_()) if req else _t
    return {"bn": {k: f(v) if k in ("gamma", "beta") else _t(v) for k, v in g["bn"].items()},
            "trunk": [(f(W), f(b)) for W, b in g["trunk"]], "mean": (f(g["mean"][0]), f(g["mean"][1])),
            "var": (f(g["var"][0]), f(g["var"][1]))}


def test_bgm_log_posterior_gradient_matches_autograd():
    OB, m, z, x, mask = _bgm_setup()
    tz = _t(z).requires_grad_()
    mu, s2 = _torch_varnet(_tg(m["g"], False), tz, training=False)
    lp = -((_t(mask) * ((_t(x) - mu) ** 2 / (2 * s2) + 0.5 * torch.log(s2))).sum(1) + (tz ** 2).sum(1) / 2)
    lp.sum().backward()
    lp_o, gr_o = OB.log_posterior_and_grad(m, z, x, mask)
    assert np.allclose(lp_o, lp.detach().numpy(), rtol=1e-12)
    assert np.allclose(gr_o, tz.grad.numpy(), rtol=1e-9, atol=1e-12)
    assert np.allclose(OB.log_posterior(m, z, x, mask), lp_o, rtol=1e-12)


def test_bgm_fit_gradients_with_batchnorm_match_autograd():
    OB, m, z, x, _ = _bgm_setup(seed=3)
    tg = _tg(m["g"], True)
    tz = _t(z).requires_grad_()
    mu, s2 = _torch_varnet(tg, tz, training=True)
    loss = (((_t(x) - mu) ** 2) / (2 * s2) + 0.5 * torch.log(s2)).sum(1).mean()
    loss.backward()
    l, _, gr, dz, _ = OB.g_loss_and_grads(m, z, x)
    assert np.isclose(l, loss.item(), rtol=1e-12)
    assert np.allclose(dz, tz.grad.numpy(), rtol=1e-8, atol=1e-12)
    assert np.allclose(gr["gamma"], tg["bn"]["gamma"].grad.numpy(), rtol=1e-8, atol=1e-12)
    assert np.allclose(gr["beta"], tg["bn"]["beta"].grad.numpy(), rtol=1e-8, atol=1e-12)
    for (dW, db), (W, b) in zip(gr["trunk"], tg["trunk"]):
        assert np.allclose(dW, W.grad.numpy(), rtol=1e-8, atol=1e-12) and np.allclose(db, b.grad.numpy(), rtol=1e-8, atol=1e-12)
    for k in ("mean", "var"):
        assert np.allclose(gr[k][0], tg[k][0].grad.numpy(), rtol=1e-8, atol=1e-12)
        assert np.allclose(gr[k][1], tg[k][1].grad.numpy(), rtol=1e-8, atol=1e-12)


def test_bgm_hmc_samples_known_gaussian_posterior():
    """Known-answer test of the HMC restatement: with all features masked out the posterior is the
    N(0, I) prior; sample mean / variance must match and the step size must adapt upwards."""
    OB, m, z, x, mask = _bgm_setup(seed=5, n=400, p=7, q=3)
    out, info = OB.hmc_sampler(m, x, np.zeros_like(mask), n_mcmc=150, burn_in=100, step_size=0.05, n_leapfrog=5,
                               seed=9, return_info=True)
    flat = out.reshape(-1, 3)
    assert np.abs(flat.mean(0)).max() < 0.05 and np.abs(flat.var(0) - 1).max() < 0.1
    assert info["step"] > 0.05 and 0.5 < info["accept_rate"] <= 1.0


# ---------------------------------------------------------------------------------------------
# EGM warm start (oracle/egm.py): WGAN-GP double backward through batch-statistics BatchNorm
# ---------------------------------------------------------------------------------------------
def _egm_setup(rs, q=10, pdim=23):
    from oracle import egm as OE
    from oracle import nets as N
    nets = {"g": N.init_mlp(rs, [q, 64, 64, pdim + 1], np.float64), "e": N.init_mlp(rs, [pdim, 64, 64, q], np.float64),
            "f": N.init_mlp(rs, [3, 64, 32, 8, 2], np.float64), "h": N.init_mlp(rs, [2, 64, 32, 8, 2], np.float64)}
    for k in nets:
        nets[k] = [(W, 0.1 * rs.randn(*b.shape)) for W, b in nets[k]]
    dz = OE.init_disc(rs, q, [64, 32, 8], np.float64)
    dz["b"] = [0.1 * rs.randn(*b.shape) for b in dz["b"]]
    dz["gamma"] = [1 + 0.2 * rs.randn(*b.shape) for b in dz["gamma"]]
    dz["beta"] = [0.1 * rs.randn(*b.shape) for b in dz["beta"]]
    return nets, dz


def _tdisc(d, x):
    h = x
    L = len(d["gamma"])
    for l in range(L):
        h = h @ d["W"][l] + d["b"][l]
        h = torch.tanh((h - h.mean(0)) / torch.sqrt(h.var(0, unbiased=False) + 1e-3) * d["gamma"][l] + d["beta"][l])
    return h @ d["W"][L] + d["b"][L]


def _max_err(ours, theirs):
    scale = max(float(np.abs(b).max()) for b in theirs)
    return max(float(np.abs(a - b).max()) for a, b in zip(ours, theirs)) / scale


def test_egm_disc_step_double_backward_matches_autograd():
    from oracle import egm as OE
    rs = np.random.RandomState(0)
    nets, dz = _egm_setup(rs)
    B, q, pdim = 32, 10, 23
    z, v = rs.randn(B, q), rs.randn(B, pdim)
    eps = 0.37
    tdz = {k: [_t(a).requires_grad_() for a in vv] for k, vv in dz.items()}
    te = [(_t(W), _t(b)) for W, b in nets["e"]]
    z_ = _fwd(te, _t(v))
    zhat = (_t(z) * eps + z_ * (1 - eps)).requires_grad_(True)
    dz_loss = -_tdisc(tdz, _t(z)).mean() + _tdisc(tdz, z_).mean()
    (gz,) = torch.autograd.grad(_tdisc(tdz, zhat).sum(), zhat, create_graph=True)
    gp = ((torch.sqrt((gz ** 2).sum(1)) - 1) ** 2).mean()
    d_loss = dz_loss + 10 * gp
    plist = tdz["W"] + tdz["b"] + tdz["gamma"] + tdz["beta"]
    tg = [g.numpy() for g in torch.autograd.grad(d_loss, plist)]
    l1, l2, gr = OE.disc_step_grads(nets, dz, z, v, eps)
    assert abs(l1 - dz_loss.item()) < 1e-12 and abs(l2 - d_loss.item()) < 1e-12
    assert _max_err(OE.disc_param_list(gr), tg) < 1e-12
    # the penalty alone, on its own batch
    zh = rs.randn(B, q)
    tzh = _t(zh).requires_grad_(True)
    (gz,) = torch.autograd.grad(_tdisc(tdz, tzh).sum(), tzh, create_graph=True)
    gp = ((torch.sqrt((gz ** 2).sum(1)) - 1) ** 2).mean()
    tg = [np.zeros(tuple(p_.shape)) if g is None else g.numpy() for g, p_ in zip(torch.autograd.grad(gp, plist, allow_unused=True), plist)]
    gpo, gr = OE.gradient_penalty_and_grads(dz, zh)
    assert abs(gpo - gp.item()) < 1e-12 and _max_err(OE.disc_param_list(gr), tg) < 1e-12


def _tdisc_fixed(d, x):
    h = x
    L = len(d["gamma"])
    for l in range(L):
        h = h @ d["W"][l] + d["b"][l]
        h = torch.tanh(h / np.sqrt(1.0 + 1e-3) * d["gamma"][l] + d["beta"][l])
    return h @ d["W"][L] + d["b"][L]


def test_egm_disc_step_fixed_norm_matches_autograd():
    """`disc_norm='fixed'`: BatchNormalization in inference mode on its initial moving averages (a constant scale)."""
    from oracle import egm as OE
    rs = np.random.RandomState(3)
    nets, dz = _egm_setup(rs)
    dz["fixed_norm"] = True
    B, q, pdim = 32, 10, 23
    z, v = rs.randn(B, q), rs.randn(B, pdim)
    eps = 0.61
    tdz = {k: [_t(a).requires_grad_() for a in vv] for k, vv in dz.items() if isinstance(vv, list)}
    te = [(_t(W), _t(b)) for W, b in nets["e"]]
    z_ = _fwd(te, _t(v))
    zhat = (_t(z) * eps + z_ * (1 - eps)).requires_grad_(True)
    dz_loss = -_tdisc_fixed(tdz, _t(z)).mean() + _tdisc_fixed(tdz, z_).mean()
    (gz,) = torch.autograd.grad(_tdisc_fixed(tdz, zhat).sum(), zhat, create_graph=True)
    gp = ((torch.sqrt((gz ** 2).sum(1)) - 1) ** 2).mean()
    d_loss = dz_loss + 10 * gp
    plist = tdz["W"] + tdz["b"] + tdz["gamma"] + tdz["beta"]
    tg = [np.zeros(tuple(p_.shape)) if g is None else g.numpy()
          for g, p_ in zip(torch.autograd.grad(d_loss, plist, allow_unused=True), plist)]
    l1, l2, gr = OE.disc_step_grads(nets, dz, z, v, eps)
    assert abs(l1 - dz_loss.item()) < 1e-12 and abs(l2 - d_loss.item()) < 1e-12
    assert _max_err(OE.disc_param_list(gr), tg) < 1e-12
    # generator side: d(-mean D(e(v)))/d e(v) through the fixed-norm discriminator
    p = dict(v_dim=pdim, z_dims=[1, 1, 1, 7], binary_treatment=False, use_z_rec=True, lr=2e-4)
    x, y = rs.rand(B, 1), rs.randn(B, 1)
    tn = {k: _torch_net(nets[k]) for k in nets}
    tdz2 = {k: [_t(a) for a in vv] for k, vv in dz.items() if isinstance(vv, list)}
    tz, tv, tx, ty = _t(z), _t(v), _t(x), _t(y)
    g_z = _fwd(tn["g"], tz)
    z_ = _fwd(tn["e"], tv)
    z__ = _fwd(tn["e"], g_z[:, :pdim])
    v__ = _fwd(tn["g"], z_)[:, :pdim]
    fo = _fwd(tn["f"], torch.cat([z_[:, :1], z_[:, 1:2], tx], 1))
    ho = _fwd(tn["h"], torch.cat([z_[:, :1], z_[:, 2:3]], 1))
    sig = (g_z[:, -1] ** 2).mean() + (fo[:, -1] ** 2).mean() + (ho[:, -1] ** 2).mean()
    loss = (-_tdisc_fixed(tdz2, z_).mean() + ((tv - v__) ** 2).mean() + ((tz - z__) ** 2).mean() + ((ho[:, :1] - tx) ** 2).mean() +
            ((fo[:, :1] - ty) ** 2).mean() + 0.001 * sig)
    pl = [a for k in ("g", "e", "f", "h") for Wb in tn[k] for a in Wb]
    tg = [g.numpy() for g in torch.autograd.grad(loss, pl)]
    losses, gr = OE.gen_step_grads(nets, dz, p, z, v, x, y)
    assert abs(losses[-1] - loss.item()) < 1e-12
    assert _max_err(OE.gen_param_list(gr), tg) < 1e-12


def test_egm_gen_step_gradients_match_autograd():
    from oracle import egm as OE
    rs = np.random.RandomState(1)
    nets, dz = _egm_setup(rs)
    B, q, pdim = 32, 10, 23
    z, v, x, y = rs.randn(B, q), rs.randn(B, pdim), rs.rand(B, 1), rs.randn(B, 1)
    for binary in (False, True):
        p = dict(v_dim=pdim, z_dims=[1, 1, 1, 7], binary_treatment=binary, use_z_rec=True, lr=2e-4)
        xx = (x > 0.5).astype(np.float64) if binary else x
        tn = {k: _torch_net(nets[k]) for k in nets}
        tdz = {k: [_t(a) for a in vv] for k, vv in dz.items()}
        tz, tv, tx, ty = _t(z), _t(v), _t(xx), _t(y)
        g_z = _fwd(tn["g"], tz)
        z_ = _fwd(tn["e"], tv)
        z__ = _fwd(tn["e"], g_z[:, :pdim])
        v__ = _fwd(tn["g"], z_)[:, :pdim]
        fo = _fwd(tn["f"], torch.cat([z_[:, :1], z_[:, 1:2], tx], 1))
        ho = _fwd(tn["h"], torch.cat([z_[:, :1], z_[:, 2:3]], 1))
        sig = (g_z[:, -1] ** 2).mean() + (fo[:, -1] ** 2).mean() + (ho[:, -1] ** 2).mean()
        l2x = torch.nn.functional.binary_cross_entropy_with_logits(ho[:, :1], tx) if binary else ((ho[:, :1] - tx) ** 2).mean()
        loss = (-_tdisc(tdz, z_).mean() + ((tv - v__) ** 2).mean() + ((tz - z__) ** 2).mean() + l2x +
                ((fo[:, :1] - ty) ** 2).mean() + 0.001 * sig)
        pl = [a for k in ("g", "e", "f", "h") for Wb in tn[k] for a in Wb]
        tg = [g.numpy() for g in torch.autograd.grad(loss, pl)]
        losses, gr = OE.gen_step_grads(nets, dz, p, z, v, xx, y)
        assert abs(losses[-1] - loss.item()) < 1e-12 and abs(losses[3] - l2x.item()) < 1e-12
        assert _max_err(OE.gen_param_list(gr), tg) < 1e-12


def _bgm_egm_setup(rs, q=6, p=19):
    from oracle import egm as OE
    from oracle import nets as N
    g = N.init_varnet(rs, q, (64, 64, 64), p, np.float64)
    g["bn"].update(gamma=1 + 0.1 * rs.randn(q), beta=0.1 * rs.randn(q), mean=0.2 * rs.randn(q), var=0.5 + rs.rand(q))
    g["trunk"] = [(W, 0.1 * rs.randn(*b.shape)) for W, b in g["trunk"]]
    g["mean"] = (g["mean"][0], 0.1 * rs.randn(p)); g["var"] = (g["var"][0], 0.1 * rs.randn(p))
    e = [(W, 0.1 * rs.randn(*b.shape)) for W, b in N.init_mlp(rs, [p, 64, 64, q], np.float64)]
    ds = []
    for in_dim in (q, p):
        d = OE.init_disc(rs, in_dim, [64, 32, 8], np.float64)
        d["b"] = [0.1 * rs.randn(*b.shape) for b in d["b"]]
        d["gamma"] = [1 + 0.2 * rs.randn(*b.shape) for b in d["gamma"]]
        d["beta"] = [0.1 * rs.randn(*b.shape) for b in d["beta"]]
        ds.append(d)
    return g, e, ds[0], ds[1]


def _tg_train(tg, z):
    """BaseVariationalNet(training=True) in torch."""
    h = (z - z.mean(0)) / torch.sqrt(z.var(0, unbiased=False) + 1e-3) * tg["gamma"] + tg["beta"]
    for W, b in tg["trunk"]:
        h = h @ W + b
        h = torch.maximum(h, 0.2 * h)
    return h @ tg["mean"][0] + tg["mean"][1], _sp(h @ tg["var"][0] + tg["var"][1]) + 1e-6


def test_bgm_egm_steps_match_autograd():
    from oracle import egm as OE
    rs = np.random.RandomState(3)
    q, p, B = 6, 19, 32
    g, e, dz, dx = _bgm_egm_setup(rs, q, p)
    z, x, n1, n2 = rs.randn(B, q), rs.randn(B, p), rs.randn(B, p), rs.randn(B, p)
    r = lambda a: _t(a).requires_grad_()
    tg = {"gamma": r(g["bn"]["gamma"]), "beta": r(g["bn"]["beta"]), "trunk": [(r(W), r(b)) for W, b in g["trunk"]],
          "mean": (r(g["mean"][0]), r(g["mean"][1])), "var": (r(g["var"][0]), r(g["var"][1]))}
    te = _torch_net(e)
    tdz = {k: [r(a) for a in v] for k, v in dz.items()}
    tdx = {k: [r(a) for a in v] for k, v in dx.items()}
    tz, tx, tn1, tn2 = _t(z), _t(x), _t(n1), _t(n2)
    # ---- generator step
    alpha = 0.3
    mu1, s21 = _tg_train(tg, tz)
    x_ = tn1 * torch.sqrt(s21) + mu1
    z_ = _fwd(te, tx)
    z__ = _fwd(te, x_)
    mu2, s22 = _tg_train(tg, z_)
    x__ = tn2 * torch.sqrt(s22) + mu2
    loss = (((0.9 - _tdisc(tdx, x_)) ** 2).mean() + ((0.9 - _tdisc(tdz, z_)) ** 2).mean() +
            10 * (((tx - x__) ** 2).mean() + ((tz - z__) ** 2).mean()) + alpha * (s21 ** 2).mean())
    gl = [tg["gamma"], tg["beta"]] + [a for Wb in tg["trunk"] for a in Wb] + list(tg["mean"]) + list(tg["var"])
    el = [a for Wb in te for a in Wb]
    ref = [t_.numpy() for t_ in torch.autograd.grad(loss, gl + el)]
    losses, gr, _ = OE.bgm_gen_step_grads(g, e, dz, dx, z, x, n1, n2, alpha)
    ours = OE.g_grad_list(gr["g"]) + [a for Wb in gr["e"] for a in Wb]
    assert abs(losses[-1] - loss.item()) < 1e-12 and _max_err(ours, ref) < 1e-12
    # ---- discriminator step (gamma > 0: gradient penalties on both discriminators)
    gamma, ez, ex = 0.7, 0.31, 0.64
    with torch.no_grad():
        z_ = _fwd(te, tx)
        mu, s2 = _tg_train(tg, tz)
        x_ = tn1 * torch.sqrt(s2) + mu
    zh = (tz * ez + z_ * (1 - ez)).requires_grad_(True)
    xh = (tx * ex + x_ * (1 - ex)).requires_grad_(True)
    dz_loss = (((0.9 - _tdisc(tdz, tz)) ** 2).mean() + ((0.1 - _tdisc(tdz, z_)) ** 2).mean()) / 2
    dx_loss = (((0.9 - _tdisc(tdx, tx)) ** 2).mean() + ((0.1 - _tdisc(tdx, x_)) ** 2).mean()) / 2
    (gz,) = torch.autograd.grad(_tdisc(tdz, zh).sum(), zh, create_graph=True)
    (gx,) = torch.autograd.grad(_tdisc(tdx, xh).sum(), xh, create_graph=True)
    d_loss = dz_loss + dx_loss + gamma * (((torch.sqrt((gz ** 2).sum(1)) - 1) ** 2).mean() + ((torch.sqrt((gx ** 2).sum(1)) - 1) ** 2).mean())
    pl = tdz["W"] + tdz["b"] + tdz["gamma"] + tdz["beta"] + tdx["W"] + tdx["b"] + tdx["gamma"] + tdx["beta"]
    ref = [t_.numpy() for t_ in torch.autograd.grad(d_loss, pl)]
    losses, gr, _ = OE.bgm_disc_step_grads(g, e, dz, dx, z, x, n1, ez, ex, gamma)
    ours = OE.disc_param_list(gr["dz"]) + OE.disc_param_list(gr["dx"])
    assert abs(losses[2] - d_loss.item()) < 1e-12 and abs(losses[0] - dz_loss.item()) < 1e-12
    assert _max_err(ours, ref) < 1e-11
