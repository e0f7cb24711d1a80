"""Oracle restatement of BGM with a Bayesian generator, ``use_bnn=True`` (TEST INFRASTRUCTURE).

Follows /root/reference/src/bayesgm/models/networks/bnn.py:40-99 (BayesianVariationalNet) and the call sites of
``self.g_net`` in /root/reference/src/bayesgm/models/bgm/base.py (update_g_net :145-164 incl. the ``use_bnn``
KL term :155-157, update_latent_variable_sgd :167-187, train_disc_step :190-244, train_gen_step :246-289 -- whose KL
term is commented out :280-283 --, evaluate :444-476, generate :478-509, predict_on_posteriors :511-525,
get_log_posterior :665-705, tfp_mcmc_sampler :709-830).  Only g_net is Bayesian; e_net, dz_net, dx_net stay
deterministic (:67-79).

tensorflow-probability 0.18.0 / keras 2.10 are absent here -> PARITY UNPINNED; restated from their published semantics
(see oracle/bnn.py for DenseFlipout), with the differences of THIS network:

  * BatchNormalization on z honours ``training`` (bnn.py:85): batch statistics + moving-average update
    (momentum 0.99, biased variance) when training, moving statistics when ``training=False``.
  * hidden DenseFlipout stack with LeakyReLU(0.2), then TWO sibling DenseFlipout heads on the same trunk output
    (mean_layer, var_layer; var = softplus(.) + 1e-6), each with its own perturbation and its own sign vectors.
  * kernel prior N(0, 0.1^2) (kernel_prior_fn, bnn.py:55-58) and the SAME prior on the bias (bias_prior_fn): the bias
    posterior is a point mass, for which TFP's registered KL(Deterministic || p) is -log p(bias), i.e.
    sum b^2 / (2 * 0.01) + log(0.1) + log(2 pi) / 2 -- part of ``sum(g_net.losses)``.
  * DenseFlipout perturbs the kernel in EVERY call, also with ``training=False``: the log posterior that HMC sees, the
    predictive draws, evaluate and generate are all stochastic in the weights.

Noise: the layout of oracle/bnn.py (Philox4x32-10) with net id 0 and the layers ordered trunk..., mean head, var head;
the var head's input signs are its own (sign_layout over (in, out) pairs).

net = {"gamma","beta","mean_mv","var_mv": [q], "trunk": [(loc, rho, bias)...], "mean": (loc, rho, bias), "var": (...)}
"""
import numpy as np
from . import rng as R
from . import bnn as BN
from . import nets as N
from .nets import lrelu, softplus, sigmoid, LEAK, BN_EPS, BN_MOMENTUM

PRIOR_SCALE = 0.1
NET_ID = 0
EPS = 1e-6
STREAM_PREDICT = 0x40000000      # + row block: predict_on_posteriors
STREAM_DECODE = 0x50000000       # evaluate / generate


def init_vnet(rs, q, units, p, dtype=np.float32):
    def layer(i, o):
        return ((0.1 * rs.standard_normal((i, o))).astype(dtype), (-3.0 + 0.1 * rs.standard_normal((i, o))).astype(dtype),
                (0.1 * rs.standard_normal(o)).astype(dtype))
    dims = [q] + list(units)
    return {"gamma": np.ones(q, dtype), "beta": np.zeros(q, dtype), "mean_mv": np.zeros(q, dtype), "var_mv": np.ones(q, dtype),
            "trunk": [layer(dims[i], dims[i + 1]) for i in range(len(dims) - 1)],
            "mean": layer(dims[-1], p), "var": layer(dims[-1], p)}


def cast_vnet(net, dtype):
    out = {k: net[k].astype(dtype) for k in ("gamma", "beta", "mean_mv", "var_mv")}
    out["trunk"] = [tuple(a.astype(dtype) for a in L) for L in net["trunk"]]
    out["mean"] = tuple(a.astype(dtype) for a in net["mean"])
    out["var"] = tuple(a.astype(dtype) for a in net["var"])
    return out


def layers_of(net):
    return list(net["trunk"]) + [net["mean"], net["var"]]


def shapes(net):
    return [L[0].shape for L in layers_of(net)]


def draw(net, B, key, stream, row0=0, dtype=np.float32, rows=None):
    return BN.draw_noise(shapes(net), B, key, stream, NET_ID, dtype=dtype, row0=row0, rows=rows)


def _flip(h, layer, noise, l):
    loc, rho, bias = layer
    return h @ loc + ((h * noise["sin"][l]) @ (BN.scale_of(rho) * noise["eps"][l])) * noise["sout"][l] + bias


def vforward(net, z, noise, training=True, eps=EPS):
    """BayesianVariationalNet.call (bnn.py:83-95) -> (mean, var, cache)."""
    t = z.dtype.type
    if training:
        mu_b, var_b = z.mean(axis=0), z.var(axis=0)
    else:
        mu_b, var_b = net["mean_mv"], net["var_mv"]
    inv = 1.0 / np.sqrt(var_b + t(BN_EPS))
    zhat = (z - mu_b) * inv
    h = zhat * net["gamma"] + net["beta"]
    acts, pres = [h], []
    T = len(net["trunk"])
    for l, L in enumerate(net["trunk"]):
        pre = _flip(h, L, noise, l)
        pres.append(pre)
        h = lrelu(pre)
        acts.append(h)
    mean = _flip(h, net["mean"], noise, T)
    s_raw = _flip(h, net["var"], noise, T + 1)
    return mean, softplus(s_raw) + t(eps), dict(zhat=zhat, inv=inv, acts=acts, pres=pres, s_raw=s_raw, mu_b=mu_b,
                                                var_b=var_b, noise=noise, training=training)


def vbackward(net, c, dmean, ds_raw, want_dz=True):
    """Gradients for upstream dLoss/dmean, dLoss/d(s_raw) [B x p].  -> ({"gamma","beta","layers": [(dloc, drho, dbias)]}
    with layers ordered trunk..., mean, var; dz | None)."""
    noise = c["noise"]
    T = len(net["trunk"])
    Ls = layers_of(net)
    grads = [None] * (T + 2)
    h = c["acts"][-1]
    dh = 0.0
    for l, d in ((T + 1, ds_raw), (T, dmean)):
        loc, rho, _ = Ls[l]
        ds = d * noise["sout"][l]
        grads[l] = (h.T @ d, ((h * noise["sin"][l]).T @ ds) * noise["eps"][l] * sigmoid(rho), d.sum(axis=0))
        dh = dh + d @ loc.T + (ds @ (BN.scale_of(rho) * noise["eps"][l]).T) * noise["sin"][l]
    for l in reversed(range(T)):
        loc, rho, _ = Ls[l]
        dh = dh * np.where(c["pres"][l] > 0, 1.0, LEAK).astype(dh.dtype)
        a = c["acts"][l]
        ds = dh * noise["sout"][l]
        grads[l] = (a.T @ dh, ((a * noise["sin"][l]).T @ ds) * noise["eps"][l] * sigmoid(rho), dh.sum(axis=0))
        dh = dh @ loc.T + (ds @ (BN.scale_of(rho) * noise["eps"][l]).T) * noise["sin"][l]
    g = {"gamma": (dh * c["zhat"]).sum(axis=0), "beta": dh.sum(axis=0), "layers": grads}
    dz = None
    if want_dz:
        dzh = dh * net["gamma"]
        if c["training"]:
            dz = c["inv"] * (dzh - dzh.mean(axis=0) - c["zhat"] * (dzh * c["zhat"]).mean(axis=0))
        else:
            dz = dzh * c["inv"]
    return g, dz


def vkl(net):
    """sum(g_net.losses): kernel KL against N(0, 0.1^2) plus -log N(bias; 0, 0.1^2) per layer.  -> (value, grads)."""
    t = net["gamma"].dtype.type
    s = t(PRIOR_SCALE)
    s2 = s * s
    val = 0.0
    gl = []
    for loc, rho, bias in layers_of(net):
        sg = BN.scale_of(rho)
        val = val + (np.log(s / sg) + (sg * sg + loc * loc) / (2 * s2) - t(0.5)).sum()
        val = val + (bias * bias / (2 * s2) + np.log(s) + t(0.5 * np.log(2 * np.pi))).sum()
        gl.append((loc / s2, (-1.0 / sg + sg / s2) * sigmoid(rho), bias / s2))
    return val, {"gamma": np.zeros_like(net["gamma"]), "beta": np.zeros_like(net["beta"]), "layers": gl}


def flat_params(net):
    """The build's flat order: gamma, beta, moving mean, moving variance, then loc, rho, bias per layer (trunk, mean, var)."""
    return [net["gamma"], net["beta"], net["mean_mv"], net["var_mv"]] + [a for L in layers_of(net) for a in L]


def flat_grads(g):
    z = np.zeros_like(g["gamma"])
    return [g["gamma"], g["beta"], z, z.copy()] + [a for L in g["layers"] for a in L]


def move_stats(net, c):
    t = c["mu_b"].dtype.type
    net["mean_mv"] = net["mean_mv"] * t(BN_MOMENTUM) + c["mu_b"] * t(1 - BN_MOMENTUM)
    net["var_mv"] = net["var_mv"] * t(BN_MOMENTUM) + c["var_b"] * t(1 - BN_MOMENTUM)


def loss_and_grads(net, z, x, noise, inv_B=None):
    """loss_x of update_g_net (:148-153) / loss_px_z (:172-175) without the KL term, training-mode BN.
    -> (loss_x, loss_mse, grads, dz, cache)."""
    t = z.dtype.type
    Bn = len(z)
    w = t(1.0 / Bn if inv_B is None else inv_B)
    mean, s2, c = vforward(net, z, noise, training=True)
    d = x - mean
    loss_b = (d ** 2 / (2 * s2) + 0.5 * np.log(s2)).sum(axis=1)
    dmean = -d / s2 * w
    ds = (-d ** 2 / (2 * s2 * s2) + 0.5 / s2) * sigmoid(c["s_raw"]) * w
    g, dz = vbackward(net, c, dmean, ds)
    return loss_b.sum() * w, (d ** 2).mean(), g, dz, c


class FitState(object):
    """BGM.fit with use_bnn (bgm/base.py:399-413): theta step = stream 2t, Z step = stream 2t + 1 of the seed's key."""

    def __init__(self, net, data_z, lr_theta, lr_z, kl_weight, seed):
        from .fit import AdamState
        self.net, self.data_z, self.lr_theta, self.lr_z, self.kl_weight, self.seed = net, data_z, lr_theta, lr_z, kl_weight, seed
        self.opt = AdamState(flat_params(net))
        self.t = 0

    def step(self, data, idx):
        from .fit import adam_lr_t, B1, B2, ADAM_EPS
        net = self.net
        t = self.data_z.dtype.type
        B = len(idx)
        zb, xb = self.data_z[idx].copy(), data[idx]
        n1 = draw(net, B, self.seed, 2 * self.t, dtype=zb.dtype)
        loss_x, mse, g, _, c = loss_and_grads(net, zb, xb, n1)
        klv, gk = vkl(net)
        g = BN.add_grads(g, gk, t(self.kl_weight))
        move_stats(net, c)
        ps = flat_params(net)
        self.opt.apply(ps, flat_grads(g), self.lr_theta)
        n2 = draw(net, B, self.seed, 2 * self.t + 1, dtype=zb.dtype)
        lz, _, _, dz, c2 = loss_and_grads(net, zb, xb, n2)
        move_stats(net, c2)
        dz = dz + zb / t(B)
        loss_post = lz + ((zb ** 2).sum(axis=1) / 2).mean()
        self.t += 1
        lr_t = t(adam_lr_t(self.lr_z, self.t))
        m_, v_ = t(1 - B1) * dz, t(1 - B2) * dz * dz             # fresh slots every minibatch (:402)
        self.data_z[idx] = zb - lr_t * m_ / (np.sqrt(v_) + t(ADAM_EPS))
        return loss_x + t(self.kl_weight) * klv, mse, loss_post


# ---------------------------------------------------------------------------------------- inference (training=False)
def log_posterior_and_grad(net, z, x, mask, noise):
    """get_log_posterior (:665-705) and its gradient w.r.t. z for ONE call of g_net(z, training=False) with `noise`."""
    t = z.dtype.type
    mean, s2, c = vforward(net, z, noise, training=False)
    mk = np.ones_like(x) if mask is None else mask.astype(z.dtype)
    d = x - mean
    logp = -((mk * (d ** 2 / (2 * s2) + 0.5 * np.log(s2))).sum(axis=1) + (z ** 2).sum(axis=1) / 2)
    dmu = mk * d / s2
    ds = mk * (d ** 2 / (2 * s2 * s2) - 0.5 / s2) * sigmoid(c["s_raw"])
    _, dz = vbackward(net, c, dmu, ds)
    return logp, dz - z


def hmc_stream(it, leap, n_leapfrog):
    """Noise stream of the gradient evaluation of leapfrog step `leap` of transition `it`; stream 0 = bootstrap."""
    return 1 + it * n_leapfrog + leap


def hmc_sampler(net, x, mask, n_mcmc, burn_in, step_size=0.01, n_leapfrog=10, seed=42, row0=0, return_info=False, frozen=False):
    """tfp_mcmc_sampler (:709-830) on the stochastic target: every evaluation is one g_net call over ALL rows (one
    perturbation shared by the rows, per-row signs keyed by the global row); the cached log-prob / gradient of the current
    state are NOT refreshed (TFP caches them in the kernel results).  frozen=True (build option): every evaluation reuses
    the perturbation of call 0, i.e. HMC on the deterministic target of ONE weight draw."""
    from . import bgm as OB
    n, q = len(x), len(net["gamma"])
    z = OB.hmc_init_state(n, q, seed, row0).astype(x.dtype)
    t = z.dtype.type
    rows = np.arange(row0, row0 + n)
    lp, gr = log_posterior_and_grad(net, z, x, mask, draw(net, n, seed, 0, row0, z.dtype))
    n_adapt = int(burn_in * 0.8)
    step = float(step_size)
    out, n_acc, steps = [], 0, []
    for it in range(burn_in + n_mcmc):
        mom = R.normals(rows, it, q, R.TAG_MOM, seed).astype(z.dtype)
        u = R.uniforms(rows, it, R.TAG_HACC, seed).astype(z.dtype)
        e = t(step)
        h0 = -lp + (mom ** 2).sum(axis=1) / 2
        zc, pc = z.copy(), mom + e / 2 * gr
        for l in range(n_leapfrog):
            zc = zc + e * pc
            lpc, grc = log_posterior_and_grad(net, zc, x, mask, draw(net, n, seed, 0 if frozen else hmc_stream(it, l, n_leapfrog), row0, z.dtype))
            pc = pc + (e if l < n_leapfrog - 1 else e / 2) * grc
        h1 = -lpc + (pc ** 2).sum(axis=1) / 2
        lr = -(h1 - h0)
        lr = np.where(np.isfinite(lr), lr, -np.inf)
        acc = np.log(u) < lr
        z = np.where(acc[:, None], zc, z)
        lp = np.where(acc, lpc, lp)
        gr = np.where(acc[:, None], grc, gr)
        steps.append(step)
        if it < n_adapt:
            step = OB.adapt_step(step, lr)
        if it >= burn_in:
            out.append(z.copy())
            n_acc += int(acc.sum())
    out = np.array(out)
    if return_info:
        return out, dict(step=step, accept_rate=n_acc / max(1, n_mcmc * n), steps=np.array(steps))
    return out


def decode(net, z, seed, stream, row0=0, rows=None):
    """g_net(z, training=False) -> (mean, var) for one call; sign rows keyed row0 + i (or rows[i])."""
    mean, s2, _ = vforward(net, z, draw(net, len(z), seed, stream, row0, z.dtype, rows=rows), training=False)
    return mean, s2


def predict_on_posteriors(net, post_z, seed, block=0, row0=0, burn_in=0, bs=None, off=0):
    """predict_on_posteriors (:511-525) for (a part of) one row block of `bs` rows: ONE g_net call over the flattened
    [n_mcmc * n] rows, stream STREAM_PREDICT + block; draw d of the block's row r (r = off + i for the i-th row handed in)
    keys its signs with d * bs + r -- independent of how the block is split over ranks.  x-noise as in oracle.bgm (tag 6,
    global row row0 + i)."""
    n_mcmc, n, q = post_z.shape
    bs = n if bs is None else bs
    p = net["mean"][0].shape[1]
    flat = post_z.reshape(n_mcmc * n, q)
    ids = (np.arange(n_mcmc)[:, None] * bs + off + np.arange(n)[None, :]).reshape(-1)
    mean, s2 = decode(net, flat, seed, STREAM_PREDICT + block, rows=ids)
    rows = np.arange(row0, row0 + n)
    out = np.empty((n_mcmc, n, p), dtype=post_z.dtype)
    for d in range(n_mcmc):
        eps = R.normals_seq(rows, burn_in + d, p, R.TAG_XNOISE, seed).astype(post_z.dtype)
        out[d] = mean[d * n:(d + 1) * n] + np.sqrt(s2[d * n:(d + 1) * n]) * eps
    return out


# ---------------------------------------------------------------------------------------- EGM warm start
def egm_gen_step_grads(g, e, dz, dx, z, x, n1, n2, alpha, noise1, noise2):
    """train_gen_step (:246-289) with the Bayesian generator (no KL term, :280-283 is commented out)."""
    from .egm import disc_forward, disc_backward
    B, q = z.shape
    p = x.shape[1]
    mu1, s21, c1 = vforward(g, z, noise1, training=True)
    x_ = n1 * np.sqrt(s21) + mu1
    reg = (s21 ** 2).mean()
    z_, ce1 = N.mlp_forward_cache(e, x)
    z__, ce2 = N.mlp_forward_cache(e, x_)
    mu2, s22, c2 = vforward(g, z_, noise2, training=True)
    x__ = n2 * np.sqrt(s22) + mu2
    dxo, cdx = disc_forward(dx, x_)
    dzo, cdz = disc_forward(dz, z_)
    l2_x = ((x - x__) ** 2).mean()
    l2_z = ((z - z__) ** 2).mean()
    g_adv = ((0.9 - dxo) ** 2).mean()
    e_adv = ((0.9 - dzo) ** 2).mean()
    total = g_adv + e_adv + 10.0 * (l2_x + l2_z) + alpha * reg
    dx__ = 10.0 * (-2.0 / (B * p)) * (x - x__)
    gg2, dz_ = vbackward(g, c2, dx__, dx__ * n2 * 0.5 / np.sqrt(s22) * sigmoid(c2["s_raw"]))
    dz__ = 10.0 * (-2.0 / (B * q)) * (z - z__)
    ge2, dx_ = N.mlp_backward(e, ce2, dz__)
    _, dx_d = disc_backward(dx, cdx, -2.0 * (0.9 - dxo) / B)
    dx_ = dx_ + dx_d
    ds21 = dx_ * n1 * 0.5 / np.sqrt(s21) + alpha * 2.0 * s21 / (B * p)
    gg1, _ = vbackward(g, c1, dx_, ds21 * sigmoid(c1["s_raw"]), want_dz=False)
    _, dz_d = disc_backward(dz, cdz, -2.0 * (0.9 - dzo) / B)
    ge1, _ = N.mlp_backward(e, ce1, dz_ + dz_d)
    grads = {"g": BN.add_grads(gg1, gg2), "e": [(wa + wb, ba + bb) for (wa, ba), (wb, bb) in zip(ge1, ge2)]}
    return np.array([g_adv, e_adv, l2_z, l2_x, reg, total]), grads, [c1, c2]


def egm_disc_step_grads(g, e, dz, dx, z, x, n1, eps_z, eps_x, gamma, noise1):
    """train_disc_step (:190-244) with the Bayesian generator."""
    from .egm import disc_forward, disc_backward, zero_disc_grads, gradient_penalty_and_grads
    B = z.shape[0]
    z_ = N.mlp_forward(e, x)
    mu, s2, c = vforward(g, z, noise1, training=True)
    x_ = n1 * np.sqrt(s2) + mu
    gz, gx = zero_disc_grads(dz), zero_disc_grads(dx)
    losses = []
    for d, gr, real, fake in ((dz, gz, z, z_), (dx, gx, x, x_)):
        o_r, c_r = disc_forward(d, real)
        o_f, c_f = disc_forward(d, fake)
        losses.append((((0.9 - o_r) ** 2).mean() + ((0.1 - o_f) ** 2).mean()) / 2.0)
        disc_backward(d, c_r, -(0.9 - o_r) / B, gr)
        disc_backward(d, c_f, -(0.1 - o_f) / B, gr)
    d_loss = losses[0] + losses[1]
    if gamma != 0.0:
        gpz, _ = gradient_penalty_and_grads(dz, z * eps_z + z_ * (1.0 - eps_z), gz, scale=gamma)
        gpx, _ = gradient_penalty_and_grads(dx, x * eps_x + x_ * (1.0 - eps_x), gx, scale=gamma)
        d_loss = d_loss + gamma * (gpz + gpx)
    return np.array([losses[0], losses[1], d_loss]), {"dz": gz, "dx": gx}, [c]


class EgmState(object):
    """g (Bayesian, training-mode BN incl. moving statistics), e, dz, dx and the two Adam(lr, 0.5, 0.9) optimizers.
    Noise streams of step s (0-based count of ALL steps of the session): generator calls 2s, 2s + 1."""

    def __init__(self, g, e, dz, dx, params, seed):
        from .egm import Adam, disc_param_list
        self.g, self.e, self.dz, self.dx, self.p, self.seed = g, e, dz, dx, params, seed
        self.g_opt = Adam(flat_params(g) + [a for Wb in e for a in Wb], params["lr"], 0.5, 0.9)
        self.d_opt = Adam(disc_param_list(dz) + disc_param_list(dx), params["lr"], 0.5, 0.9)
        self.s = 0

    def disc_step(self, z, x, n1, eps_z, eps_x):
        from .egm import disc_param_list
        no = draw(self.g, len(z), self.seed, 2 * self.s, dtype=z.dtype)
        losses, gr, caches = egm_disc_step_grads(self.g, self.e, self.dz, self.dx, z, x, n1, eps_z, eps_x, self.p["gamma"], no)
        for c in caches:
            move_stats(self.g, c)
        self.d_opt.step(disc_param_list(gr["dz"]) + disc_param_list(gr["dx"]))
        self.s += 1
        return losses

    def gen_step(self, z, x, n1, n2):
        no1 = draw(self.g, len(z), self.seed, 2 * self.s, dtype=z.dtype)
        no2 = draw(self.g, len(z), self.seed, 2 * self.s + 1, dtype=z.dtype)
        losses, gr, caches = egm_gen_step_grads(self.g, self.e, self.dz, self.dx, z, x, n1, n2, self.p["alpha"], no1, no2)
        for c in caches:
            move_stats(self.g, c)
        ps = flat_params(self.g) + [a for Wb in self.e for a in Wb]
        self.g_opt.step(flat_grads(gr["g"]) + [a for Wb in gr["e"] for a in Wb])
        self.s += 1
        return losses
