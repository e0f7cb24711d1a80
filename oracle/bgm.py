"""Oracle restatement of the BGM hot path (TEST INFRASTRUCTURE).

Follows /root/reference/src/bayesgm/models/bgm/base.py (cited per function) for deterministic
networks (``use_bnn=False``): g_net = BaseVariationalNet (networks/base.py:53-117).
TFP semantics (tensorflow-probability 0.18, not installable here -> "parity unpinned") restated from
its public documentation:
  tfp.mcmc.HamiltonianMonteCarlo(step_size, num_leapfrog_steps): identity mass, momentum ~ N(0,I),
    leapfrog  p += e/2 g(z); [z += e p; p += e g(z)] x L with the last kick halved; accept iff
    log u < -(H1 - H0), H = -logp(z) + |p|^2/2, independently per chain (row);
  tfp.mcmc.SimpleStepSizeAdaptation(num_adaptation_steps, target_accept_prob=0.75,
    adaptation_rate=0.01): one scalar step size shared by all chains; after each of the first
    num_adaptation_steps transitions  step *= (1+rate) if logmeanexp_chains(min(0, log_accept_ratio))
    > log(target) else step /= (1+rate);
  tfp.mcmc.sample_chain(num_results, num_burnin_steps): burn-in transitions discarded, then one
    result per transition.
RNG: the build's Philox spec (oracle/rng.py) -- tag 0 initial state (bgm/base.py:778), tag 4 momentum,
tag 5 accept uniform, tag 6 posterior-predictive noise (networks/base.py:113-117).
"""
import numpy as np
from . import rng as R
from .nets import (init_varnet, varnet_forward, varnet_bn_affine, softplus, sigmoid, lrelu, LEAK,  # noqa: F401
                   BN_EPS, BN_MOMENTUM)

EPS = 1e-6


def init_model(seed, z_dim, x_dim, g_units=(64,) * 5, dtype=np.float32):
    rs = np.random.RandomState(seed)
    return {"z_dim": int(z_dim), "x_dim": int(x_dim), "g": init_varnet(rs, z_dim, g_units, x_dim, dtype)}


def cast_model(m, dtype):
    g = m["g"]
    out = dict(m)
    out["g"] = {"bn": {k: v.astype(dtype) for k, v in g["bn"].items()},
                "trunk": [(W.astype(dtype), b.astype(dtype)) for W, b in g["trunk"]],
                "mean": (g["mean"][0].astype(dtype), g["mean"][1].astype(dtype)),
                "var": (g["var"][0].astype(dtype), g["var"][1].astype(dtype))}
    return out


def obs_mask_of(data):
    """bgm/base.py:578-592: observed = not NaN; missing values are fed as 0 and ignored."""
    miss = np.isnan(data)
    return (~miss), np.where(miss, 0.0, data).astype(data.dtype)


def log_posterior(m, z, x, mask=None):
    """get_log_posterior, bgm/base.py:665-705 (g_net(training=False) -> BN moving statistics).
    mask [n, p] of {0,1}: the reference gathers observed features by index and multiplies padded
    positions by obs_mask (:689-700), algebraically a 0/1 mask over the p features."""
    mu, s2 = varnet_forward(m["g"], z, training=False)
    ll = (x - mu) ** 2 / (2 * s2) + 0.5 * np.log(s2)
    if mask is not None:
        ll = ll * mask
    return -(ll.sum(axis=1) + (z ** 2).sum(axis=1) / 2)


def log_posterior_and_grad(m, z, x, mask=None):
    """(logp [n], dlogp/dz [n, q]) by hand-derived backward through the inference-mode net."""
    g = m["g"]
    t = z.dtype.type
    scale, shift = varnet_bn_affine(g)
    zn = z * scale + shift
    acts, pres = [zn], []
    h = zn
    for W, b in g["trunk"]:
        p_ = h @ W + b
        pres.append(p_)
        h = lrelu(p_)
        acts.append(h)
    mu = h @ g["mean"][0] + g["mean"][1]
    s_raw = h @ g["var"][0] + g["var"][1]
    s2 = softplus(s_raw) + t(EPS)
    mk = np.ones_like(x) if mask is None else mask.astype(z.dtype)
    d = x - mu
    logp = -((mk * (d ** 2 / (2 * s2) + 0.5 * np.log(s2))).sum(axis=1) + (z ** 2).sum(axis=1) / 2)
    dmu = mk * d / s2                                   # d logp / d mu
    ds = mk * (d ** 2 / (2 * s2 * s2) - 0.5 / s2) * sigmoid(s_raw)
    dh = dmu @ g["mean"][0].T + ds @ g["var"][0].T
    for i in reversed(range(len(g["trunk"]))):
        dh = dh * np.where(pres[i] > 0, 1.0, LEAK).astype(z.dtype)
        dh = dh @ g["trunk"][i][0].T
    return logp, dh * scale - z


def hmc_init_state(n, q, seed, row0=0):
    return R.normals(np.arange(row0, row0 + n), 0, q, R.TAG_INIT, seed)


def hmc_transition(m, z, x, mask, step, n_leapfrog, it, seed, row0=0, lp=None, gr=None):
    """One HamiltonianMonteCarlo.one_step for all chains.  Returns (z, lp, gr, log_accept_ratio, accepted)."""
    n, q = z.shape
    t = z.dtype.type
    rows = np.arange(row0, row0 + n)
    if lp is None:
        lp, gr = log_posterior_and_grad(m, z, x, mask)
    mom = R.normals(rows, it, q, R.TAG_MOM, seed).astype(z.dtype)
    u = R.uniforms(rows, it, R.TAG_HACC, seed).astype(z.dtype)
    e = t(step)
    h0 = -lp + (mom ** 2).sum(axis=1) / 2
    zc, pc = z.copy(), mom + e / 2 * gr
    lpc, grc = lp, gr
    for l in range(n_leapfrog):
        zc = zc + e * pc
        lpc, grc = log_posterior_and_grad(m, zc, x, mask)
        pc = pc + (e if l < n_leapfrog - 1 else e / 2) * grc
    h1 = -lpc + (pc ** 2).sum(axis=1) / 2
    log_ratio = -(h1 - h0)
    log_ratio = np.where(np.isfinite(log_ratio), log_ratio, -np.inf)
    acc = np.log(u) < log_ratio
    z = np.where(acc[:, None], zc, z)
    lp = np.where(acc, lpc, lp)
    gr = np.where(acc[:, None], grc, gr)
    return z, lp, gr, log_ratio, acc


def adapt_step(step, log_ratio, target=0.75, rate=0.01):
    """SimpleStepSizeAdaptation update with reduce_logmeanexp over chains."""
    lap = np.minimum(log_ratio.astype(np.float64), 0.0)
    mx = lap.max()
    log_mean = mx + np.log(np.mean(np.exp(lap - mx))) if np.isfinite(mx) else -np.inf
    return step * (1.0 + rate) if log_mean > np.log(target) else step / (1.0 + rate)


def hmc_sampler(m, x, mask, n_mcmc, burn_in, step_size=0.01, n_leapfrog=10, seed=42, row0=0,
                return_info=False):
    """tfp_mcmc_sampler, bgm/base.py:709-830 -> samples [n_mcmc, n, q]."""
    n = len(x)
    q = m["z_dim"]
    z = hmc_init_state(n, q, seed, row0).astype(x.dtype)
    lp, gr = log_posterior_and_grad(m, z, x, mask)
    n_adapt = int(burn_in * 0.8)
    step = float(step_size)
    out, n_acc, steps = [], 0, []
    for it in range(burn_in + n_mcmc):
        z, lp, gr, lr, acc = hmc_transition(m, z, x, mask, step, n_leapfrog, it, seed, row0, lp, gr)
        steps.append(step)
        if it < n_adapt:
            step = adapt_step(step, lr)
        if it >= burn_in:
            out.append(z.copy())
            n_acc += int(acc.sum())
    out = np.array(out)
    if return_info:
        return out, dict(step=step, accept_rate=n_acc / max(1, n_mcmc * n), steps=np.array(steps))
    return out


def predict_on_posteriors(m, post_z, seed, row0=0, burn_in=0):
    """bgm/base.py:511-525: x ~ N(mu(z), sigma^2(z)); noise = Philox tag 6 at iteration burn_in + d,
    sequential layout over the p features."""
    n_mcmc, n, _ = post_z.shape
    p = m["x_dim"]
    rows = np.arange(row0, row0 + n)
    out = np.empty((n_mcmc, n, p), dtype=post_z.dtype)
    for d in range(n_mcmc):
        mu, s2 = varnet_forward(m["g"], post_z[d], training=False)
        eps = R.normals_seq(rows, burn_in + d, p, R.TAG_XNOISE, seed).astype(post_z.dtype)
        out[d] = mu + np.sqrt(s2) * eps
    return out


def predict(m, data, alpha=0.05, n_mcmc=5000, burn_in=5000, step_size=0.01, n_leapfrog=10, seed=42,
            return_samples=False):
    """predict, bgm/base.py:527-663."""
    assert 0 < alpha < 1
    obs, clean = obs_mask_of(data)
    mask = obs.astype(data.dtype)
    post = hmc_sampler(m, clean, mask, n_mcmc, burn_in, step_size, n_leapfrog, seed)
    pred = predict_on_posteriors(m, post, seed, burn_in=burn_in)
    miss = ~obs
    same = np.all(miss == miss[0])
    if same:
        mi = np.where(miss[0])[0]
        if mi.size == 0:
            interval = np.zeros((len(data), 0, 2), np.float32)
        else:
            ds = pred[:, :, mi]
            interval = np.stack([np.quantile(ds, alpha / 2, axis=0), np.quantile(ds, 1 - alpha / 2, axis=0)], -1)
    else:
        interval = []
        for i in range(len(data)):
            mi = np.where(miss[i])[0]
            if mi.size == 0:
                interval.append(np.zeros((0, 2), np.float32))
                continue
            ds = pred[:, i, mi]
            interval.append(np.stack([np.quantile(ds, alpha / 2, axis=0), np.quantile(ds, 1 - alpha / 2, axis=0)], -1))
    if return_samples:
        return pred, interval
    imputed = pred.mean(axis=0)
    imputed = miss * imputed + obs * clean
    return imputed, interval


# ---------------------------------------------------------------------------- fit
def g_train_forward(g, z, eps=EPS):
    """BaseVariationalNet.call(training=True): BN with batch statistics.  Returns outputs + cache."""
    bn = g["bn"]
    t = z.dtype.type
    mu_b = z.mean(axis=0)
    var_b = z.var(axis=0)
    inv = 1.0 / np.sqrt(var_b + t(BN_EPS))
    zhat = (z - mu_b) * inv
    zn = zhat * bn["gamma"] + bn["beta"]
    acts, pres = [zn], []
    h = zn
    for W, b in g["trunk"]:
        p_ = h @ W + b
        pres.append(p_)
        h = lrelu(p_)
        acts.append(h)
    mean = h @ g["mean"][0] + g["mean"][1]
    s_raw = h @ g["var"][0] + g["var"][1]
    return mean, softplus(s_raw) + t(eps), dict(zhat=zhat, inv=inv, acts=acts, pres=pres, s_raw=s_raw,
                                                mu_b=mu_b, var_b=var_b)


def g_loss_and_grads(m, z, x, want="theta"):
    """loss_x of update_g_net (bgm/base.py:148-153) / loss_px_z (:172-175), batch mean, training-mode BN.
    Returns (loss_x, loss_mse, grads dict, dz)."""
    g = m["g"]
    t = z.dtype.type
    Bn = len(z)
    mean, s2, c = g_train_forward(g, z)
    d = x - mean
    loss_b = (d ** 2 / (2 * s2) + 0.5 * np.log(s2)).sum(axis=1)
    dmean = -d / s2 / t(Bn)
    ds = (-d ** 2 / (2 * s2 * s2) + 0.5 / s2) * sigmoid(c["s_raw"]) / t(Bn)
    h = c["acts"][-1]
    grads = {"mean": (h.T @ dmean, dmean.sum(0)), "var": (h.T @ ds, ds.sum(0)), "trunk": []}
    dh = dmean @ g["mean"][0].T + ds @ g["var"][0].T
    tg = [None] * len(g["trunk"])
    for i in reversed(range(len(g["trunk"]))):
        dh = dh * np.where(c["pres"][i] > 0, 1.0, LEAK).astype(z.dtype)
        tg[i] = (c["acts"][i].T @ dh, dh.sum(0))
        dh = dh @ g["trunk"][i][0].T
    grads["trunk"] = tg
    dzn = dh
    grads["gamma"] = (dzn * c["zhat"]).sum(0)
    grads["beta"] = dzn.sum(0)
    dzhat = dzn * g["bn"]["gamma"]
    # batch-norm backward
    dz = c["inv"] * (dzhat - dzhat.mean(0) - c["zhat"] * (dzhat * c["zhat"]).mean(0))
    return loss_b.mean(), (d ** 2).mean(), grads, dz, c


def bn_update_stats(g, c):
    t = c["mu_b"].dtype.type
    g["bn"]["mean"] = g["bn"]["mean"] * t(BN_MOMENTUM) + c["mu_b"] * t(1 - BN_MOMENTUM)
    g["bn"]["var"] = g["bn"]["var"] * t(BN_MOMENTUM) + c["var_b"] * t(1 - BN_MOMENTUM)


class BgmFitState(object):
    """Optimizer state of BGM.fit (bgm/base.py:88-89, :390)."""

    def __init__(self, m, data_z, lr_theta, lr_z):
        from .fit import AdamState
        self.m, self.data_z, self.lr_theta, self.lr_z = m, data_z, lr_theta, lr_z
        self.opt = AdamState(self.params())
        self.zt = 0

    def params(self):
        g = self.m["g"]
        out = [g["bn"]["gamma"], g["bn"]["beta"]]
        for W, b in g["trunk"]:
            out += [W, b]
        return out + [g["mean"][0], g["mean"][1], g["var"][0], g["var"][1]]


def _flat_bgm_grads(gr):
    out = [gr["gamma"], gr["beta"]]
    for dW, db in gr["trunk"]:
        out += [dW, db]
    return out + [gr["mean"][0], gr["mean"][1], gr["var"][0], gr["var"][1]]


def fit_step(st, data, idx):
    """One minibatch of the loop body bgm/base.py:399-413.  Returns (loss_x, loss_mse_x, loss_postrior_z)."""
    from .fit import adam_lr_t, B1, B2, ADAM_EPS
    m = st.m
    t = st.data_z.dtype.type
    zb, xb = st.data_z[idx].copy(), data[idx]
    loss_x, mse_x, gr, _, c = g_loss_and_grads(m, zb, xb)
    bn_update_stats(m["g"], c)
    st.opt.apply(st.params(), _flat_bgm_grads(gr), st.lr_theta)
    lz, _, _, dz, c2 = g_loss_and_grads(m, zb, xb)          # training=True again, updated networks
    bn_update_stats(m["g"], c2)
    dz = dz + zb / t(len(idx))
    loss_post = lz + ((zb ** 2).sum(axis=1) / 2).mean()
    st.zt += 1
    lr_t = t(adam_lr_t(st.lr_z, st.zt))
    m_, v_ = t(1 - B1) * dz, t(1 - B2) * dz * dz             # fresh slots every minibatch (:402)
    zb = zb - lr_t * m_ / (np.sqrt(v_) + t(ADAM_EPS))
    st.data_z[idx] = zb
    return loss_x, mse_x, loss_post
