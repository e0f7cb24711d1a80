"""Oracle restatement of the CausalBGM hot path (TEST INFRASTRUCTURE).

Follows /root/reference/src/bayesgm/models/causalbgm/base.py (cited per
function).  Deterministic nets only (``use_bnn=False``); the Bayesian-net
variant is SURVEY.md section 8(f) row N2.  Parity status: unpinned against TF
(see oracle/__init__.py); RNG streams are the build's own (oracle/rng.py).

model = {"g": net, "f": net, "h": net, "e": net (optional),
         "z_dims": [z0,z1,z2,z3], "v_dim": p, "binary_treatment": bool,
         optional "sigma_v"/"sigma_x"/"sigma_y": fixed std-devs}
"""
import numpy as np
from . import rng as R
from .nets import (mlp_forward, mlp_forward_cache, mlp_backward, softplus, sigmoid,
                   init_mlp, cast_net)

EPS = 1e-6


def init_model(seed, z_dims, v_dim, binary_treatment=False,
               g_units=(64,) * 5, e_units=(64,) * 5, f_units=(64, 32, 8), h_units=(64, 32, 8),
               dtype=np.float32, **fixed_sigmas):
    """Network shapes of base.py:64-81."""
    rs = np.random.RandomState(seed)
    q = int(sum(z_dims))
    m = {"z_dims": list(z_dims), "v_dim": int(v_dim), "binary_treatment": bool(binary_treatment)}
    m["g"] = init_mlp(rs, [q] + list(g_units) + [v_dim + 1], dtype)
    m["e"] = init_mlp(rs, [v_dim] + list(e_units) + [q], dtype)
    m["f"] = init_mlp(rs, [z_dims[0] + z_dims[1] + 1] + list(f_units) + [2], dtype)
    m["h"] = init_mlp(rs, [z_dims[0] + z_dims[2]] + list(h_units) + [2], dtype)
    m.update(fixed_sigmas)
    return m


def cast_model(m, dtype):
    out = dict(m)
    for k in ("g", "e", "f", "h"):
        if k in m:
            out[k] = cast_net(m[k], dtype)
    return out


def split_z(m, z):
    z0d, z1d, z2d, _ = m["z_dims"]
    z0 = z[..., :z0d]
    z1 = z[..., z0d:z0d + z1d]
    z2 = z[..., z0d + z1d:z0d + z1d + z2d]
    return z0, z1, z2


def _sig2(m, key, raw, t):
    """sigma^2 = params['sigma_*']**2 if fixed else softplus(raw)+eps (base.py:781-798)."""
    if key in m:
        return t(m[key]) ** 2 + 0 * raw
    return softplus(raw) + t(EPS)


def log_posterior(m, x, y, v, z, prior=None):
    """get_log_posterior, base.py:765-817.  x,y [n,1]; v [n,p]; z [n,q] -> [n].  prior = (mu [n,q], sigma^2 [n]): the conditional
    latent prior of IdentifiableCausalBGM (identifiable.py:521-555) instead of N(0, I)."""
    t = z.dtype.type
    p = m["v_dim"]
    z0, z1, z2 = split_z(m, z)
    g_out = mlp_forward(m["g"], z)
    mu_v = g_out[:, :p]
    s2v = _sig2(m, "sigma_v", g_out[:, -1], t)
    h_out = mlp_forward(m["h"], np.concatenate([z0, z2], axis=-1))
    mu_x = h_out[:, :1]
    s2x = _sig2(m, "sigma_x", h_out[:, -1], t)
    f_out = mlp_forward(m["f"], np.concatenate([z0, z1, x], axis=-1))
    mu_y = f_out[:, :1]
    s2y = _sig2(m, "sigma_y", f_out[:, -1], t)
    loss_v = ((v - mu_v) ** 2).sum(axis=1) / (2 * s2v) + t(p) * np.log(s2v) / 2
    if m["binary_treatment"]:
        l = mu_x[:, 0]
        # tf.nn.sigmoid_cross_entropy_with_logits: max(l,0) - l*x + log1p(exp(-|l|))
        loss_x = np.maximum(l, 0) - l * x[:, 0] + np.log1p(np.exp(-np.abs(l)))
    else:
        loss_x = ((x - mu_x) ** 2).sum(axis=1) / (2 * s2x) + np.log(s2x) / 2
    loss_y = ((y - mu_y) ** 2).sum(axis=1) / (2 * s2y) + np.log(s2y) / 2
    if prior is None:
        loss_prior = (z ** 2).sum(axis=1) / 2
    else:     # Z | U ~ N(mu(U), sigma^2(U) I): identifiable.py:541-551 (prior = (mu [n, q], sigma^2 [n]))
        mu_p, s2_p = prior
        loss_prior = ((z - mu_p) ** 2).sum(axis=1) / (2 * s2_p) + t(z.shape[1]) * np.log(s2_p) / 2
    return -(loss_v + loss_x + loss_y + loss_prior)


def mh_init_state(n, q, seed, row0=0):
    """current_state ~ N(0,1)  (base.py:842), build RNG spec tag 0."""
    return R.normals(np.arange(row0, row0 + n), 0, q, R.TAG_INIT, seed)


def mh_transition(m, x, y, v, state, logp, it, q_sd, seed, row0=0, eps=None, u=None, prior=None):
    """One iteration of the while-loop body, base.py:860-871.

    The reference evaluates get_log_posterior on the current state every
    iteration (:866); for deterministic nets that value equals the cached one,
    so the oracle (like the HIP kernel) carries ``logp`` along.
    Returns (state, logp, accepted[bool n])."""
    n, q = state.shape
    rows = np.arange(row0, row0 + n)
    if eps is None:
        eps = R.normals(rows, it, q, R.TAG_PROP, seed)
    if u is None:
        u = R.uniforms(rows, it, R.TAG_ACC, seed)
    t = state.dtype.type
    prop = state + t(q_sd) * eps.astype(state.dtype)
    lp_prop = log_posterior(m, x, y, v, prop, prior)
    ratio = np.exp(np.minimum(lp_prop - logp, 0))
    acc = u.astype(state.dtype) < ratio
    state = np.where(acc[:, None], prop, state)
    logp = np.where(acc, lp_prop, logp)
    return state, logp, acc


def mh_sampler(m, data, burn_in, n_keep, q_sd, seed, row0=0, adaptive=False,
               initial_q_sd=1.0, target=0.25, tol=0.05, adj_int=50, window=100,
               return_acc=False, prior=None):
    """metropolis_hastings_sampler, base.py:820-904 -> samples [n_keep, n, q]."""
    x, y, v = data
    n = len(x)
    q = int(sum(m["z_dims"]))
    dt = v.dtype
    state = mh_init_state(n, q, seed, row0).astype(dt)
    logp = log_posterior(m, x, y, v, state, prior)
    if adaptive:
        q_sd = initial_q_sd
    samples, recent, acc_hist = [], [], []
    counter = 0
    while len(samples) < n_keep:
        state, logp, acc = mh_transition(m, x, y, v, state, logp, counter, q_sd, seed, row0, prior=prior)
        recent.append(acc)
        acc_hist.append(acc.sum())
        if len(recent) > window:
            recent = recent[-window:]
        if adaptive and counter < burn_in and counter % adj_int == 0 and counter > 0:
            rate = np.sum(recent) / (len(recent) * n)
            if rate < target - tol:
                q_sd *= 0.9
            elif rate > target + tol:
                q_sd *= 1.1
        if counter >= burn_in:
            samples.append(state.copy())
        counter += 1
    samples = np.array(samples)
    if return_acc:
        return samples, np.array(acc_hist), q_sd
    return samples


def infer_from_latent_posterior(m, post_z, x_values=None, sample_y=True, seed=0, row0=0,
                                burn_in=0):
    """base.py:671-763.  post_z [n_keep, n, q].
    binary  -> ITE draws [n_keep, n];  continuous -> ADRF draws [len(x_values), n_keep].
    Outcome noise: build RNG spec tag 3 at iteration burn_in + d (draw d), sequential
    layout, feature index = dose index (binary: 0 -> x=1, 1 -> x=0)."""
    n_keep, n, _ = post_z.shape
    t = post_z.dtype.type
    rows = np.arange(row0, row0 + n)

    def f_at(z, xval):
        z0, z1, _ = split_z(m, z)
        xin = np.full((n, 1), xval, dtype=z.dtype)
        out = mlp_forward(m["f"], np.concatenate([z0, z1, xin], axis=-1))
        mu = out[:, 0]
        s2 = _sig2(m, "sigma_y", out[:, 1], t)
        return mu, s2

    if m["binary_treatment"]:
        ite = np.empty((n_keep, n), dtype=post_z.dtype)
        for d in range(n_keep):
            nz = R.normals_seq(rows, burn_in + d, 2, R.TAG_YNOISE, seed).astype(post_z.dtype)
            mu1, s1 = f_at(post_z[d], 1.0)
            mu0, s0 = f_at(post_z[d], 0.0)
            if sample_y:
                ite[d] = (mu1 + np.sqrt(s1) * nz[:, 0]) - (mu0 + np.sqrt(s0) * nz[:, 1])
            else:
                ite[d] = mu1 - mu0
        return ite
    xs = np.atleast_1d(np.asarray(x_values, dtype=np.float64))
    out = np.empty((len(xs), n_keep), dtype=post_z.dtype)
    for d in range(n_keep):
        nz = R.normals_seq(rows, burn_in + d, len(xs), R.TAG_YNOISE, seed).astype(post_z.dtype)
        for k, xv in enumerate(xs):
            mu, s2 = f_at(post_z[d], t(xv))
            yk = mu + np.sqrt(s2) * nz[:, k] if sample_y else mu
            out[k, d] = yk.mean()
    return out


def predict(m, data, alpha=0.01, n_mcmc=3000, burn_in=5000, x_values=None, q_sd=1.0,
            sample_y=True, bs=10000, seed=0):
    """predict, base.py:573-668.  Row indices of the RNG spec are global, so the
    result does not depend on ``bs`` (the reference's bs only bounds memory)."""
    x, y, v = data
    n = len(x)
    adaptive = (q_sd is None) or (q_sd <= 0)
    if m["binary_treatment"]:
        mean = np.zeros(n, np.float32)
        lo = np.zeros(n, np.float32)
        hi = np.zeros(n, np.float32)
        for s in range(0, n, bs):
            e = min(s + bs, n)
            pz = mh_sampler(m, (x[s:e], y[s:e], v[s:e]), burn_in, n_mcmc, q_sd, seed, row0=s,
                            adaptive=adaptive)
            eff = infer_from_latent_posterior(m, pz, x_values, sample_y, seed, row0=s, burn_in=burn_in)
            mean[s:e] = eff.mean(axis=0)
            hi[s:e] = np.quantile(eff, 1 - alpha / 2, axis=0)
            lo[s:e] = np.quantile(eff, alpha / 2, axis=0)
        return mean, np.stack([lo, hi], axis=1)
    if x_values is None:
        raise ValueError("For continuous treatment, 'x_values' must not be None.")
    xs = np.atleast_1d(np.asarray(x_values, dtype=float))
    sums = np.zeros((len(xs), n_mcmc), np.float64)
    seen = 0
    for s in range(0, n, bs):
        e = min(s + bs, n)
        pz = mh_sampler(m, (x[s:e], y[s:e], v[s:e]), burn_in, n_mcmc, q_sd, seed, row0=s,
                        adaptive=adaptive)
        eff = infer_from_latent_posterior(m, pz, xs, sample_y, seed, row0=s, burn_in=burn_in)
        sums += eff.astype(np.float64) * (e - s)
        seen += e - s
    ce = (sums / float(seen)).astype(np.float32)
    adrf = ce.mean(axis=1)
    hi = np.quantile(ce, 1 - alpha / 2, axis=1)
    lo = np.quantile(ce, alpha / 2, axis=1)
    return adrf, np.stack([lo, hi], axis=1)


def percentile_nearest(a, qpct):
    """tfp.stats.percentile(x, q) default interpolation='nearest' (base.py:558-559):
    sorted[round((n-1) * q/100)] with round-half-even (tf.round)."""
    s = np.sort(np.asarray(a).ravel())
    idx = int(np.round((len(s) - 1) * qpct / 100.0))
    return s[idx]


def evaluate(m, data, data_z=None, nb_intervals=200):
    """evaluate, base.py:534-570 -> (causal_pre, mse_x, mse_y, mse_v)."""
    x, y, v = data
    t = v.dtype.type
    if data_z is None:
        data_z = mlp_forward(m["e"], v)
    z0, z1, z2 = split_z(m, data_z)
    v_pred = mlp_forward(m["g"], data_z)[:, :m["v_dim"]]
    y_pred = mlp_forward(m["f"], np.concatenate([z0, z1, x], axis=-1))[:, :1]
    x_pred = mlp_forward(m["h"], np.concatenate([z0, z2], axis=-1))[:, :1]
    if m["binary_treatment"]:
        x_pred = sigmoid(x_pred)
    mse_v = ((v - v_pred) ** 2).mean()
    mse_x = ((x - x_pred) ** 2).mean()
    mse_y = ((y - y_pred) ** 2).mean()
    n = len(x)
    if m["binary_treatment"]:
        pos = mlp_forward(m["f"], np.concatenate([z0, z1, np.ones((n, 1), v.dtype)], -1))[:, :1]
        neg = mlp_forward(m["f"], np.concatenate([z0, z1, np.zeros((n, 1), v.dtype)], -1))[:, :1]
        return pos - neg, mse_x, mse_y, mse_v
    x_min = percentile_nearest(x, 5.0)
    x_max = percentile_nearest(x, 95.0)
    xs = np.linspace(x_min, x_max, nb_intervals).astype(v.dtype)
    dose = np.array([mlp_forward(m["f"], np.concatenate(
        [z0, z1, np.full((n, 1), xv, v.dtype)], -1))[:, 0].mean() for xv in xs], dtype=v.dtype)
    return dose, mse_x, mse_y, mse_v


def mh_reference_loop(m, data, n_iter, q_sd=1.0, rng=None):
    """The reference's MH loop AS WRITTEN (base.py:842-871), for CPU-baseline timing only:
    host NumPy RNG (Mersenne Twister), fresh proposal each iteration, and TWO
    get_log_posterior evaluations per iteration (:865 proposed, :866 current).
    Returns (state, n_accepted)."""
    x, y, v = data
    rng = np.random if rng is None else rng
    n = len(x)
    q = int(sum(m["z_dims"]))
    state = rng.normal(0, 1, size=(n, q)).astype('float32')
    n_acc = 0
    for _ in range(n_iter):
        prop = state + rng.normal(0, q_sd, size=(n, q)).astype('float32')
        lp_prop = log_posterior(m, x, y, v, prop)
        lp_cur = log_posterior(m, x, y, v, state)
        ratio = np.exp(np.minimum(lp_prop - lp_cur, 0))
        idx = rng.rand(n) < ratio
        state[idx] = prop[idx]
        n_acc += int(idx.sum())
    return state, n_acc
