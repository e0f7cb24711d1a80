"""Oracle restatement of IdentifiableCausalBGM's additions to CausalBGM (TEST INFRASTRUCTURE; parity unpinned, see oracle/__init__.py).

Follows /root/reference/src/bayesgm/models/causalbgm/identifiable.py:
  prior network p(z | u): BaseFullyConnectedNet(n_segments -> prior_units -> q + 1)            :76-78
  update_latent_variable_sgd (conditional prior, joint step on the batch latents and the prior net) :150-226
  fit loop (random segments U, incomplete last batch skipped, fresh batch_z Variable per step)       :279-325
  get_log_posterior / metropolis_hastings_sampler with data_u                                    :521-614 (oracle/causal.py, `prior=`)
The batch latents are a NEW tf.Variable in every minibatch (:304), so the persistent posterior_optimizer creates fresh Adam slots
for it each time while its `iterations` counter runs on: the update is lr_t * m / (sqrt(v) + eps) with m = (1 - b1) g,
v = (1 - b2) g^2 and t the global step count (the same situation as BGM.fit, oracle/bgm.py fit_step)."""
import numpy as np

from .nets import init_mlp, mlp_forward_cache, mlp_backward, softplus, sigmoid
from .fit import AdamState, adam_lr_t, B1, B2, ADAM_EPS, flat_params, flat_grads, z_loss_and_grad
from .causal import EPS


def init_prior_net(rng, n_segments, q, units=(64,), dtype=np.float32):
    return init_mlp(rng, [n_segments] + list(units) + [q + 1], dtype)


def prior_params(pnet, seg, dtype=None):
    """(mu [n, q], sigma^2 [n]) of the rows with segments `seg` (:197-200)."""
    k = pnet[0][0].shape[0]
    u = np.eye(k, dtype=pnet[0][0].dtype)[seg]
    out, cache = mlp_forward_cache(pnet, u)
    t = out.dtype.type
    return out[:, :-1], softplus(out[:, -1]) + t(EPS), (out, cache)


def prior_table(pnet, q):
    """[n_segments x (q + 2)]: mu, 1 / sigma^2, (q / 2) log sigma^2 per segment -- the table the sampling kernels take."""
    k = pnet[0][0].shape[0]
    mu, s2, _ = prior_params(pnet, np.arange(k))
    return np.concatenate([mu, (1.0 / s2)[:, None], (0.5 * q * np.log(s2))[:, None]], axis=1).astype(np.float32)


class IdentState(object):
    def __init__(self, m, pnet, data_z, lr_theta, lr_z):
        self.m, self.pnet, self.data_z, self.lr_theta, self.lr_z = m, pnet, data_z, lr_theta, lr_z
        self.opt = {k: AdamState(flat_params(m[k])) for k in ("g", "h", "f")}
        self.popt = AdamState(flat_params(pnet))
        self.zt = 0


def z_and_prior_step(st, bx, by, bv, idx, seg_b):
    """update_latent_variable_sgd (:150-226) on the batch rows idx: returns loss_postrior_z."""
    m = st.m
    zb = st.data_z[idx].copy()
    t = zb.dtype.type
    Bn, q = zb.shape
    loss_std, dz = z_loss_and_grad(m, zb, bx, by, bv)              # NLLs + |z|^2 / 2 (batch means) and its z gradient
    mu, s2, (out, cache) = prior_params(st.pnet, seg_b)
    d = zb - mu
    ssq = (d ** 2).sum(axis=1)
    loss_prior = (ssq / (2 * s2) + t(q) * np.log(s2) / 2).mean()
    loss = loss_std - ((zb ** 2).sum(axis=1) / 2).mean() + loss_prior
    dz = dz - zb / t(Bn) + d / s2[:, None] / t(Bn)
    # prior-net gradients: d loss_prior / d out
    dout = np.zeros_like(out)
    dout[:, :-1] = -d / s2[:, None] / t(Bn)
    ds2 = (-ssq / (2 * s2 * s2) + t(q) / (2 * s2)) / t(Bn)
    dout[:, -1] = ds2 * sigmoid(out[:, -1])
    pgrads, _ = mlp_backward(st.pnet, cache, dout)
    # latent update: fresh slots, global step count (:216-217 on the Variable created at :304)
    st.zt += 1
    lr_t = t(adam_lr_t(st.lr_z, st.zt))
    m_, v_ = t(1 - B1) * dz, t(1 - B2) * dz * dz
    st.data_z[idx] = zb - lr_t * m_ / (np.sqrt(v_) + t(ADAM_EPS))
    st.popt.apply(flat_params(st.pnet), flat_grads(pgrads), st.lr_theta)                                  # :220-222
    return loss


def fit_step(st, x, y, v, idx, seg):
    """One minibatch of the loop body :300-317: update_g/h/f_net on the batch latents, then the joint latent / prior step."""
    from .fit import g_loss_and_grads, h_loss_and_grads, f_loss_and_grads
    m = st.m
    bz, bx, by, bv = st.data_z[idx], x[idx], y[idx], v[idx]
    loss_v, mse_v, gg, _ = g_loss_and_grads(m, bz, bv)
    st.opt["g"].apply(flat_params(m["g"]), flat_grads(gg), st.lr_theta)
    loss_x, mse_x, gh, _ = h_loss_and_grads(m, bz, bx)
    st.opt["h"].apply(flat_params(m["h"]), flat_grads(gh), st.lr_theta)
    loss_y, mse_y, gf, _ = f_loss_and_grads(m, bz, bx, by)
    st.opt["f"].apply(flat_params(m["f"]), flat_grads(gf), st.lr_theta)
    loss_z = z_and_prior_step(st, bx, by, bv, idx, seg[idx])
    return loss_x, mse_x, loss_y, mse_y, loss_v, mse_v, loss_z


# ---------------------------------------------------------------------------------------------------
# use_bnn=True: the prior network is a BayesianFullyConnectedNet (identifiable.py:66-67); its KL terms enter the latent / prior
# step with kl_weight (:213-215); get_log_posterior calls it afresh at every evaluation (:541-551).  Noise: net id 4 in the
# streams of oracle/bnn.py.
# ---------------------------------------------------------------------------------------------------
PRIOR_NET_ID = 4


def init_prior_bnn(rs, n_segments, q, units=(64,), dtype=np.float32):
    from . import bnn as BN
    return BN.init_bnn(rs, [n_segments] + list(units) + [q + 1], dtype)


def bnn_prior_params(pnet, seg, noise):
    """(mu [n, q], sigma^2 [n], (out, cache)) of ONE noisy call of the Bayesian prior net on the one-hot rows of `seg` (:197-200)."""
    from . import bnn as BN
    k = pnet["gamma"].shape[0]
    u = np.eye(k, dtype=pnet["gamma"].dtype)[seg]
    out, cache = BN.forward(pnet, u, noise)
    t = out.dtype.type
    return out[:, :-1], softplus(out[:, -1]) + t(EPS), (out, cache)


def bnn_prior_noise(pnet, B, key, stream, dtype=np.float32, row0=0):
    from . import bnn as BN
    return BN.draw_noise(BN.net_dims(pnet), B, key, stream, PRIOR_NET_ID, dtype=dtype, row0=row0)


def bnn_prior_step_given_dz(pnet, popt, data_z, idx, seg_b, dz_std, p_noise, lr_z, zt, lr_theta, kl_weight, inv_b=None):
    """The conditional-prior half of update_latent_variable_sgd with use_bnn (:195-226) given dz_std, the z gradient of the batch-mean
    NLLs + |z|^2 / 2: exchanges the prior term, fresh-slot Adam on the batch latents, Adam on the prior net with the gradient of
    batch-mean prior term + kl_weight * KL.  Returns (batch-mean prior term, batch-mean |z|^2 / 2, KL sum)."""
    from . import bnn as BN
    zb = data_z[idx].copy()
    t = zb.dtype.type
    Bn, q = zb.shape
    ib = t(1.0 / Bn if inv_b is None else inv_b)
    mu, s2, (out, cache) = bnn_prior_params(pnet, seg_b, p_noise)
    d = zb - mu
    ssq = (d ** 2).sum(axis=1)
    loss_prior = (ssq / (2 * s2) + t(q) * np.log(s2) / 2).sum() * ib
    dz = dz_std - zb * ib + d / s2[:, None] * ib
    dout = np.zeros_like(out)
    dout[:, :-1] = -d / s2[:, None] * ib
    dout[:, -1] = (-ssq / (2 * s2 * s2) + t(q) / (2 * s2)) * ib * sigmoid(out[:, -1])
    g, _ = BN.backward(pnet, cache, dout, want_dx=False)
    klv, klg = BN.kl(pnet)
    g = BN.add_grads(g, klg, t(kl_weight))
    lr_t = t(adam_lr_t(lr_z, zt))
    m_, v_ = t(1 - B1) * dz, t(1 - B2) * dz * dz
    data_z[idx] = zb - lr_t * m_ / (np.sqrt(v_) + t(ADAM_EPS))
    popt.apply(BN.flat_params(pnet), BN.flat_grads(g), lr_theta)
    return loss_prior, ((zb ** 2).sum(axis=1) / 2).sum() * ib, klv


def bnn_z_and_prior_step(m, pnet, popt, data_z, idx, seg_b, bx, by, bv, z_noises, p_noise, lr_z, zt, lr_theta, kl_weight):
    """update_latent_variable_sgd with use_bnn (:150-226) on the batch rows idx.  m: oracle/bnn.py model; z_noises as oracle.bnn.z_step;
    p_noise: the prior net's call; popt: AdamState over bnn.flat_params(pnet); zt: latent step count AFTER this step.
    Returns loss_postrior_z incl. kl_weight * KL(prior net)."""
    from . import bnn as BN
    zb = data_z[idx]
    loss_std, dz = BN.z_step(m, zb, bx, by, bv, z_noises)          # NLLs + |z|^2 / 2 (batch means), z gradient incl. z / B
    lp, lz, klv = bnn_prior_step_given_dz(pnet, popt, data_z, idx, seg_b, dz, p_noise, lr_z, zt, lr_theta, kl_weight)
    return loss_std - lz + lp + zb.dtype.type(kl_weight) * klv


def bnn_log_posterior_blocks(m, pnet, seg, x, y, v, z, block_rows, seed, stream, block0=0):
    """get_log_posterior with use_bnn and the conditional prior (:497-555): per block of rows one noisy call of g, h, f (oracle/bnn.py
    log_posterior_blocks) and one of the prior net (stream `stream`, net id 4, the block's key)."""
    from . import bnn as BN
    lp = BN.log_posterior_blocks(m, x, y, v, z, block_rows, seed, stream, block0)
    n, q = z.shape
    t = z.dtype.type
    out = np.array(lp, dtype=z.dtype, copy=True)
    for b, lo in enumerate(range(0, n, block_rows)):
        hi = min(n, lo + block_rows)
        noise = bnn_prior_noise(pnet, hi - lo, BN.block_key(seed, block0 + b), stream, dtype=z.dtype)
        mu, s2, _ = bnn_prior_params(pnet, seg[lo:hi], noise)
        zz = z[lo:hi]
        out[lo:hi] += (zz ** 2).sum(axis=1) / 2 - (((zz - mu) ** 2).sum(axis=1) / (2 * s2) + t(q) * np.log(s2) / 2)
    return out


def bnn_mh_iteration(m, pnet, seg, x, y, v, z, it, q_sd, seed, block_rows, block0=0, row_base=0):
    """One iteration of metropolis_hastings_sampler with use_bnn and the conditional prior (:557-614): both states evaluated afresh
    (proposal: stream 2 it, current: 2 it + 1).  Returns (new z, accepted, lpp, lpc)."""
    from . import rng as R
    n, q = z.shape
    rows = np.arange(row_base, row_base + n)
    prop = (z + z.dtype.type(q_sd) * R.normals(rows, it, q, R.TAG_PROP, seed).astype(z.dtype)).astype(z.dtype)
    lpp = bnn_log_posterior_blocks(m, pnet, seg, x, y, v, prop, block_rows, seed, 2 * it, block0)
    lpc = bnn_log_posterior_blocks(m, pnet, seg, x, y, v, z, block_rows, seed, 2 * it + 1, block0)
    u = R.uniforms(rows, it, R.TAG_ACC, seed)
    acc = u < np.exp(np.minimum(lpp - lpc, 0))
    out = z.copy()
    out[acc] = prop[acc]
    return out, acc, lpp, lpc
