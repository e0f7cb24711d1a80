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
