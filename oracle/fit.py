"""Oracle restatement of the CausalBGM iterative-update step functions (TEST INFRASTRUCTURE).

Follows /root/reference/src/bayesgm/models/causalbgm/base.py:
  update_g_net :156-180, update_h_net :183-214, update_f_net :217-243,
  update_latent_variable_sgd :246-302, fit loop :488-505.
Optimizer = tf.keras.optimizers.Adam (TF 2.10 optimizer_v2), beta_1=0.9, beta_2=0.99,
epsilon=1e-7 (:90-93), restated from the Keras documentation ("parity unpinned"):
    lr_t = lr * sqrt(1 - b2^t) / (1 - b1^t),  t = iterations + 1
    m = b1 m + (1-b1) g ;  v = b2 v + (1-b2) g^2 ;  var -= lr_t * m / (sqrt(v) + eps)
  sparse (IndexedSlices) gradients -- the Z variable, :299-301 -- decay m and v of ALL rows,
  scatter-add the slice, and apply the update to ALL rows ("dense-decay"); `lazy=True` is the
  build's optional row-sparse variant (only the batch rows are touched).
Gradients are derived by hand (oracle/nets.mlp_backward) and cross-checked against PyTorch
autograd in tests/test_oracle_autograd.py.
"""
import numpy as np
from .nets import mlp_forward_cache, mlp_backward, softplus, sigmoid
from .causal import split_z, EPS

B1, B2, ADAM_EPS = 0.9, 0.99, 1e-7


def adam_lr_t(lr, t):
    return lr * np.sqrt(1.0 - B2 ** t) / (1.0 - B1 ** t)


class AdamState(object):
    """Slots of one Keras Adam optimizer over a list of arrays."""

    def __init__(self, params):
        self.m = [np.zeros_like(p) for p in params]
        self.v = [np.zeros_like(p) for p in params]
        self.t = 0

    def apply(self, params, grads, lr):
        self.t += 1
        lr_t = params[0].dtype.type(adam_lr_t(lr, self.t))
        for i, (p, g) in enumerate(zip(params, grads)):
            t_ = p.dtype.type
            self.m[i] = t_(B1) * self.m[i] + t_(1 - B1) * g
            self.v[i] = t_(B2) * self.v[i] + t_(1 - B2) * g * g
            p -= lr_t * self.m[i] / (np.sqrt(self.v[i]) + t_(ADAM_EPS))


def _gauss_head_grads(resid_sq_sum, s_raw, dim, fixed_sig2, Bn, t):
    """loss_b = ssq/(2 s2) + dim*log(s2)/2 ; returns (loss_b, s2, dloss/ds_raw) with the 1/B of the batch mean."""
    if fixed_sig2 is not None:
        s2 = t(fixed_sig2) + 0 * s_raw
        ds_raw = np.zeros_like(s_raw)
    else:
        s2 = softplus(s_raw) + t(EPS)
        ds2 = (-resid_sq_sum / (2 * s2 * s2) + t(dim) / (2 * s2)) / t(Bn)
        ds_raw = ds2 * sigmoid(s_raw)
    loss_b = resid_sq_sum / (2 * s2) + t(dim) * np.log(s2) / 2
    return loss_b, s2, ds_raw


def _fixed(m, key):
    return (m[key] ** 2) if key in m else None


def g_loss_and_grads(m, z, v, want_dz=False):
    """loss_v (:164-169) and its gradients.  Returns (loss, loss_mse, grads|None, dz|None)."""
    t = z.dtype.type
    p = m["v_dim"]
    Bn = len(z)
    out, cache = mlp_forward_cache(m["g"], z)
    mu = out[:, :p]
    d = v - mu
    ssq = (d ** 2).sum(axis=1)
    loss_b, s2, ds_raw = _gauss_head_grads(ssq, out[:, -1], p, _fixed(m, "sigma_v"), Bn, t)
    dout = np.zeros_like(out)
    dout[:, :p] = -d / s2[:, None] / t(Bn)
    dout[:, -1] = ds_raw
    grads, dz = mlp_backward(m["g"], cache, dout)
    return loss_b.mean(), (d ** 2).mean(), grads, dz


def h_loss_and_grads(m, z, x):
    """loss_x (:186-203).  Returns (loss_x, loss_mse_or_bce, grads, dz_full[n,q])."""
    t = z.dtype.type
    Bn = len(z)
    z0, z1, z2 = split_z(m, z)
    inp = np.concatenate([z0, z2], axis=-1)
    out, cache = mlp_forward_cache(m["h"], inp)
    mu = out[:, :1]
    dout = np.zeros_like(out)
    if m["binary_treatment"]:
        l = mu[:, 0]
        bce = np.maximum(l, 0) - l * x[:, 0] + np.log1p(np.exp(-np.abs(l)))
        loss = bce.mean()
        aux = loss
        dout[:, 0] = (sigmoid(l) - x[:, 0]) / t(Bn)
    else:
        d = x - mu
        ssq = (d ** 2).sum(axis=1)
        loss_b, s2, ds_raw = _gauss_head_grads(ssq, out[:, -1], 1, _fixed(m, "sigma_x"), Bn, t)
        loss = loss_b.mean()
        aux = (d ** 2).mean()
        dout[:, 0] = -d[:, 0] / s2 / t(Bn)
        dout[:, -1] = ds_raw
    grads, dinp = mlp_backward(m["h"], cache, dout)
    z0d, z1d, z2d, _ = m["z_dims"]
    dz = np.zeros_like(z)
    dz[:, :z0d] += dinp[:, :z0d]
    dz[:, z0d + z1d:z0d + z1d + z2d] += dinp[:, z0d:]
    return loss, aux, grads, dz


def f_loss_and_grads(m, z, x, y):
    """loss_y (:220-232)."""
    t = z.dtype.type
    Bn = len(z)
    z0, z1, _ = split_z(m, z)
    inp = np.concatenate([z0, z1, x], axis=-1)
    out, cache = mlp_forward_cache(m["f"], inp)
    d = y - out[:, :1]
    ssq = (d ** 2).sum(axis=1)
    loss_b, s2, ds_raw = _gauss_head_grads(ssq, out[:, -1], 1, _fixed(m, "sigma_y"), Bn, t)
    dout = np.zeros_like(out)
    dout[:, 0] = -d[:, 0] / s2 / t(Bn)
    dout[:, -1] = ds_raw
    grads, dinp = mlp_backward(m["f"], cache, dout)
    z0d, z1d, _, _ = m["z_dims"]
    dz = np.zeros_like(z)
    dz[:, :z0d + z1d] += dinp[:, :z0d + z1d]
    return loss_b.mean(), (d ** 2).mean(), grads, dz


def z_loss_and_grad(m, z, x, y, v):
    """loss_postrior_z of update_latent_variable_sgd (:250-295) and d/dz (batch-mean losses)."""
    t = z.dtype.type
    lv, _, _, dzg = g_loss_and_grads(m, z, v)
    lx, _, _, dzh = h_loss_and_grads(m, z, x)
    ly, _, _, dzf = f_loss_and_grads(m, z, x, y)
    prior = ((z ** 2).sum(axis=1) / 2).mean()
    dz = dzg + dzh + dzf + z / t(len(z))
    return lv + lx + ly + prior, dz


def flat_grads(grads):
    return [a for Wb in grads for a in Wb]


def flat_params(net):
    return [a for Wb in net for a in Wb]


class FitState(object):
    """Optimizer state of CausalBGM.fit (:90-93, :484)."""

    def __init__(self, m, data_z, lr_theta, lr_z):
        self.m = m
        self.data_z = data_z
        self.lr_theta, self.lr_z = lr_theta, lr_z
        self.opt = {k: AdamState(flat_params(m[k])) for k in ("g", "h", "f")}
        self.zm = np.zeros_like(data_z)
        self.zv = np.zeros_like(data_z)
        self.zt = 0


def fit_step(st, x, y, v, idx, lazy_z=False):
    """One minibatch of the loop body :494-505.  Returns the 7 losses of the tqdm postfix."""
    m = st.m
    bz = st.data_z[idx]
    bx, by, bv = x[idx], y[idx], v[idx]
    loss_v, mse_v, gg, _ = g_loss_and_grads(m, bz, bv)
    st.opt["g"].apply(flat_params(m["g"]), flat_grads(gg), st.lr_theta)
    loss_x, mse_x, gh, _ = h_loss_and_grads(m, bz, bx)
    st.opt["h"].apply(flat_params(m["h"]), flat_grads(gh), st.lr_theta)
    loss_y, mse_y, gf, _ = f_loss_and_grads(m, bz, bx, by)
    st.opt["f"].apply(flat_params(m["f"]), flat_grads(gf), st.lr_theta)
    # update_latent_variable_sgd with the UPDATED networks
    loss_z, dz = z_loss_and_grad(m, st.data_z[idx], bx, by, bv)
    adam_rows(st, idx, dz, lazy_z)
    return loss_x, mse_x, loss_y, mse_y, loss_v, mse_v, loss_z


def adam_rows(st, idx, dz, lazy):
    """Keras Adam `_resource_apply_sparse` on the [N x q] latent variable (dense-decay), or the
    build's lazy variant."""
    t = st.data_z.dtype.type
    st.zt += 1
    lr_t = t(adam_lr_t(st.lr_z, st.zt))
    if lazy:
        m_ = t(B1) * st.zm[idx] + t(1 - B1) * dz
        v_ = t(B2) * st.zv[idx] + t(1 - B2) * dz * dz
        st.zm[idx], st.zv[idx] = m_, v_
        st.data_z[idx] -= lr_t * m_ / (np.sqrt(v_) + t(ADAM_EPS))
        return
    st.zm *= t(B1)
    st.zv *= t(B2)
    st.zm[idx] += t(1 - B1) * dz
    st.zv[idx] += t(1 - B2) * dz * dz
    st.data_z -= lr_t * st.zm / (np.sqrt(st.zv) + t(ADAM_EPS))


def fit_epochs(st, data, epochs, batch_size, rng, lazy_z=False):
    """for epoch in range(epochs+1): perm = np.random.choice(N, N, replace=False); minibatches incl. the
    short last one (:488-505).  `rng` is a np.random.RandomState standing in for the global RNG."""
    x, y, v = data
    n = len(x)
    hist = []
    for _ in range(epochs + 1):
        perm = rng.choice(n, n, replace=False)
        for i in range(0, n, batch_size):
            hist.append(fit_step(st, x, y, v, perm[i:i + batch_size], lazy_z))
    return np.array(hist)
