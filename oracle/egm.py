"""Oracle restatement of the CausalBGM EGM warm start (TEST INFRASTRUCTURE).

Follows /root/reference/src/bayesgm/models/causalbgm/base.py
    train_disc_step :305-330   train_gen_step :332-377   egm_init :380-431
and the Discriminator of models/networks/base.py:338-385: Dense -> BatchNormalization -> tanh per
hidden layer, Dense(1) output.  The discriminator is always called with its default training=True,
so every call normalises with the statistics of ITS OWN batch (biased variance, epsilon 1e-3);
gradients therefore carry the cross-sample terms of batch normalisation.

All gradients here are hand-derived (NumPy, dtype of the inputs), because the HIP kernels
implement exactly these formulas.  The WGAN-GP term needs the gradient of
    GP = mean_b (|| d/dzhat sum_b' D(zhat)_b' ||_2 - 1)^2
with respect to the discriminator parameters, i.e. reverse mode through the backward pass of D
("double backward").  Written as an adjoint network:
    da_L = w_out^T (every row)
    for l = L..1:   dy = da_l * (1 - a_l^2);  dhat = dy * gamma_l
                    du = (dhat - mean_b dhat - uhat_l * mean_b(dhat * uhat_l)) / sigma_l
                    da_{l-1} = du W_l^T
    g = da_0
and then reversed op by op (the adjoints reach the forward nodes a_l, uhat_l, sigma_l and flow on
through the ordinary backward pass of D).  tests/test_oracle_autograd.py checks every function of
this file against PyTorch autograd (float64, create_graph double backward).

Parity status: "parity unpinned" (no TensorFlow here, the reference ships no golden vectors for
this path); see oracle/__init__.py.
"""
import numpy as np

from . import nets as N

BN_EPS = 1e-3
B1, B2, ADAM_EPS = 0.9, 0.99, 1e-7     # causalbgm/base.py:86-87 (g_pre_optimizer / d_pre_optimizer)


# ---------------------------------------------------------------------------------------------
# Discriminator
# ---------------------------------------------------------------------------------------------
def init_disc(rng, in_dim, units, dtype=np.float32):
    dims = [in_dim] + list(units) + [1]
    L = len(units)
    return {"W": [N.glorot_uniform(rng, dims[i], dims[i + 1], dtype) for i in range(L + 1)],
            "b": [np.zeros(dims[i + 1], dtype) for i in range(L + 1)],
            "gamma": [np.ones(dims[i + 1], dtype) for i in range(L)],
            "beta": [np.zeros(dims[i + 1], dtype) for i in range(L)]}


def cast_disc(d, dtype):
    return {k: ([a.astype(dtype) for a in v] if isinstance(v, list) else v) for k, v in d.items()}


def _fixed(d):
    """d["fixed_norm"] = True: the discriminator's BatchNormalization layers run in inference mode on their initial
    moving averages (mean 0, variance 1) -- a constant per-column scale 1/sqrt(1 + eps) -- instead of on batch statistics
    (the build's `disc_norm` option, DESIGN_HISTORY.md section 2b)."""
    return bool(d.get("fixed_norm", False))


def disc_forward(d, x):
    """-> (out [B,1], cache).  cache[l] = (a_in, uhat, sigma, a_out) for hidden layer l."""
    L = len(d["gamma"])
    a = x
    cache = []
    for l in range(L):
        u = a @ d["W"][l] + d["b"][l]
        if _fixed(d):
            mu, var = np.zeros_like(u[0]), np.ones_like(u[0])
        else:
            mu = u.mean(axis=0)
            var = ((u - mu) ** 2).mean(axis=0)
        sigma = np.sqrt(var + BN_EPS)
        uhat = (u - mu) / sigma
        a_out = np.tanh(uhat * d["gamma"][l] + d["beta"][l])
        cache.append((a, uhat, sigma, a_out))
        a = a_out
    out = a @ d["W"][L] + d["b"][L]
    return out, cache


def _bn_proj(x, uhat, fixed=False):
    """x - mean_b x - uhat * mean_b(x * uhat)   (the symmetric batch-norm backward projection); identity for fixed statistics."""
    if fixed:
        return x
    return x - x.mean(axis=0) - uhat * (x * uhat).mean(axis=0)


def zero_disc_grads(d):
    return {k: [np.zeros_like(a) for a in v] for k, v in d.items() if isinstance(v, list)}


def disc_backward(d, cache, dout, grads=None, a_bar=None, uhat_bar=None, sigma_bar=None, scale=1.0):
    """Ordinary backward pass.  dout [B,1] = dLoss/dout.  Optional extra adjoints on the forward nodes
    (lists per hidden layer) are added where they enter.  Accumulates `scale`*gradients into `grads`
    and returns (grads, dLoss/dinput)."""
    L = len(d["gamma"])
    if grads is None:
        grads = zero_disc_grads(d)
    a_last = cache[-1][3]
    if dout is not None:
        grads["W"][L] += scale * (a_last.T @ dout)
        grads["b"][L] += scale * dout.sum(axis=0)
        da = dout @ d["W"][L].T
    else:
        da = np.zeros_like(a_last)
    for l in reversed(range(L)):
        a_in, uhat, sigma, a_out = cache[l]
        if a_bar is not None:
            da = da + a_bar[l]
        dy = da * (1.0 - a_out ** 2)
        grads["gamma"][l] += scale * (dy * uhat).sum(axis=0)
        grads["beta"][l] += scale * dy.sum(axis=0)
        duhat = dy * d["gamma"][l]
        if uhat_bar is not None:
            duhat = duhat + uhat_bar[l]
        du = _bn_proj(duhat, uhat, _fixed(d)) / sigma
        if sigma_bar is not None and not _fixed(d):       # explicit use of sigma_l in the adjoint network: d sigma / d u = uhat / B
            du = du + sigma_bar[l] * uhat / uhat.shape[0]
        grads["W"][l] += scale * (a_in.T @ du)
        grads["b"][l] += scale * du.sum(axis=0)
        da = du @ d["W"][l].T
    return grads, da


def disc_input_gradient(d, cache):
    """g[b] = d(sum_b' D(x)_b')/dx_b  (the quantity penalised by WGAN-GP) and the adjoint-network cache."""
    L = len(d["gamma"])
    B = cache[0][0].shape[0]
    da = np.repeat(d["W"][L].T, B, axis=0)             # [B, n_L]
    adj = []
    for l in reversed(range(L)):
        a_in, uhat, sigma, a_out = cache[l]
        dy = da * (1.0 - a_out ** 2)
        dhat = dy * d["gamma"][l]
        du = _bn_proj(dhat, uhat, _fixed(d)) / sigma
        adj.append((da, dy, dhat, du))
        da = du @ d["W"][l].T
    return da, adj[::-1]


def gradient_penalty_and_grads(d, xhat, grads=None, scale=1.0):
    """GP = mean_b (||g_b|| - 1)^2 and scale * dGP/d(discriminator parameters), accumulated into grads."""
    L = len(d["gamma"])
    B = xhat.shape[0]
    if grads is None:
        grads = zero_disc_grads(d)
    _, cache = disc_forward(d, xhat)
    g, adj = disc_input_gradient(d, cache)
    norm = np.sqrt((g ** 2).sum(axis=1))
    gp = ((norm - 1.0) ** 2).mean()
    # reverse through the adjoint network
    da_bar = (2.0 * (norm - 1.0) / norm / B)[:, None] * g          # dGP/dg
    a_bar, uhat_bar, sigma_bar = [None] * L, [None] * L, [None] * L
    for l in range(L):
        a_in, uhat, sigma, a_out = cache[l]
        da, dy, dhat, du = adj[l]
        # da_{l-1} = du W_l^T
        grads["W"][l] += scale * (da_bar.T @ du)
        du_bar = da_bar @ d["W"][l]
        # du = (dhat - m1 - uhat m2) / sigma
        sigma_bar[l] = -(du_bar * du).sum(axis=0) / sigma
        t = du_bar / sigma
        m2 = (dhat * uhat).mean(axis=0)
        uhat_bar[l] = -(t * m2 + dhat * (t * uhat).mean(axis=0))
        if _fixed(d):       # constant statistics: nothing flows through sigma or the projection
            sigma_bar[l], uhat_bar[l] = np.zeros_like(sigma), np.zeros_like(uhat)
        dhat_bar = _bn_proj(t, uhat, _fixed(d))
        # dhat = dy * gamma
        grads["gamma"][l] += scale * (dhat_bar * dy).sum(axis=0)
        dy_bar = dhat_bar * d["gamma"][l]
        # dy = da * (1 - a^2)
        a_bar[l] = dy_bar * da * (-2.0 * a_out)
        da_bar = dy_bar * (1.0 - a_out ** 2)
    grads["W"][L] += scale * da_bar.sum(axis=0)[:, None]
    # ... and on through the forward pass
    disc_backward(d, cache, None, grads, a_bar=a_bar, uhat_bar=uhat_bar, sigma_bar=sigma_bar, scale=scale)
    return gp, grads


# ---------------------------------------------------------------------------------------------
# The two step functions
# ---------------------------------------------------------------------------------------------
def disc_step_grads(nets, dz, z, v, eps):
    """train_disc_step (:305-330): -> (dz_loss, d_loss, grads of d_loss w.r.t. the dz_net parameters)."""
    B = z.shape[0]
    z_ = N.mlp_forward(nets["e"], v)
    zhat = z * eps + z_ * (1.0 - eps)
    grads = zero_disc_grads(dz)
    out_f, cache_f = disc_forward(dz, z_)
    out_r, cache_r = disc_forward(dz, z)
    dz_loss = -out_r.mean() + out_f.mean()
    disc_backward(dz, cache_f, np.full_like(out_f, 1.0 / B), grads)
    disc_backward(dz, cache_r, np.full_like(out_r, -1.0 / B), grads)
    gp, _ = gradient_penalty_and_grads(dz, zhat, grads, scale=10.0)
    return dz_loss, dz_loss + 10.0 * gp, grads


def gen_step_grads(nets, dz, p, z, v, x, y):
    """train_gen_step (:332-377): -> (losses [e_adv, l2_v, l2_z, l2_x, l2_y, total], grads per net)."""
    B = z.shape[0]
    pdim = p["v_dim"]
    z0d, z1d, z2d, _ = p["z_dims"]
    q = z.shape[1]
    g, e, f, h = nets["g"], nets["e"], nets["f"], nets["h"]
    gz, c_g1 = N.mlp_forward_cache(g, z)
    v_ = gz[:, :pdim]
    z_, c_e1 = N.mlp_forward_cache(e, v)
    z__, c_e2 = N.mlp_forward_cache(e, v_)
    gv, c_g2 = N.mlp_forward_cache(g, z_)
    v__ = gv[:, :pdim]
    d_, c_d = disc_forward(dz, z_)
    f_in = np.concatenate([z_[:, :z0d], z_[:, z0d:z0d + z1d], x], axis=1)
    h_in = np.concatenate([z_[:, :z0d], z_[:, z0d + z1d:z0d + z1d + z2d]], axis=1)
    f_out, c_f = N.mlp_forward_cache(f, f_in)
    h_out, c_h = N.mlp_forward_cache(h, h_in)
    y_, x_ = f_out[:, :1], h_out[:, :1]
    l2_v = ((v - v__) ** 2).mean()
    l2_z = ((z - z__) ** 2).mean()
    e_adv = -d_.mean()
    if p["binary_treatment"]:
        l2_x = (np.maximum(x_, 0) - x_ * x + np.log1p(np.exp(-np.abs(x_)))).mean()
        dx_ = (N.sigmoid(x_) - x) / B
    else:
        l2_x = ((x_ - x) ** 2).mean()
        dx_ = 2.0 * (x_ - x) / B
    l2_y = ((y_ - y) ** 2).mean()
    sig = (gz[:, -1] ** 2).mean() + (f_out[:, -1] ** 2).mean() + (h_out[:, -1] ** 2).mean()
    zrec = float(p["use_z_rec"])
    total = e_adv + (l2_v + zrec * l2_z) + (l2_x + l2_y) + 0.001 * sig
    # ---- backward
    # z__ branch: e (call 2, input v_) -> g (call 1)
    dz__ = zrec * (-2.0 / (B * q)) * (z - z__)
    ge2, dv_ = N.mlp_backward(e, c_e2, dz__)
    dgz = np.zeros_like(gz)
    dgz[:, :pdim] = dv_
    dgz[:, -1] += 0.001 * 2.0 * gz[:, -1] / B
    gg1, _ = N.mlp_backward(g, c_g1, dgz)
    # v__ branch: g (call 2, input z_)
    dgv = np.zeros_like(gv)
    dgv[:, :pdim] = (-2.0 / (B * pdim)) * (v - v__)
    gg2, dz_ = N.mlp_backward(g, c_g2, dgv)
    # adversarial branch through the (fixed) discriminator
    _, dz_d = disc_backward(dz, c_d, np.full_like(d_, -1.0 / B))
    dz_ = dz_ + dz_d
    # f, h branches
    df_out = np.zeros_like(f_out)
    df_out[:, :1] = 2.0 * (y_ - y) / B
    df_out[:, -1] += 0.001 * 2.0 * f_out[:, -1] / B
    gf, df_in = N.mlp_backward(f, c_f, df_out)
    dh_out = np.zeros_like(h_out)
    dh_out[:, :1] += dx_
    dh_out[:, -1] += 0.001 * 2.0 * h_out[:, -1] / B
    gh, dh_in = N.mlp_backward(h, c_h, dh_out)
    dz_[:, :z0d] += df_in[:, :z0d] + dh_in[:, :z0d]
    dz_[:, z0d:z0d + z1d] += df_in[:, z0d:z0d + z1d]
    dz_[:, z0d + z1d:z0d + z1d + z2d] += dh_in[:, z0d:z0d + z2d]
    ge1, _ = N.mlp_backward(e, c_e1, dz_)
    add = lambda a, b: [(wa + wb, ba + bb) for (wa, ba), (wb, bb) in zip(a, b)]
    grads = {"g": add(gg1, gg2), "e": add(ge1, ge2), "f": gf, "h": gh}
    return np.array([e_adv, l2_v, l2_z, l2_x, l2_y, total]), grads


# ---------------------------------------------------------------------------------------------
# Keras Adam on lists of arrays + the EGM state
# ---------------------------------------------------------------------------------------------
class Adam(object):
    def __init__(self, params, lr, b1=B1, b2=B2):
        self.params, self.lr, self.b1, self.b2 = params, lr, b1, b2
        self.m = [np.zeros_like(a) for a in params]
        self.v = [np.zeros_like(a) for a in params]
        self.t = 0

    def step(self, grads):
        self.t += 1
        lr_t = self.lr * np.sqrt(1.0 - self.b2 ** self.t) / (1.0 - self.b1 ** self.t)
        for a, g, m, v in zip(self.params, grads, self.m, self.v):
            m *= self.b1; m += (1.0 - self.b1) * g
            v *= self.b2; v += (1.0 - self.b2) * g * g
            a -= (lr_t * m / (np.sqrt(v) + ADAM_EPS)).astype(a.dtype)


def gen_param_list(nets):
    return [a for k in ("g", "e", "f", "h") for Wb in nets[k] for a in Wb]


def disc_param_list(dz):
    return dz["W"] + dz["b"] + dz["gamma"] + dz["beta"]


class EgmState(object):
    """nets (g, e, f, h as [(W, b)...]) + latent discriminator + the two Adam optimizers."""

    def __init__(self, nets, dz, params):
        self.nets, self.dz, self.p = nets, dz, params
        self.g_opt = Adam(gen_param_list(nets), params["lr"])
        self.d_opt = Adam(disc_param_list(dz), params["lr"])

    def disc_step(self, z, v, eps):
        dz_loss, d_loss, gr = disc_step_grads(self.nets, self.dz, z, v, eps)
        self.d_opt.step(disc_param_list(gr))
        return dz_loss, d_loss

    def gen_step(self, z, v, x, y):
        losses, gr = gen_step_grads(self.nets, self.dz, self.p, z, v, x, y)
        self.g_opt.step(gen_param_list(gr))
        return losses


# =============================================================================================
# BGM's EGM warm start (bgm/base.py:190-340): generator = BaseVariationalNet called with training=True (input
# BatchNorm on batch statistics, moving averages updated by every call), encoder e, LSGAN discriminators dz (latent)
# and dx (data) with targets 0.9 / 0.1, optional gradient penalty `gamma`, Adam(lr, beta1 0.5, beta2 0.9).
# =============================================================================================
def g_backward(g, c, dmean, ds_raw, want_dz=True):
    """Backward of BaseVariationalNet (training mode) given dLoss/dmean and dLoss/d(s_raw) [B x p].
    c = cache of oracle.bgm.g_train_forward.  -> (grads dict like oracle.bgm.g_loss_and_grads, dz | None)."""
    from .bgm import LEAK as _LEAK
    h = c["acts"][-1]
    grads = {"mean": (h.T @ dmean, dmean.sum(0)), "var": (h.T @ ds_raw, ds_raw.sum(0))}
    dh = dmean @ g["mean"][0].T + ds_raw @ g["var"][0].T
    tg = [None] * len(g["trunk"])
    for i in reversed(range(len(g["trunk"]))):
        dh = dh * np.where(c["pres"][i] > 0, 1.0, _LEAK).astype(dh.dtype)
        tg[i] = (c["acts"][i].T @ dh, dh.sum(0))
        dh = dh @ g["trunk"][i][0].T
    grads["trunk"] = tg
    grads["gamma"] = (dh * c["zhat"]).sum(0)
    grads["beta"] = dh.sum(0)
    dz = None
    if want_dz:
        dzhat = dh * g["bn"]["gamma"]
        dz = c["inv"] * (dzhat - dzhat.mean(0) - c["zhat"] * (dzhat * c["zhat"]).mean(0))
    return grads, dz


def _add_g(a, b):
    out = {"gamma": a["gamma"] + b["gamma"], "beta": a["beta"] + b["beta"],
           "trunk": [(wa + wb, ba + bb) for (wa, ba), (wb, bb) in zip(a["trunk"], b["trunk"])]}
    for k in ("mean", "var"):
        out[k] = (a[k][0] + b[k][0], a[k][1] + b[k][1])
    return out


def bgm_gen_step_grads(g, e, dz, dx, z, x, n1, n2, alpha):
    """train_gen_step (bgm/base.py:246-289).  n1, n2: the reparameterisation noise of the two generator calls.
    -> (losses [g_adv, e_adv, l2_z, l2_x, reg, total], grads {'g': dict, 'e': [(dW, db)..]}, [cache1, cache2])"""
    from .bgm import g_train_forward
    B, q = z.shape
    p = x.shape[1]
    mu1, s21, c1 = g_train_forward(g, z)
    x_ = n1 * np.sqrt(s21) + mu1
    reg = (s21 ** 2).mean()
    z_, ce1 = N.mlp_forward_cache(e, x)
    z__, ce2 = N.mlp_forward_cache(e, x_)
    mu2, s22, c2 = g_train_forward(g, z_)
    x__ = n2 * np.sqrt(s22) + mu2
    dxo, cdx = disc_forward(dx, x_)
    dzo, cdz = disc_forward(dz, z_)
    l2_x = ((x - x__) ** 2).mean()
    l2_z = ((z - z__) ** 2).mean()
    g_adv = ((0.9 - dxo) ** 2).mean()
    e_adv = ((0.9 - dzo) ** 2).mean()
    total = g_adv + e_adv + 10.0 * (l2_x + l2_z) + alpha * reg
    # ---- backward
    dx__ = 10.0 * (-2.0 / (B * p)) * (x - x__)
    gg2, dz_ = g_backward(g, c2, dx__, dx__ * n2 * 0.5 / np.sqrt(s22) * N.sigmoid(c2["s_raw"]))
    dz__ = 10.0 * (-2.0 / (B * q)) * (z - z__)
    ge2, dx_ = N.mlp_backward(e, ce2, dz__)
    _, dx_d = disc_backward(dx, cdx, -2.0 * (0.9 - dxo) / B)
    dx_ = dx_ + dx_d
    ds21 = dx_ * n1 * 0.5 / np.sqrt(s21) + alpha * 2.0 * s21 / (B * p)
    gg1, _ = g_backward(g, c1, dx_, ds21 * N.sigmoid(c1["s_raw"]), want_dz=False)
    _, dz_d = disc_backward(dz, cdz, -2.0 * (0.9 - dzo) / B)
    ge1, _ = N.mlp_backward(e, ce1, dz_ + dz_d)
    grads = {"g": _add_g(gg1, gg2), "e": [(wa + wb, ba + bb) for (wa, ba), (wb, bb) in zip(ge1, ge2)]}
    return np.array([g_adv, e_adv, l2_z, l2_x, reg, total]), grads, [c1, c2]


def bgm_disc_step_grads(g, e, dz, dx, z, x, n1, eps_z, eps_x, gamma):
    """train_disc_step (bgm/base.py:190-244) -> ([dz_loss, dx_loss, d_loss], {'dz': grads, 'dx': grads}, [cache])."""
    from .bgm import g_train_forward
    B = z.shape[0]
    z_ = N.mlp_forward(e, x)
    mu, s2, c = g_train_forward(g, z)
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


def g_param_list(g):
    out = [g["bn"]["gamma"], g["bn"]["beta"]]
    for W, b in g["trunk"]:
        out += [W, b]
    return out + [g["mean"][0], g["mean"][1], g["var"][0], g["var"][1]]


def g_grad_list(gr):
    out = [gr["gamma"], gr["beta"]]
    for dW, db in gr["trunk"]:
        out += [dW, db]
    return out + [gr["mean"][0], gr["mean"][1], gr["var"][0], gr["var"][1]]


class BgmEgmState(object):
    """g (variational, training-mode BN incl. moving statistics), e, dz, dx and the two Adam(lr, 0.5, 0.9) optimizers."""

    def __init__(self, g, e, dz, dx, params):
        self.g, self.e, self.dz, self.dx, self.p = g, e, dz, dx, params
        self.g_opt = Adam(g_param_list(g) + [a for Wb in e for a in Wb], params["lr"], 0.5, 0.9)
        self.d_opt = Adam(disc_param_list(dz) + disc_param_list(dx), params["lr"], 0.5, 0.9)

    def _move(self, caches):
        from .bgm import bn_update_stats
        for c in caches:
            bn_update_stats(self.g, c)

    def disc_step(self, z, x, n1, eps_z, eps_x):
        losses, gr, caches = bgm_disc_step_grads(self.g, self.e, self.dz, self.dx, z, x, n1, eps_z, eps_x, self.p["gamma"])
        self._move(caches)
        self.d_opt.step(disc_param_list(gr["dz"]) + disc_param_list(gr["dx"]))
        return losses

    def gen_step(self, z, x, n1, n2):
        losses, gr, caches = bgm_gen_step_grads(self.g, self.e, self.dz, self.dx, z, x, n1, n2, self.p["alpha"])
        self._move(caches)
        self.g_opt.step(g_grad_list(gr["g"]) + [a for Wb in gr["e"] for a in Wb])
        return losses
