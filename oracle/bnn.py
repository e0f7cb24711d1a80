"""Oracle restatement of the Bayesian networks of CausalBGM, ``use_bnn=True`` (TEST INFRASTRUCTURE).

Follows /root/reference/src/bayesgm/models/networks/bnn.py:4-38 (BayesianFullyConnectedNet) and the
``use_bnn`` branches of /root/reference/src/bayesgm/models/causalbgm/base.py (update_g/h/f_net :156-243,
update_latent_variable_sgd :246-302, get_log_posterior :765-817, evaluate :534-570,
infer_from_latent_posterior :671-763).

The arithmetic lives in tensorflow-probability 0.18.0 / keras 2.10, absent from /root/reference and not
installable here -> PARITY UNPINNED; restated from their published semantics:

  BatchNormalization on the network input (bnn.py:14,26).  The layer is called without ``training=``
  inside a Model whose ``call`` defaults to ``training=True`` (bnn.py:24): Keras resolves the inner
  layer's mode from the outer call context, so it ALWAYS normalises with the statistics of the batch it
  is given: xn = gamma * (x - mean_B) / sqrt(var_B + 1e-3) + beta, biased variance.  (The moving averages
  are updated but never read on any path; the build does not keep them.)

  tfp.layers.DenseFlipout(units, activation=None) (bnn.py:19):
      kernel posterior  N(loc, sigma^2), sigma = finfo(f32).eps + softplus(rho);
          loc ~ N(0, 0.1^2), rho ~ N(-3, 0.1^2) at initialisation (default_mean_field_normal_fn)
      bias posterior    deterministic point mass, loc ~ N(0, 0.1^2) at initialisation (is_singular=True)
      kernel prior      N(0, 1) (default_multivariate_normal_fn); no bias prior
      call:  y = x @ loc + ((x * s_in) @ (sigma * eps)) * s_out + bias
             eps ~ N(0, 1) [in x out], ONE draw per call shared by the batch;
             s_in [B x in], s_out [B x out] independent Rademacher signs per example  (Wen et al. 2018)
      losses: KL(N(loc, sigma^2) || N(0, 1)) summed over the kernel, analytic, unscaled.

  LeakyReLU(0.2) between layers, none after the last (bnn.py:27-36).

Every call draws fresh noise.  The reference draws it from TF's global generator; the build fixes its own
counter-based stream (Philox4x32-10, oracle/rng.py) so that oracle and kernels see the same noise:

  noise key  (k0, k1) = (seed_lo, seed_hi + batch_id)         one key per batch ("block" of rows)
  eps of layer l of net n in call `stream`, element idx = i * out + j of the [in x out] kernel:
      Box-Muller output (idx & 3) of Philox(ctr = (idx >> 2, l | n << 16, stream, TAG_EPS))
  sign words of row r (index inside its batch): word w = Philox(ctr = (r, (w >> 2) | n << 16, stream,
      TAG_SIGN))[w & 3]; layer l's input signs start at word sin_w[l], its output signs at sout_w[l]
      (each layer side rounded up to whole 32-bit words); bit c & 31 of word c >> 5 set => sign -1.

net = {"gamma": [in], "beta": [in], "layers": [(loc [in x out], rho [in x out], bias [out]), ...]}
"""
import numpy as np
from . import rng as R
from .nets import lrelu, softplus, sigmoid, LEAK, BN_EPS

TAG_EPS, TAG_SIGN = 8, 9
NET_ID = {"g": 0, "e": 1, "f": 2, "h": 3}
SCALE_EPS = float(np.finfo(np.float32).eps)


def init_bnn(rs, dims, dtype=np.float32):
    layers = []
    for i in range(len(dims) - 1):
        loc = (0.1 * rs.standard_normal((dims[i], dims[i + 1]))).astype(dtype)
        rho = (-3.0 + 0.1 * rs.standard_normal((dims[i], dims[i + 1]))).astype(dtype)
        bias = (0.1 * rs.standard_normal(dims[i + 1])).astype(dtype)
        layers.append((loc, rho, bias))
    return {"gamma": np.ones(dims[0], dtype), "beta": np.zeros(dims[0], dtype), "layers": layers}


def cast_bnn(net, dtype):
    out = {"gamma": net["gamma"].astype(dtype), "beta": net["beta"].astype(dtype),
           "layers": [tuple(a.astype(dtype) for a in L) for L in net["layers"]]}
    if "norm" in net:
        out["norm"] = net["norm"]
    return out


def init_model(seed, z_dims, v_dim, binary_treatment=False, g_units=(64,) * 5, e_units=(64,) * 5,
               f_units=(64, 32, 8), h_units=(64, 32, 8), dtype=np.float32):
    """Network shapes of causalbgm/base.py:64-72."""
    rs = np.random.RandomState(seed)
    q = int(sum(z_dims))
    m = {"z_dims": list(z_dims), "v_dim": int(v_dim), "binary_treatment": bool(binary_treatment)}
    m["g"] = init_bnn(rs, [q] + list(g_units) + [v_dim + 1], dtype)
    m["e"] = init_bnn(rs, [v_dim] + list(e_units) + [q], dtype)
    m["f"] = init_bnn(rs, [z_dims[0] + z_dims[1] + 1] + list(f_units) + [2], dtype)
    m["h"] = init_bnn(rs, [z_dims[0] + z_dims[2]] + list(h_units) + [2], dtype)
    return m


def cast_model(m, dtype):
    out = dict(m)
    for k in ("g", "e", "f", "h"):
        out[k] = cast_bnn(m[k], dtype)
    return out


def net_dims(net):
    return [net["layers"][0][0].shape[0]] + [L[0].shape[1] for L in net["layers"]]


def _shapes(dims):
    """[(in, out)] per Flipout layer: from a chain of widths, or already a list of pairs (sibling heads)."""
    if len(dims) and isinstance(dims[0], (tuple, list)):
        return [tuple(int(v) for v in d) for d in dims]
    return [(int(dims[l]), int(dims[l + 1])) for l in range(len(dims) - 1)]


def sign_layout(dims):
    """Word offsets of the per-row sign bit string: (sin_w[l], sout_w[l], words per row rounded to 4)."""
    sin_w, sout_w, w = [], [], 0
    for fi, fo in _shapes(dims):
        sin_w.append(w)
        w += (fi + 31) // 32
        sout_w.append(w)
        w += (fo + 31) // 32
    return sin_w, sout_w, (w + 3) // 4 * 4


def draw_noise(dims, B, key, stream, net_id, dtype=np.float32, row0=0, rows=None):
    """The noise of ONE call of a net on a batch of B rows: {"eps": [...], "sin": [...], "sout": [...]}.
    The sign words of row i are keyed by row0 + i, or by rows[i] when `rows` is given."""
    k0, k1 = int(key) & 0xFFFFFFFF, (int(key) >> 32) & 0xFFFFFFFF
    shapes = _shapes(dims)
    eps = []
    for l, (fi, fo) in enumerate(shapes):
        n = fi * fo
        calls = np.arange((n + 3) // 4, dtype=np.uint32)
        bm = R.box_muller4(*R.philox4x32_10(calls, l | (net_id << 16), stream, TAG_EPS, k0, k1))
        eps.append(np.stack(bm, axis=1).reshape(-1)[:n].reshape(fi, fo).astype(dtype))
    sin_w, sout_w, words = sign_layout(shapes)
    rows = np.arange(row0, row0 + B, dtype=np.uint32) if rows is None else np.asarray(rows, dtype=np.uint32)
    W = np.empty((B, words), dtype=np.uint32)
    for c in range(words // 4):
        ws = R.philox4x32_10(rows, c | (net_id << 16), stream, TAG_SIGN, k0, k1)
        for e in range(4):
            W[:, 4 * c + e] = ws[e]

    def bits(w0, n):
        cols = np.arange(n)
        b = (W[:, w0 + (cols >> 5)] >> (cols & 31).astype(np.uint32)) & np.uint32(1)
        return (1.0 - 2.0 * b.astype(np.float64)).astype(dtype)

    sin = [bits(sin_w[l], shapes[l][0]) for l in range(len(shapes))]
    sout = [bits(sout_w[l], shapes[l][1]) for l in range(len(shapes))]
    return {"eps": eps, "sin": sin, "sout": sout}


def random_noise(rs, dims, B, dtype=np.float64):
    """Noise from a NumPy generator (for the autograd cross-checks)."""
    sh = _shapes(dims)
    return {"eps": [rs.standard_normal(s).astype(dtype) for s in sh],
            "sin": [rs.choice([-1.0, 1.0], size=(B, s[0])).astype(dtype) for s in sh],
            "sout": [rs.choice([-1.0, 1.0], size=(B, s[1])).astype(dtype) for s in sh]}


def batch_stats(x):
    """(mean, biased variance) per column, as tf.nn.moments inside BatchNormalization."""
    return x.mean(axis=0), x.var(axis=0)


def scale_of(rho):
    return rho.dtype.type(SCALE_EPS) + softplus(rho)


def forward(net, x, noise, stats=None):
    """One call of BayesianFullyConnectedNet (bnn.py:24-38) on the batch x.  `stats` overrides the batch
    statistics (the multi-workgroup kernels receive them from a reduction pass).  Returns (out, cache)."""
    t = x.dtype.type
    if stats is None and net.get("norm") == "fixed":      # build option bnn_norm="fixed": mean 0 / variance 1
        stats = (np.zeros(x.shape[1], x.dtype), np.ones(x.shape[1], x.dtype))
    mu, var = batch_stats(x) if stats is None else stats
    inv = 1.0 / np.sqrt(var + t(BN_EPS))
    xhat = (x - mu) * inv
    h = xhat * net["gamma"] + net["beta"]
    acts, pres = [h], []
    L = len(net["layers"])
    for l, (loc, rho, bias) in enumerate(net["layers"]):
        dW = scale_of(rho) * noise["eps"][l]
        pre = h @ loc + ((h * noise["sin"][l]) @ dW) * noise["sout"][l] + bias
        pres.append(pre)
        h = lrelu(pre) if l < L - 1 else pre
        acts.append(h)
    return h, {"acts": acts, "pres": pres, "xhat": xhat, "inv": inv, "noise": noise, "fixed_stats": stats is not None}


def backward(net, cache, dout, want_dx=True):
    """Gradients of a scalar loss with upstream d(loss)/d(out) = dout.
    Returns ({"gamma","beta","layers":[(dloc, drho, dbias)]}, dx)."""
    acts, pres, noise = cache["acts"], cache["pres"], cache["noise"]
    L = len(net["layers"])
    grads = [None] * L
    d = dout
    for l in reversed(range(L)):
        loc, rho, _ = net["layers"][l]
        if l < L - 1:
            d = d * np.where(pres[l] > 0, 1.0, LEAK).astype(d.dtype)
        hs = acts[l] * noise["sin"][l]
        ds = d * noise["sout"][l]
        ddW = hs.T @ ds
        drho = ddW * noise["eps"][l] * sigmoid(rho)
        grads[l] = (acts[l].T @ d, drho, d.sum(axis=0))
        dW = scale_of(rho) * noise["eps"][l]
        d = d @ loc.T + (ds @ dW.T) * noise["sin"][l]
    g = {"gamma": (d * cache["xhat"]).sum(axis=0), "beta": d.sum(axis=0), "layers": grads}
    dx = None
    if want_dx:
        dxh = d * net["gamma"]
        if cache["fixed_stats"]:
            dx = dxh * cache["inv"]
        else:
            dx = cache["inv"] * (dxh - dxh.mean(axis=0) - cache["xhat"] * (dxh * cache["xhat"]).mean(axis=0))
    return g, dx


def kl(net, prior_scale=1.0):
    """sum(net.losses): KL(N(loc, sigma^2) || N(0, prior_scale^2)) over every kernel.  Returns (value, grads)
    with grads in the structure of `backward` (zero for gamma, beta, bias)."""
    t = net["gamma"].dtype.type
    s2 = t(prior_scale) ** 2
    val = 0.0
    gl = []
    for loc, rho, bias in net["layers"]:
        sg = scale_of(rho)
        val = val + (np.log(t(prior_scale) / sg) + (sg * sg + loc * loc) / (2 * s2) - t(0.5)).sum()
        gl.append((loc / s2, (-1.0 / sg + sg / s2) * sigmoid(rho), np.zeros_like(bias)))
    return val, {"gamma": np.zeros_like(net["gamma"]), "beta": np.zeros_like(net["beta"]), "layers": gl}


def add_grads(a, b, wb=1.0):
    return {"gamma": a["gamma"] + wb * b["gamma"], "beta": a["beta"] + wb * b["beta"],
            "layers": [tuple(x + wb * y for x, y in zip(la, lb)) for la, lb in zip(a["layers"], b["layers"])]}


def flat_params(net):
    """Parameter order of the build's flat layout: gamma, beta, then per layer loc, rho, bias."""
    return [net["gamma"], net["beta"]] + [a for L in net["layers"] for a in L]


def flat_grads(g):
    return [g["gamma"], g["beta"]] + [a for L in g["layers"] for a in L]


# ---------------------------------------------------------------------------------------------------
# CausalBGM step functions with Bayesian nets
# ---------------------------------------------------------------------------------------------------
EPS = 1e-6


def _inputs(m, z, x):
    z0d, z1d, z2d, _ = m["z_dims"]
    fin = np.concatenate([z[:, :z0d + z1d], x], axis=1)
    hin = np.concatenate([z[:, :z0d], z[:, z0d + z1d:z0d + z1d + z2d]], axis=1)
    return fin, hin


def _scatter_dz(m, dz, dfin=None, dhin=None):
    z0d, z1d, z2d, _ = m["z_dims"]
    if dfin is not None:
        dz[:, :z0d + z1d] += dfin[:, :z0d + z1d]
    if dhin is not None:
        dz[:, :z0d] += dhin[:, :z0d]
        dz[:, z0d + z1d:z0d + z1d + z2d] += dhin[:, z0d:]
    return dz


def _gauss(ssq, s_raw, dim, Bn, t, fixed=None):
    """fixed: params['sigma_v' | 'sigma_x' | 'sigma_y'] (causalbgm/base.py:161,195,224,257,268,283): the variance is sigma^2, the
    variance head is not read (zero gradient)."""
    if fixed is not None:
        s2 = np.full_like(ssq, t(fixed) * t(fixed))
        return ssq / (2 * s2) + t(dim) * np.log(s2) / 2, s2, np.zeros_like(ssq)
    s2 = softplus(s_raw) + t(EPS)
    loss_b = ssq / (2 * s2) + t(dim) * np.log(s2) / 2
    ds_raw = (-ssq / (2 * s2 * s2) + t(dim) / (2 * s2)) / t(Bn) * sigmoid(s_raw)
    return loss_b, s2, ds_raw


def theta_step(m, name, z, x, y, v, noise, kl_weight):
    """update_g_net / update_h_net / update_f_net with use_bnn (:156-243): ONE call of the net, batch-mean NLL
    + kl_weight * sum(KL).  Returns (loss, aux, grads)."""
    t = z.dtype.type
    Bn = len(z)
    net = m[name]
    fin, hin = _inputs(m, z, x)
    inp = {"g": z, "f": fin, "h": hin}[name]
    out, c = forward(net, inp, noise)
    dout = np.zeros_like(out)
    if name == "g":
        p = m["v_dim"]
        d = v - out[:, :p]
        loss_b, s2, ds = _gauss((d ** 2).sum(axis=1), out[:, -1], p, Bn, t, m.get("sigma_v"))
        dout[:, :p] = -d / s2[:, None] / t(Bn)
        dout[:, -1] = ds
        loss, aux = loss_b.mean(), (d ** 2).mean()
    elif name == "h" and m["binary_treatment"]:
        l = out[:, 0]
        loss = (np.maximum(l, 0) - l * x[:, 0] + np.log1p(np.exp(-np.abs(l)))).mean()
        aux = loss
        dout[:, 0] = (sigmoid(l) - x[:, 0]) / t(Bn)
    else:
        tgt = x if name == "h" else y
        d = tgt[:, 0] - out[:, 0]
        loss_b, s2, ds = _gauss(d ** 2, out[:, -1], 1, Bn, t, m.get("sigma_x" if name == "h" else "sigma_y"))
        dout[:, 0] = -d / s2 / t(Bn)
        dout[:, -1] = ds
        loss, aux = loss_b.mean(), (d ** 2).mean()
    g, _ = backward(net, c, dout, want_dx=False)
    klv, klg = kl(net)
    return loss + t(kl_weight) * klv, aux, add_grads(g, klg, t(kl_weight))


def z_step(m, z, x, y, v, noises):
    """update_latent_variable_sgd with use_bnn (:246-302).  Each net is called TWICE with independent noise:
    the mean comes from the first call, the variance head from the second (:256-260, :267-271, :281-285).
    noises = {"g": (n1, n2), "h": (n1, n2), "f": (n1, n2)}.  Returns (loss, dz)."""
    t = z.dtype.type
    Bn = len(z)
    p = m["v_dim"]
    fin, hin = _inputs(m, z, x)
    dz = z / t(Bn)
    total = ((z ** 2).sum(axis=1) / 2).mean()
    # g
    o1, c1 = forward(m["g"], z, noises["g"][0])
    o2, c2 = forward(m["g"], z, noises["g"][1])
    d = v - o1[:, :p]
    loss_b, s2, ds = _gauss((d ** 2).sum(axis=1), o2[:, -1], p, Bn, t, m.get("sigma_v"))
    total = total + loss_b.mean()
    do1 = np.zeros_like(o1); do1[:, :p] = -d / s2[:, None] / t(Bn)
    do2 = np.zeros_like(o2); do2[:, -1] = ds
    dz = dz + backward(m["g"], c1, do1)[1] + backward(m["g"], c2, do2)[1]
    # h
    o1, c1 = forward(m["h"], hin, noises["h"][0])
    do1 = np.zeros_like(o1)
    if m["binary_treatment"]:
        l = o1[:, 0]
        total = total + (np.maximum(l, 0) - l * x[:, 0] + np.log1p(np.exp(-np.abs(l)))).mean()
        do1[:, 0] = (sigmoid(l) - x[:, 0]) / t(Bn)
        dh = backward(m["h"], c1, do1)[1]
    else:
        o2, c2 = forward(m["h"], hin, noises["h"][1])
        d = x[:, 0] - o1[:, 0]
        loss_b, s2, ds = _gauss(d ** 2, o2[:, -1], 1, Bn, t, m.get("sigma_x"))
        total = total + loss_b.mean()
        do1[:, 0] = -d / s2 / t(Bn)
        do2 = np.zeros_like(o2); do2[:, -1] = ds
        dh = backward(m["h"], c1, do1)[1] + backward(m["h"], c2, do2)[1]
    # f
    o1, c1 = forward(m["f"], fin, noises["f"][0])
    o2, c2 = forward(m["f"], fin, noises["f"][1])
    d = y[:, 0] - o1[:, 0]
    loss_b, s2, ds = _gauss(d ** 2, o2[:, -1], 1, Bn, t, m.get("sigma_y"))
    total = total + loss_b.mean()
    do1 = np.zeros_like(o1); do1[:, 0] = -d / s2 / t(Bn)
    do2 = np.zeros_like(o2); do2[:, -1] = ds
    df = backward(m["f"], c1, do1)[1] + backward(m["f"], c2, do2)[1]
    return total, _scatter_dz(m, dz, df, dh)


def log_posterior(m, x, y, v, z, noises, stats=None):
    """get_log_posterior with use_bnn (:765-817): one call of g, h, f on the block, batch statistics of the block.
    noises = {"g": n, "h": n, "f": n}; stats = optional {"g": (mean, var), ...} overrides."""
    t = z.dtype.type
    p = m["v_dim"]
    fin, hin = _inputs(m, z, x)
    st = stats or {}
    og, _ = forward(m["g"], z, noises["g"], st.get("g"))
    oh, _ = forward(m["h"], hin, noises["h"], st.get("h"))
    of, _ = forward(m["f"], fin, noises["f"], st.get("f"))
    s2v = softplus(og[:, -1]) + t(EPS) if m.get("sigma_v") is None else t(m["sigma_v"]) ** 2
    lv = ((v - og[:, :p]) ** 2).sum(axis=1) / (2 * s2v) + t(p) * np.log(s2v) / 2
    if m["binary_treatment"]:
        l = oh[:, 0]
        lx = np.maximum(l, 0) - l * x[:, 0] + np.log1p(np.exp(-np.abs(l)))
    else:
        s2x = softplus(oh[:, -1]) + t(EPS) if m.get("sigma_x") is None else t(m["sigma_x"]) ** 2
        lx = (x[:, 0] - oh[:, 0]) ** 2 / (2 * s2x) + np.log(s2x) / 2
    s2y = softplus(of[:, -1]) + t(EPS) if m.get("sigma_y") is None else t(m["sigma_y"]) ** 2
    ly = (y[:, 0] - of[:, 0]) ** 2 / (2 * s2y) + np.log(s2y) / 2
    return -(lv + lx + ly + (z ** 2).sum(axis=1) / 2)


# ---------------------------------------------------------------------------------------------------
# posterior sampling on a block-structured panel (predict, causalbgm/base.py:573-668 with use_bnn)
# ---------------------------------------------------------------------------------------------------
def block_key(seed, blk):
    """Noise key of block `blk`: (seed_lo, seed_hi + blk)."""
    return (int(seed) & 0xFFFFFFFF) | ((((int(seed) >> 32) + int(blk)) & 0xFFFFFFFF) << 32)


def log_posterior_blocks(m, x, y, v, z, block_rows, seed, stream, block0=0):
    """get_log_posterior applied block by block (each block = one reference call: own statistics, own noise)."""
    n = len(z)
    out = np.empty(n, dtype=z.dtype)
    for b, lo in enumerate(range(0, n, block_rows)):
        hi = min(n, lo + block_rows)
        key = block_key(seed, block0 + b)
        noises = {k: draw_noise(net_dims(m[k]), hi - lo, key, stream, NET_ID[k], dtype=z.dtype) for k in ("g", "h", "f")}
        out[lo:hi] = log_posterior(m, x[lo:hi], y[lo:hi], v[lo:hi], z[lo:hi], noises)
    return out


def mh_iteration(m, x, y, v, z, it, q_sd, seed, block_rows, block0=0, row_base=0):
    """One iteration of metropolis_hastings_sampler (:860-871) for every block; both log-posteriors are evaluated
    afresh (proposal first: stream 2 it, then the current state: stream 2 it + 1).  Returns (new z, accepted)."""
    n, q = z.shape
    rows = np.arange(row_base, row_base + n)
    prop = (z + z.dtype.type(q_sd) * R.normals(rows, it, q, R.TAG_PROP, seed).astype(z.dtype)).astype(z.dtype)
    lpp = log_posterior_blocks(m, x, y, v, prop, block_rows, seed, 2 * it, block0)
    lpc = log_posterior_blocks(m, x, y, v, z, block_rows, seed, 2 * it + 1, block0)
    u = R.uniforms(rows, it, R.TAG_ACC, seed)
    acc = u < np.exp(np.minimum(lpp - lpc, 0))
    out = z.copy()
    out[acc] = prop[acc]
    return out, acc, lpp, lpc


def effects_draw(m, z, xvals, d, n_keep_iter, sample_y, seed, block_rows, block0=0, row_base=0):
    """infer_from_latent_posterior (:671-763) for ONE kept draw z [n, q] (draw index d, kept by MH iteration
    n_keep_iter = burn_in + d): f-net at each treatment value of xvals on every block (own statistics, noise stream
    0x40000000 + d * len(xvals) + k).  Returns y [len(xvals), n] (mu, or mu + sqrt(s2) * outcome noise)."""
    n = len(z)
    t = z.dtype.type
    z0d, z1d = m["z_dims"][0], m["z_dims"][1]
    nd = len(xvals)
    out = np.empty((nd, n), dtype=z.dtype)
    nz = R.normals_seq(np.arange(row_base, row_base + n), n_keep_iter, nd, R.TAG_YNOISE, seed).astype(z.dtype)
    dims = net_dims(m["f"])
    for b, lo in enumerate(range(0, n, block_rows)):
        hi = min(n, lo + block_rows)
        key = block_key(seed, block0 + b)
        for k, xv in enumerate(xvals):
            noise = draw_noise(dims, hi - lo, key, 0x40000000 + d * nd + k, NET_ID["f"], dtype=z.dtype)
            inp = np.concatenate([z[lo:hi, :z0d + z1d], np.full((hi - lo, 1), xv, dtype=z.dtype)], axis=1)
            o, _ = forward(m["f"], inp, noise)
            yk = o[:, 0]
            if sample_y:
                yk = yk + (np.sqrt(softplus(o[:, 1]) + t(EPS)) if m.get("sigma_y") is None else t(m["sigma_y"])) * nz[lo:hi, k]
            out[k, lo:hi] = yk
    return out


def evaluate(m, data, data_z=None, x_values=None, seed=0, stream=0):
    """evaluate with use_bnn (:534-570); the panel is one batch.  Noise streams: `stream` for e, g, h, f; stream + 1 + k
    for the k-th counterfactual call of f.  Returns (z, causal_pre, mse_x, mse_y, mse_v)."""
    x, y, v = data
    t = v.dtype.type
    n = len(x)
    key = block_key(seed, 0)
    nz = lambda k, s: draw_noise(net_dims(m[k]), n, key, s, NET_ID[k], dtype=v.dtype)
    if data_z is None:
        data_z, _ = forward(m["e"], v, nz("e", stream))
    fin, hin = _inputs(m, data_z, x)
    v_pred = forward(m["g"], data_z, nz("g", stream))[0][:, :m["v_dim"]]
    y_pred = forward(m["f"], fin, nz("f", stream))[0][:, 0]
    x_pred = forward(m["h"], hin, nz("h", stream))[0][:, 0]
    if m["binary_treatment"]:
        x_pred = sigmoid(x_pred)
    mse_v, mse_x, mse_y = ((v - v_pred) ** 2).mean(), ((x[:, 0] - x_pred) ** 2).mean(), ((y[:, 0] - y_pred) ** 2).mean()
    z01 = fin[:, :-1]
    vals = [1.0, 0.0] if m["binary_treatment"] else list(x_values)
    mus = [forward(m["f"], np.concatenate([z01, np.full((n, 1), xv, dtype=v.dtype)], axis=1), nz("f", stream + 1 + k))[0][:, 0]
           for k, xv in enumerate(vals)]
    causal = (mus[0] - mus[1]) if m["binary_treatment"] else np.array([mu.mean() for mu in mus], dtype=v.dtype)
    return data_z, causal, mse_x, mse_y, mse_v


# ---------------------------------------------------------------------------------------------------
# EGM warm start with Bayesian nets (train_disc_step :305-330, train_gen_step :332-377 with use_bnn)
# ---------------------------------------------------------------------------------------------------
EGM_CALLS = ("g1", "g1s", "e1", "e2", "g2", "f", "fs", "h", "hs")    # noise stream of call c = stream + index


def egm_noises(m, B, key, stream, dtype=np.float32, disc_only=False):
    """The noise of every network call of one EGM step.  gen step: the nine calls above in this order;
    disc step: only e (call "e1", stream + 0)."""
    if disc_only:
        return {"e1": draw_noise(net_dims(m["e"]), B, key, stream, NET_ID["e"], dtype)}
    return {c: draw_noise(net_dims(m[c[0]]), B, key, stream + i, NET_ID[c[0]], dtype) for i, c in enumerate(EGM_CALLS)}


def egm_disc_step_grads(m, dz, z, v, eps, noises):
    """train_disc_step with a Bayesian encoder: z_ = e_net(v) is one noisy call (no gradient flows into e)."""
    from . import egm as OE
    B = z.shape[0]
    z_, _ = forward(m["e"], v, noises["e1"])
    zhat = z * eps + z_ * (1.0 - eps)
    grads = OE.zero_disc_grads(dz)
    out_f, cache_f = OE.disc_forward(dz, z_)
    out_r, cache_r = OE.disc_forward(dz, z)
    dz_loss = -out_r.mean() + out_f.mean()
    OE.disc_backward(dz, cache_f, np.full_like(out_f, 1.0 / B), grads)
    OE.disc_backward(dz, cache_r, np.full_like(out_r, -1.0 / B), grads)
    gp, _ = OE.gradient_penalty_and_grads(dz, zhat, grads, scale=10.0)
    return dz_loss, dz_loss + 10.0 * gp, grads


def egm_gen_step_grads(m, dz, use_z_rec, z, v, x, y, noises):
    """train_gen_step with Bayesian g, e, f, h: nine network calls, each with its own noise and its own batch
    statistics (g(z) twice: reconstruction input and variance penalty; f, h twice: mean and variance penalty).
    Returns (losses [e_adv, l2_v, l2_z, l2_x, l2_y, total], grads per net in the structure of `backward`)."""
    from . import egm as OE
    B, q = z.shape
    pdim = m["v_dim"]
    z0d, z1d, z2d, _ = m["z_dims"]
    g, e, f, h = m["g"], m["e"], m["f"], m["h"]
    gz, c_g1 = forward(g, z, noises["g1"])
    gzs, c_g1s = forward(g, z, noises["g1s"])
    v_ = gz[:, :pdim]
    z_, c_e1 = forward(e, v, noises["e1"])
    z__, c_e2 = forward(e, v_, noises["e2"])
    gv, c_g2 = forward(g, z_, noises["g2"])
    v__ = gv[:, :pdim]
    d_, c_d = OE.disc_forward(dz, z_)
    f_in, h_in = _inputs(m, z_, x)
    f_out, c_f = forward(f, f_in, noises["f"])
    f_s, c_fs = forward(f, f_in, noises["fs"])
    h_out, c_h = forward(h, h_in, noises["h"])
    h_s, c_hs = forward(h, h_in, noises["hs"])
    y_, x_ = f_out[:, :1], h_out[:, :1]
    l2_v = ((v - v__) ** 2).mean()
    l2_z = ((z - z__) ** 2).mean()
    e_adv = -d_.mean()
    if m["binary_treatment"]:
        l2_x = (np.maximum(x_, 0) - x_ * x + np.log1p(np.exp(-np.abs(x_)))).mean()
        dx_ = (sigmoid(x_) - x) / B
    else:
        l2_x = ((x_ - x) ** 2).mean()
        dx_ = 2.0 * (x_ - x) / B
    l2_y = ((y_ - y) ** 2).mean()
    sig = (gzs[:, -1] ** 2).mean() + (f_s[:, -1] ** 2).mean() + (h_s[:, -1] ** 2).mean()
    zrec = float(use_z_rec)
    total = e_adv + (l2_v + zrec * l2_z) + (l2_x + l2_y) + 0.001 * sig
    # ---- backward
    ge2, dv_ = backward(e, c_e2, zrec * (-2.0 / (B * q)) * (z - z__))
    dgz = np.zeros_like(gz); dgz[:, :pdim] = dv_
    gg1, _ = backward(g, c_g1, dgz, want_dx=False)
    dgzs = np.zeros_like(gzs); dgzs[:, -1] = 0.001 * 2.0 * gzs[:, -1] / B
    gg1s, _ = backward(g, c_g1s, dgzs, want_dx=False)
    dgv = np.zeros_like(gv); dgv[:, :pdim] = (-2.0 / (B * pdim)) * (v - v__)
    gg2, dz_ = backward(g, c_g2, dgv)
    _, dz_d = OE.disc_backward(dz, c_d, np.full_like(d_, -1.0 / B))
    dz_ = dz_ + dz_d
    df = np.zeros_like(f_out); df[:, :1] = 2.0 * (y_ - y) / B
    gf, df_in = backward(f, c_f, df)
    dfs = np.zeros_like(f_s); dfs[:, -1] = 0.001 * 2.0 * f_s[:, -1] / B
    gfs, dfs_in = backward(f, c_fs, dfs)
    dh = np.zeros_like(h_out); dh[:, :1] = dx_
    gh, dh_in = backward(h, c_h, dh)
    dhs = np.zeros_like(h_s); dhs[:, -1] = 0.001 * 2.0 * h_s[:, -1] / B
    ghs, dhs_in = backward(h, c_hs, dhs)
    dz_ = _scatter_dz(m, dz_, df_in + dfs_in, dh_in + dhs_in)
    ge1, _ = backward(e, c_e1, dz_, want_dx=False)
    grads = {"g": add_grads(add_grads(gg1, gg1s), gg2), "e": add_grads(ge1, ge2), "f": add_grads(gf, gfs), "h": add_grads(gh, ghs)}
    return np.array([e_adv, l2_v, l2_z, l2_x, l2_y, total]), grads
