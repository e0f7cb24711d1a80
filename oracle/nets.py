"""Oracle restatement of the reference's MLP modules (TEST INFRASTRUCTURE).

Follows /root/reference/src/bayesgm/models/networks/base.py:
  * BaseFullyConnectedNet  (:4-51)   Dense -> LeakyReLU(0.2) x L, linear last layer
  * BaseVariationalNet     (:53-117) BatchNorm(input) -> Dense+LeakyReLU x L ->
                                     mean head, var head = softplus + 1e-6
Keras semantics restated from the TF 2.10 documentation (TF is not installable
here -> "parity unpinned", see oracle/__init__.py):
  Dense: y = x @ W + b, W [in,out] glorot-uniform, b zeros.
  LeakyReLU(alpha): max(x, alpha*x).
  BatchNormalization: momentum 0.99, epsilon 1e-3, gamma=1, beta=0 initial,
      training -> batch mean / biased batch variance, moving stats updated by
      moving = moving*0.99 + batch*0.01; inference -> moving stats.
  tf.nn.softplus(x) = log(1 + exp(x)).
A "net" here is a list of (W, b) pairs of NumPy arrays; dtype of the
computation follows the dtype of the inputs (float32 or float64).
"""
import numpy as np

LEAK = 0.2


def glorot_uniform(rng, fan_in, fan_out, dtype=np.float32):
    lim = np.sqrt(6.0 / (fan_in + fan_out))
    return rng.uniform(-lim, lim, size=(fan_in, fan_out)).astype(dtype)


def init_mlp(rng, dims, dtype=np.float32):
    """dims = [in, h1, ..., out] -> [(W, b), ...]  (networks/base.py:17-26)."""
    return [(glorot_uniform(rng, dims[i], dims[i + 1], dtype),
             np.zeros(dims[i + 1], dtype=dtype)) for i in range(len(dims) - 1)]


def cast_net(net, dtype):
    return [(W.astype(dtype), b.astype(dtype)) for W, b in net]


def lrelu(x):
    return np.maximum(x, x.dtype.type(LEAK) * x)


def softplus(x):
    return np.maximum(x, 0) + np.log1p(np.exp(-np.abs(x)))


def sigmoid(x):
    return 1.0 / (1.0 + np.exp(-x))


def mlp_forward(net, x):
    """networks/base.py:30-51."""
    h = x
    for i, (W, b) in enumerate(net):
        h = h @ W + b
        if i < len(net) - 1:
            h = lrelu(h)
    return h


def mlp_forward_cache(net, x):
    """Forward keeping the layer inputs and pre-activations for backward."""
    acts, pres = [x], []
    h = x
    for i, (W, b) in enumerate(net):
        p = h @ W + b
        pres.append(p)
        h = lrelu(p) if i < len(net) - 1 else p
        acts.append(h)
    return h, (acts, pres)


def mlp_backward(net, cache, dout):
    """Returns (grads [(dW, db)...], dx) for upstream gradient dout."""
    acts, pres = cache
    grads = [None] * len(net)
    d = dout
    for i in reversed(range(len(net))):
        W, _ = net[i]
        if i < len(net) - 1:
            d = d * np.where(pres[i] > 0, 1.0, LEAK).astype(d.dtype)
        grads[i] = (acts[i].T @ d, d.sum(axis=0))
        d = d @ W.T
    return grads, d


# --------------------------------------------------------------------------
# BaseVariationalNet (BGM generator)  networks/base.py:53-117
# --------------------------------------------------------------------------
BN_MOMENTUM = 0.99
BN_EPS = 1e-3


def init_varnet(rng, z_dim, units, x_dim, dtype=np.float32):
    dims = [z_dim] + list(units)
    return {
        "bn": {"gamma": np.ones(z_dim, dtype), "beta": np.zeros(z_dim, dtype),
               "mean": np.zeros(z_dim, dtype), "var": np.ones(z_dim, dtype)},
        "trunk": init_mlp(rng, dims, dtype),
        "mean": (glorot_uniform(rng, dims[-1], x_dim, dtype), np.zeros(x_dim, dtype)),
        "var": (glorot_uniform(rng, dims[-1], x_dim, dtype), np.zeros(x_dim, dtype)),
    }


def varnet_bn_affine(vn, dtype=None):
    """Inference-mode BN folded to  z*scale + shift  (training=False path)."""
    bn = vn["bn"]
    scale = bn["gamma"] / np.sqrt(bn["var"] + bn["gamma"].dtype.type(BN_EPS))
    shift = bn["beta"] - bn["mean"] * scale
    return scale, shift


def varnet_forward(vn, z, eps=1e-6, training=False, update_stats=False):
    """networks/base.py:98-111.  Returns (mean, var[, cache])."""
    bn = vn["bn"]
    t = z.dtype.type
    if training:
        mu = z.mean(axis=0)
        var = z.var(axis=0)  # biased, as Keras
        zn = (z - mu) / np.sqrt(var + t(BN_EPS)) * bn["gamma"] + bn["beta"]
        if update_stats:
            bn["mean"] = bn["mean"] * t(BN_MOMENTUM) + mu * t(1 - BN_MOMENTUM)
            bn["var"] = bn["var"] * t(BN_MOMENTUM) + var * t(1 - BN_MOMENTUM)
    else:
        scale, shift = varnet_bn_affine(vn)
        zn = z * scale + shift
    h = zn
    for W, b in vn["trunk"]:
        h = lrelu(h @ W + b)
    mean = h @ vn["mean"][0] + vn["mean"][1]
    var_out = softplus(h @ vn["var"][0] + vn["var"][1]) + t(eps)
    return mean, var_out
