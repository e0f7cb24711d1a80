"""Host-side synthetic inputs of the hot path.

Public names follow the reference's ``bayesgm.datasets`` (datasets/base_sampler.py:29-84,
causal_samplers.py:40-67, prior_samplers.py:4-59, simulators.py:163-204) because callers import
them by name; the code is the build's own.  What has to be shared with the reference is the ORDER
in which NumPy's legacy Mersenne-Twister stream is consumed -- a panel is only value-identical to the
reference's if the same variates are drawn in the same sequence -- so every generator below documents
its draw order, draws from a ``RandomState`` seeded like the reference's ``np.random.seed(seed)`` and
then hands that state to NumPy's global generator (``_publish``), which is where the reference leaves
it for whatever code runs next.  Values are pinned bit-exactly by tests/golden/*.npz.
"""
import numpy as np


def _publish(rs):
    """Leave NumPy's global legacy generator in the state of `rs` (the reference draws from the global
    generator directly; later host code -- minibatch draws, MH initial states -- continues that stream)."""
    np.random.set_state(rs.get_state())


def _column_standardize(a):
    """Per column (a - mean) / std, the arithmetic of scikit-learn's StandardScaler.fit_transform -- which the reference
    applies to V -- restated so that a float32 panel comes out bit-identical: float64 column sums; variance as the
    corrected two-pass form (sum d^2 - (sum d)^2 / n) / n of the deviations d from the mean; columns whose variance is
    within rounding of zero are centred only; for a float32 input the centring and the scaling each round to float32."""
    a = np.asarray(a)
    n = a.shape[0]
    mu = a.sum(axis=0, dtype=np.float64) / n
    dev = a - mu                                               # float64
    var = ((dev ** 2).sum(axis=0) - dev.sum(axis=0) ** 2 / n) / n
    eps = np.finfo(np.float64).eps
    sd = np.sqrt(var)
    sd[(var <= n * eps * var + (n * mu * eps) ** 2) | (sd == 0.0)] = 1.0
    if a.dtype.kind != "f":
        return dev / sd
    centred = dev.astype(a.dtype)                              # first rounding (in-place `X -= mean` on the float array)
    return (centred / sd).astype(a.dtype)                      # second rounding (`X /= scale`)


class _EpochCursor:
    """Cyclic minibatch positions over a shuffled index: full batches in order; a batch that would run past the end
    is completed with the head of the CURRENT order, after which the order is reshuffled (with NumPy's global generator
    in whatever state the interleaved host draws have left it -- the reference's stream) and the walk restarts at 0."""

    def __init__(self, order, batch_size):
        self.order, self.b, self.pos = order, int(batch_size), 0

    def take(self):
        n, b = len(self.order), self.b
        if self.pos >= n:
            self.pos = 0
        lo, hi = self.pos, self.pos + b
        self.pos = hi
        if hi <= n:
            return self.order[lo:hi]
        picked = np.concatenate([self.order[lo:], self.order[:hi - n]])
        np.random.shuffle(self.order)
        self.pos = 0
        return picked


class Base_sampler(object):
    """Minibatch view of a panel (x [n,1], y [n,1], v [n,p]), float32.

    Draw order (stream seeded with ``random_seed``, default 123, whatever was seeded before): one shuffle of the row
    index at construction, one more after every wrapped batch.  ``normalize`` standardises V per column."""

    def __init__(self, x, y, v, batch_size=32, normalize=False, random_seed=123):
        if not (len(x) == len(y) == len(v)):
            raise AssertionError("x, y, v must have the same number of rows")
        rs = np.random.RandomState(random_seed)
        self.data_x = np.asarray(x, dtype=np.float32).reshape(len(x), -1)
        self.data_y = np.asarray(y, dtype=np.float32).reshape(len(y), -1)
        self.data_v = np.array(v, dtype=np.float32)
        if normalize:
            self.data_v = _column_standardize(self.data_v)
        self.batch_size = batch_size
        self.sample_size = len(self.data_x)
        self.full_index = np.arange(self.sample_size)
        rs.shuffle(self.full_index)
        _publish(rs)
        self._cursor = _EpochCursor(self.full_index, batch_size)

    def next_batch(self):
        rows = self._cursor.take()
        return self.data_x[rows], self.data_y[rows], self.data_v[rows]

    def load_all(self):
        return self.data_x, self.data_y, self.data_v


class Sim_Hirano_Imbens_sampler(Base_sampler):
    """Hirano-Imbens dose-response simulation (the panel of BASELINE.json's headline config).

    Draw order on the stream seeded with ``seed``: V ~ Exp(1) [N x v_dim] row-major; x_i ~ Exp(mean 1 / (v_i0 + v_i1));
    y_i ~ N(x_i + s_i exp(-x_i s_i), 1) with s_i = v_i0 + v_i2.  V is then standardised per column (x, y are not).
    True dose-response: utils.get_ADRF(dataset='Imbens') = x + 2 / (1 + x)^3."""

    def __init__(self, batch_size=32, N=20000, v_dim=200, seed=0):
        rs = np.random.RandomState(seed)
        v = rs.exponential(scale=1.0, size=(N, v_dim))
        x = rs.exponential(scale=1.0 / (v[:, 0] + v[:, 1]))
        s = v[:, 0] + v[:, 2]
        y = rs.normal(x + s * np.exp(-x * s), 1)
        Base_sampler.__init__(self, x[:, None], y[:, None], v, batch_size=batch_size, normalize=True)


class Sim_Sun_sampler(Base_sampler):
    """Sun et al. dose-response simulation (configs/Sim_Sun.yaml of the reference).

    Draw order on the stream seeded with ``seed``: V ~ N(0, 1) [N x v_dim] row-major; x_i ~ N(m_i, 1) with
    m_i = -2 sin(2 v_i0) + v_i1^2 - 1/3 + v_i2 - 1/2 + cos(v_i3); y_i ~ N(v_i0 - 1/2 + cos(v_i1) + v_i4^2 + v_i5 + x_i, 1).  V is then
    standardised per column.  True dose-response: utils.get_ADRF(dataset='Sun')."""

    def __init__(self, batch_size=32, N=20000, v_dim=200, seed=0):
        rs = np.random.RandomState(seed)
        v = rs.normal(0, 1, size=(N, v_dim))
        x = rs.normal(-2 * np.sin(2 * v[:, 0]) + (v[:, 1] ** 2 - 1 / 3) + (v[:, 2] - 1 / 2) + np.cos(v[:, 3]), 1)
        y = rs.normal((v[:, 0] - 1 / 2) + np.cos(v[:, 1]) + v[:, 4] ** 2 + v[:, 5] + x, 1)
        Base_sampler.__init__(self, x[:, None], y[:, None], v, batch_size=batch_size, normalize=True)


class Sim_Colangelo_sampler(Base_sampler):
    """Colangelo-Lee dose-response simulation (configs/Sim_Colangelo.yaml of the reference).

    Draw order on the stream seeded with ``seed``: eps ~ N(0, 1) [N], nu ~ N(0, 1) [N], THEN V ~ N(0, Sigma) [N x v_dim] with the
    tridiagonal Sigma (1 on the diagonal, ``rho`` beside it; NumPy's multivariate_normal, i.e. through an SVD of Sigma).  With
    theta_l = 1 / l^2: x = d Phi(a V theta) + b nu - 1/2, y = 1.2 x + x^3 + x v_0 + 1.2 V theta + eps.  V is then standardised per
    column.  True dose-response: utils.get_ADRF(dataset='Lee')."""

    def __init__(self, batch_size=32, N=20000, v_dim=100, seed=0, rho=0.5, offset=(-1, 0, 1), d=1, a=3, b=0.75):
        from scipy.special import ndtr
        rs = np.random.RandomState(seed)
        sigma = sum(np.diag(np.full(v_dim - abs(o), val), o) for o, val in zip(offset, (rho, 1.0, rho)))
        theta = 1.0 / np.arange(1, v_dim + 1) ** 2
        eps = rs.normal(0, 1, N)
        nu = rs.normal(0, 1, N)
        v = rs.multivariate_normal(np.zeros(v_dim), sigma, size=[N, ])
        lin = v @ theta
        x = d * ndtr(a * lin) + b * nu - 0.5
        y = 1.2 * x + x ** 3 + x * v[:, 0] + 1.2 * lin + eps
        Base_sampler.__init__(self, x[:, None], y[:, None], v, batch_size=batch_size, normalize=True)


class Gaussian_sampler(object):
    """Isotropic Gaussian prior sampler N(mean, sd^2 I), float32.

    Construction seeds the stream with 1024 and draws the ``N`` stored rows ``X``; ``get_batch`` / ``train`` draw from
    NumPy's GLOBAL generator at call time (so they follow whatever the caller seeded, as in the reference, where the
    EGM loop interleaves them with its minibatch index draws)."""

    def __init__(self, mean, sd=1, N=20000):
        self.mean = np.asarray(mean, dtype=np.float64)
        self.sd = sd
        self.total_size = N
        rs = np.random.RandomState(1024)
        self.X = rs.normal(self.mean, self.sd, (N, self.mean.shape[0])).astype(np.float32)
        _publish(rs)

    def get_batch(self, batch_size):
        return np.random.normal(self.mean, self.sd, (batch_size, self.mean.shape[0])).astype(np.float32)

    def train(self, batch_size, label=False):
        return self.X[np.random.randint(low=0, high=self.total_size, size=batch_size)]

    def load_all(self):
        return self.X


def simulate_z_hetero(n=20000, k=3, d=20 - 1, seed=42):
    """Heteroskedastic latent-factor panel of the BGM tutorial: X = 0.2 Z A' + 0.1 E [n x d], Y = sin(Z w) + s(Z) e with
    s = 0.1 + 0.5 sigmoid(Z u).  Draw order on the stream seeded with ``seed`` (all standard normal):
    Z [n x k], A [d x k], E [n x d], w [k], u [k], e [n].  Returns float64 (X, Y)."""
    rs = np.random.RandomState(seed)
    z = rs.standard_normal((n, k))
    load = rs.standard_normal((d, k))
    x = 0.2 * z @ load.T + 0.1 * rs.standard_normal((n, d))
    w, u = rs.standard_normal(k), rs.standard_normal(k)
    spread = 0.1 + 0.5 * 1 / (1 + np.exp(-(z @ u)))
    y = np.sin(z @ w) + spread * rs.standard_normal(n)
    _publish(rs)
    return x, y


def binarize_treatment(x):
    """Binary treatment derived from the continuous Hirano-Imbens dose, 1[x > median(x)].
    NOT in the reference (its binary data sets need external ACIC/Twins files); used by
    BASELINE.json config[1] as stated in SURVEY.md section 8(d)."""
    return (x > np.median(x)).astype('float32')
