"""Host-side input generators of the hot path (NumPy).

Mirror of the reference's ``bayesgm.datasets`` for the pieces every config of
BASELINE.json needs:

* ``Base_sampler``              -- datasets/base_sampler.py:6-84
* ``Sim_Hirano_Imbens_sampler`` -- datasets/causal_samplers.py:40-67
* ``Gaussian_sampler``          -- datasets/prior_samplers.py:4-59
* ``simulate_z_hetero``         -- datasets/simulators.py:163-204 (BGM tutorial panel)

Same names, argument meaning, dtype (float32) and -- deliberately -- the same
use of NumPy's *legacy global* RNG (``np.random.seed``), so a panel generated
here is value-identical to the reference's; pinned by tests/golden/*.npz.
"""
import math
import numpy as np


def _standardize(v):
    """sklearn StandardScaler().fit_transform semantics (population std, zero-variance
    columns left unscaled), used by Base_sampler(normalize=True)."""
    from sklearn.preprocessing import StandardScaler
    return StandardScaler().fit_transform(v)


class Base_sampler(object):
    """Cyclic mini-batch sampler over (x, y, v).  datasets/base_sampler.py:29-84.

    Side effect kept from the reference: the constructor re-seeds NumPy's global
    RNG with ``random_seed`` (:31) before shuffling the index."""

    def __init__(self, x, y, v, batch_size=32, normalize=False, random_seed=123):
        assert len(x) == len(y) == len(v)
        np.random.seed(random_seed)
        self.data_x = np.array(x, dtype='float32')
        self.data_y = np.array(y, dtype='float32')
        self.data_v = np.array(v, dtype='float32')
        if len(self.data_x.shape) == 1:
            self.data_x = self.data_x.reshape(-1, 1)
        if len(self.data_y.shape) == 1:
            self.data_y = self.data_y.reshape(-1, 1)
        self.batch_size = batch_size
        if normalize:
            self.data_v = _standardize(self.data_v)
        self.sample_size = len(x)
        self.full_index = np.arange(self.sample_size)
        np.random.shuffle(self.full_index)
        self.idx_gen = self.create_idx_generator(sample_size=self.sample_size)

    def create_idx_generator(self, sample_size, random_seed=123):
        while True:
            for step in range(math.ceil(sample_size / self.batch_size)):
                if (step + 1) * self.batch_size <= sample_size:
                    yield self.full_index[step * self.batch_size:(step + 1) * self.batch_size]
                else:
                    yield np.hstack([self.full_index[step * self.batch_size:],
                                     self.full_index[:((step + 1) * self.batch_size - sample_size)]])
                    np.random.shuffle(self.full_index)

    def next_batch(self):
        indx = next(self.idx_gen)
        return self.data_x[indx, :], self.data_y[indx, :], self.data_v[indx, :]

    def load_all(self):
        return self.data_x, self.data_y, self.data_v


class Sim_Hirano_Imbens_sampler(Base_sampler):
    """Hirano-Imbens continuous-treatment simulation.  datasets/causal_samplers.py:58-67:
    v ~ Exp(1) [N x v_dim]; x ~ Exp(scale = 1/(v0+v1)); y ~ N(x + (v0+v2) exp(-x (v0+v2)), 1);
    V standardised per column; true ADRF(x) = x + 2/(1+x)^3."""

    def __init__(self, batch_size=32, N=20000, v_dim=200, seed=0):
        np.random.seed(seed)
        v = np.random.exponential(scale=1.0, size=(N, v_dim))
        rate = v[:, 0] + v[:, 1]
        scale = 1 / rate
        x = np.random.exponential(scale=scale)
        y = np.random.normal(x + (v[:, 0] + v[:, 2]) * np.exp(-x * (v[:, 0] + v[:, 2])), 1)
        x = x.reshape(-1, 1)
        y = y.reshape(-1, 1)
        super().__init__(x, y, v, batch_size=batch_size, normalize=True)


class Sim_Sun_sampler(Base_sampler):
    """Sun et al. continuous-treatment simulation.  datasets/causal_samplers.py:88-94: v ~ N(0, I) [N x v_dim];
    x ~ N(-2 sin(2 v0) + (v1^2 - 1/3) + (v2 - 1/2) + cos(v3), 1); y ~ N((v0 - 1/2) + cos(v1) + v4^2 + v5 + x, 1);
    V standardised per column (true ADRF: utils.get_ADRF(dataset='Sun'))."""

    def __init__(self, batch_size=32, N=20000, v_dim=200, seed=0):
        np.random.seed(seed)
        v = np.random.normal(0, 1, size=(N, v_dim))
        mean_x = -2 * np.sin(2 * v[:, 0]) + (v[:, 1] ** 2 - 1 / 3) + (v[:, 2] - 1 / 2) + np.cos(v[:, 3])
        x = np.random.normal(mean_x, 1)
        y = np.random.normal((v[:, 0] - 1 / 2) + np.cos(v[:, 1]) + v[:, 4] ** 2 + v[:, 5] + x, 1)
        super().__init__(x.reshape(-1, 1), y.reshape(-1, 1), v, batch_size=batch_size, normalize=True)


class Sim_Colangelo_sampler(Base_sampler):
    """Colangelo & Lee continuous-treatment simulation.  datasets/causal_samplers.py:113-127: v ~ N(0, Sigma) with the
    tridiagonal Sigma (1 on the diagonal, rho beside it); theta_l = 1 / l^2; x = d * Phi(a v.theta) + b nu - 1/2;
    y = 1.2 x + x^3 + x v0 + 1.2 v.theta + eps; eps, nu ~ N(0, 1) drawn BEFORE v; V standardised per column
    (true ADRF: utils.get_ADRF(dataset='Lee'))."""

    def __init__(self, batch_size=32, N=20000, v_dim=100, seed=0, rho=0.5, offset=(-1, 0, 1), d=1, a=3, b=0.75):
        from scipy.stats import norm
        np.random.seed(seed)
        sigma = np.zeros((v_dim, v_dim))
        for off, val in zip(offset, (rho, 1.0, rho)):
            sigma += np.diag(np.full(v_dim - abs(off), val), off)
        theta = 1.0 / np.arange(1, v_dim + 1) ** 2
        epsilon = np.random.normal(0, 1, N)
        nu = np.random.normal(0, 1, N)
        v = np.random.multivariate_normal(np.zeros(v_dim), sigma, size=[N, ])
        x = d * norm.cdf(a * v @ theta) + b * nu - 0.5
        y = 1.2 * x + x ** 3 + x * v[:, 0] + 1.2 * (v @ theta) + epsilon
        super().__init__(x.reshape(-1, 1), y.reshape(-1, 1), v, batch_size=batch_size, normalize=True)


class Gaussian_sampler(object):
    """N(mean, sd^2 I) sampler.  datasets/prior_samplers.py:4-59 (re-seeds the global
    RNG with 1024 at construction, as the reference does)."""

    def __init__(self, mean, sd=1, N=20000):
        self.total_size = N
        self.mean = mean
        self.sd = sd
        np.random.seed(1024)
        self.X = np.random.normal(self.mean, self.sd, (self.total_size, len(self.mean)))
        self.X = self.X.astype('float32')

    def train(self, batch_size, label=False):
        indx = np.random.randint(low=0, high=self.total_size, size=batch_size)
        return self.X[indx, :]

    def get_batch(self, batch_size):
        return np.random.normal(self.mean, self.sd, (batch_size, len(self.mean))).astype('float32')

    def load_all(self):
        return self.X


def simulate_z_hetero(n=20000, k=3, d=20 - 1, seed=42):
    """Latent-factor heteroskedastic panel of the BGM tutorial.  datasets/simulators.py:163-204."""
    np.random.seed(seed)
    Z = np.random.randn(n, k)
    A = np.random.randn(d, k)
    X = 0.2 * Z @ A.T + 0.1 * np.random.randn(n, d)
    w = np.random.randn(k)
    u = np.random.randn(k)
    mean_Y = np.sin(Z @ w)
    std_Y = 0.1 + 0.5 * 1 / (1 + np.exp(-(Z @ u)))
    Y = mean_Y + std_Y * np.random.randn(n)
    return X, Y


def binarize_treatment(x):
    """Binary treatment derived from the continuous Hirano-Imbens dose, 1[x > median(x)].
    NOT in the reference (its binary data sets need external ACIC/Twins files); used by
    BASELINE.json config[1] as stated in SURVEY.md section 8(d)."""
    return (x > np.median(x)).astype('float32')
