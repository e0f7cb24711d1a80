"""Second, independent implementations of the three sampling / training transitions the oracle restates by hand.

TensorFlow cannot run here, so `oracle/` is the checker of every GPU parity test and nothing checks the checker END TO END (the autograd
tests cover gradients only).  These tests rebuild, from the reference's source lines and from nothing in `oracle/` except the shared
random inputs and the parameter arrays, one full transition of each kind with PyTorch modules, torch.distributions and autograd in
float64, written in a different form than the oracle's (library likelihoods instead of hand-expanded ones, autograd instead of
hand-derived backward passes, the textbook leapfrog instead of the merged-half-step form), and require the oracle's result:

  * one random-walk Metropolis-Hastings transition of CausalBGM (causalbgm/base.py:765-817, 860-871): proposal, both log posteriors,
    accept, for the continuous and the binary treatment model                                     -> oracle.causal.mh_transition
  * one Hamiltonian Monte Carlo transition of BGM with missing cells (bgm/base.py:665-705, 798-821; TFP's HamiltonianMonteCarlo
    one_step with identity mass): momentum draw, L leapfrog steps, Metropolis correction, log acceptance ratio   -> oracle.bgm.hmc_transition
  * one EGM warm-start iteration of CausalBGM (causalbgm/base.py:305-377, 400-416): g_d_freq WGAN-GP discriminator steps (double
    backward through the batch-normalised critic) and one generator / encoder step, each followed by its Keras-form Adam update
                                                                                                   -> oracle.egm.EgmState
"""
import math

import numpy as np
import pytest
import torch
import torch.nn as nn
import torch.nn.functional as F

from oracle import bgm as OBG
from oracle import causal as OC
from oracle import egm as OE
from oracle import rng as R

T = lambda a: torch.tensor(np.asarray(a), dtype=torch.float64)


class Mlp(nn.Module):
    """BaseFullyConnectedNet (networks/base.py:4-51): Dense -> LeakyReLU(0.2) ..., linear output."""

    def __init__(self, net):
        super().__init__()
        self.layers = nn.ModuleList()
        for W, b in net:
            lin = nn.Linear(W.shape[0], W.shape[1]).double()
            with torch.no_grad():
                lin.weight.copy_(T(W).t()); lin.bias.copy_(T(b))
            self.layers.append(lin)

    def forward(self, x):
        for i, lin in enumerate(self.layers):
            x = lin(x)
            if i + 1 < len(self.layers):
                x = F.leaky_relu(x, 0.2)
        return x

    def arrays(self):
        return [a for lin in self.layers for a in (lin.weight.detach().t().numpy(), lin.bias.detach().numpy())]


def causal_log_posterior(nets, z_dims, binary, x, y, v, z):
    """get_log_posterior (base.py:765-817) with library likelihoods: sum of Normal / Bernoulli log-probabilities + the N(0, I) prior.
    Returns the log density INCLUDING the Gaussian normalising constants the reference drops."""
    z0d, z1d, z2d, _ = z_dims
    p = v.shape[1]
    z0, z1, z2 = z[:, :z0d], z[:, z0d:z0d + z1d], z[:, z0d + z1d:z0d + z1d + z2d]
    g = nets["g"](z)
    lp = torch.distributions.Normal(g[:, :p], torch.sqrt(F.softplus(g[:, -1:]) + 1e-6)).log_prob(v).sum(1)
    h = nets["h"](torch.cat([z0, z2], 1))
    if binary:
        lp = lp + torch.distributions.Bernoulli(logits=h[:, 0]).log_prob(x[:, 0])
    else:
        lp = lp + torch.distributions.Normal(h[:, 0], torch.sqrt(F.softplus(h[:, 1]) + 1e-6)).log_prob(x[:, 0])
    f = nets["f"](torch.cat([z0, z1, x], 1))
    lp = lp + torch.distributions.Normal(f[:, 0], torch.sqrt(F.softplus(f[:, 1]) + 1e-6)).log_prob(y[:, 0])
    return lp + torch.distributions.Normal(0.0, 1.0).log_prob(z).sum(1)


@pytest.mark.parametrize("binary,z_dims,p", [(False, [1, 1, 1, 7], 200), (True, [3, 3, 6, 6], 100), (False, [2, 1, 2, 3], 17)])
def test_metropolis_hastings_transition(binary, z_dims, p):
    n, q_sd, seed, it = 300, 0.4, 2026, 7
    m = OC.cast_model(OC.init_model(3, z_dims, p, binary_treatment=binary), np.float64)
    rs = np.random.RandomState(1)
    q = sum(z_dims)
    v = rs.standard_normal((n, p))
    x = (rs.rand(n, 1) > 0.5).astype(np.float64) if binary else rs.exponential(size=(n, 1))
    y = x + rs.standard_normal((n, 1))
    state = rs.standard_normal((n, q))
    rows = np.arange(50, 50 + n)
    eps = R.normals(rows, it, q, R.TAG_PROP, seed).astype(np.float64)       # the shared random inputs
    u = R.uniforms(rows, it, R.TAG_ACC, seed).astype(np.float64)
    # ---- the independent transition (base.py:860-871)
    nets = {k: Mlp(m[k]) for k in ("g", "f", "h")}
    with torch.no_grad():
        cur, prop = T(state), T(state) + q_sd * T(eps)
        lp_cur = causal_log_posterior(nets, z_dims, binary, T(x), T(y), T(v), cur)
        lp_prop = causal_log_posterior(nets, z_dims, binary, T(x), T(y), T(v), prop)
        accept = T(u) < torch.exp(torch.clamp(lp_prop - lp_cur, max=0.0))
        new = torch.where(accept[:, None], prop, cur)
    # ---- the oracle
    lp0 = OC.log_posterior(m, x, y, v, state)
    st, lp1, acc = OC.mh_transition(m, x, y, v, state, lp0, it, q_sd, seed, row0=50)
    const = 0.5 * math.log(2 * math.pi) * (p + (0 if binary else 1) + 1 + q)     # the normalising constants the reference drops
    np.testing.assert_allclose(lp0, lp_cur.numpy() + const, rtol=1e-10, atol=1e-8)
    margin = np.abs(u - np.exp(np.minimum((lp_prop - lp_cur).numpy(), 0)))
    assert np.array_equal(acc[margin > 1e-9], accept.numpy()[margin > 1e-9]) and 0.02 < acc.mean() < 0.98
    np.testing.assert_allclose(st[margin > 1e-9], new.numpy()[margin > 1e-9], rtol=0, atol=1e-12)
    np.testing.assert_allclose(lp1[acc], (lp_prop.numpy() + const)[acc], rtol=1e-10, atol=1e-8)


class VarNet(nn.Module):
    """BaseVariationalNet in inference mode (networks/base.py:53-117): BatchNorm(moving statistics) -> trunk -> mean / softplus heads."""

    def __init__(self, g):
        super().__init__()
        q = g["bn"]["gamma"].shape[0]
        self.bn = nn.BatchNorm1d(q, eps=1e-3).double()
        with torch.no_grad():
            self.bn.weight.copy_(T(g["bn"]["gamma"])); self.bn.bias.copy_(T(g["bn"]["beta"]))
            self.bn.running_mean.copy_(T(g["bn"]["mean"])); self.bn.running_var.copy_(T(g["bn"]["var"]))
        self.trunk = Mlp(list(g["trunk"]) + [(np.zeros((g["trunk"][-1][0].shape[1], 1)), np.zeros(1))])
        self.trunk.layers = self.trunk.layers[:-1]
        self.mean, self.var = Mlp([g["mean"]]), Mlp([g["var"]])
        self.eval()

    def forward(self, z):
        h = self.bn(z)
        for lin in self.trunk.layers:
            h = F.leaky_relu(lin(h), 0.2)
        return self.mean(h), F.softplus(self.var(h)) + 1e-6


def test_hamiltonian_monte_carlo_transition():
    n, q, p, L, step, seed, it = 120, 10, 37, 10, 0.03, 99, 4
    m = OBG.cast_model(OBG.init_model(5, q, p), np.float64)
    rs = np.random.RandomState(2)
    g = m["g"]
    g["bn"]["gamma"] = 1.0 + 0.2 * rs.standard_normal(q); g["bn"]["beta"] = 0.1 * rs.standard_normal(q)
    g["bn"]["mean"] = 0.1 * rs.standard_normal(q); g["bn"]["var"] = 1.0 + 0.3 * rs.rand(q)
    xfull = rs.standard_normal((n, p))
    mask = rs.rand(n, p) > 0.2
    mask[0] = False                                              # a row with nothing observed: the posterior is the prior
    x = np.where(mask, xfull, 0.0)
    z0 = rs.standard_normal((n, q))
    rows = np.arange(n)
    mom = R.normals(rows, it, q, R.TAG_MOM, seed).astype(np.float64)        # the shared random inputs
    u = R.uniforms(rows, it, R.TAG_HACC, seed).astype(np.float64)
    net = VarNet(g)

    def logp(z):       # bgm/base.py:665-705: observed cells only
        mu, s2 = net(z)
        ll = torch.distributions.Normal(mu, torch.sqrt(s2)).log_prob(T(x)) * T(mask.astype(np.float64))
        return ll.sum(1) + torch.distributions.Normal(0.0, 1.0).log_prob(z).sum(1)

    def grad(z):
        z = z.detach().requires_grad_()
        return torch.autograd.grad(logp(z).sum(), z)[0]

    # ---- textbook leapfrog (Neal 2011, eq. 2.28-2.30), one chain per row, identity mass
    zc, pc = T(z0), T(mom)
    h_start = -logp(zc).detach() + 0.5 * (pc ** 2).sum(1)
    for _ in range(L):
        pc = pc + 0.5 * step * grad(zc)
        zc = zc + step * pc
        pc = pc + 0.5 * step * grad(zc)
    h_end = -logp(zc).detach() + 0.5 * (pc ** 2).sum(1)
    log_ratio = (h_start - h_end).numpy()
    accept = np.log(u) < log_ratio
    new = np.where(accept[:, None], zc.detach().numpy(), z0)
    # ---- the oracle
    zo, lp, gr, lr, acc = OBG.hmc_transition(m, z0, x, mask.astype(np.float64), step, L, it, seed)
    np.testing.assert_allclose(lr, log_ratio, rtol=1e-8, atol=1e-9)
    margin = np.abs(np.log(u) - log_ratio)
    ok = margin > 1e-7
    assert np.array_equal(acc[ok], accept[ok]) and acc.mean() > 0.3
    np.testing.assert_allclose(zo[ok], new[ok], rtol=1e-9, atol=1e-10)
    # the row without observations: the gradient is -z (dlogp/dz of the prior), whatever the generator
    np.testing.assert_allclose(grad(T(z0))[0].numpy(), -z0[0], atol=1e-12)
    np.testing.assert_allclose(gr[acc], grad(T(zo))[torch.from_numpy(acc)].numpy(), rtol=1e-8, atol=1e-9)


class Critic(nn.Module):
    """Discriminator (networks/base.py:338-385): Dense -> BatchNorm(batch statistics) -> tanh ..., Dense(1)."""

    def __init__(self, d):
        super().__init__()
        self.lin = nn.ModuleList()
        self.bn = nn.ModuleList()
        for l, (W, b) in enumerate(zip(d["W"], d["b"])):
            lin = nn.Linear(W.shape[0], W.shape[1]).double()
            with torch.no_grad():
                lin.weight.copy_(T(W).t()); lin.bias.copy_(T(b))
            self.lin.append(lin)
            if l < len(d["gamma"]):
                bn = nn.BatchNorm1d(W.shape[1], eps=1e-3).double()
                with torch.no_grad():
                    bn.weight.copy_(T(d["gamma"][l])); bn.bias.copy_(T(d["beta"][l]))
                self.bn.append(bn)
        self.train()                   # batch statistics (biased variance), as Keras' training mode

    def forward(self, x):
        for lin, bn in zip(self.lin[:-1], self.bn):
            x = torch.tanh(bn(lin(x)))
        return self.lin[-1](x)

    def arrays(self):          # oracle.egm.disc_param_list order: W..., b..., gamma..., beta...
        return ([l.weight.detach().t().numpy() for l in self.lin] + [l.bias.detach().numpy() for l in self.lin] +
                [b.weight.detach().numpy() for b in self.bn] + [b.bias.detach().numpy() for b in self.bn])

    def params(self):
        return [l.weight for l in self.lin] + [l.bias for l in self.lin] + [b.weight for b in self.bn] + [b.bias for b in self.bn]


class KerasAdam(object):
    """tf.keras.optimizers.Adam(lr, beta_1=0.9, beta_2=0.99) (base.py:95-96): theta -= lr_t m / (sqrt(v) + 1e-7),
    lr_t = lr sqrt(1 - b2^t) / (1 - b1^t)  (optimizer_v2/adam.py _resource_apply_dense, epsilon outside the bias correction)."""

    def __init__(self, params, lr, b1=0.9, b2=0.99):
        self.p, self.lr, self.b1, self.b2, self.t = list(params), lr, b1, b2, 0
        self.m = [torch.zeros_like(a) for a in self.p]
        self.v = [torch.zeros_like(a) for a in self.p]

    def step(self, grads):
        self.t += 1
        lr_t = self.lr * math.sqrt(1 - self.b2 ** self.t) / (1 - self.b1 ** self.t)
        with torch.no_grad():
            for a, g, m, v in zip(self.p, grads, self.m, self.v):
                m.mul_(self.b1).add_((1 - self.b1) * g)
                v.mul_(self.b2).add_((1 - self.b2) * g * g)
                a.sub_(lr_t * m / (v.sqrt() + 1e-7))


@pytest.mark.parametrize("binary", [False, True])
def test_egm_warm_start_iteration(binary):
    z_dims, p, B, g_d_freq = [1, 1, 1, 4], 12, 16, 5
    q = sum(z_dims)
    prm = dict(z_dims=z_dims, v_dim=p, binary_treatment=binary, use_z_rec=True, lr=2e-3)
    m = OC.cast_model(OC.init_model(7, z_dims, p, binary_treatment=binary, g_units=(10, 9), e_units=(11,), f_units=(8, 5), h_units=(7, 4)), np.float64)
    rs = np.random.RandomState(4)
    for k in ("g", "e", "f", "h"):
        m[k] = [(W, 0.05 * rs.standard_normal(b.shape)) for W, b in m[k]]
    dz = OE.cast_disc(OE.init_disc(rs, q, (9, 6)), np.float64)
    dz["gamma"] = [1.0 + 0.2 * rs.standard_normal(a.shape) for a in dz["gamma"]]
    dz["beta"] = [0.1 * rs.standard_normal(a.shape) for a in dz["beta"]]
    nets = {k: Mlp(m[k]) for k in ("g", "e", "f", "h")}
    critic = Critic(dz)
    gen_params = [a for k in ("g", "e", "f", "h") for lin in nets[k].layers for a in (lin.weight, lin.bias)]
    d_opt, g_opt = KerasAdam(critic.params(), prm["lr"]), KerasAdam(gen_params, prm["lr"])
    st = OE.EgmState({k: [(W.copy(), b.copy()) for W, b in m[k]] for k in ("g", "e", "f", "h")}, OE.cast_disc(dz, np.float64), prm)
    st.dz = {k: ([a.copy() for a in v_] if isinstance(v_, list) else v_) for k, v_ in dz.items()}
    st.d_opt = OE.Adam(OE.disc_param_list(st.dz), prm["lr"])
    batches = []
    for _ in range(g_d_freq + 1):
        v = rs.standard_normal((B, p))
        x = (rs.rand(B, 1) > 0.5).astype(np.float64) if binary else rs.exponential(size=(B, 1))
        batches.append((rs.standard_normal((B, q)), v, x, x + rs.standard_normal((B, 1)), float(rs.rand())))
    # ---- the iteration (base.py:400-416): g_d_freq critic steps, then one generator / encoder step
    for zb, v, x, y, eps in batches[:g_d_freq]:
        # train_disc_step (:305-330)
        z_fake = nets["e"](T(v))
        z_hat = (T(zb) * eps + z_fake * (1 - eps))
        d_hat = critic(z_hat)
        grad_z = torch.autograd.grad(d_hat.sum(), z_hat, create_graph=True)[0]
        gp = ((grad_z.pow(2).sum(1).sqrt() - 1.0) ** 2).mean()
        dz_loss = -critic(T(zb)).mean() + critic(z_fake).mean()
        d_loss = dz_loss + 10 * gp
        d_opt.step(torch.autograd.grad(d_loss, critic.params()))
        o_dz_loss, o_d_loss = st.disc_step(zb, v, eps)
        assert abs(o_dz_loss - float(dz_loss)) < 1e-10 and abs(o_d_loss - float(d_loss)) < 1e-9
    zb, v, x, y, _ = batches[g_d_freq]
    # train_gen_step (:332-377)
    z0d, z1d, z2d, _ = z_dims
    gz = nets["g"](T(zb))
    v_ = gz[:, :p]
    z_ = nets["e"](T(v))
    z__ = nets["e"](v_)
    v__ = nets["g"](z_)[:, :p]
    f_out = nets["f"](torch.cat([z_[:, :z0d + z1d], T(x)], 1))
    h_out = nets["h"](torch.cat([z_[:, :z0d], z_[:, z0d + z1d:z0d + z1d + z2d]], 1))
    l2_x = F.binary_cross_entropy_with_logits(h_out[:, :1], T(x)) if binary else F.mse_loss(h_out[:, :1], T(x))
    sig = (gz[:, -1] ** 2).mean() + (f_out[:, -1] ** 2).mean() + (h_out[:, -1] ** 2).mean()
    total = (-critic(z_).mean() + F.mse_loss(v__, T(v)) + 1.0 * F.mse_loss(z__, T(zb)) + l2_x + F.mse_loss(f_out[:, :1], T(y)) + 0.001 * sig)
    g_opt.step(torch.autograd.grad(total, gen_params))
    losses = st.gen_step(zb, v, x, y)
    assert abs(losses[-1] - float(total)) < 1e-10
    # ---- after the iteration: every parameter of the critic and of g, e, f, h
    for a, b in zip(critic.arrays(), OE.disc_param_list(st.dz)):
        np.testing.assert_allclose(b, a, rtol=1e-7, atol=1e-9)
    for k in ("g", "e", "f", "h"):
        for a, b in zip(nets[k].arrays(), [t_ for Wb in st.nets[k] for t_ in Wb]):
            np.testing.assert_allclose(b, a, rtol=1e-7, atol=1e-9)
        assert np.abs(nets[k].arrays()[0] - m[k][0][0]).max() > 1e-4        # the step moved the net
