"""oracle/bgm_bnn.py (BGM with the Bayesian generator) against an independent PyTorch-autograd second opinion (float64)."""
import numpy as np
import pytest
import torch

from oracle import bgm_bnn as OV
from oracle import bnn as OB


def _torch_net(net):
    tn = {k: torch.tensor(net[k], dtype=torch.float64, requires_grad=(k in ("gamma", "beta"))) for k in ("gamma", "beta", "mean_mv", "var_mv")}
    tn["layers"] = [tuple(torch.tensor(a, dtype=torch.float64, requires_grad=True) for a in L) for L in OV.layers_of(net)]
    return tn


def _torch_forward(tn, z, noise, training):
    T = len(tn["layers"]) - 2
    if training:
        mu, var = z.mean(0), z.var(0, unbiased=False)
    else:
        mu, var = tn["mean_mv"], tn["var_mv"]
    h = (z - mu) / torch.sqrt(var + 1e-3) * tn["gamma"] + tn["beta"]

    def flip(h, l):
        loc, rho, b = tn["layers"][l]
        sg = float(np.finfo(np.float32).eps) + torch.nn.functional.softplus(rho)
        return h @ loc + ((h * torch.tensor(noise["sin"][l])) @ (sg * torch.tensor(noise["eps"][l]))) * torch.tensor(noise["sout"][l]) + b
    for l in range(T):
        h = torch.nn.functional.leaky_relu(flip(h, l), 0.2)
    return flip(h, T), torch.nn.functional.softplus(flip(h, T + 1)) + 1e-6


@pytest.mark.parametrize("training", [True, False])
def test_backward_matches_autograd(training):
    rs = np.random.RandomState(3)
    q, p, B = 5, 11, 9
    net = OV.cast_vnet(OV.init_vnet(rs, q, [16, 12], p), np.float64)
    net["mean_mv"] = rs.standard_normal(q) * 0.3
    net["var_mv"] = 0.5 + rs.uniform(size=q)
    net["gamma"] = 1.0 + 0.1 * rs.standard_normal(q)
    z, x = rs.standard_normal((B, q)), rs.standard_normal((B, p))
    noise = OB.random_noise(rs, OV.shapes(net), B)
    mean, s2, c = OV.vforward(net, z, noise, training=training)
    wm, ws = rs.standard_normal((B, p)), rs.standard_normal((B, p))
    # loss = sum(wm * mean) + sum(ws * s_raw)
    g, dz = OV.vbackward(net, c, wm, ws)
    tn = _torch_net(net)
    zt = torch.tensor(z, requires_grad=True)
    mt, s2t = _torch_forward(tn, zt, noise, training)
    np.testing.assert_allclose(mt.detach().numpy(), mean, rtol=1e-10, atol=1e-12)
    np.testing.assert_allclose(s2t.detach().numpy(), s2, rtol=1e-10, atol=1e-12)
    s_raw = torch.log(torch.expm1(s2t - 1e-6))
    loss = (torch.tensor(wm) * mt).sum() + (torch.tensor(ws) * s_raw).sum()
    loss.backward()
    np.testing.assert_allclose(dz, zt.grad.numpy(), rtol=1e-7, atol=1e-10)
    np.testing.assert_allclose(g["gamma"], tn["gamma"].grad.numpy(), rtol=1e-7, atol=1e-10)
    np.testing.assert_allclose(g["beta"], tn["beta"].grad.numpy(), rtol=1e-7, atol=1e-10)
    for (a, b_, c_), L in zip(g["layers"], tn["layers"]):
        np.testing.assert_allclose(a, L[0].grad.numpy(), rtol=1e-7, atol=1e-10)
        np.testing.assert_allclose(b_, L[1].grad.numpy(), rtol=1e-7, atol=1e-10)
        np.testing.assert_allclose(c_, L[2].grad.numpy(), rtol=1e-7, atol=1e-10)


def test_kl_matches_torch_distributions():
    rs = np.random.RandomState(5)
    net = OV.cast_vnet(OV.init_vnet(rs, 4, [8], 6), np.float64)
    val, gk = OV.vkl(net)
    tn = _torch_net(net)
    tot = 0.0
    for loc, rho, b in tn["layers"]:
        sg = float(np.finfo(np.float32).eps) + torch.nn.functional.softplus(rho)
        qd = torch.distributions.Normal(loc, sg)
        pd = torch.distributions.Normal(torch.zeros_like(loc), 0.1)
        tot = tot + torch.distributions.kl_divergence(qd, pd).sum()
        tot = tot - torch.distributions.Normal(torch.zeros_like(b), 0.1).log_prob(b).sum()     # KL(Deterministic || prior)
    tot.backward()
    np.testing.assert_allclose(val, tot.item(), rtol=1e-10)
    for (a, b_, c_), L in zip(gk["layers"], tn["layers"]):
        np.testing.assert_allclose(a, L[0].grad.numpy(), rtol=1e-8, atol=1e-10)
        np.testing.assert_allclose(b_, L[1].grad.numpy(), rtol=1e-8, atol=1e-10)
        np.testing.assert_allclose(c_, L[2].grad.numpy(), rtol=1e-8, atol=1e-10)


def test_logpost_grad_matches_autograd():
    rs = np.random.RandomState(7)
    q, p, n = 4, 9, 6
    net = OV.cast_vnet(OV.init_vnet(rs, q, [16, 16], p), np.float64)
    z, x = rs.standard_normal((n, q)), rs.standard_normal((n, p))
    mask = (rs.uniform(size=(n, p)) > 0.3).astype(np.float64)
    noise = OV.draw(net, n, 11, 3, dtype=np.float64)
    lp, gr = OV.log_posterior_and_grad(net, z, x, mask, noise)
    tn = _torch_net(net)
    zt = torch.tensor(z, requires_grad=True)
    mt, s2t = _torch_forward(tn, zt, noise, False)
    lpt = -((torch.tensor(mask) * ((torch.tensor(x) - mt) ** 2 / (2 * s2t) + 0.5 * torch.log(s2t))).sum(1) + (zt ** 2).sum(1) / 2)
    lpt.sum().backward()
    np.testing.assert_allclose(lp, lpt.detach().numpy(), rtol=1e-10)
    np.testing.assert_allclose(gr, zt.grad.numpy(), rtol=1e-7, atol=1e-10)


def test_fit_steps_reduce_the_loss_and_move_the_statistics():
    rs = np.random.RandomState(1)
    q, p, n = 3, 8, 256
    net = OV.init_vnet(rs, q, [16, 16], p)
    zt = rs.standard_normal((n, q)).astype(np.float32)
    data = (zt @ rs.standard_normal((q, p)) + 0.1 * rs.standard_normal((n, p))).astype(np.float32)
    st = OV.FitState(net, zt.copy(), 5e-3, 5e-3, 5e-5, seed=9)
    first = last = None
    for ep in range(6):
        perm = rs.permutation(n)
        tot = 0.0
        for k in range(0, n - 31, 32):
            tot += st.step(data, perm[k:k + 32])[1]
        first = tot if first is None else first
        last = tot
    assert last < 0.8 * first
    assert np.abs(net["mean_mv"]).max() > 0 and np.all(np.isfinite(net["var_mv"]))


def test_hmc_all_missing_rows_sample_the_prior():
    rs = np.random.RandomState(2)
    q, p, n = 2, 5, 64
    net = OV.init_vnet(rs, q, [8], p)
    x = np.zeros((n, p), np.float32)
    mask = np.zeros((n, p), np.float32)
    post, info = OV.hmc_sampler(net, x, mask, 60, 60, step_size=0.3, n_leapfrog=5, seed=4, return_info=True)
    assert post.shape == (60, n, q)
    assert abs(post.mean()) < 0.1 and abs(post.var() - 1.0) < 0.2 and info["accept_rate"] > 0.5
