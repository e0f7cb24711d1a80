"""Split precision in the frozen-noise HMC of BGM with the Bayesian generator (opt-in, bgm_bvn_set_precision(2) /
params['hmc_precision'] = 'f16x3'; csrc/bgmfx_kernels.h): posterior means and the run's perturbation streamed as packed fp16 fragments,
every Flipout product (y = h loc + ((h * s_in) dW) * s_out + b, both directions) on v_mfma_f32_16x16x32_f16 with hi / lo splits and
fp32 accumulation; signs, likelihood and leapfrog fp32.

Criteria are the fp32 kernel's own (tests/test_gpu_bgm_bnn.py): the chains share the Philox streams (initial state, momenta,
acceptance uniforms, perturbation, sign words) with oracle/bgm_bnn.py hmc_sampler(frozen=True) and with bgmf_hmc_kernel, so after a
short run all rows agree except those whose accept / reject decision sat on the threshold.
reference: BGM.tfp_mcmc_sampler bgm/base.py:709-830 on get_log_posterior :665-705 with g_net = BayesianVariationalNet networks/bnn.py:40-99."""
import numpy as np
import pytest
import torch

from oracle import bgm_bnn as OV
from test_gpu_bgm_bnn import _net, _engine, _params, _linear_panel  # noqa: E402

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("q,units,p,n", [(3, (64,) * 3, 40, 100), (10, (64,) * 5, 50, 150), (16, (64,) * 5, 23, 70)])
def test_split_precision_chains_follow_oracle(q, units, p, n):
    """3 / 5 hidden layers; q = 16: the whole latent tile; p not a multiple of 16: masked head columns; a row count that is no
    multiple of the 16-chain tile; 25 % missing cells."""
    net = _net(q, units, p, seed=16)
    rs = np.random.RandomState(17)
    x = rs.standard_normal((n, p)).astype(np.float32)
    x[rs.uniform(size=x.shape) < 0.25] = np.nan
    eng = _engine(net, q, units, p, hmc_frozen_noise=True)
    eng.set_precision("f16x3")
    seed = 43
    out = eng.hmc_sample(x, n_mcmc=3, burn_in=5, step_size=0.03, n_leapfrog=4, seed=seed, row_base=9)
    mask = (~np.isnan(x)).astype(np.float32)
    xc = np.where(np.isnan(x), 0.0, x).astype(np.float32)
    ref, info = OV.hmc_sampler(OV.cast_vnet(net, np.float64), xc.astype(np.float64), mask.astype(np.float64), 3, 5, 0.03, 4, seed, row0=9,
                               return_info=True, frozen=True)
    got = out["draws"].cpu().numpy()
    assert got.shape == ref.shape
    assert abs(float(out["step"].item()) - info["step"]) < 1e-6
    close = np.abs(got - ref).max(axis=(0, 2)) < 1e-3
    print("MEASURED f16x3 frozen chains close to the oracle: %d of %d" % (close.sum(), n))
    assert close.mean() > 0.95
    eng.begin(net)                                   # a new session of the same engine keeps the mode
    again = eng.hmc_sample(x, n_mcmc=3, burn_in=5, step_size=0.03, n_leapfrog=4, seed=seed, row_base=9)
    assert torch.equal(again["draws"], out["draws"])
    eng.close()


@pytest.mark.parametrize("q,units,p,n", [(10, (64,) * 5, 500, 2500), (10, (64,) * 3, 33, 16 * 8 * 300 + 5)])
def test_split_precision_chains_equal_the_fp32_kernel(q, units, p, n):
    """BASELINE C4's shape, and a panel with more row tiles than one pass of the grid covers (idle waves in the last pass keep the
    stream moving); a run continued from the stored state (init = False) is part of the comparison."""
    net = _net(q, units, p, seed=18)
    rs = np.random.RandomState(19)
    x = rs.standard_normal((n, p)).astype(np.float32)
    x[rs.uniform(size=x.shape) < 0.1] = np.nan
    res = []
    for mode in ("fp32", "f16x3"):
        eng = _engine(net, q, units, p, hmc_frozen_noise=True)
        eng.set_precision(mode)
        dev = eng.device
        xd = torch.from_numpy(x).to(dev)
        state, logp, grad = torch.empty((n, q), device=dev), torch.empty(n, device=dev), torch.empty((n, q), device=dev)
        step = torch.full((1,), 0.02, device=dev)
        acc = torch.zeros(4, device=dev, dtype=torch.int32)
        eng.hmc_run(xd, state, logp, grad, step, 0, 2, 2 ** 30, 5, 11, init=True, row_base=5, acc_count=acc)
        eng.hmc_run(xd, state, logp, grad, step, 2, 2, 2 ** 30, 5, 11, row_base=5, acc_count=acc)
        res.append((state.cpu().numpy(), logp.cpu().numpy(), grad.cpu().numpy(), acc.cpu().numpy()))
        eng.close()
    (s0, l0, g0, a0), (s1, l1, g1, a1) = res
    close = np.abs(s0 - s1).max(axis=1) < 1e-3
    print("MEASURED f16x3 chains equal to the fp32 kernel's: %d of %d, acceptance %s vs %s" % (close.sum(), n, a0.tolist(), a1.tolist()))
    assert close.mean() > 0.99 and np.abs(a0 - a1).max() <= 0.01 * n
    assert np.abs(l0 - l1)[close].max() < 2e-3 * max(1.0, np.abs(l0).max())
    g_close = np.abs(g0 - g1).max(axis=1) < 2e-3 * np.abs(g0).max()
    assert g_close[close].mean() > 0.99, g_close[close].mean()


def test_split_precision_samples_the_prior_when_nothing_is_observed():
    """Known answer: with every cell missing the target is the latent prior N(0, I), whatever the generator."""
    q, units, p = 10, (64,) * 5, 20
    net = _net(q, units, p, seed=21)
    x = np.full((512, p), np.nan, np.float32)
    eng = _engine(net, q, units, p, hmc_frozen_noise=True)
    eng.set_precision("f16x3")
    out = eng.hmc_sample(x, n_mcmc=200, burn_in=100, step_size=0.1, n_leapfrog=5, seed=5)
    d = out["draws"].cpu().numpy().reshape(-1, q)
    assert np.abs(d.mean(0)).max() < 0.03 and np.abs(d.var(0) - 1).max() < 0.06
    eng.close()


@pytest.mark.parametrize("q,units,p,n", [(1, (64,) * 3, 5, 3), (16, (64,) * 3, 1, 17), (2, (64,) * 5, 16, 1)])
def test_split_precision_at_the_edges_of_the_shapes(q, units, p, n):
    """One latent dimension / the full latent tile, a single feature, exactly one 16-feature block, fewer rows than a tile, one row:
    chains against the float64 oracle and against the fp32 kernel."""
    net = _net(q, units, p, seed=31)
    rs = np.random.RandomState(32)
    x = rs.standard_normal((n, p)).astype(np.float32)
    if p > 1:
        x[rs.uniform(size=x.shape) < 0.2] = np.nan
    seed = 9
    got = {}
    for mode in ("fp32", "f16x3"):
        eng = _engine(net, q, units, p, hmc_frozen_noise=True)
        eng.set_precision(mode)
        got[mode] = eng.hmc_sample(x, n_mcmc=2, burn_in=4, step_size=0.03, n_leapfrog=3, seed=seed, row_base=100)["draws"].cpu().numpy()
        eng.close()
    mask = (~np.isnan(x)).astype(np.float64)
    xc = np.where(np.isnan(x), 0.0, x).astype(np.float64)
    ref = OV.hmc_sampler(OV.cast_vnet(net, np.float64), xc, mask, 2, 4, 0.03, 3, seed, row0=100, frozen=True)
    assert got["f16x3"].shape == ref.shape
    close = np.abs(got["f16x3"] - ref).max(axis=(0, 2)) < 1e-3
    same = np.abs(got["f16x3"] - got["fp32"]).max(axis=(0, 2)) < 1e-3
    assert close.sum() >= n - 1 and same.sum() >= n - 1, (close, same)


def test_split_precision_rows_do_not_depend_on_the_launch_they_ride_in():
    """A row's chain is a function of (row_base + row, seed) alone: slices of the panel reproduce the whole run bit for bit."""
    q, units, p, n = 10, (64,) * 5, 100, 5000
    net = _net(q, units, p, seed=3)
    x = np.random.RandomState(4).standard_normal((n, p)).astype(np.float32)
    eng = _engine(net, q, units, p, hmc_frozen_noise=True)
    eng.set_precision("f16x3")
    dev = eng.device
    xd = torch.from_numpy(x).to(dev)

    def run(lo, hi):
        m = hi - lo
        st, lg, gd = torch.empty((m, q), device=dev), torch.empty(m, device=dev), torch.empty((m, q), device=dev)
        step = torch.full((1,), 0.02, device=dev)
        eng.hmc_run(xd[lo:hi], st, lg, gd, step, 0, 2, 0, 3, 9, init=True, row_base=lo)
        return st.cpu().numpy()
    whole = run(0, n)
    np.testing.assert_array_equal(whole[:1008], run(0, 1008))
    np.testing.assert_array_equal(whole[n - 3000:], run(n - 3000, n))
    np.testing.assert_array_equal(whole, run(0, n))
    eng.close()


def test_split_precision_is_refused_where_the_kernel_does_not_serve():
    for q, units, frozen in [(20, (64,) * 5, True), (10, (64,) * 4, True), (10, (32,) * 3, True), (10, (64,) * 5, False)]:
        net = _net(q, units, 30, seed=1)
        eng = _engine(net, q, units, 30, hmc_frozen_noise=frozen)
        with pytest.raises(RuntimeError, match="f16x3 serves"):
            eng.set_precision("f16x3")
        eng.set_precision("fp32")
        eng.close()


def test_split_precision_through_the_class(tmp_path):
    """BGM(use_bnn=True, hmc_precision='f16x3').predict: the imputation agrees with the fp32 run of the same model and seed."""
    from bayesgm_amd.models import BGM
    n, p, q = 400, 12, 4
    data = _linear_panel(n, p, q)
    miss = data.copy()
    rs = np.random.RandomState(2)
    hole = rs.uniform(size=miss.shape) < 0.2
    miss[hole] = np.nan
    out = []
    for prec in ("fp32", "f16x3"):
        params = _params(tmp_path, p, q, bnn_mcmc_noise="frozen")
        params["g_units"] = [64, 64, 64]
        if prec != "fp32":
            params["hmc_precision"] = prec
        model = BGM(params, random_seed=7)
        model.fit(data, batch_size=32, epochs=10, epochs_per_eval=10, use_egm_init=False, verbose=0)
        imputed, _ = model.predict(miss[:64], n_mcmc=30, burn_in=30, step_size=0.05, num_leapfrog_steps=4, seed=3, bs=64)
        out.append(np.asarray(imputed))
    d = np.abs(out[0] - out[1])
    print("MEASURED imputation f16x3 vs fp32: max %.3e, mean %.3e (scale %.2f)" % (d.max(), d.mean(), np.abs(out[0]).std()))
    assert np.median(d) < 1e-3 and d.mean() < 0.02 * np.abs(out[0]).std() + 1e-3


def test_split_precision_at_one_gpus_share_of_config_c4():
    """BASELINE configs[4] with the Bayesian generator at the size one GPU of eight holds (625 000 x 500, 10 % missing, 10 leapfrog
    steps, frozen noise): two transitions on bgmf_hmc_kernel (fp32) and bgmfx_hmc_kernel (f16x3) from the same streams agree except
    at threshold decisions; rows from the head, the middle and the tail of the panel follow the float64 oracle chain of those rows."""
    q, units, p, n, L, seed = 10, (64,) * 5, 500, 625000, 10, 23
    net = _net(q, units, p, seed=28)
    res = {}
    x = None
    for mode in ("fp32", "f16x3"):
        eng = _engine(net, q, units, p, hmc_frozen_noise=True)
        eng.set_precision(mode)
        dev = eng.device
        if x is None:
            g = torch.Generator(device=dev).manual_seed(5)
            x = torch.randn(n, p, device=dev, generator=g)
            x[torch.rand(n, p, device=dev, generator=g) < 0.1] = float("nan")
        state, logp, grad = torch.empty((n, q), device=dev), torch.empty(n, device=dev), torch.empty((n, q), device=dev)
        step = torch.full((1,), 0.02, device=dev)
        acc = torch.zeros(2, device=dev, dtype=torch.int32)
        eng.hmc_run(x, state, logp, grad, step, 0, 2, 2 ** 30, L, seed, init=True, acc_count=acc)
        res[mode] = (state.cpu().numpy(), logp.cpu().numpy(), acc.cpu().numpy())
        eng.close()
    (s0, l0, a0), (s1, l1, a1) = res["fp32"], res["f16x3"]
    close = np.abs(s0 - s1).max(axis=1) < 1e-3
    print("MEASURED C4 share (Bayesian): rows equal between fp32 and f16x3 after two transitions: %.5f, accepted %s vs %s"
          % (close.mean(), a0.tolist(), a1.tolist()))
    assert close.mean() > 0.995 and np.abs(a0 - a1).max() <= 2e-3 * n
    assert np.abs(l0 - l1)[close].max() < 2e-3 * np.abs(l0).max()
    n64 = OV.cast_vnet(net, np.float64)
    for lo in (0, 312504, n - 40):
        hi = min(n, lo + 40)
        xs = x[lo:hi].cpu().numpy()
        mask = (~np.isnan(xs)).astype(np.float64)
        xc = np.where(np.isnan(xs), 0.0, xs).astype(np.float64)
        ref = OV.hmc_sampler(n64, xc, mask, 2, 0, 0.02, L, seed, row0=lo, frozen=True)
        ok = np.abs(s1[lo:hi] - ref[-1]).max(axis=1) < 1e-3
        assert ok.mean() >= 0.9, (lo, ok.mean())
