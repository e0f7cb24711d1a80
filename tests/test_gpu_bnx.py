"""Split-precision ("f16 x 3") form of the Bayesian-network sampling kernels (csrc/bnx_kernels.h inside bnf_mh_kernel /
bnf_effects_kernel, bgm_bnn_set_precision(h, 2)) against oracle/bnn.py in float64 through the C ABI: same Philox streams, sign words
and perturbation draws as the fp32 kernels of test_gpu_bnf.py, same tolerances as their tests."""
import numpy as np
import pytest
import torch

from oracle import bnn as OB
from test_gpu_bnf import _model, _panel, _engine, f64

pytestmark = pytest.mark.gpu


def _x3(m, **kw):
    eng = _engine(m, **kw)
    eng.set_precision("f16x3")
    return eng


@pytest.mark.parametrize("binary,p,z_dims,n,bs", [
    (False, 200, (1, 1, 1, 7), 700, 300),      # the bench shape: K = 16 first layers, 13 output tiles, ragged last block
    (True, 100, (3, 3, 6, 6), 520, 520),       # q = 18: the first layers are one K = 32 block (KS = 5)
    (False, 37, (2, 1, 2, 3), 100, 64),        # p % 4 != 0; variance column in the middle of a tile
    (False, 191, (1, 1, 1, 7), 90, 33),        # p + 1 = 192: the variance column is the last of an even tile count
    (True, 50, (5, 5, 5, 5), 130, 40),         # q = 20 (KS = 6)
    (False, 50, (4, 4, 4, 4), 75, 75),         # q = 16: the treatment sits alone in the second half of the K = 32 block
    (False, 60, (2, 3, 4, 6), 64, 64),         # q = 15, KS = 4: a full K = 16 block
])
def test_logpost_blocks_match_oracle(binary, p, z_dims, n, bs):
    m = _model(binary, z_dims=z_dims, p=p)
    z, x, y, v = _panel(m, n)
    eng = _x3(m)
    T = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(eng.device)
    seed, stream = (3 << 32) | 1234, 77
    got = eng.logpost(T(x[:, 0]), T(y[:, 0]), T(v), T(z), bs, seed, stream, block0=2).cpu().numpy()
    ref = OB.log_posterior_blocks(OB.cast_model(m, np.float64), f64(x), f64(y), f64(v), f64(z), bs, seed, stream, block0=2)
    assert np.abs(got - ref).max() < 2e-5 * np.abs(ref).max() + 2e-3, np.abs(got - ref).max()
    # and as close to float64 as the fp32 kernels are (within a factor)
    eng.set_precision("fp32")
    got32 = eng.logpost(T(x[:, 0]), T(y[:, 0]), T(v), T(z), bs, seed, stream, block0=2).cpu().numpy()
    e3, e32 = np.abs(got - ref).max(), np.abs(got32 - ref).max()
    assert e3 < 4.0 * e32 + 1e-4 * (1.0 + np.abs(ref).max() * 1e-3), (e3, e32)
    assert np.abs(got - got32).max() > 0.0          # the split kernels did run
    eng.close()


def test_split_blobs_follow_parameter_updates():
    m = _model(False, p=50)
    z, x, y, v = _panel(m, 64)
    eng = _x3(m)
    T = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(eng.device)
    a = eng.logpost(T(x[:, 0]), T(y[:, 0]), T(v), T(z), 64, 5, 1).cpu().numpy()
    m2 = _model(False, p=50, seed=3)
    from bayesgm_amd.bnn_engine import flatten_bnn, NETS
    eng.write(np.concatenate([flatten_bnn(m2[k]) for k in NETS]))
    b = eng.logpost(T(x[:, 0]), T(y[:, 0]), T(v), T(z), 64, 5, 1).cpu().numpy()
    ref = OB.log_posterior_blocks(OB.cast_model(m2, np.float64), f64(x), f64(y), f64(v), f64(z), 64, 5, 1)
    assert np.abs(a - b).max() > 1.0
    assert np.abs(b - ref).max() < 2e-5 * np.abs(ref).max() + 2e-3
    eng.close()


@pytest.mark.parametrize("binary,p,z_dims", [(False, 50, (1, 1, 1, 7)), (True, 100, (3, 6, 3, 6))])
def test_mh_iterations_match_oracle(binary, p, z_dims):
    m = _model(binary, p=p, z_dims=z_dims)
    n, bs = 600, 256
    z, x, y, v = _panel(m, n)
    eng = _x3(m)
    T = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(eng.device)
    seed = (9 << 32) | 4321
    state = T(z)
    acc = torch.zeros(1, dtype=torch.int32, device=eng.device)
    accb = torch.zeros((2, 3), dtype=torch.int32, device=eng.device)
    eng.mh_run(T(x[:, 0]), T(y[:, 0]), T(v), state, bs, it_begin=5, n_iters=2, burn_in=0, q_sd=0.3, seed=seed, row_base=1000,
               acc_count=acc, acc_blocks=accb)
    zo = f64(z)
    m64 = OB.cast_model(m, np.float64)
    n_acc, fragile = 0, np.zeros(n, bool)
    for it in (5, 6):
        zo, a, lpp, lpc = OB.mh_iteration(m64, f64(x), f64(y), f64(v), zo, it, 0.3, seed, bs, row_base=1000)
        n_acc += int(a.sum())
        u = OB.R.uniforms(np.arange(1000, 1000 + n), it, OB.R.TAG_ACC, seed)
        fragile |= np.abs(u - np.exp(np.minimum(lpp - lpc, 0))) < 1e-3
    got = state.cpu().numpy()
    ok = ~fragile
    assert ok.sum() >= 0.98 * n
    assert np.abs(got[ok] - zo[ok]).max() < 1e-5
    assert abs(int(acc[0]) - n_acc) <= int(fragile.sum())
    assert int(accb.sum()) == int(acc[0])
    eng.close()


@pytest.mark.parametrize("binary,z_dims", [(False, (1, 1, 1, 7)), (True, (1, 1, 1, 7)), (False, (3, 6, 3, 6))])
def test_mh_effects_match_oracle(binary, z_dims):
    m = _model(binary, p=50, z_dims=z_dims)
    n, bs, burn, keep = 300, 128, 2, 3
    z, x, y, v = _panel(m, n)
    eng = _x3(m)
    dev = eng.device
    T = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    seed = (2 << 32) | 555
    q = z.shape[1]
    state = torch.zeros(n, q, device=dev)
    draws = torch.zeros(keep, n, q, device=dev)
    xs = np.linspace(0.0, 3.0, 21).astype(np.float32)
    adrf = torch.zeros(len(xs), keep, device=dev, dtype=torch.float64)
    ite = torch.zeros(n, keep, device=dev)
    eng.mh_run(T(x[:, 0]), T(y[:, 0]), T(v), state, bs, 0, burn + keep, burn, 0.4, seed, init=True, row_base=40, block0=1,
               draws=draws, n_keep=keep, effect=2 if binary else 1, sample_y=True, x_values=None if binary else T(xs),
               adrf_sum=None if binary else adrf, ite=ite if binary else None)
    dr = draws.cpu().numpy()
    alone = eng.effects(draws, bs, seed, it0=burn, x_values=None if binary else xs, sample_y=True, row_base=40, block0=1).cpu().numpy()
    fused = ite.t().cpu().numpy() if binary else (adrf / n).float().cpu().numpy()
    assert np.abs(alone - fused).max() < 1e-5
    m64 = OB.cast_model(m, np.float64)
    for d in range(keep):
        ref = OB.effects_draw(m64, f64(dr[d]), [1.0, 0.0] if binary else f64(xs), d, burn + d, True, seed, bs, block0=1, row_base=40)
        if binary:
            assert np.abs(ite[:, d].cpu().numpy() - (ref[0] - ref[1])).max() < 1e-3
        else:
            assert np.abs(adrf[:, d].cpu().numpy() / n - ref.mean(axis=1)).max() < 2e-4
    alone0 = eng.effects(draws, bs, seed, it0=burn, x_values=None if binary else xs, sample_y=False, row_base=40, block0=1).cpu().numpy()
    ref0 = OB.effects_draw(m64, f64(dr[1]), [1.0, 0.0] if binary else f64(xs), 1, burn + 1, False, seed, bs, block0=1, row_base=40)
    if binary:
        assert np.abs(alone0[1] - (ref0[0] - ref0[1])).max() < 1e-3
    else:
        assert np.abs(alone0[:, 1] - ref0.mean(axis=1)).max() < 2e-4
    eng.close()


def test_precision_modes_outside_the_default_shape_kernels_are_refused():
    m = _model(False, p=50, g_units=[32, 32])
    eng = _engine(m, g_units=[32, 32])
    with pytest.raises(RuntimeError, match="split precision exists on the default-shape"):
        eng.set_precision("f16x3")
    eng.close()
    m = _model(False, p=50)
    eng = _engine(m)
    with pytest.raises(RuntimeError, match="no bf16 form"):
        eng.set_precision("bf16x3")
    eng.close()


def test_class_predict_in_split_precision(tmp_path):
    """CausalBGM(use_bnn=True, mh_precision='f16x3'): same weights, seeds and draws as the fp32 model; the two predicts differ by the
    chains whose accept decision sat within the arithmetic's distance of the threshold (a few per thousand), the dose-response by less
    than the posterior spread."""
    from bayesgm_amd.models import CausalBGM
    from bayesgm_amd.datasets import Sim_Hirano_Imbens_sampler
    from test_gpu_bnn import _params
    x, y, v = Sim_Hirano_Imbens_sampler(N=2000, v_dim=20, seed=0).load_all()
    xs = np.linspace(0, 3, 7)
    out = {}
    for mode in ("fp32", "f16x3"):
        model = CausalBGM(dict(_params(tmp_path, False), mh_precision=mode, save_res=False), random_seed=3)
        eff, interval = model.predict((x, y, v), alpha=0.05, n_mcmc=30, burn_in=30, x_values=xs, q_sd=0.5, bs=500, verbose=0)
        lp = model.get_log_posterior(x[:200], y[:200], v[:200], np.zeros((200, 10), np.float32))
        out[mode] = (eff, interval, model.last_acceptance_rate, lp)
        assert np.isfinite(eff).all() and np.all(interval[:, 0] <= interval[:, 1])
    assert np.abs(out["fp32"][3] - out["f16x3"][3]).max() < 2e-5 * np.abs(out["fp32"][3]).max() + 2e-3
    assert np.abs(out["fp32"][3] - out["f16x3"][3]).max() > 0.0
    assert abs(out["fp32"][2] - out["f16x3"][2]) < 5e-3
    assert np.abs(out["fp32"][0] - out["f16x3"][0]).max() < 0.05
    with pytest.raises(ValueError, match="mh_precision"):
        CausalBGM(dict(_params(tmp_path, False), mh_precision="bf16x3", save_res=False), random_seed=3)
