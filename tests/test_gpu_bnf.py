"""Fixed-normalisation Bayesian-network sampling kernels (csrc/bnf_kernels.h: persistent workgroups, LDS-resident loc, perturbations
streamed from L2, pre-generated sign words) against oracle/bnn.py through the C ABI.  params['bnn_norm'] = "fixed" is the shipped
default; the batch-statistics kernels are covered by test_gpu_bnn.py."""
import numpy as np
import pytest
import torch

from oracle import bnn as OB

pytestmark = pytest.mark.gpu


def _model(binary, z_dims=(1, 1, 1, 7), p=50, seed=0, **units):
    m = OB.init_model(seed, list(z_dims), p, binary, **units)
    rs = np.random.RandomState(seed + 11)
    for k in ("g", "e", "f", "h"):
        m[k]["norm"] = "fixed"
        m[k]["gamma"] = (1.0 + 0.2 * rs.standard_normal(m[k]["gamma"].shape)).astype(np.float32)
        m[k]["beta"] = (0.1 * rs.standard_normal(m[k]["beta"].shape)).astype(np.float32)
    return m


def _panel(m, n, seed=1):
    rs = np.random.RandomState(seed)
    q = sum(m["z_dims"])
    z = rs.standard_normal((n, q)).astype(np.float32)
    v = rs.standard_normal((n, m["v_dim"])).astype(np.float32)
    x = ((rs.rand(n, 1) > 0.5) if m["binary_treatment"] else rs.exponential(size=(n, 1))).astype(np.float32)
    y = (x + rs.standard_normal((n, 1))).astype(np.float32)
    return z, x, y, v


def _engine(m, **units):
    from bayesgm_amd.bnn_engine import BnnEngine
    eng = BnnEngine(m["v_dim"], m["z_dims"], m["binary_treatment"], kl_weight=1e-4, max_batch=32, norm_mode=1, **units)
    eng.begin(m)
    return eng


f64 = lambda a: a.astype(np.float64)


@pytest.mark.parametrize("binary,p,z_dims,n,bs", [
    (False, 200, (1, 1, 1, 7), 700, 300),      # the bench shape: 13 output tiles, ragged last block, row groups that end mid-block
    (True, 100, (3, 3, 6, 6), 520, 520),       # q = 18: two input k-tiles (KS = 5)
    (False, 37, (2, 1, 2, 3), 100, 64),        # p % 4 != 0: scalar data-row loads; variance column in the middle of a tile
    (False, 191, (1, 1, 1, 7), 90, 33),        # p + 1 = 192: the variance column is the last of an even tile count
    (True, 50, (5, 5, 5, 5), 130, 40),         # q = 20 (KS = 6)
    (False, 50, (4, 4, 4, 4), 75, 75),         # q = 16: the treatment sits alone in the second k-tile
])
def test_logpost_blocks_match_oracle(binary, p, z_dims, n, bs):
    m = _model(binary, z_dims=z_dims, p=p)
    z, x, y, v = _panel(m, n)
    eng = _engine(m)
    T = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(eng.device)
    seed, stream = (3 << 32) | 1234, 77
    got = eng.logpost(T(x[:, 0]), T(y[:, 0]), T(v), T(z), bs, seed, stream, block0=2).cpu().numpy()
    ref = OB.log_posterior_blocks(OB.cast_model(m, np.float64), f64(x), f64(y), f64(v), f64(z), bs, seed, stream, block0=2)
    assert np.abs(got - ref).max() < 2e-5 * np.abs(ref).max() + 2e-3, np.abs(got - ref).max()
    eng.close()


def test_logpost_follows_parameter_updates():
    """The packed blob is rebuilt when theta changes (bgm_bnn_write)."""
    m = _model(False, p=50)
    z, x, y, v = _panel(m, 64)
    eng = _engine(m)
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
    eng = _engine(m)
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
        fragile |= np.abs(u - np.exp(np.minimum(lpp - lpc, 0))) < 1e-3     # accept decisions within fp32 noise
    got = state.cpu().numpy()
    ok = ~fragile
    assert ok.sum() >= 0.98 * n          # rows whose accept decision lies within fp32 noise of u: measured 1 - 4 of 600 (expected 2 x 2e-3 x n)
    assert np.abs(got[ok] - zo[ok]).max() < 1e-5
    assert abs(int(acc[0]) - n_acc) <= int(fragile.sum())
    assert int(accb.sum()) == int(acc[0])
    eng.close()


@pytest.mark.parametrize("binary,z_dims", [(False, (1, 1, 1, 7)), (True, (1, 1, 1, 7)), (False, (3, 6, 3, 6))])
def test_mh_effects_match_oracle(binary, z_dims):
    """Kept draws + causal effects of the sampler against the oracle evaluated on the kernel's own draws; 21 doses exercise the
    second round of the lane-group-distributed outcome noise."""
    m = _model(binary, p=50, z_dims=z_dims)
    n, bs, burn, keep = 300, 128, 2, 3
    z, x, y, v = _panel(m, n)
    eng = _engine(m)
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
    init = OB.R.normals(np.arange(40, 40 + n), 0, q, OB.R.TAG_INIT, seed)
    assert np.abs(dr[-1] - state.cpu().numpy()).max() == 0.0
    assert np.abs(dr[0] - init).max() < 10.0 and np.abs(dr[0] - dr[-1]).max() > 0.0
    m64 = OB.cast_model(m, np.float64)
    for d in range(keep):
        ref = OB.effects_draw(m64, f64(dr[d]), [1.0, 0.0] if binary else f64(xs), d, burn + d, True, seed, bs, block0=1, row_base=40)
        if binary:
            assert np.abs(ite[:, d].cpu().numpy() - (ref[0] - ref[1])).max() < 1e-3
        else:
            assert np.abs(adrf[:, d].cpu().numpy() / n - ref.mean(axis=1)).max() < 2e-4
    # without outcome noise
    alone0 = eng.effects(draws, bs, seed, it0=burn, x_values=None if binary else xs, sample_y=False, row_base=40, block0=1).cpu().numpy()
    ref0 = OB.effects_draw(m64, f64(dr[1]), [1.0, 0.0] if binary else f64(xs), 1, burn + 1, False, seed, bs, block0=1, row_base=40)
    if binary:
        assert np.abs(alone0[1] - (ref0[0] - ref0[1])).max() < 1e-3
    else:
        assert np.abs(alone0[:, 1] - ref0.mean(axis=1)).max() < 2e-4
    eng.close()


def test_full_size_panel_blocks_are_independent_sampler_runs():
    """BASELINE-size panel (N = 10^6, p = 200, bs = 10^4): block b of the lock-step sampler equals a stand-alone run on that block's
    rows with the same block id and row offset, bit for bit (rows are independent; nothing is accumulated across rows), for a
    middle block and the ragged last one; the log-posterior of sampled rows matches the oracle."""
    m = _model(False, p=200)
    n, bs, p = 1000000 - 3700, 10000, 200
    eng = _engine(m)
    dev = eng.device
    g = torch.Generator(device=dev); g.manual_seed(5)
    v = torch.randn(n, p, device=dev, generator=g)
    x = torch.rand(n, device=dev, generator=g)
    y = torch.randn(n, device=dev, generator=g)
    seed = (11 << 32) | 2024
    state = torch.empty(n, 10, device=dev)
    acc = torch.zeros(1, device=dev, dtype=torch.int32)
    eng.mh_run(x, y, v, state, bs, 0, 3, 0, 0.7, seed, init=True, acc_count=acc)
    assert torch.isfinite(state).all() and 0 < int(acc[0]) < 3 * n
    for blk in (37, n // bs):                      # the last block has 6300 rows
        lo, hi = blk * bs, min(n, (blk + 1) * bs)
        sub = torch.empty(hi - lo, 10, device=dev)
        eng.mh_run(x[lo:hi].contiguous(), y[lo:hi].contiguous(), v[lo:hi].contiguous(), sub, bs, 0, 3, 0, 0.7, seed, init=True,
                   row_base=lo, block0=blk)
        assert torch.equal(sub, state[lo:hi])
    lp = eng.logpost(x, y, v, state, bs, seed, 9).cpu().numpy()
    blk = 61
    rows = np.arange(blk * bs, blk * bs + 48)
    ref = OB.log_posterior_blocks(OB.cast_model(m, np.float64), f64(x[rows].cpu().numpy()[:, None]), f64(y[rows].cpu().numpy()[:, None]),
                                  f64(v[rows].cpu().numpy()), f64(state[rows].cpu().numpy()), bs, seed, 9, block0=blk)
    assert np.abs(lp[rows] - ref).max() < 2e-5 * np.abs(ref).max() + 2e-3
    eng.close()
