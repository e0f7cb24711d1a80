"""GPU parity tests of the CausalBGM posterior-sampling path: HIP kernels (through the
C ABI) vs the NumPy oracle on identical seeded inputs.

Tolerances (fp32 path, exact-fp32 MFMA; oracle evaluated in float64):
  log-posterior      |hip - oracle64| <= 2e-6 * |oracle64| + 2e-4   (values are O(1e2..1e3))
  chain states       identical Philox stream => identical decisions except where a
                     uniform lands within rounding of the acceptance ratio; >= 99 % of
                     rows must match the oracle chain to 1e-4 after the whole run
  effects (ADRF/ITE) computed from the SAME draws: <= 2e-4 abs
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from oracle import causal as OC  # noqa: E402


def _engine(m, **kw):
    import torch  # noqa: F401
    from bayesgm_amd.engine import CausalEngine
    eng = CausalEngine(m["v_dim"], m["z_dims"], binary_treatment=m["binary_treatment"],
                       sigma_v=m.get("sigma_v"), sigma_x=m.get("sigma_x"), sigma_y=m.get("sigma_y"), **kw)
    eng.set_model(g=m["g"], f=m["f"], h=m["h"], e=m["e"])
    return eng


def _data(n, p, seed, binary=False):
    rs = np.random.RandomState(seed)
    v = rs.randn(n, p).astype(np.float32)
    x = rs.exponential(size=(n, 1)).astype(np.float32)
    if binary:
        x = (x > np.median(x)).astype(np.float32)
    y = (x + rs.randn(n, 1)).astype(np.float32)
    return x, y, v


def _model(seed, z_dims, p, binary=False, scale=1.0, **kw):
    m = OC.init_model(seed, z_dims, p, binary_treatment=binary, **kw)
    # non-zero biases so the bias path is exercised (Keras initialises them to 0)
    rs = np.random.RandomState(seed + 99)
    for k in ("g", "f", "h", "e"):
        m[k] = [((W * scale).astype(np.float32), (0.1 * rs.randn(*b.shape)).astype(np.float32)) for W, b in m[k]]
    return m


def _as64(m, *arrs):
    return OC.cast_model(m, np.float64), [a.astype(np.float64) for a in arrs]


CASES = [
    dict(z_dims=[1, 1, 1, 7], p=200, binary=False, n=1000),   # configs/Sim_Hirano_Imbens.yaml shape
    dict(z_dims=[3, 3, 6, 6], p=100, binary=True, n=777),     # cli defaults, binary treatment
    dict(z_dims=[1, 1, 1, 7], p=20, binary=False, n=16),      # exactly one tile
    dict(z_dims=[1, 1, 1, 7], p=20, binary=False, n=1),       # single row
    dict(z_dims=[3, 3, 6, 6], p=25, binary=False, n=33),      # ragged tail, p not multiple of 4
    # shapes between the compiled ones run on the next larger compiled shape (zero-padded tiles / K rows)
    dict(z_dims=[1, 1, 1, 7], p=50, binary=False, n=70),      # 4 output tiles -> the 7-tile shape
    dict(z_dims=[3, 3, 6, 6], p=150, binary=True, n=41),      # q+1 = 19, 10 output tiles
    dict(z_dims=[2, 2, 2, 6], p=120, binary=False, n=35),     # q+1 = 13 -> the two-K-tile first layer; 8 tiles -> 10
    dict(z_dims=[1, 1, 1, 2], p=5, binary=False, n=20),       # one output tile -> the 2-tile shape
    dict(z_dims=[1, 1, 1, 7], p=207, binary=False, n=18),     # the largest v_dim (208 outputs)
]


@pytest.mark.parametrize("case", CASES)
def test_logpost_matches_oracle(case):
    m = _model(1, case["z_dims"], case["p"], case["binary"])
    x, y, v = _data(case["n"], case["p"], 2, case["binary"])
    z = np.random.RandomState(3).randn(case["n"], sum(case["z_dims"])).astype(np.float32)
    eng = _engine(m)
    got = eng.logpost(x.ravel(), y.ravel(), v, z).cpu().numpy()
    m64, (x64, y64, v64, z64) = _as64(m, x, y, v, z)
    ref = OC.log_posterior(m64, x64, y64, v64, z64)
    err = np.abs(got - ref)
    assert np.all(err <= 2e-6 * np.abs(ref) + 2e-4), (err.max(), np.abs(ref).max())


def test_logpost_fixed_sigmas_and_large_weights():
    m = _model(5, [1, 1, 1, 7], 200, False, scale=1.7, sigma_v=0.8, sigma_x=1.3, sigma_y=0.5)
    x, y, v = _data(300, 200, 6)
    z = np.random.RandomState(7).randn(300, 10).astype(np.float32)
    got = _engine(m).logpost(x.ravel(), y.ravel(), v, z).cpu().numpy()
    m64, (x64, y64, v64, z64) = _as64(m, x, y, v, z)
    ref = OC.log_posterior(m64, x64, y64, v64, z64)
    assert np.all(np.abs(got - ref) <= 2e-6 * np.abs(ref) + 5e-4)


def test_logpost_rows_are_independent():
    """Size-independent property: the value of a row does not depend on the panel it sits in."""
    m = _model(8, [1, 1, 1, 7], 200)
    x, y, v = _data(5000, 200, 9)
    z = np.random.RandomState(10).randn(5000, 10).astype(np.float32)
    eng = _engine(m)
    full = eng.logpost(x.ravel(), y.ravel(), v, z).cpu().numpy()
    part = eng.logpost(x[1234:1300].ravel(), y[1234:1300].ravel(), v[1234:1300], z[1234:1300]).cpu().numpy()
    assert np.array_equal(full[1234:1300], part)


def test_encoder_matches_oracle():
    from oracle.nets import mlp_forward
    for z_dims, p, n in (([1, 1, 1, 7], 200, 500), ([3, 3, 6, 6], 100, 100), ([1, 1, 1, 7], 20, 17),
                         ([3, 3, 6, 6], 150, 33), ([1, 1, 1, 7], 50, 40), ([1, 1, 1, 2], 5, 9)):
        m = _model(11, z_dims, p)
        _, _, v = _data(n, p, 12)
        got = _engine(m).encode(v).cpu().numpy()
        ref = mlp_forward(OC.cast_model(m, np.float64)["e"], v.astype(np.float64))
        assert got.shape == ref.shape
        assert np.abs(got - ref).max() <= 1e-5 * max(1.0, np.abs(ref).max())


@pytest.mark.parametrize("case", [dict(z_dims=[1, 1, 1, 7], p=200, binary=False, n=200),
                                  dict(z_dims=[3, 3, 6, 6], p=100, binary=True, n=150),
                                  dict(z_dims=[1, 1, 1, 7], p=20, binary=False, n=40),
                                  dict(z_dims=[1, 1, 1, 7], p=50, binary=False, n=60),
                                  dict(z_dims=[2, 2, 2, 6], p=150, binary=True, n=50)])
def test_mh_chain_matches_oracle_chain(case):
    import torch
    from bayesgm_amd import _lib
    burn, keep, q_sd, seed = 25, 35, 0.3, 1234567890123
    m = _model(21, case["z_dims"], case["p"], case["binary"])
    x, y, v = _data(case["n"], case["p"], 22, case["binary"])
    eng = _engine(m)
    out = eng.mh_sample(x, y, v, burn, keep, q_sd, seed, want_draws=True, chunk=17)  # odd chunking on purpose
    draws = out["draws"].cpu().numpy()
    acc = out["acc_count"].cpu().numpy()
    ref, ref_acc, _ = OC.mh_sampler(m, (x, y, v), burn, keep, q_sd, seed, return_acc=True)
    assert draws.shape == ref.shape == (keep, case["n"], sum(case["z_dims"]))
    # initial state + first transition must agree everywhere (no decision yet compounded)
    row_ok = np.all(np.abs(draws[-1] - ref[-1]) <= 1e-4, axis=1)
    assert row_ok.mean() >= 0.99, row_ok.mean()
    assert np.abs(acc.astype(np.int64) - ref_acc).max() <= max(2, case["n"] // 50)
    assert 0.02 < acc.sum() / (acc.size * case["n"]) < 0.98
    # cached log-posterior of the final state equals a fresh evaluation (reference recomputes it, base.py:866)
    lp = eng.logpost(x.ravel(), y.ravel(), v, out["state"]).cpu().numpy()
    assert np.abs(lp - out["logp"].cpu().numpy()).max() <= 1e-3
    assert np.array_equal(out["state"].cpu().numpy(), draws[-1])
    # determinism + chunking invariance: one launch, same seed -> identical bits
    out2 = eng.mh_sample(x, y, v, burn, keep, q_sd, seed, want_draws=True)
    assert torch.equal(out2["draws"], out["draws"])


def test_mh_row_base_offsets_rng_stream():
    """Rows [s:e] run with row_base=s reproduce the same rows of the full run (predict's bs-blocking)."""
    import torch
    m = _model(31, [1, 1, 1, 7], 20)
    x, y, v = _data(96, 20, 32)
    eng = _engine(m)
    full = eng.mh_sample(x, y, v, 10, 5, 0.5, 77, want_draws=True)["draws"]
    part = eng.mh_sample(x[32:80], y[32:80], v[32:80], 10, 5, 0.5, 77, want_draws=True, row_base=32)["draws"]
    assert torch.equal(full[:, 32:80], part)


@pytest.mark.parametrize("n_doses", [1, 3, 4, 7, 16, 17, 20, 33])
def test_adrf_effects_match_oracle_on_same_draws(n_doses):
    """Dose counts cover the kernel's pass plan: Philox calls in groups of four (each lane group evaluates its own call: 16, 20,
    33 doses), a remainder of one to three shared calls (1 ... 7, 17, 20, 33) and a partial last call (1, 3, 7, 17, 33)."""
    from bayesgm_amd import _lib
    burn, keep, seed = 10, 12, 99
    m = _model(41, [1, 1, 1, 7], 200)
    x, y, v = _data(333, 200, 42)
    xs = np.linspace(0, 3, n_doses)
    eng = _engine(m)
    for sample_y in (True, False):
        out = eng.mh_sample(x, y, v, burn, keep, 0.4, seed, want_draws=True, effect=_lib.EFFECT_ADRF,
                            x_values=xs, sample_y=sample_y)
        draws = out["draws"].cpu().numpy()
        ref = OC.infer_from_latent_posterior(OC.cast_model(m, np.float64), draws.astype(np.float64), xs, sample_y,
                                             seed, burn_in=burn)
        got = out["adrf"].cpu().numpy()
        assert got.shape == (n_doses, keep)
        assert np.abs(got - ref).max() <= 2e-4, np.abs(got - ref).max()


def test_ite_effects_and_quantiles_match_oracle_on_same_draws():
    from bayesgm_amd import _lib
    burn, keep, seed = 8, 40, 5
    m = _model(51, [3, 3, 6, 6], 100, binary=True)
    x, y, v = _data(130, 100, 52, binary=True)
    eng = _engine(m)
    out = eng.mh_sample(x, y, v, burn, keep, 0.4, seed, want_draws=True, effect=_lib.EFFECT_ITE, sample_y=True)
    draws = out["draws"].cpu().numpy()
    ref = OC.infer_from_latent_posterior(OC.cast_model(m, np.float64), draws.astype(np.float64), None, True, seed,
                                         burn_in=burn)  # [keep, n]
    ite = out["ite"].cpu().numpy()  # [n, keep]
    assert np.abs(ite.T - ref).max() <= 2e-4
    mean, lo, hi = eng.row_mean_quantiles(out["ite"], 0.005, 0.995)
    assert np.allclose(mean.cpu().numpy(), ite.mean(axis=1), atol=1e-6)
    assert np.allclose(lo.cpu().numpy(), np.quantile(ite, 0.005, axis=1), atol=1e-6)
    assert np.allclose(hi.cpu().numpy(), np.quantile(ite, 0.995, axis=1), atol=1e-6)


def test_row_quantiles_edge_cases():
    import torch
    from bayesgm_amd.engine import CausalEngine
    eng = CausalEngine(20, [1, 1, 1, 7])
    rs = np.random.RandomState(0)
    # up to 32768 values per row: LDS sort; beyond: radix selection of the order statistics (negative values, ties, constants)
    for m_, ql, qh in ((1, 0.1, 0.9), (2, 0.25, 0.75), (3000, 0.005, 0.995), (4096, 0.0, 1.0), (777, 0.5, 0.5), (32768, 0.005, 0.995),
                       (32769, 0.005, 0.995), (70001, 0.025, 0.975), (100000, 0.0, 1.0), (65536, 0.5, 0.5)):
        a = rs.randn(9, m_).astype(np.float32)
        a[0] = 3.0  # constant row
        if m_ > 10:
            a[1, ::3] = np.round(a[1, ::3], 1)      # ties
        t = torch.from_numpy(a).cuda()
        mean, lo, hi = eng.row_mean_quantiles(t, ql, qh)
        assert np.allclose(mean.cpu().numpy(), a.mean(axis=1), atol=2e-6)
        assert np.allclose(lo.cpu().numpy(), np.quantile(a, ql, axis=1), atol=1e-6)
        assert np.allclose(hi.cpu().numpy(), np.quantile(a, qh, axis=1), atol=1e-6)


def test_mh_adaptive_q_sd_follows_reference_schedule():
    """Adaptive proposal scale (q_sd <= 0 / None): base.py:880-892; compare q_sd trajectory end point."""
    m = _model(61, [1, 1, 1, 7], 20)
    x, y, v = _data(64, 20, 62)
    eng = _engine(m)
    out = eng.mh_sample(x, y, v, 260, 10, None, 3, adaptive=True)
    _, _, q_ref = OC.mh_sampler(m, (x, y, v), 260, 10, None, 3, adaptive=True, return_acc=True)
    assert abs(out["q_sd"] - q_ref) <= 1e-12 or abs(np.log(out["q_sd"] / q_ref)) <= np.log(1.1) * 1.01


def test_errors_are_loud():
    from bayesgm_amd.engine import CausalEngine
    with pytest.raises(RuntimeError):
        CausalEngine(200, [1, 1, 1, 7], g_units=[64, 0])            # not a width (any positive width is accepted: tests/test_gpu_widths.py)
    eng = CausalEngine(200, [1, 1, 1, 7])
    with pytest.raises(RuntimeError):                               # weights not set
        eng.logpost(np.zeros(4, np.float32), np.zeros(4, np.float32), np.zeros((4, 200), np.float32),
                    np.zeros((4, 10), np.float32))


def test_full_size_panel_properties():
    """BASELINE.json's headline shape (N = 1e6 rows, p = 200, z_dims [1,1,1,7]) through size-independent properties:
    (i) determinism -- two runs with the same seed give bit-identical states, cached log-posteriors and ADRF draws;
    (ii) blocking / sharding invariance -- rows [s:e) run on their own with row_base = s reproduce the same rows of
         the full run bit for bit (the Philox stream is keyed by the global row, not by the launch geometry);
    (iii) a random sample of rows agrees with the float64 oracle chain run on those rows alone;
    (iv) the ADRF of a small block (assembled from per-wave-slot partial sums) equals the oracle's mean of per-row
         effects computed from the same draws."""
    import torch
    from bayesgm_amd import _lib
    n, p, burn, keep = 1_000_000, 200, 8, 4
    m = _model(71, [1, 1, 1, 7], p)
    rs = np.random.RandomState(72)
    v = rs.standard_normal((n, p)).astype(np.float32)
    x = rs.exponential(size=(n, 1)).astype(np.float32)
    y = (x + rs.standard_normal((n, 1))).astype(np.float32)
    eng = _engine(m)
    xd, yd, vd = (torch.from_numpy(a).cuda() for a in (x.reshape(-1), y.reshape(-1), v))
    doses = np.linspace(0, 3, 5).astype(np.float32)

    def run(lo, hi, want_draws=False):
        return eng.mh_sample(xd[lo:hi], yd[lo:hi], vd[lo:hi], burn, keep, 1.0, 5, want_draws=want_draws, row_base=lo,
                             effect=_lib.EFFECT_ADRF, x_values=doses, sample_y=True)
    a = run(0, n)
    b = run(0, n)
    assert torch.equal(a["state"], b["state"]) and torch.equal(a["logp"], b["logp"]) and torch.equal(a["adrf"], b["adrf"])
    lo, hi = 345_600, 345_600 + 70_001                                                                  # ragged block
    part = run(lo, hi)
    assert torch.equal(a["state"][lo:hi], part["state"]) and torch.equal(a["logp"][lo:hi], part["logp"])
    acc_rate = float(a["acc_count"].sum().item()) / (n * (burn + keep))
    assert 0.0 < acc_rate < 1.0
    # (iii) oracle on a sample of rows (each row is an independent chain keyed by its global index)
    idx = np.sort(rs.choice(n, 40, replace=False))
    got = a["state"].cpu().numpy()[idx]
    ref = np.stack([OC.mh_sampler(m, (x[i:i + 1], y[i:i + 1], v[i:i + 1]), burn, keep, 1.0, 5, row0=int(i))[-1, 0] for i in idx])
    # measured: 40 of 40 chains identical to 1e-4; one chain may flip an accept decision that lies within fp32 rounding of its uniform
    assert np.all(np.abs(got - ref) <= 1e-4, axis=1).sum() >= len(idx) - 1
    # (iv) ADRF of a 64-row block vs the oracle's effects on the same draws
    small = run(lo, lo + 64, want_draws=True)
    m64 = OC.cast_model(m, np.float64)
    ref_adrf = OC.infer_from_latent_posterior(m64, small["draws"].cpu().numpy().astype(np.float64), x_values=doses,
                                              sample_y=True, seed=5, row0=lo, burn_in=burn)
    assert np.abs(small["adrf"].cpu().numpy() - ref_adrf).max() <= 5e-4


def test_binary_treatment_full_size_panel_properties():
    """BASELINE.json configs[1] at full size (binary treatment, N = 1e5 rows, p = 100, z_dims [3,3,6,6] = the reference CLI's
    defaults, src/main.py) through size-independent properties: determinism; sharding invariance of states, cached log-posteriors
    and per-row ITE draws for a ragged block; the float64 oracle chain on a sample of rows; the per-row ITE of a block and its
    row quantiles against the oracle's effects on the same draws."""
    import torch
    from bayesgm_amd import _lib
    n, p, burn, keep, seed = 100_000, 100, 8, 16, 7
    zd = [3, 3, 6, 6]
    m = _model(73, zd, p, binary=True)
    x, y, v = _data(n, p, 74, binary=True)
    eng = _engine(m)
    xd, yd, vd = (torch.from_numpy(a).cuda() for a in (x.reshape(-1), y.reshape(-1), v))

    def run(lo, hi, want_draws=False):
        return eng.mh_sample(xd[lo:hi], yd[lo:hi], vd[lo:hi], burn, keep, 0.5, seed, want_draws=want_draws, row_base=lo,
                             effect=_lib.EFFECT_ITE, sample_y=True)
    a = run(0, n)
    b = run(0, n)
    assert torch.equal(a["state"], b["state"]) and torch.equal(a["logp"], b["logp"]) and torch.equal(a["ite"], b["ite"])
    lo, hi = 34_560, 34_560 + 7_001
    part = run(lo, hi, want_draws=True)
    assert torch.equal(a["state"][lo:hi], part["state"]) and torch.equal(a["logp"][lo:hi], part["logp"])
    assert torch.equal(a["ite"][lo:hi], part["ite"])
    acc_rate = float(a["acc_count"].sum().item()) / (n * (burn + keep))
    assert 0.0 < acc_rate < 1.0
    idx = np.sort(np.random.RandomState(75).choice(n, 40, replace=False))
    got = a["state"].cpu().numpy()[idx]
    ref = np.stack([OC.mh_sampler(m, (x[i:i + 1], y[i:i + 1], v[i:i + 1]), burn, keep, 0.5, seed, row0=int(i))[-1, 0] for i in idx])
    # measured: 40 of 40 chains identical to 1e-4; one chain may flip an accept decision that lies within fp32 rounding of its uniform
    assert np.all(np.abs(got - ref) <= 1e-4, axis=1).sum() >= len(idx) - 1
    draws = part["draws"].cpu().numpy()[:, :96]
    ref_ite = OC.infer_from_latent_posterior(OC.cast_model(m, np.float64), draws.astype(np.float64), None, True, seed, row0=lo,
                                             burn_in=burn)                                                  # [keep, 96]
    ite = part["ite"].cpu().numpy()[:96]
    assert np.abs(ite.T - ref_ite).max() <= 2e-4
    mean, ql, qh = eng.row_mean_quantiles(a["ite"], 0.005, 0.995)
    full = a["ite"].cpu().numpy()
    assert np.allclose(mean.cpu().numpy(), full.mean(axis=1), atol=1e-6)
    assert np.allclose(ql.cpu().numpy(), np.quantile(full, 0.005, axis=1), atol=1e-6)
    assert np.allclose(qh.cpu().numpy(), np.quantile(full, 0.995, axis=1), atol=1e-6)


@pytest.mark.parametrize("binary", [False, True])
def test_standalone_effects_from_draws_match_oracle_and_the_fused_pass(binary):
    """infer_from_latent_posterior on a given draw tensor (bgm_causal_effects): equals the oracle on the same draws and
    the numbers the MH kernel's fused effect pass produced for those draws (same noise counters)."""
    from bayesgm_amd import _lib
    z_dims, p, n, burn, keep, seed = [1, 1, 1, 7], 50, 83, 6, 5, 21
    m = _model(41, z_dims, p, binary)
    x, y, v = _data(n, p, 42, binary)
    eng = _engine(m)
    doses = np.linspace(0, 3, 7).astype(np.float32)
    eff = _lib.EFFECT_ITE if binary else _lib.EFFECT_ADRF
    fused = eng.mh_sample(x, y, v, burn, keep, 1.0, seed, want_draws=True, effect=eff, x_values=None if binary else doses, sample_y=True)
    draws = fused["draws"]
    alone = eng.effects(x, draws, burn, seed, x_values=None if binary else doses, sample_y=True)
    ref = OC.infer_from_latent_posterior(OC.cast_model(m, np.float64), draws.cpu().numpy().astype(np.float64),
                                         x_values=None if binary else doses, sample_y=True, seed=seed, burn_in=burn)
    got = alone.cpu().numpy()
    assert got.shape == ref.shape and np.abs(got - ref).max() <= 2e-4
    fused_out = fused["ite"].t().cpu().numpy() if binary else fused["adrf"].cpu().numpy()
    assert np.abs(got - fused_out).max() <= 1e-5
    # the class method (reference signature)
    from bayesgm_amd.models import CausalBGM
    params = dict(dataset="t", output_dir="/tmp", save_res=False, save_model=False, binary_treatment=binary, use_bnn=False,
                  z_dims=z_dims, v_dim=p, lr_theta=1e-4, lr_z=1e-4, g_units=[64] * 5, f_units=[64, 32, 8], h_units=[64, 32, 8],
                  e_units=[64] * 5, dz_units=[64, 32, 8], kl_weight=1e-4, lr=2e-4, g_d_freq=5, use_z_rec=True)
    model = CausalBGM(params, random_seed=1)
    out = model.infer_from_latent_posterior(draws.cpu().numpy(), x_values=None if binary else doses, sample_y=False)
    assert out.shape == ref.shape and np.all(np.isfinite(out))
    if not binary:
        with pytest.raises(ValueError):
            model.infer_from_latent_posterior(draws.cpu().numpy())


@pytest.mark.parametrize("precision", ["fp32", "bf16x3", "f16x3"])
def test_outcome_cache_is_bit_identical(precision):
    """bgm_causal_set_outcome_cache: a retained iteration in which no chain of a wave moved takes the outcome net's (mean, sd) at every
    dose from the previous evaluation (csrc/causal_kernels.h causal_effects_cached).  Same Philox streams with the cache on and off: the
    ADRF draw sums, the chains and the acceptance counts are equal to the last bit, and the cache is actually used."""
    from bayesgm_amd import _lib
    m = _model(11, [1, 1, 1, 7], 200)
    x, y, v = _data(3000, 200, 12)          # 188 tiles, the last one ragged
    xs = np.linspace(0, 3, 20)
    eng = _engine(m)
    eng.set_precision(precision)
    outs = {}
    for on in ('wave', False):
        eng.set_outcome_cache(on)
        eng.outcome_cache_stats(reset=True)
        out = eng.mh_sample(x, y, v, 40, 60, 1.0, 5, want_draws=True, effect=_lib.EFFECT_ADRF, x_values=xs)
        outs[on] = (out["adrf"].cpu().numpy(), out["draws"].cpu().numpy(), out["acc_count"].cpu().numpy(), eng.outcome_cache_stats())
    eng.set_outcome_cache(True)
    (a1, d1, c1, s1), (a0, d0, c0, s0) = outs['wave'], outs[False]
    print("served from cache: %d of %d retained tile-iterations (off: %d of %d)" % (s1 + s0))
    assert s1[1] == 188 * 60 and s0 == (0, 188 * 60)
    assert s1[0] > 0.05 * s1[1]                              # q_sd = 1: most proposals are rejected
    assert np.array_equal(a1, a0) and np.array_equal(d1, d0) and np.array_equal(c1, c0)
    # 7 doses (two passes, the second partial) and sample_y=False
    for kw in (dict(x_values=np.linspace(0, 2, 7)), dict(x_values=xs, sample_y=False)):
        res = []
        for on in ('wave', False):
            eng.set_outcome_cache(on)
            res.append(eng.mh_sample(x, y, v, 10, 30, 1.0, 6, effect=_lib.EFFECT_ADRF, **kw)["adrf"].cpu().numpy())
        eng.set_outcome_cache(True)
        assert np.array_equal(res[0], res[1])


def test_outcome_cache_binary_treatment_is_bit_identical():
    """The same for the individual treatment effects of a binary treatment: per wave (causal_ite_cached: the two arms' (mean, sd) in
    registers) and per chain (the event form of the retained phase, below)."""
    from bayesgm_amd import _lib
    m = _model(13, [3, 3, 6, 6], 100, True)
    x, y, v = _data(2000, 100, 14, True)
    eng = _engine(m)
    res = {}
    for on in ("wave", True, False):
        eng.set_outcome_cache(on)
        eng.outcome_cache_stats(reset=True)
        out = eng.mh_sample(x, y, v, 30, 50, 1.0, 9, effect=_lib.EFFECT_ITE)
        res[on] = (out["ite"].cpu().numpy(), eng.outcome_cache_stats())
    eng.set_outcome_cache(True)
    print("served from cache: per wave %d of %d tile-iterations, per chain %d of %d chain-iterations" % (res["wave"][1] + res[True][1]))
    assert res["wave"][1][1] == 125 * 50 and res["wave"][1][0] > 0 and res[False][1][0] == 0
    assert res[True][1][1] == 2000 * 50 and res[True][1][0] > 0.8 * 2000 * 50
    assert np.array_equal(res["wave"][0], res[False][0]) and np.array_equal(res[True][0], res[False][0])


@pytest.mark.parametrize("case", [dict(z_dims=[3, 3, 6, 6], p=100, n=2500), dict(z_dims=[1, 1, 1, 7], p=200, n=1000),
                                  dict(z_dims=[3, 3, 6, 6], p=100, n=777, sample_y=False), dict(z_dims=[1, 1, 1, 7], p=20, n=40000),
                                  dict(z_dims=[3, 3, 6, 6], p=100, n=100000)])      # (the last one: BASELINE configs[1] at full size)
def test_event_form_for_binary_treatment_is_bit_identical(case):
    """Outcome cache mode 2 with a binary treatment (round 6; csrc/causal_event_kernels.h): transitions that append an event per accepted
    move, the outcome net at the two arms on dense 16-event tiles, one thread per chain writing y(1) - y(0) of every retained draw.
    Against the fused kernel with both arms evaluated at every retained draw (mode 0), same Philox streams: the ITE matrix, chains,
    acceptance counts are equal to the last bit -- in one segment, in several (a small event budget), and with the retained phase split
    over several calls; the ITE draws equal the float64 restatement of infer_from_latent_posterior on the same latent draws.
    reference: causalbgm/base.py:686-733, 860-899."""
    from bayesgm_amd import _lib
    m = _model(31, case["z_dims"], case["p"], True)
    x, y, v = _data(case["n"], case["p"], 32, True)
    kw = dict(effect=_lib.EFFECT_ITE, sample_y=case.get("sample_y", True), want_draws=True)
    eng = _engine(m)
    eng.set_outcome_cache(False)
    ref = eng.mh_sample(x, y, v, 25, 70, 1.0, 5, **kw)
    eng.set_outcome_cache(True)
    eng.outcome_cache_stats(reset=True)
    got = eng.mh_sample(x, y, v, 25, 70, 1.0, 5, **kw)
    served, total = eng.outcome_cache_stats()
    assert total == case["n"] * 70                                         # chain-iterations: the event form ran
    events = total - served
    assert events == int(got["acc_count"].cpu().numpy()[26:].sum()) + case["n"]
    print("event form, binary: %d of %d retained chain-iterations without an outcome-net evaluation" % (served, total))
    for k in ("ite", "draws", "acc_count", "state", "logp"):
        assert np.array_equal(got[k].cpu().numpy(), ref[k].cpu().numpy()), k
    if case["n"] <= 1000:
        oref = OC.infer_from_latent_posterior(OC.cast_model(m, np.float64), got["draws"].cpu().numpy().astype(np.float64), None,
                                              case.get("sample_y", True), 5, burn_in=25)
        assert np.abs(got["ite"].cpu().numpy() - np.asarray(oref).T).max() <= 5e-4
    n_slots = eng.mh_slots(case["n"])
    tiles = (case["n"] + 15) // 16
    per_iter = n_slots * ((tiles + n_slots - 1) // n_slots) * 16 * (4 * sum(case["z_dims"]) + 4 + 32)
    fixed = 2 * tiles * 64 * 2 * 4 + tiles * 8 + n_slots * 4
    eng.set_event_budget(fixed + 9 * per_iter + 100)
    seg = eng.mh_sample(x, y, v, 25, 70, 1.0, 5, **kw)
    chunked = eng.mh_sample(x, y, v, 25, 70, 1.0, 5, chunk=30, **kw)
    eng.set_event_budget(0)
    for other in (seg, chunked):
        for k in ("ite", "draws", "acc_count", "state"):
            assert np.array_equal(other[k].cpu().numpy(), ref[k].cpu().numpy()), k


@pytest.mark.parametrize("case", [dict(z_dims=[1, 1, 1, 7], p=200, n=3000, doses=20), dict(z_dims=[1, 1, 1, 7], p=200, n=1000, doses=7),
                                  dict(z_dims=[3, 3, 6, 6], p=100, n=2500, doses=32), dict(z_dims=[1, 1, 1, 7], p=20, n=40000, doses=5),
                                  dict(z_dims=[1, 1, 1, 7], p=200, n=777, doses=20, sample_y=False)])
def test_event_form_of_the_retained_phase_is_bit_identical(case):
    """Outcome cache mode 2 (csrc/causal_event_kernels.h): the retained iterations run as transitions that append an event per accepted
    move, the outcome net runs on dense 16-event tiles, a spread pass adds mean + sd * noise per (row, draw).  Against the fused kernel
    with every dose evaluated at every retained draw (mode 0), same Philox streams: chains, acceptance counts and the per-slot ADRF
    partial sums are equal to the last bit -- in one segment, in several (a small event budget), and with the retained phase split
    over several calls.  reference: causalbgm/base.py:671-763, 860-899."""
    from bayesgm_amd import _lib
    binary = False
    m = _model(21, case["z_dims"], case["p"], binary)
    x, y, v = _data(case["n"], case["p"], 22)
    xs = np.linspace(0, 3, case["doses"])
    kw = dict(effect=_lib.EFFECT_ADRF, x_values=xs, sample_y=case.get("sample_y", True), want_draws=True)
    eng = _engine(m)
    eng.set_outcome_cache(False)
    ref = eng.mh_sample(x, y, v, 25, 70, 1.0, 5, **kw)
    eng.set_outcome_cache(True)
    eng.outcome_cache_stats(reset=True)
    got = eng.mh_sample(x, y, v, 25, 70, 1.0, 5, **kw)
    served, total = eng.outcome_cache_stats()
    acc = float(got["acc_count"].cpu().numpy()[25:].sum()) / (case["n"] * 70)
    print("event form: %d of %d retained chain-iterations without an outcome-net evaluation (acceptance %.3f)" % (served, total, acc))
    assert total == case["n"] * 70                                         # the event form ran
    events = total - served                                                # one per accepted move + one per chain at the first retained iteration
    assert events == int(got["acc_count"].cpu().numpy()[26:].sum()) + case["n"]
    for k in ("adrf_partial", "draws", "acc_count", "state", "logp"):
        assert np.array_equal(got[k].cpu().numpy(), ref[k].cpu().numpy()), k
    assert np.array_equal(got["adrf"].cpu().numpy(), ref["adrf"].cpu().numpy())
    if case["n"] <= 1000:      # and against the float64 restatement of infer_from_latent_posterior on the same draws (oracle/causal.py)
        oref = OC.infer_from_latent_posterior(OC.cast_model(m, np.float64), got["draws"].cpu().numpy().astype(np.float64), xs,
                                              case.get("sample_y", True), 5, burn_in=25)
        assert np.abs(got["adrf"].cpu().numpy() - oref).max() <= 2e-4
    # several segments: a budget that holds ~9 retained iterations of this panel
    n_slots = eng.mh_slots(case["n"])
    tiles = (case["n"] + 15) // 16
    per_iter = n_slots * ((tiles + n_slots - 1) // n_slots) * 16 * (4 * sum(case["z_dims"]) + 4 + 32 * ((case["doses"] + 3) // 4))
    eng.set_event_budget(9 * per_iter + 100)
    seg = eng.mh_sample(x, y, v, 25, 70, 1.0, 5, **kw)
    # the retained phase over three calls (chunks of 30 iterations: 25 + 5, 30, 30, 10)
    chunked = eng.mh_sample(x, y, v, 25, 70, 1.0, 5, chunk=30, **kw)
    eng.set_event_budget(0)
    for other in (seg, chunked):
        for k in ("adrf_partial", "draws", "acc_count", "state"):
            assert np.array_equal(other[k].cpu().numpy(), ref[k].cpu().numpy()), k


def test_event_form_falls_back_where_it_does_not_exist():
    """more than 32 doses, split precision, a conditional prior: mode 2 runs the fused kernel with the per-wave cache; results unchanged"""
    from bayesgm_amd import _lib
    m = _model(23, [1, 1, 1, 7], 200)
    x, y, v = _data(1500, 200, 24)
    xs = np.linspace(0, 3, 33)
    eng = _engine(m)
    res = {}
    for on in (True, False):
        eng.set_outcome_cache(on)
        eng.outcome_cache_stats(reset=True)
        res[on] = (eng.mh_sample(x, y, v, 10, 30, 1.0, 6, effect=_lib.EFFECT_ADRF, x_values=xs)["adrf"].cpu().numpy(), eng.outcome_cache_stats())
    eng.set_outcome_cache(True)
    assert res[True][1][1] == 94 * 30 and res[True][1][0] > 0           # tile-iterations: the per-wave cache
    assert np.array_equal(res[True][0], res[False][0])
    with pytest.raises(ValueError):
        eng.set_outcome_cache("sometimes")


def test_event_form_honours_its_budget_and_falls_back_when_nothing_fits():
    """the segment length follows the budget down to ONE retained iteration; when not even that fits (or the allocation
    fails) the retained phase runs on the fused kernel with the per-wave cache instead of failing -- same sums to the last bit."""
    from bayesgm_amd import _lib
    m = _model(25, [1, 1, 1, 7], 200)
    x, y, v = _data(2000, 200, 26)
    xs = np.linspace(0, 3, 20)
    kw = dict(effect=_lib.EFFECT_ADRF, x_values=xs, want_draws=True)
    eng = _engine(m)
    eng.set_outcome_cache(False)
    ref = eng.mh_sample(x, y, v, 10, 24, 1.0, 8, **kw)
    eng.set_outcome_cache(True)
    n_slots = eng.mh_slots(2000)
    tiles = (2000 + 15) // 16
    per_iter = n_slots * ((tiles + n_slots - 1) // n_slots) * 16 * (4 * 10 + 4 + 32 * 5)
    fixed = 2 * tiles * 5 * 64 * 2 * 4 + tiles * 8 + n_slots * 4
    # (i) room for exactly one retained iteration per segment: 24 segments of the event form
    eng.set_event_budget(fixed + per_iter + 8)
    eng.outcome_cache_stats(reset=True)
    one = eng.mh_sample(x, y, v, 10, 24, 1.0, 8, **kw)
    served, total = eng.outcome_cache_stats()
    assert total == 2000 * 24                                   # chain-iterations: the event form ran
    # (ii) not even one iteration fits: per-wave cache on the fused kernel (tile-iterations in the statistics), no error
    eng.set_event_budget(1024)
    eng.outcome_cache_stats(reset=True)
    none = eng.mh_sample(x, y, v, 10, 24, 1.0, 8, **kw)
    served, total = eng.outcome_cache_stats()
    assert total == tiles * 24
    eng.set_event_budget(0)
    for other in (one, none):
        for k in ("adrf_partial", "draws", "acc_count", "state", "adrf"):
            assert np.array_equal(other[k].cpu().numpy(), ref[k].cpu().numpy()), k
    # the mode argument: bools, names and the C ABI's numbers (True is mode 2, the integer 1 is mode 1)
    assert [eng.outcome_cache_mode(a) for a in (False, "off", 0, "wave", 1, True, "chain", 2)] == [0, 0, 0, 1, 1, 2, 2, 2]


@pytest.mark.parametrize("precision", ["bf16x3", "f16x3"])
def test_event_form_with_split_precision_transitions(precision):
    """mh_precision = 'bf16x3' / 'f16x3' with outcome cache mode 2: the transitions run on the split-precision kernel and append events,
    the events' outcome-net tiles run in fp32 (causal_event_f_kernel on the fp32 sampling blob) -- chains, acceptance counts and states are
    bit-identical to the fused split-precision run; the ADRF differs from it only by the outcome net's split-precision error (measured
    <= 4e-6 on these panels; bound 2e-5) and is at least as close to the fp32 run."""
    from bayesgm_amd import _lib
    m = _model(31, [1, 1, 1, 7], 200)
    x, y, v = _data(2000, 200, 32)
    xs = np.linspace(0, 3, 20)
    kw = dict(effect=_lib.EFFECT_ADRF, x_values=xs, want_draws=True)
    eng = _engine(m)
    eng.set_outcome_cache(False)
    fp32 = eng.mh_sample(x, y, v, 20, 50, 1.0, 5, **kw)
    eng.set_precision(precision)
    fused = eng.mh_sample(x, y, v, 20, 50, 1.0, 5, **kw)
    eng.set_outcome_cache(True)
    eng.outcome_cache_stats(reset=True)
    ev = eng.mh_sample(x, y, v, 20, 50, 1.0, 5, **kw)
    served, total = eng.outcome_cache_stats()
    eng.set_precision("fp32")
    assert total == 2000 * 50 and served > 0.5 * total                   # the event form ran
    for k in ("draws", "acc_count", "state", "logp"):
        assert np.array_equal(ev[k].cpu().numpy(), fused[k].cpu().numpy()), k
    a_ev, a_fused, a_fp32 = (o["adrf"].cpu().numpy() for o in (ev, fused, fp32))
    print("ADRF max |event - fused| %.2e, |fused - fp32| %.2e, |event - fp32| %.2e" % (np.abs(a_ev - a_fused).max(), np.abs(a_fused - a_fp32).max(),
                                                                                 np.abs(a_ev - a_fp32).max()))
    assert np.abs(a_ev - a_fused).max() <= 2e-5
