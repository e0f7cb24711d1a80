"""General-shape sampling path of the deterministic CausalBGM (csrc/bnf_det_api.hip: the streamed-fragment kernels of bnf_kernels.h
without the Flipout half) against oracle/causal.py through the C ABI: the shapes no LDS-resident compiled kernel contains --
sum(z_dims) = 18 / 20 at the data widths of the reference's Semi_acic.yaml / Sim_Colangelo.yaml, sum(z_dims) up to 31, and data widths
beyond 207 where g's last layer is streamed from L2 ("wide", here p = 500 and p = 1001).  Same tolerances as test_gpu_causal.py.
BGM_FORCE_GENERAL=1 (set for a subprocess below) sends a shape the resident kernels DO hold through the same path."""
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from oracle import causal as OC  # noqa: E402

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _engine(m, **kw):
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


def _model(seed, z_dims, p, binary=False, **kw):
    m = OC.init_model(seed, z_dims, p, binary_treatment=binary, **kw)
    rs = np.random.RandomState(seed + 99)
    for k in ("g", "f", "h", "e"):
        m[k] = [(W.astype(np.float32), (0.1 * rs.randn(*b.shape)).astype(np.float32)) for W, b in m[k]]
    return m


def _as64(m, *arrs):
    return OC.cast_model(m, np.float64), [a.astype(np.float64) for a in arrs]


CASES = [
    dict(z_dims=[3, 6, 3, 6], p=177, binary=True, n=333),      # configs/Semi_acic.yaml: sum(z_dims) = 18 at 12 output tiles
    dict(z_dims=[5, 5, 5, 5], p=100, binary=False, n=130),     # configs/Sim_Colangelo.yaml: sum(z_dims) = 20
    dict(z_dims=[8, 8, 8, 7], p=50, binary=False, n=65),       # sum(z_dims) = 31
    dict(z_dims=[1, 1, 1, 7], p=500, binary=False, n=200),     # wide: g's last layer (128 KB) streamed from L2
    dict(z_dims=[3, 3, 6, 6], p=1001, binary=True, n=48),      # wide, p % 4 != 0, 63 output tiles
    dict(z_dims=[4, 4, 4, 4], p=207, binary=False, n=40),      # the treatment alone in the second k-tile, largest resident data width
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
    assert np.all(err <= 1e-5 * np.abs(ref) + 1e-3), (err.max(), np.abs(ref).max())


def test_logpost_fixed_sigmas():
    m = _model(5, [3, 6, 3, 6], 177, False, sigma_v=0.8, sigma_x=1.3, sigma_y=0.5)
    x, y, v = _data(100, 177, 6)
    z = np.random.RandomState(7).randn(100, 18).astype(np.float32)
    got = _engine(m).logpost(x.ravel(), y.ravel(), v, z).cpu().numpy()
    m64, (x64, y64, v64, z64) = _as64(m, x, y, v, z)
    ref = OC.log_posterior(m64, x64, y64, v64, z64)
    assert np.all(np.abs(got - ref) <= 1e-5 * np.abs(ref) + 1e-3)


@pytest.mark.parametrize("case", [dict(z_dims=[3, 6, 3, 6], p=177, binary=True, n=150),
                                  dict(z_dims=[5, 5, 5, 5], p=100, binary=False, n=90),
                                  dict(z_dims=[1, 1, 1, 7], p=500, binary=False, n=70)])
def test_mh_chain_and_effects_match_oracle(case):
    from bayesgm_amd import _lib
    burn, keep, q_sd, seed = 20, 15, 0.3, 1234567890123
    m = _model(21, case["z_dims"], case["p"], case["binary"])
    x, y, v = _data(case["n"], case["p"], 22, case["binary"])
    eng = _engine(m)
    xs = np.linspace(0, 3, 21)
    kw = dict(effect=_lib.EFFECT_ITE) if case["binary"] else dict(effect=_lib.EFFECT_ADRF, x_values=xs)
    out = eng.mh_sample(x, y, v, burn, keep, q_sd, seed, want_draws=True, chunk=11, sample_y=True, **kw)
    draws = out["draws"].cpu().numpy()
    acc = out["acc_count"].cpu().numpy()
    ref, ref_acc, _ = OC.mh_sampler(m, (x, y, v), burn, keep, q_sd, seed, return_acc=True)
    assert draws.shape == ref.shape
    row_ok = np.all(np.abs(draws[-1] - ref[-1]) <= 1e-4, axis=1)
    assert row_ok.mean() >= 0.97, row_ok.mean()
    assert np.abs(acc.astype(np.int64) - ref_acc).max() <= max(2, int((~row_ok).sum()))
    ref_eff = OC.infer_from_latent_posterior(OC.cast_model(m, np.float64), draws.astype(np.float64), None if case["binary"] else xs, True,
                                             seed, burn_in=burn)
    if case["binary"]:
        assert np.abs(out["ite"].cpu().numpy().T - ref_eff).max() <= 5e-4
    else:
        assert np.abs(out["adrf"].cpu().numpy() - ref_eff).max() <= 2e-4
    # the stand-alone form on the kept draws
    alone = eng.effects(x, out["draws"], burn, seed, x_values=None if case["binary"] else xs, sample_y=True)
    alone = alone.cpu().numpy()
    assert np.abs((alone if case["binary"] else alone) - (ref_eff if not case["binary"] else ref_eff)).max() <= 5e-4


@pytest.mark.parametrize("case", [dict(z_dims=[3, 6, 3, 6], p=177, binary=True, n=200), dict(z_dims=[1, 1, 1, 7], p=500, binary=False, n=120)])
def test_evaluate_matches_oracle(case):
    m = _model(31, case["z_dims"], case["p"], case["binary"])
    x, y, v = _data(case["n"], case["p"], 32, case["binary"])
    z = np.random.RandomState(33).randn(case["n"], sum(case["z_dims"])).astype(np.float32)
    import torch
    eng = _engine(m)
    xs = np.linspace(0.1, 2.9, 200)
    T = lambda a_: torch.from_numpy(np.ascontiguousarray(a_)).to(eng.device)
    sums, causal = eng.evaluate(T(x.ravel()), T(y.ravel()), T(v), T(z), x_values=None if case["binary"] else xs)
    sums = sums.cpu().numpy()
    n = case["n"]
    gv, gx, gy = sums[0] / (n * case["p"]), sums[1] / n, sums[2] / n
    causal = causal.cpu().numpy() if case["binary"] else causal.cpu().numpy() / n
    # reconstruction errors and plug-in effects from first principles (oracle network forward)
    from oracle.nets import mlp_forward
    m64 = OC.cast_model(m, np.float64)
    z64 = z.astype(np.float64)
    z0d, z1d, z2d, _ = case["z_dims"]
    g_out = mlp_forward(m64["g"], z64)
    mv = ((v - g_out[:, :case["p"]]) ** 2).mean()
    h_out = mlp_forward(m64["h"], np.concatenate([z64[:, :z0d], z64[:, z0d + z1d:z0d + z1d + z2d]], axis=1))[:, 0]
    xp = 1.0 / (1.0 + np.exp(-h_out)) if case["binary"] else h_out
    mx = ((x[:, 0] - xp) ** 2).mean()
    fy = lambda xv: mlp_forward(m64["f"], np.concatenate([z64[:, :z0d + z1d], xv], axis=1))[:, 0]
    my = ((y[:, 0] - fy(x.astype(np.float64))) ** 2).mean()
    assert abs(gv - mv) <= 1e-4 * mv and abs(gx - mx) <= 1e-4 * max(mx, 1e-3) and abs(gy - my) <= 1e-4 * my
    if case["binary"]:
        ref = fy(np.ones((case["n"], 1))) - fy(np.zeros((case["n"], 1)))
        assert np.abs(np.asarray(causal) - ref).max() <= 2e-4
    else:
        ref = np.array([fy(np.full((case["n"], 1), t)).mean() for t in xs])
        assert np.abs(np.asarray(causal) - ref).max() <= 2e-4


def test_forced_general_path_equals_resident_kernels():
    """The bench shape through both kernel families: chains, effects and log-posteriors agree to rounding."""
    code = r'''
import numpy as np, sys, json
sys.path.insert(0, %r)
from oracle import causal as OC
from bayesgm_amd.engine import CausalEngine
from bayesgm_amd import _lib
m = OC.init_model(3, [1, 1, 1, 7], 200)
rs = np.random.RandomState(4)
n = 400
v = rs.randn(n, 200).astype(np.float32); x = rs.exponential(size=(n, 1)).astype(np.float32); y = (x + rs.randn(n, 1)).astype(np.float32)
eng = CausalEngine(200, [1, 1, 1, 7]); eng.set_model(g=m["g"], f=m["f"], h=m["h"], e=m["e"])
z = rs.randn(n, 10).astype(np.float32)
lp = eng.logpost(x.ravel(), y.ravel(), v, z).cpu().numpy()
out = eng.mh_sample(x, y, v, 30, 10, 0.4, 77, want_draws=True, effect=_lib.EFFECT_ADRF, x_values=np.linspace(0, 3, 20))
np.savez(sys.argv[1], lp=lp, draws=out["draws"].cpu().numpy(), adrf=out["adrf"].cpu().numpy())
''' % ROOT
    import tempfile
    with tempfile.TemporaryDirectory() as d:
        res = {}
        for tag, env in (("resident", {}), ("general", {"BGM_FORCE_GENERAL": "1"})):
            path = os.path.join(d, tag + ".npz")
            subprocess.run([sys.executable, "-c", code, path], check=True, env=dict(os.environ, **env), timeout=600)
            res[tag] = np.load(path)
        a, b = res["resident"], res["general"]
        assert np.abs(a["lp"] - b["lp"]).max() <= 1e-5 * np.abs(a["lp"]).max() + 1e-3
        same = np.all(np.abs(a["draws"][-1] - b["draws"][-1]) <= 1e-4, axis=1).mean()
        assert same >= 0.97, same
        assert np.abs(a["adrf"] - b["adrf"]).max() <= 5e-3


# ---------------------------------------------------------------------------------------------------------------------------
# fit of the same shapes: no LDS blob holds them, the row-tile chains (fit_chain.h) read the canonical parameters in place; short
# minibatches (the last one of an epoch, a rank's share under data parallelism) run on the same kernels with the tail rows masked
# ---------------------------------------------------------------------------------------------------------------------------
from oracle import fit as OF      # noqa: E402


def _flat(grads):
    return np.concatenate([np.concatenate([dW.ravel(), db.ravel()]) for dW, db in grads])


@pytest.mark.parametrize("case", [dict(z_dims=[3, 6, 3, 6], p=177, binary=True, n=300, B=32),     # Semi_acic.yaml
                                  dict(z_dims=[3, 6, 3, 6], p=177, binary=True, n=300, B=20),     # ... a short last minibatch
                                  dict(z_dims=[5, 5, 5, 5], p=100, binary=False, n=300, B=32),    # Sim_Colangelo.yaml
                                  dict(z_dims=[5, 5, 5, 5], p=100, binary=False, n=300, B=7),
                                  dict(z_dims=[4, 4, 4, 15], p=50, binary=False, n=100, B=4)])      # sum(z_dims) = 27; head inputs within one tile
def test_fit_gradients_match_oracle(case):
    import torch
    m = _model(7, case["z_dims"], case["p"], case["binary"])
    x, y, v = _data(case["n"], case["p"], 8, case["binary"])
    z = np.random.RandomState(9).randn(case["n"], sum(case["z_dims"])).astype(np.float32)
    eng = _engine(m)
    dev = eng.device
    xd, yd, vd, zd = (torch.from_numpy(a).to(dev) for a in (x.ravel(), y.ravel(), v, z))
    B = case["B"]
    idx_np = np.random.RandomState(3).choice(case["n"], B, replace=False).astype(np.int32)
    idx = torch.from_numpy(idx_np).to(dev)
    npar = eng.fit_begin(case["n"], 32)
    grad = torch.empty(npar, device=dev)
    loss = torch.zeros(8, device=dev, dtype=torch.float64)
    eng.fit_theta_grad(xd, yd, vd, zd, idx, B, grad, loss)
    m64 = OC.cast_model(m, np.float64)
    bz, bx, by, bv = (a[idx_np].astype(np.float64) for a in (z, x, y, v))
    lv, mse_v, gg, _ = OF.g_loss_and_grads(m64, bz, bv)
    lx, _, gh, _ = OF.h_loss_and_grads(m64, bz, bx)
    ly, mse_y, gf, _ = OF.f_loss_and_grads(m64, bz, bx, by)
    got = grad.cpu().numpy()
    o = 0
    for part in (_flat(gg), _flat(gf), _flat(gh)):
        g_ = got[o:o + part.size]
        assert np.abs(g_ - part).max() <= 5e-5 * np.abs(part).max() + 1e-7, (np.abs(g_ - part).max(), np.abs(part).max())
        o += part.size
    l = loss.cpu().numpy()
    assert np.allclose([l[0] / B, l[2] / B, l[4] / B], [lv, lx, ly], rtol=5e-5)
    zm = torch.zeros_like(zd); zv = torch.zeros_like(zd)
    z_before = zd.clone()
    loss.zero_()
    eng.fit_z_step(xd, yd, vd, zd, zm, zv, idx, B, 1e-3, lazy=True, loss=loss)
    lz_ref, dz_ref = OF.z_loss_and_grad(m64, bz, bx, by, bv)
    assert np.isclose(loss.cpu().numpy()[6] / B, lz_ref, rtol=5e-5)
    gm = zm.cpu().numpy()[idx_np] / 0.1
    assert np.abs(gm - dz_ref).max() <= 5e-5 * np.abs(dz_ref).max() + 1e-8
    untouched = np.setdiff1d(np.arange(case["n"]), idx_np)
    assert torch.equal(zd[untouched], z_before[untouched])
    eng.fit_end()


def test_fit_steps_then_sampling_with_the_trained_parameters():
    """Four Adam iterations (the last minibatch short) on the Semi_acic shape track the oracle; evaluate / log-posterior DURING the
    fit session read the device parameters (general sampling path), and after fit_end the host copies."""
    import torch
    z_dims, p, n = [3, 6, 3, 6], 177, 96
    m = _model(11, z_dims, p, True)
    x, y, v = _data(n, p, 12, True)
    z = np.random.RandomState(13).randn(n, sum(z_dims)).astype(np.float32)
    eng = _engine(m)
    dev = eng.device
    xd, yd, vd, zd = (torch.from_numpy(a).to(dev) for a in (x.ravel(), y.ravel(), v, z.copy()))
    zm = torch.zeros_like(zd); zv = torch.zeros_like(zd)
    B, lr = 32, 1e-3
    npar = eng.fit_begin(n, B)
    grad = torch.empty(npar, device=dev)
    st = OF.FitState(OC.cast_model(m, np.float64), z.astype(np.float64), lr, lr)
    x64, y64, v64 = (a.astype(np.float64) for a in (x, y, v))
    rs = np.random.RandomState(5)
    for step in range(4):
        idx_np = rs.choice(n, B if step < 3 else 17, replace=False).astype(np.int32)
        idx = torch.from_numpy(idx_np).to(dev)
        eng.fit_theta_grad(xd, yd, vd, zd, idx, len(idx_np), grad)
        eng.fit_theta_apply(grad, lr)
        eng.fit_z_step(xd, yd, vd, zd, zm, zv, idx, len(idx_np), lr, lazy=False)
        OF.fit_step(st, x64, y64, v64, idx_np, lazy_z=False)
    assert np.abs(zd.cpu().numpy() - st.data_z).max() <= 3e-4
    lp = eng.logpost(xd, yd, vd, zd).cpu().numpy()
    ref = OC.log_posterior(dict(st.m), x64, y64, v64, st.data_z)
    assert np.abs(lp - ref).max() <= 1e-3          # measured 4.1e-5 on |log p| <= 143 (fp32 kernels vs the float64 oracle after four Adam steps)
    eng.fit_end()
    lp2 = eng.logpost(xd, yd, vd, zd).cpu().numpy()
    assert np.abs(lp - lp2).max() <= 1e-4 * np.abs(lp).max()


@pytest.mark.parametrize("name,z_dims,p,binary", [("Semi_acic", [3, 6, 3, 6], 177, True), ("Sim_Colangelo", [5, 5, 5, 5], 100, False)])
def test_reference_yaml_shapes_through_the_class_with_deterministic_nets(tmp_path, name, z_dims, p, binary):
    """configs/Semi_acic.yaml and configs/Sim_Colangelo.yaml with use_bnn = False: EGM warm start, iterative updates (short last
    minibatch), evaluation every epoch and predict run end to end -- warm start and fit on the row-tile chains, evaluate / predict on
    the general sampling path."""
    from bayesgm_amd.models import CausalBGM
    rs = np.random.RandomState(0)
    n = 203
    v = rs.randn(n, p).astype(np.float32)
    x = ((rs.rand(n, 1) > 0.5) if binary else rs.exponential(size=(n, 1))).astype(np.float32)
    y = (x + 0.3 * v[:, :1] + rs.randn(n, 1)).astype(np.float32)
    params = dict(dataset=name, output_dir=str(tmp_path), save_res=False, save_model=False, binary_treatment=binary, use_bnn=False,
                  z_dims=z_dims, v_dim=p, lr_theta=1e-4, lr_z=1e-4, lr=2e-4, g_d_freq=5, use_z_rec=True, kl_weight=1e-4,
                  g_units=[64] * 5, e_units=[64] * 5, f_units=[64, 32, 8], h_units=[64, 32, 8], dz_units=[64, 32, 8])
    m = CausalBGM(params, timestamp="t", random_seed=1)
    m.fit((x, y, v), epochs=2, epochs_per_eval=1, batch_size=32, use_egm_init=True, egm_n_iter=30, egm_batches_per_eval=15, verbose=0)
    assert m.data_z.shape == (n, sum(z_dims)) and np.isfinite(np.asarray(m.best_causal_pre)).all()
    if binary:
        eff, iv = m.predict((x, y, v), alpha=0.05, n_mcmc=24, burn_in=16, q_sd=0.5, verbose=0)
        assert eff.shape == (n,) and iv.shape == (n, 2) and np.isfinite(eff).all() and (iv[:, 0] <= iv[:, 1]).all()
    else:
        xs = np.linspace(0, 2, 5)
        eff, iv = m.predict((x, y, v), alpha=0.05, n_mcmc=24, burn_in=16, x_values=xs, q_sd=0.5, verbose=0)
        assert eff.shape == (5,) and iv.shape == (5, 2) and np.isfinite(eff).all()
    assert 0.0 < m.last_acceptance_rate < 1.0
