"""GPU parity tests of the native EGM warm start (egm_kernels.h) vs the NumPy oracle (oracle/egm.py), whose
gradients are themselves checked against PyTorch autograd in tests/test_oracle_autograd.py.

Tolerances (fp32 kernels vs float64 oracle): losses 2e-5 relative; gradients 5e-5 of the largest gradient entry
of the same step; after 5 alternating Adam steps the parameters agree to 2e-5 absolute (lr 2e-4)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from oracle import egm as OE      # noqa: E402
from oracle import nets as N      # noqa: E402


def _setup(binary, p=23, z_dims=(1, 1, 1, 7), B=32, n=200, seed=0):
    import torch
    from bayesgm_amd.engine import CausalEngine
    from bayesgm_amd import _lib
    rs = np.random.RandomState(seed)
    q = sum(z_dims)
    g_units, e_units, f_units, h_units, dz_units = [64] * 5, [64] * 5, [64, 32, 8], [64, 32, 8], [64, 32, 8]
    nets = {"g": N.init_mlp(rs, [q] + g_units + [p + 1]), "e": N.init_mlp(rs, [p] + e_units + [q]),
            "f": N.init_mlp(rs, [z_dims[0] + z_dims[1] + 1] + f_units + [2]),
            "h": N.init_mlp(rs, [z_dims[0] + z_dims[2]] + h_units + [2])}
    for k in nets:
        nets[k] = [(W, (0.1 * rs.randn(*b.shape)).astype(np.float32)) for W, b in nets[k]]
    dz = OE.init_disc(rs, q, dz_units)
    dz["b"] = [(0.1 * rs.randn(*b.shape)).astype(np.float32) for b in dz["b"]]
    dz["gamma"] = [(1 + 0.2 * rs.randn(*b.shape)).astype(np.float32) for b in dz["gamma"]]
    dz["beta"] = [(0.1 * rs.randn(*b.shape)).astype(np.float32) for b in dz["beta"]]
    eng = CausalEngine(p, list(z_dims), binary_treatment=binary, g_units=g_units, f_units=f_units, h_units=h_units,
                       e_units=e_units)
    eng.set_model(g=nets["g"], f=nets["f"], h=nets["h"], e=nets["e"])
    v = rs.randn(n, p).astype(np.float32)
    x = (rs.rand(n) > 0.5).astype(np.float32) if binary else rs.rand(n).astype(np.float32)
    y = rs.randn(n).astype(np.float32)
    params = dict(v_dim=p, z_dims=list(z_dims), binary_treatment=binary, use_z_rec=True, lr=2e-4)
    dev = dict(v=torch.from_numpy(v).cuda(), x=torch.from_numpy(x).cuda(), y=torch.from_numpy(y).cuda())
    return eng, nets, dz, params, (x, y, v), dev, rs, dz_units


def _rel(a, b):
    return float(np.abs(a - b).max() / (np.abs(b).max() + 1e-30))


@pytest.mark.parametrize("case", [dict(binary=False, p=23, z_dims=(1, 1, 1, 7), B=32),
                                  dict(binary=True, p=200, z_dims=(3, 3, 6, 6), B=32),
                                  dict(binary=False, p=50, z_dims=(2, 1, 3, 4), B=17),
                                  dict(binary=False, p=23, z_dims=(1, 1, 1, 7), B=32, disc_norm="fixed"),
                                  dict(binary=True, p=200, z_dims=(3, 3, 6, 6), B=19, disc_norm="fixed"),
                                  # the shapes of the register-chained discriminator step (egm_chain.h): one and two 16-row tiles
                                  dict(binary=False, p=200, z_dims=(1, 1, 1, 7), B=32, disc_norm="fixed"),
                                  dict(binary=True, p=37, z_dims=(2, 3, 4, 5), B=16, disc_norm="fixed"),
                                  dict(binary=False, p=100, z_dims=(3, 3, 3, 3), B=32, disc_norm="fixed"),
                                  # the binary-treatment configs' latent layout: q = 18, two latent input tiles
                                  dict(binary=True, p=177, z_dims=(3, 3, 6, 6), B=32, disc_norm="fixed"),
                                  dict(binary=False, p=60, z_dims=(5, 5, 5, 16), B=32, disc_norm="fixed"),     # q = 31, the widest configurable latent
                                  dict(binary=True, p=177, z_dims=(3, 3, 6, 6), B=16, disc_norm="fixed")])     # two latent tiles at B = 16: phase machine
def test_egm_step_gradients_match_oracle(case):
    import torch
    B = case["B"]
    eng, nets, dz, params, (x, y, v), dev, rs, dz_units = _setup(case["binary"], case["p"], case["z_dims"], B)
    q = sum(case["z_dims"])
    if case.get("disc_norm") == "fixed":      # discriminator BatchNormalization in inference mode (bgm_set_disc_norm)
        eng.set_disc_norm("fixed")
        dz["fixed_norm"] = True
    eng.egm_begin(B, dz_units, params["lr"], True, dz)
    n_gen = sum(W.size + b.size for k in ("g", "e", "f", "h") for W, b in nets[k])
    n_dz = CausalEngineFlat(dz).size
    z = rs.randn(B, q).astype(np.float32)
    idx = rs.choice(len(x), B, replace=False).astype(np.int32)
    zd, idd = torch.from_numpy(z).cuda(), torch.from_numpy(idx).cuda()
    nets64 = {k: N.cast_net(nets[k], np.float64) for k in nets}
    dz64 = OE.cast_disc(dz, np.float64)
    # ---- discriminator step
    out_d = torch.zeros(2, device="cuda")
    eng.egm_disc_step(zd, idd, dev["v"], 0.37, apply=False, out=out_d)
    l1, l2, gr = OE.disc_step_grads(nets64, dz64, z.astype(np.float64), v[idx].astype(np.float64), 0.37)
    ref = np.concatenate([a.ravel() for a in OE.disc_param_list(gr)])
    got = eng.egm_read(3, n_dz)
    od = out_d.cpu().numpy()
    assert abs(od[0] - l1) <= 2e-5 * abs(l1) + 1e-6 and abs(od[1] - l2) <= 2e-5 * abs(l2) + 1e-5
    assert _rel(got, ref) <= 5e-5, _rel(got, ref)
    # ---- generator step
    out_g = torch.zeros(6, device="cuda")
    eng.egm_gen_step(zd, idd, dev["v"], dev["x"], dev["y"], apply=False, out=out_g)
    losses, gr = OE.gen_step_grads(nets64, dz64, params, z.astype(np.float64), v[idx].astype(np.float64),
                                   x[idx].astype(np.float64).reshape(-1, 1), y[idx].astype(np.float64).reshape(-1, 1))
    ref = np.concatenate([a.ravel() for a in OE.gen_param_list(gr)])
    got = eng.egm_read(2, n_gen)
    assert np.all(np.abs(out_g.cpu().numpy() - losses) <= 2e-5 * np.abs(losses) + 1e-6), (out_g.cpu().numpy(), losses)
    assert _rel(got, ref) <= 5e-5, _rel(got, ref)
    # apply=False left the parameters alone
    assert np.array_equal(eng.egm_read(1, n_dz), CausalEngineFlat(dz))
    eng.egm_end()


def CausalEngineFlat(dz):
    from bayesgm_amd.engine import CausalEngine
    return CausalEngine.flatten_disc(dz)


@pytest.mark.parametrize("disc_norm,p,z_dims", [("batch", 23, (1, 1, 1, 7)), ("fixed", 23, (1, 1, 1, 7)), ("fixed", 100, (1, 1, 1, 7)),
                                                ("fixed", 200, (1, 1, 1, 7)), ("fixed", 177, (3, 3, 6, 6))])
def test_egm_alternating_adam_steps_track_oracle(disc_norm, p, z_dims):
    """p = 100 / 200 with fixed normalisation: both steps run as register-chained row tiles (egm_chain.h, egm_chain_gen.h), whose
    Adam step also maintains the transposed weight mirror the next step's backward chains read."""
    import torch
    B = 32
    eng, nets, dz, params, (x, y, v), dev, rs, dz_units = _setup(False, p, z_dims, B)
    q = sum(z_dims)
    if disc_norm == "fixed":
        eng.set_disc_norm("fixed")
        dz["fixed_norm"] = True
    eng.egm_begin(B, dz_units, params["lr"], True, dz)
    st = OE.EgmState({k: N.cast_net(nets[k], np.float64) for k in nets}, OE.cast_disc(dz, np.float64), params)
    for it in range(5):
        for _ in range(2):
            z = rs.randn(B, q).astype(np.float32)
            idx = rs.choice(len(x), B, replace=False).astype(np.int32)
            eps = float(rs.rand())
            eng.egm_disc_step(torch.from_numpy(z).cuda(), torch.from_numpy(idx).cuda(), dev["v"], eps)
            st.disc_step(z.astype(np.float64), v[idx].astype(np.float64), eps)
        z = rs.randn(B, q).astype(np.float32)
        idx = rs.choice(len(x), B, replace=False).astype(np.int32)
        eng.egm_gen_step(torch.from_numpy(z).cuda(), torch.from_numpy(idx).cuda(), dev["v"], dev["x"], dev["y"])
        st.gen_step(z.astype(np.float64), v[idx].astype(np.float64), x[idx].astype(np.float64).reshape(-1, 1),
                    y[idx].astype(np.float64).reshape(-1, 1))
    ref_g = np.concatenate([a.ravel() for a in OE.gen_param_list(st.nets)])
    ref_d = np.concatenate([a.ravel() for a in OE.disc_param_list(st.dz)])
    got_g, got_d = eng.egm_read(0, ref_g.size), eng.egm_read(1, ref_d.size)
    # The biases of the discriminator's hidden layers sit in front of a BatchNorm: their exact gradient is zero, the
    # fp32 gradient is rounding noise, and Adam's normalisation turns noise into steps of up to ~lr.  They do not
    # influence the function; exclude them from the parameter comparison.
    n_w = sum(a.size for a in st.dz["W"])
    inert = np.zeros(ref_d.size, bool)
    if disc_norm == "batch":
        inert[n_w:n_w + sum(a.size for a in st.dz["b"][:-1])] = True
    assert np.abs(got_g - ref_g).max() <= 2e-5 and np.abs(got_d - ref_d)[~inert].max() <= 2e-5
    assert (not inert.any()) or np.abs(got_d - ref_d)[inert].max() <= 12 * params["lr"]
    # parameters actually moved (Adam's first steps are ~lr each)
    assert np.abs(got_g - np.concatenate([a.ravel() for k in ("g", "e", "f", "h") for Wb in nets[k] for a in Wb])).max() > 5e-4
    # end of session: the trained networks are installed in the handle
    eng.egm_end()
    g_tr = eng.get_weights(0, [q] + [64] * 5 + [p + 1])
    assert np.abs(g_tr[0][0] - st.nets["g"][0][0]).max() <= 2e-5


@pytest.mark.parametrize("disc_norm,p,B", [("fixed", 200, 32), ("fixed", 200, 16), ("batch", 23, 8), ("fixed", 50, 4)])
def test_egm_split_step_equals_fused_step(disc_norm, p, B):
    """The data-parallel form -- step with apply = 0, bgm_causal_egm_grad, [all-reduce], bgm_causal_egm_apply -- takes the Adam steps the
    fused step takes (chains at p = 200: the transposed mirror is maintained by the apply kernel; small B: the rank share at 4 / 8 GPUs)."""
    import torch
    z_dims = (1, 1, 1, 7)
    q = sum(z_dims)
    outs = []
    for split in (False, True):
        eng, nets, dz, params, (x, y, v), dev, rs, dz_units = _setup(False, p, z_dims, B, seed=5)
        if disc_norm == "fixed":
            eng.set_disc_norm("fixed")
            dz["fixed_norm"] = True
        eng.egm_begin(B, dz_units, params["lr"], True, dz)
        n_gen, n_dz = eng.egm_sizes()
        bg, bd = torch.empty(n_gen, device="cuda"), torch.empty(n_dz, device="cuda")
        for it in range(4):
            for _ in range(2):
                z = torch.from_numpy(rs.randn(B, q).astype(np.float32)).cuda()
                idx = torch.from_numpy(rs.choice(len(x), B, replace=False).astype(np.int32)).cuda()
                eps = float(rs.rand())
                eng.egm_disc_step(z, idx, dev["v"], eps, apply=not split)
                if split:
                    eng.egm_grad(1, 1.0, bd)
                    eng.egm_apply(1, bd)
            z = torch.from_numpy(rs.randn(B, q).astype(np.float32)).cuda()
            idx = torch.from_numpy(rs.choice(len(x), B, replace=False).astype(np.int32)).cuda()
            eng.egm_gen_step(z, idx, dev["v"], dev["x"], dev["y"], apply=not split)
            if split:
                eng.egm_grad(0, 1.0, bg)
                eng.egm_apply(0, bg)
        outs.append((eng.egm_read(0, n_gen), eng.egm_read(1, n_dz)))
        eng.egm_end()
    (g0, d0), (g1, d1) = outs
    keep = np.ones(d0.size, bool)
    if disc_norm == "batch":       # hidden-layer biases in front of a BatchNorm: zero true gradient, rounding noise amplified by Adam (see above)
        n_w = sum(a.size for a in dz["W"])
        keep[n_w:n_w + sum(a.size for a in dz["b"][:-1])] = False
    # same arithmetic per parameter; the compiler may contract the two Adam expressions differently (1 ulp of a step of ~lr)
    assert np.abs(g0 - g1).max() <= 1e-7 and np.abs(d0 - d1)[keep].max() <= 1e-7, (np.abs(g0 - g1).max(), np.abs(d0 - d1)[keep].max())


def test_two_rank_egm_equals_one_process_on_the_same_global_minibatches(tmp_path):
    """Data-parallel warm start (north_star's collective): two ranks, each with ITS rows only, all-reducing the dz and the
    fused g | e | f | h gradients, end with identical networks that equal -- to fp32 summation order -- those of ONE process stepping on
    the global minibatches the two ranks formed.  Deterministic and Bayesian networks."""
    import json
    import os
    import sys
    from conftest import run_two_ranks
    out = str(tmp_path / "dp_egm.npz")
    r = run_two_ranks("dp_egm_smoke.py", timeout=400, extra_args=(out,))
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    objs = [json.loads(l) for l in r.stdout.splitlines() if l.startswith('{"rank"')]
    assert len(objs) == 2
    for o in objs:
        for key in ("det", "bnn"):
            assert o[key]["spread"] == 0.0 and o[key]["finite"] and o[key]["moved"] > 1e-3, o
    two = np.load(out)
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "scripts"))
    import dp_egm_common as CM
    for use_bnn, key in ((False, "det"), (True, "bnn")):
        m = CM.build(use_bnn, 0)
        m._egm_emulate_world = 2
        m.egm_init(CM.DATA, egm_n_iter=CM.N_ITER, batch_size=CM.BATCH, egm_batches_per_eval=CM.PER_EVAL, verbose=0)
        one = CM.flat_weights(m)
        d = float(np.abs(one - two[key]).max())
        print("MEASURED two-rank EGM vs one process (%s): max |dtheta| %.3g" % (key, d))
        assert d <= 2e-5, (key, d)


# ---------------------------------------------------------------------------------------------
# BGM EGM (bgm_egm_kernels.h) vs oracle.egm.bgm_*_step_grads
# ---------------------------------------------------------------------------------------------
def _bgm_setup(p, q=10, B=32, n=128, seed=0, gamma=0.0, alpha=0.0):
    import torch
    from bayesgm_amd.engine import BgmEngine
    rs = np.random.RandomState(seed)
    g = N.init_varnet(rs, q, (64,) * 5, p)
    g["bn"].update(gamma=(1 + 0.1 * rs.randn(q)).astype(np.float32), beta=(0.1 * rs.randn(q)).astype(np.float32),
                   mean=(0.2 * rs.randn(q)).astype(np.float32), var=(0.5 + rs.rand(q)).astype(np.float32))
    g["trunk"] = [(W, (0.1 * rs.randn(*b.shape)).astype(np.float32)) for W, b in g["trunk"]]
    g["mean"] = (g["mean"][0], (0.1 * rs.randn(p)).astype(np.float32))
    g["var"] = (g["var"][0], (0.1 * rs.randn(p)).astype(np.float32))
    e = [(W, (0.1 * rs.randn(*b.shape)).astype(np.float32)) for W, b in N.init_mlp(rs, [p] + [64] * 5 + [q])]
    ds = []
    for in_dim in (q, p):
        d = OE.init_disc(rs, in_dim, [64, 32, 8])
        d["b"] = [(0.1 * rs.randn(*b.shape)).astype(np.float32) for b in d["b"]]
        d["gamma"] = [(1 + 0.2 * rs.randn(*b.shape)).astype(np.float32) for b in d["gamma"]]
        d["beta"] = [(0.1 * rs.randn(*b.shape)).astype(np.float32) for b in d["beta"]]
        ds.append(d)
    eng = BgmEngine(p, q, g_units=[64] * 5)
    eng.set_weights(g)
    eng.egm_begin(B, [64] * 5, [64, 32, 8], [64, 32, 8], 1e-3, gamma, alpha, e, ds[0], ds[1])
    x = rs.randn(n, p).astype(np.float32)
    return eng, g, e, ds[0], ds[1], x, rs


def _c64(g, e, dz, dx):
    from oracle import bgm as OB
    g64 = {"bn": {k: v.astype(np.float64) for k, v in g["bn"].items()}, "trunk": N.cast_net(g["trunk"], np.float64),
           "mean": tuple(a.astype(np.float64) for a in g["mean"]), "var": tuple(a.astype(np.float64) for a in g["var"])}
    return g64, N.cast_net(e, np.float64), OE.cast_disc(dz, np.float64), OE.cast_disc(dx, np.float64)


def _g_flat(g):
    from bayesgm_amd.engine import flatten_varnet
    return flatten_varnet(g).astype(np.float64)


@pytest.mark.parametrize("case", [dict(p=20, gamma=0.0, alpha=0.0, B=32), dict(p=100, gamma=0.7, alpha=0.3, B=32),
                                  dict(p=37, gamma=1.0, alpha=0.0, B=19)])
def test_bgm_egm_step_gradients_match_oracle(case):
    import torch
    p, B, q = case["p"], case["B"], 10
    eng, g, e, dz, dx, x, rs = _bgm_setup(p, q, B, gamma=case["gamma"], alpha=case["alpha"])
    g64, e64, dz64, dx64 = _c64(g, e, dz, dx)
    z = rs.randn(B, q).astype(np.float32)
    xb = x[rs.choice(len(x), B, replace=False)]
    n1, n2 = rs.randn(B, p).astype(np.float32), rs.randn(B, p).astype(np.float32)
    d = lambda a: torch.from_numpy(a).cuda()
    # ---- discriminator step
    out_d = torch.zeros(3, device="cuda")
    eng.egm_disc_step(d(z), d(xb), d(n1), 0.31, 0.64, apply=False, out=out_d)
    losses, gr, caches = OE.bgm_disc_step_grads(g64, e64, dz64, dx64, z.astype(np.float64), xb.astype(np.float64),
                                                n1.astype(np.float64), 0.31, 0.64, case["gamma"])
    ref = np.concatenate([a.ravel() for a in OE.disc_param_list(gr["dz"]) + OE.disc_param_list(gr["dx"])])
    got = eng.egm_read(3)
    assert np.all(np.abs(out_d.cpu().numpy() - losses) <= 3e-5 * np.abs(losses) + 1e-6), (out_d.cpu().numpy(), losses)
    n_dz = OE.disc_param_list(gr["dz"]).__len__() and sum(a.size for a in OE.disc_param_list(gr["dz"]))
    for lo, hi in ((0, n_dz), (n_dz, ref.size)):        # per discriminator: relative to its largest gradient entry
        assert _rel(got[lo:hi], ref[lo:hi]) <= 1e-4, _rel(got[lo:hi], ref[lo:hi])
    # the generator call moved the BatchNorm moving statistics although g is not trained in this step
    from oracle import bgm as OB
    OB.bn_update_stats(g64, caches[0])
    th = eng.egm_read(0)
    assert np.abs(th[2 * q:3 * q] - g64["bn"]["mean"]).max() <= 1e-6 and np.abs(th[3 * q:4 * q] - g64["bn"]["var"]).max() <= 1e-6
    # ---- generator step (continues from the moved statistics)
    out_g = torch.zeros(6, device="cuda")
    eng.egm_gen_step(d(z), d(xb), d(n1), d(n2), apply=False, out=out_g)
    losses, gr, caches = OE.bgm_gen_step_grads(g64, e64, dz64, dx64, z.astype(np.float64), xb.astype(np.float64),
                                               n1.astype(np.float64), n2.astype(np.float64), case["alpha"])
    got = eng.egm_read(2)
    assert np.all(np.abs(out_g.cpu().numpy() - losses) <= 3e-5 * np.abs(losses) + 1e-6), (out_g.cpu().numpy(), losses)
    gl = OE.g_grad_list(gr["g"])
    ref_g = np.concatenate([gl[0].ravel(), gl[1].ravel(), np.zeros(2 * q)] + [a.ravel() for a in gl[2:]])
    ref_e = np.concatenate([a.ravel() for Wb in gr["e"] for a in Wb])
    assert _rel(got[:ref_g.size], ref_g) <= 1e-4, _rel(got[:ref_g.size], ref_g)
    assert _rel(got[ref_g.size:], ref_e) <= 1e-4, _rel(got[ref_g.size:], ref_e)
    eng.egm_end()


def test_bgm_egm_alternating_adam_steps_track_oracle_and_encoder():
    import torch
    from oracle import bgm as OB
    p, q, B = 20, 10, 32
    eng, g, e, dz, dx, x, rs = _bgm_setup(p, q, B, gamma=0.5, alpha=0.2)
    g64, e64, dz64, dx64 = _c64(g, e, dz, dx)
    st = OE.BgmEgmState(g64, e64, dz64, dx64, dict(lr=1e-3, gamma=0.5, alpha=0.2))
    d = lambda a: torch.from_numpy(a).cuda()
    for it in range(4):
        z = rs.randn(B, q).astype(np.float32); xb = x[rs.choice(len(x), B, replace=False)]
        n1 = rs.randn(B, p).astype(np.float32); ez, ex = float(rs.rand()), float(rs.rand())
        eng.egm_disc_step(d(z), d(xb), d(n1), ez, ex)
        st.disc_step(z.astype(np.float64), xb.astype(np.float64), n1.astype(np.float64), ez, ex)
        z = rs.randn(B, q).astype(np.float32); xb = x[rs.choice(len(x), B, replace=False)]
        n1, n2 = rs.randn(B, p).astype(np.float32), rs.randn(B, p).astype(np.float32)
        eng.egm_gen_step(d(z), d(xb), d(n1), d(n2))
        st.gen_step(z.astype(np.float64), xb.astype(np.float64), n1.astype(np.float64), n2.astype(np.float64))
    got_g = eng.egm_read(0)
    ref_g = np.concatenate([_g_flat(st.g)] + [a.ravel() for Wb in st.e for a in Wb])
    assert np.abs(got_g - ref_g).max() <= 5e-5, np.abs(got_g - ref_g).max()
    got_d = eng.egm_read(1)
    ref_d = np.concatenate([a.ravel() for a in OE.disc_param_list(st.dz) + OE.disc_param_list(st.dx)])
    inert = np.zeros(ref_d.size, bool)          # hidden-layer biases in front of a BatchNorm: zero exact gradient (see above)
    o = 0
    for dd in (st.dz, st.dx):
        n_w = sum(a.size for a in dd["W"])
        inert[o + n_w:o + n_w + sum(a.size for a in dd["b"][:-1])] = True
        o += sum(a.size for a in OE.disc_param_list(dd))
    assert np.abs(got_d - ref_d)[~inert].max() <= 5e-5
    # encoder pass over a whole panel == oracle MLP with the trained encoder
    z_enc = eng.egm_encode(x).cpu().numpy()
    assert np.abs(z_enc - N.mlp_forward(st.e, x.astype(np.float64))).max() <= 1e-4
    # end of session: the trained generator (incl. moved BatchNorm statistics) is installed in the engine
    eng.egm_end()
    g_tr = eng.get_weights()
    assert np.abs(g_tr["bn"]["mean"] - st.g["bn"]["mean"]).max() <= 1e-5 and np.abs(g_tr["trunk"][0][0] - st.g["trunk"][0][0]).max() <= 5e-5
