"""Class-level parity: the public `CausalBGM.fit` / `CausalBGM.predict` / `BGM.fit` (orchestration + kernels, through the C ABI)
against the oracle's restatement of the reference's loops on the committed C0-sized panel
(tests/golden/hirano_imbens_N2000_p20_seed0.npz: x, y, v produced by the reference's own generator).

  fit      /root/reference/src/bayesgm/models/causalbgm/base.py:476-532: Z ~ N(0,1) from NumPy's global stream, per epoch
           np.random.choice(N, N, replace=False), minibatches of 32 incl. the SHORT LAST one (2000 = 62 x 32 + 16),
           epochs + 1 passes, evaluation at epoch % epochs_per_eval == 0, best epoch by mse_y.
  predict  :573-668: `bs`-blocks (3 blocks, the last one short), fixed and ADAPTIVE proposal scale (adapted per block from the
           block's own 100-iteration acceptance window), adrf_draw_sums weighting, mean and np.quantile intervals.

Both sides consume the same host stream (np.random state captured after the constructor) and the same Philox streams.
Tolerances: fp32 MFMA kernels vs a float64 oracle; per-epoch mean losses 2e-5 relative; parameters and latents within 2 %
of the distance they travelled (Adam normalises steps to ~lr, so rounding can shift a step's direction for near-zero
gradients) -- observed values are printed by the test."""
import os

import numpy as np
import pytest

gpu = pytest.mark.gpu

from oracle import causal as OC  # noqa: E402
from oracle import fit as OF     # noqa: E402

GOLD = os.path.join(os.path.dirname(__file__), "golden", "hirano_imbens_N2000_p20_seed0.npz")


def _params(binary=False, p=20, z_dims=(1, 1, 1, 7)):
    return dict(dataset="t", output_dir="gpurun_out/t", save_res=False, save_model=False, binary_treatment=binary, use_bnn=False,
                z_dims=list(z_dims), v_dim=p, lr_theta=1e-4, lr_z=1e-4, g_units=[64] * 5, f_units=[64, 32, 8], h_units=[64, 32, 8],
                e_units=[64] * 5, dz_units=[64, 32, 8], kl_weight=1e-4, lr=2e-4, g_d_freq=5, use_z_rec=True)


def _oracle_model(model, dtype, binary=False):
    m = dict(g=model.nets["g"], f=model.nets["f"], h=model.nets["h"], e=model.nets["e"], z_dims=list(model._p["z_dims"]),
             v_dim=int(model._p["v_dim"]), binary_treatment=binary)
    return OC.cast_model(m, dtype)


def _flat(net):
    return np.concatenate([np.concatenate([np.asarray(W, np.float64).ravel(), np.asarray(b, np.float64).ravel()]) for W, b in net])


@gpu
@pytest.mark.parametrize("lr", [1e-4, 1e-3])
def test_causalbgm_fit_trace_matches_oracle(lr):
    from bayesgm_amd.models import CausalBGM
    g = np.load(GOLD)
    x, y, v = g["x"], g["y"], g["v"]
    n, q = len(x), 10
    prm = dict(_params(), lr_theta=lr, lr_z=lr)
    model = CausalBGM(prm, random_seed=5)
    m = _oracle_model(model, np.float64)
    init = {k: _flat(m[k]) for k in "gfh"}
    host_state = np.random.get_state()
    model.fit((x, y, v), epochs=2, epochs_per_eval=1, batch_size=32, use_egm_init=False, verbose=0)
    # ---- the oracle on the same host stream
    np.random.set_state(host_state)
    z0 = np.random.normal(0, 1, size=(n, q)).astype('float32')
    st = OF.FitState(m, z0.astype(np.float64), lr, lr)
    data64 = (x.astype(np.float64), y.astype(np.float64), v.astype(np.float64))
    rows = np.array([32] * 62 + [16], np.float64)
    worst = 0.0
    mse_y_trace = []
    for epoch in range(3):
        hist = OF.fit_epochs(st, data64, 0, 32, np.random)               # one pass: choice(N, N) + 63 minibatches
        assert hist.shape == (63, 7)
        want = (hist * rows[:, None]).sum(axis=0) / n                     # loss_x, mse_x, loss_y, mse_y, loss_v, mse_v, loss_z
        got = model.fit_history[epoch]
        for key, w in zip(("loss_x", "loss_mse_x", "loss_y", "loss_mse_y", "loss_v", "loss_mse_v", "loss_postrior_z"), want):
            err = abs(got[key] - w) / max(1e-6, abs(w))
            worst = max(worst, err)
            assert err <= 2e-5, (epoch, key, got[key], w)
        dose, mse_x, mse_y, mse_v = OC.evaluate(m, data64, data_z=st.data_z)
        mse_y_trace.append(mse_y)
        for key, w in (("mse_x", mse_x), ("mse_y", mse_y), ("mse_v", mse_v)):
            assert abs(got[key] - w) <= 2e-5 * abs(w), (epoch, key, got[key], w)
        if epoch == int(np.argmin(mse_y_trace)):
            best_dose = dose
    assert model.best_epoch == int(np.argmin(mse_y_trace))
    np.testing.assert_allclose(model.best_causal_pre, best_dose, rtol=0, atol=5e-5)
    # ---- parameters and latents after 189 steps
    for k in "gfh":
        moved = np.abs(_flat(m[k]) - init[k]).max()
        diff = np.abs(_flat(model.nets[k]) - _flat(m[k])).max()
        print("fit lr=%g net %s: moved %.3e, |hip - oracle| %.3e" % (lr, k, moved, diff))
        assert diff <= 0.02 * moved + 1e-7, (k, diff, moved)
    dz = np.abs(model.data_z.cpu().numpy() - st.data_z).max()
    moved_z = np.abs(st.data_z - z0).max()
    print("fit lr=%g latents: moved %.3e, |hip - oracle| %.3e, worst epoch-loss rel err %.2e" % (lr, moved_z, dz, worst))
    assert dz <= 0.02 * moved_z + 1e-7


@gpu
@pytest.mark.parametrize("q_sd", [1.0, -1.0])
def test_causalbgm_predict_continuous_blocks_vs_oracle(q_sd):
    """3 `bs`-blocks (250 + 250 + 100 rows); q_sd = -1 adapts the proposal scale per block at iterations 50 and 100."""
    from bayesgm_amd.models import CausalBGM
    g = np.load(GOLD)
    x, y, v = g["x"][:600], g["y"][:600], g["v"][:600]
    model = CausalBGM(_params(), random_seed=11)
    m = _oracle_model(model, np.float32)
    xs = np.linspace(0, 3, 7)
    seed = (model._base_seed * 1000003 + model._seed_counter + 1) & 0x7FFFFFFFFFFFFFFF
    adrf, interval = model.predict((x, y, v), alpha=0.1, n_mcmc=60, burn_in=130, x_values=xs, q_sd=q_sd, sample_y=True, bs=250,
                                   verbose=0)
    ref_adrf, ref_int = OC.predict(m, (x, y, v), alpha=0.1, n_mcmc=60, burn_in=130, x_values=xs, q_sd=q_sd, sample_y=True,
                                   bs=250, seed=seed)
    assert adrf.shape == (7,) and interval.shape == (7, 2)
    print("predict q_sd=%g: |adrf - oracle| %.2e, |interval - oracle| %.2e" % (q_sd, np.abs(adrf - ref_adrf).max(),
                                                                              np.abs(interval - ref_int).max()))
    # one accept/reject decision that flips in fp32 moves a 600-row mean by ~1/600 of a unit
    np.testing.assert_allclose(adrf, ref_adrf, rtol=0, atol=3e-3)
    np.testing.assert_allclose(interval, ref_int, rtol=0, atol=6e-3)
    assert np.all(interval[:, 0] <= adrf) and np.all(adrf <= interval[:, 1])


@gpu
@pytest.mark.parametrize("q_sd", [1.0, 0.0])
def test_causalbgm_predict_binary_blocks_vs_oracle(q_sd):
    """Binary treatment: per-row ITE means and np.quantile intervals over 3 blocks, z_dims of the CLI default."""
    from bayesgm_amd.models import CausalBGM
    from bayesgm_amd.datasets import binarize_treatment
    g = np.load(GOLD)
    x, y, v = binarize_treatment(g["x"][:600]), g["y"][:600], g["v"][:600]
    model = CausalBGM(_params(binary=True, z_dims=(3, 3, 6, 6)), random_seed=12)
    m = _oracle_model(model, np.float32, binary=True)
    seed = (model._base_seed * 1000003 + model._seed_counter + 1) & 0x7FFFFFFFFFFFFFFF
    ite, interval = model.predict((x, y, v), alpha=0.05, n_mcmc=50, burn_in=120, q_sd=q_sd, sample_y=True, bs=250, verbose=0)
    ref_ite, ref_int = OC.predict(m, (x, y, v), alpha=0.05, n_mcmc=50, burn_in=120, q_sd=q_sd, sample_y=True, bs=250, seed=seed)
    assert ite.shape == (600,) and interval.shape == (600, 2)
    ok = (np.abs(ite - ref_ite) <= 1e-4) & (np.abs(interval - ref_int).max(axis=1) <= 1e-4)
    print("binary predict q_sd=%g: rows matching the oracle %.4f" % (q_sd, ok.mean()))
    assert ok.mean() >= 0.97          # a chain that flips one fp32 accept decision leaves the oracle's path
    assert abs(float(ite.mean()) - float(ref_ite.mean())) <= 2e-3            # ATE = mean(ITE) (tutorial cell 31)


@gpu
def test_bgm_fit_trace_matches_oracle():
    """BGM.fit (bgm/base.py:343-442): Z ~ N(0,1) from the host stream, np.random.choice permutation per epoch, the incomplete
    last batch SKIPPED (2000 rows -> 62 minibatches), theta step (training-mode BatchNorm, moving averages) + fresh-slot Z
    step per minibatch, one evaluation per epoch (which draws its noise seed from the host stream) -- against
    oracle.bgm.fit_step on the same host stream."""
    from bayesgm_amd.models import BGM
    from oracle import bgm as OB
    g = np.load(GOLD)
    data = np.concatenate([g["x"], g["y"], g["v"][:, :18]], axis=1).astype(np.float32)          # [2000 x 20]
    n, q, lr = len(data), 10, 1e-3
    prm = dict(dataset="t", output_dir="gpurun_out/t", save_res=False, save_model=False, use_bnn=False, z_dim=q, x_dim=20,
               lr_theta=lr, lr_z=lr, g_units=[64] * 5, e_units=[64] * 5, dz_units=[64, 32, 8], dx_units=[64, 32, 8],
               kl_weight=5e-5, lr=1e-3, g_d_freq=1, use_z_rec=True, alpha=0.0, gamma=0.0)
    model = BGM(prm, random_seed=9)
    m = OB.cast_model({"z_dim": q, "x_dim": 20, "g": model.g}, np.float64)
    host_state = np.random.get_state()
    model.fit(data, batch_size=32, epochs=2, epochs_per_eval=1, use_egm_init=False, verbose=0)
    np.random.set_state(host_state)
    z0 = np.random.normal(0, 1, size=(n, q)).astype('float32')
    st = OB.BgmFitState(m, z0.astype(np.float64), lr, lr)
    data64 = data.astype(np.float64)
    for epoch in range(3):
        perm = np.random.choice(n, n, replace=False)
        hist = np.array([OB.fit_step(st, data64, perm[k * 32:(k + 1) * 32]) for k in range(62)])
        np.random.randint(0, 2 ** 31 - 1)                     # the evaluation's noise seed (use_x_sd=True)
        got = model.fit_history[epoch]
        assert abs(got["loss_x"] - hist[:, 0].mean()) <= 2e-5 * abs(hist[:, 0].mean()), (epoch, got, hist[:, 0].mean())
        assert abs(got["loss_mse_x"] - hist[:, 1].mean()) <= 2e-5 * abs(hist[:, 1].mean()), (epoch, got, hist[:, 1].mean())
    dz = np.abs(model.data_z.cpu().numpy() - st.data_z).max()
    moved = np.abs(st.data_z - z0).max()
    print("BGM fit: latents moved %.3e, |hip - oracle| %.3e" % (moved, dz))
    assert dz <= 0.02 * moved + 1e-7
    for k in ("mean", "var"):          # BatchNorm moving averages after 2 x 186 training-mode calls
        np.testing.assert_allclose(model.g["bn"][k], m["g"]["bn"][k], rtol=0, atol=2e-5)
    wd = max(np.abs(model.g[k][0] - m["g"][k][0]).max() for k in ("mean", "var"))
    wm = max(np.abs(m["g"][k][0] - np.asarray(BGM(prm, random_seed=9).g[k][0], np.float64)).max() for k in ("mean", "var"))
    print("BGM fit: head weights moved %.3e, |hip - oracle| %.3e" % (wm, wd))
    assert wd <= 0.02 * wm + 1e-7


@gpu
def test_checkpoint_manager_semantics(tmp_path):
    """tf.train.CheckpointManager(max_to_keep=5) + restore-latest-at-construction (causalbgm/base.py:112-128, 524-530): a fit that
    improves mse_y at every evaluation saves ckpt-<epoch>; at most five stay; a model constructed with the same timestamp
    restores the latest one -- parameters, and (installed by its next fit) the Adam slots and step counters."""
    from bayesgm_amd.models import CausalBGM
    g = np.load(GOLD)
    x, y, v = g["x"][:512], g["y"][:512], g["v"][:512]
    prm = dict(_params(), output_dir=str(tmp_path), save_model=True, lr_theta=1e-3, lr_z=1e-3)
    model = CausalBGM(prm, timestamp="run1", random_seed=4)
    model.fit((x, y, v), epochs=7, epochs_per_eval=1, batch_size=64, use_egm_init=False, verbose=0)
    files = sorted(f for f in os.listdir(model.checkpoint_path) if f.startswith("ckpt-"))
    assert 1 <= len(files) <= 5, files
    latest = model.ckpt_manager.latest_checkpoint
    assert latest is not None and os.path.basename(latest) in files
    d = np.load(latest)
    assert {"opt_m", "opt_v", "opt_steps", "g_W0", "e_W0"} <= set(d.files) and "data_z" not in d.files
    epoch_saved = int(os.path.basename(latest)[5:-4])
    assert int(d["opt_steps"][0]) == 8 * (epoch_saved + 1)          # 8 minibatches of 64 per epoch
    again = CausalBGM(prm, timestamp="run1", random_seed=99)        # other seed: the weights must come from the checkpoint
    for k in "gfhe":
        for (W, b), i in zip(again.nets[k], range(99)):
            assert np.array_equal(W, d["%s_W%d" % (k, i)]) and np.array_equal(b, d["%s_b%d" % (k, i)])
    assert again._restored_opt is not None and again._restored_opt["t_theta"] == int(d["opt_steps"][0])
    again.fit((x, y, v), epochs=0, epochs_per_eval=1, batch_size=64, use_egm_init=False, verbose=0)   # slots installed, steps go on
    assert again._restored_opt is None
    fresh = CausalBGM(prm, timestamp="run2", random_seed=99)        # another run directory: nothing to restore
    assert fresh.ckpt_manager.latest_checkpoint is None


def test_checkpoint_manager_prunes_on_cpu_arrays(tmp_path):
    from bayesgm_amd.models._checkpoint import CheckpointManager
    m = CheckpointManager(str(tmp_path / "c"), max_to_keep=5)
    assert m.latest_checkpoint is None
    for e in range(8):
        m.save("ckpt-%d.npz" % e, dict(a=np.arange(3) + e))
    names = sorted(f for f in os.listdir(m.directory) if f.endswith(".npz"))
    assert names == ["ckpt-%d.npz" % e for e in range(3, 8)]
    assert os.path.basename(m.latest_checkpoint) == "ckpt-7.npz" and np.load(m.latest_checkpoint)["a"][0] == 7


@gpu
@pytest.mark.parametrize("z_adam", ["replay", "lazy", "dense"])
def test_epoch_loop_inside_the_library_equals_the_host_loop(z_adam):
    """CausalBGM.fit(host_loop=False) -- one bgm_causal_fit_epoch call per epoch, the latent phase of a minibatch on a second stream
    beside the theta phase of the next -- gives the networks, the latent table and the per-epoch losses of the per-minibatch calls
    from Python bit for bit (2000 = 62 x 32 + 16: the short last minibatch included)."""
    from bayesgm_amd.models import CausalBGM
    g = np.load(GOLD)
    x, y, v = g["x"], g["y"], g["v"]
    res = []
    for host_loop in (True, False):
        model = CausalBGM(dict(_params(), lr_theta=1e-3, lr_z=1e-3), random_seed=5)
        model.fit((x, y, v), epochs=2, epochs_per_eval=1, batch_size=32, use_egm_init=False, verbose=0, z_adam=z_adam, host_loop=host_loop)
        res.append((model.data_z.cpu().numpy().copy(), {k: _flat(model.nets[k]) for k in "gfh"}, [dict(h) for h in model.fit_history]))
    (za, wa, ha), (zb, wb, hb) = res
    assert np.array_equal(za, zb)
    for k in "gfh":
        assert np.array_equal(wa[k], wb[k]), k
    for a, b in zip(ha, hb):
        for key in ("loss_v", "loss_x", "loss_y", "loss_postrior_z", "mse_v", "mse_y"):
            assert abs(a[key] - b[key]) <= 1e-6 * max(1.0, abs(a[key])), (key, a[key], b[key])     # (fp64 atomics: summation order)
