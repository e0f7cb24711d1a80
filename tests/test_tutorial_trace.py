"""The reference's published run as a statistical pin of the Bayesian-network path ("parity unpinned" for
everything the reference ships no vector for -- the notebook outputs are the one exception).

tests/golden/tutorial_trace.json holds the numbers of docs/source/causalbgm/tutorial_py.ipynb (extracted by
tests/golden/make_tutorial_trace.py): EGM log, per-epoch minibatch losses, panel MSEs, MH acceptance rate, ADRF RMSE / MAPE.
 * CPU: the committed logs of the build's own runs of the same setting (profiles/r02_accuracy/, scripts/accuracy_runs.py)
   are laid next to it: the default reading (BatchNormalization layers that the reference calls without `training=` run in
   inference mode) must sit inside the envelope, the literal batch-statistics reading must not -- this is the evidence the
   defaults of params['bnn_norm'] / params['disc_norm'] rest on (DESIGN_HISTORY.md section 2b).
 * GPU: one full run of the tutorial (N = 20000, p = 200, 30000 EGM iterations + 100 epochs, predict 5000 + 3000 at 20
   doses, bs = 20000; about three minutes) must land inside the same envelope.
Envelope = published value +- a tolerance that covers the seed-to-seed spread observed over the build's runs (stated per
statistic below; window medians / means as defined in scripts/compare_trace.py)."""
import json
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "scripts"))
from compare_trace import parse_log, summary  # noqa: E402

# statistic -> absolute tolerance around the published value.  Five runs of the default reading (seeds 123, 7, 11, 2026 and
# the GPU test's) set the widths: four of them agree with the published generator-side statistics to the third digit
# (l2_loss_z 0.23-0.25 vs 0.247, MSE_v 0.9668-0.9671 vs 0.9665); seed 2026 settles in a slightly worse generator optimum
# (l2_loss_z 0.44, MSE_v 0.987) with the best ADRF of all (RMSE 0.016) -- the tolerances of those rows cover it.
ENVELOPE = {
    "egm_early_med_l2_loss_z": 0.10, "egm_late_med_l2_loss_z": 0.25, "egm_late_med_l2_loss_v": 0.04, "egm_late_med_l2_loss_y": 0.15,
    "egm_late_med_dz_loss": 0.30, "egm_late_med_gp": 0.01,
    "fit_mean_loss_py_z": 0.08, "fit_mean_loss_pv_z": 2.5, "fit_mean_loss_mse_v": 0.03, "fit_mean_loss_mse_y": 0.2,
    "fit_mean_loss_postrior_z": 3.0, "fit_last20_loss_py_z": 0.08,
    "eval_mean_mse_y": 0.12, "eval_mean_mse_v": 0.03, "eval_mean_mse_x": 0.5,
    "acceptance": 0.03, "adrf_rmse": 0.015, "adrf_mape": 0.006,
}


def _published():
    ref = json.load(open(os.path.join(ROOT, "tests", "golden", "tutorial_trace.json")))
    s = summary(np.array(ref["egm"], float), np.array(ref["minibatch"], float), np.array(ref["eval"], float))
    s.update(acceptance=ref["acceptance_rate"], adrf_rmse=ref["adrf_rmse"], adrf_mape=ref["adrf_mape"])
    return s


def _outside(stats, pub):
    return {k: (stats[k], pub[k]) for k, tol in ENVELOPE.items() if abs(stats[k] - pub[k]) > tol}


def _log_stats(text):
    egm, mb, ev, res = parse_log(text)
    s = summary(egm, mb, ev)
    s.update(acceptance=res["acceptance"], adrf_rmse=res["adrf_rmse"], adrf_mape=res["adrf_mape"])
    return s


def test_committed_runs_against_published_trace():
    pub = _published()
    d = os.path.join(ROOT, "profiles", "r02_accuracy")
    for name in ("bnn_fixed_dfix_s123.log", "bnn_fixed_dfix_s7.log", "bnn_fixed_dfix_s11.log"):     # the default reading
        out = _outside(_log_stats(open(os.path.join(d, name)).read()), pub)
        assert not out, (name, out)
    out = _outside(json.load(open(os.path.join(d, "bnn_fixed_dfix_s2026_stats.json")))["stats"], pub)
    assert not out, out
    # the literal Keras-2.10 reading (batch statistics in the Bayesian nets and the discriminator) is far outside
    out = _outside(_log_stats(open(os.path.join(d, "bnn_batch_s123.log")).read()), pub)
    assert {"adrf_rmse", "adrf_mape", "eval_mean_mse_y", "egm_late_med_gp", "egm_late_med_l2_loss_z", "acceptance"} <= set(out), out
    # inference-mode Bayesian nets but a batch-statistics discriminator: the critic statistics and the ADRF error give it away
    out = _outside(_log_stats(open(os.path.join(d, "bnn_fixed_s123.log")).read()), pub)
    assert {"adrf_rmse", "adrf_mape", "egm_late_med_gp", "acceptance", "fit_mean_loss_postrior_z"} <= set(out), out


# ---- round 3: eight seeds of the product (HIP class) and eight of the ORACLE (scripts/oracle_tutorial.py: oracle/ end to end on the
# CPU, float32 NumPy, no HIP library) on the tutorial setting.  What sixteen runs show, and what the assertions below pin:
#  * the warm start has TWO optima.  Most seeds end where the published run ended (late l2_loss_z 0.23-0.25 vs 0.247, MSE_v
#    0.9662-0.9674 vs 0.9665, acceptance 0.097-0.101 vs 0.0948); some (HIP seeds 2026 and 99: 2 of 8) end in a second one
#    (l2_loss_z 0.42-0.44, MSE_v 0.983-0.987, acceptance 0.12; oracle seeds 123 and 11, and seed 1 half way: 3 of 8) that every
#    one of those statistics gives away and in which the dose-response error can be anything (HIP seed 2026: RMSE 0.016, seed 99:
#    0.53; the three oracle runs: 0.017-0.030).  A run is classified by (l2_loss_z, MSE_v) alone;
#  * for runs in the published optimum the envelope around the published numbers is several times tighter than round 2's on the
#    PANEL statistics (MSE_v +-0.002, acceptance +-0.01, late l2_loss_z +-0.04).  The statistics that average the LAST MINIBATCH of
#    each epoch (fit_mean_*, what the reference's progress bar shows) carry their own sampling error: 101 minibatches of 32 rows
#    give a standard error of 0.0034 on loss_mse_v, 0.35 on loss_pv_z, 0.014 on loss_py_z, 0.4 on loss_postrior_z (per-epoch spread
#    / sqrt(101), the same in the published trace), and two independent runs differ by sqrt(2) of that; their widths are ~4.2
#    standard errors.  (The product's eight runs do not show this spread: like the reference, the class consumes NumPy's global
#    stream AFTER Gaussian_sampler's constructor has re-seeded it with 1024, so every seed draws the same minibatch sequence --
#    their per-epoch loss_mse_v traces correlate at 0.99.  The oracle script draws from RandomState(seed) and does show it.);
#  * the ADRF error of a single run is NOT a statistic one run can pin to 0.0188: main-mode runs spread over 0.016-0.043 with the
#    published value at the product family's median.  The claim is therefore distributional: median within 0.006 of 0.0188 (oracle
#    runs, whose curve averages the first 5000 rows only: up to 0.012 above), every main-mode run below 0.05 (a flat or shifted
#    curve is > 0.3) -- a run at 0.037 does not "reproduce 0.0188", six runs with median 0.0195 do.
TIGHT = {
    "egm_early_med_l2_loss_z": 0.10, "egm_late_med_l2_loss_z": 0.04, "egm_late_med_l2_loss_v": 0.03, "egm_late_med_l2_loss_y": 0.20,
    "egm_late_med_dz_loss": 0.30, "egm_late_med_gp": 0.004,
    "fit_mean_loss_py_z": 0.06, "fit_mean_loss_pv_z": 1.5, "fit_mean_loss_mse_v": 0.015, "fit_mean_loss_mse_y": 0.15,
    "fit_mean_loss_postrior_z": 2.0, "fit_last20_loss_py_z": 0.11,
    "eval_mean_mse_y": 0.10, "eval_mean_mse_v": 0.002, "eval_mean_mse_x": 0.30,
    "acceptance": 0.010,
}
SEEDS = (123, 7, 11, 2026, 1, 42, 99, 314)
ORACLE_RUNS_COMMITTED = 8          # finished oracle runs under profiles/r03_oracle_anchor/ (a run is ~2 h of one host core)


def _main_mode(s, pub):
    return abs(s["egm_late_med_l2_loss_z"] - pub["egm_late_med_l2_loss_z"]) < 0.08 and abs(s["eval_mean_mse_v"] - pub["eval_mean_mse_v"]) < 0.006


def _check_family(stats, pub, min_main, rmse_above=0.006, mape_above=0.003):
    main = {k: s for k, s in stats.items() if _main_mode(s, pub)}
    other = {k: s for k, s in stats.items() if k not in main}
    assert len(main) >= min_main, sorted(other)
    for name, s in main.items():
        out = {k: (s[k], pub[k]) for k, tol in TIGHT.items() if abs(s[k] - pub[k]) > tol}
        assert not out, (name, out)
        assert s["adrf_rmse"] <= 0.05 and s["adrf_mape"] <= 0.02, (name, s["adrf_rmse"], s["adrf_mape"])
    assert -0.006 <= np.median([s["adrf_rmse"] for s in main.values()]) - pub["adrf_rmse"] <= rmse_above
    assert -0.003 <= np.median([s["adrf_mape"] for s in main.values()]) - pub["adrf_mape"] <= mape_above
    for name, s in other.items():          # the second optimum (or a run between the two) announces itself in the log
        assert s["egm_late_med_l2_loss_z"] > 0.33 and s["eval_mean_mse_v"] > 0.975 and s["acceptance"] > 0.108, (name, s)
    return main, other


def test_round3_product_runs_against_published_trace():
    """Eight seeds of CausalBGM(use_bnn=True) on the round-3 kernels (bnf_* sampling kernels, replayed latent Adam; logs:
    profiles/r03_accuracy/, one gpurun call of scripts/accuracy_runs.py per seed)."""
    pub = _published()
    stats = {"hip_s%d" % sd: _log_stats(open(os.path.join(ROOT, "profiles", "r03_accuracy", "bnn_s%d.log" % sd)).read()) for sd in SEEDS}
    main, other = _check_family(stats, pub, min_main=6)
    assert sorted(other) == ["hip_s2026", "hip_s99"]


def test_oracle_runs_against_published_trace():
    """The CHECKER anchored to reference-held numbers: oracle/ (bnn.py, egm.py, fit.py) run end to end on the tutorial by
    scripts/oracle_tutorial.py -- NumPy float32 on the CPU, nothing of the HIP library -- must land where the reference's published
    run landed, on the same 18 statistics and inside the same TIGHT envelope as the product.  (MH chains and the dose-response
    curve of the oracle runs use the first 5000 of the 20000 rows: the row average carries ~0.007 more Monte-Carlo error per dose
    than the published one, inside the distributional bound.)"""
    pub = _published()
    d = os.path.join(ROOT, "profiles", "r03_oracle_anchor")
    stats = {}
    for sd in SEEDS:
        text = open(os.path.join(d, "oracle_s%d.log" % sd)).read()
        egm, mb, ev, res = parse_log(text)
        if res is None:
            continue                                                              # (a run that was cut short: no RESULT line)
        assert res.get("oracle") is True, sd
        assert len(egm) == 61 and len(mb) == 101 and len(ev) == 11, sd            # the reference's logging cadence
        stats["oracle_s%d" % sd] = _log_stats(text)
    assert len(stats) >= ORACLE_RUNS_COMMITTED, sorted(stats)
    main, other = _check_family(stats, pub, min_main=(5 * len(stats) + 7) // 8, rmse_above=0.012, mape_above=0.006)
    assert sorted(other) == ["oracle_s1", "oracle_s11", "oracle_s123"] and len(main) == 5
    # the two implementations agree with each other as families: medians of the main-mode runs, statistic by statistic
    hip = {"hip_s%d" % sd: _log_stats(open(os.path.join(ROOT, "profiles", "r03_accuracy", "bnn_s%d.log" % sd)).read()) for sd in SEEDS}
    hip_main = [s for s in hip.values() if _main_mode(s, pub)]
    for k, tol in TIGHT.items():
        a, b = np.median([s[k] for s in main.values()]), np.median([s[k] for s in hip_main])
        assert abs(a - b) <= 1.5 * tol, (k, a, b)          # (each within tol of the published value)


@pytest.mark.gpu
def test_tutorial_run_reproduces_published_trace(capsys):
    from bayesgm_amd.models import CausalBGM
    from bayesgm_amd.datasets import Sim_Hirano_Imbens_sampler
    from bayesgm_amd.utils import get_ADRF
    x, y, v = Sim_Hirano_Imbens_sampler(N=20000, v_dim=200, seed=0).load_all()
    params = dict(dataset="Sim_Hirano_Imbens", output_dir="gpurun_out/tut", save_res=False, save_model=False, binary_treatment=False,
                  use_bnn=True, z_dims=[1, 1, 1, 7], v_dim=200, lr_theta=1e-4, lr_z=1e-4, g_units=[64] * 5, f_units=[64, 32, 8],
                  h_units=[64, 32, 8], e_units=[64] * 5, dz_units=[64, 32, 8], kl_weight=1e-4, lr=2e-4, g_d_freq=5, use_z_rec=True)
    model = CausalBGM(params, random_seed=123)
    model.fit((x, y, v), epochs=100, epochs_per_eval=10, use_egm_init=True, egm_n_iter=30000, egm_batches_per_eval=500, verbose=1)
    xs = np.linspace(0, 3, 20)
    adrf, interval = model.predict((x, y, v), alpha=0.01, n_mcmc=3000, burn_in=5000, x_values=xs, q_sd=1.0, bs=20000)
    truth = get_ADRF(x_values=list(xs), dataset="Imbens")
    text = capsys.readouterr().out
    egm, mb, ev, _ = parse_log(text)
    assert len(egm) == 61 and len(mb) == 101 and len(ev) == 11           # the reference's logging cadence
    s = summary(egm, mb, ev)
    s.update(acceptance=model.last_acceptance_rate, adrf_rmse=float(np.sqrt(np.mean((adrf - truth) ** 2))),
             adrf_mape=float(np.mean(np.abs((adrf - truth) / truth))))
    pub = _published()
    print({k: round(v, 4) for k, v in s.items()})
    assert _main_mode(s, pub)                               # seed 123 ends in the published optimum (see the round-3 notes above)
    out = {k: (s[k], pub[k]) for k, tol in TIGHT.items() if abs(s[k] - pub[k]) > tol}
    assert not out, out
    assert s["adrf_rmse"] <= 0.05 and s["adrf_mape"] <= 0.02
    assert abs(float(np.mean(adrf)) - float(np.mean(truth))) <= 0.01      # |average effect error| over the dose grid


def test_second_optimum_warning_on_the_committed_runs():
    """bayesgm_amd/diagnostics.py: the one warnings.warn of CausalBGM.fit.  Replayed on the committed logs exactly as the class sees
    them -- the EGM log lines give the late l2_loss_z, every fit evaluation gives a panel MSE_v -- it fires at the FIRST evaluation of
    the five runs that ended in the second optimum (product seeds 2026 and 99 -- the latter the run with ADRF RMSE 0.53 -- and oracle
    seeds 1, 11, 123) and on no evaluation of the eleven runs that reproduce the published trace."""
    import warnings
    from bayesgm_amd import diagnostics as D
    runs = [("hip_s%d" % sd, os.path.join(ROOT, "profiles", "r03_accuracy", "bnn_s%d.log" % sd)) for sd in SEEDS]
    runs += [("oracle_s%d" % sd, os.path.join(ROOT, "profiles", "r03_oracle_anchor", "oracle_s%d.log" % sd)) for sd in SEEDS]
    fired = {}
    for name, path in runs:
        egm, _, ev, _ = parse_log(open(path, errors="replace").read())
        l2z = D.late_l2_loss_z(egm[:, 0], egm[:, 3], 30000)
        msgs = [D.second_optimum_message(l2z, mv) for mv in ev[:, 3]]
        fired[name] = [m is not None for m in msgs]
        if any(fired[name]):
            assert fired[name][0] and "another random_seed" in msgs[0], name        # at the first evaluation, naming the remedy
    assert sorted(k for k, v in fired.items() if any(v)) == ["hip_s2026", "hip_s99", "oracle_s1", "oracle_s11", "oracle_s123"]
    with pytest.warns(D.SecondOptimumWarning, match="second optimum"):
        assert D.warn_if_second_optimum(0.436, 0.9985) is True                      # product seed 99
    with warnings.catch_warnings():
        warnings.simplefilter("error")
        assert D.warn_if_second_optimum(0.248, 0.9744) is False                     # product seed 123 (published: 0.247 / 0.9665)
        assert D.warn_if_second_optimum(0.436, 0.9985, already=True) is True        # once per fit
        assert D.warn_if_second_optimum(None, 0.99) is False                        # no warm start in this process
        # unstandardised covariates: the statistic is MSE_v / var(V): a panel with var(V) = 4 and MSE_v 3.2 explains 20 %
        assert D.warn_if_second_optimum(0.436, 3.2, v_var=4.0) is False
        assert D.warn_if_second_optimum(0.436, 0.99, v_var=0.0) is False            # nothing to explain: not diagnosed
    with pytest.warns(D.SecondOptimumWarning):
        assert D.warn_if_second_optimum(0.436, 3.95, v_var=4.0) is True
