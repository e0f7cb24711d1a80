"""Run-time diagnostics of CausalBGM.fit: the second optimum of the EGM warm start.

Sixteen end-to-end runs of the published tutorial setting (eight of the product, eight of the NumPy oracle: DESIGN_HISTORY.md section 2c,
profiles/r03_accuracy/, profiles/r03_oracle_anchor/) end in one of two places.  Eleven land where the reference's published run
landed: late `l2_loss_z` 0.22 - 0.25 (published 0.247), panel `MSE_v` 0.964 - 0.976, MH acceptance 0.097 - 0.101.  Five land in a
second optimum of the warm start -- late `l2_loss_z` 0.35 - 0.44, `MSE_v` 0.975 - 0.999 at EVERY evaluation, acceptance 0.11 - 0.12
-- in which the dose-response curve can be anything (ADRF RMSE 0.016 - 0.030 for four of them, 0.53 for product seed 99: a curve
shifted by +0.65).  Both statistics are printed by `fit` (the reference prints the same lines); this module turns them into ONE
`warnings.warn` so that a run of the drop-in does not return such a curve silently.  The reference itself gives no warning.
"""
import warnings

import numpy as np

MIN_EGM_ITER = 10000        # warm starts shorter than this are not diagnosed (the sixteen runs used the reference's 30 000)
L2Z_LATE_MAX = 0.33     # median l2_loss_z of the EGM log lines of the last third of the warm start: main optimum <= 0.253, second >= 0.351
MSE_V_MAX = 0.975       # panel MSE_v of a fit evaluation (standardised V): second optimum >= 0.9753 at every evaluation


class SecondOptimumWarning(RuntimeWarning):
    pass


def late_l2_loss_z(iters, values, n_iter):
    """Median of the logged l2_loss_z over the last third of the warm start (iterations >= 2/3 n_iter); nan without such lines."""
    it = np.asarray(iters, float)
    v = np.asarray(values, float)
    sel = it >= (2.0 / 3.0) * float(n_iter)
    return float(np.median(v[sel])) if sel.any() else float("nan")


def second_optimum_message(l2z_late, mse_v, v_var=1.0):
    """The warning text when BOTH symptoms are present, else None.  `v_var`: mean per-column variance of the panel's V -- the threshold
    on MSE_v was calibrated on standardised covariates (variance 1), so the statistic compared is MSE_v / v_var, the fraction of V's
    variance the generator leaves unexplained; a panel whose V carries no usable variance (v_var <= 0) is not diagnosed."""
    if l2z_late is None or not np.isfinite(l2z_late) or mse_v is None or not np.isfinite(mse_v):
        return None
    if v_var is None or not np.isfinite(v_var) or v_var <= 0:
        return None
    mse_v = mse_v / v_var
    if l2z_late > L2Z_LATE_MAX and mse_v > MSE_V_MAX:
        return ("CausalBGM.fit: the EGM warm start may have ended in its second optimum (late l2_loss_z %.3f > %.2f and panel "
                "MSE_v / var(V) %.4f > %.3f; thresholds calibrated on sixteen runs of the reference's Hirano-Imbens tutorial, where runs "
                "that reproduce the published trace show l2_loss_z <= 0.25 and MSE_v <= 0.976 -- on data with weakly informative "
                "covariates a high MSE_v alone is normal).  On the tutorial data the estimated dose-response curve / treatment effects "
                "were unreliable in this state (one of five such runs returned an ADRF shifted by +0.65).  Remedy: re-run with another "
                "random_seed (or a longer egm_n_iter) and compare the same two log lines; params['second_optimum_check'] = False "
                "silences this check." % (l2z_late, L2Z_LATE_MAX, mse_v, MSE_V_MAX))
    return None


def warn_if_second_optimum(l2z_late, mse_v, already=False, v_var=1.0):
    """Emit the warning once; returns True when it was (or had been) emitted."""
    if already:
        return True
    msg = second_optimum_message(l2z_late, mse_v, v_var)
    if msg is None:
        return False
    warnings.warn(msg, SecondOptimumWarning, stacklevel=3)
    return True


_NOTICED = set()


def notice_once(key, message):
    """One line on stderr, once per process: a default of this build that differs from the reference as written was taken implicitly."""
    import sys
    if key in _NOTICED:
        return
    _NOTICED.add(key)
    print("bayesgm_amd: " + message, file=sys.stderr)
