"""Lay the log of one end-to-end run (scripts/accuracy_runs.py) next to the trace the reference published
(tests/golden/tutorial_trace.json): window statistics of the EGM log, of the per-epoch minibatch losses, the panel MSEs,
the MH acceptance rate and the ADRF error.  usage: python scripts/compare_trace.py LOG [LOG ...]"""
import json
import os
import re
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
NUM = r"\[([-\d.]+)\]"


def parse_log(text):
    egm = re.findall(r"EGM Initialization Iter \[(\d+)\] : e_loss_adv %s, l2_loss_v %s, l2_loss_z %s, l2_loss_x %s, "
                     r"l2_loss_y %s, g_e_loss %s, dz_loss %s, d_loss %s" % ((NUM,) * 8), text)
    mb = re.findall(r"Epoch \[(\d+)/\d+\]: loss_px_z %s, loss_mse_x %s, loss_py_z %s, loss_mse_y %s, loss_pv_z %s, "
                    r"loss_mse_v %s, loss_postrior_z %s" % ((NUM,) * 7), text)
    ev = re.findall(r"Epoch \[(\d+)/\d+\]: MSE_x: ([-\d.]+), MSE_y: ([-\d.]+), MSE_v: ([-\d.]+)", text)
    res = re.search(r"RESULT (\{.*\})", text)

    def table(rows, width):      # a run resumed from a checkpoint (scripts/oracle_tutorial.py) prints some lines twice: keep the last
        a = np.array(rows, float).reshape(-1, width)
        keep = {int(k): i for i, k in enumerate(a[:, 0])}
        return a[sorted(keep.values())]
    return table(egm, 9), table(mb, 8), table(ev, 4), json.loads(res.group(1)) if res else None


def summary(egm, mb, ev):
    """Seed-robust statistics: medians over windows of the noisy per-batch lines, means of the panel MSEs."""
    s = {}
    if len(egm):
        late = egm[egm[:, 0] >= 20000]
        early = egm[(egm[:, 0] >= 1000) & (egm[:, 0] <= 5000)]
        for j, k in enumerate(["e_loss_adv", "l2_loss_v", "l2_loss_z", "l2_loss_x", "l2_loss_y", "g_e_loss", "dz_loss"], 1):
            s["egm_late_med_" + k] = float(np.median(late[:, j])) if len(late) else float("nan")
        s["egm_early_med_l2_loss_z"] = float(np.median(early[:, 3])) if len(early) else float("nan")
        s["egm_late_med_gp"] = float(np.median((late[:, 8] - late[:, 7]) / 10.0)) if len(late) else float("nan")
        s["egm_iter0_gp"] = float((egm[0, 8] - egm[0, 7]) / 10.0)
    if len(mb):
        for j, k in enumerate(["loss_px_z", "loss_mse_x", "loss_py_z", "loss_mse_y", "loss_pv_z", "loss_mse_v", "loss_postrior_z"], 1):
            s["fit_mean_" + k] = float(np.mean(mb[:, j]))
        s["fit_last20_loss_py_z"] = float(np.mean(mb[-20:, 3]))
        s["fit_last20_loss_px_z"] = float(np.mean(mb[-20:, 1]))
    if len(ev):
        s["eval_mean_mse_x"], s["eval_mean_mse_y"], s["eval_mean_mse_v"] = (float(a) for a in ev[:, 1:].mean(0))
        s["eval_last_mse_y"] = float(ev[-1, 2])
    return s


def main():
    ref = json.load(open(os.path.join(HERE, "..", "tests", "golden", "tutorial_trace.json")))
    cols = [("published v%s" % ref["package_version"],
             dict(summary(np.array(ref["egm"], float), np.array(ref["minibatch"], float), np.array(ref["eval"], float)),
                  acceptance=ref["acceptance_rate"], adrf_rmse=ref["adrf_rmse"], adrf_mape=ref["adrf_mape"]))]
    for path in sys.argv[1:]:
        egm, mb, ev, res = parse_log(open(path, errors="replace").read())
        s = summary(egm, mb, ev)
        if res:
            s.update(acceptance=res["acceptance"], adrf_rmse=res["adrf_rmse"], adrf_mape=res["adrf_mape"])
        cols.append((os.path.basename(path).replace(".log", ""), s))
    keys = list(cols[0][1].keys())
    w = max(len(k) for k in keys)
    print(" " * w + "".join("%18s" % c[0][:17] for c in cols))
    for k in keys:
        print(k.ljust(w) + "".join("%18s" % ("%.4f" % c[1][k] if k in c[1] else "-") for c in cols))


if __name__ == "__main__":
    main()
