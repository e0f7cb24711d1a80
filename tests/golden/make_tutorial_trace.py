"""Extract the numbers the reference itself published for the hot path (run in the build container only).

    python tests/golden/make_tutorial_trace.py

The only outputs of the reference's TensorFlow path that exist anywhere in /root/reference are the cell outputs of
docs/source/causalbgm/tutorial_py.ipynb (package v1.0.1, use_bnn=True, Hirano-Imbens N=20000 p=200, default fit,
predict(n_mcmc=3000, burn_in=5000, x_values=linspace(0,3,20), q_sd=1.0, bs=20000)): the EGM log every 500 iterations,
the last-minibatch losses of the 101 epochs, the panel MSEs every 10 epochs, the final MH acceptance rate and the ADRF
RMSE / MAPE.  They are written to tutorial_trace.json -- numbers only -- and used as a statistical envelope for the
build's own end-to-end runs (tests/test_tutorial_trace.py, scripts/compare_trace.py, DESIGN_HISTORY.md section 7)."""
import json
import os
import re

NB = "/root/reference/docs/source/causalbgm/tutorial_py.ipynb"
HERE = os.path.dirname(os.path.abspath(__file__))


def cell_text(cell):
    return "".join("".join(o.get("text", [])) for o in cell.get("outputs", []))


def main():
    nb = json.load(open(NB))
    code = [c for c in nb["cells"] if c["cell_type"] == "code"]
    fit = next(c for c in code if "model.fit(" in "".join(c["source"]) and "Hirano" not in "".join(c["source"])
               and "EGM Initialization Iter" in cell_text(c))
    t = cell_text(fit)
    num = r"\[([-\d.]+)\]"
    egm = re.findall(r"EGM Initialization Iter \[(\d+)\] : e_loss_adv %s, l2_loss_v %s, l2_loss_z %s, l2_loss_x %s, "
                     r"l2_loss_y %s, g_e_loss %s, dz_loss %s, d_loss %s" % ((num,) * 8), t)
    mb = re.findall(r"Epoch (\d+)/100: 100%%.*?loss_px_z: %s, loss_mse_x: %s, loss_py_z: %s, loss_mse_y: %s, loss_pv_z: %s, "
                    r"loss_mse_v: %s, loss_postrior_z: %s" % ((num,) * 7), t)
    ev = re.findall(r"Epoch \[(\d+)/100\]: MSE_x: ([-\d.]+), MSE_y: ([-\d.]+), MSE_v: ([-\d.]+)", t)
    pred = next(c for c in code if "model.predict(" in "".join(c["source"]) and "Final MCMC Acceptance Rate" in cell_text(c))
    acc = float(re.search(r"Final MCMC Acceptance Rate: ([\d.]+)", cell_text(pred)).group(1))
    res = next(c for c in code if "RMSE (Root Mean Squared Error)" in cell_text(c))
    rmse = float(re.search(r"RMSE \(Root Mean Squared Error\): ([\d.]+)", cell_text(res)).group(1))
    mape = float(re.search(r"MAPE \(Mean Absolute Percentage Error\): ([\d.]+)", cell_text(res)).group(1))
    ver = re.search(r"Currently use version (\S+) of bayesgm", cell_text(code[0])).group(1)
    out = dict(
        source="docs/source/causalbgm/tutorial_py.ipynb cell outputs (continuous-treatment section)",
        package_version=ver, N=20000, v_dim=200, use_bnn=True,
        egm_columns=["iter", "e_loss_adv", "l2_loss_v", "l2_loss_z", "l2_loss_x", "l2_loss_y", "g_e_loss", "dz_loss", "d_loss"],
        egm=[[int(r[0])] + [float(a) for a in r[1:]] for r in egm],
        minibatch_columns=["epoch", "loss_px_z", "loss_mse_x", "loss_py_z", "loss_mse_y", "loss_pv_z", "loss_mse_v", "loss_postrior_z"],
        minibatch=[[int(r[0])] + [float(a) for a in r[1:]] for r in mb],
        eval_columns=["epoch", "mse_x", "mse_y", "mse_v"],
        eval=[[int(r[0])] + [float(a) for a in r[1:]] for r in ev],
        acceptance_rate=acc, adrf_rmse=rmse, adrf_mape=mape)
    assert len(out["egm"]) == 61 and len(out["minibatch"]) == 101 and len(out["eval"]) == 11, \
        (len(out["egm"]), len(out["minibatch"]), len(out["eval"]))
    with open(os.path.join(HERE, "tutorial_trace.json"), "w") as f:
        json.dump(out, f, indent=0)
    print("wrote tutorial_trace.json:", len(out["egm"]), "EGM lines,", len(out["minibatch"]), "minibatch lines,",
          len(out["eval"]), "evaluation lines; acceptance", acc, "rmse", rmse, "mape", mape, "version", ver)


if __name__ == "__main__":
    main()
