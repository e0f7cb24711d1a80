"""The reference's published tutorial run (docs/source/causalbgm/tutorial_py.ipynb: Hirano-Imbens N = 20000, p = 200, use_bnn = True,
30000 EGM iterations + 100 epochs at B = 32, predict 5000 + 3000 MH transitions at 20 doses) executed END TO END ON THE ORACLE
(oracle/bnn.py, oracle/egm.py, oracle/fit.py: NumPy float32 on the CPU, no HIP library, no GPU) -- the anchor of the CHECKER, not of the
product: tests/test_tutorial_trace.py asserts that the statistics of the committed oracle logs (profiles/r03_oracle_anchor/) sit in
the envelope of the numbers the reference published.  Build container only (takes ~1 h per run on 4 host threads).

Prints the log lines the reference prints (EGM every 500 iterations; last-minibatch losses per epoch; panel MSEs every 10 epochs) and
one RESULT line, in the format scripts/compare_trace.py parses.

usage: python scripts/oracle_tutorial.py NAME [seed=123] [N=20000] [egm=30000] [epochs=100] [burn_in=5000] [n_mcmc=3000] [mh_rows=0]
  mh_rows > 0: the MH chains / dose-response curve run on the first mh_rows rows only (one block; the published run uses all N).
A run takes hours: its state is pickled to gpurun_out/oracle_ckpt/NAME.pkl after the EGM phase, every 10 epochs and every 250 MH
iterations, and a restart with the same NAME resumes there (append the output to the same log: `>> log`)."""
import json
import os
import pickle
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import bnn as OB          # noqa: E402
from oracle import egm as OE          # noqa: E402
from oracle import fit as OF          # noqa: E402
from bayesgm_amd.datasets import Sim_Hirano_Imbens_sampler      # noqa: E402  (NumPy only; pinned bit-exactly against the reference)
from bayesgm_amd.utils import get_ADRF                          # noqa: E402

name = sys.argv[1]
kv = dict(a.split("=", 1) for a in sys.argv[2:])
N = int(float(kv.get("N", 20000)))
seed = int(kv.get("seed", 123))
EGM_IT, EPOCHS = int(kv.get("egm", 30000)), int(kv.get("epochs", 100))
BURN, KEEP = int(kv.get("burn_in", 5000)), int(kv.get("n_mcmc", 3000))
MH_ROWS = int(kv.get("mh_rows", 0)) or N
B, P, ZD = 32, 200, [1, 1, 1, 7]
LR, LR_THETA, LR_Z, KLW, GD = 2e-4, 1e-4, 1e-4, 1e-4, 5
f32 = np.float32

x, y, v = Sim_Hirano_Imbens_sampler(N=N, v_dim=P, seed=0).load_all()
rs = np.random.RandomState(seed)
m = OB.init_model(seed, ZD, P, False)
for k in ("g", "e", "f", "h"):
    m[k]["norm"] = "fixed"                      # params['bnn_norm'] = "fixed": the shipped reading (DESIGN_HISTORY.md section 2b)
q = sum(ZD)
dz = OE.init_disc(rs, q, [64, 32, 8])
dz["fixed_norm"] = True                         # params['disc_norm'] = "fixed"
noise_key = (seed << 20) ^ 0x5DEECE66D
stream = [0]


def streams(n):
    s = stream[0]
    stream[0] += n
    return s


def net_params(net):
    return [net["gamma"], net["beta"]] + [a for L in net["layers"] for a in L]


def net_grads(g):
    return OB.flat_grads(g)


CKPT = os.path.join(ROOT, "gpurun_out", "oracle_ckpt", name + ".pkl")
os.makedirs(os.path.dirname(CKPT), exist_ok=True)


class ZState:
    pass


def save_ckpt(**state):
    state.update(m=m, stream=stream[0], rs=rs.get_state(), elapsed=time.time() - t0)
    with open(CKPT + ".tmp", "wb") as f:
        pickle.dump(state, f)
    os.replace(CKPT + ".tmp", CKPT)


ck = None
if os.path.exists(CKPT):
    with open(CKPT, "rb") as f:
        ck = pickle.load(f)
    m, stream[0] = ck["m"], ck["stream"]
    rs.set_state(ck["rs"])
    print("(resumed from %s: phase %s)" % (os.path.relpath(CKPT, ROOT), ck["phase"]), flush=True)
t0 = time.time() - (ck["elapsed"] if ck else 0.0)
# ---- EGM warm start (causalbgm/base.py:380-431)
if ck is None:
    print("EGM Initialization Starts ...")
g_opt = OE.Adam([a for k in ("g", "e", "f", "h") for a in net_params(m[k])], LR)
d_opt = OE.Adam(OE.disc_param_list(dz), LR)
for it in range(EGM_IT + 1 if ck is None else 0):
    for _ in range(GD):
        idx = rs.choice(N, B, replace=False)
        bz = rs.normal(0, 1, (B, q)).astype(f32)
        eps = f32(rs.uniform(0.0, 1.0))
        dz_loss, d_loss, gr = OB.egm_disc_step_grads(m, dz, bz, v[idx], eps, OB.egm_noises(m, B, noise_key, streams(1), disc_only=True))
        d_opt.step(OE.disc_param_list(gr))
    bz = rs.normal(0, 1, (B, q)).astype(f32)
    idx = rs.choice(N, B, replace=False)
    lg, gr = OB.egm_gen_step_grads(m, dz, True, bz, v[idx], x[idx], y[idx], OB.egm_noises(m, B, noise_key, streams(9)))
    g_opt.step([a for k in ("g", "e", "f", "h") for a in net_grads(gr[k])])
    if it % 500 == 0:
        print("EGM Initialization Iter [%d] : e_loss_adv [%.4f], l2_loss_v [%.4f], l2_loss_z [%.4f], l2_loss_x [%.4f], l2_loss_y [%.4f], "
              "g_e_loss [%.4f], dz_loss [%.4f], d_loss [%.4f]" % (it, lg[0], lg[1], lg[2], lg[3], lg[4], lg[5], dz_loss, d_loss), flush=True)
if ck is None:
    print("EGM Initialization Ends.  (%.0f s)" % (time.time() - t0))

# ---- iterative updates (base.py:434-532)
if ck is None:
    data_z, _, _, _, _ = OB.evaluate(m, (x, y, v), None, [0.0], noise_key, streams(1))   # base.py:479: Z = e(V), one noisy call on the panel
    data_z = data_z.astype(f32)
    opt = {k: OF.AdamState(net_params(m[k])) for k in ("g", "h", "f")}
    zst = ZState()
    zst.data_z, zst.zm, zst.zv, zst.zt, zst.lr_z = data_z, np.zeros_like(data_z), np.zeros_like(data_z), 0, LR_Z
    first_epoch = 0
    save_ckpt(phase="fit", opt=opt, zst=zst, next_epoch=0)
    print("Iterative Updating Starts ...")
else:
    opt, zst = ck.get("opt"), ck.get("zst")
    first_epoch = ck["next_epoch"] if ck["phase"] == "fit" else EPOCHS + 1
t1 = time.time()
for epoch in range(first_epoch, EPOCHS + 1):
    perm = rs.choice(N, N, replace=False)
    for i in range(0, N, B):
        idx = perm[i:i + B]
        bz, bx, by, bv = zst.data_z[idx], x[idx], y[idx], v[idx]
        s0 = streams(3)
        out = {}
        for nm in ("g", "h", "f"):                                    # update_g_net, update_h_net, update_f_net (:495-497)
            noise = OB.draw_noise(OB.net_dims(m[nm]), len(idx), noise_key, s0, OB.NET_ID[nm])
            loss, aux, gr = OB.theta_step(m, nm, bz, bx, by, bv, noise, KLW)
            opt[nm].apply(net_params(m[nm]), net_grads(gr), LR_THETA)
            out[nm] = (loss, aux)
        noises = {nm: tuple(OB.draw_noise(OB.net_dims(m[nm]), len(idx), noise_key, s0 + 1 + c, OB.NET_ID[nm]) for c in (0, 1))
                  for nm in ("g", "h", "f")}
        lz, dzv = OB.z_step(m, zst.data_z[idx], bx, by, bv, noises)   # update_latent_variable_sgd (:499)
        OF.adam_rows(zst, idx, dzv.astype(f32), False)
    print("Epoch [%d/%d]: loss_px_z [%.4f], loss_mse_x [%.4f], loss_py_z [%.4f], loss_mse_y [%.4f], loss_pv_z [%.4f], loss_mse_v [%.4f], "
          "loss_postrior_z [%.4f]" % (epoch, EPOCHS, out["h"][0], out["h"][1], out["f"][0], out["f"][1], out["g"][0], out["g"][1], lz), flush=True)
    if epoch % 10 == 0:
        _, _, mse_x, mse_y, mse_v = OB.evaluate(m, (x, y, v), zst.data_z, np.linspace(0.0, 3.0, 5), noise_key, streams(8))
        print("Epoch [%d/%d]: MSE_x: %.4f, MSE_y: %.4f, MSE_v: %.4f\n" % (epoch, EPOCHS, mse_x, mse_y, mse_v), flush=True)
        save_ckpt(phase="fit", opt=opt, zst=zst, next_epoch=epoch + 1)
if ck is None or ck["phase"] == "fit":
    t_fit = time.time() - t0
    print("fit %.0f s (iterative part %.0f s of this process)" % (t_fit, time.time() - t1))
else:
    t_fit = ck["t_fit"]

# ---- predict (base.py:573-668): one block of MH_ROWS rows, q_sd = 1
t2 = time.time()
xs = np.linspace(0, 3, 20).astype(f32)
truth = get_ADRF(x_values=list(xs), dataset="Imbens")
n = MH_ROWS
xm, ym, vm = x[:n], y[:n], v[:n]
pseed = seed + 17
z = OB.R.normals(np.arange(n), 0, q, OB.R.TAG_INIT, pseed).astype(f32)
acc_tail, adrf_draws, first_it, t_pred0 = 0, np.zeros((len(xs), KEEP)), 0, 0.0
if ck is not None and ck["phase"] == "mh":
    z, acc_tail, adrf_draws, first_it, t_pred0 = ck["z"], ck["acc_tail"], ck["adrf_draws"], ck["next_it"], ck["t_pred"]
t2 -= t_pred0
for it in range(first_it, BURN + KEEP):
    z, acc, _, _ = OB.mh_iteration(m, xm, ym, vm, z, it, 1.0, pseed, n)
    if it >= BURN + KEEP - 100:
        acc_tail += int(acc.sum())
    if it >= BURN:
        d = it - BURN
        adrf_draws[:, d] = OB.effects_draw(m, z, xs, d, it, True, pseed, n).mean(axis=1)
    if it % 500 == 0:
        print("MH iteration %d (%.0f s)" % (it, time.time() - t2), flush=True)
    if it % 250 == 249:
        save_ckpt(phase="mh", t_fit=t_fit, z=z, acc_tail=acc_tail, adrf_draws=adrf_draws, next_it=it + 1, t_pred=time.time() - t2)
adrf = adrf_draws.mean(axis=1)
acc_rate = acc_tail / (100.0 * n)
print("Final MCMC Acceptance Rate: %.4f" % acc_rate)
res = dict(name=name, args=kv, fit_s=t_fit, predict_s=time.time() - t2, adrf_rmse=float(np.sqrt(np.mean((adrf - truth) ** 2))),
           adrf_mape=float(np.mean(np.abs((adrf - truth) / truth))), acceptance=float(acc_rate), adrf=[float(a) for a in adrf],
           truth=[float(t) for t in truth], oracle=True)
print("RESULT " + json.dumps(res))
