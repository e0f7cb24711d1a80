import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import numpy as np, torch
from tests.test_gpu_fit import _setup, _engine
n, B, lr, steps = 4096, 32, 1e-3, 400
m, x, y, v, z = _setup([1, 1, 1, 7], 200, False, n, 21)
rs = np.random.RandomState(6)
order = [rs.choice(n, B, replace=False).astype(np.int32) for _ in range(steps)]
st = {}
for mode in (0, 2):
    eng = _engine(m); dev = eng.device
    xd, yd, vd, zd = (torch.from_numpy(a).to(dev) for a in (x.ravel(), y.ravel(), v, z.copy()))
    zm = torch.zeros_like(zd); zv = torch.zeros_like(zd)
    npar = eng.fit_begin(n, B); grad = torch.empty(npar, device=dev)
    st[mode] = dict(eng=eng, xd=xd, yd=yd, vd=vd, zd=zd, zm=zm, zv=zv, grad=grad)
for k, idx_np in enumerate(order):
    for mode in (0, 2):
        s = st[mode]; eng = s["eng"]
        idx = torch.from_numpy(idx_np).cuda()
        if mode == 2:
            eng.fit_z_sync(s["zd"], s["zm"], s["zv"], idx, lr)
        eng.fit_theta_grad(s["xd"], s["yd"], s["vd"], s["zd"], idx, B, s["grad"])
        eng.fit_theta_apply(s["grad"], lr)
        eng.fit_z_step(s["xd"], s["yd"], s["vd"], s["zd"], s["zm"], s["zv"], idx, B, lr, lazy=mode)
    # compare after flushing a COPY of mode 2? compare batch rows only (current in both)
    a = st[0]["zd"][idx_np.astype(np.int64)].cpu().numpy(); b = st[2]["zd"][idx_np.astype(np.int64)].cpu().numpy()
    ma = st[0]["zm"][idx_np.astype(np.int64)].cpu().numpy(); mb = st[2]["zm"][idx_np.astype(np.int64)].cpu().numpy()
    ga = (st[0]["grad"] - st[2]["grad"]).abs().max().item()
    d = np.abs(a - b).max()
    if k < 12 or d > 1e-5:
        print(k, "z diff", d, "m diff", np.abs(ma - mb).max(), "grad diff", ga, "max|grad|", st[0]["grad"].abs().max().item())
    if d > 1e-4:
        r = np.unravel_index(np.abs(a - b).argmax(), a.shape)
        row = idx_np[r[0]]
        print("row", row, "feature", r[1], "uses", [j for j, o in enumerate(order[:k + 1]) if row in o], a[r], b[r], ma[r], mb[r])
        break
