"""Timing of the minibatch steps of CausalBGM(use_bnn=True) outside the default widths (the general Flipout forward / backward of
csrc/bnn_kernels.h): theta step, latent step and the epoch call, B = 32, on N rows.
usage: python scripts/probe_bnn_fit_wide.py [N=20000] [reps=100]"""
import sys, time
import numpy as np
import torch
sys.path.insert(0, ".")
from bayesgm_amd.models import CausalBGM

n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 20000
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 100
p, z_dims, batch = 200, [1, 1, 1, 7], 32
q = sum(z_dims)
dev = torch.device("cuda", 0)
g = torch.Generator(device=dev).manual_seed(3)
xs = torch.rand(n, device=dev, generator=g); ys = torch.randn(n, device=dev, generator=g); vs = torch.randn(n, p, device=dev, generator=g)
idx = torch.randperm(n, device=dev, generator=g)[:batch].to(torch.int32)
perm = torch.randperm(n, device=dev, generator=g).to(torch.int32)[:(n // batch) * batch]


def timed(fn):
    for _ in range(5):
        fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return 1e6 * (time.perf_counter() - t0) / reps


for name, u in (("[64] x 5 (default)", {}), ("[128, 128]", dict(g_units=[128, 128], e_units=[128, 128], f_units=[128, 128], h_units=[128, 128])),
                ("[256] x 3", dict(g_units=[256] * 3, e_units=[256] * 3, f_units=[256] * 3, h_units=[256] * 3))):
    params = dict(dataset="probe", output_dir=".", save_res=False, save_model=False, binary_treatment=False, use_bnn=True, z_dims=z_dims, v_dim=p,
                  lr_theta=1e-4, lr_z=1e-4, g_units=[64] * 5, f_units=[64, 32, 8], h_units=[64, 32, 8], kl_weight=1e-4, lr=2e-4, g_d_freq=5,
                  use_z_rec=True, e_units=[64] * 5, dz_units=[64, 32, 8])
    params.update(u)
    m = CausalBGM(params, timestamp="probe", random_seed=0, device=0)
    be = m.engine
    z = torch.randn(n, q, device=dev, generator=g); zm, zv = torch.zeros_like(z), torch.zeros_like(z)
    t_us = timed(lambda: be.theta_step(z, idx, xs, ys, vs, 1e-4, 1, 0))
    def latent():
        be.z_sync(z, zm, zv, idx, 1e-4)
        be.z_step(xs, ys, vs, z, zm, zv, idx, 1e-4, 1, 1, lazy=2)
    l_us = timed(latent)
    be.z_sync(z, zm, zv, None, 1e-4)
    nb = min(len(perm) // batch, 400)
    be.fit_epoch(xs, ys, vs, z, zm, zv, perm[:20 * batch], batch, 1e-4, 1e-4, 2, 1, 0)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    be.fit_epoch(xs, ys, vs, z, zm, zv, perm[:nb * batch], batch, 1e-4, 1e-4, 2, 1, 0)
    torch.cuda.synchronize()
    e_us = 1e6 * (time.perf_counter() - t0) / nb
    print("use_bnn=True %s, B=%d: theta step %.1f us, latent step %.1f us, epoch call %.1f us per minibatch" % (name, batch, t_us, l_us, e_us), flush=True)

# the EGM warm start's two steps at the same widths (general kernels of csrc/bnn_egm_kernels.h)
if len(sys.argv) > 3 and sys.argv[3] == "egm":
    rs = np.random.RandomState(5)
    dims = [q, 64, 32, 8, 1]
    dz = {"W": [(rs.uniform(-1, 1, (dims[i], dims[i + 1])) * np.sqrt(6.0 / (dims[i] + dims[i + 1]))).astype(np.float32) for i in range(len(dims) - 1)],
          "b": [np.zeros(d, np.float32) for d in dims[1:]], "gamma": [np.ones(d, np.float32) for d in dims[1:-1]], "beta": [np.zeros(d, np.float32) for d in dims[1:-1]]}
    zb = torch.randn(batch, q, device=dev, generator=g)
    for name, u in (("[128, 128]", dict(g_units=[128, 128], e_units=[128, 128], f_units=[128, 128], h_units=[128, 128])),
                    ("[256] x 3", dict(g_units=[256] * 3, e_units=[256] * 3, f_units=[256] * 3, h_units=[256] * 3))):
        params = dict(dataset="probe", output_dir=".", save_res=False, save_model=False, binary_treatment=False, use_bnn=True, z_dims=z_dims, v_dim=p,
                      lr_theta=1e-4, lr_z=1e-4, kl_weight=1e-4, lr=2e-4, g_d_freq=5, use_z_rec=True, dz_units=[64, 32, 8])
        params.update(u)
        be = CausalBGM(params, timestamp="probe", random_seed=0, device=0).engine
        be.set_disc_norm("fixed")
        be.egm_begin(dz, batch, 2e-4, 1)
        d_us = timed(lambda: be.egm_disc_step(zb, idx, vs, 0.5, 1, 0))
        g_us = timed(lambda: be.egm_gen_step(zb, idx, vs, xs, ys, 1, 1))
        be.egm_end()
        print("use_bnn=True %s EGM, B=%d: discriminator step %.1f us, generator step %.1f us, iteration (5 + 1) %.2f ms" % (name, batch, d_us, g_us, 1e-3 * (5 * d_us + g_us)), flush=True)
