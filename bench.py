#!/usr/bin/env python
"""bench.py -- headline benchmark of the BGM / CausalBGM posterior-sampling hot path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

Metric (BASELINE.json): posterior samples/sec, whole job = MCMC transitions of per-observation
latent chains per second, CausalBGM continuous treatment, Sim_Hirano_Imbens panel N=1e6 (per GPU,
weak scaling), p=200, z_dims [1,1,1,7], burn_in=5000, n_mcmc=3000, q_sd=1, 20 doses on [0,3],
sample_y=True (causalbgm/base.py:573 defaults).  One "step" = one CausalBGM.predict over the whole
panel with the inputs already resident in HBM: burn_in + n_mcmc Metropolis-Hastings transitions of
every row, the fused dose-response inference for the retained draws, the slot reduction, (N>1) the
ADRF all-reduce, and the posterior mean / quantiles.

The JSON line also carries
  roofline     -- the dominant kernel instance by share of kernel time (at the BASELINE iteration counts that is the keep-phase
                  instance causal_mh_kernel<EFFECT=1>: one transition + the outcome net at 20 doses per retained draw; the
                  burn-in instance <EFFECT=0> is listed beside it under "instances"), fp32 MFMA: algorithmic FLOP/launch
                  (2*MACs(g+f+h) = 69,696 FLOP per row-transition, + 2*MACs(f) per row and dose for a retained draw,
                  SURVEY.md 8d) over its hipEvent-measured duration, against the 157.3 TF dense fp32-MFMA peak.
  cpu_baseline -- the oracle's literal restatement of the reference loop (NumPy, host RNG, two
                  log-posterior evaluations per iteration) timed on this box's host cores on a
                  bounded sample (bs=10000 rows x a few iterations).
  parity       -- (with the cpu_baseline leg) ADRF of the HIP path vs the oracle on 256 rows, same weights and
                  Philox streams: the "matches the reference within a stated fp32 tolerance" number of this run.
  encoder / config_c1 / config_c4_share / bayesian_nets / fit_dp -- (round 6) one driver-run number per BASELINE config and
                  north_star target: e(V) over the panel against the fp32-MFMA peak; configs[1] (binary treatment, ITE + intervals)
                  and one GPU's share of configs[4] (BGM imputation, fp32 and the opt-in split precision) through the classes;
                  the reference's default Bayesian nets at the BASELINE iteration counts with their own `roofline` instances; and,
                  at every N, `fit_dp`: the data-parallel minibatch loop with its per-step RCCL gradient all-reduce issued inside the
                  library (configs[3], fit side) -- at N > 1 after the headline is complete and under a watchdog.
Weights are glorot-uniform random (seed 0): throughput does not depend on their values.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_FP32_MFMA_TFLOPS = 157.3  # MI355X_MICROARCH.md "Peak FP32 (matrix)"


def plan_rows(n, world, rank, scaling):
    """(rows of this rank, first global row of this rank, rows of the whole job).  weak: every rank its own panel of n rows;
    strong (BASELINE configs[3]: ONE N-row panel over all GPUs): the contiguous share bayesgm_amd.parallel.shard_range gives the rank."""
    n = int(n)
    if scaling == "strong":
        base, rem = divmod(n, world)
        lo = rank * base + min(rank, rem)
        return base + (1 if rank < rem else 0), lo, n
    return n, rank * n, n * world


def make_panel(n, p, seed, device, lo=0, n_gen=None):
    """Hirano-Imbens panel, generated per rank on the host with the reference's generator
    restatement (bayesgm_amd.datasets, pinned by tests/golden) and moved to HBM."""
    import torch
    from bayesgm_amd.datasets import Sim_Hirano_Imbens_sampler
    x, y, v = Sim_Hirano_Imbens_sampler(N=n if n_gen is None else n_gen, v_dim=p, seed=seed).load_all()
    if n_gen is not None:          # strong scaling: rows [lo, lo + n) of the one n_gen-row panel every rank generates identically
        x, y, v = (np.ascontiguousarray(a[lo:lo + n]) for a in (x, y, v))
    return (torch.from_numpy(x).to(device), torch.from_numpy(y).to(device), torch.from_numpy(v).to(device))


def general_width_leg(p, z_dims, device):
    """The general-width engine (csrc/gx_api.hip: models whose hidden widths are not the compiled defaults) at two shapes it exists
    for, glorot weights, 20 burn-in transitions each: `[128, 128]` nets, and the bench's own default widths forced through it
    (BGM_FORCE_GX) next to the resident kernels' rate -- fp32 and the opt-in split precision ('f16x3', hidden widths <= 128)."""
    import torch
    from bayesgm_amd.engine import CausalEngine
    rs = np.random.RandomState(0)

    def net(dims):
        return [((rs.uniform(-1, 1, (a, b)) * np.sqrt(6.0 / (a + b))).astype(np.float32), np.zeros(b, np.float32)) for a, b in zip(dims[:-1], dims[1:])]
    q, (z0, z1, z2, _) = sum(z_dims), z_dims
    out = {"sample": "CausalEngine.mh_sample, 20 burn-in transitions, p=%d, z_dims %s, glorot weights; TFLOP/s = 2 MACs(g + f + h) per row-transition (algorithmic, also for f16x3: its fraction is a speed relative to the fp32 peak, not a utilisation of the fp16 pipe)" % (p, list(z_dims))}
    shapes = (("default_widths_forced_through_the_engine", dict(g_units=[64] * 5, e_units=[64] * 5, f_units=[64, 32, 8], h_units=[64, 32, 8]), 1000000, True),
              ("w128", dict(g_units=[128, 128], e_units=[128, 128], f_units=[128, 128], h_units=[128, 128]), 500000, False))
    for name, u, n, force in shapes:
        g = torch.Generator(device=device).manual_seed(0)
        v = torch.randn(n, p, device=device, generator=g); x = torch.rand(n, device=device, generator=g); y = torch.randn(n, device=device, generator=g)
        nets = dict(g=net([q] + u["g_units"] + [p + 1]), e=net([p] + u["e_units"] + [q]), f=net([z0 + z1 + 1] + u["f_units"] + [2]),
                    h=net([z0 + z2] + u["h_units"] + [2]))
        macs = sum(w.shape[0] * w.shape[1] for k in "gfh" for w, _ in nets[k])
        rec = {"rows": n, "hidden": {k: u[k] for k in ("g_units", "f_units", "h_units")}}
        try:
            if force:
                os.environ["BGM_FORCE_GX"] = "1"
            eng = CausalEngine(p, list(z_dims), **u)
            eng.set_model(**nets)
            for mode in ("fp32", "f16x3"):
                eng.set_precision(mode)
                eng.mh_sample(x, y, v, 2, 0, 1.0, 1)
                torch.cuda.synchronize(device); t0 = time.time()
                eng.mh_sample(x, y, v, 20, 0, 1.0, 1)
                torch.cuda.synchronize(device); dt = (time.time() - t0) / 20
                rec[mode] = {"ms_per_iteration": 1e3 * dt, "transitions_per_s": n / dt, "tflops_algorithmic": 2 * macs * n / dt / 1e12,
                             "frac_of_fp32_mfma_peak": 2 * macs * n / dt / 1e12 / 157.3}
            eng.close()
        finally:
            os.environ.pop("BGM_FORCE_GX", None)
        out[name] = rec
    return out


def cpu_baseline(params, p, z_dims, budget_s=20.0):
    """Reference loop as written, on the host cores (kind = "port")."""
    from oracle import causal as OC
    from bayesgm_amd.datasets import Sim_Hirano_Imbens_sampler
    try:
        import threadpoolctl
        cores = max(i["num_threads"] for i in threadpoolctl.threadpool_info()) if threadpoolctl.threadpool_info() else 1
    except Exception:
        cores = os.cpu_count() or 1
    m = OC.init_model(0, z_dims, p)
    bs = 10000
    x, y, v = Sim_Hirano_Imbens_sampler(N=bs, v_dim=p, seed=1).load_all()
    rng = np.random.RandomState(0)
    OC.mh_reference_loop(m, (x, y, v), 1, 1.0, rng)  # warm BLAS
    t0 = time.time()
    OC.mh_reference_loop(m, (x, y, v), 3, 1.0, rng)
    per_it = (time.time() - t0) / 3
    n_it = int(max(6, min(400, budget_s / max(per_it, 1e-6))))
    # two timed halves of the sample: `value` is the faster one (a host shared with other jobs only ever slows a half down), both are
    # reported -- the spread says how much of a round-to-round change of gpu_over_cpu is the host, not the GPU
    half, rates, dt = n_it // 2, [], 0.0
    for _ in range(2):
        t0 = time.time()
        OC.mh_reference_loop(m, (x, y, v), half, 1.0, rng)
        d = time.time() - t0
        dt += d
        rates.append(bs * half / d)
    return {"value": max(rates), "unit": "posterior samples/s (MH transitions/s)", "cores": int(cores),
            "kind": "port", "halves": rates,
            "sample": f"oracle.causal.mh_reference_loop: bs={bs} rows x 2 x {half} iterations (value = the faster half), p={p}, "
                      f"2 log-posterior evals/iter + NumPy RNG as causalbgm/base.py:860-871; {dt:.1f} s"}


def parity_leg(model, x, y, v, z_dims, p, x_values, rows=256, burn_in=300, n_keep=100):
    """ADRF of the HIP path against the oracle (the checker) on the first `rows` rows: same weights, same Philox streams."""
    from oracle import causal as OC
    sub = slice(0, rows)
    xs, ys, vs = x[sub].cpu().numpy(), y[sub].cpu().numpy(), v[sub].cpu().numpy()
    m = dict(g=model.nets["g"], f=model.nets["f"], h=model.nets["h"], e=model.nets["e"], z_dims=list(z_dims), v_dim=p,
             binary_treatment=False)
    seed = 20260928
    out = model.engine.mh_sample(xs.ravel(), ys.ravel(), vs, burn_in, n_keep, 1.0, seed, effect=1, x_values=x_values)
    post = OC.mh_sampler(m, (xs, ys, vs), burn_in, n_keep, 1.0, seed)
    ref = OC.infer_from_latent_posterior(m, post, x_values, True, seed, burn_in=burn_in)
    got = out["adrf"].cpu().numpy()
    return {"adrf_max_abs_diff_vs_oracle": float(np.abs(got.mean(axis=1) - ref.mean(axis=1)).max()),
            "adrf_draws_max_abs_diff_vs_oracle": float(np.abs(got - ref).max()),
            "sample": f"first {rows} rows, {burn_in} burn-in + {n_keep} retained transitions, {len(x_values)} doses, sample_y=True, "
                      "same weights and Philox streams on both sides (oracle/causal.py)"}


def fit_leg(model, x, y, v, n_loc, steps=2000, batch=32):
    """Secondary measurement (fit throughput reported separately): minibatch iterations of CausalBGM.fit at
    the reference's batch size on the bench panel -- theta gradient + Adam + latent step per iteration.  The latent optimizer is
    Keras' dense-decay Adam on the [N x q] table in both legs: "replay" (the classes' default: the untouched rows' zero-gradient
    steps deferred until a row is next used, csrc/z_replay.h) and "dense" (one sweep over the table per minibatch)."""
    import torch
    eng = model.engine
    dev = eng.device
    res = {}
    for mode, lazy in (("replay", 2), ("dense", 0), ("replay_host_loop", 2)):
        g = torch.Generator(device=dev).manual_seed(1)
        z = torch.randn(n_loc, eng.q, device=dev, generator=g)
        zm, zv = torch.zeros_like(z), torch.zeros_like(z)
        npar = eng.fit_begin(n_loc, batch)
        grad = torch.empty(npar, device=dev)
        perm = torch.randperm(n_loc, device=dev, generator=g).to(torch.int32)

        def run(k, s0):
            if mode != "replay_host_loop":          # what CausalBGM.fit does in a single process: one library call per epoch
                eng.fit_epoch(x, y, v, z, zm, zv, perm[s0 * batch:(s0 + k) * batch], batch, 1e-4, 1e-4, lazy)
                return
            for s_ in range(s0, s0 + k):            # the per-minibatch calls from Python (the form used under torch.distributed)
                idx = perm[s_ * batch:(s_ + 1) * batch]
                eng.fit_z_sync(z, zm, zv, idx, 1e-4)
                eng.fit_theta_grad(x, y, v, z, idx, batch, grad)
                eng.fit_theta_apply(grad, 1e-4)
                eng.fit_z_step(x, y, v, z, zm, zv, idx, batch, 1e-4, lazy=lazy)
        try:
            run(20, 0)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            run(steps, 20)
            if lazy == 2:
                eng.fit_z_sync(z, zm, zv, None, 1e-4)          # the flush a fit ends with, inside the timed region
            torch.cuda.synchronize()
            res[mode] = time.perf_counter() - t0
        finally:
            eng.fit_end()
    dt = res["replay"]
    return {"value": batch * steps / dt, "unit": "observations x epochs / s", "us_per_minibatch": 1e6 * dt / steps,
            "us_per_minibatch_dense_sweep": 1e6 * res["dense"] / steps,
            "us_per_minibatch_host_loop": 1e6 * res["replay_host_loop"] / steps,
            "sample": f"{steps} DISJOINT minibatches of a permutation, B={batch}, N={n_loc}, deterministic nets, issued by one "
                      "bgm_causal_fit_epoch call as CausalBGM.fit does (latent phase of minibatch k beside the theta phase of k + 1, "
                      "ring of parameter buffers, streams ordered by device-side counters); latent optimizer = dense-decay Adam in replay form (value, us_per_minibatch), as a "
                      "sweep over the table per minibatch (us_per_minibatch_dense_sweep: no overlap possible), and the replay form "
                      "issued per minibatch from Python (us_per_minibatch_host_loop: the form used under torch.distributed)",
            "flop_per_observation": 348480}


def fit_dp_leg(model, x, y, v, n_loc, world, device, steps=2000, batch=32):
    """BASELINE configs[3] ("data-parallel ... RCCL grad all-reduce over xGMI"), the fit side: `steps` minibatches of CausalBGM.fit's
    data-parallel form on every rank -- `batch` LOCAL rows per rank and step (a global minibatch of batch x world rows: weak scaling),
    the fused g|f|h gradient summed over the ranks once per step.  Over RCCL (one GPU per rank) that is ONE bgm_causal_fit_epoch_dp
    call: gradient tiles, ncclAllReduce and Adam enqueued from C++ on the parameter stream, the latent phase beside the next theta
    phase; with one rank the same call runs on a one-rank communicator (what the N=1 driver run reports next to `fit`); a process
    group without a GPU per rank (gloo, the one-device development aid) keeps the per-minibatch host loop, as the classes do.
    Every rank runs it; timing is the max over ranks.  reference: the loop that is sharded, causalbgm/base.py:488-514."""
    import torch
    import torch.distributed as dist
    from bayesgm_amd import parallel
    eng = model.engine
    comm = parallel.fit_comm(device) if world > 1 else parallel.DeviceComm(device, world=1, rank_=0)
    g = torch.Generator(device=device).manual_seed(1)
    z = torch.randn(n_loc, eng.q, device=device, generator=g)
    zm, zv = torch.zeros_like(z), torch.zeros_like(z)
    steps = int(min(steps, n_loc // batch - 20))
    npar = eng.fit_begin(n_loc, batch)
    grad = torch.empty(npar, device=device)
    perm = torch.randperm(n_loc, device=device, generator=g).to(torch.int32)

    def run(k, s0):
        if comm is not None:
            eng.fit_epoch_dp(comm, x, y, v, z, zm, zv, perm[s0 * batch:(s0 + k) * batch], batch, 1e-4, 1e-4, 2)
            return
        for s_ in range(s0, s0 + k):
            idx = perm[s_ * batch:(s_ + 1) * batch]
            eng.fit_z_sync(z, zm, zv, idx, 1e-4)
            eng.fit_theta_grad(x, y, v, z, idx, batch * world, grad)
            parallel.all_reduce_sum_(grad)
            eng.fit_theta_apply(grad, 1e-4)
            eng.fit_z_step(x, y, v, z, zm, zv, idx, batch * world, 1e-4, lazy=2)
    try:
        run(20, 0)
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        run(steps, 20)
        eng.fit_z_sync(z, zm, zv, None, 1e-4)          # the flush a fit ends with, inside the timed region
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        # the collective alone: the gradient buffer summed over the ranks, back to back on the stream
        reps = 200
        red = (lambda: comm.all_reduce_sum_(grad)) if comm is not None else (lambda: parallel.all_reduce_sum_(grad))
        grad.zero_()
        red()
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for _ in range(reps):
            red()
        torch.cuda.synchronize()
        ar_us = 1e6 * (time.perf_counter() - t1) / reps
    finally:
        eng.fit_end()
    t = torch.tensor([dt, ar_us], dtype=torch.float64, device=device)
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    dt, ar_us = float(t[0]), float(t[1])
    n_in = int(comm.info()["world"]) if comm is not None else world
    path = ("bgm_causal_fit_epoch_dp (loop + ncclAllReduce inside the library; %s)" % comm.info()["library"]) if comm is not None else \
           "per-minibatch host loop, torch.distributed all-reduce between the calls (no GPU per rank: backend %s)" % dist.get_backend()
    if world == 1:
        comm.close()
    return {"value": batch * world * steps / dt, "unit": "observations x epochs / s (whole job)", "us_per_minibatch": 1e6 * dt / steps,
            "minibatches": steps, "rows_per_rank_and_step": batch, "global_minibatch": batch * world, "n_ranks_in_collective": n_in,
            "gradient_floats": int(npar), "allreduce_us_alone": ar_us, "allreduce_share_of_minibatch": ar_us / (1e6 * dt / steps),
            "path": path, "scaling": "weak (the global minibatch grows with the ranks; the reference's algorithm at batch_size = 32 x ranks)"}


def training_leg(params, x, y, v, device, n=20000, batch=32, reps=200):
    """Secondary measurement: latencies of the training-side steps at the reference batch size on the first `n` rows of the bench
    panel (the tutorial's N) -- EGM discriminator / generator step and the minibatch theta / latent steps, for deterministic and
    Bayesian networks.  Each is one call through the C ABI (1-3 launches); see DESIGN_HISTORY.md section 4c."""
    import torch
    from bayesgm_amd.models import CausalBGM
    q, p = sum(params["z_dims"]), params["v_dim"]
    xs, ys, vs = x[:n].contiguous(), y[:n].contiguous(), v[:n].contiguous()
    g = torch.Generator(device=device).manual_seed(3)
    zb = torch.randn(batch, q, device=device, generator=g)
    idx = torch.randperm(n, device=device, generator=g)[:batch].to(torch.int32)
    z = torch.randn(n, q, device=device, generator=g)
    zm, zv = torch.zeros_like(z), torch.zeros_like(z)

    def timed(fn):
        for _ in range(10):
            fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            fn()
        torch.cuda.synchronize()
        return 1e6 * (time.perf_counter() - t0) / reps

    perm = torch.randperm(n, device=device, generator=g).to(torch.int32)[:(n // batch) * batch]

    def epoch_us(fn):
        """microseconds per minibatch of one epoch (n // batch disjoint minibatches) issued by ONE library call, as the classes do"""
        fn(perm[:20 * batch])
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        fn(perm)
        torch.cuda.synchronize()
        return 1e6 * (time.perf_counter() - t0) / (n // batch)

    rs = np.random.RandomState(5)
    dims = [q] + list(params["dz_units"]) + [1]
    dz = {"W": [(rs.uniform(-1, 1, (dims[i], dims[i + 1])) * np.sqrt(6.0 / (dims[i] + dims[i + 1]))).astype(np.float32) for i in range(len(dims) - 1)],
          "b": [np.zeros(d, np.float32) for d in dims[1:]], "gamma": [np.ones(d, np.float32) for d in dims[1:-1]],
          "beta": [np.zeros(d, np.float32) for d in dims[1:-1]]}
    out = {"sample": f"B={batch}, N={n}, p={p}, dz_units {list(params['dz_units'])}, g_d_freq {params['g_d_freq']}; microseconds per call, {reps} calls each"}
    # deterministic nets
    m = CausalBGM(dict(params, use_bnn=False), timestamp="bench_train_det", random_seed=0, device=device.index)
    eng = m.engine
    eng.set_disc_norm("fixed")            # the models' default (DESIGN_HISTORY.md section 2b)
    eng.egm_begin(batch, list(params["dz_units"]), float(params["lr"]), bool(params["use_z_rec"]), dz)
    try:
        d_us = timed(lambda: eng.egm_disc_step(zb, idx, vs, 0.5))
        g_us = timed(lambda: eng.egm_gen_step(zb, idx, vs, xs, ys))
    finally:
        eng.egm_end()
    npar = eng.fit_begin(n, batch)
    grad = torch.empty(npar, device=device)
    try:
        def step():
            eng.fit_z_sync(z, zm, zv, idx, 1e-4)
            eng.fit_theta_grad(xs, ys, vs, z, idx, batch, grad)
            eng.fit_theta_apply(grad, 1e-4)
            eng.fit_z_step(xs, ys, vs, z, zm, zv, idx, batch, 1e-4, lazy=2)
        f_us = timed(step)
        eng.fit_z_sync(z, zm, zv, None, 1e-4)
        fe_us = epoch_us(lambda pm: eng.fit_epoch(xs, ys, vs, z, zm, zv, pm, batch, 1e-4, 1e-4, 2))
    finally:
        eng.fit_end()
    out["deterministic"] = {"egm_disc_step_us": d_us, "egm_gen_step_us": g_us, "egm_iteration_ms": 1e-3 * (params["g_d_freq"] * d_us + g_us),
                            "fit_minibatch_us": f_us, "fit_minibatch_us_epoch_call": fe_us}
    # Bayesian nets (the reference's default)
    mb = CausalBGM(dict(params, use_bnn=True), timestamp="bench_train_bnn", random_seed=0, device=device.index)
    be = mb.engine
    be.set_disc_norm("fixed")
    be.egm_begin(dz, batch, float(params["lr"]), 1 if params["use_z_rec"] else 0)
    try:
        d_us = timed(lambda: be.egm_disc_step(zb, idx, vs, 0.5, 1, 0))
        g_us = timed(lambda: be.egm_gen_step(zb, idx, vs, xs, ys, 1, 1))
    finally:
        be.egm_end()
    t_us = timed(lambda: be.theta_step(z, idx, xs, ys, vs, 1e-4, 1, 0))
    zz = z.clone()
    def latent():
        be.z_sync(zz, zm, zv, idx, 1e-4)
        be.z_step(xs, ys, vs, zz, zm, zv, idx, 1e-4, 1, 1, lazy=2)
    l_us = timed(latent)
    be.z_sync(zz, zm, zv, None, 1e-4)
    be_us = epoch_us(lambda pm: be.fit_epoch(xs, ys, vs, zz, zm, zv, pm, batch, 1e-4, 1e-4, 2, 1, 0))
    out["bayesian"] = {"egm_disc_step_us": d_us, "egm_gen_step_us": g_us, "egm_iteration_ms": 1e-3 * (params["g_d_freq"] * d_us + g_us),
                       "theta_step_us": t_us, "latent_step_us": l_us, "fit_minibatch_us_epoch_call": be_us}
    out["fit_minibatch_us_epoch_call_note"] = ("theta + latent step of one minibatch when the epoch's minibatches are issued by one "
                                               "bgm_causal_fit_epoch / bgm_bnn_fit_epoch call (what CausalBGM.fit does in a single "
                                               "process): the latent phase of minibatch k overlaps the theta phase of minibatch k + 1")
    return out



def wide_bayesian_leg(device, n=100000, p=200, batch=32):
    """Secondary measurement: use_bnn=True outside the default widths (csrc/bnw_kernels.h sampling path, the general minibatch step kernels of
    csrc/bnn_kernels.h): one MH iteration / one kept iteration with 20 doses on n rows in blocks of 1e4, and the microseconds per 32-row
    minibatch of the epoch call.  Fractions are algorithmic Flipout FLOP (two products per layer, two evaluations per iteration) over the
    fp32-MFMA peak.  DESIGN.md section 4j."""
    import torch
    from bayesgm_amd.models import CausalBGM
    z_dims, bs, iters = [1, 1, 1, 7], 10000, 5
    q = sum(z_dims)
    g = torch.Generator(device=device).manual_seed(0)
    v = torch.randn(n, p, device=device, generator=g); x = torch.rand(n, device=device, generator=g); y = torch.randn(n, device=device, generator=g)
    out = {"sample": f"N={n}, p={p}, z_dims {z_dims}, blocks of {bs} rows, {iters} iterations timed; fit: B={batch} on the first 20000 rows"}
    for name, w in (("[128,128]", [128, 128]), ("[256]x3", [256] * 3)):
        u = dict(g_units=w, e_units=w, f_units=w, h_units=w)
        params = dict(dataset="bench_wide", output_dir=".", save_res=False, save_model=False, binary_treatment=False, use_bnn=True, z_dims=z_dims, v_dim=p,
                      lr_theta=1e-4, lr_z=1e-4, kl_weight=1e-4, lr=2e-4, g_d_freq=5, use_z_rec=True, dz_units=[64, 32, 8], **u)
        eng = CausalBGM(params, timestamp="bench_wide", random_seed=0, device=device.index).engine      # (random-init weights, inference-mode normalisation)
        state = torch.empty(n, q, device=device)
        eng.mh_run(x, y, v, state, bs, 0, 2, 0, 1.0, 1, init=True)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        eng.mh_run(x, y, v, state, bs, 2, iters, 0, 1.0, 1)
        torch.cuda.synchronize(); t = (time.perf_counter() - t0) / iters
        dims = {"g": [q] + w + [p + 1], "f": [z_dims[0] + z_dims[1] + 1] + w + [2], "h": [z_dims[0] + z_dims[2]] + w + [2]}
        macs = sum(a * b for net in ("g", "f", "h") for a, b in zip(dims[net][:-1], dims[net][1:]))
        macs_f = sum(a * b for a, b in zip(dims["f"][:-1], dims["f"][1:]))
        xs = torch.linspace(0, 3, 20, device=device)
        adrf = torch.zeros(20, iters, device=device, dtype=torch.float64)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        eng.mh_run(x, y, v, state, bs, 100, iters, 100, 1.0, 1, n_keep=iters, effect=1, x_values=xs, adrf_sum=adrf)
        torch.cuda.synchronize(); t2 = (time.perf_counter() - t0) / iters
        fl, fe = 8 * macs * n, 20 * 4 * macs_f * n
        ent = {"mh_iteration_ms": 1e3 * t, "transitions_per_s": n / t, "frac_of_fp32_mfma_peak": fl / t / 1e12 / 157.3,
               "kept_iteration_ms": 1e3 * t2, "outcome_net_frac_of_fp32_mfma_peak": fe / max(t2 - t, 1e-9) / 1e12 / 157.3,
               "flop_per_row_transition": 8 * macs}
        # the minibatch loop at these widths (one bgm_bnn_fit_epoch call, 400 minibatches)
        nf = 20000
        be = eng
        z = torch.randn(nf, q, device=device, generator=g); zm, zv = torch.zeros_like(z), torch.zeros_like(z)
        perm = torch.randperm(nf, device=device, generator=g).to(torch.int32)
        xs_, ys_, vs_ = x[:nf].contiguous(), y[:nf].contiguous(), v[:nf].contiguous()
        be.fit_epoch(xs_, ys_, vs_, z, zm, zv, perm[:20 * batch], batch, 1e-4, 1e-4, 2, 1, 0)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        be.fit_epoch(xs_, ys_, vs_, z, zm, zv, perm[:400 * batch], batch, 1e-4, 1e-4, 2, 1, 0)
        torch.cuda.synchronize()
        ent["fit_minibatch_us_epoch_call"] = 1e6 * (time.perf_counter() - t0) / 400
        out[name] = ent
    return out


def bgm_hmc_leg(device, n=200000, p=500, q=10, L=10, iters=4):
    """Secondary measurement (not `value`): the HMC transition kernels of BGM imputation at BASELINE config C4's shape (x_dim 500, z_dim 10,
    g_units [64] x 5, 10 leapfrog steps, 10 % cells missing) -- deterministic generator (bgm_hmc_kernel, head weights streamed) and the
    Bayesian generator under both noise modes (bgmf_hmc_kernel).  FLOP: L gradient evaluations x 4 MACs(g) per row-transition (forward + backward to the input, 2 FLOP per MAC), twice
    that with Flipout's two products per layer.  Random-init weights; timing only."""
    import time
    import torch
    from bayesgm_amd.engine import BgmEngine
    from bayesgm_amd.bvn_engine import BvnEngine
    rs = np.random.RandomState(0)
    macs = q * 64 + 4 * 4096 + 2 * 64 * p
    x = torch.randn(n, p, device=device)
    x[torch.rand(n, p, device=device) < 0.1] = float("nan")
    step = torch.full((1,), 0.01, device=device)

    def timed(eng):
        state, logp, grad = torch.empty((n, q), device=device), torch.empty(n, device=device), torch.empty((n, q), device=device)
        eng.hmc_run(x, state, logp, grad, step, 0, 1, 2 ** 30, L, 1, init=True)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        eng.hmc_run(x, state, logp, grad, step, 1, iters, 2 ** 30, L, 1)
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / iters

    def glorot(a, b):
        lim = np.sqrt(6.0 / (a + b))
        return rs.uniform(-lim, lim, (a, b)).astype(np.float32)

    def entry(kernel, dt, products):
        return {"kernel": kernel, "ms_per_transition": 1e3 * dt, "transitions_per_s": n / dt,
                "frac_of_fp32_mfma_peak": L * 4 * macs * products * n / dt / 1e12 / 157.3}
    out = {"sample": "N=%d rows, x_dim=%d, z_dim=%d, g_units [64] x 5, %d leapfrog steps, 10 %% cells missing, %d transitions timed" % (n, p, q, L, iters)}
    g = {"bn": dict(gamma=np.ones(q, np.float32), beta=np.zeros(q, np.float32), mean=np.zeros(q, np.float32), var=np.ones(q, np.float32)),
         "trunk": [(glorot(q if i == 0 else 64, 64), np.zeros(64, np.float32)) for i in range(5)],
         "mean": (glorot(64, p), np.zeros(p, np.float32)), "var": (glorot(64, p), np.zeros(p, np.float32))}
    eng = BgmEngine(p, q, g_units=[64] * 5)
    eng.set_weights(g)
    out["deterministic"] = entry("bgm_hmc_kernel (head weights streamed)", timed(eng), 1)
    # the same transitions in split precision (opt-in, params['hmc_precision'] = 'f16x3': csrc/bgm_kernels.h);
    # frac_of_fp32_mfma_peak is then a speed in fp32-peak equivalents, not a utilisation of the fp16 pipe
    eng.set_precision("f16x3")
    out["deterministic_f16x3"] = entry("bgm_hmc_kernel<PREC = 2> (every product on v_mfma_f32_16x16x32_f16 with hi / lo fp16 splits, three products per contraction; the generator streamed through LDS as fp16 fragments)", timed(eng), 1)
    out["deterministic_f16x3"]["speedup_vs_fp32"] = out["deterministic"]["ms_per_transition"] / out["deterministic_f16x3"]["ms_per_transition"]
    eng.set_precision("fp32")

    def flip(a, b):
        return ((0.1 * rs.standard_normal((a, b))).astype(np.float32), (-3.0 + 0.1 * rs.standard_normal((a, b))).astype(np.float32),
                (0.1 * rs.standard_normal(b)).astype(np.float32))
    vnet = {"gamma": np.ones(q, np.float32), "beta": np.zeros(q, np.float32), "mean_mv": np.zeros(q, np.float32), "var_mv": np.ones(q, np.float32),
            "trunk": [flip(q if i == 0 else 64, 64) for i in range(5)], "mean": flip(64, p), "var": flip(64, p)}
    for mode, frozen in (("bayesian_frozen_noise", True), ("bayesian_fresh_noise", False)):
        be = BvnEngine(p, q, g_units=[64] * 5, hmc_frozen_noise=frozen)
        be.begin(vnet)
        out[mode] = entry("bgmf_hmc_kernel (posterior means LDS-resident, perturbation streamed)", timed(be), 2)
        if frozen:      # the same transitions in split precision (opt-in, params['hmc_precision'] = 'f16x3': csrc/bgmfx_kernels.h)
            be.set_precision("f16x3")
            out[mode + "_f16x3"] = entry("bgmfx_hmc_kernel (posterior means and perturbation streamed as fp16 fragments, three fp16 products per contraction)", timed(be), 2)
            out[mode + "_f16x3"]["speedup_vs_fp32"] = out[mode]["ms_per_transition"] / out[mode + "_f16x3"]["ms_per_transition"]
        be.close()
    return out


def bayesian_leg(params, data, x_values, n_loc, args, device):
    """Secondary measurement (not `value`): the same predict with the reference's default Bayesian nets (use_bnn=True,
    DESIGN_HISTORY.md section 7) on a tenth of the iterations -- all blocks advance in lock step, a handful of launches per iteration,
    so the per-transition rate does not depend on the iteration count.  fp32 (the default arithmetic: `value`) and the opt-in split
    precision of the sampling kernels (params['mh_precision'] = 'f16x3', csrc/bnx_kernels.h: object `f16x3`)."""
    import torch
    from bayesgm_amd.models import CausalBGM
    div = max(1, int(args.bayesian_divisor))
    burn, keep, bs = max(1, args.burn_in // div), max(1, args.n_mcmc // div), 10000
    flop_row = 2 * 2 * 2 * 34848 if (args.p == 200) else None      # two states x two GEMMs per Flipout layer x 2 FLOP/MAC
    flop_eff_row = len(x_values) * 2 * 2 * 2512                    # per kept iteration: 20 doses x two products x 2 FLOP/MAC x MACs(f)

    def one(mode):
        m = CausalBGM(dict(params, use_bnn=True, mh_precision=mode), timestamp="bench_bnn", random_seed=0, device=device.index)
        m.predict(data, alpha=0.01, n_mcmc=2, burn_in=2, x_values=x_values, q_sd=1.0, sample_y=True, bs=bs, verbose=0)   # packs, allocates
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        adrf, _ = m.predict(data, alpha=0.01, n_mcmc=keep, burn_in=burn, x_values=x_values, q_sd=1.0, sample_y=True, bs=bs, verbose=0)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        out = {"value": n_loc * (burn + keep) / dt, "unit": "MH transitions/s", "seconds": dt,
               "sample": f"CausalBGM(use_bnn=True{'' if mode == 'fp32' else ', mh_precision=' + repr(mode)}).predict, N={n_loc}, bs={bs}, burn_in={burn}, n_mcmc={keep}, 20 doses",
               "ms_per_iteration": 1e3 * dt / (burn + keep), "acceptance_rate": m.last_acceptance_rate,
               "adrf_head": [float(t) for t in np.asarray(adrf)[:3]], "flop_per_row_transition": flop_row}
        # per-iteration cost of the two launch groups (HIP events on the stream the library launches on): a burn-in iteration =
        # perturbations + sign words + sampler kernel(s); a kept iteration adds the 20 outcome-net calls.  Fractions are of the fp32-MFMA
        # peak with the ALGORITHMIC work (2 states x 2 products x 34 848 MAC; 20 doses x 2 products x 2 512 MAC) -- for the split-precision
        # run they say how many fp32-peak-equivalents the fp16 pipe delivers, not a utilisation.
        eng = m.engine
        xs = torch.as_tensor(np.asarray(x_values, np.float32), device=eng.device)
        # `data` holds this rank's device tensors behind the Shard wrapper of main()
        x_, y_, v_ = (a_.t if hasattr(a_, "t") else torch.as_tensor(np.ascontiguousarray(a_, dtype=np.float32), device=eng.device) for a_ in data)
        x_, y_ = x_.reshape(-1).float().contiguous(), y_.reshape(-1).float().contiguous()
        v_ = v_.float().contiguous()
        state = torch.empty((n_loc, eng.q), device=eng.device)
        its = 40
        eng.mh_run(x_, y_, v_, state, bs, 0, 2, 0, 1.0, 1, init=True)
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
        adrf_s = torch.zeros((len(x_values), its), device=eng.device, dtype=torch.float64)
        ev[0].record()
        eng.mh_run(x_, y_, v_, state, bs, 2, its, 10 ** 6, 1.0, 1)
        ev[1].record()
        eng.mh_run(x_, y_, v_, state, bs, 100, its, 100, 1.0, 1, n_keep=its, effect=1, x_values=xs, adrf_sum=adrf_s)
        ev[2].record()
        torch.cuda.synchronize()
        t_mh, t_keep = ev[0].elapsed_time(ev[1]) / its, ev[1].elapsed_time(ev[2]) / its
        out["burn_in_iteration_ms"], out["kept_iteration_ms"] = t_mh, t_keep
        if flop_row:
            out["sampler_frac_of_fp32_mfma_peak"] = flop_row * n_loc / (t_mh * 1e-3) / (PEAK_FP32_MFMA_TFLOPS * 1e12)
            out["effects_frac_of_fp32_mfma_peak"] = flop_eff_row * n_loc / (max(t_keep - t_mh, 1e-9) * 1e-3) / (PEAK_FP32_MFMA_TFLOPS * 1e12)
            # the roofline of THIS model class: the two launch groups of an iteration as instances, HIP-event time
            # on the library's stream over `its` iterations of the whole panel, algorithmic FLOP as above
            t_eff = max(t_keep - t_mh, 1e-9)
            share_mh = t_mh * (burn + keep) / (t_mh * (burn + keep) + t_eff * keep)
            inst = [{"kernel": "bnf_mh_kernel + perturbation / sign kernels (one MH iteration of every row: two fresh log posteriors on Flipout nets)",
                     "avg_iteration_ms": t_mh, "iterations_timed": its, "flop_per_row_iteration": flop_row, "flop_per_iteration": flop_row * n_loc,
                     "achieved": flop_row * n_loc / (t_mh * 1e-3) / 1e12, "frac": out["sampler_frac_of_fp32_mfma_peak"],
                     "share_of_predict_kernel_time": share_mh},
                    {"kernel": "bnf_effects_kernel (a kept iteration's %d fresh-noise outcome-net calls)" % len(x_values),
                     "avg_iteration_ms": t_eff, "iterations_timed": its, "flop_per_row_iteration": flop_eff_row, "flop_per_iteration": flop_eff_row * n_loc,
                     "achieved": flop_eff_row * n_loc / (t_eff * 1e-3) / 1e12, "frac": out["effects_frac_of_fp32_mfma_peak"],
                     "share_of_predict_kernel_time": 1.0 - share_mh}]
            dom = inst[0] if share_mh >= 0.5 else inst[1]
            out["roofline"] = {"bound": "mfma", "kernel": dom["kernel"], "achieved": dom["achieved"], "peak": PEAK_FP32_MFMA_TFLOPS, "unit": "TFLOP/s",
                               "frac": dom["frac"], "instances": inst, "traffic": None,
                               "whole_predict_achieved": (flop_row * (burn + keep) + flop_eff_row * keep) * n_loc / dt / 1e12,
                               "note": "algorithmic FLOP (for f16x3: fp32-peak equivalents delivered by the fp16 pipe, not a utilisation)"}
        del m
        torch.cuda.empty_cache()
        return out

    out = one("fp32")
    x3 = one("f16x3")
    if flop_row:      # executed fp16 work: 3 products per contraction
        x3["sampler_frac_of_f16_mfma_peak"] = 3 * flop_row * n_loc / (x3["burn_in_iteration_ms"] * 1e-3) / 2.5e15
    x3["speedup_vs_fp32"] = x3["value"] / out["value"]
    x3["adrf_max_abs_diff_vs_fp32"] = float(np.max(np.abs(np.asarray(x3["adrf_head"]) - np.asarray(out["adrf_head"]))))
    out["f16x3"] = x3
    return out


def encoder_leg(model, v, n_loc, p, z_dims, check_against_oracle, reps=20):
    """north_star's encoder target (">= 40 % MFMA roofline on the encoder GEMM at 1 GPU"): z = e(V) over the whole bench panel
    (causalbgm/base.py:479, the latent initialisation of fit) on causal_encode_kernel -- six fused layers, V read once.  fp32 MFMA
    bound (74.6 FLOP per byte of V >> the machine balance); the HBM side is reported next to it."""
    import torch
    eng = model.engine
    q = sum(z_dims)
    z = eng.encode(v)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        z = eng.encode(v)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    macs = p * 64 + 4 * 64 * 64 + 64 * q
    flop = 2.0 * macs * n_loc
    out = {"kernel": "causal_encode_kernel (e_net over the panel: %d -> 64 x 5 -> %d, fused, V read once)" % (p, q), "rows": n_loc,
           "ms_per_pass": ms, "passes_timed": reps, "rows_per_s": n_loc / (ms * 1e-3), "flop_per_row": 2.0 * macs,
           "bound": "mfma", "achieved": flop / (ms * 1e-3) / 1e12, "peak": PEAK_FP32_MFMA_TFLOPS, "unit": "TFLOP/s",
           "frac": flop / (ms * 1e-3) / 1e12 / PEAK_FP32_MFMA_TFLOPS, "target_frac": 0.40,
           "algorithmic_bytes_per_row": 4 * p + 4 * q, "hbm_GBps": n_loc * (4 * p + 4 * q) / (ms * 1e-3) / 1e9,
           "frac_of_hbm_peak": n_loc * (4 * p + 4 * q) / (ms * 1e-3) / 8e12}
    if check_against_oracle:       # (the checker, with the cpu_baseline / parity legs only)
        from oracle.nets import mlp_forward
        e64 = [(np.asarray(W, np.float64), np.asarray(b, np.float64)) for W, b in model.nets["e"]]
        ref = mlp_forward(e64, v[:512].cpu().numpy().astype(np.float64))
        out["max_abs_err_vs_oracle_512_rows"] = float(np.abs(z[:512].cpu().numpy() - ref).max())
    return out


def config_c1_leg(device, n=100000, p=100, burn_in=5000, n_mcmc=3000):
    """BASELINE configs[1]: CausalBGM binary treatment, N = 1e5, p = 100, z_dims [3,3,6,6] (cli defaults), fp32, one GPU:
    predict = MH chains + per-row ITE draws + posterior mean and (alpha/2, 1-alpha/2) quantiles per row
    (causalbgm/base.py:573-668, 686-733), product defaults (outcome cache on); glorot weights, Hirano-Imbens covariates with the
    treatment binarised at its median (the reference has no binary simulator: SURVEY 8d)."""
    import torch
    from bayesgm_amd.models import CausalBGM
    from bayesgm_amd.datasets import Sim_Hirano_Imbens_sampler, binarize_treatment
    x, y, v = Sim_Hirano_Imbens_sampler(N=n, v_dim=p, seed=0).load_all()
    xb = binarize_treatment(x)
    z_dims = [3, 3, 6, 6]
    params = dict(dataset="bench_c1", output_dir=".", save_res=False, save_model=False, binary_treatment=True, use_bnn=False,
                  z_dims=z_dims, v_dim=p, lr_theta=1e-4, lr_z=1e-4, g_units=[64] * 5, f_units=[64, 32, 8], h_units=[64, 32, 8],
                  e_units=[64] * 5, dz_units=[64, 32, 8], kl_weight=1e-4, lr=2e-4, g_d_freq=5, use_z_rec=True)
    m = CausalBGM(params, timestamp="bench_c1", random_seed=0, device=device.index)
    data = tuple(torch.from_numpy(np.ascontiguousarray(a)).to(device) for a in (xb, y, v))
    m.predict(data, alpha=0.01, n_mcmc=4, burn_in=4, verbose=0)
    times, first_call = {}, {}
    res = {}
    for mode in (True, True, False, False):      # product default, then the outcome net at every retained draw (the reference's work);
        m.engine.set_outcome_cache(mode)         # each twice: the first call of a mode sizes its buffers, the second is what is reported
        m.engine.outcome_cache_stats(reset=True)
        m._seed_counter = 0
        torch.cuda.synchronize(); t0 = time.perf_counter()
        ite, interval = m.predict(data, alpha=0.01, n_mcmc=n_mcmc, burn_in=burn_in, verbose=0)
        torch.cuda.synchronize(); dt_ = time.perf_counter() - t0
        if mode in times:
            first_call[mode] = times[mode]
        times[mode] = dt_
        res[mode] = (np.asarray(ite), np.asarray(interval), m.engine.outcome_cache_stats(reset=True))
    m.engine.set_outcome_cache(True)
    macs = (sum(z_dims) * 64 + 4 * 4096 + 64 * (p + 1)) + ((z_dims[0] + z_dims[1] + 1) * 64 + 2048 + 256 + 16) + ((z_dims[0] + z_dims[2]) * 64 + 2048 + 256 + 16)
    dt = times[True]
    served, total = res[True][2]
    return {"workload": f"CausalBGM.predict binary treatment, N={n}, p={p}, z_dims {z_dims}, burn_in={burn_in}, n_mcmc={n_mcmc}, ITE + per-row intervals, product defaults",
            "predict_seconds": dt, "value": n * (burn_in + n_mcmc) / dt, "unit": "MH transitions/s", "predict_seconds_cache_off": times[False],
            "predict_seconds_first_call": first_call.get(True), "predict_seconds_cache_off_first_call": first_call.get(False),
            "tflops_algorithmic_transitions": 2 * macs * n * (burn_in + n_mcmc) / dt / 1e12,
            "frac_of_fp32_mfma_peak_transitions_only": 2 * macs * n * (burn_in + n_mcmc) / dt / 1e12 / PEAK_FP32_MFMA_TFLOPS,
            "acceptance_rate": m.last_acceptance_rate, "ate": float(res[True][0].mean()), "shapes": [list(res[True][0].shape), list(res[True][1].shape)],
            "outcome_cache_served_fraction": served / max(1, total), "outcome_cache_counted_in": "retained chain-iterations (event form)" if total == n * n_mcmc else "retained tile-iterations (16 chains)",
            "ite_identical_with_cache_off": bool(np.array_equal(res[True][0], res[False][0]) and np.array_equal(res[True][1], res[False][1]))}


def config_c4_share_leg(device, n=625000, p=500, q=10, burn_in=1000, n_mcmc=1000, use_bnn=False, precision="fp32"):
    """BASELINE configs[4], one GPU's share: BGM missing-data imputation, N = 5e6 / 8 = 625 000 rows, p = 500, 10 % of the cells
    missing (MCAR), 1000 burn-in (800 with the shared step-size adaptation) + 1000 retained HMC transitions of 10 leapfrog steps per
    row, posterior-predictive imputation + per-cell intervals through BGM.predict (bgm/base.py:527-663, 709-830); glorot weights.
    FLOP: L = 10 gradient evaluations x 4 MACs(g) per row-transition executed (the gradient at the current state is cached, as TFP does)."""
    import torch
    from bayesgm_amd.models import BGM
    bp = dict(dataset="bench_c4", output_dir=".", save_res=False, save_model=False, use_bnn=use_bnn, z_dim=q, x_dim=p,
              lr_theta=5e-3, lr_z=5e-3, g_units=[64] * 5, e_units=[64] * 5, dz_units=[64, 32, 8], dx_units=[64, 32, 8],
              kl_weight=5e-5, lr=1e-3, g_d_freq=1, use_z_rec=True, alpha=0.0, gamma=0.0)
    if precision != "fp32":
        bp["hmc_precision"] = precision
    bm = BGM(bp, timestamp="bench_c4", random_seed=0)
    g = torch.Generator(device=device).manual_seed(0)
    data = torch.randn(n, p, device=device, generator=g)
    data[torch.rand(n, p, device=device, generator=g) < 0.1] = float("nan")
    n_missing = int(torch.isnan(data).sum().item())
    data = data.cpu()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    imp, interval = bm.predict(data, n_mcmc=n_mcmc, burn_in=burn_in)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    macs = q * 64 + 4 * 4096 + 2 * 64 * p
    prod = 2 if use_bnn else 1
    return {"hmc_precision": precision, "imputed_checksum": float(np.nansum(np.asarray(imp)[:4096].astype(np.float64))),
            "workload": f"BGM(use_bnn={use_bnn}).predict imputation, N={n} (one GPU's share of 5e6 over 8), p={p}, z_dim={q}, 10 % cells missing, "
                        f"burn_in={burn_in}, n_mcmc={n_mcmc}, 10 leapfrog steps, host array in / host arrays out",
            "predict_seconds": dt, "value": n * (burn_in + n_mcmc) / dt, "unit": "HMC transitions/s", "missing_cells": n_missing,
            "phases_seconds": {k: v_ for k, v_ in getattr(bm, "last_predict_timing", {}).items() if not k.startswith("_")},
            "tflops_executed": n * (burn_in + n_mcmc) * 40 * macs * prod / dt / 1e12,
            "frac_of_fp32_mfma_peak": n * (burn_in + n_mcmc) * 40 * macs * prod / dt / 1e12 / PEAK_FP32_MFMA_TFLOPS,
            "acceptance_rate": bm.last_acceptance_rate, "imputed_shape": list(np.asarray(imp).shape),
            "eight_gpu_job_note": "rows are independent chains; the only exchange is the 8-byte step-size all-reduce per adaptation step (DESIGN 5)"}


def end_to_end_leg(params, x, y, v, x_values, n_loc, args, use_bnn, epochs=100, tag="det", seed=123):
    """The whole job a user runs on the bench panel, seconds per phase: CausalBGM(params).fit(...) with the
    reference's defaults (egm_init: 30000 iterations of 5 discriminator + 1 generator step at B = 32; `epochs` epochs of N / 32
    minibatches, evaluation every 5 epochs; causalbgm/base.py:380-532) followed by predict with the bench's MCMC settings in the
    product's default configuration.  Returns (object, trained model).  The fit is a latency chain of 32-row minibatches:
    its time is minibatches x microseconds per minibatch, whatever the GPU's arithmetic rate."""
    import contextlib
    import torch
    from bayesgm_amd.models import CausalBGM
    from bayesgm_amd.utils import get_ADRF
    t = {}
    with contextlib.redirect_stdout(sys.stderr):
        import warnings
        from bayesgm_amd import diagnostics
        caught_cm = warnings.catch_warnings(record=True)
        caught = caught_cm.__enter__()
        warnings.simplefilter("always")
        m = CausalBGM(dict(params, use_bnn=use_bnn), timestamp="bench_e2e_" + tag, random_seed=seed)
        egm = m.egm_init

        def timed_egm(*a_, **k_):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            r = egm(*a_, **k_)
            torch.cuda.synchronize(); t["egm_init"] = time.perf_counter() - t0
            return r
        m.egm_init = timed_egm
        torch.cuda.synchronize(); t0 = time.perf_counter()
        m.fit((x, y, v), epochs=epochs, epochs_per_eval=5, batch_size=32, use_egm_init=True, egm_n_iter=30000, egm_batches_per_eval=500, verbose=0)
        torch.cuda.synchronize(); t["fit_total"] = time.perf_counter() - t0
        m.egm_init = egm
        m._seed_counter = 0
        t0 = time.perf_counter()
        adrf, interval = m.predict((x, y, v), alpha=0.01, n_mcmc=args.n_mcmc, burn_in=args.burn_in, x_values=x_values, q_sd=1.0, sample_y=True,
                                   verbose=0, **({"bs": 10000} if use_bnn else {}))
        torch.cuda.synchronize(); t["predict"] = time.perf_counter() - t0
        x3 = None
        if use_bnn:      # the same predict (same seeds) with the sampling kernels in split precision (params['mh_precision'] = 'f16x3')
            m.engine.set_precision("f16x3")
            m._seed_counter = 0
            t0 = time.perf_counter()
            adrf3, _ = m.predict((x, y, v), alpha=0.01, n_mcmc=args.n_mcmc, burn_in=args.burn_in, x_values=x_values, q_sd=1.0, sample_y=True, verbose=0, bs=10000)
            torch.cuda.synchronize(); dt3 = time.perf_counter() - t0
            m.engine.set_precision("fp32")
            x3 = {"predict_seconds": dt3, "predict_transitions_per_s": n_loc * (args.burn_in + args.n_mcmc) / dt3, "acceptance_rate": m.last_acceptance_rate,
                  "adrf_max_abs_diff_vs_fp32": float(np.max(np.abs(np.asarray(adrf3) - np.asarray(adrf))))}
        caught_cm.__exit__(None, None, None)
    second_optimum = any(issubclass(w.category, diagnostics.SecondOptimumWarning) for w in caught)
    truth = get_ADRF(x_values=list(x_values), dataset="Imbens")
    err = np.asarray(adrf) - truth
    if x3 is not None:
        e3 = np.asarray(adrf3) - truth
        x3["adrf_rmse"] = float(np.sqrt(np.mean(e3 ** 2))); x3["average_effect_abs_error"] = float(abs(e3.mean()))
    n_mb = (epochs + 1) * ((n_loc + 31) // 32)
    fit_loop = t["fit_total"] - t.get("egm_init", 0.0)
    out = {"model": "CausalBGM(use_bnn=%s)" % use_bnn, "rows": n_loc, "epochs": epochs, "egm_iterations": 30000,
           "seconds": {"egm_init": t.get("egm_init"), "fit_epochs": fit_loop, "predict": t["predict"], "total": t["fit_total"] + t["predict"]},
           "minibatches": n_mb, "us_per_minibatch_incl_evaluations": 1e6 * fit_loop / n_mb,
           "observations_x_epochs_per_s": (epochs + 1) * n_loc / fit_loop,
           "predict_transitions_per_s": n_loc * (args.burn_in + args.n_mcmc) / t["predict"], "acceptance_rate": m.last_acceptance_rate,
           "adrf_rmse": float(np.sqrt(np.mean(err ** 2))), "average_effect_abs_error": float(abs(err.mean())), "best_epoch": getattr(m, "best_epoch", None),
           "random_seed": seed, "second_optimum_warning": second_optimum, "egm_late_l2_loss_z": getattr(m, "_egm_late_l2z", None),
           "sample": "fit((x, y, v), epochs=%d, epochs_per_eval=5, batch_size=32, use_egm_init=True, egm_n_iter=30000) + predict(n_mcmc=%d, burn_in=%d, "
                     "20 doses) on the bench panel itself (N=%d, p=%d); epochs + 1 passes as the reference loops range(epochs + 1) (base.py:488)"
                     % (epochs, args.n_mcmc, args.burn_in, n_loc, args.p)}
    if x3 is not None:
        out["predict_f16x3"] = x3
    return out, m


def accuracy_leg(params, data, x_values, n_loc, args, m=None, with_published=True):
    """The accuracy half of the metric ("... + ATE abs-error"): the model class of the headline number (deterministic nets) TRAINED with
    the reference's default schedule (30000 EGM iterations + 100 epochs, causalbgm/base.py:434 defaults) -- `m`, the end-to-end leg's
    model trained on the bench panel itself, or (m = None) a model trained here on the tutorial panel (Hirano-Imbens N = 20000) --
    predicts on the bench panel with the bench's MCMC settings, in fp32 and in split precision; errors against the analytic
    dose-response curve utils.get_ADRF(..., 'Imbens').  `published_configuration`: the reference's default Bayesian-network model on
    the tutorial's own setting (one run; the distribution over seeds: profiles/r03_accuracy/, tests/test_tutorial_trace.py)."""
    import contextlib
    import torch
    from bayesgm_amd.models import CausalBGM
    from bayesgm_amd.datasets import Sim_Hirano_Imbens_sampler
    from bayesgm_amd.utils import get_ADRF
    truth = get_ADRF(x_values=list(x_values), dataset="Imbens")
    x, y, v = Sim_Hirano_Imbens_sampler(N=20000, v_dim=args.p, seed=0).load_all()
    out = {}
    trained_on = "the bench panel (end_to_end leg)"
    with contextlib.redirect_stdout(sys.stderr):
        if m is None:
            trained_on = "Sim_Hirano_Imbens N=20000 seed 0"
            m = CausalBGM(dict(params, use_bnn=False), timestamp="bench_acc", random_seed=123)
            t0 = time.perf_counter()
            m.fit((x, y, v), epochs=100, epochs_per_eval=100, use_egm_init=True, egm_n_iter=30000, egm_batches_per_eval=30000, verbose=0)
            torch.cuda.synchronize()
            out["fit_seconds"] = time.perf_counter() - t0
        for mode in ("fp32", "bf16x3", "f16x3"):
            m.engine.set_precision(mode)
            m._seed_counter = 0
            m.engine.outcome_cache_stats(reset=True)
            t0 = time.perf_counter()
            adrf, interval = m.predict(data, alpha=0.01, n_mcmc=args.n_mcmc, burn_in=args.burn_in, x_values=x_values, q_sd=1.0,
                                       sample_y=True, verbose=0)
            torch.cuda.synchronize()
            err = adrf - truth
            served, total = m.engine.outcome_cache_stats(reset=True)
            out[mode] = {"outcome_cache_served_fraction": (served / total) if total else None, "adrf_rmse": float(np.sqrt(np.mean(err ** 2))), "adrf_mape": float(np.mean(np.abs(err / truth))),
                         "average_effect_abs_error": float(abs(err.mean())), "max_abs_error": float(np.abs(err).max()),
                         "interval_coverage": float(np.mean((interval[:, 0] <= truth) & (truth <= interval[:, 1]))),
                         "acceptance_rate": m.last_acceptance_rate, "predict_seconds": time.perf_counter() - t0}
        m.engine.set_precision("fp32")
        # The PUBLISHED configuration (docs/source/causalbgm/tutorial_py.ipynb): use_bnn=True, fit and predict on the same N = 20000
        # panel, predict(n_mcmc=3000, burn_in=5000, x_values=linspace(0, 3, 20), q_sd=1.0, bs=20000)
        import warnings
        from bayesgm_amd import diagnostics
        with warnings.catch_warnings(record=True) as caught:
            warnings.simplefilter("always")
            if with_published:
                mb = CausalBGM(dict(params, use_bnn=True), timestamp="bench_acc_bnn", random_seed=123)
                t0 = time.perf_counter()
                mb.fit((x, y, v), epochs=100, epochs_per_eval=10, use_egm_init=True, egm_n_iter=30000, egm_batches_per_eval=500, verbose=0)
                torch.cuda.synchronize()
                fit_s = time.perf_counter() - t0
                t0 = time.perf_counter()
                adrf, interval = mb.predict((x, y, v), alpha=0.01, n_mcmc=3000, burn_in=5000, x_values=x_values, q_sd=1.0, sample_y=True, bs=20000)
                torch.cuda.synchronize()
                err = adrf - truth
                out["published_configuration"] = {
                    "adrf_rmse": float(np.sqrt(np.mean(err ** 2))), "adrf_mape": float(np.mean(np.abs(err / truth))),
                    "average_effect_abs_error": float(abs(err.mean())), "acceptance_rate": mb.last_acceptance_rate,
                    "fit_seconds": fit_s, "predict_seconds": time.perf_counter() - t0,
                    "egm_late_l2_loss_z": getattr(mb, "_egm_late_l2z", None),
                    "second_optimum_warning": any(issubclass(w.category, diagnostics.SecondOptimumWarning) for w in caught),
                    "sample": "CausalBGM(use_bnn=True), random_seed 123: fit (30000 EGM iterations + 100 epochs) and predict(n_mcmc=3000, "
                              "burn_in=5000, q_sd=1.0, bs=20000) on Sim_Hirano_Imbens N=20000 p=%d seed 0 -- the tutorial's setting" % args.p}
    out["interval_coverage_note"] = (f"the interval is the posterior interval of a MEAN over {n_loc} rows (width ~ sd/sqrt(N)): it "
                                     "covers Monte-Carlo error of the chains, not the fit's bias, so coverage of the truth well below "
                                     "1 - alpha is expected and is not a calibration statement")
    out["sample"] = (f"CausalBGM(use_bnn=False) trained on {trained_on} (reference defaults: 30000 EGM "
                     f"iterations + 100 epochs, batch 32), predict on the bench panel N={n_loc} with burn_in={args.burn_in}, "
                     f"n_mcmc={args.n_mcmc}, {len(x_values)} doses, product defaults (outcome cache on); truth = x + 2/(1+x)^3")
    out["reference_published"] = {"adrf_rmse": 0.0188, "adrf_mape": 0.0103,
                                  "note": "docs/source/causalbgm/tutorial_py.ipynb (use_bnn=True, N=20000 train = test panel)"}
    return out


PEAK_BF16_MFMA_TFLOPS = 2500.0  # MI355X_MICROARCH.md "Peak BF16/FP16 MFMA" (dense)


def outcome_cache_leg(model, data, x_values, n_loc, args, seed_counter, adrf_ref, value_ref, mode=True):
    """Secondary measurement: the same predict with the outcome-net cache.  mode True (the product's default, bgm_causal_set_outcome_cache(2)):
    per CHAIN, the event form of the retained phase (csrc/causal_event_kernels.h: transitions that append an event per accepted move,
    the outcome net on dense 16-event tiles, a spread pass per (row, draw)); mode 'wave' (round 4): a retained iteration in which none
    of the 16 chains of a wave moved reuses the previous (mean, sd).  Same Philox streams as the last headline step: the ADRF must be
    identical to the last bit in both."""
    import torch
    eng = model.engine
    eng.set_outcome_cache(mode)
    try:
        eng.outcome_cache_stats(reset=True)
        eng.timing_enable(True)
        eng.timing_read(kind=-1, reset=True)
        model._seed_counter = seed_counter - 1
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        adrf, _ = model.predict(data, alpha=0.01, n_mcmc=args.n_mcmc, burn_in=args.burn_in, x_values=x_values, q_sd=1.0,
                                sample_y=True, verbose=0)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        n_k, ms_k = eng.timing_read(kind=1, reset=True)
        eng.timing_enable(False)
        served, total = eng.outcome_cache_stats(reset=True)
    finally:
        eng.set_outcome_cache(False)
    v = n_loc * (args.burn_in + args.n_mcmc) / dt
    unit = "retained chain-iterations" if mode is True else "retained tile-iterations (16 chains)"
    return {"mode": "per chain (event form)" if mode is True else "per wave of 16 chains", "value": v, "unit": "MH transitions/s", "seconds": dt,
            "speedup_vs_headline": v / value_ref, "keep_phase_ms": ms_k / max(1, n_k), "served": served, "of": total, "counted_in": unit,
            "served_fraction": served / max(1, total), "acceptance_rate": model.last_acceptance_rate,
            "adrf_max_abs_diff_vs_headline": float(np.abs(np.asarray(adrf) - np.asarray(adrf_ref)).max()),
            "note": "the headline `value` and `roofline` are measured with the cache OFF, every dose evaluated at every retained draw"}


def bf16x3_leg(model, data, x_values, n_loc, args, z_dims, flop_row_transition, flop_row_keep, seed_counter, mode="bf16x3"):
    """Secondary measurement (not `value`, which stays fp32 = the reference's arithmetic): the same predict with the opt-in
    split-precision kernels (params['mh_precision'] = 'bf16x3', DESIGN_HISTORY.md section 4b).  `achieved` prices the ALGORITHMIC FLOP
    (the fp32 count) -- an "fp32-equivalent" rate; `executed_bf16_tflops` counts the three bf16 products per contraction that
    the matrix pipe actually runs, against the dense bf16 peak."""
    import torch
    eng = model.engine
    eng.set_precision(mode)
    try:
        model.predict(data, alpha=0.01, n_mcmc=8, burn_in=8, x_values=x_values, q_sd=1.0, sample_y=True, verbose=0)   # packs
        torch.cuda.synchronize()
        eng.timing_enable(True)
        eng.timing_read(kind=-1, reset=True)
        model._seed_counter = seed_counter - 1       # the Philox streams of the last fp32 step: the two ADRFs are comparable
        t0 = time.perf_counter()
        adrf, _ = model.predict(data, alpha=0.01, n_mcmc=args.n_mcmc, burn_in=args.burn_in, x_values=x_values, q_sd=1.0,
                                sample_y=True, verbose=0)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        n_b, ms_b = eng.timing_read(kind=0, reset=False)
        n_k, ms_k = eng.timing_read(kind=1, reset=True)
        eng.timing_enable(False)
        # the same call with the product's default outcome-net cache (off for the numbers above, as for the headline)
        eng.set_outcome_cache(True)
        eng.outcome_cache_stats(reset=True)
        model._seed_counter = seed_counter - 1
        t0 = time.perf_counter()
        adrf_c, _ = model.predict(data, alpha=0.01, n_mcmc=args.n_mcmc, burn_in=args.burn_in, x_values=x_values, q_sd=1.0,
                                  sample_y=True, verbose=0)
        torch.cuda.synchronize()
        dt_c = time.perf_counter() - t0
        served, total = eng.outcome_cache_stats(reset=True)
        cached = {"value": n_loc * (args.burn_in + args.n_mcmc) / dt_c, "seconds": dt_c, "served_fraction": served / max(1, total),
                  "adrf_max_abs_diff": float(np.abs(np.asarray(adrf_c) - np.asarray(adrf)).max())}
    finally:
        eng.set_outcome_cache(False)
        eng.set_precision("fp32")
    flop = (flop_row_transition * args.burn_in + flop_row_keep * args.n_mcmc) * n_loc
    kern_s = (ms_b + ms_k) * 1e-3
    ach = flop / kern_s / 1e12 if kern_s > 0 else None
    return {"value": n_loc * (args.burn_in + args.n_mcmc) / dt, "unit": "MH transitions/s", "seconds": dt,
            "sample": f"CausalBGM(mh_precision='{mode}').predict, N={n_loc}, burn_in={args.burn_in}, n_mcmc={args.n_mcmc}, "
                      f"{len(x_values)} doses (one call)",
            "acceptance_rate": model.last_acceptance_rate, "adrf_head": [float(a) for a in adrf[:3]], "adrf": np.asarray(adrf),
            "burn_in_kernel_ms": ms_b / max(1, n_b), "keep_kernel_ms": ms_k / max(1, n_k), "with_outcome_cache": cached,
            "roofline": {"bound": "mfma", "kernel": "causal_mh_bx3_kernel (burn-in + keep launches)",
                         "achieved": ach, "unit": "TFLOP/s (algorithmic fp32-equivalent FLOP)",
                         "executed_bf16_tflops": 3.0 * ach if ach else None, "peak": PEAK_BF16_MFMA_TFLOPS,
                         "frac": (3.0 * ach / PEAK_BF16_MFMA_TFLOPS) if ach else None,
                         "note": "three 16-bit products per contraction; frac = executed bf16 / fp16 FLOP/s over the dense peak of the "
                                 "16-bit matrix pipe (the same for both formats)"}}


def self_launch(args):
    """`python bench.py --gpus N` with no WORLD_SIZE in the environment: run this same command line as N ranks, one per GPU, under
    torch.distributed.run on 127.0.0.1 (a free port), exactly the form the docstring gives.  Fails loudly when the box has fewer than N
    devices (BGM_BENCH_SINGLE_DEVICE=1, the one-GPU development aid, and --plumbing-only, which touches no device, are exempt).
    Returns the launcher's exit code; the ranks' stdout / stderr are this process's."""
    import socket
    import subprocess
    if not args.plumbing_only and os.environ.get("BGM_BENCH_SINGLE_DEVICE") != "1":
        import torch
        have = torch.cuda.device_count()
        if have < args.gpus:
            print(f"bench.py: --gpus {args.gpus} but this box shows {have} HIP device(s); refusing to run fewer ranks than asked for",
                  file=sys.stderr)
            return 2
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")       # dmabuf IPC: RCCL's intra-node transport needs it on this driver
    env.setdefault("OMP_NUM_THREADS", str(max(1, (os.cpu_count() or 8) // args.gpus)))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    print("bench.py: launching %d ranks: %s" % (args.gpus, " ".join(cmd)), file=sys.stderr)
    return subprocess.call(cmd, env=env)


def plumbing_only(args, world):
    """The launch path without a device (see --plumbing-only)."""
    import torch
    import torch.distributed as dist
    rank = int(os.environ.get("RANK", "0"))
    if world != args.gpus:
        print(f"--gpus {args.gpus} but WORLD_SIZE={world}", file=sys.stderr)
        return 2
    seen = world
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend="gloo")
        t1 = torch.ones(1)
        dist.all_reduce(t1)
        seen = int(t1.item())
        rows = [None] * world
        dist.all_gather_object(rows, plan_rows(args.n, world, rank, args.scaling))
    else:
        rows = [plan_rows(args.n, 1, 0, args.scaling)]
    if rank == 0:
        print(json.dumps({"plumbing_only": True, "n_gpus": world, "n_ranks_in_collective": seen, "scaling": args.scaling,
                          "rows": [list(r) for r in rows]}))
    if world > 1:
        dist.destroy_process_group()
    return 0 if seen == args.gpus else 3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--rows", "--n", dest="n", type=float, default=1e6, help="rows per GPU")
    ap.add_argument("--covariates", "--p", dest="p", type=int, default=200)
    ap.add_argument("--burn-in", type=int, default=5000)
    ap.add_argument("--n-mcmc", type=int, default=3000)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-bayesian", action="store_true", help="skip the secondary use_bnn=True measurement (N=1 only)")
    ap.add_argument("--no-fit", action="store_true", help="skip the secondary fit-throughput measurement (N=1 only)")
    ap.add_argument("--no-bgm", action="store_true", help="skip the secondary BGM HMC measurement at config C4's shape (N=1 only)")
    ap.add_argument("--no-general-width", action="store_true", help="skip the secondary measurement of the general-width engine (N=1 only)")
    ap.add_argument("--scaling", choices=("weak", "strong"), default="weak",
                    help="weak: --rows per GPU (default); strong: --rows in TOTAL, sharded over the GPUs (BASELINE configs[3]: N=1e6 over 8 GPUs)")
    ap.add_argument("--no-bf16x3", action="store_true", help="skip the secondary split-precision (bf16 x 3) measurement (N=1 only)")
    ap.add_argument("--no-accuracy", action="store_true", help="skip the accuracy leg (fit on the tutorial panel + ADRF error; N=1 only)")
    ap.add_argument("--no-end-to-end", action="store_true", help="skip the end-to-end leg (egm_init + fit(epochs=100) + predict on the bench panel, deterministic nets; N=1 only)")
    ap.add_argument("--end-to-end-bnn", type=int, default=0, metavar="EPOCHS",
                    help="also run the end-to-end job with the reference's default Bayesian nets for EPOCHS epochs (100 = the full default job, ~6 minutes at N=1e6; off by default)")
    ap.add_argument("--bayesian-divisor", type=int, default=1, help="the use_bnn=True leg runs burn_in / n_mcmc divided by this (1 = the BASELINE counts)")
    ap.add_argument("--no-configs", action="store_true", help="skip the legs of the other BASELINE configurations (C1 binary treatment, C4 one-GPU share) and the encoder leg (N=1 only)")
    ap.add_argument("--no-c4-f16x3", action="store_true", help="skip the split-precision run of the C4 share")
    ap.add_argument("--c4-bnn", action="store_true", help="also run the C4 share with the Bayesian generator (use_bnn=True): fp32 (~100 s) and f16x3 (~55 s)")
    ap.add_argument("--fit-dp-timeout", type=int, default=240, help="seconds the N>1 fit_dp leg may take before the line is printed without it")
    ap.add_argument("--plumbing-only", action="store_true",
                    help="launch / rendezvous check without a device: every rank joins the process group over gloo on the CPU, the rank "
                         "count is all-reduced and rank 0 prints {n_gpus, n_ranks_in_collective}; no kernel runs (tests/test_bench_launch.py)")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # bare `python bench.py --gpus N`: this process becomes the launcher of N ranks (one per GPU) and relays rank 0's JSON line
        raise SystemExit(self_launch(args))

    import torch
    import torch.distributed as dist
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.plumbing_only:
        raise SystemExit(plumbing_only(args, world))
    # stdout carries ONE JSON line.  Native libraries print there too (RCCL writes a version banner through C stdio when a communicator
    # is created, flushed at exit): from here on descriptor 1 is stderr for everything, and the line is written to the saved descriptor.
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)

    def emit(line):
        sys.stdout.flush()
        os.write(json_fd, (line + "\n").encode())

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    # Dev aid (1-GPU box): BGM_BENCH_SINGLE_DEVICE=1 BGM_BENCH_BACKEND=gloo runs all ranks on cuda:0 over gloo so
    # that the N>1 code path (sharding, ADRF all-reduce, max-over-ranks timing) can be exercised without N GPUs.
    if os.environ.get("BGM_BENCH_SINGLE_DEVICE") == "1":
        local_rank = 0
    backend = os.environ.get("BGM_BENCH_BACKEND", "nccl")
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=device)
        else:
            dist.init_process_group(backend=backend)

    from bayesgm_amd.models import CausalBGM
    from bayesgm_amd import parallel

    p = args.p
    n_loc, row_lo, n_total = plan_rows(args.n, world, rank, args.scaling)
    z_dims = [1, 1, 1, 7]
    params = dict(dataset="Sim_Hirano_Imbens", output_dir=".", save_res=False, save_model=False,
                  binary_treatment=False, use_bnn=False, z_dims=z_dims, v_dim=p, lr_theta=1e-4, lr_z=1e-4,
                  g_units=[64] * 5, f_units=[64, 32, 8], h_units=[64, 32, 8], kl_weight=1e-4, lr=2e-4,
                  g_d_freq=5, use_z_rec=True, e_units=[64] * 5, dz_units=[64, 32, 8])
    model = CausalBGM(params, timestamp="bench", random_seed=0, device=local_rank)
    eng = model.engine
    # The headline evaluates the outcome net at all 20 doses for EVERY retained draw, as the reference does (causalbgm/base.py:671-763).
    # The product's default (params['outcome_cache'] = True) skips those evaluations for the chains that did not move since the last
    # retained draw -- identical ADRF, measured below as the secondary object `outcome_cache`; it is switched off for `value` and `roofline`.
    eng.set_outcome_cache(False)
    if args.scaling == "strong":
        x, y, v = make_panel(n_loc, p, seed=0, device=device, lo=row_lo, n_gen=n_total)
    else:
        x, y, v = make_panel(n_loc, p, seed=rank, device=device)  # each rank its own panel (weak scaling)
    x_values = np.linspace(0, 3, 20)
    n_ranks_seen = world
    if world > 1:      # the collective the run depends on works, and every rank is there
        t1 = torch.ones(1, device=device)
        dist.all_reduce(t1)
        n_ranks_seen = int(t1.item())
        if n_ranks_seen != args.gpus:       # nothing is timed unless every requested rank is inside the collective
            raise SystemExit(f"--gpus {args.gpus} but {n_ranks_seen} ranks answered the all-reduce")

    class Shard:  # predict() shards data[lo:hi] by rank; hand it this rank's rows for any slice
        def __init__(self, t): self.t = t
        def __len__(self): return n_total
        def __getitem__(self, s): return self.t
    data = (Shard(x), Shard(y), Shard(v))

    def step():
        return model.predict(data, alpha=0.01, n_mcmc=args.n_mcmc, burn_in=args.burn_in, x_values=x_values,
                             q_sd=1.0, sample_y=True, verbose=0)

    for _ in range(args.warmup):
        step()
    eng.timing_enable(True)
    eng.timing_read(kind=-1, reset=True)
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        adrf, interval = step()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    t = torch.tensor([elapsed], dtype=torch.float64, device=device)
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    elapsed = float(t.item())
    eng.timing_enable(False)
    seed_counter_last = model._seed_counter

    # the one collective of the step (C3: ADRF draw sums [n_doses x n_mcmc], float64) timed alone, for its share of a step
    allreduce_ms = None
    if world > 1:
        buf = torch.zeros((len(x_values), args.n_mcmc), dtype=torch.float64, device=device)
        dist.all_reduce(buf)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for _ in range(10):
            dist.all_reduce(buf)
        torch.cuda.synchronize()
        allreduce_ms = 1e2 * (time.perf_counter() - t1)

    # which kernels each rank ran: the sampling path, and the minibatch-step path a data-parallel fit at the reference batch size
    # (32 rows in total, 32 // world per rank) would take on this rank (bgm_causal_describe inside a fit session)
    b_fit = max(1, 32 // world)
    eng.fit_begin(n_loc, b_fit)
    try:
        path = f"rank {rank}: rows [{row_lo}, {row_lo + n_loc}) {eng.describe(b_fit)}"
    finally:
        eng.fit_end()
    paths = [path]
    if world > 1:
        paths = [None] * world
        dist.all_gather_object(paths, path)
    if rank == 0:
        for line in paths:
            print(line, file=sys.stderr)

    out = None
    if rank == 0:
        iters = args.burn_in + args.n_mcmc
        value = n_total * iters * args.steps / elapsed
        info = eng.mh_info(n_loc)
        n_burn, ms_burn = eng.timing_read(kind=0, reset=False)
        n_keep_l, ms_keep = eng.timing_read(kind=1, reset=False)
        # HBM traffic cannot be collected from inside this process: use the committed PMC passes of the same kernels on the
        # same panel shape, if there are any (per-launch read traffic is independent of the iteration count, see the file)
        traffic = {}
        traffic_src = None
        traffic_write = None
        for name in ("r05_pmc_traffic.json", "r04_pmc_traffic.json", "r02_pmc_traffic.json", "r01_pmc_traffic.json"):
            try:
                with open(os.path.join(ROOT, "profiles", name)) as f:
                    pm = json.load(f)
                if pm["rows"] == n_loc and pm["p"] == p:
                    traffic = {"burn_in": pm.get("hbm_read_bytes_per_launch"), "keep": pm.get("hbm_read_bytes_per_launch_keep")}
                    traffic_src = "profiles/" + name
                    w = pm.get("write")
                    if w:      # the write side: raw WRITE_SIZE of the fused keep kernel scales with the retained iterations
                        traffic_write = {"raw_bytes_per_retained_iteration": w["keep_raw_bytes_per_retained_iteration"],
                                         "raw_bytes_per_launch": w["keep_raw_bytes_per_retained_iteration"] * args.n_mcmc,
                                         "algorithmic": w["keep_algorithmic_bytes_per_launch"], "explanation": w["keep_explanation"],
                                         "event_form_raw_bytes_per_retained_iteration": w["event_form_raw_bytes_per_retained_iteration"]}
                    break
            except (OSError, KeyError, ValueError):
                pass
        # algorithmic work: a transition = 2 MACs(g+f+h) FLOP per row (the cached current log-posterior is not
        # re-evaluated); a retained draw adds the outcome net at every dose: 2 MACs(f) FLOP per row and dose
        macs_f = (z_dims[0] + z_dims[1] + 1) * 64 + 64 * 32 + 32 * 8 + 8 * 2
        n_doses = len(x_values)
        flop_keep_row = info.flop_per_row_transition + n_doses * 2 * macs_f
        inst = []
        if n_burn:
            avg = ms_burn / n_burn
            flop = info.flop_per_row_transition * n_loc * args.burn_in
            inst.append({"kernel": "causal_mh_kernel<EFFECT=0> (burn-in: transitions only)", "avg_launch_ms": avg, "launches": n_burn,
                         "flop_per_row_iteration": info.flop_per_row_transition, "flop_per_launch": flop,
                         "achieved": flop / (avg * 1e-3) / 1e12, "traffic": traffic.get("burn_in")})
        if n_keep_l:
            avg = ms_keep / n_keep_l
            flop = flop_keep_row * n_loc * args.n_mcmc
            inst.append({"kernel": "causal_mh_kernel<EFFECT=1> (keep phase: transition + outcome net at %d doses per retained draw)" % n_doses,
                         "avg_launch_ms": avg, "launches": n_keep_l, "flop_per_row_iteration": flop_keep_row, "flop_per_launch": flop,
                         "achieved": flop / (avg * 1e-3) / 1e12, "traffic": traffic.get("keep")})
        roof = None
        if inst:
            tot_ms = sum(k["avg_launch_ms"] * k["launches"] for k in inst)
            for k in inst:
                k["frac"] = k["achieved"] / PEAK_FP32_MFMA_TFLOPS
                k["share_of_kernel_time"] = k["avg_launch_ms"] * k["launches"] / tot_ms
            dom = max(inst, key=lambda k: k["share_of_kernel_time"])       # the dominant instance is the one reported
            roof = {"bound": "mfma", "kernel": dom["kernel"], "achieved": dom["achieved"], "peak": PEAK_FP32_MFMA_TFLOPS,
                    "unit": "TFLOP/s", "frac": dom["frac"], "traffic": dom["traffic"], "traffic_unit": "HBM read bytes per launch",
                    "traffic_source": traffic_src, "traffic_write": traffic_write if dom["kernel"].startswith("causal_mh_kernel<EFFECT=1>") else None,
                    # per launch every row's x, y, v row is read once, its chain state (q floats + the cached log-posterior) read
                    # and written back once: the same accounting as profiles/r02_pmc_traffic.json (852 B per row at p = 200, q = 10)
                    "algorithmic_bytes_per_launch": n_loc * (4 * p + 8 + 4 * sum(z_dims) + 4),
                    "avg_launch_ms": dom["avg_launch_ms"], "launches": dom["launches"], "flop_per_launch": dom["flop_per_launch"],
                    "share_of_kernel_time": dom["share_of_kernel_time"], "instances": inst,
                    "whole_predict_achieved": sum(k["flop_per_launch"] * k["launches"] for k in inst) / (tot_ms * 1e-3) / 1e12}
        out = {
            "metric": "posterior samples/sec (whole node), CausalBGM N=1e6 p=200",
            "value": value, "unit": "MH transitions/s (rows x (burn_in+n_mcmc) / t_predict)",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True, "scaling": args.scaling,
            "vs_baseline": None, "dtype": "f32",
            "data": "synthetic (Sim_Hirano_Imbens generator; throughput legs on random-init glorot weights, accuracy leg on trained weights)",
            "config": {"workload": f"CausalBGM.predict continuous treatment, N={n_loc} rows/GPU ({n_total} in total), p={p}, "
                                   f"z_dims {z_dims}, burn_in={args.burn_in}, n_mcmc={args.n_mcmc}, q_sd=1.0, "
                                   f"20 doses, sample_y=True",
                       "rows_per_gpu": n_loc, "rows_total": n_total, "p": p,
                       "parallelism": f"dp{world} (rows sharded, ADRF all-reduce)"},
            "n_ranks_in_collective": n_ranks_seen, "adrf_allreduce_ms": allreduce_ms,
            "adrf_allreduce_share_of_step": (allreduce_ms / (1e3 * elapsed / args.steps)) if allreduce_ms is not None else None,
            "retained_draws_per_s": n_total * args.n_mcmc * args.steps / elapsed,
            "acceptance_rate": model.last_acceptance_rate,
            "adrf_head": [float(a) for a in adrf[:3]],
            "roofline": roof,
            "kernel_paths": paths,
        }
        if not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline(params, p, z_dims)
            out["gpu_over_cpu"] = value / out["cpu_baseline"]["value"]
            out["parity"] = parity_leg(model, x, y, v, z_dims, p, x_values)
        if not args.no_configs and world == 1:      # (before anything retrains `model`: the encoder of the bench weights)
            out["encoder"] = encoder_leg(model, v, n_loc, p, z_dims, check_against_oracle=not args.no_cpu_baseline)
        if world == 1:
            out["outcome_cache"] = outcome_cache_leg(model, data, x_values, n_loc, args, seed_counter_last, adrf, value, mode=True)
            out["outcome_cache"]["per_wave"] = outcome_cache_leg(model, data, x_values, n_loc, args, seed_counter_last, adrf, value, mode="wave")
        if not args.no_bf16x3 and world == 1:      # before the fit leg, which trains (changes) the weights of `model`
            out["bf16x3"] = bf16x3_leg(model, data, x_values, n_loc, args, z_dims, info.flop_per_row_transition, flop_keep_row, seed_counter_last)
            out["bf16x3"]["speedup_vs_fp32"] = out["bf16x3"]["value"] / value
            out["bf16x3"]["adrf_max_abs_diff_vs_fp32"] = float(np.abs(out["bf16x3"].pop("adrf") - adrf).max())
            # the same kernels on fp16 operands (params['mh_precision'] = 'f16x3'): 22 instead of 16 mantissa bits per contraction
            f16 = bf16x3_leg(model, data, x_values, n_loc, args, z_dims, info.flop_per_row_transition, flop_keep_row, seed_counter_last, mode="f16x3")
            f16["speedup_vs_fp32"] = f16["value"] / value
            f16["adrf_max_abs_diff_vs_fp32"] = float(np.abs(f16.pop("adrf") - adrf).max())
            out["bf16x3"]["f16x3"] = f16
        trained = None
        if not args.no_end_to_end and world == 1:
            out["end_to_end"], trained = end_to_end_leg(params, x, y, v, x_values, n_loc, args, use_bnn=False)
        if not args.no_accuracy and world == 1:
            out["accuracy"] = accuracy_leg(params, data, x_values, n_loc, args, m=trained)
        if args.end_to_end_bnn > 0 and world == 1:
            out["end_to_end_bayesian"], _ = end_to_end_leg(params, x, y, v, x_values, n_loc, args, use_bnn=True, epochs=args.end_to_end_bnn, tag="bnn")
        if not args.no_bayesian and world == 1:
            out["bayesian_nets"] = bayesian_leg(params, data, x_values, n_loc, args, device)
        if not args.no_bgm and world == 1:
            out["bgm_hmc"] = bgm_hmc_leg(device)
        if not args.no_configs and world == 1:
            import contextlib
            with contextlib.redirect_stdout(sys.stderr):      # (the classes print progress lines as the reference does)
                out["config_c1"] = config_c1_leg(device)
                out["config_c4_share"] = config_c4_share_leg(device)
                if not args.no_c4_f16x3:      # the same job with the opt-in split-precision heads
                    x3 = config_c4_share_leg(device, precision="f16x3")
                    x3["speedup_vs_fp32"] = out["config_c4_share"]["predict_seconds"] / x3["predict_seconds"]
                    out["config_c4_share"]["f16x3"] = x3
                if args.c4_bnn:      # the Bayesian generator (frozen noise): fp32 and the opt-in split precision (bgmfx_kernels.h)
                    out["config_c4_share"]["use_bnn"] = config_c4_share_leg(device, use_bnn=True)
                    bx3 = config_c4_share_leg(device, use_bnn=True, precision="f16x3")
                    bx3["speedup_vs_fp32"] = out["config_c4_share"]["use_bnn"]["predict_seconds"] / bx3["predict_seconds"]
                    out["config_c4_share"]["use_bnn"]["f16x3"] = bx3
        if not args.no_general_width and world == 1:
            out["general_width_engine"] = general_width_leg(p, z_dims, device)
            import contextlib
            with contextlib.redirect_stdout(sys.stderr):
                out["wide_bayesian_nets"] = wide_bayesian_leg(device)
        if not args.no_fit and world == 1:
            out["fit"] = fit_leg(model, x, y, v, n_loc)
            out["fit_dp"] = fit_dp_leg(model, x, y, v, n_loc, 1, device)      # the data-parallel call on a one-rank communicator
            out["fit_dp"]["vs_single_process_epoch_call"] = out["fit_dp"]["us_per_minibatch"] / out["fit"]["us_per_minibatch"]
            out["training_steps"] = training_leg(params, x, y, v, device)
    if world > 1 and not args.no_fit:
        # every rank: the data-parallel minibatch loop with its gradient all-reduce (configs[3], fit side) -- AFTER everything of the
        # headline is measured, and under a watchdog: this is where the library's own RCCL communicator is created, and the headline
        # line must survive whatever a first multi-GPU run does here (a failure or a stall is reported in the line, not instead of it)
        import threading
        finished = threading.Event()

        def watchdog():
            if not finished.wait(args.fit_dp_timeout):
                if rank == 0:
                    out["fit_dp"] = {"error": "no result within %d s (the numbers above were complete before this leg started)" % args.fit_dp_timeout}
                    emit(json.dumps(out))
                os._exit(0)
        threading.Thread(target=watchdog, daemon=True).start()
        try:
            fit_dp = fit_dp_leg(model, x, y, v, n_loc, world, device)
        except Exception as e:
            fit_dp = {"error": "%s: %s" % (type(e).__name__, e)}
        finished.set()
        if rank == 0:
            out["fit_dp"] = fit_dp
    if rank == 0:
        emit(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
