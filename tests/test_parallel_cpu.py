"""World-size-2 gloo tests (CPU) of the N>1 host logic: row sharding, unequal-length gathers, and the two
data-parallel identities the multi-GPU path relies on -- checked with the oracle as the compute:
  (C1) sum over ranks of local theta-gradients taken with inv_B = 1/B_global  ==  gradient of the
       global-batch mean loss (so one all-reduce(SUM) reproduces the single-process step);
  (C3) predict: all-reduce(SUM) of n_loc-weighted ADRF draw means / n_total == single-process ADRF draws,
       independent of the sharding because the Philox stream is keyed by the GLOBAL row index.
The GPU kernels themselves cannot run here (no CPU fallback by design)."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, fn, ret):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        ret[rank] = fn(rank, world)
    finally:
        dist.destroy_process_group()


def _run(fn, world=2):
    port = 29500 + (os.getpid() % 2000)
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(world, port, fn, ret), nprocs=world, join=True)
    return [ret[r] for r in range(world)]


def _shard_fn(rank, world):
    from bayesgm_amd import parallel
    assert parallel.is_dist() and parallel.rank() == rank and parallel.world_size() == world
    n = 11
    lo, hi = parallel.shard_range(n)
    full = torch.arange(n * 3, dtype=torch.float32).reshape(n, 3)
    got = parallel.all_gather_rows(full[lo:hi].clone(), n)
    t = torch.tensor([float(rank + 1)])
    parallel.all_reduce_sum_(t)
    return (lo, hi, bool(torch.equal(got, full)), float(t.item()))


def test_shard_range_and_gather():
    from bayesgm_amd import parallel
    # single process behaviour
    assert parallel.shard_range(10) == (0, 10) and parallel.world_size() == 1
    cover = [parallel.shard_range(11, r, 4) for r in range(4)]
    assert cover[0][0] == 0 and cover[-1][1] == 11 and all(a[1] == b[0] for a, b in zip(cover, cover[1:]))
    assert max(h - l for l, h in cover) - min(h - l for l, h in cover) <= 1
    res = _run(_shard_fn)
    assert res[0][:2] == (0, 6) and res[1][:2] == (6, 11)
    assert all(r[2] for r in res) and all(r[3] == 3.0 for r in res)


def _dp_grad_fn(rank, world):
    from oracle import causal as OC, fit as OF
    from bayesgm_amd import parallel
    rs = np.random.RandomState(0)
    m = OC.cast_model(OC.init_model(1, [1, 1, 1, 7], 20), np.float64)
    n, B = 40, 16
    z = rs.randn(n, 10); v = rs.randn(n, 20); x = rs.exponential(size=(n, 1)); y = rs.randn(n, 1)
    idx = rs.choice(n, B, replace=False)
    # single-process reference gradient on the global batch
    _, _, gg, _ = OF.g_loss_and_grads(m, z[idx], v[idx])
    ref = np.concatenate([np.concatenate([a.ravel(), b.ravel()]) for a, b in gg])
    # this rank's half of the batch, batch-mean factor of the GLOBAL batch: scale local-mean grads by B_loc/B
    loc = idx[rank::world]
    _, _, gl, _ = OF.g_loss_and_grads(m, z[loc], v[loc])
    mine = np.concatenate([np.concatenate([a.ravel(), b.ravel()]) for a, b in gl]) * (len(loc) / B)
    t = torch.from_numpy(mine)
    parallel.all_reduce_sum_(t)
    return float(np.abs(t.numpy() - ref).max() / np.abs(ref).max())


def test_dp_gradient_allreduce_equals_global_batch_gradient():
    res = _run(_dp_grad_fn)
    assert all(r < 1e-12 for r in res), res


def _egm_share_fn(rank, world):
    """Host logic of the data-parallel warm start: the ranks' shares of a global draw partition its first world * b_loc slots, every
    mapped row lies in the rank's shard, and the mean of the ranks' oracle gradients on their shares (each scaled 1 / world, summed by
    the all-reduce) is the oracle gradient of the global minibatch."""
    from oracle import egm as OE
    from oracle import nets as NN
    from bayesgm_amd import parallel, host_rng
    n, B, q, p, g_d_freq, n_it = 101, 8, 10, 12, 2, 3
    np.random.seed(4)
    idx, z, eps, _ = host_rng.egm_block_from(np.random.get_state(), n, B, q, n_it, g_d_freq)
    lo, hi = parallel.shard_range(n)
    b_loc = B // world
    li, lz = host_rng.egm_rank_share(idx, z, n, hi - lo, b_loc, rank)
    ok = li.shape == (n_it, g_d_freq + 1, b_loc) and li.min() >= 0 and li.max() < hi - lo
    ok = ok and np.array_equal(lz, z[:, :, rank * b_loc:(rank + 1) * b_loc])
    rs = np.random.RandomState(0)
    z_dims = [1, 1, 1, 7]
    nets = {"g": NN.init_mlp(rs, [q, 16, p + 1]), "e": NN.init_mlp(rs, [p, 16, q]), "f": NN.init_mlp(rs, [3, 8, 2]), "h": NN.init_mlp(rs, [2, 8, 2])}
    nets = {k: NN.cast_net(v, np.float64) for k, v in nets.items()}
    dz = OE.cast_disc(OE.init_disc(rs, q, [8, 4]), np.float64)
    dz["fixed_norm"] = True                      # per-row normalisation: a mean over rows splits over ranks exactly
    v = rs.randn(n, p); x = rs.rand(n, 1); y = rs.randn(n, 1)
    params = dict(v_dim=p, z_dims=z_dims, binary_treatment=False, use_z_rec=True, lr=2e-4)
    rows = lo + li[0, g_d_freq].astype(np.int64)
    _, gr = OE.gen_step_grads(nets, dz, params, lz[0, g_d_freq].astype(np.float64), v[rows], x[rows], y[rows])
    mine = torch.from_numpy(np.concatenate([a.ravel() for a in OE.gen_param_list(gr)]) / world)
    parallel.all_reduce_sum_(mine)
    rows_all = torch.zeros(world * b_loc, dtype=torch.int64)
    rows_all[rank * b_loc:(rank + 1) * b_loc] = torch.from_numpy(rows)
    parallel.all_reduce_sum_(rows_all)
    ra = rows_all.numpy()
    _, gr_all = OE.gen_step_grads(nets, dz, params, z[0, g_d_freq, :world * b_loc].astype(np.float64), v[ra], x[ra], y[ra])
    ref = np.concatenate([a.ravel() for a in OE.gen_param_list(gr_all)])
    _, _, gd = OE.disc_step_grads(nets, dz, lz[0, 0].astype(np.float64), v[lo + li[0, 0]], float(eps[0, 0, 0]))
    mine_d = torch.from_numpy(np.concatenate([a.ravel() for a in OE.disc_param_list(gd)]) / world)
    parallel.all_reduce_sum_(mine_d)
    rows_d = torch.zeros(world * b_loc, dtype=torch.int64)
    rows_d[rank * b_loc:(rank + 1) * b_loc] = torch.from_numpy(lo + li[0, 0].astype(np.int64))
    parallel.all_reduce_sum_(rows_d)
    _, _, gd_all = OE.disc_step_grads(nets, dz, z[0, 0, :world * b_loc].astype(np.float64), v[rows_d.numpy()], float(eps[0, 0, 0]))
    ref_d = np.concatenate([a.ravel() for a in OE.disc_param_list(gd_all)])
    return (bool(ok), float(np.abs(mine.numpy() - ref).max() / np.abs(ref).max()), float(np.abs(mine_d.numpy() - ref_d).max() / np.abs(ref_d).max()))


def test_dp_egm_shares_and_gradient_allreduce_equal_the_global_minibatch():
    res = _run(_egm_share_fn)
    assert all(ok and a < 1e-10 and b < 1e-10 for ok, a, b in res), res


def _adrf_fn(rank, world):
    from oracle import causal as OC
    from bayesgm_amd import parallel
    rs = np.random.RandomState(3)
    m = OC.init_model(2, [1, 1, 1, 7], 20)
    n = 37
    v = rs.randn(n, 20).astype(np.float32); x = rs.exponential(size=(n, 1)).astype(np.float32)
    y = rs.randn(n, 1).astype(np.float32)
    xs = np.array([0.5, 1.5])
    burn, keep, seed = 6, 5, 11
    lo, hi = parallel.shard_range(n)
    pz = OC.mh_sampler(m, (x[lo:hi], y[lo:hi], v[lo:hi]), burn, keep, 0.5, seed, row0=lo)
    eff = OC.infer_from_latent_posterior(m, pz, xs, True, seed, row0=lo, burn_in=burn)      # means over local rows
    sums = torch.from_numpy(eff.astype(np.float64) * (hi - lo))
    parallel.all_reduce_sum_(sums)
    got = (sums / n).numpy()
    pz_all = OC.mh_sampler(m, (x, y, v), burn, keep, 0.5, seed)
    ref = OC.infer_from_latent_posterior(m, pz_all, xs, True, seed, burn_in=burn)
    return float(np.abs(got - ref).max())


def test_sharded_adrf_equals_single_process():
    res = _run(_adrf_fn)
    assert all(r < 1e-5 for r in res), res


def _bnn_dp_fn(rank, world):
    """Host logic of the data-parallel Bayesian-net paths: block-wise sharding of predict with a variable-length gather,
    and the KL weighting of the gradient all-reduce (every rank adds kl_weight * B_loc / B_global of the KL gradient)."""
    from oracle import bnn as OB
    from bayesgm_amd import parallel
    # (1) whole blocks per rank, gathered in rank order
    n, bs = 2350, 300
    n_blocks = (n + bs - 1) // bs
    b_lo, b_hi = parallel.shard_range(n_blocks)
    lo, hi = min(n, b_lo * bs), min(n, b_hi * bs)
    full = torch.arange(n, dtype=torch.float32).reshape(n, 1)
    got = parallel.all_gather_rows_var(full[lo:hi].clone())
    ok_gather = bool(torch.equal(got, full)) and lo % bs == 0
    # (2) sum over ranks of [local NLL gradient with 1/B_global + kl_weight * (B_loc / B_global) * dKL] == global objective's
    #     gradient when every rank normalises with ITS batch statistics and noise (the stated data-parallel semantics)
    rs = np.random.RandomState(0)
    m = OB.init_model(1, [1, 1, 1, 3], 9, False, g_units=(8, 8), e_units=(8,), f_units=(6, 4), h_units=(6, 4), dtype=np.float64)
    B, klw = 12, 0.3
    z = rs.randn(B, 6); v = rs.randn(B, 9); x = rs.exponential(size=(B, 1)); y = rs.randn(B, 1)
    loc = np.arange(B)[rank::world]
    noise = OB.draw_noise(OB.net_dims(m["f"]), len(loc), 77 + 97 * rank, 3, OB.NET_ID["f"], dtype=np.float64)
    # local step as the kernel computes it: batch mean over the GLOBAL batch, KL share B_loc / B
    _, _, g_loc = OB.theta_step(m, "f", z[loc], x[loc], y[loc], v[loc], noise, 0.0)
    _, klg = OB.kl(m["f"])
    share = len(loc) / B
    mine = np.concatenate([a.ravel() for a in OB.flat_grads(OB.add_grads(g_loc, klg, klw * share))])
    mine_nll = np.concatenate([a.ravel() for a in OB.flat_grads(g_loc)]) * share     # local mean -> global mean
    t = torch.from_numpy(mine_nll + (mine - np.concatenate([a.ravel() for a in OB.flat_grads(g_loc)])))
    parallel.all_reduce_sum_(t)
    # the KL part of the reduced gradient must be exactly kl_weight * dKL, whatever the split
    t_kl = torch.from_numpy(np.concatenate([a.ravel() for a in OB.flat_grads(klg)]) * klw * share)
    parallel.all_reduce_sum_(t_kl)
    ref_kl = np.concatenate([a.ravel() for a in OB.flat_grads(klg)]) * klw
    return ok_gather, float(np.abs(t_kl.numpy() - ref_kl).max()), bool(np.isfinite(t.numpy()).all())


def test_bayesian_dp_host_logic():
    res = _run(_bnn_dp_fn)
    assert all(r[0] for r in res)
    assert all(r[1] < 1e-12 for r in res) and all(r[2] for r in res)


def test_bayesian_generator_block_plan_covers_every_row_once_for_any_sharding():
    """BGM(use_bnn=True).predict: predictive calls are blocks of `bs` GLOBAL rows; ranks own contiguous shards.  Whatever the
    world size and memory chunk, the union of the ranks' calls covers each row exactly once, every call stays inside one
    block and one chunk, and (block, offset-in-block) of a row do not depend on the sharding."""
    from bayesgm_amd.models.bgm_bnn import block_plan
    from bayesgm_amd.parallel import shard_range
    for n, bs, chunk in ((257, 100, 100), (1025, 64, 256), (1000, 100, 10 ** 9), (37, 100, 50), (640, 64, 1)):
        ref = {}
        for world in (1, 2, 3, 8):
            seen = {}
            for r in range(world):
                lo, hi = shard_range(n, r, world)
                for s, e, calls in block_plan(lo, hi - lo, bs, chunk):
                    assert 0 <= s < e <= hi - lo
                    assert (lo + e) % bs == 0 or e == hi - lo                 # chunks end on block boundaries (or at the shard end)
                    assert calls[0][0] == s and calls[-1][1] == e
                    for b, be, blk, off in calls:
                        assert s <= b < be <= e and off + (be - b) <= bs and (lo + b) // bs == blk == (lo + be - 1) // bs
                        for i in range(b, be):
                            g = lo + i
                            assert g not in seen
                            seen[g] = (blk, off + (i - b))
            assert sorted(seen) == list(range(n))
            if not ref:
                ref = seen
            assert seen == ref


def _strong_plan_fn(rank, world):
    """bench.py --scaling strong (BASELINE configs[3]: ONE N-row panel over all GPUs): every rank's share, its first row, the job size."""
    import torch.distributed as dist
    sys.path.insert(0, ROOT)
    import bench
    from bayesgm_amd import parallel
    out = []
    for n in (1_000_000, 1_000_003, 37, 2):
        n_loc, lo, n_total = bench.plan_rows(n, world, rank, "strong")
        assert (lo, lo + n_loc) == parallel.shard_range(n)                 # the rows CausalBGM.predict gives this rank
        t = torch.tensor([n_loc], dtype=torch.int64)
        dist.all_reduce(t)
        spans = [None] * world
        dist.all_gather_object(spans, (lo, lo + n_loc))
        out.append((int(t.item()), n_total, spans))
        # whole-job throughput of a strong-scaling step counts the job's rows once, a weak-scaling one every rank's panel
        assert bench.plan_rows(n, world, rank, "weak") == (n, rank * n, n * world)
    return out


def test_strong_scaling_shards_partition_the_one_panel():
    res = _run(_strong_plan_fn)
    for per_rank in res:
        for total, n_total, spans in per_rank:
            assert total == n_total
            assert spans[0][0] == 0 and spans[-1][1] == n_total
            assert all(spans[i][1] == spans[i + 1][0] for i in range(len(spans) - 1))


def test_strong_scaling_panel_is_the_slice_of_the_single_gpu_panel():
    """make_panel(..., lo, n_gen): rank r of a strong-scaling run holds rows [lo, lo + n) of the panel a one-GPU run of the same
    --rows holds (same generator seed), so the SCALE line at N GPUs is the configs[3] workload, not N different panels."""
    sys.path.insert(0, ROOT)
    import bench
    full = [t.numpy() for t in bench.make_panel(101, 7, seed=0, device="cpu")]
    for world in (2, 3):
        parts = []
        for r in range(world):
            n_loc, lo, n_total = bench.plan_rows(101, world, r, "strong")
            parts.append([t.numpy() for t in bench.make_panel(n_loc, 7, seed=0, device="cpu", lo=lo, n_gen=n_total)])
        for k in range(3):
            assert np.array_equal(np.concatenate([p_[k] for p_ in parts]), full[k])


def _shared_seed_fn(rank, world):
    from bayesgm_amd import parallel
    return (parallel.shared_seed(17), parallel.shared_seed(None))


def test_shared_seed_is_one_value_on_every_rank():
    """random_seed=None is the reference default (cli, main.py): under torch.distributed rank 0's entropy seed is
    broadcast so that initial weights and the replicated EGM warm start agree on every rank; a user seed is kept; a
    single process keeps None (NumPy's own entropy seeding)."""
    from bayesgm_amd import parallel
    assert parallel.shared_seed(None) is None and parallel.shared_seed(5) == 5
    res = _run(_shared_seed_fn)
    assert res[0][0] == res[1][0] == 17
    assert isinstance(res[0][1], int) and res[0][1] == res[1][1]


def test_check_n_mcmc_limits():
    """predict() rejects a nonsensical draw count before any sampling happens; 32768 is no longer a limit."""
    from bayesgm_amd import parallel
    parallel.check_n_mcmc(1)
    parallel.check_n_mcmc(parallel.MAX_INTERVAL_DRAWS)
    parallel.check_n_mcmc(100000)
    for bad in (0, parallel.MAX_INTERVAL_DRAWS + 1):
        with pytest.raises(ValueError, match="n_mcmc"):
            parallel.check_n_mcmc(bad)
