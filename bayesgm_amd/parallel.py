"""One-process-per-GPU helpers (torch.distributed; backend "nccl" is RCCL on ROCm).

The hot path shards by observation: rows are split into
contiguous per-rank ranges, MCMC chains never communicate, and the only
exchanges are (C1) the network-gradient all-reduce in fit, (C3) the ADRF
partial sums [n_doses x n_keep] once per predict and gathers of per-row results.
"""
import os

import torch
import torch.distributed as dist


def is_dist():
    return dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1


def rank():
    return dist.get_rank() if is_dist() else 0


def world_size():
    return dist.get_world_size() if is_dist() else 1


def shard_range(n, r=None, w=None):
    """Contiguous row range [lo, hi) owned by rank r of w (sizes differ by at most 1)."""
    r = rank() if r is None else r
    w = world_size() if w is None else w
    base, rem = divmod(n, w)
    lo = r * base + min(r, rem)
    return lo, lo + base + (1 if r < rem else 0)


def _device_view(t):
    """RCCL moves device memory only.  A host tensor handed to a collective is staged through the current HIP device when the backend is
    nccl (returns (device tensor, host tensor to copy the result back into)); BGM_STRICT_COLLECTIVES=1 turns that case into an error
    under ANY backend, so that the two-rank gloo runs of a one-GPU box find what an RCCL run would have to stage."""
    if t.is_cuda:
        return t, None
    if os.environ.get("BGM_STRICT_COLLECTIVES") == "1":
        raise RuntimeError("collective on a host tensor of shape %s: the data-parallel path keeps its exchanges in HBM" % (tuple(t.shape),))
    if dist.get_backend() != "nccl":
        return t, None
    return t.to(torch.device("cuda", torch.cuda.current_device())), t


def all_reduce_sum_(t):
    if is_dist():
        d, host = _device_view(t)
        dist.all_reduce(d, op=dist.ReduceOp.SUM)
        if host is not None:
            host.copy_(d)
    return t


def broadcast_(t, src=0):
    """In-place broadcast of a tensor from rank `src` (a host-side random draw every rank must agree on)."""
    if is_dist():
        d, host = _device_view(t)
        dist.broadcast(d, src)
        if host is not None:
            host.copy_(d)
    return t


def all_reduce_max_(t):
    """In-place MAX all-reduce (no-op when not distributed)."""
    if is_dist():
        d, host = _device_view(t)
        dist.all_reduce(d, op=dist.ReduceOp.MAX)
        if host is not None:
            host.copy_(d)
    return t


def all_gather_rows(t, n_total):
    """Concatenate per-rank row blocks (first dim) of possibly unequal length."""
    if not is_dist():
        return t
    w = world_size()
    t, host = _device_view(t)
    sizes = [shard_range(n_total, r, w)[1] - shard_range(n_total, r, w)[0] for r in range(w)]
    mx = max(sizes)
    pad = torch.zeros((mx,) + tuple(t.shape[1:]), dtype=t.dtype, device=t.device)
    pad[:t.shape[0]] = t
    outs = [torch.empty_like(pad) for _ in range(w)]
    dist.all_gather(outs, pad)
    res = torch.cat([o[:s] for o, s in zip(outs, sizes)], dim=0)
    return res if host is None else res.cpu()


def all_gather_rows_var(t):
    """Concatenate per-rank row blocks (first dim) whose lengths are only known locally (rank order)."""
    if not is_dist():
        return t
    w = world_size()
    t, host = _device_view(t)
    n = torch.tensor([t.shape[0]], dtype=torch.int64, device=t.device)
    ns = [torch.zeros_like(n) for _ in range(w)]
    dist.all_gather(ns, n)
    sizes = [int(s.item()) for s in ns]
    mx = max(max(sizes), 1)
    pad = torch.zeros((mx,) + tuple(t.shape[1:]), dtype=t.dtype, device=t.device)
    pad[:t.shape[0]] = t
    outs = [torch.empty_like(pad) for _ in range(w)]
    dist.all_gather(outs, pad)
    res = torch.cat([o[:s] for o, s in zip(outs, sizes)], dim=0)
    return res if host is None else res.cpu()


def shared_seed(random_seed):
    """Seed every rank derives its host RNG streams and initial weights from.

    A user seed is used as is.  With ``random_seed=None`` (the reference's default) a single process keeps NumPy's
    entropy seeding (returns None); under torch.distributed rank 0 draws a seed and broadcasts it, because the
    replicated phases (weight initialisation, EGM warm start, evaluation noise) must be identical on every rank
    before the gradient all-reduce of `fit` can keep the replicas identical."""
    if random_seed is not None or not is_dist():
        return random_seed
    import numpy as np
    box = [int(np.random.SeedSequence().generate_state(1)[0] & 0x7FFFFFFF) if rank() == 0 else None]
    dist.broadcast_object_list(box, src=0)
    return int(box[0])

MAX_INTERVAL_DRAWS = 1 << 24   # kept draws per row; bgm_row_mean_quantiles sorts a row in LDS up to 32768 values and selects the
#                                order statistics by radix passes beyond (aux_kernels.hip); the bound is the iteration counter's headroom


def check_n_mcmc(n_mcmc):
    """predict() reduces n_mcmc kept draws per row to mean + posterior interval; reject a nonsensical count before the sampler runs."""
    if int(n_mcmc) < 1 or int(n_mcmc) > MAX_INTERVAL_DRAWS:
        raise ValueError("n_mcmc must be in [1, %d]; got %r" % (MAX_INTERVAL_DRAWS, n_mcmc))
