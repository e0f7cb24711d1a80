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


class DeviceComm:
    """An RCCL communicator owned by the HIP library (csrc/comm_api.hip): what the data-parallel epoch calls
    (bgm_causal_fit_epoch_dp / bgm_bnn_fit_epoch_dp) enqueue their gradient all-reduce on -- one ncclAllReduce per minibatch between
    the gradient kernels and the Adam step, issued from C++ on the library's stream, no Python in between.

    ``DeviceComm(device)`` under torch.distributed: rank 0 asks RCCL for a unique id, the 128 bytes are broadcast over the process
    group, every rank joins with its device.  ``DeviceComm(device, world=1)``: a one-rank communicator (tests, single process)."""

    def __init__(self, device, world=None, rank_=None):
        import ctypes as C
        import numpy as np
        from . import _lib
        self._lib = _lib
        lib = _lib.load()
        dev = torch.device(device) if not isinstance(device, torch.device) else device
        index = dev.index if dev.index is not None else torch.cuda.current_device()
        w = world_size() if world is None else int(world)
        r = rank() if rank_ is None else int(rank_)
        ident = np.zeros(128, np.uint8)
        if r == 0:
            _lib.check(lib.bgm_comm_unique_id(ident.ctypes.data_as(C.c_void_p)), "bgm_comm_unique_id")
        if w > 1:
            t = torch.from_numpy(ident).to(torch.device("cuda", index))
            broadcast_(t, 0)
            ident = t.cpu().numpy()
        h = C.c_void_p()
        with torch.cuda.device(index):
            _lib.check(lib.bgm_comm_create(index, ident.ctypes.data_as(C.c_void_p), w, r, C.byref(h)), "bgm_comm_create")
        self.handle, self.world, self.rank, self.device_index = h, w, r, index

    def info(self):
        import ctypes as C
        w, r, buf = C.c_int32(), C.c_int32(), C.create_string_buffer(256)
        self._lib.check(self._lib.load().bgm_comm_info(self.handle, C.byref(w), C.byref(r), buf, 256), "bgm_comm_info")
        return {"world": w.value, "rank": r.value, "library": buf.value.decode()}

    def all_reduce_sum_(self, t):
        """In-place sum of a float32 device tensor over the ranks, on torch's current stream."""
        import ctypes as C
        assert t.is_cuda and t.dtype == torch.float32 and t.is_contiguous()
        self._lib.check(self._lib.load().bgm_comm_all_reduce_f32(self.handle, t.data_ptr(), t.numel(),
                                                                 C.c_void_p(torch.cuda.current_stream(t.device).cuda_stream)), "bgm_comm_all_reduce_f32")
        return t

    def close(self):
        if getattr(self, "handle", None) is not None and self.handle.value:
            torch.cuda.synchronize(self.device_index)
            self._lib.load().bgm_comm_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


_fit_comms = {}


def fit_comm(device):
    """The communicator of the data-parallel epoch calls for this process's `device`, or None when the minibatch loop stays in Python:
    single process, or a process group whose ranks do not each own a GPU (backend gloo: the two-rank runs on ONE device of the test
    suite -- RCCL refuses two ranks on one device).  BGM_DP_HOST_LOOP=1 forces the host loop (A/B)."""
    if not is_dist() or dist.get_backend() != "nccl" or os.environ.get("BGM_DP_HOST_LOOP") == "1":
        return None
    dev = torch.device(device) if not isinstance(device, torch.device) else device
    key = dev.index if dev.index is not None else torch.cuda.current_device()
    if key not in _fit_comms:
        _fit_comms[key] = DeviceComm(torch.device("cuda", key))
    return _fit_comms[key]


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
