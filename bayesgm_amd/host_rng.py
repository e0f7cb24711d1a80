"""Host random numbers of a block of EGM iterations, from NumPy's legacy global generator in the reference's order
(causalbgm/base.py:399-416), drawn by a small C routine (csrc/host/host_rng.c).

np.random.choice(n, B, replace=False) permutes all n indices per call; at the tutorial's n = 20 000 that is ~0.3 ms, 180 000
times per warm start -- more than the GPU side of the whole fit once the step kernels were rebuilt.  The C routine runs the
same Fisher-Yates sweeps, polar-method normals and 53-bit uniforms on the same MT19937 state, bit-identically (outputs and
final generator state: tests/test_host_rng.py), about 1.4x faster; the floor is the ~24 000 sequential Mersenne-Twister draws
per index batch that stream parity with the reference prescribes.  This is host bookkeeping, not the GPU product path: without
a C compiler the NumPy loop below is used (same numbers)."""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SRC = os.path.join(_HERE, "csrc", "host", "host_rng.c")
_SO = os.path.join(_HERE, "libbgm_hostrng.so")
_lib = None


def build(force=False):
    if force or not os.path.exists(_SO) or os.path.getmtime(_SRC) > os.path.getmtime(_SO):
        # link to a private name and rename: ranks of one torchrun job reach this at the same time, and a rank must never map a
        # half-written library (os.replace is atomic; whoever loses the race just replaces an identical file)
        tmp = "%s.%d.tmp" % (_SO, os.getpid())
        subprocess.check_call(["gcc", "-O3", "-shared", "-fPIC", "-o", tmp, _SRC, "-lm"])
        os.replace(tmp, _SO)
    return _SO


def _load():
    global _lib
    if _lib is None:
        try:
            _lib = C.CDLL(build())
            _lib.bgm_host_egm_block.restype = C.c_int
        except Exception:            # no compiler / read-only tree: NumPy path
            _lib = False
    return _lib


def egm_block_numpy(n, batch_size, q, n_it, g_d_freq, n_eps=1):
    """The reference's draw order, call by call."""
    steps = g_d_freq + 1
    idx = np.empty((n_it, steps, batch_size), np.int32)
    z = np.empty((n_it, steps, batch_size, q), np.float32)
    eps = np.empty((n_it, g_d_freq, n_eps), np.float64)
    mean = np.zeros(q)
    for i in range(n_it):
        for j in range(g_d_freq):
            idx[i, j] = np.random.choice(n, batch_size, replace=False)
            z[i, j] = np.random.normal(mean, 1.0, (batch_size, q)).astype(np.float32)
            eps[i, j] = np.random.uniform(0.0, 1.0, size=n_eps)
        z[i, g_d_freq] = np.random.normal(mean, 1.0, (batch_size, q)).astype(np.float32)
        idx[i, g_d_freq] = np.random.choice(n, batch_size, replace=False)
    return idx, z, eps


def egm_block(n, batch_size, q, n_it, g_d_freq, n_eps=1):
    """-> idx [n_it, g_d_freq + 1, B] int32, z [n_it, g_d_freq + 1, B, q] float32, eps [n_it, g_d_freq, n_eps] float64; advances
    np.random's global state exactly as the call-by-call loop does."""
    lib = _load()
    if not lib or n_it == 0:
        return egm_block_numpy(n, batch_size, q, n_it, g_d_freq, n_eps)
    st = np.random.get_state(legacy=True)
    key = np.ascontiguousarray(st[1], dtype=np.uint32).copy()
    pos, has_gauss, gauss = C.c_int(int(st[2])), C.c_int(int(st[3])), C.c_double(float(st[4]))
    steps = g_d_freq + 1
    idx = np.empty((n_it, steps, batch_size), np.int32)
    z = np.empty((n_it, steps, batch_size, q), np.float32)
    eps = np.empty((n_it, g_d_freq, n_eps), np.float64)
    rc = lib.bgm_host_egm_block(key.ctypes.data_as(C.c_void_p), C.byref(pos), C.byref(has_gauss), C.byref(gauss), int(n), int(batch_size),
                                int(q), int(n_it), int(g_d_freq), int(n_eps), idx.ctypes.data_as(C.c_void_p),
                                z.ctypes.data_as(C.c_void_p), eps.ctypes.data_as(C.c_void_p))
    if rc != 0:
        raise ValueError("bgm_host_egm_block: bad argument")
    np.random.set_state(("MT19937", key, pos.value, has_gauss.value, gauss.value))
    return idx, z, eps


# ---------------------------------------------------------------------------------------------------------------------------
# The same draws on a PRIVATE copy of the legacy MT19937 state (what the prefetching worker thread of egm_init uses): nothing here
# reads or writes np.random's global generator.
def egm_block_from(state, n, batch_size, q, n_it, g_d_freq, n_eps=1):
    """-> (idx, z, eps, state_after): the draws egm_block would make if np.random's state were `state` (a legacy get_state() tuple)."""
    lib = _load()
    steps = g_d_freq + 1
    small = n <= 200000 or batch_size * 20 > n           # np.random.choice's full permutation (bit-identical to the reference)
    if lib and n_it and small:
        key = np.ascontiguousarray(state[1], dtype=np.uint32).copy()
        pos, has_gauss, gauss = C.c_int(int(state[2])), C.c_int(int(state[3])), C.c_double(float(state[4]))
        idx = np.empty((n_it, steps, batch_size), np.int32)
        z = np.empty((n_it, steps, batch_size, q), np.float32)
        eps = np.empty((n_it, g_d_freq, n_eps), np.float64)
        rc = lib.bgm_host_egm_block(key.ctypes.data_as(C.c_void_p), C.byref(pos), C.byref(has_gauss), C.byref(gauss), int(n), int(batch_size),
                                    int(q), int(n_it), int(g_d_freq), int(n_eps), idx.ctypes.data_as(C.c_void_p),
                                    z.ctypes.data_as(C.c_void_p), eps.ctypes.data_as(C.c_void_p))
        if rc != 0:
            raise ValueError("bgm_host_egm_block: bad argument")
        return idx, z, eps, ("MT19937", key, pos.value, has_gauss.value, gauss.value)
    rs = np.random.RandomState()
    rs.set_state(state)

    def choice():
        if small:
            return rs.choice(n, batch_size, replace=False)
        while True:                                       # large panels: k distinct indices by rejection (same law, O(k))
            c = rs.randint(0, n, size=batch_size)
            if len(np.unique(c)) == batch_size:
                return c
    idx = np.empty((n_it, steps, batch_size), np.int32)
    z = np.empty((n_it, steps, batch_size, q), np.float32)
    eps = np.empty((n_it, g_d_freq, n_eps), np.float64)
    mean = np.zeros(q)
    for i in range(n_it):
        for j in range(g_d_freq):
            idx[i, j] = choice()
            z[i, j] = rs.normal(mean, 1.0, (batch_size, q)).astype(np.float32)
            eps[i, j] = rs.uniform(0.0, 1.0, size=n_eps) if n_eps > 1 else rs.uniform(0.0, 1.0)
        z[i, g_d_freq] = rs.normal(mean, 1.0, (batch_size, q)).astype(np.float32)
        idx[i, g_d_freq] = choice()
    return idx, z, eps, rs.get_state()


def egm_rank_share(idx, z, n_total, n_loc, b_loc, rank):
    """Data-parallel warm start: every rank draws the GLOBAL minibatches from the shared stream (np.random stays in lockstep on all
    ranks and moves as in a single-process run) and keeps slots [rank * b_loc, (rank + 1) * b_loc) of each: their prior samples as
    drawn, their panel rows mapped onto the rank's own n_loc rows (floor(idx * n_loc / n_total): uniform over the shard)."""
    sl = slice(rank * b_loc, (rank + 1) * b_loc)
    loc = (idx[:, :, sl].astype(np.int64) * int(n_loc)) // int(n_total)
    return np.ascontiguousarray(loc.astype(np.int32)), np.ascontiguousarray(z[:, :, sl])


def _same_state(a, b):
    return a[2] == b[2] and a[3] == b[3] and a[4] == b[4] and np.array_equal(a[1], b[1])


class EgmDrawPipeline(object):
    """Host random numbers of the EGM warm start, one block of iterations ahead of the GPU.  The worker thread draws block k + 1 on
    a private copy of the state block k ended in; np.random's GLOBAL state is moved only on the caller's thread, when a block is
    handed over (`take`), to exactly where the reference's sequential loop would have left it.  If anything consumed np.random in
    between (an evaluation hook, user code), the prefetched block is discarded and redrawn from the stream as it is now; after an
    exception the global state is the one after the last block that was actually used."""

    def __init__(self, n, batch_size, q, g_d_freq, n_eps=1):
        from concurrent.futures import ThreadPoolExecutor
        self._args = (n, batch_size, q)
        self._tail = (g_d_freq, n_eps)
        self._pool = ThreadPoolExecutor(max_workers=1)
        self._pending = None
        self.redrawn = 0

    def _draw(self, n_it, state):
        n, b, q = self._args
        return egm_block_from(state, n, b, q, n_it, *self._tail)

    def request(self, n_it):
        self._base = np.random.get_state()
        self._n_it = n_it
        self._pending = self._pool.submit(self._draw, n_it, self._base)

    def take(self, next_n_it=0):
        idx, z, eps, after = self._pending.result()
        now = np.random.get_state()
        if not _same_state(now, self._base):
            idx, z, eps, after = self._draw(self._n_it, now)
            self.redrawn += 1
        np.random.set_state(after)
        self._pending = None
        if next_n_it:
            self.request(next_n_it)
        return idx, z, eps

    def close(self):
        self._pool.shutdown(wait=True)

