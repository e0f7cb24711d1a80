"""The C routine that draws the host random numbers of an EGM block (bayesgm_amd/host_rng.py) against the call-by-call NumPy loop
it replaces: every index, normal and uniform, and the generator state afterwards, bit for bit."""
import numpy as np
import pytest

from bayesgm_amd import host_rng


@pytest.mark.parametrize("n,B,q,n_it,gd,n_eps", [(20000, 32, 10, 4, 5, 1), (333, 32, 9, 5, 2, 1), (1000, 7, 3, 11, 1, 1), (64, 64, 2, 3, 3, 1)])
def test_block_is_bit_identical_to_numpy(n, B, q, n_it, gd, n_eps):
    if not host_rng._load():
        pytest.skip("no C compiler here")
    np.random.seed(42)
    np.random.normal(size=3)                 # an odd count leaves a cached Gaussian pending
    a = host_rng.egm_block_numpy(n, B, q, n_it, gd, n_eps)
    sa = np.random.get_state(legacy=True)
    np.random.seed(42)
    np.random.normal(size=3)
    b = host_rng.egm_block(n, B, q, n_it, gd, n_eps)
    sb = np.random.get_state(legacy=True)
    for x, y in zip(a, b):
        assert x.dtype == y.dtype and np.array_equal(x, y)
    assert np.array_equal(sa[1], sb[1]) and sa[2:] == sb[2:]
    assert np.random.uniform() == (np.random.set_state(sa), np.random.uniform())[1]


@pytest.mark.parametrize("n", [500, 300000])          # full-permutation draws (C routine) / rejection draws (large panels)
def test_prefetch_pipeline_consumes_the_global_stream_like_the_sequential_loop(n):
    """EgmDrawPipeline: blocks drawn ahead on a private state; the global generator moves only at hand-over, to where the call-by-call
    loop would be; a consumer of np.random between two blocks makes the prefetched block be redrawn from the stream as it then is."""
    B, q, gd = 32, 10, 5
    sizes = [1, 3, 2]
    np.random.seed(11)
    pipe = host_rng.EgmDrawPipeline(n, B, q, gd)
    pipe.request(sizes[0])
    got = []
    for k, n_it in enumerate(sizes):
        before = np.random.get_state()
        blk = pipe.take(sizes[k + 1] if k + 1 < len(sizes) else 0)
        assert not host_rng._same_state(before, np.random.get_state())       # handed over: the global stream advanced by this block only
        got.append(blk)
    tail = np.random.uniform(size=3)
    pipe.close()
    assert pipe.redrawn == 0
    np.random.seed(11)
    for n_it, (idx, z, eps) in zip(sizes, got):
        ridx, rz, reps, st = host_rng.egm_block_from(np.random.get_state(), n, B, q, n_it, gd)
        np.random.set_state(st)
        assert np.array_equal(idx, ridx) and np.array_equal(z, rz) and np.array_equal(eps, reps)
        assert idx.min() >= 0 and idx.max() < n and all(len(np.unique(r)) == B for r in idx.reshape(-1, B))
    assert np.array_equal(tail, np.random.uniform(size=3))
    if n <= 200000:        # the private-state draws are the global-stream draws of the reference's loop
        np.random.seed(11)
        for n_it, (idx, z, eps) in zip(sizes, got):
            ridx, rz, reps = host_rng.egm_block_numpy(n, B, q, n_it, gd)
            assert np.array_equal(idx, ridx) and np.array_equal(z, rz) and np.array_equal(eps, reps)
    # interference: something draws from np.random while block 2 is already prefetched
    np.random.seed(12)
    pipe = host_rng.EgmDrawPipeline(n, B, q, gd)
    pipe.request(2)
    a = pipe.take(2)
    stolen = np.random.normal(size=5)                       # e.g. an evaluation hook
    b = pipe.take(0)
    pipe.close()
    assert pipe.redrawn == 1
    np.random.seed(12)
    ra = host_rng.egm_block_from(np.random.get_state(), n, B, q, 2, gd)
    np.random.set_state(ra[3])
    assert np.array_equal(np.random.normal(size=5), stolen)
    rb = host_rng.egm_block_from(np.random.get_state(), n, B, q, 2, gd)
    assert np.array_equal(a[0], ra[0]) and np.array_equal(b[0], rb[0]) and np.array_equal(b[1], rb[1])

