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
