"""Host-side logic added in round 6 that needs no device: the outcome-cache mode argument (True == 1 in a dict literal),
the data-parallel communicator policy without a process group, the bench's row plan, the library's new exports."""
import numpy as np


def test_outcome_cache_mode_argument():
    from bayesgm_amd.engine import CausalEngine
    f = CausalEngine.outcome_cache_mode
    assert [f(a) for a in (False, "off", 0, np.int64(0))] == [0, 0, 0, 0]
    assert [f(a) for a in ("wave", 1, np.int32(1))] == [1, 1, 1]                 # the integer 1 is the C ABI's mode 1 ...
    assert [f(a) for a in (True, np.bool_(True), "chain", 2)] == [2, 2, 2, 2]    # ... True is "the default cache" = mode 2
    import pytest
    for bad in ("sometimes", 3, -1, None, 1.0):
        with pytest.raises(ValueError):
            f(bad)


def test_fit_communicator_policy_without_a_process_group():
    """single process: no communicator, the classes take the single-process epoch call"""
    from bayesgm_amd import parallel
    assert parallel.fit_comm("cuda:0") is None
    assert parallel.world_size() == 1 and parallel.shard_range(10) == (0, 10)


def test_round6_exports_and_struct_layout():
    import ctypes as C
    from bayesgm_amd import _lib
    a = _lib.BnnMhArgs()
    assert a.block_row0 == 0 and _lib.BnnMhArgs.block_row0.offset == C.sizeof(_lib.BnnMhArgs) - 8      # appended: older callers' layout unchanged
    import subprocess
    out = subprocess.run(["nm", "-D", _lib.LIB_PATH], capture_output=True, text=True).stdout
    for name in ("bgm_comm_create", "bgm_causal_fit_epoch_dp", "bgm_bnn_fit_epoch_dp", "bgm_bgm_set_precision"):
        assert (" T " + name) in out, name


def test_bench_row_plan_and_new_flags():
    import bench
    assert bench.plan_rows(1e6, 8, 3, "weak") == (1000000, 3000000, 8000000)
    assert bench.plan_rows(10, 4, 3, "strong") == (2, 8, 10) and sum(bench.plan_rows(10, 4, r, "strong")[0] for r in range(4)) == 10
