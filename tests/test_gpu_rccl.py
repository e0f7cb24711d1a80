"""The data-parallel path over RCCL (torch.distributed backend "nccl"), one rank per GPU.  These tests need two HIP devices: on the
one-GPU development / test boxes they SKIP (the same scripts run there over gloo with both ranks on GPU 0: tests/test_gpu_fit.py,
test_gpu_egm.py, test_gpu_bnn.py, test_gpu_bgm.py, test_gpu_identifiable.py); on a multi-GPU node they are the first place RCCL executes.
SURVEY.md 8e; reference: the minibatch loops of causalbgm/base.py:434-532 and bgm/base.py:343-442, and predict's block loop :573-668."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _need_two():
    import torch
    if torch.cuda.device_count() < 2:
        pytest.skip("RCCL tests need two HIP devices; this box has %d" % torch.cuda.device_count())


def _lines(r):
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-3000:])
    return [json.loads(l) for l in r.stdout.splitlines() if l.startswith("{")]


@pytest.mark.parametrize("script,key", [("dp_causal_smoke.py", "spread"), ("dp_bnn_smoke.py", "spread"),
                                        ("dp_bgm_fit_smoke.py", "param_spread"), ("dp_bgm_bnn_smoke.py", "param_spread"),
                                        ("dp_ident_smoke.py", None), ("dp_ident_bnn_smoke.py", "spread")])
def test_two_rank_run_over_rccl(script, key):
    """fit + predict with the gradient / ADRF all-reduces on RCCL: replicas stay bit-identical (the scripts assert it too)"""
    _need_two()
    from conftest import run_two_ranks
    rows = _lines(run_two_ranks(script, timeout=600, backend="nccl"))
    assert len(rows) == 2
    if key:
        assert all(r_[key] == 0.0 for r_ in rows), rows


def test_rccl_fit_runs_inside_the_library_and_equals_the_gloo_host_loop():
    """Over RCCL the epoch is ONE library call per rank with the gradient all-reduce enqueued from C++ (bgm_causal_fit_epoch_dp /
    bgm_bnn_fit_epoch_dp on a communicator of the library's own); over gloo (both ranks on one device) the per-minibatch host loop
    with torch's all-reduce between the calls.  Same minibatches, same sums of two terms: the trained ADRF agrees."""
    _need_two()
    from conftest import run_two_ranks
    for script in ("dp_causal_smoke.py", "dp_bnn_smoke.py"):
        a = _lines(run_two_ranks(script, timeout=600, backend="nccl"))
        b = _lines(run_two_ranks(script, timeout=600, backend="gloo"))
        assert all(r_["fit_path"].startswith("library_epoch_dp") for r_ in a), a[0]["fit_path"]
        assert all(r_["fit_path"] == "host_loop" for r_ in b), b[0]["fit_path"]
        np.testing.assert_allclose(a[0]["adrf"], b[0]["adrf"], rtol=0, atol=2e-5)


def test_rccl_predict_equals_gloo_predict():
    """the untrained, seeded predict of dp_causal_smoke.py does not depend on the transport: RCCL on two devices = gloo on one"""
    _need_two()
    from conftest import run_two_ranks
    a = _lines(run_two_ranks("dp_causal_smoke.py", timeout=600, backend="nccl"))[0]
    b = _lines(run_two_ranks("dp_causal_smoke.py", timeout=600, backend="gloo"))[0]
    np.testing.assert_allclose(a["adrf_untrained"], b["adrf_untrained"], rtol=0, atol=1e-6)
    np.testing.assert_allclose(a["interval_untrained"], b["interval_untrained"], rtol=0, atol=1e-6)


def test_bare_bench_over_rccl():
    """`python bench.py --gpus 2` from a plain shell: two ranks, RCCL, both inside the collective before anything is timed"""
    _need_two()
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT", "BGM_BENCH_SINGLE_DEVICE",
                                                            "BGM_BENCH_BACKEND")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--rows", "65536", "--burn-in", "100", "--n-mcmc", "50",
                        "--steps", "1", "--warmup", "1"], cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    d = _lines(r)[-1]
    assert d["n_gpus"] == 2 and d["n_ranks_in_collective"] == 2 and d["config"]["rows_total"] == 131072
    assert d["adrf_allreduce_ms"] is not None and d["value"] > 0
