import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (gfx950) device")


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


def _have_gpu():
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False


def pytest_collection_modifyitems(config, items):
    # GPU tests must FAIL (not skip) on a GPU box if the extension is missing; on a box
    # without a GPU they are deselected by `-m "not gpu"`; if someone runs them anyway
    # without a device they are skipped with a clear reason.
    if _have_gpu():
        return
    skip = pytest.mark.skip(reason="no HIP device visible")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


def run_two_ranks(script, timeout=240, attempts=2, extra_args=(), backend="gloo"):
    """Launch scripts/<script> with two ranks (torch.distributed.run on 127.0.0.1, a free port) in its own process group; a run that
    does not finish in `timeout` seconds is killed WITH its ranks and retried once.  backend "gloo": both ranks on GPU 0 (what a
    one-GPU box can run); backend "nccl": rank r on GPU r over RCCL (needs two devices).  Returns the CompletedProcess."""
    import signal
    import socket
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    if backend == "gloo":
        env["BGM_DEVICE"] = "0"
        env["BGM_STRICT_COLLECTIVES"] = "1"  # a host tensor in a collective is an error here: RCCL would have to stage it (parallel._device_view)
    else:
        env.pop("BGM_DEVICE", None)          # the scripts fall back to LOCAL_RANK
    last = None
    for attempt in range(attempts):
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
        proc = subprocess.Popen([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
                                 "127.0.0.1", "--master-port", str(port), os.path.join(root, "scripts", script), backend] + list(extra_args),
                                cwd=root, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, start_new_session=True)
        try:
            out, err = proc.communicate(timeout=timeout)
        except subprocess.TimeoutExpired:
            os.killpg(proc.pid, signal.SIGKILL)
            proc.communicate()
            last = "two-rank run of %s did not finish within %d s (attempt %d)" % (script, timeout, attempt)
            continue
        return subprocess.CompletedProcess(proc.args, proc.returncode, out, err)
    raise AssertionError(last)
