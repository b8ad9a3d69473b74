import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "ms-gs_amd"), os.path.join(ROOT, "ms-gs_amd", "host"), os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


# thread pools sized for the container's CPU quota, not the host's CPU count (hostinfo.py): the GPU boxes show 256 CPUs and
# grant 16 — an oversized pool gets the whole process parked by the cgroup for most of every 100 ms period
from hostinfo import limit_thread_pools  # noqa: E402
limit_thread_pools(reserve=0)


# The two-rank rehearsal of bench.py's N > 1 path (tests/test_bench_multirank_gpu.py) has to run as a FRESH child process,
# and on the GPU boxes a process that has initialised the GPU must not start another program: the child is therefore started
# HERE, when a `-m gpu` session is configured — before anything in this process touches the GPU (torch.cuda.device_count()
# does not initialise it; torch.cuda.is_available() further down does) — and the test only collects its output.
BENCH_REHEARSAL = {"proc": None, "log": None, "torchrun": None, "torchrun_log": None}


def _start_bench_rehearsal(config):
    import subprocess
    import tempfile
    expr = (config.getoption("markexpr", "") or "").strip()
    if "gpu" not in expr or "not gpu" in expr or os.environ.get("MSGS_NO_BENCH_REHEARSAL") == "1":
        return
    keyword = config.getoption("keyword", "") or ""
    paths = [str(a) for a in config.args]
    if keyword or any(a.endswith(".py") or "::" in a for a in paths):
        # a selection of tests: only when the rehearsal's own file is among them
        if not any("test_bench_multirank_gpu" in a for a in paths):
            return
    try:
        import torch
        if torch.cuda.device_count() < 1:
            return
    except Exception:
        return
    log = tempfile.NamedTemporaryFile(prefix="msgs_bench_rehearsal_", suffix=".log", delete=False)
    env = dict(os.environ, MSGS_BENCH_BACKEND="gloo", MSGS_BENCH_TIMEOUT="900")
    env.pop("WORLD_SIZE", None)
    BENCH_REHEARSAL["proc"] = subprocess.Popen(
        [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1", "--no-cpu-baseline",
         "--no-pyramid"], env=env, stdout=subprocess.PIPE, stderr=log, text=True, cwd=ROOT)
    BENCH_REHEARSAL["log"] = log.name
    # the driver's own way to start N > 1: torch.distributed.run around bench.py (WORLD_SIZE set, no self-launch)
    log2 = tempfile.NamedTemporaryFile(prefix="msgs_bench_rehearsal_torchrun_", suffix=".log", delete=False)
    BENCH_REHEARSAL["torchrun"] = subprocess.Popen(
        [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
         "--master-port", "29533", os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1",
         "--no-cpu-baseline", "--no-pyramid"], env=env, stdout=subprocess.PIPE, stderr=log2, text=True, cwd=ROOT)
    BENCH_REHEARSAL["torchrun_log"] = log2.name


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    _start_bench_rehearsal(config)


def pytest_unconfigure(config):
    for key in ("proc", "torchrun"):
        p = BENCH_REHEARSAL.get(key)
        if p is not None and p.poll() is None:   # the test did not run (deselected / interrupted): do not leave it behind
            p.terminate()


def _has_gpu():
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False


def pytest_collection_modifyitems(config, items):
    # the rehearsal child is already running beside this session (it shares the GPU with the first tests for ~40 s: they check
    # results, not times): its result is collected LAST, so that with `-x` a failure of the rehearsal cannot hide the other tests
    items.sort(key=lambda it: 1 if "test_bench_multirank_gpu" in it.nodeid else 0)
    if _has_gpu():
        return
    skip = pytest.mark.skip(reason="no GPU in this container")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)
