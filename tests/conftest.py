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


# Child processes of a `-m gpu` session (the two-rank rehearsals of bench.py's N > 1 path, the fallback rehearsal, the
# MSGS_BLOCKING_SYNC run: tests/test_bench_multirank_gpu.py, tests/test_blocking_sync_gpu.py) have to be FRESH processes, and on
# the GPU boxes a process that has initialised the GPU must not start another program: ONE supervisor (tests/_rehearsals.py,
# which never touches the GPU) is therefore started HERE, when the session is configured — before anything in this process
# touches the GPU (torch.cuda.device_count() does not initialise it; torch.cuda.is_available() further down does) — and runs
# the jobs one after the other (they share the GPU with the session's tests, which check results, not times; two 1 M-Gaussian
# bench runs at once beside the max-size tests were a memory risk).  The tests only collect `<name>.rc/.out/.err`.
REHEARSALS = {"proc": None, "dir": None, "jobs": ()}
_BENCH_ARGS = ["--gpus", "2", "--steps", "3", "--warmup", "1", "--no-cpu-baseline", "--no-pyramid"]


def _start_rehearsals(config):
    import json
    import subprocess
    import tempfile
    expr = (config.getoption("markexpr", "") or "").strip()
    if "gpu" not in expr or "not gpu" in expr or os.environ.get("MSGS_NO_BENCH_REHEARSAL") == "1":
        return
    keyword = config.getoption("keyword", "") or ""
    paths = [str(a) for a in config.args]
    want = {"bench": True, "blocking": True}
    if keyword or any(a.endswith(".py") or "::" in a for a in paths):
        # a selection of tests: only the jobs whose collecting file is among them
        want = {"bench": any("test_bench_multirank_gpu" in a for a in paths),
                "blocking": any("test_blocking_sync_gpu" in a for a in paths)}
    if not any(want.values()):
        return
    try:
        import torch
        if torch.cuda.device_count() < 1:
            return
    except Exception:
        return
    out_dir = tempfile.mkdtemp(prefix="msgs_rehearsals_")
    bench = os.path.join(ROOT, "bench.py")
    gloo = {"MSGS_BENCH_BACKEND": "gloo", "MSGS_BENCH_TIMEOUT": "900"}
    jobs = []
    if want["blocking"]:
        jobs.append({"name": "blocking_sync", "argv": [sys.executable, os.path.join(ROOT, "tests", "child_blocking_sync.py")],
                     "env": {"MSGS_BLOCKING_SYNC": "1"}, "timeout": 600})
    if want["bench"]:
        jobs += [
            # `python3 bench.py --gpus 2`: its own launcher
            {"name": "self_launch", "argv": [sys.executable, bench] + _BENCH_ARGS, "env": gloo, "unset_env": ["WORLD_SIZE"],
             "timeout": 900},
            # the driver's own way to start N > 1: torch.distributed.run around bench.py (WORLD_SIZE set, no self-launch)
            {"name": "torchrun", "argv": [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                                          "--master-addr", "127.0.0.1", "--master-port", bench] + _BENCH_ARGS,
             "free_port_arg": "--master-port", "env": gloo, "unset_env": ["WORLD_SIZE"], "timeout": 900},
            # rank 1 fails its first factored step: both ranks must land on the dense fallback and the line must say so
            {"name": "fallback", "argv": [sys.executable, bench] + _BENCH_ARGS + ["--no-two-view"],
             "env": dict(gloo, MSGS_BENCH_FAIL_FACTORED="1"), "unset_env": ["WORLD_SIZE"], "timeout": 900},
        ]
    spec = os.path.join(out_dir, "spec.json")
    with open(spec, "w") as f:
        json.dump({"dir": out_dir, "cwd": ROOT, "jobs": jobs}, f)
    REHEARSALS["proc"] = subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "_rehearsals.py"), spec], cwd=ROOT,
                                          start_new_session=True)
    REHEARSALS["dir"] = out_dir
    REHEARSALS["jobs"] = tuple(j["name"] for j in jobs)


def collect_rehearsal(name, timeout=1500.0):
    """(returncode, stdout, stderr tail) of job `name`, waiting for it; None when the supervisor was not started for it"""
    import time
    if REHEARSALS["proc"] is None or name not in REHEARSALS["jobs"]:
        return None
    d = REHEARSALS["dir"]
    rc_file = os.path.join(d, name + ".rc")
    deadline = time.time() + timeout
    while not os.path.exists(rc_file):
        if REHEARSALS["proc"].poll() is not None and not os.path.exists(rc_file):
            raise RuntimeError(f"the rehearsal supervisor exited ({REHEARSALS['proc'].returncode}) without running '{name}'")
        if time.time() > deadline:
            raise TimeoutError(f"rehearsal '{name}' did not finish within {timeout:.0f} s")
        time.sleep(0.5)
    rc = int(open(rc_file).read().strip())
    out = open(os.path.join(d, name + ".out")).read()
    err = open(os.path.join(d, name + ".err")).read()[-4000:]
    return rc, out, err


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    _start_rehearsals(config)


def pytest_unconfigure(config):
    import shutil
    import signal
    p = REHEARSALS.get("proc")
    if p is not None and p.poll() is None:   # the tests did not collect everything (deselected / interrupted): the supervisor
        try:                                 # and the job it is running are OUR process group (start_new_session): stop them
            os.killpg(p.pid, signal.SIGTERM)
        except Exception:
            p.terminate()
    if REHEARSALS.get("dir") and (p is None or p.poll() == 0):
        shutil.rmtree(REHEARSALS["dir"], ignore_errors=True)


def _has_gpu():
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False


def pytest_collection_modifyitems(config, items):
    # the rehearsal child is already running beside this session (it shares the GPU with the first tests for ~40 s: they check
    # results, not times): its result is collected LAST, so that with `-x` a failure of the rehearsal cannot hide the other tests
    items.sort(key=lambda it: 1 if ("test_bench_multirank_gpu" in it.nodeid or "test_blocking_sync_gpu" in it.nodeid) else 0)
    if _has_gpu():
        return
    skip = pytest.mark.skip(reason="no GPU in this container")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)
