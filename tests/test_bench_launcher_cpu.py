"""bench.py's rank start-up without a GPU (MSGS_BENCH_LAUNCH_ONLY=1 stops each rank after the rendezvous and one gloo
collective): `python3 bench.py --gpus N` must start its own N ranks when no launcher set WORLD_SIZE, relay rank 0's JSON
line, keep working under torch.distributed.run, and refuse a --gpus / WORLD_SIZE mismatch."""
import json
import os
import socket
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def _env(**kw):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT", "MASTER_ADDR")}
    env.update(MSGS_BENCH_LAUNCH_ONLY="1", **kw)
    return env


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def test_plain_invocation_starts_its_own_ranks():
    for n in (2, 4):
        r = subprocess.run([sys.executable, BENCH, "--gpus", str(n), "--steps", "1", "--warmup", "0"], env=_env(),
                           capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, r.stderr[-2000:]
        line = json.loads(r.stdout.strip().splitlines()[-1])
        assert line == {"launch_only": True, "n_gpus": n, "rank_sum": float(n * (n - 1) // 2)}


def test_single_gpu_invocation_stays_in_process():
    r = subprocess.run([sys.executable, BENCH], env=_env(), capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    assert json.loads(r.stdout.strip().splitlines()[-1])["n_gpus"] == 1


def test_under_torch_distributed_run():
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                        "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), BENCH, "--gpus", "2"],
                       env=_env(), capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [json.loads(l) for l in r.stdout.splitlines() if l.startswith("{")]
    assert lines == [{"launch_only": True, "n_gpus": 2, "rank_sum": 1.0}]


def test_world_size_mismatch_is_refused():
    r = subprocess.run([sys.executable, BENCH, "--gpus", "4"], env=_env(WORLD_SIZE="2", RANK="0", LOCAL_RANK="0"),
                       capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and "WORLD_SIZE=2" in r.stderr


def test_a_dead_rank_takes_the_others_down_quickly():
    """rank 1 exits before the rendezvous: the launcher must stop rank 0 (which would otherwise wait for the
    process-group timeout) and return the failing status"""
    import time
    t0 = time.time()
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--steps", "1", "--warmup", "0"], env=_env(MSGS_BENCH_FAIL_RANK="1"),
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 3, (r.returncode, r.stderr[-1000:])
    assert "the other ranks were stopped" in r.stderr
    assert time.time() - t0 < 120


def test_rehearsal_supervisor_runs_its_jobs_one_after_the_other(tmp_path):
    """tests/_rehearsals.py (the supervisor of the `-m gpu` session's child processes): jobs run sequentially, each with its own
    environment, a free port is inserted where asked, `<name>.rc` appears last, a failing or overrunning job does not stop the
    next one."""
    import time
    spec = {"dir": str(tmp_path), "cwd": ROOT, "jobs": [
        {"name": "a", "argv": [sys.executable, "-c", "import os,sys,time; time.sleep(0.5); print(os.environ['X_JOB'], sys.argv[1:])",
                               "--port"], "free_port_arg": "--port", "env": {"X_JOB": "first"}},
        {"name": "b", "argv": [sys.executable, "-c", "import sys; sys.stderr.write('boom'); sys.exit(7)"]},
        {"name": "c", "argv": [sys.executable, "-c", "import time; time.sleep(30)"], "timeout": 0.5},
        {"name": "d", "argv": [sys.executable, "-c", "import os; print(os.environ.get('WORLD_SIZE', 'unset'))"],
         "unset_env": ["WORLD_SIZE"]},
    ]}
    sp = tmp_path / "spec.json"
    sp.write_text(json.dumps(spec))
    t0 = time.time()
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "_rehearsals.py"), str(sp)],
                       env=dict(os.environ, WORLD_SIZE="9"), timeout=120)
    assert r.returncode == 0 and time.time() - t0 < 60
    rc = {n: int((tmp_path / f"{n}.rc").read_text()) for n in "abcd"}
    assert rc == {"a": 0, "b": 7, "c": 124, "d": 0}
    out_a = (tmp_path / "a.out").read_text()
    assert out_a.startswith("first ['--port', '") and int(out_a.split("'")[3]) > 0
    assert (tmp_path / "b.err").read_text() == "boom"
    assert (tmp_path / "d.out").read_text().strip() == "unset"
    # sequential: every .rc is written after the previous job's
    m = [os.path.getmtime(tmp_path / f"{n}.rc") for n in "abcd"]
    assert m == sorted(m)
