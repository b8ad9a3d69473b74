"""bench.py's multi-rank path in front of the driver: `python3 bench.py --gpus 2` (its own launcher, two ranks, the C4 workload,
the factored gradient exchange complete inside every step) rehearsed on ONE GPU over gloo — RCCL refuses two ranks on one device,
and an 8-GPU node is the driver's to use.  The child process is started by tests/conftest.py when the `-m gpu` session is
configured, before this pytest process touches the GPU (a process that has initialised the GPU must not start another program
on these boxes); this test waits for it and checks the relayed JSON line."""
import json

import pytest

from conftest import BENCH_REHEARSAL

pytestmark = pytest.mark.gpu


def test_two_rank_rehearsal_of_bench_over_gloo():
    proc = BENCH_REHEARSAL["proc"]
    if proc is None:
        pytest.skip("the rehearsal child was not started (not a plain `-m gpu` session, or MSGS_NO_BENCH_REHEARSAL=1)")
    try:
        out, _ = proc.communicate(timeout=900)
    except Exception:
        proc.kill()
        raise
    err = open(BENCH_REHEARSAL["log"]).read()[-3000:] if BENCH_REHEARSAL["log"] else ""
    assert proc.returncode == 0, err
    lines = [l for l in out.splitlines() if l.startswith("{")]
    assert len(lines) == 1, (out[-2000:], err)
    j = json.loads(lines[0])
    assert j["n_gpus"] == 2 and j["steps"] == 3 and j["scaling"] == "weak" and j["unit"] == "Mpixels/s"
    assert j["config"]["exchange"] == "factored", j["config"]
    assert j["config"]["backend"] == "gloo" and j["config"]["rccl_ranks"] is None
    assert "C4" in j["config"]["workload"] and j["value"] > 0
    ex = j["exchange"]
    assert ex["headline_exchange"] == "factored"
    for k in ("dense_serial_allreduce", "dense_pipelined", "without_exchange"):
        assert ex[k]["ms_per_step"] > 0 and ex[k]["value"] > 0, k
    assert ex["factored_bytes_received_per_gpu"] > 0 and ex["dense_allreduce_bytes"] == 4 * 59 * 1_000_000
    assert j["two_views_per_rank"]["fwd_bwd_ms_per_view"] > 0


def test_two_rank_rehearsal_under_torch_distributed_run():
    """the driver's launch line for N > 1 (`python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1
    --master-port P bench.py --gpus N ...`): WORLD_SIZE / RANK / LOCAL_RANK come from the launcher, bench.py does not start ranks"""
    proc = BENCH_REHEARSAL["torchrun"]
    if proc is None:
        pytest.skip("the rehearsal child was not started (not a plain `-m gpu` session, or MSGS_NO_BENCH_REHEARSAL=1)")
    try:
        out, _ = proc.communicate(timeout=900)
    except Exception:
        proc.kill()
        raise
    err = open(BENCH_REHEARSAL["torchrun_log"]).read()[-3000:] if BENCH_REHEARSAL["torchrun_log"] else ""
    assert proc.returncode == 0, err
    lines = [l for l in out.splitlines() if l.startswith("{")]
    assert len(lines) == 1, (out[-2000:], err)
    j = json.loads(lines[0])
    assert j["n_gpus"] == 2 and j["steps"] == 3 and j["warmup"] == 1 and j["value"] > 0
    assert j["config"]["exchange"] == "factored" and j["config"]["backend"] == "gloo"
