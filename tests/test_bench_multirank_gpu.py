"""bench.py's multi-rank path in front of the driver: `python3 bench.py --gpus 2` (its own launcher, two ranks, the C4 workload,
the factored gradient exchange complete inside every step) rehearsed on ONE GPU over gloo — RCCL refuses two ranks on one device,
and an 8-GPU node is the driver's to use.  The child processes are started by tests/conftest.py (through the supervisor
tests/_rehearsals.py, one job after the other) when the `-m gpu` session is configured, before this pytest process touches the GPU
(a process that has initialised the GPU must not start another program on these boxes); these tests wait for them and check the
relayed JSON lines."""
import json

import pytest

from conftest import collect_rehearsal

pytestmark = pytest.mark.gpu


def _line(name):
    got = collect_rehearsal(name)
    if got is None:
        pytest.skip("the rehearsal child was not started (not a plain `-m gpu` session, or MSGS_NO_BENCH_REHEARSAL=1)")
    rc, out, err = got
    assert rc == 0, err
    lines = [l for l in out.splitlines() if l.startswith("{")]
    assert len(lines) == 1, (out[-2000:], err)
    return json.loads(lines[0]), err


def test_two_rank_rehearsal_of_bench_over_gloo():
    j, _ = _line("self_launch")
    assert j["n_gpus"] == 2 and j["steps"] == 3 and j["scaling"] == "weak" and j["unit"] == "Mpixels/s"
    assert j["config"]["exchange"] == "factored", j["config"]
    assert j["config"]["backend"] == "gloo" and j["config"]["rccl_ranks"] is None
    assert "C4" in j["config"]["workload"] and j["value"] > 0
    ex = j["exchange"]
    assert ex["headline_exchange"] == "factored"
    for k in ("dense_serial_allreduce", "dense_pipelined", "without_exchange"):
        assert ex[k]["ms_per_step"] > 0 and ex[k]["value"] > 0, k
    # bytes a GPU receives per step at two ranks: one peer's {dL/drgb [P,3] | camera centre | pad} row + the ring-equivalent
    # all-reduce of the 11 non-SH floats
    P = 1_000_000
    assert ex["factored_bytes_received_per_gpu"] == 4 * ((3 * P + 4) + 2 * 1 * (11 * P) // 2)
    assert ex["dense_allreduce_bytes"] == 4 * 59 * P
    assert j["two_views_per_rank"]["fwd_bwd_ms_per_view"] > 0


def test_two_rank_rehearsal_under_torch_distributed_run():
    """the driver's launch line for N > 1 (`python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1
    --master-port P bench.py --gpus N ...`): WORLD_SIZE / RANK / LOCAL_RANK come from the launcher, bench.py does not start ranks"""
    j, _ = _line("torchrun")
    assert j["n_gpus"] == 2 and j["steps"] == 3 and j["warmup"] == 1 and j["value"] > 0
    assert j["config"]["exchange"] == "factored" and j["config"]["backend"] == "gloo"


def test_factored_exchange_failing_on_one_rank_lands_both_ranks_on_the_dense_fallback():
    """bench.py's last branch that no real run had ever executed: the factored exchange fails on ONE rank's first step
    (MSGS_BENCH_FAIL_FACTORED=1 -> rank 1), the ranks agree over the CPU group, BOTH replace their step by the dense flat
    all-reduce after every view, and the line says so everywhere it names the exchange."""
    j, _ = _line("fallback")
    assert j["n_gpus"] == 2 and j["value"] > 0 and j["steps"] == 3
    assert j["config"]["exchange"] == "dense_serial_allreduce (fallback)", j["config"]
    assert j["config"]["backend"] == "gloo" and j["config"]["rccl_ranks"] is None
    ex = j["exchange"]
    assert ex["headline_exchange"] == "dense_serial_allreduce (fallback)"
    assert ex["factored_bytes_received_per_gpu"] is None and ex["dense_allreduce_bytes"] == 4 * 59 * 1_000_000
    # the headline now IS the dense serial exchange: the same step measured twice, within a wide margin of each other
    assert 0.5 < ex["dense_serial_allreduce"]["ms_per_step"] / j["ms_per_step"] < 2.0
