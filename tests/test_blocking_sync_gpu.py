"""MSGS_BLOCKING_SYNC=1 (instance count through a 16-byte device-to-host copy + hipStreamSynchronize instead of the polled pinned
words) together with deferred forwards on ONE stream: the copy is enqueued at resolve time and reads device words inside the
launch's stage-1 scratch, which therefore has to outlive the launch.  The switch is latched per process: the run
happens in a child (tests/child_blocking_sync.py) started by tests/conftest.py before this process touches the GPU."""
import pytest

from conftest import collect_rehearsal

pytestmark = pytest.mark.gpu


def test_deferred_forwards_on_one_stream_with_the_blocking_count_readback():
    got = collect_rehearsal("blocking_sync")
    if got is None:
        pytest.skip("the child was not started (not a plain `-m gpu` session, or MSGS_NO_BENCH_REHEARSAL=1)")
    rc, out, err = got
    assert rc == 0, err
    assert "BLOCKING_SYNC_OK" in out, (out[-2000:], err)
