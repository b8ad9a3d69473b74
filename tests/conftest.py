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


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def _has_gpu():
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False


def pytest_collection_modifyitems(config, items):
    if _has_gpu():
        return
    skip = pytest.mark.skip(reason="no GPU in this container")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)
