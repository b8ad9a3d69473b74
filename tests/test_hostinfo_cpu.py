"""hostinfo.usable_cpus / limit_thread_pools: the thread pools follow the CPUs the container may USE (affinity ∩ cgroup
quota), not the CPUs it can see — profiles/r3_notes.md ("host stalls")."""
import builtins
import io
import os

import hostinfo


def test_usable_cpus_is_bounded_by_affinity_and_positive():
    n = hostinfo.usable_cpus()
    assert 1 <= n <= len(os.sched_getaffinity(0))


def test_cgroup_v2_quota_caps_the_count(monkeypatch):
    real_open = builtins.open

    def fake_open(path, *a, **k):
        if path == "/sys/fs/cgroup/cpu.max":
            return io.StringIO("250000 100000\n")          # 2.5 CPUs -> 2
        return real_open(path, *a, **k)
    monkeypatch.setattr(builtins, "open", fake_open)
    monkeypatch.setattr(os, "sched_getaffinity", lambda pid: set(range(256)))
    assert hostinfo.usable_cpus() == 2


def test_unlimited_quota_leaves_the_affinity_count(monkeypatch):
    real_open = builtins.open

    def fake_open(path, *a, **k):
        if path == "/sys/fs/cgroup/cpu.max":
            return io.StringIO("max 100000\n")
        return real_open(path, *a, **k)
    monkeypatch.setattr(builtins, "open", fake_open)
    monkeypatch.setattr(os, "sched_getaffinity", lambda pid: set(range(12)))
    assert hostinfo.usable_cpus() == 12


def test_limit_thread_pools_respects_an_explicit_choice(monkeypatch):
    import torch
    before = torch.get_num_threads()
    monkeypatch.setenv("OMP_NUM_THREADS", "3")
    try:
        assert hostinfo.limit_thread_pools() == 3
        assert torch.get_num_threads() <= 3
    finally:
        torch.set_num_threads(before)
