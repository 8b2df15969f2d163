"""hostcpu: the intra-op pool is fitted to what the cgroup grants (never grown), and explicit settings are respected."""
import builtins
import io

import torch

from deeprank_gnn_amd import hostcpu


def _fake_open(files):
    real = builtins.open

    def opener(path, *a, **k):
        if path in files:
            if files[path] is None:
                raise OSError(path)
            return io.StringIO(files[path])
        return real(path, *a, **k)
    return opener


def test_quota_of_both_cgroup_versions(monkeypatch):
    monkeypatch.setattr(builtins, "open", _fake_open({"/sys/fs/cgroup/cpu.max": "1600000 100000\n"}))
    assert hostcpu.cpu_quota() == 16.0
    monkeypatch.setattr(builtins, "open", _fake_open({"/sys/fs/cgroup/cpu.max": "max 100000\n"}))
    assert hostcpu.cpu_quota() is None
    v1 = {"/sys/fs/cgroup/cpu.max": None, "/sys/fs/cgroup/cpu/cpu.cfs_quota_us": "250000\n",
          "/sys/fs/cgroup/cpu/cpu.cfs_period_us": "100000\n"}
    monkeypatch.setattr(builtins, "open", _fake_open(v1))
    assert hostcpu.cpu_quota() == 2.5
    v1["/sys/fs/cgroup/cpu/cpu.cfs_quota_us"] = "-1\n"
    assert hostcpu.cpu_quota() is None
    assert hostcpu.granted_cpus() >= 1


def test_pool_is_fitted_once_and_never_grown(monkeypatch):
    before = torch.get_num_threads()
    try:
        monkeypatch.delenv("OMP_NUM_THREADS", raising=False)
        monkeypatch.delenv("DRGNN_KEEP_TORCH_THREADS", raising=False)
        monkeypatch.setattr(hostcpu, "granted_cpus", lambda: 6)
        torch.set_num_threads(2)
        assert hostcpu.fit_torch_threads(force=True) == 2               # 2 < 6 // 2: left alone
        torch.set_num_threads(max(4, before))
        monkeypatch.setenv("LOCAL_WORLD_SIZE", "1")
        assert hostcpu.fit_torch_threads(force=True) == 3
        monkeypatch.setenv("LOCAL_WORLD_SIZE", "8")                     # 8 ranks share the quota: one thread each
        assert hostcpu.fit_torch_threads(force=True) == 1
        torch.set_num_threads(4)
        monkeypatch.setenv("OMP_NUM_THREADS", "4")                      # an explicit setting wins
        assert hostcpu.fit_torch_threads(force=True) == 4
        monkeypatch.delenv("OMP_NUM_THREADS")
        monkeypatch.setenv("DRGNN_KEEP_TORCH_THREADS", "1")
        assert hostcpu.fit_torch_threads(force=True) == 4
        assert hostcpu.fit_torch_threads() == 4                         # later calls do nothing
    finally:
        torch.set_num_threads(before)
