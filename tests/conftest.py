import pathlib
import sys

import pytest

ROOT = pathlib.Path(__file__).resolve().parent.parent
if str(ROOT) not in sys.path:
    sys.path.insert(0, str(ROOT))


def cpu_share():
    """CPUs this process may really use: the cgroup quota when there is one (a GPU box shows 256 CPUs and grants 16), else the affinity mask."""
    import os
    n = len(os.sched_getaffinity(0))
    for path, parse in (("/sys/fs/cgroup/cpu.max", lambda t: t.split()), ("/sys/fs/cgroup/cpu/cpu.cfs_quota_us", lambda t: (t, open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read()))):
        try:
            q, per = parse(open(path).read())
            if q.strip() not in ("max", "-1"):
                n = min(n, max(1, int(int(q) / int(per))))
            break
        except (OSError, ValueError):
            continue
    return n


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # The torch oracle works on arrays of a few thousand particles: with torch's default of one thread per visible CPU (128 on a GPU box whose cgroup
    # grants 16) its small ops spend their time in the thread pool - 4.7 s instead of 0.3 s for six substeps at 2,000 particles
    # (tools/cpu_threads_probe.py, profiles/r05_cpu_threads.txt: that, not the GPU, was two thirds of the GPU suite's 700 s).
    import os
    share = cpu_share()
    os.environ.setdefault("OMP_NUM_THREADS", str(share))          # (the C++ oracle port's OpenMP loops: the granted CPUs, not the visible ones)
    try:
        import torch
        torch.set_num_threads(max(1, min(8, share)))
    except ImportError:
        pass


@pytest.fixture(scope="session")
def hip_lib():
    """Build (if stale) and load libsoftmac_hip.so - works without a GPU."""
    from softmac_amd import build, _ffi
    build.build(verbose=False)
    return _ffi.load_library()
