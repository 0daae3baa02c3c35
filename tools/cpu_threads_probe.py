"""What CPU share does this box give, and what does torch's thread count do to the torch oracle at test size?  (GPU suite run time: the oracle dominates it)"""
import os, sys, time, pathlib
ROOT = pathlib.Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
import numpy as np, torch
import helpers as H
print("os.cpu_count", os.cpu_count(), "affinity", len(os.sched_getaffinity(0)), "torch threads", torch.get_num_threads(), "interop", torch.get_num_interop_threads())
for f in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "/sys/fs/cgroup/cpu/cpu.cfs_period_us"):
    try:
        print(f, open(f).read().strip())
    except OSError:
        pass
st = np.load(H.GOLDEN / "grip_state_2k.npz")["state"]
cfg = H.sim_cfg(len(st), n_grid=64, dt=2e-4, ptype=0)
P = H.oracle_params(cfg, 1e-3)
for nt in (torch.get_num_threads(), 16, 8, 4, 1):
    torch.set_num_threads(nt)
    t = time.time()
    orc = H.OracleRollout(P, st).forward(6)
    orc.backward({6: (np.ones((len(st), 3)), None, None, None)})
    print(f"torch threads {nt}: 6 substeps fwd + bwd of the torch oracle at 2,000 particles: {time.time() - t:.2f} s")
