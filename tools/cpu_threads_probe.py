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

# the C++ oracle port at full size: all visible hardware threads (what bench.py's cpu_baseline used through round 4) against the cgroup's share
from oracle import mpm_cpu
from softmac_amd import scenes
from helpers import O
cfg, env_dt, state, specs, s13 = scenes.s_grip(1 << 20, 128, 8, "float64", 0, seed=1)
port = mpm_cpu.CpuPort(H.oracle_params(cfg, env_dt), specs)
fr = tuple(t.numpy() for t in O.state24_split(state))
pst = np.array([s for s in s13])
hw = port.threads()
for nt in (hw, 64, 32, mpm_cpu.cpu_share(), 8):
    port.set_threads(nt)
    t = time.time()
    out = port.substep(0, *fr, pst)
    g = port.substep_grad(0, *fr, np.ones((1 << 20, 3)), np.zeros((1 << 20, 3)), np.zeros((1 << 20, 3, 3)), np.zeros((1 << 20, 3, 3)), pst=pst)
    print(f"C++ port, {nt} OpenMP threads (share {mpm_cpu.cpu_share()}, visible {hw}): 1 substep fwd + bwd at 1M particles / 128^3: {time.time() - t:.2f} s")
