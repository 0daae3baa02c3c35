"""Is the fast/slow mode of the gather kernels a property of the process or of the allocation?  Builds several
simulators in ONE process and prints each one's mean k_g2p / k_g2p_grad time."""
import sys, pathlib
sys.path.insert(0, str(pathlib.Path(__file__).resolve().parent.parent))
import numpy as np
import bench

class A: pass
a = A(); a.particles = 1 << 20; a.grid = 128; a.precision = "float32"; a.steps = 16; a.warmup = 16; a.sort_interval = 0
a.recompute_backward = False; a.workload = "s-grip"
keep = []
for trial in range(5):
    sim, run, cfg = bench.build_sim(a, 0, 1)
    keep.append(sim)                                  # keep earlier allocations alive: the next one lands elsewhere
    run.run_substeps(0, 16)
    sim.clear_grads(); sim.add_grad(16, gx=np.zeros((a.particles, 3)))
    run.run_substeps_grad(0, 16)
    sim.profile(True)
    run.run_substeps(16, 16)
    run.run_substeps_grad(16, 16)
    prof = sim.profile_report()
    print(trial, {k: round(v[0] / max(v[1], 1), 4) for k, v in prof.items() if k in ("g2p", "g2p_grad", "p2g", "p2g_grad")}, flush=True)
