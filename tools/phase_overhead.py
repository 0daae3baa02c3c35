import sys, time, pathlib
sys.path.insert(0, '/root/repo')
import numpy as np, torch
import bench
from softmac_amd.parallel import HipSlabEngine, SlabRunner
class A: pass
a = A(); a.particles = 1 << 20; a.grid = 128; a.precision = "float32"; a.steps = 64; a.warmup = 16; a.sort_interval = 0
a.recompute_backward = False; a.workload = "s-grip"
sim, run, cfg = bench.build_sim(a, 0, 1)
eng = HipSlabEngine(sim)
sr = SlabRunner(eng, 0, 1, 32, 96, has_contact=True)
W, K = 16, 64
for name, r in (("batched C loop", run), ("python phases", sr), ("batched C loop", run), ("python phases", sr)):
    r.run_substeps(0, W); sim.clear_grads(); sim.add_grad(W, gx=np.zeros((a.particles, 3))); r.run_substeps_grad(0, W)
    sim.clear_grads(); sim.add_grad(W + K, gx=np.zeros((a.particles, 3)))
    sim.sync(); t0 = time.perf_counter()
    r.run_substeps(W, K); r.run_substeps_grad(W, K)
    sim.sync(); dt = time.perf_counter() - t0
    print(name, round(K / dt, 1), "substeps/s", round(dt / K * 1e6, 1), "us per pair")
