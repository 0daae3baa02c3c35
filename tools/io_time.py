import sys, time
sys.path.insert(0, '/root/repo')
import numpy as np
import bench
class A: pass
a = A(); a.particles = 1 << 20; a.grid = 128; a.precision = "float32"; a.steps = 8; a.warmup = 2; a.sort_interval = 0
a.recompute_backward = False; a.workload = "s-grip"
sim, run, cfg = bench.build_sim(a, 0, 1)
from softmac_amd import scenes
_, _, state, _, _ = scenes.s_grip(1 << 20, 128)
for name, fn in (("reset (N,24) f64", lambda: sim.reset(state)), ("get_x", lambda: sim.get_x(0)), ("get_state (N,24)", lambda: sim.get_state(0)),
                 ("add_grad gx", lambda: sim.add_grad(1, gx=state[:, :3])), ("get_grad (gx, gv)", lambda: sim.get_grad(1))):
    fn(); sim.sync()
    t0 = time.perf_counter(); fn(); sim.sync(); dt = time.perf_counter() - t0
    print(f"{name:22s} {dt*1e3:8.1f} ms")
