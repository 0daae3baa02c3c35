"""Which allocation decides the per-process fast/slow mode of the gather kernels?  Prints device addresses next to
the mean k_g2p / k_g2p_grad times of a short S-grip run."""
import ctypes as C
import sys, pathlib
sys.path.insert(0, str(pathlib.Path(__file__).resolve().parent.parent))
import numpy as np
import bench

class A: pass
a = A(); a.particles = 1 << 20; a.grid = 128; a.precision = "float32"; a.steps = 16; a.warmup = 16; a.sort_interval = 0
a.recompute_backward = False; a.workload = "s-grip"
sim, run, cfg = bench.build_sim(a, 0, 1)
run.run_substeps(0, 16)
sim.clear_grads(); sim.add_grad(16, gx=np.zeros((a.particles, 3)))
run.run_substeps_grad(0, 16)
sim.profile(True)
run.run_substeps(16, 16)
run.run_substeps_grad(16, 16)
prof = sim.profile_report()
out = {}
for name in ("grid_in", "grid_out", "grid_out.grad", "state", "state.grad", "slab"):
    p = C.c_void_p(); n = C.c_int64(); b = C.c_int32()
    sim._h.call("smac_grid_device_ptr", name.encode(), C.byref(p), C.byref(n), C.byref(b))
    out[name] = hex(p.value or 0)
print({k: round(v[0] / max(v[1], 1), 4) for k, v in prof.items() if k in ("g2p", "g2p_grad", "p2g", "p2g_grad")}, out)
