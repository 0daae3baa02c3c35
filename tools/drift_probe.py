"""Where does S-grip 4M / 256^3 raise the drift error?  Forward substeps in batches of 5, printing repairs / re-sorts; on the error: the handle's counters."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import bench
a = bench.parse_args(["--steps", "20", "--warmup", "5", "--particles", sys.argv[1] if len(sys.argv) > 1 else "4194304", "--grid", sys.argv[2] if len(sys.argv) > 2 else "256",
                      "--sort-interval", sys.argv[3] if len(sys.argv) > 3 else "40"])
import torch; torch.cuda.init()
sim, run, cfg = bench.build_sim(a, 0, 1, frames=130)
f = 0
try:
    while f < 120:
        sim.run_substeps(f, 5); f += 5
        sim.sync()
        x = sim.get_state(f)
        v = np.abs(x[:, 3:6]).max()
        print(f, "resorts", sim.get_param("resorts"), "repairs", sim.get_param("drift_repairs"), "vmax", round(float(v), 3), "cells/substep", round(float(v) * cfg.dt * a.grid, 4), flush=True)
except Exception as e:
    print("FAILED at batch starting", f, str(e)[:120])
    for k in ("drift_repairs", "resorts", "hit_overflows"):
        try:
            print(k, sim.get_param(k))
        except Exception as e2:
            print(k, "?", e2)
