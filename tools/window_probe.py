"""Where does the first pass over a range of frames cost more than a repeat of it?  (round 4, session Z)

Times K-step windows (forward K, backward K) of the benchmark scene three ways - the same frames again and again, windows that advance through the
episode, and the advancing windows a second time - and prints the per-kernel HIP-event profile of a first pass and of a repeated pass."""
import argparse, json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench

ap = argparse.ArgumentParser()
ap.add_argument("--steps", type=int, default=20)
ap.add_argument("--warmup", type=int, default=5)
ap.add_argument("--windows", type=int, default=4)
ap.add_argument("--sort-interval", type=int, default=40)
a = ap.parse_args()
args = bench.parse_args(["--steps", str(a.steps), "--warmup", str(a.warmup), "--sort-interval", str(a.sort_interval)])
K, W, R = a.steps, a.warmup, a.windows
sim, run, cfg = bench.build_sim(args, 0, 1, frames=W + (R + 1) * K + 2)
rng = np.random.default_rng(7)
gx = rng.standard_normal((int(cfg.n_particles), 3))

def window(f0, prof=False):
    sim.clear_grads(); sim.add_grad(f0 + K, gx=gx)
    for m in sim.primitives: m.clear_ext_f()
    if prof: sim.profile(True)
    sim.sync(); t0 = time.perf_counter()
    run.run_substeps(f0, K); run.run_substeps_grad(f0, K)
    sim.sync(); t = time.perf_counter() - t0
    rep = None
    if prof:
        rep = {k: (round(v[0] / K * 1e3, 1), v[1]) for k, v in sim.profile_report().items() if v[1] > 0}; sim.profile(False)
    return round(1e3 * t / K, 5), rep

sim.clear_grads(); sim.add_grad(W, gx=gx); run.run_substeps(0, W); run.run_substeps_grad(0, W)
out = {"K": K, "W": W, "interval": a.sort_interval}
out["same_frames"] = [window(W)[0] for _ in range(4)]
out["advancing_first_pass"] = [window(W + (r + 1) * K)[0] for r in range(R)]
out["advancing_second_pass"] = [window(W + (r + 1) * K)[0] for r in range(R)]
print(json.dumps(out))
# profiles: a fresh range vs the same range again (needs one more window of frames: reuse the last one after a reset of nothing - the first pass of a NEW sim)
sim._h.close()
sim, run, cfg = bench.build_sim(args, 0, 1, frames=W + 3 * K + 2)
sim.clear_grads(); sim.add_grad(W, gx=gx); run.run_substeps(0, W); run.run_substeps_grad(0, W)
t1, p1 = window(W, prof=True)
t2, p2 = window(W, prof=True)
t3, p3 = window(W + K, prof=True)
print(json.dumps({"first_pass_us_per_step": p1, "repeat": p2, "next_range_first_pass": p3, "ms": [t1, t2, t3]}))
