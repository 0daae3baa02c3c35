#!/usr/bin/env python3
"""What a windowed episode (softmac_amd/engine/windowed.py: checkpoint-every-K state frames with recompute) costs against the fully resident one, on the
benchmark scene (S-grip, 1M particles, 128^3, float32): T substeps forward + a seed on the last frame + the whole backward sweep.
    python tools/windowed_cost.py [T] [K]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import helpers as H  # noqa: E402
from softmac_amd import scenes  # noqa: E402
from softmac_amd.engine.windowed import WindowedEpisode  # noqa: E402

T = int(sys.argv[1]) if len(sys.argv) > 1 else 200
K = int(sys.argv[2]) if len(sys.argv) > 2 else 50
N, grid = 1 << 20, 128
rng = np.random.default_rng(7)
seed = dict(gx=rng.standard_normal((N, 3)), gv=rng.standard_normal((N, 3)))


def scene(max_steps):
    cfg, env_dt, state, specs, s13 = scenes.s_grip(N, grid, max_steps=max_steps, precision="float32")
    pst = lambda f: [np.concatenate([s[:3] + s[7:10] * cfg.dt * f, s[3:]]) for s in s13]
    return cfg, env_dt, state, specs, pst


cfg, env_dt, state, specs, pst = scene(T + 2)
sim, prm = H.build_engine(cfg, env_dt, specs, [pst(f) for f in range(T + 2)])
for rep in range(2):
    sim.reset(state)
    sim.sync()
    t0 = time.perf_counter()
    sim.run_substeps(0, T)
    sim.clear_grads()
    sim.add_grad(T, **seed)
    sim.run_substeps_grad(0, T)
    sim.sync()
    resident = time.perf_counter() - t0
g_res = np.hstack([a.reshape(N, -1) for a in sim.get_grad_full(0)])
del sim, prm

nwin = (T + K - 1) // K
cfg, env_dt, state, specs, pst = scene(K + 1 + nwin + 1)
sim, prm = H.build_engine(cfg, env_dt, specs, None)
ep = WindowedEpisode(sim, K, prim_state=pst)
for rep in range(2):
    ep.reset(state)
    sim.sync()
    t0 = time.perf_counter()
    ep.forward(T)
    sim.sync()
    t1 = time.perf_counter()
    g, _ = ep.backward({T: seed})
    sim.sync()
    t2 = time.perf_counter()
g_win = np.hstack([a.reshape(N, -1) for a in g])
d = np.abs(g_win - g_res).max(axis=1) / np.abs(g_res).max()
frame_mb = 24 * 4 * N / 1e6
print(f"S-grip 1M particles, T = {T} substeps forward + backward (seed upload and the final get_grad included in both)")
print(f"  resident : {T + 2} state frames = {frame_mb * (T + 2) / 1e3:.1f} GB   {resident * 1e3:8.1f} ms   {T / resident:7.0f} substeps/s")
print(f"  windowed : {K + 1 + nwin + 1} state frames = {frame_mb * (K + 1 + nwin + 1) / 1e3:.1f} GB   {(t2 - t0) * 1e3:8.1f} ms   {T / (t2 - t0):7.0f} substeps/s   "
      f"(K = {K}: forward {1e3 * (t1 - t0):.1f} ms, backward incl. the recompute {1e3 * (t2 - t1):.1f} ms)   x {(t2 - t0) / resident:.2f}")
print(f"  final adjoint, windowed vs resident: 99th percentile {np.quantile(d, 0.99):.1e}, max {d.max():.1e} (relative to the field's maximum)")
