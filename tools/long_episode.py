#!/usr/bin/env python3
"""An episode of the benchmark scene (S-grip, 1M particles, 128^3): N substeps forward, a loss seed on the last frame (+ one per env step), the whole
backward sweep - wall clock per phase with everything the episode needs (re-sorts, checkpoint, adjoint re-ordering at epoch boundaries), with the
fused backward step (default) and with SMAC_FUSED_PG=0, and the difference of the two final gradients.   python tools/long_episode.py [substeps]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import helpers as H  # noqa: E402
from softmac_amd import scenes  # noqa: E402

n_sub = int(sys.argv[1]) if len(sys.argv) > 1 else 200
N, grid, env = 1 << 20, 128, 50


def episode(fused):
    os.environ["SMAC_FUSED_PG"] = "1" if fused else "0"
    cfg, env_dt, state, specs, s13 = scenes.s_grip(N, grid, max_steps=n_sub + 4, precision="float32")
    pst = [[np.concatenate([s[:3] + s[7:10] * cfg.dt * f, s[3:]]) for s in s13] for f in range(n_sub + 4)]
    sim, prm = H.build_engine(cfg, env_dt, specs, pst)
    sim.reset(state)
    out = {}
    for rep in range(2):                                  # second pass: warm
        sim.reset(state)
        sim.sync()
        t0 = time.perf_counter()
        sim.run_substeps(0, n_sub)
        sim.sync()
        t1 = time.perf_counter()
        sim.clear_grads()
        rng = np.random.default_rng(7)
        for f in range(n_sub, 0, -env):
            sim.add_grad(f, gx=rng.standard_normal((N, 3)), gv=rng.standard_normal((N, 3)))
        sim.sync()
        t2 = time.perf_counter()
        sim.profile(True)
        sim.run_substeps_grad(0, n_sub)
        sim.sync()
        t3 = time.perf_counter()
        rep_prof = sim.profile_report()
        sim.profile(False)
        out = dict(fwd_ms=(t1 - t0) * 1e3, bwd_ms=(t3 - t2) * 1e3, prof=rep_prof)
    g = np.hstack([a.reshape(N, -1) for a in sim.get_grad_full(0)])
    x = sim.get_state(n_sub)[:, :3]
    return out, g, x


a, ga, xa = episode(True)
b, gb, xb = episode(False)
c, gc, xc = episode(False)                     # the same path once more: what two handles of ONE path differ by (float atomics of drifted lanes)
for name, o in (("fused", a), ("apart", b)):
    p = o["prof"]
    print(f"{name}: {n_sub} substeps forward {o['fwd_ms']:.1f} ms ({o['fwd_ms'] / n_sub * 1e3:.0f} us each), backward {o['bwd_ms']:.1f} ms "
          f"({o['bwd_ms'] / n_sub * 1e3:.0f} us each; with kernel timers on) -> {n_sub / ((o['fwd_ms'] + o['bwd_ms']) * 1e-3):.0f} substeps/s fwd+bwd; "
          f"launches: p2g_g2p_grad {p.get('p2g_g2p_grad', (0, 0))[1]}, p2g_grad {p.get('p2g_grad', (0, 0))[1]}, g2p_grad {p.get('g2p_grad', (0, 0))[1]}, "
          f"sort {p.get('sort', (0, 0))[1]}, reorder_adjoint {p.get('reorder_adjoint', (0, 0))[1]}")
print(f"final x: fused vs apart {np.abs(xa - xb).max():.1e} (absolute);  d loss / d state[0]: {np.abs(ga - gb).max() / np.abs(gb).max():.1e} of its largest entry; "
      f"finite: {bool(np.isfinite(ga).all())}")
print(f"apart vs apart (two handles): final x {np.abs(xc - xb).max():.1e};  d loss / d state[0]: {np.abs(gc - gb).max() / np.abs(gb).max():.1e}")
