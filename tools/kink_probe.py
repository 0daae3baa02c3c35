"""Where does the f32 adjoint of a 10-substep window at 1M particles leave the 1e-5 bar?  (round 5, after tests/test_gpu_window_parity.py's first run:
state 3e-7, adjoint 2e-2 on a handful of particles OUTSIDE the SVD-adjoint clamp tiers.)

Runs S-grip 1M / 128^3, one env step forward + backward: the C++ oracle port (f64), the HIP path in float64 and in float32.  For every particle: the
distance of its singular values to the plastic clip bounds [1 - 2e-3, 1 + 3e-3] (mpm_simulator.py:226-229) - the kink of the reference's own function:
d clip(s) / d s jumps from 1 to 0 there - minimised over the window's frames, the distance to the SVD-adjoint clamp, and whether it sits in a contact
band.  Prints the error of each class and the worst particles.   python tools/kink_probe.py [--substeps 10]"""
import argparse
import json
import pathlib
import sys

import numpy as np

ROOT = pathlib.Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
import helpers as H  # noqa: E402
from helpers import O  # noqa: E402
from oracle import mpm_cpu  # noqa: E402
from softmac_amd import scenes  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--substeps", type=int, default=10)
ap.add_argument("--particles", type=int, default=1 << 20)
ap.add_argument("--grid", type=int, default=128)
ap.add_argument("--out", default=None)
ap.add_argument("--precisions", default="float64,float32")
a = ap.parse_args()
N, n_sub = a.particles, a.substeps
cfg, env_dt, state, specs, s13 = scenes.s_grip(N, a.grid, max_steps=n_sub + 4, precision="float32")
pst = [[np.concatenate([s[:3] + s[7:10] * cfg.dt * f, s[3:]]) for s in s13] for f in range(n_sub + 4)]
P = H.oracle_params(cfg, env_dt)
port = mpm_cpu.CpuPort(P, specs)
frames = [tuple(t.numpy() for t in O.state24_split(state))]
for f in range(n_sub):
    frames.append(port.substep(f, *frames[-1], np.array(pst[f]))[:4])
rng = np.random.default_rng(17)
seed = [rng.standard_normal((N, 3)), rng.standard_normal((N, 3)), 0.01 * rng.standard_normal((N, 3, 3)), 0.01 * rng.standard_normal((N, 3, 3))]
g = list(seed)
for f in range(n_sub - 1, -1, -1):
    g = list(port.substep_grad(f, *frames[f], *g, pst=np.array(pst[f]))[:4])
ref = dict(gx=g[0], gv=g[1], gC=g[2].reshape(N, 9), gF=g[3].reshape(N, 9))

# ---- classes from the oracle's frames
LO, HI = 1.0 - 2e-3, 1.0 + 3e-3
d_clip = np.full(N, np.inf)          # min over frames and singular values of |s - bound|
d_clamp = np.full(N, np.inf)         # min over frames of the smallest |s_i^2 - s_j^2|
for f in range(n_sub):
    x, v, C, F = frames[f]
    Ft = (np.eye(3)[None] + cfg.dt * C) @ F
    s = np.linalg.svd(Ft, compute_uv=False)
    d_clip = np.minimum(d_clip, np.minimum(np.abs(s - LO), np.abs(s - HI)).min(axis=1))
    s2 = s ** 2
    d_clamp = np.minimum(d_clamp, np.minimum(np.abs(s2[:, 0] - s2[:, 1]), np.minimum(np.abs(s2[:, 1] - s2[:, 2]), np.abs(s2[:, 0] - s2[:, 2]))))
x0 = frames[0][0]

out = {"particles": N, "substeps": n_sub}
for precision in a.precisions.split(","):
    c2, e2, st2, sp2, s2_ = scenes.s_grip(N, a.grid, max_steps=n_sub + 4, precision=precision)
    sim, prm = H.build_engine(c2, e2, sp2, pst)
    sim.reset(state)
    sim.run_substeps(0, n_sub)
    sim.clear_grads()
    sim.add_grad(n_sub, gx=seed[0], gv=seed[1], gC=seed[2], gF=seed[3])
    sim.run_substeps_grad(0, n_sub)
    st = sim.get_state(n_sub)
    gx, gv, gF, gC = sim.get_grad_full(0)
    got = dict(gx=gx, gv=gv, gC=np.asarray(gC).reshape(N, 9), gF=np.asarray(gF).reshape(N, 9))
    per = np.zeros(N)
    for k in ref:
        per = np.maximum(per, np.abs(got[k] - ref[k]).max(axis=1) / np.abs(ref[k]).max())
    hits = sim.contact_counts()[0]
    rec = {"state_err_x": H.rel_err(st[:, :3], frames[n_sub][0]), "contact_particles_last_substep": int(hits), "max_err": float(per.max())}
    for name, lim in (("1e-5", 1e-5), ("1e-4", 1e-4), ("1e-3", 1e-3), ("1e-2", 1e-2)):
        rec[f"particles_over_{name}"] = int((per > lim).sum())
    # error by distance to the clip kink
    bins = [0, 1e-9, 1e-8, 1e-7, 1e-6, 1e-5, 1e-4, np.inf]
    rec["by_distance_to_clip_bound"] = []
    for lo, hi in zip(bins[:-1], bins[1:]):
        m = (d_clip >= lo) & (d_clip < hi)
        rec["by_distance_to_clip_bound"].append({"range": [lo, hi if np.isfinite(hi) else None], "particles": int(m.sum()), "max_err": float(per[m].max()) if m.any() else 0.0,
                                                 "over_1e-5": int((per[m] > 1e-5).sum())})
    rec["by_distance_to_svd_clamp"] = []
    cb = [0, 1e-6, 1e-5, 1e-4, 3e-4, 1e-3, 3e-3, 1e-2, np.inf]
    for lo, hi in zip(cb[:-1], cb[1:]):
        m = (d_clamp >= lo) & (d_clamp < hi)
        rec["by_distance_to_svd_clamp"].append({"range": [lo, hi if np.isfinite(hi) else None], "particles": int(m.sum()), "max_err": float(per[m].max()) if m.any() else 0.0,
                                                "median_err": float(np.median(per[m])) if m.any() else 0.0, "over_1e-5": int((per[m] > 1e-5).sum())})
    worst = np.argsort(-per)[:25]
    rec["worst"] = [{"p": int(p), "err": float(per[p]), "d_clip": float(d_clip[p]), "d_clamp": float(d_clamp[p]), "x0": [round(float(c), 4) for c in x0[p]]} for p in worst]
    out[precision] = rec
    print(f"\n== HIP {precision} vs the f64 port, {n_sub} substeps fwd + bwd: max adjoint error {per.max():.2e}; particles over 1e-5: {(per > 1e-5).sum()}, over 1e-3: {(per > 1e-3).sum()}")
    for r in rec["by_distance_to_clip_bound"]:
        print(f"   distance to a clip bound in {r['range']}: {r['particles']:8d} particles, max error {r['max_err']:.2e}, over 1e-5: {r['over_1e-5']}")
    for r in rec["by_distance_to_svd_clamp"]:
        print(f"   smallest |s_i^2 - s_j^2| over the window in {r['range']}: {r['particles']:8d} particles, max error {r['max_err']:.2e}, median {r['median_err']:.2e}, over 1e-5: {r['over_1e-5']}")
    for w in rec["worst"][:12]:
        print("   ", w)
    sim._h.close()
if a.out:
    json.dump(out, open(a.out, "w"), indent=1)
