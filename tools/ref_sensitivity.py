"""How far does the f64 REFERENCE function move when its state is stored in float32?  (round 5; CPU only)

S-grip at full size, one env step (10 substeps) forward + backward with the C++ oracle port (oracle/mpm_cpu.cpp, f64 arithmetic):
  A: as it is;
  B: the same arithmetic, but after every forward substep v, C, F are rounded to float32 and x to 2^-32 (the device's storage in float32 mode).
The plastic return map clips singular values at [1 - 2e-3, 1 + 3e-3] (mpm_simulator.py:226-229): d clip / d s jumps from 1 to 0 there, so a particle whose
singular value sits within the storage rounding of a bound takes the other branch in B and its adjoint changes by O(1) of its own size - and its grid
neighbours inherit a share at every further substep.  Prints how many particles' frame-0 adjoints differ between A and B by more than 1e-5 / 1e-4 / 1e-3
of the field's maximum: the floor under ANY float32-storage implementation's distance from the f64 reference on this window.
    python tools/ref_sensitivity.py [--particles N] [--grid G] [--substeps K] [--out file.json]"""
import argparse
import json
import pathlib
import sys
import time

import numpy as np

ROOT = pathlib.Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
import helpers as H  # noqa: E402
from helpers import O  # noqa: E402
from oracle import mpm_cpu  # noqa: E402
from softmac_amd import scenes  # noqa: E402


MODE = "E"


def f32_storage(fr):
    x, v, C, F = fr
    if MODE == "F":             # F itself in float32 (6e-8 absolute on entries near 1): what a float32 implementation WITHOUT the device's E = F - I rows would store
        Fs = F.astype(np.float32).astype(np.float64)
    else:                       # the device's rows: E = F - I in float32 (1e-10 absolute)
        Fs = np.eye(3) + (F - np.eye(3)).astype(np.float32).astype(np.float64)
    return (np.round(x * 2.0 ** 32) / 2.0 ** 32, v.astype(np.float32).astype(np.float64), C.astype(np.float32).astype(np.float64), Fs)


def window(port, cfg, state, pst, n_sub, seed, rounded):
    N = len(state)
    frames = [tuple(t.numpy() for t in O.state24_split(state))]
    if rounded:
        frames[0] = f32_storage(frames[0])
    for f in range(n_sub):
        fr = port.substep(f, *frames[-1], np.array(pst[f]))[:4]
        frames.append(f32_storage(fr) if rounded else fr)
    g = list(seed)
    for f in range(n_sub - 1, -1, -1):
        g = list(port.substep_grad(f, *frames[f], *g, pst=np.array(pst[f]))[:4])
    return frames, dict(gx=g[0], gv=g[1], gC=g[2].reshape(N, 9), gF=g[3].reshape(N, 9))


def compare(a, b):
    N = len(a["gx"])
    per = np.zeros(N)
    for k in a:
        per = np.maximum(per, np.abs(a[k] - b[k]).max(axis=1) / np.abs(a[k]).max())
    return per


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--particles", type=int, default=1 << 20)
    ap.add_argument("--grid", type=int, default=128)
    ap.add_argument("--substeps", type=int, default=10)
    ap.add_argument("--out", default=None)
    ap.add_argument("--store", default="E", choices=["E", "F"], help="what the float32 storage holds: E = F - I (the device) or F itself")
    a = ap.parse_args()
    global MODE
    MODE = a.store
    N, n_sub = a.particles, a.substeps
    cfg, env_dt, state, specs, s13 = scenes.s_grip(N, a.grid, max_steps=n_sub + 4, precision="float64")
    pst = [[np.concatenate([s[:3] + s[7:10] * cfg.dt * f, s[3:]]) for s in s13] for f in range(n_sub + 4)]
    port = mpm_cpu.CpuPort(H.oracle_params(cfg, env_dt), specs)
    rng = np.random.default_rng(17)
    seed = [rng.standard_normal((N, 3)), rng.standard_normal((N, 3)), 0.01 * rng.standard_normal((N, 3, 3)), 0.01 * rng.standard_normal((N, 3, 3))]
    t0 = time.time()
    fa, ga = window(port, cfg, state, pst, n_sub, seed, False)
    fb, gb = window(port, cfg, state, pst, n_sub, seed, True)
    per = compare(ga, gb)
    xs = H.rel_err(fb[n_sub][0], fa[n_sub][0])
    Fs = H.rel_err(fb[n_sub][3], fa[n_sub][3])
    out = {"particles": N, "n_grid": a.grid, "substeps": n_sub, "float32_storage_of": a.store, "seconds": time.time() - t0, "state_distance": {"x": xs, "F": Fs}, "max": float(per.max()),
           "particles_over": {t: int((per > float(t)).sum()) for t in ("1e-5", "1e-4", "1e-3", "1e-2")}}
    print(json.dumps(out))
    if a.out:
        json.dump(out, open(a.out, "w"), indent=1)


if __name__ == "__main__":
    main()
