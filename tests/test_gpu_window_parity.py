"""Window-length parity at the sizes BASELINE.json names.

* C3, the headline configuration: S-grip, 1,048,576 particles / 128^3, float32 - ONE ENV STEP of the reference's loop (`substeps` = 10 substeps,
  taichi_env.py:101-102, 128-131) forward and backward through the batched entry points (`smac_substeps[_grad]`: the fused kernels are asserted to be
  the ones that ran) against the C++ oracle port (oracle/mpm_cpu.cpp, pinned to the torch oracle by tests/test_cpu_port.py), under the tiered bounds of
  helpers.F32_TOL - and the SIZE of each tier is recorded: how many of the 1,048,576 particles enter the reference's own SVD-adjoint clamp during
  the window, how many share a grid node with one, and the largest error in each tier (gpurun_out/r05_window_parity.json -> profiles/).
* C2 at its real size: S-elastic, 262,144 particles / 64^3, forward only, 8 substeps, both precisions."""
import json
import os
import pathlib
import types

import numpy as np
import pytest
import torch

import helpers as H
from helpers import O
from softmac_amd import scenes

pytestmark = pytest.mark.gpu


def _record(key, payload):
    out = pathlib.Path(H.ROOT) / "gpurun_out"
    try:
        out.mkdir(exist_ok=True)
        path = out / "r05_window_parity.json"
        cur = json.load(open(path)) if path.exists() else {}
        cur[key] = payload
        json.dump(cur, open(path, "w"), indent=1)
    except OSError:
        pass
    print(f"\n[{key}] " + json.dumps(payload))


def test_headline_env_step_fwd_bwd_vs_cpu_port_f32():
    from oracle import mpm_cpu
    N, n_sub = 1 << 20, 10
    cfg, env_dt, state, specs, s13 = scenes.s_grip(N, 128, max_steps=n_sub + 4, precision="float32")
    assert int(round(env_dt / cfg.dt)) == n_sub
    pst = [[np.concatenate([s[:3] + s[7:10] * cfg.dt * f, s[3:]]) for s in s13] for f in range(n_sub + 4)]
    sim, prm = H.build_engine(cfg, env_dt, specs, pst)
    P = H.oracle_params(cfg, env_dt)
    port = mpm_cpu.CpuPort(P, specs)
    # ---- oracle: the env step, frame by frame
    frames = [tuple(t.numpy() for t in O.state24_split(state))]
    ext_ref = np.zeros((len(specs), 6))
    for f in range(n_sub):
        nx, nv, nC, nF, ext = port.substep(f, *frames[-1], np.array(pst[f]))
        frames.append((nx, nv, nC, nF))
        ext_ref += ext
    rng = np.random.default_rng(17)
    seed = [rng.standard_normal((N, 3)), rng.standard_normal((N, 3)), 0.01 * rng.standard_normal((N, 3, 3)), 0.01 * rng.standard_normal((N, 3, 3))]
    g = list(seed)
    pg_ref = []
    for f in range(n_sub - 1, -1, -1):
        r = port.substep_grad(f, *frames[f], *g, pst=np.array(pst[f]))
        g = list(r[:4])
        pg_ref.insert(0, r[4])
    # ---- HIP: the same env step through the batched entry points
    sim.reset(state)
    sim.profile(True)
    sim.run_substeps(0, n_sub)
    sim.clear_grads()
    sim.add_grad(n_sub, gx=seed[0], gv=seed[1], gC=seed[2], gF=seed[3])
    sim.run_substeps_grad(0, n_sub)
    counts = sim.profile_report()
    sim.profile(False)
    assert counts.get("g2p_p2g", (0, 0))[1] == n_sub - 1 and counts.get("p2g_g2p_grad", (0, 0))[1] == n_sub - 1, counts   # the fused kernels were the ones compared
    st = sim.get_state(n_sub)
    x, v, C, F = frames[n_sub]
    errs = dict(x=H.rel_err(st[:, 0:3], x), v=H.rel_err(st[:, 3:6], v), F=H.rel_err(st[:, 6:15], F.reshape(N, 9)), C=H.rel_err(st[:, 15:24], C.reshape(N, 9)))
    ts, tg = H.F32_TOL["state"], H.F32_TOL["grad"]
    assert errs["x"] < ts and errs["v"] < ts and errs["F"] < ts, errs
    assert errs["C"] < H.c_tol(ts, cfg.n_grid, v, C), errs
    e_ext = H.rel_err(np.array([m.ext_f.to_numpy() for m in prm]), ext_ref)
    assert e_ext < 50 * ts and np.abs(ext_ref[1:]).max() > 0, e_ext
    # ---- adjoint at frame 0, by tier.  The clamp margin follows the F difference just measured (helpers.clamp_zone)
    dF = float(np.abs(st[:, 6:15] - F.reshape(N, 9)).max())
    orc = types.SimpleNamespace(frames=[tuple(torch.as_tensor(a) for a in fr) for fr in frames])
    zone, near = H.clamp_zone(orc, P, n_sub, margin=4.0 * dF + 1e-7, neighbours=True)
    gx, gv, gF, gC = sim.get_grad_full(0)
    tiers = {}
    for name, got, ref in (("gx", gx, g[0]), ("gv", gv, g[1]), ("gC", gC, g[2]), ("gF", gF, g[3])):
        tiers[name] = H.rel_err_tiers(np.asarray(got).reshape(N, -1), np.asarray(ref).reshape(N, -1), zone, near)
    pg = np.array([[m.get_all_states_grad(f) for m in prm] for f in range(n_sub)])
    e_pg = float(np.abs(pg - np.array(pg_ref)).max() / max(np.abs(np.array(pg_ref)).max(), 1e-30))
    _record("c3_env_step_f32", dict(
        particles=N, substeps=n_sub, state_errors=errs, ext_f_error=e_ext, clamp_margin=4.0 * dF + 1e-7,
        tier_sizes=dict(clamp=int(zone.sum()), near_clamp=int(near.sum()), rest=int(N - zone.sum() - near.sum())),
        tier_fraction=dict(clamp=float(zone.mean()), near_clamp=float(near.mean())),
        adjoint_errors_rest_near_clamp={k: [float(e) for e in v3] for k, v3 in tiers.items()},
        bounds=dict(rest=tg, near_clamp=H.F32_TOL["near_clamp"], clamp=H.F32_TOL["clamp"]), primitive_state_grad_error=e_pg,
        launches={k: int(c[1]) for k, c in counts.items() if c[1] > 0}))
    for name, (out, nr, zn) in tiers.items():
        assert out < tg, (name, tiers)
        assert nr < H.F32_TOL["near_clamp"] and zn < H.F32_TOL["clamp"], (name, tiers)
    assert e_pg < 10 * tg, e_pg


@pytest.mark.parametrize("precision,tol", [("float64", 1e-9), ("float32", H.F32_TOL["state"])])
def test_c2_real_size_forward_vs_cpu_port(precision, tol):
    """BASELINE config C2 as written: 262,144 particles, 64^3, elastic, no contact, forward only - 8 substeps through the batched loop."""
    from oracle import mpm_cpu
    N, n_sub = 1 << 18, 8
    cfg, env_dt, state, specs, s13 = scenes.s_elastic(N, 64, max_steps=n_sub + 4, precision=precision)
    sim, _ = H.build_engine(cfg, env_dt)
    port = mpm_cpu.CpuPort(H.oracle_params(cfg, env_dt), [])
    fr = tuple(t.numpy() for t in O.state24_split(state))
    for f in range(n_sub):
        fr = port.substep(f, *fr)[:4]
    sim.reset(state)
    sim.profile(True)
    sim.run_substeps(0, n_sub)
    counts = sim.profile_report()
    sim.profile(False)
    assert counts.get("g2p_p2g", (0, 0))[1] == n_sub - 1, counts
    st = sim.get_state(n_sub)
    x, v, C, F = fr
    errs = dict(x=H.rel_err(st[:, 0:3], x), v=H.rel_err(st[:, 3:6], v), F=H.rel_err(st[:, 6:15], F.reshape(N, 9)), C=H.rel_err(st[:, 15:24], C.reshape(N, 9)))
    _record(f"c2_forward_{precision}", dict(particles=N, n_grid=64, substeps=n_sub, state_errors=errs))
    assert errs["x"] < tol and errs["v"] < tol and errs["F"] < tol, errs
    assert errs["C"] < (tol if precision == "float64" else H.c_tol(tol, cfg.n_grid, v, C)), errs
