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


GAP_C = 1.5e-9      # float32 bound of a particle whose smallest |s_i^2 - s_j^2| is `gap` (outside the reference's clamp): GAP_C / gap of the field's maximum


def _kink_tiers(frame, P, dt, delta):
    """Particles whose F_tmp has a singular value within `delta` of a bound of the plastic clip [1 - 2e-3, 1 + 3e-3] (mpm_simulator.py:226-229) in this frame,
    and the particles that share a grid node with one.  d clip(s) / d s jumps from 1 to 0 at a bound: the reference's own function has no derivative there,
    and an implementation whose s differs by less than the distance takes the other one-sided value - an O(1) change of that particle's adjoint, of which its
    stencil neighbours receive a share in the SAME substep (through grid_v_in.grad).  `delta` is what float32 arithmetic can differ by from f64 on identical
    inputs: e = s - 1 comes out of the Jacobi SVD of E = F_tmp - I with a few ulp of |e| <= 3e-3, i.e. ~1e-9."""
    x, v, C, F = frame
    N = len(x)
    Ft = (np.eye(3)[None] + dt * C) @ F
    s = np.linalg.svd(Ft, compute_uv=False)
    d = np.minimum(np.abs(s - (1.0 - 2e-3)), np.abs(s - (1.0 + 3e-3))).min(axis=1)
    kink = d < delta
    near = np.zeros(N, dtype=bool)
    if kink.any():
        n = int(P.n_grid)
        base = np.clip((x * P.inv_dx - 0.5).astype(np.int64), 0, n - 1) + 2
        g = np.zeros((n + 4, n + 4, n + 4), dtype=bool)
        g[base[kink, 0], base[kink, 1], base[kink, 2]] = True
        dil = np.zeros_like(g)
        for ox in range(-2, 3):
            for oy in range(-2, 3):
                for oz in range(-2, 3):
                    dil[2:-2, 2:-2, 2:-2] |= g[2 + ox:n + 2 + ox, 2 + oy:n + 2 + oy, 2 + oz:n + 2 + oz]
        near = dil[base[:, 0], base[:, 1], base[:, 2]] & ~kink
    return kink, near, d


def test_headline_env_step_fwd_bwd_vs_cpu_port():
    """S-grip at the headline size, one env step (10 substeps) forward + backward through the batched entry points, against the C++ oracle port.

    float64 mode: every particle, state 1e-9, frame-0 adjoint 1e-8 - the whole window, end to end.

    float32 mode, in three statements (measured first by tools/kink_probe.py and tools/ref_sensitivity.py, profiles/r05_window_parity.md):
      (a) state after the window: every particle within F32_TOL (measured: x 2e-9, v 3e-7, F 7e-8).
      (b) the adjoint of EVERY substep of the window, on identical inputs: for frames f = 9, 4, 0 the device's adjoint frame A[f] against the port's
          substep_grad applied to the DEVICE's own state S[f] and adjoint A[f + 1] - 1e-5 of the field's maximum for every particle outside the tiers,
          F32_TOL's bounds inside the SVD-adjoint clamp tiers, and two things this size brings to light: the plastic clip's kink (`_kink_tiers`) and
          the continuation of the clamp tier to gaps above it - a particle whose smallest |s_i^2 - s_j^2| is `gap` is bounded by GAP_C / gap where that
          exceeds 1e-5 (gaps below 1.5e-4: eps / gap is what float32 keeps of K = 1 / gap times a difference of O(1) terms).  The sizes of all tiers are recorded.  The frames in between are covered by linearity: the backward sweep is the product of these maps.
      (c) end to end against the pure f64 window the same bar can NOT hold in float32 storage, and not because of the adjoint kernels: the device's state
          differs from the f64 rollout by 7e-8 in F (float32 grid velocities -> C -> F), about 1e-4 of the particles sit closer than that to a clip bound in
          some frame, their derivative takes the other one-sided value and their neighbours inherit a share at every further substep.  The f64 port ITSELF
          moves by more when its F is merely stored in float32 (tools/ref_sensitivity.py --store F: 91 % of the particles beyond 1e-5, largest 0.29) and by
          1.7e-6 when it is stored as the device stores it (E = F - I).  Recorded: how many particles lie beyond 1e-5 / 1e-4 / 1e-3; asserted: the
          decomposition - device vs the port's chain along the DEVICE's states (what the adjoint kernels add) stays below the distance between that chain
          and the pure f64 window (what the state difference does to the reference's own derivative)."""
    from oracle import mpm_cpu
    N, n_sub = 1 << 20, 10
    cfg, env_dt, state, specs, s13 = scenes.s_grip(N, 128, max_steps=n_sub + 4, precision="float32")
    assert int(round(env_dt / cfg.dt)) == n_sub
    pst = [[np.concatenate([s[:3] + s[7:10] * cfg.dt * f, s[3:]]) for s in s13] for f in range(n_sub + 4)]
    P = H.oracle_params(cfg, env_dt)
    port = mpm_cpu.CpuPort(P, specs)
    # ---- oracle: the env step, frame by frame, and its adjoint
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
    ref0 = dict(gx=g[0], gv=g[1], gC=g[2].reshape(N, 9), gF=g[3].reshape(N, 9))

    def per_particle(got, ref):
        per = np.zeros(N)
        for k in ref:
            per = np.maximum(per, np.abs(np.asarray(got[k]).reshape(N, -1) - ref[k].reshape(N, -1)).max(axis=1) / np.abs(ref[k]).max())
        return per

    def run(precision):
        c2, e2, _, sp2, _ = scenes.s_grip(N, 128, max_steps=n_sub + 4, precision=precision)
        sim, prm = H.build_engine(c2, e2, sp2, pst)
        sim.reset(state)
        sim.profile(True)
        sim.run_substeps(0, n_sub)
        sim.clear_grads()
        sim.add_grad(n_sub, gx=seed[0], gv=seed[1], gC=seed[2], gF=seed[3])
        sim.run_substeps_grad(0, n_sub)
        counts = sim.profile_report()
        sim.profile(False)
        assert counts.get("g2p_p2g", (0, 0))[1] == n_sub - 1, counts                     # the fused kernels were the ones compared
        if precision == "float32":
            assert counts.get("p2g_g2p_grad", (0, 0))[1] == n_sub - 1, counts
        return sim, prm, counts

    def adj(sim, f):
        gx, gv, gF, gC = sim.get_grad_full(f)
        return dict(gx=np.asarray(gx), gv=np.asarray(gv), gC=np.asarray(gC).reshape(N, 9), gF=np.asarray(gF).reshape(N, 9))

    # ---- float64 mode: end to end, every particle
    sim, prm, _ = run("float64")
    st = sim.get_state(n_sub)
    x, v, C, F = frames[n_sub]
    e64 = dict(x=H.rel_err(st[:, 0:3], x), v=H.rel_err(st[:, 3:6], v), F=H.rel_err(st[:, 6:15], F.reshape(N, 9)), C=H.rel_err(st[:, 15:24], C.reshape(N, 9)))
    g64 = float(per_particle(adj(sim, 0), ref0).max())
    sim._h.close()
    assert max(e64.values()) < 1e-9 and g64 < 1e-8, (e64, g64)

    # ---- float32 mode
    sim, prm, counts = run("float32")
    st = sim.get_state(n_sub)
    errs = dict(x=H.rel_err(st[:, 0:3], x), v=H.rel_err(st[:, 3:6], v), F=H.rel_err(st[:, 6:15], F.reshape(N, 9)), C=H.rel_err(st[:, 15:24], C.reshape(N, 9)))
    ts, tg = H.F32_TOL["state"], H.F32_TOL["grad"]
    assert errs["x"] < ts and errs["v"] < ts and errs["F"] < ts, errs                  # (a)
    assert errs["C"] < H.c_tol(ts, cfg.n_grid, v, C), errs
    e_ext = H.rel_err(np.array([m.ext_f.to_numpy() for m in prm]), ext_ref)
    assert e_ext < 50 * ts and np.abs(ext_ref[1:]).max() > 0, e_ext
    pg = np.array([[m.get_all_states_grad(f) for m in prm] for f in range(n_sub)])
    e_pg = float(np.abs(pg - np.array(pg_ref)).max() / max(np.abs(np.array(pg_ref)).max(), 1e-30))

    def dev_frame(f):
        s = sim.get_state(f)
        return (s[:, 0:3].copy(), s[:, 3:6].copy(), s[:, 15:24].reshape(N, 3, 3).copy(), s[:, 6:15].reshape(N, 3, 3).copy())

    dev = {f: dev_frame(f) for f in range(n_sub)}
    A = {f: adj(sim, f) for f in range(n_sub + 1)}
    A[n_sub] = dict(gx=seed[0], gv=seed[1], gC=seed[2].reshape(N, 9), gF=seed[3].reshape(N, 9))       # (the seed as it was given, f64)
    # (b) every checked substep on identical inputs, with tiers
    per_frame = {}
    chain = A[n_sub]
    chain_frames = {}
    for f in range(n_sub - 1, -1, -1):
        r = port.substep_grad(f, *dev[f], chain["gx"], chain["gv"], chain["gC"].reshape(N, 3, 3), chain["gF"].reshape(N, 3, 3), pst=np.array(pst[f]))
        chain = dict(gx=r[0], gv=r[1], gC=r[2].reshape(N, 9), gF=r[3].reshape(N, 9))                  # the port's chain ALONG the device's states (for (c))
        if f in (n_sub - 1, 4, 0):
            a1 = A[f + 1]
            r1 = port.substep_grad(f, *dev[f], a1["gx"], a1["gv"], a1["gC"].reshape(N, 3, 3), a1["gF"].reshape(N, 3, 3), pst=np.array(pst[f]))
            one = dict(gx=r1[0], gv=r1[1], gC=r1[2].reshape(N, 9), gF=r1[3].reshape(N, 9))
            per = per_particle(A[f], one)
            orc1 = types.SimpleNamespace(frames=[tuple(torch.as_tensor(a) for a in dev[f])])
            zone, near = H.clamp_zone(orc1, P, 1, neighbours=True)             # (the suite's margin: helpers.clamp_zone)
            kink, knear, dk = _kink_tiers(dev[f], P, cfg.dt, 2e-9)
            # outside the clamp the SVD adjoint multiplies a difference of O(1) terms by K = 1 / |s_i^2 - s_j^2| (mpm_simulator.py:140-157): float32 keeps
            # eps / gap of that particle's own term.  Measured at this size (three particles of 1,048,576 in one substep, gaps 1.7e-5 ... 7e-5, errors
            # 1.0e-5 ... 2.1e-5): the bound of an ill-conditioned particle is GAP_C / gap where that exceeds 1e-5, i.e. for gaps below 1.5e-4
            s2 = np.linalg.svd((np.eye(3)[None] + cfg.dt * dev[f][2]) @ dev[f][3], compute_uv=False) ** 2
            gap = np.minimum(np.abs(s2[:, 0] - s2[:, 1]), np.minimum(np.abs(s2[:, 1] - s2[:, 2]), np.abs(s2[:, 0] - s2[:, 2])))
            ill = (gap < GAP_C / tg) & ~(zone | near | kink)
            ill_worst = float((per[ill] * gap[ill]).max() / GAP_C) if ill.any() else 0.0       # worst error in units of its bound GAP_C / gap
            # (a flipped kink changes that particle's OWN adjoint rows; its stencil neighbours feel it one substep EARLIER, through the grid: in the substep itself they
            #  stay inside the bar - measured 6.5e-6 next to a particle that moved by 7e-2 - and are not a tier)
            rest = ~(zone | near | kink | ill)
            pick = lambda m: float(per[m].max()) if m.any() else 0.0
            # who is beyond the bar outside every tier (diagnostics for the record: which field, how close to the clamp / a clip bound, where)
            off = np.nonzero(rest & (per > tg))[0][:10]
            offenders = [dict(p=int(q), err=float(per[q]), per_field={k: float(np.abs(A[f][k][q] - one[k][q]).max() / np.abs(one[k]).max()) for k in one},
                              clamp_gap=float(gap[q]), clip_distance=float(dk[q]),
                              x=[round(float(c), 4) for c in dev[f][0][q]], speed=float(np.abs(dev[f][1][q]).max())) for i, q in enumerate(off)]
            per_frame[f] = dict(rest=pick(rest), offenders=offenders, ill_conditioned=dict(particles=int(ill.sum()), over_1e5=int((per[ill] > tg).sum()), max=pick(ill),
                                                                                           worst_in_units_of_its_bound=ill_worst), clamp=pick(zone), near_clamp=pick(near & ~kink & ~knear), kink=pick(kink), stencil_neighbours_of_kink=pick(knear & ~zone & ~near),
                                sizes=dict(clamp=int(zone.sum()), near_clamp=int(near.sum()), kink=int(kink.sum()), near_kink=int(knear.sum()), rest=int(rest.sum())),
                                over_1e5_in_rest=int((per[rest] > tg).sum()))
    # (c) end to end: what the adjoint kernels add (device vs the chain along its own states) against what the state difference does to the reference's derivative
    per_impl = per_particle(A[0], chain)
    per_sens = per_particle(chain, ref0)
    per_total = per_particle(A[0], ref0)
    count = lambda per: {t: int((per > float(t)).sum()) for t in ("1e-5", "1e-4", "1e-3", "1e-2")}
    _record("c3_env_step", dict(
        particles=N, substeps=n_sub, f64_mode=dict(state_errors=e64, adjoint_error_every_particle=g64),
        f32_state_errors=errs, f32_ext_f_error=e_ext, f32_primitive_state_grad_error=e_pg,
        f32_single_substeps_on_identical_inputs=per_frame,
        f32_end_to_end=dict(vs_pure_f64_window=dict(max=float(per_total.max()), particles_over=count(per_total)),
                            adjoint_kernels_only__device_vs_port_chain_along_device_states=dict(max=float(per_impl.max()), particles_over=count(per_impl)),
                            reference_sensitivity__port_chain_along_device_states_vs_pure_f64=dict(max=float(per_sens.max()), particles_over=count(per_sens))),
        bounds=dict(rest=tg, near_clamp=H.F32_TOL["near_clamp"], clamp=H.F32_TOL["clamp"], kink=0.2,
                    ill_conditioned=f"{GAP_C:g} / gap for gap < {GAP_C / tg:g}"),
        launches={k: int(c[1]) for k, c in counts.items() if c[1] > 0}))
    sim._h.close()
    for f, r in per_frame.items():                                                     # (b)
        assert r["rest"] < tg, (f, per_frame)
        assert r["ill_conditioned"]["worst_in_units_of_its_bound"] < 1.0, (f, per_frame)
        assert r["near_clamp"] < H.F32_TOL["near_clamp"] and r["clamp"] < H.F32_TOL["clamp"], (f, per_frame)
        assert r["kink"] < 0.2 and r["sizes"]["kink"] < 64, (f, per_frame)                 # the kink carve-out: a few particles of a million per substep
    assert e_pg < 10 * tg, e_pg
    # (c) the adjoint kernels' own share of the end-to-end distance is the small one
    assert np.median(per_total) < 1e-6 and per_total.max() < 0.2
    assert int((per_impl > 1e-3).sum()) <= max(int((per_sens > 1e-3).sum()), 20), (count(per_impl), count(per_sens))


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
