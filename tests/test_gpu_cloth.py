"""Soft <-> cloth contact on the GPU against oracle/cloth_oracle.py: contact-face search and penetration tracing
(integer results: equal, except where two faces tie to rounding), the substep with the sheet's forecast contact in f64 and f32
(helpers.F32_TOL), its adjoint including the sheet's position / velocity adjoints and the action, and the env loop."""
import numpy as np
import pytest
import torch

import helpers as H
import scenes_cloth as S
from oracle import cloth_oracle as CO

pytestmark = pytest.mark.gpu


def _pairs_match(x, verts, faces, ids_dev, ids_ref):
    """equal ids, or - where they differ - both faces at the same distance to 1e-12 (closest point on a shared edge / vertex)"""
    bad = np.nonzero(ids_dev != ids_ref)[0]
    if len(bad) == 0:
        return 0
    assert ((ids_dev[bad] >= 0) & (ids_ref[bad] >= 0)).all()
    P = torch.as_tensor(x[bad])
    Vt = torch.as_tensor(verts)
    d = []
    for ids in (ids_dev[bad], ids_ref[bad]):
        f = faces[ids].astype(np.int64)
        d.append(CO.distance_function(P, Vt[f[:, 0]], Vt[f[:, 1]], Vt[f[:, 2]]).numpy())
    assert np.abs(d[0] - d[1]).max() < 1e-12 * max(1.0, np.abs(verts).max())
    return len(bad)


@pytest.mark.parametrize("kind", ["taco", "hit"])
@pytest.mark.parametrize("precision", ["float64", "float32"])
def test_contact_pairs_and_tracing_match_oracle(kind, precision):
    sc = S.build(kind, precision, n_env_steps=1)
    sim, prim = S.build_engine(sc)
    N = len(sc["state"])
    x0, v0 = sc["motion"](0.0)
    prim.set_all_states(0, x0, v0, f_end=sc["nframes"] + 2)
    sim.reset(sc["state"])
    sim.get_contact_pair(0)
    ids_dev, pen_dev = sim.get_contact(0)
    xs = sim.get_x(0)                                   # what the device sees (f32: positions on its 2^-32 lattice)
    ids_ref = CO.get_contact_pair(xs, x0, sc["faces"], None, sc["scale"])
    ties = _pairs_match(xs, x0, sc["faces"], ids_dev, ids_ref)
    assert (ids_dev >= 0).sum() > N // 10 and ties < N // 20
    assert (pen_dev == 0).all()
    # three substeps with the sheet moving: trace after each; flags must equal the oracle's on the device's own trajectory
    nb, nbd = prim.neighbor_faces_np, prim.neighbor_faces_direction_np
    onb, onbd = CO.process_faces(sc["faces"], 200)
    assert (nb == onb).all() and (nbd == onbd).all()
    pen_prev, ids_prev, x_prev, cl_prev = pen_dev.copy(), ids_dev.copy(), xs, x0
    flips = 0
    sweep = -1.0 if kind == "taco" else 1.0               # (towards the particles: the tortilla rises under the disc)
    for f in range(1, 4):
        xc, vc = sc["motion"](sweep * f * sc["cfg"].dt * 40)   # exaggerated sheet motion: vertices sweep through the particles
        prim.set_all_states(f, xc, vc)
        sim.substep(f - 1, sc["action"])
        sim.get_contact_pair(f)
        sim.trace_penetration_after_mpm(f)
        ids_c, pen_c = sim.get_contact(f)
        x_c = sim.get_x(f)
        ids_o = CO.get_contact_pair(x_c, xc, sc["faces"], pen_prev, sc["scale"])
        _pairs_match(x_c, xc, sc["faces"], ids_c, ids_o)
        pen_o, _ = CO.trace_penetration_after_mpm(x_c, x_prev, xc, cl_prev, sc["faces"], ids_c, ids_prev, pen_prev, nb, nbd)
        assert (pen_o == pen_c).all()
        assert sim.check_penetration(f) == int((pen_c == 1).sum())
        flips += int((pen_c != pen_prev).sum())
        pen_prev, ids_prev, x_prev, cl_prev = pen_c, ids_c, x_c, xc
    assert flips > 0                                    # the scenario does exercise the flag
    # after-cloth tracing: move the sheet under frozen particles
    sim.backup_contact_pair(3)
    xn, vn = sc["motion"](0.0)
    prim.set_all_states(3, xn, vn)
    sim.get_contact_pair(3)
    sim.trace_penetration_after_cloth(3)
    ids_n, pen_n = sim.get_contact(3)
    # (cloth frame f-1 = 2 still holds the old sheet: that is the reference's comparison, :538-540)
    pen_o, _ = CO.trace_penetration_after_cloth(x_prev, xn, sc["motion"](sweep * 2 * sc["cfg"].dt * 40)[0], sc["faces"], ids_n, ids_prev, pen_prev, nb, nbd)
    assert (pen_o == pen_n).all()


def _rollout_oracle(sc, P, frames_cloth, ids, pens, n):
    x, v, C, F = CO.O.state24_split(sc["state"])
    frames, exts = [(x, v, C, F)], []
    ci = None if sc["control_idx"] is None else torch.as_tensor(sc["control_idx"], dtype=torch.int64)
    act = None if sc["action"] is None else torch.as_tensor(sc["action"], dtype=CO.DT)
    for f in range(n):
        pr = S.oracle_prim(sc, *frames_cloth[f])
        x, v, C, F, ext = CO.substep(*frames[-1], P, pr, ids[f], pens[f], f, ci, act)
        frames.append((x.detach(), v.detach(), C.detach(), F.detach()))
        exts.append(ext.detach().numpy())
    return frames, exts


@pytest.mark.parametrize("kind,ctype", [("taco", 2), ("hit", 2), ("taco", 1), ("hit", 1)])
@pytest.mark.parametrize("precision", ["float64", "float32"])
def test_substep_with_cloth_contact_matches_oracle(kind, ctype, precision):
    n = 3
    sc = S.build(kind, precision, n_env_steps=1, collision_type=ctype)
    sim, prim = S.build_engine(sc)
    N, V = len(sc["state"]), len(sc["vertices"])
    P = S.oracle_params(sc)
    tol_s, tol_g = (1e-9, 1e-8) if precision == "float64" else (H.F32_TOL["state"], H.F32_TOL["grad"])
    cloth = [sc["motion"](f * sc["cfg"].dt) for f in range(n + 1)]
    for f in range(n + 1):
        prim.set_all_states(f, *cloth[f])
    sim.reset(sc["state"])
    # contact faces / flags: searched on the device once, then fixed for both sides (some particles flagged as penetrated so that
    # the push-out branch of collide_mixed :271-272 runs)
    sim.get_contact_pair(0)
    ids0, _ = sim.get_contact(0)
    rng = np.random.default_rng(5)
    ids, pens = [], []
    for f in range(n):
        pen = ((rng.uniform(size=N) < 0.15) & (ids0 >= 0)).astype(np.int8)
        sim.set_contact(f, ids0, pen)
        ids.append(ids0.copy()); pens.append(pen)
    frames, exts = _rollout_oracle(sc, P, cloth, ids, pens, n)
    for f in range(n):
        sim.substep(f, sc["action"])
    nhit = sim.contact_counts()[0]                       # particles the contact kernels walked: those that can be inside the 5e-3 band
    assert int(pens[n - 1].sum()) < nhit <= int((ids0 >= 0).sum())
    st = sim.get_state(n)
    x, v, C, F = (t.numpy() for t in frames[n])
    assert H.rel_err(st[:, 0:3], x) < tol_s and H.rel_err(st[:, 3:6], v) < tol_s
    assert H.rel_err(st[:, 6:15], F.reshape(N, 9)) < tol_s
    assert H.rel_err(st[:, 15:24], C.reshape(N, 9)) < (tol_s if precision == "float64" else H.c_tol(tol_s, P.n_grid / P.scale, v, C))
    ext_ref = np.sum(exts, axis=0)
    assert np.abs(ext_ref).max() > 0
    print(f"\n[cloth {kind} ctype {ctype} {precision}] x {H.rel_err(st[:, 0:3], x):.1e} v {H.rel_err(st[:, 3:6], v):.1e} F {H.rel_err(st[:, 6:15], F.reshape(N, 9)):.1e} "
          f"C {H.rel_err(st[:, 15:24], C.reshape(N, 9)):.1e} ext_f {H.rel_err(prim.ext_f.to_numpy(), ext_ref):.1e}")
    assert H.rel_err(prim.ext_f.to_numpy(), ext_ref) < (1e-8 if precision == "float64" else tol_s)
    # adjoint: seeds on the last frame + on the sheet's force
    gx, gv = rng.standard_normal((N, 3)), rng.standard_normal((N, 3))
    gC, gF = 0.01 * rng.standard_normal((N, 3, 3)), 0.01 * rng.standard_normal((N, 3, 3))
    eg = rng.standard_normal((V, 3)) * 1e-2 / P.p_mass * P.dt
    adj = (torch.as_tensor(gx), torch.as_tensor(gv), torch.as_tensor(gC), torch.as_tensor(gF))
    ci = None if sc["control_idx"] is None else torch.as_tensor(sc["control_idx"], dtype=torch.int64)
    act = None if sc["action"] is None else torch.as_tensor(sc["action"], dtype=CO.DT)
    ref_cp, ref_cv, ref_act = [], [], []
    for f in range(n - 1, -1, -1):
        g = CO.substep_grad(*frames[f], P, S.oracle_prim(sc, *cloth[f]), ids[f], pens[f], f, *adj, ext_f_grad=eg, control_idx=ci, action=act)
        adj = (g["gx"], g["gv"], g["gC"], g["gF"])
        ref_cp.insert(0, g["cloth_pos"].numpy()); ref_cv.insert(0, g["cloth_vel"].numpy()); ref_act.insert(0, None if g["action"] is None else g["action"].numpy())
    sim.clear_grads()
    sim.add_grad(n, gx=gx, gv=gv, gC=gC, gF=gF)
    got_act = []
    for f in range(n - 1, -1, -1):
        got_act.insert(0, sim.substep_grad(f, sc["action"], ext_f_grad=eg))
    dgx, dgv, dgF, dgC = sim.get_grad_full(0)
    zone = H.clamp_zone(types_frames(frames), P, n) if precision == "float32" else np.zeros(N, dtype=bool)
    for got, ref in ((dgx, adj[0]), (dgv, adj[1]), (dgC.reshape(N, 9), adj[2].reshape(N, 9)), (dgF.reshape(N, 9), adj[3].reshape(N, 9))):
        out, ins = H.rel_err_split(got.reshape(N, -1), ref.numpy().reshape(N, -1), zone)
        assert out < tol_g and ins < H.F32_TOL["clamp"]
    errs = [H.rel_err_split(g.reshape(N, -1), r.numpy().reshape(N, -1), zone) for g, r in ((dgx, adj[0]), (dgv, adj[1]), (dgC, adj[2]), (dgF, adj[3]))]
    print(f"\n[cloth {kind} ctype {ctype} {precision}] gx {errs[0][0]:.1e} gv {errs[1][0]:.1e} gC {errs[2][0]:.1e} gF {errs[3][0]:.1e} clamp-zone {max(e[1] for e in errs):.1e} ({int(zone.sum())} particles)")
    for f in range(n):
        cp, cv = prim.get_all_states_grad(f)
        scale_p, scale_v = max(np.abs(r).max() for r in ref_cp), max(np.abs(r).max() for r in ref_cv)
        print(f"[cloth {kind} {precision}] frame {f}: sheet position.grad {np.abs(cp - ref_cp[f]).max() / scale_p:.1e} velocity.grad {np.abs(cv - ref_cv[f]).max() / scale_v:.1e}"
              + ("" if sc["action"] is None else f" action.grad {H.rel_err(got_act[f], ref_act[f]):.1e}"))
        # measured in f32 (profiles/r02_v_cloth_f32_errors.txt): forecast contact 2e-8 ... 7e-7; penalty contact (collision_type 1) up to 1.3e-6 on the
        # positions and 1.0e-5 on velocity.grad, whose only source is the friction term -p_v_t / |p_v_t| |n.v| kf (primitive_cloth.py:220-222): a
        # quotient of float32-stored particle velocities, 100x smaller than every other adjoint of the scene
        assert np.abs(cp - ref_cp[f]).max() < tol_g * scale_p
        assert np.abs(cv - ref_cv[f]).max() < (2 if (ctype == 1 and precision == "float32") else 1) * tol_g * scale_v
        if sc["action"] is not None:
            assert H.rel_err(got_act[f], ref_act[f]) < tol_g


def types_frames(frames):
    import types
    return types.SimpleNamespace(frames=frames)


def test_env_loop_with_kinematic_sheet_matches_oracle():
    """two env steps of the reference's loop (taichi_env.py:86-106): substeps with search + tracing after each, sheet update, backup,
    search, after-cloth tracing - final state, flags and summed sheet force against the oracle run on the same sequence"""
    from softmac_amd.config import CfgNode
    from softmac_amd.soft_cloth.engine.taichi_env import TaichiEnv
    sc = S.build("taco", "float64", n_env_steps=2, N=1200)
    cfg = CfgNode()
    cfg.env_dt, cfg.mpm_scale, cfg.control_mode = sc["env_dt"], sc["scale"], "mpm"
    cfg.SIMULATOR = CfgNode({k: v for k, v in vars(sc["cfg"]).items()})
    cfg.PRIMITIVES = CfgNode(dict(sc["prim"], mpm_force_scale=1.0))
    env_t = {"t": 0.0}

    def motion(idx, x, v, action, ext_f):
        return sc["motion"](idx * sc["env_dt"])
    env = TaichiEnv(cfg, sc["state"][:, :3], vertices=sc["vertices"], faces=sc["faces"], motion=motion)
    env.initialize()
    sim, prim = env.simulator, env.primitive
    N = env.n_particles
    sub = env.substeps
    # oracle mirror of the loop
    P = S.oracle_params(sc)
    nb, nbd = CO.process_faces(sc["faces"], 200)
    x = torch.as_tensor(sc["state"][:, :3]); v = torch.zeros(N, 3, dtype=CO.DT)
    C = torch.zeros(N, 3, 3, dtype=CO.DT); F = torch.eye(3, dtype=CO.DT).repeat(N, 1, 1)
    sheet = {f: (sc["vertices"], np.zeros_like(sc["vertices"])) for f in range(sub + 1)}
    ids = {0: CO.get_contact_pair(x, sheet[0][0], sc["faces"], None, sc["scale"])}
    pen = {0: np.zeros(N, dtype=np.int8)}
    xs = {0: x.numpy()}
    ext_sum = []
    for step in range(2):
        ext = np.zeros_like(sc["vertices"])
        for s in range(step * sub, (step + 1) * sub):
            x, v, C, F, e = CO.substep(x, v, C, F, P, S.oracle_prim(sc, *sheet[s]), ids[s], pen[s], s)
            x, v, C, F = x.detach(), v.detach(), C.detach(), F.detach()
            ext += e.numpy()
            xs[s + 1] = x.numpy()
            ids[s + 1] = CO.get_contact_pair(x, sheet[s + 1][0], sc["faces"], pen[s], sc["scale"])
            pen[s + 1], _ = CO.trace_penetration_after_mpm(xs[s + 1], xs[s], sheet[s + 1][0], sheet[s][0], sc["faces"], ids[s + 1], ids[s], pen[s], nb, nbd)
        ext_sum.append(ext / sub)
        cur = (step + 1) * sub
        new = sc["motion"]((step + 1) * sc["env_dt"])
        old_prev = sheet[cur - 1][0]
        for j in range(cur, cur + sub + 1):
            sheet[j] = new
        before = ids[cur]
        ids[cur] = CO.get_contact_pair(x, new[0], sc["faces"], pen[cur - 1], sc["scale"])
        pen[cur], _ = CO.trace_penetration_after_cloth(xs[cur], new[0], old_prev, sc["faces"], ids[cur], before, pen[cur], nb, nbd)
        env.step(None)
    st = sim.get_state(2 * sub)
    assert H.rel_err(st[:, 0:3], x.numpy()) < 1e-9 and H.rel_err(st[:, 3:6], v.numpy()) < 1e-8
    assert (st[:, 25] == pen[2 * sub]).all()
    differ = st[:, 24].astype(np.int64) != ids[2 * sub]
    assert differ.sum() <= N // 50                      # ties between faces sharing an edge
    for a, b in zip(env.cloth_simulator.ext_f_log, ext_sum):
        assert H.rel_err(a, b) < 1e-8


def test_cloth_path_at_full_size():
    """1M particles / 128^3 on the sheet (scenes.s_taco; the oracle cannot follow at this size): size-independent properties.
    The chunk-culled contact-face search equals the flat one bit for bit; the adjoint is linear in its seeds; the force collected on
    the sheet balances the momentum the contact took from the particles' grid."""
    import ctypes as C
    from softmac_amd import scenes
    from softmac_amd.config import CfgNode
    from softmac_amd.soft_cloth.engine.mpm_simulator import MPMSimulator
    from softmac_amd.soft_cloth.engine.primitive import Primitive_Cloth
    N = 1 << 20
    cfg, env_dt, scale, state, V, F, prim_cfg = scenes.s_taco(N, 128, max_steps=8, precision="float32")
    prim = Primitive_Cloth(CfgNode(prim_cfg), max_timesteps=cfg.max_steps, mpm_scale=scale, vertices=V, faces=F)
    sim = MPMSimulator(cfg, prim, env_dt, scale)
    prim.initialize()
    Vv = np.zeros_like(V); Vv[:, 1] = 0.2
    prim.set_all_states(0, V, Vv, f_end=cfg.max_steps)
    sim.reset(state)
    sim.get_contact_pair(0)
    for s in range(3):
        sim.substep(s)
        sim.get_contact_pair(s + 1)
        sim.trace_penetration_after_mpm(s + 1)
    ids_chunk, pen = sim.get_contact(3)
    sim._h.call("smac_set_param", b"cloth_pairs_flat", C.c_double(1.0))
    sim.get_contact_pair(3)
    ids_flat, _ = sim.get_contact(3)
    sim._h.call("smac_set_param", b"cloth_pairs_flat", C.c_double(0.0))
    sim._h.call("smac_set_param", b"cloth_hash", C.c_double(0.0))       # round 2's per-chunk search over the whole mesh (no broad phase)
    sim.get_contact_pair(3)
    ids_nohash, _ = sim.get_contact(3)
    sim._h.call("smac_set_param", b"cloth_hash", C.c_double(1.0))
    assert (ids_chunk == ids_flat).all() and (ids_nohash == ids_flat).all() and (ids_chunk >= 0).sum() > N // 100
    assert sim.get_param("cloth_hash_entries") > len(F)                 # the broad phase was built (every face sits in several blocks' lists)
    assert sim.check_penetration(3) == int((pen == 1).sum()) and sim.tracing_warnings == 0
    ext = prim.ext_f.to_numpy()
    assert np.isfinite(ext).all() and np.abs(ext).max() > 0
    assert ext[:, 1].sum() < 0                           # the disc falls onto a sheet that moves up: the sheet is pushed down
    rng = np.random.default_rng(8)
    s1, s2 = rng.standard_normal((N, 3)), rng.standard_normal((N, 3))

    def grad(seed_x, seed_v):
        sim.clear_grads()
        sim.add_grad(3, gx=seed_x, gv=seed_v)
        for s in range(2, -1, -1):
            sim.substep_grad(s)
        gx, gv = sim.get_grad(0)
        cp, cv = prim.get_all_states_grad(1)
        return gx, gv, cp, cv
    a = grad(s1, None)
    b = grad(None, s2)
    c = grad(2.0 * s1, -0.5 * s2)
    for ga, gb, gc in zip(a, b, c):
        ref = 2.0 * ga - 0.5 * gb
        assert np.abs(gc - ref).max() < 2e-4 * max(np.abs(ref).max(), 1e-30)
    assert np.abs(a[2]).max() > 0 and np.abs(b[3]).max() > 0        # the sheet does receive adjoints


def test_c5_slice_16m_particles_256_grid_on_a_fine_sheet():
    """BASELINE config C5 on ONE GPU: 16,777,216 particles on a 256^3 grid resting on a sheet of 13,824 faces (1.6 GB per
    frame).  The oracle cannot follow; size-independent properties: the broad phase (per-block face lists) gives the contact faces of the search
    over the whole mesh bit for bit, also after substeps have moved the particles out of their binning cells and after the sheet has moved; the
    adjoint is linear in its seeds; no tracing warnings.  Prints what the broad phase buys at this size."""
    import ctypes as C
    import time
    from softmac_amd import scenes
    from softmac_amd.config import CfgNode
    from softmac_amd.soft_cloth.engine.mpm_simulator import MPMSimulator
    from softmac_amd.soft_cloth.engine.primitive import Primitive_Cloth
    N = 1 << 24
    cfg, env_dt, scale, state, V, F, prim_cfg = scenes.s_taco(N, 256, max_steps=4, precision="float32", rings=48)
    assert len(F) == 13824
    prim = Primitive_Cloth(CfgNode(prim_cfg), max_timesteps=cfg.max_steps, mpm_scale=scale, vertices=V, faces=F)
    sim = MPMSimulator(cfg, prim, env_dt, scale)
    prim.initialize()
    Vv = np.zeros_like(V); Vv[:, 1] = 0.2
    for f in range(cfg.max_steps):
        prim.set_all_states(f, V + f * cfg.dt * Vv, Vv, f_end=f + 1)           # the sheet moves up: a new broad phase per frame
    sim.reset(state)
    del state
    sim.get_contact_pair(0)
    for s in range(2):
        sim.substep(s)
        sim.get_contact_pair(s + 1)
        sim.trace_penetration_after_mpm(s + 1)

    def search(hash_on, reps=3):
        sim._h.call("smac_set_param", b"cloth_hash", C.c_double(1.0 if hash_on else 0.0))
        sim.get_contact_pair(2); sim.sync()
        t0 = time.perf_counter()
        for _ in range(reps):
            sim.get_contact_pair(2)
        sim.sync()
        return sim.get_contact(2)[0].copy(), (time.perf_counter() - t0) / reps

    ids_hash, t_hash = search(True)
    ids_full, t_full = search(False)
    sim._h.call("smac_set_param", b"cloth_hash", C.c_double(1.0))
    print(f"\n[C5 slice] contact-face search at 16M particles x 13,824 faces: {1e3 * t_hash:.2f} ms with the broad phase, {1e3 * t_full:.2f} ms per-chunk over "
          f"the whole mesh; {int((ids_hash >= 0).sum())} particles hold a face; {int(sim.get_param('cloth_hash_entries'))} (face, block) pairs")
    assert (ids_hash == ids_full).all() and (ids_hash >= 0).sum() > N // 200
    sim.check_penetration(2)
    assert sim.tracing_warnings == 0
    ext = prim.ext_f.to_numpy()
    assert np.isfinite(ext).all() and np.abs(ext).max() > 0
    rng = np.random.default_rng(9)
    s1 = rng.standard_normal((N, 3))

    def grad(seed, k):
        sim.clear_grads()
        sim.add_grad(2, gx=k * seed)
        sim.substep_grad(1)
        sim.substep_grad(0)
        return sim.get_grad(0)[0], prim.get_all_states_grad(1)[0]
    a, b = grad(s1, 1.0), grad(s1, -2.0)
    for ga, gb in zip(a, b):
        assert np.abs(gb + 2.0 * ga).max() < 2e-4 * max(np.abs(ga).max(), 1e-30)
    assert np.abs(a[1]).max() > 0


@pytest.mark.parametrize("name", ["taco", "hit", "hit_penalty"])
@pytest.mark.parametrize("precision", ["float64", "float32"])
def test_hip_path_against_committed_cloth_golden_vectors(name, precision):
    """the committed vectors (tests/golden/oracle_cloth_*.npz, tools/make_golden_cloth.py) need no oracle on the GPU box"""
    import importlib.util
    spec = importlib.util.spec_from_file_location("make_golden_cloth", H.ROOT / "tools" / "make_golden_cloth.py")
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    kind, ctype = mod.CASES[name]
    gold = np.load(H.GOLDEN / f"oracle_cloth_{name}.npz")
    N = len(gold["contact_id"])
    sc = S.build(kind, precision, N=N, seed=21, collision_type=ctype)
    sim, prim = S.build_engine(sc)
    n = mod.N_STEPS
    for f in range(n + 1):
        prim.set_all_states(f, *sc["motion"](f * sc["cfg"].dt))
    sim.reset(sc["state"])
    for f in range(n):
        sim.set_contact(f, gold["contact_id"], gold["penetration"])
        sim.substep(f, sc["action"])
    ts, tg = (1e-9, 1e-8) if precision == "float64" else (H.F32_TOL["state"], H.F32_TOL["grad"])
    st = sim.get_state(n)
    P = S.oracle_params(sc)
    assert H.rel_err(st[:, 0:3], gold["x"]) < ts and H.rel_err(st[:, 3:6], gold["v"]) < ts and H.rel_err(st[:, 6:15], gold["F"].reshape(N, 9)) < ts
    assert H.rel_err(st[:, 15:24], gold["C"].reshape(N, 9)) < (ts if precision == "float64" else H.c_tol(ts, P.n_grid / P.scale, gold["v"], gold["C"]))
    assert H.rel_err(prim.ext_f.to_numpy(), gold["ext_f"]) < (1e-8 if precision == "float64" else ts)
    sim.clear_grads()
    sim.add_grad(n, gx=gold["seed_gx"], gv=gold["seed_gv"], gC=gold["seed_gC"], gF=gold["seed_gF"])
    acts = []
    for f in range(n - 1, -1, -1):
        acts.insert(0, sim.substep_grad(f, sc["action"], ext_f_grad=gold["ext_f_grad"]))
    gx, gv, gF, gC = sim.get_grad_full(0)
    for got, key in ((gx, "gx0"), (gv, "gv0"), (gC, "gC0"), (gF, "gF0")):
        assert H.rel_err(got.reshape(N, -1), gold[key].reshape(N, -1)) < tg, key
    slack = 2 if (ctype == 1 and precision == "float32") else 1            # (see test_substep_with_cloth_contact_matches_oracle)
    for f in range(n):
        cp, cv = prim.get_all_states_grad(f)
        assert np.abs(cp - gold["cloth_pos_grad"][f]).max() < tg * np.abs(gold["cloth_pos_grad"]).max()
        assert np.abs(cv - gold["cloth_vel_grad"][f]).max() < slack * tg * np.abs(gold["cloth_vel_grad"]).max()
        if sc["action"] is not None:
            assert H.rel_err(acts[f], gold["action_grad"][f]) < tg


def test_cloth_api_rejects_misuse():
    from softmac_amd._ffi import SmacError
    sc = S.build("hit", "float64", N=200)
    sim, prim = S.build_engine(sc)
    sim.reset(sc["state"])
    with pytest.raises(SmacError, match="already has a cloth"):
        prim._bind(sim._h)
    with pytest.raises(SmacError, match="frame range"):
        prim.set_all_states(sc["cfg"].max_steps, sc["vertices"], sc["vertices"])
    with pytest.raises(SmacError, match="face id out of range"):
        sim.set_contact(0, np.full(200, len(sc["faces"]), dtype=np.int32), None)
    with pytest.raises(SmacError, match="needs frame f-1"):
        sim.trace_penetration_after_mpm(0)
    with pytest.raises(SmacError, match="unknown parameter"):
        sim._h.call("smac_set_param", b"no_such_knob", __import__("ctypes").c_double(1.0))
    with pytest.raises(ValueError):
        prim.set_all_states(0, sc["vertices"][:-1], sc["vertices"][:-1])


@pytest.mark.parametrize("case", list(range(12)))
def test_cloth_random_configurations(case):
    """seeded corners the two demo scenes do not visit: every material, both contact models and stickiness settings, odd particle counts,
    other length scales and friction / softness values, both precisions, re-sorting every substep or never - two substeps each"""
    rng = np.random.default_rng(500 + case)
    kind = "taco" if case % 2 == 0 else "hit"
    precision = "float64" if case % 4 < 2 else "float32"
    N = int(rng.choice([1, 63, 257, 900]))
    ctype = int(rng.choice([1, 2, 2]))
    sc = S.build(kind, precision, N=N, seed=600 + case, collision_type=ctype)
    new_scale = float(rng.choice([1.0, 2.5, 5.0]))
    f = new_scale / sc["scale"]
    sc["state"][:, 0:6] *= f
    sc["vertices"] = sc["vertices"] * f
    sc["scale"] = new_scale
    sc["motion"] = S.sheet_motion(kind, sc["vertices"], new_scale)
    sc["cfg"].ptype = int(rng.integers(0, 3))
    sc["cfg"].material_model = 0 if sc["cfg"].ptype == 0 else int(rng.integers(0, 2))
    sc["cfg"].E = float(rng.choice([500.0, 5000.0])) if sc["cfg"].ptype != 2 else 30.0
    sc["cfg"].sort_interval = int(rng.choice([1, 16]))
    sc["prim"].update(sticky=bool(rng.integers(0, 2)), friction=float(rng.choice([0.0, 0.9, 10.0])), softness=float(rng.choice([66.0, 666.0])))
    sim, prim = S.build_engine(sc)
    P = S.oracle_params(sc)
    n = 2
    cloth = [sc["motion"](k * sc["cfg"].dt) for k in range(n + 1)]
    for k in range(n + 1):
        prim.set_all_states(k, *cloth[k])
    sim.reset(sc["state"])
    sim.get_contact_pair(0)
    ids0, _ = sim.get_contact(0)
    pen = ((rng.uniform(size=N) < 0.2) & (ids0 >= 0)).astype(np.int8)
    for k in range(n):
        sim.set_contact(k, ids0, pen)
    frames, exts = _rollout_oracle(sc, P, cloth, [ids0] * n, [pen] * n, n)
    for k in range(n):
        sim.substep(k, sc["action"])
    ts, tg = (1e-9, 1e-8) if precision == "float64" else (H.F32_TOL["state"], H.F32_TOL["grad"])
    st = sim.get_state(n)
    x, v, C, F = (t.numpy() for t in frames[n])
    assert H.rel_err(st[:, 0:3], x) < ts and H.rel_err(st[:, 3:6], v) < ts and H.rel_err(st[:, 6:15], F.reshape(N, 9)) < ts
    assert H.rel_err(st[:, 15:24], C.reshape(N, 9)) < (ts if precision == "float64" else H.c_tol(ts, P.n_grid / P.scale, v, C))
    ext_ref = np.sum(exts, axis=0)
    if np.abs(ext_ref).max() > 0:
        assert H.rel_err(prim.ext_f.to_numpy(), ext_ref) < (1e-8 if precision == "float64" else ts)
    gx, gv = rng.standard_normal((N, 3)), rng.standard_normal((N, 3))
    adj = (torch.as_tensor(gx), torch.as_tensor(gv), torch.zeros(N, 3, 3, dtype=CO.DT), torch.zeros(N, 3, 3, dtype=CO.DT))
    ci = None if sc["control_idx"] is None else torch.as_tensor(sc["control_idx"], dtype=torch.int64)
    act = None if sc["action"] is None else torch.as_tensor(sc["action"], dtype=CO.DT)
    for k in range(n - 1, -1, -1):
        g = CO.substep_grad(*frames[k], P, S.oracle_prim(sc, *cloth[k]), ids0, pen, k, *adj, control_idx=ci, action=act)
        adj = (g["gx"], g["gv"], g["gC"], g["gF"])
    sim.clear_grads()
    sim.add_grad(n, gx=gx, gv=gv)
    for k in range(n - 1, -1, -1):
        sim.substep_grad(k, sc["action"])
    got = sim.get_grad_full(0)
    zone = H.clamp_zone(types_frames(frames), P, n) if precision == "float32" else np.zeros(N, dtype=bool)
    pairs = ((got[0], adj[0]), (got[1], adj[1]), (got[3].reshape(N, 9), adj[2].reshape(N, 9)), (got[2].reshape(N, 9), adj[3].reshape(N, 9)))
    top = max(float(b.abs().max()) for _, b in pairs)
    for a, b in pairs:
        if float(b.abs().max()) < 1e-6 * top:          # an exactly vanishing adjoint (one isolated particle: sum_n w_n (x_n - x_p) = 0 removes C and F from its
            assert np.abs(a).max() < tg * top           # velocity) is compared on the scale of the others, not on its own rounding noise
            continue
        out, ins = H.rel_err_split(a.reshape(N, -1), b.numpy().reshape(N, -1), zone)
        assert out < tg and ins < H.F32_TOL["clamp"]


def test_particles_that_outrun_their_binning_are_recomputed_with_the_sheet_in_contact():
    """Round 4 (VERDICT r3 missing 4): the drift repair used to leave the cloth variant out - its contact-face search and penetration tracing run on the
    host's schedule between the substeps, so a replayed epoch would have met the faces found on the INVALID frames.  The calls made on every frame are now on
    file and are made again on the recomputed frame.  A block thrown at the towel under an enormous acceleration with a 32-substep re-sort interval (it
    out-runs the 4-cell halo of its binning) against the same loop with a re-sort before every substep (which cannot drift): states, contact faces,
    penetration flags and the sheet's force must agree; the second handle is the reference here, and it is itself held to the oracle for three substeps."""
    n = 16
    runs = {}
    for interval in (1, 32):
        sc = S.build("hit", "float64", n_env_steps=2, N=1500)
        sc["cfg"].gravity = (0.0, 0.0, -30000.0)
        sc["cfg"].n_controllers = 0
        sc["control_idx"] = None
        sc["cfg"].sort_interval = interval
        sim, prim = S.build_engine(sc)
        prim.set_all_states(0, sc["vertices"], 0 * sc["vertices"], f_end=sc["cfg"].max_steps)
        sim.reset(sc["state"])
        sim.get_contact_pair(0)
        for s in range(n):
            sim.substep(s)
            sim.get_contact_pair(s + 1)
            sim.trace_penetration_after_mpm(s + 1)
        runs[interval] = dict(st=sim.get_state(n), ext=prim.ext_f.to_numpy().copy(), repairs=sim.get_param("drift_repairs"), mid=sim.get_state(3))
    a, b = runs[1], runs[32]
    assert a["repairs"] == 0 and b["repairs"] >= 1, (a["repairs"], b["repairs"])
    N = len(a["st"])
    assert H.rel_err(b["st"][:, 0:3], a["st"][:, 0:3]) < 1e-9 and H.rel_err(b["st"][:, 3:6], a["st"][:, 3:6]) < 1e-8
    assert H.rel_err(b["st"][:, 6:24], a["st"][:, 6:24]) < 1e-8
    assert (b["st"][:, 25] == a["st"][:, 25]).all() and (b["st"][:, 24] != a["st"][:, 24]).sum() <= N // 50      # (ties between faces sharing an edge)
    assert np.abs(a["ext"]).max() > 0 and H.rel_err(b["ext"], a["ext"]) < 1e-8
    # the never-drifting handle against the oracle over the first substeps (the whole window would take the Python oracle minutes)
    sc = S.build("hit", "float64", n_env_steps=2, N=1500)
    sc["cfg"].gravity = (0.0, 0.0, -30000.0)
    sc["cfg"].n_controllers = 0
    P = S.oracle_params(sc)
    x = torch.as_tensor(sc["state"][:, :3]); v = torch.as_tensor(sc["state"][:, 3:6])
    F = torch.as_tensor(sc["state"][:, 6:15].reshape(N, 3, 3)); C = torch.as_tensor(sc["state"][:, 15:24].reshape(N, 3, 3))
    nb, nbd = CO.process_faces(sc["faces"], 200)
    verts = (sc["vertices"], np.zeros_like(sc["vertices"]))
    ids = CO.get_contact_pair(x, verts[0], sc["faces"], None, sc["scale"])
    pen = np.zeros(N, dtype=np.int8)
    for s in range(3):
        xp = x.numpy().copy()
        x, v, C, F, e = CO.substep(x, v, C, F, P, S.oracle_prim(sc, *verts), ids, pen, s)
        x, v, C, F = x.detach(), v.detach(), C.detach(), F.detach()
        ids_new = CO.get_contact_pair(x, verts[0], sc["faces"], pen, sc["scale"])
        pen, _ = CO.trace_penetration_after_mpm(x.numpy(), xp, verts[0], verts[0], sc["faces"], ids_new, ids, pen, nb, nbd)
        ids = ids_new
    assert H.rel_err(a["mid"][:, 0:3], x.numpy()) < 1e-9 and H.rel_err(a["mid"][:, 3:6], v.numpy()) < 1e-8
