"""BASELINE config C5 as written - "mixed soft-rigid-cloth coupling": ONE handle with rigid SDF primitives AND the triangle-mesh sheet, a cloud of
two materials, forecast contact.  The reference has no simulator with both kinds of primitive; the composition (primitives in index order, then the
sheet; a two-entry table for the reference's per-particle mu / lam / yield fields) is this build's and is stated in oracle/mixed_oracle.py, which
composes the two line-by-line restatements.  Small scene against that oracle in f64 and f32; the 16M-particle / 256^3 scene through size-independent
properties."""
import types

import numpy as np
import pytest
import torch

import helpers as H
import scenes_cloth as S
from helpers import O
from oracle import mixed_oracle as MO

pytestmark = pytest.mark.gpu


def _scene(precision, N=2500, seed=4):
    sc = S.build("hit", precision, n_env_steps=1, N=N, seed=seed)
    c = sc["cfg"]
    c.ptype, c.n_controllers, c.E, c.yield_stress, c.gravity = 0, 0, 2000.0, 30.0, (0.0, -3.0, 0.0)      # von-Mises plasticine (soft_cloth :231-232)
    sc["control_idx"], sc["action"] = None, None
    sc["mat_id"] = (sc["state"][:, 0] > 0.5).astype(np.int32)             # two blocks side by side
    sc["mat2"] = dict(E=800.0, nu=0.3, yield_stress=12.0)
    palm = H.load_palm()
    sc["rigid_spec"] = dict(palm, friction=0.6, softness=666.0, contact=True)
    # the palm (a box, half extents 0.30 x 0.15 x 0.075) under the block, its top face 2 mm below the lowest particles, rising and turning slowly
    s13 = np.concatenate([[0.5, 0.248, 0.53], [1.0, 0.0, 0.0, 0.0], [0.0, 0.3, 0.05], [0.2, 0.0, 0.1]])
    sc["rigid_states"] = []
    for f in range(sc["nframes"] + 2):
        st = s13.copy()
        st[:3] += f * c.dt * s13[7:10]
        sc["rigid_states"].append(st)
    return sc


def _engine(sc):
    from softmac_amd.config import CfgNode
    from softmac_amd.engine.primitive import Mesh, Primitives
    from softmac_amd.soft_cloth.engine.mpm_simulator import MPMSimulator
    from softmac_amd.soft_cloth.engine.primitive import Primitive_Cloth
    pc = CfgNode()
    for k, v in sc["prim"].items():
        setattr(pc, k, v)
    pc.mpm_force_scale = 1.0
    sheet = Primitive_Cloth(pc, max_timesteps=sc["cfg"].max_steps, mpm_scale=1.0, vertices=sc["vertices"], faces=sc["faces"])
    rc = CfgNode()
    rc.friction, rc.enable_external_force, rc.urdf_path = sc["rigid_spec"]["friction"], True, ""
    mesh = Mesh(sdf=sc["rigid_spec"], cfg=rc, max_timesteps=sc["cfg"].max_steps)
    rigid = Primitives(primitives=[mesh])
    sim = MPMSimulator(sc["cfg"], sheet, sc["env_dt"], 1.0, rigid_primitives=rigid)
    sheet.initialize()
    rigid.initialize()
    mesh.friction[None] = sc["rigid_spec"]["friction"]
    sim.primitives_contact = [True]
    for f, st in enumerate(sc["rigid_states"]):
        mesh.set_all_states(f, st)
    sim.set_materials(sc["mat_id"], sc["mat2"]["E"], sc["mat2"]["nu"], sc["mat2"]["yield_stress"])
    return sim, sheet, mesh


def _oracle_params(sc):
    import dataclasses
    P = S.oracle_params(sc)
    return P, dataclasses.replace(P, E=sc["mat2"]["E"], nu=sc["mat2"]["nu"], yield_stress=sc["mat2"]["yield_stress"])


def _rigid(sc, f):
    st, s = sc["rigid_states"][f], sc["rigid_spec"]
    return O.make_prim(st[:3], st[3:7], st[7:10], st[10:13], s["sdf"], s["normal"], s["lower"], s["upper"], s["dx"], s["friction"], s["softness"], True)


@pytest.mark.parametrize("precision", ["float64", "float32"])
def test_mixed_rigid_and_sheet_substep_matches_the_composed_oracle(precision):
    n = 3
    sc = _scene(precision)
    sim, sheet, mesh = _engine(sc)
    N, V = len(sc["state"]), len(sc["vertices"])
    P, P2 = _oracle_params(sc)
    assert P.ptype == 0 and P.scale == 1.0
    tol_s, tol_g = (1e-9, 1e-8) if precision == "float64" else (H.F32_TOL["state"], H.F32_TOL["grad"])
    cloth = [sc["motion"](f * sc["cfg"].dt) for f in range(n + 1)]
    for f in range(n + 1):
        sheet.set_all_states(f, *cloth[f])
    sim.reset(sc["state"])
    sim.get_contact_pair(0)
    ids0, _ = sim.get_contact(0)
    rng = np.random.default_rng(5)
    ids, pens = [], []
    for f in range(n):
        pen = ((rng.uniform(size=N) < 0.15) & (ids0 >= 0)).astype(np.int8)
        sim.set_contact(f, ids0, pen)
        ids.append(ids0.copy()); pens.append(pen)
    # the scene does hold particles in reach of the palm only, of the sheet only, and of both
    xs = torch.as_tensor(sc["state"][:, :3])
    band = (O.prim_sdf(_rigid(sc, 0), xs) <= 5e-3).numpy()
    assert band.sum() > 20 and (ids0 >= 0).sum() > 100 and (band & (ids0 >= 0)).sum() >= 3, (band.sum(), (ids0 >= 0).sum(), (band & (ids0 >= 0)).sum())
    # oracle rollout
    x, v, C, F = O.state24_split(sc["state"])
    frames, ext_r, ext_c = [(x, v, C, F)], [], []
    for f in range(n):
        x, v, C, F, er, ec = MO.substep(*frames[-1], P, [_rigid(sc, f)], S.oracle_prim(sc, *cloth[f]), ids[f], pens[f], f, P2, sc["mat_id"])
        frames.append((x.detach(), v.detach(), C.detach(), F.detach()))
        ext_r.append(er[0].detach().numpy()); ext_c.append(ec.detach().numpy())
    for f in range(n):
        sim.substep(f)
    walked = sim.contact_counts()[0]
    assert 0 < walked <= int((band | (ids0 >= 0)).sum()) + N // 50     # ONE entry per particle, whichever primitives reach it (the band moves with the palm)
    st = sim.get_state(n)
    x, v, C, F = (t.numpy() for t in frames[n])
    errs = dict(x=H.rel_err(st[:, 0:3], x), v=H.rel_err(st[:, 3:6], v), F=H.rel_err(st[:, 6:15], F.reshape(N, 9)), C=H.rel_err(st[:, 15:24], C.reshape(N, 9)))
    er_ref, ec_ref = np.sum(ext_r, axis=0), np.sum(ext_c, axis=0)
    assert np.abs(er_ref).max() > 0 and np.abs(ec_ref).max() > 0
    e_r, e_c = H.rel_err(mesh.ext_f.to_numpy(), er_ref), H.rel_err(sheet.ext_f.to_numpy(), ec_ref)
    print(f"\n[mixed {precision}] " + " ".join(f"{k} {e:.1e}" for k, e in errs.items()) + f" ext_f rigid {e_r:.1e} sheet {e_c:.1e} ({walked} particles in contact)")
    assert errs["x"] < tol_s and errs["v"] < tol_s and errs["F"] < tol_s
    assert errs["C"] < (tol_s if precision == "float64" else H.c_tol(tol_s, P.n_grid, v, C))
    assert e_r < (1e-8 if precision == "float64" else tol_s) and e_c < (1e-8 if precision == "float64" else tol_s)
    # a second material that changes nothing would not be a test: the two blocks do differ
    Fone = MO.substep(*frames[0], P, [_rigid(sc, 0)], S.oracle_prim(sc, *cloth[0]), ids[0], pens[0], 0, None, None)[3]
    assert (Fone - frames[1][3]).abs().max() > 1e-6
    # adjoint: seeds on the last frame, on the palm's wrench and on the sheet's force
    gx, gv = rng.standard_normal((N, 3)), rng.standard_normal((N, 3))
    gC, gF = 0.01 * rng.standard_normal((N, 3, 3)), 0.01 * rng.standard_normal((N, 3, 3))
    eg_c = rng.standard_normal((V, 3)) * 1e-2 / P.p_mass * P.dt
    eg_r = rng.standard_normal(6) * 1e-2 / P.p_mass * P.dt
    adj = (torch.as_tensor(gx), torch.as_tensor(gv), torch.as_tensor(gC), torch.as_tensor(gF))
    ref_p, ref_cp, ref_cv = [], [], []
    for f in range(n - 1, -1, -1):
        g = MO.substep_grad(*frames[f], P, [_rigid(sc, f)], S.oracle_prim(sc, *cloth[f]), ids[f], pens[f], f, *adj, ext_r_grad=[eg_r], ext_c_grad=eg_c, P2=P2,
                            mat_id=sc["mat_id"])
        adj = (g["gx"], g["gv"], g["gC"], g["gF"])
        ref_p.insert(0, g["prims"][0].numpy()); ref_cp.insert(0, g["sheet_pos"].numpy()); ref_cv.insert(0, g["sheet_vel"].numpy())
    sim.clear_grads()
    sim.add_grad(n, gx=gx, gv=gv, gC=gC, gF=gF)
    for f in range(n - 1, -1, -1):
        sim.substep_grad(f, None, ext_f_grad=eg_c, rigid_ext_f_grad=[eg_r])
    dgx, dgv, dgF, dgC = sim.get_grad_full(0)
    zone = H.clamp_zone(types.SimpleNamespace(frames=frames), P, n) if precision == "float32" else np.zeros(N, dtype=bool)
    gerr = {}
    for name, got, ref in (("gx", dgx, adj[0]), ("gv", dgv, adj[1]), ("gC", dgC, adj[2]), ("gF", dgF, adj[3])):
        out, ins = H.rel_err_split(got.reshape(N, -1), ref.numpy().reshape(N, -1), zone)
        gerr[name] = out
        assert out < tol_g and ins < H.F32_TOL["clamp"], (name, out, ins)
    scale_r = max(np.abs(r).max() for r in ref_p)
    scale_p, scale_v = max(np.abs(r).max() for r in ref_cp), max(np.abs(r).max() for r in ref_cv)
    worst = [0.0, 0.0, 0.0]
    for f in range(n):
        cp, cv = sheet.get_all_states_grad(f)
        gr = mesh.get_all_states_grad(f)
        worst = [max(worst[0], np.abs(gr - ref_p[f]).max() / scale_r), max(worst[1], np.abs(cp - ref_cp[f]).max() / scale_p), max(worst[2], np.abs(cv - ref_cv[f]).max() / scale_v)]
    print(f"[mixed {precision}] " + " ".join(f"{k} {e:.1e}" for k, e in gerr.items()) + f" rigid state.grad {worst[0]:.1e} sheet position.grad {worst[1]:.1e} velocity.grad {worst[2]:.1e}")
    assert scale_r > 0 and worst[0] < tol_g          # (measured 2.5e-15 in f64, 2.7e-7 in f32: profiles/r04_d_mixed_errors.txt)
    assert worst[1] < tol_g and worst[2] < tol_g


def test_c5_mixed_16m_particles_256_grid_rigid_plus_sheet_two_materials():
    """BASELINE config C5 as written, on ONE GPU: 16,777,216 particles / 256^3, two material blocks, one rigid SDF primitive (the reference's cached
    gripper palm) pressed into the cylinder from above, the sticky sheet of 13,824 faces under it.  The oracle cannot follow this size; size-independent
    properties: every particle in reach of a primitive is ONE entry of the contact list, both primitives receive a finite non-zero wrench / force, the
    materials differ where the selector says so (the yield stress caps the deviatoric log strain of each block at ITS ratio), the adjoint through two
    substeps is linear in its seeds for the particles, the palm's state and the sheet's vertices."""
    from softmac_amd import scenes
    from softmac_amd.config import CfgNode
    from softmac_amd.engine.primitive import Mesh, Primitives
    from softmac_amd.soft_cloth.engine.mpm_simulator import MPMSimulator
    from softmac_amd.soft_cloth.engine.primitive import Primitive_Cloth
    N = 1 << 24
    cfg, env_dt, state, V, F, sheet_cfg, rigid, s13, mat_id, mat2 = scenes.s_mixed(N, 256, max_steps=4, precision="float32", rings=48, palm=H.load_palm())
    assert len(F) == 13824 and 0.3 < mat_id.mean() < 0.7
    sheet = Primitive_Cloth(CfgNode(sheet_cfg), max_timesteps=cfg.max_steps, mpm_scale=1.0, vertices=V, faces=F)
    rc = CfgNode()
    rc.friction, rc.enable_external_force, rc.urdf_path = rigid["friction"], True, ""
    mesh = Mesh(sdf=rigid, cfg=rc, max_timesteps=cfg.max_steps)
    sim = MPMSimulator(cfg, sheet, env_dt, 1.0, rigid_primitives=Primitives(primitives=[mesh]))
    sheet.initialize()
    mesh.softness[None] = 666.0
    mesh.friction[None] = rigid["friction"]
    sim.primitives_contact = [True]
    Vv = np.zeros_like(V)
    for f in range(cfg.max_steps):
        sheet.set_all_states(f, V, Vv, f_end=f + 1)
        st = s13.copy()
        st[:3] += f * cfg.dt * s13[7:10]
        mesh.set_all_states(f, st)
    sim.set_materials(mat_id, mat2["E"], mat2["nu"], mat2["yield_stress"])
    sim.reset(state)
    x0 = state[:, :3].copy()
    del state
    sim.get_contact_pair(0)
    ids0, _ = sim.get_contact(0)
    for s in range(2):
        sim.substep(s)
        sim.get_contact_pair(s + 1)
        sim.trace_penetration_after_mpm(s + 1)
    walked = sim.contact_counts()[0]
    top = x0[:, 1].max()
    near_palm = int((x0[:, 1] > top - 8e-3).sum())                       # a generous bound of the palm's 5 mm band (it moves 0.04 mm per substep)
    held = int((ids0 >= 0).sum())
    print(f"\n[C5 mixed] {walked} particles on the contact list ({held} hold a face of the sheet at frame 0, <= {near_palm} within 8 mm of the palm)")
    assert walked > N // 400 and walked <= held + near_palm + N // 100
    er, ec = mesh.ext_f.to_numpy(), sheet.ext_f.to_numpy()
    assert np.isfinite(er).all() and np.abs(er[:3]).max() > 0 and er[1] > 0          # the block pushes the descending palm UP
    assert np.isfinite(ec).all() and np.abs(ec).max() > 0
    # the two blocks yield at their own ratio: |dev log sigma| of a yielded particle sits ON its block's yield surface (soft_cloth :181-186)
    Fm = sim.get_state(2)[:, 6:15].reshape(N, 3, 3)
    idx = np.random.default_rng(0).choice(N, 200000, replace=False)
    sv = np.linalg.svd(Fm[idx], compute_uv=False)
    eps = np.log(np.maximum(sv, 0.05))
    dev = np.sqrt(((eps - eps.mean(1, keepdims=True)) ** 2).sum(1) + 1e-8)
    mu1 = cfg.E / (2 * (1 + cfg.nu)); mu2 = mat2["E"] / (2 * (1 + mat2["nu"]))
    r1, r2 = cfg.yield_stress / (2 * mu1), mat2["yield_stress"] / (2 * mu2)
    m = mat_id[idx]
    assert dev[m == 0].max() < r1 * (1 + 1e-3) + 2e-4 and dev[m == 1].max() < r2 * (1 + 1e-3) + 2e-4 and abs(r1 - r2) > 1e-3, (dev[m == 0].max(), r1, dev[m == 1].max(), r2)
    del Fm
    rng = np.random.default_rng(9)
    s1 = rng.standard_normal((N, 3))

    def grad(seed, k):
        sim.clear_grads()
        sim.add_grad(2, gx=k * seed)
        sim.substep_grad(1)
        sim.substep_grad(0)
        return sim.get_grad(0)[0], sheet.get_all_states_grad(1)[0], mesh.get_all_states_grad(1)
    a, b = grad(s1, 1.0), grad(s1, -2.0)
    for ga, gb in zip(a, b):
        assert np.abs(gb + 2.0 * ga).max() < 2e-4 * max(np.abs(ga).max(), 1e-30)
    assert np.abs(a[1]).max() > 0 and np.abs(a[2]).max() > 0
