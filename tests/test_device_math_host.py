"""Device arithmetic (softmac_amd/csrc/smac_math.hpp) compiled for the HOST by tests/harness and
compared with the oracle: the hand-derived constitutive adjoint, the Jacobi SVD, the forecast-contact
forward-mode adjoint and the kinematics - in both scalar types."""
import ctypes
import subprocess

import numpy as np
import pytest
import torch

import helpers as H
from helpers import O

HARNESS = H.ROOT / "tests" / "harness"
dp = ctypes.POINTER(ctypes.c_double)
ip = ctypes.POINTER(ctypes.c_int)


def P(a):
    return a.ctypes.data_as(dp)


@pytest.fixture(scope="module")
def lib():
    out = HARNESS / "_build"
    out.mkdir(exist_ok=True)
    so = out / "libmath_harness.so"
    src = HARNESS / "math_harness.cpp"
    hdr = H.ROOT / "softmac_amd" / "csrc" / "smac_math.hpp"
    if not so.exists() or so.stat().st_mtime < max(src.stat().st_mtime, hdr.stat().st_mtime):
        subprocess.check_call(["g++", "-O2", "-shared", "-fPIC", "-o", str(so), str(src)])
    return ctypes.CDLL(str(so))


@pytest.mark.parametrize("prec,tol", [(64, 1e-10), (32, 5e-6)])
@pytest.mark.parametrize("ptype,model", [(0, 0), (1, 0), (2, 0), (0, 1), (1, 1), (2, 1)])
@pytest.mark.parametrize("scale", [1e-3, 5e-2])
def test_constitutive_forward_and_adjoint(lib, prec, tol, ptype, model, scale):
    rng = np.random.default_rng(ptype * 7 + model)
    n = 500
    Pm = O.SimParams(ptype=ptype, material_model=model, E=3e3)
    Et = scale * rng.standard_normal((n, 3, 3)); G = rng.standard_normal((n, 3, 3)); gFn = rng.standard_normal((n, 3, 3))
    Ft = torch.tensor(np.eye(3) + Et, requires_grad=True)
    U, sig, V = O.svd3(Ft) if model == 0 else (None, None, None)
    nF, stress = O.constitutive(Ft, U, sig, V, Pm)
    (gref,) = torch.autograd.grad((stress * torch.tensor(G)).sum() + (nF * torch.tensor(gFn)).sum(), Ft)
    En, st, gEt = np.zeros_like(Et), np.zeros_like(Et), np.zeros_like(Et)
    lib.h_constitutive(prec, n, ptype, model, ctypes.c_double(Pm.mu), ctypes.c_double(Pm.lam), P(Et), P(G), P(gFn), P(En), P(st), P(gEt))
    assert np.abs(En + np.eye(3) - nF.detach().numpy()).max() < tol
    assert H.rel_err(st, stress.detach().numpy()) < tol
    assert H.rel_err(gEt, gref.numpy()) < tol


@pytest.mark.parametrize("prec,tol,scale", [(64, 1e-10, 2e-3), (64, 1e-10, 3e-2), (64, 1e-10, 0.3),      # 0.3: includes sigma < 0.05 (the :175 clamp)
                                            (32, 5e-6, 2e-3), (32, 5e-6, 3e-2), (32, 5e-6, 0.1)])
def test_von_mises_forward_and_adjoint(lib, prec, tol, scale):
    """soft_cloth's return mapping (soft_cloth/engine/mpm_simulator.py:172-188) in the device's SVD-basis form against torch autograd
    through the oracle's restatement: particles inside the yield surface, on it, and far outside"""
    from oracle import cloth_oracle as CO
    rng = np.random.default_rng(11)
    n = 600
    Pm = CO.ClothSimParams(ptype=0, material_model=0, E=5000.0, nu=0.2, yield_stress=60.0)
    Et = scale * rng.standard_normal((n, 3, 3)); G = rng.standard_normal((n, 3, 3)); gFn = rng.standard_normal((n, 3, 3))
    Ft = torch.tensor(np.eye(3) + Et, requires_grad=True)
    U, sig, V = O.svd3(Ft)
    nF, stress = CO.constitutive(Ft, U, sig, V, Pm)
    (gref,) = torch.autograd.grad((stress * torch.tensor(G)).sum() + (nF * torch.tensor(gFn)).sum(), Ft)
    En, st, gEt = np.zeros_like(Et), np.zeros_like(Et), np.zeros_like(Et)
    lib.h_constitutive_von_mises(prec, n, ctypes.c_double(Pm.mu), ctypes.c_double(Pm.lam), ctypes.c_double(Pm.yield_stress / (2 * Pm.mu)),
                                 P(Et), P(G), P(gFn), P(En), P(st), P(gEt))
    yields = (nF.detach() - Ft.detach()).abs().amax((1, 2)) > 0
    assert (scale > 1e-2) == bool(yields.float().mean() > 0.5)
    assert np.abs(En + np.eye(3) - nF.detach().numpy()).max() < tol
    assert H.rel_err(st, stress.detach().numpy()) < tol
    if prec == 32:                                      # the clamp-zone particles of the reference's backward_svd (helpers.F32_TOL) are bounded apart
        s2 = torch.linalg.svdvals(Ft.detach()).numpy() ** 2
        gap = np.minimum(np.abs(s2[:, 0] - s2[:, 1]), np.minimum(np.abs(s2[:, 1] - s2[:, 2]), np.abs(s2[:, 0] - s2[:, 2])))
        out, ins = H.rel_err_split(gEt.reshape(n, -1), gref.numpy().reshape(n, -1), gap < 4e-6)
        assert out < tol and ins < H.F32_TOL["clamp"]
    else:
        assert H.rel_err(gEt, gref.numpy()) < tol


@pytest.mark.parametrize("prec,tol", [(64, 1e-12), (32, 2e-6)])
def test_jacobi_svd(lib, prec, tol):
    rng = np.random.default_rng(0)
    n = 1000
    E = np.concatenate([1e-4 * rng.standard_normal((n // 2, 3, 3)), 0.2 * rng.standard_normal((n // 2, 3, 3))])
    E[0] = 0                                            # F = I exactly (every demo's first substep)
    E[1] = np.diag([1e-3, 1e-3, 1e-3])                  # F = c I (pour fixture)
    U, e, V = np.zeros_like(E), np.zeros((n, 3)), np.zeros_like(E)
    lib.h_svd(prec, n, P(E), P(U), P(e), P(V))
    F = np.eye(3) + E
    rec = np.einsum("nij,nj,nkj->nik", U, 1 + e, V)
    assert np.abs(rec - F).max() < tol
    assert np.abs(np.einsum("nji,njk->nik", U, U) - np.eye(3)).max() < tol * 10
    assert np.abs(np.einsum("nji,njk->nik", V, V) - np.eye(3)).max() < tol * 10
    # strains keep RELATIVE accuracy (this is why the kernels work on E = F - I)
    s_ref = np.linalg.svd(F[:n // 2], compute_uv=False)
    assert np.abs(np.sort(1 + e[:n // 2], axis=1)[:, ::-1] - s_ref).max() < (1e-14 if prec == 64 else 2e-7)


# prec 48: the two-width chain of float32 mode (collide_mixed_hybrid, round 5): the signed distance in f64, everything else in f32 - 1e-7 of the field's
# maximum where the all-f32 chain (prec 32, never shipped) makes 3e-5: what lets the device leave the 2,000-instruction f64 chain and keep the 1e-5 bar
@pytest.mark.parametrize("prec,tol", [(64, 1e-10), (48, 2e-6), (32, 2e-3)])
def test_collide_mixed_forward_and_adjoint(lib, prec, tol):
    d = H.load_palm()
    rng = np.random.default_rng(1)
    n = 1500
    he = np.array([0.3, 0.15, 0.075])
    loc = rng.uniform(-1, 1, (n, 3)) * he
    ax = rng.integers(0, 3, n); sgn = rng.choice([-1, 1], n)
    loc[np.arange(n), ax] = sgn * (he[ax] + rng.uniform(-0.004, 0.008, n))
    q = np.array([0.9, 0.1, -0.3, 0.2]); q /= np.linalg.norm(q); q *= 1.02
    pos0 = np.array([0.5, 0.4, 0.5])
    st13 = np.concatenate([pos0, q, [0.1, -0.2, 0.05], [0.3, 0.2, -0.4]])
    world = O.qrot(torch.tensor(q / np.linalg.norm(q)), torch.tensor(loc)).numpy() + pos0
    vel = 0.5 * rng.standard_normal((n, 3))
    if prec == 48:
        vel = vel.astype(np.float32).astype(np.float64)        # (the device hands this chain a float velocity: the grid gather's)
    g_v = rng.standard_normal((n, 3)); g_ext = rng.standard_normal(6)
    Pm = O.SimParams(n_grid=64, dt=2e-4)
    prim = O.make_prim(st13[:3], st13[3:7], st13[7:10], st13[10:], d["sdf"], d["normal"], d["lower"], d["upper"], d["dx"], friction=0.3)
    x = torch.tensor(world, requires_grad=True); v = torch.tensor(vel, requires_grad=True)
    leaves = [t.requires_grad_(True) for t in (prim.position, prim.rotation, prim.v, prim.w)]
    ov, ext = O.collide_mixed(prim, x, v, Pm.p_mass, Pm.dt, 0.25)
    gr = torch.autograd.grad((ov * torch.tensor(g_v)).sum() + (ext * torch.tensor(g_ext)).sum(), [x, v] + leaves)
    out_v, out_ext = np.zeros((n, 3)), np.zeros((n, 6))
    act = np.zeros(n, dtype=np.int32); g_pos, g_vin, g_state = np.zeros((n, 3)), np.zeros((n, 3)), np.zeros(13)
    res = np.asarray(d["res"], dtype=np.int32)
    lib.h_collide_mixed(prec, n, P(np.ascontiguousarray(d["sdf"])), P(np.ascontiguousarray(d["normal"])), res.ctypes.data_as(ip),
                        P(d["lower"]), P(d["upper"]), ctypes.c_double(d["dx"]), ctypes.c_double(0.3), ctypes.c_double(666.0), P(st13),
                        P(world), P(vel), ctypes.c_double(Pm.p_mass), ctypes.c_double(Pm.dt), ctypes.c_double(0.25), P(g_v), P(g_ext),
                        P(out_v), P(out_ext), act.ctypes.data_as(ip), P(g_pos), P(g_vin), P(g_state))
    assert act.sum() > 500
    assert np.abs(out_v - ov.detach().numpy()).max() < tol * (np.abs(ov.detach().numpy()).max() if prec == 48 else 1.0)
    assert H.rel_err(out_ext.sum(0), ext.detach().numpy()) < tol
    assert H.rel_err(g_pos, gr[0].numpy()) < tol and H.rel_err(g_vin, gr[1].numpy()) < tol
    assert H.rel_err(g_state, torch.cat(gr[2:]).numpy()) < tol


@pytest.mark.parametrize("prec,tol", [(64, 1e-14), (32, 1e-6)])
def test_forward_kinematics(lib, prec, tol):
    s13 = np.array([0.5, 0.4, 0.5, 0.9, 0.1, -0.3, 0.2, 0.1, -0.2, 0.05, 0.3, 0.2, -0.4])
    s13[3:7] /= np.linalg.norm(s13[3:7])
    out7 = np.zeros(7)
    lib.h_forward_kinematics(prec, P(s13), ctypes.c_double(2e-4), P(out7))
    t = lambda a: torch.tensor(a, dtype=O.DT)
    p, r = O.forward_kinematics(t(s13[:3]), t(s13[3:7]), t(s13[7:10]), t(s13[10:]), 2e-4)
    assert np.abs(out7[:3] - p.numpy()).max() < tol and np.abs(out7[3:] - r.numpy()).max() < tol


@pytest.mark.parametrize("prec,tol", [(64, 1e-10), (32, 2e-3)])
@pytest.mark.parametrize("kind", [1, 0])
def test_collide_particle_and_grid(lib, prec, tol, kind):
    """collision_type 1 (penalty impulse in p2g) and 0 (grid-node projection in grid_op) vs the oracle, with adjoints."""
    d = H.load_palm()
    rng = np.random.default_rng(4 + kind)
    n = 1200
    he = np.array([0.3, 0.15, 0.075])
    loc = rng.uniform(-1, 1, (n, 3)) * he
    ax = rng.integers(0, 3, n); sgn = rng.choice([-1, 1], n)
    loc[np.arange(n), ax] = sgn * (he[ax] + rng.uniform(-0.004, 0.008, n))
    q = np.array([0.9, 0.1, -0.3, 0.2]); q /= np.linalg.norm(q)
    pos0 = np.array([0.5, 0.4, 0.5])
    st13 = np.concatenate([pos0, q, [0.1, -0.2, 0.05], [0.3, 0.2, -0.4]])
    world = O.qrot(torch.tensor(q), torch.tensor(loc)).numpy() + pos0
    vel = 0.5 * rng.standard_normal((n, 3)); mass = rng.uniform(1e-5, 1e-4, n)
    g_out = rng.standard_normal((n, 3)); g_ext = rng.standard_normal(6)
    dt = 2e-4
    prim = O.make_prim(st13[:3], st13[3:7], st13[7:10], st13[10:], d["sdf"], d["normal"], d["lower"], d["upper"], d["dx"], friction=0.3)
    x = torch.tensor(world, requires_grad=True); v = torch.tensor(vel, requires_grad=True); m = torch.tensor(mass, requires_grad=True)
    leaves = [t.requires_grad_(True) for t in (prim.position, prim.rotation, prim.v, prim.w)]
    if kind == 1:
        out, ext = O.collide_particle(prim, x, v, dt)
    else:
        out, ext = O.collide_grid(prim, x.detach(), v, dt, m)
    L = (out * torch.tensor(g_out)).sum() + (ext * torch.tensor(g_ext)).sum()
    ins = ([x] if kind == 1 else []) + [v] + ([m] if kind == 0 else []) + leaves
    gr = torch.autograd.grad(L, ins, allow_unused=True)
    gr = [torch.zeros_like(i) if g is None else g for g, i in zip(gr, ins)]
    out3, out_ext = np.zeros((n, 3)), np.zeros((n, 6))
    act = np.zeros(n, dtype=np.int32); g_in = np.zeros((n, 20))
    res = np.asarray(d["res"], dtype=np.int32)
    lib.h_collide_other(prec, kind, n, P(np.ascontiguousarray(d["sdf"])), P(np.ascontiguousarray(d["normal"])), res.ctypes.data_as(ip),
                        P(d["lower"]), P(d["upper"]), ctypes.c_double(d["dx"]), ctypes.c_double(0.3), ctypes.c_double(666.0), P(st13),
                        P(world), P(vel), P(mass), ctypes.c_double(dt), P(g_out), P(g_ext), P(out3), P(out_ext), act.ctypes.data_as(ip), P(g_in))
    assert act.sum() > 300
    assert H.rel_err(out3, out.detach().numpy()) < tol
    assert H.rel_err(out_ext.sum(0), ext.detach().numpy()) < tol
    k = 0
    if kind == 1:
        assert H.rel_err(g_in[:, 0:3], gr[k].numpy()) < tol; k += 1
    assert H.rel_err(g_in[:, 3:6], gr[k].numpy()) < tol; k += 1
    if kind == 0:
        assert H.rel_err(g_in[:, 6], gr[k].numpy()) < tol; k += 1
    assert H.rel_err(g_in[:, 7:20].sum(0), torch.cat(gr[k:]).numpy()) < tol
