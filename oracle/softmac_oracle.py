"""CPU oracle for the SoftMAC per-substep MPM hot path.  TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this
module.  The product (softmac_amd/) never imports, links or calls anything under oracle/.

What it is
----------
A float64 restatement, kernel by kernel, of the reference's MLS-MPM substep
(`softmac/engine/mpm_simulator.py`) and of the contact device functions it inlines
(`softmac/engine/primitive/primitive_base.py`, `mesh.py`, `primitive_utils.py`), written with
vectorised torch ops so that `torch.autograd` provides the adjoint of every kernel
*independently* of the hand-derived adjoints in the HIP kernels and in oracle/mpm_cpu.c.

PARITY UNPINNED at three third-party boundaries (none of them has source under /root/reference
and the reference has no tests or golden vectors; Taichi is not installed here - a plain
ModuleNotFoundError, nothing was refused - so the reference cannot run):
  * `ti.svd` (Taichi 1.4.1 built-in; call site mpm_simulator.py:133).  The oracle uses an exact
    f64 SVD; every consumer (R = U V^T, U clip(S) V^T, the `backward_svd` formula) is invariant
    to the SVD gauge, so any accurate SVD agrees mathematically.
  * Taichi's source-to-source reverse-mode AD (`kernel.grad`, call sites mpm_simulator.py:361-374,
    389-394).  The oracle follows Taichi's documented conventions: no gradient through integer
    casts or branch predicates or `ti.cast(cond, dtype)` flags; sub-gradient routing of
    min/max/abs follows torch (ties are avoided in every fixture).  The SVD adjoint is NOT
    autograd's: it is the reference's own hand-written `backward_svd` (mpm_simulator.py:140-157)
    including its +-1e-6 clamp, wrapped as a custom autograd Function.
  * Taichi's typing of python literals / captured floats (believed f32 constants; 3e-8 relative).
What pins it instead: analytic invariants, central finite differences (tests/test_oracle_*.py),
the reference's own data files (envs/*.npy initial states, the palm SDF cache) as inputs, and
agreement between this file and the independent plain-C restatement oracle/mpm_cpu.c.

Layout conventions (reference: AOS fields, here plain tensors)
  x,v: (N,3)   C,F: (N,3,3)   grid vectors: (n,n,n,3)   grid scalars: (n,n,n)
"""
from __future__ import annotations

import dataclasses
import math
from typing import List, Optional, Sequence

import torch

DT = torch.float64

MODEL_COROTATED, MODEL_NEOHOOKEAN = 0, 1          # mpm_simulator.py:4-5
MAT_PLASTIC, MAT_ELASTIC, MAT_LIQUID = 0, 1, 2     # :7-9
CONTACT_GRID, CONTACT_PARTICLE, CONTACT_MIXED = 0, 1, 2  # :11-13

SDF_INF = 1e10                                     # mesh.py:12


# ----------------------------------------------------------------------------------------------
# parameters (mpm_simulator.py:17-52)
# ----------------------------------------------------------------------------------------------
@dataclasses.dataclass
class SimParams:
    n_grid: int = 64
    dt: float = 2e-4
    E: float = 3e3
    nu: float = 0.2
    ptype: int = MAT_PLASTIC
    material_model: int = MODEL_COROTATED
    gravity: Sequence[float] = (0.0, -9.8, 0.0)
    ground_friction: float = 20.0
    collision_type: int = CONTACT_MIXED
    substeps: int = 5
    n_control: int = 0

    def __post_init__(self):
        self.dx = 1.0 / self.n_grid                       # :32
        self.inv_dx = float(self.n_grid)
        self.p_vol = (self.dx * 0.5) ** 2                 # :34 (sic: squared even in 3-D)
        self.p_mass = self.p_vol * 1.0                    # :35
        mu = self.E / (2 * (1 + self.nu))                 # :41
        lam = self.E * self.nu / ((1 + self.nu) * (1 - 2 * self.nu))
        if self.ptype == MAT_ELASTIC:                     # :42-43
            mu, lam = 0.3 * mu, 0.3 * lam
        elif self.ptype == MAT_LIQUID:                    # :44-45
            mu = 0.0
        self.mu, self.lam = mu, lam


@dataclasses.dataclass
class RigidPrim:
    """One `Mesh` primitive at one frame (primitive_base.py:26-43, mesh.py:35-43)."""
    position: torch.Tensor            # (3,)
    rotation: torch.Tensor            # (4,) quaternion w,x,y,z
    v: torch.Tensor                   # (3,) body-frame linear velocity (see collider_v)
    w: torch.Tensor                   # (3,)
    sdf_table: torch.Tensor           # (rx,ry,rz)
    normal_table: torch.Tensor        # (rx,ry,rz,3)
    lower: torch.Tensor               # (3,)
    upper: torch.Tensor               # (3,)
    sdf_dx: float
    friction: float = 0.9
    softness: float = 666.0           # primitives.py:55-56
    contact: bool = True              # mpm_simulator.py:70 primitives_contact[i]


# ----------------------------------------------------------------------------------------------
# quaternion helpers (primitive_utils.py)
# ----------------------------------------------------------------------------------------------
def length(x):                                              # primitive_utils.py:3-5
    return torch.sqrt((x * x).sum(-1) + 1e-8)


def qrot(rot, v):                                           # primitive_utils.py:7-13
    qvec = rot[..., 1:4].expand(v.shape)
    uv = torch.linalg.cross(qvec, v)
    uuv = torch.linalg.cross(qvec, uv)
    return v + 2 * (rot[..., 0:1] * uv + uuv)


def qmul(q, r):                                             # primitive_utils.py:19-27
    t = torch.outer(r, q)                                   # terms = r.outer_product(q)
    w = t[0, 0] - t[1, 1] - t[2, 2] - t[3, 3]
    x = t[0, 1] + t[1, 0] - t[2, 3] + t[3, 2]
    y = t[0, 2] + t[1, 3] + t[2, 0] - t[3, 1]
    z = t[0, 3] - t[1, 2] + t[2, 1] + t[3, 0]
    out = torch.stack([w, x, y, z])
    return out / torch.sqrt((out * out).sum())


def w2quat(axis_angle):                                     # primitive_utils.py:29-40
    w = torch.sqrt((axis_angle * axis_angle).sum() + 1e-12)  # norm(eps) = sqrt(|a|^2 + eps)
    v = (axis_angle / w) * torch.sin(w / 2)
    return torch.cat([torch.cos(w / 2).reshape(1), v])


def inv_trans(pos, position, rotation):                     # primitive_utils.py:42-46
    iq = torch.stack([rotation[0], -rotation[1], -rotation[2], -rotation[3]])
    iq = iq / torch.sqrt((iq * iq).sum())
    return qrot(iq, pos - position)


def forward_kinematics(position, rotation, v, w, dt):       # primitive_base.py:280-283
    return position + v * dt, qmul(w2quat(w * dt), rotation)


# ----------------------------------------------------------------------------------------------
# SDF tables (mesh.py:45-113)
# ----------------------------------------------------------------------------------------------
def _trilinear(prim: RigidPrim, local_pos, table):
    """8-tap trilinear lookup used by both `_sdf` (mesh.py:55-65) and `_normal` (:99-109).
    Returns (value, in_box).  Outside the box the caller substitutes its constant."""
    in_box = ((local_pos >= prim.lower) & (local_pos < prim.upper)).all(-1)   # :50-52 / :93-95
    safe = torch.where(in_box[:, None], local_pos, prim.lower.expand_as(local_pos))
    pos = (safe - prim.lower) * (1.0 / prim.sdf_dx)
    base = pos.detach().to(torch.int64)                      # ti.cast(pos, i32): no gradient
    fx = pos - base.to(DT)
    w = [1.0 - fx, fx]
    res = table.shape[:3]
    out = 0
    for i in (0, 1):
        for j in (0, 1):
            for k in (0, 1):
                weight = w[i][:, 0] * w[j][:, 1] * w[k][:, 2]
                ii = (base[:, 0] + i).clamp(max=res[0] - 1)
                jj = (base[:, 1] + j).clamp(max=res[1] - 1)
                kk = (base[:, 2] + k).clamp(max=res[2] - 1)
                val = table[ii, jj, kk]
                out = out + (weight[:, None] * val if val.dim() == 2 else weight * val)
    return out, in_box


def prim_sdf(prim: RigidPrim, pos):                         # primitive_base.py:53-56 + mesh.py:45-68
    local = inv_trans(pos, prim.position, prim.rotation)
    val, in_box = _trilinear(prim, local, prim.sdf_table)
    return torch.where(in_box, val, torch.full_like(val, SDF_INF))


def prim_normal(prim: RigidPrim, pos):                      # primitive_base.py:58-61 + mesh.py:90-113
    local = inv_trans(pos, prim.position, prim.rotation)
    val, in_box = _trilinear(prim, local, prim.normal_table)
    safe = torch.where(in_box[:, None], val, torch.tensor([0.0, 1.0, 0.0], dtype=DT).expand_as(val))
    n = safe / torch.sqrt((safe * safe).sum(-1, keepdim=True))   # .normalized(), mesh.py:110
    n = torch.where(in_box[:, None], n, torch.tensor([0.0, 1.0, 0.0], dtype=DT).expand_as(n))
    return qrot(prim.rotation, n)                            # NB: un-normalised rotation, as in :61


def collider_v(prim: RigidPrim, r):                         # primitive_base.py:63-70
    quat = prim.rotation / torch.sqrt((prim.rotation * prim.rotation).sum())
    inv_quat = torch.stack([quat[0], -quat[1], -quat[2], -quat[3]])
    r_local = qrot(inv_quat, r)
    v_local = prim.v + torch.linalg.cross(prim.w.expand_as(r_local), r_local)
    return qrot(quat, v_local)


# ----------------------------------------------------------------------------------------------
# contact models (primitive_base.py:72-181).  Each returns (new velocity / impulse, ext_f(6))
# ----------------------------------------------------------------------------------------------
def collide_mixed(prim: RigidPrim, p_pos, p_v, p_mass, dt, life):   # primitive_base.py:139-181
    dist = prim_sdf(prim, p_pos)
    active = dist <= 5e-3                                   # :142-143
    p_v_in = p_v
    D = prim_normal(prim, p_pos)
    r = p_pos - prim.position
    cv = collider_v(prim, r)
    input_v = p_v - cv                                       # :149
    nc = (input_v * D).sum(-1)                               # :150
    approaching = nc < 0                                     # :152
    p_v_t = input_v - nc[:, None] * D                        # :153
    p_v_t_norm = length(p_v_t)                               # :154 (eps 1e-8)
    fr = p_v_t / p_v_t_norm[:, None] * torch.clamp(p_v_t_norm + nc * prim.friction, min=0.0)[:, None]
    flag = (approaching & (torch.sqrt((p_v_t * p_v_t).sum(-1)) > 1e-30)).to(DT)[:, None]  # :156
    p_v_t2 = fr * flag + p_v_t * (1 - flag)                  # :157
    v_hit = cv + p_v_t2                                      # :159
    influence = torch.clamp(torch.exp(-dist * prim.softness), max=1.0)          # :162
    v_soft = cv + input_v * (1 - influence)[:, None] + p_v_t2 * influence[:, None]  # :163
    v_hit = torch.where((dist > 0)[:, None], v_soft, v_hit)  # :161
    pv = torch.where(approaching[:, None], v_hit, p_v)
    # move penetrated particles to surface (:165-170)
    x_new = pv * dt + p_pos
    sdf2 = prim_sdf(prim, x_new)
    pen = sdf2 < 0
    n2 = prim_normal(prim, x_new)
    sdf2s = torch.where(pen, sdf2, torch.zeros_like(sdf2))
    pv = torch.where(pen[:, None], pv - (sdf2s / dt)[:, None] * n2 * life, pv)
    # force on rigid body (:172-179)
    b_f = p_mass * (p_v_in - pv) * (1.0 / dt)
    b_t = torch.linalg.cross(r, b_f)
    act = active[:, None]
    out_v = torch.where(act, pv, p_v)
    ext = torch.cat([torch.where(act, b_f, torch.zeros_like(b_f)).sum(0),
                     torch.where(act, b_t, torch.zeros_like(b_t)).sum(0)])
    return out_v, ext


def collide_particle(prim: RigidPrim, p_pos, p_v, dt):      # primitive_base.py:105-137
    dist = prim_sdf(prim, p_pos)
    c = dist - 5e-3
    active = c < 0.0
    cs = torch.where(active, c, torch.zeros_like(c))
    D = prim_normal(prim, p_pos)
    r = p_pos - prim.position
    cv = collider_v(prim, r)
    input_v = p_v - cv
    nc = (input_v * D).sum(-1)
    p_v_t = input_v - nc[:, None] * D
    f1 = -D * cs[:, None] * 50.0                             # :120-121
    p_v_t_norm = torch.sqrt((p_v_t * p_v_t).sum(-1) + 1e-8)  # :124
    f2 = -p_v_t / p_v_t_norm[:, None] * torch.abs(nc)[:, None] * prim.friction   # :126
    p_f = (f1 + f2)
    b_f = -(f1 + f2)
    b_t = torch.linalg.cross(r, b_f)
    act = active[:, None]
    zero = torch.zeros_like(p_f)
    ext = torch.cat([torch.where(act, b_f, zero).sum(0), torch.where(act, b_t, zero).sum(0)])
    return torch.where(act, p_f, zero) * dt, ext             # :137


def collide_grid(prim: RigidPrim, grid_pos, v_out, dt, grid_m):     # primitive_base.py:72-103
    dist = prim_sdf(prim, grid_pos)
    influence = torch.clamp(torch.exp(-dist * prim.softness), max=1.0)           # :75
    active = ((prim.softness > 0) & (influence > 0.1)) | (dist <= 0)             # :76
    v_in = v_out
    D = prim_normal(prim, grid_pos)
    r = grid_pos - prim.position
    cv = collider_v(prim, r)
    input_v = v_out - cv
    nc = (input_v * D).sum(-1)
    g_t = input_v - torch.clamp(nc, max=0.0)[:, None] * D    # :86
    g_t_norm = length(g_t)
    fr = g_t / g_t_norm[:, None] * torch.clamp(g_t_norm + nc * prim.friction, min=0.0)[:, None]
    flag = ((nc < 0) & (torch.sqrt((g_t * g_t).sum(-1)) > 1e-30)).to(DT)[:, None]
    g_t2 = fr * flag + g_t * (1 - flag)
    new_v = cv + input_v * (1 - influence)[:, None] + g_t2 * influence[:, None]  # :92
    b_f = grid_m[:, None] * (v_in - new_v) * (1.0 / dt)
    b_t = torch.linalg.cross(r, b_f)
    act = active[:, None]
    zero = torch.zeros_like(b_f)
    ext = torch.cat([torch.where(act, b_f, zero).sum(0), torch.where(act, b_t, zero).sum(0)])
    return torch.where(act, new_v, v_out), ext


# ----------------------------------------------------------------------------------------------
# SVD with the reference's hand-written adjoint (mpm_simulator.py:130-157, 184-192)
# ----------------------------------------------------------------------------------------------
def _clamp_ref(a):                                          # mpm_simulator.py:184-192
    return torch.where(a >= 0, torch.clamp(a, min=1e-6), torch.clamp(a, max=-1e-6))


def backward_svd(gu, gsigma, gv, u, sig, v):                # mpm_simulator.py:140-157
    vt, ut = v.transpose(-1, -2), u.transpose(-1, -2)
    sigma_term = u @ gsigma @ vt
    s = torch.diagonal(sig, dim1=-2, dim2=-1) ** 2
    diff = s[:, None, :] - s[:, :, None]                    # [i,j] = s[j] - s[i]
    Fm = 1.0 / _clamp_ref(diff)
    Fm = Fm * (1 - torch.eye(3, dtype=DT))                  # i == j -> 0
    u_term = u @ ((Fm * (ut @ gu - gu.transpose(-1, -2) @ u)) @ sig) @ vt
    v_term = u @ (sig @ ((Fm * (vt @ gv - gv.transpose(-1, -2) @ v)) @ vt))
    return u_term + v_term + sigma_term


class _SVD3(torch.autograd.Function):
    """U, sig(3x3 diagonal matrix), V = svd(F), U,V in SO(3) for det F > 0 (ti.svd contract)."""

    @staticmethod
    def forward(ctx, F):
        U, S, Vh = torch.linalg.svd(F)
        V = Vh.transpose(-1, -2)
        # rotation variant: move reflections into the last singular value's sign
        du, dv = torch.linalg.det(U), torch.linalg.det(V)
        U = U.clone(); V = V.clone(); S = S.clone()
        U[:, :, 2] *= torch.sign(du)[:, None]
        V[:, :, 2] *= torch.sign(dv)[:, None]
        S[:, 2] *= torch.sign(du) * torch.sign(dv)
        sig = torch.diag_embed(S)
        ctx.save_for_backward(U, sig, V)
        return U, sig, V

    @staticmethod
    def backward(ctx, gu, gsig, gv):
        U, sig, V = ctx.saved_tensors
        return backward_svd(gu, gsig, gv, U, sig, V)


def svd3(F):
    return _SVD3.apply(F)


# ----------------------------------------------------------------------------------------------
# MPM kernels (mpm_simulator.py)
# ----------------------------------------------------------------------------------------------
def bspline(x, inv_dx):                                      # :215-217 (same at :302-304, 409-411, 434-436)
    xs = x * inv_dx
    base = (xs.detach() - 0.5).to(torch.int64)              # .cast(int): truncation, no gradient
    fx = xs - base.to(DT)
    w = [0.5 * (1.5 - fx) ** 2, 0.75 - (fx - 1.0) ** 2, 0.5 * (fx - 0.5) ** 2]
    return base, fx, w


def _flat(base, i, j, k, n):
    return ((base[:, 0] + i) * n + (base[:, 1] + j)) * n + (base[:, 2] + k)


def compute_F_tmp(C, F, dt):                                 # :125-128
    return (torch.eye(3, dtype=DT) + dt * C) @ F


def constitutive(F_tmp, U, sig, V, P: SimParams):            # :219-250 -> (new_F, stress)
    eye = torch.eye(3, dtype=DT)
    new_F = F_tmp
    J = torch.linalg.det(F_tmp)                              # :222 (pre-projection F_tmp)
    if P.material_model == MODEL_COROTATED:
        if P.ptype == MAT_PLASTIC:                           # :226-229
            s = torch.diagonal(sig, dim1=-2, dim2=-1)
            s_new = torch.minimum(torch.maximum(s, torch.tensor(1 - 2e-3, dtype=DT)),
                                  torch.tensor(1 + 3e-3, dtype=DT))
            new_F = U @ torch.diag_embed(s_new) @ V.transpose(-1, -2)
        elif P.ptype == MAT_LIQUID:                          # :233
            new_F = eye * torch.pow(J, 1.0 / 3.0)[:, None, None]
        r = U @ V.transpose(-1, -2)                          # :234
        stress = 2 * P.mu * (new_F - r) @ new_F.transpose(-1, -2) \
            + eye * (P.lam * J * (J - 1))[:, None, None]     # :235-236
    else:                                                    # :237-245
        if P.ptype == MAT_LIQUID:
            sq = torch.sqrt(J)
            new_F = torch.diag_embed(torch.stack([sq, sq, torch.ones_like(sq)], -1))
        stress = P.mu * (new_F @ new_F.transpose(-1, -2)) \
            + eye * (P.lam * torch.log(J) - P.mu)[:, None, None]
    return new_F, stress


def p2g(x, v, C, F_tmp, U, sig, V, P: SimParams, prims: Sequence[RigidPrim] = (),
        control_idx=None, action=None):                      # :198-262
    """Returns new_F (N,3,3), grid_v_in (n,n,n,3), grid_m (n,n,n), ext_f list (particle contact)."""
    n = P.n_grid
    N = x.shape[0]
    impulse = torch.zeros_like(x)
    ext_fs = [torch.zeros(6, dtype=DT) for _ in prims]
    if P.collision_type == CONTACT_PARTICLE:                 # :203-206
        for i, pr in enumerate(prims):
            if pr.contact:
                imp, ext = collide_particle(pr, x, v, P.dt)
                impulse = impulse + imp
                ext_fs[i] = ext_fs[i] + ext
    if P.n_control > 0 and control_idx is not None:          # :209-213
        sel = control_idx >= 0
        a = action[control_idx.clamp(min=0)]
        impulse = impulse + torch.where(sel[:, None], 6e-4 * a * P.dt, torch.zeros_like(a))
    base, fx, w = bspline(x, P.inv_dx)
    new_F, stress = constitutive(F_tmp, U, sig, V, P)
    stress = (-P.dt * P.p_vol * 4 * P.inv_dx * P.inv_dx) * stress        # :247
    affine = stress + P.p_mass * C                                       # :248
    gv = torch.zeros(n * n * n, 3, dtype=DT)
    gm = torch.zeros(n * n * n, dtype=DT)
    for i in range(3):
        for j in range(3):
            for k in range(3):
                off = torch.tensor([i, j, k], dtype=DT)
                dpos = (off - fx) * P.dx                                  # :254
                weight = w[i][:, 0] * w[j][:, 1] * w[k][:, 2]            # :255-257
                idx = _flat(base, i, j, k, n)
                mom = P.p_mass * v + (affine @ dpos[:, :, None])[:, :, 0] + impulse
                gv = gv.index_add(0, idx, weight[:, None] * mom)          # :261
                gm = gm.index_add(0, idx, weight * P.p_mass)              # :262
    return new_F, gv.reshape(n, n, n, 3), gm.reshape(n, n, n), ext_fs


def boundary_condition(v_out, P: SimParams):                 # :268-281, v_out (n,n,n,3)
    n = P.n_grid
    I = torch.arange(n)
    lo = I < 3
    hi = I > n - 3
    comps = []
    for d in range(3):
        shape = [1, 1, 1]
        shape[d] = n
        vd = v_out[..., d]
        vd = torch.where(lo.reshape(shape) & (vd < 0), torch.zeros_like(vd), vd)
        vd = torch.where(hi.reshape(shape) & (vd > 0), torch.zeros_like(vd), vd)
        comps.append(vd)
    out = torch.stack(comps, -1)
    if P.ground_friction >= 10.0:                            # :278-279 sticky floor
        out = torch.where(lo.reshape(1, n, 1, 1), torch.zeros_like(out), out)
    return out


def _grid_velocity(grid_m, grid_v_in, P: SimParams):         # :286-288 / :399-401
    has = grid_m > 1e-10
    m_safe = torch.where(has, grid_m, torch.ones_like(grid_m))
    v_out = (1.0 / m_safe)[..., None] * grid_v_in + P.dt * torch.tensor(P.gravity, dtype=DT)
    return has, v_out


def grid_op(grid_m, grid_v_in, P: SimParams, prims: Sequence[RigidPrim] = ()):   # :283-297
    n = P.n_grid
    has, v_out = _grid_velocity(grid_m, grid_v_in, P)
    ext_fs = [torch.zeros(6, dtype=DT) for _ in prims]
    if P.collision_type == CONTACT_GRID:                     # :290-294
        I = torch.stack(torch.meshgrid(torch.arange(n), torch.arange(n), torch.arange(n), indexing="ij"), -1)
        pos = (I.to(DT) * P.dx).reshape(-1, 3)
        hv = has.reshape(-1)
        vf = v_out.reshape(-1, 3)
        mf = grid_m.reshape(-1)
        sel = hv.nonzero()[:, 0]
        for i, pr in enumerate(prims):
            if pr.contact:
                nv, ext = collide_grid(pr, pos[sel], vf[sel], P.dt, mf[sel])
                vf = vf.index_put((sel,), nv)
                ext_fs[i] = ext_fs[i] + ext
        v_out = vf.reshape(n, n, n, 3)
    v_out = boundary_condition(v_out, P)
    return torch.where(has[..., None], v_out, torch.zeros_like(v_out)), ext_fs


def grid_op_mixed1(grid_m, grid_v_in, P: SimParams):         # :396-404
    has, v_out = _grid_velocity(grid_m, grid_v_in, P)
    v_out = boundary_condition(v_out, P)
    grid_v_mixed = torch.where(has[..., None], v_out, torch.zeros_like(v_out))
    return grid_v_mixed                                      # grid_v_out (cleared) += grid_v_mixed


def grid_op_mixed2(x, grid_v_mixed, P: SimParams):           # :406-419
    n = P.n_grid
    base, fx, w = bspline(x, P.inv_dx)
    g = grid_v_mixed.reshape(-1, 3)
    new_v = torch.zeros_like(x)
    for i in range(3):
        for j in range(3):
            for k in range(3):
                weight = w[i][:, 0] * w[j][:, 1] * w[k][:, 2]
                new_v = new_v + weight[:, None] * g[_flat(base, i, j, k, n)]
    return new_v


def grid_op_mixed3(x, v_tmp, prims: Sequence[RigidPrim], P: SimParams, f: int):   # :421-429
    v_tgt = v_tmp
    life = 1.0 / (P.substeps - f % P.substeps)               # :425
    ext_fs = []
    for pr in prims:
        if pr.contact:
            v_tgt, ext = collide_mixed(pr, x, v_tgt, P.p_mass, P.dt, life)
        else:
            ext = torch.zeros(6, dtype=DT)
        ext_fs.append(ext)
    return v_tgt, ext_fs


def grid_op_mixed4(x, v_tmp, v_tgt, grid_m, grid_v_out, P: SimParams):            # :431-443
    n = P.n_grid
    base, fx, w = bspline(x, P.inv_dx)
    gm = grid_m.reshape(-1)
    out = grid_v_out.reshape(-1, 3)
    alpha = 2.0
    for i in range(3):
        for j in range(3):
            for k in range(3):
                weight = w[i][:, 0] * w[j][:, 1] * w[k][:, 2]
                idx = _flat(base, i, j, k, n)
                has = (gm[idx] > 1e-10).to(DT)
                out = out.index_add(0, idx, -(alpha * weight * has)[:, None] * (v_tmp - v_tgt))
    return out.reshape(n, n, n, 3)


def g2p(x, grid_v_out, P: SimParams):                        # :299-318
    n = P.n_grid
    base, fx, w = bspline(x, P.inv_dx)
    g = grid_v_out.reshape(-1, 3)
    new_v = torch.zeros_like(x)
    new_C = torch.zeros(x.shape[0], 3, 3, dtype=DT)
    for i in range(3):
        for j in range(3):
            for k in range(3):
                dpos = torch.tensor([i, j, k], dtype=DT) - fx
                weight = w[i][:, 0] * w[j][:, 1] * w[k][:, 2]
                g_v = g[_flat(base, i, j, k, n)]
                new_v = new_v + weight[:, None] * g_v
                new_C = new_C + 4 * P.inv_dx * weight[:, None, None] * g_v[:, :, None] * dpos[:, None, :]
    new_x = x + P.dt * new_v                                 # :318
    return new_x, new_v, new_C


# ----------------------------------------------------------------------------------------------
# orchestration (mpm_simulator.py:320-378)
# ----------------------------------------------------------------------------------------------
def substep(x, v, C, F, P: SimParams, prims: Sequence[RigidPrim] = (), f: int = 0,
            control_idx=None, action=None, return_intermediates=False):
    """One forward substep.  Returns (x', v', C', F', ext_f list[6-vector per primitive])."""
    F_tmp = compute_F_tmp(C, F, P.dt)                        # :324
    if P.material_model == MODEL_COROTATED:
        U, sig, V = svd3(F_tmp)                              # :325-326
    else:
        U = sig = V = None
    new_F, grid_v_in, grid_m, ext_p = p2g(x, v, C, F_tmp, U, sig, V, P, prims, control_idx, action)  # :327
    ext_fs = ext_p
    if P.collision_type == CONTACT_MIXED:                    # :333-334, 383-387
        grid_v_mixed = grid_op_mixed1(grid_m, grid_v_in, P)
        grid_v_out = grid_v_mixed
        if any(pr.contact for pr in prims):
            v_tmp = grid_op_mixed2(x, grid_v_mixed, P)
            v_tgt, ext_m = grid_op_mixed3(x, v_tmp, prims, P, f)
            grid_v_out = grid_op_mixed4(x, v_tmp, v_tgt, grid_m, grid_v_out, P)
            ext_fs = [a + b for a, b in zip(ext_fs, ext_m)]
    else:
        grid_v_out, ext_g = grid_op(grid_m, grid_v_in, P, prims)    # :335-336
        ext_fs = [a + b for a, b in zip(ext_fs, ext_g)]
    new_x, new_v, new_C = g2p(x, grid_v_out, P)              # :337
    if return_intermediates:
        inter = dict(F_tmp=F_tmp, grid_v_in=grid_v_in, grid_m=grid_m, grid_v_out=grid_v_out)
        if P.collision_type == CONTACT_MIXED:
            inter["grid_v_mixed"] = grid_v_mixed
        return new_x, new_v, new_C, new_F, ext_fs, inter
    return new_x, new_v, new_C, new_F, ext_fs


def substep_grad(x, v, C, F, P: SimParams, prims: Sequence[RigidPrim], f: int,
                 gx1, gv1, gC1, gF1, ext_f_grad: Optional[Sequence[torch.Tensor]] = None,
                 control_idx=None, action=None):
    """Adjoint of one substep (what `substep_grad`, mpm_simulator.py:339-378, accumulates).

    Inputs: state at frame f, adjoints of frame f+1, seeds on each primitive's ext_f.
    Returns dict with gx,gv,gC,gF (the `+=` increments of frame f's adjoints),
    per-primitive (gpos,grot,gv,gw) increments at frame f, and action grad.
    """
    leaves = [t.detach().clone().requires_grad_(True) for t in (x, v, C, F)]
    pl = []
    prims_l = []
    for pr in prims:
        st = [t.detach().clone().requires_grad_(True) for t in (pr.position, pr.rotation, pr.v, pr.w)]
        pl.append(st)
        prims_l.append(dataclasses.replace(pr, position=st[0], rotation=st[1], v=st[2], w=st[3]))
    act = None
    if action is not None:
        act = action.detach().clone().requires_grad_(True)
    nx, nv, nC, nF, ext = substep(*leaves, P, prims_l, f, control_idx, act)
    total = (nx * gx1).sum() + (nv * gv1).sum() + (nC * gC1).sum() + (nF * gF1).sum()
    if ext_f_grad is not None:
        for e, g in zip(ext, ext_f_grad):
            total = total + (e * g).sum()
    inputs = list(leaves) + [t for st in pl for t in st] + ([act] if act is not None else [])
    grads = torch.autograd.grad(total, inputs, allow_unused=True)
    grads = [torch.zeros_like(i) if g is None else g for g, i in zip(grads, inputs)]
    out = dict(gx=grads[0], gv=grads[1], gC=grads[2], gF=grads[3], prims=[], action=None)
    k = 4
    for _ in prims:
        out["prims"].append(tuple(grads[k:k + 4]))
        k += 4
    if act is not None:
        out["action"] = grads[k]
    return out


# ----------------------------------------------------------------------------------------------
# helpers used by tests / golden generation
# ----------------------------------------------------------------------------------------------
def state24_split(s):                                        # layout: mpm_simulator.py:481-489, 503-512
    s = torch.as_tensor(s, dtype=DT)
    return s[:, 0:3].clone(), s[:, 3:6].clone(), s[:, 15:24].reshape(-1, 3, 3).clone(), \
        s[:, 6:15].reshape(-1, 3, 3).clone()                 # x, v, C, F


def make_prim(position, rotation, v, w, sdf, normal, lower, upper, dx, friction=0.9, softness=666.0,
              contact=True):
    t = lambda a: torch.as_tensor(a, dtype=DT)
    return RigidPrim(t(position), t(rotation), t(v), t(w), t(sdf), t(normal), t(lower), t(upper),
                     float(dx), float(friction), float(softness), bool(contact))


def exp2quat(e):                                             # rigid_simulator_vel.py:46-55
    mag = math.sqrt(sum(float(c) ** 2 for c in e))
    if mag > 1e-10:
        s = abs(math.sin(mag / 2)) / mag
        return [math.cos(mag / 2), e[0] * s, e[1] * s, e[2] * s]
    return [1.0, 0.0, 0.0, 0.0]
