"""CPU oracle for the soft <-> cloth variant of the substep.  TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.

Restates, in float64 torch (autograd supplies every adjoint), what `soft_cloth/engine/mpm_simulator.py` and
`soft_cloth/engine/primitive/primitive_cloth.py` add to the substep of `softmac/engine/mpm_simulator.py`:
  * a length scale `mpm_scale` (dx = scale / n_grid, :30-35), walls only (no floor rule, :275-286);
  * von-Mises return mapping for the plastic material (:172-188, used at :232);
  * contact with ONE triangle-mesh primitive whose vertices move kinematically: per particle a contact face and a
    penetration flag (`get_contact_pair` :447-461, `trace_penetration_after_mpm` :484-510, `..._after_cloth` :520-545 -
    integer work, no gradient), the face's signed distance / normal (`primitive_cloth.py:142-164`), the forecast contact
    `collide_mixed` (:233-280) and the penalty contact `collide_particle` (:198-231) with their barycentric force splat
    onto the three vertices (:274-278);
  * the face-neighbourhood tables of `process_faces.py`.
The MPM kernels that are unchanged (b-spline, p2g scatter, grid_op_mixed1/2/4, g2p, the SVD with the reference's
backward_svd) are the ones of oracle/softmac_oracle.py.

PARITY UNPINNED like softmac_oracle.py (Taichi is not installed; diffcloth_py - the cloth dynamics - is closed source and
absent: the sheet is driven kinematically here).  Pinned by: finite differences of every adjoint, geometric
invariants (a flat sheet's signed distance, symmetry under face re-labelling), the reference's own meshes as inputs
(`envs/assets/tortilla/tortilla.obj`, `towel/towel.obj`, copied as data under tests/golden/).
"""
from __future__ import annotations

import collections
import dataclasses
from typing import Optional, Sequence

import numpy as np
import torch

from . import softmac_oracle as O

DT = O.DT


@dataclasses.dataclass
class ClothSimParams(O.SimParams):
    """soft_cloth/engine/mpm_simulator.py:16-52"""
    scale: float = 1.0
    yield_stress: float = 50.0

    def __post_init__(self):
        super().__post_init__()
        self.dx = 1.0 / self.n_grid * self.scale              # :31
        self.inv_dx = float(self.n_grid) / self.scale
        self.p_vol = (self.dx * 0.5) ** 2                     # :34
        self.p_mass = self.p_vol * 1.0
        self.ground_friction = 0.0                            # boundary_condition :275-286 has no floor rule


@dataclasses.dataclass
class ClothPrim:
    """`Primitive_Cloth` at one frame (primitive_cloth.py:26-69)"""
    position: torch.Tensor            # (V,3)
    velocity: torch.Tensor            # (V,3)
    faces: torch.Tensor               # (Fc,3) int64
    friction: float = 0.9
    softness: float = 666.0
    cloth_force_scale: float = 1.0
    sticky: bool = False
    mpm_scale: float = 1.0


# ----------------------------------------------------------------------------------------------
# process_faces.py:5-53: for every face the first n_neighbours faces in breadth-first order over shared edges, with a
# flag telling whether the neighbour's orientation is inverted relative to the face (edge traversed in the same direction)
# ----------------------------------------------------------------------------------------------
def process_faces(faces: np.ndarray, n_neighbours: int = 200):
    faces = np.asarray(faces)
    nf = faces.shape[0]
    edge_faces = collections.defaultdict(list)
    for i in range(nf):
        for j in range(3):
            a, b = int(faces[i, j]), int(faces[i, (j + 1) % 3])
            edge_faces[(min(a, b), max(a, b))].append(i)
    nb = np.zeros((nf, n_neighbours), dtype=np.int32)
    nbd = np.zeros((nf, n_neighbours), dtype=np.int8)
    for i in range(nf):
        found = []
        queue = collections.deque([(i, False)])
        visited = np.zeros(nf, dtype=bool)
        while queue:
            cur, inv = queue.popleft()
            if visited[cur]:
                continue
            found.append((cur, inv))
            if len(found) > n_neighbours:
                break
            visited[cur] = True
            for j in range(3):
                a, b = int(faces[cur, j]), int(faces[cur, (j + 1) % 3])
                for g in edge_faces[(min(a, b), max(a, b))]:
                    if g == cur:
                        continue
                    same_dir = any(int(faces[g, q]) == a and int(faces[g, (q + 1) % 3]) == b for q in range(3))
                    queue.append((g, (not inv) if same_dir else inv))
        found = found[1:]
        found += [(i, False)] * (n_neighbours - len(found))
        nb[i] = [c for c, _ in found]
        nbd[i] = [1 if v else 0 for _, v in found]
    return nb, nbd


# ----------------------------------------------------------------------------------------------
# triangle geometry (primitive_cloth.py:18-24, 75-196), batched over the leading dimension
# ----------------------------------------------------------------------------------------------
def length(x):                                               # :18-20
    return torch.sqrt((x * x).sum(-1) + 1e-14)


def normalize(n):                                            # :22-24
    return n / length(n)[..., None]


def closest_point_on_edge(p, x0, x1):                        # :83-96
    v, w = x1 - x0, p - x0
    c1, c2 = (w * v).sum(-1), (v * v).sum(-1)
    c2s = torch.where(c2 != 0, c2, torch.ones_like(c2))
    mid = x0 + v * (c1 / c2s)[..., None]
    out = torch.where((c1 > 0)[..., None], mid, x0)
    return torch.where((c1 >= c2)[..., None], x1, out)


def _safe_div(a, b):
    return a / torch.where(b != 0, b, torch.ones_like(b))


def barycentric_coordinate(p, x0, x1, x2):                   # :98-113
    A, B, Cc = x1 - x0, x2 - x0, p - x0
    dxy = A[..., 0] * B[..., 1] - A[..., 1] * B[..., 0]
    use_xz = dxy.abs() < 1e-10
    dxz = A[..., 0] * B[..., 2] - A[..., 2] * B[..., 0]
    w1 = torch.where(use_xz, _safe_div(Cc[..., 0] * B[..., 2] - Cc[..., 2] * B[..., 0], dxz),
                     _safe_div(Cc[..., 0] * B[..., 1] - Cc[..., 1] * B[..., 0], dxy))
    w2 = torch.where(use_xz, _safe_div(Cc[..., 0] * A[..., 2] - Cc[..., 2] * A[..., 0], -dxz),
                     _safe_div(Cc[..., 0] * A[..., 1] - Cc[..., 1] * A[..., 0], -dxy))
    return w1, w2, 1 - w1 - w2


def point_in_triangle(p, x0, x1, x2):                        # :115-118
    w1, w2, w3 = barycentric_coordinate(p, x0, x1, x2)
    return (w1 >= 0) & (w2 >= 0) & (w3 >= 0)


def _plane_or_edge(p, x0, x1, x2):
    """shared body of distance_function :120-135 / sdf_and_normal :142-158 -> (d, n) before any sign rule"""
    n = normalize(torch.linalg.cross(x1 - x0, x2 - x0))
    d = (n * (p - x0)).sum(-1)
    inside = point_in_triangle(p - d[..., None] * n, x0, x1, x2)
    d_e = torch.full_like(d, 1e6)
    n_e = n
    xs = (x0, x1, x2)
    for i in range(3):
        pt = closest_point_on_edge(p, xs[i], xs[(i + 1) % 3])
        dt_ = length(p - pt)
        better = dt_ < d_e
        n_e = torch.where(better[..., None], normalize(p - pt), n_e)
        d_e = torch.where(better, dt_, d_e)
    return torch.where(inside, d, d_e), torch.where(inside[..., None], n, n_e)


def distance_function(p, x0, x1, x2):                        # :120-140 (unsigned)
    d, _ = _plane_or_edge(p, x0, x1, x2)
    return d.abs()


def sdf_and_normal(p, penetrated, x0, x1, x2):               # :142-164
    d, n = _plane_or_edge(p, x0, x1, x2)
    flip = (penetrated == 0) == (d < 0)                      # :160
    return torch.where(flip, -d, d), torch.where(flip[..., None], -n, n)


def in_bounding_box(p, x0, x1, x2, threshold):               # :166-179
    lo = torch.minimum(x0, torch.minimum(x1, x2)) - threshold
    hi = torch.maximum(x0, torch.maximum(x1, x2)) + threshold
    return ((p > lo) & (p < hi)).all(-1)


def check_side(p, x0, x1, x2):                               # :189-196 (normal not normalised)
    n = torch.linalg.cross(x1 - x0, x2 - x0)
    return (n * (p - x0)).sum(-1) > 0


def _face_vertices(prim: ClothPrim, face_id):
    vid = prim.faces[face_id]                                # (M,3)
    return vid, prim.position[vid[:, 0]], prim.position[vid[:, 1]], prim.position[vid[:, 2]]


def _splat(prim: ClothPrim, vid, weights, c_f):              # :227-229 / :276-278
    ext = torch.zeros_like(prim.position)
    for i in range(3):
        ext = ext.index_add(0, vid[:, i], c_f * weights[i][:, None])
    return ext


def collide_mixed(prim: ClothPrim, p_pos, p_v, p_mass, dt, life, face_id, penetrated):   # :233-280
    """rows = the particles with a contact face.  Returns (new p_v, ext_f (V,3))."""
    vid, x0, x1, x2 = _face_vertices(prim, face_id)
    dist, D = sdf_and_normal(p_pos, penetrated, x0, x1, x2)
    active = dist <= 5e-3 * prim.mpm_scale                   # :236-237
    w1, w2, w3 = barycentric_coordinate(p_pos - D * dist[:, None], x0, x1, x2)
    cv = w1[:, None] * prim.velocity[vid[:, 0]] + w2[:, None] * prim.velocity[vid[:, 1]] + w3[:, None] * prim.velocity[vid[:, 2]]
    input_v = p_v - cv                                       # :247
    nc = (input_v * D).sum(-1)
    influence = torch.exp(torch.clamp(-dist * prim.softness, max=0.0))[:, None]   # min(exp(.), 1) without the overflow of a deep penetration
    if not prim.sticky:                                      # :250-262
        p_v_t = input_v - torch.clamp(nc, max=0.0)[:, None] * D
        nrm = length(p_v_t)
        fr = p_v_t / nrm[:, None] * torch.clamp(nrm + nc * prim.friction, min=0.0)[:, None]
        flag = ((nc < 0) & (torch.sqrt((p_v_t * p_v_t).sum(-1)) > 1e-30)).to(DT)[:, None]
        p_v_t = fr * flag + p_v_t * (1 - flag)
        hit = cv + p_v_t
        hit = torch.where((dist > 0)[:, None], cv + input_v * (1 - influence) + p_v_t * influence, hit)
        pv = torch.where((nc < 0)[:, None], hit, p_v)
    else:                                                    # :263-268
        pv = torch.where((dist > 0)[:, None], cv + input_v * (1 - influence), cv)
    pv = torch.where((dist < 0)[:, None], -(dist / dt)[:, None] * D * life, pv)        # :271-272
    c_f = p_mass * (p_v - pv) * (1.0 / dt) * prim.cloth_force_scale                       # :274
    c_f = torch.where(active[:, None], c_f, torch.zeros_like(c_f))
    return torch.where(active[:, None], pv, p_v), _splat(prim, vid, (w1, w2, w3), c_f)


def collide_particle(prim: ClothPrim, p_pos, p_v, dt, face_id, penetrated):              # :198-231
    """Returns (impulse rows, ext_f (V,3))."""
    vid, x0, x1, x2 = _face_vertices(prim, face_id)
    dist, D = sdf_and_normal(p_pos, penetrated, x0, x1, x2)
    c = dist - 5e-3 * prim.mpm_scale
    active = c < 0.0
    w1, w2, w3 = barycentric_coordinate(p_pos - D * dist[:, None], x0, x1, x2)
    cv = w1[:, None] * prim.velocity[vid[:, 0]] + w2[:, None] * prim.velocity[vid[:, 1]] + w3[:, None] * prim.velocity[vid[:, 2]]
    input_v = p_v - cv
    nc = (input_v * D).sum(-1)
    p_v_t = input_v - nc[:, None] * D
    f1 = -D * c[:, None] * 140.0                             # :217-218
    kf = prim.friction * 0.001
    nrm = torch.sqrt((p_v_t * p_v_t).sum(-1) + 1e-8)
    f2 = -p_v_t / nrm[:, None] * nc.abs()[:, None] * kf
    zero = torch.zeros_like(f1)
    p_f = torch.where(active[:, None], (f1 + f2) * 0.3, zero)
    c_f = torch.where(active[:, None], -(f1 + f2) * 0.01, zero)
    return p_f * dt, _splat(prim, vid, (w1, w2, w3), c_f)


# ----------------------------------------------------------------------------------------------
# contact faces and penetration flags (mpm_simulator.py:447-553): integer work, no gradient
# ----------------------------------------------------------------------------------------------
def get_contact_pair(x, cloth_pos, faces, penetration_prev, scale, chunk=4096):   # :447-461
    """-> contact_id (N,) int32: the face with the smallest distance among the faces whose padded bounding box holds the
    particle (all faces for a particle that was penetrated at the previous frame); -1 if none; first minimum wins"""
    x = torch.as_tensor(x, dtype=DT)
    cloth_pos = torch.as_tensor(cloth_pos, dtype=DT)
    faces = torch.as_tensor(np.asarray(faces), dtype=torch.int64)
    x0, x1, x2 = cloth_pos[faces[:, 0]], cloth_pos[faces[:, 1]], cloth_pos[faces[:, 2]]
    N = x.shape[0]
    out = np.full(N, -1, dtype=np.int32)
    pen = np.zeros(N, dtype=bool) if penetration_prev is None else np.asarray(penetration_prev).astype(bool)
    for s in range(0, N, chunk):
        p = x[s:s + chunk, None, :]
        cand = in_bounding_box(p, x0[None], x1[None], x2[None], 1e-2 * scale) | torch.as_tensor(pen[s:s + chunk])[:, None]
        d = distance_function(p.expand(-1, faces.shape[0], -1), x0[None].expand(p.shape[0], -1, -1),
                              x1[None].expand(p.shape[0], -1, -1), x2[None].expand(p.shape[0], -1, -1))
        d = torch.where(cand, d, torch.full_like(d, float("inf")))
        dmin, arg = d.min(dim=1)
        first = (d == dmin[:, None]).to(torch.int64).argmax(dim=1)          # first minimum (strict `<` in the sequential loop)
        ok = dmin < 1e10
        out[s:s + chunk] = torch.where(ok, first, torch.full_like(first, -1)).numpy().astype(np.int32)
    return out


def _trace(contact_cur, contact_prev, pen_start, side_cur_fn, side_prev_fn, nb, nbd):
    N = len(contact_cur)
    pen = np.array(pen_start, dtype=np.int8).copy()
    warnings = 0
    for i in range(N):
        fc, fp = int(contact_cur[i]), int(contact_prev[i])
        if fc == -1 or fp == -1:
            pen[i] = 0
            continue
        inverse, neighbouring = 0, False
        if fc != fp:
            hit = np.nonzero(nb[fc] == fp)[0]
            if len(hit):
                neighbouring, inverse = True, int(nbd[fc, hit[0]])
        else:
            neighbouring = True
        if neighbouring:
            if (side_cur_fn(i, fc) == side_prev_fn(i, fp)) == bool(inverse):
                pen[i] = 1 - pen[i]
        else:
            warnings += 1
    return pen, warnings


def _side(p, cloth_pos, faces, face):
    v = faces[face]
    return bool(check_side(torch.as_tensor(p, dtype=DT), *(torch.as_tensor(cloth_pos[int(q)], dtype=DT) for q in v)))


def trace_penetration_after_mpm(x_cur, x_prev, cloth_cur, cloth_prev, faces, contact_cur, contact_prev, pen_prev, nb, nbd):   # :484-510
    faces = np.asarray(faces)
    return _trace(contact_cur, contact_prev, pen_prev,
                  lambda i, fc: _side(x_cur[i], cloth_cur, faces, fc), lambda i, fp: _side(x_prev[i], cloth_prev, faces, fp), nb, nbd)


def trace_penetration_after_cloth(x_cur, cloth_cur, cloth_prev, faces, contact_cur, contact_before, pen_cur, nb, nbd):        # :520-545
    faces = np.asarray(faces)
    return _trace(contact_cur, contact_before, pen_cur,
                  lambda i, fc: _side(x_cur[i], cloth_cur, faces, fc), lambda i, fp: _side(x_cur[i], cloth_prev, faces, fp), nb, nbd)


# ----------------------------------------------------------------------------------------------
# the substep (mpm_simulator.py:325-337) with the cloth's changes
# ----------------------------------------------------------------------------------------------
def compute_von_mises(F, U, sig, V, yield_stress, mu):       # :172-188
    s = torch.clamp(torch.diagonal(sig, dim1=-2, dim2=-1), min=0.05)          # :175 (only the diagonal is read afterwards)
    eps = torch.log(s)
    eps_hat = eps - eps.sum(-1, keepdim=True) / 3
    nrm = torch.sqrt((eps_hat * eps_hat).sum(-1) + 1e-8)      # norm(), :200-202
    dg = nrm - yield_stress / (2 * mu)
    eps_new = eps - (dg / nrm)[:, None] * eps_hat
    F_new = U @ torch.diag_embed(torch.exp(eps_new)) @ V.transpose(-1, -2)
    return torch.where((dg > 0)[:, None, None], F_new, F)


def constitutive(F_tmp, U, sig, V, P: ClothSimParams):       # :226-252
    if P.material_model == O.MODEL_COROTATED and P.ptype == O.MAT_PLASTIC:
        eye = torch.eye(3, dtype=DT)
        J = torch.linalg.det(F_tmp)
        new_F = compute_von_mises(F_tmp, U, sig, V, P.yield_stress, P.mu)
        r = U @ V.transpose(-1, -2)
        return new_F, 2 * P.mu * (new_F - r) @ new_F.transpose(-1, -2) + eye * (P.lam * J * (J - 1))[:, None, None]
    return O.constitutive(F_tmp, U, sig, V, P)               # elastic / liquid / neo-Hookean: as in softmac


def p2g(x, v, C, F_tmp, U, sig, V, P: ClothSimParams, impulse):               # :204-269
    n = P.n_grid
    base, fx, w = O.bspline(x, P.inv_dx)
    new_F, stress = constitutive(F_tmp, U, sig, V, P)
    stress = (-P.dt * P.p_vol * 4 * P.inv_dx * P.inv_dx) * stress             # :254
    affine = stress + P.p_mass * C
    gv = torch.zeros(n * n * n, 3, dtype=DT)
    gm = torch.zeros(n * n * n, dtype=DT)
    for i in range(3):
        for j in range(3):
            for k in range(3):
                dpos = (torch.tensor([i, j, k], dtype=DT) - fx) * P.dx
                weight = w[i][:, 0] * w[j][:, 1] * w[k][:, 2]
                idx = O._flat(base, i, j, k, n)
                gv = gv.index_add(0, idx, weight[:, None] * (P.p_mass * v + (affine @ dpos[:, :, None])[:, :, 0] + impulse))
                gm = gm.index_add(0, idx, weight * P.p_mass)
    return new_F, gv.reshape(n, n, n, 3), gm.reshape(n, n, n)


def substep(x, v, C, F, P: ClothSimParams, prim: Optional[ClothPrim], contact_id, penetration, f: int = 0,
            control_idx=None, action=None):
    """One forward substep of soft_cloth's MPMSimulator.  contact_id / penetration: this frame's (N,) integer arrays.
    Returns (x', v', C', F', ext_f (V,3))."""
    N = x.shape[0]
    F_tmp = O.compute_F_tmp(C, F, P.dt)
    U = sig = V = None
    if P.material_model == O.MODEL_COROTATED:
        U, sig, V = O.svd3(F_tmp)
    sel = torch.as_tensor(np.nonzero(np.asarray(contact_id) >= 0)[0], dtype=torch.int64) if prim is not None else torch.zeros(0, dtype=torch.int64)
    face = torch.as_tensor(np.asarray(contact_id), dtype=torch.int64)[sel]
    pen = torch.as_tensor(np.asarray(penetration), dtype=torch.int64)[sel]
    ext = torch.zeros_like(prim.position) if prim is not None else None
    impulse = torch.zeros_like(x)
    if P.collision_type == O.CONTACT_PARTICLE and len(sel):                   # :208-213
        imp, e = collide_particle(prim, x[sel], v[sel], P.dt, face, pen)
        impulse = impulse.index_add(0, sel, imp)
        ext = ext + e
    if P.n_control > 0 and control_idx is not None:                           # :216-220
        on = control_idx >= 0
        a = action[control_idx.clamp(min=0)]
        impulse = impulse + torch.where(on[:, None], 6e-4 * a * P.dt, torch.zeros_like(a))
    new_F, grid_v_in, grid_m = p2g(x, v, C, F_tmp, U, sig, V, P, impulse)
    if P.collision_type == O.CONTACT_MIXED:                                   # :380-384
        grid_v_mixed = O.grid_op_mixed1(grid_m, grid_v_in, P)
        grid_v_out = grid_v_mixed
        v_tmp = O.grid_op_mixed2(x, grid_v_mixed, P)
        v_tgt = v_tmp
        if len(sel):                                                          # :419-428
            life = 1.0 / (P.substeps - f % P.substeps)
            vt, e = collide_mixed(prim, x[sel], v_tmp[sel], P.p_mass, P.dt, life, face, pen)
            v_tgt = v_tmp.index_put((sel,), vt)
            ext = ext + e
        grid_v_out = O.grid_op_mixed4(x, v_tmp, v_tgt, grid_m, grid_v_out, P)
    else:
        grid_v_out, _ = O.grid_op(grid_m, grid_v_in, P, ())
    new_x, new_v, new_C = O.g2p(x, grid_v_out, P)
    return new_x, new_v, new_C, new_F, ext


def substep_grad(x, v, C, F, P, prim, contact_id, penetration, f, gx1, gv1, gC1, gF1, ext_f_grad=None, control_idx=None, action=None):
    """adjoint of one substep: increments of frame f's x/v/C/F adjoints, of the cloth's position / velocity adjoints at
    frame f (`position.grad[f]`, `velocity.grad[f]`) and of the action"""
    leaves = [t.detach().clone().requires_grad_(True) for t in (x, v, C, F)]
    pl = None
    inputs = list(leaves)
    if prim is not None:
        cp, cvl = prim.position.detach().clone().requires_grad_(True), prim.velocity.detach().clone().requires_grad_(True)
        pl = dataclasses.replace(prim, position=cp, velocity=cvl)
        inputs += [cp, cvl]
    act = None
    if action is not None:
        act = action.detach().clone().requires_grad_(True)
        inputs.append(act)
    nx, nv, nC, nF, ext = substep(*leaves, P, pl, contact_id, penetration, f, control_idx, act)
    total = (nx * gx1).sum() + (nv * gv1).sum() + (nC * gC1).sum() + (nF * gF1).sum()
    if ext_f_grad is not None and ext is not None:
        total = total + (ext * torch.as_tensor(ext_f_grad, dtype=DT).reshape(ext.shape)).sum()
    grads = torch.autograd.grad(total, inputs, allow_unused=True)
    grads = [torch.zeros_like(i) if g is None else g for g, i in zip(grads, inputs)]
    out = dict(gx=grads[0], gv=grads[1], gC=grads[2], gF=grads[3], cloth_pos=None, cloth_vel=None, action=None)
    k = 4
    if prim is not None:
        out["cloth_pos"], out["cloth_vel"] = grads[4], grads[5]
        k = 6
    if act is not None:
        out["action"] = grads[k]
    return out
