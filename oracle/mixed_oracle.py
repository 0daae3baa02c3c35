"""TEST INFRASTRUCTURE - CPU restatement (torch float64 + autograd) of the MIXED soft-rigid-cloth substep of BASELINE config C5: one particle cloud of
two materials in forecast contact with rigid SDF primitives AND one triangle-mesh sheet.  Only tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg may import this; the product (softmac_amd/) never does.

PARITY UNPINNED, and by construction: the reference has NO simulator that holds both kinds of primitive.  softmac's grid_op_mixed3 loops its rigid
primitives (/root/reference/softmac/engine/mpm_simulator.py:421-429 -> primitive_base.py:139-181), soft_cloth's has the one sheet
(/root/reference/soft_cloth/engine/mpm_simulator.py:419-428 -> primitive_cloth.py:233-280).  This file COMPOSES the two restatements that exist
(oracle/softmac_oracle.py, oracle/cloth_oracle.py - each follows its reference file line by line) in one mixed3 pass, in a stated order:

    v_tgt = v_tmp;  for each rigid primitive in index order: v_tgt = collide_mixed(prim, x, v_tgt);  then, for the particles that hold a contact
    face: v_tgt = sheet.collide_mixed(x, v_tgt)

and takes the per-particle Lame parameters / yield stress the reference's fields allow (mpm_simulator.py:47-49, soft_cloth :47-50; filled uniformly
there, :86-90) from a two-entry table selected by `mat_id`.  Everything else (compute_F_tmp, svd, p2g, grid_op_mixed1/2/4, g2p, the von-Mises return
map of soft_cloth :172-188, walls only :275-286) is the cloth oracle's substep unchanged.  Scenes live on the unit domain (mpm_scale = 1): the rigid
primitives' SDF tables are unit-domain objects."""
from __future__ import annotations

import dataclasses
from typing import Optional, Sequence

import numpy as np
import torch

from . import cloth_oracle as CL
from . import softmac_oracle as O

DT = O.DT


def constitutive_two(F_tmp, U, sig, V, P: CL.ClothSimParams, P2: Optional[CL.ClothSimParams], mat_id):
    """new_F and stress per particle: entry 0 (P) or entry 1 (P2) of the material table"""
    nF, st = CL.constitutive(F_tmp, U, sig, V, P)
    if P2 is None or mat_id is None:
        return nF, st
    nF2, st2 = CL.constitutive(F_tmp, U, sig, V, P2)
    m = torch.as_tensor(np.asarray(mat_id) != 0)[:, None, None]
    return torch.where(m, nF2, nF), torch.where(m, st2, st)


def p2g(x, v, C, new_F, stress, P: CL.ClothSimParams):       # soft_cloth :254-269 with the constitutive update done by the caller
    n = P.n_grid
    base, fx, w = O.bspline(x, P.inv_dx)
    stress = (-P.dt * P.p_vol * 4 * P.inv_dx * P.inv_dx) * stress
    affine = stress + P.p_mass * C
    gv = torch.zeros(n * n * n, 3, dtype=DT)
    gm = torch.zeros(n * n * n, dtype=DT)
    for i in range(3):
        for j in range(3):
            for k in range(3):
                dpos = (torch.tensor([i, j, k], dtype=DT) - fx) * P.dx
                weight = w[i][:, 0] * w[j][:, 1] * w[k][:, 2]
                idx = O._flat(base, i, j, k, n)
                gv = gv.index_add(0, idx, weight[:, None] * (P.p_mass * v + (affine @ dpos[:, :, None])[:, :, 0]))
                gm = gm.index_add(0, idx, weight * P.p_mass)
    return gv.reshape(n, n, n, 3), gm.reshape(n, n, n)


def substep(x, v, C, F, P: CL.ClothSimParams, prims: Sequence[O.RigidPrim], sheet: Optional[CL.ClothPrim], contact_id, penetration, f: int = 0,
            P2: Optional[CL.ClothSimParams] = None, mat_id=None):
    """Returns (x', v', C', F', ext_f of the rigid primitives [6 each], ext_f of the sheet (V, 3))."""
    assert P.scale == 1.0 and P.collision_type == O.CONTACT_MIXED
    F_tmp = O.compute_F_tmp(C, F, P.dt)
    U, sig, V = O.svd3(F_tmp)
    new_F, stress = constitutive_two(F_tmp, U, sig, V, P, P2, mat_id)
    grid_v_in, grid_m = p2g(x, v, C, new_F, stress, P)
    grid_v_mixed = O.grid_op_mixed1(grid_m, grid_v_in, P)
    v_tmp = O.grid_op_mixed2(x, grid_v_mixed, P)
    life = 1.0 / (P.substeps - f % P.substeps)                # :425
    v_tgt = v_tmp
    ext_r = []
    for pr in prims:                                          # softmac :426-429: the rigid primitives in index order
        if pr.contact:
            v_tgt, e = O.collide_mixed(pr, x, v_tgt, P.p_mass, P.dt, life)
        else:
            e = torch.zeros(6, dtype=DT)
        ext_r.append(e)
    ext_c = None
    if sheet is not None:                                     # soft_cloth :419-428: then the sheet, on the velocity the primitives left
        ext_c = torch.zeros_like(sheet.position)
        sel = torch.as_tensor(np.nonzero(np.asarray(contact_id) >= 0)[0], dtype=torch.int64)
        if len(sel):
            face = torch.as_tensor(np.asarray(contact_id), dtype=torch.int64)[sel]
            pen = torch.as_tensor(np.asarray(penetration), dtype=torch.int64)[sel]
            vt, e = CL.collide_mixed(sheet, x[sel], v_tgt[sel], P.p_mass, P.dt, life, face, pen)
            v_tgt = v_tgt.index_put((sel,), vt)
            ext_c = ext_c + e
    grid_v_out = O.grid_op_mixed4(x, v_tmp, v_tgt, grid_m, grid_v_mixed, P)
    new_x, new_v, new_C = O.g2p(x, grid_v_out, P)
    return new_x, new_v, new_C, new_F, ext_r, ext_c


def substep_grad(x, v, C, F, P, prims, sheet, contact_id, penetration, f, gx1, gv1, gC1, gF1, ext_r_grad=None, ext_c_grad=None, P2=None, mat_id=None):
    """adjoint of one substep by autograd: increments of frame f's particle adjoints, of every rigid primitive's (position, rotation, v, w) adjoints
    and of the sheet's vertex position / velocity adjoints"""
    leaves = [t.detach().clone().requires_grad_(True) for t in (x, v, C, F)]
    inputs = list(leaves)
    prims_l = []
    for pr in prims:
        st = [t.detach().clone().requires_grad_(True) for t in (pr.position, pr.rotation, pr.v, pr.w)]
        prims_l.append(dataclasses.replace(pr, position=st[0], rotation=st[1], v=st[2], w=st[3]))
        inputs += st
    sl = None
    if sheet is not None:
        cp, cv = sheet.position.detach().clone().requires_grad_(True), sheet.velocity.detach().clone().requires_grad_(True)
        sl = dataclasses.replace(sheet, position=cp, velocity=cv)
        inputs += [cp, cv]
    nx, nv, nC, nF, ext_r, ext_c = substep(*leaves, P, prims_l, sl, contact_id, penetration, f, P2, mat_id)
    total = (nx * gx1).sum() + (nv * gv1).sum() + (nC * gC1).sum() + (nF * gF1).sum()
    if ext_r_grad is not None:
        for e, g in zip(ext_r, ext_r_grad):
            total = total + (e * torch.as_tensor(g, dtype=DT)).sum()
    if ext_c_grad is not None and ext_c is not None:
        total = total + (ext_c * torch.as_tensor(ext_c_grad, dtype=DT).reshape(ext_c.shape)).sum()
    grads = torch.autograd.grad(total, inputs, allow_unused=True)
    grads = [torch.zeros_like(i) if g is None else g for g, i in zip(grads, inputs)]
    out = dict(gx=grads[0], gv=grads[1], gC=grads[2], gF=grads[3], prims=[], sheet_pos=None, sheet_vel=None)
    k = 4
    for _ in prims:
        out["prims"].append(torch.cat([g.reshape(-1) for g in grads[k:k + 4]]))        # 13 = pos3 quat4 v3 w3
        k += 4
    if sheet is not None:
        out["sheet_pos"], out["sheet_vel"] = grads[k], grads[k + 1]
    return out
