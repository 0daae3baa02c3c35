"""Engines for the slab-decomposition tests.

OracleSlabEngine: a stand-in for the HIP engine built from the oracle's kernels on dense torch grids, with the
same phase / halo interface as softmac_amd.parallel.HipSlabEngine.  It lets the world_size-2 gloo test exercise
SlabRunner's exchange logic on the CPU (test infrastructure only)."""
import numpy as np
import torch

import helpers as H
from helpers import O


class OracleSlabEngine:
    def __init__(self, P, state24, prim_specs=(), prim_states=None):
        self.P, self.n = P, P.n_grid
        self.frames = [O.state24_split(state24)]
        self.specs, self.pstates = list(prim_specs), prim_states
        n = self.n
        z = lambda: torch.zeros(n, n, n, 4, dtype=O.DT)
        self.fields = {k: z() for k in ("grid_in", "grid_mixed", "grid_out", "grid_in.grad", "grid_mixed.grad", "grid_out.grad")}
        self.ck = {}
        self.adj = {}
        self.ext = [torch.zeros(6, dtype=O.DT) for _ in self.specs]
        self.pgrad = {}
        self.newF = None
        self.shift = 0                                # migration: duplicate frames before the current segment

    # ---- helpers
    def prims_at(self, f, leaves=False):
        out, lv = [], []
        for i, s in enumerate(self.specs):
            st = torch.as_tensor(np.asarray(self.pstates[f][i]), dtype=O.DT)
            parts = [st[:3].clone(), st[3:7].clone(), st[7:10].clone(), st[10:13].clone()]
            if leaves:
                parts = [p.requires_grad_(True) for p in parts]
            lv.append(parts)
            out.append(O.RigidPrim(*parts, torch.as_tensor(s["sdf"], dtype=O.DT), torch.as_tensor(s["normal"], dtype=O.DT),
                                   torch.as_tensor(s["lower"], dtype=O.DT), torch.as_tensor(s["upper"], dtype=O.DT), float(s["dx"]),
                                   float(s.get("friction", 0.9)), float(s.get("softness", 666.0)), bool(s.get("contact", True))))
        return out, lv

    def get_adj(self, f, n=None):
        if f not in self.adj:
            N = n if n is not None else self.frames[f][0].shape[0]
            self.adj[f] = [torch.zeros(N, 3, dtype=O.DT), torch.zeros(N, 3, dtype=O.DT), torch.zeros(N, 3, 3, dtype=O.DT),
                           torch.zeros(N, 3, 3, dtype=O.DT)]
        return self.adj[f]

    def new_buffer(self, nplanes):
        return torch.zeros(nplanes, self.n, self.n, 4, dtype=O.DT)

    def halo_pack(self, field, plane0, nplanes, buf, minus_mixed=0):
        v = self.fields[field][plane0:plane0 + nplanes]
        if minus_mixed:
            v = v - self.fields["grid_mixed"][plane0:plane0 + nplanes]
        buf.copy_(v)

    def halo_unpack_add(self, field, plane0, nplanes, buf):
        self.fields[field][plane0:plane0 + nplanes] += buf

    # ---- migration interface (rows = x3 v3 F9 C9 / gx3 gv3 gF9 gC9), same as HipSlabEngine
    def set_segment(self, n_live, shift):
        self.shift = int(shift)

    def get_state(self, f):
        x, v, C, F = self.frames[f]
        n = x.shape[0]
        return np.hstack([x.numpy(), v.numpy(), F.reshape(n, 9).numpy(), C.reshape(n, 9).numpy()])

    def set_state(self, f, st):
        fr = O.state24_split(st)
        while len(self.frames) <= f:
            self.frames.append(None)
        self.frames[f] = fr

    def get_grad_rows(self, f):
        gx, gv, gC, gF = self.get_adj(f)
        n = gx.shape[0]
        return np.hstack([gx.numpy(), gv.numpy(), gF.reshape(n, 9).numpy(), gC.reshape(n, 9).numpy()])

    def add_grad_rows(self, f, g):
        n = len(g)
        self.adj.pop(f, None) if (f in self.adj and self.adj[f][0].shape[0] != n) else None
        self.frames_n = n
        a = self.get_adj(f, n)
        t = lambda b, shp: torch.as_tensor(np.ascontiguousarray(b), dtype=O.DT).reshape(shp)
        a[0] = a[0] + t(g[:, 0:3], (n, 3)); a[1] = a[1] + t(g[:, 3:6], (n, 3))
        a[3] = a[3] + t(g[:, 6:15], (n, 3, 3)); a[2] = a[2] + t(g[:, 15:24], (n, 3, 3))

    def _contact(self, x, vmix, gm, prims, f):
        """correction added to grid_v_out by mixed2-4, and the per-primitive wrench"""
        f = f - self.shift                            # the substep phase of `life` follows the physical substep (smac_set_segment)
        v_tmp = O.grid_op_mixed2(x, vmix, self.P)
        v_tgt, ext = O.grid_op_mixed3(x, v_tmp, prims, self.P, f)
        corr = O.grid_op_mixed4(x, v_tmp, v_tgt, gm, torch.zeros_like(vmix), self.P)
        return corr, ext

    # ---- forward phases
    def phase(self, f, k):
        P = self.P
        x, v, C, F = self.frames[f]
        if k == 0:
            Ft = O.compute_F_tmp(C, F, P.dt)
            U, sig, V = O.svd3(Ft) if P.material_model == 0 else (None, None, None)
            self.newF, gv, gm, _ = O.p2g(x, v, C, Ft, U, sig, V, P)
            self.fields["grid_in"] = torch.cat([gm[..., None], gv], -1)
        elif k == 1:
            gin = self.fields["grid_in"]
            vmix = O.grid_op_mixed1(gin[..., 0], gin[..., 1:], P)
            vout = vmix
            prims, _ = self.prims_at(f)
            if any(p.contact for p in prims) and x.shape[0] > 0:
                corr, ext = self._contact(x, vmix, gin[..., 0], prims, f)
                vout = vmix + corr
                self.ext = [a + b for a, b in zip(self.ext, ext)]
            pad = torch.zeros(self.n, self.n, self.n, 1, dtype=O.DT)
            self.fields["grid_mixed"] = torch.cat([vmix, pad], -1)
            self.fields["grid_out"] = torch.cat([vout, pad], -1)
        else:
            self.ck[f] = {k_: self.fields[k_].clone() for k_ in ("grid_in", "grid_mixed", "grid_out")}
            nx, nv, nC = O.g2p(x, self.fields["grid_out"][..., :3], P)
            nxt = (nx, nv, nC, self.newF)
            if len(self.frames) > f + 1:
                self.frames[f + 1] = nxt
            else:
                self.frames.append(nxt)

    # ---- backward phases (autograd per piece)
    def grad_phase(self, f, k, ext_f_grad=None):
        P = self.P
        x, v, C, F = self.frames[f]
        a1 = self.get_adj(f + 1)
        a0 = self.get_adj(f)
        if k == 0:
            for k_, val in self.ck[f].items():
                self.fields[k_] = val.clone()
            for k_ in ("grid_in.grad", "grid_mixed.grad", "grid_out.grad"):
                self.fields[k_] = torch.zeros(self.n, self.n, self.n, 4, dtype=O.DT)
            self._eg = ext_f_grad
            xl = x.clone().requires_grad_(True)
            vo = self.fields["grid_out"][..., :3].clone().requires_grad_(True)
            nx, nv, nC = O.g2p(xl, vo, P)
            L = (nx * a1[0]).sum() + (nv * a1[1]).sum() + (nC * a1[2]).sum()
            gx, gvo = torch.autograd.grad(L, [xl, vo])
            a0[0] = a0[0] + gx
            self.fields["grid_out.grad"][..., :3] = gvo
        elif k == 1:
            prims, leaves = self.prims_at(f, leaves=True)
            if any(p.contact for p in prims) and x.shape[0] > 0:
                xl = x.clone().requires_grad_(True)
                vm = self.fields["grid_mixed"][..., :3].clone().requires_grad_(True)
                corr, ext = self._contact(xl, vm, self.fields["grid_in"][..., 0], prims, f)
                L = (corr * self.fields["grid_out.grad"][..., :3]).sum()
                if self._eg is not None:
                    for e, g in zip(ext, self._eg):
                        L = L + (e * torch.as_tensor(g, dtype=O.DT)).sum()
                inputs = [xl, vm] + [t for lv in leaves for t in lv]
                gr = torch.autograd.grad(L, inputs, allow_unused=True)
                gr = [torch.zeros_like(i) if g is None else g for g, i in zip(gr, inputs)]
                a0[0] = a0[0] + gr[0]
                self.fields["grid_mixed.grad"][..., :3] = gr[1]
                for i in range(len(prims)):
                    self.pgrad.setdefault(f, [np.zeros(13) for _ in prims])
                    self.pgrad[f][i] += torch.cat(gr[2 + 4 * i: 6 + 4 * i]).numpy()
        else:
            gin = self.fields["grid_in"]
            gm = gin[..., 0].clone().requires_grad_(True)
            gv = gin[..., 1:].clone().requires_grad_(True)
            vmix = O.grid_op_mixed1(gm, gv, P)
            up = self.fields["grid_out.grad"][..., :3] + self.fields["grid_mixed.grad"][..., :3]
            agm, agv = torch.autograd.grad((vmix * up).sum(), [gm, gv], allow_unused=True)
            agm = torch.zeros_like(gm) if agm is None else agm
            leaves = [t.clone().requires_grad_(True) for t in (x, v, C, F)]
            Ft = O.compute_F_tmp(leaves[2], leaves[3], P.dt)
            U, sig, V = O.svd3(Ft) if P.material_model == 0 else (None, None, None)
            nF, pgv, pgm, _ = O.p2g(leaves[0], leaves[1], leaves[2], Ft, U, sig, V, P)
            L = (nF * a1[3]).sum() + (pgv * agv).sum() + (pgm * agm).sum()
            gr = torch.autograd.grad(L, leaves)
            for i in range(4):
                a0[i] = a0[i] + gr[i]


class ClothOracleSlabEngine(OracleSlabEngine):
    """The soft <-> cloth substep (oracle/cloth_oracle.py) cut into the same three phases: the sheet's forecast contact takes the place of the SDF
    primitives' (its v_out corrections and grid_v_mixed.grad partials cross the slab boundary the same way), contact faces and penetration flags are
    per-particle inputs of each frame (rank-local).  Test infrastructure for tests/test_slabs.py (world-2 gloo, CPU)."""

    def __init__(self, P, state24, cloth_frames, faces, prim_kw, contact):
        """cloth_frames[f] = (pos (V,3), vel (V,3)); contact[f] = (contact_id (N,), penetration (N,)) of THIS rank's particles"""
        super().__init__(P, state24)
        from oracle import cloth_oracle as CO
        self.CO, self.cloth, self.faces, self.kw, self.contact = CO, cloth_frames, faces, prim_kw, contact
        V = np.asarray(cloth_frames[0][0]).shape[0]
        self.ext = torch.zeros(V, 3, dtype=O.DT)                 # per-vertex force (this rank's partial sum)
        self.cgrad = {}                                          # frame -> (position.grad, velocity.grad) partial sums
        self.ext_f_grad = None                                   # (V,3) seed on the sheet's force

    def _prim(self, f, leaves=False):
        pos = torch.as_tensor(np.asarray(self.cloth[f][0]), dtype=O.DT).clone()
        vel = torch.as_tensor(np.asarray(self.cloth[f][1]), dtype=O.DT).clone()
        if leaves:
            pos.requires_grad_(True); vel.requires_grad_(True)
        return self.CO.ClothPrim(pos, vel, torch.as_tensor(np.asarray(self.faces).astype(np.int64)), mpm_scale=self.P.scale, **self.kw)

    def _cloth_contact(self, x, vmix, gm, prim, f):
        CO, P = self.CO, self.P
        cid, pen = self.contact[f]
        sel = torch.as_tensor(np.nonzero(np.asarray(cid) >= 0)[0], dtype=torch.int64)
        v_tmp = O.grid_op_mixed2(x, vmix, P)
        v_tgt, ext = v_tmp, torch.zeros_like(prim.position)
        if len(sel):
            life = 1.0 / (P.substeps - (f - self.shift) % P.substeps)
            vt, ext = CO.collide_mixed(prim, x[sel], v_tmp[sel], P.p_mass, P.dt, life, torch.as_tensor(np.asarray(cid), dtype=torch.int64)[sel],
                                       torch.as_tensor(np.asarray(pen), dtype=torch.int64)[sel])
            v_tgt = v_tmp.index_put((sel,), vt)
        corr = O.grid_op_mixed4(x, v_tmp, v_tgt, gm, torch.zeros_like(vmix), P)
        return corr, ext

    def _p2g(self, x, v, C, F):
        CO, P = self.CO, self.P
        Ft = O.compute_F_tmp(C, F, P.dt)
        U, sig, V = O.svd3(Ft) if P.material_model == 0 else (None, None, None)
        return CO.p2g(x, v, C, Ft, U, sig, V, P, torch.zeros_like(x))

    def phase(self, f, k):
        P = self.P
        x, v, C, F = self.frames[f]
        if k == 0:
            self.newF, gv, gm = self._p2g(x, v, C, F)
            self.fields["grid_in"] = torch.cat([gm[..., None], gv], -1)
        elif k == 1:
            gin = self.fields["grid_in"]
            vmix = O.grid_op_mixed1(gin[..., 0], gin[..., 1:], P)
            corr, ext = self._cloth_contact(x, vmix, gin[..., 0], self._prim(f), f)
            self.ext = self.ext + ext.detach()
            pad = torch.zeros(self.n, self.n, self.n, 1, dtype=O.DT)
            self.fields["grid_mixed"] = torch.cat([vmix, pad], -1)
            self.fields["grid_out"] = torch.cat([vmix + corr, pad], -1)
        else:
            super().phase(f, 2)

    def grad_phase(self, f, k, ext_f_grad=None):
        P = self.P
        x, v, C, F = self.frames[f]
        a1, a0 = self.get_adj(f + 1), self.get_adj(f)
        if k == 0:
            super().grad_phase(f, 0, None)
        elif k == 1:
            prim = self._prim(f, leaves=True)
            xl = x.clone().requires_grad_(True)
            vm = self.fields["grid_mixed"][..., :3].clone().requires_grad_(True)
            corr, ext = self._cloth_contact(xl, vm, self.fields["grid_in"][..., 0], prim, f)
            L = (corr * self.fields["grid_out.grad"][..., :3]).sum()
            if self.ext_f_grad is not None:
                L = L + (ext * torch.as_tensor(self.ext_f_grad, dtype=O.DT)).sum()
            inputs = [xl, vm, prim.position, prim.velocity]
            gr = torch.autograd.grad(L, inputs, allow_unused=True)
            gr = [torch.zeros_like(i) if g is None else g for g, i in zip(gr, inputs)]
            a0[0] = a0[0] + gr[0]
            self.fields["grid_mixed.grad"][..., :3] = gr[1]
            self.cgrad[f] = (gr[2].numpy(), gr[3].numpy())
        else:
            gin = self.fields["grid_in"]
            gm = gin[..., 0].clone().requires_grad_(True)
            gv = gin[..., 1:].clone().requires_grad_(True)
            vmix = O.grid_op_mixed1(gm, gv, P)
            up = self.fields["grid_out.grad"][..., :3] + self.fields["grid_mixed.grad"][..., :3]
            agm, agv = torch.autograd.grad((vmix * up).sum(), [gm, gv], allow_unused=True)
            agm = torch.zeros_like(gm) if agm is None else agm
            leaves = [t.clone().requires_grad_(True) for t in (x, v, C, F)]
            nF, pgv, pgm = self._p2g(*leaves)
            L = (nF * a1[3]).sum() + (pgv * agv).sum() + (pgm * agm).sum()
            gr = torch.autograd.grad(L, leaves)
            for i in range(4):
                a0[i] = a0[i] + gr[i]
