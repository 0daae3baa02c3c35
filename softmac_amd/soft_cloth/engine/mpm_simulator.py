"""MPMSimulator of the soft <-> cloth variant - host-side mirror of the reference class
(soft_cloth/engine/mpm_simulator.py:14-784) over libsoftmac_hip.so.

What differs from softmac's simulator (engine/mpm_simulator.py): the constructor takes ONE cloth primitive and a length scale
(:16), the plastic material returns by von Mises (:172-188), walls only (:275-286), and the per-particle contact face /
penetration flag with their search and tracing kernels (:447-561).

Length scale.  The particle kernels work on the unit domain.  A simulation on [0, s)^3 with Young's modulus E, yield stress Y,
gravity g, control action a is the unit-domain simulation of x/s, v/s with E/s^2, Y/s^2, g/s, a/s^3 and a mass threshold of
1e-10/s^2 (C and F are dimensionless; every term of p2g :254-269 then scales by s^3, grid_op by s).  This class converts at its
boundary, so callers see the reference's physical quantities; the cloth primitive is physical on the device too."""
from __future__ import annotations

import ctypes as C
import types

import numpy as np

from ... import _ffi
from ...engine.mpm_simulator import CONTACT_MIXED, CONTACT_PARTICLE, MAT_PLASTIC, MODEL_COROTATED, MPMSimulator as _Base


class MPMSimulator(_Base):
    def __init__(self, cfg, primitive, env_dt=2e-3, scale=1., rigid_primitives=()):
        """rigid_primitives (this build's extension, BASELINE config C5 "mixed soft-rigid-cloth"): softmac Mesh primitives alongside the sheet - forecast
        contact only, scale 1 (their SDF tables are unit-domain objects); the contact chain takes them in index order, then the sheet."""
        s = self.scale = float(scale)
        if len(rigid_primitives) and (s != 1.0 or int(cfg.collision_type) != CONTACT_MIXED):
            raise ValueError("rigid primitives alongside the sheet need mpm_scale 1 and collision_type 2 (forecast contact)")
        unit = types.SimpleNamespace(**{k: getattr(cfg, k) for k in ("dim", "dtype", "quality", "yield_stress", "n_particles", "dt", "ptype",
                                                                     "material_model", "nu", "max_steps", "n_controllers", "collision_type")})
        for k in ("n_grid", "precision", "device", "grad_enabled", "sort_interval", "recompute_backward", "adjoint_frames", "slab_flags"):
            if getattr(cfg, k, None) is not None:
                setattr(unit, k, getattr(cfg, k))
        unit.E = cfg.E / (s * s)
        unit.gravity = tuple(float(g) / s for g in cfg.gravity)
        unit.ground_friction = 0.0                         # no floor rule in this variant (:275-286)
        super().__init__(unit, rigid_primitives, env_dt)
        self.ground_friction = getattr(cfg, "ground_friction", 0.0)
        self.default_gravity = cfg.gravity
        self._yield_stress = cfg.yield_stress
        self.dx, self.inv_dx = 1 / self.n_grid * s, float(self.n_grid) / s          # :31 (physical, as the reference reports them)
        self.p_vol = (self.dx * 0.5) ** 2
        self.p_mass = self.p_vol * self.p_rho
        self._mu, self._lam = self._mu * s * s, self._lam * s * s
        self.primitive = primitive
        self.n_triangles = primitive.num_faces
        self.n_vertices = primitive.num_vertices
        if self.material_model == MODEL_COROTATED and self.ptype == MAT_PLASTIC:       # :231-232
            self._h.call("smac_set_param", b"plasticity", C.c_double(1.0))
            self._h.call("smac_set_param", b"yield_ratio", C.c_double(float(cfg.yield_stress) / (2.0 * self._mu)))
        self._h.call("smac_set_param", b"mass_eps", C.c_double(1e-10 / (s * s)))        # :291 on the physical mass
        primitive._bind(self._h)

    # ------------------------------------------------------------------ unit <-> physical
    def _read_field(self, name, f, grad):
        a = super()._read_field(name, f, grad)
        if name in ("x", "v"):
            a *= (1.0 / self.scale) if grad else self.scale
        return a

    def substep(self, s, action=None):
        a = None if action is None else np.asarray(action, dtype=np.float64) / self.scale ** 3
        super().substep(s, a)

    def substep_grad(self, s, action=None, ext_f_grad=None, rigid_ext_f_grad=None):
        """ext_f_grad: the sheet's (V, 3) seed (:343-346); rigid_ext_f_grad: one 6-vector per rigid primitive of a mixed scene (softmac :342-344)"""
        if ext_f_grad is not None:                         # :343-346
            self.primitive.set_ext_f_grad(np.asarray(ext_f_grad, dtype=np.float64).reshape(-1, 3))
        a = None if action is None else np.asarray(action, dtype=np.float64) / self.scale ** 3
        g = super().substep_grad(s, a, rigid_ext_f_grad)
        return None if g is None else g / self.scale ** 3

    def set_action(self, action):
        super().set_action(np.asarray(action, dtype=np.float64) / self.scale ** 3)

    # ------------------------------------------------------------------ contact faces and penetration flags (:447-561)
    def get_contact_pair(self, f):
        self._h.call("smac_cloth_contact_pair", int(f))

    def backup_contact_pair(self, f):
        self._h.call("smac_cloth_backup_contact_pair", int(f))

    def trace_penetration_after_mpm(self, f):
        self._h.call("smac_cloth_trace_penetration", int(f), 0)

    def trace_penetration_after_cloth(self, f):
        self._h.call("smac_cloth_trace_penetration", int(f), 1)

    def check_penetration(self, f):
        total, warn = C.c_int32(0), C.c_int32(0)
        self._h.call("smac_cloth_check_penetration", int(f), C.byref(total), C.byref(warn))
        self.tracing_warnings = int(warn.value)            # particles whose previous face was not a listed neighbour (the reference prints, :509)
        return int(total.value)

    def get_contact(self, f):
        ids, pen = np.zeros(self.n_particles, dtype=np.int32), np.zeros(self.n_particles, dtype=np.int8)
        self._h.call("smac_cloth_get_contact", int(f), ids.ctypes.data_as(_ffi.c_int32_p), pen.ctypes.data_as(_ffi.c_int8_p))
        return ids, pen

    def set_contact(self, f, contact_id=None, penetration=None):
        ids = None if contact_id is None else np.ascontiguousarray(contact_id, dtype=np.int32).reshape(self.n_particles)
        pen = None if penetration is None else np.ascontiguousarray(penetration, dtype=np.int8).reshape(self.n_particles)
        self._h.call("smac_cloth_set_contact", int(f), None if ids is None else ids.ctypes.data_as(_ffi.c_int32_p),
                     None if pen is None else pen.ctypes.data_as(_ffi.c_int8_p))

    def get_penetration(self, f):                          # :721-724
        return self.get_contact(f)[1]

    # ------------------------------------------------------------------ IO (:566-724)
    def readframe(self, f, x, v, F, C, contact_id=None, penetration=None):
        super().readframe(f, x, v, F, C)
        x *= self.scale
        v *= self.scale
        if contact_id is not None or penetration is not None:
            ids, pen = self.get_contact(f)
            if contact_id is not None:
                contact_id[:, 0] = ids
            if penetration is not None:
                penetration[:, 0] = pen

    def get_state(self, f):                                # :604-615: (N, 26) = x3 v3 F9 C9 contact_id penetration
        st = super().get_state(f)
        st[:, 0:6] *= self.scale
        ids, pen = self.get_contact(f)
        return np.hstack([st, ids.astype(np.float64)[:, None], pen.astype(np.float64)[:, None]])

    def set_state(self, f, state):                         # :617-618
        x, v, F, Cm = state[:4]
        super().set_state(f, (np.asarray(x, dtype=np.float64) / self.scale, np.asarray(v, dtype=np.float64) / self.scale, F, Cm))

    def reset(self, x):                                    # :645-650
        x = np.asarray(x, dtype=np.float64)
        N = self.n_particles
        if x.shape[1] == self.dim:                         # reset_kernel :620-629
            super().reset(x / self.scale)
            self.set_contact(0, None, np.zeros(N, dtype=np.int8))
        else:                                              # reset_all_kernel :631-643 (26 columns; 24 = the state without the contact columns)
            st = x[:, :24].copy()
            st[:, 0:6] /= self.scale
            super().reset(st)
            if x.shape[1] >= 26:
                self.set_contact(0, x[:, 24].astype(np.int32), x[:, 25].astype(np.int8))
            else:
                self.set_contact(0, np.full(N, -1, dtype=np.int32), np.zeros(N, dtype=np.int8))
        self.cur = 0

    def set_x(self, f, x):
        super().set_x(f, np.asarray(x, dtype=np.float64) / self.scale)

    def set_v(self, f, v):
        super().set_v(f, np.asarray(v, dtype=np.float64) / self.scale)

    def get_grad(self, f):                                 # :710-714
        gx, gv = super().get_grad(f)
        return gx / self.scale, gv / self.scale

    def get_grad_full(self, f):
        gx, gv, gF, gC = super().get_grad_full(f)
        return gx / self.scale, gv / self.scale, gF, gC

    def add_grad(self, f, gx=None, gv=None, gF=None, gC=None):
        s = self.scale
        super().add_grad(f, None if gx is None else np.asarray(gx, dtype=np.float64) * s, None if gv is None else np.asarray(gv, dtype=np.float64) * s, gF, gC)

    def set_x_grad(self, f, x_grad):                       # :658-659 (an assignment in the reference)
        cur, _ = self.get_grad(f)
        self.add_grad(f, gx=np.asarray(x_grad, dtype=np.float64).reshape(cur.shape) - cur)

    def compute_grid_m_kernel(self, f):
        return super().compute_grid_m_kernel(f) * self.scale ** 2

    def get_observation(self, f):                          # :780-784
        step = self.n_particles // 200
        x, v = self.get_x(f), self.get_v(f)
        return np.hstack([x[::step][:200], v[::step][:200]]).reshape(-1)
