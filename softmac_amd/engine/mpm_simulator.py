"""MPMSimulator - host-side mirror of the reference class
(/root/reference/softmac/engine/mpm_simulator.py:16-618) over libsoftmac_hip.so.

Same constructor, attributes and method names as the reference so that TaichiEnv, the losses
and the demos can drive it; every method forwards to one C-ABI entry point (include/softmac_hip.h).
The public dtype stays float64 numpy (reference :482-485); the device arithmetic type is chosen
by `cfg.precision` ("float32" default / "float64") - see DESIGN.md "precision".
"""
from __future__ import annotations

import ctypes as C
import weakref

import numpy as np

from .. import _ffi

MODEL_COROTATED = 0
MODEL_NEOHOOKEAN = 1

MAT_PLASTIC = 0
MAT_ELASTIC = 1
MAT_LIQUID = 2

CONTACT_GRID = 0
CONTACT_PARTICLE = 1
CONTACT_MIXED = 2


class _FrameField:
    """`sim.x[f]`-style read access to one trajectory field (+ `.grad`), as the losses use it
    (reference loss_pour.py:12).  Indexing returns numpy copies; the data lives on the GPU."""

    def __init__(self, sim, name, grad=False):
        self._sim, self._name, self._grad = sim, name, grad
        if not grad:
            self.grad = _FrameField(sim, name, grad=True)

    def __getitem__(self, idx):
        f, rest = (idx, None) if not isinstance(idx, tuple) else (idx[0], idx[1:])
        arr = self._sim._read_field(self._name, int(f), self._grad)
        return arr if rest is None else arr[rest if len(rest) > 1 else rest[0]]

    def to_numpy(self, f):
        return self[f]


LIVE_SIMULATORS = weakref.WeakSet()      # what compat/taichi's `ti.ad.clear_all_gradients()` reaches


class MPMSimulator:
    def __init__(self, cfg, primitives=(), env_dt=2e-3, rigid_velocity_control=False):
        dim = self.dim = cfg.dim
        assert dim == 3, "the MI355X path implements the 3-D simulator only"
        assert cfg.dtype == 'float64'                       # reference :19 (public dtype)
        self.dtype = np.float64
        self._yield_stress = cfg.yield_stress
        self.ground_friction = cfg.ground_friction
        self.default_gravity = cfg.gravity
        self.n_primitive = len(primitives)

        quality = cfg.quality * 0.5                         # :26-30
        n_particles = self.n_particles = int(cfg.n_particles)
        self.capacity = n_particles                         # slab migration changes n_particles within this capacity (set_segment)
        n_grid = self.n_grid = int(128 * quality)
        if getattr(cfg, "n_grid", None):                    # extension: explicit grid size (128^3 / 256^3 runs)
            n_grid = self.n_grid = int(cfg.n_grid)

        self.dx, self.inv_dx = 1 / n_grid, float(n_grid)
        self.dt = cfg.dt
        self.p_vol, self.p_rho = (self.dx * 0.5) ** 2, 1    # :34
        self.p_mass = self.p_vol * self.p_rho

        self.ptype = cfg.ptype
        self.material_model = cfg.material_model
        E, nu = cfg.E, cfg.nu
        self._mu, self._lam = E / (2 * (1 + nu)), E * nu / ((1 + nu) * (1 - 2 * nu))   # :41
        if self.ptype == 1:
            self._mu, self._lam = 0.3 * self._mu, 0.3 * self._lam
        elif self.ptype == 2:
            self._mu = 0.0

        self.max_steps = int(cfg.max_steps)
        self.substeps = int(env_dt / self.dt)               # :52
        self.res = (n_grid, n_grid, n_grid)
        self.primitives = primitives
        self.primitives_contact = [True for _ in range(self.n_primitive)]   # :70, mutated by demos
        self.rigid_velocity_control = rigid_velocity_control
        self.n_control = int(cfg.n_controllers)
        self.collision_type = int(cfg.collision_type)
        self.cur = 0

        precision = getattr(cfg, "precision", "float32")
        self.precision = 64 if str(precision) in ("float64", "64", "f64") else 32
        self.device = int(getattr(cfg, "device", 0))
        self.grad_enabled = bool(getattr(cfg, "grad_enabled", True))

        c = _ffi.SmacConfig()
        c.abi_version = _ffi.ABI_VERSION
        c.precision = self.precision
        c.device = self.device
        c.n_particles = n_particles
        c.n_grid = n_grid
        c.max_frames = self.max_steps
        c.grad_enabled = 1 if self.grad_enabled else 0
        c.substeps = max(self.substeps, 1)
        c.ptype = self.ptype
        c.material_model = self.material_model
        c.collision_type = self.collision_type
        c.n_control = self.n_control
        c.n_primitives = self.n_primitive
        c.rigid_velocity_control = 1 if rigid_velocity_control else 0
        c.sort_interval = int(getattr(cfg, "sort_interval", 0))
        c.flags = (1 if getattr(cfg, "recompute_backward", False) else 0) | int(getattr(cfg, "slab_flags", 0))
        c.adjoint_frames = int(getattr(cfg, "adjoint_frames", 0))            # 0: one per state frame; k >= 3: rolling (long episodes)
        c.dt = self.dt
        c.mu, c.lam = self._mu, self._lam
        c.p_vol, c.p_mass = self.p_vol, self.p_mass
        g = tuple(cfg.gravity)
        c.gravity[0], c.gravity[1], c.gravity[2] = float(g[0]), float(g[1]), float(g[2])
        c.ground_friction = float(self.ground_friction)
        c.yield_stress = float(self._yield_stress)
        self._h = _ffi.Handle(c)                            # raises without a GPU: no CPU fallback
        LIVE_SIMULATORS.add(self)

        for i, p in enumerate(primitives):                  # device-side primitive slots
            p._bind(self._h, i)
        self._contact_pushed = None

        self.x = _FrameField(self, "x")
        self.v = _FrameField(self, "v")
        self.C = _FrameField(self, "C")
        self.F = _FrameField(self, "F")

    # ------------------------------------------------------------------ plumbing
    def initialize(self):                                   # :86-90 (gravity/mu/lam already in the handle)
        pass

    def _push_contact_flags(self):
        flags = tuple(bool(b) for b in self.primitives_contact)
        if flags != self._contact_pushed:
            for i, p in enumerate(self.primitives):
                p._push_params(contact=flags[i])
            self._contact_pushed = flags

    def _read_field(self, name, f, grad):
        N = self.n_particles
        shape = (N, 3) if name in ("x", "v") else (N, 3, 3)
        out = np.zeros(shape, dtype=np.float64)
        args = {"x": 0, "v": 1, "F": 2, "C": 3}[name]
        ptrs = [None, None, None, None]
        ptrs[args] = _ffi.dptr(out)
        self._h.call("smac_get_grad" if grad else "smac_get_frame", f, *ptrs)
        return out

    def sync(self):
        self._h.call("smac_sync")

    def set_segment(self, n_live, frame_shift=0):
        """Slab migration (softmac_amd/parallel.py): from now on frames hold `n_live` <= capacity particles; `frame_shift` duplicate
        frames precede the current segment (keeps the substep phase of the forecast contact on the physical substep)."""
        self._h.call("smac_set_segment", int(n_live), int(frame_shift))
        self.n_particles = int(n_live)

    # ------------------------------------------------------------------ hot path (:320-378)
    def substep(self, s, action=None):
        self._push_contact_flags()
        a = None if action is None else _ffi.as_f64(np.asarray(action).reshape(self.n_control, self.dim))
        self._h.call("smac_substep", int(s), _ffi.dptr(a))

    def substep_grad(self, s, action=None, ext_f_grad=None):
        self._push_contact_flags()
        a = None if action is None else _ffi.as_f64(np.asarray(action).reshape(self.n_control, self.dim))
        e = None
        if ext_f_grad is not None:
            e = _ffi.as_f64(np.stack([np.asarray(g, dtype=np.float64).reshape(6) for g in ext_f_grad]), (self.n_primitive, 6))
        out = None if action is None else np.zeros((self.n_control, self.dim))
        self._h.call("smac_substep_grad", int(s), _ffi.dptr(a), _ffi.dptr(e), _ffi.dptr(out))
        if action is None:
            return None
        return out.reshape(np.asarray(action).shape)

    def run_substeps(self, s0, count, action=None):
        """Batched forward: frames s0 .. s0+count-1 in one FFI call (the loop of taichi_env.py:101-102); `action`: the particle
        action held over the window (control_mode "mpm")."""
        self._push_contact_flags()
        if action is None:
            self._h.call("smac_substeps", int(s0), int(count))
            return
        a = _ffi.as_f64(np.asarray(action, dtype=np.float64).reshape(self.n_control, self.dim))
        self._h.call("smac_substeps_action", int(s0), int(count), _ffi.dptr(a))

    def run_substeps_grad(self, s0, count, ext_f_grad=None, action=None):
        """Batched backward: substeps s0+count-1 .. s0 in one FFI call (the loop of taichi_env.py:128-133).  With `action` it returns the
        sum over the window of the per-substep action gradients (what TaichiEnv.step_grad accumulates), read back once."""
        self._push_contact_flags()
        e = None
        if ext_f_grad is not None:
            e = _ffi.as_f64(np.stack([np.asarray(g, dtype=np.float64).reshape(6) for g in ext_f_grad]), (self.n_primitive, 6))
        if action is None:
            self._h.call("smac_substeps_grad", int(s0), int(count), _ffi.dptr(e))
            return None
        a = _ffi.as_f64(np.asarray(action, dtype=np.float64).reshape(self.n_control, self.dim))
        out = np.zeros((self.n_control, self.dim))
        self._h.call("smac_substeps_grad_action", int(s0), int(count), _ffi.dptr(a), _ffi.dptr(e), _ffi.dptr(out))
        return out.reshape(np.asarray(action).shape)

    # ------------------------------------------------------------------ IO (:448-574)
    def get_state(self, f):
        out = np.empty((self.n_particles, 24))
        self._h.call("smac_get_state", int(f), _ffi.dptr(out))
        return out

    def set_state(self, f, state):
        N = self.n_particles
        x, v, F, Cm = (_ffi.as_f64(s, shp) for s, shp in zip(state[:4], ((N, 3), (N, 3), (N, 3, 3), (N, 3, 3))))
        self._h.call("smac_set_frame", int(f), _ffi.dptr(x), _ffi.dptr(v), _ffi.dptr(F), _ffi.dptr(Cm))

    def readframe(self, f, x, v, F, C):                     # :449-456 (fills the caller's arrays)
        N = self.n_particles
        for a, shp in ((x, (N, 3)), (v, (N, 3)), (F, (N, 3, 3)), (C, (N, 3, 3))):
            if not (isinstance(a, np.ndarray) and a.dtype == np.float64 and a.flags.c_contiguous and a.shape == shp):
                raise ValueError(f"readframe: expected a C-contiguous float64 array of shape {shp}")
        self._h.call("smac_get_frame", int(f), _ffi.dptr(x), _ffi.dptr(v), _ffi.dptr(F), _ffi.dptr(C))

    def setframe(self, f, x, v, F, C):                      # :458-466
        self.set_state(f, (x, v, F, C))

    def reset(self, x):
        x = _ffi.as_f64(x)
        assert x.shape[0] == self.n_particles and x.shape[1] in (3, 24)
        self._h.call("smac_reset", _ffi.dptr(x), int(x.shape[1]))
        self.cur = 0

    def copyframe(self, source, target):
        self._h.call("smac_copy_frame", int(source), int(target))

    def get_x(self, f):
        return self._read_field("x", int(f), False)

    def get_v(self, f):
        return self._read_field("v", int(f), False)

    def set_x(self, f, x):
        x = _ffi.as_f64(x, (self.n_particles, 3))
        self._h.call("smac_set_frame", int(f), _ffi.dptr(x), None, None, None)

    def set_v(self, f, v):
        v = _ffi.as_f64(v, (self.n_particles, 3))
        self._h.call("smac_set_frame", int(f), None, _ffi.dptr(v), None, None)

    def get_grad(self, f):
        N = self.n_particles
        gx, gv = np.zeros((N, 3)), np.zeros((N, 3))
        self._h.call("smac_get_grad", int(f), _ffi.dptr(gx), _ffi.dptr(gv), None, None)
        return gx, gv

    def get_grad_full(self, f):
        N = self.n_particles
        gx, gv, gF, gC = np.zeros((N, 3)), np.zeros((N, 3)), np.zeros((N, 3, 3)), np.zeros((N, 3, 3))
        self._h.call("smac_get_grad", int(f), _ffi.dptr(gx), _ffi.dptr(gv), _ffi.dptr(gF), _ffi.dptr(gC))
        return gx, gv, gF, gC

    def add_grad(self, f, gx=None, gv=None, gF=None, gC=None):
        """`x.grad[f, i] += ...` as done by the loss kernels (reference loss_pour.py:130-140)."""
        N = self.n_particles
        arrs = [None if a is None else _ffi.as_f64(np.asarray(a).reshape(shp) if np.asarray(a).size == int(np.prod(shp)) else a, shp)
                for a, shp in zip((gx, gv, gF, gC), ((N, 3), (N, 3), (N, 3, 3), (N, 3, 3)))]
        self._h.call("smac_add_grad", int(f), *[_ffi.dptr(a) for a in arrs])

    def add_grad_device(self, f, gx=None, gv=None, gF=None, gC=None):
        """add_grad from arrays that already live on the GPU: each argument is a device pointer (int) or anything with `.data_ptr()` (a contiguous
        float64 torch tensor of shape (N, 3 | 3 | 9 | 9) on this handle's device).  No host round trip, no synchronisation: a loss evaluated on the
        device seeds the backward pass the way the reference's Taichi loss kernels do (loss_pour.py:130-140)."""
        def ptr(a):
            if a is None:
                return None
            if hasattr(a, "data_ptr"):
                if str(getattr(a, "dtype", "")) != "torch.float64" or not a.is_contiguous():
                    raise ValueError("add_grad_device: contiguous float64 tensors")
                return int(a.data_ptr())
            return int(a)
        self._h.call("smac_add_grad_device", int(f), ptr(gx), ptr(gv), ptr(gF), ptr(gC))

    def clear_grads(self):
        self._h.call("smac_clear_grads")

    def carry_grad(self, src, dst):
        """clear_grads(), except that the particle adjoint of frame `src` survives as the adjoint of frame `dst` (engine/windowed.py)"""
        self._h.call("smac_carry_grad", int(src), int(dst))

    # ------------------------------------------------------------------ control (:579-602)
    def set_action(self, action):                           # :589-592 (the device copy persists until the next set_action)
        a = _ffi.as_f64(np.asarray(action, dtype=np.float64).reshape(self.n_control, self.dim))
        self._h.call("smac_set_action", _ffi.dptr(a))

    def set_control_idx(self, idx=None):
        idx = np.ascontiguousarray(idx, dtype=np.int32)
        if idx.shape != (self.n_particles,):
            raise ValueError(f"set_control_idx: expected shape ({self.n_particles},), got {idx.shape}")
        if self.n_control == 0:
            idx = idx * 0
        self._h.call("smac_set_control_idx", idx.ctypes.data_as(_ffi.c_int32_p))

    def set_materials(self, ids, E2, nu2, yield_stress2=None):
        """Two kinds of particles in one cloud (the reference's mu / lam / yield_stress are per-particle fields, mpm_simulator.py:47-49, filled
        uniformly at :86-90; here: a two-entry table and one selector per particle).  ids[p] = 0: the material of the config; 1: (E2, nu2[,
        yield_stress2]) through the same rules (:40-45).  ids=None: one material again."""
        import ctypes as C
        if ids is None:
            self._h.call("smac_set_material_ids", None)
            return
        ids = np.ascontiguousarray(ids, dtype=np.int32)
        if ids.shape != (self.n_particles,):
            raise ValueError(f"set_materials: expected shape ({self.n_particles},), got {ids.shape}")
        mu2, lam2 = E2 / (2 * (1 + nu2)), E2 * nu2 / ((1 + nu2) * (1 - 2 * nu2))            # :41
        if self.ptype == 1:
            mu2, lam2 = 0.3 * mu2, 0.3 * lam2
        elif self.ptype == 2:
            mu2 = 0.0
        scale2 = getattr(self, "scale", 1.0) ** 2                                            # (soft_cloth mirror: unit-domain moduli are E / s^2)
        self._h.call("smac_set_param", b"mu2", C.c_double(mu2 / scale2))
        self._h.call("smac_set_param", b"lam2", C.c_double(lam2 / scale2))
        if yield_stress2 is None:
            yield_stress2 = self._yield_stress                  # the reference fills yield_stress uniformly from the config (:86-90)
        if mu2 > 0:
            self._h.call("smac_set_param", b"yield_ratio2", C.c_double(float(yield_stress2) / (2.0 * mu2)))
        self._h.call("smac_set_material_ids", ids.ctypes.data_as(_ffi.c_int32_p))

    def compute_grid_m_kernel(self, f):
        out = np.zeros(self.res, dtype=np.float64)
        self._h.call("smac_compute_grid_m", int(f), _ffi.dptr(out))
        return out

    # ------------------------------------------------------------------ measurement helpers
    def count_active_cells(self, f):
        n = C.c_int64(0)
        self._h.call("smac_count_active_cells", int(f), C.byref(n))
        return int(n.value)

    # ------------------------------------------------------------------ losses on the device (losses/loss_*.py)
    def loss_set_target(self, target):
        t = _ffi.as_f64(target)
        assert t.ndim == 2 and t.shape[1] == 3
        self._h.call("smac_loss_set_target", _ffi.dptr(t), int(t.shape[0]))

    def loss_chamfer(self, f, weight=1.0, add_grad=False):
        """Unweighted bidirectional chamfer distance between x[f] and the target; with add_grad,
        x.grad[f] += weight * d(chamfer)/dx (what the reference's tape leaves there)."""
        out = C.c_double(0.0)
        self._h.call("smac_loss_chamfer", int(f), C.c_double(float(weight)), 1 if add_grad else 0, C.byref(out))
        return float(out.value)

    def loss_min_dist(self, f, id_begin, id_end, center, offset=0.01, weight=1.0, add_grad=False):
        """min over particles [id_begin, id_end) of max(|x - center|^2 - offset, 0) at frame f, and the gradient of
        weight * value^2 with respect to `center`; with add_grad the winner's x.grad[f] is seeded too."""
        c = _ffi.as_f64(np.asarray(center, dtype=np.float64).reshape(3))
        out = np.zeros(4)
        self._h.call("smac_loss_min_dist", int(f), int(id_begin), int(id_end), _ffi.dptr(c), C.c_double(float(offset)),
                     C.c_double(float(weight)), 1 if add_grad else 0, _ffi.dptr(out))
        return float(out[0]), out[1:].copy()

    def get_param(self, name):
        v = C.c_double(0.0)
        self._h.call("smac_get_param", name.encode(), C.byref(v))
        return float(v.value)

    def contact_counts(self):
        """(particles inside a contact band, work items holding one) after the most recent forward substep."""
        a, b = C.c_int32(0), C.c_int32(0)
        self._h.call("smac_contact_counts", C.byref(a), C.byref(b))
        return int(a.value), int(b.value)

    def timer_start(self):
        self._h.call("smac_timer_start")

    def timer_stop(self):
        ms = C.c_double(0)
        self._h.call("smac_timer_stop", C.byref(ms))
        return ms.value

    def profile(self, on=True):
        self._h.call("smac_profile_enable", 1 if on else 0)
        self._h.call("smac_profile_reset")

    def profile_report(self):
        n = self._h.lib.smac_profile_count(self._h.h)
        rows = {}
        for i in range(n):
            name = C.create_string_buffer(64)
            ms, cnt = C.c_double(0), C.c_int64(0)
            self._h.call("smac_profile_get", i, name, 64, C.byref(ms), C.byref(cnt))
            rows[name.value.decode()] = (ms.value, int(cnt.value))
        return rows
