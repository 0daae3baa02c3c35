"""Primitive - host-side mirror of /root/reference/softmac/engine/primitive/primitive_base.py:8-336.

The per-frame rigid state (position3, rotation4, v3, w3), its adjoint, the wrench accumulator
`ext_f` and the velocity-control action buffer live on the GPU inside the simulator handle
(one slot per primitive); this class keeps the reference's method names and field-like accessors
(`friction[None]`, `ext_f.to_numpy()`, `position[f]`, `position.grad[f]`) and forwards them to the
C ABI.  Until the simulator binds the primitive to a handle, writes are kept in a host mirror and
flushed at bind time (the reference builds primitives before the simulator, taichi_env.py:32-38).
"""
from __future__ import annotations

import numpy as np

from ... import _ffi


class _Scalar:
    """`field[None]` get/set for friction / softness (reference :26-27)."""

    def __init__(self, owner, key, value):
        self._owner, self._key, self._value = owner, key, float(value)

    def __getitem__(self, _):
        return self._value

    def __setitem__(self, _, value):
        self._value = float(value)
        self._owner._push_params()


class _StateView:
    """`prim.position[f]` / `.rotation[f]` / `.v[f]` / `.w[f]` reads (+ `.grad[f]`)."""

    def __init__(self, owner, lo, hi, grad=False):
        self._owner, self._lo, self._hi, self._grad = owner, lo, hi, grad
        if not grad:
            self.grad = _StateView(owner, lo, hi, grad=True)

    def __getitem__(self, f):
        o = self._owner
        if self._grad:
            return o._get_state_grad(int(f), int(f) + 1)[self._lo:self._hi]
        return o._get_state13(int(f))[self._lo:self._hi]


class _ExtF:
    def __init__(self, owner):
        self._owner = owner

    def to_numpy(self):
        o = self._owner
        out = np.zeros(6)
        if o._h is not None:
            o._h.call("smac_prim_get_ext_f", o._slot, _ffi.dptr(out))
        return out

    def __getitem__(self, _):
        return self.to_numpy()


class Primitive:
    state_dim = 7

    def __init__(self, cfg=None, dim=3, max_timesteps=2048, dtype=np.float64, rigid_velocity_control=False, **kwargs):
        self.cfg = self.default_config() if cfg is None else cfg
        self.dim = dim
        self.max_timesteps = int(max_timesteps)
        self.dtype = dtype
        self.rotation_dim = 4
        self.angular_velocity_dim = 3
        self.friction = _Scalar(self, "friction", getattr(self.cfg, "friction", 0.9))
        self.softness = _Scalar(self, "softness", 0.0)
        self.position = _StateView(self, 0, 3)
        self.rotation = _StateView(self, 3, 7)
        self.v = _StateView(self, 7, 10)
        self.w = _StateView(self, 10, 13)
        self.enable_external_force = getattr(self.cfg, "enable_external_force", True)
        self.ext_f = _ExtF(self)
        self.rigid_velocity_control = rigid_velocity_control
        self._h, self._slot = None, -1
        self._contact = True
        self._pending = []            # (f0, f1, state13) writes issued before binding
        self._sdf = None              # dict(sdf, normal, lower, upper, dx, res) set by Mesh

    # ------------------------------------------------------------------ binding
    def _bind(self, handle, slot):
        self._h, self._slot = handle, int(slot)
        if self._sdf is not None:
            s = self._sdf
            res = np.ascontiguousarray(s["res"], dtype=np.int32)
            self._h.call("smac_prim_upload_sdf", self._slot, _ffi.dptr(_ffi.as_f64(s["sdf"])),
                         _ffi.dptr(_ffi.as_f64(s["normal"])), res.ctypes.data_as(_ffi.c_int32_p),
                         _ffi.dptr(_ffi.as_f64(s["lower"])), _ffi.dptr(_ffi.as_f64(s["upper"])), float(s["dx"]))
        self._push_params()
        for f0, f1, st in self._pending:
            self._h.call("smac_prim_set_state", self._slot, f0, f1, _ffi.dptr(st))
        self._pending = []

    def _push_params(self, contact=None):
        if contact is not None:
            self._contact = bool(contact)
        if self._h is not None:
            has_table = self._sdf is not None
            self._h.call("smac_prim_set_params", self._slot, self.friction[None], self.softness[None],
                         1 if (self._contact and has_table) else 0)

    def _get_state13(self, f):
        out = np.zeros(13)
        if self._h is None:
            for f0, f1, st in self._pending:
                if f0 <= f < f1:
                    out = st.copy()
            return out
        self._h.call("smac_prim_get_state", self._slot, int(f), _ffi.dptr(out))
        return out

    def _get_state_grad(self, f0, f1):
        out = np.zeros(13)
        if self._h is not None:
            self._h.call("smac_prim_get_state_grad", self._slot, int(f0), int(f1), _ffi.dptr(out))
        return out

    # ------------------------------------------------------------------ reference API
    def clear_ext_f(self):                                   # :183-187
        if self._h is not None:
            self._h.call("smac_prim_clear_ext_f", self._slot)

    def set_ext_f_grad(self, ext_f_grad):                    # :189-192 - forwarded by substep_grad(ext_f_grad=...)
        self._ext_f_grad = np.asarray(ext_f_grad, dtype=np.float64).reshape(6)

    def get_state(self, f):                                  # :248-251
        return self._get_state13(f)[:7].copy()

    def set_state(self, f, state):                           # :253-256
        ss = self._get_state13(f)
        ss[:len(state)] = state
        self._write(f, f + 1, ss)

    def set_all_states(self, f, state):                      # :258-260
        self._write(int(f), int(f) + 1, _ffi.as_f64(np.asarray(state, dtype=np.float64).reshape(13)))

    def set_all_states_range(self, f0, f1, state):
        """Batched set_all_states for frames [f0, f1): one FFI call instead of 2*(f1-f0) launches
        (reference rigid_simulator.py:200-201)."""
        self._write(int(f0), int(f1), _ffi.as_f64(np.asarray(state, dtype=np.float64).reshape(13)))

    def _write(self, f0, f1, st13):
        st13 = _ffi.as_f64(st13)
        if self._h is None:
            self._pending.append((f0, f1, st13.copy()))
        else:
            self._h.call("smac_prim_set_state", self._slot, f0, f1, _ffi.dptr(st13))

    def set_states_trajectory(self, f0, states):
        """one 13-vector per frame of [f0, f0 + len(states)) in ONE transfer (an episode's or a window's prescribed trajectory)"""
        st = _ffi.as_f64(np.asarray(states, dtype=np.float64).reshape(-1, 13))
        if self._h is None:
            for j in range(len(st)):
                self._pending.append((f0 + j, f0 + j + 1, st[j].copy()))
        else:
            self._h.call("smac_prim_set_states", self._slot, int(f0), int(f0) + len(st), _ffi.dptr(st))

    def get_states_grad_trajectory(self, f0, f1):
        """(f1 - f0, 13): get_all_states_grad of every frame of [f0, f1) in ONE transfer"""
        out = np.zeros((int(f1) - int(f0), 13))
        if self._h is not None:
            self._h.call("smac_prim_get_state_grads", self._slot, int(f0), int(f1), _ffi.dptr(out))
        return out

    def get_all_states_grad(self, f):                        # :262-265
        return self._get_state_grad(int(f), int(f) + 1)

    def get_all_states_grad_sum(self, f0, f1):
        return self._get_state_grad(int(f0), int(f1))

    def add_state_grad(self, f, g13):
        g = _ffi.as_f64(np.asarray(g13, dtype=np.float64).reshape(13))
        self._h.call("smac_prim_add_state_grad", self._slot, int(f), _ffi.dptr(g))

    def initialize(self):                                    # :267-269
        self.friction[None] = getattr(self.cfg, "friction", 0.9)
        self.reset()

    def reset(self):                                         # :271-275
        self._pending = []
        if self._h is not None:
            self._h.call("smac_prim_reset", self._slot)

    def forward_kinematics(self, f, dt=None):                # :280-283 (dt is the simulator's dt)
        self._h.call("smac_prim_forward_kinematics", self._slot, int(f))

    def forward_kinematics_grad(self, f, dt=None):
        self._h.call("smac_prim_forward_kinematics_grad", self._slot, int(f))

    def set_action(self, s, n, action):                      # :311-313
        a = _ffi.as_f64(np.asarray(action, dtype=np.float64).reshape(6))
        self._h.call("smac_prim_set_action", self._slot, int(s), int(n), _ffi.dptr(a))

    def get_action_grad(self, s, n):                         # :315-319
        g = np.zeros(6)
        self._h.call("smac_prim_get_action_grad", self._slot, int(s), int(n), _ffi.dptr(g))
        return g

    def get_action_grads(self, s0, s1, n):
        """(s1 - s0, 6): get_action_grad of the env steps [s0, s1) in one launch and one transfer (an episode's backward())"""
        g = np.zeros((int(s1) - int(s0), 6))
        self._h.call("smac_prim_get_action_grads", self._slot, int(s0), int(s1), int(n), _ffi.dptr(g))
        return g

    @classmethod
    def default_config(cls):                                 # :328-335
        from ...config import CfgNode as CN
        cfg = CN()
        cfg.friction = 0.9
        cfg.enable_external_force = True
        cfg.urdf_path = ''
        return cfg
