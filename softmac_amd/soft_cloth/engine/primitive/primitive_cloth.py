"""Primitive_Cloth - host-side mirror of the reference class (soft_cloth/engine/primitive/primitive_cloth.py:26-387): one
triangle-mesh sheet whose vertex positions / velocities are set per frame from outside, which collides with the MPM particles
(collide_mixed :233-280 on the device, csrc/smac_cloth.hpp) and collects the contact force per vertex (`ext_f`).

The data lives in the MPMSimulator's handle (include/softmac_hip.h, smac_cloth_*); this object is bound to it when the
simulator is constructed.  Arrays are float64 numpy in physical units, as in the reference."""
from __future__ import annotations

import ctypes as C

import numpy as np

from ...._ffi import as_f64, c_int8_p, c_int32_p, dptr
from ....config.cfgnode import CfgNode as CN
from ....engine.primitive.sdf_cache import load_obj
from .process_faces import process


def make_cls_config(self, cfg=None, **kwargs):
    _cfg = self.default_config()
    if cfg is not None:
        if isinstance(cfg, str):
            _cfg.merge_from_file(cfg)
        else:
            _cfg.merge_from_other_cfg(cfg)
    if len(kwargs) > 0:
        _cfg.merge_from_list(sum(list(kwargs.items()), ()))
    return _cfg


class _ExtF:
    def __init__(self, prim):
        self._p = prim

    def to_numpy(self):
        out = np.zeros((self._p.num_vertices, 3))
        self._p._handle().call("smac_cloth_get_ext_f", dptr(out))
        return out


class Primitive_Cloth:
    def __init__(self, cfg=None, dim=3, max_timesteps=4096, dtype=np.float64, mesh_path="", mpm_scale=1., vertices=None, faces=None, **kwargs):
        """`mesh_path`: an OBJ file (the reference loads it with trimesh, :44); alternatively `vertices` (n,3) / `faces` (m,3)."""
        self.cfg = make_cls_config(self, cfg, **kwargs)
        self.dim = dim
        self.max_timesteps = max_timesteps
        self.dtype = dtype
        self.mpm_scale = float(mpm_scale)
        if vertices is None:
            vertices, faces = load_obj(mesh_path)
        self.rest_vertices = np.ascontiguousarray(vertices, dtype=np.float64)
        self.mesh_faces = np.ascontiguousarray(faces, dtype=np.int32)
        self.num_vertices = self.rest_vertices.shape[0]
        self.num_faces = self.mesh_faces.shape[0]
        self.n_neighbors = 200                                                # :48
        self.neighbor_faces_np, self.neighbor_faces_direction_np = process(self.mesh_faces, self.n_neighbors)   # :53
        self.sticky = bool(self.cfg.sticky)
        self.ext_f = _ExtF(self)
        self._h = None

    # ---- binding (MPMSimulator.__init__)
    def _bind(self, handle):
        self._h = handle
        handle.call("smac_cloth_create", int(self.num_vertices), int(self.num_faces), self.mesh_faces.ctypes.data_as(c_int32_p), int(self.n_neighbors),
                    np.ascontiguousarray(self.neighbor_faces_np, dtype=np.int32).ctypes.data_as(c_int32_p),
                    np.ascontiguousarray(self.neighbor_faces_direction_np, dtype=np.int8).ctypes.data_as(c_int8_p),
                    C.c_double(float(self.cfg.friction)), C.c_double(float(self.cfg.softness)), C.c_double(float(self.cfg.cloth_force_scale)),
                    1 if self.cfg.sticky else 0, C.c_double(self.mpm_scale))

    def _handle(self):
        if self._h is None:
            raise RuntimeError("Primitive_Cloth is not bound to a simulator yet (construct MPMSimulator(cfg, primitive, ...))")
        return self._h

    # ---- IO (:285-365)
    def clear_ext_f(self):
        self._handle().call("smac_cloth_clear_ext_f")

    def set_ext_f_grad(self, ext_f_grad):
        g = as_f64(np.asarray(ext_f_grad, dtype=np.float64).reshape(self.num_vertices, 3))
        self._handle().call("smac_cloth_set_ext_f_grad", dptr(g))

    def clear_all_states(self):
        z = np.zeros((self.num_vertices, 3))
        self._handle().call("smac_cloth_set_state", 0, int(self.max_timesteps), dptr(z), dptr(z))

    def get_all_states(self, f):
        x, v = np.zeros((self.num_vertices, 3)), np.zeros((self.num_vertices, 3))
        self._handle().call("smac_cloth_get_state", int(f), dptr(x), dptr(v))
        return x, v

    def get_all_states_grad(self, f):
        x, v = np.zeros((self.num_vertices, 3)), np.zeros((self.num_vertices, 3))
        self._handle().call("smac_cloth_get_state_grad", int(f), dptr(x), dptr(v))
        return x, v

    def set_all_states(self, f, x, v, f_end=None):
        """frame f (the reference's signature), or frames [f, f_end) in one call"""
        x = as_f64(np.asarray(x, dtype=np.float64).reshape(-1, 3), (self.num_vertices, 3))
        v = as_f64(np.asarray(v, dtype=np.float64).reshape(-1, 3), (self.num_vertices, 3))
        self._handle().call("smac_cloth_set_state", int(f), int(f + 1 if f_end is None else f_end), dptr(x), dptr(v))

    def get_vertices(self, f):
        return self.get_all_states(f)[0]

    def initialize(self):                                                     # :367-376 (the contact parameters went in at _bind)
        self.clear_all_states()
        self.clear_ext_f()

    @classmethod
    def default_config(cls):                                                  # :379-387
        cfg = CN()
        cfg.friction = 0.9
        cfg.softness = 666.
        cfg.cloth_force_scale = 1.0
        cfg.mpm_force_scale = 1.0
        cfg.sticky = False
        return cfg
