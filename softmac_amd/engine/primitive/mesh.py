"""Mesh primitive: a rigid body described by a voxelised signed-distance table
(/root/reference/softmac/engine/primitive/mesh.py:18-43).  The lookup itself (mesh.py:45-113)
runs on the GPU (csrc/smac_math.hpp prim_sdf / prim_normal)."""
from __future__ import annotations

import numpy as np

from .primitive_base import Primitive


class Mesh(Primitive):
    def __init__(self, mesh_path=None, color=None, sdf=None, **kwargs):
        super().__init__(**kwargs)
        self.mesh_path = mesh_path
        self.urdf_path = getattr(self.cfg, "urdf_path", "")
        self.color = color
        if sdf is None:
            from .sdf_cache import load_or_build_sdf
            sdf, self.mesh_rest = load_or_build_sdf(mesh_path)
        else:
            self.mesh_rest = None
        self.set_sdf(sdf)

    def set_sdf(self, sdf):
        """sdf: dict with the reference's keys (mesh.py:235-241): sdf, normal, position=(lower,upper), dx, res
        (or lower/upper given directly)."""
        lower, upper = (sdf["position"] if "position" in sdf else (sdf["lower"], sdf["upper"]))
        dx = np.asarray(sdf["dx"], dtype=np.float64).reshape(-1)[0]
        table = np.ascontiguousarray(sdf["sdf"], dtype=np.float64)
        self.sdf_dx = float(dx)
        self.inv_sdf_dx = 1 / self.sdf_dx
        self.sdf_res = list(table.shape)
        self._sdf = dict(sdf=table, normal=np.ascontiguousarray(sdf["normal"], dtype=np.float64),
                         lower=np.asarray(lower, dtype=np.float64), upper=np.asarray(upper, dtype=np.float64),
                         dx=self.sdf_dx, res=np.asarray(table.shape, dtype=np.int32))
        if self._h is not None:
            self._bind(self._h, self._slot)
