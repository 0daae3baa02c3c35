"""Mesh -> SDF tables, the reference's `Mesh.task` / `trimesh2sdf` (/root/reference/softmac/engine/primitive/mesh.py:167-241)
without trimesh: the sampling box is chosen on the host exactly as the reference does, the distance / sign / normal of
every sample is computed by the HIP kernel behind `smac_mesh_to_sdf` (softmac_amd/csrc/smac_voxel.hpp).  No CPU
fallback: without a GPU this raises, like every other compute entry point."""
from __future__ import annotations

import ctypes as C

import numpy as np

from ... import _ffi


def merge_vertices(vertices, faces, digits=8):
    """What `trimesh.load(path, force='mesh')` does to an OBJ before the reference hashes it (mesh.py:139-148):
    vertices that coincide to 1e-8 are merged, first occurrence kept, original order preserved."""
    key = np.round(np.asarray(vertices, dtype=np.float64) * 10 ** digits).astype(np.int64)
    first, keep = {}, []
    inv = np.zeros(len(key), dtype=np.int64)
    for i, k in enumerate(map(tuple, key)):
        if k not in first:
            first[k] = len(keep)
            keep.append(i)
        inv[i] = first[k]
    return np.asarray(vertices, dtype=np.float64)[keep], inv[np.asarray(faces, dtype=np.int64)]


def sampling_box(vertices, margin=None, dx=None):
    """mesh.py:170-176 (dx, margin) and :191-193, :233-234 (res, lower, upper): returns dx, res, first sample, last sample."""
    lo, hi = vertices.min(0), vertices.max(0)
    length = float(np.max(hi - lo))
    if dx is None:
        dx = min(0.01, length / 80)
    if margin is None:
        margin = max(dx * 3, 0.01)
    center = (lo + hi) / 2
    res = np.ceil((hi - lo + margin * 2) / dx).astype(int)
    lower = center - res * dx / 2.0 + dx / 2.0
    upper = lower + (res - 1) * dx
    return float(dx), res, lower, upper


def mesh_to_sdf(vertices, faces, margin=None, dx=None, device=0):
    """Returns the dict the reference caches: sdf (res), normal (res, 3), position (lower, upper), dx (3,), res."""
    vertices = np.ascontiguousarray(vertices, dtype=np.float64)
    faces32 = np.ascontiguousarray(faces, dtype=np.int32)
    dx, res, lower, upper = sampling_box(vertices, margin, dx)
    lib = _ffi.load_library()
    sdf = np.zeros(tuple(res), dtype=np.float64)
    normal = np.zeros(tuple(res) + (3,), dtype=np.float64)
    res32 = np.ascontiguousarray(res, dtype=np.int32)
    lower = np.ascontiguousarray(lower, dtype=np.float64)
    rc = lib.smac_mesh_to_sdf(int(device), _ffi.dptr(vertices), len(vertices), faces32.ctypes.data_as(_ffi.c_int32_p), len(faces32),
                              _ffi.dptr(lower), res32.ctypes.data_as(_ffi.c_int32_p), C.c_double(dx), _ffi.dptr(sdf), _ffi.dptr(normal))
    if rc != 0:
        msg = lib.smac_last_error(None)
        raise _ffi.SmacError(f"smac_mesh_to_sdf failed ({rc}): {msg.decode() if msg else ''}")
    return {"sdf": sdf, "normal": normal, "position": (lower, upper), "dx": np.ones(3) * dx, "res": res}
