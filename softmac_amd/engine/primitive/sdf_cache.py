"""SDF cache loading for Mesh primitives (/root/reference/softmac/engine/primitive/mesh.py:136-165).

The reference stores `{signature, sdf, meshes}` pickles next to the OBJ, named by
sha256("v2" + vertices.tobytes() + faces.tobytes()).  This module parses the OBJ without trimesh, recomputes the
signature (after the vertex merge trimesh applies on load) and loads a matching cache with a numpy-only restricted
unpickler (never plain pickle.load on foreign blobs).  A missing cache is built on the GPU (voxelize.py ->
smac_mesh_to_sdf, the MI355X replacement of mesh.py:167-241) and written next to the mesh in the reference's format
when the directory is writable."""
from __future__ import annotations

import hashlib
import importlib
import pathlib
import pickle

import numpy as np

_ALLOWED = {("numpy.core.multiarray", "_reconstruct"), ("numpy._core.multiarray", "_reconstruct"), ("numpy", "ndarray"),
            ("numpy", "dtype"), ("numpy.core.multiarray", "scalar"), ("numpy._core.multiarray", "scalar")}


class _NumpyOnlyUnpickler(pickle.Unpickler):
    def find_class(self, module, name):
        if (module, name) not in _ALLOWED:
            raise pickle.UnpicklingError(f"forbidden global {module}.{name}")
        return getattr(importlib.import_module(module.replace("numpy.core", "numpy._core")), name)


def load_obj(path):
    """Vertices (n,3) float64 and triangle faces (m,3) int64, in file order (fan-triangulating polygons)."""
    v, f = [], []
    for line in open(path):
        p = line.split()
        if not p:
            continue
        if p[0] == "v":
            v.append([float(c) for c in p[1:4]])
        elif p[0] == "f":
            idx = [int(t.split("/")[0]) for t in p[1:]]
            idx = [i - 1 if i > 0 else len(v) + i for i in idx]
            for k in range(1, len(idx) - 1):
                f.append([idx[0], idx[k], idx[k + 1]])
    return np.asarray(v, dtype=np.float64), np.asarray(f, dtype=np.int64)


def signature(vertices, faces):
    h = hashlib.sha256()
    h.update(bytes("v2", encoding="utf-8"))
    h.update(np.ascontiguousarray(vertices, dtype=np.float64).tobytes())
    h.update(np.ascontiguousarray(faces, dtype=np.int64).tobytes())
    return h.hexdigest()


def load_or_build_sdf(mesh_path, device=0, write_cache=True):
    from .voxelize import merge_vertices, mesh_to_sdf
    mesh_path = pathlib.Path(mesh_path)
    vertices, faces = merge_vertices(*load_obj(mesh_path))
    sig = signature(vertices, faces)
    cache = mesh_path.parent.absolute() / sig
    if cache.exists():
        with open(cache, "rb") as fh:
            blob = _NumpyOnlyUnpickler(fh).load()
        return blob["sdf"], (vertices, faces)
    sdf = mesh_to_sdf(vertices, faces, device=device)                # mesh.py:155-157
    if write_cache:
        try:
            with open(cache, "wb") as fh:                            # mesh.py:159-161, same layout
                pickle.dump({"signature": sig, "sdf": sdf, "meshes": [(vertices, faces)]}, fh)
        except OSError:
            pass                                                     # read-only asset directory: rebuild next time
    return sdf, (vertices, faces)
