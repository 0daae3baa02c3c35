"""The scene's rigid bodies as one indexable collection.

Contract kept from the reference's `Primitives` (/root/reference/softmac/engine/primitive/primitives.py:15-60), because `TaichiEnv`, the
rigid simulators, the losses and the demos use it: `Primitives(cfg.PRIMITIVES, max_timesteps, rigid_velocity_control)` makes one `Mesh`
per collision mesh of every URDF in the config list (in document order, paired with the document's visual colours); `len()`, indexing
(a tuple index means its first entry, as Taichi-style `prims[i,]` call sites write it), `.urdfs`, `.primitives`, `initialize()` (softness
666 on every body), `set_softness`, `reset()`.  The two XPath expressions are the URDF schema, not code.  Built here on a small URDF
reader that returns records; `primitives=` takes ready-made `Mesh` objects (synthetic scenes, tests, bench.py).
"""
from __future__ import annotations

import xml.etree.ElementTree as ET
from pathlib import Path
from typing import NamedTuple

import numpy as np

from .mesh import Mesh

DEFAULT_SOFTNESS = 666.0
COLLISION_MESHES = ".//collision/geometry/mesh"
VISUAL_COLORS = ".//visual/material/color"


class UrdfBody(NamedTuple):
    mesh_path: Path
    rgba: np.ndarray


def read_urdf_bodies(urdf_path):
    """collision meshes of a URDF (paths relative to the URDF's directory) with the visual colours of the same document, pairwise"""
    doc = ET.parse(urdf_path).getroot()
    folder = Path(urdf_path).parent
    meshes = []
    for node in doc.findall(COLLISION_MESHES):
        name = node.attrib.get("filename", "")
        if not name:
            raise ValueError(f"{urdf_path}: a collision mesh without a filename")
        meshes.append(folder / name)
    colours = [np.array(node.attrib.get("rgba", "").split()[:4], dtype=np.float64) for node in doc.findall(VISUAL_COLORS)]
    return [UrdfBody(m, c) for m, c in zip(meshes, colours)]


class Primitives:
    def __init__(self, cfgs=(), max_timesteps=2048, rigid_velocity_control=False, primitives=None):
        self.urdfs = []
        if primitives is not None:
            self.primitives = list(primitives)
            return
        self.primitives = []
        for cfg in cfgs:
            self.urdfs.append(cfg)
            self.primitives += [Mesh(body.mesh_path, color=body.rgba, cfg=cfg, max_timesteps=max_timesteps, rigid_velocity_control=rigid_velocity_control)
                                for body in read_urdf_bodies(cfg.urdf_path)]

    def load_info_from_urdf(self, urdf_path):
        """the reference's accessor: (mesh paths, colours) of one URDF"""
        bodies = read_urdf_bodies(urdf_path)
        return [b.mesh_path for b in bodies], [b.rgba for b in bodies]

    def __len__(self):
        return len(self.primitives)

    def __iter__(self):
        return iter(self.primitives)

    def __getitem__(self, index):
        return self.primitives[index[0] if isinstance(index, tuple) else index]

    def set_softness(self, softness=DEFAULT_SOFTNESS):
        for body in self.primitives:
            body.softness[None] = softness

    def initialize(self):
        self.set_softness(DEFAULT_SOFTNESS)

    def reset(self):
        for body in self.primitives:
            body.reset()
