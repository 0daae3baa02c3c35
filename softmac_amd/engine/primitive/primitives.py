"""Primitives - list-like container built from URDF collision meshes
(/root/reference/softmac/engine/primitive/primitives.py:15-60)."""
from __future__ import annotations

import os
import xml.etree.ElementTree as ET
from pathlib import Path

import numpy as np

from .mesh import Mesh


class Primitives:
    def __init__(self, cfgs=(), max_timesteps=2048, rigid_velocity_control=False, primitives=None):
        self.primitives = []
        self.urdfs = []
        if primitives is not None:                      # pre-built Mesh objects (tests / synthetic scenes)
            self.primitives = list(primitives)
            return
        for i in cfgs:
            self.urdfs.append(i)
            mesh_paths, colors = self.load_info_from_urdf(i.urdf_path)
            for mesh_path, color in zip(mesh_paths, colors):
                self.primitives.append(Mesh(mesh_path, color=color, cfg=i, max_timesteps=max_timesteps,
                                            rigid_velocity_control=rigid_velocity_control))

    def load_info_from_urdf(self, urdf_path):           # primitives.py:26-41
        root = ET.parse(urdf_path).getroot()
        mesh_elements = root.findall(".//collision/geometry/mesh")
        mesh_file_paths = [Path(os.path.dirname(urdf_path)) / m.attrib.get("filename", "") for m in mesh_elements]
        color_elements = root.findall(".//visual/material/color")
        colors = [np.array([float(c) for c in e.attrib.get("rgba", "").split()[:4]]) for e in color_elements]
        return mesh_file_paths, colors

    def set_softness(self, softness=666.):
        for i in self.primitives:
            i.softness[None] = softness

    def __getitem__(self, item):
        if isinstance(item, tuple):
            item = item[0]
        return self.primitives[item]

    def __len__(self):
        return len(self.primitives)

    def initialize(self):
        self.set_softness(666.)

    def reset(self):
        for i in self.primitives:
            i.reset()
