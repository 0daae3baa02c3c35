#!/usr/bin/env python3
"""Derive small test fixtures from the reference's own DATA files (run in the build container,
where /root/reference exists; the outputs under tests/golden/ are committed and travel).

Only data is read - particle-state arrays and a cached SDF table - never source.  The SDF
cache is a pickle, so it is opened with a numpy-only restricted unpickler.

  grip_state_2k.npz   2000-particle subsample of softmac/envs/grip/grip_mpm_init_state.npy
  pour_state_1k.npz   1000-particle subsample of softmac/envs/pour/pour_mpm_init_state_corotated.npy
  palm_sdf.npz        sdf/normal tables of softmac/assets/gripper/palm.obj (the cached 6895...c4d5 blob)
  door_sdf.npz        sdf/normal tables of softmac/assets/door/door.obj (the cached e7ab...561a blob; four touching boxes)
  pour_scene.npz      BASELINE config C1 inputs: the full 5000-particle pour state + target, and the glass / bowl collision
                      meshes (vertices / faces of assets/glass/glass.obj, assets/bowl/bowl.obj after trimesh's vertex merge)
  grip_scene.npz      the full 10000-particle settled grip state (equilibrium pin, tests/test_equilibrium.py) + target, and
                      the gripper finger mesh (assets/gripper/finger.obj)
"""
import importlib
import pathlib
import pickle
import sys

import numpy as np

REF = pathlib.Path("/root/reference/softmac")
OUT = pathlib.Path(__file__).resolve().parent.parent / "tests" / "golden"

_ALLOWED = {("numpy.core.multiarray", "_reconstruct"), ("numpy._core.multiarray", "_reconstruct"),
            ("numpy", "ndarray"), ("numpy", "dtype"),
            ("numpy.core.multiarray", "scalar"), ("numpy._core.multiarray", "scalar")}


class NumpyOnlyUnpickler(pickle.Unpickler):
    def find_class(self, module, name):
        if (module, name) not in _ALLOWED:
            raise pickle.UnpicklingError(f"forbidden global {module}.{name}")
        return getattr(importlib.import_module(module.replace("numpy.core", "numpy._core")), name)


def main():
    OUT.mkdir(parents=True, exist_ok=True)
    rng = np.random.default_rng(0)
    grip = np.load(REF / "envs/grip/grip_mpm_init_state.npy")
    idx = np.sort(rng.choice(len(grip), 2000, replace=False))
    np.savez_compressed(OUT / "grip_state_2k.npz", state=grip[idx], index=idx)
    pour = np.load(REF / "envs/pour/pour_mpm_init_state_corotated.npy")
    idx = np.sort(rng.choice(len(pour), 1000, replace=False))
    np.savez_compressed(OUT / "pour_state_1k.npz", state=pour[idx], index=idx)
    with open(REF / "assets/gripper/68956732a79bf09d8703ab990a2e2319bf5492c792294e9a86632db03b5ac4d5", "rb") as f:
        blob = NumpyOnlyUnpickler(f).load()
    sdf = blob["sdf"]
    np.savez_compressed(OUT / "palm_sdf.npz", sdf=sdf["sdf"], normal=sdf["normal"],
                        lower=np.asarray(sdf["position"][0]), upper=np.asarray(sdf["position"][1]),
                        dx=float(sdf["dx"][0]), res=np.asarray(sdf["res"]),
                        vertices=blob["meshes"][0][0], faces=blob["meshes"][0][1])
    with open(REF / "assets/door/e7ab3378b317f8d1d4de18fa5bfa4d98e79629e714104b720ebcf0470dfc561a", "rb") as f:
        blob = NumpyOnlyUnpickler(f).load()
    sdf = blob["sdf"]
    np.savez_compressed(OUT / "door_sdf.npz", sdf=sdf["sdf"], normal=sdf["normal"].astype(np.float32),
                        lower=np.asarray(sdf["position"][0]), upper=np.asarray(sdf["position"][1]),
                        dx=float(sdf["dx"][0]), res=np.asarray(sdf["res"]),
                        vertices=blob["meshes"][0][0], faces=blob["meshes"][0][1])
    sys.path.insert(0, str(OUT.parent.parent))
    from softmac_amd.engine.primitive.sdf_cache import load_obj
    from softmac_amd.engine.primitive.voxelize import merge_vertices
    gv, gf = merge_vertices(*load_obj(REF / "assets/glass/glass.obj"))
    bv, bf = merge_vertices(*load_obj(REF / "assets/bowl/bowl.obj"))
    np.savez_compressed(OUT / "pour_scene.npz", state=pour, target=np.load(REF / "envs/pour/pour_mpm_target_position_corotated.npy"),
                        glass_vertices=gv, glass_faces=gf.astype(np.int32), bowl_vertices=bv, bowl_faces=bf.astype(np.int32))
    fv, ff = merge_vertices(*load_obj(REF / "assets/gripper/finger.obj"))
    np.savez_compressed(OUT / "grip_scene.npz", state=grip, target=np.load(REF / "envs/grip/grip_mpm_target_position.npy"),
                        finger_vertices=fv, finger_faces=ff.astype(np.int32))
    for p in sorted(OUT.glob("*.npz")):
        print(p.name, p.stat().st_size)


if __name__ == "__main__":
    sys.exit(main())
