"""BASELINE config C1 - the reference's default pour scene (softmac/config/demo_pour_config.py:9-64): the full 5000-particle
liquid state `envs/pour/pour_mpm_init_state_corotated.npy` (+ the (0, 0.04, 0) offset of :31-36) sitting in the glass, the
glass (friction 0.1) and the bowl (friction 1.0) as Mesh primitives in forecast contact, n_grid 64, dt = env_dt = 1e-3,
E 22, liquid.  The reference drives the glass through Jade (out of scope); here it is tilted kinematically, poses advanced
with the reference's forward_kinematics rule.  The SDF caches of both meshes are missing from the reference checkout
(.MISSING_LARGE_BLOBS); they are rebuilt from the OBJ geometry (tests/golden/pour_scene.npz) by the voxeliser."""
import numpy as np
import torch

import helpers as H
from helpers import O

GLASS_POS = (0.7, 0.23488457 + 0.04 + 0.04, 0.5)             # demo_pour_config.py:48
BOWL_POS = (0.34, 0.08737724 + 0.04, 0.5)                    # :50


def load():
    d = np.load(H.GOLDEN / "pour_scene.npz")
    state = d["state"].copy()
    state[:, :3] += np.array([0.0, 0.04, 0.0])               # SHAPES offset
    return state, d["target"], (d["glass_vertices"], d["glass_faces"]), (d["bowl_vertices"], d["bowl_faces"])


def spec_from_table(t, friction):
    return dict(sdf=t["sdf"], normal=t["normal"], lower=np.asarray(t["position"][0]), upper=np.asarray(t["position"][1]),
                dx=float(np.asarray(t["dx"]).reshape(-1)[0]), res=np.asarray(t["res"]), friction=friction, softness=666.0, contact=True)


def primitive_states(nframes, dt):
    """glass: tilting about z at 2 rad/s while moving towards the bowl; bowl: at rest.  Poses by forward_kinematics."""
    q = torch.tensor([1.0, 0.0, 0.0, 0.0], dtype=O.DT)
    gp = torch.tensor(GLASS_POS, dtype=O.DT)
    gv, gw = torch.tensor([-0.08, 0.03, 0.01], dtype=O.DT), torch.tensor([0.1, -0.05, 2.0], dtype=O.DT)
    bowl = np.concatenate([BOWL_POS, [1, 0, 0, 0], np.zeros(6)])
    out = []
    for f in range(nframes + 1):
        out.append([np.concatenate([gp.numpy(), q.numpy(), gv.numpy(), gw.numpy()]), bowl.copy()])
        gp, q = O.forward_kinematics(gp, q, gv, gw, dt)
    return out


def build(tables, precision="float64", nsteps=4):
    state, target, _, _ = load()
    cfg = H.sim_cfg(len(state), n_grid=64, dt=1e-3, E=22.0, nu=0.2, ptype=2, material_model=0, ground_friction=0.0,
                    collision_type=2, max_steps=nsteps + 2, precision=precision)
    specs = [spec_from_table(tables[0], 0.1), spec_from_table(tables[1], 1.0)]           # [Glass, Bowl] (:64)
    return dict(cfg=cfg, env_dt=1e-3, state=state, target=target, specs=specs, pstates=primitive_states(nsteps, 1e-3), nsteps=nsteps,
                ext_f_grad=[np.linspace(-1e-2, 1e-2, 6), np.linspace(2e-2, -2e-2, 6)])
