"""Scenes for the soft <-> cloth contact path, shaped after the reference's two demos:
  taco  (soft_cloth/config/demo_taco_config.py): mpm_scale 5, n_grid 64, dt 2e-4 / env_dt 2e-3, plastic (von Mises, E 5000, yield 60),
        gravity (0,-5,0), STICKY sheet = the reference's tortilla mesh scaled 1.5 at (2.5, 2.0, 2.5); a disc of plasticine on it;
  hit   (demo_hit_config.py): scale 1, elastic E 500, no gravity, friction 10, non-sticky, one particle controller; the reference's
        hanging towel mesh, a block of particles flying into it.
The sheet moves kinematically (DiffClothAI is out of scope): each vertex follows a smooth prescribed wave."""
import types

import numpy as np

import helpers as H
from softmac_amd.engine.primitive.sdf_cache import load_obj

CLOTH = H.GOLDEN / "cloth"


def _cfg(N, kind, precision, max_steps):
    c = types.SimpleNamespace(dim=3, dtype="float64", quality=1, n_particles=N, dt=2e-4, nu=0.2, max_steps=max_steps, material_model=0,
                              collision_type=2, precision=precision, ground_friction=0.0)
    if kind == "taco":
        c.yield_stress, c.E, c.gravity, c.ptype, c.n_controllers = 60.0, 5000.0, (0.0, -5.0, 0.0), 0, 0
    else:
        c.yield_stress, c.E, c.gravity, c.ptype, c.n_controllers = 50.0, 500.0, (0.0, 0.0, 0.0), 1, 1
    return c


def sheet_motion(kind, rest, scale):
    """vertex positions / velocities as smooth functions of time (so that position and velocity frames are consistent)"""
    amp, om = 0.01 * scale, 40.0
    axis = 1 if kind == "taco" else 2

    def at(t):
        x = rest.copy()
        v = np.zeros_like(rest)
        ph = 3.0 * rest[:, 0] / scale + 2.0 * rest[:, (axis + 1) % 3] / scale
        x[:, axis] += amp * np.sin(om * t + ph) - amp * np.sin(ph)
        v[:, axis] = amp * om * np.cos(om * t + ph)
        return x, v
    return at


def build(kind, precision="float64", n_env_steps=1, N=2500, seed=0, collision_type=2):
    rng = np.random.default_rng(seed)
    if kind == "taco":
        scale = 5.0
        V, F = load_obj(CLOTH / "tortilla.obj")
        rest = V * 1.5 + np.array([2.5, 2.0, 2.5])                       # CLOTH.transform (demo_taco_config.py:73-76)
        r = 0.6 * np.sqrt(rng.uniform(0, 1, N))
        th = rng.uniform(0, 2 * np.pi, N)
        x = np.stack([2.5 + r * np.cos(th), rng.uniform(2.004, 2.10, N), 2.5 + r * np.sin(th)], 1)
        prim = dict(friction=1.0, softness=666.0, cloth_force_scale=1.0, sticky=True)
        v0 = np.tile([0.0, -0.4, 0.0], (N, 1)) + 0.05 * rng.standard_normal((N, 3))
    else:
        scale = 1.0
        rest, F = load_obj(CLOTH / "towel.obj")
        x = rng.uniform([0.44, 0.40, 0.505], [0.56, 0.50, 0.555], (N, 3))
        prim = dict(friction=10.0, softness=666.0, cloth_force_scale=1.0, sticky=False)
        v0 = np.tile([0.1, 0.05, -0.5], (N, 1)) + 0.02 * rng.standard_normal((N, 3))
    env_dt = 2e-3
    substeps = 10
    nframes = n_env_steps * substeps
    cfg = _cfg(N, kind, precision, nframes + 2)
    cfg.collision_type = collision_type                                   # 2: forecast contact (both demos); 1: penalty contact inside p2g
    Fm = np.eye(3)[None] + 0.01 * rng.standard_normal((N, 3, 3))
    Cm = 0.5 * rng.standard_normal((N, 3, 3))
    state = np.hstack([x, v0, Fm.reshape(N, 9), Cm.reshape(N, 9)])
    control_idx = (np.arange(N) % 3 == 0).astype(np.int32) - 1 if cfg.n_controllers else None          # every third particle driven
    return dict(kind=kind, cfg=cfg, env_dt=env_dt, scale=scale, substeps=substeps, state=state, vertices=rest, faces=F.astype(np.int32), prim=prim,
                motion=sheet_motion(kind, rest, scale), nframes=nframes, control_idx=control_idx,
                action=(3.0 * rng.standard_normal((1, 3)) if cfg.n_controllers else None))


def oracle_params(sc):
    from oracle import cloth_oracle as CO
    c = sc["cfg"]
    return CO.ClothSimParams(n_grid=int(128 * c.quality * 0.5), dt=c.dt, E=c.E, nu=c.nu, ptype=c.ptype, material_model=c.material_model, gravity=tuple(c.gravity),
                             collision_type=c.collision_type, substeps=sc["substeps"], n_control=c.n_controllers, scale=sc["scale"], yield_stress=c.yield_stress)


def oracle_prim(sc, x, v):
    import torch
    from oracle import cloth_oracle as CO
    return CO.ClothPrim(torch.as_tensor(x, dtype=CO.DT), torch.as_tensor(v, dtype=CO.DT), torch.as_tensor(sc["faces"].astype(np.int64)),
                        friction=sc["prim"]["friction"], softness=sc["prim"]["softness"], cloth_force_scale=sc["prim"]["cloth_force_scale"],
                        sticky=sc["prim"]["sticky"], mpm_scale=sc["scale"])


def build_engine(sc):
    """HIP simulator + bound cloth primitive (requires a GPU)"""
    from softmac_amd.config import CfgNode
    from softmac_amd.soft_cloth.engine.mpm_simulator import MPMSimulator
    from softmac_amd.soft_cloth.engine.primitive import Primitive_Cloth
    pc = CfgNode()
    for k, v in sc["prim"].items():
        setattr(pc, k, v)
    pc.mpm_force_scale = 1.0
    prim = Primitive_Cloth(pc, max_timesteps=sc["cfg"].max_steps, mpm_scale=sc["scale"], vertices=sc["vertices"], faces=sc["faces"])
    sim = MPMSimulator(sc["cfg"], prim, sc["env_dt"], sc["scale"])
    prim.initialize()
    if sc["control_idx"] is not None:
        sim.set_control_idx(sc["control_idx"])
    return sim, prim
