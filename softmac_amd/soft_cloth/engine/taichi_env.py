"""TaichiEnv of the soft <-> cloth variant: the env loop of the reference (soft_cloth/engine/taichi_env.py:86-141) around the HIP
simulator, with the kinematic sheet driver in place of DiffClothAI.  Renderer, shapes and losses are not part of this path."""
from __future__ import annotations

import numpy as np

from .cloth_simulator import KinematicCloth
from .mpm_simulator import MPMSimulator
from .primitive import Primitive_Cloth


class TaichiEnv:
    def __init__(self, cfg, init_particles, vertices=None, faces=None, mesh_path="", motion=None, motion_grad=None):
        self.cfg = cfg
        self.env_dt = cfg.env_dt
        self.mpm_scale = cfg.mpm_scale
        self.substeps = int(cfg.env_dt / cfg.SIMULATOR.dt)
        self.init_particles = np.asarray(init_particles, dtype=np.float64)
        self.primitive = Primitive_Cloth(cfg.PRIMITIVES, max_timesteps=cfg.SIMULATOR.max_steps, mesh_path=mesh_path, mpm_scale=self.mpm_scale,
                                         vertices=vertices, faces=faces)
        cfg.SIMULATOR.defrost()
        self.n_particles = cfg.SIMULATOR.n_particles = len(self.init_particles)
        self.simulator = MPMSimulator(cfg.SIMULATOR, self.primitive, self.env_dt, self.mpm_scale)
        self.cloth_simulator = KinematicCloth(self.primitive, self.substeps, self.env_dt, motion=motion, motion_grad=motion_grad)
        self.control_mode = getattr(cfg, "control_mode", "mpm")
        self.action_list = []
        self._is_copy = False

    def set_copy(self, is_copy: bool):                     # :40-41
        self._is_copy = is_copy

    def set_control_mode(self, mode):                      # :133-135
        assert mode in ("mpm", "cloth")
        self.control_mode = mode

    def initialize(self):                                  # :46-63
        self.primitive.initialize()
        self.simulator.initialize()
        self.cloth_simulator.initialize()
        self.simulator.reset(self.init_particles)
        self.simulator.get_contact_pair(0)
        self.action_list = []

    def step(self, action=None):                           # :86-106
        sim = self.simulator
        start = 0 if self._is_copy else sim.cur
        sim.cur = start + self.substeps
        mpm_action = action if self.control_mode == "mpm" else None
        cloth_action = action if self.control_mode == "cloth" else None
        self.action_list.append(action)
        for s in range(start, sim.cur):
            sim.substep(s, mpm_action)
            sim.get_contact_pair(s + 1)
            sim.trace_penetration_after_mpm(s + 1)
        self.cloth_simulator.step(start // self.substeps, cloth_action)
        sim.backup_contact_pair(sim.cur)
        sim.get_contact_pair(sim.cur)
        sim.trace_penetration_after_cloth(sim.cur)
        if self._is_copy:                                  # :92-95 copy to the first frame for rendering (the sheet's frames and the contact columns go along: smac_copy_frame)
            sim.copyframe(sim.cur, 0)
            sim.cur = 0

    def step_grad(self, action=None):                      # :108-127
        sim = self.simulator
        start = sim.cur
        sim.cur = start - self.substeps
        mpm_action = action if self.control_mode == "mpm" else None
        cloth_action_grad, ext_f_grad = self.cloth_simulator.step_grad(sim.cur // self.substeps)
        mpm_action_grad = None if action is None else np.zeros(np.asarray(action).shape)
        for s in range(start - 1, sim.cur - 1, -1):
            g = sim.substep_grad(s, action=mpm_action, ext_f_grad=ext_f_grad)
            if g is not None:
                mpm_action_grad += g
        if action is None:
            return None
        return mpm_action_grad if self.control_mode == "mpm" else cloth_action_grad

    forward = step

    def backward(self):                                    # :129-141
        total = self.simulator.cur // self.substeps
        grads = []
        for s in range(total - 1, -1, -1):
            grads = [self.step_grad(self.action_list[s])] + grads
        gx, gv = self.cloth_simulator.get_ext_state_grad(0)     # :126-128 what reaches the sheet's initial state through the first env step's frames
        self.cloth_simulator.dL_dx += gx
        self.cloth_simulator.dL_dv += gv
        return grads

    @property
    def cur(self):
        return self.simulator.cur
