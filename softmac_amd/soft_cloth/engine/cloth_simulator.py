"""Kinematic stand-in for the reference's ClothSimulator (soft_cloth/engine/cloth_simulator.py, which wraps the closed-source
DiffClothAI).  Same place in the env loop and the same calls on the primitive: once per env step it reads and clears the contact
force, produces the sheet's next vertex positions / velocities and writes them to the primitive's frames of the next env step
(:60-81).  Here the motion is prescribed: `motion(step_index, x, v, action, ext_f) -> (x_new, v_new)`.  The backward pass hands the
adjoints that reach the sheet (position.grad / velocity.grad summed over the env step's frames, :126-136) to `motion_grad` if given."""
from __future__ import annotations

import numpy as np


class KinematicCloth:
    def __init__(self, primitive, substeps=20, env_dt=2e-3, x_init=None, v_init=None, motion=None, motion_grad=None):
        self.primitive = primitive
        self.substeps = substeps
        self.dt = env_dt
        self.x_init = np.array(primitive.rest_vertices if x_init is None else x_init, dtype=np.float64)
        self.v_init = np.zeros_like(self.x_init) if v_init is None else np.array(v_init, dtype=np.float64)
        self.motion = motion or (lambda idx, x, v, action, ext_f: (x + env_dt * v, v))
        self.motion_grad = motion_grad
        self.x = self.v = None
        self.ext_f_log = []
        self.dL_dx = np.zeros_like(self.x_init)            # adjoint of the sheet's state carried backwards (cloth_simulator.py:83-86, 127-128 of taichi_env.py)
        self.dL_dv = np.zeros_like(self.x_init)

    def initialize(self):                                  # :129-135
        self.x, self.v = self.x_init.copy(), self.v_init.copy()
        self.ext_f_log = []
        self.dL_dx[:] = 0.0
        self.dL_dv[:] = 0.0
        self.primitive.set_all_states(0, self.x, self.v, f_end=self.substeps + 1)

    def step(self, s, action=None):                        # :60-81
        idx = s + 1
        ext_f = self.primitive.ext_f.to_numpy() / self.substeps
        self.primitive.clear_ext_f()
        self.ext_f_log.append(ext_f)
        self.x, self.v = self.motion(idx, self.x, self.v, action, ext_f)
        first, last = idx * self.substeps, min((idx + 1) * self.substeps + 1, self.primitive.max_timesteps)   # (the reference's fields hold 2048 frames)
        if first < last:
            self.primitive.set_all_states(first, self.x, self.v, f_end=last)

    def get_ext_state_grad(self, s):                       # :126-136
        gx, gv = np.zeros_like(self.x_init), np.zeros_like(self.x_init)
        for j in range(s * self.substeps, (s + 1) * self.substeps):
            a, b = self.primitive.get_all_states_grad(j)
            gx += a
            gv += b
        return gx, gv

    def step_grad(self, idx):
        gx, gv = self.get_ext_state_grad(idx + 1)
        if self.motion_grad is not None:
            return self.motion_grad(idx, gx, gv)
        return None, None

    def get_observation(self):
        return np.concatenate([self.x.reshape(-1), self.v.reshape(-1)])
