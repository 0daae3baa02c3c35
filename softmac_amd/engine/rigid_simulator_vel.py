"""Kinematic rigid bodies driven by 6-D velocity actions - the Jade-free coupling of `demo_pour_vel.py`.

Surface kept from the reference's `RigidSimulatorVelocityControl` (/root/reference/softmac/engine/rigid_simulator_vel.py:6-71), which
`TaichiEnv` drives: `step(s, action)` hands env step s's action (6 numbers per body: angular then linear velocity) to the NEXT env step's
substep frames and clears the wrench accumulators (:20-32); `step_grad(s)` reads the adjoint of that action back (:34-44); `reset()` turns
the config's `init_state` - per body an exponential-map rotation + position, then all bodies' angular + linear velocities - into the
13-scalar device state `pos3 quat4 v3 w3` of the first env step's frames (:60-71).  Poses advance inside `MPMSimulator.substep`
(`forward_kinematics` on the device).  What is this build's: the layout is decoded once into (n, 6) tables, the device is written one
range per body instead of one call per substep frame, and `step_grad_all` returns a whole episode's action gradients in one round trip per body.
"""
from __future__ import annotations

import numpy as np
import torch

ACTION_DOF = 6                       # per body: w3 then v3 (primitive_base.py:298-304)


def exp_map_to_quat(e):
    """rotation vector -> unit quaternion (w, x, y, z); the reference's `exp2quat` (:46-55), identity below 1e-10"""
    e = np.asarray(e, dtype=np.float64)
    angle = float(np.linalg.norm(e))
    if angle <= 1e-10:
        return np.array([1.0, 0.0, 0.0, 0.0])
    return np.concatenate([[np.cos(angle / 2)], e * (abs(np.sin(angle / 2)) / angle)])


def decode_init_state(init_state, n):
    """`init_state` = n x (rotvec3, pos3) followed by n x (w3, v3)  ->  (n, 13) device states"""
    flat = np.asarray(init_state, dtype=np.float64)
    if flat.size != 2 * ACTION_DOF * n:
        raise AssertionError(f"init_state holds {flat.size} numbers, 12 per primitive expected ({n} primitives)")
    pose, vel = flat[:ACTION_DOF * n].reshape(n, ACTION_DOF), flat[ACTION_DOF * n:].reshape(n, ACTION_DOF)
    out = np.zeros((n, 13))
    out[:, 0:3] = pose[:, 3:]
    out[:, 3:7] = [exp_map_to_quat(r) for r in pose[:, :3]] if n else np.zeros((0, 4))
    out[:, 7:10] = vel[:, 3:]
    out[:, 10:13] = vel[:, :3]
    return out


class RigidSimulatorVelocityControl:
    def __init__(self, cfg, primitives, substeps=20, env_dt=2e-3):
        self.cfg, self.primitives = cfg, primitives
        self.n_primitive = len(primitives)
        self.substeps, self.dt = substeps, env_dt
        self.max_steps = 2048 // substeps
        self.gravity = cfg.gravity
        self.init_state = np.array(cfg.init_state, dtype=np.float64)
        self._states0 = decode_init_state(self.init_state, self.n_primitive)

    def _bodies(self):
        return ((i, self.primitives[i], slice(i * ACTION_DOF, (i + 1) * ACTION_DOF)) for i in range(self.n_primitive))

    def step(self, s, action):
        for _, body, dof in self._bodies():
            body.clear_ext_f()
            a = action[dof]
            body.set_action(s + 1, self.substeps, a.detach().cpu().numpy() if isinstance(a, torch.Tensor) else a)

    def step_grad(self, s, action=None):
        if self.n_primitive == 0:
            return None, None
        grad = np.zeros(self.n_primitive * ACTION_DOF)
        for _, body, dof in self._bodies():
            grad[dof] = body.get_action_grad(s + 1, self.substeps)
        return torch.tensor(grad), None

    def step_grad_all(self, total_steps):
        """step_grad of env steps 0 .. total_steps - 1 at once: (total_steps, 6 n_primitive)"""
        grad = np.zeros((total_steps, self.n_primitive * ACTION_DOF))
        for _, body, dof in self._bodies():
            grad[:, dof] = body.get_action_grads(1, total_steps + 1, self.substeps)
        return torch.tensor(grad)

    def exp2quat(self, e):
        return exp_map_to_quat(e)

    def initialize(self):
        pass

    def reset(self):
        for i, body, _ in self._bodies():
            body.set_all_states_range(0, self.substeps, self._states0[i])
