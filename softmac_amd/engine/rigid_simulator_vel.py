"""RigidSimulatorVelocityControl - kinematic rigid bodies driven by 6-D velocity actions
(/root/reference/softmac/engine/rigid_simulator_vel.py:6-71; the reference's unused nimblephysics import is
dropped).  Poses advance inside MPMSimulator.substep through forward_kinematics on the device."""
import numpy as np
import torch


class RigidSimulatorVelocityControl:
    def __init__(self, cfg, primitives, substeps=20, env_dt=2e-3):
        self.cfg = cfg
        self.primitives = primitives
        self.n_primitive = len(self.primitives)
        self.substeps = substeps
        self.max_steps = 2048 // substeps
        self.gravity = cfg.gravity
        self.dt = env_dt
        assert len(cfg.init_state) == 12 * self.n_primitive
        self.init_state = np.array(cfg.init_state, dtype=np.float64)

    def step(self, s, action):                                   # :20-32
        if self.n_primitive == 0:
            return
        for i in range(self.n_primitive):
            self.primitives[i].clear_ext_f()
            a = action[i * 6: i * 6 + 6]
            if isinstance(a, torch.Tensor):
                a = a.detach().cpu().numpy()
            self.primitives[i].set_action(s + 1, self.substeps, a)

    def step_grad(self, s, action=None):                         # :34-44
        if self.n_primitive == 0:
            return None, None
        g = np.zeros(self.n_primitive * 6)
        for i in range(self.n_primitive):
            g[i * 6: i * 6 + 6] = self.primitives[i].get_action_grad(s + 1, self.substeps)
        return torch.tensor(g), None

    def step_grad_all(self, total_steps):
        """step_grad of the env steps 0 .. total_steps - 1 at once: (total_steps, 6 n_primitive); one device round trip per primitive, not per env step"""
        g = np.zeros((total_steps, self.n_primitive * 6))
        for i in range(self.n_primitive):
            g[:, i * 6: i * 6 + 6] = self.primitives[i].get_action_grads(1, total_steps + 1, self.substeps)
        return torch.tensor(g)

    def exp2quat(self, e):                                       # :46-55
        mag = np.linalg.norm(e)
        if mag > 1e-10:
            q = np.zeros(4)
            q[0] = np.cos(mag / 2)
            q[1:] = e * np.abs(np.sin(mag / 2)) / mag
            return q
        return np.array([1., 0., 0., 0.])

    def initialize(self):
        pass

    def reset(self):                                             # :60-71
        n = self.n_primitive
        for i in range(n):
            state = np.zeros(13)
            pose = self.init_state[i * 6: i * 6 + 6]
            vel = self.init_state[i * 6 + 6 * n: i * 6 + 6 + 6 * n]
            state[:3] = pose[3:]
            state[3:7] = self.exp2quat(pose[:3])
            state[7:10] = vel[3:]
            state[10:] = vel[:3]
            self.primitives[i].set_all_states_range(0, self.substeps, state)     # one FFI call instead of 2*substeps launches
