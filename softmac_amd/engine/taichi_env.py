"""TaichiEnv - scene assembly and env-step orchestration with the reference's surface
(/root/reference/softmac/engine/taichi_env.py:15-162), over the HIP engine.  No Taichi is involved; the name is
kept so the demos' `from ...taichi_env import TaichiEnv` keeps working.

Built this round: velocity-controlled rigid bodies (`cfg.rigid_velocity_control = True`, reference
rigid_simulator_vel.py) - a complete Jade-free differentiable rollout.  The force-controlled path needs
nimblephysics ("Jade", closed third-party, absent) and raises.  The renderer is a stub."""
import numpy as np
import torch

from .mpm_simulator import MPMSimulator
from .primitive import Primitives
from .shapes import Shapes


class _NullRenderer:
    def initialize(self): pass
    def reset(self): pass
    def set_particles(self, x, colors): pass
    def set_primitives(self, f): pass
    def render(self): return None


class TaichiEnv:
    def __init__(self, cfg, primitives=None, loss=None):
        """cfg: CfgNode as produced by softmac_amd.config.load.  `primitives` lets tests / synthetic scenes pass a
        ready Primitives container instead of URDF-driven construction."""
        self.cfg = cfg.ENV
        cfg.defrost()
        self.env_dt = cfg.env_dt
        self.control_mode = cfg.control_mode
        assert self.control_mode in ("mpm", "rigid")
        self.rigid_velocity_control = cfg.rigid_velocity_control

        self.primitives = primitives if primitives is not None else Primitives(
            cfg.PRIMITIVES, max_timesteps=cfg.SIMULATOR.max_steps, rigid_velocity_control=self.rigid_velocity_control)
        self.shapes = Shapes(cfg.SHAPES)
        self.init_particles, self.particle_colors = self.shapes.get()
        self.n_particles = cfg.SIMULATOR.n_particles = len(self.init_particles)

        self.simulator = MPMSimulator(cfg.SIMULATOR, self.primitives, self.env_dt,
                                      rigid_velocity_control=self.rigid_velocity_control)
        self.substeps = self.simulator.substeps
        if self.rigid_velocity_control:
            from .rigid_simulator_vel import RigidSimulatorVelocityControl
            self.rigid_simulator = RigidSimulatorVelocityControl(cfg.RIGID, self.primitives, self.substeps, self.env_dt)
        else:
            raise NotImplementedError("force-controlled rigid bodies need nimblephysics (Jade), which is not available; "
                                      "use cfg.rigid_velocity_control = True")
        self.renderer = _NullRenderer()
        # the loss named by the config (reference :49-54 evaluates cfg.ENV.loss_type: PourLoss, GripLoss, DoorLoss, TransportLoss) or one handed in
        if loss is None and getattr(self.cfg, "loss_type", ""):
            from . import losses
            if not hasattr(losses, self.cfg.loss_type):
                raise ValueError(f"cfg.ENV.loss_type = {self.cfg.loss_type!r}: not one of {losses.__all__}")
            loss = getattr(losses, self.cfg.loss_type)(self.cfg.loss, self.simulator)
        self.use_loss = loss is not None
        self.loss = loss
        self._is_copy = False
        self.initialize()

    def set_copy(self, is_copy: bool):
        self._is_copy = is_copy

    def _parts(self):
        """what initialize / reset walk, in the reference's order (:64-82): primitives, particles, rigid bodies, renderer, loss"""
        parts = [self.primitives, self.simulator, self.rigid_simulator, self.renderer]
        return parts + ([self.loss] if self.loss else [])

    def initialize(self):
        for part in self._parts():
            part.initialize()
        self.reset()

    def reset(self):
        for part in self._parts():
            if part is self.simulator:
                part.reset(self.init_particles)
            else:
                part.reset()
        self.action_list = []

    def _route(self, action):
        """(particle action, rigid action): an env's action drives either the particle controllers or the rigid bodies (:95-96)"""
        return (action, None) if self.control_mode == "mpm" else (None, action)

    def render(self, f=None):
        return self.renderer.render()

    def step(self, action=None):                                 # :93-115
        start = 0 if self._is_copy else self.simulator.cur
        self.simulator.cur = start + self.substeps
        mpm_action, rigid_action = self._route(action)
        self.action_list.append(action)
        self.simulator.run_substeps(start, self.substeps, mpm_action)    # one FFI call for the env step's substeps (:101-102)
        self.rigid_simulator.step(start // self.substeps, rigid_action)
        if self._is_copy:
            self.simulator.copyframe(self.simulator.cur, 0)
            self.simulator.cur = 0

    def step_grad(self, action=None):                            # :117-137
        start = self.simulator.cur
        self.simulator.cur = start - self.substeps
        mpm_action, rigid_action = self._route(action)
        rigid_action_grad, ext_f_grad_list = self.rigid_simulator.step_grad(self.simulator.cur // self.substeps, rigid_action)
        # the reverse loop of :128-133 as ONE call: the library runs the env step's substeps back to back (no host round trip, no stream
        # sync per substep; in float32 it reverses substep f's P2G and substep f-1's G2P in one launch) and sums action.grad on the device
        tmp = self.simulator.run_substeps_grad(self.simulator.cur, self.substeps, ext_f_grad_list, mpm_action)
        if action is None:
            return None
        mpm_action_grad = tmp if tmp is not None else np.zeros(np.asarray(action).shape)
        return torch.tensor(mpm_action_grad) if self.control_mode == "mpm" else rigid_action_grad

    def backward(self):                                          # :139-151
        total_steps = self.simulator.cur // self.substeps
        if self.control_mode == "rigid" and self.rigid_velocity_control and total_steps > 0:
            # Velocity control: RigidSimulatorVelocityControl.step_grad only READS the primitives' adjoints of an env step's frames
            # (get_action_grad, rigid_simulator_vel.py:34-44) and hands no ext_f.grad back, so the reverse loop over env steps is one sweep over
            # all substeps followed by the reads - same numbers, no host round trip per env step, and the library's fused backward step is
            # not interrupted at every env-step boundary.  (step_grad remains for callers that walk the env steps themselves.)
            cur = self.simulator.cur
            self.simulator.run_substeps_grad(0, cur)
            self.simulator.cur = 0
            return self.rigid_simulator.step_grad_all(total_steps)
        action_grad = []
        for s in range(total_steps - 1, -1, -1):
            action_grad = [self.step_grad(self.action_list[s])] + action_grad
        return torch.vstack(action_grad)

    def compute_loss(self, f=None, **kwargs):
        assert self.loss is not None
        if f is None:
            if self._is_copy:
                self.loss.clear()                                # :156-158: copy mode starts the loss from zero
                f = 0
            else:
                f = self.simulator.cur
        return self.loss.compute_loss(f, **kwargs)
