"""DoorLoss / TransportLoss with the reference's surface (/root/reference/softmac/engine/losses/loss_door.py:6-140,
loss_transport.py:6-157):  loss = pose_weight * pose + velocity_weight * |v|^2 + contact_weight * sum_k min_dist_k^2,
where min_dist_k = min over the particles of controller k of max(|x_i - rigid.position[f]|^2 - 0.01, 0).

The min-distance term runs on the device (`smac_loss_min_dist`); pose / velocity terms read 13 scalars of the primitive
and are evaluated on the host.  `with loss.tape():` stands in for `ti.ad.Tape(loss=env.loss.loss)` as in loss_chamfer.py."""
import contextlib

import numpy as np

from .loss_chamfer import LIVE_LOSSES, ScalarField


class _ContactDistanceLoss:
    n_controllers = 1

    def __init__(self, cfg, mpm_sim):
        self.cfg = cfg
        self.sim = mpm_sim
        self.dim = mpm_sim.dim
        self.n_particles = mpm_sim.n_particles
        self.n_particles_per_controller = self.n_particles // self.n_controllers
        self.rigid = mpm_sim.primitives[0]
        self.pose_weight = self.velocity_weight = self.contact_weight = 0.0
        self.loss = ScalarField(0.0, self)
        self._recording = False
        LIVE_LOSSES.add(self)

    def initialize(self):
        w = self.cfg.weight
        self.pose_weight, self.velocity_weight, self.contact_weight = float(w[0]), float(w[1]), float(w[2])

    def clear(self):
        self.loss = ScalarField(0.0, self)

    reset = clear

    @contextlib.contextmanager
    def tape(self):
        self._recording = True
        try:
            yield self
        finally:
            self._recording = False

    def pose_terms(self, s13):
        raise NotImplementedError

    def compute_loss(self, f):
        s13 = self.rigid._get_state13(f)
        g = np.zeros(13)
        pose = vel = contact = 0.0
        if self.contact_weight > 0:
            n = self.n_particles_per_controller
            for k in range(self.n_controllers):
                d, gc = self.sim.loss_min_dist(f, k * n, (k + 1) * n, s13[:3], offset=0.01, weight=self.contact_weight,
                                               add_grad=self._recording)
                contact += d * d
                g[:3] += gc
        if self.velocity_weight > 0:
            vel = float((s13[7:10] ** 2).sum())
            g[7:10] += self.velocity_weight * 2.0 * s13[7:10]
        if self.pose_weight > 0:
            pose, gp = self.pose_terms(s13)
            g += self.pose_weight * gp
        if self._recording:
            self.rigid.add_state_grad(f, g)
        self.loss = ScalarField(self.loss + pose * self.pose_weight + vel * self.velocity_weight + contact * self.contact_weight, self)
        return {"loss": self.loss, "pose_loss": pose * self.pose_weight, "vel_loss": vel * self.velocity_weight,
                "contact_loss": contact * self.contact_weight}


class DoorLoss(_ContactDistanceLoss):                                   # loss_door.py:36-37
    def pose_terms(self, s13):
        g = np.zeros(13)
        c = np.cos(np.pi / 8)
        g[3] = 2.0 * (s13[3] - c)
        return float((s13[3] - c) ** 2), g


class TransportLoss(_ContactDistanceLoss):                              # loss_transport.py:12, 41-44
    n_controllers = 2

    def __init__(self, cfg, mpm_sim):
        super().__init__(cfg, mpm_sim)
        self.target = np.zeros(3)

    def set_target(self, target):
        self.target = np.asarray(target, dtype=np.float64).reshape(3)

    def pose_terms(self, s13):
        g = np.zeros(13)
        d = s13[:3] - self.target
        g[:3] = 2.0 * d
        return float((d ** 2).sum()), g
