"""PourLoss / GripLoss with the reference's surface (/root/reference/softmac/engine/losses/loss_pour.py:6-169,
loss_grip.py:6-170), over the HIP engine.

    loss = chamfer_weight * chamfer(x[f], target) + pose_weight * pose(f) + velocity_weight * velocity(f)

* the chamfer term and its gradient run on the device (`smac_loss_chamfer`, uniform-grid nearest neighbours instead
  of the reference's two O(N^2) sweeps);
* the pose / velocity terms read 13 scalars of the controlled primitive and are evaluated on the host;
* there is no Taichi tape.  The reference wraps `env.compute_loss(f)` in `with ti.ad.Tape(loss=env.loss.loss)`, whose
  exit back-propagates d loss = 1 into `x.grad[f]` and the primitive's `position/rotation/v/w.grad[f]`
  (loss_pour.py:130-140).  Here `with env.loss.tape():` does the same: every `compute_loss(f)` inside the block also
  accumulates those seeds (the total is a plain sum over frames, so each term's seed does not depend on the others).
"""
import contextlib
import os
import weakref

import numpy as np

LIVE_LOSSES = weakref.WeakSet()          # what a `ti.ad.Tape(loss=env.loss.loss)` of the demos can be resolved against (compat/taichi)


class ScalarField(float):
    """`env.loss.loss` as the demos use it (demo_pour.py:171, 200): a number that also answers `.to_numpy()` / `[None]` like the reference's 0-d
    Taichi field, and knows the loss object it belongs to (`Tape(loss=field)` records on that object)."""
    owner = None

    def __new__(cls, value=0.0, owner=None):
        f = super().__new__(cls, value)
        f.owner = owner
        return f

    def to_numpy(self):
        return np.float64(self)

    def __getitem__(self, idx):
        return float(self)


class ChamferPoseLoss:
    def __init__(self, cfg, mpm_sim):
        self.cfg = cfg
        self.sim = mpm_sim
        self.dim = mpm_sim.dim
        self.n_particles = mpm_sim.n_particles
        self.rigid_control = mpm_sim.primitives[0] if len(mpm_sim.primitives) else None    # loss_pour.py:13
        self.chamfer_weight = self.pose_weight = self.velocity_weight = 0.0
        self.loss = ScalarField(0.0, self)
        self._recording = False
        self.target_x = None
        LIVE_LOSSES.add(self)

    # ---- reference surface ------------------------------------------------------------------
    def load_target_position(self, path):                                                    # :29-31
        pos = np.load(path if os.path.isabs(path) else os.path.join(os.getcwd(), path))
        self.set_target(pos)

    def set_target(self, pos):
        self.target_x = np.ascontiguousarray(pos, dtype=np.float64).reshape(-1, 3)
        self.sim.loss_set_target(self.target_x)

    def initialize(self):                                                                    # :33-39
        w = self.cfg.weight
        self.chamfer_weight, self.pose_weight, self.velocity_weight = float(w[0]), float(w[1]), float(w[2])
        path = getattr(self.cfg, "target_path", None)
        if path and self.target_x is None:
            self.load_target_position(path)

    def clear(self):                                                                         # :155-158
        self.loss = ScalarField(0.0, self)

    reset = clear

    @contextlib.contextmanager
    def tape(self):
        """Stands in for `ti.ad.Tape(loss=env.loss.loss)`: losses computed inside also seed their gradients."""
        self._recording = True
        try:
            yield self
        finally:
            self._recording = False

    # ---- terms --------------------------------------------------------------------------------
    def pose_terms(self, s13):
        """value and gradient (13,) of the pose penalty for one primitive state; loss_pour.py:76."""
        g = np.zeros(13)
        v = 10.0 * (s13[1] - 0.4) ** 2
        g[1] = 20.0 * (s13[1] - 0.4)
        return v, g

    def velocity_terms(self, s13):                                                           # :86-87
        g = np.zeros(13)
        v = float((s13[7:10] ** 2).sum() + 0.1 * (s13[10:13] ** 2).sum())
        g[7:10] = 2.0 * s13[7:10]
        g[10:13] = 0.2 * s13[10:13]
        return v, g

    def compute_loss(self, f):                                                               # :118-152
        chamfer = pose = vel = 0.0
        if self.chamfer_weight > 0:
            chamfer = self.sim.loss_chamfer(f, weight=self.chamfer_weight, add_grad=self._recording)
        if (self.pose_weight > 0 or self.velocity_weight > 0) and self.rigid_control is not None:
            s13 = self.rigid_control._get_state13(f)
            g = np.zeros(13)
            if self.pose_weight > 0:
                pose, gp = self.pose_terms(s13)
                g += self.pose_weight * gp
            if self.velocity_weight > 0:
                vel, gv = self.velocity_terms(s13)
                g += self.velocity_weight * gv
            if self._recording:
                self.rigid_control.add_state_grad(f, g)
        total = chamfer * self.chamfer_weight + pose * self.pose_weight + vel * self.velocity_weight
        self.loss = ScalarField(self.loss + total, self)       # the reference's loss field accumulates until clear()
        return {"loss": self.loss, "chamfer_loss": chamfer * self.chamfer_weight, "pose_loss": pose * self.pose_weight,
                "vel_loss": vel * self.velocity_weight}


class PourLoss(ChamferPoseLoss):
    pass


class GripLoss(ChamferPoseLoss):
    def pose_terms(self, s13):                                                               # loss_grip.py:74-79
        v, g = super().pose_terms(s13)
        a = abs(s13[3])
        sgn = 1.0 if s13[3] >= 0 else -1.0
        lo, hi = min(0.0, a - 0.5), max(0.0, a - 0.9)
        v += lo ** 2 + hi ** 2
        g[3] += (2.0 * lo * (1.0 if a - 0.5 < 0 else 0.0) + 2.0 * hi * (1.0 if a - 0.9 > 0 else 0.0)) * sgn
        return v, g
