"""Checkpoint-every-K state frames with recompute, on top of the C ABI (no reference counterpart: the reference keeps the whole
trajectory resident - 2048 frames of its fields, mpm_simulator.py:60-67 - and rolls over a 2-frame window only without gradients, taichi_env.py `_is_copy`).

A handle holds `max_steps` state frames of 96 B per particle (float32).  An episode of T substeps needs T + 1 of them when it is fully resident; here it
needs `window + 1` working frames plus one frame per window:

    slots 0 .. K            the window being simulated (logical substep t = w K + j lives in slot j)
    slot  K + 1 + w         the state the w-th window started from (a device copy: smac_copy_frame)

forward():   per window - file slot 0, upload the primitives' states of the window, smac_substeps(0, K), copy slot K to slot 0.
backward():  per window, last first - bring the filed state back to slot 0, run the window forward again (which also re-files the grid checkpoints
             and hit lists the backward kernels restore from), add the loss seeds of the window's frames and the adjoint handed down by the window
             after it to the working slots, smac_substeps_grad(0, K); the adjoint of slot 0 goes to the window before.

Cost: one extra forward pass (+ 40 % of a forward + backward pair) and three frame-sized device copies per window (the state in and out of its file,
the adjoint's hand-down: smac_carry_grad keeps the adjoint of slot 0 as the adjoint of the previous window's last slot, particle order tag included).
`window` must be a multiple of the env step's substeps: the forecast contact's `life` (mpm_simulator.py:425) follows the substep's phase inside its env step.
Primitive states are prescribed per logical frame (`prim_state(t)` -> one 13-vector per primitive); velocity-controlled episodes keep using TaichiEnv."""
from __future__ import annotations

import numpy as np


class WindowedEpisode:
    def __init__(self, sim, window, prim_state=None):
        assert window >= 1 and window % max(sim.substeps, 1) == 0, "window must be a multiple of the env step's substeps"
        self.sim, self.K, self.prim_state = sim, int(window), prim_state
        self.windows = []                     # substeps of each simulated window (all K except possibly the last)
        assert sim.max_steps >= self.K + 3, "the handle needs window + 1 working frames and at least two filed ones"

    # ---- helpers
    def _file_slot(self, w):
        slot = self.K + 1 + w
        assert slot < self.sim.max_steps, f"max_steps = {self.sim.max_steps} holds {self.sim.max_steps - self.K - 1} windows of {self.K} substeps"
        return slot

    def _upload_primitives(self, t0, n):
        if self.prim_state is None:
            return
        traj = np.array([self.prim_state(t0 + j) for j in range(n + 1)], dtype=np.float64)     # (n + 1, primitives, 13)
        for i, m in enumerate(self.sim.primitives):
            m.set_states_trajectory(0, traj[:, i])

    @property
    def T(self):
        return sum(self.windows)

    # ---- the episode
    def reset(self, state):
        self.sim.reset(state)
        self.windows = []

    def forward(self, n_substeps):
        """n_substeps more substeps (whole windows, a shorter one at the end); the state after them is in slot `self.end_slot`"""
        assert not self.windows or self.windows[-1] == self.K, "the episode already ended on a short window"
        left = int(n_substeps)
        while left > 0:
            n = min(self.K, left)
            w = len(self.windows)
            if w > 0:
                self.sim.copyframe(self.K, 0)              # (the previous window was a full one)
            self.sim.copyframe(0, self._file_slot(w))
            self._upload_primitives(w * self.K, n)
            self.sim.run_substeps(0, n)
            self.windows.append(n)
            left -= n
        return self.end_slot

    @property
    def end_slot(self):
        return self.windows[-1] if self.windows else 0

    def get_state(self):
        return self.sim.get_state(self.end_slot)

    def backward(self, seeds):
        """seeds: {logical frame t: dict(gx=, gv=, gF=, gC=)} (any subset).  Returns (adjoint of frame 0 as (gx, gv, gF, gC),
        {t: [13-vector per primitive]} adjoints of the prescribed primitive states of every frame)."""
        sim, K = self.sim, self.K
        carried = False
        prim_grads = {}
        for w in range(len(self.windows) - 1, -1, -1):
            n, t0 = self.windows[w], w * K
            sim.copyframe(self._file_slot(w), 0)
            self._upload_primitives(t0, n)
            sim.run_substeps(0, n)                          # recompute: states, grid checkpoints and hit lists of this window
            if carried:
                sim.carry_grad(0, n)                        # the adjoint the later window left on its first frame seeds this window's last (on the device)
            else:
                sim.clear_grads()
            for t, g in seeds.items():
                if t0 < t <= t0 + n or (t == 0 and w == 0):
                    sim.add_grad(t - t0, **g)
            sim.run_substeps_grad(0, n)
            carried = True
            pg = [m.get_states_grad_trajectory(0, n) for m in sim.primitives]
            for j in range(n):
                prim_grads[t0 + j] = [g[j] for g in pg]
        return sim.get_grad_full(0), prim_grads
