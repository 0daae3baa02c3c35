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


def loss_seeds(loss, frames):
    """`seed_fn` for WindowedEpisode.backward / WindowedEnvEpisode.backward out of one of engine/losses' loss objects: the reference's

        with ti.ad.Tape(loss=env.loss.loss):            # demo_pour.py:171-176
            for f in frames: env.compute_loss(f)

    evaluated window by window on the way back - a frame of the episode only exists (again) while its window is being reversed.  Logical frame t of the
    window [t0, t0 + n] lives in slot t - t0; frame t0 belongs to the window before it (where it is slot n of that window), except frame 0.  The loss value
    accumulates over the windows as it does over the reference's frames: read `loss.loss` after backward()."""
    frames = sorted(int(t) for t in frames)

    def seed(t0, n):
        with loss.tape():
            for t in frames:
                if t0 < t <= t0 + n or (t == 0 and t0 == 0):
                    loss.compute_loss(t - t0)
    return seed


class WindowedEpisode:
    """`particle_action(e)` (optional; control_mode "mpm", the door demo's controller - mpm_simulator.py:208-213, 579-602): the (n_control, 3) action
    held over env step e (substeps e n .. e n + n - 1, n = sim.substeps, as TaichiEnv.step holds it: taichi_env.py:99-102).  backward() then also returns
    {e: action.grad summed over the env step's substeps} - what TaichiEnv.step_grad accumulates (taichi_env.py:128-133)."""

    def __init__(self, sim, window, prim_state=None, particle_action=None):
        assert window >= 1 and window % max(sim.substeps, 1) == 0, "window must be a multiple of the env step's substeps"
        self.sim, self.K, self.prim_state, self.particle_action = sim, int(window), prim_state, particle_action
        self.windows = []                     # substeps of each simulated window (all K except possibly the last)
        assert sim.max_steps >= self.K + 3, "the handle needs window + 1 working frames and at least two filed ones"
        assert particle_action is None or sim.n_control > 0, "particle actions need a handle with n_control > 0"

    def _run(self, t0, n):
        """substeps t0 .. t0 + n - 1 of the episode, in slots 0 .. n"""
        if self.particle_action is None:
            self.sim.run_substeps(0, n)
            return
        m = max(self.sim.substeps, 1)
        for j in range(0, n, m):                            # one batched call per env step: the action is constant inside it
            self.sim.run_substeps(j, min(m, n - j), self.particle_action((t0 + j) // m))

    def _run_grad(self, t0, n, action_grads):
        if self.particle_action is None:
            self.sim.run_substeps_grad(0, n)
            return
        m = max(self.sim.substeps, 1)
        for j in range(((n - 1) // m) * m, -1, -m):
            e = (t0 + j) // m
            action_grads[e] = self.sim.run_substeps_grad(j, min(m, n - j), action=self.particle_action(e))

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
            self._run(w * self.K, n)
            self.windows.append(n)
            left -= n
        return self.end_slot

    @property
    def end_slot(self):
        return self.windows[-1] if self.windows else 0

    def get_state(self):
        return self.sim.get_state(self.end_slot)

    def backward(self, seeds, seed_fn=None):
        """seeds: {logical frame t: dict(gx=, gv=, gF=, gC=)} (any subset).  seed_fn(t0, n) (optional) is called once per window, after its frames
        [t0, t0 + n] have been recomputed into slots 0 .. n and before they are reversed: a loss evaluated on the device adds its gradients there
        (e.g. `smac_loss_chamfer(slot, add_grad=1)` through engine/losses: logical frame t lives in slot t - t0).  Returns (adjoint of frame 0 as
        (gx, gv, gF, gC), {t: [13-vector per primitive]} adjoints of the prescribed primitive states of every frame) and, with particle actions,
        {env step: action gradient} as a third value."""
        sim, K = self.sim, self.K
        carried = False
        prim_grads, action_grads = {}, {}
        for w in range(len(self.windows) - 1, -1, -1):
            n, t0 = self.windows[w], w * K
            sim.copyframe(self._file_slot(w), 0)
            self._upload_primitives(t0, n)
            self._run(t0, n)                                # recompute: states, grid checkpoints and hit lists of this window
            if carried:
                sim.carry_grad(0, n)                        # the adjoint the later window left on its first frame seeds this window's last (on the device)
            else:
                sim.clear_grads()
            for t, g in seeds.items():
                if t0 < t <= t0 + n or (t == 0 and w == 0):
                    sim.add_grad(t - t0, **g)
            if seed_fn is not None:
                seed_fn(t0, n)
            self._run_grad(t0, n, action_grads)
            carried = True
            pg = [m.get_states_grad_trajectory(0, n + 1) for m in sim.primitives]
            for j in range(n + 1):
                # (slot n is the next window's slot 0: that window has filed the adjoint its substeps left on the frame, this one adds what a loss seeded on it)
                row = [np.asarray(g[j], dtype=np.float64) for g in pg]
                prim_grads[t0 + j] = [a + b for a, b in zip(prim_grads[t0 + j], row)] if (t0 + j) in prim_grads else row
        if self.particle_action is not None:
            return sim.get_grad_full(0), prim_grads, action_grads
        return sim.get_grad_full(0), prim_grads


class WindowedEnvEpisode:
    """The same for the reference's env loop with velocity-controlled primitives (TaichiEnv.step / backward, taichi_env.py:93-151): an episode of any
    number of env steps in a handle that holds `window` of them.

    Working frames 0 .. (window + 1) n - 1 (n = substeps per env step; an action sets the velocities of the NEXT env step's n frames, primitive_base.py:
    298-304, so the last env step of a window writes one env step past its end); the w-th window's first frame is filed at slot (window + 1) n + w n
    (smac_copy_frame moves the particles and the primitives' n frames from there on, as the reference's copyframe does).

    step(action):  TaichiEnv.step; when the window is full its last frame - particles, poses, the velocities the last action set - becomes frame 0 first.
    backward(seeds): per window, last first: filed frame back to slot 0, the window's actions stepped again, the later window's particle adjoint
        (smac_carry_grad) and POSE adjoint of its first frame handed to this window's last frame, the window's loss seeds, one backward sweep.  The
        velocity adjoints of a window's first n frames belong to the action given in the LAST env step of the window before it."""

    def __init__(self, env, window):
        sim = env.simulator
        assert env.control_mode == "rigid" and env.rigid_velocity_control, "velocity-controlled primitives only (the reference's RigidSimulatorVelocityControl)"
        self.env, self.sim, self.W, self.n = env, sim, int(window), int(env.substeps)
        assert self.W >= 1
        self.first_file = (self.W + 1) * self.n
        assert sim.max_steps >= self.first_file + 2 * self.n, "the handle needs (window + 1) env steps of working frames and room for filed frames"
        self.actions, self.windows = [], []

    def _file_slot(self, w):
        slot = self.first_file + w * self.n
        assert slot + self.n <= self.sim.max_steps, f"max_steps = {self.sim.max_steps} holds {(self.sim.max_steps - self.first_file) // self.n} windows"
        return slot

    def reset(self):
        self.env.reset()
        self.actions, self.windows = [], []

    def step(self, action):
        sim = self.sim
        if not self.windows or self.windows[-1] == self.W:
            if self.windows:
                sim.copyframe(self.W * self.n, 0)          # the full window's last frame opens the next one
            sim.copyframe(0, self._file_slot(len(self.windows)))
            self.windows.append(0)
            sim.cur = 0
            self.env.action_list = []
        self.env.step(action)
        self.actions.append(action)
        self.windows[-1] += 1

    @property
    def frame(self):
        """slot of the episode's current last frame"""
        return self.sim.cur

    def backward(self, seeds, seed_fn=None):
        """seeds: {logical frame t (in substeps): dict(gx=, gv=, gF=, gC=)}; seed_fn(t0, n): as in WindowedEpisode.backward (the place for a loss
        that seeds on the device).  Returns the action gradients, (env steps, 6 x primitives)."""
        import torch
        env, sim, n = self.env, self.sim, self.n
        prims = list(sim.primitives)
        out = np.zeros((len(self.actions), 6 * len(prims)))
        carried, pose_carry = False, None
        k_end = len(self.actions)
        for w in range(len(self.windows) - 1, -1, -1):
            nw = self.windows[w]
            k0 = k_end - nw
            sim.copyframe(self._file_slot(w), 0)
            sim.cur = 0
            env.action_list = []
            for k in range(nw):                            # recompute: states, checkpoints, the primitives' trajectories of this window
                env.step(self.actions[k0 + k])
            n_sub = nw * n
            if carried:
                sim.carry_grad(0, n_sub)
            else:
                sim.clear_grads()
            if pose_carry is not None:
                for m, g in zip(prims, pose_carry):
                    m.add_state_grad(n_sub, g)
            for t, g in seeds.items():
                if k0 * n < t <= k0 * n + n_sub or (t == 0 and w == 0):
                    sim.add_grad(t - k0 * n, **g)
            if seed_fn is not None:
                seed_fn(k0 * n, n_sub)
            sim.run_substeps_grad(0, n_sub)
            sim.cur = 0
            for i, m in enumerate(prims):                  # env step s of this window carries the action given one env step earlier
                g = m.get_action_grads(0, nw, n)
                for s in range(nw):
                    if k0 + s - 1 >= 0:
                        out[k0 + s - 1, 6 * i: 6 * i + 6] += g[s]
            pose_carry = []
            for m in prims:
                g13 = m.get_states_grad_trajectory(0, 1)[0].copy()
                g13[7:] = 0.0                              # (the velocity adjoints of this frame are in the action gradient above)
                pose_carry.append(g13)
            carried = True
            k_end = k0
        return torch.tensor(out)
