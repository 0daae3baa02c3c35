"""End-to-end differentiable rollout through the reference's env surface (TaichiEnv + velocity-controlled rigid body,
SURVEY 8f-1): actions -> primitive velocities -> device forward kinematics -> forecast contact -> particles -> loss,
and back.  Checked against the oracle with torch.autograd through the whole chain."""
import numpy as np
import pytest
import torch

import helpers as H
from helpers import O

pytestmark = pytest.mark.gpu


def _scene(precision):
    from softmac_amd.config import CfgNode, get_cfg_defaults
    n_grid, N = 32, 1200
    state = H.make_cloud(N, n_grid, seed=41, lo=(0.36, 0.1, 0.36), hi=(0.64, 0.3, 0.64), v_std=0.1, F_std=5e-3)
    cfg = get_cfg_defaults()
    cfg.control_mode = "rigid"
    cfg.rigid_velocity_control = True
    cfg.env_dt = 4e-4
    S = cfg.SIMULATOR
    S.dt = 2e-4; S.E = 3e3; S.nu = 0.2; S.ptype = 0; S.material_model = 0; S.gravity = (0., -9.8, 0.)
    S.ground_friction = 20.; S.collision_type = 2; S.max_steps = 16; S.n_grid = n_grid; S.precision = precision
    cfg.SHAPES = [{"shape": "predefined", "state": state}]
    q = np.array([0.995, 0.02, 0.03, 0.09]); q /= np.linalg.norm(q)
    ang = 2 * np.arccos(q[0]); axis = q[1:] / np.linalg.norm(q[1:])
    pose = np.concatenate([axis * ang, [0.5, 0.3 + 0.15 - 0.003, 0.5]])          # exp-map rotation, position
    vel = np.array([0.1, 0.05, -0.2, 0.02, -0.3, 0.01])                          # w, v
    cfg.RIGID.init_state = tuple(np.concatenate([pose, vel]))
    return cfg, state, pose, vel


@pytest.mark.parametrize("precision,ts,tg", [("float64", 1e-9, 1e-7), ("float32", H.F32_TOL["state"], H.F32_TOL["grad"])])      # (measured 6.7e-7: profiles/r04_g_f32_bounds.txt)
def test_velocity_control_rollout_and_action_gradients(precision, ts, tg):
    from softmac_amd.config import CfgNode
    from softmac_amd.engine.primitive import Mesh, Primitives
    from softmac_amd.engine.taichi_env import TaichiEnv
    cfg, state, pose, vel = _scene(precision)
    palm = H.load_palm()
    pc = CfgNode(); pc.friction = 0.4; pc.enable_external_force = True; pc.urdf_path = ""
    mesh = Mesh(sdf=palm, cfg=pc, max_timesteps=cfg.SIMULATOR.max_steps, rigid_velocity_control=True)
    env = TaichiEnv(cfg, primitives=Primitives(primitives=[mesh]))
    mesh.friction[None] = 0.4
    K, n = 3, env.substeps
    assert n == 2
    rng = np.random.default_rng(8)
    actions = 0.3 * rng.standard_normal((K, 6))
    for k in range(K):
        env.step(torch.tensor(actions[k]))
    T = K * n
    r = rng.standard_normal((len(state), 3))
    env.simulator.clear_grads()
    env.simulator.add_grad(T, gx=r)
    got = env.backward().numpy()
    x_got = env.simulator.get_x(T)

    # ---- oracle: the same rollout in torch, autograd through kinematics, contact and MPM
    P = O.SimParams(n_grid=32, dt=2e-4, E=3e3, nu=0.2, ptype=0, material_model=0, gravity=(0., -9.8, 0.), ground_friction=20.,
                    collision_type=2, substeps=n)
    A = torch.tensor(actions, requires_grad=True)
    sim = env.rigid_simulator
    pos = torch.tensor(pose[3:]); rot = torch.tensor(sim.exp2quat(pose[:3]))
    x, v, C, F = O.state24_split(state)
    t = lambda a: torch.as_tensor(a, dtype=O.DT)
    for f in range(T):
        k = f // n
        vw = torch.tensor(vel) if k == 0 else A[k - 1]           # set_action(s+1, ...): action k drives env step k+1
        pv, pw = vw[3:], vw[:3]
        prim = O.RigidPrim(pos, rot, pv, pw, t(palm["sdf"]), t(palm["normal"]), t(palm["lower"]), t(palm["upper"]), float(palm["dx"]),
                           0.4, 666.0, True)
        x, v, C, F, _ = O.substep(x, v, C, F, P, [prim], f)
        pos, rot = O.forward_kinematics(pos, rot, pv, pw, P.dt)
    L = (x * torch.tensor(r)).sum()
    (gA,) = torch.autograd.grad(L, A)
    assert H.rel_err(x_got, x.detach().numpy()) < ts
    assert np.abs(gA[-1]).max() == 0 and np.abs(got[-1]).max() == 0          # the last action only sets the next step's velocities
    assert H.note(f"env action.grad {precision}", H.rel_err(got, gA.numpy()), tg) < tg, (got, gA)
    # pose reached through device forward kinematics
    st = mesh.get_state(T)
    assert np.abs(st[:3] - pos.detach().numpy()).max() < ts * 10 and np.abs(st[3:7] - rot.detach().numpy()).max() < max(ts * 10, 1e-6)


def test_loss_to_action_gradient_by_finite_differences():
    """The whole loop of demo_pour_vel.py (:76-111) - reset, K env steps, chamfer + pose + velocity loss at every env
    step inside the tape, env.backward() - against central differences of the total loss in the actions (f64,
    neo-Hookean so that no SVD-clamp convention enters, see tests/test_gpu_fullsize.py)."""
    import types
    from softmac_amd.config import CfgNode
    from softmac_amd.engine.losses import PourLoss
    from softmac_amd.engine.primitive import Mesh, Primitives
    from softmac_amd.engine.taichi_env import TaichiEnv
    cfg, state, pose, vel = _scene("float64")
    cfg.SIMULATOR.ptype = 1
    cfg.SIMULATOR.material_model = 1
    palm = H.load_palm()
    pc = CfgNode(); pc.friction = 0.4; pc.enable_external_force = True; pc.urdf_path = ""
    mesh = Mesh(sdf=palm, cfg=pc, max_timesteps=cfg.SIMULATOR.max_steps, rigid_velocity_control=True)
    env = TaichiEnv(cfg, primitives=Primitives(primitives=[mesh]))
    mesh.friction[None] = 0.4
    rng = np.random.default_rng(12)
    target = state[:, :3] + np.array([0.02, -0.03, 0.01]) + 0.002 * rng.standard_normal((len(state), 3))
    loss = PourLoss(types.SimpleNamespace(weight=(1.0, 0.5, 0.25), target_path=None), env.simulator)
    loss.set_target(target)
    loss.initialize()
    env.loss = loss
    K = 3
    actions = 0.3 * rng.standard_normal((K, 6))

    def rollout(acts, record):
        env.reset()
        env.simulator.clear_grads()
        total = 0.0
        ctx = loss.tape() if record else contextlib.nullcontext()
        with ctx:
            for k in range(K):
                env.step(torch.tensor(acts[k]))
                loss.clear()
                total += env.compute_loss()["loss"]
        return total

    import contextlib
    base = rollout(actions, True)
    grad = env.backward().numpy()
    assert np.abs(grad[:-1]).max() > 0
    for (k, c) in ((0, 4), (0, 1), (1, 3), (1, 2)):            # components of (w, v) of the first two actions
        eps = 1e-6
        ap, am = actions.copy(), actions.copy()
        ap[k, c] += eps; am[k, c] -= eps
        fd = (rollout(ap, False) - rollout(am, False)) / (2 * eps)
        assert abs(fd - grad[k, c]) < 2e-5 * max(abs(fd), 1e-3), (k, c, fd, grad[k, c])
    assert abs(rollout(actions, False) - base) < 1e-12 * abs(base)
