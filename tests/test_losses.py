"""Device chamfer loss + PourLoss/GripLoss mirrors (reference losses/loss_pour.py, loss_grip.py).

CPU: the numpy oracle (oracle/loss_oracle.py) against finite differences and scipy's k-d tree; the host-side pose /
velocity penalties against finite differences.
GPU: smac_loss_chamfer against the oracle - overlapping clouds, clouds far apart (coarse-grid path), exact ties on a
lattice (the reference's first-minimum rule), seeds added on top of existing adjoints after a re-sort, both
precisions - and the tape stand-in seeding x.grad and the primitive's state adjoints."""
import numpy as np
import pytest

import helpers as H
from oracle import loss_oracle as L
from softmac_amd.engine.losses import GripLoss, PourLoss
from softmac_amd.engine.losses.loss_chamfer import ChamferPoseLoss


def _clouds(n, m, seed, shift=(0.0, 0.0, 0.0), spread=0.2):
    rng = np.random.default_rng(seed)
    x = 0.4 + spread * rng.random((n, 3))
    t = 0.4 + spread * rng.random((m, 3)) + np.asarray(shift)
    return x, t


def test_oracle_chamfer_gradient_by_finite_differences():
    x, t = _clouds(300, 280, 0)
    loss, g, _, _ = L.chamfer(x, t)
    rng = np.random.default_rng(1)
    d = rng.standard_normal(x.shape)
    eps = 1e-7
    fd = (L.chamfer(x + eps * d, t)[0] - L.chamfer(x - eps * d, t)[0]) / (2 * eps)
    assert abs(fd - (g * d).sum()) < 1e-6 * abs(fd)


def test_oracle_brute_force_agrees_with_kdtree():
    x, t = _clouds(3000, 2500, 2, shift=(0.05, 0, 0))
    a, b = L.chamfer(x, t), L.chamfer_kdtree(x, t)
    assert abs(a[0] - b[0]) < 1e-12 * a[0] and np.abs(a[1] - b[1]).max() < 1e-12
    assert (a[2] == b[2]).all() and (a[3] == b[3]).all()


def test_oracle_first_minimum_rule():
    x = np.array([[0.5, 0.5, 0.5]])
    t = np.array([[0.6, 0.5, 0.5], [0.4, 0.5, 0.5], [0.5, 0.6, 0.5]])          # three targets at the same distance
    _, g, nn_cur, _ = L.chamfer(x, t)
    assert nn_cur[0] == 0


@pytest.mark.parametrize("cls", [PourLoss, GripLoss])
def test_pose_and_velocity_penalties_by_finite_differences(cls):
    obj = cls.__new__(cls)
    rng = np.random.default_rng(3)
    for trial in range(6):
        s = rng.standard_normal(13)
        s[3] = [0.3, -0.3, 0.7, -0.95, 0.97, 0.6][trial]                         # the three branches of |q_w|
        for fn in (obj.pose_terms, obj.velocity_terms):
            v, g = fn(s)
            for k in range(13):
                e = np.zeros(13); e[k] = 1e-6
                fd = (fn(s + e)[0] - fn(s - e)[0]) / 2e-6
                assert abs(fd - g[k]) < 1e-6 * max(1.0, abs(g[k]))


# ------------------------------------------------------------------------------------------- GPU
def _sim(n, precision, n_grid=64, max_steps=8, **kw):
    cfg = H.sim_cfg(n, n_grid=n_grid, precision=precision, max_steps=max_steps, **kw)
    sim, prims = H.build_engine(cfg, 2e-3)
    return cfg, sim, prims


@pytest.mark.gpu
@pytest.mark.parametrize("precision,tol", [("float64", 1e-12), ("float32", 2e-5)])   # f32: thousands of targets add onto one particle
@pytest.mark.parametrize("shift", [(0.0, 0.0, 0.0), (0.01, -0.02, 0.0), (0.35, 0.3, -0.3)])
def test_device_chamfer_vs_oracle(precision, tol, shift):
    n, m = 20000, 17000
    x, t = _clouds(n, m, 5, shift=shift)
    cfg, sim, _ = _sim(n, precision)
    sim.reset(x)
    sim.loss_set_target(t)
    xs = sim.get_x(0)                                    # what the device holds (rounded in f32 mode)
    ref, g_ref, _, _ = L.chamfer_kdtree(xs, t)
    got = sim.loss_chamfer(0, weight=0.7, add_grad=True)
    assert abs(got - ref) < 1e-10 * ref
    gx, _ = sim.get_grad(0)
    assert H.rel_err(gx, 0.7 * g_ref) < tol
    assert abs(sim.loss_chamfer(0) - got) < 1e-12 * got   # value only: no second seed (sum order differs)
    assert H.rel_err(sim.get_grad(0)[0], 0.7 * g_ref) < tol


@pytest.mark.gpu
def test_device_chamfer_first_minimum_on_a_lattice():
    g = np.arange(8) / 64.0 + 0.3
    x = np.stack(np.meshgrid(g, g, g, indexing="ij"), -1).reshape(-1, 3)
    t = x + np.array([0.5 / 64.0, 0.0, 0.0])            # every particle has two targets at exactly the same distance
    rng = np.random.default_rng(0)
    x, t = x[rng.permutation(len(x))], t[rng.permutation(len(t))]
    cfg, sim, _ = _sim(len(x), "float64")
    sim.reset(x)
    sim.loss_set_target(t)
    ref, g_ref, _, _ = L.chamfer(x, t)
    got = sim.loss_chamfer(0, weight=1.0, add_grad=True)
    assert abs(got - ref) < 1e-12 * ref
    assert np.abs(sim.get_grad(0)[0] - g_ref).max() < 1e-12


@pytest.mark.gpu
def test_chamfer_seed_on_top_of_existing_adjoints_after_resort():
    n = 6000
    state = H.make_cloud(n, 64, seed=9, lo=(0.35, 0.35, 0.35), hi=(0.65, 0.65, 0.65))
    cfg, sim, _ = _sim(n, "float64", sort_interval=2)
    sim.reset(state)
    sim.run_substeps(0, 5)                               # frames 1..5, re-binned twice on the way
    _, t = _clouds(n, 5000, 11)
    sim.loss_set_target(t)
    rng = np.random.default_rng(2)
    seed = rng.standard_normal((n, 3))
    sim.clear_grads()
    sim.add_grad(5, gx=seed)                             # stored in the caller's order ...
    val = sim.loss_chamfer(5, weight=2.0, add_grad=True)  # ... the device adds in the frame's own order
    x5 = sim.get_x(5)
    ref, g_ref, _, _ = L.chamfer_kdtree(x5, t)
    assert abs(val - ref) < 1e-10 * ref
    assert np.abs(sim.get_grad(5)[0] - (seed + 2.0 * g_ref)).max() < 1e-10
    sim.substep_grad(4)                                  # and the backward pass consumes it
    assert np.isfinite(sim.get_grad(4)[0]).all()


@pytest.mark.gpu
def test_pour_loss_tape_seeds_particles_and_primitive():
    import types
    n = 4000
    state = H.make_cloud(n, 64, seed=4)
    d = H.load_palm()
    spec = dict(d, friction=0.5, softness=666.0, contact=True)
    s13 = np.array([0.5, 0.9, 0.5, 1, 0, 0, 0, 0.1, -0.2, 0.05, 0.3, 0.2, -0.4], dtype=np.float64)
    cfg = H.sim_cfg(n, n_grid=64, precision="float64", max_steps=8)
    sim, prims = H.build_engine(cfg, 2e-3, [spec], [[s13]] * 8)
    sim.reset(state)
    sim.run_substeps(0, 2)
    _, t = _clouds(n, n, 6, shift=(0.1, 0, 0))
    loss = PourLoss(types.SimpleNamespace(weight=(1.5, 0.3, 0.2), target_path=None), sim)
    loss.set_target(t)
    loss.initialize()
    sim.clear_grads()
    plain = loss.compute_loss(2)                          # outside the tape: values only
    assert np.abs(sim.get_grad(2)[0]).max() == 0
    loss.clear()
    with loss.tape():
        info = loss.compute_loss(2)
    ref, g_ref, _, _ = L.chamfer_kdtree(sim.get_x(2), t)
    assert abs(info["chamfer_loss"] - 1.5 * ref) < 1e-10 * ref and abs(info["loss"] - plain["loss"]) < 1e-12
    assert abs(info["pose_loss"] - 0.3 * 10 * (0.9 - 0.4) ** 2) < 1e-12
    assert abs(info["vel_loss"] - 0.2 * ((s13[7:10] ** 2).sum() + 0.1 * (s13[10:] ** 2).sum())) < 1e-12
    assert np.abs(sim.get_grad(2)[0] - 1.5 * g_ref).max() < 1e-10
    gp = prims[0].get_all_states_grad(2)
    expect = np.zeros(13)
    expect[1] = 0.3 * 20 * (0.9 - 0.4)
    expect[7:10] = 0.2 * 2 * s13[7:10]
    expect[10:] = 0.2 * 0.2 * s13[10:]
    assert np.abs(gp - expect).max() < 1e-12


@pytest.mark.gpu
def test_fullsize_chamfer_far_and_near():
    """1M particles against 1M targets: the case the reference's O(N^2) kernels cannot run."""
    import time
    from softmac_amd import scenes
    cfg, env_dt, state, specs, s13 = scenes.s_grip(1 << 20, 128, max_steps=4, precision="float32")
    sim, _ = H.build_engine(cfg, env_dt)
    sim.reset(state)
    xs = sim.get_x(0)
    for shift in ((0.004, 0.002, -0.003), (0.0, 0.45, 0.0)):
        t = state[:, :3] + np.asarray(shift)
        sim.loss_set_target(t)
        sim.clear_grads()
        t0 = time.time()
        got = sim.loss_chamfer(0, weight=1.0, add_grad=True)
        dt = time.time() - t0
        ref, g_ref, _, _ = L.chamfer_kdtree(xs, t)
        assert abs(got - ref) < 1e-9 * ref, (shift, got, ref)
        assert H.rel_err(sim.get_grad(0)[0], g_ref) < 5e-6
        assert dt < 20.0


@pytest.mark.gpu
@pytest.mark.parametrize("precision,tol", [("float64", 1e-12), ("float32", 2e-6)])
def test_door_and_transport_losses(precision, tol):
    """min-distance contact term on the device + host pose / velocity terms against numpy, including the seeds the tape
    stand-in leaves in x.grad[f] and in the primitive's state adjoint; after two re-sorts, so ids and slots differ."""
    import types
    from softmac_amd.engine.losses import DoorLoss, TransportLoss
    n = 5000
    state = H.make_cloud(n, 64, seed=21, lo=(0.35, 0.35, 0.35), hi=(0.65, 0.65, 0.65))
    d = H.load_palm()
    spec = dict(d, friction=0.5, softness=666.0, contact=False)
    q = np.array([0.95, 0.1, -0.2, 0.15]); q /= np.linalg.norm(q)
    s13 = np.concatenate([[0.52, 0.9, 0.47], q, [0.1, -0.2, 0.05], [0.3, 0.2, -0.4]])
    cfg = H.sim_cfg(n, n_grid=64, precision=precision, max_steps=8, sort_interval=2)
    sim, prims = H.build_engine(cfg, 2e-3, [spec], [[s13]] * 8)
    sim.reset(state)
    sim.run_substeps(0, 4)
    x = sim.get_x(4)
    for cls, ncon in ((DoorLoss, 1), (TransportLoss, 2)):
        loss = cls(types.SimpleNamespace(weight=(0.7, 0.3, 2.0)), sim)
        loss.initialize()
        if ncon == 2:
            loss.set_target([0.4, 0.8, 0.5])
        sim.clear_grads()
        with loss.tape():
            info = loss.compute_loss(4)
        gx_ref = np.zeros((n, 3)); gp_ref = np.zeros(13); contact = 0.0
        per = n // ncon
        for k in range(ncon):
            xs = x[k * per:(k + 1) * per]
            dd = np.maximum(((xs - s13[:3]) ** 2).sum(1) - 0.01, 0.0)
            i = int(dd.argmin()); v = dd[i]
            contact += v * v
            gx_ref[k * per + i] += 2.0 * 4.0 * v * (xs[i] - s13[:3])
            gp_ref[:3] -= 2.0 * 4.0 * v * (xs[i] - s13[:3])
        assert contact > 0
        gp_ref[7:10] += 0.3 * 2 * s13[7:10]
        if ncon == 1:
            pose = (s13[3] - np.cos(np.pi / 8)) ** 2
            gp_ref[3] += 0.7 * 2 * (s13[3] - np.cos(np.pi / 8))
        else:
            pose = ((s13[:3] - [0.4, 0.8, 0.5]) ** 2).sum()
            gp_ref[:3] += 0.7 * 2 * (s13[:3] - [0.4, 0.8, 0.5])
        assert abs(info["contact_loss"] - 2.0 * contact) < max(tol, 1e-9) * 2.0 * contact * 10
        ptol = 1e-12 if precision == "float64" else 1e-7            # the primitive's state lives on the device in R
        assert abs(info["pose_loss"] - 0.7 * pose) < ptol and abs(info["vel_loss"] - 0.3 * (s13[7:10] ** 2).sum()) < ptol
        assert H.rel_err(sim.get_grad(4)[0], gx_ref) < tol * 10
        assert H.rel_err(prims[0].get_all_states_grad(4), gp_ref) < tol * 10
