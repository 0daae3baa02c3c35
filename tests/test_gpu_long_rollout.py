"""Trajectory-length scaling.  The reference keeps x/v/C/F AND their gradients for every
step (mpm_simulator.py:53-56) - 3.9 GB + 3.9 GB at its 5000 particles; at 1M particles a 2000-substep episode
(demo_grip_config.py:25, demo_grip.py:189-191) is 200 GB of state alone.  The state frames stay resident here too (they
are the tape), but the adjoint frames roll: substep_grad(f) needs .grad[f] and .grad[f+1] plus whatever a loss seeded
ahead of the sweep (smac_config.adjoint_frames), and the per-frame contact hit lists are sized for 1/8 of the particles
instead of all of them."""
import time

import numpy as np
import pytest

import helpers as H
from softmac_amd import scenes
from softmac_amd._ffi import SmacError
from test_gpu_parity import _palm_scene

pytestmark = pytest.mark.gpu


def _small(precision, adjoint_frames, nsteps=12):
    state = np.load(H.GOLDEN / "grip_state_2k.npz")["state"]
    specs, pstates = _palm_scene(state, nsteps)
    cfg = H.sim_cfg(len(state), n_grid=64, dt=2e-4, ptype=0, precision=precision, max_steps=nsteps + 2, sort_interval=4,
                    adjoint_frames=adjoint_frames)
    sim, prims = H.build_engine(cfg, 1e-3, specs, pstates)
    sim.reset(state)
    sim.run_substeps(0, nsteps)
    return sim, prims, len(state)


@pytest.mark.parametrize("precision,tol", [("float64", 1e-11), ("float32", 1e-5)])
def test_rolling_adjoint_frames_equal_resident_ones(precision, tol):
    n = 12
    rng = np.random.default_rng(2)
    out = []
    for k in (0, 6):                                           # 0: one adjoint frame per state frame; 6: pool of 6
        sim, prims, N = _small(precision, k, n)
        rng = np.random.default_rng(2)
        sim.clear_grads()
        for f in (n, 7, 3):                                    # seeds ahead of the sweep hold a slot until it reaches them
            sim.add_grad(f, gx=rng.standard_normal((N, 3)), gv=rng.standard_normal((N, 3)) if f == n else None)
        sim.run_substeps_grad(0, n)
        gx, gv, gF, gC = sim.get_grad_full(0)
        out.append((np.hstack([gx, gv]), gF, gC, np.array([m.get_all_states_grad(3) for m in prims])))
        if k:
            with pytest.raises(SmacError, match="released"):
                sim.get_grad(5)                                # the sweep passed it long ago
            assert np.abs(sim.get_grad(1)[0]).max() > 0        # the two most recent frames stay readable
    for a, b in zip(out[0], out[1]):
        assert H.rel_err(b, a) < tol


def test_rolling_pool_too_small_is_an_error_not_a_wrong_gradient():
    sim, prims, N = _small("float64", 3, 8)
    sim.clear_grads()
    sim.add_grad(8, gx=np.ones((N, 3)))
    sim.add_grad(5, gx=np.ones((N, 3)))
    sim.add_grad(2, gx=np.ones((N, 3)))                        # three seeded frames fill the pool of 3 ...
    with pytest.raises(SmacError, match="rolling adjoint storage exhausted"):
        sim.run_substeps_grad(0, 8)                            # ... so .grad[7] has nowhere to go


def test_2000_substep_episode_at_one_million_particles():
    """BASELINE config C3's size for the length of the reference's grip demo: 2000 substeps forward + backward of the 1M-
    particle S-grip scene on one GPU (about 230 GB: 202 GB of state frames, grid checkpoints, 4 rolling adjoint frames).
    First a 64-substep episode on the SAME handle must reproduce the gradient of a small fully-resident handle."""
    nlong, npre = 2000, 64

    def engine(max_steps, adjoint_frames):
        cfg, env_dt, state, specs, s13 = scenes.s_grip(1 << 20, 128, max_steps=max_steps, precision="float32")
        cfg.adjoint_frames = adjoint_frames
        # the fingers close for 300 substeps (9 mm each) and then hold
        pst = [[np.concatenate([s[:3] + s[7:10] * cfg.dt * min(f, 300), s[3:7], s[7:10] * (1.0 if f < 300 else 0.0), s[10:]]) for s in s13]
               for f in range(max_steps)]
        sim, prm = H.build_engine(cfg, env_dt, specs, pst)
        sim.reset(state)
        return sim, prm, cfg.n_particles

    seed = np.random.default_rng(4).standard_normal((1 << 20, 3))
    ref_sim, _, N = engine(npre + 2, 0)
    ref_sim.run_substeps(0, npre)
    ref_sim.clear_grads(); ref_sim.add_grad(npre, gx=seed)
    ref_sim.run_substeps_grad(0, npre)
    g_ref = np.hstack(ref_sim.get_grad(0))
    del ref_sim

    sim, prm, N = engine(nlong + 2, 4)
    sim.run_substeps(0, npre)
    sim.clear_grads(); sim.add_grad(npre, gx=seed)
    sim.run_substeps_grad(0, npre)
    g = np.hstack(sim.get_grad(0))
    per = np.abs(g - g_ref).reshape(N, -1).max(1) / np.abs(g_ref).max()
    assert np.sort(per)[int(0.999 * N)] < 1e-5 and per.max() < 1e-2, (np.sort(per)[int(0.999 * N)], per.max())   # (clamp-zone particles: helpers.F32_TOL)

    t0 = time.perf_counter()
    sim.run_substeps(npre, nlong - npre)
    sim.sync()
    t1 = time.perf_counter()
    sim.clear_grads(); sim.add_grad(nlong, gx=seed)
    sim.run_substeps_grad(0, nlong)
    gx, gv = sim.get_grad(0)
    t2 = time.perf_counter()
    print(f"[long rollout] forward {nlong - npre} substeps {t1 - t0:.2f} s, backward {nlong} substeps {t2 - t1:.2f} s")
    x = sim.get_x(nlong)
    assert np.isfinite(x).all() and np.isfinite(gx).all() and np.isfinite(gv).all()
    assert np.abs(gx).max() > 0 and x[:, 1].min() > 0.0
