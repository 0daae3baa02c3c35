"""The batched forward loop (`run_substeps`) runs substep f's G2P and substep f+1's P2G in ONE launch (k_g2p_p2g, round 4) wherever the two substeps
share a binning: x, v, C of frame f+1 stay in registers, forward_kinematics to frame f+1 and the emptying of the next hit counter move into
k_grid_op's launch, the next substep's hits go to the second of two lists.  It must give what the launch-per-kernel loop gives - every frame, the
primitives' wrench sums, the hit counts, and the adjoints a backward sweep computes from the frames and checkpoints it leaves behind - across
re-sorts, with batches that end inside an epoch, and for the float64 mode.  This file compares the library with itself (SMAC_FUSED_FWD = 1 / 0); the
comparison with the ORACLE is tests/test_gpu_parity.py::test_batched_sweep_against_the_oracle, whose batched legs take this path."""
import os

import numpy as np
import pytest

import helpers as H
from softmac_amd import scenes

pytestmark = pytest.mark.gpu


def _engine(fused, n_sub, sort_interval, precision, n=1 << 15, grid=64):
    old = os.environ.get("SMAC_FUSED_FWD")
    os.environ["SMAC_FUSED_FWD"] = "1" if fused else "0"
    try:
        cfg, env_dt, state, specs, s13 = scenes.s_grip(n, grid, max_steps=n_sub + 4, precision=precision, dt=1e-4)
        cfg.sort_interval = sort_interval
        pst = [[np.concatenate([s[:3] + s[7:10] * cfg.dt * f, s[3:]]) for s in s13] for f in range(n_sub + 4)]
        sim, prm = H.build_engine(cfg, env_dt, specs, pst)
    finally:
        if old is None:
            os.environ.pop("SMAC_FUSED_FWD", None)
        else:
            os.environ["SMAC_FUSED_FWD"] = old
    sim.reset(state)
    return sim, prm, cfg.n_particles


def _rel(a, b):
    a, b = np.asarray(a, dtype=np.float64).reshape(len(a), -1), np.asarray(b, dtype=np.float64).reshape(len(b), -1)
    s = np.abs(b).max()
    return 0.0 if s == 0 else float(np.abs(a - b).max() / s)


@pytest.mark.parametrize("precision,batches,sort_interval", [("float32", (24,), 8), ("float32", (5, 1, 7, 11), 6), ("float64", (14,), 5)])
def test_fused_forward_step_equals_the_two_kernels(precision, batches, sort_interval):
    n_sub = sum(batches)
    out = []
    for fused in (True, False):
        sim, prm, N = _engine(fused, n_sub, sort_interval, precision)
        sim.profile(True)
        f = 0
        for b in batches:                                      # a batch that ends inside an epoch: its last substep runs the plain G2P, the next batch's first the plain P2G
            sim.run_substeps(f, b)
            f += b
        counts = sim.profile_report()
        sim.profile(False)
        n_fused = counts.get("g2p_p2g", (0, 0))[1]
        assert (n_fused > 0) == fused, counts
        if fused:                                              # every substep that has a successor inside its batch and its epoch
            expect = sum(1 for b0, b in zip(np.cumsum((0,) + batches[:-1]), batches) for g in range(b0, b0 + b - 1) if (g + 1) % sort_interval != 0)
            assert n_fused == expect, (n_fused, expect, counts)
        states = [sim.get_state(g) for g in range(0, n_sub + 1, 3)]
        ext = np.array([m.ext_f.to_numpy() for m in prm])
        hits = sim.contact_counts()
        rng = np.random.default_rng(5)
        sim.clear_grads()
        sim.add_grad(n_sub, gx=rng.standard_normal((N, 3)), gv=rng.standard_normal((N, 3)))
        sim.run_substeps_grad(0, n_sub)
        grads = np.hstack([a.reshape(N, -1) for a in sim.get_grad_full(0)])
        pgrad = np.array([m.get_all_states_grad(g) for m in prm for g in range(n_sub)])
        out.append((states, ext, hits, grads, pgrad))
    (sa, ea, ha, ga, pa), (sb, eb, hb, gb, pb) = out
    # same arithmetic in the same order per particle; what differs is the order in which a chunk's particles reach the LDS tile (integers in f32 mode: none)
    # (the hit lists - hence the contact corrections' f64 sums - arrive in another order: the float64 leg is the sharp one, the float32 leg carries the
    #  parity suite's own bound, per field)
    tol = 1e-12 if precision == "float64" else H.F32_TOL["state"]
    worst = 0.0
    for g, (a, b) in enumerate(zip(sa, sb)):
        for c0, c1 in ((0, 3), (3, 6), (6, 15), (15, 24)):
            worst = max(worst, _rel(a[:, c0:c1], b[:, c0:c1]))
            assert _rel(a[:, c0:c1], b[:, c0:c1]) <= tol, ("state frame", 3 * g, "columns", c0, c1, _rel(a[:, c0:c1], b[:, c0:c1]))
    H.note(f"fused forward vs plain, {precision}, worst field error over the frames", worst, tol)
    assert ha == hb, (ha, hb)
    if ea is not None:
        assert _rel(ea, eb) <= (1e-10 if precision == "float64" else 1e-5), _rel(ea, eb)
    d = np.abs(ga - gb).max(axis=1) / np.abs(gb).max()
    assert np.quantile(d, 0.99) < (1e-10 if precision == "float64" else 1e-5) and d.max() < (1e-8 if precision == "float64" else 1e-1), (float(np.quantile(d, 0.99)), float(d.max()))
    assert _rel(pa, pb) <= (1e-9 if precision == "float64" else 1e-3), _rel(pa, pb)
