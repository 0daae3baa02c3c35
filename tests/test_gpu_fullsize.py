"""BASELINE.json's full size (S-grip: 1,048,576 particles, 128^3, plasticine, three gripper primitives) on the GPU:

* one substep forward + adjoint against the C++ oracle port (oracle/mpm_cpu.cpp, itself pinned to the torch
  oracle at small sizes by tests/test_cpu_port.py) - direct parity at full size;
* size-independent properties: mass / linear-momentum conservation of P2G -> grid -> G2P, the adjoint
  dot-product identity <J d, r> = <d, J^T r> by central differences, linearity of the adjoint in its seeds,
  and invariance of a 20-substep rollout + backward to how often the particles are re-binned."""
import numpy as np
import pytest

import helpers as H
from helpers import O
from softmac_amd import scenes

pytestmark = pytest.mark.gpu

N_FULL, GRID_FULL = 1 << 20, 128


def _engine(precision, max_steps=8, sort_interval=None, n=N_FULL, grid=GRID_FULL, prims=True, **over):
    cfg, env_dt, state, specs, s13 = scenes.s_grip(n, grid, max_steps=max_steps, precision=precision)
    for k, v in over.items():
        setattr(cfg, k, v)
    if sort_interval is not None:
        cfg.sort_interval = sort_interval
    if not prims:
        specs, s13 = [], []
    # primitives move with their constant velocity (what bench.py does)
    pst = [[np.concatenate([s[:3] + s[7:10] * cfg.dt * f, s[3:]]) for s in s13] for f in range(max_steps)]
    sim, prm = H.build_engine(cfg, env_dt, specs, pst if prims else None)
    return cfg, env_dt, state, specs, pst, sim, prm


@pytest.mark.parametrize("precision,tol_s,tol_g", [("float64", 1e-10, 1e-8), ("float32", H.F32_TOL["state"], H.F32_TOL["gx"])])
def test_fullsize_one_substep_vs_cpu_port(precision, tol_s, tol_g):
    # f32: helpers.F32_TOL (fixed-point positions and the f64 contact chain keep forecast contact inside 1e-5)
    from oracle import mpm_cpu
    cfg, env_dt, state, specs, pst, sim, prm = _engine(precision)
    P = H.oracle_params(cfg, env_dt)
    port = mpm_cpu.CpuPort(P, specs)
    x, v, C, F = (t.numpy() for t in O.state24_split(state))
    p0 = np.array(pst[0])
    rx, rv, rC, rF, rext = port.substep(0, x, v, C, F, p0)
    sim.reset(state)
    sim.substep(0)
    st = sim.get_state(1)
    N = cfg.n_particles
    assert H.rel_err(st[:, 0:3], rx) < tol_s and H.rel_err(st[:, 3:6], rv) < tol_s
    assert H.rel_err(st[:, 6:15], rF.reshape(N, 9)) < tol_s and H.rel_err(st[:, 15:24], rC.reshape(N, 9)) < tol_s
    for i, m in enumerate(prm):
        scale = max(np.abs(rext).max(), 1e-12)
        assert np.abs(m.ext_f.to_numpy() - rext[i]).max() / scale < max(50 * tol_s, 1e-8)
    assert np.abs(rext[1:]).max() > 0                                        # the fingers do touch the block
    rng = np.random.default_rng(5)
    g = [rng.standard_normal((N, 3)), rng.standard_normal((N, 3)), 0.01 * rng.standard_normal((N, 3, 3)),
         0.01 * rng.standard_normal((N, 3, 3))]
    ref = port.substep_grad(0, x, v, C, F, *g, pst=p0)
    sim.clear_grads()
    sim.add_grad(1, gx=g[0], gv=g[1], gC=g[2], gF=g[3])
    sim.substep_grad(0)
    gx, gv, gF, gC = sim.get_grad_full(0)
    # particles inside the reference's SVD-adjoint clamp (helpers.F32_TOL "clamp") are bounded separately in f32
    Ft = (np.eye(3)[None] + cfg.dt * C) @ F
    s2 = np.linalg.svd(Ft, compute_uv=False) ** 2
    zone = np.minimum(np.abs(s2[:, 0] - s2[:, 1]), np.minimum(np.abs(s2[:, 1] - s2[:, 2]), np.abs(s2[:, 0] - s2[:, 2]))) < 4e-6
    for name, got, rf in (("gx", gx, ref[0]), ("gv", gv, ref[1]), ("gC", gC, ref[2]), ("gF", gF, ref[3])):
        eo, ei = H.rel_err_split(got, rf, zone)
        assert eo < tol_g and ei < (tol_g if precision == "float64" else H.F32_TOL["clamp"]), (name, eo, ei, int(zone.sum()))
    for i, m in enumerate(prm):
        scale = max(np.abs(ref[4]).max(), 1e-9)
        assert np.abs(m.get_all_states_grad(0) - ref[4][i]).max() / scale < 10 * tol_g


@pytest.mark.parametrize("precision,tol", [("float64", 1e-9), ("float32", 2e-5)])
def test_fullsize_mass_and_momentum_conservation(precision, tol):
    """No gravity, no contact, block away from the walls: sum of grid mass = N p_mass and the particles'
    total linear momentum is unchanged by P2G -> grid -> G2P (APIC + quadratic B-splines, the stress and
    affine terms sum to zero over a stencil)."""
    cfg, env_dt, state, specs, pst, sim, prm = _engine(precision, prims=False, gravity=(0., 0., 0.), ptype=1)
    sim.reset(state)
    m = sim.compute_grid_m_kernel(0)
    p_mass = (0.5 / cfg.n_grid) ** 2
    assert abs(m.sum() / (cfg.n_particles * p_mass) - 1) < tol
    assert sim.count_active_cells(0) == int((m > 0).sum())
    sim.substep(0)
    v0 = state[:, 3:6]
    v1 = sim.get_v(1)
    scale = np.abs(v0).sum(0).max()
    assert np.abs(v1.sum(0) - v0.sum(0)).max() / scale < tol


@pytest.mark.parametrize("ptype,model,tol", [(1, 1, 2e-6), (0, 0, 2e-3)])
def test_fullsize_adjoint_dot_product_and_linearity(ptype, model, tol):
    """f64: <d, J^T r> from the adjoint kernels equals the central difference of L(s) = <r, step2(s)>;
    and the adjoint is linear in its seeds.  Neo-Hookean (no SVD) must agree to the difference quotient's
    accuracy.  With the SVD materials the reference's `backward_svd` clamps 1/(s_i^2 - s_j^2) at 1e6
    (mpm_simulator.py:141-158), so for the ~3e-4 of the particles whose singular values nearly coincide the
    reference gradient - which the kernels reproduce, see the parity tests - is not the true derivative."""
    cfg, env_dt, state, specs, pst, sim, prm = _engine("float64", ptype=ptype, material_model=model)
    N = cfg.n_particles
    rng = np.random.default_rng(11)
    r = [rng.standard_normal((N, 3)), rng.standard_normal((N, 3)), 0.01 * rng.standard_normal((N, 3, 3)),
         0.01 * rng.standard_normal((N, 3, 3))]
    r2 = [rng.standard_normal(a.shape) * s for a, s in zip(r, (1, 1, 0.01, 0.01))]
    d = np.hstack([rng.standard_normal((N, 3)) * 1e-3, rng.standard_normal((N, 3)), 1e-2 * rng.standard_normal((N, 9)),
                   rng.standard_normal((N, 9))])

    def loss(st0):
        sim.reset(st0)
        sim.run_substeps(0, 2)
        s = sim.get_state(2)
        return (s[:, 0:3] * r[0]).sum() + (s[:, 3:6] * r[1]).sum() + (s[:, 15:24] * r[2].reshape(N, 9)).sum() + \
            (s[:, 6:15] * r[3].reshape(N, 9)).sum()

    def grads(seed):
        sim.clear_grads()
        sim.add_grad(2, gx=seed[0], gv=seed[1], gC=seed[2], gF=seed[3])
        sim.run_substeps_grad(0, 2)
        gx, gv, gF, gC = sim.get_grad_full(0)
        return np.hstack([gx, gv, gF.reshape(N, 9), gC.reshape(N, 9)])

    eps = 1e-6
    lp, lm = loss(state + eps * d), loss(state - eps * d)
    loss(state)
    g = grads(r)
    fd, an = (lp - lm) / (2 * eps), (g * d).sum()
    assert abs(fd - an) / abs(an) < tol, (fd, an)
    g2 = grads(r2)
    g3 = grads([2.5 * a + b for a, b in zip(r, r2)])
    assert H.rel_err(g3, 2.5 * g + g2) < 1e-11


def _rebinning_outputs(precision, si, nsub=20, branches=True):
    cfg, env_dt, state, specs, pst, sim, prm = _engine(precision, max_steps=nsub + 4, sort_interval=si)
    N = cfg.n_particles
    sim.reset(state)
    sim.run_substeps(0, nsub)
    st = sim.get_state(nsub)
    # Which side of each discontinuity of the reference's function every particle is on, per frame, from the rollout's own states (the batched
    # SVDs of 1M matrices per frame are most of this test's time):
    #   bits 0-5  yield clip :226-229: singular value k (sorted) above 1 + 3e-3 / below 1 - 2e-3 - the clip's adjoint is a step there
    #   bit  6    inside the SVD-adjoint clamp |s_j^2 - s_i^2| < 1e-6 (mpm_simulator.py:184-192): K = 1e6 multiplies the difference itself
    #   bits 8-11 collide_mixed's branches per finger (primitive_base.py:152 inside the 5e-3 band, :168 forecast position inside the body)
    side = np.zeros((nsub, N), dtype=np.uint16) if branches else None
    base0 = None
    for f in (range(0, nsub, 4) if branches else ()):         # (every fourth frame: these statistics are printed, not asserted, and were 50 s of the suite)
        s = sim.get_state(f)
        if base0 is None:
            base0 = np.floor(s[:, 0:3] * cfg.n_grid - 0.5).astype(np.int64)
        Ft = (np.eye(3)[None] + cfg.dt * s[:, 15:24].reshape(N, 3, 3)) @ s[:, 6:15].reshape(N, 3, 3)
        sv = np.linalg.svd(Ft, compute_uv=False)
        s2 = sv ** 2
        gap = np.minimum(np.abs(s2[:, 0] - s2[:, 1]), np.minimum(np.abs(s2[:, 1] - s2[:, 2]), np.abs(s2[:, 0] - s2[:, 2])))
        b = np.zeros(N, dtype=np.uint16)
        for k in range(3):
            b |= ((sv[:, k] - 1.0 > 3e-3).astype(np.uint16) << (2 * k)) | ((sv[:, k] - 1.0 < -2e-3).astype(np.uint16) << (2 * k + 1))
        b |= (gap < 1e-6).astype(np.uint16) << 6
        j = 0
        for i, sp in enumerate(specs):
            if not sp["contact"]:
                continue
            c = pst[f][i][:3]
            xx, xf = s[:, 0:3], s[:, 0:3] + cfg.dt * s[:, 3:6]                  # x and the forecast position x + v dt (:166); finger = y-axis cylinder
            d0 = np.maximum(np.hypot(xx[:, 0] - c[0], xx[:, 2] - c[2]) - 0.05, np.abs(xx[:, 1] - c[1]) - 0.1)
            d1 = np.maximum(np.hypot(xf[:, 0] - c[0], xf[:, 2] - c[2]) - 0.05, np.abs(xf[:, 1] - c[1]) - 0.1)
            b |= ((d0 < 5e-3).astype(np.uint16) << (8 + 2 * j)) | ((d1 < 0).astype(np.uint16) << (9 + 2 * j))
            j += 1
        side[f] = b
    rng = np.random.default_rng(3)
    sim.clear_grads()
    sim.add_grad(nsub, gx=rng.standard_normal((N, 3)), gv=rng.standard_normal((N, 3)))
    sim.run_substeps_grad(0, nsub)
    gx, gv, gF, gC = sim.get_grad_full(0)
    ext = np.array([m.ext_f.to_numpy() for m in prm])
    return st, np.hstack([gx, gv]), gF, ext, dict(side=side, base=base0, n_grid=cfg.n_grid)


@pytest.mark.parametrize("precision", ["float64", "float32"])
def test_fullsize_rebinning_invariance(precision):
    """A 20-substep S-grip rollout and its backward pass must not depend on how often the particles are re-binned
    (sort_interval 1 / 16 / never): only the summation order changes.  Max-norm PER PARTICLE (r1's relative L2 hid a 6.7 %
    outlier in one particle's F adjoint).  f64 rollouts agree to 1e-8 for every particle (a deterministic check of the re-ordering code).
    In f32 two rollouts are a rounding apart in state, and the reference's function is DISCONTINUOUS in places: the SVD-adjoint clamp
    (K = 1e6 times the singular-value difference itself inside |s_j^2 - s_i^2| < 1e-6), the yield clip of sigma (its adjoint switches between
    ma_ii and 0) and the contact branches.  Plasticity PUTS particles on the clip boundary every substep: 48,298 of the 1,048,576 particles are
    within 3.6e-7 (4 x the largest F difference between two rollouts) of a clip bound at some frame, so which of them the DEVICE's float32
    evaluation puts on the other side cannot be told from the stored states (round 3 tried: recomputing every branch decision of both
    rollouts in f64 from their frames finds 31 differing decisions, whose 2-cell neighbourhoods explain 18 of the 51 particles over 1e-3).
    What the test can state without knowing the flips: the state agrees to 1e-5 for every particle; every adjoint stays within 1e-1 of the
    field's max; the particles beyond 1e-3 are at most 128 = 1.2e-4 of the cloud.  That last number is an observed property of this scene,
    not a derivation: 9, 32, 35, 35, 37, 41, 54, 72 over eight builds of round 2, 51, 59, 82 in round 3 - it moves with every change of the
    summation order and has been fixed at 128 since commit 88268e1; the branch statistics are printed so that a change of regime is visible.
    Round 5 put a derivation beside the observation (tests/test_gpu_window_parity.py): against the f64 oracle on the same window, every SINGLE substep's
    adjoint on identical inputs holds 1e-5 outside recorded carve-outs of a few dozen particles; what two float32 rollouts (or a float32 and a float64 one) differ
    by END TO END is the f64 function's own sensitivity to their 1e-7 state difference - the f64 port moves by more when only its F is stored in float32."""
    f32 = precision == "float32"
    outs = [_rebinning_outputs(precision, si, branches=f32) for si in (1, 16, 1000)]
    N = outs[0][0].shape[0]
    per = lambda a, b: np.abs(np.asarray(a) - np.asarray(b)).reshape(N, -1).max(1) / np.abs(np.asarray(b)).max()
    # f32: two 20-substep rollouts that differ only in summation order agree to 1e-5 on every particle's state; adjoints to 1e-3 away from
    # the flipped branches, 1e-1 next to them (a flipped branch changes that particle's adjoint by O(1) of ITS size)
    ts, tg, tz = (1e-9, 1e-8, 1e-8) if not f32 else (1e-5, 1e-3, 1e-1)
    info = outs[0][4]
    for o in outs[1:]:
        es, eg, ef = per(o[0], outs[0][0]), per(o[1], outs[0][1]), per(o[2], outs[0][2])
        assert es.max() < ts
        over = (eg > tg) | (ef > tg)
        if not f32:
            assert not over.any()
            print(f"[float64] state {es.max():.1e}  gx,gv {eg.max():.1e}  gF {ef.max():.1e}")
            continue
        a, b = info["side"], o[4]["side"]
        flipped = ((a != b) | (((a | b) >> 6) & 1).astype(bool)).any(axis=0)       # a branch taken differently in some frame, or the clamp active
        n = info["n_grid"]
        occ = np.zeros((n, n, n), dtype=bool)
        c = np.clip(info["base"][flipped], 0, n - 1)
        occ[c[:, 0], c[:, 1], c[:, 2]] = True
        import scipy.ndimage
        occ = scipy.ndimage.binary_dilation(occ, structure=np.ones((3, 3, 3), dtype=bool), iterations=2)   # shares a grid node: within 2 cells
        ba = np.clip(info["base"], 0, n - 1)
        explained = occ[ba[:, 0], ba[:, 1], ba[:, 2]]
        stray = over & ~explained
        print(f"[float32] state {es.max():.1e}  gx,gv {eg.max():.1e}  gF {ef.max():.1e};  particles whose branch decisions, recomputed in f64 from the two "
              f"rollouts' frames, differ (or that sit in the clamp) {int(flipped.sum())}, their 2-cell neighbourhoods {explained.mean():.2%} of the cloud;  "
              f"over 1e-3: {int(over.sum())}, of which outside those neighbourhoods {int(stray.sum())}")
        assert int(over.sum()) <= 128
        assert eg.max() < tz and ef.max() < tz
        assert H.rel_err(o[3], outs[0][3]) < max(100 * ts, 1e-8)


@pytest.mark.parametrize("precision", ["float64", "float32"])
def test_c4_slice_4m_particles_256_grid_vs_cpu_port(precision):
    """BASELINE config C4 on ONE GPU (VERDICT r1 next #2b): S-pour - 4,194,304 liquid particles, 256^3 grid, the reference's bowl
    (voxelised here) in forecast contact - one substep forward + adjoint against the C++ oracle port, and P2G -> grid -> G2P mass
    conservation.  The slab-decomposed form of this size is what `bench.py --gpus N --scaling weak` runs per rank."""
    from oracle import mpm_cpu
    from softmac_amd.engine.primitive import voxelize
    d = np.load(H.GOLDEN / "pour_scene.npz")
    bowl = voxelize.mesh_to_sdf(d["bowl_vertices"], d["bowl_faces"])
    cfg, env_dt, state, specs, s13 = scenes.s_pour(1 << 22, 256, max_steps=4, precision=precision, bowl_table=bowl)
    pst = [[s.copy() for s in s13] for _ in range(4)]
    sim, prm = H.build_engine(cfg, env_dt, specs, pst)
    N = cfg.n_particles
    P = H.oracle_params(cfg, env_dt)
    port = mpm_cpu.CpuPort(P, specs)
    x, v, C, F = (t.numpy() for t in O.state24_split(state))
    p0 = np.array(pst[0])
    rx, rv, rC, rF, rext = port.substep(0, x, v, C, F, p0)
    sim.reset(state)
    m = sim.compute_grid_m_kernel(0)
    assert abs(m.sum() / (N * (0.5 / cfg.n_grid) ** 2) - 1) < (1e-9 if precision == "float64" else 2e-5)
    sim.substep(0)
    assert sim.contact_counts()[0] > 1000                                       # the column does sit in the bowl's contact band
    st = sim.get_state(1)
    ts = 1e-10 if precision == "float64" else H.F32_TOL["state"]
    assert H.rel_err(st[:, 0:3], rx) < ts and H.rel_err(st[:, 3:6], rv) < ts and H.rel_err(st[:, 6:15], rF.reshape(N, 9)) < ts
    assert H.rel_err(st[:, 15:24], rC.reshape(N, 9)) < (ts if precision == "float64" else H.c_tol(ts, cfg.n_grid, rv, rC))
    assert np.abs(prm[0].ext_f.to_numpy() - rext[0]).max() / max(np.abs(rext).max(), 1e-12) < max(50 * ts, 1e-8)
    rng = np.random.default_rng(6)
    g = [rng.standard_normal((N, 3)), rng.standard_normal((N, 3)), 0.01 * rng.standard_normal((N, 3, 3)), 0.01 * rng.standard_normal((N, 3, 3))]
    ref = port.substep_grad(0, x, v, C, F, *g, pst=p0)
    sim.clear_grads()
    sim.add_grad(1, gx=g[0], gv=g[1], gC=g[2], gF=g[3])
    sim.substep_grad(0)
    gx, gv, gF, gC = sim.get_grad_full(0)
    tg = 1e-8 if precision == "float64" else H.F32_TOL["grad"]
    for name, got, rf in (("gx", gx, ref[0]), ("gv", gv, ref[1]), ("gC", gC, ref[2]), ("gF", gF, ref[3])):
        assert H.rel_err(got, rf) < tg, (name, H.rel_err(got, rf))               # (a liquid with mu = 0 takes no SVD: no clamp zone)
    # the batched sweep at this size (float32: fused backward step, restore-ahead on the second grid buffer set - 1 GB at 256^3 -, checkpoint
    # save inside k_g2p, empty-block flags) against the substep-by-substep sweep of the same handle
    sim.run_substeps(0, 3)
    seed = rng.standard_normal((N, 3))
    sim.clear_grads()
    sim.add_grad(3, gx=seed)
    for f in (2, 1, 0):
        sim.substep_grad(f)
    one_by_one = sim.get_grad_full(0)
    sim.clear_grads()
    sim.add_grad(3, gx=seed)
    sim.profile(True)
    sim.run_substeps_grad(0, 3)
    counts = sim.profile_report()
    sim.profile(False)
    if precision == "float32":
        assert counts.get("p2g_g2p_grad", (0, 0))[1] == 2
    for name, got, rf in zip(("gx", "gv", "gF", "gC"), sim.get_grad_full(0), one_by_one):
        assert H.rel_err(got, rf) < (1e-10 if precision == "float64" else 1e-5), (name, H.rel_err(got, rf))
