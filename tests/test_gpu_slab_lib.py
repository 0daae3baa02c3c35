"""The slab loop inside the library (smac_substeps_slab[_grad], RCCL) on ONE GPU.

A world-1 communicator in self-loop mode - left neighbour = right neighbour = this rank, on periodic planes - runs the whole in-library path:
phase kernels, two-sided plane pack, ncclGroupStart / ncclSend x 2 / ncclRecv x 2 / ncclGroupEnd on the communication stream, event hand-offs,
two-sided unpack.  Its result must equal the SAME self-loop driven from Python through the phase entry points and smac_halo_pack /
smac_halo_unpack_add - the building blocks the gloo world-2 / world-4 tests (tests/test_slabs.py) compare with the single-domain oracle."""
import numpy as np
import pytest

import helpers as H

pytestmark = pytest.mark.gpu

N_GRID, N, NSUB = 32, 4000, 6
LEFT0, RIGHT0, NP = 11, 19, 2


def _scene(precision):
    state = H.make_cloud(N, N_GRID, seed=51, lo=(0.3, 0.08, 0.3), hi=(0.7, 0.3, 0.7), v_std=0.3, F_std=5e-3)
    palm = H.load_palm()
    top = state[:, 1].max()
    q = np.array([0.995, 0.02, 0.03, 0.09]); q /= np.linalg.norm(q)
    s = np.concatenate([[0.5, top + 0.15 - 0.004, 0.5], q, [0.02, -0.3, 0.01], [0.1, 0.05, -0.2]])
    pst = []
    for f in range(NSUB + 2):
        pst.append([s.copy()])
        s[:3] = s[:3] + 2e-4 * s[7:10]
    spec = dict(palm, friction=0.4, softness=666.0, contact=True)
    cfg = H.sim_cfg(N, n_grid=N_GRID, dt=2e-4, ptype=0, ground_friction=20.0, precision=precision, max_steps=NSUB + 2, sort_interval=4,
                    slab_flags=6)                          # no wall at either x end: neighbours there
    return cfg, state, [spec], pst


def _seeds():
    rng = np.random.default_rng(52)
    return {NSUB: (rng.standard_normal((N, 3)), rng.standard_normal((N, 3)), 0.01 * rng.standard_normal((N, 3, 3)), 0.01 * rng.standard_normal((N, 3, 3))),
            2: (rng.standard_normal((N, 3)), None, None, None)}


def _collect(sim, prims):
    out = dict(st=sim.get_state(NSUB), ext=prims[0].ext_f.to_numpy().copy())
    for k, a in zip(("gx", "gv", "gF", "gC"), sim.get_grad_full(0)):
        out[k] = a.reshape(N, -1)
    out["pg"] = np.array([prims[0].get_all_states_grad(f) for f in range(NSUB)])
    return out


def _run(runner_factory, precision):
    cfg, state, specs, pst = _scene(precision)
    sim, prims = H.build_engine(cfg, 2e-3, specs, pst)
    sim.reset(state)
    run = runner_factory(sim)
    run.run_substeps(0, NSUB)
    sim.clear_grads()
    for f, s in _seeds().items():
        sim.add_grad(f, gx=s[0], gv=s[1], gC=s[2], gF=s[3])
    eg = [np.linspace(-1e-2, 1e-2, 6)]
    run.run_substeps_grad(0, NSUB, eg)
    out = _collect(sim, prims)
    out["hits"] = sim.contact_counts()[0]
    return out, run, sim


def _python_self_loop(sim):
    """the reference: SlabRunner's phase sequence with the exchange done by hand on this rank's own two plane sets"""
    from softmac_amd.parallel import HipSlabEngine, SlabRunner

    class SelfLoop(SlabRunner):
        def exchange(self, field, minus_mixed=0, contact_only=False):
            a, b = self._buffers("L")[0], self._buffers("R")[0]
            self.e.halo_pack(field, self.left0, self.np, a, minus_mixed)       # both packs BEFORE any unpack: partials, not totals
            self.e.halo_pack(field, self.right0, self.np, b, minus_mixed)
            self.e.halo_unpack_add(field, self.right0, self.np, a)             # what goes out on the left comes in on the right ...
            self.e.halo_unpack_add(field, self.left0, self.np, b)              # ... and vice versa

    return SelfLoop(HipSlabEngine(sim, use_torch_stream=False), 0, 1, LEFT0, RIGHT0, NP, has_contact=True)


@pytest.mark.parametrize("precision,tol", [("float64", 1e-11), ("float32", 2e-5)])
def test_backward_exchange_packed_and_added_inside_the_grid_kernels(precision, tol, monkeypatch):
    """Round 5 (VERDICT r4 next #5c): where no contact primitive reaches a shared plane, the exchange of grid_v_out.grad is packed by k_reduce_aout and added by
    k_grid_op_grad (no k_halo_pack2 / k_halo_unpack_add2 launches) - same numbers as the Python phase loop with its explicit pack / unpack, and as the library
    loop with the fusion switched off."""
    from softmac_amd.parallel import LibSlabRunner
    import test_gpu_slab_lib as me
    scene0 = me._scene

    def no_contact_scene(prec):
        cfg, state, specs, pst = scene0(prec)
        return cfg, state, [dict(specs[0], contact=False)], pst

    monkeypatch.setattr(me, "_scene", no_contact_scene)

    def py_loop(sim):
        r = _python_self_loop(sim)
        r.has_contact = False                             # (SlabRunner: no contact exchanges)
        r.contact_side = {"L": False, "R": False}
        return r

    ref, _, _ = _run(py_loop, precision)
    got, run, sim = _run(lambda s: LibSlabRunner(s, 0, 1, LEFT0, RIGHT0, NP, has_contact=(False, False), self_loop=True), precision)
    assert run.exchanges() == 2 * NSUB                    # {m,p} forward, grid_v_out.grad backward
    monkeypatch.setenv("SMAC_HALO_FUSE", "0")
    plain, run0, _ = _run(lambda s: LibSlabRunner(s, 0, 1, LEFT0, RIGHT0, NP, has_contact=(False, False), self_loop=True), precision)
    for k in ("st", "gx", "gv", "gF", "gC"):
        assert H.rel_err(got[k], ref[k]) < tol, (k, H.rel_err(got[k], ref[k]))
        assert H.rel_err(got[k], plain[k]) < tol, (k, H.rel_err(got[k], plain[k]))
    run.close(); run0.close()


@pytest.mark.parametrize("precision,tol", [("float64", 1e-11), ("float32", 2e-5)])
def test_in_library_rccl_self_exchange_equals_the_python_phase_loop(precision, tol):
    from softmac_amd.parallel import LibSlabRunner
    ref, _, _ = _run(_python_self_loop, precision)
    got, run, sim = _run(lambda s: LibSlabRunner(s, 0, 1, LEFT0, RIGHT0, NP, has_contact=(True, True), self_loop=True), precision)
    assert run.exchanges() == 4 * NSUB                    # 2 forward + 2 backward exchanges per substep, all through RCCL
    assert ref["hits"] > 10 and np.abs(ref["ext"]).max() > 0
    for k in ("st", "gx", "gv", "gF", "gC", "ext", "pg"):
        assert H.rel_err(got[k], ref[k]) < tol, (k, H.rel_err(got[k], ref[k]))
    # the exchange is not a no-op: without it the result differs
    cfg, state, specs, pst = _scene(precision)
    plain, prims = H.build_engine(cfg, 2e-3, specs, pst)
    plain.reset(state)
    plain.run_substeps(0, NSUB)
    assert H.rel_err(plain.get_state(NSUB), ref["st"]) > 1e-4
    run.close()


def test_stubbed_communication_moves_the_same_data(monkeypatch):
    """SMAC_COMM_STUB=1 (tools/exchange_overhead.py measures host enqueue time with it): the RCCL calls replaced by device copies."""
    from softmac_amd.parallel import LibSlabRunner
    ref, r0, _ = _run(lambda s: LibSlabRunner(s, 0, 1, LEFT0, RIGHT0, NP, has_contact=(True, True), self_loop=True), "float64")
    monkeypatch.setenv("SMAC_COMM_STUB", "1")
    got, r1, _ = _run(lambda s: LibSlabRunner(s, 0, 1, LEFT0, RIGHT0, NP, has_contact=(True, True), self_loop=True), "float64")
    for k in ("st", "gx", "gv"):
        assert H.rel_err(got[k], ref[k]) < 1e-11
    r0.close(); r1.close()


def test_primitive_reductions_and_the_slab_range_guard():
    from softmac_amd._ffi import SmacError
    from softmac_amd.parallel import LibSlabRunner
    cfg, state, specs, pst = _scene("float64")
    sim, prims = H.build_engine(cfg, 2e-3, specs, pst)
    sim.reset(state)
    # bases of this cloud span planes 9 .. 21; a rank that owns [12, 18) with tolerance 0 must refuse it
    run = LibSlabRunner(sim, 0, 1, LEFT0, RIGHT0, NP, has_contact=(True, True), self_loop=True, own=(12, 18), drift_tol=0)
    run.run_substeps(0, 1)
    with pytest.raises(SmacError, match="left the x-planes"):
        sim.sync()
    run.close()
    sim2, prims2 = H.build_engine(cfg, 2e-3, specs, pst)
    sim2.reset(state)
    run2 = LibSlabRunner(sim2, 0, 1, LEFT0, RIGHT0, NP, has_contact=(True, True), self_loop=True, own=(9, 22), drift_tol=1)
    run2.run_substeps(0, 2)
    e = prims2[0].ext_f.to_numpy().copy()
    tot = run2.allreduce_ext_f(clear=True)                 # world 1: the sum over ranks is the rank's own value
    assert np.abs(e).max() > 0 and np.allclose(tot[0], e) and np.abs(prims2[0].ext_f.to_numpy()).max() == 0
    run2.allreduce_state_grad(0, 2)
    run2.close()


def test_cloth_variant_under_the_slab_loop():
    """SURVEY 8 f4 + 8e: the soft <-> cloth substep cut into the slab phases - the sheet's contact exchanges its v_out
    corrections and grid_v_mixed.grad partials like an SDF primitive's - through the in-library RCCL loop (world-1 self exchange) against the
    Python phase loop; contact-face search and penetration tracing stay rank-local calls between the substeps, as in the reference's env loop
    (soft_cloth/engine/taichi_env.py:86-106).  Then the sheet's per-vertex force and vertex adjoints go through the in-library reductions."""
    import scenes_cloth as SC
    from softmac_amd.parallel import HipSlabEngine, LibSlabRunner, SlabRunner
    n = 4
    left0, right0 = 28, 34                                           # planes through the disc of plasticine (x in [1.9, 3.1] of 5 -> cells 24 .. 40 of 64)

    def run(make_runner):
        sc = SC.build("taco", "float64", n_env_steps=1)
        sc["cfg"].slab_flags = 6
        sim, prim = SC.build_engine(sc)
        for f in range(n + 1):
            prim.set_all_states(f, *sc["motion"](f * sc["cfg"].dt))
        sim.reset(sc["state"])
        runner = make_runner(sim)
        sim.get_contact_pair(0)
        for f in range(n):
            runner.run_substeps(f, 1)
            sim.get_contact_pair(f + 1)
            sim.trace_penetration_after_mpm(f + 1)
        N = len(sc["state"])
        rng = np.random.default_rng(61)
        sim.clear_grads()
        sim.add_grad(n, gx=rng.standard_normal((N, 3)), gv=rng.standard_normal((N, 3)))
        prim.set_ext_f_grad(1e-2 * rng.standard_normal((len(sc["vertices"]), 3)))
        for f in range(n - 1, -1, -1):
            runner.run_substeps_grad(f, 1)
        gx, gv = sim.get_grad(0)
        cp, cv = prim.get_all_states_grad(1)
        return dict(st=sim.get_state(n), gx=gx, gv=gv, ext=prim.ext_f.to_numpy().copy(), cp=cp, cv=cv, hits=sim.contact_counts()[0]), runner, sim, prim

    class SelfLoop(SlabRunner):
        def exchange(self, field, minus_mixed=0, contact_only=False):
            a, b = self._buffers("L")[0], self._buffers("R")[0]
            self.e.halo_pack(field, self.left0, self.np, a, minus_mixed)
            self.e.halo_pack(field, self.right0, self.np, b, minus_mixed)
            self.e.halo_unpack_add(field, self.right0, self.np, a)
            self.e.halo_unpack_add(field, self.left0, self.np, b)

    ref, _, _, _ = run(lambda s: SelfLoop(HipSlabEngine(s, use_torch_stream=False), 0, 1, left0, right0, 2, has_contact=True))
    got, runner, sim, prim = run(lambda s: LibSlabRunner(s, 0, 1, left0, right0, 2, has_contact=(True, True), self_loop=True))
    assert ref["hits"] > 20 and np.abs(ref["ext"]).max() > 0 and np.abs(ref["cp"]).max() > 0
    assert runner.exchanges() == 4 * n
    for k in ("st", "gx", "gv", "ext", "cp", "cv"):
        assert H.rel_err(got[k], ref[k]) < 1e-10, (k, H.rel_err(got[k], ref[k]))
    before = prim.ext_f.to_numpy().copy()
    runner.allreduce_ext_f()                                          # world 1: sums of one rank
    runner.allreduce_state_grad(0, n)
    assert np.allclose(prim.ext_f.to_numpy(), before) and np.allclose(prim.get_all_states_grad(1)[0], got["cp"])
    runner.close()


@pytest.mark.parametrize("precision,tol", [("float64", 1e-10), ("float32", H.F32_TOL["state"])])      # (measured 1.1e-6 / 5.9e-7: profiles/r04_g_f32_bounds.txt)
def test_device_side_migration_round_trip(precision, tol):
    """smac_migrate / smac_migrate_grad (VERDICT r2 missing #3: migration went through get_state on the host).  World-1 self exchange: what leaves
    on one side re-enters on the other with the SAME coordinates, so the particle set is unchanged and only its order, its segment bookkeeping and
    the tape of hand-overs are exercised - rows and ids cross RCCL as bytes both ways, the frame index jumps at every migration point, the
    adjoint walks back through both.  The rollout with two migrations must equal the one without, particle by particle (matched by global id)."""
    from softmac_amd.parallel import LibSlabRunner
    n_grid, N = 32, 2400
    state = H.make_cloud(N, n_grid, seed=33, lo=(0.25, 0.3, 0.36), hi=(0.72, 0.5, 0.64), v_std=0.3, F_std=5e-3)
    state[: N // 2, 3] += 20.0                      # half the cloud flies right, half left: both neighbours receive
    state[N // 2:, 3] -= 20.0
    own = (12, 20)
    rng = np.random.default_rng(71)
    seed = (rng.standard_normal((N, 3)), rng.standard_normal((N, 3)))

    def make():
        cfg = H.sim_cfg(N, n_grid=n_grid, dt=2e-4, ptype=1, ground_friction=0.0, precision=precision, max_steps=16, sort_interval=4, slab_flags=6)
        sim, _ = H.build_engine(cfg, 1e-3)
        sim.reset(state)
        return sim, LibSlabRunner(sim, 0, 1, 10, 20, 2, has_contact=(False, False), self_loop=True)

    sim, run = make()
    run.run_substeps(0, 8)
    ref_end = sim.get_state(8)
    sim.clear_grads()
    sim.add_grad(8, gx=seed[0], gv=seed[1])
    run.run_substeps_grad(0, 8)
    ref_g = np.hstack(sim.get_grad(0))
    run.close()

    sim, run = make()
    run.run_substeps(0, 3)
    f = run.migrate(3, own)
    ids1 = run.ids()
    assert f == 4 and run.moved > 20 and sorted(ids1.tolist()) == list(range(N)) and not (ids1 == np.arange(N)).all()
    run.run_substeps(4, 3)
    f = run.migrate(7, own)
    ids2 = run.ids()
    assert f == 8 and sorted(ids2.tolist()) == list(range(N))
    run.run_substeps(8, 2)
    end = sim.get_state(10)
    assert H.note(f"migration state {precision}", H.rel_err(end, ref_end[ids2]), tol) < tol
    sim.clear_grads()
    sim.add_grad(10, gx=seed[0][ids2], gv=seed[1][ids2])
    run.run_substeps_grad(8, 2)
    run.migrate_grad()
    assert (run.ids() == ids1).all()
    run.run_substeps_grad(4, 3)
    run.migrate_grad()
    assert (run.ids() == np.arange(N)).all()
    run.run_substeps_grad(0, 3)
    g = np.hstack(sim.get_grad(0))
    assert H.note(f"migration grad {precision}", H.rel_err(g, ref_g), tol) < tol, H.rel_err(g, ref_g)
    run.close()


def test_the_slab_loop_takes_the_fused_particle_launches():
    """Round 4: G2P of substep f + P2G of substep f + 1, and p2g.grad of substep f + g2p.grad of substep f - 1, cross no halo exchange - the library's slab
    loop runs them as the single-GPU loop does (k_g2p_p2g / k_p2g_g2p_grad), its grid passes stay in pieces around the exchanges.  Same result as the
    Python phase loop (plain kernels), and the fused launches are asserted taken.  float32 (the fused backward kernel is its own), one seed at the end."""
    from softmac_amd.parallel import LibSlabRunner

    def run(factory):
        cfg, state, specs, pst = _scene("float32")
        sim, prims = H.build_engine(cfg, 2e-3, specs, pst)
        sim.reset(state)
        r = factory(sim)
        sim.profile(True)
        r.run_substeps(0, NSUB)
        sim.clear_grads()
        s = _seeds()[NSUB]
        sim.add_grad(NSUB, gx=s[0], gv=s[1], gC=s[2], gF=s[3])
        r.run_substeps_grad(0, NSUB, [np.linspace(-1e-2, 1e-2, 6)])
        counts = sim.profile_report()
        sim.profile(False)
        return _collect(sim, prims), counts, r

    ref, c_ref, _ = run(_python_self_loop)
    got, c_got, r = run(lambda s: LibSlabRunner(s, 0, 1, LEFT0, RIGHT0, NP, has_contact=(True, True), self_loop=True))
    assert c_ref.get("g2p_p2g", (0, 0))[1] == 0 and c_ref.get("p2g_g2p_grad", (0, 0))[1] == 0, c_ref          # the phase entry points from Python: plain kernels
    # sort_interval 4 over 6 substeps: substeps 0-3 and 4-5 share a binning -> fused forward launches after substeps 0, 1, 2 and 4; fused backward ones likewise
    assert c_got.get("g2p_p2g", (0, 0))[1] == 4 and c_got.get("p2g_g2p_grad", (0, 0))[1] == 4, c_got
    for k in ("st", "gx", "gv", "gF", "gC", "ext", "pg"):
        assert H.rel_err(got[k], ref[k]) < 2e-5, (k, H.rel_err(got[k], ref[k]))
    r.close()
