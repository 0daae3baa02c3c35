"""Round-3 bookkeeping around the re-sort and the contact lists (no reference counterpart: the reference keeps particles in creation order):

* a particle that is still in the cell of the previous binning keeps its rank there, so a re-sort moves only what has to move
  (`smac_get_param("resort_moved")` counts, on request, the particles the last binning put into another slot);
* the hit counters of even and odd frames alternate (k_g2p's launch carries the checkpoint save and empties the NEXT frame's counter): the count a
  forward substep reports must follow the frame through single calls, batched calls and a backward pass in between."""
import numpy as np
import pytest

import helpers as H
from test_gpu_parity import _palm_scene

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("precision", ["float32", "float64"])
def test_a_cloud_at_rest_keeps_every_slot_across_resorts(precision):
    n_grid, N = 32, 2500                      # 2.8 particles per cell: no cell comes near the 15 ranks a block's bins hold
    cfg = H.sim_cfg(N, n_grid=n_grid, dt=2e-4, ptype=1, gravity=(0., 0., 0.), precision=precision, sort_interval=1, max_steps=8)
    state = H.make_cloud(N, n_grid, seed=5, lo=(0.3, 0.3, 0.3), hi=(0.6, 0.6, 0.6), v_std=0.0, C_std=0.0, F_std=0.0)
    sim, _ = H.build_engine(cfg, 1e-3, [], None)
    sim.reset(state)
    assert sim.get_param("resorts") == 1 and sim.get_param("resort_moved") > N / 2      # the first binning, from the caller's order
    for f in range(3):
        sim.substep(f)                        # sort_interval 1: frame f is re-binned first (frame 0 already is)
    assert sim.get_param("resorts") == 3
    assert sim.get_param("resort_moved") == 0
    assert H.rel_err(sim.get_x(3), state[:, :3]) < 1e-6                                   # (nothing moved: F = I, v = 0, no gravity)


@pytest.mark.parametrize("guests", [0, 1])
def test_a_drifting_cloud_is_re_binned_by_short_shifts(guests, monkeypatch):
    monkeypatch.setenv("SMAC_GUESTS", str(guests))          # (1, the default since round 5: blocks hand their particles beyond whole chunks to a face neighbour)
    n_grid, N = 32, 20000
    cfg = H.sim_cfg(N, n_grid=n_grid, dt=2e-4, ptype=1, gravity=(0., -9.8, 0.), precision="float32", sort_interval=4, max_steps=16)
    state = H.make_cloud(N, n_grid, seed=6, lo=(0.3, 0.4, 0.3), hi=(0.6, 0.7, 0.6), v_std=0.05)
    state[:, 3] += 1.0                                                                    # 1 m/s: 0.026 cells per re-sort interval
    sim, _ = H.build_engine(cfg, 1e-3, [], None)
    sim.reset(state)
    sim.run_substeps(0, 9)                    # re-binned at frames 4 and 8
    assert sim.get_param("resorts") == 3
    moved, far = sim.get_param("resort_moved"), sim.get_param("resort_far")
    # Every particle behind the first one that changed cell shifts by a slot or two (the bins close up), so most slots change - but by the running
    # balance of arrivals and departures in front of them, a few dozen slots, which is what keeps the row moves coalesced.  With the ranks handed
    # out afresh (round 2: arrival order of atomics) a re-sort scattered each block's ~1,400 particles here over the block's whole range.
    print(f"\n[re-sort] {int(moved)} of {N} particles changed slot at the last re-binning, {int(far)} by more than a chunk (256 slots)")
    # only a particle that changed its cell between the two binnings (frames 4 and 8) can move far; one that stayed keeps block, rank and cell
    base = lambda x: np.clip((x * n_grid - 0.5).astype(np.int64), 0, n_grid - 3)
    changed = int((base(sim.get_x(4)) != base(sim.get_x(8))).any(axis=1).sum())
    print(f"          {changed} particles changed their cell between the two binnings")
    # (with guests: which particles of a block's facing layer take the tickets is settled by the order of their atomics, binning by binning - a guest of the last
    #  binning that is none now, or the reverse, moves to another block's range: a few per block that hands particles over)
    assert 0 < changed < 0.2 * N and (far <= changed if not guests else far <= changed + 0.05 * N)
    assert np.isfinite(sim.get_state(9)).all()


def test_contact_counts_follow_the_frame_through_batches_and_a_backward_pass():
    d = np.load(H.GOLDEN / "grip_state_2k.npz")
    state = d["state"]
    specs, pstates = _palm_scene(state, 6)
    cfg = H.sim_cfg(len(state), n_grid=64, dt=2e-4, ptype=0, material_model=0, ground_friction=20.0, precision="float32", max_steps=8)
    N = len(state)
    sim, _ = H.build_engine(cfg, 1e-3, specs, pstates)
    sim.reset(state)
    counts = []
    for f in range(5):                        # even and odd frames use different counters
        sim.substep(f)
        counts.append(sim.contact_counts()[0])
    assert min(counts) > 0
    # a backward sweep (one counter, set by the checkpoint restore) and the same forward substeps again: the same counts
    rng = np.random.default_rng(0)
    sim.clear_grads()
    sim.add_grad(5, gx=rng.standard_normal((N, 3)))
    for f in (4, 3, 2):
        sim.substep_grad(f)
    again = []
    for f in (2, 3, 4):
        sim.substep(f)
        again.append(sim.contact_counts()[0])
    assert again == counts[2:], (again, counts)
    # the batched entry point ends on the same frame with the same list; an odd-length batch after an even one
    sim2, _ = H.build_engine(cfg, 1e-3, specs, pstates)
    sim2.reset(state)
    sim2.run_substeps(0, 2)
    assert sim2.contact_counts()[0] == counts[1]
    sim2.run_substeps(2, 3)
    assert sim2.contact_counts()[0] == counts[4]
    sim2.clear_grads()
    sim2.add_grad(5, gx=rng.standard_normal((N, 3)))
    sim2.run_substeps_grad(0, 5)
    sim2.run_substeps(0, 4)
    assert sim2.contact_counts()[0] == counts[3]


@pytest.mark.parametrize("precision,tol", [("float64", 1e-10), ("float32", 2e-4)])
def test_guests_cut_the_chunk_count_and_change_nothing_else(precision, tol, monkeypatch):
    """Round 5 (smac_sort.hpp "guests"): the fused particle kernels are limited by their workgroup slots - time = chunks x a chunk's lifetime / resident workgroups.
    At 8 particles per cell a 4^3 block holds about 512 +- 40: three blocks in ten need a third chunk for a few dozen particles.  The re-sort hands such a remainder
    to a face neighbour with room (the particles of the block's outermost cell layer are binned one cell further: the wide tile takes a particle one node outside its
    chunk's block on the fast path).  Here: a box at 8 per cell; fewer chunks with guests, the same states and gradients over substeps with a re-sort in between."""
    n_grid, N = 32, 8 * 16 ** 3
    state = H.make_cloud(N, n_grid, seed=11, lo=(0.252, 0.252, 0.252), hi=(0.752, 0.752, 0.752), v_std=0.05, C_std=0.5, F_std=0.005)
    rng = np.random.default_rng(3)
    gx = rng.standard_normal((N, 3))
    out = {}
    for guests in (0, 1):
        monkeypatch.setenv("SMAC_GUESTS", str(guests))
        cfg = H.sim_cfg(N, n_grid=n_grid, dt=2e-4, ptype=1, gravity=(0., -9.8, 0.), precision=precision, sort_interval=3, max_steps=8)
        sim, _ = H.build_engine(cfg, 1e-3, [], None)
        sim.reset(state)
        sim.run_substeps(0, 6)                # re-binned at frames 3 (and 6)
        chunks = sim.get_param("chunks")
        sim.clear_grads()
        sim.add_grad(6, gx=gx)
        sim.run_substeps_grad(0, 6)
        out[guests] = (chunks, sim.get_state(6), sim.get_grad(0))
        assert sim.get_param("drift_repairs") == 0
    (c0, s0, g0), (c1, s1, g1) = out[0], out[1]
    print(f"\n[guests] {int(c0)} chunks -> {int(c1)}")
    assert c1 <= 0.95 * c0
    assert H.rel_err(s1, s0) < tol
    for a, b in zip(g1, g0):
        assert H.rel_err(a, b) < tol
