"""Slab decomposition = single domain.  Two ranks over gloo each own half of the particles of one scene and
exchange halo planes through softmac_amd.parallel.SlabRunner; every particle's state, every adjoint, the wrench
and the primitive-state adjoints must equal the single-domain oracle run.
CPU: oracle stand-in engine (exercises the exchange logic).  GPU: the real HIP engine, 2 ranks on the one GPU."""
import pathlib
import socket
import subprocess
import sys

import numpy as np
import pytest

import helpers as H
import scenes_slab as S

HERE = pathlib.Path(__file__).resolve().parent


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close()
    return p


def _run_ranks(engine, precision, tmp_path, world=2, scene="static", env=None):
    port = _free_port()
    import os
    procs = [subprocess.Popen([sys.executable, str(HERE / "slab_worker.py"), "--engine", engine, "--rank", str(r), "--world", str(world),
                               "--port", str(port), "--out", str(tmp_path), "--precision", precision, "--scene", scene],
                              env=None if env is None else dict(os.environ, **env)) for r in range(world)]
    for p in procs:
        assert p.wait(timeout=900) == 0
    return [np.load(tmp_path / f"rank{r}.npz") for r in range(world)]


def _run_two_ranks(engine, precision, tmp_path):
    return _run_ranks(engine, precision, tmp_path)


def _check_moving(parts, world, tol_state, tol_grad):
    """migration: every particle's final state (wherever it ended up) and its frame-0 adjoint (on the rank that owned it then)
    equal the single-domain oracle; particles did change hands"""
    sc = S.build_moving("float64", world)
    P = H.oracle_params(sc["cfg"], sc["env_dt"])
    n = sc["nsteps"]
    N = len(sc["state"])
    orc = H.OracleRollout(P, sc["state"]).forward(n)
    rng = np.random.default_rng(77)
    seed_end = np.hstack([rng.standard_normal((N, 3)), rng.standard_normal((N, 3)), 0.01 * rng.standard_normal((N, 9)), 0.01 * rng.standard_normal((N, 9))])
    seed_1 = np.hstack([rng.standard_normal((N, 3)), np.zeros((N, 21))])
    split = lambda r: (r[:, 0:3], r[:, 3:6], r[:, 15:24].reshape(N, 3, 3), r[:, 6:15].reshape(N, 3, 3))      # rows are x v F C -> (x, v, C, F)
    adj, _, _ = orc.backward({n: split(seed_end), 1: (seed_1[:, 0:3], None, None, None)})
    x, v, C, F = (t.numpy() for t in orc.frames[n])
    ref_end = np.hstack([x, v, F.reshape(N, 9), C.reshape(N, 9)])
    ref_g0 = np.hstack([adj[0][0].numpy(), adj[0][1].numpy(), adj[0][3].numpy().reshape(N, 9), adj[0][2].numpy().reshape(N, 9)])
    assert sorted(np.concatenate([p["ids_end"] for p in parts]).tolist()) == list(range(N))                  # nobody lost, nobody doubled
    assert sum(int(p["moved"]) for p in parts) > 20 * (world - 1)                                              # ownership really changed
    # the slabs' rows put back into single-domain order: errors are then measured as everywhere else (max-norm relative to the FIELD's max,
    # clamp-zone tiers of helpers.F32_TOL) instead of per rank
    end, g0 = np.zeros((N, 24)), np.zeros((N, 24))
    for p in parts:
        end[p["ids_end"]] = p["st_end"]
        g0[p["ids0"]] = p["g0"]
    f32 = tol_state > 1e-8
    zone, near = H.clamp_zone(orc, P, n, neighbours=True) if f32 else (np.zeros(N, dtype=bool), np.zeros(N, dtype=bool))
    errs = {}
    for name, sl in (("x", slice(0, 3)), ("v", slice(3, 6)), ("F", slice(6, 15)), ("C", slice(15, 24))):
        lim = tol_state
        if name == "C" and f32:                                  # C in float32: difference quotient of a 20 m/s velocity field (helpers.c_tol)
            lim = H.c_tol(tol_state, sc["n_grid"], ref_end[:, 3:6], ref_end[:, 15:24])
        errs[name] = (H.rel_err(end[:, sl], ref_end[:, sl]), lim)
        assert errs[name][0] < lim, (name, errs)
    gerrs = {}
    for name, sl in (("gx", slice(0, 3)), ("gv", slice(3, 6)), ("gF", slice(6, 15)), ("gC", slice(15, 24))):
        lim = tol_grad
        if name == "gx" and f32:                                 # x.grad of a fast cloud: the adjoint's twin of c_tol (helpers.gx_tol)
            lim = H.gx_tol(tol_grad, sc["n_grid"], ref_end[:, 3:6], seed_end[:, 3:6], seed_end[:, 15:24], ref_g0[:, 0:3])
        out, nr, zn = H.rel_err_tiers(g0[:, sl], ref_g0[:, sl], zone, near)
        gerrs[name] = (out, nr, zn, lim)
    print(f"\n[migration, world {world}, {'f32' if f32 else 'f64'}] state {errs}  adjoint (outside, next to, inside the clamp zone, bound) {gerrs}")
    for name, (out, nr, zn, lim) in gerrs.items():
        assert out < lim, (name, gerrs)
        assert nr < max(lim, H.F32_TOL["near_clamp"] if f32 else lim) and zn < max(lim, H.F32_TOL["clamp"] if f32 else lim), (name, gerrs)


@pytest.mark.parametrize("world", [2, 4])
def test_migration_between_slabs_matches_single_domain_cpu(tmp_path, world):
    _check_moving(_run_ranks("oracle", "float64", tmp_path, world, "moving"), world, 1e-11, 1e-9)


@pytest.mark.gpu
# (f32: the cloud flies at 20 m/s, 850 x its velocity-gradient scale: C and x.grad have the cancellation floors of helpers.c_tol / gx_tol,
#  stated there; every other field is held to F32_TOL)
@pytest.mark.parametrize("precision,ts,tg", [("float64", 1e-9, 1e-8), ("float32", H.F32_TOL["state"], H.F32_TOL["grad"])])
def test_migration_between_slabs_matches_single_domain_gpu(tmp_path, precision, ts, tg):
    _check_moving(_run_ranks("hip", precision, tmp_path, 2, "moving"), 2, ts, tg)


def _check(parts, tol_state, tol_grad):
    sc = S.build()
    P = H.oracle_params(sc["cfg"], sc["env_dt"])
    orc = H.OracleRollout(P, sc["state"], sc["specs"], sc["pstates"]).forward(sc["nsteps"])
    adj, pg, _ = orc.backward(S.seeds(sc), sc["ext_f_grad"])
    n = sc["nsteps"]
    ref = dict(zip("xvCF", [t.numpy() for t in orc.frames[n]]))
    refg = dict(gx=adj[0][0].numpy(), gv=adj[0][1].numpy(), gC=adj[0][2].numpy(), gF=adj[0][3].numpy())
    assert sum(len(p["idx"]) for p in parts) == len(sc["state"]) and min(len(p["idx"]) for p in parts) > 200
    N = len(sc["state"])
    f32 = tol_state > 1e-8
    zone, near = H.clamp_zone(orc, P, n, neighbours=True) if f32 else (np.zeros(N, dtype=bool), np.zeros(N, dtype=bool))
    full = {k: np.zeros(ref[k].shape) for k in "xvCF"}
    fullg = {k: np.zeros(refg[k].shape) for k in refg}
    for p in parts:
        for k in "xvCF":
            full[k][p["idx"]] = p[k]
        for k in refg:
            fullg[k][p["idx"]] = p[k]
    for k in "xvCF":
        assert H.rel_err(full[k], ref[k]) < tol_state, k
    gerrs = {k: H.rel_err_tiers(fullg[k], refg[k], zone, near) for k in refg}
    print(f"\n[two slabs, {'f32' if f32 else 'f64'}] adjoint errors (outside, next to, inside the clamp zone): {gerrs}")
    for k, (out, nr, zn) in gerrs.items():
        assert out < tol_grad, (k, gerrs)
        assert nr < (H.F32_TOL["near_clamp"] if f32 else tol_grad) and zn < (H.F32_TOL["clamp"] if f32 else tol_grad), (k, gerrs)
    ext = sum(p["ext"] for p in parts)
    assert H.rel_err(ext, np.sum(np.array(orc.ext), axis=0)) < max(tol_state * 100, 1e-8)
    pgr = sum(p["pgrad"] for p in parts)
    assert H.rel_err(pgr, np.array(pg)[:-1]) < tol_grad * 10


def test_two_slabs_match_single_domain_cpu(tmp_path):
    _check(_run_two_ranks("oracle", "float64", tmp_path), 1e-11, 1e-9)


@pytest.mark.gpu
@pytest.mark.parametrize("precision,ts,tg", [("float64", 1e-9, 1e-8), ("float32", H.F32_TOL["state"], H.F32_TOL["grad"])])
def test_two_slabs_match_single_domain_gpu(tmp_path, precision, ts, tg):
    _check(_run_two_ranks("hip", precision, tmp_path), ts, tg)


@pytest.mark.gpu
@pytest.mark.parametrize("precision,ts,tg", [("float64", 1e-9, 1e-8), ("float32", H.F32_TOL["state"], H.F32_TOL["grad"])])
def test_in_library_slab_loop_between_two_ranks_gpu(tmp_path, precision, ts, tg):
    """`smac_substeps_slab[_grad]` had only ever run as a world-1 self exchange (two RCCL ranks cannot share the one GPU of a
    development box).  SMAC_COMM_STUB=2 swaps RCCL for the IPC link of smac_comm.hpp - exported device mailboxes, host-synchronous - and runs the SAME
    loop between two processes: rank 0 has only a right neighbour, rank 1 only a left one, so the left / right slot mapping, the one-sided pack and
    unpack-add and the 2 + 2 exchanges per substep pair run as they would over RCCL.  Same scene and same bar as the Python loop: every particle, every
    adjoint, the wrench and the palm's state adjoints against the single-domain oracle; the link's all-reduces give every rank the sums."""
    parts = _run_ranks("lib", precision, tmp_path, env={"SMAC_COMM_STUB": "2"})
    _check(parts, ts, tg)
    sc = S.build()
    assert all(int(p["exchanges"]) == 4 * sc["nsteps"] for p in parts)
    ext, pgr = sum(p["ext"] for p in parts), sum(p["pgrad"] for p in parts)
    for p in parts:
        assert np.abs(p["ext_total"] - ext).max() <= 1e-12 * np.abs(ext).max() and np.abs(p["pgrad_total"] - pgr).max() <= 1e-12 * np.abs(pgr).max()


@pytest.mark.gpu
@pytest.mark.parametrize("precision,ts,tg", [("float64", 1e-9, 1e-8), ("float32", H.F32_TOL["state"], H.F32_TOL["grad"])])
def test_device_side_migration_between_two_ranks_gpu(tmp_path, precision, ts, tg):
    """`smac_migrate` / `smac_migrate_grad` between two real ranks (IPC link): counts, rows + global ids and, backwards, adjoint rows cross as device-side
    byte messages - against the single-domain oracle, as the host-staged Python migration is"""
    _check_moving(_run_ranks("lib", precision, tmp_path, 2, "moving", env={"SMAC_COMM_STUB": "2"}), 2, ts, tg)


@pytest.mark.gpu
def test_pending_exchange_is_released_when_the_neighbour_fails_gpu(tmp_path):
    """the healthy rank really has a pending exchange - it is inside smac_substeps_slab at the link's barrier - when its neighbour fails.
    parallel.FailureWatch publishes out of band, the healthy rank's watch thread calls smac_comm_abort from a second host thread, both ranks raise."""
    import json
    _run_ranks("lib", "float64", tmp_path, 2, "lib_failure", env={"SMAC_COMM_STUB": "2"})
    res = [json.load(open(tmp_path / f"rank{r}.json")) for r in range(2)]
    assert "injected failure on rank 1" in res[0]["raised"] and "injected failure on rank 1" in res[1]["raised"]
    assert res[0]["own_error"] and "abort" in res[0]["own_error"]          # rank 0's pending call came back with the link's abort error ...
    assert res[0]["seconds"] < 20.0 and res[1]["seconds"] < 20.0           # ... long before the link's 60 s timeout


@pytest.mark.gpu
def test_in_library_loop_and_migration_among_three_ranks_gpu(tmp_path):
    """World 3 over the IPC link: rank 1 is a MIDDLE slab - two different peers, both slots of every pack / unpack-add in use, arrivals from both sides
    in one migration - which neither the world-1 self exchange (both neighbours = this rank) nor the two-rank runs (one neighbour each) execute."""
    _check_moving(_run_ranks("lib", "float64", tmp_path, 3, "moving", env={"SMAC_COMM_STUB": "2"}), 3, 1e-9, 1e-8)


@pytest.mark.gpu
def test_contact_exchange_shortcut_keeps_the_single_domain_result_gpu(tmp_path):
    """`parallel.contact_sides`: a boundary no primitive can reach skips the two contact exchanges.  The strong-scaling bench scene at
    262,144 particles cut in two at the block's centre (the fingers grip its x ends): both ranks drop them, contact still happens inside
    each slab, and states / adjoints / wrenches equal the single-domain HIP run."""
    import slab_worker as W
    from softmac_amd import scenes
    G = W.GRIP_STRONG
    n = G["nsteps"]
    parts = _run_ranks("hip", "float64", tmp_path, 2, "grip_strong")
    assert all((p["sides"] == 0).all() for p in parts) and sum(int(p["hits"]) for p in parts) > 20
    cfg, env_dt, state, specs, s13 = scenes.s_grip(G["particles"], G["grid"], n + 2, "float64", 0)
    sim, prims = H.build_engine(cfg, env_dt, specs, W.grip_strong_states(s13, n + 2, cfg.dt))
    sim.reset(state)
    sim.run_substeps(0, n)
    rng = np.random.default_rng(5)
    N = G["particles"]
    gx, gv = rng.standard_normal((N, 3)), rng.standard_normal((N, 3))
    sim.clear_grads()
    sim.add_grad(n, gx=gx, gv=gv)
    sim.run_substeps_grad(0, n)
    st = sim.get_state(n)
    g = sim.get_grad_full(0)
    assert sorted(np.concatenate([p["idx"] for p in parts]).tolist()) == list(range(N))
    for p in parts:
        i = p["idx"]
        assert H.rel_err(p["st"], st[i]) < 1e-9
        for k, ref in zip(("gx", "gv", "gF", "gC"), g):
            assert np.abs(p[k] - ref[i]).max() < 1e-8 * np.abs(ref).max(), k
    ext = sum(p["ext"] for p in parts)
    assert H.rel_err(ext, np.array([m.ext_f.to_numpy() for m in prims])) < 1e-9


def test_cloth_variant_in_two_slabs_matches_single_domain_cpu(tmp_path):
    """the soft <-> cloth substep under the slab decomposition (world 2, gloo, oracle stand-in engine behind the same
    SlabRunner) against the single-domain cloth oracle: states, adjoints, and the two per-rank partial sums the reductions add up - the force on
    the sheet's vertices and the sheet's vertex adjoints."""
    import slab_worker as W
    from oracle import cloth_oracle as CO
    import scenes_cloth as SC
    import torch
    parts = _run_ranks("oracle", "float64", tmp_path, 2, "cloth")
    sc, P, cloth_fr, contact, seeds, eg, n = W.cloth_scene()
    N, V = len(sc["state"]), len(sc["vertices"])
    x, v, C, F = H.O.state24_split(sc["state"])
    frames, ext = [(x, v, C, F)], torch.zeros(V, 3, dtype=CO.DT)
    for f in range(n):
        prim = SC.oracle_prim(sc, *cloth_fr[f])
        x, v, C, F, e = CO.substep(*frames[-1], P, prim, contact[f][0], contact[f][1], f)
        frames.append((x.detach(), v.detach(), C.detach(), F.detach()))
        ext = ext + e.detach()
    g = [torch.as_tensor(s_) for s_ in seeds]
    cg = []
    for f in range(n - 1, -1, -1):
        out = CO.substep_grad(*frames[f], P, SC.oracle_prim(sc, *cloth_fr[f]), contact[f][0], contact[f][1], f, g[0], g[1], g[2], g[3], ext_f_grad=eg)
        g = [out["gx"], out["gv"], out["gC"], out["gF"]]
        cg.append((out["cloth_pos"].numpy(), out["cloth_vel"].numpy()))
    cg = np.array(cg[::-1])
    assert sum(len(p["idx"]) for p in parts) == N and min(int(p["hits"]) for p in parts) > 20      # both slabs touch the sheet
    ref = dict(zip("xvCF", [t.numpy() for t in frames[n]]))
    refg = dict(gx=g[0].numpy(), gv=g[1].numpy(), gC=g[2].numpy(), gF=g[3].numpy())
    for p in parts:
        for k in "xvCF":
            assert H.rel_err(p[k], ref[k][p["idx"]]) < 1e-11, k
        for k in refg:
            assert np.abs(p[k] - refg[k][p["idx"]]).max() < 1e-9 * np.abs(refg[k]).max(), k
    assert np.abs(ext.numpy()).max() > 0 and H.rel_err(sum(p["ext"] for p in parts), ext.numpy()) < 1e-10
    assert np.abs(cg).max() > 0 and H.rel_err(sum(p["cgrad"] for p in parts), cg) < 1e-9


def test_contact_sides_predicate(monkeypatch):
    from softmac_amd.parallel import contact_sides
    spec = dict(lower=[-0.06, -0.11, -0.06], upper=[0.06, 0.11, 0.06], contact=True)
    ident = [1.0, 0.0, 0.0, 0.0]
    still = np.array([0.30, 0.3, 0.5] + ident + [0.0] * 6)
    # planes 62..65 of a 128 grid = x in [0.484, 0.516): the box reaches 0.36 + 0.005 + 4.5 cells = 0.40 -> free; moved to x = 0.42 it touches
    assert contact_sides([spec], [still], 128, 62, 62, 4, 1, 3) == (False, False)
    near = still.copy(); near[0] = 0.42
    assert contact_sides([spec], [near], 128, 62, 62, 4, 1, 3) == (True, True)
    assert contact_sides([spec], [near], 128, 62, 62, 4, 0, 2) == (False, True)            # no left neighbour
    assert contact_sides([dict(spec, contact=False)], [near], 128, 62, 62, 4, 1, 3) == (False, False)
    moving = np.stack([still + np.concatenate([[0.004 * f, 0, 0], np.zeros(10)]) for f in range(40)])     # slides into reach during the window
    assert contact_sides([spec], [moving], 128, 62, 62, 4, 1, 3) == (True, True)
    turned = still.copy(); turned[3:7] = [np.cos(np.pi / 4), 0.0, 0.0, np.sin(np.pi / 4)]                 # 90 degrees about z: the long axis now lies along x
    turned[0] = 0.345
    assert contact_sides([spec], [turned], 128, 62, 62, 4, 1, 3) == (True, True)
    assert contact_sides([spec], [np.concatenate([[0.345], still[1:]])], 128, 62, 62, 4, 1, 3) == (False, False)
    monkeypatch.setenv("SMAC_SLAB_ALL_EXCHANGES", "1")
    assert contact_sides([spec], [still], 128, 62, 62, 4, 1, 3) == (True, True)


@pytest.mark.parametrize("mode", ["fine", "rank0_fails", "rank1_fails_later"])
def test_rendezvous_failure_reaches_every_rank(tmp_path, mode):
    """bench.py --gpus N brings the in-library RCCL loop up on every rank or on none: a rank 0 that cannot create the RCCL id sends the failure through
    the broadcast the others are waiting in (instead of raising in front of it and leaving them there), and a failure on any one rank afterwards is
    known to all before any of them posts an exchange (parallel.all_ranks_ok) - bench.py then falls back to the Python loop on every rank."""
    import json
    port = str(_free_port())
    procs = [subprocess.Popen([sys.executable, str(HERE / "rendezvous_worker.py"), str(r), "2", port, str(tmp_path), mode]) for r in range(2)]
    for p in procs:
        assert p.wait(timeout=120) == 0
    res = [json.load(open(tmp_path / f"rank{r}.json")) for r in range(2)]
    assert res[0]["failed"] == res[1]["failed"]
    if mode == "fine":
        assert all(r["uid_ok"] and r["err"] is None for r in res) and res[0]["failed"] == []
    elif mode == "rank0_fails":
        assert all(r["err"] and "librccl" in r["err"] for r in res) and len(res[0]["failed"]) == 2
    else:
        assert res[0]["err"] is None and res[0]["failed"] == ["rank 1: SmacError: smac_comm_init failed"]


@pytest.mark.parametrize("mode", ["fine", "rank1_fails", "rank1_fails_peer_pending"])
def test_run_time_failure_on_one_rank_surfaces_on_all(tmp_path, mode):
    """an error inside the collective loop returns on ONE rank (drift, the slab-range guard, a HIP error) while its neighbour waits in the
    matching receive.  The failing rank's library aborts its communicator (softmac_hip.hip slab_guard); a host with a control plane then calls
    parallel.agreed_failure at the end of the window: every rank learns of the failure, drops its own communicator (smac_comm_abort) and raises the
    same error.  (bench.py, which runs under a launcher, ends the failing rank with a non-zero status instead and lets the launcher stop the peers.)"""
    import json
    port = str(_free_port())
    procs = [subprocess.Popen([sys.executable, str(HERE / "agreed_failure_worker.py"), str(r), "2", port, str(tmp_path), mode]) for r in range(2)]
    for p in procs:
        assert p.wait(timeout=120) == 0
    res = [json.load(open(tmp_path / f"rank{r}.json")) for r in range(2)]
    if mode == "fine":
        assert all(r["raised"] is None and not r["aborted"] for r in res)
    else:
        # rank1_fails_peer_pending: rank 0 sits in an exchange rank 1 never answers and cannot reach the all_gather by itself; rank 1 publishes
        # through parallel.FailureWatch, rank 0's watch thread aborts its runner, the pending call returns and both ranks raise - within seconds, not after
        # gloo's timeout
        assert "rank 1:" in res[0]["raised"] and "left the halo" in res[0]["raised"] and "rank 1:" in res[1]["raised"]
        if mode == "rank1_fails":
            assert res[0]["raised"] == res[1]["raised"]
        else:
            assert "rank 0:" in res[0]["raised"] and "aborted" in res[0]["raised"] and res[0]["seconds"] < 20.0
        assert all(r["aborted"] for r in res)
