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


def _run_ranks(engine, precision, tmp_path, world=2, scene="static"):
    port = _free_port()
    procs = [subprocess.Popen([sys.executable, str(HERE / "slab_worker.py"), "--engine", engine, "--rank", str(r), "--world", str(world),
                               "--port", str(port), "--out", str(tmp_path), "--precision", precision, "--scene", scene]) for r in range(world)]
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
    for p in parts:
        for sl in (slice(0, 3), slice(3, 6), slice(6, 15), slice(15, 24)):
            lim = tol_state
            if sl.start == 15 and tol_state > 1e-8:             # C in float32: difference quotient of a 20 m/s velocity field (helpers.c_tol)
                mine = ref_end[p["ids_end"]]                     # (rel_err divides by THIS rank's largest |C|: so must the absolute floor)
                lim = H.c_tol(tol_state, sc["n_grid"], mine[:, 3:6], mine[:, 15:24])
            assert H.rel_err(p["st_end"][:, sl], ref_end[p["ids_end"]][:, sl]) < lim
            assert H.rel_err(p["g0"][:, sl], ref_g0[p["ids0"]][:, sl]) < tol_grad * (np.abs(ref_g0[:, sl]).max() / max(np.abs(ref_g0[p["ids0"]][:, sl]).max(), 1e-300))


@pytest.mark.parametrize("world", [2, 4])
def test_migration_between_slabs_matches_single_domain_cpu(tmp_path, world):
    _check_moving(_run_ranks("oracle", "float64", tmp_path, world, "moving"), world, 1e-11, 1e-9)


@pytest.mark.gpu
# (f32 gradients: the cloud flies at 20 m/s, 850 x its velocity-gradient scale - the cancellation of helpers.c_tol enters the adjoint too)
@pytest.mark.parametrize("precision,ts,tg", [("float64", 1e-9, 1e-8), ("float32", H.F32_TOL["state"], 2e-4)])
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
    for p in parts:
        idx = p["idx"]
        for k in "xvCF":
            assert H.rel_err(p[k], ref[k][idx]) < tol_state, k
        for k in refg:
            assert H.rel_err(p[k], refg[k][idx]) < tol_grad, k
    ext = sum(p["ext"] for p in parts)
    assert H.rel_err(ext, np.sum(np.array(orc.ext), axis=0)) < max(tol_state * 100, 1e-8)
    pgr = sum(p["pgrad"] for p in parts)
    assert H.rel_err(pgr, np.array(pg)[:-1]) < tol_grad * 10


def test_two_slabs_match_single_domain_cpu(tmp_path):
    _check(_run_two_ranks("oracle", "float64", tmp_path), 1e-11, 1e-9)


@pytest.mark.gpu
@pytest.mark.parametrize("precision,ts,tg", [("float64", 1e-9, 1e-8), ("float32", H.F32_TOL["state"], 5e-5)])
def test_two_slabs_match_single_domain_gpu(tmp_path, precision, ts, tg):
    _check(_run_two_ranks("hip", precision, tmp_path), ts, tg)
