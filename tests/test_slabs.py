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


def _run_two_ranks(engine, precision, tmp_path):
    port = _free_port()
    procs = [subprocess.Popen([sys.executable, str(HERE / "slab_worker.py"), "--engine", engine, "--rank", str(r), "--world", "2",
                               "--port", str(port), "--out", str(tmp_path), "--precision", precision]) for r in range(2)]
    for p in procs:
        assert p.wait(timeout=600) == 0
    return [np.load(tmp_path / f"rank{r}.npz") for r in range(2)]


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
