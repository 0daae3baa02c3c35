"""oracle/mpm_cpu.cpp (plain C++/OpenMP f64, the CPU baseline) against oracle/softmac_oracle.py (torch f64 +
autograd): two independent restatements of the reference must agree to rounding."""
import numpy as np
import pytest
import torch

import helpers as H
import scenes_golden as G
from helpers import O
from oracle import mpm_cpu


@pytest.mark.parametrize("ptype,model", [(0, 0), (1, 0), (2, 0), (0, 1), (1, 1), (2, 1)])
def test_materials(ptype, model):
    P = O.SimParams(n_grid=32, dt=2e-4, ptype=ptype, material_model=model, E=22.0 if ptype == 2 else 3e3, ground_friction=20.0)
    st = H.make_cloud(800, 32, seed=ptype + 5 * model, lo=(0.3, 0.05, 0.3), hi=(0.7, 0.4, 0.7))
    x, v, C, F = O.state24_split(st)
    port = mpm_cpu.CpuPort(P)
    nx, nv, nC, nF, _ = port.substep(0, x.numpy(), v.numpy(), C.numpy(), F.numpy())
    rx, rv, rC, rF, _ = O.substep(x, v, C, F, P)
    for a, b in ((nx, rx), (nv, rv), (nC, rC), (nF, rF)):
        assert H.rel_err(a, b.numpy()) < 1e-11
    rng = np.random.default_rng(1)
    g = [rng.standard_normal(t.shape) for t in (nx, nv, nC, nF)]
    ref = O.substep_grad(x, v, C, F, P, (), 0, *[torch.tensor(a) for a in g])
    got = port.substep_grad(0, x.numpy(), v.numpy(), C.numpy(), F.numpy(), *g)
    for a, k in zip(got[:4], ("gx", "gv", "gC", "gF")):
        assert H.rel_err(a, ref[k].numpy()) < 1e-9, k


def test_forecast_contact_reference_fixture():
    sc = G.build("grip_contact")
    P = H.oracle_params(sc["cfg"], sc["env_dt"])
    x, v, C, F = O.state24_split(sc["state"])
    prims = H.OracleRollout(P, sc["state"], sc["specs"], sc["pstates"]).prims_at(1)
    pst = np.array(sc["pstates"][1])
    port = mpm_cpu.CpuPort(P, sc["specs"])
    nx, nv, nC, nF, ext = port.substep(1, x.numpy(), v.numpy(), C.numpy(), F.numpy(), pst)
    rx, rv, rC, rF, rext = O.substep(x, v, C, F, P, prims, 1)
    for a, b in ((nx, rx), (nv, rv), (nC, rC), (nF, rF)):
        assert H.rel_err(a, b.numpy()) < 1e-11
    assert H.rel_err(ext[0], rext[0].numpy()) < 1e-10
    rng = np.random.default_rng(2)
    g = [rng.standard_normal(t.shape) for t in (nx, nv, nC, nF)]
    eg = [rng.standard_normal(6)]
    ref = O.substep_grad(x, v, C, F, P, prims, 1, *[torch.tensor(a) for a in g], ext_f_grad=[torch.tensor(eg[0])])
    got = port.substep_grad(1, x.numpy(), v.numpy(), C.numpy(), F.numpy(), *g, pst=pst, ext_f_grad=eg)
    for a, k in zip(got[:4], ("gx", "gv", "gC", "gF")):
        assert H.rel_err(a, ref[k].numpy()) < 1e-9, k
    assert H.rel_err(got[4][0], torch.cat(ref["prims"][0]).numpy()) < 1e-9
