"""oracle/mixed_oracle.py is a COMPOSITION of the two restatements (rigid primitives in index order, then the sheet; two-entry material table).  What
can be pinned without a reference that has such a simulator: it reduces to each of the two oracles when the other primitive is absent, the order of
the chain is the stated one, the material selector picks per particle, and its autograd adjoint agrees with central finite differences."""
import dataclasses

import numpy as np
import torch

import helpers as H
import scenes_cloth as S
from helpers import O
from oracle import cloth_oracle as CO
from oracle import mixed_oracle as MO


def _scene(N=300):
    sc = S.build("hit", "float64", n_env_steps=1, N=N, seed=4)
    c = sc["cfg"]
    c.ptype, c.n_controllers, c.E, c.yield_stress, c.gravity = 0, 0, 2000.0, 30.0, (0.0, -3.0, 0.0)
    P = S.oracle_params(sc)
    P2 = dataclasses.replace(P, E=800.0, nu=0.3, yield_stress=12.0)
    palm = H.load_palm()
    st = np.concatenate([[0.5, 0.248, 0.53], [1.0, 0.0, 0.0, 0.0], [0.0, 0.3, 0.05], [0.2, 0.0, 0.1]])
    rigid = O.make_prim(st[:3], st[3:7], st[7:10], st[10:13], palm["sdf"], palm["normal"], palm["lower"], palm["upper"], palm["dx"], 0.6, 666.0, True)
    sheet = S.oracle_prim(sc, *sc["motion"](0.0))
    x = sc["state"][:, :3]
    ids = CO.get_contact_pair(x, sc["motion"](0.0)[0], sc["faces"], None, 1.0)
    pen = ((np.arange(N) % 7 == 0) & (ids >= 0)).astype(np.int8)
    mat = (x[:, 0] > 0.5).astype(np.int32)
    return sc, P, P2, rigid, sheet, ids, pen, mat


def test_reduces_to_the_two_oracles_and_keeps_the_stated_order():
    sc, P, P2, rigid, sheet, ids, pen, mat = _scene()
    fr = O.state24_split(sc["state"])
    # sheet only = the cloth oracle
    a = MO.substep(*fr, P, [], sheet, ids, pen, 0)
    b = CO.substep(*fr, P, sheet, ids, pen, 0)
    for u, w in zip(a[:4], b[:4]):
        assert (u - w).abs().max() < 1e-13
    assert (a[5] - b[4]).abs().max() < 1e-12 * max(1.0, b[4].abs().max())
    # rigid only: mixed3 = softmac's loop on the cloth oracle's substep (von Mises, walls only): the rigid wrench is non-zero, the sheet is absent
    r = MO.substep(*fr, P, [rigid], None, -np.ones(len(ids), dtype=np.int64), np.zeros(len(ids), dtype=np.int8), 0)
    assert r[5] is None and r[4][0].abs().max() > 0
    none = MO.substep(*fr, P, [], None, ids, pen, 0)
    assert (r[1] - none[1]).abs().max() > 1e-6                                   # the palm does act
    # both: particles in reach of both primitives get the sheet applied to the velocity the palm left (not to v_tmp): swapping the order changes them
    both = MO.substep(*fr, P, [rigid], sheet, ids, pen, 0)
    band = (O.prim_sdf(rigid, fr[0]) <= 5e-3).numpy() & (ids >= 0)
    assert band.sum() >= 1
    assert (both[1] - a[1]).abs().max() > 1e-6 and (both[1] - r[1]).abs().max() > 1e-6
    # the material selector: entry 1 everywhere = the oracle run with P2; a mixed selector differs from both on the particles of the other entry
    all2 = MO.substep(*fr, P, [rigid], sheet, ids, pen, 0, P2, np.ones(len(ids), dtype=np.int32))
    p2 = MO.substep(*fr, P2, [rigid], sheet, ids, pen, 0)
    assert (all2[3] - p2[3]).abs().max() < 1e-13 and (all2[0] - p2[0]).abs().max() < 1e-13
    mix = MO.substep(*fr, P, [rigid], sheet, ids, pen, 0, P2, mat)
    assert (mix[3][mat == 0] - both[3][mat == 0]).abs().max() < 1e-13 and (mix[3][mat == 1] - p2[3][mat == 1]).abs().max() < 1e-13


def test_adjoint_against_central_differences():
    sc, P, P2, rigid, sheet, ids, pen, mat = _scene(N=150)
    fr = O.state24_split(sc["state"])
    rng = np.random.default_rng(1)
    N, V = fr[0].shape[0], sheet.position.shape[0]
    seeds = [torch.as_tensor(rng.standard_normal(t.shape)) for t in fr]
    er, ec = rng.standard_normal(6) * 1e-3, rng.standard_normal((V, 3)) * 1e-3

    def loss(x, v, C, F, rp=None, sp=None):
        rg = rigid if rp is None else dataclasses.replace(rigid, position=rp)
        sh = sheet if sp is None else dataclasses.replace(sheet, position=sp)
        out = MO.substep(x, v, C, F, P, [rg], sh, ids, pen, 0, P2, mat)
        tot = sum((o * s).sum() for o, s in zip(out[:4], seeds))
        return float(tot + (out[4][0] * torch.as_tensor(er)).sum() + (out[5] * torch.as_tensor(ec)).sum())
    g = MO.substep_grad(*fr, P, [rigid], sheet, ids, pen, 0, *seeds, ext_r_grad=[er], ext_c_grad=ec, P2=P2, mat_id=mat)
    h = 1e-6
    for name, k, ref in (("x", 0, g["gx"]), ("v", 1, g["gv"])):
        d = torch.as_tensor(rng.standard_normal(fr[k].shape))
        args_p = [t.clone() for t in fr]; args_m = [t.clone() for t in fr]
        args_p[k] = fr[k] + h * d; args_m[k] = fr[k] - h * d
        fd = (loss(*args_p) - loss(*args_m)) / (2 * h)
        an = float((ref * d).sum())
        assert abs(fd - an) < 2e-5 * max(1.0, abs(an)), (name, fd, an)
    d = torch.as_tensor(rng.standard_normal(3))
    fd = (loss(*fr, rp=rigid.position + h * d) - loss(*fr, rp=rigid.position - h * d)) / (2 * h)
    an = float((g["prims"][0][:3] * d).sum())
    assert abs(fd - an) < 2e-5 * max(1.0, abs(an)), ("rigid position", fd, an)
    d = torch.as_tensor(rng.standard_normal((V, 3)))
    fd = (loss(*fr, sp=sheet.position + h * d) - loss(*fr, sp=sheet.position - h * d)) / (2 * h)
    an = float((g["sheet_pos"] * d).sum())
    assert abs(fd - an) < 2e-5 * max(1.0, abs(an)), ("sheet position", fd, an)
