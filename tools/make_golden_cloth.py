#!/usr/bin/env python3
"""Golden vectors of the cloth oracle (oracle/cloth_oracle.py), made here in the build container: for the two demo-shaped scenes of
tests/scenes_cloth.py a 3-substep rollout with fixed contact faces / penetration flags - last frame, accumulated sheet force, adjoints at
frame 0 and on the sheet for fixed seeds.  tests/test_cloth_oracle.py re-derives them on the CPU (regression pin of the oracle);
tests/test_gpu_cloth.py compares the HIP path with them on the GPU box."""
import pathlib
import sys

import numpy as np
import torch

ROOT = pathlib.Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
import helpers as H  # noqa: E402
import scenes_cloth as S  # noqa: E402
from oracle import cloth_oracle as CO  # noqa: E402

N_STEPS = 3


def run(kind, ctype=2, N=800):
    sc = S.build(kind, "float64", N=N, seed=21, collision_type=ctype)
    P = S.oracle_params(sc)
    V = len(sc["vertices"])
    cloth = [sc["motion"](f * sc["cfg"].dt) for f in range(N_STEPS + 1)]
    x, v, C, F = CO.O.state24_split(sc["state"])
    ids = CO.get_contact_pair(x, cloth[0][0], sc["faces"], None, sc["scale"])
    rng = np.random.default_rng(22)
    pen = ((rng.uniform(size=N) < 0.15) & (ids >= 0)).astype(np.int8)
    ci = None if sc["control_idx"] is None else torch.as_tensor(sc["control_idx"], dtype=torch.int64)
    act = None if sc["action"] is None else torch.as_tensor(sc["action"], dtype=CO.DT)
    frames, ext = [(x, v, C, F)], np.zeros((V, 3))
    for f in range(N_STEPS):
        out = CO.substep(*frames[-1], P, S.oracle_prim(sc, *cloth[f]), ids, pen, f, ci, act)
        frames.append(tuple(t.detach() for t in out[:4]))
        ext += out[4].detach().numpy()
    seeds = [rng.standard_normal((N, 3)), rng.standard_normal((N, 3)), 0.01 * rng.standard_normal((N, 3, 3)), 0.01 * rng.standard_normal((N, 3, 3))]
    eg = rng.standard_normal((V, 3)) * 1e-2 / P.p_mass * P.dt
    adj = tuple(torch.as_tensor(s) for s in seeds)
    cp, cv, ag = [], [], []
    for f in range(N_STEPS - 1, -1, -1):
        g = CO.substep_grad(*frames[f], P, S.oracle_prim(sc, *cloth[f]), ids, pen, f, *adj, ext_f_grad=eg, control_idx=ci, action=act)
        adj = (g["gx"], g["gv"], g["gC"], g["gF"])
        cp.insert(0, g["cloth_pos"].numpy()); cv.insert(0, g["cloth_vel"].numpy())
        ag.insert(0, np.zeros((1, 3)) if g["action"] is None else g["action"].numpy())
    xl, vl, Cl, Fl = (t.numpy() for t in frames[-1])
    return dict(contact_id=ids, penetration=pen, seed_gx=seeds[0], seed_gv=seeds[1], seed_gC=seeds[2], seed_gF=seeds[3], ext_f_grad=eg,
                x=xl, v=vl, C=Cl, F=Fl, ext_f=ext, gx0=adj[0].numpy(), gv0=adj[1].numpy(), gC0=adj[2].numpy(), gF0=adj[3].numpy(),
                cloth_pos_grad=np.stack(cp), cloth_vel_grad=np.stack(cv), action_grad=np.stack(ag))


CASES = {"taco": ("taco", 2), "hit": ("hit", 2), "hit_penalty": ("hit", 1)}

if __name__ == "__main__":
    for name, (kind, ctype) in CASES.items():
        out = run(kind, ctype)
        path = H.GOLDEN / f"oracle_cloth_{name}.npz"
        np.savez_compressed(path, **out)
        print(name, path.stat().st_size, int((out["contact_id"] >= 0).sum()), "contact particles")
