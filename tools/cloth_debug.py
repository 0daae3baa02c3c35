"""bisect the taco-scene mismatch: one substep, f64, per-field errors for variants of the scene"""
import sys, pathlib
ROOT = pathlib.Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
import numpy as np, torch
import helpers as H, scenes_cloth as S
from oracle import cloth_oracle as CO

def run(tag, kind="taco", scale=None, ptype=None, ctype=None, gravity=None, sticky=None, n=1):
    sc = S.build(kind, "float64")
    if scale is not None:
        f = scale / sc["scale"]
        sc["state"][:, 0:6] *= f; sc["vertices"] = sc["vertices"] * f; sc["scale"] = scale
        rest = sc["vertices"]; sc["motion"] = S.sheet_motion(kind, rest, scale)
    if ptype is not None: sc["cfg"].ptype = ptype
    if ctype is not None: sc["cfg"].collision_type = ctype
    if gravity is not None: sc["cfg"].gravity = gravity
    if sticky is not None: sc["prim"]["sticky"] = sticky
    sim, prim = S.build_engine(sc)
    P = S.oracle_params(sc)
    N = len(sc["state"])
    cloth = [sc["motion"](f * sc["cfg"].dt) for f in range(n + 1)]
    for f in range(n + 1): prim.set_all_states(f, *cloth[f])
    sim.reset(sc["state"])
    sim.get_contact_pair(0)
    ids0, _ = sim.get_contact(0)
    pen = np.zeros(N, dtype=np.int8)
    x, v, C, F = CO.O.state24_split(sc["state"])
    for f in range(n):
        sim.set_contact(f, ids0, pen)
        x, v, C, F, ext = CO.substep(x, v, C, F, P, S.oracle_prim(sc, *cloth[f]), ids0, pen, f)
        sim.substep(f, None)
    st = sim.get_state(n)
    e = lambda a, b: H.rel_err(a, b.detach().numpy().reshape(N, -1))
    print(f"{tag:34s} contacts {(ids0>=0).sum():5d} x {e(st[:,0:3],x):.2e} v {e(st[:,3:6],v):.2e} F {e(st[:,6:15],F):.2e} C {e(st[:,15:24],C):.2e} ext {H.rel_err(prim.ext_f.to_numpy(), ext.detach().numpy()):.2e}", flush=True)

run("taco as is")
run("taco no contact", ctype=0)
run("taco elastic", ptype=1)
run("taco no gravity", gravity=(0.0, 0.0, 0.0))
run("taco scale 1", scale=1.0)
run("taco not sticky", sticky=False)
run("taco no contact elastic nograv", ctype=0, ptype=1, gravity=(0.0, 0.0, 0.0))
run("hit as is", kind="hit")
run("hit scale 3", kind="hit", scale=3.0)
