#!/usr/bin/env python3
"""Where does the device's rounding error sit?  Runs small parity scenes on the GPU (through the C ABI) against the
f64 oracle and reports, per field, the max-norm and L2 relative errors plus the worst particles with what makes them
special (singular-value gaps of F_tmp, plastic clip status, contact band).  Diagnostics only - never asserts.

    SMAC_LIB=path/to/variant.so python tools/prec_probe.py [--precision float32] [--out gpurun_out/prec.json]
"""
from __future__ import annotations

import argparse
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import helpers as H  # noqa: E402
import scenes_golden as G  # noqa: E402


def scenes():
    out = []
    st2k = np.load(H.GOLDEN / "grip_state_2k.npz")["state"]
    out.append(dict(name="grip2k_plastic", cfg=H.sim_cfg(len(st2k), n_grid=64, dt=2e-4, ptype=0), env_dt=1e-3, state=st2k, n=4,
                    specs=[], pst=None, eg=None))
    sc = G.build("grip_contact")
    out.append(dict(name="grip2k_contact", cfg=sc["cfg"], env_dt=sc["env_dt"], state=sc["state"], n=3, specs=sc["specs"],
                    pst=sc["pstates"], eg=sc["ext_f_grad"]))
    for ct in (1, 0):
        c2 = H.sim_cfg(len(st2k), n_grid=64, dt=2e-4, ptype=0, collision_type=ct)
        out.append(dict(name=f"grip2k_ctype{ct}", cfg=c2, env_dt=1e-3, state=st2k, n=3, specs=sc["specs"], pst=sc["pstates"],
                        eg=[np.random.default_rng(6).standard_normal(6) * 1e-2]))
    for ptype, model in ((1, 0), (0, 0), (2, 0), (0, 1)):
        N, ng = 3000, 32
        cfg = H.sim_cfg(N, n_grid=ng, dt=2e-4, ptype=ptype, material_model=model, ground_friction=0.0, E=3e3 if ptype != 2 else 22.0)
        out.append(dict(name=f"cloud_p{ptype}m{model}", cfg=cfg, env_dt=1e-3,
                        state=H.make_cloud(N, ng, seed=ptype * 2 + model, lo=(0.3, 0.05, 0.3), hi=(0.7, 0.4, 0.7)), n=3, specs=[], pst=None, eg=None))
    pour = np.load(H.GOLDEN / "pour_state_1k.npz")["state"]
    out.append(dict(name="pour1k_liquid", cfg=H.sim_cfg(len(pour), n_grid=64, dt=1e-3, E=22.0, ptype=2, ground_friction=0.0), env_dt=1e-3,
                    state=pour, n=3, specs=[], pst=None, eg=None))
    return out


def describe(sc, orc, idx):
    """what is special about particle idx: smallest singular-value gap of F_tmp over the window, clip status, cell coordinates"""
    dt = sc["cfg"].dt
    gmin, clipped = 1e30, False
    for f in range(len(orc.frames) - 1):
        x, v, C, F = orc.frames[f]
        Ft = (np.eye(3) + dt * C[idx].numpy()) @ F[idx].numpy()
        s = np.linalg.svd(Ft, compute_uv=False)
        e = np.sort(s - 1.0)
        gmin = min(gmin, float(np.diff(e).min()))
        clipped |= bool((e < -2e-3).any() or (e > 3e-3).any())
    x0 = orc.frames[0][0][idx].numpy() * sc["cfg"].n_grid
    return dict(e=[float(a) for a in e], min_gap=gmin, clipped=clipped, cell=[round(float(a), 3) for a in x0],
                v=[round(float(a), 3) for a in orc.frames[0][1][idx].numpy()])


def run(sc, precision):
    cfg = sc["cfg"]
    cfg.precision = precision
    P = H.oracle_params(cfg, sc["env_dt"])
    n = sc["n"]
    orc = H.OracleRollout(P, sc["state"], sc["specs"], sc["pst"]).forward(n)
    sim, prims = H.build_engine(cfg, sc["env_dt"], sc["specs"], sc["pst"])
    sim.reset(sc["state"])
    sim.run_substeps(0, n)
    N = cfg.n_particles
    rep = {}

    def fields(got, ref, tag):
        got, ref = np.asarray(got, dtype=np.float64).reshape(N, -1), np.asarray(ref, dtype=np.float64).reshape(N, -1)
        scale = np.abs(ref).max()
        per = np.abs(got - ref).max(axis=1) / (scale if scale > 0 else 1.0)
        worst = np.argsort(per)[::-1][:4]
        rep[tag] = dict(max=float(per.max()), l2=float(np.linalg.norm(got - ref) / max(np.linalg.norm(ref), 1e-300)),
                        n_over_1e5=int((per > 1e-5).sum()), n_over_1e4=int((per > 1e-4).sum()),
                        worst=[dict(i=int(i), err=float(per[i]), **describe(sc, orc, int(i))) for i in worst])

    for f in (1, n):
        st = sim.get_state(f)
        x, v, C, F = orc.frames[f]
        fields(st[:, 0:3], x.numpy(), f"x[{f}]")
        fields(st[:, 3:6], v.numpy(), f"v[{f}]")
        fields(st[:, 6:15], F.numpy(), f"F[{f}]")
        fields(st[:, 15:24], C.numpy(), f"C[{f}]")
    rng = np.random.default_rng(100)
    seeds = {n: (rng.standard_normal((N, 3)), rng.standard_normal((N, 3)), 0.01 * rng.standard_normal((N, 3, 3)), 0.01 * rng.standard_normal((N, 3, 3))),
             1: (rng.standard_normal((N, 3)), None, None, None)}
    adj, pg, _ = orc.backward(seeds, sc["eg"])
    sim.clear_grads()
    for f, s in seeds.items():
        sim.add_grad(f, gx=s[0], gv=s[1], gC=s[2], gF=s[3])
    # per-frame gradient errors: where along the backward chain does the error enter?
    for f in range(n - 1, -1, -1):
        sim.substep_grad(f, None, sc["eg"])
        if f in (n - 1, 0):
            gx, gv, gF, gC = sim.get_grad_full(f)
            fields(gx, adj[f][0].numpy(), f"gx[{f}]")
            fields(gv, adj[f][1].numpy(), f"gv[{f}]")
            fields(gC, adj[f][2].numpy(), f"gC[{f}]")
            fields(gF, adj[f][3].numpy(), f"gF[{f}]")
    if sc["specs"]:
        ext_ref = np.sum(np.array(orc.ext), axis=0)
        got = np.array([m.ext_f.to_numpy() for m in prims])
        rep["ext_f"] = dict(max=float(np.abs(got - ext_ref).max() / max(np.abs(ext_ref).max(), 1e-300)))
        pgerr = 0.0
        for f in range(n):
            for i, m in enumerate(prims):
                ref = pg[f][i]
                pgerr = max(pgerr, float(np.abs(m.get_all_states_grad(f) - ref).max() / max(np.abs(ref).max(), 1e-9)))
        rep["prim_grad"] = dict(max=pgerr)
        st0 = sim.get_state(0)
        rep["n_contact"] = int(sim.contact_counts()[0])
    return rep


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--precision", default="float32")
    ap.add_argument("--out", default="")
    ap.add_argument("--only", default="")
    ap.add_argument("--fuzz", default="", help="comma-separated case ids of tests/test_gpu_fuzz.py to probe instead of the fixed scenes")
    args = ap.parse_args()
    res = {"lib": os.environ.get("SMAC_LIB", "default"), "precision": args.precision, "scenes": {}}
    todo = scenes()
    if args.fuzz:
        import test_gpu_fuzz as FZ
        todo = []
        for c in [int(a) for a in args.fuzz.split(",")]:
            fc = FZ.fuzz_case(c)
            cfg = fc["cfg"]
            print(f"fuzz {c}: N {cfg.n_particles} ptype {cfg.ptype} model {cfg.material_model} ctype {cfg.collision_type} prim {fc['with_prim']} "
                  f"wall {fc['near_wall']} gf {cfg.ground_friction} g {cfg.gravity} steps {fc['steps']} sort {cfg.sort_interval}", flush=True)
            todo.append(dict(name=f"fuzz{c}", cfg=cfg, env_dt=2e-3, state=fc["state"], n=fc["steps"], specs=list(fc["specs"]), pst=fc["pstates"], eg=None))
    for sc in todo:
        if args.only and args.only not in sc["name"]:
            continue
        r = run(sc, args.precision)
        res["scenes"][sc["name"]] = r
        line = " ".join(f"{k}={v['max']:.1e}" for k, v in r.items() if isinstance(v, dict) and "max" in v)
        print(f"[{res['lib']}] {sc['name']}: {line}", flush=True)
        for k in ("gF[0]", "gx[0]", "v[%d]" % sc["n"], "C[%d]" % sc["n"]):
            if k in r:
                print("    ", k, "n>1e-5:", r[k]["n_over_1e5"], "l2", f"{r[k]['l2']:.1e}", "worst:",
                      [(w["i"], f"{w['err']:.1e}", f"gap {w['min_gap']:.1e}", "clip" if w["clipped"] else "", w["cell"], w["v"]) for w in r[k]["worst"][:3]], flush=True)
    if args.out:
        os.makedirs(os.path.dirname(args.out) or ".", exist_ok=True)
        json.dump(res, open(args.out, "w"), indent=1)


if __name__ == "__main__":
    main()
