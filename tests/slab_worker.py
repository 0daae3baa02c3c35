"""One rank of the 2-slab test (spawned by tests/test_slabs.py): runs SlabRunner over either the oracle
stand-in engine (CPU, gloo) or the real HIP engine (GPU box; gloo staging through the host because both ranks
share the one GPU), and writes its particles' results to <out>/rank<r>.npz."""
import argparse
import os
import pathlib
import sys

import numpy as np
import torch
import torch.distributed as dist

HERE = pathlib.Path(__file__).resolve().parent
sys.path.insert(0, str(HERE)); sys.path.insert(0, str(HERE.parent))
import helpers as H  # noqa: E402
import scenes_slab as S  # noqa: E402
from softmac_amd.parallel import SlabRunner  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--engine", default="oracle"); ap.add_argument("--rank", type=int); ap.add_argument("--world", type=int)
    ap.add_argument("--port", type=int); ap.add_argument("--out"); ap.add_argument("--precision", default="float64")
    ap.add_argument("--scene", default="static")
    a = ap.parse_args()
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(a.port))
    dist.init_process_group("gloo", rank=a.rank, world_size=a.world)
    if a.scene == "moving":
        return moving(a)
    if a.scene == "grip_strong":
        return grip_strong(a)
    if a.scene == "cloth":
        return cloth(a)
    if a.scene == "lib_failure":
        return lib_failure(a)
    sc = S.build(a.precision)
    idx = S.owned(sc, a.rank, a.world)
    state = sc["state"][idx]
    nsteps = sc["nsteps"]
    seeds = S.seeds(sc)
    if a.engine == "oracle":
        import slab_engines as E
        eng = E.OracleSlabEngine(H.oracle_params(sc["cfg"], sc["env_dt"]), state, sc["specs"], sc["pstates"])
        run = SlabRunner(eng, a.rank, a.world, sc["split"], sc["split"], 2, has_contact=bool(sc["specs"]))
        run.run_substeps(0, nsteps)
        for f, s in seeds.items():
            adj = eng.get_adj(f)
            for k in range(4):
                if s[k] is not None:
                    adj[k] = adj[k] + torch.as_tensor(s[k][idx]).reshape(adj[k].shape)
        run.run_substeps_grad(0, nsteps, sc["ext_f_grad"])
        x, v, C, F = eng.frames[nsteps]
        g = eng.get_adj(0)
        out = dict(x=x.numpy(), v=v.numpy(), C=C.numpy(), F=F.numpy(), gx=g[0].numpy(), gv=g[1].numpy(), gC=g[2].numpy(), gF=g[3].numpy(),
                   ext=np.array([e.numpy() for e in eng.ext]) if sc["specs"] else np.zeros((0, 6)),
                   pgrad=np.array([eng.pgrad.get(f, [np.zeros(13)] * len(sc["specs"])) for f in range(nsteps)]))
    elif a.engine == "lib":
        # the slab loop INSIDE the library between two real ranks on the one GPU: SMAC_COMM_STUB=2 (set by the test) swaps RCCL for the IPC link of
        # smac_comm.hpp - distinct peers, left / right slot mapping, two-sided pack / unpack-add, the reductions of the primitives' sums
        from softmac_amd.parallel import LibSlabRunner, rendezvous_unique_id
        assert os.environ.get("SMAC_COMM_STUB") == "2"
        cfg = sc["cfg"]
        cfg.n_particles = len(state)
        sim, prims = H.build_engine(cfg, sc["env_dt"], sc["specs"], sc["pstates"])
        sim.reset(state)
        run = LibSlabRunner(sim, a.rank, a.world, sc["split"], sc["split"], 2, has_contact=(True, True), unique_id=rendezvous_unique_id(a.rank))
        run.run_substeps(0, nsteps)
        sim.clear_grads()
        for f, s in seeds.items():
            sim.add_grad(f, **{k: (None if s[i] is None else s[i][idx]) for i, k in enumerate(("gx", "gv", "gC", "gF"))})
        run.run_substeps_grad(0, nsteps, sc["ext_f_grad"])
        st = sim.get_state(nsteps)
        N = len(state)
        gx, gv, gF, gC = sim.get_grad_full(0)
        out = dict(x=st[:, 0:3], v=st[:, 3:6], F=st[:, 6:15].reshape(N, 3, 3), C=st[:, 15:24].reshape(N, 3, 3), gx=gx, gv=gv, gC=gC, gF=gF,
                   ext=np.array([m.ext_f.to_numpy() for m in prims]), pgrad=np.array([[m.get_all_states_grad(f) for m in prims] for f in range(nsteps)]),
                   exchanges=run.exchanges())
        out["ext_total"] = run.allreduce_ext_f()                  # the all-reduces over the link: every rank then holds the sums
        run.allreduce_state_grad(0, nsteps)
        out["pgrad_total"] = np.array([[m.get_all_states_grad(f) for m in prims] for f in range(nsteps)])
        run.close()
    else:
        from softmac_amd.parallel import HipSlabEngine
        cfg = sc["cfg"]
        cfg.n_particles = len(state)
        sim, prims = H.build_engine(cfg, sc["env_dt"], sc["specs"], sc["pstates"])
        sim.reset(state)
        eng = HipSlabEngine(sim, use_torch_stream=True)
        run = SlabRunner(eng, a.rank, a.world, sc["split"], sc["split"], 2, has_contact=bool(sc["specs"]))
        run.run_substeps(0, nsteps)
        sim.clear_grads()
        for f, s in seeds.items():
            sim.add_grad(f, **{k: (None if s[i] is None else s[i][idx]) for i, k in enumerate(("gx", "gv", "gC", "gF"))})
        run.run_substeps_grad(0, nsteps, sc["ext_f_grad"])
        st = sim.get_state(nsteps)
        N = len(state)
        gx, gv, gF, gC = sim.get_grad_full(0)
        out = dict(x=st[:, 0:3], v=st[:, 3:6], F=st[:, 6:15].reshape(N, 3, 3), C=st[:, 15:24].reshape(N, 3, 3), gx=gx, gv=gv, gC=gC, gF=gF,
                   ext=np.array([m.ext_f.to_numpy() for m in prims]) if sc["specs"] else np.zeros((0, 6)),
                   pgrad=np.array([[m.get_all_states_grad(f) for m in prims] for f in range(nsteps)]))
    np.savez(pathlib.Path(a.out) / f"rank{a.rank}.npz", idx=idx, **out)
    dist.barrier()
    dist.destroy_process_group()


def lib_failure(a):
    """rank 1 fails before its first exchange and publishes through parallel.FailureWatch; rank 0 is INSIDE smac_substeps_slab, waiting at the
    IPC link's pair barrier for a neighbour that never comes (the test transport's form of a pending receive).  Its watch thread sees the file and calls
    smac_comm_abort from the second host thread; the barrier wait returns with an error and both ranks raise through agreed_failure within seconds - the
    link's own timeout is 60 s."""
    import json
    import time
    from softmac_amd.parallel import FailureWatch, LibSlabRunner, agreed_failure, rendezvous_unique_id
    assert os.environ.get("SMAC_COMM_STUB") == "2"
    sc = S.build(a.precision)
    idx = S.owned(sc, a.rank, a.world)
    cfg = sc["cfg"]
    cfg.n_particles = len(idx)
    sim, prims = H.build_engine(cfg, sc["env_dt"], sc["specs"], sc["pstates"])
    sim.reset(sc["state"][idx])
    run = LibSlabRunner(sim, a.rank, a.world, sc["split"], sc["split"], 2, has_contact=(True, True), unique_id=rendezvous_unique_id(a.rank))
    watch = FailureWatch(a.rank, a.world, runner=run, directory=a.out)
    dist.barrier()
    t0, err = time.monotonic(), None
    try:
        if a.rank == 1:
            raise RuntimeError("injected failure on rank 1 before its first exchange")
        run.run_substeps(0, sc["nsteps"])
        sim.sync()
    except Exception as e:                                         # noqa: BLE001
        err = f"{type(e).__name__}: {e}"
    raised = None
    try:
        agreed_failure(err, run, watch=watch)
    except RuntimeError as e:
        raised = str(e)
    json.dump({"raised": raised, "own_error": err, "seconds": time.monotonic() - t0}, open(pathlib.Path(a.out) / f"rank{a.rank}.json", "w"))
    watch.close()
    np.savez(pathlib.Path(a.out) / f"rank{a.rank}.npz", idx=idx)
    dist.barrier()
    dist.destroy_process_group()


def cloth_scene(n=3):
    """the taco scene of tests/scenes_cloth.py (von-Mises plasticine on the sticky tortilla, scale 5) cut at the disc's centre plane; contact faces
    searched per frame on the particles the oracle rollout puts there, 15 % of the contact particles flagged as penetrated (push-out branch)"""
    import scenes_cloth as SC
    from oracle import cloth_oracle as CO
    sc = SC.build("taco", "float64", n_env_steps=1, N=1500)
    P = SC.oracle_params(sc)
    cloth = [sc["motion"](f * sc["cfg"].dt) for f in range(n + 1)]
    N = len(sc["state"])
    rng = np.random.default_rng(91)
    x, v, C, F = H.O.state24_split(sc["state"])
    ids0 = CO.get_contact_pair(x, torch.as_tensor(cloth[0][0]), torch.as_tensor(sc["faces"].astype(np.int64)), np.zeros(N, dtype=np.int64), sc["scale"])
    ids0 = np.asarray(ids0)
    contact = []
    for f in range(n):
        pen = ((rng.uniform(size=N) < 0.15) & (ids0 >= 0)).astype(np.int64)
        contact.append((ids0.copy(), pen))
    seeds = (rng.standard_normal((N, 3)), rng.standard_normal((N, 3)), 0.01 * rng.standard_normal((N, 3, 3)), 0.01 * rng.standard_normal((N, 3, 3)))
    eg = 1e-2 * rng.standard_normal((len(sc["vertices"]), 3))
    return sc, P, cloth, contact, seeds, eg, n


def cloth(a):
    import slab_engines as E
    sc, P, cloth_fr, contact, seeds, eg, n = cloth_scene()
    split = P.n_grid // 2                                          # the disc is centred at x = 2.5 of 5
    base = np.floor(sc["state"][:, 0] * P.inv_dx - 0.5).astype(int)
    idx = np.nonzero(base < split if a.rank == 0 else base >= split)[0]
    eng = E.ClothOracleSlabEngine(P, sc["state"][idx], cloth_fr, sc["faces"], dict(friction=sc["prim"]["friction"], softness=sc["prim"]["softness"],
                                  cloth_force_scale=sc["prim"]["cloth_force_scale"], sticky=sc["prim"]["sticky"]), [(c[idx], p[idx]) for c, p in contact])
    eng.ext_f_grad = eg
    run = SlabRunner(eng, a.rank, a.world, split, split, 2, has_contact=True)
    run.run_substeps(0, n)
    adj = eng.get_adj(n)
    for k in range(4):
        adj[k] = adj[k] + torch.as_tensor(seeds[k][idx]).reshape(adj[k].shape)
    run.run_substeps_grad(0, n)
    x, v, C, F = eng.frames[n]
    g = eng.get_adj(0)
    V = len(sc["vertices"])
    cg = np.array([[eng.cgrad.get(f, (np.zeros((V, 3)), np.zeros((V, 3))))[k] for k in range(2)] for f in range(n)])
    np.savez(pathlib.Path(a.out) / f"rank{a.rank}.npz", idx=idx, x=x.numpy(), v=v.numpy(), C=C.numpy(), F=F.numpy(), gx=g[0].numpy(), gv=g[1].numpy(),
             gC=g[2].numpy(), gF=g[3].numpy(), ext=eng.ext.numpy(), cgrad=cg, hits=int((contact[0][0][idx] >= 0).sum()))
    dist.barrier()
    dist.destroy_process_group()


GRIP_STRONG = dict(particles=1 << 18, grid=128, nsteps=6)


def grip_strong_states(s13, nframes, dt):
    return [[np.concatenate([st[:3] + f * dt * st[7:10], st[3:]]) for st in s13] for f in range(nframes)]


def grip_strong(a):
    """the strong-scaling bench scene at reduced size, HIP engine: the block cut at planes no gripper finger reaches, so that
    `contact_sides` drops the two contact exchanges - the result must still equal the single-domain run (tests/test_slabs.py)"""
    from softmac_amd import scenes
    from softmac_amd.parallel import HipSlabEngine, contact_sides
    G = GRIP_STRONG
    n = G["nsteps"]
    cfg, env_dt, state, specs, s13, slab, own = scenes.s_grip_strong(a.rank, a.world, G["particles"], G["grid"], n + 2, a.precision, 0)
    pst = grip_strong_states(s13, n + 2, cfg.dt)
    sim, prims = H.build_engine(cfg, env_dt, specs, pst)
    sim.reset(state)
    sides = contact_sides(specs, [np.stack([pst[f][i] for f in range(n + 2)]) for i in range(len(specs))], G["grid"], slab[0], slab[1], slab[2], a.rank, a.world)
    run = SlabRunner(HipSlabEngine(sim, use_torch_stream=True), a.rank, a.world, slab[0], slab[1], slab[2], has_contact=sides)
    run.run_substeps(0, n)
    hits = sim.contact_counts()[0]
    rng = np.random.default_rng(5)
    N = G["particles"]
    gx, gv = rng.standard_normal((N, 3)), rng.standard_normal((N, 3))
    sim.clear_grads()
    sim.add_grad(n, gx=gx[own], gv=gv[own])
    run.run_substeps_grad(0, n)
    g = sim.get_grad_full(0)
    np.savez(pathlib.Path(a.out) / f"rank{a.rank}.npz", idx=own, st=sim.get_state(n), gx=g[0], gv=g[1], gF=g[2], gC=g[3], sides=np.array(sides), hits=hits,
             ext=np.array([m.ext_f.to_numpy() for m in prims]))
    dist.barrier()
    dist.destroy_process_group()


def moving(a):
    """migration test: segments of `migrate_every` substeps, SlabRunner.migrate between them, migrate_grad on the way back"""
    sc = S.build_moving(a.precision, a.world)
    ids0 = S.owned_range(sc, a.rank)
    state = sc["state"][ids0]
    n, M, tol = sc["nsteps"], sc["migrate_every"], sc["drift_tol"]
    lo, hi = sc["bounds"][a.rank], sc["bounds"][a.rank + 1]
    npl = 2 + 2 * tol
    left0, right0 = max(lo - tol, 0), min(hi - tol, sc["n_grid"] - npl)
    if a.engine == "oracle":
        import slab_engines as E
        eng = E.OracleSlabEngine(H.oracle_params(sc["cfg"], sc["env_dt"]), state)
        sim = None
    else:
        from softmac_amd.parallel import HipSlabEngine
        cfg = sc["cfg"]
        cfg.n_particles = len(sc["state"])                     # capacity: any rank may end up with many more particles
        cfg.slab_flags = (2 if a.rank > 0 else 0) | (4 if a.rank < a.world - 1 else 0)
        sim, prims = H.build_engine(cfg, sc["env_dt"])
        sim.set_segment(len(state), 0)
        sim.reset(state)
        eng = HipSlabEngine(sim, use_torch_stream=True)
    if a.engine == "lib":
        return moving_lib(a, sc, sim, ids0, (lo, hi), left0, right0, npl, n, M)
    run = SlabRunner(eng, a.rank, a.world, left0, right0, npl, has_contact=False, own=(lo, hi), ids=ids0)
    f, starts = 0, []
    for seg in range(n // M):
        starts.append(f)
        run.run_substeps(f, M)
        f += M
        if seg < n // M - 1:
            f = run.migrate(f)
    ids_end = run.ids.copy()
    st_end = eng.get_state(f)
    rng = np.random.default_rng(77)
    N = len(sc["state"])
    seed_end = np.hstack([rng.standard_normal((N, 3)), rng.standard_normal((N, 3)), 0.01 * rng.standard_normal((N, 9)), 0.01 * rng.standard_normal((N, 9))])
    seed_1 = np.hstack([rng.standard_normal((N, 3)), np.zeros((N, 21))])
    if sim is not None:
        sim.clear_grads()
    eng.add_grad_rows(f, seed_end[ids_end])
    for seg in range(n // M - 1, -1, -1):
        if seg == 0:                                           # a seed inside the first segment (frame 1), in that segment's ordering
            eng.add_grad_rows(1, seed_1[ids0])
        run.run_substeps_grad(starts[seg], M)
        if seg > 0:
            run.migrate_grad()
    assert (run.ids == ids0).all()
    g0 = eng.get_grad_rows(0)
    moved = int(len(np.setdiff1d(ids_end, ids0)))
    np.savez(pathlib.Path(a.out) / f"rank{a.rank}.npz", ids0=ids0, ids_end=ids_end, st_end=st_end, g0=g0, moved=moved)
    dist.barrier()
    dist.destroy_process_group()


def moving_lib(a, sc, sim, ids0, own, left0, right0, npl, n, M):
    """the migration scene through smac_substeps_slab / smac_migrate between two real ranks (IPC link): particles, their global ids and - on the way
    back - their adjoint rows change hands as device-side byte messages"""
    from softmac_amd.parallel import LibSlabRunner, rendezvous_unique_id
    assert os.environ.get("SMAC_COMM_STUB") == "2"
    run = LibSlabRunner(sim, a.rank, a.world, left0, right0, npl, has_contact=(False, False), own=own, unique_id=rendezvous_unique_id(a.rank))
    run.set_ids(ids0)
    f, starts = 0, []
    for seg in range(n // M):
        starts.append(f)
        run.run_substeps(f, M)
        f += M
        if seg < n // M - 1:
            f = run.migrate(f, own)
    ids_end = run.ids().copy()
    st_end = sim.get_state(f)
    rng = np.random.default_rng(77)
    N = len(sc["state"])
    seed_end = np.hstack([rng.standard_normal((N, 3)), rng.standard_normal((N, 3)), 0.01 * rng.standard_normal((N, 9)), 0.01 * rng.standard_normal((N, 9))])
    seed_1 = np.hstack([rng.standard_normal((N, 3)), np.zeros((N, 21))])
    rows = lambda r: dict(gx=r[:, 0:3], gv=r[:, 3:6], gF=r[:, 6:15].reshape(-1, 3, 3), gC=r[:, 15:24].reshape(-1, 3, 3))
    sim.clear_grads()
    sim.add_grad(f, **rows(seed_end[ids_end]))
    for seg in range(n // M - 1, -1, -1):
        if seg == 0:
            sim.add_grad(1, gx=seed_1[ids0][:, 0:3])
        run.run_substeps_grad(starts[seg], M)
        if seg > 0:
            run.migrate_grad()
    assert (run.ids() == ids0).all()
    gx, gv, gF, gC = sim.get_grad_full(0)
    g0 = np.hstack([gx, gv, gF.reshape(-1, 9), gC.reshape(-1, 9)])
    moved = int(len(np.setdiff1d(ids_end, ids0)))
    np.savez(pathlib.Path(a.out) / f"rank{a.rank}.npz", ids0=ids0, ids_end=ids_end, st_end=st_end, g0=g0, moved=moved)
    run.close()
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
