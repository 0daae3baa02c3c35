#!/usr/bin/env python3
"""bench.py - MPM substeps/s (forward + backward) on the S-grip workload (BASELINE config C3:
1,048,576 particles, 128^3 grid, plasticine + gripper SDF forecast contact), synthetic seeded data.
`--workload s-pour` (C4: 4,194,304 liquid particles, 256^3, the reference's bowl) and `--workload s-mixed` (C5: 16,777,216
particles, 256^3, two material blocks + rigid palm + sheet) are SURVEY 8(d)'s other two scenes under the same window scheme.

One "step" = one substep() + one substep_grad() (its forward recompute included), run as K forward
substeps followed by K backward substeps with inputs resident in HBM.  Contract: SURVEY/driver
(`python bench.py --gpus N --steps K --warmup W`, one JSON line on rank 0).
"""
from __future__ import annotations

import argparse
import json
import math
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_HBM_GBS = 8000.0         # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
LIB_SORT_INTERVAL = 80        # the library's default re-sort interval (softmac_hip.hip: smac_config.sort_interval = 0)


def baseline_metric():
    """The headline metric, named exactly as BASELINE.json names it."""
    try:
        return json.load(open(os.path.join(ROOT, "BASELINE.json")))["metric"]
    except Exception:
        return "MPM substeps/s (fwd+bwd) at 1M particles/128\u00b3 grid, 1/2/4/8 MI355X"


_TABLES = {}


def gripper_tables(dev=0):
    """SURVEY 8(d)'s gripper: palm SDF = the reference's cached table, finger SDF = finger.obj through the library's voxeliser (once per process)"""
    if dev not in _TABLES:
        from softmac_amd import scenes
        _TABLES[dev] = scenes.gripper_tables(os.path.join(ROOT, "tests", "golden"), dev)
    return _TABLES[dev]


def bowl_table(dev=0):
    """S-pour's bowl: the reference's bowl mesh (arrays in tests/golden/pour_scene.npz, made from assets by tools/make_fixtures.py) through the library's
    voxeliser with the reference's sampling rule (mesh.py:170-233) - its cache blob is one of those missing from the reference checkout"""
    if ("bowl", dev) not in _TABLES:
        from softmac_amd.engine.primitive import voxelize
        d = np.load(os.path.join(ROOT, "tests", "golden", "pour_scene.npz"))
        _TABLES[("bowl", dev)] = voxelize.mesh_to_sdf(d["bowl_vertices"], d["bowl_faces"], device=dev)
    return _TABLES[("bowl", dev)]


def build_sim(args, rank, world, precision=None, frames=None):
    from softmac_amd import scenes
    from softmac_amd.config import CfgNode
    from softmac_amd.engine.mpm_simulator import MPMSimulator
    from softmac_amd.engine.primitive import Mesh, Primitives
    precision = precision or args.precision
    frames = frames or (args.warmup + args.steps + 2)
    mig = getattr(args, "migrate_every", 0) if world > 1 else 0
    if mig > 0:
        frames += (args.warmup + args.steps) // mig + 2           # a migration starts a new segment one frame further on
    dev = int(os.environ.get("SMAC_FORCE_DEVICE", os.environ.get("LOCAL_RANK", 0))) if world > 1 else 0
    slab = _own = None
    tables = gripper_tables(dev) if args.workload == "s-grip" else None
    if args.workload == "s-grip" and world > 1 and args.scaling == "strong":
        # the metric's own case: ONE 1M-particle / 128^3 scene cut into `world` x-slabs of one global grid
        cfg, env_dt, state, specs, s13, slab, _own = scenes.s_grip_strong(rank, world, args.particles, args.grid, frames, precision, dev, tables=tables)
    elif args.workload == "s-grip" and world > 1:
        cfg, env_dt, state, specs, s13, lh = scenes.s_grip_slab(rank, world, args.particles, args.grid, frames, precision, dev)
        slab = (lh[0], lh[1], 2)
    elif args.workload == "s-grip":
        cfg, env_dt, state, specs, s13 = scenes.s_grip(args.particles, args.grid, frames, precision, dev, seed=1 + rank, tables=tables)
    elif args.workload == "s-pour":
        cfg, env_dt, state, specs, s13 = scenes.s_pour(args.particles, args.grid, frames, precision, dev, seed=2 + rank, bowl_table=bowl_table(dev))
    else:
        cfg, env_dt, state, specs, s13 = scenes.s_elastic(args.particles, args.grid, frames, precision, dev, seed=rank)
    cfg.recompute_backward = args.recompute_backward
    cfg.sort_interval = args.sort_interval
    meshes = []
    for s in specs:
        pc = CfgNode(); pc.friction = s["friction"]; pc.enable_external_force = True; pc.urdf_path = ""
        meshes.append(Mesh(sdf=s, cfg=pc, max_timesteps=frames))
    prims = Primitives(primitives=meshes)
    n_own = len(state)
    if mig > 0 and slab is not None:
        if not (args.scaling == "strong" and args.slab_runner == "lib"):
            sys.exit("bench.py: --migrate-every needs --scaling strong and --slab-runner lib (device-side migration)")
        cfg.n_particles = n_own + n_own // 4 + 1024               # capacity: room for arrivals
    sim = MPMSimulator(cfg, prims, env_dt)
    if mig > 0 and slab is not None:
        sim.set_segment(n_own, 0)
    prims.initialize()
    for m, s in zip(meshes, specs):
        m.friction[None] = s["friction"]
    sim.primitives_contact = [bool(s["contact"]) for s in specs]
    for i, m in enumerate(meshes):
        for f in range(frames):
            st = s13[i].copy()
            st[:3] += f * cfg.dt * st[7:10]
            m.set_all_states(f, st)
    sim.reset(state)
    runner = sim
    if slab is not None:                                   # slab decomposition: halo exchange of the shared x-planes over RCCL
        from softmac_amd.parallel import HipSlabEngine, LibSlabRunner, SlabRunner, agree_contact_sides, contact_sides, rendezvous_unique_id
        traj = [np.stack([s13[i] + np.concatenate([f * cfg.dt * s13[i][7:10], np.zeros(10)]) for f in range(frames)]) for i in range(len(specs))]
        sides = contact_sides(specs, traj, args.grid, slab[0], slab[1], slab[2], rank, world)
        sides = agree_contact_sides(sides, rank, world)    # a boundary's two ranks must post the same exchanges
        fallback = None
        if args.slab_runner == "lib":
            # the loop inside the library: ncclSend / ncclRecv on its own stream, no Python between the phases; the process group (gloo, CPU)
            # only carries the RCCL id, the barriers and the max-over-ranks of the wall clock
            from softmac_amd.parallel import all_ranks_ok
            tol = (slab[2] - 2) // 2                        # range of stencil bases this rank owns, from its shared planes (scenes.s_grip_strong)
            own = (slab[0] + tol if rank > 0 else 0, slab[1] + tol if rank < world - 1 else args.grid)
            err = None
            try:
                runner = LibSlabRunner(sim, rank, world, slab[0], slab[1], slab[2], has_contact=sides, own=own, unique_id=rendezvous_unique_id(rank))
            except Exception as e:                          # noqa: BLE001
                err, runner = f"{type(e).__name__}: {e}", None
            failed = all_ranks_ok(err)                      # every rank takes the same decision
            if failed:
                # the RCCL communicator did not come up on every rank: the run is not lost - the Python loop over the control group (gloo: halo planes
                # staged through the host) carries the same exchange, slowly; the line says so
                if runner is not None:
                    try:
                        runner.close()
                    except Exception:                       # noqa: BLE001
                        pass
                fallback = "in-library RCCL loop unavailable (" + "; ".join(failed)[:400] + "): Python SlabRunner over gloo, halo planes staged through the host"
                if rank == 0:
                    print("bench.py: " + fallback, file=sys.stderr, flush=True)
                runner = SlabRunner(HipSlabEngine(sim, use_torch_stream=True), rank, world, slab[0], slab[1], slab[2], has_contact=sides)
        else:
            runner = SlabRunner(HipSlabEngine(sim, use_torch_stream=True), rank, world, slab[0], slab[1], slab[2], has_contact=sides)
        runner.contact_sides_note = sides
        runner.fallback_note = fallback
        if mig > 0:
            if fallback is not None:
                sys.exit("bench.py: --migrate-every: the in-library runner did not come up")
            runner.own_range = own
            runner.set_ids(np.asarray(_own, dtype=np.int64))
    return sim, runner, cfg


def kernel_sources_sha1():
    import hashlib
    h = hashlib.sha1()
    # everything that can change the bytes a launch moves: the kernels, the launch structure (save-in-g2p, restore-ahead, skip-empty flags, ride-along
    # parts live in softmac_hip.hip), the cloth contact kernels, and for N > 1 the migration / communication code
    for name in ("smac_kernels.hpp", "smac_math.hpp", "smac_sort.hpp", "softmac_hip.hip", "smac_cloth.hpp", "smac_cloth_kernels.hpp", "smac_migrate.hpp", "smac_comm.hpp"):
        h.update(open(os.path.join(ROOT, "softmac_amd", "csrc", name), "rb").read())
    return h.hexdigest()


def algorithmic_bytes(N, G_t, s):
    """per substep, fwd = 48 s N + 20 s G_t ; bwd = 72 s N + 40 s G_t."""
    return dict(fwd=48 * s * N + 20 * s * G_t, bwd=72 * s * N + 40 * s * G_t)


# compulsory bytes of each kernel taken alone (DESIGN.md "kernels"): scalars per particle, scalars per touched cell
KERNEL_BYTES = {
    # (g2p: its launch also carries the frame's checkpoint save since round 3 - "grid_checkpoint"'s 20 G_t, of which the save is 14 - priced apart as before)
    "p2g": (24 + 9, 4), "grid_op": (0, 4 + 6), "contact": (0, 0), "g2p": (3 + 15, 3),
    "g2p_grad": (3 + 15 + 3, 3 + 3), "contact_grad": (0, 0), "grid_op_grad": (0, 4 + 6 + 4), "p2g_grad": (24 + 9 + 3 + 24, 4),
    "clear_grid": (0, 10), "grid_checkpoint": (0, 20), "reduce_agvout": (0, 6), "forward_kinematics": (0, 0),
    "sort": (48, 0), "reorder_adjoint": (48, 0),
    # p2g.grad of substep f + g2p.grad of substep f-1 in one launch: p2g.grad's rows, + x of frame f-1 and its x.grad; both gather tiles + the slab
    "p2g_g2p_grad": (24 + 9 + 3 + 24 + 3 + 3, 4 + 3 + 3),
    # g2p of substep f + p2g of substep f+1 in one launch (round 4): x of frame f and F of frame f+1 in, x v C of frame f+1 and F of frame f+2 out; the gather tile + the slab
    "g2p_p2g": (3 + 9 + 15 + 9, 3 + 4),
}


def cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def _cpu_window(port, cfg, state, s13, nsub, N):
    """nsub forward then nsub backward substeps of the C++ port; returns seconds"""
    x, v = state[:, 0:3].copy(), state[:, 3:6].copy()
    F, C = state[:, 6:15].reshape(N, 3, 3).copy(), state[:, 15:24].reshape(N, 3, 3).copy()
    rng = np.random.default_rng(0)
    frames, psts = [(x, v, C, F)], []
    t0 = time.perf_counter()
    for f in range(nsub):
        pst = np.array([a + np.concatenate([f * cfg.dt * a[7:10], np.zeros(10)]) for a in s13]) if len(s13) else None
        psts.append(pst)
        nx, nv, nC, nF, _ = port.substep(f, *frames[-1], pst)
        frames.append((nx, nv, nC, nF))
    g = [rng.standard_normal((N, 3)), np.zeros((N, 3)), np.zeros((N, 3, 3)), np.zeros((N, 3, 3))]
    for f in range(nsub - 1, -1, -1):
        g = list(port.substep_grad(f, *frames[f], *g, pst=psts[f])[:4])
    return time.perf_counter() - t0


def cpu_baseline(args):
    """The plain C++/OpenMP f64 oracle port (oracle/mpm_cpu.cpp: the reference's decomposition - dense grid, one pass per Taichi kernel,
    atomics; NOT Taichi) timed on this box's host cores on bounded samples of the workload (SURVEY 8d "CPU baseline beside it"):
      value      the full S-grip workload (C3), `cpu_steps` forward + backward substeps, all host threads;
      c2_window  S-elastic (C2: 262,144 particles, 64^3), the survey's 64 + 64 window (shortened to fit ~12 s), all host threads;
      single_thread  the same C2 workload on ONE thread, 1 + 1 substeps."""
    from softmac_amd import scenes
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import helpers as H
    from oracle import mpm_cpu
    if args.workload == "s-pour":
        cfg, env_dt, state, specs, s13 = scenes.s_pour(args.particles, args.grid, 8, "float64", 0, seed=2, bowl_table=bowl_table(0))
        what = "liquid column over the reference's bowl, 1 primitive"
    elif args.workload == "s-elastic":
        cfg, env_dt, state, specs, s13 = scenes.s_elastic(args.particles, args.grid, 8, "float64", 0, seed=0)
        what = "elastic block, no primitives"
    else:
        cfg, env_dt, state, specs, s13 = scenes.s_grip(args.particles, args.grid, 8, "float64", 0, seed=1, tables=gripper_tables(0))
        what = "3 primitives"
    port = mpm_cpu.CpuPort(H.oracle_params(cfg, env_dt), specs)
    N = args.particles
    # threads: twice the CPUs this process is GRANTED (cgroup quota: 16 of the 256 visible on a GPU box), not every visible one - measured on that box at 1M
    # particles (tools/cpu_threads_probe.py, profiles/r05_cpu_threads.txt): 0.90 s per substep pair on 32 threads, 0.95 on 16, 1.04 on 64, 1.38 on 128
    hw = port.threads()
    threads = max(1, min(hw, 2 * mpm_cpu.cpu_share()))
    port.set_threads(threads)
    nsub = args.cpu_steps
    dt = _cpu_window(port, cfg, state, s13, nsub, N)
    out = {"value": nsub / dt, "unit": "substeps/s (fwd+bwd)", "cores": threads, "kind": "port", "cpu_model": cpu_model(),
           "sample": f"{nsub} forward + {nsub} backward substeps of the full workload ({N} particles, {args.grid}^3, {what}), "
                     f"f64 C++/OpenMP restatement of the reference's kernel decomposition (not Taichi), {dt:.1f} s on {threads} threads"}
    if args.workload != "s-grip":
        return out
    try:
        c2, env2, st2, sp2, s2 = scenes.s_elastic(1 << 18, 64, 8, "float64", 0, seed=0)
        port2 = mpm_cpu.CpuPort(H.oracle_params(c2, env2), sp2)
        t1 = _cpu_window(port2, c2, st2, s2, 1, 1 << 18)                 # one pair to size the window
        n2 = int(max(2, min(64, 12.0 / max(t1, 1e-3))))
        t2 = _cpu_window(port2, c2, st2, s2, n2, 1 << 18)
        out["c2_window"] = {"value": n2 / t2, "unit": "substeps/s (fwd+bwd)", "cores": threads,
                            "sample": f"S-elastic (C2: 262,144 particles, 64^3, no primitives), {n2} forward + {n2} backward substeps "
                                      f"(the survey's window is 64 + 64; shortened to fit the bench's time budget), {t2:.1f} s"}
        port2.set_threads(1)
        t3 = _cpu_window(port2, c2, st2, s2, 1, 1 << 18)
        port2.set_threads(threads)
        out["single_thread"] = {"value": 1.0 / t3, "unit": "substeps/s (fwd+bwd)", "cores": 1,
                                "sample": f"S-elastic (C2) on ONE thread, 1 forward + 1 backward substep, {t3:.1f} s"}
    except Exception as e:                                               # noqa: BLE001
        out["c2_error"] = f"{type(e).__name__}: {e}"[:200]
    return out


def env_loop_record(args, seed_gx):
    """The reference's OWN loop shape (taichi_env.py:93-151) on the same workload: TaichiEnv.step(action) per env step (its substeps, then the
    velocity-controlled rigid step: clear_ext_f + set_action per primitive, rigid_simulator_vel.py:20-32), a loss seed on the last frame, then
    env.backward() (step_grad per env step in reverse: get_action_grad per primitive + the substeps' adjoints).  Host orchestration included."""
    from softmac_amd import scenes
    from softmac_amd.config import CfgNode, get_cfg_defaults
    from softmac_amd.engine.primitive import Mesh, Primitives
    from softmac_amd.engine.taichi_env import TaichiEnv
    import torch
    scfg, env_dt, state, specs, s13 = scenes.s_grip(args.particles, args.grid, 8, args.precision, 0, seed=1, tables=gripper_tables(0))
    n_env = max(5, args.steps // 10)         # at least 5 env steps: an episode starts from the caller's particle order (reset), whose first binning
    w_env = max(1, (args.warmup + 9) // 10)  # costs 3 x a re-sort of a binned frame - the reference's episodes have 400 env steps (demo_grip.py:189-191)
    cfg = get_cfg_defaults()
    cfg.control_mode = "rigid"
    cfg.rigid_velocity_control = True
    cfg.env_dt = env_dt
    S = cfg.SIMULATOR
    for k in ("dt", "E", "nu", "ptype", "material_model", "gravity", "ground_friction", "collision_type", "yield_stress", "precision"):
        setattr(S, k, getattr(scfg, k))
    S.n_grid = args.grid
    S.max_steps = (n_env + w_env) * 10 + 2
    S.sort_interval = args.sort_interval
    cfg.SHAPES = [{"shape": "predefined", "state": state}]
    n = len(specs)
    pose = np.zeros((n, 6)); vel = np.zeros((n, 6))
    for i, st in enumerate(s13):
        pose[i, 3:] = st[:3]                     # identity rotation (exp map 0), position
        vel[i, :3], vel[i, 3:] = st[10:13], st[7:10]
    cfg.RIGID.init_state = tuple(np.concatenate([pose.reshape(-1), vel.reshape(-1)]))
    meshes = []
    for sp in specs:
        pc = CfgNode(); pc.friction = sp["friction"]; pc.enable_external_force = True; pc.urdf_path = ""
        meshes.append(Mesh(sdf=sp, cfg=pc, max_timesteps=S.max_steps, rigid_velocity_control=True))
    env = TaichiEnv(cfg, primitives=Primitives(primitives=meshes))
    for m, sp in zip(meshes, specs):
        m.friction[None] = sp["friction"]
    env.simulator.primitives_contact = [bool(sp["contact"]) for sp in specs]
    action = torch.tensor(vel.reshape(-1))       # constant closing speed: (w, v) per primitive
    sim = env.simulator

    def episode(k_env):
        """returns (action gradient, seconds spent in the env steps + in backward()); reset and the loss seed's upload are not in it
        (96 + 25 MB of host arrays at this size: the reference pays them once per epoch of 400 env steps, demo_grip.py:135-165)"""
        env.reset()
        sim.clear_grads()
        sim.sync()
        t0 = time.perf_counter()
        for _ in range(k_env):
            env.step(action)
        sim.sync()
        t1 = time.perf_counter()
        sim.add_grad(sim.cur, gx=seed_gx)
        sim.sync()
        t2 = time.perf_counter()
        g = env.backward()
        sim.sync()
        return g, (t1 - t0) + (time.perf_counter() - t2)

    episode(w_env)
    runs = [episode(n_env) for _ in range(max(args.repeats, 1))]     # the same statistic as `value`: the median of `repeats` identical episodes
    walls = sorted(w for _, w in runs)
    g, wall = runs[0][0], walls[len(walls) // 2]
    sim._h.close()
    return {"value": n_env * 10 / wall, "unit": "substeps/s", "ms_per_step": 1e3 * wall / (n_env * 10), "env_steps": n_env, "substeps_per_env_step": 10,
            "repeats": len(runs), "ms_per_step_all": [round(1e3 * w / (n_env * 10), 5) for _, w in runs],
            "action_grad_norm": float(np.linalg.norm(g.numpy())),
            "note": "TaichiEnv.step x env_steps, then TaichiEnv.backward(), with velocity-controlled primitives (the reference's loop shape, "
                    "taichi_env.py:93-151): host orchestration per env step included, reset and the loss seed's upload not; same particles / grid / "
                    "primitives as `value`"}


def workload_text(args, N, cfg):
    head = f"{args.workload}: {N} particles, {args.grid}^3 grid, dt {cfg.dt:g}, "
    if args.workload == "s-grip":
        return head + ("plastic fixed-corotated, 3 gripper SDF primitives (2 in forecast contact; palm = the reference's cached SDF table, fingers = "
                       "finger.obj through smac_mesh_to_sdf), fwd+bwd")
    if args.workload == "s-pour":
        return head + ("liquid (ptype 2, E 22: demo_pour_config.py:9,20,28) falling into the reference's bowl (bowl.obj through smac_mesh_to_sdf, friction 1.0) "
                       "in forecast contact, fwd+bwd - SURVEY 8(d) S-pour = BASELINE C4's scene on ONE GPU")
    return head + "elastic fixed-corotated, no primitives, fwd+bwd"


def metric_name(args):
    """BASELINE.json's metric for the configuration it is quoted on (s-grip at its default size); any other workload says what it is"""
    if args.workload == "s-grip" and args.particles == 1 << 20 and args.grid == 128:
        return baseline_metric()
    return f"MPM substeps/s (fwd+bwd), {args.workload} at {args.particles} particles / {args.grid}^3 (not the headline configuration)"


def transport_note(run, sim, dist, world):
    """What carried the shared planes in THIS run, read back from the objects that did it - never a constant."""
    from softmac_amd.parallel import LibSlabRunner
    if isinstance(run, LibSlabRunner):
        kind, ranks = int(sim.get_param("comm_transport")), int(sim.get_param("comm_world"))
        name = {1: "RCCL (ncclSend / ncclRecv inside libsoftmac_hip, its own communicator)",
                2: "IPC link between processes sharing GPUs (SMAC_COMM_STUB=2: a test transport, host-synchronous - NOT a scaling number)",
                3: "device-copy stub (SMAC_COMM_STUB=1: no bytes leave the GPU - NOT a scaling number)"}.get(kind, "none: the communicator is gone")
        return f"{name}; {ranks} ranks in the library's communicator (launched: {world}); control plane torch.distributed/{dist.get_backend()}"
    be = dist.get_backend()
    how = "RCCL through torch.distributed" if be == "nccl" else f"torch.distributed/{be}: halo planes staged through the HOST - NOT a scaling number"
    return f"{how}; {dist.get_world_size()} ranks in the process group"


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=None, help="ranks (one per GPU); spawned here when not already under torchrun")
    ap.add_argument("--steps", type=int, default=None, help="substep pairs per window (default 32; s-mixed: 5 - a frame of 16M particles is 3.2 GB)")
    ap.add_argument("--warmup", type=int, default=4)
    ap.add_argument("--repeats", type=int, default=0, help="number of K-step windows (0: window_plan - as many as give the re-sorts their steady-state share, at least 8)")
    ap.add_argument("--workload", default="s-grip", choices=["s-grip", "s-elastic", "s-pour", "s-mixed"],
                    help="SURVEY 8(d)'s scenes: s-grip = BASELINE C3 (the metric's configuration), s-elastic = C2, s-pour = C4 (4M / 256^3 on one GPU), s-mixed = C5 (16M / 256^3)")
    ap.add_argument("--scaling", default="strong", choices=["strong", "weak"],
                    help="N > 1: strong = the metric's one 1M-particle scene cut into N slabs; weak = N bars of 1M particles each")
    ap.add_argument("--particles", type=int, default=None, help="default: the workload's size (s-grip 1,048,576; s-elastic 262,144; s-pour 4,194,304; s-mixed 16,777,216)")
    ap.add_argument("--grid", type=int, default=None, help="default: the workload's grid (128 / 64 / 256 / 256)")
    ap.add_argument("--precision", default="float32", choices=["float32", "float64"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-f64", action="store_true", help="skip the float64 sub-record (the mode that equals the reference's dtype)")
    ap.add_argument("--no-cloth", action="store_true", help="skip the soft <-> cloth sub-record (tools/bench_cloth.py in a child process)")
    ap.add_argument("--no-env-loop", action="store_true", help="skip the env_loop sub-record (the reference's TaichiEnv.step / backward loop on the same workload)")
    ap.add_argument("--cpu-steps", type=int, default=None, help="substep pairs of the CPU baseline's sample (default 8; s-pour 2)")
    ap.add_argument("--recompute-backward", action="store_true", help="substep_grad recomputes the forward grid (reference style)")
    ap.add_argument("--sort-interval", type=int, default=0, help="0: the library's default")
    ap.add_argument("--slab-runner", default=os.environ.get("SMAC_SLAB_RUNNER", "lib"), choices=["lib", "python"],
                    help="N > 1: lib (default since round 5) = smac_substeps_slab, the loop inside the library with its own RCCL communicator and the fused particle "
                         "launches - its distinct-peer code has run between 2 and 3 ranks over the IPC test transport; if the communicator does not come up on "
                         "every rank the run falls back, on every rank together, to python = parallel.SlabRunner on torch.distributed (plain kernels, slower)")
    ap.add_argument("--first-window-limit", type=float, default=float(os.environ.get("SMAC_FIRST_WINDOW_LIMIT", 240.0)),
                    help="N > 1: seconds the warm-up window (the first place an N-rank run can hang: first exchange, first migration) may take before the rank "
                         "publishes a failure, aborts its communicator and exits with status 3 - a fresh non-zero exit, nothing is re-executed")
    ap.add_argument("--migrate-every", type=int, default=0,
                    help="N > 1, strong scaling, --slab-runner lib: hand particles over between the slabs (smac_migrate, device side) every k substeps INSIDE the "
                         "timed window; the handles get 25 %% spare capacity and one extra frame per migration")
    ap.add_argument("--launch-check", action="store_true",
                    help="spawn the ranks, rendezvous (gloo), report - no simulator, no GPU call (CPU test of the launcher)")
    args = ap.parse_args(argv)
    size = {"s-grip": (1 << 20, 128), "s-elastic": (1 << 18, 64), "s-pour": (1 << 22, 256), "s-mixed": (1 << 24, 256)}[args.workload]
    if args.particles is None:
        args.particles = size[0]
    if args.grid is None:
        args.grid = size[1]
    if args.steps is None:
        args.steps = 5 if args.workload == "s-mixed" else 32
    if args.cpu_steps is None:
        args.cpu_steps = 2 if args.workload == "s-pour" else 8
    return args


def launch_children(args, argv):
    """`python bench.py --gpus N` outside torchrun: start N fresh ranks (one per GPU) BEFORE anything here touches the GPU -
    this parent never imports torch or the HIP library - and pass their exit code on."""
    import socket
    import subprocess
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")          # dmabuf IPC (RCCL / cross-process device memory on this host driver)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + list(argv)
    return subprocess.call(cmd, env=env)


def launch_check(args, world, rank):
    import torch.distributed as dist
    dist.init_process_group(os.environ.get("SMAC_DIST_BACKEND", "gloo"))
    mine = {"rank": rank, "local_rank": int(os.environ.get("LOCAL_RANK", -1)), "pid": os.getpid()}
    ranks = [None] * world
    dist.all_gather_object(ranks, mine)
    if rank == 0:
        print(json.dumps({"launch_check": True, "n_gpus": world, "requested": args.gpus, "backend": dist.get_backend(),
                          "ranks": sorted(ranks, key=lambda r: r["rank"])}), flush=True)
    dist.barrier()
    dist.destroy_process_group()


def window_plan(K, interval):
    """Windows of EXACTLY K substep pairs that advance through the episode, as many of them as it takes for the re-sorts inside them to have their
    steady-state share: R * K is a multiple of the re-sort interval (so K = 20 at an interval of 80 has a re-sort in every fourth window),
    and R >= 8 so that the GPU's clocks have settled for most of them (the first two windows after a cold start run 5-8 % slower:
    profiles/r04_z_first_pass.txt).  Every window is timed and counted - nothing is dropped, `value` comes from their sum."""
    block = math.lcm(K, max(interval, 1)) // K
    if block * K > 480:                                       # (an interval that shares no factor with K: fall back to whole windows, share approximate)
        block = 1
    R = block * -(-8 // block)
    while R > block and R * K > 480:                          # every frame of the episode is resident (212 MB per frame at 1M particles): a long K gets fewer windows
        R -= block
    return max(R, 1)


def timed_windows(args, sim, run, reducer, seed_gx, barrier, dist, windows=None):
    """W warm-up substep pairs, then R windows of EXACTLY K forward + K backward substeps; window r covers frames [W + r K, W + (r + 1) K) - the
    episode goes on, it is not the same K frames again and again (frames that were simulated before keep their binning, an adjoint seed stored under
    it and warm pages: profiles/r04_z_first_pass.txt).  The loss seed of a window is added INSIDE its timed region, after the forward pass, from a
    buffer that is resident in HBM (smac_add_grad_device) - the reference's loss kernels write x.grad on the device after the forward pass too
    (losses/loss_pour.py:130-140).  Returns (per-window wall seconds after MAX over ranks, per-window device ms)."""
    import torch
    K, W = args.steps, args.warmup
    R = windows or window_plan(K, args.sort_interval)
    env = max(sim.substeps, 1)
    seed_dev = torch.from_numpy(np.ascontiguousarray(np.vstack([seed_gx, seed_gx[: max(len(seed_gx) // 4, 1) + 1024]]), dtype=np.float64)).to(f"cuda:{sim.device}")
    torch.cuda.synchronize()

    M = getattr(args, "migrate_every", 0) if reducer is not None else 0
    segs = []                                                # (first frame, substeps) of the segments of the last forward pass; a migration between two of them

    def forward(f0, n):
        if reducer is None:
            run.run_substeps(f0, n)
            return f0 + n
        if M > 0:                                            # migrating window: k substeps, hand-over (frame index + 1), k substeps, ...
            del segs[:]
            f, done = f0, 0
            while done < n:
                m = min(M, n - done)
                run.run_substeps(f, m)
                segs.append((f, m))
                f, done = f + m, done + m
                reducer.allreduce_ext_f(clear=True)
                if done < n:
                    f = run.migrate(f, run.own_range)
            return f
        f = f0
        while f < f0 + n:                                   # per env step: substeps, then the wrench sums of all slabs
            m = min(env - f % env, f0 + n - f)
            run.run_substeps(f, m)
            f += m
            reducer.allreduce_ext_f(clear=True)
        return f

    def backward(f0, n):
        if M > 0 and reducer is not None:
            for i in range(len(segs) - 1, -1, -1):
                run.run_substeps_grad(*segs[i])
                reducer.allreduce_state_grad(segs[i][0], segs[i][0] + segs[i][1])
                if i > 0:
                    run.migrate_grad()
            return
        run.run_substeps_grad(f0, n)
        if reducer is not None:
            reducer.allreduce_state_grad(f0, f0 + n)

    def seed(f_end):                                         # x.grad[f_end] += seed, device to device on the handle's stream
        sim.add_grad_device(f_end, gx=seed_dev[: sim.n_particles])

    def pair(f0, n):                                         # one window: forward, the loss seed, backward
        f_end = forward(f0, n)
        seed(f_end)
        backward(f0, n)
        return f_end

    sim.clear_grads()
    watch = getattr(args, "failure_watch", None)
    if watch is not None:
        watch.arm(args.first_window_limit, f"the warm-up window ({W} substep pairs: the first exchanges of the {getattr(args, 'gpus', None)}-rank run)")
    pair(0, W)
    sim.sync()
    if watch is not None:
        watch.disarm()
        watch.check()
    walls, devs = [], []
    f0 = W
    starts = []
    for _ in range(R):
        sim.clear_grads()                                    # (ti.ad.clear_all_gradients() between episodes; outside the timed region as the reference's is outside its tape)
        for m in sim.primitives:
            m.clear_ext_f()
        barrier()
        t0 = time.perf_counter()
        sim.timer_start()
        f_end = pair(f0, K)
        dev_ms = sim.timer_stop()
        barrier()
        wall = time.perf_counter() - t0
        if dist is not None:
            t = torch.tensor([wall], device="cuda" if dist.get_backend() == "nccl" else "cpu")
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            wall = float(t.item())
        walls.append(wall)
        devs.append(dev_ms)
        starts.append(f0)
        f0 = f_end
    timed_windows.migrations = max(len(segs) - 1, 0)
    timed_windows.moved = int(getattr(run, "moved", 0))
    timed_windows.pair = pair
    timed_windows.starts = starts
    return walls, devs, forward, backward


def run_mixed(args):
    """S-mixed (BASELINE config C5): 16,777,216 particles / 256^3, two material blocks, the reference's cached gripper palm pressed into the
    cylinder from above, the sticky sheet under it - ONE handle with both kinds of primitive.  The loop is the reference's soft_cloth env loop
    (soft_cloth/engine/taichi_env.py:86-106: substep, contact-face search, penetration tracing per substep) forward, then substep_grad in reverse; windows
    of exactly K substep pairs advancing through one episode, every window counted, the loss seed added from HBM inside the window - the scheme of
    `timed_windows`.  A frame of this scene is 3.2 GB (state + adjoint), hence the short default K = 5."""
    import torch
    torch.cuda.init()
    from softmac_amd import scenes
    from softmac_amd.config import CfgNode
    from softmac_amd.engine.primitive import Mesh, Primitives
    from softmac_amd.soft_cloth.engine.mpm_simulator import MPMSimulator
    from softmac_amd.soft_cloth.engine.primitive import Primitive_Cloth
    K, W = args.steps, args.warmup
    R = args.repeats if args.repeats > 0 else 8
    frames = W + R * K + 2
    N = args.particles
    palm = np.load(os.path.join(ROOT, "tests", "golden", "palm_sdf.npz"))
    palm = dict(sdf=palm["sdf"], normal=palm["normal"], lower=palm["lower"], upper=palm["upper"], dx=float(palm["dx"]), res=palm["res"])
    rings = max(12, 48 * args.grid // 256)
    cfg, env_dt, state, V, F, sheet_cfg, rigid, s13, mat_id, mat2 = scenes.s_mixed(N, args.grid, max_steps=frames, precision=args.precision, rings=rings, palm=palm)
    cfg.sort_interval = args.sort_interval
    sheet = Primitive_Cloth(CfgNode(sheet_cfg), max_timesteps=cfg.max_steps, mpm_scale=1.0, vertices=V, faces=F)
    rc = CfgNode()
    rc.friction, rc.enable_external_force, rc.urdf_path = rigid["friction"], True, ""
    mesh = Mesh(sdf=rigid, cfg=rc, max_timesteps=cfg.max_steps)
    sim = MPMSimulator(cfg, sheet, env_dt, 1.0, rigid_primitives=Primitives(primitives=[mesh]))
    sheet.initialize()
    mesh.softness[None] = 666.0
    mesh.friction[None] = rigid["friction"]
    sim.primitives_contact = [True]
    Vv = np.zeros_like(V)
    sheet.set_all_states(0, V, Vv, f_end=cfg.max_steps)
    for f in range(cfg.max_steps):
        st = s13.copy()
        st[:3] += f * cfg.dt * s13[7:10]
        mesh.set_all_states(f, st)
    sim.set_materials(mat_id, mat2["E"], mat2["nu"], mat2["yield_stress"])
    sim.reset(state)
    rng = np.random.default_rng(7)
    seed_dev = torch.from_numpy(rng.standard_normal((N, 3))).to(f"cuda:{sim.device}")
    del state
    sim.get_contact_pair(0)

    def forward(f0, n):
        for s_ in range(f0, f0 + n):
            sim.substep(s_)
            sim.get_contact_pair(s_ + 1)
            sim.trace_penetration_after_mpm(s_ + 1)

    def backward(f0, n):
        for s_ in range(f0 + n - 1, f0 - 1, -1):
            sim.substep_grad(s_)

    def pair(f0, n):
        forward(f0, n)
        sim.add_grad_device(f0 + n, gx=seed_dev)
        backward(f0, n)

    sim.clear_grads()
    pair(0, W)
    walls, devs, starts = [], [], []
    f0 = W
    for _ in range(R):
        sim.clear_grads()
        sim.sync()
        t0 = time.perf_counter()
        sim.timer_start()
        pair(f0, K)
        dev_ms = sim.timer_stop()
        sim.sync()
        walls.append(time.perf_counter() - t0)
        devs.append(dev_ms)
        starts.append(f0)
        f0 += K
    wall, dev_ms = sum(walls) / len(walls), sum(devs) / len(devs)
    sim.profile(True)
    for f0 in starts[-2:]:                                  # per-kernel HIP-event profile over the last two windows once more (out of the timed region)
        sim.clear_grads()
        pair(f0, K)
    prof = sim.profile_report()
    sim.profile(False)
    prof_substeps = 2 * K
    G_t = sim.count_active_cells(W)
    n_hits = sim.contact_counts()[0]
    ids, pen = sim.get_contact(W)
    kern = {k: v for k, v in prof.items() if v[1] > 0}
    dom = max((k for k in kern if k not in ("sort", "reorder_adjoint")), key=lambda k: kern[k][0])
    avg_ms = kern[dom][0] / kern[dom][1]
    sbytes = 4 if args.precision == "float32" else 8
    pp, pc = KERNEL_BYTES.get(dom, (0, 0))
    alg = (pp * N + pc * G_t) * sbytes
    achieved = alg / (avg_ms * 1e-3) / 1e9
    ab = algorithmic_bytes(N, G_t, sbytes)
    sub_gbs = (ab["fwd"] + ab["bwd"]) * (K / (dev_ms * 1e-3)) / 1e9
    out = {"metric": metric_name(args), "value": K / wall, "unit": "substeps/s", "n_gpus": 1, "steps": K, "warmup": W, "ms_per_step": 1e3 * wall / K,
           "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f32" if args.precision == "float32" else "f64", "data": "synthetic",
           "config": {"workload": (f"s-mixed: {N} particles, {args.grid}^3 grid, dt {cfg.dt:g}, von-Mises plasticine in TWO material blocks (E {cfg.E:g} / {mat2['E']:g}, yield stress "
                                   f"{cfg.yield_stress:g} / {mat2['yield_stress']:g}), one rigid SDF primitive (the reference's cached gripper palm) AND a sticky triangle-mesh sheet of "
                                   f"{len(F)} faces in forecast contact; per substep the contact-face search and the penetration tracing of the reference's soft_cloth loop; fwd+bwd - "
                                   "SURVEY 8(d) S-mixed = BASELINE C5 on ONE GPU"),
                      "n_grid": args.grid, "touched_cells": G_t, "contact_particles": n_hits, "particles_holding_a_face": int((ids >= 0).sum()), "penetrated": int((pen == 1).sum()),
                      "resort_interval": args.sort_interval, "resorts_in_windows": int(kern.get("sort", (0, 0))[1]),
                      "windows": (f"{R} windows of exactly {K} substep pairs each, advancing through one episode (frames {W} .. {W + R * K}); the loss seed is added from HBM "
                                  "inside each window; value = K / mean window time, no window dropped"),
                      "parallelism": "1 gpu"},
           "repeats": R, "aggregate": "mean", "ms_per_step_all": [round(1e3 * w / K, 5) for w in walls], "spread": (max(walls) - min(walls)) / wall,
           "device_ms_per_step": dev_ms / K, "drift_repairs": int(sim.get_param("drift_repairs")),
           "roofline": {"bound": "hbm", "kernel": dom, "achieved": achieved, "peak": PEAK_HBM_GBS, "unit": "GB/s", "frac": achieved / PEAK_HBM_GBS, "traffic": None,
                        "traffic_note": "no PMC pass was taken for this workload", "avg_launch_ms": avg_ms, "algorithmic_bytes_per_launch": alg},
           "roofline_substep": {"algorithmic_bytes_fwd_bwd": ab["fwd"] + ab["bwd"], "achieved": sub_gbs, "peak": PEAK_HBM_GBS, "unit": "GB/s", "frac": sub_gbs / PEAK_HBM_GBS},
           "kernels_ms": {k: round(v[0] / v[1], 4) for k, v in kern.items()},
           "kernels_ms_per_step": {k: round(v[0] / prof_substeps, 4) for k, v in kern.items()}}
    sim._h.close()
    if not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline_mixed(args, palm)
    print(json.dumps(out), flush=True)


def cpu_baseline_mixed(args, palm):
    """CPU baseline of S-mixed: the composed torch-f64 oracle (oracle/mixed_oracle.py: NOT Taichi, NOT the reference - the reference has no simulator with
    both primitive kinds) on a BOUNDED SAMPLE: the same scene generator at 1/64 of the particles on a grid of a quarter the resolution per axis (same 8
    particles per cell, same shapes), one substep forward + its adjoint; `value` is that rate scaled by (sample particles / full particles) - linear in
    N, which is what the dense per-particle torch ops cost - and says so."""
    import torch
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import helpers as H                                     # noqa: F401  (puts the repo root on sys.path for `oracle`)
    import dataclasses
    from oracle import cloth_oracle as CO, mixed_oracle as MO, softmac_oracle as O
    from softmac_amd import scenes
    n_s, g_s = min(max(args.particles // 64, 1 << 14), 1 << 17), max(args.grid // 4, 32)      # (131,072 particles: ~20-50 s of torch work)
    cfg, env_dt, state, V, F, sheet_cfg, rigid, s13, mat_id, mat2 = scenes.s_mixed(n_s, g_s, max_steps=4, precision="float64", rings=12, palm=palm)
    P = CO.ClothSimParams(n_grid=g_s, dt=cfg.dt, E=cfg.E, nu=cfg.nu, ptype=0, material_model=0, gravity=tuple(cfg.gravity), yield_stress=cfg.yield_stress,
                          collision_type=2, substeps=int(round(env_dt / cfg.dt)), scale=1.0)
    P2 = dataclasses.replace(P, E=mat2["E"], nu=mat2["nu"], yield_stress=mat2["yield_stress"])
    x, v, C, Fm = O.state24_split(state)
    faces = torch.as_tensor(F.astype(np.int64))
    sheet = CO.ClothPrim(position=torch.as_tensor(V), velocity=torch.zeros(V.shape, dtype=O.DT), faces=faces, friction=sheet_cfg["friction"], softness=sheet_cfg["softness"],
                         cloth_force_scale=sheet_cfg["cloth_force_scale"], sticky=sheet_cfg["sticky"])
    prim = O.make_prim(s13[:3], s13[3:7], s13[7:10], s13[10:13], rigid["sdf"], rigid["normal"], rigid["lower"], rigid["upper"], rigid["dx"], rigid["friction"], rigid["softness"], True)
    from oracle import mpm_cpu
    torch.set_num_threads(max(1, min(torch.get_num_threads(), mpm_cpu.cpu_share())))     # (the granted CPUs: 128 torch threads on a 16-CPU quota run this sample 10 x slower)
    threads = torch.get_num_threads()
    t0 = time.perf_counter()
    ids = np.asarray(CO.get_contact_pair(x, sheet.position, faces, np.zeros(n_s, dtype=np.int64), 1.0))
    pen = np.zeros(n_s, dtype=np.int64)
    nx, nv, nC, nF, er, ec = MO.substep(x, v, C, Fm, P, [prim], sheet, ids, pen, 0, P2, mat_id)
    g = MO.substep_grad(x, v, C, Fm, P, [prim], sheet, ids, pen, 0, torch.ones_like(nx), torch.zeros_like(nv), torch.zeros_like(nC), torch.zeros_like(nF), P2=P2, mat_id=mat_id)
    dt = time.perf_counter() - t0
    assert torch.isfinite(g["gx"]).all()
    return {"value": (1.0 / dt) * n_s / args.particles, "unit": "substeps/s (fwd+bwd)", "cores": threads, "kind": "port", "cpu_model": cpu_model(),
            "sample": (f"1 forward + 1 backward substep (contact-face search included) of the SAME scene generator at {n_s} particles / {g_s}^3 (1/{args.particles // n_s} of the "
                       f"particles, 8 per cell as at full size), composed torch-f64 oracle (oracle/mixed_oracle.py; not Taichi), {dt:.1f} s on {threads} torch threads; "
                       f"value = that rate x {n_s}/{args.particles} (linear in N)"),
            "sample_substeps_per_s": 1.0 / dt}


def main():
    argv = sys.argv[1:]
    args = parse_args(argv)
    under_launcher = "WORLD_SIZE" in os.environ
    if not under_launcher and args.gpus and args.gpus > 1:
        sys.exit(launch_children(args, argv))
    world = int(os.environ.get("WORLD_SIZE", 1))
    rank = int(os.environ.get("RANK", 0))
    if args.gpus is not None and args.gpus != world:
        sys.exit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}: refusing to report a {world}-rank run as {args.gpus} GPUs")
    if args.launch_check:
        return launch_check(args, world, rank)
    if args.sort_interval <= 0:
        args.sort_interval = LIB_SORT_INTERVAL              # the library's default (smac_config.sort_interval = 0): what a caller of the engine gets
    if args.workload == "s-mixed":
        if world > 1:
            sys.exit("bench.py: --workload s-mixed runs on one GPU (the slab runners do not carry the sheet's contact search)")
        return run_mixed(args)
    dist = None
    import torch
    torch.cuda.init()           # torch's HIP runtime comes up BEFORE libsoftmac_hip loads the system's (the other order leaves torch without a device)
    if world > 1:
        import torch.distributed as dist
        torch.cuda.set_device(int(os.environ.get("SMAC_FORCE_DEVICE", os.environ.get("LOCAL_RANK", 0))))
        # in-library runner: the data path's RCCL communicator lives in libsoftmac_hip; torch.distributed is the control plane only (gloo)
        dist.init_process_group(os.environ.get("SMAC_DIST_BACKEND", "gloo" if args.slab_runner == "lib" else "nccl"))      # "nccl" is RCCL on ROCm

    R = args.repeats if args.repeats > 0 else window_plan(args.steps, args.sort_interval)
    sim, run, cfg = build_sim(args, rank, world, frames=args.warmup + R * args.steps + 2)
    N_local, K, W = int(cfg.n_particles), args.steps, args.warmup
    N = args.particles
    sbytes = 4 if args.precision == "float32" else 8
    rng = np.random.default_rng(7 + rank)
    seed_gx = rng.standard_normal((N_local, 3))
    reducer = None
    watch = None
    if world > 1:
        from softmac_amd.parallel import FailureWatch, LibSlabRunner, PrimitiveReducer
        reducer = run if isinstance(run, LibSlabRunner) else PrimitiveReducer(sim)
        # out-of-band failure channel: a rank that fails inside the loop publishes; its neighbours' watch threads abort their communicators so
        # that their pending receives return, and every rank exits non-zero instead of waiting for the launcher's timeout
        watch = FailureWatch(rank, world, runner=run if hasattr(run, "abort") else None)
        args.failure_watch = watch

    def barrier():
        sim.sync()
        if dist is not None:
            import torch
            dist.barrier()
            torch.cuda.synchronize()

    try:
        walls, devs, forward, backward = timed_windows(args, sim, run, reducer, seed_gx, barrier, dist, windows=R)
    except Exception as e:                                  # noqa: BLE001
        if world == 1:
            raise
        # One rank failed inside the collective loop (drift, the slab-range guard, a HIP / RCCL error): its neighbours sit in an exchange nobody will
        # answer.  The library has aborted this rank's communicator (smac_substeps_slab: slab_guard); nothing is retried in-process - this rank ends
        # NOW with a non-zero status and the launcher (torch.distributed.run) stops the others.  os._exit: no destructor may wait on the dead exchange.
        print(f"bench.py: rank {rank} failed inside the collective run: {type(e).__name__}: {e}", file=sys.stderr, flush=True)
        try:
            if watch is not None and not watch.failure:     # (a rank that was stopped BY the watch does not publish: the first failure is the one to read)
                watch.publish(f"{type(e).__name__}: {e}")
            if hasattr(run, "abort"):
                run.abort()
        except Exception:                                   # noqa: BLE001
            pass
        sys.stdout.flush()
        os._exit(1)
    # every window counts: K / (mean window time) - the windows without and with a re-sort in their steady-state proportion
    wall, dev_ms = sum(walls) / len(walls), sum(devs) / len(devs)
    strong = world > 1 and args.scaling == "strong" and args.workload == "s-grip"
    value = (1 if (strong or world == 1) else world) * K / wall

    # per-kernel HIP-event profile over the same R windows once more (out of the timed region; the frames keep their binning, the launches are the same)
    M_ = getattr(args, "migrate_every", 0) if reducer is not None else 0
    again = timed_windows.starts if M_ == 0 else timed_windows.starts[-1:]      # (a migrated window cannot be replayed from its first frame's old segment)
    sim.profile(True)
    for f0 in again:
        sim.clear_grads()
        timed_windows.pair(f0, K)
    prof = sim.profile_report()
    sim.profile(False)
    prof_substeps = K * len(again)
    # SURVEY 8(d) asks for the two directions separately as well: the same windows with a sync between the two directions
    split = None
    if world == 1:
        import torch
        sdev = torch.from_numpy(np.ascontiguousarray(seed_gx, dtype=np.float64)).to(f"cuda:{sim.device}")
        torch.cuda.synchronize()
        tf = tb = 0.0
        for f0 in again:
            sim.clear_grads()
            barrier()
            t0 = time.perf_counter()
            forward(f0, K)
            barrier()
            t1 = time.perf_counter()
            sim.add_grad_device(f0 + K, gx=sdev)
            backward(f0, K)
            barrier()
            t2 = time.perf_counter()
            tf, tb = tf + (t1 - t0) / len(again), tb + (t2 - t1) / len(again)
        split = (tf, tb)
    G_t = sim.count_active_cells(W)
    n_hits, n_hit_chunks = sim.contact_counts()
    counts = [N_local, G_t, n_hits]
    if dist is not None:
        allc = [None] * world
        dist.all_gather_object(allc, counts)
    else:
        allc = [counts]

    if rank == 0:
        kern = {k: v for k, v in prof.items() if v[1] > 0}
        per_step = {k: v[0] / prof_substeps for k, v in kern.items()}            # ms per substep pair, amortised (sort: 1 per interval)
        dom = max((k for k in kern if k not in ("sort", "reorder_adjoint")), key=lambda k: kern[k][0])
        avg_ms = kern[dom][0] / kern[dom][1]
        pp, pc = KERNEL_BYTES[dom]
        alg = (pp * N_local + pc * G_t) * sbytes
        achieved = alg / (avg_ms * 1e-3) / 1e9
        Gsum = sum(c[1] for c in allc)
        ab = algorithmic_bytes(sum(c[0] for c in allc), Gsum, sbytes)
        steps_per_s_dev = K / (dev_ms * 1e-3)
        sub_gbs = (ab["fwd"] + ab["bwd"]) * steps_per_s_dev / 1e9 / world            # per GPU (all ranks' bytes / ranks)
        # HBM bytes per launch from the PMC passes (tools/pmc_traffic.py; separate --pmc runs as the guide prescribes): a committed
        # measurement, so it is only quoted while the kernels it was taken on are the kernels that just ran (hash of the kernel sources)
        traffic, traffic_note = None, None
        tpath = os.path.join(ROOT, "profiles", "traffic_latest.json")
        if os.path.exists(tpath):
            try:
                tj = json.load(open(tpath))
                if tj.get("kernel_sources_sha1") == kernel_sources_sha1():
                    traffic = tj.get(dom)
                    traffic_note = tj.get("measured", "profiles/traffic_latest.json")
                else:
                    traffic_note = "stale: the kernel sources changed since profiles/traffic_latest.json was measured - dropped"
            except Exception:
                traffic = None
        if world == 1:
            par = "1 gpu"
        elif strong:
            par = (f"strong scaling: the one {N}-particle scene cut into {world} x-slabs of one global {args.grid}^3 grid, balanced by particle count "
                   f"({[c[0] for c in allc]} particles per rank); per substep neighbour-only send/recv of the 4 shared grid planes (over what: `transport`) "
                   f"(fwd: m,p + contact corrections; bwd: grid_v_out.grad + grid_v_mixed.grad; the two contact exchanges only across "
                   f"boundaries a gripper finger can reach - rank 0: {getattr(run, 'contact_sides_note', None)}); ext_f all-reduced per env step, "
                   f"primitive adjoints per window")
        else:
            par = (f"weak scaling: {world} x-slabs of one bar, {N} particles each; value counts every slab's substep; per substep "
                   f"neighbour-only send/recv of 2 shared grid planes (over what: `transport`)")
        out = {
            "metric": metric_name(args),
            "value": value, "unit": "substeps/s", "n_gpus": world, "steps": K, "warmup": W,
            "ms_per_step": 1e3 * wall / K, "higher_is_better": True,
            "scaling": "strong" if (strong or world == 1) else "weak", "vs_baseline": None,
            "dtype": "f32" if args.precision == "float32" else "f64", "data": "synthetic",
            "config": {"workload": workload_text(args, N, cfg),
                       "particles_per_gpu": [c[0] for c in allc] if world > 1 else N, "n_grid": args.grid,
                       "touched_cells": Gsum if world > 1 else G_t, "contact_particles": sum(c[2] for c in allc),
                       "backward": "forward grid recomputed in substep_grad (reference style)" if args.recompute_backward
                       else "forward grid restored from the per-frame checkpoint saved by substep",
                       "resort_interval": args.sort_interval, "resorts_in_windows": int(kern.get("sort", (0, 0))[1]), "substeps_in_windows": prof_substeps,
                       "windows": (f"{len(walls)} windows of exactly {K} substep pairs each, advancing through one episode (frames {W} .. {W + len(walls) * K}); "
                                   "the loss seed is added from HBM inside each window; value = K / mean window time, no window dropped"),
                       "parallelism": par},
            "slab_runner": None if world == 1 else (getattr(run, "fallback_note", None) or (("in-library slab loop over the IPC test transport" if os.environ.get("SMAC_COMM_STUB") == "2" else "in-library RCCL loop") if args.slab_runner == "lib" else "Python SlabRunner")),
            "transport": None if world == 1 else transport_note(run, sim, dist, world),
            "migrations_in_window": int(getattr(timed_windows, "migrations", 0)), "particles_migrated": int(getattr(timed_windows, "moved", 0)),
            "multi_gpu_note": None if world == 1 else ("no N > 1 run on N GPUs existed when this code was committed: the in-library slab loop has run between two and three "
                                                         "ranks over the IPC test transport on one GPU (tests/test_slabs.py) and as a world-1 RCCL self exchange"),
            "repeats": len(walls), "aggregate": "mean", "ms_per_step_all": [round(1e3 * w / K, 5) for w in walls],
            "spread": (max(walls) - min(walls)) / wall, "drift_repairs": int(sim.get_param("drift_repairs")), "resorts": int(sim.get_param("resorts")), "chunks": int(sim.get_param("chunks")),
            "device_ms_per_step": dev_ms / K,
            "roofline": {"bound": "hbm", "kernel": dom, "achieved": achieved, "peak": PEAK_HBM_GBS, "unit": "GB/s",
                         "frac": achieved / PEAK_HBM_GBS, "traffic": traffic, "traffic_note": traffic_note,
                         "avg_launch_ms": avg_ms, "algorithmic_bytes_per_launch": alg},
            "roofline_substep": {"algorithmic_bytes_fwd_bwd": ab["fwd"] + ab["bwd"], "achieved": sub_gbs,
                                 "peak": PEAK_HBM_GBS, "unit": "GB/s", "frac": sub_gbs / PEAK_HBM_GBS},
            "kernels_ms": {k: round(v[0] / v[1], 4) for k, v in kern.items()},
            "kernels_ms_per_step": {k: round(v, 4) for k, v in per_step.items()},
        }
        if split is not None:
            for name, t, key in (("fwd_only", split[0], "fwd"), ("bwd_only", split[1], "bwd")):
                gbs = ab[key] * K / t / 1e9
                out[name] = {"value": K / t, "unit": "substeps/s", "ms_per_step": 1e3 * t / K, "algorithmic_bytes": ab[key],
                             "achieved": gbs, "peak": PEAK_HBM_GBS, "frac": gbs / PEAK_HBM_GBS,
                             "note": f"{K} {'forward' if key == 'fwd' else 'backward'} substeps of the same window, one sync before and after"}
    if world == 1 and args.precision == "float32" and not args.no_f64 and args.workload in ("s-grip", "s-elastic"):
        # the mode that computes in the reference's own dtype (mpm_simulator.py:19) and meets 1e-9: one window, same workload
        del run
        sim._h.close()
        del sim
        a64 = argparse.Namespace(**vars(args))
        a64.precision = "float64"
        R64 = math.lcm(K, args.sort_interval) // K if math.lcm(K, args.sort_interval) <= 640 else 1      # one block of windows with the re-sorts' share
        sim, run, cfg = build_sim(a64, rank, world, frames=W + R64 * K + 2)
        w64, d64, _, _ = timed_windows(a64, sim, run, None, seed_gx, lambda: sim.sync(), None, windows=R64)
        w64, d64 = [sum(w64) / len(w64)], [sum(d64) / len(d64)]
        out["f64"] = {"value": K / w64[0], "unit": "substeps/s", "ms_per_step": 1e3 * w64[0] / K, "device_ms_per_step": d64[0] / K,
                      "dtype": "f64", "note": "same workload, arithmetic and storage in float64 (parity 1e-9 state / 1e-8 gradients)"}
    if world == 1 and not args.no_env_loop and args.workload == "s-grip":
        try:
            sim._h.close()
            out["env_loop"] = env_loop_record(args, seed_gx)
            out["env_loop"]["vs_value"] = out["env_loop"]["value"] / out["value"]
        except Exception as e:                                                     # noqa: BLE001
            out["env_loop"] = {"error": f"{type(e).__name__}: {e}"[:300]}
    if world == 1 and args.precision == "float32" and not args.no_cloth and args.workload == "s-grip":
        # the soft <-> cloth path at the same size, in a child process: whatever happens there cannot touch the metric line
        try:
            sim._h.close()
            import subprocess
            r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "bench_cloth.py"), "--particles", str(args.particles), "--grid", str(args.grid)],
                               capture_output=True, text=True, timeout=300)
            line = [l for l in r.stdout.splitlines() if l.startswith("{")]
            out["cloth"] = json.loads(line[-1]) if line else {"error": (r.stderr or "no output")[-300:]}
        except Exception as e:                                                     # noqa: BLE001
            out["cloth"] = {"error": f"{type(e).__name__}: {e}"[:300]}
    if rank == 0:
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(args)
        print(json.dumps(out), flush=True)
    if world == 1:
        try:
            sim._h.close()                                   # (destroys the handle: a -DSMAC_PHASE_CLOCK build dumps its markers there, tools/phase_clock.py)
        except Exception:                                    # noqa: BLE001 - closed already by a sub-record
            pass
    if watch is not None:
        watch.close()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
