#!/usr/bin/env python3
"""bench.py - MPM substeps/s (forward + backward) on the S-grip workload (BASELINE config C3:
1,048,576 particles, 128^3 grid, plasticine + gripper SDF forecast contact), synthetic seeded data.

One "step" = one substep() + one substep_grad() (its forward recompute included), run as K forward
substeps followed by K backward substeps with inputs resident in HBM.  Contract: SURVEY/driver
(`python bench.py --gpus N --steps K --warmup W`, one JSON line on rank 0).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_HBM_GBS = 8000.0         # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec


def baseline_metric():
    """The headline metric, named exactly as BASELINE.json names it."""
    try:
        return json.load(open(os.path.join(ROOT, "BASELINE.json")))["metric"]
    except Exception:
        return "MPM substeps/s (fwd+bwd) at 1M particles/128\u00b3 grid, 1/2/4/8 MI355X"


def build_sim(args, rank, world):
    from softmac_amd import scenes
    from softmac_amd.config import CfgNode
    from softmac_amd.engine.mpm_simulator import MPMSimulator
    from softmac_amd.engine.primitive import Mesh, Primitives
    frames = args.warmup + args.steps + 2
    dev = int(os.environ.get("SMAC_FORCE_DEVICE", os.environ.get("LOCAL_RANK", 0))) if world > 1 else 0
    slab = None
    if args.workload == "s-grip" and world > 1:
        cfg, env_dt, state, specs, s13, slab = scenes.s_grip_slab(rank, world, args.particles, args.grid, frames, args.precision, dev)
    elif args.workload == "s-grip":
        cfg, env_dt, state, specs, s13 = scenes.s_grip(args.particles, args.grid, frames, args.precision, dev, seed=1 + rank)
    else:
        cfg, env_dt, state, specs, s13 = scenes.s_elastic(args.particles, args.grid, frames, args.precision, dev, seed=rank)
    cfg.recompute_backward = args.recompute_backward
    cfg.sort_interval = args.sort_interval
    meshes = []
    for s in specs:
        pc = CfgNode(); pc.friction = s["friction"]; pc.enable_external_force = True; pc.urdf_path = ""
        meshes.append(Mesh(sdf=s, cfg=pc, max_timesteps=frames))
    prims = Primitives(primitives=meshes)
    sim = MPMSimulator(cfg, prims, env_dt)
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
    if slab is not None:                                   # slab decomposition: halo exchange of 2 shared x-planes over RCCL
        from softmac_amd.parallel import HipSlabEngine, SlabRunner
        runner = SlabRunner(HipSlabEngine(sim, use_torch_stream=True), rank, world, slab[0], slab[1], 2, has_contact=True)
    return sim, runner, cfg


def algorithmic_bytes(N, G_t, s):
    """per substep, fwd = 48 s N + 20 s G_t ; bwd = 72 s N + 40 s G_t."""
    return dict(fwd=48 * s * N + 20 * s * G_t, bwd=72 * s * N + 40 * s * G_t)


# compulsory bytes of each kernel taken alone (DESIGN.md "kernels"): scalars per particle, scalars per touched cell
KERNEL_BYTES = {
    "p2g": (24 + 9, 4), "grid_op": (0, 4 + 6), "contact": (0, 0), "g2p": (3 + 15, 3),
    "g2p_grad": (3 + 15 + 3, 3 + 3), "contact_grad": (0, 0), "grid_op_grad": (0, 4 + 6 + 4), "p2g_grad": (24 + 9 + 3 + 24, 4),
    "clear_grid": (0, 10), "grid_checkpoint": (0, 20), "reduce_agvout": (0, 6), "forward_kinematics": (0, 0),
    "sort": (48, 0), "reorder_adjoint": (48, 0),
}


def cpu_baseline(args):
    """The plain C++/OpenMP f64 oracle port (oracle/mpm_cpu.cpp: the reference's decomposition - dense grid,
    one pass per Taichi kernel, atomics) timed on this box's host cores on a bounded sample of the SAME workload."""
    from softmac_amd import scenes
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import helpers as H
    from oracle import mpm_cpu
    cfg, env_dt, state, specs, s13 = scenes.s_grip(args.particles, args.grid, 8, "float64", 0, seed=1)
    P = H.oracle_params(cfg, env_dt)
    port = mpm_cpu.CpuPort(P, specs)
    N = args.particles
    x, v = state[:, 0:3].copy(), state[:, 3:6].copy()
    F, C = state[:, 6:15].reshape(N, 3, 3).copy(), state[:, 15:24].reshape(N, 3, 3).copy()
    nsub = args.cpu_steps
    rng = np.random.default_rng(0)
    frames = [(x, v, C, F)]
    psts = []
    t0 = time.perf_counter()
    for f in range(nsub):
        pst = np.array([a + np.concatenate([f * cfg.dt * a[7:10], np.zeros(10)]) for a in s13])
        psts.append(pst)
        nx, nv, nC, nF, _ = port.substep(f, *frames[-1], pst)
        frames.append((nx, nv, nC, nF))
    g = [rng.standard_normal((N, 3)), np.zeros((N, 3)), np.zeros((N, 3, 3)), np.zeros((N, 3, 3))]
    for f in range(nsub - 1, -1, -1):
        g = list(port.substep_grad(f, *frames[f], *g, pst=psts[f])[:4])
    dt = time.perf_counter() - t0
    return {"value": nsub / dt, "unit": "substeps/s (fwd+bwd)", "cores": port.threads(), "kind": "port",
            "sample": f"{nsub} forward + {nsub} backward substeps of the full workload ({N} particles, {args.grid}^3, 3 primitives), "
                      f"f64 C++/OpenMP restatement of the reference's kernel decomposition (not Taichi), {dt:.1f} s on {port.threads()} threads"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=32)
    ap.add_argument("--warmup", type=int, default=4)
    ap.add_argument("--workload", default="s-grip", choices=["s-grip", "s-elastic"])
    ap.add_argument("--particles", type=int, default=1 << 20)
    ap.add_argument("--grid", type=int, default=128)
    ap.add_argument("--precision", default="float32", choices=["float32", "float64"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-steps", type=int, default=8)
    ap.add_argument("--recompute-backward", action="store_true", help="substep_grad recomputes the forward grid (reference style)")
    ap.add_argument("--sort-interval", type=int, default=0)
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", 1))
    rank = int(os.environ.get("RANK", 0))
    dist = None
    if world > 1:
        import torch
        import torch.distributed as dist
        torch.cuda.set_device(int(os.environ.get("SMAC_FORCE_DEVICE", os.environ.get("LOCAL_RANK", 0))))
        dist.init_process_group(os.environ.get("SMAC_DIST_BACKEND", "nccl"))      # "nccl" is RCCL on ROCm

    sim, run, cfg = build_sim(args, rank, world)
    N, K, W = args.particles, args.steps, args.warmup
    sbytes = 4 if args.precision == "float32" else 8
    rng = np.random.default_rng(7 + rank)
    seed_gx = rng.standard_normal((N, 3))

    def barrier():
        sim.sync()
        if dist is not None:
            import torch
            dist.barrier()
            torch.cuda.synchronize()

    # warmup: W fwd + W bwd on frames [0, W)
    run.run_substeps(0, W)
    sim.clear_grads()
    sim.add_grad(W, gx=seed_gx)
    run.run_substeps_grad(0, W)
    sim.clear_grads()
    sim.add_grad(W + K, gx=seed_gx)
    for m in sim.primitives:
        m.clear_ext_f()
    barrier()

    # timed: K forward substeps then K backward substeps, frames [W, W+K)
    t0 = time.perf_counter()
    sim.timer_start()
    run.run_substeps(W, K)
    run.run_substeps_grad(W, K)
    dev_ms = sim.timer_stop()
    barrier()
    wall = time.perf_counter() - t0
    if dist is not None:
        import torch
        t = torch.tensor([wall], device="cuda" if dist.get_backend() == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        wall = float(t.item())
    value = world * K / wall

    # per-kernel HIP-event profile over a second identical pass (kept out of the timed region)
    sim.clear_grads()
    sim.add_grad(W + K, gx=seed_gx)
    sim.profile(True)
    run.run_substeps(W, K)
    run.run_substeps_grad(W, K)
    prof = sim.profile_report()
    sim.profile(False)
    G_t = sim.count_active_cells(W)
    n_hits, n_hit_chunks = sim.contact_counts()

    if rank == 0:
        kern = {k: v for k, v in prof.items() if v[1] > 0}
        dom = max(kern, key=lambda k: kern[k][0])
        avg_ms = kern[dom][0] / kern[dom][1]
        pp, pc = KERNEL_BYTES[dom]
        alg = (pp * N + pc * G_t) * sbytes
        achieved = alg / (avg_ms * 1e-3) / 1e9
        ab = algorithmic_bytes(N, G_t, sbytes)
        sub_gbs = (ab["fwd"] + ab["bwd"]) * (K / (dev_ms * 1e-3)) / 1e9
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "traffic_latest.json")
        if os.path.exists(tpath):
            try:
                traffic = json.load(open(tpath)).get(dom)
            except Exception:
                traffic = None
        out = {
            "metric": baseline_metric(),
            "value": value, "unit": "substeps/s", "n_gpus": world, "steps": K, "warmup": W,
            "ms_per_step": 1e3 * wall / K, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32" if args.precision == "float32" else "f64", "data": "synthetic",
            "config": {"workload": (f"{args.workload}: {N} particles, {args.grid}^3 grid, plastic fixed-corotated, "
                                    f"3 gripper SDF primitives (2 in forecast contact), fwd+bwd") if args.workload == "s-grip" else
                                   f"{args.workload}: {N} particles, {args.grid}^3 grid, elastic fixed-corotated, no primitives, fwd+bwd",
                       "particles_per_gpu": N, "n_grid": args.grid, "touched_cells": G_t, "contact_particles": n_hits,
                       "backward": "forward grid recomputed in substep_grad (reference style)" if args.recompute_backward
                       else "forward grid restored from the per-frame checkpoint saved by substep",
                       "parallelism": "1 gpu" if world == 1 else
                       f"{world} x-slabs of one bar, {N} particles each; per substep neighbour-only RCCL send/recv of 2 shared grid planes "
                       f"(fwd: m,p + contact corrections; bwd: grid_v_out.grad + grid_v_mixed.grad)"},
            "device_ms_per_step": dev_ms / K,
            "roofline": {"bound": "hbm", "kernel": dom, "achieved": achieved, "peak": PEAK_HBM_GBS, "unit": "GB/s",
                         "frac": achieved / PEAK_HBM_GBS, "traffic": traffic,
                         "avg_launch_ms": avg_ms, "algorithmic_bytes_per_launch": alg},
            "roofline_substep": {"algorithmic_bytes_fwd_bwd": ab["fwd"] + ab["bwd"], "achieved": sub_gbs,
                                 "peak": PEAK_HBM_GBS, "unit": "GB/s", "frac": sub_gbs / PEAK_HBM_GBS},
            "kernels_ms": {k: round(v[0] / v[1], 4) for k, v in kern.items()},
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(args)
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
