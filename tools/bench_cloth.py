"""Timing of the soft <-> cloth path at full size (scenes.s_taco: 1M particles, 128^3, mpm_scale 5, von-Mises plasticine on a sticky
sheet): one env step of the reference's loop (soft_cloth/engine/taichi_env.py:86-106 - substep, contact-face search, penetration
tracing per substep) forward, then its backward pass.  Prints one JSON line.   python tools/bench_cloth.py [--particles N] [--grid G]"""
import argparse, json, sys, pathlib, time
sys.path.insert(0, str(pathlib.Path(__file__).resolve().parent.parent))
import ctypes as C
import numpy as np
from softmac_amd import scenes
from softmac_amd.config import CfgNode
from softmac_amd.soft_cloth.engine.mpm_simulator import MPMSimulator
from softmac_amd.soft_cloth.engine.primitive import Primitive_Cloth

ap = argparse.ArgumentParser()
ap.add_argument("--particles", type=int, default=1 << 20)
ap.add_argument("--grid", type=int, default=128)
ap.add_argument("--precision", default="float32")
ap.add_argument("--env-steps", type=int, default=3)
a = ap.parse_args()
cfg, env_dt, scale, state, V, F, prim_cfg = scenes.s_taco(a.particles, a.grid, max_steps=10 * a.env_steps + 4, precision=a.precision)
pc = CfgNode(prim_cfg)
prim = Primitive_Cloth(pc, max_timesteps=cfg.max_steps, mpm_scale=scale, vertices=V, faces=F)
sim = MPMSimulator(cfg, prim, env_dt, scale)
prim.initialize()
sub = sim.substeps
prim.set_all_states(0, V, np.zeros_like(V), f_end=cfg.max_steps)
sim.reset(state)
sim.get_contact_pair(0)


def timed(fn):
    sim.sync(); t = time.perf_counter(); fn(); sim.sync(); return (time.perf_counter() - t) * 1e3


def env_step(start):
    for s in range(start, start + sub):
        sim.substep(s)
        sim.get_contact_pair(s + 1)
        sim.trace_penetration_after_mpm(s + 1)


def env_step_grad(start):
    for s in range(start + sub - 1, start - 1, -1):
        sim.substep_grad(s)


env_step(0)                                               # warm-up (first re-sort, allocations)
fwd = [timed(lambda k=k: env_step(k * sub)) for k in range(1, a.env_steps)]
n = a.env_steps * sub
sim.clear_grads()
sim.add_grad(n, gx=np.ones((a.particles, 3)))
bwd = [timed(lambda k=k: env_step_grad(k * sub)) for k in range(a.env_steps - 1, 0, -1)]
f = n
t_sub = timed(lambda: [sim.substep(s) for s in range(f - sub, f)]) / sub          # (re-running the last env step's substeps alone)
t_pair = timed(lambda: [sim.get_contact_pair(f) for _ in range(10)]) / 10
sim._h.call("smac_set_param", b"cloth_pairs_flat", C.c_double(1.0))
t_pair_flat = timed(lambda: [sim.get_contact_pair(f) for _ in range(3)]) / 3
sim._h.call("smac_set_param", b"cloth_pairs_flat", C.c_double(0.0))
t_trace = timed(lambda: [sim.trace_penetration_after_mpm(f) for _ in range(10)]) / 10
ids, pen = sim.get_contact(f)
print(json.dumps({"workload": f"s-taco: {a.particles} particles, {a.grid}^3, mpm_scale 5, von Mises, sticky sheet of {len(F)} faces", "dtype": a.precision,
                  "env_step_forward_ms": round(float(np.median(fwd)), 3), "env_step_backward_ms": round(float(np.median(bwd)), 3), "substeps_per_env_step": sub,
                  "substeps_per_s_fwd_bwd_incl_search_and_tracing": round(sub / (np.median(fwd) + np.median(bwd)) * 1e3, 1),
                  "substep_forward_ms": round(t_sub, 4), "contact_pair_ms_chunk_culled": round(t_pair, 4), "contact_pair_ms_flat": round(t_pair_flat, 4),
                  "trace_penetration_ms": round(t_trace, 4), "particles_with_contact_face": int((ids >= 0).sum()), "penetrated": int((pen == 1).sum())}))
