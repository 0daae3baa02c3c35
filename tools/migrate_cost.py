"""What one particle migration costs on the device (smac_migrate) at the bench's size, against the host path of round 2
(SlabRunner.migrate: get_state of the slab frame, numpy masks, set_state).  One GPU, world-1 self exchange: the particles that leave on one side
re-enter on the other, so rows and ids really cross RCCL as bytes."""
import sys, time, pathlib
sys.path.insert(0, str(pathlib.Path(__file__).resolve().parent.parent))
import numpy as np
import bench
from softmac_amd.parallel import LibSlabRunner

a = bench.parse_args(["--steps", "8", "--warmup", "2", "--no-cpu-baseline", "--no-f64", "--no-cloth", "--repeats", "1"])
sim, run, cfg = bench.build_sim(a, 0, 1)
N = int(cfg.n_particles)
lr = LibSlabRunner(sim, 0, 1, 40, 86, 4, has_contact=(False, False), self_loop=True)
lr.run_substeps(0, 4)
st = sim.get_state(4)
base = np.floor(st[:, 0] * a.grid - 0.5).astype(int)
for lo, hi in ((int(base.min()) + 1, int(base.max())), (int(base.min()) + 4, int(base.max()) - 3)):
    movers = int(((base < lo) | (base >= hi)).sum())
    sim.sync(); t0 = time.perf_counter()
    f = lr.migrate(4, (lo, hi))
    sim.sync(); t1 = time.perf_counter()
    lr.migrate_grad()
    sim.sync(); t2 = time.perf_counter()
    print(f"own range [{lo}, {hi}): {movers} of {N} particles change hands; smac_migrate {1e3 * (t1 - t0):.2f} ms, smac_migrate_grad {1e3 * (t2 - t1):.2f} ms", flush=True)
t0 = time.perf_counter(); s2 = sim.get_state(4); t1 = time.perf_counter()
x, v, F, C = s2[:, 0:3], s2[:, 3:6], s2[:, 6:15].reshape(N, 3, 3), s2[:, 15:24].reshape(N, 3, 3)
sim.set_state(5, (x, v, F, C)); sim.sync(); t2 = time.perf_counter()
print(f"host path of round 2 at the same size: get_state {1e3 * (t1 - t0):.1f} ms + set_state {1e3 * (t2 - t1):.1f} ms (+ the numpy masks and the send / recv of the rows)")
