"""Host + launch overhead of the Python-driven slab loop WITHOUT the communication itself: one process pretends to be the middle rank of three
(two neighbours), packs / unpacks the shared planes as SlabRunner does, and skips the send/recv (the unpacked data is its own).  The difference
to the batched C loop is what the loop costs per substep pair before any byte moves."""
import sys, time, pathlib
sys.path.insert(0, str(pathlib.Path(__file__).resolve().parent.parent))
import numpy as np, torch
import torch.distributed as dist
import bench
from softmac_amd.parallel import HipSlabEngine, SlabRunner

a = bench.parse_args(["--steps", "32", "--warmup", "8", "--no-cpu-baseline", "--no-f64", "--no-cloth", "--repeats", "1"])
sim, run, cfg = bench.build_sim(a, 0, 1)
eng = HipSlabEngine(sim)
sr = SlabRunner(eng, 1, 3, 40, 86, nplanes=4, has_contact=True)       # middle rank: both neighbours exist


class _Done:
    def wait(self):
        return None


calls = {"n": 0}


def fake_batch(ops):
    calls["n"] += 1
    for op in ops[1::2]:                   # every irecv gets the matching isend's buffer (same process)
        pass
    return [_Done()]


dist.batch_isend_irecv = fake_batch
dist.P2POp = lambda fn, t, peer, group=None: (fn, t, peer)
W, K = 8, 32
N = int(cfg.n_particles)
for name, r in (("batched C loop", run), ("python phases + pack/unpack, no comm", sr), ("batched C loop", run), ("python phases + pack/unpack, no comm", sr)):
    r.run_substeps(0, W); sim.clear_grads(); sim.add_grad(W, gx=np.zeros((N, 3))); r.run_substeps_grad(0, W)
    sim.clear_grads(); sim.add_grad(W + K, gx=np.zeros((N, 3)))
    sim.sync(); torch.cuda.synchronize(); t0 = time.perf_counter()
    r.run_substeps(W, K)
    t1 = time.perf_counter()               # host time to ENQUEUE the forward half
    r.run_substeps_grad(W, K)
    t2 = time.perf_counter()
    sim.sync(); torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print(f"{name:40s} {K / dt:7.1f} substeps/s  {dt / K * 1e6:7.1f} us per pair  (host enqueue: fwd {1e6 * (t1 - t0) / K:6.1f} us, bwd {1e6 * (t2 - t1) / K:6.1f} us per substep)", flush=True)
print("exchanges per pair:", calls["n"] / (2 * (W + K)))
