"""What the slab loop costs per substep pair BEFORE any byte crosses a link, on one GPU at the bench's size:

  batched C loop          smac_substeps / smac_substeps_grad, no slabs                                      (the single-GPU path)
  python phase loop       parallel.SlabRunner: 3 phase calls + pack / unpack per side from Python, communication faked     (round 2's slab loop)
  library loop, stub      smac_substeps_slab[_grad] with SMAC_COMM_STUB=1: phases, two-sided pack / unpack, event hand-offs, device copies for the
                          RCCL calls - this rank pretends to be a middle slab (both neighbours), left = right = itself
  library loop, RCCL      the same with the real ncclSend / ncclRecv group on the communication stream (world-1 self exchange)

Host enqueue time is what bounds strong scaling: at 8 GPUs a rank has about 45 us of kernels per pair."""
import os, sys, time, pathlib
sys.path.insert(0, str(pathlib.Path(__file__).resolve().parent.parent))
import numpy as np, torch
import torch.distributed as dist
import bench
from softmac_amd.parallel import HipSlabEngine, LibSlabRunner, SlabRunner

a = bench.parse_args(["--steps", "32", "--warmup", "8", "--no-cpu-baseline", "--no-f64", "--no-cloth", "--repeats", "1"])
W, K = 8, 32


class _Done:
    def wait(self):
        return None


calls = {"n": 0}


def fake_batch(ops):
    calls["n"] += 1
    return [_Done()]


def measure(name, sim, r, N):
    r.run_substeps(0, W); sim.clear_grads(); sim.add_grad(W, gx=np.zeros((N, 3))); r.run_substeps_grad(0, W)
    sim.clear_grads(); sim.add_grad(W + K, gx=np.zeros((N, 3)))
    sim.sync(); torch.cuda.synchronize(); t0 = time.perf_counter()
    r.run_substeps(W, K)
    t1 = time.perf_counter()               # host time to ENQUEUE the forward half
    r.run_substeps_grad(W, K)
    t2 = time.perf_counter()
    sim.sync(); torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print(f"{name:40s} {K / dt:7.1f} substeps/s  {dt / K * 1e6:7.1f} us per pair  (host enqueue: fwd {1e6 * (t1 - t0) / K:6.1f} us, bwd {1e6 * (t2 - t1) / K:6.1f} us per substep)", flush=True)


def build(stub, own_stream=0):
    os.environ["SMAC_COMM_STREAM"] = str(own_stream)
    if stub:
        os.environ["SMAC_COMM_STUB"] = "1"
    else:
        os.environ.pop("SMAC_COMM_STUB", None)
    sim, run, cfg = bench.build_sim(a, 0, 1)
    return sim, run, int(cfg.n_particles)


which = sys.argv[1:] or ["c", "python", "stub", "rccl"]
for rep in range(2):
    if "c" in which or "python" in which:
        sim, run, N = build(False)
        if "c" in which:
            measure("batched C loop (no slabs)", sim, run, N)
        if "python" in which:
            real = (dist.batch_isend_irecv, dist.P2POp)
            dist.batch_isend_irecv = fake_batch
            dist.P2POp = lambda fn, t, peer, group=None: (fn, t, peer)
            sr = SlabRunner(HipSlabEngine(sim), 1, 3, 40, 86, nplanes=4, has_contact=True)       # middle rank: both neighbours exist
            measure("python phase loop, communication faked", sim, sr, N)
            dist.batch_isend_irecv, dist.P2POp = real
        sim._h.close()
    for label, stub, own in (("stub", True, 0), ("rccl", False, 0), ("stub2", True, 1), ("rccl2", False, 1)):
        if label not in which and label.rstrip("2") not in which:
            continue
        sim, run, N = build(stub, own)
        lr = LibSlabRunner(sim, 0, 1, 40, 86, 4, has_contact=(True, True), self_loop=True)
        measure("library loop, " + ("device copies" if stub else "RCCL self exchange") + (", own stream + 2 events" if own else ", kernels' stream"), sim, lr, N)
        print("    exchanges per pair:", lr.exchanges() / (W + K))
        lr.close()
        sim._h.close()
