"""Is the forward path host-bound?    `tools/exchange_overhead.py` found 106.6 us of host time per forward substep of the
batched loop against 122 us of device time.  Two readings: the four launches COST the host 27 us each (then an 8-way strong-scaling rank, with 45 us of
kernels per pair, is host-bound), or the host merely waits for room in a queue the device drains at its own pace (back-pressure).  This tool separates
them on the benchmark scene:

  empty queue   one smac_substep call after a stream sync, timed on the host: nothing ahead of it in the queue - the true cost of enqueueing its launches
  full queue    K substeps in one smac_substeps call without a sync: host time per substep next to the device time per substep

and, with a small cloud (5,000 particles, whose kernels take a few us), the same two numbers where the device cannot be the one that holds the host up."""
import pathlib
import sys
import time

import numpy as np

sys.path.insert(0, str(pathlib.Path(__file__).resolve().parent.parent))
import bench  # noqa: E402


def measure(label, argv, K=40):
    a = bench.parse_args(argv + ["--no-cpu-baseline", "--no-f64", "--no-cloth", "--no-env-loop", "--repeats", "1", "--steps", str(K), "--warmup", "8"])
    a.sort_interval = 1000                                      # no re-sort inside the measured calls
    sim, run, cfg = bench.build_sim(a, 0, 1)
    run.run_substeps(0, 8)
    sim.sync()
    empty = []
    for f in range(8, 8 + K // 2):
        sim.sync()
        t0 = time.perf_counter()
        sim.substep(f)
        empty.append(time.perf_counter() - t0)
    sim.sync()
    f0 = 8 + K // 2
    t0 = time.perf_counter()
    run.run_substeps(f0, K // 2)
    t_host = time.perf_counter() - t0
    sim.sync()
    t_all = time.perf_counter() - t0
    e = np.array(empty[2:]) * 1e6
    print(f"{label:34s} empty queue: {np.median(e):6.1f} us per substep call (min {e.min():.1f});   full queue: host {1e6 * t_host / (K // 2):6.1f} us per substep, "
          f"device {1e6 * t_all / (K // 2):6.1f} us per substep", flush=True)


if __name__ == "__main__":
    measure("S-grip 1M / 128^3 (benchmark)", [])
    measure("S-elastic 5k / 64^3", ["--workload", "s-elastic", "--particles", "5000", "--grid", "64"])
