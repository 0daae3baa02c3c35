"""One process of the fast/slow-mode study of k_g2p (profiles/HISTORY.md 7): builds the bench scene, runs a few forward substeps and prints the
mean k_g2p time measured with HIP events (run under `rocprofv3 --pmc ...` the kernel trace carries durations and counters too)."""
import sys, pathlib
sys.path.insert(0, str(pathlib.Path(__file__).resolve().parent.parent))
import bench

a = bench.parse_args(["--steps", "12", "--warmup", "4", "--no-cpu-baseline", "--no-f64", "--no-cloth", "--repeats", "1"])
sim, run, cfg = bench.build_sim(a, 0, 1)
run.run_substeps(0, 4)
sim.profile(True)
run.run_substeps(4, 12)
prof = sim.profile_report()
print("g2p_us %.1f p2g_us %.1f" % (1e3 * prof["g2p"][0] / prof["g2p"][1], 1e3 * prof["p2g"][0] / prof["p2g"][1]), flush=True)
