"""`python bench.py --gpus N` must really start N ranks (VERDICT r1 #3 / ADVICE): the launcher itself is driven here on the
CPU (`--launch-check`: spawn, gloo rendezvous, report - no simulator, the product has no CPU path); on the GPU box the
2-rank slab bench goes through the same launcher with both ranks on the one card."""
import json
import math
import os
import pathlib
import subprocess
import sys

import pytest

ROOT = pathlib.Path(__file__).resolve().parent.parent


def _run(args, env=None, timeout=600):
    e = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        e.pop(k, None)
    e.update(env or {})
    return subprocess.run([sys.executable, str(ROOT / "bench.py")] + args, capture_output=True, text=True, env=e, timeout=timeout, cwd=str(ROOT))


def _last_json(out):
    lines = [l for l in out.splitlines() if l.startswith("{")]
    assert lines, out
    return json.loads(lines[-1])


@pytest.mark.parametrize("n", [2, 4])
def test_gpus_flag_spawns_that_many_ranks(n):
    r = _run(["--gpus", str(n), "--launch-check"])
    assert r.returncode == 0, r.stderr[-2000:]
    d = _last_json(r.stdout)
    assert d["launch_check"] and d["n_gpus"] == n and d["requested"] == n
    assert [x["rank"] for x in d["ranks"]] == list(range(n))
    assert len({x["pid"] for x in d["ranks"]}) == n                      # n distinct processes


def test_world_size_mismatch_fails_loudly():
    r = _run(["--gpus", "4", "--launch-check"], env={"WORLD_SIZE": "2", "RANK": "0", "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": "29999"})
    assert r.returncode != 0 and "WORLD_SIZE=2" in (r.stderr + r.stdout)


def test_single_process_launch_check():
    r = _run(["--gpus", "1", "--launch-check"], env={"WORLD_SIZE": "1", "RANK": "0", "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": "29998"})
    assert r.returncode == 0, r.stderr[-2000:]
    assert _last_json(r.stdout)["n_gpus"] == 1


@pytest.mark.gpu
def test_two_rank_strong_scaling_bench_through_the_launcher():
    """2 ranks on the one GPU of the test box (gloo staging, SMAC_FORCE_DEVICE=0): the strong-scaling slab path end to end.  Two ranks cannot
    share one device under RCCL, so this leg drives the Python SlabRunner (`--slab-runner python`); the in-library RCCL runner that `--gpus N`
    uses by default is exercised on one GPU by tests/test_gpu_slab_lib.py (world-1 self exchange)."""
    r = _run(["--gpus", "2", "--steps", "12", "--warmup", "2", "--repeats", "2", "--sort-interval", "12", "--particles", "65536", "--grid", "64", "--no-cpu-baseline",
              "--slab-runner", "python"],
             env={"SMAC_DIST_BACKEND": "gloo", "SMAC_FORCE_DEVICE": "0"}, timeout=900)
    assert r.returncode == 0, (r.stdout[-1000:], r.stderr[-3000:])
    d = _last_json(r.stdout)
    assert d["n_gpus"] == 2 and d["scaling"] == "strong" and d["value"] > 0
    assert sum(d["config"]["particles_per_gpu"]) == 65536
    assert d["config"]["resorts_in_windows"] >= 1


def test_window_plan_gives_the_resorts_their_steady_state_share():
    """bench.py times R windows of exactly K substep pairs that advance through one episode; R K is a multiple of the re-sort interval wherever that fits
    in 480 resident frames, so that windows without and with a re-sort appear in their steady-state proportion."""
    sys.path.insert(0, str(ROOT))
    import bench
    for K, I in ((20, 40), (32, 40), (40, 40), (20, 20), (64, 40), (7, 40), (100, 40), (20, 80), (32, 80)):
        R = bench.window_plan(K, I)
        assert R >= 1 and (R * K) % I == 0, (K, I, R)
        assert R * K <= 480 or R == math.lcm(K, I) // K, (K, I, R)
    assert bench.window_plan(20, 40) == 8 and bench.window_plan(32, 40) == 10          # the driver's flags, the default flags (rounds 4 - 5)
    assert bench.window_plan(20, bench.LIB_SORT_INTERVAL) == 8 and bench.window_plan(32, bench.LIB_SORT_INTERVAL) == 10      # ... and at the library's default interval of today
    assert bench.window_plan(1000, 40) == 1                                           # one window is the least there is


def test_workloads_take_the_sizes_and_time_steps_of_the_survey():
    """S-grip 1M / 128^3 at dt 1e-4, S-elastic 262,144 / 64^3, S-pour 4M / 256^3 at 2.5e-4, S-mixed 16M / 256^3; `metric` is BASELINE.json's only for
    the headline configuration; S-grip's dt keeps c dt / dx at every resolution (1e-4 at 256^3 diverged)."""
    sys.path.insert(0, str(ROOT))
    import bench
    from softmac_amd import scenes
    want = {"s-grip": (1 << 20, 128, 32), "s-elastic": (1 << 18, 64, 32), "s-pour": (1 << 22, 256, 32), "s-mixed": (1 << 24, 256, 5)}
    for w, (n, g, k) in want.items():
        a = bench.parse_args(["--workload", w])
        assert (a.particles, a.grid, a.steps) == (n, g, k), (w, a)
        assert (bench.metric_name(a) == bench.baseline_metric()) == (w == "s-grip")
    assert "not the headline" in bench.metric_name(bench.parse_args(["--particles", "65536", "--grid", "64"]))
    for g in (64, 128, 256):
        assert abs(scenes.grip_dt(g) * g * 57.735 - 0.739) < 1e-3          # c dt / dx with c = sqrt((lam + 2 mu) / rho) = 57.7 m/s
    cfg = scenes.s_grip(4096, 256, 4)[0]
    assert cfg.dt == 5e-5
    assert scenes.s_pour(4096, 256, 4)[0].dt == 2.5e-4
