"""Seeded random configurations of the substep against the oracle: odd particle counts (1, 63, 65 ...), clouds touching
the walls and the floor, every material / model, every contact type, both precisions, various re-sort intervals.
Each case is small (a few substeps on a 32^3 grid); together they walk the corners the fixed scenes do not."""
import numpy as np
import pytest

import helpers as H
from test_gpu_parity import _compare_rollout, _palm_scene

pytestmark = pytest.mark.gpu

CASES = list(range(40))


def fuzz_case(case):
    """the seeded configuration of one case (also used by tools/prec_probe.py --fuzz)"""
    rng = np.random.default_rng(1000 + case)
    N = int(rng.choice([1, 2, 63, 64, 65, 127, 300, 777, 1500]))
    n_grid = 32
    ptype, model = int(rng.integers(0, 3)), int(rng.integers(0, 2))
    precision = "float64" if case % 2 == 0 else "float32"
    collision_type = int(rng.choice([2, 2, 1, 0]))
    with_prim = bool(rng.integers(0, 2))
    near_wall = bool(rng.integers(0, 2))
    lo = np.array([0.3, 0.3, 0.3]); hi = np.array([0.7, 0.55, 0.7])
    if near_wall:                                   # hug the floor and one side wall (boundary clamps, sticky floor)
        lo = np.array([0.06, 0.04, 0.3]); hi = np.array([0.4, 0.3, 0.7])
    state = H.make_cloud(N, n_grid, seed=case, lo=tuple(lo), hi=tuple(hi), v_std=float(rng.choice([0.1, 1.0, 3.0])),
                         F_std=float(rng.choice([1e-3, 2e-2])))
    steps = int(rng.integers(2, 6))
    cfg = H.sim_cfg(N, n_grid=n_grid, dt=2e-4, ptype=ptype, material_model=model, E=22.0 if ptype == 2 else 3e3,
                    ground_friction=float(rng.choice([0.0, 1.5, 20.0])), collision_type=collision_type, precision=precision,
                    sort_interval=int(rng.choice([1, 2, 3, 16])), max_steps=8,
                    gravity=(0.0, -9.8, 0.0) if rng.integers(0, 2) else (0.0, 0.0, 0.0))
    specs, pstates = ((), None)
    if with_prim:
        # palm box lowered onto the top of the cloud
        specs, pstates = _palm_scene(np.concatenate([state[:, :3], state[:, 3:]], axis=1), steps)
        top = state[:, 1].max()
        for f in range(len(pstates)):
            pstates[f][0][1] = top + 0.15 - 0.004 - 2e-4 * 0.3 * f
    return dict(cfg=cfg, state=state, steps=steps, specs=specs, pstates=pstates, near_wall=near_wall, with_prim=with_prim)


@pytest.mark.parametrize("case", CASES)
def test_random_configuration(case):
    c = fuzz_case(case)
    _compare_rollout(c["cfg"], 2e-3, c["state"], c["steps"], c["specs"], c["pstates"], seed=case)   # tolerances: helpers.F32_TOL / 1e-9, 1e-8
