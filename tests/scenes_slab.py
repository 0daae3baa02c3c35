"""The 2-slab test scene: a block of plasticine straddling the split plane, with the palm pressing on it right
at the interface, so that P2G sums, contact corrections and every adjoint cross the slab boundary."""
import numpy as np

import helpers as H


def build(precision="float64"):
    n_grid, N = 32, 1600
    state = H.make_cloud(N, n_grid, seed=31, lo=(0.36, 0.1, 0.36), hi=(0.64, 0.3, 0.64), v_std=0.2, F_std=5e-3)
    cfg = H.sim_cfg(N, n_grid=n_grid, dt=2e-4, ptype=0, material_model=0, ground_friction=20.0, precision=precision, max_steps=8)
    palm = H.load_palm()
    spec = dict(palm, friction=0.4, softness=666.0, contact=True)
    q = np.array([0.995, 0.02, 0.03, 0.09]); q /= np.linalg.norm(q)
    s = np.concatenate([[0.5, 0.3 + 0.15 - 0.004, 0.5], q, [0.02, -0.3, 0.01], [0.1, 0.05, -0.2]])
    nsteps = 2
    pstates = []
    for f in range(nsteps + 1):
        pstates.append([s.copy()])
        s[:3] = s[:3] + 2e-4 * np.array([0.02, -0.3, 0.01])
    return dict(cfg=cfg, env_dt=1e-3, state=state, nsteps=nsteps, specs=[spec], pstates=pstates, split=16,
                ext_f_grad=[np.linspace(-1e-2, 1e-2, 6)], n_grid=n_grid)


def owned(sc, rank, world):
    """rank 0 owns particles whose stencil base.x < split (in every frame of the short window), rank 1 the rest"""
    base = (sc["state"][:, 0] * sc["n_grid"] - 0.5).astype(int)
    left = base < sc["split"]
    return np.nonzero(left if rank == 0 else ~left)[0]


def seeds(sc):
    N = len(sc["state"])
    rng = np.random.default_rng(77)
    n = sc["nsteps"]
    return {n: (rng.standard_normal((N, 3)), rng.standard_normal((N, 3)), 0.01 * rng.standard_normal((N, 3, 3)),
                0.01 * rng.standard_normal((N, 3, 3))), 1: (rng.standard_normal((N, 3)), None, None, None)}


def build_moving(precision="float64", world=2):
    """Migration scene: an elastic cloud flying in +x at 20 m/s (0.128 cells per substep at n_grid 32): within the 8-substep
    window every slab boundary is crossed by particles, so ownership must change hands (SlabRunner.migrate every 4 substeps)."""
    n_grid, N = 32, 2400
    state = H.make_cloud(N, n_grid, seed=33, lo=(0.2, 0.3, 0.36), hi=(0.72, 0.5, 0.64), v_std=0.3, F_std=5e-3)
    state[:, 3] += 20.0
    cfg = H.sim_cfg(N, n_grid=n_grid, dt=2e-4, ptype=1, material_model=0, ground_friction=0.0, precision=precision, max_steps=16,
                    gravity=(0.0, -9.8, 0.0), sort_interval=4)
    base = (state[:, 0] * n_grid - 0.5).astype(int)
    from softmac_amd.scenes import balanced_slab_bounds
    bounds = balanced_slab_bounds(base, world, n_grid, min_width=4)
    return dict(cfg=cfg, env_dt=1e-3, state=state, nsteps=8, migrate_every=4, specs=[], pstates=None, ext_f_grad=None, n_grid=n_grid,
                bounds=bounds, drift_tol=1)


def owned_range(sc, rank):
    base = (sc["state"][:, 0] * sc["n_grid"] - 0.5).astype(int)
    lo, hi = sc["bounds"][rank], sc["bounds"][rank + 1]
    return np.nonzero((base >= lo) & (base < hi))[0]
