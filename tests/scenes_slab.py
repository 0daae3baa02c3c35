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
