"""Golden-vector scenes shared by tools/make_golden.py, tests/test_oracle.py and the GPU parity tests."""
import numpy as np

import helpers as H

SCENES = ("grip_contact", "pour_liquid", "cloud_elastic")


def palm_states(state, nframes, dt):
    top = state[:, 1].max()
    q = np.array([0.995, 0.02, 0.03, 0.09]); q /= np.linalg.norm(q)
    s = np.concatenate([[0.5, top + 0.15 - 0.004, 0.5], q, [0.02, -0.3, 0.01], [0.1, 0.05, -0.2]])
    out = []
    for f in range(nframes + 1):
        out.append([s.copy()])
        s[:3] = s[:3] + dt * np.array([0.02, -0.3, 0.01])
    return out


def build(name):
    if name == "grip_contact":
        state = np.load(H.GOLDEN / "grip_state_2k.npz")["state"]
        cfg = H.sim_cfg(len(state), n_grid=64, dt=2e-4, ptype=0, material_model=0, ground_friction=20.0)
        spec = dict(H.load_palm(), friction=0.4, softness=666.0, contact=True)
        return dict(name=name, cfg=cfg, env_dt=1e-3, state=state, nsteps=3, specs=[spec],
                    pstates=palm_states(state, 3, 2e-4), ext_f_grad=[np.linspace(-1e-2, 1e-2, 6)], seed=21)
    if name == "pour_liquid":
        state = np.load(H.GOLDEN / "pour_state_1k.npz")["state"]
        cfg = H.sim_cfg(len(state), n_grid=64, dt=1e-3, E=22.0, ptype=2, material_model=0, ground_friction=0.0)
        return dict(name=name, cfg=cfg, env_dt=1e-3, state=state, nsteps=3, specs=[], pstates=None, ext_f_grad=None, seed=22)
    if name == "cloud_elastic":
        N, n_grid = 1500, 32
        state = H.make_cloud(N, n_grid, seed=23, lo=(0.3, 0.05, 0.3), hi=(0.7, 0.4, 0.7))
        cfg = H.sim_cfg(N, n_grid=n_grid, dt=2e-4, ptype=1, material_model=0, ground_friction=0.0)
        return dict(name=name, cfg=cfg, env_dt=1e-3, state=state, nsteps=3, specs=[], pstates=None, ext_f_grad=None, seed=23)
    raise KeyError(name)


def seeds_for(sc):
    N = sc["cfg"].n_particles
    rng = np.random.default_rng(sc["seed"])
    n = sc["nsteps"]
    return {n: (rng.standard_normal((N, 3)), rng.standard_normal((N, 3)), 0.01 * rng.standard_normal((N, 3, 3)),
                0.01 * rng.standard_normal((N, 3, 3))), 1: (rng.standard_normal((N, 3)), None, None, None)}


def run_oracle(sc):
    P = H.oracle_params(sc["cfg"], sc["env_dt"])
    orc = H.OracleRollout(P, sc["state"], sc["specs"], sc["pstates"]).forward(sc["nsteps"])
    adj, pg, _ = orc.backward(seeds_for(sc), sc["ext_f_grad"])
    N = sc["cfg"].n_particles
    x, v, C, F = orc.frames[-1]
    out = dict(x=x.numpy(), v=v.numpy(), C=C.numpy().reshape(N, 9), F=F.numpy().reshape(N, 9),
               gx=adj[0][0].numpy(), gv=adj[0][1].numpy(), gC=adj[0][2].numpy().reshape(N, 9),
               gF=adj[0][3].numpy().reshape(N, 9))
    if sc["specs"]:
        out["ext_f"] = np.sum(np.array(orc.ext), axis=0)
        out["prim_grad"] = np.array(pg)[:-1]             # (nsteps, P, 13)
    return out
