"""Random sequences of the entry points - single substeps, batches, backward sweeps cut into batches and single calls, reads in between, forward again from an
earlier frame - on two handles of one scene: one with the round-3 launch structure (fused backward step, checkpoint save inside k_g2p, restore-ahead on the
second grid buffer set, alternating hit counters), one with all of it switched off (SMAC_FUSED_PG / SMAC_SAVE_IN_G2P / SMAC_RESTORE_AHEAD = 0: one kernel per
step, as round 1 had it).  Same arithmetic either way, so every read must agree; what this walks is the host-side state those switches added (which buffer set
is current, which counter is empty, whether a sweep is in flight) under orders of calls the fixed tests do not use."""
import os

import numpy as np
import pytest

import helpers as H
from softmac_amd import scenes

pytestmark = pytest.mark.gpu

OFF = {"SMAC_FUSED_PG": "0", "SMAC_SAVE_IN_G2P": "0", "SMAC_RESTORE_AHEAD": "0"}
T = 14                                   # frames a case may reach


def _engine(env, sort_interval):
    old = {k: os.environ.get(k) for k in OFF}
    os.environ.update(env)
    try:
        cfg, env_dt, state, specs, s13 = scenes.s_grip(1 << 14, 64, max_steps=T + 4, precision="float32", dt=1e-4)      # (the scene the bounds below were measured on: dt as at 128^3)
        cfg.sort_interval = sort_interval
        pst = [[np.concatenate([s[:3] + s[7:10] * cfg.dt * f, s[3:]]) for s in s13] for f in range(T + 4)]
        sim, prm = H.build_engine(cfg, env_dt, specs, pst)
    finally:
        for k, v in old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v
    sim.reset(state)
    return sim, prm, cfg.n_particles


def _agree(a, b, what):
    a, b = np.asarray(a, dtype=np.float64).reshape(len(a), -1), np.asarray(b, dtype=np.float64).reshape(len(b), -1)
    scale = np.abs(b).max()
    if scale == 0:
        assert np.abs(a).max() == 0, what
        return
    d = np.abs(a - b).max(axis=1) / scale
    # two handles = two particle orders = f32 roundings apart; a particle on the other side of a branch of the reference's function (yield clip, contact
    # band) moves its own adjoint by O(1e-4) and its stencil neighbours' by less (tests/test_gpu_fused_backward.py): 99th percentile tight, maximum loose
    assert np.quantile(d, 0.99) < 1e-5 and d.max() < 1e-1, (what, float(np.quantile(d, 0.99)), float(d.max()))


@pytest.mark.parametrize("case", range(6))
def test_random_call_sequences_agree_with_the_plain_launch_structure(case):
    rng = np.random.default_rng(500 + case)
    sort_interval = int(rng.choice([3, 5, 1000]))
    fast, pf, N = _engine({}, sort_interval)
    plain, pp, _ = _engine(OFF, sort_interval)
    both = (fast, plain)
    cur, log = int(rng.integers(3, 8)), []
    for s in both:
        s.run_substeps(0, cur)
    for step in range(int(rng.integers(12, 18))):
        op = rng.choice(["batch", "single", "sweep", "rewind", "read"], p=[0.25, 0.15, 0.35, 0.1, 0.15])
        if op == "batch" and cur < T:
            k = int(rng.integers(1, min(6, T - cur) + 1))
            for s in both:
                s.run_substeps(cur, k)
            cur += k
            assert abs(fast.contact_counts()[0] - plain.contact_counts()[0]) <= 2 and plain.contact_counts()[0] > 0, (log, op)   # (a particle on the band's edge may differ)
        elif op == "single" and cur < T:
            for s in both:
                s.substep(cur)
            cur += 1
            assert abs(fast.contact_counts()[0] - plain.contact_counts()[0]) <= 2 and plain.contact_counts()[0] > 0, (log, op)   # (a particle on the band's edge may differ)
        elif op == "sweep" and cur >= 1:
            top = int(rng.integers(1, cur + 1))
            bottom = int(rng.integers(0, top))
            seeds = {top: rng.standard_normal((N, 3))}
            if top - bottom > 2 and rng.integers(0, 2):
                seeds[int(rng.integers(bottom + 1, top))] = rng.standard_normal((N, 3))
            cuts = [top]
            while cuts[-1] > bottom:                          # the sweep top-1 ... bottom in pieces: batches and single calls
                cuts.append(int(rng.integers(bottom, cuts[-1])))
            singles = [bool(rng.integers(0, 3) == 0) for _ in cuts[1:]]
            peek = bool(rng.integers(0, 2))
            for s in both:
                s.clear_grads()
                for f, g in seeds.items():
                    s.add_grad(f, gx=g)
                for hi, lo, single in zip(cuts, cuts[1:], singles):
                    if single:
                        for f in range(hi - 1, lo - 1, -1):
                            s.substep_grad(f)
                    else:
                        s.run_substeps_grad(lo, hi - lo)
                    if peek:
                        s.get_x(lo)                            # a read between the pieces of a sweep
            ga, gb = fast.get_grad_full(bottom), plain.get_grad_full(bottom)
            for name, x, y in zip(("gx", "gv", "gF", "gC"), ga, gb):
                _agree(x, y, (log, op, name, top, bottom, cuts, singles))
            _agree(np.array([m.get_all_states_grad(bottom) for m in pf]), np.array([m.get_all_states_grad(bottom) for m in pp]), (log, op, "primitives"))
        elif op == "rewind" and cur >= 2:
            cur = int(rng.integers(max(cur - 6, 0), cur))     # the next forward call recomputes from an earlier frame
        else:
            _agree(fast.get_state(cur), plain.get_state(cur), (log, op, "state", cur))
        log.append((op if op != "sweep" or cur >= 1 else "read", cur))
    _agree(fast.get_state(cur), plain.get_state(cur), (log, "final state", cur))
    print(f"\n[api sequence {case}] re-sort every {sort_interval}: " + " ".join(f"{o}->{c}" for o, c in log))
    assert fast.get_param("hit_overflows") == 0 and fast.get_param("drift_repairs") == plain.get_param("drift_repairs")


def test_a_seed_from_device_memory_equals_the_host_seed():
    """smac_add_grad_device (round 4): the loss kernels of the reference add to x.grad on the device (losses/loss_pour.py:130-140); a seed handed over as
    device pointers must be the seed handed over as host arrays - before the frame exists (identity order) and after (the frame's binning), all four fields."""
    import torch
    rng = np.random.default_rng(3)
    sim, prm, n = _engine({}, 4)
    g = {k: rng.standard_normal((n,) + shp) for k, shp in (("gx", (3,)), ("gv", (3,)), ("gF", (3, 3)), ("gC", (3, 3)))}
    gd = {k: torch.from_numpy(v.reshape(n, -1).copy()).cuda() for k, v in g.items()}
    torch.cuda.synchronize()
    for when in ("before the frame exists", "after the forward pass"):
        got = []
        for dev in (False, True):
            sim.clear_grads()
            if when.startswith("after"):
                sim.run_substeps(0, 6)                      # frame 6 lies under the second binning (interval 4)
            (sim.add_grad_device if dev else sim.add_grad)(6, **(gd if dev else g))
            if when.startswith("before"):
                sim.run_substeps(0, 6)
            sim.run_substeps_grad(0, 6)
            got.append([np.asarray(a) for a in sim.get_grad_full(0)])
        for a, b in zip(*got):
            _agree(a, b, when)
    with pytest.raises(Exception):
        sim.add_grad_device(6, gx=g["gx"].ctypes.data)      # a host pointer is refused, not dereferenced
