"""Demo-side helpers with the reference's names (/root/reference/softmac/utils.py:11-119): `prepare` (log directory + config), `render`,
`make_gif_from_numpy`, `adjust_action_with_ext_force`.  SURVEY 2 #11 marks them API-SURFACE ONLY; they are here so that a demo's
`from softmac.utils import ...` resolves (compat/softmac aliases this package).  Rendering is out of scope (pyrender / imageio are absent):
`render` returns whatever `env.render` returns (None frames from the null renderer) and `make_gif_from_numpy` stores the frames as an .npy stack
when imageio is missing."""
from __future__ import annotations

import json
import shutil
from pathlib import Path

import numpy as np
import torch


def make_gif_from_numpy(images, logdir, name=None):
    out = Path(logdir) / ((name or "movie") + ".gif")
    frames = [im for im in images if im is not None]
    try:
        import imageio.v2 as imageio
    except ImportError:
        if frames:
            np.save(out.with_suffix(".npy"), np.stack(frames))
        return None
    with imageio.get_writer(out, mode="I", loop=0) as writer:
        for im in frames:
            writer.append_data(im)
    return out


def render(env, action=None, n_steps=100, interval=10):
    """one image per `interval` env steps: of the recorded trajectory (action None) or of a replay of `action` in copy mode (:31-50)"""
    images = []
    replay = action is not None
    if replay:
        env.initialize()
        was_copy = env._is_copy
        env.set_copy(True)
    for i in range(n_steps):
        if replay:
            env.step(action[i])
        if i % interval == 0:
            images.append(env.render(0 if replay else i * env.substeps))
    if replay:
        env.set_copy(was_copy)
    return images


def prepare(args):
    """logs/<exp_name>/ with the config copied in, args.json, empty figs/ and actions/ (:53-71); returns (log_dir, cfg)"""
    from .config import load
    log_dir = Path("logs") / args.exp_name
    log_dir.mkdir(parents=True, exist_ok=True)
    cfg = load(args.config)
    shutil.copyfile(args.config, log_dir / "config.py")
    (log_dir / "args.json").write_text(json.dumps(vars(args), indent=4))
    for sub in ("figs", "actions"):
        d = log_dir / sub
        if d.exists():
            shutil.rmtree(d)
        d.mkdir()
    return log_dir, cfg


def adjust_action_with_ext_force(env, actions):
    """Actions optimised without external force, corrected by the wrench the particles put on each primitive (:76-119): per env step the MPM
    substeps run, every force-enabled primitive's mean wrench (+ its weight, force control only) is subtracted from that step's action, then the
    rigid simulator steps.  Velocity-controlled bodies have no mass model here (no Jade): the gravity term is taken only when the rigid simulator
    provides `skeletons`."""
    assert env.control_mode == "rigid" and not env._is_copy
    rs = env.rigid_simulator
    out = []
    for t in range(actions.shape[0]):
        start = env.simulator.cur
        env.simulator.cur = start + env.substeps
        env.simulator.run_substeps(start, env.substeps)
        for i in range(rs.n_primitive):
            prim = env.primitives[i]
            if not getattr(prim, "enable_external_force", False):
                continue
            wrench = torch.as_tensor(prim.ext_f.to_numpy(), dtype=actions.dtype) / env.substeps
            force, torque = wrench[:3].clone(), wrench[3:]
            if hasattr(rs, "skeletons"):
                force += rs.skeletons[i].getMass() * torch.as_tensor(rs.gravity, dtype=actions.dtype)
            actions[t, i * 6: i * 6 + 3] -= torque
            actions[t, i * 6 + 3: i * 6 + 6] -= force
        rs.step(start // env.substeps, actions[t])
        out.append(actions[t])
    return torch.vstack(out)
