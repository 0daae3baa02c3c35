"""Import-level drop-in (INTEGRATION route A, compat/): with compat/ on the path a script written against the reference's package names resolves to this
build - `softmac.*` ARE the softmac_amd modules, `yacs.config.CfgNode` is the build's CfgNode, `taichi.ad` offers the two calls the demos make.  CPU part
here (imports, aliasing, a reference-style config file); the GPU part runs the demo's epoch body through those names."""
import subprocess
import sys
import textwrap

import numpy as np
import pytest

import helpers as H

COMPAT = str(H.ROOT / "compat")


def _run(code, **kw):
    env_path = f"{H.ROOT}:{COMPAT}"
    return subprocess.run([sys.executable, "-c", textwrap.dedent(code)], capture_output=True, text=True, timeout=300, env={**__import__('os').environ, "PYTHONPATH": env_path}, **kw)


def test_aliases_are_the_same_module_objects():
    out = _run("""
        import taichi as ti
        from yacs.config import CfgNode as CN
        import softmac, softmac_amd
        import softmac.engine.taichi_env as a
        import softmac_amd.engine.taichi_env as b
        from softmac.utils import make_gif_from_numpy, render, prepare, adjust_action_with_ext_force
        from softmac.engine.losses import PourLoss
        import softmac_amd.config as cfgmod
        assert a is b and a.TaichiEnv is b.TaichiEnv
        assert CN is cfgmod.CfgNode
        assert callable(ti.ad.clear_all_gradients) and hasattr(ti.ad, "Tape")
        ti.ad.clear_all_gradients()                       # no simulator alive: nothing to do, no error
        print("ok")
    """)
    assert out.returncode == 0 and out.stdout.strip().endswith("ok"), out.stderr


def test_reference_style_config_file_loads(tmp_path):
    (tmp_path / "demo_cfg.py").write_text(textwrap.dedent("""
        from yacs.config import CfgNode as CN
        _C = CN()
        cfg = _C
        _C.control_mode = "rigid"
        _C.rigid_velocity_control = True
        _C.env_dt = 1e-3
        _C.SIMULATOR = CN()
        _C.SIMULATOR.dt = 1e-3
        _C.SIMULATOR.ptype = 2
        _C.SHAPES = [{"shape": "box", "init_pos": "(0.5, 0.5, 0.5)", "width": "(0.1, 0.1, 0.1)", "n_particles": 100}]
        _C.ENV = CN()
        _C.ENV.loss_type = "PourLoss"
    """))
    out = _run(f"""
        from softmac.config import load
        cfg = load({str(tmp_path / 'demo_cfg.py')!r})
        assert cfg.rigid_velocity_control is True and cfg.SIMULATOR.ptype == 2 and cfg.SIMULATOR.E == 5e3 and cfg.ENV.loss_type == "PourLoss"
        print("ok")
    """)
    assert out.returncode == 0 and out.stdout.strip().endswith("ok"), out.stderr


@pytest.mark.gpu
def test_demo_epoch_body_through_the_reference_names(tmp_path):
    """the body of demo_pour_vel.py's epoch loop (:78-106) - clear_all_gradients, reset, steps, Tape around compute_loss, backward - written against
    `taichi`, `softmac.engine.taichi_env` and a yacs-style config; the action gradient must equal the one the build's own names give"""
    cfg_file = tmp_path / "cfg.py"
    tgt = tmp_path / "target.npy"
    rng = np.random.default_rng(0)
    np.save(tgt, rng.uniform(0.4, 0.6, (400, 3)))
    palm = H.load_palm()
    np.savez(tmp_path / "palm.npz", **{k: np.asarray(v) for k, v in palm.items()})
    cfg_file.write_text(textwrap.dedent(f"""
        from yacs.config import CfgNode as CN
        _C = CN()
        cfg = _C
        _C.control_mode = "rigid"
        _C.rigid_velocity_control = True
        _C.env_dt = 1e-3
        _C.SIMULATOR = CN()
        _C.SIMULATOR.dt = 2.5e-4
        _C.SIMULATOR.ptype = 2
        _C.SIMULATOR.E = 22.0
        _C.SIMULATOR.material_model = 0
        _C.SIMULATOR.gravity = (0.0, -9.8, 0.0)
        _C.SIMULATOR.max_steps = 64
        _C.SIMULATOR.collision_type = 2
        _C.SHAPES = [{{"shape": "box", "init_pos": "(0.5, 0.42, 0.5)", "width": "(0.12, 0.06, 0.12)", "n_particles": 600}}]
        _C.RIGID = CN()
        _C.RIGID.init_state = (0.0, 0.0, 0.0, 0.5, 0.235, 0.5, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0)
        _C.ENV = CN()
        _C.ENV.loss_type = "PourLoss"
        _C.ENV.loss = CN()
        _C.ENV.loss.weight = (1.0, 1.0, 0.1)
        _C.ENV.loss.target_path = {str(tgt)!r}
    """))
    code = f"""
        import sys
        sys.path.insert(0, {str(H.ROOT / 'tests')!r})
        import numpy as np, torch
        import taichi as ti
        from softmac.engine.taichi_env import TaichiEnv
        from softmac.config import load
        from softmac.engine.primitive import Mesh, Primitives
        from softmac_amd.config import CfgNode
        palm = dict(np.load({str(tmp_path / 'palm.npz')!r}))
        palm["dx"] = float(palm["dx"])
        def make_env():
            cfg = load({str(cfg_file)!r})
            pc = CfgNode(); pc.friction = 0.9; pc.enable_external_force = True; pc.urdf_path = ""
            mesh = Mesh(sdf=palm, cfg=pc, max_timesteps=cfg.SIMULATOR.max_steps, rigid_velocity_control=True)
            return TaichiEnv(cfg, primitives=Primitives(primitives=[mesh]))
        actions = np.zeros((3, 6)); actions[:, 4] = 0.3; actions[:, 2] = 0.5
        # --- the demo's epoch body, reference names only
        env = make_env()
        ti.ad.clear_all_gradients()
        env.reset()
        for i in range(3):
            env.step(actions[i])
        with ti.ad.Tape(loss=env.loss.loss):
            for f in range(0, env.simulator.cur + 1, 4):
                info = env.compute_loss(f)
        g_demo = env.backward().numpy()
        loss_demo = float(env.loss.loss.to_numpy())
        # --- the same through the build's own surface
        env2 = make_env()
        env2.simulator.clear_grads()
        env2.reset()
        for i in range(3):
            env2.step(actions[i])
        env2.loss.clear()
        with env2.loss.tape():
            for f in range(0, env2.simulator.cur + 1, 4):
                env2.compute_loss(f)
        g_own = env2.backward().numpy()
        assert np.isfinite(g_demo).all() and np.abs(g_demo).max() > 0
        assert np.abs(g_demo - g_own).max() <= 1e-4 * np.abs(g_own).max(), (g_demo, g_own)
        assert abs(loss_demo - float(env2.loss.loss)) <= 1e-6 * abs(loss_demo) and loss_demo > 0
        print("ok", loss_demo)
    """
    out = _run(code)
    assert out.returncode == 0 and "ok" in out.stdout, out.stdout + out.stderr
