"""yacs-compatible CfgNode subset (reference: softmac/config/utils.py, default_config.py)."""
import textwrap

import pytest

from softmac_amd.config import CfgNode, load, make_cls_config


def test_defaults_match_reference_tree():
    cfg = load()
    assert cfg.SIMULATOR.dtype == "float64" and cfg.SIMULATOR.collision_type == 2 and cfg.SIMULATOR.material_model == 1
    assert cfg.env_dt == 2e-3 and cfg.control_mode == "rigid" and cfg.rigid_velocity_control is False
    with pytest.raises(AttributeError):
        cfg.env_dt = 1.0                                     # frozen


def test_merge_from_python_file(tmp_path):
    f = tmp_path / "demo_cfg.py"
    f.write_text(textwrap.dedent("""
        from softmac_amd.config import CfgNode as CN
        _C = CN()
        cfg = _C
        _C.env_dt = 1e-3
        _C.SIMULATOR = CN()
        _C.SIMULATOR.E = 22.
        _C.SIMULATOR.ptype = 2
        G = CN(); G.friction = 0.1; G.urdf_path = "assets/glass/glass.urdf"
        _C.PRIMITIVES = [G]
    """))
    cfg = load(str(f))
    assert cfg.env_dt == 1e-3 and cfg.SIMULATOR.E == 22. and cfg.SIMULATOR.nu == 0.2    # merged over defaults
    assert cfg.PRIMITIVES[0].friction == 0.1
    c2 = cfg.clone(); c2.defrost(); c2.SIMULATOR.n_particles = 5
    assert cfg.SIMULATOR.n_particles == 9000


def test_make_cls_config_and_merge_from_list():
    class Thing:
        @classmethod
        def default_config(cls):
            c = CfgNode(); c.friction = 0.9; c.urdf_path = ""
            return c
    over = CfgNode(); over.friction = 0.2
    c = make_cls_config(Thing(), over, urdf_path="a.urdf")
    assert c.friction == 0.2 and c.urdf_path == "a.urdf"
