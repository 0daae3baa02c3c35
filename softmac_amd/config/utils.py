"""load / make_cls_config / purge_cfg with the reference's semantics
(/root/reference/softmac/config/utils.py:4-40) and its default tree (default_config.py:4-95)."""
import math

from .cfgnode import CfgNode


def get_cfg_defaults():
    C = CfgNode()
    C.control_mode = "rigid"
    C.rigid_velocity_control = False
    C.env_dt = 2e-3
    S = C.SIMULATOR = CfgNode()
    S.dim = 3
    S.quality = 1
    S.yield_stress = 50.
    S.dtype = "float64"
    S.max_steps = 1024
    S.n_particles = 9000
    S.E = 5e3
    S.nu = 0.2
    S.ground_friction = 1.5
    S.gravity = (0, 0, 0)
    S.ptype = 0
    S.material_model = 1
    S.dt = 1e-4
    S.n_controllers = 0
    S.collision_type = 2
    C.PRIMITIVES = list()
    C.SHAPES = list()
    R = C.RIGID = CfgNode()
    R.gravity = (0., 0., 0.)
    R.init_state = ()
    R.enable_floor = True
    Rn = C.RENDERER = CfgNode()
    Rn.mode = "rgb_array"
    Rn.light_rot = (-math.pi / 4, 0)
    Rn.camera_pos = (0.5, 0.8, 2.8)
    Rn.camera_rot = (-0.2, 0)
    E = C.ENV = CfgNode()
    E.loss_type = ""
    L = E.loss = CfgNode()
    L.soft_contact = False
    L.weight = (10., 10., 1.)
    L.target_path = ''
    E.n_observed_particles = 200
    C.VARIANTS = list()
    return C


def make_cls_config(owner, cfg=None, **overrides):
    """a class's default config tree (`owner.default_config()`) with a file name or another tree, then keyword overrides, laid over it
    (/root/reference/softmac/config/utils.py:4-13; the keyword form is `merge_from_list`'s flat key / value sequence)"""
    tree = owner.default_config()
    if isinstance(cfg, str):
        tree.merge_from_file(cfg)
    elif cfg is not None:
        tree.merge_from_other_cfg(cfg)
    if overrides:
        flat = []
        for key, value in overrides.items():
            flat += [key, value]
        tree.merge_from_list(flat)
    return tree


def purge_cfg(cfg):
    """a node that names its variant in `TYPE` keeps, of its sub-trees, only the one of that name (utils.py:15-30); applied recursively"""
    keep = cfg.get("TYPE", None)
    for name in [k for k, v in cfg.items() if isinstance(v, CfgNode)]:
        if keep is None or name == keep:
            purge_cfg(cfg[name])
        else:
            del cfg[name]


def load(path=None, opts=None):
    """defaults <- yaml file <- flat option list, purged and frozen (utils.py:32-40)"""
    cfg = get_cfg_defaults()
    for source, merge in ((path, cfg.merge_from_file), (opts, cfg.merge_from_list)):
        if source is not None:
            merge(source)
    purge_cfg(cfg)
    cfg.freeze()
    return cfg
