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


def make_cls_config(self, cfg=None, **kwargs):
    _cfg = self.default_config()
    if cfg is not None:
        if isinstance(cfg, str):
            _cfg.merge_from_file(cfg)
        else:
            _cfg.merge_from_other_cfg(cfg)
    if len(kwargs) > 0:
        _cfg.merge_from_list(sum(list(kwargs.items()), ()))
    return _cfg


def purge_cfg(cfg):
    target_key = cfg.get('TYPE', None)
    removed = []
    for k, v in cfg.items():
        if isinstance(v, CfgNode):
            if target_key is not None and (k != target_key):
                removed.append(k)
            else:
                purge_cfg(v)
    for k in removed:
        del cfg[k]


def load(path=None, opts=None):
    cfg = get_cfg_defaults()
    if path is not None:
        cfg.merge_from_file(path)
    if opts is not None:
        cfg.merge_from_list(opts)
    purge_cfg(cfg)
    cfg.freeze()
    return cfg
