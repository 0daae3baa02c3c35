"""Minimal yacs-compatible CfgNode (yacs is not installable here) + `load`, mirroring
/root/reference/softmac/config/utils.py:4-40 and default_config.py."""
from .cfgnode import CfgNode
from .utils import load, make_cls_config, get_cfg_defaults

__all__ = ["CfgNode", "load", "make_cls_config", "get_cfg_defaults"]
