"""A small re-creation of the subset of yacs.config.CfgNode the reference uses:
attribute access, clone, merge_from_file(.py exporting `cfg`), merge_from_other_cfg,
merge_from_list, freeze/defrost, get/items/del."""
from __future__ import annotations

import copy
import importlib.util
import pathlib


class CfgNode(dict):
    _FROZEN = "__frozen__"

    def __init__(self, init_dict=None):
        super().__init__()
        object.__setattr__(self, CfgNode._FROZEN, False)
        for k, v in (init_dict or {}).items():
            self[k] = CfgNode(v) if isinstance(v, dict) and not isinstance(v, CfgNode) else v

    def __getattr__(self, name):
        if name in self:
            return self[name]
        raise AttributeError(name)

    def __setattr__(self, name, value):
        if self.is_frozen():
            raise AttributeError(f"Attempted to set {name} to {value}, but CfgNode is immutable")
        self[name] = value

    def is_frozen(self):
        return object.__getattribute__(self, CfgNode._FROZEN)

    def _set_frozen(self, flag):
        object.__setattr__(self, CfgNode._FROZEN, flag)
        for v in self.values():
            if isinstance(v, CfgNode):
                v._set_frozen(flag)
            elif isinstance(v, (list, tuple)):
                for i in v:
                    if isinstance(i, CfgNode):
                        i._set_frozen(flag)

    def freeze(self):
        self._set_frozen(True)

    def defrost(self):
        self._set_frozen(False)

    def clone(self):
        return copy.deepcopy(self)

    def __deepcopy__(self, memo):
        out = CfgNode()
        for k, v in self.items():
            dict.__setitem__(out, k, copy.deepcopy(v, memo))
        object.__setattr__(out, CfgNode._FROZEN, self.is_frozen())
        return out

    def merge_from_other_cfg(self, other):
        for k, v in other.items():
            if isinstance(v, CfgNode) and isinstance(self.get(k), CfgNode):
                self[k].merge_from_other_cfg(v)
            else:
                dict.__setitem__(self, k, copy.deepcopy(v))

    def merge_from_file(self, path):
        path = pathlib.Path(path)
        if path.suffix != ".py":
            raise ValueError("only python config files exporting `cfg` are supported")
        spec = importlib.util.spec_from_file_location("_softmac_cfg_" + path.stem, path)
        mod = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(mod)
        self.merge_from_other_cfg(_to_cfgnode(mod.cfg))

    def merge_from_list(self, lst):
        assert len(lst) % 2 == 0
        for key, value in zip(lst[0::2], lst[1::2]):
            node = self
            parts = key.split(".")
            for p in parts[:-1]:
                node = node[p]
            dict.__setitem__(node, parts[-1], value)

    def __repr__(self):
        return "CfgNode(" + dict.__repr__(self) + ")"


def _to_cfgnode(obj):
    """Accept our CfgNode, a plain dict, or any mapping-like yacs node."""
    if isinstance(obj, CfgNode):
        return obj
    out = CfgNode()
    for k, v in dict(obj).items():
        if hasattr(v, "items") and not isinstance(v, CfgNode):
            v = _to_cfgnode(v)
        elif isinstance(v, (list, tuple)):
            v = type(v)(_to_cfgnode(i) if hasattr(i, "items") and not isinstance(i, CfgNode) else i for i in v)
        dict.__setitem__(out, k, v)
    return out
