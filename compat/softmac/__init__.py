"""`softmac` = `softmac_amd` under the reference's package name.  Every `softmac.<x>` import resolves to the `softmac_amd.<x>` module object itself
(a meta-path finder; no second copy of any module, so registries and the loaded library are shared)."""
import importlib
import importlib.abc
import importlib.machinery
import sys

import softmac_amd

_SRC, _DST = "softmac_amd", "softmac"


class _Alias(importlib.abc.MetaPathFinder, importlib.abc.Loader):
    def find_spec(self, fullname, path=None, target=None):
        if fullname != _DST and not fullname.startswith(_DST + "."):
            return None
        real = _SRC + fullname[len(_DST):]
        try:
            mod = importlib.import_module(real)
        except ModuleNotFoundError as e:
            if e.name == real:
                return None
            raise
        spec = importlib.machinery.ModuleSpec(fullname, self, is_package=hasattr(mod, "__path__"))
        spec._aliased = mod
        return spec

    def create_module(self, spec):
        return spec._aliased

    def exec_module(self, module):
        pass


sys.meta_path.insert(0, _Alias())
__path__ = softmac_amd.__path__
for _k, _v in vars(softmac_amd).items():
    if not _k.startswith("__"):
        globals()[_k] = _v
