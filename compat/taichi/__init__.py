"""The two things the reference's demos ask of `taichi` itself (compat/README.md): `ti.ad.clear_all_gradients()` and `ti.ad.Tape(loss=...)`.
No kernel language, no runtime: the engine is libsoftmac_hip."""
from . import ad  # noqa: F401

__version__ = "0.0-softmac-amd-shim"
