"""ctypes binding of include/softmac_hip.h (libsoftmac_hip.so).

There is no CPU fallback: if the library is missing or no GPU is visible, creating a simulator
raises.  Loading the library itself works without a GPU (the "does it export every symbol" test).
"""
from __future__ import annotations

import ctypes as C
import pathlib

import numpy as np

_LIB = None
LIB_PATH = pathlib.Path(__file__).resolve().parent / "lib" / "libsoftmac_hip.so"

ABI_VERSION = 2
MAX_PRIMS = 4

c_double_p = C.POINTER(C.c_double)
c_int32_p = C.POINTER(C.c_int32)
c_int8_p = C.POINTER(C.c_int8)


class SmacConfig(C.Structure):
    _fields_ = [
        ("abi_version", C.c_int32), ("precision", C.c_int32), ("device", C.c_int32),
        ("n_particles", C.c_int32), ("n_grid", C.c_int32), ("max_frames", C.c_int32),
        ("grad_enabled", C.c_int32), ("substeps", C.c_int32), ("ptype", C.c_int32),
        ("material_model", C.c_int32), ("collision_type", C.c_int32), ("n_control", C.c_int32),
        ("n_primitives", C.c_int32), ("rigid_velocity_control", C.c_int32),
        ("sort_interval", C.c_int32), ("flags", C.c_int32), ("adjoint_frames", C.c_int32), ("reserved0", C.c_int32),
        ("dt", C.c_double), ("mu", C.c_double), ("lam", C.c_double),
        ("p_vol", C.c_double), ("p_mass", C.c_double), ("gravity", C.c_double * 3),
        ("ground_friction", C.c_double), ("yield_stress", C.c_double),
    ]


H = C.c_void_p
# name -> (restype, argtypes); mirrors include/softmac_hip.h one to one
SIGNATURES = {
    "smac_last_error": (C.c_char_p, [H]),
    "smac_abi_version": (C.c_int, []),
    "smac_device_count": (C.c_int, []),
    "smac_mesh_to_sdf": (C.c_int, [C.c_int, c_double_p, C.c_int, c_int32_p, C.c_int, c_double_p, c_int32_p, C.c_double,
                                   c_double_p, c_double_p]),
    "smac_create": (C.c_int, [C.POINTER(SmacConfig), C.POINTER(H)]),
    "smac_destroy": (C.c_int, [H]),
    "smac_sync": (C.c_int, [H]),
    "smac_reset": (C.c_int, [H, c_double_p, C.c_int]),
    "smac_set_frame": (C.c_int, [H, C.c_int, c_double_p, c_double_p, c_double_p, c_double_p]),
    "smac_get_frame": (C.c_int, [H, C.c_int, c_double_p, c_double_p, c_double_p, c_double_p]),
    "smac_get_state": (C.c_int, [H, C.c_int, c_double_p]),
    "smac_copy_frame": (C.c_int, [H, C.c_int, C.c_int]),
    "smac_get_grad": (C.c_int, [H, C.c_int, c_double_p, c_double_p, c_double_p, c_double_p]),
    "smac_add_grad": (C.c_int, [H, C.c_int, c_double_p, c_double_p, c_double_p, c_double_p]),
    "smac_add_grad_device": (C.c_int, [H, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "smac_clear_grads": (C.c_int, [H]),
    "smac_carry_grad": (C.c_int, [H, C.c_int, C.c_int]),
    "smac_set_control_idx": (C.c_int, [H, c_int32_p]),
    "smac_set_material_ids": (C.c_int, [H, c_int32_p]),
    "smac_set_action": (C.c_int, [H, c_double_p]),
    "smac_set_segment": (C.c_int, [H, C.c_int, C.c_int]),
    "smac_compute_grid_m": (C.c_int, [H, C.c_int, c_double_p]),
    "smac_substep": (C.c_int, [H, C.c_int, c_double_p]),
    "smac_substep_grad": (C.c_int, [H, C.c_int, c_double_p, c_double_p, c_double_p]),
    "smac_substeps": (C.c_int, [H, C.c_int, C.c_int]),
    "smac_substeps_grad": (C.c_int, [H, C.c_int, C.c_int, c_double_p]),
    "smac_substeps_action": (C.c_int, [H, C.c_int, C.c_int, c_double_p]),
    "smac_substeps_grad_action": (C.c_int, [H, C.c_int, C.c_int, c_double_p, c_double_p, c_double_p]),
    "smac_prim_upload_sdf": (C.c_int, [H, C.c_int, c_double_p, c_double_p, c_int32_p, c_double_p, c_double_p, C.c_double]),
    "smac_prim_set_params": (C.c_int, [H, C.c_int, C.c_double, C.c_double, C.c_int]),
    "smac_prim_set_state": (C.c_int, [H, C.c_int, C.c_int, C.c_int, c_double_p]),
    "smac_prim_get_state": (C.c_int, [H, C.c_int, C.c_int, c_double_p]),
    "smac_prim_get_state_grad": (C.c_int, [H, C.c_int, C.c_int, C.c_int, c_double_p]),
    "smac_prim_get_action_grads": (C.c_int, [H, C.c_int, C.c_int, C.c_int, C.c_int, c_double_p]),
    "smac_prim_set_states": (C.c_int, [H, C.c_int, C.c_int, C.c_int, c_double_p]),
    "smac_prim_get_state_grads": (C.c_int, [H, C.c_int, C.c_int, C.c_int, c_double_p]),
    "smac_prim_add_state_grad": (C.c_int, [H, C.c_int, C.c_int, c_double_p]),
    "smac_prim_forward_kinematics": (C.c_int, [H, C.c_int, C.c_int]),
    "smac_prim_forward_kinematics_grad": (C.c_int, [H, C.c_int, C.c_int]),
    "smac_prim_get_ext_f": (C.c_int, [H, C.c_int, c_double_p]),
    "smac_prim_clear_ext_f": (C.c_int, [H, C.c_int]),
    "smac_prim_set_action": (C.c_int, [H, C.c_int, C.c_int, C.c_int, c_double_p]),
    "smac_prim_get_action_grad": (C.c_int, [H, C.c_int, C.c_int, C.c_int, c_double_p]),
    "smac_prim_reset": (C.c_int, [H, C.c_int]),
    "smac_timer_start": (C.c_int, [H]),
    "smac_timer_stop": (C.c_int, [H, c_double_p]),
    "smac_profile_enable": (C.c_int, [H, C.c_int]),
    "smac_profile_reset": (C.c_int, [H]),
    "smac_profile_count": (C.c_int, [H]),
    "smac_profile_get": (C.c_int, [H, C.c_int, C.c_char_p, C.c_int, c_double_p, C.POINTER(C.c_int64)]),
    "smac_count_active_cells": (C.c_int, [H, C.c_int, C.POINTER(C.c_int64)]),
    "smac_contact_counts": (C.c_int, [H, c_int32_p, c_int32_p]),
    "smac_set_param": (C.c_int, [H, C.c_char_p, C.c_double]),
    "smac_get_param": (C.c_int, [H, C.c_char_p, c_double_p]),
    "smac_cloth_create": (C.c_int, [H, C.c_int, C.c_int, c_int32_p, C.c_int, c_int32_p, c_int8_p, C.c_double, C.c_double, C.c_double, C.c_int,
                                    C.c_double]),
    "smac_cloth_set_state": (C.c_int, [H, C.c_int, C.c_int, c_double_p, c_double_p]),
    "smac_cloth_get_state": (C.c_int, [H, C.c_int, c_double_p, c_double_p]),
    "smac_cloth_get_state_grad": (C.c_int, [H, C.c_int, c_double_p, c_double_p]),
    "smac_cloth_get_ext_f": (C.c_int, [H, c_double_p]),
    "smac_cloth_clear_ext_f": (C.c_int, [H]),
    "smac_cloth_set_ext_f_grad": (C.c_int, [H, c_double_p]),
    "smac_cloth_contact_pair": (C.c_int, [H, C.c_int]),
    "smac_cloth_backup_contact_pair": (C.c_int, [H, C.c_int]),
    "smac_cloth_trace_penetration": (C.c_int, [H, C.c_int, C.c_int]),
    "smac_cloth_get_contact": (C.c_int, [H, C.c_int, c_int32_p, c_int8_p]),
    "smac_cloth_set_contact": (C.c_int, [H, C.c_int, c_int32_p, c_int8_p]),
    "smac_cloth_check_penetration": (C.c_int, [H, C.c_int, c_int32_p, c_int32_p]),
    "smac_loss_set_target": (C.c_int, [H, c_double_p, C.c_int]),
    "smac_loss_chamfer": (C.c_int, [H, C.c_int, C.c_double, C.c_int, c_double_p]),
    "smac_loss_min_dist": (C.c_int, [H, C.c_int, C.c_int, C.c_int, c_double_p, C.c_double, C.c_double, C.c_int, c_double_p]),
    "smac_grid_device_ptr": (C.c_int, [H, C.c_char_p, C.POINTER(C.c_void_p), C.POINTER(C.c_int64), C.POINTER(C.c_int32)]),
    "smac_stream_handle": (C.c_int, [H, C.POINTER(C.c_void_p)]),
    "smac_set_stream": (C.c_int, [H, C.c_void_p]),
    "smac_substep_phase": (C.c_int, [H, C.c_int, C.c_int]),
    "smac_substep_grad_phase": (C.c_int, [H, C.c_int, c_double_p, C.c_int]),
    "smac_halo_pack": (C.c_int, [H, C.c_char_p, C.c_int, C.c_int, C.c_void_p, C.c_int]),
    "smac_halo_unpack_add": (C.c_int, [H, C.c_char_p, C.c_int, C.c_int, C.c_void_p]),
    "smac_comm_unique_id": (C.c_int, [C.c_char_p]),
    "smac_comm_init": (C.c_int, [H, C.c_char_p, C.c_int, C.c_int]),
    "smac_comm_slab": (C.c_int, [H, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int]),
    "smac_substeps_slab": (C.c_int, [H, C.c_int, C.c_int]),
    "smac_substeps_slab_grad": (C.c_int, [H, C.c_int, C.c_int, c_double_p]),
    "smac_comm_allreduce_ext_f": (C.c_int, [H, c_double_p, C.c_int]),
    "smac_comm_allreduce_prim_grad": (C.c_int, [H, C.c_int, C.c_int]),
    "smac_comm_destroy": (C.c_int, [H]),
    "smac_comm_abort": (C.c_int, [H]),
    "smac_migrate": (C.c_int, [H, C.c_int, C.c_int, C.c_int, c_int32_p]),
    "smac_migrate_grad": (C.c_int, [H]),
    "smac_set_ids": (C.c_int, [H, C.POINTER(C.c_int64)]),
    "smac_get_ids": (C.c_int, [H, C.POINTER(C.c_int64)]),
}


def load_library(path=None):
    """dlopen libsoftmac_hip.so and declare every prototype.  Raises if the library is absent."""
    global _LIB
    if _LIB is not None and path is None:
        return _LIB
    import os
    if path is None and os.environ.get("SMAC_LIB"):          # experiments: an alternative build of the same ABI
        path = os.environ["SMAC_LIB"]
    p = pathlib.Path(path) if path else LIB_PATH
    if not p.exists():
        raise RuntimeError(
            f"{p} not found: build it with `python -m softmac_amd.build` (hipcc, gfx950). "
            "softmac_amd has no CPU fallback.")
    lib = C.CDLL(str(p))
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)            # AttributeError if a declared symbol is not exported
        fn.restype = res
        fn.argtypes = args
    if lib.smac_abi_version() != ABI_VERSION:
        raise RuntimeError("libsoftmac_hip.so ABI version mismatch")
    if path is None:
        _LIB = lib
    return lib


def as_f64(a, shape=None):
    a = np.ascontiguousarray(a, dtype=np.float64)
    if shape is not None and tuple(a.shape) != tuple(shape):
        raise ValueError(f"expected shape {tuple(shape)}, got {a.shape}")
    return a


def dptr(a):
    return None if a is None else a.ctypes.data_as(c_double_p)


class SmacError(RuntimeError):
    pass


class Handle:
    """Owns one smac_handle; every method raises SmacError with smac_last_error on failure."""

    def __init__(self, cfg: SmacConfig):
        self.lib = load_library()
        self.h = H()
        rc = self.lib.smac_create(C.byref(cfg), C.byref(self.h))
        if rc != 0:
            msg = self.lib.smac_last_error(None)
            self.h = None
            raise SmacError(f"smac_create failed ({rc}): {msg.decode() if msg else ''}")
        self.cfg = cfg

    def call(self, name, *args):
        if self.h is None:
            raise SmacError("handle destroyed")
        rc = getattr(self.lib, name)(self.h, *args)
        if rc != 0:
            msg = self.lib.smac_last_error(self.h)
            raise SmacError(f"{name} failed ({rc}): {msg.decode() if msg else ''}")

    def close(self):
        if getattr(self, "h", None):
            self.lib.smac_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
