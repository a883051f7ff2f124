"""ctypes binding of include/mgn_hip.h -- one prototype per declared symbol, nothing else.

The product path fails loudly when the HIP extension is missing: `load()` raises, there is no
fallback implementation anywhere in this package.
"""
from __future__ import annotations

import ctypes as C
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "lib", "libmgn_hip.so")

MGN_DEVICE_NONE = -2
MGN_OK, MGN_E_ARG, MGN_E_HIP, MGN_E_STATE, MGN_E_OOM, MGN_E_UNSUPPORTED, MGN_E_RCCL = 0, -1, -2, -3, -4, -5, -6
STATUS_NAMES = {0: "MGN_OK", -1: "MGN_E_ARG", -2: "MGN_E_HIP", -3: "MGN_E_STATE", -4: "MGN_E_OOM", -5: "MGN_E_UNSUPPORTED",
                -6: "MGN_E_RCCL"}
MGN_COMM_ID_BYTES = 128
MGN_COMM_RCCL, MGN_COMM_HOST = 0, 1


class MgnConfig(C.Structure):
    _fields_ = [(n, C.c_int32) for n in
                ("Fn", "Fe", "O", "L", "hidden_layers", "mps", "dtype", "rank", "nranks", "device", "n_edge_sets", "Fe2", "ln_mode", "ln_dims")]


class MgnRolloutDesc(C.Structure):
    _fields_ = [("solver", C.c_int32), ("t0", C.c_float), ("t1", C.c_float), ("dt", C.c_float), ("saves_dt", C.c_float),
                ("n_saves", C.c_int32), ("abstol", C.c_float), ("reltol", C.c_float),
                ("x0", C.POINTER(C.c_float)), ("node_type_onehot", C.POINTER(C.c_float)), ("ef_raw", C.POINTER(C.c_float)),
                ("val_mask", C.POINTER(C.c_float)), ("inflow_mask", C.POINTER(C.c_uint8)), ("inflow_data", C.POINTER(C.c_float)),
                ("n_frames", C.c_int32), ("out", C.POINTER(C.c_float)),
                ("n_accept", C.c_int32), ("n_reject", C.c_int32), ("n_rhs", C.c_int32),
                ("inflow_rule", C.c_int32), ("time_f64", C.c_int32),
                ("t0_f64", C.c_double), ("t1_f64", C.c_double), ("dt_f64", C.c_double), ("saves_dt_f64", C.c_double)]


ABI_VERSION = 4      # MGN_ABI_VERSION of include/mgn_hip.h these mirrors were written against (tests/test_julia_shim.py compares)

_f32p = C.POINTER(C.c_float)
_i32p = C.POINTER(C.c_int32)
_i64p = C.POINTER(C.c_int64)
_f64p = C.POINTER(C.c_double)
_H = C.c_void_p

# name -> (restype, argtypes): must list exactly the symbols of include/mgn_hip.h (tests check this)
PROTOTYPES = {
    "mgn_abi_version": (C.c_int, []),
    "mgn_create": (C.c_int, [C.POINTER(MgnConfig), C.POINTER(_H)]),
    "mgn_destroy": (None, [_H]),
    "mgn_last_error": (C.c_char_p, [_H]),
    "mgn_set_stream": (C.c_int, [_H, C.c_void_p]),
    "mgn_synchronize": (C.c_int, [_H]),
    "mgn_param_count": (C.c_size_t, [C.POINTER(MgnConfig)]),
    "mgn_set_params": (C.c_int, [_H, _f32p, C.c_size_t]),
    "mgn_get_params": (C.c_int, [_H, _f32p, C.c_size_t]),
    "mgn_set_norms": (C.c_int, [_H, _f32p, _f32p, _f32p, _f32p, _f32p, _f32p]),
    "mgn_set_graph": (C.c_int, [_H, C.c_int32, C.c_int64, _i32p, _i32p, C.c_int32, _f32p, C.c_int32]),
    "mgn_partition_nodes": (C.c_int, [C.c_int32, _f32p, C.c_int32, C.c_int32, _i32p]),
    "mgn_set_graph_local": (C.c_int, [_H, C.c_int32, _i32p, C.c_int64, C.c_int64, _i32p, _i32p, _i64p, C.c_int32]),
    "mgn_set_edge_set": (C.c_int, [_H, C.c_int32, C.c_int64, _i32p, _i32p, C.c_int32]),
    "mgn_set_edge_features": (C.c_int, [_H, C.c_int32, _f32p]),
    "mgn_edge_set_info": (C.c_int, [_H, C.c_int32, _i64p, _i64p]),
    "mgn_edge_latents_import": (C.c_int, [_H, C.c_int32, _f32p]),
    "mgn_edge_latents_export": (C.c_int, [_H, C.c_int32, _f32p]),
    "mgn_partition_info": (C.c_int, [_H, _i32p, _i32p, _i64p]),
    "mgn_owned_nodes": (C.c_int, [_H, _i32p]),
    "mgn_local_edges": (C.c_int, [_H, _i64p]),
    "mgn_halo_counts": (C.c_int, [_H, _i32p, _i32p]),
    "mgn_halo_nodes": (C.c_int, [_H, _i32p]),
    "mgn_halo_send_index": (C.c_int, [_H, _i32p]),
    "mgn_local_graph": (C.c_int, [_H, _i32p, _i32p, _i32p]),
    "mgn_node_owner": (C.c_int, [_H, _i32p]),
    "mgn_boundary_count": (C.c_int, [_H, _i32p]),
    "mgn_forward": (C.c_int, [_H, _f32p, _f32p, _f32p]),
    "mgn_ode_step": (C.c_int, [_H, _f32p, _f32p, _f32p, _f32p, _f32p]),
    "mgn_set_static": (C.c_int, [_H, _f32p, _f32p, _f32p]),
    "mgn_triangles_to_edges": (C.c_int, [_i32p, C.c_int64, _i32p, _i32p, _i64p]),
    "mgn_world_edges": (C.c_int, [_f32p, C.c_int32, C.c_int32, C.c_float, _i32p, _i32p, C.c_int64, C.c_int32, _i32p, _i32p, _i64p]),
    "mgn_edge_features": (C.c_int, [_f32p, C.c_int32, _i32p, _i32p, C.c_int64, C.c_int32, _f32p]),
    "mgn_triangles_to_edges_dev": (C.c_int, [_H, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_int64, _i64p]),
    "mgn_set_static_mesh": (C.c_int, [_H, C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, C.c_int32, C.c_void_p]),
    "mgn_world_edges_dev": (C.c_int, [_H, C.c_int32, C.c_void_p, C.c_int32, C.c_float, _i64p]),
    "mgn_edge_set_export": (C.c_int, [_H, C.c_int32, _i32p, _i32p]),
    "mgn_rollout": (C.c_int, [_H, C.POINTER(MgnRolloutDesc)]),
    "mgn_step": (C.c_int, [_H, _f32p, _f32p, _f32p, _i32p, C.c_int64, C.c_int32, _f32p, C.c_size_t, _f32p]),
    "mgn_ode_vjp": (C.c_int, [_H, _f32p, _f32p, _f32p, _f32p, _f32p, _f32p, _f32p, _f32p, C.c_size_t]),
    "mgn_forward_vjp": (C.c_int, [_H, _f32p, _f32p, _f32p, _f32p, _f32p, _f32p, C.c_size_t]),
    "mgn_processor_steps": (C.c_int, [_H, _f32p, _f32p, C.c_int32]),
    "mgn_latents_import": (C.c_int, [_H, _f32p, _f32p]),
    "mgn_latents_export": (C.c_int, [_H, _f32p, _f32p]),
    "mgn_latents_randn": (C.c_int, [_H, C.c_uint64]),
    "mgn_latents_checksum": (C.c_int, [_H, _f64p, _f64p, _f64p, _f64p]),
    "mgn_processor_steps_dev": (C.c_int, [_H, C.c_int32]),
    "mgn_fwd_upload": (C.c_int, [_H, _f32p, _f32p]),
    "mgn_fwd_encode": (C.c_int, [_H]),
    "mgn_proc_begin": (C.c_int, [_H]),
    "mgn_proc_edge": (C.c_int, [_H, C.c_int32]),
    "mgn_proc_node": (C.c_int, [_H, C.c_int32, C.c_int32]),
    "mgn_proc_node_phase": (C.c_int, [_H, C.c_int32, C.c_int32]),
    "mgn_proc_edge_phase": (C.c_int, [_H, C.c_int32, C.c_int32]),
    "mgn_edge_boundary_tiles": (C.c_int, [_H, C.c_int32, _i32p, _i32p]),
    "mgn_fwd_decode": (C.c_int, [_H]),
    "mgn_fwd_download": (C.c_int, [_H, _f32p]),
    "mgn_halo_bytes_per_row": (C.c_int, [_H]),
    "mgn_halo_pack": (C.c_int, [_H, C.c_void_p]),
    "mgn_halo_unpack": (C.c_int, [_H, C.c_void_p]),
    "mgn_comm_unique_id": (C.c_int, [C.c_void_p, C.c_int32]),
    "mgn_comm_init": (C.c_int, [_H, C.c_void_p, C.c_size_t, C.c_int32]),
    "mgn_comm_init_file": (C.c_int, [_H, C.c_char_p, C.c_int32]),
    "mgn_comm_destroy": (C.c_int, [_H]),
    "mgn_comm_barrier": (C.c_int, [_H]),
    "mgn_comm_allreduce": (C.c_int, [_H, _f64p, C.c_int32, C.c_int32]),
    "mgn_halo_exchange": (C.c_int, [_H]),
    "mgn_halo_exchange_host": (C.c_int, [_H, _f32p, _f32p, C.c_int32]),
    "mgn_tfrecord_open": (C.c_int, [C.c_char_p, C.c_int32, C.POINTER(_H)]),
    "mgn_tfrecord_next": (C.c_int, [_H]),
    "mgn_tfrecord_feature_count": (C.c_int, [_H]),
    "mgn_tfrecord_feature_name": (C.c_char_p, [_H, C.c_int32]),
    "mgn_tfrecord_feature": (C.c_int, [_H, C.c_char_p, _i32p, C.POINTER(C.c_void_p), _i64p]),
    "mgn_tfrecord_error": (C.c_char_p, [_H]),
    "mgn_tfrecord_close": (None, [_H]),
    "mgn_crc32c": (C.c_uint32, [C.c_void_p, C.c_size_t]),
    "mgn_feature_stats": (C.c_int, [_H, _f32p, C.c_int64, C.c_int32, C.POINTER(C.c_double), C.POINTER(C.c_double)]),
    "mgn_profile_enable": (C.c_int, [_H, C.c_int32]),
    "mgn_profile_read": (C.c_int, [_H, _f64p, _i64p]),
}

_lib = None


def load(path: str | None = None):
    """dlopen libmgn_hip.so and attach prototypes.  Raises if the extension has not been built."""
    global _lib
    if _lib is not None and path is None:
        return _lib
    p = path or os.environ.get("MGN_LIB_PATH") or LIB_PATH   # MGN_LIB_PATH: A/B experiments only
    if not os.path.exists(p):
        raise RuntimeError(
            f"HIP extension {p} is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
            "(hipcc --offload-arch=gfx950).  This package has no CPU fallback.")
    # PyTorch-ROCm wheels bundle their own libamdhip64; two HIP runtimes in one process fight over the device and the one
    # that initialises second reports "No HIP GPUs are available".  Importing torch first makes this library resolve to
    # the runtime that is already loaded (device tensors at the boundary, torch.distributed for the halo exchange).
    if os.environ.get("MGN_TORCH_PRELOAD", "1") != "0" and "torch" not in sys.modules:
        try:
            import torch  # noqa: F401
        except ImportError:
            pass
    lib = C.CDLL(p, mode=C.RTLD_LOCAL)
    for name, (res, args) in PROTOTYPES.items():
        fn = getattr(lib, name)  # AttributeError if the symbol is not exported
        fn.restype = res
        fn.argtypes = args
    if lib.mgn_abi_version() != ABI_VERSION:   # the struct mirrors above are written for ONE version of include/mgn_hip.h
        raise RuntimeError(f"{p} has ABI version {lib.mgn_abi_version()}, these bindings are written for {ABI_VERSION}: rebuild the library")
    if path is None:
        _lib = lib
    return lib


def f32(a):
    return a.ctypes.data_as(_f32p) if a is not None else None


def i32(a):
    return a.ctypes.data_as(_i32p) if a is not None else None


def i64(a):
    return a.ctypes.data_as(_i64p) if a is not None else None
