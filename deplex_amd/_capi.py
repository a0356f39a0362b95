"""ctypes binding of the C-ABI in include/deplex_amd.h (libdeplex_amd.so).

The shared library holds the hand-written HIP kernels; there is no Python/CPU fallback.  If the library is
missing this module raises at import time of the symbols (``lib()``), loudly, instead of degrading.
"""
from __future__ import annotations

import ctypes as C
import os
from pathlib import Path

import numpy as np

_PKG = Path(__file__).resolve().parent
# DPX_LIB_PATH selects another build of the same library (kernel tuning experiments); default is the in-tree build
LIB_PATH = Path(os.environ["DPX_LIB_PATH"]) if os.environ.get("DPX_LIB_PATH") else _PKG / "lib" / "libdeplex_amd.so"

DPX_OK = 0
DPX_ERR_PATCH_ZERO = -1
DPX_ERR_SHAPE = -2
DPX_ERR_UNDEFINED = -3
DPX_ERR_HIP = -4
DPX_ERR_ARG = -5
DPX_ERR_CAPACITY = -6
DPX_ERR_NO_DEVICE = -7
DPX_ERR_IO = -8
K_NAMES = ("planefit", "cellgraph", "regions", "labels")


class DpxConfig(C.Structure):
    """POD mirror of deplex::config::Config (reference cpp/deplex/include/deplex/config.h:51-81)."""
    _fields_ = [
        ("patch_size", C.c_int32),
        ("histogram_bins_per_coord", C.c_int32),
        ("min_cos_angle_merge", C.c_float),
        ("max_merge_dist", C.c_float),
        ("min_region_growing_candidate_size", C.c_int32),
        ("min_region_growing_cells_activated", C.c_int32),
        ("min_region_planarity_score", C.c_float),
        ("depth_sigma_coeff", C.c_float),
        ("depth_sigma_margin", C.c_float),
        ("min_pts_per_cell", C.c_int32),
        ("depth_discontinuity_threshold", C.c_float),
        ("max_number_depth_discontinuity", C.c_int32),
        ("ransac_refinement", C.c_int32),
        ("ransac_max_iterations", C.c_int32),
        ("ransac_threshold", C.c_float),
        ("ransac_inliers_ratio", C.c_float),
    ]


CELL_DTYPE = np.dtype([("sum", "<f4", 3), ("pp", "<f4", 6), ("normal", "<f4", 3), ("d", "<f4"), ("mse", "<f4"),
                       ("tol", "<f4"), ("info", "<u4")])
PLANE_DTYPE = np.dtype([("sum", "<f4", 3), ("pp", "<f4", 6), ("mean", "<f4", 3), ("normal", "<f4", 3), ("d", "<f4"),
                        ("mse", "<f4"), ("score", "<f4"), ("n_pts", "<i4"), ("n_cells", "<i4"),
                        ("merge_label", "<i4")])
STATS_DTYPE = np.dtype([(k, "<i4") for k in ("planar_cells", "grows", "activations", "segments", "merges", "overflow")])
assert CELL_DTYPE.itemsize == 64 and PLANE_DTYPE.itemsize == 84

# name -> (restype, argtypes); every symbol include/deplex_amd.h declares
SYMBOLS = {
    "dpx_version": (C.c_char_p, []),
    "dpx_last_error": (C.c_char_p, []),
    "dpx_default_config": (None, [C.POINTER(DpxConfig)]),
    "dpx_config_from_ini": (C.c_int, [C.c_char_p, C.POINTER(DpxConfig)]),
    "dpx_create": (C.c_int, [C.POINTER(DpxConfig), C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.POINTER(C.c_void_p)]),
    "dpx_destroy": (None, [C.c_void_p]),
    "dpx_process_pcd": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p]),
    "dpx_process_pcd_batch": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p]),
    "dpx_process_depth_batch": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.c_float, C.c_float, C.c_float, C.c_float, C.c_void_p]),
    "dpx_process_depth_batch_cells": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.c_float, C.c_float, C.c_float, C.c_float, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.POINTER(C.c_int64)]),
    "dpx_get_packed_planes": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.POINTER(C.c_int64)]),
    "dpx_expand_cell_labels": (C.c_int, [C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_int32, C.c_void_p]),
    "dpx_host_alloc": (C.c_void_p, [C.c_size_t]),
    "dpx_host_free": (None, [C.c_void_p]),
    "dpx_process_depth_batch_device": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.c_float, C.c_float, C.c_float, C.c_float, C.c_void_p, C.c_void_p]),
    "dpx_process_pcd_batch_device": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p]),
    "dpx_pipeline_create": (C.c_int, [C.POINTER(DpxConfig), C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.POINTER(C.c_void_p)]),
    "dpx_pipeline_destroy": (None, [C.c_void_p]),
    "dpx_pipeline_submit_depth": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.c_float, C.c_float, C.c_float, C.c_float, C.c_void_p]),
    "dpx_pipeline_submit_depth_cells": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.c_float, C.c_float, C.c_float, C.c_float, C.c_void_p]),
    "dpx_pipeline_submit_depth_host": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.c_float, C.c_float, C.c_float, C.c_float, C.c_void_p]),
    "dpx_pipeline_submit_depth_cells_host": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.c_float, C.c_float, C.c_float, C.c_float, C.c_void_p]),
    "dpx_pipeline_submit_depth_many": (C.c_int, [C.c_void_p, C.c_int32, C.POINTER(C.c_void_p), C.c_int32, C.c_float, C.c_float, C.c_float, C.c_float, C.POINTER(C.c_void_p)]),
    "dpx_pipeline_submit_pcd": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p]),
    "dpx_pipeline_sync": (C.c_int, [C.c_void_p]),
    "dpx_pipeline_flush": (C.c_int, [C.c_void_p]),
    "dpx_pipeline_slot": (C.c_void_p, [C.c_void_p, C.c_int32]),
    "dpx_prepare_intrinsics": (C.c_int, [C.c_void_p, C.c_float, C.c_float, C.c_float, C.c_float, C.POINTER(C.c_int32)]),
    "dpx_backproject": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, C.c_float, C.c_float, C.c_float, C.c_float, C.c_void_p, C.c_int32]),
    "dpx_grid_dims": (C.c_int, [C.c_void_p, C.POINTER(C.c_int32), C.POINTER(C.c_int32)]),
    "dpx_get_cells": (C.c_int, [C.c_void_p, C.c_int32, C.c_void_p]),
    "dpx_get_cell_edges": (C.c_int, [C.c_void_p, C.c_int32, C.c_void_p]),
    "dpx_get_cell_labels": (C.c_int, [C.c_void_p, C.c_int32, C.c_void_p]),
    "dpx_get_planes": (C.c_int, [C.c_void_p, C.c_int32, C.c_void_p, C.c_int32, C.POINTER(C.c_int32)]),
    "dpx_get_frame_stats": (C.c_int, [C.c_void_p, C.c_int32, C.c_void_p]),
    "dpx_debug_region_phases": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, C.c_void_p]),
    "dpx_debug_cores_phases": (C.c_int, [C.c_void_p, C.c_int32, C.c_void_p]),
    "dpx_get_options": (C.c_int, [C.c_void_p, C.c_char_p, C.c_int32]),
    "dpx_debug_read_cores": (C.c_int64, [C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, C.c_int64]),
    "dpx_set_profiling": (C.c_int, [C.c_void_p, C.c_int32]),
    "dpx_get_kernel_ms": (C.c_int, [C.c_void_p, C.c_void_p]),
    "dpx_time_planefit": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_float, C.c_float, C.c_float, C.c_float, C.c_int32, C.c_void_p, C.POINTER(C.c_float)]),
    "dpx_time_planefit_group": (C.c_int, [C.c_void_p, C.POINTER(C.c_void_p), C.c_int32, C.c_int32, C.c_int32, C.c_float, C.c_float, C.c_float, C.c_float, C.c_int32, C.c_void_p, C.POINTER(C.c_float)]),
    "dpx_planefit_bytes": (C.c_int64, [C.c_void_p, C.c_int32, C.c_int32]),
    "dpx_read_depth_png": (C.c_int, [C.c_char_p, C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.POINTER(C.POINTER(C.c_uint16))]),
    "dpx_free": (None, [C.c_void_p]),
    "dpx_debug_residency": (C.c_int, [C.c_void_p, C.POINTER(C.c_double)]),
    "dpx_debug_trace": (C.c_int, [C.c_int32, C.c_void_p, C.c_uint32, C.POINTER(C.c_uint32)]),
    "dpx_test_math": (C.c_int, [C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int32]),
    "dpx_test_eig3": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int32]),
}

_LIB = None


class DeplexError(RuntimeError):
    """Raised for every failing C-ABI call; `.code` is the DPX_ERR_* status.  The reference throws
    std::runtime_error (pybind: RuntimeError) with the same message for its own throwing cases."""

    def __init__(self, code: int, msg: str):
        super().__init__(msg)
        self.code = code


def _preload_hip_runtime() -> None:
    """PyTorch-ROCm wheels bundle their own libamdhip64.so (same SONAME as /opt/rocm's).  Two HIP runtimes in one
    process cannot both own the GPU, so bind to the copy torch will use: load it by path first (without importing
    torch); libdeplex_amd.so's NEEDED libamdhip64.so.7 then resolves to it, and a later `import torch` finds the
    same file already mapped."""
    import importlib.util
    try:
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        spec = None
    if spec is None or not spec.origin:
        return
    cand = Path(spec.origin).parent / "lib" / "libamdhip64.so"
    if cand.exists():
        try:
            C.CDLL(str(cand), mode=getattr(os, "RTLD_GLOBAL", 256) | getattr(os, "RTLD_NOW", 2))
        except OSError:
            pass


def lib():
    global _LIB
    if _LIB is None:
        if not LIB_PATH.exists():
            raise ImportError(
                f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "(hipcc --offload-arch=gfx950). deplex_amd has no CPU fallback.")
        _preload_hip_runtime()
        L = C.CDLL(str(LIB_PATH), mode=getattr(os, "RTLD_NOW", 2))
        for name, (res, args) in SYMBOLS.items():
            fn = getattr(L, name)  # AttributeError if the library does not export a declared symbol
            fn.restype = res
            fn.argtypes = args
        _LIB = L
    return _LIB


def check(rc: int) -> None:
    if rc != DPX_OK:
        raise DeplexError(rc, lib().dpx_last_error().decode(errors="replace"))
