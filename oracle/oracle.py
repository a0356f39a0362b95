"""ctypes front-end of the CPU oracle (TEST INFRASTRUCTURE — see oracle/dpx_oracle.h).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from pathlib import Path

import numpy as np

_DIR = Path(__file__).resolve().parent
_LIB = None


class OrcConfig(C.Structure):
    # mirrors deplex::config::Config (reference cpp/deplex/include/deplex/config.h:51-81)
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


CELL_DTYPE = np.dtype(
    [
        ("sum", "<f4", 3),
        ("pp", "<f4", 6),
        ("mean", "<f4", 3),
        ("normal", "<f4", 3),
        ("d", "<f4"),
        ("mse", "<f4"),
        ("score", "<f4"),
        ("tol", "<f4"),
        ("valid", "<i4"),
        ("planar", "<i4"),
        ("bin", "<i4"),
        ("used_ql", "<i4"),
    ]
)
PLANE_DTYPE = np.dtype(
    [
        ("sum", "<f4", 3),
        ("pp", "<f4", 6),
        ("mean", "<f4", 3),
        ("normal", "<f4", 3),
        ("d", "<f4"),
        ("mse", "<f4"),
        ("score", "<f4"),
        ("n_pts", "<i4"),
        ("n_cells", "<i4"),
        ("merge_label", "<i4"),
    ]
)
STATS_DTYPE = np.dtype(
    [(k, "<i4") for k in ("nh", "nv", "planar_cells", "grows", "activations", "max_grow", "segments", "merges",
                          "ql_fallbacks", "eig_solves")]
)


def build(force: bool = False) -> Path:
    """Compile oracle/liboracle.so (and oracle/_ref when /root/reference exists)."""
    so = _DIR / "liboracle.so"
    src_m = max((_DIR / "dpx_oracle.cpp").stat().st_mtime, (_DIR / "dpx_oracle.h").stat().st_mtime,
                (_DIR.parent / "deplex_amd" / "csrc" / "dpx_math.h").stat().st_mtime)
    if force or not so.exists() or so.stat().st_mtime < src_m:
        subprocess.run(["make", "-C", str(_DIR), "liboracle.so"], check=True, capture_output=True)
    if Path("/root/reference/libs/dsyev/src").is_dir() and (force or not (_DIR / "_ref" / "libdsyev_ref.so").exists()):
        subprocess.run(["make", "-C", str(_DIR), "ref"], check=True, capture_output=True)
    return so


def lib():
    global _LIB
    if _LIB is None:
        so = build()
        L = C.CDLL(str(so))
        L.orc_set_math.argtypes = [C.c_int]
        L.orc_get_math.restype = C.c_int
        L.orc_default_config.argtypes = [C.POINTER(OrcConfig)]
        L.orc_backproject.argtypes = [C.c_void_p, C.c_int32, C.c_int32, C.c_float, C.c_float, C.c_float, C.c_float,
                                      C.c_void_p]
        L.orc_process.restype = C.c_int
        L.orc_process.argtypes = [C.POINTER(OrcConfig), C.c_int32, C.c_int32, C.c_void_p, C.c_int64, C.c_void_p,
                                  C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_char_p, C.c_int32]
        L.orc_process_depth_batch.restype = C.c_int
        L.orc_process_depth_batch.argtypes = [C.POINTER(OrcConfig), C.c_int32, C.c_int32, C.c_void_p, C.c_int32,
                                              C.c_float, C.c_float, C.c_float, C.c_float, C.c_void_p, C.c_int32]
        L.orc_eig3.restype = C.c_int
        L.orc_eig3.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        L.orc_points_mse.restype = C.c_float
        L.orc_points_mse.argtypes = [C.c_void_p, C.c_int64]
        _LIB = L
    return _LIB


def default_config(**overrides) -> OrcConfig:
    cfg = OrcConfig()
    lib().orc_default_config(C.byref(cfg))
    for k, v in overrides.items():
        if not hasattr(cfg, k):
            raise AttributeError(k)
        setattr(cfg, k, v)
    return cfg


def config_from_ini(path) -> OrcConfig:
    """The reference's .ini reader (cpp/deplex/src/deplex/config.cpp:28-80): key=value lines, no trimming,
    '#' comments, unknown keys ignored (the reference prints a warning)."""
    keys = {
        "patchSize": ("patch_size", int), "histogramBinsPerCoord": ("histogram_bins_per_coord", int),
        "minCosAngleForMerge": ("min_cos_angle_merge", float), "maxMergeDist": ("max_merge_dist", float),
        "minRegionGrowingCandidateSize": ("min_region_growing_candidate_size", int),
        "minRegionGrowingCellsActivated": ("min_region_growing_cells_activated", int),
        "minRegionPlanarityScore": ("min_region_planarity_score", float),
        "depthSigmaCoeff": ("depth_sigma_coeff", float), "depthSigmaMargin": ("depth_sigma_margin", float),
        "minPtsPerCell": ("min_pts_per_cell", int),
        "depthDiscontinuityThreshold": ("depth_discontinuity_threshold", float),
        "maxNumberDepthDiscontinuity": ("max_number_depth_discontinuity", int),
        "ransacRefinement": ("ransac_refinement", int), "ransacMaxIterations": ("ransac_max_iterations", int),
        "ransacThreshold": ("ransac_threshold", float), "ransacInliersRatio": ("ransac_inliers_ratio", float),
    }
    cfg = default_config()
    with open(path) as fh:
        for line in fh.read().split("\n"):
            if not line or line[0] == "#" or "=" not in line or line[0] == "=":
                continue
            k, v = line.split("=", 1)
            if k in keys:
                name, conv = keys[k]
                setattr(cfg, name, _stoi(v) if conv is int else _stof(v))
    return cfg


def _stoi(v: str) -> int:
    """std::stoi: longest leading integer prefix, ValueError (std::invalid_argument) if none."""
    import re
    m = re.match(r"\s*[+-]?\d+", v)
    if not m:
        raise ValueError("stoi")
    return int(m.group(0))


def _stof(v: str) -> float:
    """std::stof: longest leading decimal floating prefix."""
    import re
    m = re.match(r"\s*[+-]?(\d+\.?\d*|\.\d+)([eE][+-]?\d+)?", v)
    if not m:
        raise ValueError("stof")
    return float(m.group(0))


def set_math(flavour: str) -> None:
    lib().orc_set_math({"glibc": 0, "devmath": 1}[flavour])


def backproject(depth: np.ndarray, fx, fy, cx, cy) -> np.ndarray:
    depth = np.ascontiguousarray(depth, dtype=np.uint16)
    h, w = depth.shape
    xyz = np.empty((h * w, 3), dtype=np.float32)
    lib().orc_backproject(depth.ctypes.data, h, w, np.float32(fx), np.float32(fy), np.float32(cx), np.float32(cy),
                          xyz.ctypes.data)
    return xyz


class OracleError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(msg)
        self.code = code


def process(cfg: OrcConfig, height: int, width: int, xyz: np.ndarray, debug: bool = False, planes_cap: int = 0):
    """PlaneExtractor(height, width, cfg).process(xyz).  Returns labels, or (labels, cells, planes, cell_labels,
    stats) with debug=True.  Raises OracleError with the reference's message for its throwing cases."""
    xyz = np.ascontiguousarray(xyz, dtype=np.float32)
    n = xyz.shape[0] if xyz.ndim == 2 else 0
    labels = np.zeros(max(n, 1), dtype=np.int32)
    P = max(cfg.patch_size, 1)
    nh, nv = width // P, height // P
    cells = np.zeros(nh * nv, dtype=CELL_DTYPE) if debug else None
    if planes_cap <= 0:  # one plane per cell is the most the algorithm can produce
        planes_cap = max(4096, nh * nv)
    planes = np.zeros(planes_cap, dtype=PLANE_DTYPE) if debug else None
    cl = np.zeros(nh * nv, dtype=np.int32) if debug else None
    stats = np.zeros(1, dtype=STATS_DTYPE)
    err = C.create_string_buffer(512)
    rc = lib().orc_process(C.byref(cfg), height, width, xyz.ctypes.data if n else None, n, labels.ctypes.data,
                           cells.ctypes.data if debug else None, planes.ctypes.data if debug else None, planes_cap,
                           cl.ctypes.data if debug else None, stats.ctypes.data, err, 512)
    if rc != 0:
        raise OracleError(rc, err.value.decode())
    labels = labels[:n]
    if not debug:
        return labels
    return labels, cells, planes[: int(stats["segments"][0])], cl.reshape(nv, nh), stats[0]


def process_depth_batch(cfg: OrcConfig, depth: np.ndarray, fx, fy, cx, cy, threads: int = 1) -> np.ndarray:
    depth = np.ascontiguousarray(depth, dtype=np.uint16)
    b, h, w = depth.shape
    labels = np.empty((b, h * w), dtype=np.int32)
    rc = lib().orc_process_depth_batch(C.byref(cfg), h, w, depth.ctypes.data, b, np.float32(fx), np.float32(fy),
                                       np.float32(cx), np.float32(cy), labels.ctypes.data, threads)
    if rc != 0:
        raise OracleError(rc, "oracle batch failed")
    return labels


def eig3(A: np.ndarray):
    A = np.ascontiguousarray(A, dtype=np.float64)
    Q = np.zeros((3, 3)); w = np.zeros(3)
    rc = lib().orc_eig3(A.ctypes.data, Q.ctypes.data, w.ctypes.data)
    return rc, Q, w


_REF = None


def ref_dsyevh3(A: np.ndarray):
    """The reference's own compiled dsyevh3 (oracle/_ref/libdsyev_ref.so); None if not built."""
    global _REF
    so = _DIR / "_ref" / "libdsyev_ref.so"
    if _REF is None:
        if not so.exists():
            return None
        _REF = C.CDLL(str(so))
        _REF.dsyevh3.restype = C.c_int
        _REF.dsyevh3.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
    A = np.ascontiguousarray(A, dtype=np.float64)
    Q = np.zeros((3, 3)); w = np.zeros(3)
    rc = _REF.dsyevh3(A.ctypes.data, Q.ctypes.data, w.ctypes.data)
    return rc, Q, w
