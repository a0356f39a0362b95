"""Python mirror of the reference's `deplex.plane_extraction` module over the C-ABI.

Reference surface (cpp/pybind/plane_extraction/plane_extraction.cpp:28-37):
    Config(path: str)
    PlaneExtractor(image_height, image_width, config=Config()).process(pcd_array) -> 1-D int32 labels
Added (not in the reference): batched depth entry points and the per-plane / per-cell records the
acceptance criteria grade.
"""
from __future__ import annotations

import ctypes as C
from typing import Optional

import numpy as np

from . import _capi
from ._capi import DeplexError, DpxConfig

_FIELDS = [f[0] for f in DpxConfig._fields_]
MAX_GROUP = 32  # submissions per pipeline group (dpx::kMaxGroup)


class Config:
    """deplex::config::Config — defaults of cpp/deplex/include/deplex/config.h:51-81; `Config(path)` parses
    an .ini exactly like cpp/deplex/src/deplex/config.cpp:28-80 (the parser lives in the C-ABI library)."""

    def __init__(self, path: Optional[str] = None, **overrides):
        self._c = DpxConfig()
        if path is None:
            _capi.lib().dpx_default_config(C.byref(self._c))
        else:
            _capi.check(_capi.lib().dpx_config_from_ini(str(path).encode(), C.byref(self._c)))
        for k, v in overrides.items():
            setattr(self, k, v)

    def __getattr__(self, name):
        if name in _FIELDS:
            v = getattr(self.__dict__["_c"], name)
            return bool(v) if name == "ransac_refinement" else v
        raise AttributeError(name)

    def __setattr__(self, name, value):
        if name == "_c":
            object.__setattr__(self, name, value)
        elif name in _FIELDS:
            setattr(self._c, name, int(value) if isinstance(value, (bool, np.bool_)) else value)
        else:
            raise AttributeError(f"Config has no parameter {name!r}")

    def copy(self) -> "Config":
        c = Config()
        C.memmove(C.byref(c._c), C.byref(self._c), C.sizeof(DpxConfig))
        return c

    def __repr__(self):
        return "Config(" + ", ".join(f"{k}={getattr(self, k)!r}" for k in _FIELDS) + ")"


class PlaneExtractor:
    """deplex::PlaneExtractor (cpp/deplex/include/deplex/plane_extractor.h:28-56).

    Construction errors and process() shape errors raise DeplexError (a RuntimeError) carrying the
    reference's messages (plane_extractor.cpp:161-164, :188-194).
    """

    def __init__(self, image_height: int, image_width: int, config: Optional[Config] = None, *, max_batch: int = 1,
                 device: int = 0):
        self._h = C.c_void_p()
        cfg = config if config is not None else Config()
        self.config = cfg
        self.image_height, self.image_width = int(image_height), int(image_width)
        self.max_batch = int(max_batch)
        self.device = int(device)
        _capi.check(_capi.lib().dpx_create(C.byref(cfg._c), self.image_height, self.image_width, self.max_batch,
                                          self.device, C.byref(self._h)))
        nh, nv = C.c_int32(), C.c_int32()
        _capi.check(_capi.lib().dpx_grid_dims(self._h, C.byref(nh), C.byref(nv)))
        self.nh, self.nv = nh.value, nv.value

    def __del__(self):
        h = getattr(self, "_h", None)
        if getattr(self, "_borrowed", False):
            return
        if h is not None and h.value:
            try:
                _capi.lib().dpx_destroy(h)
            except Exception:
                pass
            self._h = C.c_void_p()

    close = __del__

    # ---- reference API -------------------------------------------------------------------------------
    def process(self, pcd_array) -> np.ndarray:
        """Labels for one organised cloud.  Accepts any (N, 3) float array (float64 is cast to float32 with
        round-to-nearest, like pybind11's Eigen caster does for the reference)."""
        pcd = np.ascontiguousarray(np.asarray(pcd_array), dtype=np.float32)
        if pcd.ndim != 2 or pcd.shape[1] != 3:
            raise TypeError("process(): pcd_array must have shape (N, 3)")  # pybind's Eigen caster rejects it too
        n = pcd.shape[0]
        labels = np.zeros(max(n, 1), dtype=np.int32)
        _capi.check(_capi.lib().dpx_process_pcd(self._h, pcd.ctypes.data if pcd.size else None, n,
                                               labels.ctypes.data))
        return labels[:n]

    # ---- batched / fused entry points ----------------------------------------------------------------
    def _check_depth(self, depth):
        depth = np.ascontiguousarray(depth, dtype=np.uint16)
        if depth.ndim == 2:
            depth = depth[None]
        b, h, w = depth.shape
        if (h, w) != (self.image_height, self.image_width):
            raise DeplexError(_capi.DPX_ERR_SHAPE, f"Error! Number of points doesn't match image shape: {h * w} != "
                              f"{self.image_height} x {self.image_width}")
        return depth

    def process_depth_batch(self, depth, fx, fy, cx, cy, out: Optional[np.ndarray] = None) -> np.ndarray:
        """depth: (B, H, W) uint16 -> labels (B, H*W) int32; back-projection fused on the device.  `out`: a C-contiguous
        int32 array of that shape to write into (e.g. a slice of a page-locked buffer, deplex_amd.pinned_empty)."""
        depth = self._check_depth(depth)
        b, h, w = depth.shape
        if out is None:
            out = np.empty((b, h * w), dtype=np.int32)
        elif out.dtype != np.int32 or out.size != b * h * w or not out.flags.c_contiguous:
            raise TypeError("out must be a C-contiguous int32 array of B * H * W elements")
        _capi.check(_capi.lib().dpx_process_depth_batch(self._h, depth.ctypes.data, b, fx, fy, cx, cy, out.ctypes.data))
        return out.reshape(b, h * w)

    def process_depth_batch_cells(self, depth, fx, fy, cx, cy, out: Optional[np.ndarray] = None, planes: bool = True):
        """Cell-level output of the same computation (dpx_process_depth_batch_cells): (cell_labels (B, nv, nh) int32 —
        the merged 1-based label of every cell, what the pixel map holds for all pixels of that cell — and, with
        planes=True, a list of B plane-record arrays).  80x fewer bytes come back over PCIe than with the pixel map."""
        depth = self._check_depth(depth)
        b = depth.shape[0]
        if out is None:
            out = np.empty((b, self.nv, self.nh), dtype=np.int32)
        elif out.dtype != np.int32 or out.size != b * self.nv * self.nh or not out.flags.c_contiguous:
            raise TypeError("out must be a C-contiguous int32 array of B * nv * nh elements")
        if not planes:
            _capi.check(_capi.lib().dpx_process_depth_batch_cells(self._h, depth.ctypes.data, b, fx, fy, cx, cy, out.ctypes.data,
                                                                 None, None, 0, None))
            return out.reshape(b, self.nv, self.nh), None
        counts = np.zeros(max(b, 1), dtype=np.int32)
        total = C.c_int64()
        _capi.check(_capi.lib().dpx_process_depth_batch_cells(self._h, depth.ctypes.data, b, fx, fy, cx, cy, out.ctypes.data,
                                                             counts.ctypes.data, None, 0, C.byref(total)))
        recs = np.zeros(max(int(total.value), 1), dtype=_capi.PLANE_DTYPE)  # (sized from the count: one copy out of the staging)
        _capi.check(_capi.lib().dpx_get_packed_planes(self._h, recs.ctypes.data, int(total.value), None))
        recs = recs[: int(total.value)]
        offs = np.concatenate([[0], np.cumsum(counts[:b])])
        return out.reshape(b, self.nv, self.nh), [recs[offs[k]:offs[k + 1]] for k in range(b)]

    def expand_cell_labels(self, cell_labels) -> np.ndarray:
        """Impl::toImageLabels (plane_extractor.cpp:455-470) on the host: (B, nv, nh) cell labels -> (B, H*W) pixel labels."""
        cl = np.ascontiguousarray(cell_labels, dtype=np.int32).reshape(-1, self.nv, self.nh)
        out = np.empty((cl.shape[0], self.image_height * self.image_width), dtype=np.int32)
        _capi.check(_capi.lib().dpx_expand_cell_labels(self.image_height, self.image_width, int(self.config.patch_size),
                                                      cl.ctypes.data, cl.shape[0], out.ctypes.data))
        return out

    def process_pcd_batch(self, pcd) -> np.ndarray:
        pcd = np.ascontiguousarray(pcd, dtype=np.float32)
        b = pcd.shape[0]
        if pcd.shape[1:] != (self.image_height * self.image_width, 3):
            raise DeplexError(_capi.DPX_ERR_SHAPE, "Error! Number of points doesn't match image shape: "
                              f"{pcd.shape[1]} != {self.image_height} x {self.image_width}")
        labels = np.empty((b, pcd.shape[1]), dtype=np.int32)
        _capi.check(_capi.lib().dpx_process_pcd_batch(self._h, pcd.ctypes.data, b, labels.ctypes.data))
        return labels

    def process_depth_device(self, d_depth_ptr: int, batch: int, fx, fy, cx, cy, d_labels_ptr: int, stream: int = 0):
        """Stream-ordered, device pointers (e.g. torch tensors' data_ptr()); no synchronisation."""
        _capi.check(_capi.lib().dpx_process_depth_batch_device(self._h, d_depth_ptr, batch, fx, fy, cx, cy,
                                                              d_labels_ptr, stream))

    def process_pcd_device(self, d_xyz_ptr: int, batch: int, d_labels_ptr: int, stream: int = 0):
        _capi.check(_capi.lib().dpx_process_pcd_batch_device(self._h, d_xyz_ptr, batch, d_labels_ptr, stream))

    def prepare_intrinsics(self, fx, fy, cx, cy) -> bool:
        """Runs the one-time exact-division check for these intrinsics; True if the fast sequence is exact."""
        ok = C.c_int32()
        _capi.check(_capi.lib().dpx_prepare_intrinsics(self._h, fx, fy, cx, cy, C.byref(ok)))
        return bool(ok.value)

    # ---- records of the last call -----------------------------------------------------------------------
    def cells(self, frame: int = 0) -> np.ndarray:
        out = np.zeros(self.nh * self.nv, dtype=_capi.CELL_DTYPE)
        _capi.check(_capi.lib().dpx_get_cells(self._h, frame, out.ctypes.data))
        return out

    def cell_edges(self, frame: int = 0) -> np.ndarray:
        out = np.zeros(max(self.nh * self.nv, 1), dtype=np.uint8)
        _capi.check(_capi.lib().dpx_get_cell_edges(self._h, frame, out.ctypes.data))
        return out[: self.nh * self.nv]

    def cell_labels(self, frame: int = 0) -> np.ndarray:
        out = np.zeros(max(self.nh * self.nv, 1), dtype=np.int32)
        _capi.check(_capi.lib().dpx_get_cell_labels(self._h, frame, out.ctypes.data))
        return out[: self.nh * self.nv].reshape(self.nv, self.nh)

    def planes(self, frame: int = 0) -> np.ndarray:
        st = self.frame_stats(frame)
        n = int(st["segments"])
        out = np.zeros(max(n, 1), dtype=_capi.PLANE_DTYPE)
        got = C.c_int32()
        _capi.check(_capi.lib().dpx_get_planes(self._h, frame, out.ctypes.data, n, C.byref(got)))
        return out[:n]

    def options(self) -> dict:
        """Which kernels this handle runs (dpx_get_options): {"regions": "tier2" | "cores" | ..., ...}."""
        buf = C.create_string_buffer(512)
        n = _capi.lib().dpx_get_options(self._h, buf, 512)
        if n < 0:
            _capi.check(n)
        return dict(kv.split("=", 1) for kv in buf.value.decode().split(";") if "=" in kv)

    def frame_stats(self, frame: int = 0):
        out = np.zeros(1, dtype=_capi.STATS_DTYPE)
        _capi.check(_capi.lib().dpx_get_frame_stats(self._h, frame, out.ctypes.data))
        return out[0]

    # ---- the reference's DEBUG_DEPLEX stage dumps (plane_extractor.cpp:204-263, 514-549) ---------------------
    def dump_debug(self, directory, labels, frame: int = 0) -> dict:
        """Writes, for one frame of the last call, the CSV files a DEBUG_DEPLEX build of the reference leaves in its
        working directory, byte-compatible in layout, so the two can be diffed stage by stage:
        dbg_1_planar_cells.csv (H x W, cell id inside planar cells, else 0), dbg_3_labels.csv (label map before
        refinement = merged cell labels expanded), dbg_4_refined_labels.csv (only with ransac_refinement: the final
        map).  `labels` is that frame's output of process*().  Returns the reference's three [DebugInfo] counters."""
        from pathlib import Path
        d = Path(directory)
        d.mkdir(parents=True, exist_ok=True)
        h, w, P = self.image_height, self.image_width, max(int(self.config.patch_size), 1)
        P = min(P, h, w)
        cells = self.cells(frame)
        planar = (cells["info"] & 2) != 0
        img = np.zeros((h, w), dtype=np.int32)
        ids = np.where(planar, np.arange(self.nh * self.nv, dtype=np.int32), 0).reshape(self.nv, self.nh)
        img[: self.nv * P, : self.nh * P] = np.repeat(np.repeat(ids, P, axis=0), P, axis=1)
        np.savetxt(d / "dbg_1_planar_cells.csv", img, fmt="%d", delimiter=",")
        merged = np.zeros((h, w), dtype=np.int32)
        merged[: self.nv * P, : self.nh * P] = np.repeat(np.repeat(self.cell_labels(frame), P, axis=0), P, axis=1)
        np.savetxt(d / "dbg_3_labels.csv", merged, fmt="%d", delimiter=",")
        final = np.asarray(labels, dtype=np.int32).reshape(h, w)
        if self.config.ransac_refinement:
            np.savetxt(d / "dbg_4_refined_labels.csv", final, fmt="%d", delimiter=",")
        planes = self.planes(frame)
        return {"planar_cells": int(planar.sum()), "plane_segments": int(len(planes)),
                "planes_after_merge": int(len(np.unique(planes["merge_label"]))) if len(planes) else 0}

    # ---- measurement ------------------------------------------------------------------------------------
    def set_profiling(self, enable: bool) -> None:
        _capi.check(_capi.lib().dpx_set_profiling(self._h, int(enable)))

    def kernel_ms(self) -> dict:
        ms = (C.c_float * 4)()
        _capi.check(_capi.lib().dpx_get_kernel_ms(self._h, ms))
        return dict(zip(_capi.K_NAMES, list(ms)))

    def time_planefit(self, d_in_ptr: int, input_kind: int, batch: int, fx, fy, cx, cy, iters: int, stream: int = 0) -> float:
        ms = C.c_float()
        _capi.check(_capi.lib().dpx_time_planefit(self._h, d_in_ptr, input_kind, batch, fx, fy, cx, cy, iters, stream,
                                                 C.byref(ms)))
        return ms.value

    def time_planefit_group(self, d_in_ptrs, input_kind: int, per_sub: int, fx, fy, cx, cy, iters: int, stream: int = 0) -> float:
        """One plane-fit launch over len(d_in_ptrs) device buffers of per_sub frames each (the launch shape of a
        Pipeline group), repeated `iters` times on `stream`; average ms per launch from HIP events on that stream."""
        ms = C.c_float()
        arr = (C.c_void_p * len(d_in_ptrs))(*[int(p) for p in d_in_ptrs])
        _capi.check(_capi.lib().dpx_time_planefit_group(self._h, arr, len(d_in_ptrs), input_kind, per_sub, fx, fy, cx, cy,
                                                       iters, stream, C.byref(ms)))
        return ms.value

    def planefit_bytes(self, input_kind: int, batch: int) -> int:
        return int(_capi.lib().dpx_planefit_bytes(self._h, input_kind, batch))


class Pipeline:
    """Streamed, pipelined processing of depth batches that already live in device memory (dpx_pipeline_*):
    `slots` groups of `group` submissions in flight, one stream per slot; a group is launched when it is full (or at
    sync()): one plane-fit launch over all its frames, then region growing and label expansion.  Results are complete
    after sync()."""

    def __init__(self, image_height: int, image_width: int, config: Optional[Config] = None, *, batch: int = 64,
                 slots: int = 16, group: int = 1, device: int = 0):
        self._p = C.c_void_p()
        self._keep = []  # host arrays of submissions in flight (submit_depth_host)
        cfg = config if config is not None else Config()
        self.config, self.device = cfg, int(device)
        self.image_height, self.image_width, self.batch, self.slots = int(image_height), int(image_width), int(batch), int(slots)
        self.group = max(1, min(MAX_GROUP, int(group)))
        _capi.check(_capi.lib().dpx_pipeline_create(C.byref(cfg._c), self.image_height, self.image_width, self.batch,
                                                   int(device), self.slots, int(group), C.byref(self._p)))

    def submit_depth(self, d_depth_ptr: int, batch: int, fx, fy, cx, cy, d_labels_ptr: int) -> None:
        _capi.check(_capi.lib().dpx_pipeline_submit_depth(self._p, d_depth_ptr, batch, fx, fy, cx, cy, d_labels_ptr))

    def submit_depth_cells(self, d_depth_ptr: int, batch: int, fx, fy, cx, cy, d_cell_labels_ptr: int) -> None:
        """Cell-level output: int32 [batch][nv][nh] merged cell labels into device memory, no pixel label map (no label
        expansion launch for the group)."""
        _capi.check(_capi.lib().dpx_pipeline_submit_depth_cells(self._p, d_depth_ptr, batch, fx, fy, cx, cy, d_cell_labels_ptr))

    def submit_depth_host(self, depth: np.ndarray, fx, fy, cx, cy, out: np.ndarray, cells: bool = False) -> None:
        """Host-buffer submission (dpx_pipeline_submit_depth_host / _cells_host): `depth` (batch, H, W) uint16 and `out`
        (batch, H*W) int32 — or (batch, nv, nh) with cells=True — are HOST arrays, ideally page-locked (pinned_empty);
        the slot's stream carries the copy in, the kernels and the copy out.  Both arrays are kept alive by the pipeline
        and must not be touched until sync()."""
        if depth.dtype != np.uint16 or not depth.flags.c_contiguous or out.dtype != np.int32 or not out.flags.c_contiguous:
            raise TypeError("depth must be C-contiguous uint16, out C-contiguous int32")
        # The C side queues asynchronous copies of batch*H*W*2 bytes in and batch*H*W*4 (or batch*nv*nh*4) bytes out: a wrong
        # shape would be read or written out of bounds in host memory after this call has returned.
        if depth.ndim != 3 or tuple(depth.shape) != (self.batch, self.image_height, self.image_width):
            raise TypeError(f"depth must have shape ({self.batch}, {self.image_height}, {self.image_width}), got {tuple(depth.shape)}")
        if cells:
            ex0 = self.slot(0)
            want = self.batch * ex0.nv * ex0.nh
        else:
            want = self.batch * self.image_height * self.image_width
        if out.size != want:
            raise TypeError(f"out must hold {want} int32 values ({'cell labels' if cells else 'pixel labels'}), got {out.size}")
        batch = int(depth.shape[0])
        fn = _capi.lib().dpx_pipeline_submit_depth_cells_host if cells else _capi.lib().dpx_pipeline_submit_depth_host
        self._keep.append((depth, out))
        _capi.check(fn(self._p, depth.ctypes.data, batch, fx, fy, cx, cy, out.ctypes.data))

    def submit_depth_many(self, d_depth_ptrs, batch: int, fx, fy, cx, cy, d_labels_ptrs) -> None:
        """len(d_depth_ptrs) submissions in one call (dpx_pipeline_submit_depth_many): same order and grouping as single calls."""
        n = len(d_depth_ptrs)
        if n != len(d_labels_ptrs):
            raise ValueError("one output buffer per input buffer")
        ins = (C.c_void_p * n)(*[int(p) for p in d_depth_ptrs])
        outs = (C.c_void_p * n)(*[int(p) for p in d_labels_ptrs])
        _capi.check(_capi.lib().dpx_pipeline_submit_depth_many(self._p, n, ins, batch, fx, fy, cx, cy, outs))

    def submit_pcd(self, d_xyz_ptr: int, batch: int, d_labels_ptr: int) -> None:
        """Organised clouds (float32 [batch][H*W][3]) in device memory, the reference API's own input form."""
        _capi.check(_capi.lib().dpx_pipeline_submit_pcd(self._p, d_xyz_ptr, batch, d_labels_ptr))

    def sync(self) -> None:
        _capi.check(_capi.lib().dpx_pipeline_sync(self._p))
        self._keep.clear()

    def flush(self) -> None:
        """Launch what has been submitted so far without waiting (a partially filled group goes as it is)."""
        _capi.check(_capi.lib().dpx_pipeline_flush(self._p))

    def slot(self, k: int) -> "PlaneExtractor":
        """A non-owning PlaneExtractor view of slot k (result getters, profiling)."""
        h = _capi.lib().dpx_pipeline_slot(self._p, k)
        if not h:
            raise IndexError(k)
        ex = PlaneExtractor.__new__(PlaneExtractor)
        ex._h = C.c_void_p(h)
        ex.image_height, ex.image_width, ex.max_batch = self.image_height, self.image_width, self.batch * self.group
        nh, nv = C.c_int32(), C.c_int32()
        _capi.check(_capi.lib().dpx_grid_dims(ex._h, C.byref(nh), C.byref(nv)))
        ex.nh, ex.nv = nh.value, nv.value
        ex.config, ex.device = self.config, self.device
        ex._borrowed = True
        ex._owner = self  # the view's handle lives inside the pipeline: keep it alive as long as the view
        return ex

    def __del__(self):
        p = getattr(self, "_p", None)
        if p is not None and p.value:
            try:
                _capi.lib().dpx_pipeline_destroy(p)
            except Exception:
                pass
            self._p = C.c_void_p()
