"""Frame sharding across GPUs.  Frames are independent (an extractor holds only per-call scratch, reference
plane_extractor.cpp:281,428), so a batch shards into contiguous frame ranges, one per device, with no data-path
collective: one extractor and one host thread per device, results written to disjoint slices of
one page-locked host buffer.  `shard_range` is also what the one-process-per-GPU form uses (bench.py --gpus N, tests/dist_worker.py).
"""
from __future__ import annotations

import ctypes as C
import weakref
from concurrent.futures import ThreadPoolExecutor
from typing import Optional, Sequence

import numpy as np

from . import _capi
from .plane_extraction import Config, PlaneExtractor


def shard_range(n_frames: int, rank: int, world: int) -> tuple:
    """frames[g*B/G : (g+1)*B/G] for rank g of G; covers 0..n_frames exactly once, sizes differ by at most 1."""
    if world < 1 or not (0 <= rank < world) or n_frames < 0:
        raise ValueError("bad shard arguments")
    return (n_frames * rank) // world, (n_frames * (rank + 1)) // world


def pinned_empty(shape, dtype=np.int32) -> np.ndarray:
    """An uninitialised page-locked host array (hipHostMalloc through the C-ABI; freed with the array)."""
    dt = np.dtype(dtype)
    n = int(np.prod(shape, dtype=np.int64)) * dt.itemsize
    p = _capi.lib().dpx_host_alloc(max(n, 1))
    if not p:
        raise _capi.DeplexError(_capi.DPX_ERR_HIP, _capi.lib().dpx_last_error().decode(errors="replace"))
    buf = (C.c_char * max(n, 1)).from_address(p)
    arr = np.frombuffer(buf, dtype=dt, count=int(np.prod(shape, dtype=np.int64))).reshape(shape)
    weakref.finalize(buf, _capi.lib().dpx_host_free, p)  # (the array keeps `buf` alive through .base)
    return arr


class ShardedExtractor:
    """deplex::PlaneExtractor over several GPUs of one node for callers that hold a batch of HOST frames.

    `devices` lists HIP device ids (a device may appear more than once: that many shards share it).  process_depth_batch()
    splits the batch into contiguous frame ranges (shard_range), one per entry of `devices`; each range is fed by its own
    host thread, in chunks of `max_batch` frames, through the entry's `Pipeline` of `slots` slots with host-buffer
    submissions (dpx_pipeline_submit_depth_host / _cells_host): every slot's stream carries copy in, kernels, copy out,
    so one chunk copies in while another computes and a third copies out — the streamed rates of DESIGN.md section 6
    (37 k pixel-label / 90 k cell-label frames/s per device at 640x480) instead of one blocking call at a time (26 k /
    56 k).  A last chunk shorter than `max_batch`, and the cell-level form WITH plane records, go through the entry's plain
    extractor.  Results land in the caller's (ideally page-locked: pinned_empty) output at the range's offset.  There is
    no exchange between devices.  HIP maps streams to GPU_MAX_HW_QUEUES hardware queues (default 4): set it to at least
    `slots` per device before HIP initialises (the library warns once otherwise).  Measured on one GPU only in this
    repository (devices=[0, 0]); a node with several GPUs has not been available to the builder.
    """

    def __init__(self, image_height: int, image_width: int, config: Optional[Config] = None, *,
                 devices: Sequence[int] = (0,), max_batch: int = 64, slots: int = 3):
        from .plane_extraction import Pipeline
        if not devices:
            raise ValueError("ShardedExtractor needs at least one device")
        self.devices = [int(d) for d in devices]
        self.max_batch, self.slots = int(max_batch), max(1, int(slots))
        self.image_height, self.image_width = int(image_height), int(image_width)
        self.config = config if config is not None else Config()
        self.extractors = [PlaneExtractor(image_height, image_width, self.config, max_batch=self.max_batch, device=d)
                           for d in self.devices]
        self.pipelines = [Pipeline(image_height, image_width, self.config, batch=self.max_batch, slots=self.slots, group=1, device=d)
                          for d in self.devices]
        self.nh, self.nv = self.extractors[0].nh, self.extractors[0].nv
        self._pool = ThreadPoolExecutor(max_workers=len(self.devices), thread_name_prefix="dpx-shard")

    def shards(self, n_frames: int):
        return [shard_range(n_frames, g, len(self.devices)) for g in range(len(self.devices))]

    def _run(self, depth, worker):
        futs = [self._pool.submit(worker, g, lo, hi, depth) for g, (lo, hi) in enumerate(self.shards(depth.shape[0])) if hi > lo]
        for f in futs:
            f.result()  # (re-raises a shard's DeplexError)
        return depth.shape[0]

    def _stream(self, g, lo, hi, d, fx, fy, cx, cy, out, cells):
        """Shard g: whole chunks through its pipeline, the remainder through its extractor."""
        pipe, ex, mb = self.pipelines[g], self.extractors[g], self.max_batch
        s = lo
        while s + mb <= hi:
            pipe.submit_depth_host(d[s:s + mb], fx, fy, cx, cy, out[s:s + mb], cells=cells)
            s += mb
        if s < hi:
            if cells:
                ex.process_depth_batch_cells(d[s:hi], fx, fy, cx, cy, out=out[s:hi], planes=False)
            else:
                ex.process_depth_batch(d[s:hi], fx, fy, cx, cy, out=out[s:hi])
        pipe.sync()

    def process_depth_batch(self, depth, fx, fy, cx, cy, out: Optional[np.ndarray] = None) -> np.ndarray:
        """depth (B, H, W) uint16 in host memory -> labels (B, H*W) int32 (`out`, or a new page-locked array)."""
        depth = np.ascontiguousarray(depth, dtype=np.uint16)
        if depth.ndim == 2:
            depth = depth[None]
        b, n = depth.shape[0], self.image_height * self.image_width
        if depth.shape[1:] != (self.image_height, self.image_width):
            raise TypeError(f"depth must have shape (B, {self.image_height}, {self.image_width})")
        if out is None:
            out = pinned_empty((b, n), np.int32)
        elif out.dtype != np.int32 or out.size != b * n or not out.flags.c_contiguous:
            raise TypeError("out must be a C-contiguous int32 array of B * H * W elements")
        out2 = out.reshape(b, n)
        self._run(depth, lambda g, lo, hi, d: self._stream(g, lo, hi, d, fx, fy, cx, cy, out2, False))
        return out2

    def process_depth_batch_cells(self, depth, fx, fy, cx, cy, planes: bool = True):
        """Cell-level output for the whole batch: (B, nv, nh) labels and — with planes=True — a list of B plane-record arrays
        (PlaneExtractor.process_depth_batch_cells, one blocking call per chunk: the records come out of the handle's
        staging); planes=False streams like process_depth_batch and returns (labels, None)."""
        depth = np.ascontiguousarray(depth, dtype=np.uint16)
        if depth.ndim == 2:
            depth = depth[None]
        if depth.shape[1:] != (self.image_height, self.image_width):
            raise TypeError(f"depth must have shape (B, {self.image_height}, {self.image_width})")
        b = depth.shape[0]
        out = pinned_empty((b, self.nv, self.nh), np.int32)
        if not planes:
            self._run(depth, lambda g, lo, hi, d: self._stream(g, lo, hi, d, fx, fy, cx, cy, out, True))
            return out, None
        recs = [None] * b

        def worker(g, lo, hi, d):
            ex = self.extractors[g]
            for s in range(lo, hi, self.max_batch):
                e = min(s + self.max_batch, hi)
                _, pl = ex.process_depth_batch_cells(d[s:e], fx, fy, cx, cy, out=out[s:e])
                recs[s:e] = pl

        self._run(depth, worker)
        return out, recs

    def close(self):
        self._pool.shutdown(wait=True)
        self.pipelines = []
        for ex in self.extractors:
            ex.close()
        self.extractors = []

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
