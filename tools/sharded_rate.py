"""Host -> host rate of ShardedExtractor (two shards on device 0) against single blocking calls and the plain streamed
pipeline: 640x480, chunks of 64 frames.  usage: sharded_rate.py [frames]"""
import os, sys, time
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
from pathlib import Path
ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
import numpy as np
import synth
import deplex_amd as D

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
base = synth.tum_stream(64)
K = synth.read_K(synth.TUM_K); intr = tuple(float(v) for v in (K[0, 0], K[1, 1], K[0, 2], K[1, 2]))
depth = D.pinned_empty((n, 480, 640), np.uint16)
for s in range(0, n, 64):
    depth[s:s + 64] = base[: min(64, n - s)]
out = D.pinned_empty((n, 480 * 640), np.int32)
for devices in ([0], [0, 0]):
    sh = D.ShardedExtractor(480, 640, D.Config(), devices=devices, max_batch=64, slots=3)
    for form in ("pixel", "cells"):
        best = 0.0
        for rep in range(4):
            t0 = time.perf_counter()
            if form == "pixel":
                sh.process_depth_batch(depth, *intr, out=out)
            else:
                sh.process_depth_batch_cells(depth, *intr, planes=False)
            dt = time.perf_counter() - t0
            if rep:
                best = max(best, n / dt)
        print(f"ShardedExtractor devices={devices} slots=3, {form} labels: {best:,.0f} frames/s host to host ({n} frames of 640x480)", flush=True)
    sh.close()
ex = D.PlaneExtractor(480, 640, max_batch=64)
best = 0.0
for rep in range(3):
    t0 = time.perf_counter()
    for s in range(0, n, 64):
        ex.process_depth_batch(depth[s:s + 64], *intr, out=out[s:s + 64])
    dt = time.perf_counter() - t0
    if rep:
        best = max(best, n / dt)
print(f"one blocking call per 64 frames (what ShardedExtractor did until round 3): {best:,.0f} frames/s", flush=True)
