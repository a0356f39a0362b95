#!/usr/bin/env python3
"""Streamed rate of the pipeline with pixel label maps against cell-level output (dpx_pipeline_submit_depth_cells):
the TUM stream at 640x480 x 64 and the synthetic 1280x960 x 64 stream, 4 groups of 8 batches in flight, inputs and
outputs resident in HBM.  Prints one JSON line per workload.

    python tools/cells_stream_rate.py [--steps 1500]
"""
import argparse
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=1500)
    args = ap.parse_args()
    import torch

    import deplex_amd as D
    from tests import synth

    dev = torch.device("cuda", 0)
    for name, H, W, B, steps in (("tum_stream 640x480x64", 480, 640, 64, args.steps), ("synth 1280x960x64", 960, 1280, 64, args.steps // 4)):
        intr = synth.intrinsics_for(H, W)
        R, S, G = 8, 4, 8
        if H == 480:
            frames = [synth.tum_stream(B, seed0=B * r) for r in range(R)]
        else:
            frames = [np.tile(synth.synth_batch(8, H, W, seed0=1000 + 97 * r), (B // 8, 1, 1)) for r in range(R)]
        ins = [torch.from_numpy(np.ascontiguousarray(f).view(np.int16)).to(dev) for f in frames]
        pipe = D.Pipeline(H, W, D.Config(), batch=B, slots=S, group=G)
        ex = pipe.slot(0)
        for k in range(S):
            pipe.slot(k).prepare_intrinsics(*intr)
        NB = S * G
        px = torch.empty((NB, B, H * W), dtype=torch.int32, device=dev)
        cl = torch.empty((NB, B, ex.nv, ex.nh), dtype=torch.int32, device=dev)
        rec = {"workload": name, "groups_in_flight": S, "batches_per_launch": G}
        for kind, fn, out in (("pixel_labels", pipe.submit_depth, px), ("cell_labels", pipe.submit_depth_cells, cl)):
            for i in range(128):
                fn(ins[i % R].data_ptr(), B, *intr, out[i % NB].data_ptr())
            pipe.sync(); torch.cuda.synchronize(dev)
            t0 = time.perf_counter()
            for i in range(steps):
                fn(ins[i % R].data_ptr(), B, *intr, out[i % NB].data_ptr())
            pipe.sync(); torch.cuda.synchronize(dev)
            dt = time.perf_counter() - t0
            rec[kind] = {"frames_per_s": round(B * steps / dt, 1), "ms_per_step": round(dt / steps * 1e3, 4)}
        # same results in both forms
        a = ex.expand_cell_labels(cl[:R].cpu().numpy().reshape(-1, ex.nv, ex.nh))
        rec["cell_labels_expand_to_pixel_labels"] = bool(np.array_equal(a, px[:R].cpu().numpy().reshape(-1, H * W)))
        rec["ratio"] = round(rec["cell_labels"]["frames_per_s"] / rec["pixel_labels"]["frames_per_s"], 3)
        print(json.dumps(rec), flush=True)
        del pipe, px, cl


if __name__ == "__main__":
    main()
