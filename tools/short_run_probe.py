"""The driver's protocol (5 warm-up + 20 timed steps of 64 TUM frames) taken apart: host time of the submit loop vs the
whole timed region, for one submit call per step and for one call per group (dpx_pipeline_submit_depth_many), over a few
pipeline shapes.  usage: short_run_probe.py"""
import os, sys, time
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
from pathlib import Path
ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
import numpy as np, torch
import deplex_amd as D, synth
B, STEPS, WARM = 64, 20, 5
K = synth.read_K(synth.TUM_K); intr = tuple(float(v) for v in (K[0, 0], K[1, 1], K[0, 2], K[1, 2]))
dev = torch.device("cuda", 0)
ins = [torch.from_numpy(synth.tum_stream(B, seed0=r * B).view(np.int16)).to(dev) for r in range(8)]
for S, G in ((16, 5), (4, 5), (8, 5), (16, 4), (16, 10), (2, 10), (1, 20), (16, 2)):
    pipe = D.Pipeline(480, 640, batch=B, slots=S, group=G)
    for k in range(S): pipe.slot(k).prepare_intrinsics(*intr)
    NB = S * G
    lab = torch.empty((min(NB, 40), B, 480 * 640), dtype=torch.int32, device=dev)
    ip = [t.data_ptr() for t in ins]; op = [lab[k % len(lab)].data_ptr() for k in range(NB)]
    res = {}
    for mode in ("per step", "per group"):
        best = None
        for rep in range(5):
            for i in range(WARM): pipe.submit_depth(ip[i % 8], B, *intr, op[i % NB])
            pipe.sync(); torch.cuda.synchronize()
            t0 = time.perf_counter()
            if mode == "per step":
                for i in range(STEPS): pipe.submit_depth(ip[i % 8], B, *intr, op[i % NB])
            else:
                for g0 in range(0, STEPS, G):
                    idx = range(g0, min(g0 + G, STEPS))
                    pipe.submit_depth_many([ip[i % 8] for i in idx], B, *intr, [op[i % NB] for i in idx])
            t1 = time.perf_counter()
            pipe.sync(); torch.cuda.synchronize()
            t2 = time.perf_counter()
            if best is None or t2 - t0 < best[1]: best = (t1 - t0, t2 - t0)
        res[mode] = best
    print(f"slots {S:2d} x group {G:2d}: " + "; ".join(f"{m}: submit loop {1e3 * a:.3f} ms, total {1e3 * b:.3f} ms = {B * STEPS / b / 1e6:.3f} M frames/s" for m, (a, b) in res.items()), flush=True)
    del pipe, lab
