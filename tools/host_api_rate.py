"""End-to-end rate of the host-pointer entry points (depth frames in host memory -> results in host memory), PCIe
included: the pixel-label form (dpx_process_depth_batch, 0.6 MB in + 1.2 MB out per 640x480 frame), the cell-level form
(dpx_process_depth_batch_cells, 0.6 MB in + ~15 KB out) and the sharded entry point with two extractors on the one GPU.
usage: host_api_rate.py [batch]"""
import sys, time
from pathlib import Path
ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
import numpy as np
import deplex_amd as D, synth

B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
depth = synth.tum_stream(B)
K = synth.read_K(synth.TUM_K)
intr = tuple(float(v) for v in (K[0, 0], K[1, 1], K[0, 2], K[1, 2]))
N = 480 * 640


def rate(fn, reps=12):
    fn(); fn()
    best = 1e9
    for _ in range(reps):
        t0 = time.perf_counter(); fn(); best = min(best, time.perf_counter() - t0)
    return best


ex = D.PlaneExtractor(480, 640, max_batch=B)
pin_in = D.pinned_empty(depth.shape, np.uint16); pin_in[:] = depth
pin_out = D.pinned_empty((B, N), np.int32)
pin_cl = D.pinned_empty((B, ex.nv, ex.nh), np.int32)
pag_out = np.empty((B, N), np.int32)
for name, fn, out_bytes in (
        ("pixel labels, pageable buffers", lambda: ex.process_depth_batch(depth, *intr, out=pag_out), 4 * N),
        ("pixel labels, page-locked buffers", lambda: ex.process_depth_batch(pin_in, *intr, out=pin_out), 4 * N),
        ("cell labels + planes, pageable in", lambda: ex.process_depth_batch_cells(depth, *intr, out=pin_cl), 4 * ex.nv * ex.nh),
        ("cell labels + planes, page-locked in", lambda: ex.process_depth_batch_cells(pin_in, *intr, out=pin_cl), 4 * ex.nv * ex.nh),
        ("cell labels only, page-locked in", lambda: ex.process_depth_batch_cells(pin_in, *intr, out=pin_cl, planes=False), 4 * ex.nv * ex.nh)):
    dt = rate(fn)
    print(f"{name:40s}: {dt * 1e3:6.2f} ms per {B} frames = {B / dt:8.0f} frames/s; H2D {B * 2 * N / dt / 1e9:5.1f} GB/s, "
          f"D2H {B * out_bytes / dt / 1e9:5.2f} GB/s", flush=True)
sh = D.ShardedExtractor(480, 640, devices=[0, 0], max_batch=B // 2)
dt = rate(lambda: sh.process_depth_batch(pin_in, *intr, out=pin_out))
print(f"{'sharded [0, 0], pixel labels, pinned':40s}: {dt * 1e3:6.2f} ms = {B / dt:8.0f} frames/s", flush=True)
dt = rate(lambda: sh.process_depth_batch_cells(pin_in, *intr))
print(f"{'sharded [0, 0], cell labels + planes':40s}: {dt * 1e3:6.2f} ms = {B / dt:8.0f} frames/s", flush=True)

# streamed: host-buffer submissions of a pipeline (the slot's stream carries copy in, kernels, copy out; the copies of one
# slot overlap the kernels of the others)
R = 4
pins = [D.pinned_empty(depth.shape, np.uint16) for _ in range(R)]
for r in range(R):
    pins[r][:] = synth.tum_stream(B, seed0=B * r)
for slots in (2, 3, 4, 6):
    pipe = D.Pipeline(480, 640, D.Config(), batch=B, slots=slots, group=1)
    for k in range(slots):
        pipe.slot(k).prepare_intrinsics(*intr)
    for cells, shape in ((False, (B, N)), (True, (B, ex.nv, ex.nh))):
        outs = [D.pinned_empty(shape, np.int32) for _ in range(slots + 1)]
        n = 48

        def run():
            for i in range(n):
                pipe.submit_depth_host(pins[i % R], *intr, outs[i % len(outs)], cells=cells)
            pipe.sync()

        run()
        t0 = time.perf_counter(); run(); dt = (time.perf_counter() - t0) / n
        print(f"{'pipeline host submissions, ' + ('cell labels' if cells else 'pixel labels') + f', {slots} slots':48s}: {dt * 1e3:6.2f} ms per {B} frames = "
              f"{B / dt:8.0f} frames/s; H2D {B * 2 * N / dt / 1e9:5.1f} GB/s, D2H {outs[0].nbytes / dt / 1e9:5.2f} GB/s", flush=True)
    del pipe
