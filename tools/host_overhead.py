"""Host-side submit cost vs GPU time for the single-stream handles and the 3-stage pipeline (diagnostic)."""
import os, sys, time
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
from pathlib import Path
ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
import numpy as np, torch
import deplex_amd as D, synth
B = 64
depth = synth.tum_stream(B)
K = synth.read_K(synth.TUM_K); intr = tuple(float(v) for v in (K[0, 0], K[1, 1], K[0, 2], K[1, 2]))
dev = torch.device("cuda", 0)
d = torch.from_numpy(depth.view(np.int16)).to(dev)
S = 16
lab = torch.empty((S * 4, B, 480 * 640), dtype=torch.int32, device=dev)
def run(name, submit, sync, n=256):
    for i in range(32): submit(i)
    sync()
    t0 = time.perf_counter()
    for i in range(n): submit(i)
    t1 = time.perf_counter()
    sync()
    t2 = time.perf_counter()
    print(f"{name}: host {1e6*(t1-t0)/n:.1f} us/submit, total {1e6*(t2-t0)/n:.1f} us/step -> {B*n/(t2-t0):.0f} fps")
for prof in (False, True):
    exs = [D.PlaneExtractor(480, 640, max_batch=B) for _ in range(S)]
    streams = [torch.cuda.Stream(device=dev) for _ in range(S)]
    for e in exs: e.prepare_intrinsics(*intr); e.set_profiling(prof)
    run(f"handles x{S} prof={prof}", lambda i: exs[i % S].process_depth_device(d.data_ptr(), B, *intr, lab[i % len(lab)].data_ptr(), streams[i % S].cuda_stream), lambda: torch.cuda.synchronize())
    del exs
    for rs in (1, 4):
        pipe = D.Pipeline(480, 640, batch=B, slots=S, group=rs)
        for k in range(S): pipe.slot(k).prepare_intrinsics(*intr); pipe.slot(k).set_profiling(prof)
        run(f"pipeline slots={S} group={rs} prof={prof}", lambda i: pipe.submit_depth(d.data_ptr(), B, *intr, lab[i % len(lab)].data_ptr()), lambda: (pipe.sync(), torch.cuda.synchronize()))
        del pipe
