"""Diagnostic: can the plane-fit launch (VALU-bound, no LDS) overlap with the region + label kernels of OTHER groups when
nothing orders them?  Runs the pipeline without its own plane fits (DPX_SKIP=F: real cell records from the first calls)
while an independent stream issues plane-fit group launches back to back, and compares with each running alone."""
import os as _os
_os.environ.setdefault("DPX_LIB_PATH", str(__import__("pathlib").Path(__file__).resolve().parents[1] / "deplex_amd" / "lib" / "variants" / "diag.so"))  # the -DDPX_DIAG build
import os, sys, threading, time
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
os.environ["DPX_SKIP"] = sys.argv[1] if len(sys.argv) > 1 else "F"
from pathlib import Path
ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
import numpy as np, torch
import deplex_amd as D
import synth
B, S, G = 64, 16, 8
K = synth.read_K(synth.TUM_K); intr = tuple(float(v) for v in (K[0, 0], K[1, 1], K[0, 2], K[1, 2]))
dev = torch.device("cuda", 0)
d_in = [torch.from_numpy(synth.tum_stream(B, seed0=r * B).view(np.int16)).to(dev) for r in range(8)]
d_out = torch.empty((S * G, B, 480 * 640), dtype=torch.int32, device=dev)
pipe = D.Pipeline(480, 640, D.Config(), batch=B, slots=S, group=G)
ex2 = D.PlaneExtractor(480, 640, max_batch=G * B)
st2 = torch.cuda.Stream(device=dev)
ptrs = [d.data_ptr() for d in d_in]
def run_pipe(steps):
    t = time.perf_counter()
    for i in range(steps): pipe.submit_depth(d_in[i % 8].data_ptr(), B, *intr, d_out[i % (S * G)].data_ptr())
    pipe.sync()
    return (time.perf_counter() - t) / steps * 1e6
def run_pf(iters, out):
    out.append(ex2.time_planefit_group(ptrs, 0, B, *intr, iters, st2.cuda_stream) * 1e3 / G)
run_pipe(512); run_pf(20, [])
steps = 2048
alone_pipe = run_pipe(steps)
o = []; run_pf(200, o); alone_pf = o[0]
iters = int(alone_pipe * steps / (alone_pf * G) * 1.5)
o = []
th = threading.Thread(target=run_pf, args=(iters, o)); th.start()
both_pipe = run_pipe(steps)
th.join()
print(f"skip={os.environ['DPX_SKIP']}: pipeline alone {alone_pipe:.1f} us/batch, plane fit alone {alone_pf:.1f} us/batch; together: pipeline {both_pipe:.1f}, "
      f"plane fit {o[0]:.1f} us/batch (the plane-fit series ran {iters} launches, longer than the pipeline)")
