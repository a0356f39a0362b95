"""Region-growing launch times (HIP events, production kernels) for the large-frame configs: 3840x2160 x 8 (4 distinct
frames), 1280x960 x 256 (8 distinct), 640x480 x 512 (16 distinct).  usage: region_time.py [4k|1280|640 ...]"""
import os, sys, time
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
from pathlib import Path
ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
import numpy as np, torch
import deplex_amd as D, synth
CFG = {"4k": (2160, 3840, 8, 4), "4k64": (2160, 3840, 64, 4), "4k64noisy": (2160, 3840, 64, 1, 0), "4k64clean": (2160, 3840, 64, 1, 3), "1280": (960, 1280, 256, 8), "640": (480, 640, 512, 16)}
dev = torch.device("cuda", 0)
for name in (sys.argv[1:] or ["4k", "1280", "640"]):
    h, w, B, nuniq = CFG[name][:4]
    uniq = synth.synth_batch(nuniq, h, w, seed0=CFG[name][4] if len(CFG[name]) > 4 else 0)
    d = torch.from_numpy(uniq.view(np.int16)).to(dev).repeat(B // nuniq, 1, 1)
    lab = torch.empty((B, h * w), dtype=torch.int32, device=dev)
    ex = D.PlaneExtractor(h, w, max_batch=B)
    st = torch.cuda.Stream()
    ex.set_profiling(True)
    intr = synth.intrinsics_for(h, w)
    best = None
    for rep in range(4):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        ex.process_depth_device(d.data_ptr(), B, *intr, lab.data_ptr(), st.cuda_stream)
        st.synchronize(); dt = (time.perf_counter() - t0) * 1e3
        km = ex.kernel_ms()
        if rep and (best is None or dt < best[0]):
            best = (dt, km)
    print(f"{name}: {w}x{h} x {B}: call {best[0]:.2f} ms; kernels " + ", ".join(f"{k} {v:.3f}" for k, v in best[1].items()), flush=True)
