"""BASELINE configs 3 and 5 (1280x960 batch 256; 3840x2160 batch 8 per GPU): functional check + timing."""
import sys, time, os
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
from pathlib import Path
ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
import numpy as np, torch
import deplex_amd as D, synth
from oracle import oracle as O
O.set_math("devmath")
dev = torch.device("cuda", 0)
for (h, w, B, nuniq) in [(960, 1280, 256, 8), (2160, 3840, 8, 2)]:
    t = time.time()
    uniq = synth.synth_batch(nuniq, h, w, seed0=0)
    depth = np.concatenate([uniq] * (B // nuniq))
    print(f"{w}x{h}: generated {nuniq} frames in {time.time()-t:.1f}s", flush=True)
    intr = synth.intrinsics_for(h, w)
    d = torch.from_numpy(depth.view(np.int16)).to(dev)
    lab = torch.empty((B, h * w), dtype=torch.int32, device=dev)
    ex = D.PlaneExtractor(h, w, max_batch=B)
    st = torch.cuda.Stream()
    ex.set_profiling(True)
    for rep in range(3):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        ex.process_depth_device(d.data_ptr(), B, *intr, lab.data_ptr(), st.cuda_stream)
        st.synchronize(); dt = time.perf_counter() - t0
    print(f"  batch {B}: {dt*1e3:.2f} ms -> {B/dt:.0f} fps; kernels {ex.kernel_ms()}", flush=True)
    got = lab[:nuniq].cpu().numpy()
    t = time.time()
    ref = O.process_depth_batch(O.default_config(), uniq, *intr, threads=min(nuniq, 8))
    print(f"  oracle {nuniq} frames {time.time()-t:.1f}s; bit-exact: {np.array_equal(got, ref)}; planes {[int(x.max()) for x in got]}", flush=True)
    assert np.array_equal(got, ref), "label maps differ from the oracle"
    assert np.array_equal(lab[nuniq:2*nuniq].cpu().numpy(), got)
