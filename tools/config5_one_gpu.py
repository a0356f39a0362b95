"""BASELINE configs[4] on ONE GPU: 64 synthetic 3840x2160 frames in one call.  Region growing is latency-bound per
frame (one workgroup each), so 64 frames take about as long as 8: sharding this config over 8 GPUs buys no time."""
import os, sys, time
from pathlib import Path
ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
import numpy as np, torch
import deplex_amd as D, synth
from oracle import oracle as O
O.set_math("devmath")
h, w, nuniq = 2160, 3840, 4
uniq = synth.synth_batch(nuniq, h, w, seed0=0)
intr = synth.intrinsics_for(h, w)
dev = torch.device("cuda", 0)
ref = O.process_depth_batch(O.default_config(), uniq, *intr, threads=4)
for B in (8, 64):
    depth = np.concatenate([uniq] * (B // nuniq))
    d = torch.from_numpy(depth.view(np.int16)).to(dev)
    lab = torch.empty((B, h * w), dtype=torch.int32, device=dev)
    ex = D.PlaneExtractor(h, w, max_batch=B)
    st = torch.cuda.Stream()
    for rep in range(3):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        ex.process_depth_device(d.data_ptr(), B, *intr, lab.data_ptr(), st.cuda_stream)
        st.synchronize(); dt = time.perf_counter() - t0
    ok = all(np.array_equal(lab[k].cpu().numpy(), ref[k % nuniq]) for k in range(0, B, max(1, B // 8)))
    print(f"3840x2160 batch {B} on one GPU: {dt*1e3:.1f} ms -> {B/dt:.0f} frames/s; bit-exact: {ok}", flush=True)
    assert ok
    del ex, d, lab
