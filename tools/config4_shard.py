"""BASELINE configs[3]: synthetic 640x480, batch 4096 sharded over 8 GPUs = 512 frames per GPU, no collectives.
One GPU's share: a single handle with max_batch 512 (one shot) and the streamed pipeline (8 submissions of 64)."""
import os, sys, time
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
from pathlib import Path
ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
import numpy as np, torch
import deplex_amd as D, synth
from deplex_amd.sharding import shard_range
from oracle import oracle as O
O.set_math("devmath")
h, w, total, world = 480, 640, 4096, 8
lo, hi = shard_range(total, 0, world)
B = hi - lo
uniq = synth.synth_batch(16, h, w, seed0=lo)
depth = np.concatenate([uniq] * (B // 16))
intr = synth.intrinsics_for(h, w)
dev = torch.device("cuda", 0)
d = torch.from_numpy(depth.view(np.int16)).to(dev)
lab = torch.empty((B, h * w), dtype=torch.int32, device=dev)
ex = D.PlaneExtractor(h, w, max_batch=B)
st = torch.cuda.Stream()
for rep in range(3):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    ex.process_depth_device(d.data_ptr(), B, *intr, lab.data_ptr(), st.cuda_stream)
    st.synchronize(); dt = time.perf_counter() - t0
print(f"rank 0 shard [{lo}, {hi}): one shot {dt*1e3:.2f} ms -> {B/dt:.0f} fps")
ref = O.process_depth_batch(O.default_config(), uniq, *intr, threads=8)
got = lab.cpu().numpy()
assert all(np.array_equal(got[k], ref[k % 16]) for k in range(B)), "mismatch"
pipe = D.Pipeline(h, w, D.Config(), batch=64, slots=4, group=8)
lab.fill_(-1)
for rep in range(3):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for i in range(B // 64):
        pipe.submit_depth(d[i * 64:(i + 1) * 64].data_ptr(), 64, *intr, lab[i * 64:(i + 1) * 64].data_ptr())
    pipe.sync(); dt = time.perf_counter() - t0
print(f"pipeline (8 x 64): {dt*1e3:.2f} ms -> {B/dt:.0f} fps; bit-exact: {all(np.array_equal(lab[k].cpu().numpy(), ref[k % 16]) for k in range(0, B, 7))}")
