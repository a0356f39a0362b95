"""Create / use / destroy extractors, pipelines and sharded extractors in a loop and watch device and host memory."""
import sys, os, gc
from pathlib import Path
ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
import numpy as np, torch, psutil
import deplex_amd as D, synth
depth = synth.synth_batch(4, 240, 320, seed0=1)
intr = synth.intrinsics_for(240, 320)
dev = torch.device("cuda", 0)
d_in = torch.from_numpy(depth.view(np.int16)).to(dev)
d_out = torch.empty((4, 240 * 320), dtype=torch.int32, device=dev)
proc = psutil.Process()
def snap():
    torch.cuda.synchronize(); gc.collect()
    free, total = torch.cuda.mem_get_info()
    return (total - free) / 2**20, proc.memory_info().rss / 2**20
base = None
for it in range(6):
    for _ in range(40):
        ex = D.PlaneExtractor(240, 320, max_batch=4)
        ex.process_depth_batch(depth, *intr)
        ex.process_depth_batch_cells(depth, *intr)
        ex.close()
        p = D.Pipeline(240, 320, batch=4, slots=3, group=2)
        for k in range(5): p.submit_depth(d_in.data_ptr(), 4, *intr, d_out.data_ptr())
        p.sync(); del p
        sh = D.ShardedExtractor(240, 320, devices=[0, 0], max_batch=2)
        sh.process_depth_batch(depth, *intr); sh.close()
        big = D.PlaneExtractor(960, 1280, max_batch=2); big.close()
    s = snap()
    if base is None: base = s
    print(f"round {it}: device {s[0]:.0f} MiB (+{s[0]-base[0]:.0f}), host rss {s[1]:.0f} MiB (+{s[1]-base[1]:.0f})", flush=True)
assert s[0] - base[0] < 64 and s[1] - base[1] < 256, "memory grows"
print("no growth")
