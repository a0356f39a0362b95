"""PCIe-inclusive rate of the host-pointer entry point dpx_process_depth_batch (H2D depth + kernels + D2H labels),
pageable vs pinned host buffers, next to the device-resident rate."""
import os, sys, time
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
from pathlib import Path
ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
import numpy as np, torch
import deplex_amd as D, synth
from deplex_amd import _capi

B = 64
depth = synth.tum_stream(B)
K = synth.read_K(synth.TUM_K)
intr = tuple(float(v) for v in (K[0, 0], K[1, 1], K[0, 2], K[1, 2]))
ex = D.PlaneExtractor(480, 640, max_batch=B)
L = _capi.lib()
def run(name, d_host, l_host, n=20):
    for _ in range(3):
        _capi.check(L.dpx_process_depth_batch(ex._h, d_host.ctypes.data if hasattr(d_host, "ctypes") else d_host.data_ptr(), B, *intr,
                                              l_host.ctypes.data if hasattr(l_host, "ctypes") else l_host.data_ptr()))
    t0 = time.perf_counter()
    for _ in range(n):
        _capi.check(L.dpx_process_depth_batch(ex._h, d_host.ctypes.data if hasattr(d_host, "ctypes") else d_host.data_ptr(), B, *intr,
                                              l_host.ctypes.data if hasattr(l_host, "ctypes") else l_host.data_ptr()))
    dt = (time.perf_counter() - t0) / n
    by = B * 480 * 640 * 6
    print(f"{name}: {dt*1e3:.2f} ms per batch of {B} -> {B/dt:.0f} fps, {by/dt/1e9:.1f} GB/s over PCIe (2 B in + 4 B out per pixel)")
labels = np.empty((B, 480 * 640), dtype=np.int32)
run("pageable host buffers", depth, labels)
dp = torch.from_numpy(depth.view(np.int16)).pin_memory()
lp = torch.empty((B, 480 * 640), dtype=torch.int32).pin_memory()
run("pinned host buffers", dp, lp)
assert np.array_equal(lp.numpy(), labels)
