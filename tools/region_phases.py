"""Diagnostic: per-phase shader-clock cycles of the region-growing kernel (s_memtime: the shader clock, 2.4 GHz under load on MI355X; s_memrealtime is the 100 MHz one).
usage: region_phases.py [H W batch]   (default: the bench workload, 64 TUM-like 640x480 frames)"""
import os as _os
_os.environ.setdefault("DPX_LIB_PATH", str(__import__("pathlib").Path(__file__).resolve().parents[1] / "deplex_amd" / "lib" / "variants" / "diag.so"))  # the -DDPX_DIAG build
import sys
from pathlib import Path
import numpy as np
ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
import deplex_amd as D
from deplex_amd import _capi
import synth

if len(sys.argv) >= 4:
    h, w, B = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
    depth = synth.synth_batch(B, h, w, seed0=0)
    intr = synth.intrinsics_for(h, w)
else:
    h, w, B = 480, 640, 64
    depth = synth.tum_stream(64)
    K = synth.read_K(synth.TUM_K)
    intr = tuple(float(v) for v in (K[0, 0], K[1, 1], K[0, 2], K[1, 2]))
ex = D.PlaneExtractor(h, w, max_batch=B)
_capi.check(_capi.lib().dpx_debug_region_phases(ex._h, 1, 0, None))
for rep in range(2):
    ex.process_depth_batch(depth, *intr)
names = ["P0 load", "P1 seeds+bfs", "P2 accum", "P3 fits", "P4 labels", "P5 merge", "P6 final", "total",
         "sum argmax", "sum seeds+spec", "sum bfs", "rounds|spec_grows<<32", "bfs steps", "grows", "cands", "activations",
         "cyc scans", "cyc spec", "lane scan steps", "wave scan steps", "spec pops",
         "stops: large grow", "stops: bin touched", "stops: cell taken",
         "cyc claims", "cyc resolve+commit", "single-event rounds", "cyc single-event rounds", "rounds with events from several chunks",
         "grows given up beyond the 64th item", "rounds that continued a sweep without an arg-max", "cyc CSR merge walk (wave 0)"]
for f in (range(B) if B <= 8 else sorted(set((0, 1, B - 1)))):
    out = np.zeros(32, dtype=np.uint64)
    _capi.check(_capi.lib().dpx_debug_region_phases(ex._h, 1, f, out.ctypes.data))
    vals = [int(v) for v in out]
    txt = ", ".join(f"{n}={v}" for n, v in zip(names, vals) if not n.startswith("rounds"))
    print(f"frame {f}: {txt}, sweep rounds={vals[11] & 0xFFFFFFFF}, grows committed by rounds={vals[11] >> 32}", "stats", ex.frame_stats(f))
