"""Diagnostic: shader-clock cycles of the core-based region path's launches per frame.  usage: cores_phases.py H W batch [quilt]"""
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

h, w, B = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
depth = synth.synth_batch(B, h, w, seed0=0) if len(sys.argv) < 5 else np.stack([synth.quilt_depth(h, w, seed=s) for s in range(B)])
intr = synth.intrinsics_for(h, w)
ex = D.PlaneExtractor(h, w, max_batch=B)
print("options", ex.options())
_capi.check(_capi.lib().dpx_debug_region_phases(ex._h, 1, 0, None))
for rep in range(2):
    ex.process_depth_batch(depth, *intr)
tn = ["P0 load", "P1", "P2 accum", "P3 fits", "P4 labels", "P5 merge", "P6 final", "tail total"]
for f in range(B):
    a = np.zeros(32, dtype=np.uint64); b = np.zeros(32, dtype=np.uint64)
    _capi.check(_capi.lib().dpx_debug_region_phases(ex._h, 1, f, a.ctypes.data))
    _capi.check(_capi.lib().dpx_debug_cores_phases(ex._h, f, b.ctypes.data))
    a = [int(v) for v in a]; b = [int(v) for v in b]
    print(f"frame {f}: sequence={b[1]} (setup {b[0]}) [events {b[3]} seeds {b[4]} items+claims {b[5]} walk {b[6]} commit {b[7]} wide {b[8]}] order(max wave)={b[2]} rounds={b[11]} grows={b[13]} cands={b[14]} (wave-ordered {b[30] & 0xFFFFFFFF}, workgroup-ordered {b[30] >> 32}) activations={b[15]} pairs={b[29]} window waits={b[19]} (helper: {b[24]} batches, {b[26]} requests, {b[25]} cycles) wide={b[21]} skipped={b[22]} irregular={b[23]} multi-rounds={b[28]} | tail: "
          + ", ".join(f"{n}={v}" for n, v in zip(tn, a[:8])) + f", merge walk={a[31]}", "stats", ex.frame_stats(f))
