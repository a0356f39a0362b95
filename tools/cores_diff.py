"""Ad-hoc: which intermediate array of the core path differs between copies of the same frame in one batch."""
import sys
from pathlib import Path
ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
import numpy as np
import synth
import deplex_amd as D
from deplex_amd import _capi

B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
h, w = 2160, 3840
nu = 4
uniq = synth.synth_batch(nu, h, w, seed0=0)
depth = np.concatenate([uniq] * (B // nu))
intr = synth.intrinsics_for(h, w)
ex = D.PlaneExtractor(h, w, max_batch=B)
C = ex.nh * ex.nv
def rd(which, frame, n, dt=np.uint32):
    out = np.zeros(n, dtype=dt)
    got = _capi.lib().dpx_debug_read_cores(ex._h, which, frame, out.ctypes.data, out.nbytes)
    assert got > 0, got
    return out
names = {0: "parent", 1: "csize", 2: "npairs", 7: "histg", 8: "cursors", 9: "header", 13: "meta"}
for rep in range(6):
    labels = ex.process_depth_batch(depth, *intr)
    nbad = 0
    for k in range(nu, B):
        ref = k % nu
        for which, nm in names.items():
            n = {7: 400, 8: 4, 9: 48}.get(which, C)
            a, b = rd(which, k, n), rd(which, ref, n)
            if not np.array_equal(a, b):
                idx = np.nonzero(a != b)[0]
                print(f"rep {rep} frame {k} vs {ref}: {nm} differs at {len(idx)} entries, first {idx[:6]}: {a[idx[:6]]} vs {b[idx[:6]]}", flush=True)
                nbad += 1
        # core partition compare via sorted unique root sets
    print(f"rep {rep}: {nbad} array differences", flush=True)
