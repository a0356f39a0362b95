"""Ad-hoc: a large batch of tiled 4K frames through one handle, then several handles at once: every copy of a frame must
give the same label map, planes and stats.  usage: cores_stress.py [B] [H W]"""
import os, sys, threading
from pathlib import Path
ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
import numpy as np
import synth
import deplex_amd as D

B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
h, w = (int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (2160, 3840)
nu = 4
uniq = synth.synth_batch(nu, h, w, seed0=0)
depth = np.concatenate([uniq] * (B // nu))
intr = synth.intrinsics_for(h, w)

def check(tag, ex, labels):
    bad = 0
    for k in range(B):
        ref = k % nu
        if not np.array_equal(labels[k], labels[ref]):
            d = int((labels[k] != labels[ref]).sum())
            st, sr = ex.frame_stats(k), ex.frame_stats(ref)
            print(f"{tag}: frame {k} differs from its copy {ref} in {d} pixels; stats {tuple(int(v) for v in st)} vs {tuple(int(v) for v in sr)}", flush=True)
            bad += 1
    print(f"{tag}: {bad} of {B} frames differ", flush=True)
    return bad

ex = D.PlaneExtractor(h, w, max_batch=B)
print(ex.options())
bad = 0
for rep in range(3):
    bad += check(f"single handle rep {rep}", ex, ex.process_depth_batch(depth, *intr))
exs = [D.PlaneExtractor(h, w, max_batch=B) for _ in range(3)]
outs = [None] * 3
def run(i):
    outs[i] = exs[i].process_depth_batch(depth, *intr)
for rep in range(2):
    ths = [threading.Thread(target=run, args=(i,)) for i in range(3)]
    [t.start() for t in ths]; [t.join() for t in ths]
    for i in range(3):
        bad += check(f"concurrent handle {i} rep {rep}", exs[i], outs[i])
sys.exit(1 if bad else 0)
