"""Analysis aid (CPU only, oracle): the sequence of grows of createPlaneSegments (plane_extractor.cpp:297-347) on
synthetic frames — bin, seed, size per grow — to see how long the runs of one-cell ("isolated") grows are and how the
most frequent bin moves.  usage: grow_trace.py H W seed [seed ...]"""
import os, sys, tempfile
from pathlib import Path
import numpy as np
ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
import synth
from oracle import oracle as O

h, w = int(sys.argv[1]), int(sys.argv[2])
O.set_math("devmath")
for seed in map(int, sys.argv[3:]):
    depth = synth.synth_depth(h, w, seed=seed)
    path = tempfile.mktemp(prefix="growtrace_")
    os.environ["ORC_GROW_TRACE"] = path
    O.process_depth_batch(O.default_config(), depth[None], *synth.intrinsics_for(h, w), threads=1)
    del os.environ["ORC_GROW_TRACE"]
    t = np.loadtxt(path, dtype=np.int64).reshape(-1, 3)
    os.unlink(path)
    size = t[:, 2]
    iso = size == 1
    # runs of consecutive one-cell grows
    runs = []
    r = 0
    for v in iso:
        if v:
            r += 1
        else:
            if r: runs.append(r)
            r = 0
    if r: runs.append(r)
    runs = np.array(runs) if runs else np.zeros(1, dtype=np.int64)
    binchg = int((t[1:, 0] != t[:-1, 0]).sum())
    print(f"seed {seed}: grows {len(t)}, one-cell {int(iso.sum())}, <4 cells {int((size < 4).sum())}, >=64 cells {int((size >= 64).sum())}, "
          f"activations {int(size.sum())}, bin changes {binchg}; one-cell runs: n {len(runs)}, mean {runs.mean():.1f}, "
          f"median {np.median(runs):.0f}, max {runs.max()}, cells in runs >= 16: {int(runs[runs >= 16].sum())}")
    # where in the sequence the one-cell grows sit (deciles)
    dec = np.array_split(iso, 10)
    print("   one-cell share per decile of the sequence:", " ".join(f"{d.mean():.2f}" for d in dec))
    print("   size histogram:", {k: int(((size >= lo) & (size < hi)).sum()) for k, lo, hi in
                                  [("1", 1, 2), ("2-3", 2, 4), ("4-15", 4, 16), ("16-63", 16, 64), ("64-1023", 64, 1024), ("1024+", 1024, 1 << 30)]})
