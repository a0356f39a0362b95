"""Census of where the two libm flavours of the oracle disagree (CPU only, zero GPU minutes).

The reference links glibc's double-precision atan2 / cos / sin / sqrt (libs/dsyev/src/dsyevc3.c:62-72) and acos / atan2 /
sqrt (cpp/deplex/src/deplex/normals_histogram.cpp:35-37).  The device cannot call glibc; it runs the plain-arithmetic
implementations of deplex_amd/csrc/dpx_math.h, and the oracle can run either ("glibc" = what the reference computes,
"devmath" = what the device computes, bit for bit).  HIP == oracle(devmath) is what the GPU tests prove; this tool counts
how often oracle(devmath) != oracle(glibc), i.e. how far "bit-exact vs the device-maths oracle" is from "bit-exact vs a
glibc build of the reference": per frame any label difference, per cell a different histogram bin / planar bit / QL
path, per plane the largest parameter deviation.

usage: flavour_census.py [frames (default 10500)] [workers (default: all cores)] [--json out.json]
"""
import json
import multiprocessing as mp
import os
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))

# (name, share of the frames, generator kwargs)
FAMILIES = [
    ("synthetic 640x480 P10 noisy", 0.30, dict(h=480, w=640, P=10, kw={})),
    ("synthetic 320x240 P10 noisy", 0.15, dict(h=240, w=320, P=10, kw={})),
    ("synthetic 640x480 P10 noise-free (QL fallbacks, exact ties)", 0.12, dict(h=480, w=640, P=10, kw={"noise": False})),
    ("synthetic 640x480 P10 TUM scale (x5)", 0.10, dict(h=480, w=640, P=10, kw={"scale": 5.0})),
    ("synthetic 320x240 P4 noisy", 0.08, dict(h=240, w=320, P=4, kw={})),
    ("synthetic 320x240 P4 noise-free", 0.05, dict(h=240, w=320, P=4, kw={"noise": False, "dropout": False})),
    ("TUM stream variants P10 (reference frame + seeded noise / shifts)", 0.20, dict(tum=True, P=10)),
]


def work(args):
    fam_idx, seeds = args
    import synth
    from oracle import oracle as O
    name, _, g = FAMILIES[fam_idx]
    cfg = O.default_config(patch_size=g["P"])
    out = dict(frames=0, frames_label_diff=0, px_label_diff=0, cells=0, planar_cells=0, cells_bin_diff=0, cells_planar_diff=0,
               cells_ql_diff=0, frames_plane_count_diff=0, planes=0, max_plane_dev=0.0, frames_stats_diff=0)
    if g.get("tum"):
        K = synth.read_K(synth.TUM_K)
        intr = tuple(float(v) for v in (K[0, 0], K[1, 1], K[0, 2], K[1, 2]))
        h, w = 480, 640
    else:
        h, w = g["h"], g["w"]
        intr = synth.intrinsics_for(h, w)
    for s in seeds:
        depth = synth.tum_stream(1, seed0=s)[0] if g.get("tum") else synth.synth_depth(h, w, seed=s, **g["kw"])
        if g.get("tum") and s == 0:
            depth = synth.load_png_u16(synth.TUM_PNG)
        xyz = O.backproject(depth, *intr)
        res = {}
        for fl in ("glibc", "devmath"):
            O.set_math(fl)
            res[fl] = O.process(cfg, h, w, xyz, debug=True)
        (la, ca, pa, _, sa), (lb, cb, pb, _, sb) = res["glibc"], res["devmath"]
        out["frames"] += 1
        d = int((la != lb).sum())
        out["px_label_diff"] += d
        out["frames_label_diff"] += d > 0
        planar = ca["planar"].astype(bool) | cb["planar"].astype(bool)
        out["cells"] += len(ca)
        out["planar_cells"] += int(planar.sum())
        out["cells_planar_diff"] += int((ca["planar"] != cb["planar"]).sum())
        out["cells_bin_diff"] += int(((ca["bin"] != cb["bin"]) & planar).sum())
        out["cells_ql_diff"] += int((ca["used_ql"] != cb["used_ql"]).sum())
        out["frames_stats_diff"] += int(tuple(sa[k] for k in ("grows", "activations", "segments", "merges")) !=
                                        tuple(sb[k] for k in ("grows", "activations", "segments", "merges")))
        if len(pa) != len(pb):
            out["frames_plane_count_diff"] += 1
        else:
            out["planes"] += len(pa)
            for f in ("normal", "mean"):
                if len(pa):
                    # normals are unit vectors, d / mean in raw depth units (mm): deviation relative to max(1, |value|)
                    dev = np.abs(pa[f].astype(np.float64) - pb[f]) / np.maximum(1.0, np.abs(pa[f].astype(np.float64)))
                    out["max_plane_dev"] = max(out["max_plane_dev"], float(dev.max()))
            if len(pa):
                dev = np.abs(pa["d"].astype(np.float64) - pb["d"]) / np.maximum(1.0, np.abs(pa["d"].astype(np.float64)))
                out["max_plane_dev"] = max(out["max_plane_dev"], float(dev.max()))
    return fam_idx, out


def census(total=10500, workers=None, chunk=25):
    workers = workers or os.cpu_count() or 1
    tasks = []
    for i, (name, share, g) in enumerate(FAMILIES):
        n = int(round(total * share))
        base = 100000 * i if not g.get("tum") else 0
        seeds = list(range(base, base + n))
        tasks += [(i, seeds[k:k + chunk]) for k in range(0, n, chunk)]
    agg = [None] * len(FAMILIES)
    with mp.Pool(workers) as pool:
        for i, out in pool.imap_unordered(work, tasks):
            if agg[i] is None:
                agg[i] = out
            else:
                for k, v in out.items():
                    agg[i][k] = max(agg[i][k], v) if k == "max_plane_dev" else agg[i][k] + v
    rows = [dict(family=FAMILIES[i][0], **a) for i, a in enumerate(agg) if a]
    tot = {k: (max(r[k] for r in rows) if k == "max_plane_dev" else sum(r[k] for r in rows)) for k in rows[0] if k != "family"}
    return rows, tot


if __name__ == "__main__":
    argv = [a for a in sys.argv[1:] if not a.startswith("--")]
    total = int(argv[0]) if argv else 10500
    workers = int(argv[1]) if len(argv) > 1 else None
    t0 = time.time()
    rows, tot = census(total, workers)
    for r in rows:
        print(r)
    print("TOTAL", tot, f"({time.time() - t0:.0f} s)")
    if "--json" in sys.argv:
        path = sys.argv[sys.argv.index("--json") + 1]
        Path(path).write_text(json.dumps({"families": rows, "total": tot}, indent=1))
