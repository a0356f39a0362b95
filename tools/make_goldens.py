"""Generates tests/golden/oracle_goldens.json from the CPU oracle (glibc flavour, i.e. what the reference links).

Run in the build container: `python tools/make_goldens.py`.  The reference itself cannot be built here (it needs
Eigen 3.4 from the network), so these vectors come from the oracle; the four fixture rows reproduce the
independently derived provisional goldens of SURVEY.md Appendix B (same sha256 prefixes) and the reference's own
known answer (34 planes on the TUM frame).  Inputs are the reference's test images plus seeded synthetic frames
(tests/synth.py); outputs are sha256 of the int32 label map and the stage counters.
"""
import hashlib
import json
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))
import synth  # noqa: E402
from oracle import oracle as O  # noqa: E402


def f64_backproject(depth, Kpath):
    """python/tests/utils.py:5-17 of the reference: float64 back-projection, cast to float32 by the binding."""
    K = np.loadtxt(Kpath)
    h, w = depth.shape
    pts = np.zeros((w * h, 3))
    col = np.tile(np.arange(w), (h, 1)).flatten()
    row = np.transpose(np.tile(np.arange(h), (w, 1))).flatten()
    pts[:, 2] = depth.astype(np.float64).flatten()
    pts[:, 0] = (col - K[0][2]) * pts[:, 2] / K[0][0]
    pts[:, 1] = (row - K[1][2]) * pts[:, 2] / K[1][1]
    return pts.astype(np.float32)


def row(name, labels, st):
    return {"name": name, "sha256": hashlib.sha256(labels.astype("<i4").tobytes()).hexdigest(),
            "max_label": int(labels.max()) if labels.size else 0, "labelled_px": int((labels > 0).sum()),
            "distinct": int(len(np.unique(labels[labels > 0]))),
            **{k: int(st[k]) for k in ("planar_cells", "grows", "activations", "max_grow", "segments", "merges")}}


def plane_rows(planes):
    """Fitted plane records as float32 bit patterns (exact) — pins the "plane parameters within 1e-5" half of the bar."""
    return [{"normal": [int(np.float32(v).view(np.uint32)) for v in p["normal"]], "d": int(np.float32(p["d"]).view(np.uint32)),
             "mse": int(np.float32(p["mse"]).view(np.uint32)), "score": int(np.float32(p["score"]).view(np.uint32)),
             "n_pts": int(p["n_pts"]), "merge_label": int(p["merge_label"])} for p in planes]


def main():
    O.set_math("glibc")
    rows = []
    plane_gold = {}
    tum = synth.load_png_u16(synth.TUM_PNG)
    icl = synth.load_png_u16(synth.ICL_PNG)
    Kt, Ki = synth.read_K(synth.TUM_K), synth.read_K(synth.ICL_K)
    xyz = O.backproject(tum, Kt[0, 0], Kt[1, 1], Kt[0, 2], Kt[1, 2])
    l, _, pl, _, st = O.process(O.default_config(), 480, 640, xyz, debug=True)
    rows.append(row("tum_f32_default", l, st))
    plane_gold["tum_f32_default"] = plane_rows(pl)
    l, _, _, _, st = O.process(O.default_config(), 480, 640, f64_backproject(tum, synth.TUM_K), debug=True)
    rows.append(row("tum_f64_default", l, st))
    xyz = O.backproject(icl, Ki[0, 0], Ki[1, 1], Ki[0, 2], Ki[1, 2])
    l, _, pl, _, st = O.process(O.default_config(), 480, 640, xyz, debug=True)
    rows.append(row("icl_default", l, st))
    plane_gold["icl_default"] = plane_rows(pl)
    l, _, pl, _, st = O.process(O.config_from_ini(synth.ICL_INI), 480, 640, xyz, debug=True)
    rows.append(row("icl_ini", l, st))
    plane_gold["icl_ini"] = plane_rows(pl)
    for (h, w, kw, seeds) in [(240, 320, {}, range(4)), (480, 640, {}, range(4)), (480, 640, {"noise": False}, range(2)),
                              (480, 640, {"scale": 5.0}, range(2)), (960, 1280, {}, range(1))]:
        fx, fy, cx, cy = synth.intrinsics_for(h, w)
        for s in seeds:
            d = synth.synth_depth(h, w, seed=s, **kw)
            xyz = O.backproject(d, fx, fy, cx, cy)
            l, _, _, _, st = O.process(O.default_config(), h, w, xyz, debug=True)
            tag = "_".join(f"{k}{v}" for k, v in kw.items()) or "noisy"
            r = row(f"synth_{w}x{h}_{tag}_seed{s}", l, st)
            r["depth_sha256"] = hashlib.sha256(d.tobytes()).hexdigest()
            rows.append(r)
    ts = synth.tum_stream(4)
    for k in range(4):
        xyz = O.backproject(ts[k], Kt[0, 0], Kt[1, 1], Kt[0, 2], Kt[1, 2])
        l, _, _, _, st = O.process(O.default_config(), 480, 640, xyz, debug=True)
        r = row(f"tum_stream_{k}", l, st)
        r["depth_sha256"] = hashlib.sha256(ts[k].tobytes()).hexdigest()
        rows.append(r)
    out = ROOT / "tests" / "golden" / "oracle_goldens.json"
    out.write_text(json.dumps({"generator": "tools/make_goldens.py (oracle, glibc flavour)", "rows": rows}, indent=1))
    print(f"wrote {out} ({len(rows)} rows)")
    pout = ROOT / "tests" / "golden" / "plane_goldens.json"
    pout.write_text(json.dumps({"generator": "tools/make_goldens.py (oracle, glibc flavour); float32 bit patterns",
                                "frames": plane_gold}, indent=0))
    print(f"wrote {pout} ({sum(len(v) for v in plane_gold.values())} planes)")
    for r in rows[:4]:
        print(r["name"], r["sha256"][:16], r["max_label"], r["segments"])


if __name__ == "__main__":
    main()
