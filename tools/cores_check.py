"""Ad-hoc GPU check of the core-based region path (dpx_regions3.h) against the oracle: every record of every frame.
usage: cores_check.py [quick]"""
import os, sys, time, traceback
from pathlib import Path
ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
import numpy as np
import synth
import deplex_amd as D
from oracle import oracle as O
import test_gpu_parity as T

quick = len(sys.argv) > 1 and sys.argv[1] == "quick"
os.environ["DPX_REGIONS_CORES"] = "1"
K = synth.read_K(synth.TUM_K); tum_intr = tuple(float(v) for v in (K[0, 0], K[1, 1], K[0, 2], K[1, 2]))
K = synth.read_K(synth.ICL_K); icl_intr = tuple(float(v) for v in (K[0, 0], K[1, 1], K[0, 2], K[1, 2]))
cases = []
cases.append(("tum", synth.load_png_u16(synth.TUM_PNG)[None], tum_intr, O.default_config()))
cases.append(("tum stream 4", synth.tum_stream(4), tum_intr, O.default_config()))
cases.append(("icl ini", synth.load_png_u16(synth.ICL_PNG)[None], icl_intr, O.config_from_ini(synth.ICL_INI)))
for s in range(0, 6, 2):
    cases.append((f"synth640 s{s}..", synth.synth_batch(2, 480, 640, seed0=s), synth.intrinsics_for(480, 640), O.default_config()))
cases.append(("synth640 mincells1", synth.synth_batch(2, 480, 640, seed0=2), synth.intrinsics_for(480, 640), O.default_config(min_region_growing_cells_activated=1, min_region_growing_candidate_size=1)))
cases.append(("synth320x240 P4", synth.synth_batch(3, 240, 320, seed0=7), synth.intrinsics_for(240, 320), O.default_config(patch_size=4)))
cases.append(("ragged 333x517", synth.synth_batch(2, 333, 517, seed0=3), synth.intrinsics_for(333, 517), O.default_config(patch_size=7)))
cases.append(("synth1280 s0..", synth.synth_batch(3, 960, 1280, seed0=0), synth.intrinsics_for(960, 1280), O.default_config()))
cases.append(("quilt1280", np.stack([synth.quilt_depth(960, 1280, seed=s) for s in range(2)]), synth.intrinsics_for(960, 1280), O.default_config()))
cases.append(("quilt1280 jitter", np.stack([synth.quilt_depth(960, 1280, seed=3, jitter=2.0)]), synth.intrinsics_for(960, 1280), O.default_config()))
cases.append(("quilt1280 3x3 bins", np.stack([synth.quilt_depth(960, 1280, seed=5)]), synth.intrinsics_for(960, 1280), O.default_config(histogram_bins_per_coord=3)))
cases.append(("synth640 1 bin", synth.synth_batch(2, 480, 640, seed0=2), synth.intrinsics_for(480, 640), O.default_config(histogram_bins_per_coord=1)))
if not quick:
    cases.append(("quilt4k 2x2 bins (segments beyond the LDS sort)", np.stack([synth.quilt_depth(2160, 3840, seed=2)]), synth.intrinsics_for(2160, 3840), O.default_config(histogram_bins_per_coord=2)))
    cases.append(("synth4k s0..3", synth.synth_batch(4, 2160, 3840, seed0=0), synth.intrinsics_for(2160, 3840), O.default_config()))
    cases.append(("quilt4k", np.stack([synth.quilt_depth(2160, 3840, seed=1)]), synth.intrinsics_for(2160, 3840), O.default_config()))
fails = 0
for name, depth, intr, cfg in cases:
    t0 = time.time()
    try:
        T.full_compare(D, O, depth, intr, cfg)
        opt = D.PlaneExtractor(depth.shape[1], depth.shape[2], T.to_cfg(D, cfg), max_batch=1).options()
        print(f"ok   {name}  ({time.time() - t0:.1f} s) regions={opt['regions']}", flush=True)
    except Exception as e:  # noqa: BLE001
        fails += 1
        print(f"FAIL {name}: {type(e).__name__}: {str(e)[:300]}", flush=True)
        try:
            B, H, W = depth.shape
            ex = D.PlaneExtractor(H, W, T.to_cfg(D, cfg), max_batch=B)
            ex.process_depth_batch(depth, *intr)
            for k in range(B):
                xyz = O.backproject(depth[k], *intr)
                ol, oc, op, ocl, ost = O.process(cfg, H, W, xyz, debug=True)
                st = ex.frame_stats(k)
                print("     frame", k, "gpu", {n: int(st[n]) for n in ("planar_cells", "grows", "activations", "segments", "merges")},
                      "oracle", {n: int(ost[n]) for n in ("planar_cells", "grows", "activations", "segments", "merges")}, flush=True)
        except Exception:  # noqa: BLE001
            traceback.print_exc()
print(f"done: {len(cases)} cases, {fails} failures")
sys.exit(1 if fails else 0)
