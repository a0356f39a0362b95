"""Randomised parity sweep: random image sizes, patch sizes, configurations, input forms and kernel tiers against the
oracle (labels, cell records, plane records, statistics).  usage: fuzz_parity.py [cases] [seed] [share of large frames]
(tests/test_gpu_parity.py::test_seeded_fuzz runs run_case() for a fixed seed inside the driver's `pytest -m gpu`)"""
import os, sys, time
from pathlib import Path
ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
import numpy as np
import synth
import deplex_amd as D
from oracle import oracle as O
import test_gpu_parity as T

import torch


def fuzz_pipeline(rng, depth, intr, ocfg, xyz):
    """The same frames through dpx_pipeline with random slots / group / number of submissions (device buffers)."""
    B, H, W = depth.shape
    dev = torch.device("cuda", 0)
    O.set_math("devmath")
    refs = []
    for d in depth:
        refs.append(O.process(ocfg, H, W, O.backproject(d, *[np.float32(v) for v in intr]), debug=True)[0])
    ref = np.stack(refs)
    slots, group, nsub = int(rng.integers(1, 4)), int(rng.integers(1, 5)), int(rng.integers(1, 8))
    if rng.random() < 0.25:  # wide groups (up to the 32 submissions a group can hold), more submissions than one group
        group, nsub = int(rng.integers(5, 33)), int(rng.integers(1, 40))
    pipe = D.Pipeline(H, W, T.to_cfg(D, ocfg), batch=B, slots=slots, group=group)
    if xyz:
        d_in = torch.from_numpy(np.stack([O.backproject(d, *[np.float32(v) for v in intr]) for d in depth])).to(dev)
    else:
        d_in = torch.from_numpy(depth.view(np.int16)).to(dev)
    outs = [torch.full((B, H * W), -1, dtype=torch.int32, device=dev) for _ in range(nsub)]
    # depth input: some submissions hand over HOST buffers (page-locked or pageable; pixel maps or, without the
    # refinement, cell labels) between the device-buffer ones
    ex0 = pipe.slot(0)
    nvh = (ex0.nv, ex0.nh)
    host_ok = not xyz
    cells_ok = host_ok and not bool(ocfg.ransac_refinement)
    pin_in = None
    kinds = []
    for i in range(nsub):
        kind = "dev"
        if host_ok and rng.random() < 0.4:
            kind = "cells" if (cells_ok and rng.random() < 0.4) else ("pin" if rng.random() < 0.6 else "page")
        kinds.append(kind)
        if xyz:
            pipe.submit_pcd(d_in.data_ptr(), B, outs[i].data_ptr())
        elif kind == "dev":
            pipe.submit_depth(d_in.data_ptr(), B, *intr, outs[i].data_ptr())
        else:
            if pin_in is None:
                pin_in = D.pinned_empty(depth.shape, np.uint16)
                pin_in[:] = depth
            shape = (B, nvh[0], nvh[1]) if kind == "cells" else (B, H * W)
            o = np.full(shape, -1, np.int32) if kind == "page" else D.pinned_empty(shape, np.int32)
            o[...] = -1
            outs[i] = o
            pipe.submit_depth_host(depth if kind == "page" else pin_in, *intr, o, cells=(kind == "cells"))
        if rng.random() < 0.2:
            pipe.sync()
    pipe.sync()
    for i in range(nsub):
        got = outs[i].cpu().numpy() if kinds[i] == "dev" or xyz else np.asarray(outs[i])
        if kinds[i] == "cells" and not xyz:
            got = ex0.expand_cell_labels(got)
        assert np.array_equal(got, ref), f"pipeline submission {i} ({kinds[i]}; slots {slots}, group {group}, nsub {nsub})"
    del pipe


def run_case(rng, big_fraction=0.1, verbose=False):
    """One random case; raises on any difference.  Returns a one-line description (also attached to the exception)."""
    P = int(rng.choice([4, 8, 10, 10, 10, 6, 5, 7, 12, 16]))
    big = rng.random() < big_fraction
    huge = big and rng.random() < 0.25
    nh = int(rng.integers(1, 400 if huge else (200 if big else 90)))
    nv = int(rng.integers(1, 230 if huge else (130 if big else 60)))
    w = nh * P + int(rng.choice([0, 0, 0, 1, 2, 3]))
    h = nv * P + int(rng.choice([0, 0, 1, P - 1]))
    ov = {"patch_size": P}
    if rng.random() < 0.5:
        ov["histogram_bins_per_coord"] = int(rng.choice([3, 7, 20, 33, 60]))
    if rng.random() < 0.3:
        ov["min_cos_angle_merge"] = float(rng.choice([0.8, 0.93, 0.99]))
    if rng.random() < 0.3:
        ov["min_region_growing_cells_activated"] = int(rng.choice([1, 2, 8]))
    if rng.random() < 0.3:
        ov["min_region_growing_candidate_size"] = int(rng.choice([1, 3, 10]))
    if rng.random() < 0.2:
        ov["min_region_planarity_score"] = float(rng.choice([0.3, 0.8]))
    if rng.random() < 0.15:
        ov["ransac_refinement"] = 1
    envs = str(rng.choice(["", "", "DPX_REGIONS_LEAN", "DPX_REGIONS_GSTATE", "DPX_REGIONS_NOSPLIT", "DPX_FUSED_LABELS", "DPX_REGIONS_V1", "DPX_PREP_GROW_SPLIT",
                           "DPX_REGIONS_GSPLIT", "DPX_REGIONS_CORES", "DPX_REGIONS_CORES", "DPX_REGIONS_CORES", "DPX_REGIONS_V2",
                           "DPX_REGIONS_LEAN DPX_SORT_COUNTING", "DPX_REGIONS_GSTATE DPX_SORT_COUNTING"])).split()
    xyz = bool(rng.random() < 0.3)
    B = int(rng.integers(1, 4))
    seed = int(rng.integers(0, 10000))
    kw = dict(noise=bool(rng.random() < 0.8), dropout=bool(rng.random() < 0.8), scale=float(rng.choice([1.0, 5.0])))
    if rng.random() < 0.2:  # quilts of small patches: thousands of small grows next to each other (sweep rounds under load)
        qk = dict(P=P, kmax=int(rng.integers(1, 5)), ntilt=int(rng.integers(2, 13)), jitter=float(rng.choice([0.0, 0.3, 1.0])), drop=float(rng.choice([0.0, 0.05, 0.2])))
        depth = np.stack([synth.quilt_depth(h, w, seed=seed + k, **qk) for k in range(B)])
        kw = qk
    else:
        depth = np.stack([synth.synth_depth(h, w, seed=seed + k, **kw) for k in range(B)])
    desc = f"{h}x{w} P={P} B={B} seed={seed} env={' '.join(envs)} xyz={xyz} ov={ov} kw={kw}"
    if verbose:  # (to find a case that never returns: its parameters are printed first)
        print("case:", desc, flush=True)
    for e in envs:
        os.environ[e] = "1"
    try:
        if rng.random() < 0.2 and "DPX_REGIONS_V1" not in envs:
            fuzz_pipeline(rng, depth, synth.intrinsics_for(h, w), O.default_config(**ov), xyz)
        else:
            T.full_compare(D, O, depth, synth.intrinsics_for(h, w), O.default_config(**ov), xyz_api=xyz, glibc=False)
    except Exception as e:  # noqa: BLE001
        raise AssertionError(f"{desc}: {type(e).__name__}: {str(e)[:200]}") from e
    finally:
        for e in envs:
            del os.environ[e]
    return desc


def main():
    big_fraction = float(sys.argv[3]) if len(sys.argv) > 3 else 0.1  # share of large frames (up to ~4K-sized grids)
    cases = int(sys.argv[1]) if len(sys.argv) > 1 else 100
    rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
    t0 = time.time()
    fails = 0
    for it in range(cases):
        try:
            run_case(rng, big_fraction, verbose=bool(os.environ.get("DPX_FUZZ_VERBOSE")))
        except AssertionError as e:
            fails += 1
            print(f"FAIL case {it}: {e}", flush=True)
        if it % 20 == 19:
            print(f"{it + 1} cases, {fails} failures, {time.time() - t0:.0f} s", flush=True)
    print(f"done: {cases} cases, {fails} failures")
    sys.exit(1 if fails else 0)


if __name__ == "__main__":
    main()
