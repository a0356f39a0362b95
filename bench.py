"""bench.py — depth frames/s of the plane-extraction hot path on MI355X.

One step = one pass of the hot path (depth u16 -> per-pixel int32 labels + plane records) over one batch of frames
already resident in HBM.  N=1 workload: BASELINE.json configs[1] — "TUM fr3 640x480 depth stream, batch=64"
(the reference's single TUM frame + 63 seeded variants, tests/synth.py).  For N>1 every rank runs the same shape
on its own GPU with its own frames (weak scaling, no data-path collective).

Driver contract: python bench.py --gpus N --steps K --warmup W ; for N>1 launched through torch.distributed.run,
one rank per GPU.  Rank 0 prints ONE JSON line.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time
from pathlib import Path

import numpy as np

# HIP maps streams onto at most GPU_MAX_HW_QUEUES hardware queues (default 4); the bench keeps up to 20 batches in flight so
# that the latency-bound region-growing kernels of one batch overlap the bandwidth/ALU-bound kernels of the others (two
# streams that share a hardware queue run one after the other: 16 streams on 16 queues next to torch's own were slower than 14).
os.environ.setdefault("GPU_MAX_HW_QUEUES", "32")

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: 8 TB/s spec
# plane-fit kernel (depth): vector instructions the kernel EXECUTES per pixel / per cell (rocprofv3 SQ_INSTS_VALU* counters:
# 3 468.6 per wave = per cell, 24.5 of them per pixel) and the chip's measured VALU rates in T lane-operations/s: dependent
# v_mul -> v_add pairs at >= 2 waves per SIMD, v_fma_f32 (tools/ubench/pk.hip), and the kernel's own instruction mix priced
# class by class (tools/ubench/mix.hip; profiles/r03_planefit_isa_budget.md: 2.22 SIMD cycles per instruction against 1.77)
VALU_PER_PIXEL, VALU_PER_CELL = 24.5, 1019
VALU_PEAK_MULADD_TOPS, VALU_PEAK_FMA_TOPS = 62.7, 39.6
VALU_PEAK_MIX_TOPS = round(62.7 * 1.77 / 2.22, 1)


def dist_env():
    return int(os.environ.get("RANK", 0)), int(os.environ.get("LOCAL_RANK", 0)), int(os.environ.get("WORLD_SIZE", 1))


# The other BASELINE.json configs as bench lines (parity-test cases first, tests/test_gpu_parity.py::test_baseline_config_*):
#   --workload synth1280 = configs[2] (1280x960, batch 256), synth640 = one GPU's share of configs[3] (640x480, 512 of the
#   4096 frames), synth4k = one GPU's share of configs[4] (3840x2160, 8 of the 64 frames).  A step is one whole batch.
# `streams` = whole batches in flight (pipeline slots, one launch sequence each): region growing of a large frame is a
# latency-bound sequence on one wave, so a stream of batches gains from depth until the chip-wide kernels (plane fit,
# seed-list sort, tails, label expansion) fill the time: 1280x960 x 256: 101 k / 157 k / 199 k (211 k with the final sweep rounds) / 208 k / 185 k frames/s with
# 2 / 4 / 8 / 12 / 16 batches in flight; 640x480 x 512: 0.78 M / 1.19 M / 1.48 M / 1.37 M with 2 / 4 / 8 / 16; 3840x2160 x 8:
# 0.85 k / 1.6 k / 3.0 k (3.8 k with the final kernels) with 2 / 4 / 8 (8 in flight = the 64 frames of configs[4] on this one GPU).
# synth4k64 = ALL 64 frames of configs[4] per step on one GPU: bound by the chip (26.7 k frames/s with 8 or 12 batches in flight, 24 k with 16).
# synth4k = one GPU's share of configs[4], 8 frames per step.  One launch sequence per step (22 launches) is bound by the host:
# 8.3 / 11.3 / 13.7 / 14.7 / 14.4 k frames/s with 8 / 12 / 16 / 20 / 24 steps in flight, the queues idle 72 % of the time
# (tools/trace_busy.py).  With `group` steps per launch sequence (as the default workload does with 8): 21.3 k (20 x 2), 24.4 k
# (16 x 4), 24.8 k (12 x 8) — next to synth4k64's chip-bound 26.7 k.  Steps cover several rounds of the slots, or the fill and the
# drain of the pipeline are most of the timed region.
BIG_WORKLOADS = {"synth1280": dict(batch=256, unique=8, steps=32, warmup=8, streams=8), "synth640": dict(batch=512, unique=16, steps=64, warmup=16, streams=8),
                 "synth4k": dict(batch=8, unique=4, steps=384, warmup=96, streams=12, group=8), "synth4k64": dict(batch=64, unique=4, steps=48, warmup=16, streams=8)}


def make_workload(name: str, batch: int, rank: int, part: int = 0, parts: int = 1, unique: int = 0):
    """Input batch number `part` of rank `rank` (every (rank, part) pair gets its own seeds).  unique > 0: that many
    distinct frames, tiled to the batch (frame k = distinct frame k % unique) — generating hundreds of large synthetic
    frames on the host would take longer than the whole measurement."""
    import synth
    seed0 = (rank * parts + part) * batch
    if unique and name.startswith("synth") and batch % unique == 0 and unique < batch:
        w, h = {"synth640": (640, 480), "synth1280": (1280, 960), "synth4k": (3840, 2160), "synth4k64": (3840, 2160)}[name]
        uniq = synth.synth_batch(unique, h, w, seed0=seed0)
        return (np.concatenate([uniq] * (batch // unique)), synth.intrinsics_for(h, w),
                f"synthetic {w}x{h} piecewise-planar depth, batch={batch} ({unique} distinct frames, tiled)")
    if name == "tum_stream":
        K = synth.read_K(synth.TUM_K)
        depth = synth.tum_stream(batch, seed0=seed0)
        intr = tuple(float(v) for v in (K[0, 0], K[1, 1], K[0, 2], K[1, 2]))
        return depth, intr, f"TUM fr3 640x480 depth stream (reference frame + seeded variants), batch={batch}"
    if name.startswith("synth"):
        w, h = {"synth640": (640, 480), "synth1280": (1280, 960), "synth4k": (3840, 2160), "synth4k64": (3840, 2160)}[name]
        depth = synth.synth_batch(batch, h, w, seed0=seed0)
        return depth, synth.intrinsics_for(h, w), f"synthetic {w}x{h} piecewise-planar depth, batch={batch}"
    raise SystemExit(f"unknown workload {name}")


def timed_region(step_fn, sync_fn, barrier_fn, steps: int, warmup: int) -> float:
    """W untimed steps, then exactly K steps bracketed by barrier + device synchronise on both sides."""
    for i in range(warmup):
        step_fn(i)
    sync_fn()
    barrier_fn()
    sync_fn()
    t0 = time.perf_counter()
    for i in range(steps):
        step_fn(i)
    sync_fn()
    barrier_fn()
    return time.perf_counter() - t0


def cpu_baseline(depth, intr, cores: int, budget_s: float = 15.0):
    """The oracle ("port" of the reference's CPU path; the reference itself needs Eigen and cannot be built here)
    timed frame-parallel on the host cores over a bounded sample of the same workload."""
    from oracle import oracle as O
    O.set_math("glibc")
    cfg = O.default_config()
    npix = depth.shape[1] * depth.shape[2]
    sample = depth[: max(1, min(len(depth), 8, int(2.5e6 // npix)))]
    depth = depth[: min(len(depth), 64)]  # (bounded sample of the large batches)
    t = time.perf_counter()
    O.process_depth_batch(cfg, sample, *intr, threads=1)
    per_frame = (time.perf_counter() - t) / len(sample)
    single_fps = 1.0 / per_frame
    reps = max(1, int(budget_s / (per_frame * len(depth))))
    t = time.perf_counter()
    for _ in range(reps):
        O.process_depth_batch(cfg, depth, *intr, threads=cores)
    dt = time.perf_counter() - t
    return {"value": round(reps * len(depth) / dt, 1), "unit": "frames/s", "cores": cores, "kind": "port",
            "sample": f"{reps} x {len(depth)} frames of the bench workload, {cores} threads, one extractor each "
                      f"(single thread: {single_fps:.1f} frames/s)"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=0, help="timed steps (default 2000; the large-frame workloads have their own)")
    ap.add_argument("--warmup", type=int, default=-1)
    ap.add_argument("--batch", type=int, default=0, help="frames per step (default 64; --workload synth1280/synth640/synth4k: the BASELINE batch)")
    ap.add_argument("--unique", type=int, default=-1, help="distinct frames per input batch, tiled to --batch (synthetic workloads)")
    # Pipeline shape.  0 = the one default: 4 slots x groups of 8 — one hardware queue per compute pipe, measured 1.65-1.70 M
    # frames/s against 1.55-1.60 M with 16 slots and 1.52-1.55 M with 5 or 6; the large-frame
    # workloads have their own (BIG_WORKLOADS).
    ap.add_argument("--streams", type=int, default=int(os.environ.get("DPX_BENCH_STREAMS", 0)), help="groups in flight (pipeline slots); 0 = auto")
    ap.add_argument("--group", type=int, default=int(os.environ.get("DPX_BENCH_GROUP", 0)), help="submissions per launch sequence (1..32); 0 = auto")
    ap.add_argument("--flush-after", default=os.environ.get("DPX_BENCH_FLUSH_AFTER", ""),
                    help="comma-separated step counts of the timed region after which the open group is launched early (dpx_pipeline_flush)")
    ap.add_argument("--prof-every", type=int, default=16, help="record per-kernel HIP events on every n-th slot")
    ap.add_argument("--roofline-iters", type=int, default=50)
    ap.add_argument("--no-cloud-roofline", action="store_true")
    ap.add_argument("--roofline-only", action="store_true", help="only the stand-alone plane-fit series (for rocprofv3 --stats / --pmc runs)")
    ap.add_argument("--roofline-input", choices=("depth", "cloud"), default="depth", help="input form of the --roofline-only series")
    ap.add_argument("--workload", default="tum_stream")
    ap.add_argument("--input", choices=("depth", "cloud"), default="depth",
                    help="depth: u16 frames, back-projection fused (the metric); cloud: organised float32 clouds, the reference API's input form")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-sustained", action="store_true", help="skip the long-run-shape measurement that follows a short timed region")
    ap.add_argument("--sustained-steps", type=int, default=2000)
    ap.add_argument("--no-end-to-end", action="store_true", help="skip the host-to-host (PCIe-inclusive) measurement")
    ap.add_argument("--parity-frames", type=int, default=4)
    ap.add_argument("--inputs", type=int, default=8,
                    help="distinct input batches the steps rotate over (8 x 39 MB of depth frames exceed the 256 MB Infinity Cache)")
    args = ap.parse_args()
    big = BIG_WORKLOADS.get(args.workload)
    if args.batch <= 0:
        args.batch = big["batch"] if big else 64
    if args.steps <= 0:
        args.steps = big["steps"] if big else 2000
    if args.warmup < 0:
        args.warmup = big["warmup"] if big else 256
    if args.unique < 0:
        args.unique = big["unique"] if big else 0
    if big:
        args.no_cloud_roofline = True  # (the cloud form of a 256-frame 1280x960 batch is 3.8 GB of host back-projection)

    rank, local_rank, world = dist_env()
    if args.gpus != world:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}: launch N ranks with "
                         f"python -m torch.distributed.run --nproc-per-node {args.gpus} bench.py --gpus {args.gpus} ...")
    import torch
    import torch.distributed as dist
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: torch.cuda.is_available() is False and there is no CPU path")
    # DPX_BENCH_SINGLE_DEVICE=1 + DPX_BENCH_BACKEND=gloo rehearse the N>1 code path on a one-GPU box (tests only)
    if os.environ.get("DPX_BENCH_SINGLE_DEVICE") == "1":
        local_rank = 0
    backend = os.environ.get("DPX_BENCH_BACKEND", "nccl")
    torch.cuda.set_device(local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend)
    import deplex_amd as D

    # ONE default shape, whatever the run length: 4 slots x groups of 8 (one hardware queue per compute pipe of the command
    # processor).  The driver's 20 timed steps are then 2.5 groups — `value` is that short run as it is,
    # `sustained` the same shape over 2000 steps.  (Until round 3 short runs took 16 x 5 because that shape divides 20 steps
    # into whole groups: a number shaped by the protocol, dropped.)
    S = args.streams if args.streams > 0 else (big["streams"] if big else 4)  # groups in flight (pipeline slots), one stream each
    G = max(1, min(32, args.group)) if args.group > 0 else (big.get("group", 1) if big else 8)
    if big and args.inputs == 8:
        args.inputs = 2
    # The steps rotate over R distinct input batches (R divides the S*G output buffers, so buffer k always holds the
    # result of input k % R): together larger than the 256 MB Infinity Cache, i.e. the frames stream from HBM.
    R = max(1, min(args.inputs, S * G))
    while (S * G) % R:
        R -= 1
    parts = [make_workload(args.workload, args.batch, rank, r, R, args.unique) for r in range(R)]
    depths = [p[0] for p in parts]
    depth, intr, wl_desc = parts[0]
    B, H, W = depth.shape
    N = H * W
    dev = torch.device("cuda", local_rank)
    d_depths = [torch.from_numpy(d.view(np.int16)).to(dev) for d in depths]
    d_depth = d_depths[0]
    d_labels = torch.empty((S * G, B, N), dtype=torch.int32, device=dev)
    pipe = D.Pipeline(H, W, D.Config(), batch=B, slots=S, group=G, device=local_rank)
    exs = [pipe.slot(k) for k in range(S)]
    for ex in exs:
        ex.prepare_intrinsics(*intr)  # one-time exact-division check, outside the timed region
    # HIP events around every kernel of a slot cost ~35 % throughput when recorded on all slots (they serialise the
    # queues); sample them on every `prof_every`-th slot instead
    prof_slots = list(range(0, S, max(1, args.prof_every)))
    for k in prof_slots:
        exs[k].set_profiling(True)

    d_clouds = None
    if args.input == "cloud":
        from oracle import oracle as O  # back-projection of the bench frames (input preparation, outside any timed region)
        d_clouds = [torch.from_numpy(np.stack([O.backproject(f, *[np.float32(v) for v in intr]) for f in d])).to(dev) for d in depths]

    # device addresses up front: a step is one submit call (indexing a tensor per step costs more host time than the call)
    in_ptrs = [t.data_ptr() for t in (d_clouds if d_clouds is not None else d_depths)]
    out_ptrs = [d_labels[k].data_ptr() for k in range(S * G)]
    NB = S * G

    flush_after = {int(v) for v in args.flush_after.split(",") if v.strip()}
    if d_clouds is not None:
        def step(i):
            pipe.submit_pcd(in_ptrs[i % R], B, out_ptrs[i % NB])
    else:
        def step(i):
            pipe.submit_depth(in_ptrs[i % R], B, *intr, out_ptrs[i % NB])
            if (i + 1) in flush_after:
                pipe.flush()

    def sync():
        pipe.sync()
        torch.cuda.synchronize(dev)

    def barrier():
        if world > 1:
            dist.barrier()

    # The plane-fit launch the pipeline issues: ONE launch over the G submissions of a group, each with its own input
    # buffer (G * B frames; G distinct buffers of the rotation, so the launch reads G x 39 MB from HBM).
    def planefit_series(cloud: bool, iters: int):
        rstream = torch.cuda.Stream(device=dev)
        if cloud:
            from oracle import oracle as O  # back-projection of the bench frames (input preparation only)
            bufs = [torch.from_numpy(np.stack([O.backproject(f, *[np.float32(v) for v in intr]) for f in depths[j % R]])).to(dev)
                    for j in range(min(G, R))]
        else:
            bufs = d_depths
        ptrs = [bufs[j % len(bufs)].data_ptr() for j in range(G)]
        exs[0].time_planefit_group(ptrs, int(cloud), B, *intr, 3, rstream.cuda_stream)
        ms = exs[0].time_planefit_group(ptrs, int(cloud), B, *intr, iters, rstream.cuda_stream)
        return ms, exs[0].planefit_bytes(int(cloud), G * B)

    if args.roofline_only:
        cloud = args.roofline_input == "cloud"
        ms, by = planefit_series(cloud, args.roofline_iters)
        print(json.dumps({"roofline_only": True, "kernel": "k_planefit_cloud_lds" if cloud else "k_planefit_depth", "batch": B,
                          "frames_per_launch": G * B, "workload": args.workload, "avg_launch_ms": round(ms, 4), "bytes_per_launch": by,
                          "achieved_GBps": round(by / (ms * 1e-3) / 1e9, 1), "frac": round(by / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)}))
        return
    elapsed = timed_region(step, sync, barrier, args.steps, args.warmup)
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev if backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # per-kernel durations measured with HIP events on the launch streams during the timed steps (sampled slots): one
    # plane-fit / region / label launch sequence per GROUP of G submissions.  With S groups in flight these include the
    # time a launch spends sharing the GPU with the other groups' kernels.
    kms = {k: 0.0 for k in ("planefit", "cellgraph", "regions", "labels")}
    used = [k for k in prof_slots if k * G < args.steps + args.warmup]
    for k in used:
        for name, v in exs[k].kernel_ms().items():
            kms[name] += v / len(used)
    # roofline of the plane-fit kernel: the same group launch issued back to back on one stream right after the timed
    # region, bracketed by HIP events on that stream
    sync()
    pf_alone_ms, pf_bytes = planefit_series(False, args.roofline_iters)
    # the same kernel stage fed with the reference API's own input (organised float32 cloud, 12 B/point)
    cloud_ms = cloud_bytes = None
    if rank == 0 and not args.no_cloud_roofline:
        cloud_ms, cloud_bytes = planefit_series(True, max(4, args.roofline_iters // 4))

    # parity spot check against the oracle (outside the timed region; the oracle is only the checker).  EVERY rank checks
    # its own frames and buffers; the verdicts are MIN-reduced and a single failing rank makes rank 0 refuse to report.
    parity = "unchecked"
    parity_ok, parity_err = 1, ""
    if args.parity_frames > 0:
        try:
            from oracle import oracle as O
            O.set_math("devmath")
            nchk = min(args.parity_frames, B)
            last = args.steps - 1
            # the last step and the first inputs of the rotation (output buffer k holds the result of input k % R)
            nbuf = min(max(args.steps, args.warmup), S * G)  # (warm-up and timed steps both index the buffers from 0)
            check = [last % (S * G)] + [k for k in range(min(R, nbuf)) if k != last % (S * G)][:2]
            ncheck = 0
            for k in check:
                got = d_labels[k, :nchk].cpu().numpy()
                ref = O.process_depth_batch(O.default_config(), depths[k % R][:nchk], *intr, threads=min(nchk, os.cpu_count() or 1))
                if not np.array_equal(got, ref):
                    raise RuntimeError(f"rank {rank}: HIP label maps differ from the oracle (output buffer {k})")
                ncheck += nchk
            # all output buffers of the in-flight window that hold the same input must hold the same bits (catches a slot
            # being reused before its previous submission finished)
            if not all(bool(torch.equal(d_labels[k], d_labels[k % R])) for k in range(R, nbuf)):
                raise RuntimeError(f"rank {rank}: output buffers of different in-flight submissions differ")
            if os.environ.get("DPX_BENCH_FAIL_RANK") == str(rank):  # test hook (tests/test_gpu_parity.py): a rank that fails
                raise RuntimeError(f"rank {rank}: injected parity failure")
            parity = (f"bit-exact label maps vs oracle on {ncheck} frames per rank; {nbuf} in-flight output buffers consistent over "
                      f"{R} rotating inputs; checked on all {world} rank(s)")
        except Exception as e:  # noqa: BLE001  (the verdict must reach the reduction below on every rank)
            parity_ok, parity_err = 0, str(e)
        if world > 1:
            t = torch.tensor([parity_ok], dtype=torch.int32, device=dev if backend == "nccl" else "cpu")
            dist.all_reduce(t, op=dist.ReduceOp.MIN)
            all_ok = int(t.item())
        else:
            all_ok = parity_ok
        if not all_ok:
            if parity_err:
                print(f"bench: {parity_err}", file=sys.stderr, flush=True)
            raise SystemExit("bench: parity check failed on at least one rank — refusing to report a number")

    # ---- the same pipeline shape over a long run in the same record.  The driver times 20 steps; the pipeline needs a few
    # hundred to reach its steady state.  Same workload, same buffers-in-HBM rule, same barrier + synchronise bracket, MAX
    # over ranks.
    sustained = None
    if args.workload == "tum_stream" and args.input == "depth" and not args.no_sustained:
        if args.steps >= 1000 and not big:
            sus_elapsed, sus_steps, sus_S, sus_G = elapsed, args.steps, S, G
        else:
            sus_S, sus_G, sus_steps = 4, 8, args.sustained_steps
            del pipe
            pipe2 = D.Pipeline(H, W, D.Config(), batch=B, slots=sus_S, group=sus_G, device=local_rank)
            for k in range(sus_S):
                pipe2.slot(k).prepare_intrinsics(*intr)
            NB2 = sus_S * sus_G
            d_labels2 = d_labels if NB2 <= NB else torch.empty((NB2, B, N), dtype=torch.int32, device=dev)
            outs2 = [d_labels2[k].data_ptr() for k in range(NB2)]

            def step2(i):
                pipe2.submit_depth(in_ptrs[i % R], B, *intr, outs2[i % NB2])

            def sync2():
                pipe2.sync()
                torch.cuda.synchronize(dev)

            sus_elapsed = timed_region(step2, sync2, barrier, sus_steps, 256)
            if world > 1:
                t = torch.tensor([sus_elapsed], dtype=torch.float64, device=dev if backend == "nccl" else "cpu")
                dist.all_reduce(t, op=dist.ReduceOp.MAX)
                sus_elapsed = float(t.item())
            exs = [pipe2.slot(k) for k in range(sus_S)]  # (the plane-fit series below runs on a live handle)
            pipe = pipe2
        sus_fps = B * sus_steps * world / sus_elapsed
        sus_gbs = sus_fps / world * N * 6 / 1e9
        sustained = {"value": round(sus_fps, 1), "unit": "frames/s", "steps": sus_steps, "warmup": 256,
                     "ms_per_step": round(sus_elapsed / sus_steps * 1e3, 4), "groups_in_flight": sus_S, "batches_per_launch": sus_G,
                     "pipeline_roofline": {"bound": "hbm", "achieved": round(sus_gbs, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                           "frac": round(sus_gbs / HBM_PEAK_GBS, 4)},
                     "note": "same workload and rules as `value`, run right after the timed region with the long-run pipeline shape"}

    # ---- end to end: depth frames in page-locked HOST memory -> results in HOST memory, PCIe included (what a caller of the
    # reference API sees: examples/process_sequence.cpp:30-53).  Never `value`.  Rank 0 at N = 1 only (the ranks of a node
    # share the host's PCIe root and memory bandwidth; a per-rank figure measured one at a time would not add up).
    end_to_end = None
    if world == 1 and not args.no_end_to_end and args.input == "depth":
        exh = D.PlaneExtractor(H, W, D.Config(), max_batch=B, device=local_rank)
        pin_in = D.pinned_empty(depth.shape, np.uint16)
        pin_in[:] = depth
        pin_out = D.pinned_empty((B, N), np.int32)
        pin_cl = D.pinned_empty((B, exh.nv, exh.nh), np.int32)

        def rate(fn, reps):
            fn()
            t0 = time.perf_counter()
            for _ in range(reps):
                fn()
            return (time.perf_counter() - t0) / reps

        reps = max(3, min(30, int(2e9 / (B * N * 6))))
        t_pix = rate(lambda: exh.process_depth_batch(pin_in, *intr, out=pin_out), reps)
        t_cel = rate(lambda: exh.process_depth_batch_cells(pin_in, *intr, out=pin_cl), reps)
        ok_e2e = np.array_equal(pin_out, np.asarray(exh.expand_cell_labels(pin_cl)))
        if not ok_e2e:
            raise SystemExit("bench: cell-level output expanded on the host differs from the pixel label map")
        # the same over a pipeline's host-buffer submissions: every slot's stream carries copy in, kernels, copy out, so the
        # copies of one slot run beside the kernels (and, as far as the link allows, the copies) of the others
        SH = 3
        pipe_h = D.Pipeline(H, W, D.Config(), batch=B, slots=SH, group=1, device=local_rank)
        for k in range(SH):
            pipe_h.slot(k).prepare_intrinsics(*intr)
        nsub = max(2 * SH, min(48, 2 * reps))
        streamed = {}
        for cells, shape in ((False, (B, N)), (True, (B, exh.nv, exh.nh))):
            houts = [D.pinned_empty(shape, np.int32) for _ in range(SH + 1)]

            def run_h():
                for i in range(nsub):
                    pipe_h.submit_depth_host(pin_in, *intr, houts[i % len(houts)], cells=cells)
                pipe_h.sync()

            run_h()
            t0 = time.perf_counter()
            run_h()
            t_h = (time.perf_counter() - t0) / nsub
            if not all(np.array_equal(o, pin_cl if cells else pin_out) for o in houts):
                raise SystemExit("bench: results of the pipeline's host-buffer submissions differ from the single calls'")
            streamed["cell_labels" if cells else "pixel_labels"] = {
                "value": round(B / t_h, 1), "unit": "frames/s", "ms_per_batch": round(t_h * 1e3, 3),
                "h2d_GBps": round(B * N * 2 / t_h / 1e9, 2), "d2h_GBps": round(houts[0].nbytes / t_h / 1e9, 3)}
        streamed["entry"] = f"dpx_pipeline_submit_depth_host / _cells_host, {SH} slots of one batch, {nsub} submissions then sync; results equal the single calls' (checked)"
        del pipe_h
        end_to_end = {
            "streamed": streamed,
            "pixel_labels": {"value": round(B / t_pix, 1), "unit": "frames/s", "ms_per_batch": round(t_pix * 1e3, 3),
                             "h2d_GBps": round(B * N * 2 / t_pix / 1e9, 2), "d2h_GBps": round(B * N * 4 / t_pix / 1e9, 2),
                             "entry": "dpx_process_depth_batch (2 B in + 4 B out per pixel)"},
            "cell_labels": {"value": round(B / t_cel, 1), "unit": "frames/s", "ms_per_batch": round(t_cel * 1e3, 3),
                            "h2d_GBps": round(B * N * 2 / t_cel / 1e9, 2),
                            "d2h_GBps": round(B * exh.nv * exh.nh * 4 / t_cel / 1e9, 3),
                            "entry": "dpx_process_depth_batch_cells (cell labels + plane records; expanded on the host they equal the pixel map: checked)"},
            "host_buffers": "page-locked (dpx_host_alloc)", "batch": B,
            "note": "pixel_labels / cell_labels: one call at a time on one stream (tools/duplex_probe.py: 53 GB/s either way alone, 26.6 GB/s each "
                    "when one copy each way runs); streamed: submissions of several slots in flight"}
        del exh

    if rank == 0:
        total_frames = B * args.steps * world
        fps = total_frames / elapsed
        pf_ms = pf_alone_ms
        achieved = pf_bytes / (pf_ms * 1e-3) / 1e9 if pf_ms > 0 else 0.0
        in_pipe = pf_bytes / (kms["planefit"] * 1e-3) / 1e9 if kms["planefit"] > 0 else 0.0
        traffic = cloud_traffic = None
        tfile = ROOT / "profiles" / "planefit_traffic.json"
        if tfile.exists():
            try:
                tj = json.loads(tfile.read_text())
                # measured per launch of `frames_per_launch` frames; a launch's traffic is proportional to its frames
                # (every wave reads its own rows once), so other group sizes are scaled
                if tj.get("workload") == args.workload and tj.get("frames_per_launch"):
                    traffic = int(round(tj["hbm_bytes_per_launch"] / tj["frames_per_launch"] * G * B))
                    c = tj.get("cloud", {})
                    if c.get("frames_per_launch"):
                        cloud_traffic = int(round(c["hbm_bytes_per_launch"] / c["frames_per_launch"] * G * B))
            except Exception:
                traffic = None
        # lane-operations of the exact arithmetic in the kernel's ISA: VALU_PER_PIXEL per pixel
        # + VALU_PER_CELL per cell for the f64 plane fit; the chip's measured rates for that mix are in tools/ubench/pk.hip
        lane_ops = int(G * B * (VALU_PER_PIXEL * N + VALU_PER_CELL * (W // 10) * (H // 10)))
        out = {
            "metric": ("depth frames/sec (640x480)" if (W, H) == (640, 480) else f"depth frames/sec ({W}x{H})") +
                      (" [organised-cloud input]" if args.input == "cloud" else ""),
            "value": round(fps, 1), "unit": "frames/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 4), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": wl_desc, "patch_size": 10, "batch_per_gpu": B, "groups_in_flight": S, "batches_per_launch": G,
                       "shape_auto_selected": args.streams <= 0 and args.group <= 0,
                       "distinct_input_batches": R, "input_bytes_in_rotation": int(R * B * N * (12 if args.input == "cloud" else 2)),
                       "streams": f"{S} (one per group in flight), GPU_MAX_HW_QUEUES={os.environ.get('GPU_MAX_HW_QUEUES')}",
                       "arithmetic": "f32 moments (reference order), f64 3x3 eigen, int32 labels",
                       "parallelism": f"frame-shard x{world}, no collectives"},
            "parity": parity,
            "kernel_ms": {k: round(v, 4) for k, v in kms.items()},
            "roofline": {"kernel": "k_planefit_depth (depth -> per-cell moments + eigen plane fit), one launch per group of "
                                   f"{G} submissions = {G * B} frames",
                         "bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic,
                         "bytes_per_launch": pf_bytes, "avg_launch_ms": round(pf_ms, 4),
                         "valu": {"lane_ops_per_launch": lane_ops,
                                  "achieved_Tops": round(lane_ops / (pf_ms * 1e-3) / 1e12, 2),
                                  "measured_peak_Tops": {"mul_add": VALU_PEAK_MULADD_TOPS, "fma": VALU_PEAK_FMA_TOPS, "kernel_mix": VALU_PEAK_MIX_TOPS},
                                  "frac_of_mul_add_peak": round(lane_ops / (pf_ms * 1e-3) / (VALU_PEAK_MULADD_TOPS * 1e12), 3),
                                  "frac_of_kernel_mix_peak": round(lane_ops / (pf_ms * 1e-3) / (VALU_PEAK_MIX_TOPS * 1e12), 3),
                                  "budget": "profiles/r03_planefit_isa_budget.md (executed instructions by class x measured issue costs)"},
                         "measured": f"{args.roofline_iters} launches back to back on one stream after the timed region, HIP events on that stream; "
                                     f"{G} distinct input buffers per launch",
                         "in_pipeline": {"avg_launch_ms": round(kms["planefit"], 4), "achieved": round(in_pipe, 1),
                                         "note": f"HIP events around the launches of the timed steps with {S} groups of {G} batches in flight (includes queueing behind / sharing with the other groups' kernels)"}},
        }
        # the whole path against the same roofline: SURVEY 8(d) algorithmic bytes per frame (input + 4 B of label per pixel)
        in_bpp = 12 if args.input == "cloud" else 2
        pipe_gbs = fps / world * N * (in_bpp + 4) / 1e9  # per GPU
        pipe_traffic = None
        ptfile = ROOT / "profiles" / "pipeline_traffic.json"
        if ptfile.exists() and args.input == "depth":
            try:
                pipe_traffic = json.loads(ptfile.read_text()).get(args.workload, {}).get("hbm_bytes_per_frame")
            except Exception:
                pipe_traffic = None
        out["pipeline_roofline"] = {"bound": "hbm", "algorithmic_bytes_per_frame": N * (in_bpp + 4), "achieved": round(pipe_gbs, 1),
                                    "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(pipe_gbs / HBM_PEAK_GBS, 4),
                                    "traffic": pipe_traffic,
                                    "note": "whole pipeline, input -> labels, per GPU; traffic = measured L2<->memory bytes per frame over all "
                                            "kernels (2 x FETCH_SIZE + WRITE_SIZE, profiles/pipeline_traffic.json; null: not collected for this workload)"}
        if cloud_ms:
            out["roofline_cloud_input"] = {"kernel": "k_planefit_cloud_lds (organised xyz cloud -> per-cell moments + eigen plane fit), "
                                                     f"one launch per group of {G} submissions = {G * B} frames",
                                           "bound": "hbm", "achieved": round(cloud_bytes / (cloud_ms * 1e-3) / 1e9, 1), "peak": HBM_PEAK_GBS,
                                           "unit": "GB/s", "frac": round(cloud_bytes / (cloud_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                                           "traffic": cloud_traffic, "bytes_per_launch": cloud_bytes, "avg_launch_ms": round(cloud_ms, 4)}
        if sustained:
            out["sustained"] = sustained
        if end_to_end:
            out["end_to_end"] = end_to_end
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(depth, intr, min(16, os.cpu_count() or 1))  # the box's CPU share for one GPU
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
