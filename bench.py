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

# HIP maps streams onto at most GPU_MAX_HW_QUEUES hardware queues (default 4); the bench keeps 16 batches in flight so
# that the latency-bound region-growing kernel of one batch overlaps the bandwidth/ALU-bound kernels of the others.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: 8 TB/s spec


def dist_env():
    return int(os.environ.get("RANK", 0)), int(os.environ.get("LOCAL_RANK", 0)), int(os.environ.get("WORLD_SIZE", 1))


def make_workload(name: str, batch: int, rank: int):
    import synth
    if name == "tum_stream":
        K = synth.read_K(synth.TUM_K)
        depth = synth.tum_stream(batch, seed0=rank * batch)
        intr = tuple(float(v) for v in (K[0, 0], K[1, 1], K[0, 2], K[1, 2]))
        return depth, intr, f"TUM fr3 640x480 depth stream (reference frame + seeded variants), batch={batch}"
    if name.startswith("synth"):
        w, h = {"synth640": (640, 480), "synth1280": (1280, 960), "synth4k": (3840, 2160)}[name]
        depth = synth.synth_batch(batch, h, w, seed0=rank * batch)
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
    sample = depth[: min(len(depth), 8)]
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
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--batch", type=int, default=64)
    ap.add_argument("--streams", type=int, default=int(os.environ.get("DPX_BENCH_STREAMS", 16)))
    ap.add_argument("--workload", default="tum_stream")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--parity-frames", type=int, default=4)
    args = ap.parse_args()

    rank, local_rank, world = dist_env()
    import torch
    import torch.distributed as dist
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: torch.cuda.is_available() is False and there is no CPU path")
    torch.cuda.set_device(local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    import deplex_amd as D

    depth, intr, wl_desc = make_workload(args.workload, args.batch, rank)
    B, H, W = depth.shape
    N = H * W
    dev = torch.device("cuda", local_rank)
    d_depth = torch.from_numpy(depth.view(np.int16)).to(dev)
    S = max(1, args.streams)
    d_labels = torch.empty((S, B, N), dtype=torch.int32, device=dev)
    streams = [torch.cuda.Stream(device=dev) for _ in range(S)]
    exs = [D.PlaneExtractor(H, W, D.Config(), max_batch=B, device=local_rank) for _ in range(S)]
    for ex in exs:
        ex.set_profiling(True)

    def step(i):
        k = i % S
        exs[k].process_depth_device(d_depth.data_ptr(), B, *intr, d_labels[k].data_ptr(), streams[k].cuda_stream)

    def sync():
        torch.cuda.synchronize(dev)

    def barrier():
        if world > 1:
            dist.barrier()

    elapsed = timed_region(step, sync, barrier, args.steps, args.warmup)
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # per-kernel durations measured with HIP events on the launch streams during the steps above
    kms = {k: 0.0 for k in ("planefit", "cellgraph", "regions", "labels")}
    for ex in exs[: min(S, args.steps + args.warmup)]:
        for k, v in ex.kernel_ms().items():
            kms[k] += v / min(S, args.steps + args.warmup)

    # parity spot check against the oracle (outside the timed region; the oracle is only the checker)
    parity = "unchecked"
    if rank == 0 and args.parity_frames > 0:
        from oracle import oracle as O
        O.set_math("devmath")
        nchk = min(args.parity_frames, B)
        got = d_labels[(args.steps - 1) % S, :nchk].cpu().numpy()
        ref = O.process_depth_batch(O.default_config(), depth[:nchk], *intr, threads=min(nchk, os.cpu_count() or 1))
        if not np.array_equal(got, ref):
            raise SystemExit("bench: HIP label maps differ from the oracle — refusing to report a number")
        parity = f"bit-exact label maps vs oracle on {nchk} frames"

    if rank == 0:
        total_frames = B * args.steps * world
        fps = total_frames / elapsed
        pf_bytes = exs[0].planefit_bytes(0, B)
        pf_ms = kms["planefit"]
        achieved = pf_bytes / (pf_ms * 1e-3) / 1e9 if pf_ms > 0 else 0.0
        traffic = None
        tfile = ROOT / "profiles" / "planefit_traffic.json"
        if tfile.exists():
            try:
                tj = json.loads(tfile.read_text())
                if tj.get("batch") == B and tj.get("workload") == args.workload:
                    traffic = tj.get("hbm_bytes_per_launch")
            except Exception:
                traffic = None
        out = {
            "metric": "depth frames/sec (640x480)" if (W, H) == (640, 480) else f"depth frames/sec ({W}x{H})",
            "value": round(fps, 1), "unit": "frames/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 4), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": wl_desc, "patch_size": 10, "batch_per_gpu": B, "streams": S,
                       "arithmetic": "f32 moments (reference order), f64 3x3 eigen, int32 labels",
                       "parallelism": f"frame-shard x{world}, no collectives"},
            "parity": parity,
            "kernel_ms": {k: round(v, 4) for k, v in kms.items()},
            "roofline": {"kernel": "k_planefit (depth -> per-cell moments + eigen plane fit)", "bound": "hbm",
                         "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic,
                         "bytes_per_launch": pf_bytes, "avg_launch_ms": round(pf_ms, 4)},
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(depth, intr, min(16, os.cpu_count() or 1))  # the box's CPU share for one GPU
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
