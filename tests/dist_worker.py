"""Worker for tests/test_sharding.py: launched by torch.distributed.run with the gloo backend (CPU).

Exercises the N>1 plumbing of bench.py without a GPU: rank/world from the environment, contiguous frame shards
(deplex_amd.sharding), the barrier-bracketed timed region, the MAX-over-ranks reduction, and a gather of per-frame
label checksums.  Per-rank compute: DPX_DIST_USE_GPU=1 (the -m gpu test) = the HIP path through
deplex_amd.ShardedExtractor on the rank's device — the product's own sharded entry point, one device per rank — checked
on rank 0 against the oracle; otherwise (CPU box) the oracle stands in for the HIP path (tests may use the oracle)."""
import hashlib
import json
import os
import sys
from pathlib import Path

import numpy as np
import torch
import torch.distributed as dist

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))
import bench  # noqa: E402
import synth  # noqa: E402
from deplex_amd.sharding import shard_range  # noqa: E402
from oracle import oracle as O  # noqa: E402


def main():
    rank, local_rank, world = bench.dist_env()
    dist.init_process_group("gloo")
    total = 6
    h, w = 120, 160
    depth = synth.synth_batch(total, h, w, seed0=100)  # every rank generates the same global batch
    intr = synth.intrinsics_for(h, w)
    lo, hi = shard_range(total, rank, world)
    out = {}
    use_gpu = os.environ.get("DPX_DIST_USE_GPU") == "1"
    if use_gpu:
        import deplex_amd as D  # (fails loudly without the HIP library or a device: there is no CPU path)
        ndev = max(1, torch.cuda.device_count())
        sh = D.ShardedExtractor(h, w, D.Config(), devices=[local_rank % ndev], max_batch=4)

        def step(i):
            out["labels"] = np.array(sh.process_depth_batch(depth[lo:hi], *intr))
    else:
        def step(i):
            out["labels"] = O.process_depth_batch(O.default_config(), depth[lo:hi], *intr, threads=1)

    elapsed = bench.timed_region(step, lambda: None, dist.barrier, steps=2, warmup=1)
    t = torch.tensor([elapsed], dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    mine = [(lo + k, hashlib.sha256(out["labels"][k].tobytes()).hexdigest()) for k in range(hi - lo)]
    gathered = [None] * world
    dist.all_gather_object(gathered, mine)
    if rank == 0:
        flat = sorted(x for part in gathered for x in part)
        O.set_math("devmath" if use_gpu else "glibc")
        ref = O.process_depth_batch(O.default_config(), depth, *intr, threads=1)
        expect = [(k, hashlib.sha256(ref[k].tobytes()).hexdigest()) for k in range(total)]
        ok = flat == expect and float(t.item()) >= elapsed - 1e-9
        Path(os.environ["DPX_DIST_OUT"]).write_text(json.dumps({"ok": bool(ok), "world": world, "frames": len(flat),
                                                               "max_time": float(t.item()), "hip": use_gpu}))
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
