"""Multi-GPU path on CPU: frame sharding + the distributed timing plumbing with world_size 2 over gloo."""
import json
import os
import subprocess
import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parents[1]


def test_shard_range_partitions_exactly():
    from deplex_amd.sharding import shard_range
    for n in (0, 1, 7, 64, 4096, 4097):
        for world in (1, 2, 3, 4, 8):
            spans = [shard_range(n, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            sizes = [b - a for a, b in spans]
            assert max(sizes) - min(sizes) <= 1
    with pytest.raises(ValueError):
        shard_range(4, 2, 2)


def test_two_rank_gloo_frame_shard(tmp_path):
    out = tmp_path / "dist.json"
    env = dict(os.environ, DPX_DIST_OUT=str(out), MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
           "127.0.0.1", "--master-port", "29517", str(ROOT / "tests" / "dist_worker.py")]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    res = json.loads(out.read_text())
    assert res == {"ok": True, "world": 2, "frames": 6, "max_time": res["max_time"]}
