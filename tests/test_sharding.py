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


def _two_ranks(tmp_path, use_gpu: bool, port: int):
    out = tmp_path / "dist.json"
    env = dict(os.environ, DPX_DIST_OUT=str(out), MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="1", DPX_DIST_USE_GPU="1" if use_gpu else "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
           "127.0.0.1", "--master-port", str(port), str(ROOT / "tests" / "dist_worker.py")]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    res = json.loads(out.read_text())
    assert res == {"ok": True, "world": 2, "frames": 6, "max_time": res["max_time"], "hip": use_gpu}


def test_two_rank_gloo_frame_shard(tmp_path):
    _two_ranks(tmp_path, False, 29517)


@pytest.mark.gpu
def test_two_rank_gloo_frame_shard_hip_path(tmp_path):
    """The same two-rank run with the HIP path as the per-rank compute (deplex_amd.ShardedExtractor, both ranks on the
    box's one GPU), checked against the oracle on rank 0."""
    _two_ranks(tmp_path, True, 29519)


def test_expand_cell_labels_is_to_image_labels():
    """dpx_expand_cell_labels = Impl::toImageLabels (plane_extractor.cpp:455-470) on the host; no device involved."""
    import ctypes as C
    import numpy as np
    from deplex_amd import _capi
    rng = np.random.default_rng(3)
    for (h, w, P) in ((480, 640, 10), (37, 53, 4), (20, 30, 7), (5, 9, 16), (12, 12, 12)):
        nh, nv = w // P, h // P
        Pc = min(P, h, w)
        cl = rng.integers(0, 50, size=(3, nv, nh), dtype=np.int32)
        out = np.full((3, h * w), -7, dtype=np.int32)
        _capi.check(_capi.lib().dpx_expand_cell_labels(h, w, P, cl.ctypes.data, 3, out.ctypes.data))
        rows, cols = np.arange(h) // Pc, np.arange(w) // Pc
        exp = np.zeros((3, h, w), dtype=np.int32)
        ok = (rows[:, None] < nv) & (cols[None, :] < nh)
        if nv and nh:
            exp[:, ok] = cl[:, np.minimum(rows, nv - 1)[:, None], np.minimum(cols, nh - 1)[None, :]][:, ok]
        assert np.array_equal(out.reshape(3, h, w), exp), (h, w, P)
    assert _capi.lib().dpx_expand_cell_labels(4, 4, 0, None, 1, None) == _capi.DPX_ERR_ARG
