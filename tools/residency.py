"""Diagnostic: lifetime (shader-clock cycles) of the region workgroups in the running pipeline vs stand-alone.
usage: DPX_RESIDENCY=1 python tools/residency.py [steps]"""
import os, sys, ctypes as C
from pathlib import Path
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
os.environ["DPX_RESIDENCY"] = "1"
ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
import numpy as np, torch
import deplex_amd as D
from deplex_amd import _capi
import synth
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 512
B, S, G = 64, int(os.environ.get("DPX_BENCH_STREAMS", 4)), 8
K = synth.read_K(synth.TUM_K); intr = tuple(float(v) for v in (K[0, 0], K[1, 1], K[0, 2], K[1, 2]))
dev = torch.device("cuda", 0)
d_in = [torch.from_numpy(synth.tum_stream(B, seed0=r * B).view(np.int16)).to(dev) for r in range(8)]
d_out = torch.empty((S * G, B, 480 * 640), dtype=torch.int32, device=dev)
def res(ex):
    out = (C.c_double * 4)()
    _capi.check(_capi.lib().dpx_debug_residency(ex._h, out))
    return [round(v) for v in out]
for skip in ("", "f"):
    # stand-alone: one group at a time
    pipe = D.Pipeline(480, 640, D.Config(), batch=B, slots=S, group=G)
    for i in range(G): pipe.submit_depth(d_in[i % 8].data_ptr(), B, *intr, d_out[i].data_ptr())
    pipe.sync()
    for i in range(G): pipe.submit_depth(d_in[i % 8].data_ptr(), B, *intr, d_out[i].data_ptr())
    pipe.sync()
    alone = res(pipe.slot(0))
    for i in range(steps): pipe.submit_depth(d_in[i % 8].data_ptr(), B, *intr, d_out[i % (S * G)].data_ptr())
    pipe.sync()
    busy = [res(pipe.slot(k)) for k in range(S)]
    avg = [round(sum(b[j] for b in busy) / S) for j in range(4)]
    print(f"workgroup lifetime in cycles (prep, grow, tail; grow in 10 ns ticks): one group alone {alone}; {S} groups in flight {avg}")
    print(f"shader clock seen by the grow wave: alone {alone[1] / alone[3] / 10:.2f} GHz, pipeline {avg[1] / avg[3] / 10:.2f} GHz")
    break
