"""Timing-only breakdown of the plane-fit kernel (diagnostic instantiations; outputs are garbage for DIAG != 0)."""
import os as _os
_os.environ.setdefault("DPX_LIB_PATH", str(__import__("pathlib").Path(__file__).resolve().parents[1] / "deplex_amd" / "lib" / "variants" / "diag.so"))  # the -DDPX_DIAG build
import os, sys, subprocess
from pathlib import Path
ROOT = Path(__file__).resolve().parents[1]
if len(sys.argv) > 1:
    sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
    import numpy as np, torch
    import deplex_amd as D, synth
    B = int(sys.argv[2]) if len(sys.argv) > 2 else 64
    depth = synth.tum_stream(64)
    if B > 64: depth = np.concatenate([depth] * (B // 64))
    K = synth.read_K(synth.TUM_K); intr = tuple(float(v) for v in (K[0, 0], K[1, 1], K[0, 2], K[1, 2]))
    d = torch.from_numpy(depth.view(np.int16)).cuda()
    ex = D.PlaneExtractor(480, 640, max_batch=B)
    st = torch.cuda.Stream()
    ex.time_planefit(d.data_ptr(), 0, B, *intr, 5, st.cuda_stream)
    ms = min(ex.time_planefit(d.data_ptr(), 0, B, *intr, 20, st.cuda_stream) for _ in range(3))
    by = ex.planefit_bytes(0, B)
    print(f"DIAG={os.environ.get('DPX_PF_DIAG','0')} V1={os.environ.get('DPX_PLANEFIT_V1','0')} NOFD={os.environ.get('DPX_NO_FASTDIV','0')} batch={B}: {ms*1e3:.1f} us/launch, {by/ms/1e6:.0f} GB/s")
else:
    for env in ({}, {"DPX_PF_DIAG": "1"}, {"DPX_PF_DIAG": "2"}, {"DPX_NO_FASTDIV": "1"}):
        for B in ("64", "256"):
            subprocess.run([sys.executable, __file__, "run", B], env=dict(os.environ, **env))
