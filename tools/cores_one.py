"""Ad-hoc: the TUM frame through the core path with every launch synchronised and named (DPX_SYNC_EACH)."""
import os, sys
os.environ["DPX_REGIONS_CORES"] = "1"; os.environ["DPX_SYNC_EACH"] = "1"
from pathlib import Path
ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
import numpy as np, synth
import deplex_amd as D
K = synth.read_K(synth.TUM_K); intr = tuple(float(v) for v in (K[0, 0], K[1, 1], K[0, 2], K[1, 2]))
ex = D.PlaneExtractor(480, 640, max_batch=1)
print(ex.options(), flush=True)
lab = ex.process_depth_batch(synth.load_png_u16(synth.TUM_PNG)[None], *intr)
print("max label", lab.max(), ex.frame_stats(0), flush=True)
