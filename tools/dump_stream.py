"""Writes the bench workload (tests/synth.tum_stream) as raw uint16 for tools/pfbench: dump_stream.py out.bin [batch]."""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1] / "tests"))
import synth
out, batch = sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 64
synth.tum_stream(batch).tofile(out)
print(f"wrote {batch} frames to {out}")
