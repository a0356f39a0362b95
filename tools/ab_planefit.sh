#!/bin/bash
# A/B of the plane-fit group launch between the working tree and a git ref, on ONE GPU box (boxes differ by several %):
#   tools/ab_planefit.sh <ref>     (run here; builds deplex_amd/lib/variants/ref.so from <ref> with the flags and libraries
#   of deplex_amd/build.py, then calls gpurun).  A failing compile stops the script: a stale ref.so is never measured.
set -e
ref=${1:-HEAD}
tmp=$(mktemp -d)
git archive "$ref" | tar -x -C "$tmp"
mkdir -p deplex_amd/lib/variants
rm -f deplex_amd/lib/variants/ref.so
python - "$tmp" <<'PY'
import subprocess, sys, shutil
from deplex_amd import build as b
src = sys.argv[1] + "/deplex_amd/csrc/dpx_api.hip"
out = b.OUT.parent / "variants" / "ref.so"
subprocess.run([shutil.which("hipcc") or "/opt/rocm/bin/hipcc", *b.FLAGS, src, "-o", str(out), *b.LIBS], check=True)
b.build()
PY
rm -rf "$tmp"
test -s deplex_amd/lib/variants/ref.so
/usr/local/graft/bin/gpurun --timeout 600 -- 'for rep in 1 2 3; do for v in ref cur; do L=""; [ $v = ref ] && L=$PWD/deplex_amd/lib/variants/ref.so; DPX_LIB_PATH=$L python bench.py --roofline-only --group 8 --roofline-iters 100 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(\"'"$ref"' vs tree:\", \"$v\", d[\"avg_launch_ms\"], d[\"frac\"])"; done; done'
