#!/bin/bash
# A/B of the plane-fit group launch between the working tree and a git ref, on ONE GPU box (boxes differ by several %):
#   tools/ab_planefit.sh <ref>     (run here; builds deplex_amd/lib/variants/ref.so from <ref>, then calls gpurun)
set -e
ref=${1:-HEAD}
tmp=$(mktemp -d)
git archive "$ref" | tar -x -C "$tmp"
mkdir -p deplex_amd/lib/variants
hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -fPIC -shared -std=c++17 -Wno-unused-value -Wno-pass-failed "$tmp/deplex_amd/csrc/dpx_api.hip" -o deplex_amd/lib/variants/ref.so 2>/dev/null
rm -rf "$tmp"
python -c "from deplex_amd import build as b; b.build()"
/usr/local/graft/bin/gpurun --timeout 600 -- 'for rep in 1 2 3; do for v in ref cur; do L=""; [ $v = ref ] && L=$PWD/deplex_amd/lib/variants/ref.so; DPX_LIB_PATH=$L python bench.py --roofline-only --group 8 --roofline-iters 100 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(\"'"$ref"' vs tree:\", \"$v\", d[\"avg_launch_ms\"], d[\"frac\"])"; done; done'
