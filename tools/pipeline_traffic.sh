#!/bin/bash
# HBM traffic of every kernel of the pipeline (rocprofv3 PMC, separate passes for FETCH_SIZE and WRITE_SIZE; kernels run
# one at a time under counter collection, so cache hits between concurrent groups are not what the real pipeline sees).
# usage (GPU box, repo root): tools/pipeline_traffic.sh [steps]
set -e
steps=${1:-256}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/pipe_traffic
rm -rf $out && mkdir -p $out
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d $out/$c -o pmc -- python3 bench.py --steps $steps --warmup 32 --no-cpu-baseline --no-cloud-roofline --parity-frames 0 > /dev/null 2>$out/$c.err
  python3 tools/pmc_table.py $(ls $out/$c/*counter_collection.csv $out/$c/*/*counter_collection.csv 2>/dev/null | head -1) > $out/$c.txt
done
cat $out/FETCH_SIZE.txt $out/WRITE_SIZE.txt
