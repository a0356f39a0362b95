#!/bin/bash
# HBM traffic of every kernel of the pipeline (rocprofv3 PMC, separate passes for FETCH_SIZE and WRITE_SIZE as
# MI355X_MICROARCH.md prescribes; kernels run one at a time under counter collection, so cache hits between concurrent
# groups are not what the real pipeline sees).  usage (GPU box, repo root): tools/pipeline_traffic.sh <tag> [workload ...]
# Writes gpurun_out/pipe_traffic/<tag>_pipeline_traffic_<workload>.txt (per-kernel table + one JSON summary line each);
# tools/pipeline_traffic.py merges the JSON lines into profiles/pipeline_traffic.json, which bench.py reads.
tag=${1:-rXX}; shift
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/pipe_traffic
mkdir -p $out
for wl in ${@:-tum_stream synth1280 synth4k}; do
  case $wl in tum_stream) extra="--steps 256 --warmup 32";; *) extra="";; esac
  for c in FETCH_SIZE WRITE_SIZE; do
    rm -rf $out/$c
    rocprofv3 --kernel-trace --pmc $c --output-format csv -d $out/$c -o pmc -- python3 bench.py --workload $wl $extra --no-cpu-baseline --no-cloud-roofline --no-sustained --no-end-to-end --parity-frames 0 --roofline-iters 1 > $out/bench_$c.json 2>$out/$c.err
    python3 tools/pmc_table.py $(ls $out/$c/*counter_collection.csv $out/$c/*/*counter_collection.csv 2>/dev/null | head -1) > $out/$c.txt
    rm -rf $out/$c
  done
  python3 tools/pipeline_traffic.py $wl $out/FETCH_SIZE.txt $out/WRITE_SIZE.txt $out/bench_FETCH_SIZE.json > $out/${tag}_pipeline_traffic_$wl.txt
  tail -3 $out/${tag}_pipeline_traffic_$wl.txt
done
