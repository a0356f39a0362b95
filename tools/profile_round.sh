#!/bin/bash
# rocprofv3 evidence for profiles/: run on the GPU box through gpurun from the repo root.  usage: tools/profile_round.sh r01e
set -e
tag=${1:-rXX}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/prof_$tag
rm -rf $out && mkdir -p $out
for inp in depth cloud; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $out/ro_$inp -o ro -- python3 bench.py --roofline-only --roofline-input $inp > $out/${tag}_roofline_only_${inp}_bench.json 2>$out/ro_$inp.err
  cp $(ls $out/ro_$inp/*kernel_stats.csv $out/ro_$inp/*/*kernel_stats.csv 2>/dev/null | head -1) $out/${tag}_roofline_only_${inp}_kernel_stats.csv
  for c in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --kernel-trace --pmc $c --output-format csv -d $out/pmc_${inp}_$c -o pmc -- python3 bench.py --roofline-only --roofline-input $inp --roofline-iters 10 > /dev/null 2>$out/pmc_${inp}_$c.err
    python3 tools/pmc_sum.py $(ls $out/pmc_${inp}_$c/*counter_collection.csv $out/pmc_${inp}_$c/*/*counter_collection.csv 2>/dev/null | head -1) $c planefit > $out/${tag}_pmc_${inp}_$c.txt
  done
done
rocprofv3 --kernel-trace --stats --output-format csv -d $out/def -o def -- python3 bench.py --no-cpu-baseline > $out/${tag}_bench_default.json 2>$out/def.err
cp $(ls $out/def/*kernel_stats.csv $out/def/*/*kernel_stats.csv 2>/dev/null | head -1) $out/${tag}_bench_default_kernel_stats.csv
python3 tools/trace_streams.py $(ls $out/def/*kernel_trace.csv $out/def/*/*kernel_trace.csv 2>/dev/null | head -1) > $out/${tag}_bench_default_concurrency.txt
python3 tools/make_traffic_json.py $tag $out/${tag}_pmc_depth_FETCH_SIZE.txt $out/${tag}_pmc_depth_WRITE_SIZE.txt $out/${tag}_pmc_cloud_FETCH_SIZE.txt $out/${tag}_pmc_cloud_WRITE_SIZE.txt $out/${tag}_roofline_only_depth_bench.json $out/${tag}_roofline_only_cloud_bench.json > $out/planefit_traffic.json
# the driver's own protocol (20 timed steps after 5 warm-up steps)
python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline > $out/${tag}_bench_steps20.json 2>>$out/def.err
rm -rf $out/ro_* $out/pmc_*_*SIZE $out/def
ls -la $out
