#!/bin/bash
# Round-4 evidence for profiles/ (run on the GPU box through gpurun from the repo root).  usage: tools/profile_round4.sh <part> [tag]
#   part a: pipeline traffic per tier (PMC), region phases and call times of the large configs, sharded-extractor rate
#   part b: the BASELINE configs as bench lines + rocprofv3 --kernel-trace --stats of the same commands
#   part c: tools/profile_round.sh (plane-fit roofline series, PMC traffic of the plane fit, default bench under the profiler)
part=${1:-a}; tag=${2:-r04}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/prof4_$tag
mkdir -p $out
if [ "$part" = a ]; then
  tools/pipeline_traffic.sh $tag tum_stream synth1280 synth4k > $out/traffic.log 2>&1
  cp gpurun_out/pipe_traffic/${tag}_pipeline_traffic_*.txt $out/ 2>/dev/null
  python3 tools/pipeline_traffic.py --merge $out/pipeline_traffic.json $out/${tag}_pipeline_traffic_*.txt
  python3 tools/cores_phases.py 2160 3840 4 > $out/${tag}_region_phases_4k.txt 2>/dev/null
  python3 tools/region_phases.py 960 1280 8 > $out/${tag}_region_phases_1280.txt 2>/dev/null
  python3 tools/region_time.py 4k 1280 640 > $out/${tag}_region_time.txt 2>/dev/null
  python3 tools/sharded_rate.py 1024 > $out/${tag}_sharded_rate.txt 2>/dev/null
elif [ "$part" = b ]; then
  for w in synth1280:1280 synth640:640x512 synth4k:4k synth4k64:4k64; do
    wl=${w%%:*}; nm=${w##*:}
    python3 bench.py --workload $wl > $out/${tag}_bench_$nm.json 2>$out/bench_$nm.err
    rocprofv3 --kernel-trace --stats --output-format csv -d $out/st_$nm -o st -- python3 bench.py --workload $wl --no-cpu-baseline --no-end-to-end > $out/${tag}_bench_${nm}_profiled.json 2>$out/st_$nm.err
    cp $(ls $out/st_$nm/*kernel_stats.csv $out/st_$nm/*/*kernel_stats.csv 2>/dev/null | head -1) $out/${tag}_bench_${nm}_kernel_stats.csv
    rm -rf $out/st_$nm
    echo "$wl done" >> $out/progress.log
  done
else
  tools/profile_round.sh $tag > $out/profile_round.log 2>&1
  cp gpurun_out/prof_$tag/* $out/ 2>/dev/null
fi
ls $out
