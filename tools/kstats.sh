#!/bin/bash
# Per-kernel durations of single calls on the large configs: rocprofv3 --kernel-trace --stats around tools/region_time.py.
# usage (on the GPU box, from the repo root): tools/kstats.sh <tag> [4k 1280 ...]
tag=${1:-cur}; shift
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/kstats_$tag
rm -rf $out && mkdir -p $out
for cfg in ${@:-4k 1280}; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $out/st_$cfg -o st -- python3 tools/region_time.py $cfg > $out/region_time_$cfg.txt 2>$out/st_$cfg.err
  f=$(ls $out/st_$cfg/*kernel_stats.csv $out/st_$cfg/*/*kernel_stats.csv 2>/dev/null | head -1)
  [ -n "$f" ] && cp $f $out/${tag}_${cfg}_kernel_stats.csv
  rm -rf $out/st_$cfg
  cat $out/region_time_$cfg.txt
  python3 - "$out/${tag}_${cfg}_kernel_stats.csv" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows[:16]:
    print(f"  {r['Name'][:70]:70s} calls {r['Calls']:>5s} avg_us {float(r['AverageNs'])/1e3:10.1f} total_ms {float(r['TotalDurationNs'])/1e6:9.2f} {r['Percentage']}%")
PY
done
