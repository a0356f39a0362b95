#!/bin/bash
# The driver's run shape (bench.py --steps 20 --warmup 5) under a few pipeline shapes / kernel switches; median of 5 runs each.
mkdir -p gpurun_out/short
run() { name=$1; shift
  for i in 1 2 3 4 5; do
    env "$@" python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/short/${name}_$i.json 2>gpurun_out/short/${name}_$i.err || echo FAIL $name $i
  done
  python - "$name" <<'PY'
import json,sys,statistics
v=[]
for i in range(1,6):
    try: v.append(json.loads(open(f'gpurun_out/short/{sys.argv[1]}_{i}.json').read().strip().splitlines()[-1])['value'])
    except Exception as e: pass
print(sys.argv[1], 'median', round(statistics.median(v)) if v else None, [round(x) for x in v], flush=True)
PY
}
run base X=1
run s4g5 DPX_BENCH_STREAMS=4 DPX_BENCH_GROUP=5
run s16g4 DPX_BENCH_STREAMS=16 DPX_BENCH_GROUP=4
run s16g10 DPX_BENCH_STREAMS=16 DPX_BENCH_GROUP=10
run s16g7 DPX_BENCH_STREAMS=16 DPX_BENCH_GROUP=7
run fused DPX_FUSED_LABELS=1
run lab0 DPX_LAB_RESIDENT=0
run lab2 DPX_LAB_RESIDENT=2
run base2 X=1
