#!/bin/bash
# marginal cost of each pipeline stage in the bench (timing-only: DPX_SKIP builds produce garbage labels)
# usage: tools/stage_cost.sh [extra bench args]; environment switches (DPX_FUSED_LABELS, ...) pass through
for s in "" F r l Fr Fl rl; do
  DPX_LIB_PATH=$PWD/deplex_amd/lib/variants/diag.so DPX_SKIP=$s timeout -k 10 200 python bench.py --steps 400 --warmup 40 --no-cpu-baseline --parity-frames 0 --prof-every 1000 --no-cloud-roofline "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('skip=[$s]', d['value'], d['ms_per_step'])"
done
