#!/bin/bash
for f in deplex_amd/lib/variants/rt*.so; do
  for s in "" f; do
  DPX_SKIP=$s DPX_LIB_PATH=$PWD/$f timeout -k 10 200 python bench.py --steps 192 --warmup 32 --no-cpu-baseline --parity-frames 0 --prof-every 1000 --no-cloud-roofline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$f skip=[$s]', d['value'], d['ms_per_step'])"
  done
done
