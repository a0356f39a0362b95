#!/bin/bash
# times the plane-fit kernel of every experimental build under deplex_amd/lib/variants/
for f in deplex_amd/lib/variants/*.so; do
  echo "== $f"
  for d in 0 1 2; do
  DPX_PF_DIAG=$d DPX_LIB_PATH=$PWD/$f timeout -k 5 120 python tools/pf_breakdown.py run 256 2>&1 | grep DIAG
  done
done
