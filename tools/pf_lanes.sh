#!/bin/bash
# Lane utilisation of the depth plane-fit kernel on the TUM stream: executed VALU instructions vs thread-cycles
# (SQ_ACTIVE_INST_VALU counts cycles a VALU instruction is active per wave, SQ_THREAD_CYCLES_VALU the same weighted by active lanes).
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/pf_lanes; rm -rf $out; mkdir -p $out
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES --output-format csv -d $out/a -o pmc -- python3 bench.py --roofline-only --group 8 --roofline-iters 12 > /dev/null 2>$out/a.err
python3 tools/pmc_table.py $(ls $out/a/*counter_collection.csv $out/a/*/*counter_collection.csv 2>/dev/null | head -1) | grep -i "kernel\|planefit"
