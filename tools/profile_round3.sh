#!/bin/bash
# Round-3 additions to tools/profile_round.sh (run on the GPU box through gpurun from the repo root): usage tools/profile_round3.sh r03
#   * the three large BASELINE configs as bench lines, each also under rocprofv3 --kernel-trace --stats;
#   * executed-instruction counters of the depth plane-fit kernel (SQ_INSTS_VALU_* by class, three --pmc passes of at most
#     8 SQ counters) for profiles/<tag>_planefit_isa_budget.md, and the per-class issue costs of tools/ubench/mix.hip;
#   * region-growing phase stamps of the noisiest large frames.
set -e
tag=${1:-r03}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/prof3_$tag
rm -rf $out && mkdir -p $out
for w in synth1280:1280 synth640:640x512 synth4k:4k; do
  wl=${w%%:*}; nm=${w##*:}
  python3 bench.py --workload $wl > $out/${tag}_bench_$nm.json 2>$out/bench_$nm.err
  rocprofv3 --kernel-trace --stats --output-format csv -d $out/st_$nm -o st -- python3 bench.py --workload $wl --no-cpu-baseline --no-end-to-end > $out/${tag}_bench_${nm}_profiled.json 2>$out/st_$nm.err
  cp $(ls $out/st_$nm/*kernel_stats.csv $out/st_$nm/*/*kernel_stats.csv 2>/dev/null | head -1) $out/${tag}_bench_${nm}_kernel_stats.csv
  rm -rf $out/st_$nm
done
i=0
for set in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_CVT SQ_INSTS_VALU_INT32" \
           "SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU_INT64 SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_THREAD_CYCLES_VALU SQ_INSTS_VMEM_WR"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $set --output-format csv -d $out/ic_$i -o ic -- python3 bench.py --roofline-only --group 8 --roofline-iters 12 > /dev/null 2>$out/ic_$i.err
  python3 tools/pmc_table.py $(ls $out/ic_$i/*counter_collection.csv $out/ic_$i/*/*counter_collection.csv 2>/dev/null | head -1) | grep -i "kernel\|planefit" > $out/${tag}_planefit_insts_pass$i.txt
  rm -rf $out/ic_$i
done
[ -x tools/ubench/mix_bench ] || hipcc --offload-arch=gfx950 -O3 -o tools/ubench/mix_bench tools/ubench/mix.hip || echo "mix_bench not built: skipping"
[ -x tools/ubench/mix_bench ] && tools/ubench/mix_bench > $out/${tag}_ubench_mix.txt 2>&1
python3 tools/region_phases.py 2160 3840 4 > $out/${tag}_region_phases_4k.txt 2>/dev/null
python3 tools/region_phases.py 960 1280 8 > $out/${tag}_region_phases_1280.txt 2>/dev/null
python3 tools/region_time.py 4k 1280 640 > $out/${tag}_region_time.txt 2>/dev/null
ls -la $out
