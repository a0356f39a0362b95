"""Issue budget of the depth plane-fit kernel from measured data only:
  executed wave-instructions by class (rocprofv3 --pmc SQ_INSTS_VALU_*, profiles/<tag>_planefit_insts_pass{1,2,3}.txt)
  x SIMD cycles per wave-instruction of that class at 4 waves per SIMD (tools/ubench/mix.hip, profiles/<tag>_ubench_mix.txt)
against the wave lifetime the same counters report (SQ_WAVE_CYCLES) and the kernel's duration.
usage: isa_budget.py [tag (default r03)] [kernel us per launch from the bench line]"""
import re, sys
from pathlib import Path
P = Path(__file__).resolve().parents[1] / "profiles"
tag = sys.argv[1] if len(sys.argv) > 1 else "r03"
kernel_us = float(sys.argv[2]) if len(sys.argv) > 2 else None
cnt = {}
for i in (1, 2, 3):
    lines = (P / f"{tag}_planefit_insts_pass{i}.txt").read_text().splitlines()
    hdr, row = lines[0].split(), lines[1]
    vals = row.split()[-len(hdr) + 2:]
    names = {"SQ_INSTS_VALU": "VALU", "STS_VALU_ADD_F32": "ADD_F32", "Q_INSTS_VALU_CVT": "CVT", "STS_VALU_FMA_F32": "FMA_F32", "INSTS_VALU_INT32": "INT32",
             "STS_VALU_MUL_F32": "MUL_F32", "S_VALU_TRANS_F32": "TRANS_F32", "SQ_WAVES": "WAVES", "SQ_INSTS_SALU": "SALU", "SQ_INSTS_SMEM": "SMEM",
             "STS_VALU_ADD_F64": "ADD_F64", "STS_VALU_FMA_F64": "FMA_F64", "INSTS_VALU_INT64": "INT64", "STS_VALU_MUL_F64": "MUL_F64",
             "S_VALU_TRANS_F64": "TRANS_F64", "SQ_INSTS_VMEM_RD": "VMEM_RD", "_ACTIVE_INST_ANY": "ACTIVE_ANY", "ACTIVE_INST_VALU": "ACTIVE_VALU",
             "SQ_BUSY_CYCLES": "BUSY", "SQ_INSTS_VMEM_WR": "VMEM_WR", "READ_CYCLES_VALU": "THREAD_CYCLES_VALU", "SQ_WAIT_ANY": "WAIT_ANY",
             "SQ_WAIT_INST_ANY": "WAIT_INST_ANY", "SQ_WAVE_CYCLES": "WAVE_CYCLES"}
    for h, v in zip(hdr[2:], vals):
        cnt[names[h]] = float(v)
mix = {}
for l in (P / f"{tag}_ubench_mix.txt").read_text().splitlines():
    m = re.match(r"(.+?)\s+\| w1:\s*([\d.]+) \| w2:\s*([\d.]+) \| w4:\s*([\d.]+) \| w8:\s*([\d.]+)", l)
    if m:
        mix[m.group(1).strip()] = float(m.group(4))
W = cnt["WAVES"]
per = {k: v / W for k, v in cnt.items()}
cost = {"ADD_F32": mix["v_mul_f32 -> v_add_f32 (dependent pairs)"], "MUL_F32": mix["v_mul_f32 -> v_add_f32 (dependent pairs)"], "FMA_F32": mix["v_fma_f32"],
        "TRANS_F32": mix["v_rcp_f32 (transcendental)"], "CVT": mix["v_cvt_f32_u32 sdwa"], "INT32": mix["v_sad_u16 / v_pk_add_u16 / v_pk_min_u16"],
        "ADD_F64": mix["v_add_f64"], "MUL_F64": mix["v_mul_f64"], "FMA_F64": mix["v_fma_f64"], "TRANS_F64": mix["v_rcp_f64 (transcendental)"],
        "INT64": mix["v_add_f64"]}
classified = sum(per[k] for k in cost)
other = per["VALU"] - classified
print(f"waves per launch {W:.0f} (512 frames x 3072 cells / 64); executed wave-instructions per wave (= per cell, lanes are cells)")
print(f"{'class':12s} {'instr/wave':>10s} {'cyc/instr @4 waves/SIMD':>24s} {'SIMD cycles':>12s}")
tot = 0.0
for k in cost:
    c = per[k] * cost[k]
    tot += c
    print(f"{k:12s} {per[k]:10.1f} {cost[k]:24.2f} {c:12.0f}")
oc = other * mix["v_cvt_f32_u32 sdwa"]
tot += oc
print(f"{'other VALU':12s} {other:10.1f} {mix['v_cvt_f32_u32 sdwa']:24.2f} {oc:12.0f}   (v_mov / v_cmp / v_cndmask / logic: priced like the 32-bit integer and conversion classes)")
print(f"{'all VALU':12s} {per['VALU']:10.1f} {'':24s} {tot:12.0f}   = issue budget per wave; x 4 waves per SIMD = {4 * tot:.0f} SIMD cycles per generation")
print(f"scalar side per wave: SALU {per['SALU']:.0f}, SMEM {per['SMEM']:.0f}, VMEM reads {per['VMEM_RD']:.0f}, VMEM writes {per['VMEM_WR']:.0f} (own issue ports)")
life = per["WAVE_CYCLES"] * 4
print(f"wave lifetime (SQ_WAVE_CYCLES x 4): {life:.0f} cycles = executing {per['ACTIVE_ANY'] * 4:.0f} + parked on s_waitcnt {per['WAIT_ANY'] * 4:.0f} + "
      f"waiting to issue {per['WAIT_INST_ANY'] * 4:.0f}")
print(f"SIMD cycles per VALU wave-instruction in the steady state: achieved {life / (4 * per['VALU']):.2f}, budget {tot / per['VALU']:.2f} "
      f"-> the kernel issues at {4 * tot / life:.2f} of what the chip sustains for this instruction mix")
if kernel_us:
    gens = W / (256 * 4 * 4)
    print(f"kernel: {kernel_us:.1f} us per launch = {kernel_us * 2400:.0f} cycles at 2.4 GHz; {gens:.1f} generations of 4 waves per SIMD x budget "
          f"{4 * tot:.0f} = {gens * 4 * tot:.0f} cycles -> {gens * 4 * tot / (kernel_us * 2400):.2f} of the launch (the rest: first-row loads of every generation, "
          f"the uneven last generation, launch ramp)")
