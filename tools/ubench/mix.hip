// Micro-benchmark: SIMD cycles per wave-instruction of the instruction classes of the depth plane-fit kernel
// (k_planefit_depth) on gfx950, at 1 / 2 / 4 / 8 waves per SIMD.  Eight independent dependency chains per lane, cycles
// from the shader clock inside the kernel (per wave, averaged).  The per-class costs at 4 waves per SIMD (the kernel's
// occupancy), multiplied with the executed-instruction counts of the kernel (SQ_INSTS_VALU_* counters), give the issue
// budget of profiles/r03_planefit_isa_budget.md.
// Build: hipcc --offload-arch=gfx950 -O3 -o mix_bench mix.hip ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define ITERS 1024

#define F32_2(op) \
  asm volatile(op " %0, %0, %8\n" op " %1, %1, %8\n" op " %2, %2, %8\n" op " %3, %3, %8\n" op " %4, %4, %8\n" op " %5, %5, %8\n" op " %6, %6, %8\n" op " %7, %7, %8" \
               : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(b))
#define F32_3(op) \
  asm volatile(op " %0, %0, %8, %9\n" op " %1, %1, %8, %9\n" op " %2, %2, %8, %9\n" op " %3, %3, %8, %9\n" op " %4, %4, %8, %9\n" op " %5, %5, %8, %9\n" op " %6, %6, %8, %9\n" op " %7, %7, %8, %9" \
               : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(b), "v"(a))
#define F32_1(op) \
  asm volatile(op " %0, %0\n" op " %1, %1\n" op " %2, %2\n" op " %3, %3\n" op " %4, %4\n" op " %5, %5\n" op " %6, %6\n" op " %7, %7" \
               : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7))
#define F64_2(op) \
  asm volatile(op " %0, %0, %8\n" op " %1, %1, %8\n" op " %2, %2, %8\n" op " %3, %3, %8\n" op " %4, %4, %8\n" op " %5, %5, %8\n" op " %6, %6, %8\n" op " %7, %7, %8" \
               : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3), "+v"(d4), "+v"(d5), "+v"(d6), "+v"(d7) : "v"(db))
#define F64_3(op) \
  asm volatile(op " %0, %0, %8, %9\n" op " %1, %1, %8, %9\n" op " %2, %2, %8, %9\n" op " %3, %3, %8, %9\n" op " %4, %4, %8, %9\n" op " %5, %5, %8, %9\n" op " %6, %6, %8, %9\n" op " %7, %7, %8, %9" \
               : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3), "+v"(d4), "+v"(d5), "+v"(d6), "+v"(d7) : "v"(db), "v"(da))
#define F64_1(op) \
  asm volatile(op " %0, %0\n" op " %1, %1\n" op " %2, %2\n" op " %3, %3\n" op " %4, %4\n" op " %5, %5\n" op " %6, %6\n" op " %7, %7" \
               : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3), "+v"(d4), "+v"(d5), "+v"(d6), "+v"(d7))

template <int MODE>
__global__ __launch_bounds__(256) void k(float* out, unsigned long long* cyc, float a, float b) {
  float x0 = threadIdx.x * 1e-3f + a, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3, x4 = x0 + 4, x5 = x0 + 5, x6 = x0 + 6, x7 = x0 + 7;
  double d0 = x0, d1 = x1, d2 = x2, d3 = x3, d4 = x4, d5 = x5, d6 = x6, d7 = x7, da = a, db = b;
  unsigned u0 = threadIdx.x, ub = 0x00010001u;
  unsigned long long t0 = __builtin_readcyclecounter();
#pragma unroll 2
  for (int i = 0; i < ITERS; ++i) {
    if (MODE == 0) F32_2("v_add_f32");
    else if (MODE == 1) F32_2("v_mul_f32");
    else if (MODE == 2) F32_3("v_fma_f32");
    else if (MODE == 3) F64_2("v_add_f64");
    else if (MODE == 4) F64_2("v_mul_f64");
    else if (MODE == 5) F64_3("v_fma_f64");
    else if (MODE == 6) F64_1("v_rcp_f64");
    else if (MODE == 7) F64_1("v_rsq_f64");
    else if (MODE == 8) F32_1("v_rcp_f32");
    else if (MODE == 9) {  // v_cndmask_b32 (vcc)
      asm volatile("v_cndmask_b32 %0, %0, %8, vcc\nv_cndmask_b32 %1, %1, %8, vcc\nv_cndmask_b32 %2, %2, %8, vcc\nv_cndmask_b32 %3, %3, %8, vcc\n"
                   "v_cndmask_b32 %4, %4, %8, vcc\nv_cndmask_b32 %5, %5, %8, vcc\nv_cndmask_b32 %6, %6, %8, vcc\nv_cndmask_b32 %7, %7, %8, vcc"
                   : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(b) : "vcc");
    } else if (MODE == 10) {  // v_sad_u16 + v_pk_add_u16 + v_pk_min_u16 (the integer side of a pixel pair), x2 + 2
      asm volatile("v_sad_u16 %0, %1, 0, %0\nv_pk_add_u16 %2, %2, %3\nv_pk_min_u16 %4, %1, %3\nv_sad_u16 %5, %1, 0, %5\nv_pk_add_u16 %6, %6, %3\nv_pk_min_u16 %7, %1, %3\n"
                   "v_sad_u16 %0, %1, 0, %0\nv_pk_add_u16 %2, %2, %3"
                   : "+v"(u0), "+v"(ub), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7));
    } else if (MODE == 11) {  // f64 division helpers: div_scale, div_fmas, div_fixup, ldexp (2 each)
      asm volatile("v_div_scale_f64 %0, vcc, %0, %8, %0\nv_div_scale_f64 %1, vcc, %1, %8, %1\nv_div_fmas_f64 %2, %2, %8, %9\nv_div_fmas_f64 %3, %3, %8, %9\n"
                   "v_div_fixup_f64 %4, %4, %8, %9\nv_div_fixup_f64 %5, %5, %8, %9\nv_ldexp_f64 %6, %6, 1\nv_ldexp_f64 %7, %7, 1"
                   : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3), "+v"(d4), "+v"(d5), "+v"(d6), "+v"(d7) : "v"(db), "v"(da) : "vcc");
    } else if (MODE == 12) {  // conversions f32 <-> f64 (4 + 4)
      asm volatile("v_cvt_f64_f32 %0, %4\nv_cvt_f64_f32 %1, %5\nv_cvt_f64_f32 %2, %6\nv_cvt_f64_f32 %3, %7\nv_cvt_f32_f64 %4, %0\nv_cvt_f32_f64 %5, %1\nv_cvt_f32_f64 %6, %2\nv_cvt_f32_f64 %7, %3"
                   : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7));
    } else if (MODE == 13) {  // dependent mul -> add pairs (the moment chains): 8 mul + 8 add
      float t0_, t1_, t2_, t3_, t4_, t5_, t6_, t7_;
      asm volatile("v_mul_f32 %0, %8, %9\nv_mul_f32 %1, %8, %9\nv_mul_f32 %2, %8, %9\nv_mul_f32 %3, %8, %9\nv_mul_f32 %4, %8, %9\nv_mul_f32 %5, %8, %9\nv_mul_f32 %6, %8, %9\nv_mul_f32 %7, %8, %9"
                   : "=v"(t0_), "=v"(t1_), "=v"(t2_), "=v"(t3_), "=v"(t4_), "=v"(t5_), "=v"(t6_), "=v"(t7_) : "v"(a), "v"(b));
      asm volatile("v_add_f32 %0, %0, %8\nv_add_f32 %1, %1, %9\nv_add_f32 %2, %2, %10\nv_add_f32 %3, %3, %11\nv_add_f32 %4, %4, %12\nv_add_f32 %5, %5, %13\nv_add_f32 %6, %6, %14\nv_add_f32 %7, %7, %15"
                   : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7)
                   : "v"(t0_), "v"(t1_), "v"(t2_), "v"(t3_), "v"(t4_), "v"(t5_), "v"(t6_), "v"(t7_));
    } else if (MODE == 14) {  // v_cvt_f32_u32 sdwa word select
      asm volatile("v_cvt_f32_u32_sdwa %0, %8 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1\nv_cvt_f32_u32_sdwa %1, %8 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_0\n"
                   "v_cvt_f32_u32_sdwa %2, %8 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1\nv_cvt_f32_u32_sdwa %3, %8 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_0\n"
                   "v_cvt_f32_u32_sdwa %4, %8 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1\nv_cvt_f32_u32_sdwa %5, %8 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_0\n"
                   "v_cvt_f32_u32_sdwa %6, %8 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1\nv_cvt_f32_u32_sdwa %7, %8 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_0"
                   : "=v"(x0), "=v"(x1), "=v"(x2), "=v"(x3), "=v"(x4), "=v"(x5), "=v"(x6), "=v"(x7) : "v"(u0));
    }
  }
  unsigned long long t1 = __builtin_readcyclecounter();
  if ((threadIdx.x & 63) == 0) cyc[(blockIdx.x * 256 + threadIdx.x) >> 6] = t1 - t0;
  out[blockIdx.x * 256 + threadIdx.x] = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7 + static_cast<float>(d0 + d1 + d2 + d3 + d4 + d5 + d6 + d7) + u0 + ub;
}

template <int MODE>
void run(const char* name, int instrPerIter, float* d, unsigned long long* dc) {
  printf("%-52s", name);
  for (int wps : {1, 2, 4, 8}) {
    const int blocks = 256 * wps;  // 256-thread blocks: 4 waves, one per SIMD; wps blocks per CU
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, d, dc, 1.0001f, 1.0003f);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, d, dc, 1.0001f, 1.0003f);
    hipDeviceSynchronize();
    std::vector<unsigned long long> c(blocks * 4);
    hipMemcpy(c.data(), dc, c.size() * 8, hipMemcpyDeviceToHost);
    double s = 0; for (auto v : c) s += double(v);
    const double cyc_per_wave_iter = s / c.size() / ITERS;
    printf(" | w%d: %5.2f", wps, cyc_per_wave_iter / (instrPerIter * wps));  // SIMD cycles per wave-instruction
  }
  printf("\n");
}
int main() {
  float* d; hipMalloc(&d, 256 * 8 * 256 * 4);
  unsigned long long* dc; hipMalloc(&dc, 256 * 8 * 4 * 8);
  printf("SIMD cycles per wave-instruction at w waves per SIMD (shader clock)\n");
  run<0>("v_add_f32", 8, d, dc);
  run<1>("v_mul_f32", 8, d, dc);
  run<2>("v_fma_f32", 8, d, dc);
  run<13>("v_mul_f32 -> v_add_f32 (dependent pairs)", 16, d, dc);
  run<14>("v_cvt_f32_u32 sdwa", 8, d, dc);
  run<10>("v_sad_u16 / v_pk_add_u16 / v_pk_min_u16", 8, d, dc);
  run<9>("v_cndmask_b32", 8, d, dc);
  run<8>("v_rcp_f32 (transcendental)", 8, d, dc);
  run<3>("v_add_f64", 8, d, dc);
  run<4>("v_mul_f64", 8, d, dc);
  run<5>("v_fma_f64", 8, d, dc);
  run<6>("v_rcp_f64 (transcendental)", 8, d, dc);
  run<7>("v_rsq_f64 (transcendental)", 8, d, dc);
  run<11>("v_div_scale/fmas/fixup_f64, v_ldexp_f64", 8, d, dc);
  run<12>("v_cvt_f64_f32 / v_cvt_f32_f64", 8, d, dc);
  return 0;
}
