// Micro-benchmark: issue cost of packed-f32 VALU instructions on gfx950 against their scalar forms, per wave count.
// Every mode runs 8 independent dependency chains per lane (scalar) or 4 register pairs (packed), so one wave alone
// can already issue back to back; cycles come from s_memtime inside the kernel (per wave, averaged), not from wall time.
// Build: hipcc --offload-arch=gfx950 -O3 -o pk_bench pk.hip ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f2 __attribute__((ext_vector_type(2)));
#define ITERS 2048

#define PK2(op, d, a, b) asm volatile(op " %0, %1, %2" : "=v"(d) : "v"(a), "v"(b))
#define PK3(op, d, a, b, c) asm volatile(op " %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(c))

template <int MODE>
__global__ __launch_bounds__(256) void k(float* out, unsigned long long* cyc, float a, float b) {
  float x0 = threadIdx.x * 1e-3f + a, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3, x4 = x0 + 4, x5 = x0 + 5, x6 = x0 + 6, x7 = x0 + 7;
  f2 p0 = {x0, x1}, p1 = {x2, x3}, p2 = {x4, x5}, p3 = {x6, x7};
  f2 bb = {b, b * 1.0001f}, aa = {a, a * 1.0001f};
  unsigned long long t0 = __builtin_readcyclecounter();
#pragma unroll 4
  for (int i = 0; i < ITERS; ++i) {
    if (MODE == 0) {  // 8 v_add_f32
      PK2("v_add_f32", x0, x0, b); PK2("v_add_f32", x1, x1, b); PK2("v_add_f32", x2, x2, b); PK2("v_add_f32", x3, x3, b);
      PK2("v_add_f32", x4, x4, b); PK2("v_add_f32", x5, x5, b); PK2("v_add_f32", x6, x6, b); PK2("v_add_f32", x7, x7, b);
    } else if (MODE == 1) {  // 4 v_pk_add_f32
      PK2("v_pk_add_f32", p0, p0, bb); PK2("v_pk_add_f32", p1, p1, bb); PK2("v_pk_add_f32", p2, p2, bb); PK2("v_pk_add_f32", p3, p3, bb);
    } else if (MODE == 2) {  // 4 v_pk_mul_f32
      PK2("v_pk_mul_f32", p0, p0, bb); PK2("v_pk_mul_f32", p1, p1, bb); PK2("v_pk_mul_f32", p2, p2, bb); PK2("v_pk_mul_f32", p3, p3, bb);
    } else if (MODE == 3) {  // 4 v_pk_fma_f32
      PK3("v_pk_fma_f32", p0, p0, bb, aa); PK3("v_pk_fma_f32", p1, p1, bb, aa); PK3("v_pk_fma_f32", p2, p2, bb, aa); PK3("v_pk_fma_f32", p3, p3, bb, aa);
    } else if (MODE == 4) {  // 8 v_fma_f32
      PK3("v_fma_f32", x0, x0, b, a); PK3("v_fma_f32", x1, x1, b, a); PK3("v_fma_f32", x2, x2, b, a); PK3("v_fma_f32", x3, x3, b, a);
      PK3("v_fma_f32", x4, x4, b, a); PK3("v_fma_f32", x5, x5, b, a); PK3("v_fma_f32", x6, x6, b, a); PK3("v_fma_f32", x7, x7, b, a);
    } else if (MODE == 5) {  // 8 v_mul_f32
      PK2("v_mul_f32", x0, x0, b); PK2("v_mul_f32", x1, x1, b); PK2("v_mul_f32", x2, x2, b); PK2("v_mul_f32", x3, x3, b);
      PK2("v_mul_f32", x4, x4, b); PK2("v_mul_f32", x5, x5, b); PK2("v_mul_f32", x6, x6, b); PK2("v_mul_f32", x7, x7, b);
    } else if (MODE == 6) {  // moment-like: pk_mul into temp then pk_add onto accumulator (4 pairs -> 8 pk instr)
      f2 t0_, t1_, t2_, t3_;
      PK2("v_pk_mul_f32", t0_, aa, bb); PK2("v_pk_mul_f32", t1_, aa, bb); PK2("v_pk_mul_f32", t2_, aa, bb); PK2("v_pk_mul_f32", t3_, aa, bb);
      PK2("v_pk_add_f32", p0, p0, t0_); PK2("v_pk_add_f32", p1, p1, t1_); PK2("v_pk_add_f32", p2, p2, t2_); PK2("v_pk_add_f32", p3, p3, t3_);
    } else if (MODE == 7) {  // scalar form of mode 6: 8 mul + 8 add
      float t0_, t1_, t2_, t3_, t4_, t5_, t6_, t7_;
      PK2("v_mul_f32", t0_, a, b); PK2("v_mul_f32", t1_, a, b); PK2("v_mul_f32", t2_, a, b); PK2("v_mul_f32", t3_, a, b);
      PK2("v_mul_f32", t4_, a, b); PK2("v_mul_f32", t5_, a, b); PK2("v_mul_f32", t6_, a, b); PK2("v_mul_f32", t7_, a, b);
      PK2("v_add_f32", x0, x0, t0_); PK2("v_add_f32", x1, x1, t1_); PK2("v_add_f32", x2, x2, t2_); PK2("v_add_f32", x3, x3, t3_);
      PK2("v_add_f32", x4, x4, t4_); PK2("v_add_f32", x5, x5, t5_); PK2("v_add_f32", x6, x6, t6_); PK2("v_add_f32", x7, x7, t7_);
    } else if (MODE == 8) {  // 8 v_cvt_f32_u32
      unsigned u = __float_as_uint(a);
      PK2("v_add_f32", x0, x0, b);
      asm volatile("v_cvt_f32_u32_e32 %0, %1" : "=v"(x1) : "v"(u)); asm volatile("v_cvt_f32_u32_e32 %0, %1" : "=v"(x2) : "v"(u));
      asm volatile("v_cvt_f32_u32_e32 %0, %1" : "=v"(x3) : "v"(u)); asm volatile("v_cvt_f32_u32_e32 %0, %1" : "=v"(x4) : "v"(u));
      asm volatile("v_cvt_f32_u32_e32 %0, %1" : "=v"(x5) : "v"(u)); asm volatile("v_cvt_f32_u32_e32 %0, %1" : "=v"(x6) : "v"(u));
      asm volatile("v_cvt_f32_u32_e32 %0, %1" : "=v"(x7) : "v"(u)); asm volatile("v_cvt_f32_u32_e32 %0, %1" : "=v"(x1) : "v"(u));
    } else if (MODE == 9) {  // 8 v_cvt_f32_u32 sdwa word select
      unsigned u = __float_as_uint(a);
      asm volatile("v_cvt_f32_u32_sdwa %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1" : "=v"(x0) : "v"(u));
      asm volatile("v_cvt_f32_u32_sdwa %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_0" : "=v"(x1) : "v"(u));
      asm volatile("v_cvt_f32_u32_sdwa %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1" : "=v"(x2) : "v"(u));
      asm volatile("v_cvt_f32_u32_sdwa %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_0" : "=v"(x3) : "v"(u));
      asm volatile("v_cvt_f32_u32_sdwa %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1" : "=v"(x4) : "v"(u));
      asm volatile("v_cvt_f32_u32_sdwa %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_0" : "=v"(x5) : "v"(u));
      asm volatile("v_cvt_f32_u32_sdwa %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1" : "=v"(x6) : "v"(u));
      asm volatile("v_cvt_f32_u32_sdwa %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_0" : "=v"(x7) : "v"(u));
    } else if (MODE == 10) {  // 8 f64 fma
      double d0 = x0, d1 = x1, d2 = x2, d3 = x3; double da = a, db = b;
      for (int j = 0; j < 2; ++j) {
        asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(d0) : "v"(db), "v"(da)); asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(d1) : "v"(db), "v"(da));
        asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(d2) : "v"(db), "v"(da)); asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(d3) : "v"(db), "v"(da));
      }
      x0 = d0; x1 = d1; x2 = d2; x3 = d3;
    }
  }
  unsigned long long t1 = __builtin_readcyclecounter();
  if ((threadIdx.x & 63) == 0) cyc[(blockIdx.x * 256 + threadIdx.x) >> 6] = t1 - t0;
  out[blockIdx.x * 256 + threadIdx.x] = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7 + p0.x + p0.y + p1.x + p1.y + p2.x + p2.y + p3.x + p3.y;
}

template <int MODE>
void run(const char* name, int instrPerIter, int laneOpsPerIter, float* d, unsigned long long* dc) {
  printf("%-44s", name);
  for (int wps : {1, 2, 4, 8}) {
    const int blocks = 256 * wps;  // 256-thread blocks: 4 waves, one per SIMD; wps blocks per CU
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, d, dc, 1.0001f, 1.0003f);
    hipEventRecord(a);
    for (int r = 0; r < 3; ++r) hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, d, dc, 1.0001f, 1.0003f);
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b); ms /= 3;
    std::vector<unsigned long long> c(blocks * 4);
    hipMemcpy(c.data(), dc, c.size() * 8, hipMemcpyDeviceToHost);
    double s = 0; for (auto v : c) s += double(v);
    double cyc_per_wave_iter = s / c.size() / ITERS;
    // SIMD-cycles per instruction: a wave's loop time divided by (instructions of all co-resident waves on that SIMD)
    double simd_cyc_per_instr = cyc_per_wave_iter / (instrPerIter * wps);
    double tops = double(blocks) * 256 * ITERS * laneOpsPerIter / ms / 1e9;
    printf(" | w%d: %5.2f cyc/instr %6.2f Tlop/s", wps, simd_cyc_per_instr, tops);
    hipEventDestroy(a); hipEventDestroy(b);
  }
  printf("\n");
}
int main() {
  float* d; hipMalloc(&d, 256 * 8 * 256 * 4);
  unsigned long long* dc; hipMalloc(&dc, 256 * 8 * 4 * 8);
  run<0>("v_add_f32 x8", 8, 8, d, dc);
  run<5>("v_mul_f32 x8", 8, 8, d, dc);
  run<4>("v_fma_f32 x8", 8, 8, d, dc);
  run<1>("v_pk_add_f32 x4 (8 lane-ops)", 4, 8, d, dc);
  run<2>("v_pk_mul_f32 x4 (8 lane-ops)", 4, 8, d, dc);
  run<3>("v_pk_fma_f32 x4 (8 lane-ops)", 4, 8, d, dc);
  run<6>("pk_mul->pk_add x4+4 (16 lane-ops)", 8, 16, d, dc);
  run<7>("mul->add x8+8 (16 lane-ops)", 16, 16, d, dc);
  run<8>("v_cvt_f32_u32 x8 + 1 add", 9, 9, d, dc);
  run<9>("v_cvt_f32_u32_sdwa x8", 8, 8, d, dc);
  run<10>("v_fma_f64 x8 (+cvts)", 8, 8, d, dc);
  return 0;
}
