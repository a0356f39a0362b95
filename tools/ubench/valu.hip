// Micro-benchmarks that shape the plane-fit kernel: VALU issue rates on gfx950.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f2 __attribute__((ext_vector_type(2)));
#define ITERS 4096
template <int MODE>
__global__ __launch_bounds__(256) void k(float* out, float a, float b) {
  float x0 = threadIdx.x * 1e-3f + a, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3, x4 = x0 + 4, x5 = x0 + 5, x6 = x0 + 6, x7 = x0 + 7;
  f2 p0 = {x0, x1}, p1 = {x2, x3}, p2 = {x4, x5}, p3 = {x6, x7};
  f2 bb = {b, b * 1.0001f};
  for (int i = 0; i < ITERS; ++i) {
    if (MODE == 0) {  // 8 independent v_add_f32
      x0 += b; x1 += b; x2 += b; x3 += b; x4 += b; x5 += b; x6 += b; x7 += b;
    } else if (MODE == 1) {  // 4 independent v_pk_add_f32 (8 flops-lanes)
      p0 += bb; p1 += bb; p2 += bb; p3 += bb;
    } else if (MODE == 2) {  // 8 IEEE divides
      x0 = x0 / b; x1 = x1 / b; x2 = x2 / b; x3 = x3 / b; x4 = x4 / b; x5 = x5 / b; x6 = x6 / b; x7 = x7 / b;
    } else if (MODE == 3) {  // 8 reciprocal+2 fma divides
      float r = a;
#define D(x) { float q = x * r; float e = __builtin_fmaf(-q, b, x); x = __builtin_fmaf(e, r, q); }
      D(x0) D(x1) D(x2) D(x3) D(x4) D(x5) D(x6) D(x7)
    } else if (MODE == 4) {  // packed version of mode 3: 4 pk
      f2 r = {a, a};
#define DP(x) { f2 q = x * r; f2 e = __builtin_elementwise_fma(-q, bb, x); x = __builtin_elementwise_fma(e, r, q); }
      DP(p0) DP(p1) DP(p2) DP(p3)
    } else if (MODE == 5) {  // 8 v_mul + 8 v_add (mul-then-add, no contraction)
      x0 = x0 * b + a; x1 = x1 * b + a; x2 = x2 * b + a; x3 = x3 * b + a; x4 = x4 * b + a; x5 = x5 * b + a; x6 = x6 * b + a; x7 = x7 * b + a;
    } else if (MODE == 6) {  // 8 f64 add
      double d0 = x0, d1 = x1, d2 = x2, d3 = x3;
      d0 += b; d1 += b; d2 += b; d3 += b; d0 *= a; d1 *= a; d2 *= a; d3 *= a;
      x0 = d0; x1 = d1; x2 = d2; x3 = d3;
    }
  }
  out[blockIdx.x * 256 + threadIdx.x] = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7 + p0.x + p0.y + p1.x + p1.y + p2.x + p2.y + p3.x + p3.y;
}
template <int MODE>
void run(const char* name, int opsPerIter, float* d) {
  const int blocks = 256 * 8;  // 8 blocks of 256 per CU -> 8 waves per SIMD
  hipEvent_t a, b;
  hipEventCreate(&a); hipEventCreate(&b);
  hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, d, 1.0001f, 1.0003f);
  hipEventRecord(a);
  for (int r = 0; r < 5; ++r) hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, d, 1.0001f, 1.0003f);
  hipEventRecord(b);
  hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b); ms /= 5;
  double lane_ops = double(blocks) * 256 * ITERS * opsPerIter;
  printf("%-34s %8.3f ms  %8.2f T lane-ops/s  (%.2f per clk per CU @2.4GHz)\n", name, ms, lane_ops / ms / 1e9, lane_ops / (ms * 1e-3) / 2.4e9 / 256);
}
int main() {
  float* d; hipMalloc(&d, 256 * 8 * 256 * 4);
  run<0>("v_add_f32 x8", 8, d);
  run<1>("v_pk_add_f32 x4 (8 lanes-ops)", 8, d);
  run<2>("IEEE f32 divide x8", 8, d);
  run<3>("rcp-mul + 2 fma divide x8", 8, d);
  run<4>("packed rcp-mul + 2 pk_fma x4", 8, d);
  run<5>("v_mul + v_add x8 (16 ops)", 16, d);
  run<6>("f64 add+mul x4 (8 ops) + cvts", 8, d);
  return 0;
}
