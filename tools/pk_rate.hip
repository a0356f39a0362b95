// Issue rate of packed f32 VALU operations (v_pk_mul_f32 / v_pk_add_f32 / v_pk_fma_f32) against the plain ones on gfx950:
// 16 independent chains per lane, enough waves to fill the SIMDs.  build: hipcc --offload-arch=gfx950 -O3 tools/pk_rate.hip -o gpurun_out/pk_rate
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float v2f __attribute__((ext_vector_type(2)));
template <int MODE>
__global__ __launch_bounds__(256) void k(float* out, int iters, float seed) {
  float a[16];
  v2f p[16];
  for (int i = 0; i < 16; ++i) { a[i] = seed + i + threadIdx.x; p[i] = v2f{seed + i, seed - i + threadIdx.x}; }
  const float m = 1.0000001f;
  const v2f pm = v2f{1.0000001f, 0.9999999f};
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      if (MODE == 0) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a[i]) : "v"(m));
      if (MODE == 1) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(p[i]) : "v"(pm));
      if (MODE == 2) asm volatile("v_pk_mul_f32 %0, %0, %1 op_sel:[0,1] op_sel_hi:[0,1]" : "+v"(p[i]) : "v"(pm));
      if (MODE == 3) asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(p[i]) : "v"(pm));
      if (MODE == 4) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(a[i]) : "v"(m));
      if (MODE == 5) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(p[i]) : "v"(pm));
    }
  }
  float s = 0.f;
  for (int i = 0; i < 16; ++i) s += a[i] + p[i].x + p[i].y;
  out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int MODE>
void run(const char* name, float* d) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  const int blocks = 256 * 8, iters = 4096;
  hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, d, 16, 1.0f);
  hipEventRecord(e0);
  hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, d, iters, 1.0f);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms = 0.f;
  hipEventElapsedTime(&ms, e0, e1);
  const double insts = double(blocks) * 4 * iters * 16;  // wave instructions
  printf("%-28s %8.3f ms  %7.2f G wave-instr/s\n", name, ms, insts / ms / 1e6);
}
int main() {
  float* d;
  hipMalloc(&d, 256 * 8 * 256 * sizeof(float));
  run<0>("v_mul_f32", d);
  run<1>("v_pk_mul_f32", d);
  run<2>("v_pk_mul_f32 op_sel", d);
  run<3>("v_pk_fma_f32", d);
  run<4>("v_fma_f32", d);
  run<5>("v_pk_add_f32", d);
  return 0;
}
