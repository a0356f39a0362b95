// icache — does instruction-cache capacity couple kernels that share a CU?  (diagnostic, not part of the product)
//   hipcc --offload-arch=gfx950 -O3 -o tools/ubench/icache_bench tools/ubench/icache.hip && tools/ubench/icache_bench
// Victim: straight-line VALU code of ~8 / 16 / 32 KB, 4 waves per SIMD, looped.  Polluter: one wave per CU walking
// ~48 KB of straight-line scalar code on another stream.  Prints the victim's time alone and next to the polluter.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
#define R4(x) x x x x
#define R16(x) R4(R4(x))
#define R64(x) R4(R16(x))
#define R256(x) R4(R64(x))
#define R1024(x) R4(R256(x))
#define FMA asm volatile("v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %3, %3, %1, %2" : "+v"(x), "+v"(y) : "v"(a), "v"(b));  // 2 x 8 bytes

template <int KB>
__global__ __launch_bounds__(256) void victim(float* out, int iters, float a, float b) {
  float x = threadIdx.x, y = blockIdx.x;
#pragma clang loop unroll(disable)
  for (int i = 0; i < iters; ++i) {
    if (KB >= 8) { R256(FMA) R256(FMA) }      // 512 pairs = 8 KB
    if (KB >= 16) { R256(FMA) R256(FMA) }
    if (KB >= 32) { R1024(FMA) }
  }
  if (x + y == 12345.678f) out[0] = x + y;
}
#define SN asm volatile("s_nop 0\n s_nop 0\n" : "+s"(c));  // 4 + 4 bytes (nothing here may touch SCC: the loop test lives across it)
__global__ __launch_bounds__(64) void polluter(int* out, int iters, volatile int* stop) {
  int c = 0;
#pragma clang loop unroll(disable)
  for (int i = 0; i < iters; ++i) {
    R1024(SN) R1024(SN) R1024(SN) R1024(SN) R1024(SN) R1024(SN)  // 6144 x 8 B = 48 KB
  }
  if (c == 12345) out[0] = c;
}

template <int KB>
static void run(hipStream_t s1, hipStream_t s2, float* d, int* di, bool pollute) {
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  const int iters = 4096 / KB;  // same instruction count per size
  victim<KB><<<1024, 256, 0, s1>>>(d, 2, 1.0f, 0.5f);
  CK(hipStreamSynchronize(s1));
  if (pollute) polluter<<<256, 64, 0, s2>>>(di, 3000, nullptr);
  CK(hipEventRecord(a, s1));
  for (int r = 0; r < 10; ++r) victim<KB><<<1024, 256, 0, s1>>>(d, iters, 1.0f, 0.5f);
  CK(hipEventRecord(b, s1));
  CK(hipEventSynchronize(b));
  float ms; CK(hipEventElapsedTime(&ms, a, b));
  hipError_t q = pollute ? hipStreamQuery(s2) : hipSuccess;
  printf("victim %2d KB %s: %.3f ms per launch%s\n", KB, pollute ? "with polluter" : "alone        ", ms / 10, pollute ? (q == hipErrorNotReady ? " (polluter still running)" : " (POLLUTER ALREADY DONE)") : "");
  CK(hipDeviceSynchronize());
}
int main() {
  float* d; int* di; CK(hipMalloc(&d, 4096)); CK(hipMalloc(&di, 4096));
  hipStream_t s1, s2; CK(hipStreamCreate(&s1)); CK(hipStreamCreate(&s2));
  for (int p = 0; p < 2; ++p) { run<8>(s1, s2, d, di, p); run<16>(s1, s2, d, di, p); run<32>(s1, s2, d, di, p); }
  // polluter on its own for scale
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  CK(hipEventRecord(a, s2)); polluter<<<256, 64, 0, s2>>>(di, 300, nullptr); CK(hipEventRecord(b, s2)); CK(hipEventSynchronize(b));
  float ms; CK(hipEventElapsedTime(&ms, a, b)); printf("polluter 300 iters: %.3f ms\n", ms);
  return 0;
}
