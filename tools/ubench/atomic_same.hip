// How fast does memory serve atomics on ONE address?  `waves` wavefronts each issue one device-scope atomicAdd (lane 0) either
// all on the same word, on the 32 words of one 128-byte line, or each on a word of its own line, with or without using the returned value; time per launch by HIP
// events, many launches.  This is what made k_core_alloc / k_core_flatten / k_pair_scatter slow before their counters were
// combined per workgroup in LDS.  build: hipcc --offload-arch=gfx950 -O3 atomic_same.hip -o atomic_same
#include <hip/hip_runtime.h>
#include <cstdio>
template <int SAME, bool RET>  // SAME: 1 one word, 2 the 32 words of one 128-byte line, 0 a word of its own line per wave
__global__ __launch_bounds__(256) void k(unsigned* ctr, unsigned* sink) {
  const unsigned wave = (blockIdx.x * 256 + threadIdx.x) >> 6;
  if ((threadIdx.x & 63) == 0) {
    unsigned* p = ctr + (SAME == 1 ? 0u : (SAME == 2 ? (wave & 31u) : wave * 64u));
    if (RET) {
      const unsigned old = atomicAdd(p, 1u);
      if (old == 0xFFFFFFFFu) sink[0] = old;  // (the value is needed: the wave waits for it)
    } else {
      atomicAdd(p, 1u);
    }
  }
}
template <int SAME, bool RET>
void run(const char* name, unsigned* ctr, unsigned* sink, int waves) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  const int blocks = waves / 4, reps = 20;
  hipLaunchKernelGGL((k<SAME, RET>), dim3(blocks), dim3(256), 0, 0, ctr, sink);
  hipEventRecord(e0);
  for (int i = 0; i < reps; ++i) hipLaunchKernelGGL((k<SAME, RET>), dim3(blocks), dim3(256), 0, 0, ctr, sink);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms = 0.f;
  (void)hipEventElapsedTime(&ms, e0, e1);
  printf("%-34s %7d waves: %8.1f us per launch, %7.1f ns per atomic\n", name, waves, ms * 1e3 / reps, ms * 1e6 / reps / waves);
}
int main() {
  unsigned *ctr, *sink;
  hipMalloc(&ctr, size_t(1) << 28);
  hipMalloc(&sink, 64);
  hipMemset(ctr, 0, size_t(1) << 28);
  for (int waves : {1024, 8192, 65536}) {
    run<1, false>("same word, result unused", ctr, sink, waves);
    run<1, true>("same word, result used", ctr, sink, waves);
    run<2, true>("32 words of one line, result used", ctr, sink, waves);
    run<0, false>("own line, result unused", ctr, sink, waves);
    run<0, true>("own line, result used", ctr, sink, waves);
  }
  return 0;
}
