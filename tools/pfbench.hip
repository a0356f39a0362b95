// pfbench — stand-alone timing / diagnosis harness for the depth plane-fit kernels (not part of the product).
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -std=c++17 [-DPFBENCH_STAMPS] -o tools/pfbench tools/pfbench.hip
//   python tools/dump_stream.py /tmp/tum64.bin 64 && tools/pfbench /tmp/tum64.bin 64 [iters] [1 = print stamps]
// Runs every kernel variant on the same frames, compares the cell records bit for bit with variant 0 (the
// product kernel) and prints per-launch time; with stamps it also prints where the waves ran and for how long.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <vector>

// -DPFBENCH_STAMPS compiles the per-wave s_memtime stamps in (they cost the kernel registers: its timings are then no
// longer those of the product kernel; use the stamps for the shape of a launch, the plain build for its duration).
__device__ unsigned long long* g_stamps = nullptr;
#ifdef PFBENCH_STAMPS
#define DPX_PF_STAMP(slot)                                                                          \
  if (g_stamps && lane == 0) {                                                                      \
    g_stamps[wave * 4 + (slot)] = __builtin_readcyclecounter();                                     \
    if ((slot) == 0)                                                                                \
      g_stamps[wave * 4 + 3] = (static_cast<unsigned long long>(__builtin_amdgcn_s_getreg((31 << 11) | 20)) << 32) | \
                               __builtin_amdgcn_s_getreg((31 << 11) | 4);                           \
  }
#endif

#include "../deplex_amd/csrc/dpx_planefit_fast.h"
#if __has_include("../deplex_amd/csrc/dpx_planefit_v2.h")
#include "../deplex_amd/csrc/dpx_planefit_v2.h"
#define HAVE_V2 1
#endif

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

static dpx::Params make_params(int H, int W, int P, int B, float fx, float fy, float cx, float cy) {
  dpx::Params p{};
  p.H = H; p.W = W; p.P = P; p.nh = W / P; p.nv = H / P; p.C = p.nh * p.nv; p.N = H * W; p.n = P * P;
  p.stride_w = p.nh * P; p.sheared = p.stride_w != W; p.batch = B;
  p.fx = fx; p.fy = fy; p.cx = cx; p.cy = cy;
  p.min_cos = 0.90f; p.max_merge_dist = 500.f; p.min_score = 0.55f; p.sigma_coeff = 1.425e-6f; p.sigma_margin = 10.f;
  p.disc_thr = 160.f; p.tol_sin = sqrtf(1.0f - 0.9f * 0.9f); p.hist_B = 20; p.min_cand = 5; p.min_cells = 4; p.max_disc = 1;
  p.valid_thr = (3ll * p.n) / 3;
  return p;
}

struct Variant { const char* name; void (*launch)(const dpx::Params&, const uint16_t*, dpx_cell*, float, float, hipStream_t); };

template <int MODE>
static void launch_v0(const dpx::Params& prm, const uint16_t* d, dpx_cell* c, float rx, float ry, hipStream_t s) {
  const long long waves = static_cast<long long>(prm.batch) * prm.nv * ((prm.nh + 63) / 64);
  const unsigned blocks = static_cast<unsigned>((waves + 3) / 4);
  dpx::PfGroup g{}; for (int j = 0; j < dpx::kMaxGroup; ++j) { g.in[j] = d; g.fx[j] = prm.fx; g.fy[j] = prm.fy; g.cx[j] = prm.cx; g.cy[j] = prm.cy; g.rx[j] = rx; g.ry[j] = ry; } g.per_sub = prm.batch;
  hipLaunchKernelGGL((dpx::k_planefit_depth<10, true, MODE>), dim3(blocks), dim3(256), 0, s, prm, g, c);
}

int main(int argc, char** argv) {
  if (argc < 3) { fprintf(stderr, "usage: pfbench depth.bin B [iters] [stamps]\n"); return 2; }
  const int B = atoi(argv[2]);
  const int iters = argc > 3 ? atoi(argv[3]) : 50;
  const bool stamps = argc > 4 && atoi(argv[4]) != 0;
  const int H = 480, W = 640, P = 10;
  const size_t N = size_t(H) * W;
  std::vector<uint16_t> h(N * B);
  FILE* f = fopen(argv[1], "rb");
  if (!f) { perror(argv[1]); return 1; }
  size_t got = fread(h.data(), 2, h.size(), f);
  fclose(f);
  if (got < N) { fprintf(stderr, "short file\n"); return 1; }
  for (size_t i = got; i < h.size(); ++i) h[i] = h[i % got];
  // TUM fr3 intrinsics (tests/golden/data/configs/TUM_fr3_long_val.K)
  const float fx = 535.4f, fy = 539.2f, cx = 320.1f, cy = 247.6f;
  dpx::Params prm = make_params(H, W, P, B, fx, fy, cx, cy);
  uint16_t* d_depth; dpx_cell *d_ref, *d_out;
  CK(hipMalloc(&d_depth, h.size() * 2));
  CK(hipMemcpy(d_depth, h.data(), h.size() * 2, hipMemcpyHostToDevice));
  const size_t cells = size_t(B) * prm.C;
  CK(hipMalloc(&d_ref, cells * sizeof(dpx_cell)));
  CK(hipMalloc(&d_out, cells * sizeof(dpx_cell)));
  const float rx = 1.0f / fx, ry = 1.0f / fy;
  hipStream_t s; CK(hipStreamCreate(&s));

  std::vector<Variant> vars = {
      {"v0 product k_planefit_depth<10,fastdiv>", launch_v0<0>},
      {"v0 DIAG1 (moments only)", launch_v0<1>},
      {"v0 DIAG2 (loads+stores only)", launch_v0<2>},
#ifdef HAVE_V2
      DPX_V2_VARIANTS
#endif
  };
  std::vector<dpx_cell> ref(cells), out(cells);
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  const double bytes = double(B) * (N * 2 + prm.C * 64.0);
  for (size_t v = 0; v < vars.size(); ++v) {
    dpx_cell* dst = v == 0 ? d_ref : d_out;
    CK(hipMemsetAsync(dst, 0xFF, cells * sizeof(dpx_cell), s));
    for (int i = 0; i < 3; ++i) vars[v].launch(prm, d_depth, dst, rx, ry, s);
    CK(hipStreamSynchronize(s));
    float best = 1e9f;
    for (int rep = 0; rep < 3; ++rep) {
      CK(hipEventRecord(a, s));
      for (int i = 0; i < iters; ++i) vars[v].launch(prm, d_depth, dst, rx, ry, s);
      CK(hipEventRecord(b, s));
      CK(hipEventSynchronize(b));
      float ms; CK(hipEventElapsedTime(&ms, a, b));
      best = std::min(best, ms / iters);
    }
    CK(hipGetLastError());
    size_t bad = 0;
    if (v == 0) CK(hipMemcpy(ref.data(), d_ref, cells * sizeof(dpx_cell), hipMemcpyDeviceToHost));
    else {
      CK(hipMemcpy(out.data(), d_out, cells * sizeof(dpx_cell), hipMemcpyDeviceToHost));
      for (size_t i = 0; i < cells; ++i) bad += memcmp(&ref[i], &out[i], sizeof(dpx_cell)) != 0;
    }
    size_t valid = 0;
    if (v == 0) for (auto& c : ref) valid += (c.info & 1u);
    printf("%-48s %7.2f us/launch  %6.0f GB/s  frac %.3f  %s", vars[v].name, best * 1e3, bytes / best / 1e6, bytes / best / 1e6 / 8000.0,
           v == 0 ? "" : (bad ? "DIFFERS" : "bit-exact"));
    if (v == 0) printf("(valid cells %.1f %%)", 100.0 * valid / cells);
    if (bad) printf(" (%zu of %zu records)", bad, cells);
    printf("\n");
    if (stamps) {
      const long long waves = static_cast<long long>(B) * prm.nv * ((prm.nh + 63) / 64);
      unsigned long long* d_st; CK(hipMalloc(&d_st, waves * 4 * 8));
      CK(hipMemset(d_st, 0, waves * 4 * 8));
      CK(hipMemcpyToSymbol(HIP_SYMBOL(g_stamps), &d_st, sizeof(d_st)));
      vars[v].launch(prm, d_depth, dst, rx, ry, s);
      CK(hipStreamSynchronize(s));
      unsigned long long* nul = nullptr;
      CK(hipMemcpyToSymbol(HIP_SYMBOL(g_stamps), &nul, sizeof(nul)));
      std::vector<unsigned long long> st(waves * 4);
      CK(hipMemcpy(st.data(), d_st, waves * 4 * 8, hipMemcpyDeviceToHost));
      CK(hipFree(d_st));
      double mom = 0, fit = 0; long long cnt = 0;
      std::map<unsigned, int> per_simd;  // (xcc, se, cu, simd) -> waves
      std::map<unsigned, std::pair<unsigned long long, unsigned long long>> span;  // per xcc: min start, max end
      std::vector<double> durs;
      for (long long w = 0; w < waves; ++w) {
        const unsigned long long t0 = st[w * 4], t1 = st[w * 4 + 1], t2 = st[w * 4 + 2], id = st[w * 4 + 3];
        if (!t0 || !t2) continue;
        mom += double(t1 - t0); fit += double(t2 - t1); ++cnt;
        durs.push_back(double(t2 - t0));
        const unsigned hw = unsigned(id), xcc = unsigned(id >> 32) & 0xF;
        const unsigned simd = (hw >> 4) & 3, cu = (hw >> 8) & 0xF, sh = (hw >> 12) & 1, se = (hw >> 13) & 7;
        per_simd[(xcc << 16) | (se << 12) | (sh << 8) | (cu << 4) | simd]++;
        auto& sp = span[xcc];
        if (!sp.first || t0 < sp.first) sp.first = t0;
        if (t2 > sp.second) sp.second = t2;
      }
      std::map<int, int> hist;
      for (auto& kv : per_simd) hist[kv.second]++;
      std::sort(durs.begin(), durs.end());
      printf("    stamps: %lld waves, moments %.0f cyc, fit+store %.0f cyc (mean per wave); wave total p50 %.0f p95 %.0f max %.0f; SIMDs used %zu; waves/SIMD histogram:",
             cnt, mom / cnt, fit / cnt, durs[durs.size() / 2], durs[durs.size() * 95 / 100], durs.back(), per_simd.size());
      for (auto& kv : hist) printf(" %dw x%d", kv.first, kv.second);
      printf("\n    per-XCC span (cycles):");
      for (auto& kv : span) printf(" x%u:%llu", kv.first, kv.second.second - kv.second.first);
      printf("\n");
    }
  }
  return 0;
}
