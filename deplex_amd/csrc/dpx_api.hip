// dpx_api.hip — host side of the C-ABI declared in include/deplex_amd.h.
// Builds into libdeplex_amd.so with:  hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -fPIC -shared
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/deplex_amd.h"
#include "dpx_kernels.h"
#include "dpx_planefit_fast.h"
#include "dpx_regions.h"
#include "dpx_regions2.h"
#include "dpx_regions3.h"
#include "dpx_binsort.h"
#include "dpx_refine.h"
#include "dpx_png.h"

// Diagnostics that can return garbage (DPX_SKIP: stages left out for timing), run timing-only instantiations (DPX_PF_DIAG) or
// stamped / synchronising forms of the region kernels (dpx_debug_region_phases, DPX_SYNC_EACH) exist only in the -DDPX_DIAG
// build (deplex_amd/build.py build_variant("diag") -> lib/variants/diag.so, selected with DPX_LIB_PATH); the product library
// has neither the switches nor the code.
#ifdef DPX_DIAG
#define DPX_IF_DIAG(...) __VA_ARGS__
#define DPX_DIAG_GETENV(name) std::getenv(name)
#else
#define DPX_IF_DIAG(...)
#define DPX_DIAG_GETENV(name) static_cast<const char*>(nullptr)
#endif

namespace {

thread_local std::string g_err;

int fail(int code, const std::string& msg) {
  g_err = msg;
  return code;
}

#define DPX_HIP(expr)                                                                                      \
  do {                                                                                                     \
    hipError_t e_ = (expr);                                                                                \
    if (e_ != hipSuccess)                                                                                  \
      return fail(e_ == hipErrorNoDevice || e_ == hipErrorInvalidDevice ? DPX_ERR_NO_DEVICE : DPX_ERR_HIP, \
                  std::string(#expr) + ": " + hipGetErrorString(e_));                                      \
  } while (0)

constexpr size_t kLdsLimit = 160 * 1024 - 512;

}  // namespace

struct dpx_handle_s {
  dpx_config cfg;  // as stored by the reference: patch_size clamped to min(H, W)
  int H = 0, W = 0, max_batch = 0, device = 0;
  dpx::Params prm{};
  int pcap = 1;
  bool use_lds = true;
  size_t lds_bytes = 0;
  hipStream_t stream = nullptr;       // owned, used by the host-pointer entry points
  hipStream_t last_stream = nullptr;  // stream of the last launch (for result getters)
  // scratch
  dpx_cell* d_cells = nullptr;
  uint32_t* d_meta = nullptr;
  uint32_t* d_order = nullptr;
  uint32_t* d_grow_start = nullptr;
  float* d_cand = nullptr;
  dpx_plane* d_planes = nullptr;
  int32_t* d_cell_labels = nullptr;
  uint2* d_pairs = nullptr;
  int32_t* d_merge = nullptr;
  dpx_frame_stats* d_stats = nullptr;
  uint32_t* d_mt_init = nullptr;        // default-seeded mt19937 state for the RANSAC refinement
  size_t refine_lds = 0;
  int* d_refine_cells = nullptr;        // refinement of frames whose cell list does not fit LDS: [B][C]
  unsigned long long* d_dbg = nullptr;  // diagnostic phase stamps (dpx_debug_region_phases)
  unsigned long long* d_res = nullptr;  // diagnostic (DPX_RESIDENCY=1): workgroup lifetimes of the three region launches
  float4* d_lds_img = nullptr;          // split region kernel: LDS image per frame
  bool regions_split = false;
  bool stream_mode = false;   // slot of a dpx_pipeline: several batches in flight, so the LDS a frame's grow wave holds decides throughput
  bool tier0_split = false;   // tier 0 as three launches with a one-wave grow launch (mid-size frames in stream mode)
  float* d_mse_c = nullptr;
  uint32_t* d_cand_grow = nullptr;
  bool no_pixel_labels = false;  // set for the duration of a cell-level call: the label expansion launch is left out
  int regions_tier = -1;  // k_regions2<TIER>: 2 all state in LDS, 1 lean, 0 flags in LDS + global queues; -1 = k_regions (v1)
  size_t lds2_bytes = 0;                     // single-launch form: image + the larger of staging / sort keys
  size_t lds2_img = 0, lds2_p1 = 0, lds2_p2 = 0, lds2_p3 = 0;  // split form: image, and the three launches' allocations
  size_t lds2_p4 = 0;                        // fused preparation + grow launch (tier 2)
  bool opt_prep_grow_fused = false;
  int sort_pad = 0;                          // tier 2: slots of the in-LDS seed-list sort (power of two >= C)
  // variant switches, read from the environment at create (tests and diagnostics)
  bool opt_nofuse = false, opt_labels_v1 = false, opt_cloud_nolds = false, opt_prep_v1 = false;
  bool opt_sort_bitonic = true;              // tiers 1/0: the frame-wide bitonic seed-list sort of dpx_binsort.h; false (DPX_SORT_COUNTING=1) = the
                                             // core path's counting sort on cell keys — measured slower there (1280x960 x 256: 114 k against 211 k frames/s:
                                             // a clean frame has ~10 000 cells in ONE bin, one workgroup's sort), kept as a cross-check
  int opt_csr_chunk = 64;
  int Cpad = 0;                              // tiers 1/0: sort slots per frame (power of two >= C)
  unsigned long long* d_sort_keys = nullptr;  // [B][Cpad]
  uint32_t* d_bin_off = nullptr;              // [B][nb + 1]
  uint32_t* d_gqpos = nullptr;                // tier 0: [B][C]
  char* h_cells_stage = nullptr;              // cell-level output: page-locked staging the device writes into
  long long cells_stage_total = -1;           // plane records the staging holds from the last cell-level call (-1: none)
  size_t cells_stage_planes_off = 0;
  uint8_t* d_gflags = nullptr;
  uint8_t* d_gowner = nullptr;
  uint32_t* d_gqueue = nullptr;
  // region growing through cores (dpx_regions3.h): large frames
  bool cores = false;
  int core_T = 0, core_logT = 0;              // slots of the pair hash table per frame
  size_t lds3_sim = 0;                        // LDS of the one-wave sequence launch
  size_t lds3_order = 0;                      // dynamic LDS of the order launch (a huge grow's cells, ring, batch map)
  uint32_t* d_core = nullptr;                 // one allocation for the uint32 arrays below
  uint32_t *d_parent = nullptr, *d_csize = nullptr, *d_npairs = nullptr, *d_pstart = nullptr, *d_pfill = nullptr, *d_noe = nullptr,
           *d_oestart = nullptr, *d_oefill = nullptr, *d_hkey = nullptr, *d_hcnt = nullptr, *d_corepairs = nullptr,
           *d_histg = nullptr, *d_pbcnt = nullptr, *d_hslots = nullptr, *d_cgrow = nullptr, *d_cellq = nullptr, *d_worklist = nullptr, *d_biglist = nullptr, *d_ccursor = nullptr;
  int* d_order_task = nullptr;
  unsigned long long* d_hbest = nullptr;      // [B][T]
  uint4* d_oedges = nullptr;                  // [B][4C] one-way edges with their target core's data
  uint2* d_crec = nullptr;                    // [B][C] by core ID: first one-way edge, number of edges
  float* d_candfit = nullptr;                 // [B][pcap][12] fits of the candidate segments (k_seg_fit)
  uint32_t* d_pbcnt_own = nullptr;            // cell-level tiers 1/0: [B][bins] counters of the counting sort
  uint4* d_lists = nullptr;                   // [B][C][2] seed lists, one 32-byte entry per (core, bin) pair
  // staging for the host-pointer API (lazy)
  void* d_in = nullptr;
  size_t d_in_bytes = 0;
  int32_t* d_labels = nullptr;
  size_t d_labels_bytes = 0;
  // exact-division-by-reciprocal verification cache (dpx_planefit_fast.h): the last few intrinsics tuples seen
  struct FdEntry {
    float fx, fy, cx, cy, rx, ry;
    bool ok;
  };
  unsigned int* d_mismatch = nullptr;
  hipStream_t verify_stream = nullptr;  // the exact-division check runs here, beside whatever the handle has in flight
  // profiling: a ring of event sets, one per process call
  bool profiling = false;
  static constexpr int kProfSlots = 128;
  hipEvent_t ev[kProfSlots][7] = {};  // fit start/end, graph start, graph end = regions start, regions end, labels start/end
  bool ev_created = false;
  long long prof_calls = 0;
  int last_batch = 0;
  long long fit_calls = 0;
};

namespace {

template <typename T>
int dev_alloc(T** p, size_t count) {
  DPX_HIP(hipMalloc(reinterpret_cast<void**>(p), std::max<size_t>(count, 1) * sizeof(T)));
  return DPX_OK;
}

void free_all(dpx_handle h) {
  if (!h) return;
  (void)hipSetDevice(h->device);
  hipFree(h->d_cells);
  hipFree(h->d_meta);
  hipFree(h->d_order);
  hipFree(h->d_grow_start);
  if (h->h_cells_stage) hipHostFree(h->h_cells_stage);
  hipFree(h->d_cand);
  hipFree(h->d_planes);
  hipFree(h->d_cell_labels);
  hipFree(h->d_pairs);
  hipFree(h->d_merge);
  hipFree(h->d_stats);
  hipFree(h->d_dbg);
  hipFree(h->d_res);
  hipFree(h->d_mt_init);
  hipFree(h->d_lds_img);
  hipFree(h->d_mse_c);
  hipFree(h->d_cand_grow);
  hipFree(h->d_gflags);
  hipFree(h->d_gowner);
  hipFree(h->d_gqueue);
  hipFree(h->d_core);
  hipFree(h->d_hbest);
  hipFree(h->d_oedges);
  hipFree(h->d_crec);
  hipFree(h->d_candfit);
  hipFree(h->d_pbcnt_own);
  hipFree(h->d_lists);
  hipFree(h->d_sort_keys);
  hipFree(h->d_bin_off);
  hipFree(h->d_gqpos);
  hipFree(h->d_refine_cells);
  hipFree(h->d_in);
  hipFree(h->d_labels);
  hipFree(h->d_mismatch);
  for (auto& slot : h->ev)
    for (auto& e : slot)
      if (e) hipEventDestroy(e);
  if (h->verify_stream) hipStreamDestroy(h->verify_stream);
  if (h->stream) hipStreamDestroy(h->stream);
}

int device_cus() {  // compute units of the CURRENT device (persistent launches size their grids with it), cached per device id
  static std::mutex mu;
  static std::vector<int> cus;
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0) return 256;
  std::lock_guard<std::mutex> lk(mu);
  if (static_cast<size_t>(dev) >= cus.size()) cus.resize(static_cast<size_t>(dev) + 1, 0);
  if (cus[dev] <= 0) {
    int n = 0;
    if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256;
    cus[dev] = n;
  }
  return cus[dev];
}

bool fast_shape(const dpx::Params& prm) {
  return !prm.sheared && (prm.P == 4 || prm.P == 8 || prm.P == 10) && (prm.W % 2 == 0) && prm.C > 0;
}

// Verifies (once per intrinsics tuple and image shape) that the reciprocal-based division of the fast plane-fit kernel
// returns the IEEE quotient for every numerator this image geometry can produce.  The verdicts live in ONE process-wide
// cache (device, W, H and the four floats compared bit for bit, so a NaN tuple matches itself): the slots of a pipeline
// and every extractor of the same shape share them.  A tuple seen for the first time is checked on a stream of its own
// and only that stream is waited for — a pipeline_submit() with new intrinsics does not wait for the slot's group in
// flight (it does block the host for the ~0.2 ms the two check kernels take).
struct FdKey {
  int device, W, H;
  uint32_t fx, fy, cx, cy;
  bool operator==(const FdKey& o) const { return device == o.device && W == o.W && H == o.H && fx == o.fx && fy == o.fy && cx == o.cx && cy == o.cy; }
};
struct FdGlobal {
  FdKey key;
  dpx_handle_s::FdEntry e;
};
constexpr size_t kFdCache = 64;
std::mutex g_fd_mu;
std::vector<FdGlobal> g_fd_cache;  // most recently used first

uint32_t f32_bits(float v) {
  uint32_t u;
  std::memcpy(&u, &v, 4);
  return u;
}

int prepare_intrinsics(dpx_handle h, float fx, float fy, float cx, float cy, dpx_handle_s::FdEntry* out) {
  const bool fast = fast_shape(h->prm) && !std::getenv("DPX_NO_FASTDIV");
  const FdKey key{fast ? h->device : -1, fast ? h->W : 0, fast ? h->H : 0, f32_bits(fx), f32_bits(fy), f32_bits(cx), f32_bits(cy)};
  {
    std::lock_guard<std::mutex> lk(g_fd_mu);
    for (size_t i = 0; i < g_fd_cache.size(); ++i)
      if (g_fd_cache[i].key == key) {
        if (i != 0) std::rotate(g_fd_cache.begin(), g_fd_cache.begin() + i, g_fd_cache.begin() + i + 1);
        if (out) *out = g_fd_cache[0].e;
        return DPX_OK;
      }
  }
  dpx_handle_s::FdEntry e{fx, fy, cx, cy, 1.0f / fx, 1.0f / fy, false};
  if (fast && std::isfinite(e.rx) && std::isfinite(e.ry)) {
    if (!h->verify_stream) DPX_HIP(hipStreamCreateWithFlags(&h->verify_stream, hipStreamNonBlocking));
    if (!h->d_mismatch) DPX_HIP(hipMalloc(reinterpret_cast<void**>(&h->d_mismatch), sizeof(unsigned int)));
    hipStream_t vs = h->verify_stream;
    DPX_HIP(hipMemsetAsync(h->d_mismatch, 0, sizeof(unsigned int), vs));
    const long long tx = static_cast<long long>(h->W) * 65536, ty = static_cast<long long>(h->H) * 65536;
    hipLaunchKernelGGL(dpx::k_verify_fastdiv, dim3(static_cast<unsigned>((tx + 255) / 256)), dim3(256), 0, vs, h->W, cx, fx, e.rx, h->d_mismatch);
    hipLaunchKernelGGL(dpx::k_verify_fastdiv, dim3(static_cast<unsigned>((ty + 255) / 256)), dim3(256), 0, vs, h->H, cy, fy, e.ry, h->d_mismatch);
    unsigned int mism = 1;
    DPX_HIP(hipMemcpyAsync(&mism, h->d_mismatch, sizeof(unsigned int), hipMemcpyDeviceToHost, vs));
    DPX_HIP(hipStreamSynchronize(vs));
    e.ok = (mism == 0);
  }
  {
    std::lock_guard<std::mutex> lk(g_fd_mu);
    if (g_fd_cache.size() >= kFdCache) g_fd_cache.pop_back();
    g_fd_cache.insert(g_fd_cache.begin(), FdGlobal{key, e});
  }
  if (out) *out = e;
  return DPX_OK;
}

constexpr int kPfBlockWaves = 4;
template <int P>
void launch_planefit_fast(const dpx::Params& prm, const dpx::PfGroup& grp, dpx_cell* d_cells, hipStream_t s, bool fastdiv) {
  const long long waves = static_cast<long long>(prm.batch) * prm.nv * ((prm.nh + 63) / 64);
  // waves per workgroup (a wave is a task on its own; the kernel takes any whole number of waves up to 8).  4 = one wave
  // per SIMD is the balance point in the pipeline: single-wave workgroups fill every gap and crowd the region launches
  // out (plane fit -20 %, regions +15 %, total -4 %), 8-wave ones are crowded out themselves (-5 %).
  static const int wpb = std::getenv("DPX_PF_BLOCK_WAVES") ? std::max(1, std::min(8, std::atoi(std::getenv("DPX_PF_BLOCK_WAVES")))) : kPfBlockWaves;
  const unsigned blocks = static_cast<unsigned>((waves + wpb - 1) / wpb);
  const dim3 bd(64u * wpb);
#ifdef DPX_DIAG
  static const int diag = std::getenv("DPX_PF_DIAG") ? std::atoi(std::getenv("DPX_PF_DIAG")) : 0;  // timing-only instantiations
  if (P == 10 && diag == 1) {
    hipLaunchKernelGGL((dpx::k_planefit_depth<10, true, 1>), dim3(blocks), bd, 0, s, prm, grp, d_cells);
    return;
  }
  if (P == 10 && diag == 2) {
    hipLaunchKernelGGL((dpx::k_planefit_depth<10, true, 2>), dim3(blocks), bd, 0, s, prm, grp, d_cells);
    return;
  }
#endif
  if (false) {
  } else if (fastdiv)
    hipLaunchKernelGGL((dpx::k_planefit_depth<P, true>), dim3(blocks), bd, 0, s, prm, grp, d_cells);
  else
    hipLaunchKernelGGL((dpx::k_planefit_depth<P, false>), dim3(blocks), bd, 0, s, prm, grp, d_cells);
}

template <bool DEPTH>
void launch_planefit(const dpx::Params& prm, const void* d_in, dpx_cell* d_cells, hipStream_t s) {
  const long long total = static_cast<long long>(prm.batch) * prm.C;
  if (total == 0) return;
  const unsigned blocks = static_cast<unsigned>((total + 255) / 256);
  switch (prm.sheared ? 0 : prm.P) {
    case 4: hipLaunchKernelGGL((dpx::k_planefit<4, DEPTH>), dim3(blocks), dim3(256), 0, s, prm, d_in, d_cells); break;
    case 8: hipLaunchKernelGGL((dpx::k_planefit<8, DEPTH>), dim3(blocks), dim3(256), 0, s, prm, d_in, d_cells); break;
    case 10: hipLaunchKernelGGL((dpx::k_planefit<10, DEPTH>), dim3(blocks), dim3(256), 0, s, prm, d_in, d_cells); break;
    case 16: hipLaunchKernelGGL((dpx::k_planefit<16, DEPTH>), dim3(blocks), dim3(256), 0, s, prm, d_in, d_cells); break;
    default: hipLaunchKernelGGL((dpx::k_planefit<0, DEPTH>), dim3(blocks), dim3(256), 0, s, prm, d_in, d_cells); break;
  }
}

// LDS-staged cloud kernel: needs 16-byte aligned row segments (W % 4 == 0 and aligned clouds); takes a whole group
bool cloud_lds_path(dpx_handle h, const dpx::Params& prm) {
  const char* v1 = std::getenv("DPX_PLANEFIT_V1");
  return fast_shape(prm) && !(v1 && v1[0] == '1') && !h->opt_cloud_nolds && (prm.W & 3) == 0;
}

void launch_planefit_cloud_group(const dpx::Params& prm, const dpx::PfGroup& grp, dpx_cell* d_cells, hipStream_t s) {
  const long long waves = static_cast<long long>(prm.batch) * prm.nv * ((prm.nh + 63) / 64);
  const unsigned blocks = static_cast<unsigned>((waves + 3) / 4);
  switch (prm.P) {
    case 4: hipLaunchKernelGGL((dpx::k_planefit_cloud_lds<4>), dim3(blocks), dim3(256), 0, s, prm, grp, d_cells); return;
    case 8: hipLaunchKernelGGL((dpx::k_planefit_cloud_lds<8>), dim3(blocks), dim3(256), 0, s, prm, grp, d_cells); return;
    default: hipLaunchKernelGGL((dpx::k_planefit_cloud_lds<10>), dim3(blocks), dim3(256), 0, s, prm, grp, d_cells); return;
  }
}

void launch_planefit_cloud(dpx_handle h, const dpx::Params& prm, const void* d_in, dpx_cell* d_cells, hipStream_t s) {
  const char* v1 = std::getenv("DPX_PLANEFIT_V1");
  if (fast_shape(prm) && !(v1 && v1[0] == '1')) {
    const long long waves = static_cast<long long>(prm.batch) * prm.nv * ((prm.nh + 63) / 64);
    const unsigned blocks = static_cast<unsigned>((waves + 3) / 4);
    const float* xyz = static_cast<const float*>(d_in);
    if (cloud_lds_path(h, prm) && (reinterpret_cast<uintptr_t>(d_in) & 15u) == 0) {
      dpx::PfGroup grp{};
      for (int q = 0; q < dpx::kMaxGroup; ++q) grp.in[q] = d_in;
      grp.per_sub = prm.batch;
      launch_planefit_cloud_group(prm, grp, d_cells, s);
      return;
    }
    switch (prm.P) {
      case 4: hipLaunchKernelGGL((dpx::k_planefit_cloud<4>), dim3(blocks), dim3(256), 0, s, prm, xyz, d_cells); return;
      case 8: hipLaunchKernelGGL((dpx::k_planefit_cloud<8>), dim3(blocks), dim3(256), 0, s, prm, xyz, d_cells); return;
      case 10: hipLaunchKernelGGL((dpx::k_planefit_cloud<10>), dim3(blocks), dim3(256), 0, s, prm, xyz, d_cells); return;
      default: break;
    }
  }
  launch_planefit<false>(prm, d_in, d_cells, s);
}

// true: this frame geometry runs the tuned depth kernel (k_planefit_depth), which takes a whole group per launch
bool depth_fast_path(const dpx::Params& prm) {
  const char* v1 = std::getenv("DPX_PLANEFIT_V1");
  return fast_shape(prm) && !(v1 && v1[0] == '1');
}

// prm.batch frames = the group's submissions back to back; every submission verified for the same division flavour
void launch_planefit_depth_group(const dpx::Params& prm, const dpx::PfGroup& grp, dpx_cell* d_cells, hipStream_t s, bool fastdiv) {
  switch (prm.P) {
    case 4: launch_planefit_fast<4>(prm, grp, d_cells, s, fastdiv); return;
    case 8: launch_planefit_fast<8>(prm, grp, d_cells, s, fastdiv); return;
    default: launch_planefit_fast<10>(prm, grp, d_cells, s, fastdiv); return;
  }
}

// A group of `nsub` sub-batches of `per_sub` frames each (one submission of a dpx_pipeline group, or the whole batch of
// a single call) goes through two host-side stages on one stream:
//   stage_fit  : plane fit of all sub-batches into the handle's cell records (frame-major, sub-batch after sub-batch)
//                — one launch for the whole group on the tuned depth path
//   stage_rest : region growing (+ cell graph and seed-list sort kernels for tiers 1 / 0), label expansion and the
//                optional refinement — one region launch sequence and one label launch for all sub-batches
// (a launch per 64 frames leaves the chip under-filled at its start and end and, for the latency-bound region kernels,
// costs more than it needs to).
struct SubBatch {
  const void* d_in;
  int32_t* d_labels;
  float fx, fy, cx, cy;  // intrinsics of this sub-batch (plane fit; the refinement re-projects its depth with them)
  float rx, ry;          // verified reciprocals of fx, fy
  bool fd_ok;            // the reciprocal division is exact for this geometry (else IEEE division)
};

dpx::CoreBufs core_bufs(dpx_handle h) {
  dpx::CoreBufs cb{};
  cb.meta = h->d_meta; cb.mse_c = h->d_mse_c; cb.parent = h->d_parent; cb.csize = h->d_csize; cb.npairs = h->d_npairs; cb.pstart = h->d_pstart;
  cb.pfill = h->d_pfill; cb.noe = h->d_noe; cb.oestart = h->d_oestart; cb.oefill = h->d_oefill; cb.hkey = h->d_hkey; cb.hcnt = h->d_hcnt;
  cb.hbest = h->d_hbest; cb.hslots = h->d_hslots; cb.corepairs = h->d_corepairs; cb.oedges = h->d_oedges; cb.crec = h->d_crec; cb.keys = h->d_sort_keys; cb.histg = h->d_histg;
  cb.pbcnt = h->d_pbcnt; cb.bin_off = h->d_bin_off; cb.lists = h->d_lists;
  cb.cgrow = h->d_cgrow; cb.cursor = h->d_ccursor; cb.cellq = h->d_cellq; cb.order_task = h->d_order_task;
  cb.T = h->core_T; cb.logT = h->core_logT; cb.Cpad = h->Cpad;
  return cb;
}

// seed lists of the sorted region tiers (dpx_binsort.h): keys -> bitonic network -> per-bin cell lists
void launch_binsort(dpx_handle h, const dpx::Params& prm, hipStream_t s) {
  if (!h->opt_sort_bitonic) {
    // counting sort over the bins, then every bin's segment sorted by (mse, cell) in LDS (dpx_regions3.h)
    const int nbins = prm.hist_B * prm.hist_B, B = prm.batch;
    const dim3 cell_grid(static_cast<unsigned>((prm.C + dpx::kPrepT - 1) / dpx::kPrepT), static_cast<unsigned>(B));
    dpx::CoreBufs cb{};
    cb.keys = h->d_sort_keys; cb.bin_off = h->d_bin_off; cb.pbcnt = h->d_pbcnt; cb.Cpad = h->Cpad;
    (void)hipMemsetAsync(h->d_pbcnt, 0, static_cast<size_t>(B) * nbins * 4, s);
    hipLaunchKernelGGL(dpx::k_cell_bincount, cell_grid, dim3(dpx::kPrepT), static_cast<size_t>(nbins) * 4, s, prm, h->d_meta, h->d_pbcnt);
    hipLaunchKernelGGL(dpx::k_bin_scan, dim3(static_cast<unsigned>(B)), dim3(64), 0, s, prm, cb);
    hipLaunchKernelGGL(dpx::k_cell_scatter, cell_grid, dim3(dpx::kPrepT), static_cast<size_t>(nbins) * 4, s, prm, h->d_meta, h->d_mse_c, h->d_pbcnt, h->d_bin_off, h->d_sort_keys, h->Cpad);
    // the segments of up to kCellSortSmall keys: many small workgroups (8 KB of LDS each, waves for the smallest); the few
    // beyond — a clean frame's one big bin — one 1024-thread workgroup per frame with LDS for the whole frame's cells
    const int sort_wgs = std::max(4, std::min(std::min(64, nbins), 4096 / std::max(1, B)));
    hipLaunchKernelGGL((dpx::k_bin_sort<true>), dim3(static_cast<unsigned>(std::min(sort_wgs, nbins)), static_cast<unsigned>(B)), dim3(256), static_cast<size_t>(dpx::kCellSortSmall) * 8, s, prm, cb, h->d_order, dpx::kCellSortSmall, 0);
    const int cap = std::max(64, std::min(prm.C, dpx::kCellSortMax));
    hipLaunchKernelGGL((dpx::k_bin_sort<true>), dim3(1, static_cast<unsigned>(B)), dim3(dpx::kCellSortT), static_cast<size_t>(cap) * 8, s, prm, cb, h->d_order, cap, dpx::kCellSortSmall);
    return;
  }
  const int Cpad = h->Cpad, B = prm.batch;
  const long long slots = static_cast<long long>(B) * Cpad;
  const unsigned g256 = static_cast<unsigned>((slots + 255) / 256);
  hipLaunchKernelGGL(dpx::k_sort_build, dim3(g256), dim3(256), 0, s, prm, h->d_meta, h->d_mse_c, h->d_sort_keys, Cpad);
  const int tile = Cpad < dpx::kSortTile ? Cpad : dpx::kSortTile;
  const unsigned tiles = static_cast<unsigned>(slots / tile);
  const size_t lds = static_cast<size_t>(tile) * 8;
  hipLaunchKernelGGL(dpx::k_sort_tile, dim3(tiles), dim3(dpx::kSortThreads), lds, s, h->d_sort_keys, Cpad, tile, 2);
  for (int k = tile * 2; k <= Cpad; k <<= 1) {
    for (int j = k >> 1; j >= tile; j >>= 1)
      hipLaunchKernelGGL(dpx::k_sort_global, dim3(static_cast<unsigned>((slots / 2 + 255) / 256)), dim3(256), 0, s, h->d_sort_keys, Cpad, B, j, k);
    hipLaunchKernelGGL(dpx::k_sort_tile, dim3(tiles), dim3(dpx::kSortThreads), lds, s, h->d_sort_keys, Cpad, tile, k);
  }
  hipLaunchKernelGGL(dpx::k_sort_lists, dim3(g256), dim3(256), 0, s, prm, h->d_sort_keys, Cpad, h->d_order, h->d_bin_off);
}

int fill_sub(dpx_handle h, SubBatch& sb, const void* d_in, int32_t* d_labels, bool depth, float fx, float fy, float cx, float cy) {
  sb = SubBatch{d_in, d_labels, fx, fy, cx, cy, 1.f, 1.f, false};
  if (depth) {
    DPX_HIP(hipSetDevice(h->device));
    dpx_handle_s::FdEntry e{};
    const int rc = prepare_intrinsics(h, fx, fy, cx, cy, &e);
    if (rc) return rc;
    sb.rx = e.rx; sb.ry = e.ry; sb.fd_ok = e.ok;
  }
  return DPX_OK;
}

int stage_fit(dpx_handle h, const SubBatch* subs, int nsub, int per_sub, bool depth, hipStream_t s, bool allow_skip = true) {
  const int batch = nsub * per_sub;
  if (nsub < 1 || nsub > dpx::kMaxGroup || per_sub < 0 || batch > h->max_batch)
    return fail(DPX_ERR_ARG, "batch " + std::to_string(batch) + " exceeds max_batch " + std::to_string(h->max_batch));
  DPX_HIP(hipSetDevice(h->device));
  if (batch == 0) return DPX_OK;
  const bool prof = h->profiling;
  hipEvent_t* ev = h->ev[h->prof_calls % dpx_handle_s::kProfSlots];
  // DPX_SKIP=<letters f r l>: timing-only diagnostics that leave out a stage (results are garbage); tools/stage_cost.sh
  static const char* skip = DPX_DIAG_GETENV("DPX_SKIP");
  // 'F': the plane fit runs only the first two times on a handle, so that the later (timed) calls of the other stages
  // still work on real cell records
  const bool skip_f = allow_skip && skip && (std::strchr(skip, 'f') || (std::strchr(skip, 'F') && h->fit_calls >= 2));
  ++h->fit_calls;
  if (prof) DPX_HIP(hipEventRecord(ev[0], s));
  if (h->prm.C > 0 && !skip_f) {
    bool one_launch = depth && depth_fast_path(h->prm);
    for (int j = 1; j < nsub && one_launch; ++j) one_launch = subs[j].fd_ok == subs[0].fd_ok;
    bool cloud_group = !depth && cloud_lds_path(h, h->prm);
    for (int j = 0; j < nsub && cloud_group; ++j) cloud_group = (reinterpret_cast<uintptr_t>(subs[j].d_in) & 15u) == 0;
    if (cloud_group) {
      dpx::Params prm = h->prm;
      prm.batch = batch;
      dpx::PfGroup grp{};
      for (int j = 0; j < dpx::kMaxGroup; ++j) grp.in[j] = subs[j < nsub ? j : 0].d_in;
      grp.per_sub = per_sub;
      launch_planefit_cloud_group(prm, grp, h->d_cells, s);
    } else if (one_launch) {
      dpx::Params prm = h->prm;
      prm.batch = batch;
      dpx::PfGroup grp{};
      for (int j = 0; j < dpx::kMaxGroup; ++j) {
        const SubBatch& b = subs[j < nsub ? j : 0];
        grp.in[j] = b.d_in;
        grp.fx[j] = b.fx; grp.fy[j] = b.fy; grp.cx[j] = b.cx; grp.cy[j] = b.cy; grp.rx[j] = b.rx; grp.ry[j] = b.ry;
      }
      grp.per_sub = per_sub;
      launch_planefit_depth_group(prm, grp, h->d_cells, s, subs[0].fd_ok);
    } else {
      for (int j = 0; j < nsub; ++j) {
        dpx::Params prm = h->prm;
        prm.batch = per_sub;
        prm.fx = subs[j].fx; prm.fy = subs[j].fy; prm.cx = subs[j].cx; prm.cy = subs[j].cy;
        dpx_cell* out = h->d_cells + static_cast<size_t>(j) * per_sub * prm.C;
        if (depth && depth_fast_path(prm)) {
          dpx::PfGroup grp{};
          for (int q = 0; q < dpx::kMaxGroup; ++q) {
            grp.in[q] = subs[j].d_in;
            grp.fx[q] = subs[j].fx; grp.fy[q] = subs[j].fy; grp.cx[q] = subs[j].cx; grp.cy[q] = subs[j].cy;
            grp.rx[q] = subs[j].rx; grp.ry[q] = subs[j].ry;
          }
          grp.per_sub = per_sub;
          launch_planefit_depth_group(prm, grp, out, s, subs[j].fd_ok);
        } else if (depth) {
          launch_planefit<true>(prm, subs[j].d_in, out, s);
        } else {
          launch_planefit_cloud(h, prm, subs[j].d_in, out, s);
        }
      }
    }
  }
  if (prof) DPX_HIP(hipEventRecord(ev[1], s));
  return DPX_OK;
}

int stage_rest(dpx_handle h, const SubBatch* subs, int nsub, int per_sub, bool depth, hipStream_t s) {
  const int batch = nsub * per_sub;
  if (nsub < 1 || nsub > dpx::kMaxGroup || batch > h->max_batch) return fail(DPX_ERR_ARG, "bad sub-batch grouping");
  DPX_HIP(hipSetDevice(h->device));
  h->last_stream = s;
  h->last_batch = batch;
  if (batch == 0) return DPX_OK;
  dpx::Params prm = h->prm;
  prm.batch = batch;
  const bool prof = h->profiling;
  hipEvent_t* ev = h->ev[h->prof_calls % dpx_handle_s::kProfSlots];
  static const char* skip = DPX_DIAG_GETENV("DPX_SKIP");
  const bool skip_r = skip && std::strchr(skip, 'r'), skip_l = skip && std::strchr(skip, 'l');
  // ---- cell graph (only when the region kernel does not compute it itself) + region growing
  if (prof) DPX_HIP(hipEventRecord(ev[2], s));
  const long long total = static_cast<long long>(batch) * prm.C;
  if (total > 0 && h->regions_tier != 2)
    hipLaunchKernelGGL(dpx::k_cellgraph, dim3(static_cast<unsigned>((total + 2047) / 2048 * 8)), dim3(256), 0, s, prm, h->d_cells, h->d_meta, h->d_mse_c);  // (a multiple of 8 workgroups: one contiguous range of cells per XCD)
  const dim3 cell_grid(static_cast<unsigned>((prm.C + 255) / 256), static_cast<unsigned>(batch));
  if (total > 0 && h->cores && !skip_r) {  // cores of the symmetric edges and their (core, bin) pairs (dpx_regions3.h), before the seed lists
    const dpx::CoreBufs cb = core_bufs(h);
    const int nbins = prm.hist_B * prm.hist_B;
    const int sort_wgs = std::max(4, std::min(std::min(64, nbins), 2048 / std::max(1, batch)));
    const dim3 tile_grid(static_cast<unsigned>(((prm.nh + dpx::kTile - 1) / dpx::kTile) * ((prm.nv + dpx::kTile - 1) / dpx::kTile)), static_cast<unsigned>(batch));
    hipLaunchKernelGGL(dpx::k_core_tiles, tile_grid, dim3(dpx::kTile * dpx::kTile), 0, s, prm, cb);
    hipLaunchKernelGGL(dpx::k_core_link, tile_grid, dim3(64), 0, s, prm, cb);
    const dim3 prep_grid(static_cast<unsigned>((prm.C + dpx::kPrepT - 1) / dpx::kPrepT), static_cast<unsigned>(batch));
    hipLaunchKernelGGL(dpx::k_core_flatten, prep_grid, dim3(dpx::kPrepT), 0, s, prm, cb);
    hipLaunchKernelGGL(dpx::k_pair_count, prep_grid, dim3(dpx::kPrepT), static_cast<size_t>(nbins) * 4, s, prm, cb);
    hipLaunchKernelGGL(dpx::k_core_alloc, prep_grid, dim3(dpx::kPrepT), 0, s, prm, cb);
    hipLaunchKernelGGL(dpx::k_bin_scan, dim3(static_cast<unsigned>(batch)), dim3(64), 0, s, prm, cb);
    hipLaunchKernelGGL(dpx::k_pair_scatter, prep_grid, dim3(dpx::kPrepT), static_cast<size_t>(nbins) * 8, s, prm, cb);
    hipLaunchKernelGGL(dpx::k_edge_scatter, cell_grid, dim3(256), 0, s, prm, cb);
    hipLaunchKernelGGL((dpx::k_bin_sort<false>), dim3(static_cast<unsigned>(std::min(sort_wgs, nbins)), static_cast<unsigned>(batch)), dim3(dpx::kBinSortT), static_cast<size_t>(dpx::kBinSortMax) * 8, s, prm, cb, static_cast<uint32_t*>(nullptr), dpx::kBinSortMax, 0);
  }
  if (total > 0 && (h->regions_tier == 1 || h->regions_tier == 0) && !h->cores && !skip_r) launch_binsort(h, prm, s);
  if (prof) DPX_HIP(hipEventRecord(ev[3], s));
  dpx::RegionBufs rb{};
  rb.cells = h->d_cells; rb.meta = h->d_meta; rb.order = h->d_order; rb.grow_start = h->d_grow_start;
  rb.cand = h->d_cand; rb.planes = h->d_planes; rb.cell_labels = h->d_cell_labels; rb.pairs = h->d_pairs;
  rb.merge = h->d_merge; rb.stats = h->d_stats; rb.g_flags = h->d_gflags; rb.g_owner = h->d_gowner;
  rb.g_queue = h->d_gqueue; rb.pcap = h->pcap;
  // the region kernel's tail can write the pixel label map itself (no separate label-expansion launch)
  const bool fuse_labels = (h->regions_tier == 2 || h->regions_tier == 1) && (prm.W & 3) == 0 && !h->opt_nofuse && !skip_l && !skip_r && !h->no_pixel_labels;
  dpx::Regions2Extra rx{};
  rx.mse_c = h->d_mse_c; rx.cand_grow = h->d_cand_grow; rx.dbg = h->d_dbg; rx.lds_img = h->d_lds_img;
  rx.res = h->d_res;
  rx.img_bytes = static_cast<int>(h->lds2_img);
  rx.sort_pad = h->sort_pad;
  rx.csr_chunk = h->opt_csr_chunk;
  rx.bin_off = h->d_bin_off;
  rx.g_qpos = h->d_gqpos;
  rx.frames_per_out = per_sub;
  for (int j = 0; j < dpx::kMaxGroup; ++j) rx.labels_out[j] = (fuse_labels && j < nsub) ? subs[j].d_labels : nullptr;
  // diagnostic (DPX_SYNC_EACH=1): wait for every launch of the core path and name it on stderr — finds a launch that never ends
  static const bool sync_each = DPX_DIAG_GETENV("DPX_SYNC_EACH") != nullptr;
  auto CHK = [&](const char* name) {
    if (sync_each) {
      const hipError_t e1 = hipStreamSynchronize(s);
      std::fprintf(stderr, "[dpx] %s: %s\n", name, hipGetErrorString(e1));
      std::fflush(stderr);
    }
  };
  if (skip_r) {
  } else if (h->cores) {
    if (total > 0) {
      dpx::SimBufs sb{};
      sb.lists = h->d_lists; sb.bin_off = h->d_bin_off; sb.histg = h->d_histg; sb.corepairs = h->d_corepairs; sb.oedges = h->d_oedges; sb.crec = h->d_crec;
      sb.cursor = h->d_ccursor; sb.cgrow = h->d_cgrow; sb.worklist = h->d_worklist;
      sb.cand_grow = reinterpret_cast<uint4*>(h->d_cand_grow); sb.biglist = h->d_biglist;
      sb.hdr_out = reinterpret_cast<int*>(h->d_lds_img); sb.pcap = h->pcap;
      sb.dbg = h->d_dbg ? h->d_dbg + static_cast<size_t>(h->max_batch) * dpx::kDbgSlots : nullptr;
      dpx::OrderBufs ob{};
      ob.cellq = h->d_cellq; ob.cand_grow = reinterpret_cast<const uint4*>(h->d_cand_grow); ob.biglist = h->d_biglist;
      ob.hdr = reinterpret_cast<const int*>(h->d_lds_img); ob.queue = h->d_gqueue; ob.task = h->d_order_task; ob.pcap = h->pcap;
      ob.dbg = sb.dbg;
      const int per_frame = std::max(1, std::min(16, 512 / std::max(1, batch)));
      DPX_IF_DIAG(if (h->d_dbg) (void)hipMemsetAsync(h->d_dbg + (static_cast<size_t>(h->max_batch)) * dpx::kDbgSlots, 0, static_cast<size_t>(batch) * dpx::kDbgSlots * 8, s);
                  if (h->d_dbg) hipLaunchKernelGGL(dpx::k_regsim<true>, dim3(batch), dim3(128), h->lds3_sim, s, prm, sb); else)
      hipLaunchKernelGGL(dpx::k_regsim<false>, dim3(batch), dim3(128), h->lds3_sim, s, prm, sb);
      CHK("k_regsim");
      hipLaunchKernelGGL(dpx::k_cellgrow, cell_grid, dim3(256), 0, s, prm, core_bufs(h));
      CHK("k_cellgrow");
      const int per_frame_huge = std::max(1, std::min(4, 128 / std::max(1, batch)));
      DPX_IF_DIAG(if (h->d_dbg) {
        hipLaunchKernelGGL((dpx::k_order<true, true>), dim3(per_frame_huge, batch), dim3(dpx::kHugeWaves * 64), h->lds3_order, s, prm, ob);
        hipLaunchKernelGGL((dpx::k_order<true, false>), dim3(per_frame, batch), dim3(dpx::kOrderWaves * 64), 0, s, prm, ob);
      } else)
      {
        hipLaunchKernelGGL((dpx::k_order<false, true>), dim3(per_frame_huge, batch), dim3(dpx::kHugeWaves * 64), h->lds3_order, s, prm, ob);
        hipLaunchKernelGGL((dpx::k_order<false, false>), dim3(per_frame, batch), dim3(dpx::kOrderWaves * 64), 0, s, prm, ob);
      }
      CHK("k_order");
      // segment sums and fits chip-wide, then the per-frame remainder (ids, labels_map_, adjacency, merge walk, labels)
      dpx::TailBufs tb{};
      tb.cells = h->d_cells; tb.queue = h->d_gqueue; tb.cand_grow = reinterpret_cast<const uint4*>(h->d_cand_grow); tb.biglist = h->d_biglist;
      tb.hdr = reinterpret_cast<const int*>(h->d_lds_img); tb.cand = h->d_cand; tb.candfit = h->d_candfit; tb.task = h->d_order_task; tb.pcap = h->pcap;
      rx.candfit = h->d_candfit;
      rx.biglist = h->d_biglist;
      hipLaunchKernelGGL((dpx::k_seg_accum<true>), dim3(per_frame_huge, batch), dim3(dpx::kAccWaves * 64), 0, s, prm, tb);
      CHK("k_seg_accum<huge>");
      const int per_frame_acc = std::max(1, std::min(32, 2048 / std::max(1, batch)));
      hipLaunchKernelGGL((dpx::k_seg_accum<false>), dim3(per_frame_acc, batch), dim3(dpx::kAccWaves * 64), 0, s, prm, tb);
      CHK("k_seg_accum");
      hipLaunchKernelGGL(dpx::k_seg_fit, dim3(static_cast<unsigned>((h->pcap + 255) / 256), static_cast<unsigned>(batch)), dim3(256), 0, s, prm, tb);
      CHK("k_seg_fit");
      DPX_IF_DIAG(if (h->d_dbg) hipLaunchKernelGGL((dpx::k_regions2<0, true, 5>), dim3(batch), dim3(dpx::kTail5RT), h->lds2_p3, s, prm, rb, rx); else)
      hipLaunchKernelGGL((dpx::k_regions2<0, false, 5>), dim3(batch), dim3(dpx::kTail5RT), h->lds2_p3, s, prm, rb, rx);
      CHK("k_regions2<0, ., 5>");
    }
  } else if (h->regions_tier == 0 && h->tier0_split && !h->d_dbg) {
    hipLaunchKernelGGL((dpx::k_regions2<0, false, 1>), dim3(batch), dim3(dpx::kRT), h->lds2_p1, s, prm, rb, rx);
    hipLaunchKernelGGL((dpx::k_regions2<0, false, 2>), dim3(batch), dim3(64), h->lds2_p2, s, prm, rb, rx);
    hipLaunchKernelGGL((dpx::k_regions2<0, false, 3>), dim3(batch), dim3(dpx::kTailRT), h->lds2_p3, s, prm, rb, rx);
  } else if (h->regions_tier == 1 && h->regions_split && !h->d_dbg) {
    hipLaunchKernelGGL((dpx::k_regions2<1, false, 1>), dim3(batch), dim3(dpx::kRT), h->lds2_p1, s, prm, rb, rx);
    hipLaunchKernelGGL((dpx::k_regions2<1, false, 2>), dim3(batch), dim3(64), h->lds2_p2, s, prm, rb, rx);
    hipLaunchKernelGGL((dpx::k_regions2<1, false, 3>), dim3(batch), dim3(dpx::kTailRT), h->lds2_p3, s, prm, rb, rx);
  }
#ifdef DPX_DIAG
  else if (h->regions_tier == 2 && h->d_dbg)
    hipLaunchKernelGGL((dpx::k_regions2<2, true, 0>), dim3(batch), dim3(dpx::kRT), h->lds2_bytes, s, prm, rb, rx);
  else if (h->regions_tier == 1 && h->d_dbg)
    hipLaunchKernelGGL((dpx::k_regions2<1, true, 0>), dim3(batch), dim3(dpx::kRT), h->lds2_bytes, s, prm, rb, rx);
  else if (h->regions_tier == 0 && h->d_dbg)
    hipLaunchKernelGGL((dpx::k_regions2<0, true, 0>), dim3(batch), dim3(dpx::kRT), h->lds2_bytes, s, prm, rb, rx);
#endif
  else if (h->regions_tier == 2 && h->regions_split) {
    const bool skip_p2 = skip && std::strchr(skip, '2'), skip_p3 = skip_p2 || (skip && std::strchr(skip, '3'));
    if (h->opt_prep_grow_fused && !skip_p2) {
      hipLaunchKernelGGL((dpx::k_regions2<2, false, 4>), dim3(batch), dim3(dpx::kPrepRT), h->lds2_p4, s, prm, rb, rx);
    } else {
      if (h->opt_prep_v1)
        hipLaunchKernelGGL((dpx::k_regions2<2, false, 1>), dim3(batch), dim3(dpx::kRT), h->lds2_p1, s, prm, rb, rx);
      else
        hipLaunchKernelGGL(dpx::k_region_prep, dim3(batch), dim3(dpx::kPrepRT), h->lds2_p1, s, prm, rb, rx);
      if (!skip_p2) hipLaunchKernelGGL((dpx::k_regions2<2, false, 2>), dim3(batch), dim3(64), h->lds2_p2, s, prm, rb, rx);
    }
    if (!skip_p3) hipLaunchKernelGGL((dpx::k_regions2<2, false, 3>), dim3(batch), dim3(dpx::kTailRT), h->lds2_p3, s, prm, rb, rx);
  } else if (h->regions_tier == 2)
    hipLaunchKernelGGL((dpx::k_regions2<2, false, 0>), dim3(batch), dim3(dpx::kRT), h->lds2_bytes, s, prm, rb, rx);
  else if (h->regions_tier == 1)
    hipLaunchKernelGGL((dpx::k_regions2<1, false, 0>), dim3(batch), dim3(dpx::kRT), h->lds2_bytes, s, prm, rb, rx);
  else if (h->regions_tier == 0)
    hipLaunchKernelGGL((dpx::k_regions2<0, false, 0>), dim3(batch), dim3(dpx::kRT), h->lds2_bytes, s, prm, rb, rx);
  else if (h->use_lds)
    hipLaunchKernelGGL(dpx::k_regions<true>, dim3(batch), dim3(64), h->lds_bytes, s, prm, rb);
  else
    hipLaunchKernelGGL(dpx::k_regions<false>, dim3(batch), dim3(64), h->lds_bytes, s, prm, rb);
  if (prof) DPX_HIP(hipEventRecord(ev[4], s));
  // ---- label expansion (per sub-batch: each has its own output buffer)
  if (prof) DPX_HIP(hipEventRecord(ev[5], s));
  if (!skip_l && !fuse_labels && !h->no_pixel_labels) {
    const bool labels_v1 = h->opt_labels_v1;
    if ((prm.W & 3) == 0 && prm.nv + 1 <= 65535 && !labels_v1) {
      // one launch for the whole group: block (frame, cell row) finds its output buffer through `outs`
      dpx::LabelOuts outs{};
      for (int j = 0; j < dpx::kMaxGroup; ++j) outs.out[j] = j < nsub ? subs[j].d_labels : nullptr;
      outs.per_out = per_sub;
      const int extra = prm.H > prm.nv * prm.P ? 1 : 0;
      if (prm.nv + extra > 0) {
        // At most DPX_LAB_RESIDENT (default 1) blocks per CU, each walking a task list (see k_labels_rows); 0 = one block
        // per task.  64 blocks already write at the full 5.6-5.8 TB/s, and every group in flight has its own label
        // launch: one block per CU keeps the launch from flooding the wave slots (pipeline +1.5 %; on its own 14.3 us
        // per 64 frames against 13.5 with a block per task).
        static const int resident = std::getenv("DPX_LAB_RESIDENT") ? std::atoi(std::getenv("DPX_LAB_RESIDENT")) : 1;
        const long long tasks = static_cast<long long>(batch) * (prm.nv + extra);
        long long blocks = tasks;
        if (resident > 0 && blocks > static_cast<long long>(resident) * device_cus()) blocks = static_cast<long long>(resident) * device_cus();
        hipLaunchKernelGGL(dpx::k_labels_rows, dim3(static_cast<unsigned>(blocks)), dim3(256), 0, s, prm, h->d_cell_labels, outs, prm.nv + extra);
      }
    } else {
      dpx::Params ps = prm;
      ps.batch = per_sub;
      for (int j = 0; j < nsub; ++j) {
        const int32_t* cl = h->d_cell_labels + static_cast<size_t>(j) * per_sub * prm.C;
        const long long per_frame = static_cast<long long>(prm.H) * ((prm.W + 3) / 4);
        const long long tot = per_frame * per_sub;
        if (tot > 0)
          hipLaunchKernelGGL(dpx::k_labels, dim3(static_cast<unsigned>((tot + 255) / 256)), dim3(256), 0, s, ps, cl, subs[j].d_labels);
      }
    }
  }
  // ---- optional: RANSAC refinement of the label map (plane_extractor.cpp:265-279)
  if (h->cfg.ransac_refinement && prm.C > 0) {
    dpx::Params ps = prm;
    ps.batch = per_sub;
    for (int j = 0; j < nsub; ++j) {
      ps.fx = subs[j].fx; ps.fy = subs[j].fy; ps.cx = subs[j].cx; ps.cy = subs[j].cy;
      dpx::RefineArgs ra{};
      ra.input = subs[j].d_in;
      ra.cell_labels = h->d_cell_labels + static_cast<size_t>(j) * per_sub * prm.C;
      ra.labels = subs[j].d_labels;
      ra.mt_init = h->d_mt_init;
      ra.max_iter = h->cfg.ransac_max_iterations;
      ra.thr = static_cast<double>(h->cfg.ransac_threshold);      // SetParamThreshold(double) <- float config value
      ra.ratio = static_cast<double>(h->cfg.ransac_inliers_ratio);
      ra.g_cells = h->d_refine_cells ? h->d_refine_cells + static_cast<size_t>(j) * per_sub * prm.C : nullptr;
      if (h->d_refine_cells) {
        if (depth)
          hipLaunchKernelGGL((dpx::k_refine<true, true>), dim3(per_sub), dim3(dpx::kRefT), h->refine_lds, s, ps, ra);
        else
          hipLaunchKernelGGL((dpx::k_refine<false, true>), dim3(per_sub), dim3(dpx::kRefT), h->refine_lds, s, ps, ra);
      } else if (depth)
        hipLaunchKernelGGL((dpx::k_refine<true, false>), dim3(per_sub), dim3(dpx::kRefT), h->refine_lds, s, ps, ra);
      else
        hipLaunchKernelGGL((dpx::k_refine<false, false>), dim3(per_sub), dim3(dpx::kRefT), h->refine_lds, s, ps, ra);
    }
  }
  if (prof) {
    DPX_HIP(hipEventRecord(ev[6], s));
    ++h->prof_calls;
  }
  DPX_HIP(hipGetLastError());
  return DPX_OK;
}

int run_pipeline(dpx_handle h, const void* d_in, bool depth, int batch, float fx, float fy, float cx, float cy,
                 int32_t* d_labels, hipStream_t s) {
  if (batch < 0 || batch > h->max_batch)
    return fail(DPX_ERR_ARG, "batch " + std::to_string(batch) + " exceeds max_batch " + std::to_string(h->max_batch));
  h->last_stream = s;
  h->last_batch = batch;
  if (batch == 0) return DPX_OK;
  SubBatch sb;
  int rc = fill_sub(h, sb, d_in, d_labels, depth, fx, fy, cx, cy);
  if (rc) return rc;
  rc = stage_fit(h, &sb, 1, batch, depth, s);
  if (rc) return rc;
  return stage_rest(h, &sb, 1, batch, depth, s);
}

int ensure_staging(dpx_handle h, size_t in_bytes, size_t label_bytes) {
  if (h->d_in_bytes < in_bytes) {
    if (h->d_in) hipFree(h->d_in);
    h->d_in = nullptr;
    h->d_in_bytes = 0;
    DPX_HIP(hipMalloc(&h->d_in, in_bytes));
    h->d_in_bytes = in_bytes;
  }
  if (h->d_labels_bytes < label_bytes) {
    if (h->d_labels) hipFree(h->d_labels);
    h->d_labels = nullptr;
    h->d_labels_bytes = 0;
    DPX_HIP(hipMalloc(reinterpret_cast<void**>(&h->d_labels), label_bytes));
    h->d_labels_bytes = label_bytes;
  }
  return DPX_OK;
}

// Cell-level output of a batch, written by the device straight into a page-locked host buffer (zero-copy stores over
// PCIe, 16 bytes per lane): the merged cell labels, the per-frame plane counts, their total, and the plane records packed
// back to back (frame f's records start at the sum of the earlier frames' counts).  A device-to-host hipMemcpy of a few
// hundred KB costs a fixed ~0.37 ms on this platform (tools/duplex_probe.py), more than everything else in the call.
// Block 0 also does the serial prefix over <= max_batch counts; every block then copies its share.
__global__ __launch_bounds__(256) void k_pack_out(const dpx_frame_stats* __restrict__ stats, const dpx_plane* __restrict__ planes, int pcap, int batch,
                                                  const int32_t* __restrict__ cell_labels, long long n_cell_words, int32_t* __restrict__ out_cells,
                                                  int32_t* __restrict__ out_counts, long long* __restrict__ out_total, dpx_plane* __restrict__ out_planes,
                                                  long long cap) {
  extern __shared__ long long off[];  // batch + 1
  for (int f = threadIdx.x; f < batch; f += blockDim.x) off[f + 1] = stats[f].segments < pcap ? stats[f].segments : pcap;
  __syncthreads();
  if (threadIdx.x == 0) {
    long long o = 0;
    for (int f = 0; f < batch; ++f) {
      const long long n = off[f + 1];
      off[f] = o;
      o += n;
    }
    off[batch] = o;
  }
  __syncthreads();
  if (blockIdx.x == 0) {
    for (int f = threadIdx.x; f < batch; f += blockDim.x) out_counts[f] = static_cast<int32_t>(off[f + 1] - off[f]);
    if (threadIdx.x == 0) *out_total = off[batch];
  }
  const long long tid = static_cast<long long>(blockIdx.x) * blockDim.x + threadIdx.x, nthr = static_cast<long long>(gridDim.x) * blockDim.x;
  {  // cell labels: int4 where the word count allows (device and host buffers are 16-byte aligned)
    const long long n4 = n_cell_words >> 2;
    const int4* s4 = reinterpret_cast<const int4*>(cell_labels);
    int4* d4 = reinterpret_cast<int4*>(out_cells);
    for (long long i = tid; i < n4; i += nthr) d4[i] = s4[i];
    for (long long i = (n4 << 2) + tid; i < n_cell_words; i += nthr) out_cells[i] = cell_labels[i];
  }
  if (out_planes) {
    constexpr int kWords = static_cast<int>(sizeof(dpx_plane) / 4);
    for (int f = 0; f < batch; ++f) {
      const long long o = off[f], n = off[f + 1] - o, room = cap - o;
      const long long words = (room <= 0 ? 0 : (n < room ? n : room)) * kWords;
      const uint32_t* src = reinterpret_cast<const uint32_t*>(planes + static_cast<size_t>(f) * pcap);
      uint32_t* dst = reinterpret_cast<uint32_t*>(out_planes + o);
      for (long long i = tid; i < words; i += nthr) dst[i] = src[i];
    }
  }
}

int host_process(dpx_handle h, const void* in, bool depth, int batch, float fx, float fy, float cx, float cy, int32_t* labels) {
  if (!h) return fail(DPX_ERR_ARG, "null handle");
  if (batch < 0 || batch > h->max_batch)
    return fail(DPX_ERR_ARG, "batch " + std::to_string(batch) + " exceeds max_batch " + std::to_string(h->max_batch));
  if (batch == 0) return DPX_OK;
  if (!in || !labels) return fail(DPX_ERR_ARG, "null buffer");
  DPX_HIP(hipSetDevice(h->device));
  const size_t N = static_cast<size_t>(h->H) * h->W;
  const size_t in_bytes = N * batch * (depth ? sizeof(uint16_t) : 3 * sizeof(float));
  const size_t lab_bytes = N * batch * sizeof(int32_t);
  int rc = ensure_staging(h, in_bytes, lab_bytes);
  if (rc) return rc;
  DPX_HIP(hipMemcpyAsync(h->d_in, in, in_bytes, hipMemcpyHostToDevice, h->stream));
  rc = run_pipeline(h, h->d_in, depth, batch, fx, fy, cx, cy, h->d_labels, h->stream);
  if (rc) return rc;
  DPX_HIP(hipMemcpyAsync(labels, h->d_labels, lab_bytes, hipMemcpyDeviceToHost, h->stream));
  DPX_HIP(hipStreamSynchronize(h->stream));
  return DPX_OK;
}

int sync_last(dpx_handle h) {
  DPX_HIP(hipSetDevice(h->device));
  DPX_HIP(hipStreamSynchronize(h->last_stream));
  return DPX_OK;
}

// ---- small test kernels ---------------------------------------------------------------------------------
__global__ void k_test_math(int which, const double* a, const double* b, double* out, long long n) {
  const long long i = static_cast<long long>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (i >= n) return;
  double s, c;
  switch (which) {
    case 0: out[i] = dpxm::datan2(a[i], b[i]); break;
    case 1: dpxm::sincos_small(a[i], &s, &c); out[i] = s; break;
    case 2: dpxm::sincos_small(a[i], &s, &c); out[i] = c; break;
    case 3: out[i] = dpxm::dacos(a[i]); break;
    case 5: out[i] = static_cast<double>(dpx::est_acos(static_cast<float>(a[i]))); break;      // float32 estimates of
    case 6: out[i] = static_cast<double>(dpx::est_atan2(static_cast<float>(a[i]), static_cast<float>(b[i]))); break;  // finish_cell
    default: out[i] = dpxm::dsqrt(a[i]); break;
  }
}

__global__ void k_test_eig3(const double* A, double* Q, double* w, int* rc, long long n) {
  const long long i = static_cast<long long>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (i >= n) return;
  dpx::Sym3 S{A[6 * i], A[6 * i + 1], A[6 * i + 2], A[6 * i + 3], A[6 * i + 4], A[6 * i + 5]};
  dpx::Eig3 e;
  dpx::eig3_hybrid(S, e);
  double* q = Q + 9 * i;
  q[0] = e.q00; q[1] = e.q01; q[2] = e.q02; q[3] = e.q10; q[4] = e.q11; q[5] = e.q12; q[6] = e.q20; q[7] = e.q21; q[8] = e.q22;
  w[3 * i] = e.w0; w[3 * i + 1] = e.w1; w[3 * i + 2] = e.w2;
  rc[i] = e.path;
}

}  // namespace

extern "C" {

const char* dpx_version(void) { return "deplex_amd 0.1 (gfx950)"; }

const char* dpx_last_error(void) { return g_err.c_str(); }

void dpx_default_config(dpx_config* c) {
  if (!c) return;
  c->patch_size = 10;
  c->histogram_bins_per_coord = 20;
  c->min_cos_angle_merge = 0.90f;
  c->max_merge_dist = 500.f;
  c->min_region_growing_candidate_size = 5;
  c->min_region_growing_cells_activated = 4;
  c->min_region_planarity_score = 0.55f;
  c->depth_sigma_coeff = 1.425e-6f;
  c->depth_sigma_margin = 10.f;
  c->min_pts_per_cell = 3;
  c->depth_discontinuity_threshold = 160.f;
  c->max_number_depth_discontinuity = 1;
  c->ransac_refinement = 0;
  c->ransac_max_iterations = 1000;
  c->ransac_threshold = 1.f;
  c->ransac_inliers_ratio = 0.9f;
}

int dpx_config_from_ini(const char* path, dpx_config* cfg) {
  if (!path || !cfg) return fail(DPX_ERR_ARG, "null argument");
  dpx_default_config(cfg);
  std::ifstream ini(path);
  if (!ini.is_open()) return fail(DPX_ERR_IO, std::string("Couldn't open ini file: ") + path);
  try {
    while (ini) {
      std::string line;
      std::getline(ini, line);
      if (line.empty() || line[0] == '#') continue;
      const size_t eq = line.find_first_of('=');
      if (eq == std::string::npos || eq == 0) continue;
      const std::string key = line.substr(0, eq), value = line.substr(eq + 1);
      if (key == "patchSize") cfg->patch_size = std::stoi(value);
      else if (key == "histogramBinsPerCoord") cfg->histogram_bins_per_coord = std::stoi(value);
      else if (key == "minCosAngleForMerge") cfg->min_cos_angle_merge = std::stof(value);
      else if (key == "maxMergeDist") cfg->max_merge_dist = std::stof(value);
      else if (key == "minRegionGrowingCandidateSize") cfg->min_region_growing_candidate_size = std::stoi(value);
      else if (key == "minRegionGrowingCellsActivated") cfg->min_region_growing_cells_activated = std::stoi(value);
      else if (key == "minRegionPlanarityScore") cfg->min_region_planarity_score = std::stof(value);
      else if (key == "depthSigmaCoeff") cfg->depth_sigma_coeff = std::stof(value);
      else if (key == "depthSigmaMargin") cfg->depth_sigma_margin = std::stof(value);
      else if (key == "minPtsPerCell") cfg->min_pts_per_cell = std::stoi(value);
      else if (key == "depthDiscontinuityThreshold") cfg->depth_discontinuity_threshold = std::stof(value);
      else if (key == "maxNumberDepthDiscontinuity") cfg->max_number_depth_discontinuity = std::stoi(value);
      else if (key == "ransacRefinement") cfg->ransac_refinement = static_cast<bool>(std::stoi(value)) ? 1 : 0;
      else if (key == "ransacMaxIterations") cfg->ransac_max_iterations = std::stoi(value);
      else if (key == "ransacThreshold") cfg->ransac_threshold = std::stof(value);
      else if (key == "ransacInliersRatio") cfg->ransac_inliers_ratio = std::stof(value);
      else std::fprintf(stderr, "Unknown parameter name: %s\n", key.c_str());
    }
  } catch (const std::exception& e) {
    return fail(DPX_ERR_ARG, std::string("invalid value in ini file: ") + e.what());
  }
  return DPX_OK;
}

static int create_handle(const dpx_config* cfg, int32_t image_height, int32_t image_width, int32_t max_batch, int32_t device,
                         bool stream_mode, dpx_handle* out);

int dpx_create(const dpx_config* cfg, int32_t image_height, int32_t image_width, int32_t max_batch, int32_t device,
               dpx_handle* out) {
  return create_handle(cfg, image_height, image_width, max_batch, device, false, out);
}

static int create_handle(const dpx_config* cfg, int32_t image_height, int32_t image_width, int32_t max_batch, int32_t device,
                         bool stream_mode, dpx_handle* out) {
  if (!cfg || !out) return fail(DPX_ERR_ARG, "null argument");
  *out = nullptr;
  // plane_extractor.cpp:161-164
  if (cfg->patch_size == 0)
    return fail(DPX_ERR_PATCH_ZERO, "Error! Invalid config parameter: patchSize(" + std::to_string(cfg->patch_size) +
                                        "). patchSize has to be positive.");
  if (cfg->patch_size < 0 || cfg->min_pts_per_cell == 0 || cfg->histogram_bins_per_coord < 1 || image_height < 0 ||
      image_width < 0)
    return fail(DPX_ERR_UNDEFINED,
                "configuration undefined in the reference (negative patchSize, minPtsPerCell==0 or histogramBinsPerCoord<1)");
  if (max_batch < 1) return fail(DPX_ERR_ARG, "max_batch must be >= 1");
  int ndev = 0;
  hipError_t e = hipGetDeviceCount(&ndev);
  if (e != hipSuccess || ndev == 0) return fail(DPX_ERR_NO_DEVICE, "no HIP device available (this library has no CPU path)");
  if (device < 0 || device >= ndev) return fail(DPX_ERR_NO_DEVICE, "HIP device index out of range");
  DPX_HIP(hipSetDevice(device));

  dpx_handle h = new dpx_handle_s();
  h->cfg = *cfg;
  h->H = image_height;
  h->W = image_width;
  h->max_batch = max_batch;
  h->device = device;
  h->stream_mode = stream_mode;
  // plane_extractor.cpp:155-160
  const int nh = image_width / std::max(cfg->patch_size, 1);
  const int nv = image_height / std::max(cfg->patch_size, 1);
  h->cfg.patch_size = std::min(cfg->patch_size, std::min(image_height, image_width));
  dpx::Params& p = h->prm;
  p.H = image_height; p.W = image_width; p.P = h->cfg.patch_size; p.nh = nh; p.nv = nv;
  p.C = nh * nv;
  p.N = image_height * image_width;
  p.n = p.P * p.P;
  p.stride_w = nh * p.P;
  p.sheared = (p.stride_w != image_width) ? 1 : 0;
  p.batch = 0;
  p.fx = p.fy = 1.f; p.cx = p.cy = 0.f;
  p.min_cos = cfg->min_cos_angle_merge;
  p.max_merge_dist = cfg->max_merge_dist;
  p.min_score = cfg->min_region_planarity_score;
  p.sigma_coeff = cfg->depth_sigma_coeff;
  p.sigma_margin = cfg->depth_sigma_margin;
  p.disc_thr = cfg->depth_discontinuity_threshold;
  p.tol_sin = sqrtf(1.0f - cfg->min_cos_angle_merge * cfg->min_cos_angle_merge);  // cell_segment.cpp:106
  p.hist_B = cfg->histogram_bins_per_coord;
  p.min_cand = cfg->min_region_growing_candidate_size;
  p.min_cells = cfg->min_region_growing_cells_activated;
  p.max_disc = cfg->max_number_depth_discontinuity;
  p.valid_thr = (3ll * p.n) / cfg->min_pts_per_cell;  // cell_segment.cpp:23
  if (p.P == 0) {  // zero-sized image: no cells
    p.P = 1;
    p.n = 1;
  }
  const long long nb = static_cast<long long>(p.hist_B) * p.hist_B;
  const size_t hist_bytes = static_cast<size_t>(3 * nb + 1) * 4;
  if (hist_bytes > 64 * 1024) {
    delete h;
    return fail(DPX_ERR_CAPACITY, "histogramBinsPerCoord too large for the on-chip histogram (max 73)");
  }
  const size_t C = static_cast<size_t>(p.C);
  const size_t lds_state = hist_bytes + 6 * C + 16;
  h->opt_nofuse = std::getenv("DPX_FUSED_LABELS") == nullptr;  // fusing the label expansion into the region tail measured 3-5 % slower
  h->opt_labels_v1 = std::getenv("DPX_LABELS_V1") != nullptr;
  h->opt_cloud_nolds = std::getenv("DPX_CLOUD_NOLDS") != nullptr;
  h->opt_prep_v1 = std::getenv("DPX_PREP_V1") != nullptr;
  h->opt_sort_bitonic = std::getenv("DPX_SORT_COUNTING") == nullptr;
  // preparation and grow in one launch (tier 2, lean preparation, C small enough for its key hand-over; +1 % in long runs,
  // +2.5 % in 20-step ones); DPX_PREP_GROW_SPLIT restores the two launches (test hook)
  h->opt_prep_grow_fused = !h->opt_prep_v1 && static_cast<long long>(C) <= dpx::kFusedMaxCells && !std::getenv("DPX_PREP_GROW_SPLIT");  // test hook: tier-2 preparation with the whole image in LDS
  if (const char* e = std::getenv("DPX_CSR_CHUNK")) {  // test hook: force the wide-row walk of the merge phase
    const int v = std::atoi(e);
    h->opt_csr_chunk = v < 0 ? 0 : (v > 64 ? 64 : v);
  }
  const char* force = std::getenv("DPX_FORCE_GLOBAL_STATE");
  h->use_lds = lds_state <= kLdsLimit && !(force && force[0] == '1');
  h->lds_bytes = h->use_lds ? lds_state : hist_bytes + 16;
  h->pcap = static_cast<int>(std::max<long long>(1, cfg->min_region_growing_cells_activated <= 1
                                                      ? static_cast<long long>(C)
                                                      : static_cast<long long>(C) / cfg->min_region_growing_cells_activated));
  {
    // region-growing kernel tier (dpx_regions2.h): 2 = everything in LDS (9 B/cell + 8 B/sort slot while preparing),
    // 1 = lean (7 B/cell), 0 = flag byte only (1 B/cell, queues in global memory), else the v1 kernel
    const char* v1 = std::getenv("DPX_REGIONS_V1");
    const char* lean = std::getenv("DPX_REGIONS_LEAN");
    const char* gst = std::getenv("DPX_REGIONS_GSTATE");
    const bool want_lean = lean && lean[0] == '1', want_gst = gst && gst[0] == '1';
    size_t pad2 = 1;
    while (pad2 < C) pad2 <<= 1;
    h->regions_tier = -1;
    if (!(v1 && v1[0] == '1') && !(force && force[0] == '1') && nb < 4600) {
      const int Ci = static_cast<int>(C), nbi = static_cast<int>(nb);
      const dpx::R2Layout l2 = dpx::r2_layout(2, Ci, nbi, nh), l1 = dpx::r2_layout(1, Ci, nbi, nh), l0 = dpx::r2_layout(0, Ci, nbi, nh);
      // (grow_end: the grow phase's scratch behind the image, see r2_layout)
      const bool fits2 = C < 65000 && std::max<size_t>(std::max<size_t>(l2.tail_end, l2.grow_end), l2.image_end + 8 * pad2) <= kLdsLimit;
      const bool fits1 = C < 65000 && std::max<size_t>(l1.tail_end, l1.grow_end) <= kLdsLimit;
      const bool fits0 = C < (size_t(1) << dpx::kKeyCellBits) && std::max<size_t>(l0.tail_end, l0.grow_end) <= kLdsLimit;
      // Mid-size frames in stream mode (slots of a dpx_pipeline; DPX_REGIONS_GSPLIT=1 forces it for any handle): tier 0 in
      // split form.  Its one-wave grow launch holds the flag bytes and a small scratch (29.5 KB for 1280x960) instead of
      // tier 1's 9 B per cell (135 KB): five frames' grow waves per CU instead of one.  A frame's own latency is ~10 %
      // longer (queues and seed lists in global memory), a stream of batches faster by a third and more (DESIGN.md
      // section 5).  Taken for frames beyond tier 2 when at least two such launches fit a CU.
      const char* gsplit = std::getenv("DPX_REGIONS_GSPLIT");
      const bool force_split = gsplit && gsplit[0] == '1';
      const bool split0 = fits0 && !want_gst && !want_lean && 2 * static_cast<size_t>(dpx::r2_lean_grow_end(l0, nbi)) <= kLdsLimit &&
                          (force_split || (stream_mode && !(fits2 && !want_lean)));
      if (fits2 && !want_lean && !want_gst && !force_split) {
        h->regions_tier = 2;
        h->sort_pad = static_cast<int>(pad2);
        h->lds2_img = l2.image_end;
        h->lds2_p1 = h->opt_prep_v1 ? l2.image_end + 8 * pad2 : dpx::r2_prep_lds_bytes(l2, Ci);
        h->lds2_p2 = l2.grow_end;
        h->lds2_p3 = l2.tail_end;
        h->lds2_p4 = std::max<size_t>(dpx::r2_fused_lds_bytes(l2, Ci), l2.grow_end);
        h->lds2_bytes = std::max<size_t>(std::max<size_t>(l2.tail_end, l2.grow_end), l2.image_end + 8 * pad2);
      } else if (split0) {
        h->regions_tier = 0;
        h->tier0_split = true;
        h->lds2_img = l0.image_end;
        h->lds2_p1 = l0.image_end;
        h->lds2_p2 = dpx::r2_lean_grow_end(l0, nbi);
        if (const char* pad = std::getenv("DPX_GROW_LDS_PAD"))  // tuning hook: KB added to the one-wave grow launch (fewer frames per CU)
          h->lds2_p2 += static_cast<size_t>(std::max(0, std::atoi(pad))) * 1024;
        h->lds2_p3 = l0.tail_end;
        h->lds2_bytes = std::max<size_t>(l0.tail_end, l0.grow_end);
      } else if (fits1 && !want_gst) {
        h->regions_tier = 1;
        h->lds2_img = l1.image_end;
        h->lds2_p1 = l1.image_end;
        h->lds2_p2 = l1.grow_end;
        h->lds2_p3 = l1.tail_end;
        h->lds2_bytes = std::max<size_t>(l1.tail_end, l1.grow_end);
      } else if (fits0) {
        h->regions_tier = 0;
        h->lds2_img = l0.image_end;
        h->lds2_bytes = std::max<size_t>(l0.tail_end, l0.grow_end);
      }
    }
    // Frames beyond tier 2 grow their regions through CORES (dpx_regions3.h): cores + sequence + order launches, then the
    // tier-0 tail.  DPX_REGIONS_CORES=1 forces the path for any frame that fits it (tests on small frames), the switches that
    // name one of the cell-level tiers (DPX_REGIONS_LEAN / _GSTATE / _GSPLIT / _V1, DPX_REGIONS_V2=1) keep those as cross-checks.
    {
      const char* fc = std::getenv("DPX_REGIONS_CORES");
      const char* v2 = std::getenv("DPX_REGIONS_V2");
      const char* gsplit = std::getenv("DPX_REGIONS_GSPLIT");
      const bool force_cores = fc && fc[0] == '1';
      const bool named_old = (v1 && v1[0] == '1') || want_lean || want_gst || (gsplit && gsplit[0] == '1') || (v2 && v2[0] == '1') || (force && force[0] == '1');
      const int Ci = static_cast<int>(C), nbi = static_cast<int>(nb);
      const dpx::R2Layout l0 = dpx::r2_layout(0, Ci, nbi, nh);
      const bool fits_cores = nb < 4600 && C >= 1 && C < (size_t(1) << dpx::kKeyCellBits) && l0.tail_end <= kLdsLimit &&
                              dpx::s3_layout(nbi, Ci).end <= kLdsLimit && dpx::order_lds_bytes(Ci) + 8192 <= kLdsLimit;
      // default: frames beyond tier 1's capacity (the 4K class: twice as fast there).  Mid-size frames (1280x960) keep the
      // cell-level tiers: on frames with thousands of one-way edges the core path's sequence is the slower one
      const dpx::R2Layout l1c = dpx::r2_layout(1, Ci, nbi, nh);
      const bool big = !(C < 65000 && std::max<size_t>(l1c.tail_end, l1c.grow_end) <= kLdsLimit);
      if (fits_cores && (force_cores || (!named_old && h->regions_tier != 2 && big))) {
        h->cores = true;
        h->core_logT = 6;
        while ((size_t(1) << h->core_logT) < 2 * C) ++h->core_logT;
        h->core_T = 1 << h->core_logT;
        h->lds3_sim = dpx::s3_layout(nbi, Ci).end;
        h->lds3_order = dpx::order_lds_bytes(Ci);
        h->regions_tier = 0;
        h->tier0_split = false;
        h->sort_pad = 0;
        h->lds2_img = 192;  // what the tail launch takes over from the sequence launch: the header
        h->lds2_p1 = h->lds2_p2 = h->lds2_p4 = 0;
        h->lds2_p3 = l0.tail_end;
        h->lds2_bytes = l0.tail_end;
      }
    }
    if (h->regions_tier == 1 || h->regions_tier == 0) {
      h->Cpad = 1;
      while (static_cast<size_t>(h->Cpad) < C) h->Cpad <<= 1;
    }
  }
  const size_t B = static_cast<size_t>(max_batch);
  int rc = DPX_OK;
  auto A = [&](int r) { if (rc == DPX_OK) rc = r; };
  A(dev_alloc(&h->d_cells, B * C));
  A(dev_alloc(&h->d_meta, B * C));
  A(dev_alloc(&h->d_order, B * C));
  A(dev_alloc(&h->d_grow_start, B * (C + 1)));
  A(dev_alloc(&h->d_cand, B * h->pcap * 12));
  A(dev_alloc(&h->d_planes, B * h->pcap));
  A(dev_alloc(&h->d_cell_labels, B * C));
  A(dev_alloc(&h->d_pairs, B * 2 * C));
  A(dev_alloc(&h->d_merge, B * h->pcap));
  A(dev_alloc(&h->d_stats, B));
  A(dev_alloc(&h->d_mse_c, B * C));
  A(dev_alloc(&h->d_cand_grow, B * h->pcap * 4));
  if (!h->use_lds && h->regions_tier < 0) {
    A(dev_alloc(&h->d_gflags, B * C));
    A(dev_alloc(&h->d_gowner, B * C));
    A(dev_alloc(&h->d_gqueue, B * C));
  }
  if (h->regions_tier == 1 || h->regions_tier == 0) {
    A(dev_alloc(&h->d_sort_keys, B * static_cast<size_t>(h->Cpad)));
    A(dev_alloc(&h->d_bin_off, B * static_cast<size_t>(nb + 1)));
    if (!h->cores) A(dev_alloc(&h->d_pbcnt_own, B * static_cast<size_t>(nb)));
    if (!h->cores) h->d_pbcnt = h->d_pbcnt_own;
  }
  if (h->regions_tier == 0) {
    A(dev_alloc(&h->d_gqpos, B * C));  // (the core path's tail still uses it: bit row of the merge walk's wide rows)
    A(dev_alloc(&h->d_gqueue, B * C));
  }
  if (h->cores) {
    const size_t T = static_cast<size_t>(h->core_T);
    A(dev_alloc(&h->d_core, B * (16 * C + 2 * T + 2 * static_cast<size_t>(nb) + 2 * dpx::kCtr) + 32));
    A(dev_alloc(&h->d_candfit, B * static_cast<size_t>(h->pcap) * 12));
    A(dev_alloc(&h->d_oedges, B * 4 * C));
    A(dev_alloc(&h->d_crec, B * C));
    A(dev_alloc(&h->d_hbest, B * T));
    A(dev_alloc(&h->d_lists, B * C * 2));
    if (rc == DPX_OK) {
      uint32_t* q = h->d_core;
      auto take = [&](uint32_t*& dst, size_t n) { dst = q; q += B * n; };
      take(h->d_parent, C); take(h->d_csize, C); take(h->d_npairs, C); take(h->d_pstart, C); take(h->d_pfill, C); take(h->d_noe, C);
      take(h->d_oestart, C); take(h->d_oefill, C); take(h->d_corepairs, C); take(h->d_cgrow, C); take(h->d_cellq, C); take(h->d_worklist, 3 * C);
      take(h->d_biglist, C); take(h->d_hslots, C); take(h->d_hkey, T); take(h->d_hcnt, T); take(h->d_histg, static_cast<size_t>(nb)); take(h->d_pbcnt, static_cast<size_t>(nb));
      q = reinterpret_cast<uint32_t*>((reinterpret_cast<uintptr_t>(q) + 127u) & ~static_cast<uintptr_t>(127u));  // (a 128-byte line per frame and counter set)
      take(h->d_ccursor, dpx::kCtr);
      h->d_order_task = reinterpret_cast<int*>(q);
      // the pair table starts empty; every call leaves it empty again (k_pair_scatter)
      if (hipMemset(h->d_hkey, 0xFF, B * T * 4) != hipSuccess || hipMemset(h->d_hcnt, 0, B * T * 4) != hipSuccess ||
          hipMemset(h->d_hbest, 0xFF, B * T * 8) != hipSuccess)
        rc = fail(DPX_ERR_HIP, "hipMemset failed");
    }
  }
  if (std::getenv("DPX_RESIDENCY")) {
    A(dev_alloc(&h->d_res, B * 4));
    if (rc == DPX_OK) (void)hipMemset(h->d_res, 0, B * 4 * sizeof(unsigned long long));
  }
  if (rc == DPX_OK && hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking) != hipSuccess)
    rc = fail(DPX_ERR_HIP, "hipStreamCreate failed");
  if (rc == DPX_OK && h->lds_bytes > 48 * 1024) {
    hipError_t e2 = h->use_lds ? hipFuncSetAttribute(reinterpret_cast<const void*>(dpx::k_regions<true>),
                                                     hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(h->lds_bytes))
                               : hipFuncSetAttribute(reinterpret_cast<const void*>(dpx::k_regions<false>),
                                                     hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(h->lds_bytes));
    if (e2 != hipSuccess) rc = fail(DPX_ERR_HIP, std::string("hipFuncSetAttribute: ") + hipGetErrorString(e2));
  }
  if (rc == DPX_OK && cfg->ransac_refinement) {
    uint32_t mt[624];
    mt[0] = 5489u;  // std::mt19937 default seed
    for (uint32_t i = 1; i < 624; ++i) mt[i] = 1812433253u * (mt[i - 1] ^ (mt[i - 1] >> 30)) + i;
    A(dev_alloc(&h->d_mt_init, 624));
    if (rc == DPX_OK && hipMemcpy(h->d_mt_init, mt, sizeof(mt), hipMemcpyHostToDevice) != hipSuccess) rc = fail(DPX_ERR_HIP, "hipMemcpy failed");
    const size_t refine_fixed = (624 + 32 + 16 + 2 * (static_cast<size_t>(nv) + 1)) * 4 + 64;
    h->refine_lds = refine_fixed + C * 4;
    const char* gc = std::getenv("DPX_REFINE_GCELLS");  // test hook: cell list in global scratch for any frame size
    if (h->refine_lds > kLdsLimit || (gc && gc[0] == '1')) {  // the label's cell list goes to global scratch
      h->refine_lds = refine_fixed;
      A(dev_alloc(&h->d_refine_cells, B * C));
    }
    if (h->refine_lds > kLdsLimit) rc = fail(DPX_ERR_CAPACITY, "image too tall for the RANSAC refinement kernel's row tables");
    if (rc == DPX_OK && h->refine_lds > 48 * 1024) {
      const void* fns[4] = {reinterpret_cast<const void*>(dpx::k_refine<true, false>), reinterpret_cast<const void*>(dpx::k_refine<false, false>),
                            reinterpret_cast<const void*>(dpx::k_refine<true, true>), reinterpret_cast<const void*>(dpx::k_refine<false, true>)};
      for (const void* fn : fns)
        if (hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(h->refine_lds)) != hipSuccess)
          rc = fail(DPX_ERR_HIP, "hipFuncSetAttribute failed");
    }
  }
  if (rc == DPX_OK && (h->regions_tier == 1 || h->regions_tier == 0) && !h->cores && !h->opt_sort_bitonic &&
      static_cast<size_t>(std::min(static_cast<int>(C), dpx::kCellSortMax)) * 8 > 48 * 1024) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(dpx::k_bin_sort<true>), hipFuncAttributeMaxDynamicSharedMemorySize,
                            std::min(static_cast<int>(C), dpx::kCellSortMax) * 8) != hipSuccess)
      rc = fail(DPX_ERR_HIP, "hipFuncSetAttribute failed");
  }
  if (rc == DPX_OK && h->cores && h->lds3_order + 8192 > 48 * 1024) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(dpx::k_order<false, true>), hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(h->lds3_order)) != hipSuccess
        DPX_IF_DIAG(|| hipFuncSetAttribute(reinterpret_cast<const void*>(dpx::k_order<true, true>), hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(h->lds3_order)) != hipSuccess))
      rc = fail(DPX_ERR_HIP, "hipFuncSetAttribute failed");
  }
  if (rc == DPX_OK && h->cores && h->lds3_sim > 48 * 1024) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(dpx::k_regsim<false>), hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(h->lds3_sim)) != hipSuccess
        DPX_IF_DIAG(|| hipFuncSetAttribute(reinterpret_cast<const void*>(dpx::k_regsim<true>), hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(h->lds3_sim)) != hipSuccess))
      rc = fail(DPX_ERR_HIP, "hipFuncSetAttribute failed");
  }
  if (rc == DPX_OK && (h->regions_tier == 2 || h->regions_tier == 1 || h->tier0_split || h->cores)) {
    const char* nosplit = std::getenv("DPX_REGIONS_NOSPLIT");
    h->regions_split = h->tier0_split || h->cores || !(nosplit && nosplit[0] == '1');
    if (h->regions_split) {
      A(dev_alloc(&h->d_lds_img, B * (h->lds2_img / 16)));
      const void* fns[3];
      if (h->regions_tier == 2) {
        fns[0] = h->opt_prep_v1 ? reinterpret_cast<const void*>(dpx::k_regions2<2, false, 1>) : reinterpret_cast<const void*>(dpx::k_region_prep);
        fns[1] = reinterpret_cast<const void*>(dpx::k_regions2<2, false, 2>);
        fns[2] = reinterpret_cast<const void*>(dpx::k_regions2<2, false, 3>);
      } else if (h->regions_tier == 1) {
        fns[0] = reinterpret_cast<const void*>(dpx::k_regions2<1, false, 1>);
        fns[1] = reinterpret_cast<const void*>(dpx::k_regions2<1, false, 2>);
        fns[2] = reinterpret_cast<const void*>(dpx::k_regions2<1, false, 3>);
      } else {
        fns[0] = reinterpret_cast<const void*>(dpx::k_regions2<0, false, 1>);
        fns[1] = reinterpret_cast<const void*>(dpx::k_regions2<0, false, 2>);
        fns[2] = h->cores ? reinterpret_cast<const void*>(dpx::k_regions2<0, false, 5>) : reinterpret_cast<const void*>(dpx::k_regions2<0, false, 3>);
      }
      const size_t need[3] = {h->lds2_p1, h->lds2_p2, h->lds2_p3};
      for (int i = 0; i < 3 && rc == DPX_OK; ++i)
        if (need[i] > 48 * 1024 &&
            hipFuncSetAttribute(fns[i], hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(need[i])) != hipSuccess)
          rc = fail(DPX_ERR_HIP, "hipFuncSetAttribute failed");
      if (rc == DPX_OK && h->regions_tier == 2 && h->lds2_p4 > 48 * 1024 &&
          hipFuncSetAttribute(reinterpret_cast<const void*>(dpx::k_regions2<2, false, 4>), hipFuncAttributeMaxDynamicSharedMemorySize,
                              static_cast<int>(h->lds2_p4)) != hipSuccess)
        rc = fail(DPX_ERR_HIP, "hipFuncSetAttribute failed");
    }
  }
  if (rc == DPX_OK && h->regions_tier >= 0 && h->lds2_bytes > 48 * 1024) {
    const void* fn = h->regions_tier == 2   ? reinterpret_cast<const void*>(dpx::k_regions2<2, false, 0>)
                     : h->regions_tier == 1 ? reinterpret_cast<const void*>(dpx::k_regions2<1, false, 0>)
                                            : reinterpret_cast<const void*>(dpx::k_regions2<0, false, 0>);
    hipError_t e3 = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(h->lds2_bytes));
    if (e3 != hipSuccess) rc = fail(DPX_ERR_HIP, std::string("hipFuncSetAttribute: ") + hipGetErrorString(e3));
  }
  if (rc != DPX_OK) {
    const std::string keep = g_err;
    free_all(h);
    delete h;
    g_err = keep;
    return rc;
  }
  h->last_stream = h->stream;
  *out = h;
  return DPX_OK;
}

void dpx_destroy(dpx_handle h) {
  if (!h) return;
  (void)hipSetDevice(h->device);
  (void)hipDeviceSynchronize();
  free_all(h);
  delete h;
}

int dpx_process_pcd(dpx_handle h, const float* xyz, int64_t n_points, int32_t* labels) {
  if (!h) return fail(DPX_ERR_ARG, "null handle");
  // plane_extractor.cpp:188-194
  if (n_points != static_cast<int64_t>(h->W) * h->H)
    return fail(DPX_ERR_SHAPE, "Error! Number of points doesn't match image shape: " + std::to_string(n_points) +
                                   " != " + std::to_string(h->H) + " x " + std::to_string(h->W));
  if (n_points == 0) return DPX_OK;
  return host_process(h, xyz, false, 1, 1.f, 1.f, 0.f, 0.f, labels);
}

int dpx_process_pcd_batch(dpx_handle h, const float* xyz, int32_t batch, int32_t* labels) {
  return host_process(h, xyz, false, batch, 1.f, 1.f, 0.f, 0.f, labels);
}

int dpx_process_depth_batch(dpx_handle h, const uint16_t* depth, int32_t batch, float fx, float fy, float cx, float cy,
                            int32_t* labels) {
  return host_process(h, depth, true, batch, fx, fy, cx, cy, labels);
}

int dpx_process_depth_batch_cells(dpx_handle h, const uint16_t* depth, int32_t batch, float fx, float fy, float cx, float cy,
                                  int32_t* cell_labels, int32_t* n_planes, dpx_plane* planes, int64_t planes_cap, int64_t* planes_total) {
  if (!h) return fail(DPX_ERR_ARG, "null handle");
  if (batch < 0 || batch > h->max_batch)
    return fail(DPX_ERR_ARG, "batch " + std::to_string(batch) + " exceeds max_batch " + std::to_string(h->max_batch));
  if (h->cfg.ransac_refinement) return fail(DPX_ERR_ARG, "cell-level output is not defined with ransac_refinement (it edits single pixels)");
  if (planes_total) *planes_total = 0;
  if (batch == 0) {  // an empty batch has no planes: dpx_get_packed_planes answers 0 records (no stale total of an earlier call)
    h->cells_stage_total = 0;
    return DPX_OK;
  }
  if (!depth || !cell_labels) return fail(DPX_ERR_ARG, "null buffer");
  if (planes && planes_cap < 0) return fail(DPX_ERR_ARG, "negative planes_cap");
  DPX_HIP(hipSetDevice(h->device));
  const size_t N = static_cast<size_t>(h->H) * h->W, C = static_cast<size_t>(h->prm.C);
  int rc = ensure_staging(h, N * batch * sizeof(uint16_t), 16);
  if (rc) return rc;
  // page-locked staging, allocated at the first cell-level call: [cell labels][counts][total][packed plane records]
  const size_t B = static_cast<size_t>(h->max_batch);
  const size_t off_counts = (B * C * sizeof(int32_t) + 15) & ~size_t(15);
  const size_t off_total = off_counts + ((B * sizeof(int32_t) + 15) & ~size_t(15));
  const size_t off_planes = off_total + 16;
  const size_t pack_cap = B * static_cast<size_t>(h->pcap);
  if (!h->h_cells_stage) {
    DPX_HIP(hipHostMalloc(reinterpret_cast<void**>(&h->h_cells_stage), off_planes + pack_cap * sizeof(dpx_plane), hipHostMallocDefault));
  }
  char* st = h->h_cells_stage;
  const bool want_planes = planes != nullptr || n_planes != nullptr || planes_total != nullptr;  // (packed whenever counts are asked for: dpx_get_packed_planes)
  DPX_HIP(hipMemcpyAsync(h->d_in, depth, N * batch * sizeof(uint16_t), hipMemcpyHostToDevice, h->stream));
  h->no_pixel_labels = true;
  rc = run_pipeline(h, h->d_in, true, batch, fx, fy, cx, cy, h->d_labels, h->stream);
  h->no_pixel_labels = false;
  if (rc) return rc;
  const long long words = static_cast<long long>(C) * batch;
  const unsigned blocks = static_cast<unsigned>(std::max<long long>(1, std::min<long long>(256, (words / 4 + 255) / 256)));
  hipLaunchKernelGGL(k_pack_out, dim3(blocks), dim3(256), (static_cast<size_t>(batch) + 1) * sizeof(long long), h->stream, h->d_stats, h->d_planes,
                     h->pcap, batch, h->d_cell_labels, words, reinterpret_cast<int32_t*>(st), reinterpret_cast<int32_t*>(st + off_counts),
                     reinterpret_cast<long long*>(st + off_total), want_planes ? reinterpret_cast<dpx_plane*>(st + off_planes) : nullptr,
                     static_cast<long long>(pack_cap));
  DPX_HIP(hipGetLastError());
  DPX_HIP(hipStreamSynchronize(h->stream));
  std::memcpy(cell_labels, st, static_cast<size_t>(words) * sizeof(int32_t));
  const long long total = *reinterpret_cast<const long long*>(st + off_total);
  h->cells_stage_total = want_planes ? total : -1;
  h->cells_stage_planes_off = off_planes;
  if (n_planes) std::memcpy(n_planes, st + off_counts, static_cast<size_t>(batch) * sizeof(int32_t));
  if (planes_total) *planes_total = total;
  if (planes) {
    const long long ncopy = total < planes_cap ? total : planes_cap;
    if (ncopy > 0) std::memcpy(planes, st + off_planes, static_cast<size_t>(ncopy) * sizeof(dpx_plane));
    if (total > planes_cap)
      return fail(DPX_ERR_CAPACITY, "plane buffer holds " + std::to_string(planes_cap) + " records, the batch has " + std::to_string(total));
  }
  return DPX_OK;
}

int dpx_get_packed_planes(dpx_handle h, dpx_plane* planes, int64_t planes_cap, int64_t* planes_total) {
  if (!h) return fail(DPX_ERR_ARG, "null handle");
  if (h->cells_stage_total == 0) {  // (also the empty batch, which never allocated the staging)
    if (planes_total) *planes_total = 0;
    return DPX_OK;
  }
  if (!h->h_cells_stage || h->cells_stage_total < 0) return fail(DPX_ERR_ARG, "no cell-level call with plane counts precedes this one");
  if (planes_total) *planes_total = h->cells_stage_total;
  if (planes_cap < 0 || (planes_cap > 0 && !planes)) return fail(DPX_ERR_ARG, "bad plane buffer");
  const long long ncopy = h->cells_stage_total < planes_cap ? h->cells_stage_total : planes_cap;
  if (ncopy > 0) std::memcpy(planes, h->h_cells_stage + h->cells_stage_planes_off, static_cast<size_t>(ncopy) * sizeof(dpx_plane));
  if (h->cells_stage_total > planes_cap)
    return fail(DPX_ERR_CAPACITY, "plane buffer holds " + std::to_string(planes_cap) + " records, the batch has " + std::to_string(h->cells_stage_total));
  return DPX_OK;
}

int dpx_expand_cell_labels(int32_t image_height, int32_t image_width, int32_t patch_size, const int32_t* cell_labels, int32_t batch,
                           int32_t* labels) {
  if (image_height < 0 || image_width < 0 || batch < 0 || patch_size <= 0) return fail(DPX_ERR_ARG, "bad shape");
  if (batch == 0 || image_height == 0 || image_width == 0) return DPX_OK;
  if (!cell_labels || !labels) return fail(DPX_ERR_ARG, "null buffer");
  // grid of plane_extractor.cpp:155-160: nh, nv from the configured patch size, P clamped to the image afterwards
  const int nh = image_width / patch_size, nv = image_height / patch_size;
  const int P = std::min(patch_size, std::min(image_height, image_width));
  for (int32_t f = 0; f < batch; ++f) {
    const int32_t* cl = cell_labels + static_cast<size_t>(f) * nh * nv;
    int32_t* out = labels + static_cast<size_t>(f) * image_height * image_width;
    for (int row = 0; row < image_height; ++row) {
      const int cr = row / P;
      int32_t* o = out + static_cast<size_t>(row) * image_width;
      for (int col = 0; col < image_width; ++col) {
        const int cc = col / P;
        o[col] = (cr < nv && cc < nh) ? cl[static_cast<size_t>(cr) * nh + cc] : 0;  // plane_extractor.cpp:464-465 (labels already merged)
      }
    }
  }
  return DPX_OK;
}

void* dpx_host_alloc(size_t bytes) {
  void* p = nullptr;
  if (hipHostMalloc(&p, bytes ? bytes : 1, hipHostMallocPortable) != hipSuccess) {
    fail(DPX_ERR_HIP, "hipHostMalloc failed");
    return nullptr;
  }
  return p;
}

void dpx_host_free(void* p) {
  if (p) hipHostFree(p);
}

int dpx_process_depth_batch_device(dpx_handle h, const void* d_depth, int32_t batch, float fx, float fy, float cx,
                                   float cy, void* d_labels, void* hip_stream) {
  if (!h) return fail(DPX_ERR_ARG, "null handle");
  if (batch > 0 && (!d_depth || !d_labels)) return fail(DPX_ERR_ARG, "null buffer");
  return run_pipeline(h, d_depth, true, batch, fx, fy, cx, cy, static_cast<int32_t*>(d_labels), static_cast<hipStream_t>(hip_stream));
}

int dpx_process_pcd_batch_device(dpx_handle h, const void* d_xyz, int32_t batch, void* d_labels, void* hip_stream) {
  if (!h) return fail(DPX_ERR_ARG, "null handle");
  if (batch > 0 && (!d_xyz || !d_labels)) return fail(DPX_ERR_ARG, "null buffer");
  return run_pipeline(h, d_xyz, false, batch, 1.f, 1.f, 0.f, 0.f, static_cast<int32_t*>(d_labels), static_cast<hipStream_t>(hip_stream));
}

int dpx_backproject(const uint16_t* depth, int32_t height, int32_t width, float fx, float fy, float cx, float cy,
                    float* xyz, int32_t device) {
  if (height < 0 || width < 0) return fail(DPX_ERR_ARG, "negative image size");
  const size_t N = static_cast<size_t>(height) * width;
  if (N == 0) return DPX_OK;
  if (!depth || !xyz) return fail(DPX_ERR_ARG, "null buffer");
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) return fail(DPX_ERR_NO_DEVICE, "no HIP device available (this library has no CPU path)");
  DPX_HIP(hipSetDevice(device));
  uint16_t* d_d = nullptr;
  float* d_x = nullptr;
  DPX_HIP(hipMalloc(reinterpret_cast<void**>(&d_d), N * 2));
  if (hipMalloc(reinterpret_cast<void**>(&d_x), N * 12) != hipSuccess) {
    hipFree(d_d);
    return fail(DPX_ERR_HIP, "hipMalloc failed");
  }
  hipError_t e = hipMemcpy(d_d, depth, N * 2, hipMemcpyHostToDevice);
  if (e == hipSuccess) {
    hipLaunchKernelGGL(dpx::k_backproject, dim3(static_cast<unsigned>((N + 255) / 256)), dim3(256), 0, 0, d_d, height, width, fx, fy, cx, cy, d_x);
    e = hipMemcpy(xyz, d_x, N * 12, hipMemcpyDeviceToHost);
  }
  hipFree(d_d);
  hipFree(d_x);
  if (e != hipSuccess) return fail(DPX_ERR_HIP, hipGetErrorString(e));
  return DPX_OK;
}

int dpx_read_depth_png(const char* path, int32_t* height, int32_t* width, uint16_t** data) {
  if (!path || !height || !width || !data) return fail(DPX_ERR_ARG, "null argument");
  *data = nullptr;
  dpxpng::Image img;
  std::string err;
  try {
    err = dpxpng::read_grey16(path, img);
  } catch (const std::exception& e) {  // (allocation failure on a hostile file: no exception crosses the C boundary)
    err = e.what();
  }
  if (!err.empty()) return fail(DPX_ERR_IO, std::string("Error: Couldn't read image ") + path);  // depth_image.cpp:33
  const size_t n = static_cast<size_t>(img.width) * img.height;
  uint16_t* buf = static_cast<uint16_t*>(std::malloc(std::max<size_t>(n, 1) * sizeof(uint16_t)));
  if (!buf) return fail(DPX_ERR_IO, "out of memory");
  std::memcpy(buf, img.grey.data(), n * sizeof(uint16_t));
  *height = img.height;
  *width = img.width;
  *data = buf;
  return DPX_OK;
}

void dpx_free(void* p) { std::free(p); }

int dpx_grid_dims(dpx_handle h, int32_t* nh, int32_t* nv) {
  if (!h) return fail(DPX_ERR_ARG, "null handle");
  if (nh) *nh = h->prm.nh;
  if (nv) *nv = h->prm.nv;
  return DPX_OK;
}

#define DPX_CHECK_FRAME()                                                            \
  if (!h || !out) return fail(DPX_ERR_ARG, "null argument");                         \
  if (frame < 0 || frame >= h->last_batch) return fail(DPX_ERR_ARG, "frame index out of range of the last call"); \
  {                                                                                  \
    int rc_ = sync_last(h);                                                          \
    if (rc_) return rc_;                                                             \
  }

int dpx_get_cells(dpx_handle h, int32_t frame, dpx_cell* out) {
  DPX_CHECK_FRAME();
  const size_t C = h->prm.C;
  if (C) DPX_HIP(hipMemcpy(out, h->d_cells + frame * C, C * sizeof(dpx_cell), hipMemcpyDeviceToHost));
  return DPX_OK;
}

int dpx_get_cell_edges(dpx_handle h, int32_t frame, uint8_t* out) {
  DPX_CHECK_FRAME();
  const size_t C = h->prm.C;
  std::vector<uint32_t> tmp(C);
  if (C && h->regions_tier == 2) {  // the fused path never materialises the cell graph: compute it on demand
    dpx::Params prm = h->prm;
    prm.batch = h->last_batch;
    const long long total = static_cast<long long>(prm.batch) * prm.C;
    hipLaunchKernelGGL(dpx::k_cellgraph, dim3(static_cast<unsigned>((total + 2047) / 2048 * 8)), dim3(256), 0, h->last_stream, prm, h->d_cells, h->d_meta, h->d_mse_c);
    DPX_HIP(hipStreamSynchronize(h->last_stream));
  }
  if (C) DPX_HIP(hipMemcpy(tmp.data(), h->d_meta + frame * C, C * 4, hipMemcpyDeviceToHost));
  for (size_t i = 0; i < C; ++i) out[i] = static_cast<uint8_t>(tmp[i] & 0x3F);
  return DPX_OK;
}

int dpx_get_cell_labels(dpx_handle h, int32_t frame, int32_t* out) {
  DPX_CHECK_FRAME();
  const size_t C = h->prm.C;
  if (C) DPX_HIP(hipMemcpy(out, h->d_cell_labels + frame * C, C * 4, hipMemcpyDeviceToHost));
  return DPX_OK;
}

int dpx_get_frame_stats(dpx_handle h, int32_t frame, dpx_frame_stats* out) {
  DPX_CHECK_FRAME();
  DPX_HIP(hipMemcpy(out, h->d_stats + frame, sizeof(dpx_frame_stats), hipMemcpyDeviceToHost));
  return DPX_OK;
}

int dpx_get_planes(dpx_handle h, int32_t frame, dpx_plane* out, int32_t cap, int32_t* n_out) {
  if (!n_out) return fail(DPX_ERR_ARG, "null argument");
  dpx_frame_stats st;
  int rc = dpx_get_frame_stats(h, frame, &st);
  if (rc) return rc;
  *n_out = st.segments;
  const int n = std::min(st.segments, cap);
  if (n > 0 && out)
    DPX_HIP(hipMemcpy(out, h->d_planes + static_cast<size_t>(frame) * h->pcap, n * sizeof(dpx_plane), hipMemcpyDeviceToHost));
  return DPX_OK;
}

int dpx_prepare_intrinsics(dpx_handle h, float fx, float fy, float cx, float cy, int32_t* fast_division) {
  if (!h) return fail(DPX_ERR_ARG, "null handle");
  DPX_HIP(hipSetDevice(h->device));
  dpx_handle_s::FdEntry e{};
  const int rc = prepare_intrinsics(h, fx, fy, cx, cy, &e);
  if (rc == DPX_OK && fast_division) *fast_division = e.ok ? 1 : 0;
  return rc;
}

int dpx_get_options(dpx_handle h, char* buf, int32_t cap) {
  if (!h || !buf || cap < 1) return fail(DPX_ERR_ARG, "null argument");
  std::string o = "regions=";
  o += h->cores ? "cores" : (h->regions_tier == 2 ? "tier2" : (h->regions_tier == 1 ? "tier1" : (h->regions_tier == 0 ? (h->tier0_split ? "tier0-split" : "tier0") : "v1")));
  o += std::string(";split=") + (h->regions_split ? "1" : "0") + ";stream=" + (h->stream_mode ? "1" : "0");
  o += std::string(";prep_grow_fused=") + (h->opt_prep_grow_fused ? "1" : "0") + ";fused_labels=" + (h->opt_nofuse ? "0" : "1");
  o += std::string(";labels_v1=") + (h->opt_labels_v1 ? "1" : "0") + ";cloud_lds=" + (h->opt_cloud_nolds ? "0" : "1") + ";prep_v1=" + (h->opt_prep_v1 ? "1" : "0");
  o += std::string(";sort=") + (h->opt_sort_bitonic ? "bitonic" : "counting");
  o += ";csr_chunk=" + std::to_string(h->opt_csr_chunk) + ";cells=" + std::to_string(h->prm.C) + ";pcap=" + std::to_string(h->pcap);
  const int n = static_cast<int>(std::min<size_t>(o.size(), static_cast<size_t>(cap) - 1));
  std::memcpy(buf, o.data(), static_cast<size_t>(n));
  buf[n] = 0;
  return n;
}

int64_t dpx_debug_read_cores(dpx_handle h, int32_t which, int32_t frame, void* out, int64_t cap_bytes) {
  if (!h || !out) return fail(DPX_ERR_ARG, "null argument");
  if (!h->cores || frame < 0 || frame >= h->max_batch) return fail(DPX_ERR_ARG, "not the core path, or bad frame");
  DPX_HIP(hipSetDevice(h->device));
  DPX_HIP(hipStreamSynchronize(h->last_stream));
  const size_t C = static_cast<size_t>(h->prm.C), f = static_cast<size_t>(frame), nb = static_cast<size_t>(h->prm.hist_B) * h->prm.hist_B;
  const void* src = nullptr;
  size_t bytes = 0;
  switch (which) {
    case 0: src = h->d_parent + f * C; bytes = C * 4; break;
    case 1: src = h->d_csize + f * C; bytes = C * 4; break;
    case 2: src = h->d_npairs + f * C; bytes = C * 4; break;
    case 3: src = h->d_pstart + f * C; bytes = C * 4; break;
    case 4: src = h->d_corepairs + f * C; bytes = C * 4; break;
    case 5: src = h->d_lists + f * C * 2; bytes = C * 32; break;
    case 6: src = h->d_bin_off + f * (nb + 1); bytes = (nb + 1) * 4; break;
    case 7: src = h->d_histg + f * nb; bytes = nb * 4; break;
    case 8: src = h->d_ccursor + f * dpx::kCtr; bytes = 16; break;
    case 9: src = reinterpret_cast<const int*>(h->d_lds_img) + f * 48; bytes = 192; break;
    case 10: src = h->d_cand_grow + f * static_cast<size_t>(h->pcap) * 4; bytes = static_cast<size_t>(h->pcap) * 16; break;
    case 11: src = h->d_gqueue + f * C; bytes = C * 4; break;
    case 12: src = h->d_cellq + f * C; bytes = C * 4; break;
    case 13: src = h->d_meta + f * C; bytes = C * 4; break;
    default: return fail(DPX_ERR_ARG, "unknown array");
  }
  if (static_cast<int64_t>(bytes) > cap_bytes) return fail(DPX_ERR_CAPACITY, "buffer too small");
  DPX_HIP(hipMemcpy(out, src, bytes, hipMemcpyDeviceToHost));
  return static_cast<int64_t>(bytes);
}

int dpx_debug_cores_phases(dpx_handle h, int32_t frame, uint64_t* out32) {
  if (!h || !out32) return fail(DPX_ERR_ARG, "null argument");
#ifndef DPX_DIAG
  return fail(DPX_ERR_ARG, "this build has no phase stamps: use the -DDPX_DIAG variant (deplex_amd.build.build_variant('diag'), DPX_LIB_PATH)");
#endif
  if (!h->d_dbg || !h->cores || frame < 0 || frame >= h->max_batch) return fail(DPX_ERR_ARG, "diagnostics not enabled, not the core path, or bad frame");
  DPX_HIP(hipSetDevice(h->device));
  DPX_HIP(hipStreamSynchronize(h->last_stream));
  DPX_HIP(hipMemcpy(out32, h->d_dbg + (static_cast<size_t>(h->max_batch) + frame) * dpx::kDbgSlots, dpx::kDbgSlots * sizeof(unsigned long long), hipMemcpyDeviceToHost));
  return DPX_OK;
}

int dpx_debug_region_phases(dpx_handle h, int32_t enable, int32_t frame, uint64_t* out32) {
  if (!h) return fail(DPX_ERR_ARG, "null handle");
#ifndef DPX_DIAG
  if (enable || out32) return fail(DPX_ERR_ARG, "this build has no phase stamps: use the -DDPX_DIAG variant (deplex_amd.build.build_variant('diag'), DPX_LIB_PATH)");
  return DPX_OK;
#else
  DPX_HIP(hipSetDevice(h->device));
  if (enable && !h->d_dbg) {
    // (two blocks of kDbgSlots words per frame: the k_regions2 stamps, then those of the core path's sequence / order launches)
    DPX_HIP(hipMalloc(reinterpret_cast<void**>(&h->d_dbg), static_cast<size_t>(h->max_batch) * 2 * dpx::kDbgSlots * sizeof(unsigned long long)));
    DPX_HIP(hipMemset(h->d_dbg, 0, static_cast<size_t>(h->max_batch) * 2 * dpx::kDbgSlots * sizeof(unsigned long long)));
    if (h->cores) {
      if (h->lds2_p3 > 48 * 1024)
        DPX_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(dpx::k_regions2<0, true, 5>), hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(h->lds2_p3)));
    } else if (h->lds2_bytes > 48 * 1024 && h->regions_tier >= 0) {
      const void* fn = h->regions_tier == 2   ? reinterpret_cast<const void*>(dpx::k_regions2<2, true, 0>)
                       : h->regions_tier == 1 ? reinterpret_cast<const void*>(dpx::k_regions2<1, true, 0>)
                                              : reinterpret_cast<const void*>(dpx::k_regions2<0, true, 0>);
      DPX_HIP(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(h->lds2_bytes)));
    }
  }
  if (out32) {
    if (!h->d_dbg || frame < 0 || frame >= h->max_batch) return fail(DPX_ERR_ARG, "diagnostics not enabled or bad frame");
    DPX_HIP(hipStreamSynchronize(h->last_stream));
    DPX_HIP(hipMemcpy(out32, h->d_dbg + static_cast<size_t>(frame) * dpx::kDbgSlots, dpx::kDbgSlots * sizeof(unsigned long long), hipMemcpyDeviceToHost));
  }
  if (!enable && h->d_dbg) {
    hipFree(h->d_dbg);
    h->d_dbg = nullptr;
  }
  return DPX_OK;
#endif
}

int dpx_debug_trace(int32_t device, uint64_t* d_buf, uint32_t cap_per_slot, uint32_t* counts) {
#ifndef DPX_TRACE
  if (d_buf) return fail(DPX_ERR_ARG, "this build has no wave trace: use the -DDPX_TRACE variant (deplex_amd.build.build_variant('trace'))");
#endif
  DPX_HIP(hipSetDevice(device));
  DPX_HIP(hipDeviceSynchronize());
  if (counts) DPX_HIP(hipMemcpyFromSymbol(counts, HIP_SYMBOL(dpx::g_trace_n), dpx::kTraceSlots * sizeof(unsigned int)));
  unsigned long long* buf = reinterpret_cast<unsigned long long*>(d_buf);
  static const std::vector<unsigned int> zeros(dpx::kTraceSlots, 0u);
  DPX_HIP(hipMemcpyToSymbol(HIP_SYMBOL(dpx::g_trace_n), zeros.data(), dpx::kTraceSlots * sizeof(unsigned int)));
  DPX_HIP(hipMemcpyToSymbol(HIP_SYMBOL(dpx::g_trace_cap), &cap_per_slot, sizeof(cap_per_slot)));
  DPX_HIP(hipMemcpyToSymbol(HIP_SYMBOL(dpx::g_trace), &buf, sizeof(buf)));
  return DPX_OK;
}

int dpx_debug_residency(dpx_handle h, double* out3) {  // out3[3]: the grow launch's lifetime in 100 MHz ticks
  if (!h || !out3) return fail(DPX_ERR_ARG, "null argument");
  if (!h->d_res) return fail(DPX_ERR_ARG, "set DPX_RESIDENCY=1 before dpx_create");
  DPX_HIP(hipSetDevice(h->device));
  DPX_HIP(hipStreamSynchronize(h->last_stream));
  std::vector<unsigned long long> v(static_cast<size_t>(h->max_batch) * 4);
  DPX_HIP(hipMemcpy(v.data(), h->d_res, v.size() * 8, hipMemcpyDeviceToHost));
  for (int k = 0; k < 4; ++k) {
    double s = 0;
    int n = 0;
    for (int f = 0; f < h->last_batch; ++f)
      if (v[static_cast<size_t>(f) * 4 + k]) { s += static_cast<double>(v[static_cast<size_t>(f) * 4 + k]); ++n; }
    out3[k] = n ? s / n : 0.0;
  }
  return DPX_OK;
}

// ---- streamed pipeline: `slots` groups of `group` batches in flight, one stream and one scratch set per slot ------
struct dpx_pipeline_s {
  std::vector<dpx_handle> slots;
  std::vector<std::vector<SubBatch>> pending;  // per slot: submissions recorded, not launched yet
  std::vector<char> pending_depth;             // per slot: input form of the pending submissions (1 depth, 0 cloud)
  std::vector<char> pending_cells;             // per slot: output form (1 = cell labels into the submissions' buffers, 0 = pixel label maps)
  // host-buffer submissions (dpx_pipeline_submit_depth*_host): per slot a device staging buffer for the group's depth
  // frames and one for its pixel label maps (allocated at the first such submission), and per pending submission the host
  // address its result goes to (nullptr: a device-buffer submission)
  std::vector<void*> stage_in, stage_out;
  std::vector<std::vector<void*>> pending_host;
  int group = 1, batch = 0;
  long long submitted = 0;
  int device = 0;
};

namespace {
int pipeline_flush(dpx_pipeline p, size_t k) {
  if (p->pending[k].empty()) return DPX_OK;
  dpx_handle h = p->slots[k];
  const int nsub = static_cast<int>(p->pending[k].size());
  int rc = stage_fit(h, p->pending[k].data(), nsub, p->batch, p->pending_depth[k] != 0, h->stream);
  const bool cells = p->pending_cells[k] != 0;
  if (cells) h->no_pixel_labels = true;  // cell-level submissions: no label expansion launch, the merged cell labels are copied out
  if (rc == DPX_OK) rc = stage_rest(h, p->pending[k].data(), nsub, p->batch, p->pending_depth[k] != 0, h->stream);
  h->no_pixel_labels = false;
  // results that do not stay where the kernels wrote them: cell labels leave the slot's buffer (device to device, or
  // straight to the host for a host-buffer submission), pixel maps of host-buffer submissions leave the staging buffer
  const size_t per = static_cast<size_t>(p->batch) * h->prm.C, per_px = static_cast<size_t>(p->batch) * h->prm.N;
  for (int j = 0; j < nsub && rc == DPX_OK; ++j) {
    void* hdst = p->pending_host[k][j];
    hipError_t e = hipSuccess;
    if (cells)
      e = hipMemcpyAsync(hdst ? hdst : static_cast<void*>(p->pending[k][j].d_labels), h->d_cell_labels + static_cast<size_t>(j) * per,
                         per * sizeof(int32_t), hdst ? hipMemcpyDeviceToHost : hipMemcpyDeviceToDevice, h->stream);
    else if (hdst)
      e = hipMemcpyAsync(hdst, p->pending[k][j].d_labels, per_px * sizeof(int32_t), hipMemcpyDeviceToHost, h->stream);
    if (e != hipSuccess) rc = fail(DPX_ERR_HIP, "hipMemcpyAsync (results) failed");
  }
  p->pending[k].clear();
  p->pending_host[k].clear();
  return rc;
}
}  // namespace

int dpx_pipeline_create(const dpx_config* cfg, int32_t image_height, int32_t image_width, int32_t batch, int32_t device,
                        int32_t slots, int32_t group, dpx_pipeline* out) {
  if (!out || slots < 1 || batch < 1) return fail(DPX_ERR_ARG, "bad pipeline arguments");
  *out = nullptr;
  if (group < 1) group = 1;
  if (group > dpx::kMaxGroup) group = dpx::kMaxGroup;
  {
    // HIP maps streams to GPU_MAX_HW_QUEUES hardware queues (default 4), read when the runtime initialises: slots beyond
    // that share a queue and a launch waiting there holds up its neighbour.  Said once.
    static bool warned = false;
    const char* q = std::getenv("GPU_MAX_HW_QUEUES");
    const int queues = q ? std::atoi(q) : 4;
    if (!warned && slots > queues) {
      warned = true;
      std::fprintf(stderr, "[deplex_amd] a pipeline of %d slots with GPU_MAX_HW_QUEUES=%d: set GPU_MAX_HW_QUEUES >= slots (per device) "
                           "before HIP initialises, or the slots' streams share hardware queues\n", slots, queues);
    }
  }
  dpx_pipeline p = new dpx_pipeline_s();
  p->device = device;
  p->group = group;
  p->batch = batch;
  for (int i = 0; i < slots; ++i) {
    dpx_handle h = nullptr;
    const int rc = create_handle(cfg, image_height, image_width, batch * group, device, /*stream_mode=*/true, &h);
    if (rc != DPX_OK) {
      const std::string keep = g_err;
      dpx_pipeline_destroy(p);
      g_err = keep;
      return rc;
    }
    p->slots.push_back(h);
  }
  p->pending.resize(p->slots.size());
  p->pending_depth.assign(p->slots.size(), 1);
  p->pending_cells.assign(p->slots.size(), 0);
  p->stage_in.assign(p->slots.size(), nullptr);
  p->stage_out.assign(p->slots.size(), nullptr);
  p->pending_host.resize(p->slots.size());
  // HIP binds a stream to a hardware queue at its first use: do that here, once, for every slot, instead of inside the
  // first groups a caller submits
  for (auto h : p->slots) (void)hipMemsetAsync(h->d_stats, 0, sizeof(dpx_frame_stats), h->stream);
  for (auto h : p->slots) (void)hipStreamSynchronize(h->stream);
  *out = p;
  return DPX_OK;
}

void dpx_pipeline_destroy(dpx_pipeline p) {
  if (!p) return;
  (void)hipSetDevice(p->device);
  (void)hipDeviceSynchronize();
  for (auto h : p->slots) dpx_destroy(h);
  for (void* b : p->stage_in) (void)hipFree(b);
  for (void* b : p->stage_out) (void)hipFree(b);
  delete p;
}

namespace {
// host = true: d_in / d_labels are HOST addresses (page-locked for the copies to be asynchronous); the frames are copied
// into the slot's staging buffer on the slot's stream right here, the results back after the group's kernels
int pipeline_submit(dpx_pipeline p, const void* d_in, bool depth, int32_t batch, float fx, float fy, float cx, float cy, void* d_labels,
                    bool cells = false, bool host = false) {
  if (!p || !d_in || !d_labels) return fail(DPX_ERR_ARG, "bad argument");
  if (cells && !p->slots.empty() && p->slots[0]->cfg.ransac_refinement)
    return fail(DPX_ERR_ARG, "cell-level output is not defined with ransac_refinement (it edits single pixels)");
  if (batch != p->batch) return fail(DPX_ERR_ARG, "pipeline submissions must have the batch size given at creation");
  long long n = p->submitted;
  size_t k = static_cast<size_t>((n / p->group) % static_cast<long long>(p->slots.size()));
  if (!p->pending[k].empty() && ((p->pending_depth[k] != 0) != depth || (p->pending_cells[k] != 0) != cells)) {
    // a group holds one input form (the refinement re-reads the input) and one output form: close the open group and
    // start a new one
    const int rcf = pipeline_flush(p, k);
    if (rcf) return rcf;
    p->submitted = ((p->submitted + p->group - 1) / p->group) * p->group;
    n = p->submitted;
    k = static_cast<size_t>((n / p->group) % static_cast<long long>(p->slots.size()));
  }
  const int j = static_cast<int>(n % p->group);
  dpx_handle h = p->slots[k];
  // the submission is only recorded here: the plane fit of the whole group is ONE launch, issued with the region
  // stage when the group is full (or at dpx_pipeline_sync)
  void* hdst = nullptr;
  if (host) {
    DPX_HIP(hipSetDevice(p->device));
    const size_t in_bytes = static_cast<size_t>(p->batch) * h->prm.N * sizeof(uint16_t), out_bytes = static_cast<size_t>(p->batch) * h->prm.N * sizeof(int32_t);
    if (!p->stage_in[k]) DPX_HIP(hipMalloc(&p->stage_in[k], in_bytes * static_cast<size_t>(p->group)));
    if (!cells && !p->stage_out[k]) DPX_HIP(hipMalloc(&p->stage_out[k], out_bytes * static_cast<size_t>(p->group)));
    void* din = static_cast<char*>(p->stage_in[k]) + in_bytes * static_cast<size_t>(j);
    DPX_HIP(hipMemcpyAsync(din, d_in, in_bytes, hipMemcpyHostToDevice, h->stream));
    hdst = d_labels;
    d_in = din;
    // (cell-level: the result is copied from the slot's cell labels; the SubBatch's own address is not written)
    d_labels = cells ? static_cast<void*>(h->d_cell_labels) : static_cast<void*>(static_cast<char*>(p->stage_out[k]) + out_bytes * static_cast<size_t>(j));
  }
  SubBatch sb;
  int rc = fill_sub(h, sb, d_in, static_cast<int32_t*>(d_labels), depth, fx, fy, cx, cy);
  if (rc) return rc;
  ++p->submitted;
  p->pending_depth[k] = depth ? 1 : 0;
  p->pending_cells[k] = cells ? 1 : 0;
  p->pending[k].push_back(sb);
  p->pending_host[k].push_back(hdst);
  if (j == p->group - 1) rc = pipeline_flush(p, k);
  return rc;
}
}  // namespace

int dpx_pipeline_submit_depth(dpx_pipeline p, const void* d_depth, int32_t batch, float fx, float fy, float cx, float cy,
                              void* d_labels) {
  return pipeline_submit(p, d_depth, true, batch, fx, fy, cx, cy, d_labels);
}

int dpx_pipeline_submit_depth_cells(dpx_pipeline p, const void* d_depth, int32_t batch, float fx, float fy, float cx, float cy,
                                    void* d_cell_labels) {
  return pipeline_submit(p, d_depth, true, batch, fx, fy, cx, cy, d_cell_labels, /*cells=*/true);
}

int dpx_pipeline_submit_depth_host(dpx_pipeline p, const uint16_t* h_depth, int32_t batch, float fx, float fy, float cx, float cy,
                                   int32_t* h_labels) {
  return pipeline_submit(p, h_depth, true, batch, fx, fy, cx, cy, h_labels, /*cells=*/false, /*host=*/true);
}

int dpx_pipeline_submit_depth_cells_host(dpx_pipeline p, const uint16_t* h_depth, int32_t batch, float fx, float fy, float cx, float cy,
                                         int32_t* h_cell_labels) {
  return pipeline_submit(p, h_depth, true, batch, fx, fy, cx, cy, h_cell_labels, /*cells=*/true, /*host=*/true);
}

int dpx_pipeline_submit_depth_many(dpx_pipeline p, int32_t n, const void* const* d_depth, int32_t batch, float fx, float fy, float cx,
                                   float cy, void* const* d_labels) {
  if (n < 0 || (n > 0 && (!d_depth || !d_labels))) return fail(DPX_ERR_ARG, "bad submission list");
  for (int32_t i = 0; i < n; ++i) {
    const int rc = pipeline_submit(p, d_depth[i], true, batch, fx, fy, cx, cy, d_labels[i]);
    if (rc) return rc;
  }
  return DPX_OK;
}

int dpx_pipeline_submit_pcd(dpx_pipeline p, const void* d_xyz, int32_t batch, void* d_labels) {
  return pipeline_submit(p, d_xyz, false, batch, 1.f, 1.f, 0.f, 0.f, d_labels);
}

int dpx_pipeline_sync(dpx_pipeline p) {
  if (!p) return fail(DPX_ERR_ARG, "null pipeline");
  DPX_HIP(hipSetDevice(p->device));
  for (size_t k = 0; k < p->slots.size(); ++k) {
    const int rc = pipeline_flush(p, k);
    if (rc) return rc;
  }
  // a partially filled group was flushed: the next submission starts a new group
  p->submitted = ((p->submitted + p->group - 1) / p->group) * p->group;
  for (auto h : p->slots) DPX_HIP(hipStreamSynchronize(h->stream));
  return DPX_OK;
}

int dpx_pipeline_flush(dpx_pipeline p) {
  if (!p) return fail(DPX_ERR_ARG, "null pipeline");
  DPX_HIP(hipSetDevice(p->device));
  for (size_t k = 0; k < p->slots.size(); ++k) {
    const int rc = pipeline_flush(p, k);
    if (rc) return rc;
  }
  p->submitted = ((p->submitted + p->group - 1) / p->group) * p->group;
  return DPX_OK;
}

dpx_handle dpx_pipeline_slot(dpx_pipeline p, int32_t slot) {
  if (!p || slot < 0 || slot >= static_cast<int>(p->slots.size())) return nullptr;
  return p->slots[slot];
}

int dpx_set_profiling(dpx_handle h, int32_t enable) {
  if (!h) return fail(DPX_ERR_ARG, "null handle");
  DPX_HIP(hipSetDevice(h->device));
  if (enable && !h->ev_created) {
    for (auto& slot : h->ev)
      // (no system-scope fence when the event is recorded: a fence writes back and invalidates the caches under every
      // kernel on the chip, and these events only time launches)
      for (auto& e : slot) DPX_HIP(hipEventCreateWithFlags(&e, hipEventDisableSystemFence));
    h->ev_created = true;
  }
  h->profiling = enable != 0;
  h->prof_calls = 0;
  return DPX_OK;
}

int dpx_get_kernel_ms(dpx_handle h, float* ms) {
  if (!h || !ms) return fail(DPX_ERR_ARG, "null argument");
  if (h->prof_calls == 0) return fail(DPX_ERR_ARG, "no profiled call recorded");
  DPX_HIP(hipSetDevice(h->device));
  const int n = static_cast<int>(std::min<long long>(h->prof_calls, dpx_handle_s::kProfSlots));
  double acc[DPX_K_COUNT] = {};
  static const int kStart[DPX_K_COUNT] = {0, 2, 3, 5}, kEnd[DPX_K_COUNT] = {1, 3, 4, 6};
  for (int i = 0; i < n; ++i) {
    DPX_HIP(hipEventSynchronize(h->ev[i][6]));
    for (int k = 0; k < DPX_K_COUNT; ++k) {
      float t = 0.f;
      DPX_HIP(hipEventElapsedTime(&t, h->ev[i][kStart[k]], h->ev[i][kEnd[k]]));
      acc[k] += t;
    }
  }
  for (int k = 0; k < DPX_K_COUNT; ++k) ms[k] = static_cast<float>(acc[k] / n);
  return DPX_OK;
}

int dpx_time_planefit_group(dpx_handle h, const void* const* d_in, int32_t nsub, int32_t input_kind, int32_t per_sub, float fx,
                            float fy, float cx, float cy, int32_t iters, void* hip_stream, float* avg_ms) {
  if (!h || !d_in || !avg_ms || iters < 1 || nsub < 1 || nsub > dpx::kMaxGroup) return fail(DPX_ERR_ARG, "bad argument");
  if (per_sub < 1 || nsub * per_sub > h->max_batch) return fail(DPX_ERR_ARG, "batch out of range");
  DPX_HIP(hipSetDevice(h->device));
  hipStream_t s = static_cast<hipStream_t>(hip_stream);
  SubBatch subs[dpx::kMaxGroup];
  for (int j = 0; j < nsub; ++j) {
    if (!d_in[j]) return fail(DPX_ERR_ARG, "null input buffer");
    // (the one-time verification of the reciprocal division synchronises: it happens here, before the timing)
    const int rc = fill_sub(h, subs[j], d_in[j], nullptr, input_kind == 0, fx, fy, cx, cy);
    if (rc) return rc;
  }
  const bool was_prof = h->profiling;
  h->profiling = false;
  hipEvent_t a = nullptr, b = nullptr;
  float ms = 0.f;
  int rc = DPX_OK;
  hipError_t e = hipEventCreate(&a);
  if (e == hipSuccess) e = hipEventCreate(&b);
  if (e == hipSuccess) e = hipEventRecord(a, s);
  if (e == hipSuccess) {
    for (int i = 0; i < iters && rc == DPX_OK; ++i) rc = stage_fit(h, subs, nsub, per_sub, input_kind == 0, s, false);
    e = hipEventRecord(b, s);
  }
  if (e == hipSuccess) e = hipEventSynchronize(b);
  if (e == hipSuccess) e = hipEventElapsedTime(&ms, a, b);
  if (a) hipEventDestroy(a);
  if (b) hipEventDestroy(b);
  h->profiling = was_prof;
  if (rc) return rc;
  if (e != hipSuccess) return fail(DPX_ERR_HIP, std::string("dpx_time_planefit: ") + hipGetErrorString(e));
  *avg_ms = ms / iters;
  return DPX_OK;
}

int dpx_time_planefit(dpx_handle h, const void* d_in, int32_t input_kind, int32_t batch, float fx, float fy, float cx,
                      float cy, int32_t iters, void* hip_stream, float* avg_ms) {
  return dpx_time_planefit_group(h, &d_in, 1, input_kind, batch, fx, fy, cx, cy, iters, hip_stream, avg_ms);
}

int64_t dpx_planefit_bytes(dpx_handle h, int32_t input_kind, int32_t batch) {
  if (!h) return 0;
  const int64_t N = static_cast<int64_t>(h->H) * h->W;
  return static_cast<int64_t>(batch) * (N * (input_kind == 0 ? 2 : 12) + static_cast<int64_t>(h->prm.C) * sizeof(dpx_cell));
}

int dpx_test_math(int32_t which, const double* a, const double* b, double* out, int64_t n, int32_t device) {
  if (n <= 0) return DPX_OK;
  DPX_HIP(hipSetDevice(device));
  double *da = nullptr, *db = nullptr, *dout = nullptr;
  DPX_HIP(hipMalloc(reinterpret_cast<void**>(&da), n * 8));
  DPX_HIP(hipMalloc(reinterpret_cast<void**>(&db), n * 8));
  DPX_HIP(hipMalloc(reinterpret_cast<void**>(&dout), n * 8));
  DPX_HIP(hipMemcpy(da, a, n * 8, hipMemcpyHostToDevice));
  DPX_HIP(hipMemcpy(db, b ? b : a, n * 8, hipMemcpyHostToDevice));
  hipLaunchKernelGGL(k_test_math, dim3(static_cast<unsigned>((n + 255) / 256)), dim3(256), 0, 0, which, da, db, dout, static_cast<long long>(n));
  DPX_HIP(hipMemcpy(out, dout, n * 8, hipMemcpyDeviceToHost));
  hipFree(da);
  hipFree(db);
  hipFree(dout);
  return DPX_OK;
}

int dpx_test_eig3(const double* A, double* Q, double* w, int32_t* rc, int64_t n, int32_t device) {
  if (n <= 0) return DPX_OK;
  DPX_HIP(hipSetDevice(device));
  double *dA = nullptr, *dQ = nullptr, *dw = nullptr;
  int* drc = nullptr;
  DPX_HIP(hipMalloc(reinterpret_cast<void**>(&dA), n * 48));
  DPX_HIP(hipMalloc(reinterpret_cast<void**>(&dQ), n * 72));
  DPX_HIP(hipMalloc(reinterpret_cast<void**>(&dw), n * 24));
  DPX_HIP(hipMalloc(reinterpret_cast<void**>(&drc), n * 4));
  DPX_HIP(hipMemcpy(dA, A, n * 48, hipMemcpyHostToDevice));
  hipLaunchKernelGGL(k_test_eig3, dim3(static_cast<unsigned>((n + 255) / 256)), dim3(256), 0, 0, dA, dQ, dw, drc, static_cast<long long>(n));
  DPX_HIP(hipMemcpy(Q, dQ, n * 72, hipMemcpyDeviceToHost));
  DPX_HIP(hipMemcpy(w, dw, n * 24, hipMemcpyDeviceToHost));
  DPX_HIP(hipMemcpy(rc, drc, n * 4, hipMemcpyDeviceToHost));
  hipFree(dA);
  hipFree(dQ);
  hipFree(dw);
  hipFree(drc);
  return DPX_OK;
}

}  // extern "C"
