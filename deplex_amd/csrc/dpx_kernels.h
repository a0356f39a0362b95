// dpx_kernels.h — HIP kernels of the depth-plane extractor (gfx950 / CDNA4, wave64).
//
//   k_planefit   K1+K2  depth (or organised cloud) -> per-cell moments, validity, 3x3 eigen plane fit,
//                       planarity, merge tolerance, histogram bin.  One 64-byte dpx_cell per cell.
//   k_cellgraph  K3     per-cell directed 4-neighbour edge bits of the region-growing predicate.
//   k_regions    K4+K5  (dpx_regions.h) histogram seeding, FIFO region growing, segment fits, merging.
//   k_labels     K6     cell labels -> per-pixel int32 label map.
//
// Reference files restated: utils/depth_image.cpp:55-78, cell_grid.cpp:21-44,69-83, cell_segment.cpp:21-110,
// cell_segment_stat.cpp:27-81, normals_histogram.cpp:21-49, plane_extractor.cpp:349-392,455-470.
// Float semantics: every f32 product and sum is rounded separately (compile with -ffp-contract=off); the
// summation orders are those of the reference's canonical build.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/deplex_amd.h"
#include "dpx_fit.h"

namespace dpx {

// Most submissions a dpx_pipeline group may hold (sizes the per-submission tables passed to the group launches as kernel
// arguments; include/deplex_amd.h documents the same limit).
constexpr int kMaxGroup = 32;

// operand types of the gfx950 LDS-DMA loads (__builtin_amdgcn_global_load_lds)
typedef __attribute__((address_space(3))) void* dpx_lds_ptr;
typedef const __attribute__((address_space(1))) void* dpx_glb_ptr;

// ---- diagnostic wave trace (dpx_debug_trace): every wave of the pipeline's kernels logs where and when it ran --------
// Entry = 4 x u64: start and end in 100 MHz real-time ticks (s_memrealtime: one clock for the whole chip), the hardware
// id (XCC << 32 | HW_ID: SE, CU, SIMD, wave slot), and kind | wave-in-block << 4 | blockIdx.x << 8.  A hardware wave slot
// holds one wave at a time, so each slot owns a ring of entries and a plain counter — no atomics; the counter is
// loaded when the wave starts and first needed when it ends, i.e. its latency hides behind the kernel.  Compiled in
// only with -DDPX_TRACE (deplex_amd/build.py build_variant("trace"), used by tools/wave_trace.py): the live values cost
// the plane-fit kernel registers it does not have.
constexpr unsigned int kTraceSlots = 1u << 16;  // XCC (3) | SE, SH, CU bits of HW_ID (7) | SIMD (2) | wave slot (4)
__device__ unsigned long long* g_trace = nullptr;
__device__ unsigned int g_trace_cap = 0;  // entries per slot
__device__ unsigned int g_trace_n[kTraceSlots] = {};
enum TraceKind { TK_PLANEFIT = 1, TK_PREP = 2, TK_GROW = 3, TK_TAIL = 4, TK_LABELS = 5 };
#ifdef DPX_TRACE
struct WaveTrace {
  unsigned long long* base;
  unsigned long long t0, id, what;
  unsigned int slot, n;
  __device__ __forceinline__ explicit WaveTrace(int kind) : base(nullptr) {
    unsigned long long* t = g_trace;
    if (t && (threadIdx.x & 63) == 0) {
      const unsigned int xcc = __builtin_amdgcn_s_getreg((31 << 11) | 20), hw = __builtin_amdgcn_s_getreg((31 << 11) | 4);
      slot = ((xcc & 7u) << 13) | (((hw >> 8) & 0x7Fu) << 6) | (((hw >> 4) & 3u) << 4) | (hw & 15u);
      n = g_trace_n[slot];
      base = t;
      t0 = __builtin_amdgcn_s_memrealtime();
      id = (static_cast<unsigned long long>(xcc) << 32) | hw;
      what = static_cast<unsigned long long>(kind) | (static_cast<unsigned long long>(threadIdx.x >> 6) << 4) |
             (static_cast<unsigned long long>(blockIdx.x) << 8);
    }
  }
  __device__ __forceinline__ ~WaveTrace() {
    if (base) {
      const unsigned int cap = g_trace_cap;
      if (n < cap) {
        unsigned long long* e = base + 4ull * (static_cast<unsigned long long>(slot) * cap + n);
        e[0] = t0;
        e[1] = __builtin_amdgcn_s_memrealtime();
        e[2] = id;
        e[3] = what;
        g_trace_n[slot] = n + 1;
      }
    }
  }
};
#else  // product build: the trace costs registers that the plane-fit kernel does not have to spare
struct WaveTrace {
  __device__ __forceinline__ explicit WaveTrace(int) {}
};
#endif

struct Params {
  int H, W, P, nh, nv, C, N;  // image, patch, grid, cells per frame, pixels per frame
  int n;                      // points per cell = P*P
  int stride_w;               // nh*P: the row stride CellGrid::cellContinuousOrganize uses (cell_grid.cpp:71)
  int sheared;                // stride_w != W
  int batch;
  float fx, fy, cx, cy;
  float min_cos, max_merge_dist, min_score, sigma_coeff, sigma_margin, disc_thr, tol_sin;
  int hist_B, min_cand, min_cells, max_disc;
  long long valid_thr;  // (3n)/min_pts_per_cell as the reference's size_t comparison sees it; <0 => never valid
};

constexpr uint32_t CELL_VALID = 1u, CELL_PLANAR = 2u, CELL_QL = 4u;
// cell meta byte (k_cellgraph output / region-growing flags)
constexpr uint32_t F_UNASSIGNED = 1u;          // planar and not yet grown
constexpr uint32_t F_EDGE0 = 4u;               // bits 2..5: edge to U, D, L, R neighbour passes the predicate
constexpr uint32_t F_INEDGE0 = 256u;           // bits 8..11: edge FROM the U, D, L, R neighbour to this cell passes it
constexpr int META_BIN_SHIFT = 12;             // bits 12..31: histogram bin + 1 (0 = none)

// ------------------------------------------------------------------------------------------------------
// Point access.  Cloud index q of cell point (i, j) is (cr*P+i)*stride_w + cc*P + j (cell_grid.cpp:74-80),
// which is pixel (q / W, q % W) of the organised cloud.
template <bool DEPTH>
struct PointSrc {
  const void* base;  // frame base pointer
  int W;
  float fx, fy, cx, cy;
  int sheared;
  __device__ __forceinline__ void get(long long q, int row, int col, float& x, float& y, float& z) const {
    if (DEPTH) {
      const uint16_t* d = static_cast<const uint16_t*>(base);
      if (sheared) {
        row = static_cast<int>(q / W);
        col = static_cast<int>(q - static_cast<long long>(row) * W);
      }
      z = static_cast<float>(d[q]);
      // depth_image.cpp:71-73: (col - cx) * z / fx, float32, product rounded before the IEEE divide
      x = ((static_cast<float>(col) - cx) * z) / fx;
      y = ((static_cast<float>(row) - cy) * z) / fy;
    } else {
      const float* p = static_cast<const float*>(base) + 3 * q;
      x = p[0];
      y = p[1];
      z = p[2];
    }
  }
  __device__ __forceinline__ float getz(long long q) const {
    if (DEPTH) return static_cast<float>(static_cast<const uint16_t*>(base)[q]);
    return static_cast<const float*>(base)[3 * q + 2];
  }
};

// Eigen's linear-vectorised column sum for a column whose first 16-byte-aligned element is index `a`
// (redux_impl<..., LinearVectorizedTraversal, NoUnrolling> with 4-float packets).
// fetch(k) returns element k.  Used only when n % 4 != 0 (odd patch sizes); the common case is fused below.
template <class F>
__device__ float eigen_colsum_generic(F fetch, int n, int a) {
  if (a > n) a = n;
  const int A = ((n - a) / 4) * 4;
  const int A2 = ((n - a) / 8) * 8;
  if (A == 0) {
    float r = fetch(0);
    for (int i = 1; i < n; ++i) r = r + fetch(i);
    return r;
  }
  float p00 = fetch(a), p01 = fetch(a + 1), p02 = fetch(a + 2), p03 = fetch(a + 3);
  if (A > 4) {
    float p10 = fetch(a + 4), p11 = fetch(a + 5), p12 = fetch(a + 6), p13 = fetch(a + 7);
    for (int i = a + 8; i < a + A2; i += 8) {
      p00 = p00 + fetch(i);
      p01 = p01 + fetch(i + 1);
      p02 = p02 + fetch(i + 2);
      p03 = p03 + fetch(i + 3);
      p10 = p10 + fetch(i + 4);
      p11 = p11 + fetch(i + 5);
      p12 = p12 + fetch(i + 6);
      p13 = p13 + fetch(i + 7);
    }
    p00 = p00 + p10;
    p01 = p01 + p11;
    p02 = p02 + p12;
    p03 = p03 + p13;
    if (A > A2) {
      p00 = p00 + fetch(a + A2);
      p01 = p01 + fetch(a + A2 + 1);
      p02 = p02 + fetch(a + A2 + 2);
      p03 = p03 + fetch(a + A2 + 3);
    }
  }
  float r = (p00 + p02) + (p01 + p03);
  for (int i = 0; i < a; ++i) r = r + fetch(i);
  for (int i = a + A; i < n; ++i) r = r + fetch(i);
  return r;
}

// Packed 8-chain accumulator for the a == 0 case (n % 4 == 0): element k goes to chain k & 7 while k < A2,
// the chains are folded pairwise when the 8-wide part ends, a trailing 4-packet is added, then the SSE
// horizontal add (p0+p2)+(p1+p3).
struct ColSum8 {
  float c[8];
  __device__ __forceinline__ void add(int k, int A2, float v) {
    if (k < A2) {
      if (k < 8)
        c[k & 7] = v;
      else
        c[k & 7] = c[k & 7] + v;
    } else {
      const int idx = k - A2;  // 0..3: trailing packet
      if (A2 == 0) {
        c[idx] = v;
      } else {
        if (idx == 0) fold();
        c[idx] = c[idx] + v;
      }
    }
  }
  __device__ __forceinline__ void fold() {
    c[0] = c[0] + c[4];
    c[1] = c[1] + c[5];
    c[2] = c[2] + c[6];
    c[3] = c[3] + c[7];
  }
  __device__ __forceinline__ float finish(int n, int A2) {
    if (A2 > 0 && n == A2) fold();
    return (c[0] + c[2]) + (c[1] + c[3]);
  }
};

// Float32 ESTIMATES of acos and atan2 for the histogram-bin shortcut in finish_cell: degree-7 polynomial fits
// (acos(x) = sqrt(1 - x) * P(x) on [0, 1]; atan(a) = a * Q(a^2) on [0, 1]; fit errors 2.5e-8 and 7.5e-8) evaluated with
// hardware reciprocal / square root (1 ulp).  Measured against float64 over dense sweeps: absolute error < 3e-7 rad for
// the kernels, < 7e-7 with the reconstruction steps — the shortcut below budgets 1e-6.  ~35 instructions for both
// instead of ~140 for the library acosf / atan2f.  Non-finite or out-of-range input yields NaN, which fails the
// shortcut's range tests and sends the lane to the exact float64 path.
__device__ __forceinline__ float est_acos(float x) {
  const float ax = __builtin_fabsf(x);
  float p = -1.2370048531e-03f;
  p = __builtin_fmaf(p, ax, 6.5808852398e-03f);
  p = __builtin_fmaf(p, ax, -1.6965471964e-02f);
  p = __builtin_fmaf(p, ax, 3.0808253412e-02f);
  p = __builtin_fmaf(p, ax, -5.0145025020e-02f);
  p = __builtin_fmaf(p, ax, 8.8974099304e-02f);
  p = __builtin_fmaf(p, ax, -2.1459849423e-01f);
  p = __builtin_fmaf(p, ax, 1.5707963018e+00f);
  const float r = __builtin_amdgcn_sqrtf(1.0f - ax) * p;
  return x < 0.0f ? 3.14159274f - r : r;
}
__device__ __forceinline__ float est_atan2(float y, float x) {
  const float ay = __builtin_fabsf(y), ax = __builtin_fabsf(x);
  const float mx = ax > ay ? ax : ay, mn = ax > ay ? ay : ax;
  const float a = mn * __builtin_amdgcn_rcpf(mx);
  const float q = a * a;
  float p = -4.6687733077e-03f;
  p = __builtin_fmaf(p, q, 2.4166189523e-02f);
  p = __builtin_fmaf(p, q, -5.9367100790e-02f);
  p = __builtin_fmaf(p, q, 9.9060968961e-02f);
  p = __builtin_fmaf(p, q, -1.4016585042e-01f);
  p = __builtin_fmaf(p, q, 1.9969235395e-01f);
  p = __builtin_fmaf(p, q, -3.3331959724e-01f);
  p = __builtin_fmaf(p, q, 9.9999989775e-01f);
  float r = a * p;
  r = ay > ax ? 1.57079637f - r : r;
  r = x < 0.0f ? 3.14159274f - r : r;
  return y < 0.0f ? -r : r;
}

// Everything after the moments: plane fit, planarity, merge tolerance, histogram bin, record.
// (dx, dy, dz) = first point - last point of the cell (cell_segment.cpp:107).
__device__ __forceinline__ void finish_cell(const Params& prm, int n, const float sum[3], const float pp[6], float dx,
                                            float dy, float dz, bool valid, dpx_cell& out) {
  PlaneFit fit;
  fit_plane(sum, pp, n, fit);

  // hasSmallPlaneError (cell_segment.cpp:99-102)
  const float mz = fit.mean[2];
  const float thr = prm.sigma_coeff * (mz * mz) + prm.sigma_margin;
  const bool planar = valid && (static_cast<double>(fit.mse) <= static_cast<double>(thr) * static_cast<double>(thr));
  // calculateMergeTolerance (cell_segment.cpp:104-110), min_merge_dist 20
  const float diam = __builtin_sqrtf(dx * dx + (dy * dy + dz * dz));
  float t = diam * prm.tol_sin;
  t = (t < 20.0f) ? 20.0f : t;                      // std::max(t, 20)
  t = (prm.max_merge_dist < t) ? prm.max_merge_dist : t;  // std::min(t, max)
  const float tol = t * t;

  // histogram bin (normals_histogram.cpp:32-45), planar cells only (plane_extractor.cpp:286-291)
  uint32_t bin1 = 0;
  if (planar) {
    const float nx = fit.normal[0], ny = fit.normal[1], nz = fit.normal[2];
    const bool is_zero = __builtin_fabsf(nx) <= 1e-5f && __builtin_fabsf(ny) <= 1e-5f && __builtin_fabsf(nz) <= 1e-5f;
    if (!is_zero) {
      const int B = prm.hist_B;
      const double kPi = 3.14159265358979323846;
      int Xq, Yq = 0;
#if !defined(DPX_EXACT_BINS_ONLY)
      // The bin indices are floors of (B-1)*angle/range.  A float32 estimate of the angles (est_acos / est_atan2, absolute
      // error < 1e-6 rad, i.e. < 2e-5 after scaling for B <= 64) decides the floor whenever the scaled value is at
      // least `margin` away from an integer — an order of magnitude more than the estimate's error plus the <1e-15
      // error of the exact f64 evaluation — and
      // only lanes closer than that to a bin edge (or with nx == ny == 0) take the exact f64 path below.  The result
      // is therefore identical to evaluating the exact path for every cell; it just skips ~230 f64 instructions.
      // (B <= 256 keeps the scaled estimate error below 2e-4.)
      const float tX = static_cast<float>(B - 1) * est_acos(-nz) * 0.318309886183790672f;
      const float tY = static_cast<float>(B - 1) * (est_atan2(nx, ny) + 3.14159265358979f) * 0.159154943091895336f;
      const float fX = tX - __builtin_floorf(tX), fY = tY - __builtin_floorf(tY);
      const float margin = B <= 64 ? 2e-4f : 1e-3f;  // >= 10x the estimate error in both regimes
      const bool sureX = fX > margin && fX < 1.0f - margin;
      const bool sureY = fY > margin && fY < 1.0f - margin && !(nx == 0.0f && ny == 0.0f);
      if (sureX && sureY && B <= 256) {
        Xq = static_cast<int>(tX);
        if (Xq > 0) Yq = static_cast<int>(tY);
      } else
#endif
      {
        const double dnx = nx, dny = ny;
        const double proj = dpxm::dsqrt(dnx * dnx + dny * dny);
        const double polar = dpxm::dacos(static_cast<double>(-nz));
        const double azim = dpxm::datan2(dnx / proj, dny / proj);
        Xq = static_cast<int>((B - 1) * (polar - 0.0) / (kPi - 0.0));
        // azim is NaN only when nx == ny == 0; the reference then has Xq == 0 (normal (0,0,-1)) and never
        // evaluates Yq, or reads out of bounds for a (0,0,+1) normal (undefined there; defined as 0 here).
        if (Xq > 0 && azim == azim) Yq = static_cast<int>((B - 1) * (azim - (-kPi)) / (kPi - (-kPi)));
      }
      bin1 = static_cast<uint32_t>(Yq * B + Xq) + 1u;
    }
  }

  if (valid) {
    out.sum[0] = sum[0]; out.sum[1] = sum[1]; out.sum[2] = sum[2];
    out.pp[0] = pp[0]; out.pp[1] = pp[1]; out.pp[2] = pp[2]; out.pp[3] = pp[3]; out.pp[4] = pp[4]; out.pp[5] = pp[5];
    out.normal[0] = fit.normal[0]; out.normal[1] = fit.normal[1]; out.normal[2] = fit.normal[2];
    out.d = fit.d;
    out.mse = fit.mse;
    out.tol = tol;
    out.info = CELL_VALID | (planar ? CELL_PLANAR : 0u) | (fit.path != 0 ? CELL_QL : 0u) | (bin1 << 8);
  } else {
    // reference leaves the default-constructed stat (mse = FLT_MAX, score 0) — cell_segment_stat.cpp:27
    out.sum[0] = out.sum[1] = out.sum[2] = 0.f;
    out.pp[0] = out.pp[1] = out.pp[2] = out.pp[3] = out.pp[4] = out.pp[5] = 0.f;
    out.normal[0] = out.normal[1] = out.normal[2] = 0.f;
    out.d = 0.f;
    out.mse = 3.402823466e+38f;
    out.tol = 0.f;
    out.info = 0u;
  }
}

// One cell: everything CellSegment::CellSegment does (cell_segment.cpp:21-35) plus the histogram bin.
template <int PT, bool DEPTH>
__device__ __forceinline__ void planefit_cell(const Params& prm, const PointSrc<DEPTH>& src, int cr, int cc,
                                               dpx_cell& out) {
  const int P = PT ? PT : prm.P;
  const int n = P * P;
  const long long q0 = static_cast<long long>(cr) * P * prm.stride_w + static_cast<long long>(cc) * P;
  const int row0 = cr * P, col0 = cc * P;

  // ---- validity: hasValidPoints + isDepthContinuous (cell_segment.cpp:57-97)
  bool cont;
  {
    // horizontal: k = n/2 .. n/2+P-1
    const int mid = n / 2;
    int disc = 0;
    float prev = src.getz(q0 + static_cast<long long>(mid / P) * prm.stride_w + (mid % P));
    for (int t = 0; t < P; ++t) {
      const int k = mid + t;
      const float cur = src.getz(q0 + static_cast<long long>(k / P) * prm.stride_w + (k % P));
      if (cur > 0 && __builtin_fabsf(cur - prev) < prm.disc_thr)
        prev = cur;
      else if (cur > 0)
        ++disc;
    }
    cont = disc < prm.max_disc;
    // vertical: k = P/2 + t*P
    disc = 0;
    prev = src.getz(q0 + (P / 2));
    for (int t = 0; t < P; ++t) {
      const float cur = src.getz(q0 + static_cast<long long>(t) * prm.stride_w + (P / 2));
      if (cur > 0 && __builtin_fabsf(cur - prev) < prm.disc_thr)
        prev = cur;
      else if (cur > 0)
        ++disc;
    }
    cont = cont && (disc < prm.max_disc);
  }

  // ---- moments (cell_segment_stat.cpp:29-33)
  float sum[3], pp[6];
  float fx0 = 0, fy0 = 0, fz0 = 0, lx = 0, ly = 0, lz = 0;  // first and last point
  int valid_pts = 0;
  const bool gemm = (n + 6) >= 20;  // Eigen GEMM vs lazy product threshold
  if ((n & 3) == 0) {
    const int A2 = (n / 8) * 8;
    ColSum8 sx, sy, sz;
    float xx = 0.f, xy = 0.f, xz = 0.f, yy = 0.f, yz = 0.f, zz = 0.f;
#pragma unroll
    for (int i = 0; i < P; ++i) {
#pragma unroll
      for (int j = 0; j < P; ++j) {
        const int k = i * P + j;
        float x, y, z;
        src.get(q0 + static_cast<long long>(i) * prm.stride_w + j, row0 + i, col0 + j, x, y, z);
        valid_pts += z > 0;
        if (k == 0) {
          fx0 = x;
          fy0 = y;
          fz0 = z;
        }
        if (k == n - 1) {
          lx = x;
          ly = y;
          lz = z;
        }
        sx.add(k, A2, x);
        sy.add(k, A2, y);
        sz.add(k, A2, z);
        if (gemm || k > 0) {
          xx = x * x + xx;
          xy = x * y + xy;
          xz = x * z + xz;
          yy = y * y + yy;
          yz = y * z + yz;
          zz = z * z + zz;
        } else {
          xx = x * x;
          xy = x * y;
          xz = x * z;
          yy = y * y;
          yz = y * z;
          zz = z * z;
        }
      }
    }
    sum[0] = sx.finish(n, A2);
    sum[1] = sy.finish(n, A2);
    sum[2] = sz.finish(n, A2);
    pp[0] = xx; pp[1] = xy; pp[2] = xz; pp[3] = yy; pp[4] = yz; pp[5] = zz;
  } else {
    // odd patch sizes: column j of the reference's temporary starts j*n floats past a 16-byte boundary
    auto fx_ = [&](int k) { float x, y, z; const int i = k / P, j = k - i * P; src.get(q0 + static_cast<long long>(i) * prm.stride_w + j, row0 + i, col0 + j, x, y, z); return x; };
    auto fy_ = [&](int k) { float x, y, z; const int i = k / P, j = k - i * P; src.get(q0 + static_cast<long long>(i) * prm.stride_w + j, row0 + i, col0 + j, x, y, z); return y; };
    auto fz_ = [&](int k) { float x, y, z; const int i = k / P, j = k - i * P; src.get(q0 + static_cast<long long>(i) * prm.stride_w + j, row0 + i, col0 + j, x, y, z); return z; };
    sum[0] = eigen_colsum_generic(fx_, n, 0);
    sum[1] = eigen_colsum_generic(fy_, n, (4 - (n % 4)) % 4);
    sum[2] = eigen_colsum_generic(fz_, n, (4 - ((2 * n) % 4)) % 4);
    float xx = 0.f, xy = 0.f, xz = 0.f, yy = 0.f, yz = 0.f, zz = 0.f;
    for (int k = 0; k < n; ++k) {
      const int i = k / P, j = k - i * P;
      float x, y, z;
      src.get(q0 + static_cast<long long>(i) * prm.stride_w + j, row0 + i, col0 + j, x, y, z);
      valid_pts += z > 0;
      if (k == 0) { fx0 = x; fy0 = y; fz0 = z; }
      if (k == n - 1) { lx = x; ly = y; lz = z; }
      if (gemm || k > 0) {
        xx = x * x + xx; xy = x * y + xy; xz = x * z + xz; yy = y * y + yy; yz = y * z + yz; zz = z * z + zz;
      } else {
        xx = x * x; xy = x * y; xz = x * z; yy = y * y; yz = y * z; zz = z * z;
      }
    }
    pp[0] = xx; pp[1] = xy; pp[2] = xz; pp[3] = yy; pp[4] = yz; pp[5] = zz;
  }

  const bool valid = prm.valid_thr >= 0 && static_cast<long long>(valid_pts) >= prm.valid_thr && cont;
  finish_cell(prm, n, sum, pp, fx0 - lx, fy0 - ly, fz0 - lz, valid, out);
}

// v1 launch shape: one thread per cell, 256-thread blocks over batch*C cells.
template <int PT, bool DEPTH>
__global__ __launch_bounds__(256) void k_planefit(Params prm, const void* __restrict__ in, dpx_cell* __restrict__ cells) {
  const long long gid = static_cast<long long>(blockIdx.x) * 256 + threadIdx.x;
  const long long total = static_cast<long long>(prm.batch) * prm.C;
  if (gid >= total) return;
  const int frame = static_cast<int>(gid / prm.C);
  const int c = static_cast<int>(gid - static_cast<long long>(frame) * prm.C);
  const int cr = c / prm.nh, cc = c - cr * prm.nh;
  PointSrc<DEPTH> src;
  src.base = DEPTH ? static_cast<const void*>(static_cast<const uint16_t*>(in) + static_cast<long long>(frame) * prm.N)
                   : static_cast<const void*>(static_cast<const float*>(in) + 3ll * frame * prm.N);
  src.W = prm.W;
  src.fx = prm.fx; src.fy = prm.fy; src.cx = prm.cx; src.cy = prm.cy;
  src.sheared = prm.sheared;
  dpx_cell rec;
  planefit_cell<PT, DEPTH>(prm, src, cr, cc, rec);
  // 64-byte record as four 16-byte stores
  float4* dst = reinterpret_cast<float4*>(cells + gid);
  const float4* s4 = reinterpret_cast<const float4*>(&rec);
  dst[0] = s4[0];
  dst[1] = s4[1];
  dst[2] = s4[2];
  dst[3] = s4[3];
}

// ------------------------------------------------------------------------------------------------------
// K3: directed edge bits.  Edge u->v is set iff both cells are planar and growSeed would accept v from u
// ignoring the dynamic unassigned/activated state (plane_extractor.cpp:377-383):
//   n_u . n_v >= min_cos  &&  (n_u . mean_v + d_u)^2 <= tol_v        (comparisons in double)
// Output: one uint32 per cell: bit 0 planar, bits 2..5 out-edges, bits 8..11 in-edges, bits 12..31 histogram bin + 1,
// plus the cell mse as a compact float array.
__global__ __launch_bounds__(256) void k_cellgraph(Params prm, const dpx_cell* __restrict__ cells, uint32_t* __restrict__ meta,
                                                   float* __restrict__ mse_c) {
  // Workgroups go to the 8 XCDs round robin, and each XCD has its own L2: with the plain order a cell record is fetched
  // from memory by the XCD of its own workgroup and again by the XCDs of the workgroups one cell row up and down (measured:
  // 4x the records' size at 3840x2160).  So XCD k takes the k-th eighth of the cells as one contiguous range (the grid is
  // a multiple of 8 workgroups).
  const unsigned chunk = gridDim.x >> 3;
  const long long gid = static_cast<long long>((blockIdx.x & 7u) * chunk + (blockIdx.x >> 3)) * 256 + threadIdx.x;
  const long long total = static_cast<long long>(prm.batch) * prm.C;
  if (gid >= total) return;
  const int frame = static_cast<int>(gid / prm.C);
  const int c = static_cast<int>(gid - static_cast<long long>(frame) * prm.C);
  const int cr = c / prm.nh, cc = c - cr * prm.nh;
  // Everything the four tests read is fetched up front, unconditionally (a missing neighbour reads the cell itself): the
  // own record's three quarters that matter and the same of the four neighbours, fifteen 16-byte loads in flight at once.
  // With the loads behind the tests' branches a wave made up to ten dependent round trips to memory.
  const float4* f4 = reinterpret_cast<const float4*>(cells + static_cast<long long>(frame) * prm.C);
  const int nb[4] = {cr >= 1 ? c - prm.nh : -1, cr + 1 < prm.nv ? c + prm.nh : -1, cc >= 1 ? c - 1 : -1, cc + 1 < prm.nh ? c + 1 : -1};
  const float4 o0 = f4[4 * c], o2 = f4[4 * c + 2], o3 = f4[4 * c + 3];  // (sum, pp0) (pp5, normal) (d, mse, tol, info)
  float4 v0[4], v2[4], v3[4];
#pragma unroll
  for (int dir = 0; dir < 4; ++dir) {
    const int v = nb[dir] >= 0 ? nb[dir] : c;
    v0[dir] = f4[4 * v];
    v2[dir] = f4[4 * v + 2];
    v3[dir] = f4[4 * v + 3];
  }
  const uint32_t info = __float_as_uint(o3.w);
  uint32_t m = 0;
  if (info & CELL_PLANAR) {
    m = F_UNASSIGNED | ((info >> 8) << META_BIN_SHIFT);
    const float nx = o2.y, ny = o2.z, nz = o2.w;
    const double du = static_cast<double>(o3.x);
    const float nf = static_cast<float>(prm.n);
    const float smx = o0.x / nf, smy = o0.y / nf, smz = o0.z / nf;  // own mean (for in-edges)
    const double tol_self = static_cast<double>(o3.z);
#pragma unroll
    for (int dir = 0; dir < 4; ++dir) {
      if (nb[dir] < 0) continue;
      if (!(__float_as_uint(v3[dir].w) & CELL_PLANAR)) continue;
      const float vnx = v2[dir].y, vny = v2[dir].z, vnz = v2[dir].w;
      const float cosf_ = dot3(nx, ny, nz, vnx, vny, vnz);
      if (!(static_cast<double>(cosf_) >= static_cast<double>(prm.min_cos))) continue;  // the cosine is symmetric
      // out-edge  c -> v : (n_c . mean_v + d_c)^2 <= tol_v
      const float mx = v0[dir].x / nf, my = v0[dir].y / nf, mzv = v0[dir].z / nf;  // mean_ = coord_sum_/nr_pts_
      const double tt = static_cast<double>(dot3(nx, ny, nz, mx, my, mzv)) + du;
      if (tt * tt <= static_cast<double>(v3[dir].z)) m |= (F_EDGE0 << dir);
      // in-edge   v -> c : (n_v . mean_c + d_v)^2 <= tol_c
      const double ti = static_cast<double>(dot3(vnx, vny, vnz, smx, smy, smz)) + static_cast<double>(v3[dir].x);
      if (ti * ti <= tol_self) m |= (F_INEDGE0 << dir);
    }
  }
  meta[gid] = m;
  mse_c[gid] = o3.y;  // compact copy for the region-growing kernel's seed search
}

// ------------------------------------------------------------------------------------------------------
// K6: toImageLabels (plane_extractor.cpp:455-470).  cell_labels already hold merge[L-1]+1 (or 0).
// One thread writes 4 consecutive pixels (16-byte store).  Pixels outside the nv*P x nh*P grid (image size
// not a multiple of P: out-of-bounds read in the reference) are defined as 0.
__global__ __launch_bounds__(256) void k_labels(Params prm, const int32_t* __restrict__ cell_labels, int32_t* __restrict__ labels) {
  const int W4 = (prm.W + 3) >> 2;
  const long long per_frame = static_cast<long long>(prm.H) * W4;
  const long long gid = static_cast<long long>(blockIdx.x) * 256 + threadIdx.x;
  if (gid >= per_frame * prm.batch) return;
  const int frame = static_cast<int>(gid / per_frame);
  const long long r = gid - frame * per_frame;
  const int row = static_cast<int>(r / W4);
  const int col = static_cast<int>(r - static_cast<long long>(row) * W4) * 4;
  const int cr = row / prm.P;
  const int32_t* cl = cell_labels + static_cast<long long>(frame) * prm.C + static_cast<long long>(cr) * prm.nh;
  int32_t v[4];
  int cc = col / prm.P;
  int rem = col - cc * prm.P;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    v[i] = (cr < prm.nv && cc < prm.nh) ? cl[cc] : 0;
    if (++rem == prm.P) {
      rem = 0;
      ++cc;
    }
  }
  int32_t* dst = labels + static_cast<long long>(frame) * prm.N + static_cast<long long>(row) * prm.W + col;
  if (col + 3 < prm.W && ((prm.W & 3) == 0)) {
    *reinterpret_cast<int4*>(dst) = make_int4(v[0], v[1], v[2], v[3]);
  } else {
    for (int i = 0; i < 4; ++i)
      if (col + i < prm.W) dst[i] = v[i];
  }
}

// K6, tuned form for W % 4 == 0: one 256-thread block per (frame, cell row).  The P image rows of a cell row carry the
// same label pattern, so a thread builds its 4-pixel int4 once and stores it to all P rows: ~1 store per 16 output
// bytes and almost no arithmetic.  The cell labels of the row (nh * 4 bytes) are read straight from global memory
// (L1/L2 hits): the kernel allocates NO LDS, so its blocks also fit on CUs whose LDS is filled by region workgroups.
// blockIdx.y == nv handles the image rows below the last full cell row (H % P != 0), which are 0.
// Frames of a grouped submission go to up to kMaxGroup output buffers (`per_out` frames each).
struct LabelOuts {
  int32_t* out[kMaxGroup];
  int per_out;
};
// Block b takes the (frame, cell row) tasks b, b + gridDim.x, ...: the host starts one block per CU (a few dozen blocks
// already write at the full HBM rate; DPX_LAB_RESIDENT=0 gives one block per task).
// Issue priority of the label-expansion waves: above the plane-fit waves they share SIMDs with (their few address / store
// instructions then do not queue behind a saturated VALU), below the latency-bound region waves (priority 3).  +1 %.
#ifndef DPX_LAB_PRIO
#define DPX_LAB_PRIO 1
#endif
__global__ __launch_bounds__(256) void k_labels_rows(Params prm, const int32_t* __restrict__ cell_labels, LabelOuts outs, int rows_per_frame) {
 WaveTrace trace_(TK_LABELS);
#if DPX_LAB_PRIO
 __builtin_amdgcn_s_setprio(DPX_LAB_PRIO);
#endif
 const int ntasks = prm.batch * rows_per_frame;
 for (int task = blockIdx.x; task < ntasks; task += gridDim.x) {
  const int frame = task / rows_per_frame;
  const int sub = frame / outs.per_out;
  int32_t* __restrict__ dst = outs.out[sub] + static_cast<long long>(frame - sub * outs.per_out) * prm.N;
  const int cr = task - frame * rows_per_frame;
  const int W4 = prm.W >> 2;
  const bool tail = cr >= prm.nv;  // rows outside the cell grid
  const int row_begin = tail ? prm.nv * prm.P : cr * prm.P;
  const int row_end = tail ? prm.H : row_begin + prm.P;
  const int32_t* __restrict__ src = cell_labels + static_cast<long long>(frame) * prm.C + static_cast<long long>(tail ? 0 : cr) * prm.nh;
  for (int q = threadIdx.x; q < W4; q += 256) {
    const int col = q << 2;
    int cc = col / prm.P;
    int rem = col - cc * prm.P;
    int4 v;
    int32_t* vp = reinterpret_cast<int32_t*>(&v);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      vp[i] = (!tail && cc < prm.nh) ? src[cc] : 0;  // columns right of the cell grid (W % P != 0) are 0
      if (++rem == prm.P) {
        rem = 0;
        ++cc;
      }
    }
    // streaming stores: the label map is written once and not read again by this pipeline
    for (int r = row_begin; r < row_end; ++r) {
#ifdef DPX_LABELS_PLAIN_STORE
      *reinterpret_cast<int4*>(dst + static_cast<long long>(r) * prm.W + col) = v;
#else
      typedef int i32x4 __attribute__((ext_vector_type(4)));
      const i32x4 nv = {v.x, v.y, v.z, v.w};
      __builtin_nontemporal_store(nv, reinterpret_cast<i32x4*>(dst + static_cast<long long>(r) * prm.W + col));
#endif
    }
  }
 }
}

// DepthImage::toPointCloud (utils/depth_image.cpp:55-78) as a stand-alone kernel for the utils API.
__global__ __launch_bounds__(256) void k_backproject(const uint16_t* __restrict__ depth, int H, int W, float fx, float fy,
                                                     float cx, float cy, float* __restrict__ xyz) {
  const long long gid = static_cast<long long>(blockIdx.x) * 256 + threadIdx.x;
  if (gid >= static_cast<long long>(H) * W) return;
  const int row = static_cast<int>(gid / W), col = static_cast<int>(gid - static_cast<long long>(row) * W);
  const float z = static_cast<float>(depth[gid]);
  xyz[3 * gid + 0] = ((static_cast<float>(col) - cx) * z) / fx;
  xyz[3 * gid + 1] = ((static_cast<float>(row) - cy) * z) / fy;
  xyz[3 * gid + 2] = z;
}

}  // namespace dpx
