// dpx_refine.h — K7: RANSAC label refinement (reference: PlaneExtractor::Impl::refineLabels,
// cpp/deplex/src/deplex/plane_extractor.cpp:472-509, with RTL::PlaneRANSAC libs/rtl/include/rtl/RANSAC.hpp:25-107 and
// PlaneEstimator libs/rtl/include/rtl/Plane.hpp:13-49).  Off by default (config.h:75).
//
// One 1024-thread workgroup per frame.  What is sequential in the reference stays sequential here — the RNG stream
// (one default-seeded std::mt19937 per process() call, continuing across labels), the iterations of a label (each may
// end the loop) and the labels themselves — and every iteration's loss, a count over all points of the label, is
// evaluated by the whole workgroup.  Points are not gathered: a label's points are the pixels of its cells, visited
// through the list of its cells; the k-th point in pixel order (needed for the three samples) is found analytically
// from per-cell-row counts because labels are constant inside a cell.
//
// RNG: std::uniform_int_distribution<int>(0, N-1) over std::mt19937 as implemented by libstdc++ 11 (the toolchain of
// this image: Lemire's multiply-shift rejection for a 32-bit engine, bits/uniform_int_dist.h).  The reference's wheels
// were built with libstdc++ 10, whose distribution draws differently — the draw sequence is toolchain-defined in the
// reference itself; parity is checked against a CPU restatement that uses the standard library directly.
#pragma once
#include "dpx_regions.h"

namespace dpx {

struct RefineArgs {
  const void* input;           // depth uint16 [B][N] or xyz float [B][N][3]
  const int32_t* cell_labels;  // [B][C] merged label per cell
  int32_t* labels;             // [B][N] in/out
  const uint32_t* mt_init;     // 624 words: state of a default-seeded mt19937
  int* g_cells;                // GCELLS instantiation: [B][C] scratch for the current label's cell list
  int max_iter;
  double thr;    // paramThreshold (double of the float config value)
  double ratio;  // paramTargetInliersRatio
};

struct Mt19937 {
  uint32_t* s;  // 624 words in LDS
  int idx;
  __device__ void twist() {
    for (int i = 0; i < 624; ++i) {
      const uint32_t y = (s[i] & 0x80000000u) | (s[(i + 1) % 624] & 0x7fffffffu);
      s[i] = s[(i + 397) % 624] ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
    }
    idx = 0;
  }
  __device__ uint32_t next() {
    if (idx >= 624) twist();
    uint32_t y = s[idx++];
    y ^= (y >> 11);
    y ^= (y << 7) & 0x9d2c5680u;
    y ^= (y << 15) & 0xefc60000u;
    y ^= (y >> 18);
    return y;
  }
  // std::uniform_int_distribution<int>(0, range-1), libstdc++ 11 _S_nd<uint64_t>
  __device__ uint32_t uniform(uint32_t range) {
    unsigned long long product = static_cast<unsigned long long>(next()) * range;
    uint32_t low = static_cast<uint32_t>(product);
    if (low < range) {
      const uint32_t threshold = (0u - range) % range;
      while (low < threshold) {
        product = static_cast<unsigned long long>(next()) * range;
        low = static_cast<uint32_t>(product);
      }
    }
    return static_cast<uint32_t>(product >> 32);
  }
};

template <bool DEPTH>
__device__ __forceinline__ void refine_point(const Params& prm, const void* base, int pix, float& x, float& y, float& z) {
  if (DEPTH) {
    const int row = pix / prm.W, col = pix - row * prm.W;
    z = static_cast<float>(static_cast<const uint16_t*>(base)[pix]);
    x = ((static_cast<float>(col) - prm.cx) * z) / prm.fx;  // depth_image.cpp:71-73
    y = ((static_cast<float>(row) - prm.cy) * z) / prm.fy;
  } else {
    const float* p = static_cast<const float*>(base) + 3ll * pix;
    x = p[0]; y = p[1]; z = p[2];
  }
}

constexpr int kRefT = 1024;

// GCELLS: the cell list of the current label lives in global scratch instead of LDS (frames with more than ~39 000
// cells, e.g. 3840x2160 at P = 10)
template <bool DEPTH, bool GCELLS = false>
__global__ __launch_bounds__(kRefT) void k_refine(Params prm, RefineArgs ra) {
  const int frame = blockIdx.x, tid = threadIdx.x, lane = tid & 63;
  const int C = prm.C, nh = prm.nh, P = prm.P, n_cell = prm.n;
  extern __shared__ __align__(16) unsigned char smem[];
  uint32_t* mt = reinterpret_cast<uint32_t*>(smem);           // 624
  int* sh = reinterpret_cast<int*>(mt + 624);                  // 32 ints of scalars
  float* shf = reinterpret_cast<float*>(sh + 32);              // 16 floats: 3 sample points (9), model (4)
  int* grp_first = reinterpret_cast<int*>(shf + 16);           // per cell-row group: first index into lcells  [nv+1]
  int* grp_px = grp_first + prm.nv + 1;                        // cumulative pixel count before the group   [nv+1]
  int* lcells = GCELLS ? ra.g_cells + static_cast<size_t>(blockIdx.x) * prm.C  // [C] cells of the current label, ascending
                       : grp_px + prm.nv + 1;
  const int32_t* cl = ra.cell_labels + static_cast<size_t>(frame) * C;
  int32_t* labels = ra.labels + static_cast<size_t>(frame) * prm.N;
  const void* base = DEPTH ? static_cast<const void*>(static_cast<const uint16_t*>(ra.input) + static_cast<size_t>(frame) * prm.N)
                           : static_cast<const void*>(static_cast<const float*>(ra.input) + 3ull * frame * prm.N);
  // sh: 0 maxl, 1 ncells, 2 loss, 3 continue flag, 4 last inlier pixel, 5 ngroups, 6..8 sample pixels
  for (int i = tid; i < 624; i += kRefT) mt[i] = ra.mt_init[i];
  if (tid < 32) sh[tid] = 0;
  __syncthreads();
  {
    int m = 0;
    for (int c = tid; c < C; c += kRefT) m = cl[c] > m ? cl[c] : m;
    atomicMax(&sh[0], m);
  }
  __syncthreads();
  const int maxl = sh[0];
  Mt19937 gen{mt, 624};  // idx 624: first draw twists, like a freshly seeded engine
  for (int l = 1; l <= maxl; ++l) {
    // ---- the cells of label l in ascending order (stable compaction by waves is not needed: one pass per 1024 cells
    //      with a block-wide ordered prefix)
    __syncthreads();
    if (tid == 0) sh[1] = 0;
    __syncthreads();
    for (int base_c = 0; base_c < C; base_c += kRefT) {
      const int c = base_c + tid;
      const bool mine = c < C && cl[c] == l;
      const unsigned long long bm = __ballot(mine);
      int* wcnt = sh + 16;  // 16 wave counts
      if (lane == 0) wcnt[tid >> 6] = __popcll(bm);
      __syncthreads();
      int off = sh[1];
      for (int w = 0; w < (tid >> 6); ++w) off += wcnt[w];
      if (mine) lcells[off + __popcll(bm & ((1ull << lane) - 1ull))] = c;
      __syncthreads();
      if (tid == 0) {
        int tot = 0;
        for (int w = 0; w < kRefT / 64; ++w) tot += wcnt[w];
        sh[1] += tot;
      }
      __syncthreads();
    }
    const int ncells = sh[1];
    if (ncells == 0) continue;  // labels_indices[label].size() == 0 (plane_extractor.cpp:487-489)
    const int n = ncells * n_cell;
    if (tid == 0) {
      // cell-row groups for the k-th-point lookup
      int ng = 0, px = 0, i = 0;
      while (i < ncells) {
        const int cr = lcells[i] / nh;
        int j = i;
        while (j < ncells && lcells[j] / nh == cr) ++j;
        grp_first[ng] = i;
        grp_px[ng] = px;
        px += (j - i) * n_cell;
        ++ng;
        i = j;
      }
      grp_first[ng] = ncells;
      grp_px[ng] = px;
      sh[5] = ng;
    }
    __syncthreads();
    // ---- FindBest (RANSAC.hpp:25-51)
    float best0 = 0.f, best1 = 0.f, best2 = 0.f, best3 = 0.f;  // kept by every thread (all compute the same)
    double bestloss = __builtin_inf();
    int iteration = 0;
    for (;;) {
      // IsContinued(iteration, N - bestloss, N) (RANSAC.hpp:77-79); -inf -> int is INT_MIN on x86-64
      const int inl = (bestloss == __builtin_inf()) ? static_cast<int>(0x80000000u) : static_cast<int>(n - bestloss);
      if (!(iteration < ra.max_iter && static_cast<double>(inl) < ra.ratio * n)) break;
      ++iteration;
      if (tid == 0) {
        // GenerateModel (:81-87): draw until the std::set holds 3 distinct indices; points used in ascending order
        int a = -1, b = -1, c = -1, cnt = 0;
        while (cnt < 3) {
          const int v = static_cast<int>(gen.uniform(static_cast<uint32_t>(n)));
          if (v == a || v == b || v == c) continue;
          if (cnt == 0) a = v; else if (cnt == 1) b = v; else c = v;
          ++cnt;
        }
        int lo = a < b ? a : b, hi = a < b ? b : a;
        int mid = c;
        if (c < lo) { mid = lo; lo = c; } else if (c > hi) { mid = hi; hi = c; }
        sh[6] = lo; sh[7] = mid; sh[8] = hi;
      }
      __syncthreads();
      if (tid < 3) {
        // k-th point of the label in pixel order
        int k = sh[6 + tid];
        const int ng = sh[5];
        int g = 0;
        while (g + 1 < ng && grp_px[g + 1] <= k) ++g;
        k -= grp_px[g];
        const int cnt = grp_first[g + 1] - grp_first[g];
        const int i = k / (cnt * P);
        const int r2 = k - i * (cnt * P);
        const int sidx = r2 / P, j = r2 - sidx * P;
        const int cell = lcells[grp_first[g] + sidx];
        const int cr = cell / nh, cc = cell - cr * nh;
        const int pix = (cr * P + i) * prm.W + cc * P + j;
        float x, y, z;
        refine_point<DEPTH>(prm, base, pix, x, y, z);
        shf[3 * tid] = x; shf[3 * tid + 1] = y; shf[3 * tid + 2] = z;
      }
      __syncthreads();
      // ComputeModel (Plane.hpp:13-44), float32, same expression order; every thread computes it (no broadcast needed)
      const float x0 = shf[0], y0 = shf[1], z0 = shf[2], x1 = shf[3], y1 = shf[4], z1 = shf[5], x2 = shf[6], y2 = shf[7], z2 = shf[8];
      const float den = x0 * y1 - x1 * y0 - x0 * y2 + x2 * y0 + x1 * y2 - x2 * y1;
      const float ma = (z0 * (y1 - y2)) / den - (z1 * (y0 - y2)) / den + (z2 * (y0 - y1)) / den;
      const float mb = (z1 * (x0 - x2)) / den - (z0 * (x1 - x2)) / den - (z2 * (x0 - x1)) / den;
      const float md = (z2 * (x0 * y1 - x1 * y0)) / den - (z1 * (x0 * y2 - x2 * y0)) / den + (z0 * (x1 * y2 - x2 * y1)) / den;
      const float mc = -1.0f;
      const float ml = __builtin_sqrtf(ma * ma + mb * mb + mc * mc);
      const float m0 = ma / ml, m1 = mb / ml, m2 = mc / ml, m3 = md / ml;
      // EvaluateModel (:89-98): loss = #points with |error| >= threshold
      int cntl = 0;
      for (int q = tid; q < n; q += kRefT) {
        const int ci = q / n_cell, r = q - ci * n_cell;
        const int cell = lcells[ci];
        const int cr = cell / nh, cc = cell - cr * nh;
        const int i = r / P, j = r - i * P;
        const int pix = (cr * P + i) * prm.W + cc * P + j;
        float x, y, z;
        refine_point<DEPTH>(prm, base, pix, x, y, z);
        const double err = static_cast<double>(m0 * x + m1 * y + m2 * z + m3);
        cntl += (__builtin_fabs(err) >= ra.thr) ? 1 : 0;
      }
      cntl = wave_sum(cntl);
      if (tid == 0) sh[2] = 0;
      __syncthreads();
      if (lane == 0) atomicAdd(&sh[2], cntl);
      __syncthreads();
      const double loss = static_cast<double>(sh[2]);
      if (loss < bestloss) {  // UpdateBest (:100-105)
        best0 = m0; best1 = m1; best2 = m2; best3 = m3;
        bestloss = loss;
      }
      __syncthreads();
    }
    // ---- FindInliers (:53-62) + relabel loop (plane_extractor.cpp:500-507): non-inliers are zeroed only up to the
    //      last inlier in pixel order
    if (tid == 0) sh[4] = -1;
    __syncthreads();
    {
      int last = -1;
      for (int q = tid; q < n; q += kRefT) {
        const int ci = q / n_cell, r = q - ci * n_cell;
        const int cell = lcells[ci];
        const int cr = cell / nh, cc = cell - cr * nh;
        const int i = r / P, j = r - i * P;
        const int pix = (cr * P + i) * prm.W + cc * P + j;
        float x, y, z;
        refine_point<DEPTH>(prm, base, pix, x, y, z);
        const double err = static_cast<double>(best0 * x + best1 * y + best2 * z + best3);
        if (__builtin_fabs(err) < ra.thr) last = pix > last ? pix : last;
      }
      atomicMax(&sh[4], last);
    }
    __syncthreads();
    const int last_inlier = sh[4];
    if (last_inlier >= 0) {
      for (int q = tid; q < n; q += kRefT) {
        const int ci = q / n_cell, r = q - ci * n_cell;
        const int cell = lcells[ci];
        const int cr = cell / nh, cc = cell - cr * nh;
        const int i = r / P, j = r - i * P;
        const int pix = (cr * P + i) * prm.W + cc * P + j;
        if (pix >= last_inlier) continue;
        float x, y, z;
        refine_point<DEPTH>(prm, base, pix, x, y, z);
        const double err = static_cast<double>(best0 * x + best1 * y + best2 * z + best3);
        if (!(__builtin_fabs(err) < ra.thr)) labels[pix] = 0;
      }
    }
  }
}

}  // namespace dpx
