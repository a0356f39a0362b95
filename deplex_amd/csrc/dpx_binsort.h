// dpx_binsort.h — seed lists for large frames: the planar cells of every frame sorted by (histogram bin, MSE, cell id).
//
// The reference picks the seed of a grow as the FIRST minimum-MSE cell among the still unassigned cells of the most
// frequent bin (plane_extractor.cpp:303-312, candidates in ascending cell id from normals_histogram.cpp:57-66).  With
// the cells of a bin sorted once by (MSE, cell id) that is "the first live entry of the bin's list", found by looking at
// one 64-entry chunk from a head pointer that only moves forward — instead of re-scanning the whole bin (hundreds to
// tens of thousands of cells on 1280x960 / 4K frames) for every one of the thousands of grows.
//
// Sort = bitonic network on 64-bit keys, one key per cell slot, Cpad = next power of two >= C slots per frame:
//   key = bin << 51 | ordered(mse) << 19 | cell   (non-planar cells and padding: all ones, sorted to the end)
// Cells whose MSE is not below INT_MAX (or NaN) can never win the reference's `mse < min_mse` test; they get the largest
// MSE key, so among themselves they sort by cell id — which is exactly the reference's fallback (first candidate).
// Tiles of kSortTile keys are sorted in LDS by one workgroup each (k_sort_tile), larger strides run as chip-wide
// compare-exchange passes (k_sort_global) followed by an in-LDS finish per tile (k_sort_local).  k_sort_lists then turns
// the sorted keys into the per-frame cell list and the bin offsets the region kernel starts from.
#pragma once
#include "dpx_kernels.h"

namespace dpx {

constexpr int kSortTile = 4096;     // keys per LDS tile (32 KB)
constexpr int kSortThreads = 512;
constexpr int kKeyCellBits = 19;    // C < 2^19
constexpr int kKeyBinShift = 51;    // 32 MSE bits above the cell bits; bin count < 2^13

__device__ __forceinline__ uint32_t sort_mse_key(float m) {
  if (!(static_cast<double>(m) < 2147483647.0)) return 0xFFFFFFFFu;
  m = m + 0.0f;  // -0 -> +0 (the reference's float compare treats them as equal)
  const uint32_t u = __float_as_uint(m);
  return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}

// keys[B][Cpad] from the cell graph (meta: bit 0 planar, bin + 1 from META_BIN_SHIFT up) and the compact mse array
__global__ __launch_bounds__(256) void k_sort_build(Params prm, const uint32_t* __restrict__ meta, const float* __restrict__ mse_c,
                                                    unsigned long long* __restrict__ keys, int Cpad) {
  const long long i = static_cast<long long>(blockIdx.x) * 256 + threadIdx.x;
  const long long total = static_cast<long long>(prm.batch) * Cpad;
  if (i >= total) return;
  const int frame = static_cast<int>(i / Cpad), c = static_cast<int>(i - static_cast<long long>(frame) * Cpad);
  unsigned long long key = ~0ull;
  if (c < prm.C) {
    const uint32_t m = meta[static_cast<size_t>(frame) * prm.C + c];
    const int bin = static_cast<int>(m >> META_BIN_SHIFT) - 1;
    if ((m & F_UNASSIGNED) && bin >= 0 && bin < prm.hist_B * prm.hist_B)
      key = (static_cast<unsigned long long>(bin) << kKeyBinShift) |
            (static_cast<unsigned long long>(sort_mse_key(mse_c[static_cast<size_t>(frame) * prm.C + c])) << kKeyCellBits) |
            static_cast<unsigned long long>(c);
  }
  keys[i] = key;
}

// compare-exchange of the tile-local pair (i, i | j) inside LDS; `base` = index of the tile's first key within its frame
__device__ __forceinline__ void sort_lds_stage(unsigned long long* t, int tile, int j, int k, int base, int tid) {
  for (int p = tid; p < tile / 2; p += kSortThreads) {
    const int lo = ((p & ~(j - 1)) << 1) | (p & (j - 1));
    const int hi = lo | j;
    const unsigned long long a = t[lo], b = t[hi];
    const bool asc = ((base + lo) & k) == 0;
    if ((a > b) == asc) {
      t[lo] = b;
      t[hi] = a;
    }
  }
}

// full bitonic sort of every tile (strides below the tile size), directions as in the frame-wide network.
// k_from == 2: sort from scratch; otherwise only the finishing strides of network stage k_from (after k_sort_global)
__global__ __launch_bounds__(kSortThreads) void k_sort_tile(unsigned long long* __restrict__ keys, int Cpad, int tile, int k_from) {
  extern __shared__ __align__(16) unsigned char smem[];
  unsigned long long* t = reinterpret_cast<unsigned long long*>(smem);
  const int tid = threadIdx.x;
  const int tiles_per_frame = Cpad / tile;
  const int frame = blockIdx.x / tiles_per_frame;
  const int base = (blockIdx.x - frame * tiles_per_frame) * tile;
  unsigned long long* g = keys + static_cast<size_t>(frame) * Cpad + base;
  for (int i = tid; i < tile; i += kSortThreads) t[i] = g[i];
  __syncthreads();
  if (k_from == 2) {
    for (int k = 2; k <= tile; k <<= 1)
      for (int j = k >> 1; j >= 1; j >>= 1) {
        sort_lds_stage(t, tile, j, k, base, tid);
        __syncthreads();
      }
  } else {
    for (int j = tile >> 1; j >= 1; j >>= 1) {
      sort_lds_stage(t, tile, j, k_from, base, tid);
      __syncthreads();
    }
  }
  for (int i = tid; i < tile; i += kSortThreads) g[i] = t[i];
}

// one compare-exchange pass of stride j >= tile of network stage k, all frames
__global__ __launch_bounds__(256) void k_sort_global(unsigned long long* __restrict__ keys, int Cpad, int batch, int j, int k) {
  const long long p = static_cast<long long>(blockIdx.x) * 256 + threadIdx.x;
  const int half = Cpad >> 1;
  if (p >= static_cast<long long>(batch) * half) return;
  const int frame = static_cast<int>(p / half), q = static_cast<int>(p - static_cast<long long>(frame) * half);
  const int lo = ((q & ~(j - 1)) << 1) | (q & (j - 1));
  const int hi = lo | j;
  unsigned long long* g = keys + static_cast<size_t>(frame) * Cpad;
  const unsigned long long a = g[lo], b = g[hi];
  const bool asc = (lo & k) == 0;
  if ((a > b) == asc) {
    g[lo] = b;
    g[hi] = a;
  }
}

// sorted keys -> order[B][C] (cell ids, bin by bin, each bin by (mse, cell)) and bin_off[B][nb + 1]
__global__ __launch_bounds__(256) void k_sort_lists(Params prm, const unsigned long long* __restrict__ keys, int Cpad,
                                                    uint32_t* __restrict__ order, uint32_t* __restrict__ bin_off) {
  const long long i = static_cast<long long>(blockIdx.x) * 256 + threadIdx.x;
  const long long total = static_cast<long long>(prm.batch) * Cpad;
  if (i >= total) return;
  const int nb = prm.hist_B * prm.hist_B;
  const int frame = static_cast<int>(i / Cpad), pos = static_cast<int>(i - static_cast<long long>(frame) * Cpad);
  const unsigned long long key = keys[i];
  const unsigned long long prev = pos > 0 ? keys[i - 1] : 0ull;
  // bins of this entry and of its predecessor; padding / non-planar entries count as bin nb
  const int b = key == ~0ull ? nb : static_cast<int>(key >> kKeyBinShift);
  const int pb = pos == 0 ? -1 : (prev == ~0ull ? nb : static_cast<int>(prev >> kKeyBinShift));
  uint32_t* bo = bin_off + static_cast<size_t>(frame) * (nb + 1);
  for (int x = pb + 1; x <= b && x <= nb; ++x) bo[x] = static_cast<uint32_t>(pos);  // first entry of bins pb+1..b
  if (pos == Cpad - 1 && b < nb)
    for (int x = b + 1; x <= nb; ++x) bo[x] = static_cast<uint32_t>(Cpad);  // (only when C == Cpad and every cell is planar)
  if (key != ~0ull && pos < prm.C) order[static_cast<size_t>(frame) * prm.C + pos] = static_cast<uint32_t>(key & ((1u << kKeyCellBits) - 1u));
}

}  // namespace dpx
