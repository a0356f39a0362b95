// The LDS-lean preparation launch of the phase-split form (k_region_prep) and the body it shares with the fused
// preparation + grow launch (k_regions2<2, ., 4>).  Included at the end of dpx_regions2.h.
#pragma once
namespace dpx {

// ---- Tier-2 preparation of the phase-split form, LDS-lean -----------------------------------------------------------
// Same results as k_regions2<2, ., 1> (the frame's LDS image for the grow launch, byte for byte where the grow and tail
// launches read it), but the workgroup keeps only what the preparation itself works on in LDS: the header, the
// histogram / list-bound arrays and the sort keys.  The per-cell arrays of the image (flag bytes, bins, queue
// positions, seed lists) go straight to the image in global memory, and the keys are sorted with the all-ascending
// ("normalised") bitonic network, which needs no padding to a power of two: comparators that would touch an index >= n
// are simply skipped (the virtual keys there are +infinity and already in place).  For 640x480 that is 5 KB + 8 B per
// planar cell <= 29.6 KB instead of 65 KB: five preparation workgroups per CU instead of two — the region launches
// are bound by how many frames' LDS fit on the chip at a time.
__host__ __device__ inline uint32_t r2_prep_keys_offset(const R2Layout& L) { return (L.flags + 7u) & ~7u; }
__host__ __device__ inline uint32_t r2_prep_lds_bytes(const R2Layout& L, int C) { return r2_prep_keys_offset(L) + 8u * static_cast<uint32_t>(C); }

// FUSED = the body runs in front of the grow phase in ONE launch (k_regions2<2, ., 4>): the per-cell arrays go to the
// LDS image instead of the image in global memory, and the sort keys lie over the image's queue-position / queue / seed-list
// arrays (written only after the sort) and 2 C bytes behind it.
__host__ __device__ inline uint32_t r2_fused_keys_offset(const R2Layout& L) { return (L.qpos + 7u) & ~7u; }
__host__ __device__ inline uint32_t r2_fused_lds_bytes(const R2Layout& L, int C) {
  const uint32_t k = r2_fused_keys_offset(L) + 8u * static_cast<uint32_t>(C);
  return ((k > L.image_end ? k : L.image_end) + 15u) & ~15u;
}
constexpr int kFusedKeysPerThread = 32;
constexpr int kFusedMaxCells = kPrepRT * kFusedKeysPerThread;
template <bool FUSED>
__device__ __forceinline__ void region_prep_body(const Params& prm, const RegionBufs& rb, const Regions2Extra& rx, unsigned char* smem,
                                                 const int frame, const unsigned long long t_res) {
  constexpr int kRT = dpx::kPrepRT;  // (this launch's workgroup size)
  constexpr int kRW = kRT / 64;
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int C = prm.C, nh = prm.nh, nv = prm.nv;
  const int nb = prm.hist_B * prm.hist_B;
  const unsigned long long lt_mask = (1ull << lane) - 1ull;
  const R2Layout lay = r2_layout(2, C, nb, nh);
  int* hdr = reinterpret_cast<int*>(smem);
  int* hist = reinterpret_cast<int*>(smem + lay.hist);
  int* cursor = reinterpret_cast<int*>(smem + lay.cursor);
  uint16_t* bin_start = reinterpret_cast<uint16_t*>(smem + lay.bin_start);
  uint16_t* nzlist = reinterpret_cast<uint16_t*>(smem + lay.nzlist);
  unsigned long long* skeys = reinterpret_cast<unsigned long long*>(smem + (FUSED ? r2_fused_keys_offset(lay) : r2_prep_keys_offset(lay)));
  // the frame's image in global memory, same byte offsets as the LDS image of the grow / tail launches
  unsigned char* img = FUSED ? smem : reinterpret_cast<unsigned char*>(rx.lds_img + static_cast<size_t>(frame) * (rx.img_bytes >> 4));
  uint8_t* g_flags = img + lay.flags;
  uint16_t* g_binof = reinterpret_cast<uint16_t*>(img + lay.binof);
  uint16_t* g_order = reinterpret_cast<uint16_t*>(img + lay.order);
  const dpx_cell* cells = rb.cells + static_cast<size_t>(frame) * C;

  for (int b = tid; b < nb; b += kRT) hist[b] = 0;
  if (tid < 48) hdr[tid] = 0;
  __syncthreads();
  {
    int planar_cnt = 0;
    // Fused K3 (k_cellgraph): planar bit, histogram bin and the four IN-edges of the growSeed predicate
    // (plane_extractor.cpp:377-383) straight from the plane-fit records.  A wave walks down a strip of 64 grid columns,
    // one grid row per step: the records of the row above and below are the previous and the next step's own loads
    // (kept in registers, the next row in flight while this one is evaluated), the left and right neighbours come from
    // the adjacent lanes.  The records are 64-byte structs, so every float4 load of a wave touches 64 cache lines
    // however it is arranged; reading each record once instead of five times (itself + as four cells' neighbour) is
    // what shortens this phase — it was bound by the CU's address coalescer (62 k of the launch's 124 k cycles).
    struct Row {
      float4 q0, q2, q3;  // {sum x, y, z, .}, {pp5, n0, n1, n2}, {d, mse, tol, info}
    };
    const int strips = (nh + 63) >> 6;
    const int rows_per_task = (nv + kRW - 1) / kRW;
    const float nf = static_cast<float>(prm.n);
    for (int task = wid; task < strips * kRW; task += kRW) {
      const int strip = task / kRW, r0 = (task - strip * kRW) * rows_per_task;
      const int r1 = r0 + rows_per_task < nv ? r0 + rows_per_task : nv;
      const int cc = strip * 64 + lane;
      const bool col_ok = cc < nh;
      const int ccl = col_ok ? cc : nh - 1;  // (idle lanes of the last strip read the last column)
      auto fetch = [&](int r, Row& row) {
        const float4* rc = reinterpret_cast<const float4*>(cells + (r * nh + ccl));
        row.q0 = rc[0];
        row.q2 = rc[2];
        row.q3 = rc[3];
      };
      Row up, cur, down;
      if (r0 < r1) {
        fetch(r0 > 0 ? r0 - 1 : r0, up);
        fetch(r0, cur);
      }
      for (int r = r0; r < r1; ++r) {
        const int c = r * nh + ccl;
        fetch(r + 1 < nv ? r + 1 : r, down);
        const float4 q0 = cur.q0, q2 = cur.q2, q3 = cur.q3;
        const uint32_t info = __float_as_uint(q3.w);
        // left / right neighbours: the adjacent lanes' records; the strip's edge lanes read them from memory
        float4 l2, l3, r2, r3;
        l2.y = __shfl_up(q2.y, 1); l2.z = __shfl_up(q2.z, 1); l2.w = __shfl_up(q2.w, 1); l3.x = __shfl_up(q3.x, 1); l3.w = __shfl_up(q3.w, 1);
        r2.y = __shfl_down(q2.y, 1); r2.z = __shfl_down(q2.z, 1); r2.w = __shfl_down(q2.w, 1); r3.x = __shfl_down(q3.x, 1); r3.w = __shfl_down(q3.w, 1);
        if (lane == 0 && cc >= 1) {
          const float4* rv = reinterpret_cast<const float4*>(cells + (c - 1));
          const float4 a = rv[2], b = rv[3];
          l2.y = a.y; l2.z = a.z; l2.w = a.w; l3.x = b.x; l3.w = b.w;
        }
        if (lane == 63 && cc + 1 < nh) {
          const float4* rv = reinterpret_cast<const float4*>(cells + (c + 1));
          const float4 a = rv[2], b = rv[3];
          r2.y = a.y; r2.z = a.z; r2.w = a.w; r3.x = b.x; r3.w = b.w;
        }
        uint32_t fl = (info & CELL_PLANAR) ? F_UNASSIGNED : 0u;
        const uint32_t bin1 = (info & CELL_PLANAR) ? (info >> 8) : 0u;
        if (fl) {
          const float smx = q0.x / nf, smy = q0.y / nf, smz = q0.z / nf;  // mean_ = coord_sum_ / nr_pts_
          const double tol_self = static_cast<double>(q3.z);
          const bool have[4] = {r >= 1, r + 1 < nv, cc >= 1, cc + 1 < nh};
          const float4 n2[4] = {up.q2, down.q2, l2, r2};
          const float4 n3[4] = {up.q3, down.q3, l3, r3};
#pragma unroll
          for (int dir = 0; dir < 4; ++dir) {
            if (!have[dir]) continue;
            const float4 v2 = n2[dir], v3 = n3[dir];
            if (!(__float_as_uint(v3.w) & CELL_PLANAR)) continue;
            const float cosf_ = dot3(q2.y, q2.z, q2.w, v2.y, v2.z, v2.w);  // same products/sum order as n_v . n_c
            if (!(static_cast<double>(cosf_) >= static_cast<double>(prm.min_cos))) continue;
            const double ti = static_cast<double>(dot3(v2.y, v2.z, v2.w, smx, smy, smz)) + static_cast<double>(v3.x);
            if (ti * ti <= tol_self) fl |= (F_EDGE0 << dir);
          }
        }
        // sort key of the seed lists (dpx_binsort.h): bin, then MSE, then cell id
        const int kb = static_cast<int>(bin1) - 1;
        const bool has = col_ok && fl && kb >= 0 && kb < nb;
        if (col_ok) {
          g_flags[c] = static_cast<uint8_t>(fl);
          g_binof[c] = static_cast<uint16_t>(bin1);
          if (!FUSED) reinterpret_cast<uint16_t*>(img + lay.qpos)[c] = static_cast<uint16_t>(0xFFFFu);  // (FUSED: the keys lie there)
          if (fl & F_UNASSIGNED) {
            ++planar_cnt;
            if (kb >= 0 && kb < nb) atomicAdd(&hist[kb], 1);
          }
        }
        const unsigned long long hm = __ballot(has);  // compact the keys (any order: they are sorted next)
        int base = 0;
        if (lane == 0 && hm) base = atomicAdd(&hdr[6], __popcll(hm));
        base = __shfl(base, 0);
        if (has)
          skeys[base + __popcll(hm & lt_mask)] = (static_cast<unsigned long long>(kb) << kKeyBinShift) |
                                                 (static_cast<unsigned long long>(sort_mse_key(q3.y)) << kKeyCellBits) |
                                                 static_cast<unsigned long long>(c);
        up = cur;
        cur = down;
      }
    }
    planar_cnt = wave_sum(planar_cnt);
    if (lane == 0) atomicAdd(&hdr[0], planar_cnt);
  }
  __syncthreads();
  if (wid == 0) {  // list bounds of every bin, compact list of the non-empty bins (ascending)
    int carry = 0;
    for (int base = 0; base < nb; base += 64) {
      const int b = base + lane;
      const int v = b < nb ? hist[b] : 0;
      const int incl = wave_incl_scan(v, lane);
      if (b < nb) {
        bin_start[b] = static_cast<uint16_t>(carry + incl - v);
        cursor[b] = carry + incl;
      }
      carry += __shfl(incl, 63);
      const unsigned long long nzm = __ballot(v > 0);
      if (v > 0) nzlist[hdr[5] + __popcll(nzm & lt_mask)] = static_cast<uint16_t>(b);
      if (lane == 0) hdr[5] += __popcll(nzm);
      wave_sync();
    }
    if (lane == 0) bin_start[nb] = static_cast<uint16_t>(carry);
  }
  __syncthreads();
  // ---- all-ascending bitonic network on exactly n keys.  Stage k: blocks of k keys; first the "mirror" step (i against
  // the block's k-1-i), then the half-cleaners j = k/4 .. 1 (i against i+j).  Steps whose pairs stay inside an aligned
  // 128-key chunk are finished by one wave per chunk without workgroup barriers.
  const int n = hdr[6];
  auto cex = [&](int lo, int hi) {
    if (hi < n) {
      const unsigned long long a = skeys[lo], b = skeys[hi];
      if (a > b) {
        skeys[lo] = b;
        skeys[hi] = a;
      }
    }
  };
  const int half = (n + 1) >> 1;  // pair indices that can have lo < n
  for (int k = 2; (k >> 1) < n; k <<= 1) {
    if (k <= 128) {
      for (int cb = wid * 128; cb < n; cb += kRW * 128) {
        {  // mirror step inside the chunk: pair p of the chunk -> block p / (k/2), offset p % (k/2)
          const int hk = k >> 1;
          const int blk = lane / hk, off = lane - blk * hk;
          cex(cb + blk * k + off, cb + blk * k + (k - 1 - off));
          wave_sync();
        }
        for (int j = k >> 2; j >= 1; j >>= 1) {
          cex(cb + (((lane & ~(j - 1)) << 1) | (lane & (j - 1))), cb + ((((lane & ~(j - 1)) << 1) | (lane & (j - 1))) | j));
          wave_sync();
        }
      }
      __syncthreads();
      continue;
    }
    {  // mirror step across the workgroup
      const int hk = k >> 1;
      for (int pi = tid; pi < ((n + k - 1) / k) * hk; pi += kRT) {
        const int blk = pi / hk, off = pi - blk * hk;
        cex(blk * k + off, blk * k + (k - 1 - off));
      }
      __syncthreads();
    }
    for (int j = k >> 2; j >= 64; j >>= 1) {
      for (int pi = tid; pi < half + j; pi += kRT) cex(((pi & ~(j - 1)) << 1) | (pi & (j - 1)), (((pi & ~(j - 1)) << 1) | (pi & (j - 1))) | j);
      __syncthreads();
    }
    for (int cb = wid * 128; cb < n; cb += kRW * 128) {
      for (int j = 32; j >= 1; j >>= 1) {
        cex(cb + (((lane & ~(j - 1)) << 1) | (lane & (j - 1))), cb + ((((lane & ~(j - 1)) << 1) | (lane & (j - 1))) | j));
        wave_sync();
      }
    }
    __syncthreads();
  }
  if (FUSED) {
    // the seed lists and the queue positions overwrite the key area: all keys go through registers first (the host
    // takes the fused launch only for C <= kFusedMaxCells = kPrepRT * kFusedKeysPerThread)
    uint16_t v[kFusedKeysPerThread];
#pragma unroll
    for (int q = 0; q < kFusedKeysPerThread; ++q) {
      const int i = q * kRT + tid;
      v[q] = i < n ? static_cast<uint16_t>(skeys[i] & ((1u << kKeyCellBits) - 1u)) : static_cast<uint16_t>(0);
    }
    __syncthreads();
#pragma unroll
    for (int q = 0; q < kFusedKeysPerThread; ++q) {
      const int i = q * kRT + tid;
      if (i < n) g_order[i] = v[q];
    }
    uint16_t* qp = reinterpret_cast<uint16_t*>(smem + lay.qpos);
    for (int c = tid; c < C; c += kRT) qp[c] = static_cast<uint16_t>(0xFFFFu);
    __syncthreads();
    return;
  }
  for (int i = tid; i < n; i += kRT) g_order[i] = static_cast<uint16_t>(skeys[i] & ((1u << kKeyCellBits) - 1u));
  // header, histogram and list bounds: the first lay.flags bytes of the image (a multiple of 4)
  {
    const uint32_t* s32 = reinterpret_cast<const uint32_t*>(smem);
    uint32_t* d32 = reinterpret_cast<uint32_t*>(img);
    for (int i = tid; i < static_cast<int>(lay.flags >> 2); i += kRT) d32[i] = s32[i];
  }
  if (rx.res && tid == 0) rx.res[frame * 4 + 0] = static_cast<unsigned long long>(__builtin_amdgcn_s_memtime()) - t_res;
}


// (8 waves per SIMD asked for = at most 64 vector registers: one preparation wave then fits next to four plane-fit waves
// of 112; left alone the compiler takes 71 = 72 allocated, which does not)
__global__ __launch_bounds__(kPrepRT, 8) void k_region_prep(Params prm, RegionBufs rb, Regions2Extra rx) {
  WaveTrace trace_(TK_PREP);
#if DPX_REGIONS_PRIO
  __builtin_amdgcn_s_setprio(DPX_REGIONS_PRIO);
#endif
  const unsigned long long t_res = rx.res ? static_cast<unsigned long long>(__builtin_amdgcn_s_memtime()) : 0ull;
  extern __shared__ __align__(16) unsigned char smem[];
  region_prep_body<false>(prm, rb, rx, smem, static_cast<int>(blockIdx.x), t_res);
}

}  // namespace dpx
