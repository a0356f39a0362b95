// dpx_regions.h — K4+K5: histogram seeding, FIFO region growing, segment fits and plane merging.
//
// One wavefront (64 lanes) owns one frame; the batch supplies the parallelism.  The reference's loop
// (plane_extractor.cpp:297-347 createPlaneSegments, :349-392 growSeed, :394-453 findMergedLabels /
// getConnectedComponents, normals_histogram.cpp:51-72) is sequential across seeds because the histogram
// changes after every grow; inside a frame the wave parallelises
//   * the histogram arg-max and the min-MSE seed search (lane-strided scans + butterfly reductions),
//   * the BFS frontier: lane = (queue slot, direction U/D/L/R), 16 slots per step.  Lanes that reach the same
//     cell are resolved to the lowest lane (= the reference's FIFO order), winners are appended in lane order
//     with ballot + prefix popcount, so the activation ORDER (which feeds float sums) equals the reference's,
//   * the float accumulation of a grown segment: one lane per moment component runs the exact sequential
//     chain in BFS order,
//   * segment plane fits (3x3 eigen in f64): one lane per grown segment, all segments of the frame at once.
//     This is legal because a grow never depends on an earlier fit (cells leave the histogram whether or not
//     their segment is accepted); only the 1-based segment ids do, and those come from an ordered ballot.
//   * merging: per row plane the adjacent columns are tested in parallel (tests read only state that is
//     constant during the row), accepted columns are folded in ascending order.
#pragma once
#include "dpx_kernels.h"

namespace dpx {

struct RegionBufs {
  const dpx_cell* cells;   // [B][C]
  const uint32_t* meta;    // [B][C]   k_cellgraph output
  uint32_t* order;         // [B][C]   planar cells grouped by histogram bin
  uint32_t* grow_start;    // [B][C+1] queue offsets of each grow
  float* cand;             // [B][pcap][12] sums(9), n_pts, grow index, accepted id (ints stored as bits)
  dpx_plane* planes;       // [B][pcap]
  int32_t* cell_labels;    // [B][C]   in: scratch, out: merged label per cell (0 = none)
  uint2* pairs;            // [B][2C]  adjacent (label_a, label_b) pairs
  int32_t* merge;          // [B][pcap]
  dpx_frame_stats* stats;  // [B]
  uint8_t* g_flags;        // global-state variant only: [B][C]
  uint8_t* g_owner;        // [B][C]
  uint32_t* g_queue;       // [B][C]
  int pcap;
};

__device__ __forceinline__ int wave_sum(int v) {
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off);
  return v;
}
__device__ __forceinline__ int wave_min(int v) {
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) {
    const int o = __shfl_xor(v, off);
    v = o < v ? o : v;
  }
  return v;
}
__device__ __forceinline__ int wave_incl_scan(int v, int lane) {
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    const int t = __shfl_up(v, off);
    if (lane >= off) v += t;
  }
  return v;
}

template <bool LDS>
__global__ __launch_bounds__(64) void k_regions(Params prm, RegionBufs rb) {
  const int frame = blockIdx.x;
  const int lane = threadIdx.x;
  const int C = prm.C, nh = prm.nh, nv = prm.nv;
  const int nb = prm.hist_B * prm.hist_B;
  const unsigned long long lt_mask = (1ull << lane) - 1ull;

  extern __shared__ __align__(16) unsigned char smem[];
  int* hist = reinterpret_cast<int*>(smem);
  int* bin_start = hist + nb;       // nb + 1
  int* cursor = bin_start + nb + 1;  // nb
  uint32_t* queue;
  uint8_t* flags;
  uint8_t* owner;
  if (LDS) {
    queue = reinterpret_cast<uint32_t*>(cursor + nb);
    flags = reinterpret_cast<uint8_t*>(queue + C);
    owner = flags + C;
  } else {
    queue = rb.g_queue + static_cast<size_t>(frame) * C;
    flags = rb.g_flags + static_cast<size_t>(frame) * C;
    owner = rb.g_owner + static_cast<size_t>(frame) * C;
  }
  const dpx_cell* cells = rb.cells + static_cast<size_t>(frame) * C;
  const uint32_t* meta = rb.meta + static_cast<size_t>(frame) * C;
  uint32_t* order = rb.order + static_cast<size_t>(frame) * C;
  uint32_t* grow_start = rb.grow_start + static_cast<size_t>(frame) * (C + 1);
  float* cand = rb.cand + static_cast<size_t>(frame) * rb.pcap * 12;
  dpx_plane* planes = rb.planes + static_cast<size_t>(frame) * rb.pcap;
  int32_t* cell_labels = rb.cell_labels + static_cast<size_t>(frame) * C;
  uint2* pairs = rb.pairs + static_cast<size_t>(frame) * 2 * C;
  int32_t* merge = rb.merge + static_cast<size_t>(frame) * rb.pcap;

  // ---------------- P0: state, histogram, bin-sorted order (normals_histogram.cpp:21-49) ----------------
  for (int b = lane; b < nb; b += 64) hist[b] = 0;
  __syncthreads();
  int planar_cnt = 0;
  for (int c = lane; c < C; c += 64) {
    const uint32_t m = meta[c];
    flags[c] = static_cast<uint8_t>(m & 0xFFu);
    owner[c] = 0xFF;
    cell_labels[c] = 0;
    if (m & F_UNASSIGNED) {
      ++planar_cnt;
      const int b = static_cast<int>(m >> META_BIN_SHIFT) - 1;
      if (b >= 0 && b < nb) atomicAdd(&hist[b], 1);
    }
  }
  int remaining = wave_sum(planar_cnt);
  const int planar_total = remaining;
  __syncthreads();
  {
    int carry = 0;
    for (int base = 0; base < nb; base += 64) {
      const int b = base + lane;
      const int v = b < nb ? hist[b] : 0;
      const int incl = wave_incl_scan(v, lane);
      if (b < nb) {
        bin_start[b] = carry + incl - v;
        cursor[b] = carry + incl - v;
      }
      carry += __shfl(incl, 63);
    }
    if (lane == 0) bin_start[nb] = carry;
  }
  __syncthreads();
  for (int c = lane; c < C; c += 64) {
    const uint32_t m = meta[c];
    if (m & F_UNASSIGNED) {
      const int b = static_cast<int>(m >> META_BIN_SHIFT) - 1;
      if (b >= 0 && b < nb) {
        const int pos = atomicAdd(&cursor[b], 1);
        order[pos] = static_cast<uint32_t>(c);
      }
    }
  }
  __syncthreads();

  // ---------------- P1: seeds and region growing (plane_extractor.cpp:302-327, :349-392) ----------------
  int tail = 0, ngrow = 0, activations = 0;
  for (int guard = 0; remaining > 0 && guard <= C; ++guard) {
    // most frequent bin, lowest index among maxima (std::max_element)
    int bv = -1, bb = 0x7fffffff;
    for (int b = lane; b < nb; b += 64) {
      const int v = hist[b];
      if (v > bv) {
        bv = v;
        bb = b;
      }
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
      const int ov = __shfl_xor(bv, off), ob = __shfl_xor(bb, off);
      if (ov > bv || (ov == bv && ob < bb)) {
        bv = ov;
        bb = ob;
      }
    }
    // seed_candidates.size() < min_region_growing_candidate_size (size_t comparison, :305)
    if (bv <= 0) break;
    if (static_cast<unsigned long long>(bv) < static_cast<unsigned long long>(static_cast<long long>(prm.min_cand))) break;
    // minimum-MSE candidate, first in ascending cell id among equals (:309-316)
    const int s = bin_start[bb], e = bin_start[bb + 1];
    float bm = __builtin_inff();
    int bi = 0x7fffffff, anyc = 0x7fffffff;
    for (int i = s + lane; i < e; i += 64) {
      const int c = static_cast<int>(order[i]);
      if (flags[c] & F_UNASSIGNED) {
        anyc = c < anyc ? c : anyc;
        const float m = cells[c].mse;
        if (static_cast<double>(m) < 2147483647.0) {
          if (m < bm || (m == bm && c < bi)) {
            bm = m;
            bi = c;
          }
        }
      }
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
      const float om = __shfl_xor(bm, off);
      const int oi = __shfl_xor(bi, off);
      if (om < bm || (om == bm && oi < bi)) {
        bm = om;
        bi = oi;
      }
    }
    anyc = wave_min(anyc);
    const int seed = bi != 0x7fffffff ? bi : anyc;  // the reference reads an uninitialised id in the second case
    if (seed == 0x7fffffff) break;

    // BFS, FIFO order.  A cell leaves F_UNASSIGNED the moment it is activated (equivalent to the reference's
    // separate activation map because activated cells are un-assigned at the end of the grow anyway).
    const int gstart = tail;
    if (lane == 0) {
      queue[tail] = static_cast<uint32_t>(seed);
      flags[seed] = static_cast<uint8_t>(flags[seed] & ~F_UNASSIGNED);
    }
    ++tail;
    int head = gstart;
    __syncthreads();
    while (head < tail) {
      const int nslots = (tail - head) < 16 ? (tail - head) : 16;
      const int slot = lane >> 2, dir = lane & 3;
      bool cnd = false;
      int v = 0;
      if (slot < nslots) {
        const int u = static_cast<int>(queue[head + slot]);
        const uint32_t f = flags[u];
        if (f & (F_EDGE0 << dir)) {
          v = u + (dir == 0 ? -nh : (dir == 1 ? nh : (dir == 2 ? -1 : 1)));
          cnd = (flags[v] & F_UNASSIGNED) != 0;
        }
      }
      // lowest lane among the claimants of a cell wins (at most 4 claimants => at most 5 rounds)
      for (int round = 0; round < 8; ++round) {
        const uint32_t cur = cnd ? owner[v] : 0u;
        const bool w = cnd && (static_cast<uint32_t>(lane) < cur);
        __syncthreads();
        if (w) owner[v] = static_cast<uint8_t>(lane);
        __syncthreads();
        if (!__any(w)) break;
      }
      const bool win = cnd && owner[v] == static_cast<uint8_t>(lane);
      const unsigned long long mask = __ballot(win);
      if (win) {
        queue[tail + __popcll(mask & lt_mask)] = static_cast<uint32_t>(v);
        flags[v] = static_cast<uint8_t>(flags[v] & ~F_UNASSIGNED);
        owner[v] = 0xFF;
      }
      tail += __popcll(mask);
      head += nslots;
      __syncthreads();
    }
    const int g = tail - gstart;
    // hist.removePoint for every grown cell (:324)
    for (int i = gstart + lane; i < tail; i += 64) {
      const int b = static_cast<int>(meta[queue[i]] >> META_BIN_SHIFT) - 1;
      if (b >= 0 && b < nb) atomicSub(&hist[b], 1);
    }
    if (lane == 0) grow_start[ngrow] = static_cast<uint32_t>(gstart);
    ++ngrow;
    remaining -= g;
    activations += g;
    __syncthreads();
  }
  if (lane == 0) grow_start[ngrow] = static_cast<uint32_t>(tail);
  __syncthreads();

  // ---------------- P2: float accumulation in BFS order (:318-323; the seed is added twice) -------------
  int ncand = 0;
  for (int gi = 0; gi < ngrow; ++gi) {
    const int gs = static_cast<int>(grow_start[gi]), ge = static_cast<int>(grow_start[gi + 1]);
    const int g = ge - gs;
    // cells_to_merge.size() < min_region_growing_cells_activated (size_t comparison, :329)
    if (static_cast<unsigned long long>(g) < static_cast<unsigned long long>(static_cast<long long>(prm.min_cells))) continue;
    if (ncand >= rb.pcap) break;  // cannot happen: grows are disjoint
    const int comp = lane < 9 ? lane : 0;
    const float* cf = reinterpret_cast<const float*>(cells);
    float acc = cf[static_cast<size_t>(queue[gs]) * 16 + comp];
    for (int base = gs; base < ge; base += 16) {
      float tmp[16];
#pragma unroll
      for (int t = 0; t < 16; ++t) {
        const int e = base + t;
        tmp[t] = e < ge ? cf[static_cast<size_t>(queue[e]) * 16 + comp] : 0.f;
      }
#pragma unroll
      for (int t = 0; t < 16; ++t)
        if (base + t < ge) acc = acc + tmp[t];
    }
    float* co = cand + static_cast<size_t>(ncand) * 12;
    if (lane < 9) co[lane] = acc;
    if (lane == 9) co[9] = __int_as_float((g + 1) * prm.n);
    if (lane == 10) co[10] = __int_as_float(gi);
    ++ncand;
  }
  __syncthreads();

  // ---------------- P3: plane fit per candidate segment, lane-parallel (:333-337) ------------------------
  int nplanes = 0;
  for (int base = 0; base < ncand; base += 64) {
    const int ci = base + lane;
    bool ok = false;
    PlaneFit fit;
    float s[3] = {0, 0, 0}, pp[6] = {0, 0, 0, 0, 0, 0};
    int npts = 0, gi = 0;
    float* co = cand + static_cast<size_t>(ci < ncand ? ci : 0) * 12;
    if (ci < ncand) {
      s[0] = co[0]; s[1] = co[1]; s[2] = co[2];
      pp[0] = co[3]; pp[1] = co[4]; pp[2] = co[5]; pp[3] = co[6]; pp[4] = co[7]; pp[5] = co[8];
      npts = __float_as_int(co[9]);
      gi = __float_as_int(co[10]);
      fit_plane(s, pp, npts, fit);
      ok = fit.score > prm.min_score;
    }
    const unsigned long long mask = __ballot(ok);
    const int id = nplanes + __popcll(mask & lt_mask);
    if (ok) {
      dpx_plane& pl = planes[id];
      pl.sum[0] = s[0]; pl.sum[1] = s[1]; pl.sum[2] = s[2];
      pl.pp[0] = pp[0]; pl.pp[1] = pp[1]; pl.pp[2] = pp[2]; pl.pp[3] = pp[3]; pl.pp[4] = pp[4]; pl.pp[5] = pp[5];
      pl.mean[0] = fit.mean[0]; pl.mean[1] = fit.mean[1]; pl.mean[2] = fit.mean[2];
      pl.normal[0] = fit.normal[0]; pl.normal[1] = fit.normal[1]; pl.normal[2] = fit.normal[2];
      pl.d = fit.d; pl.mse = fit.mse; pl.score = fit.score;
      pl.n_pts = npts;
      pl.n_cells = static_cast<int>(grow_start[gi + 1] - grow_start[gi]);
      pl.merge_label = id;
      merge[id] = id;
    }
    if (ci < ncand) co[11] = __int_as_float(ok ? id + 1 : 0);
    nplanes += __popcll(mask);
  }
  __syncthreads();

  // ---------------- P4: labels_map_ (:338-342) ------------------------------------------------------------
  for (int ci = 0; ci < ncand; ++ci) {
    const float* co = cand + static_cast<size_t>(ci) * 12;
    const int id1 = __float_as_int(co[11]);
    if (id1 == 0) continue;
    const int gi = __float_as_int(co[10]);
    const int gs = static_cast<int>(grow_start[gi]), ge = static_cast<int>(grow_start[gi + 1]);
    for (int i = gs + lane; i < ge; i += 64) cell_labels[queue[i]] = id1;
  }
  __syncthreads();

  // ---------------- P5: findMergedLabels (:394-426) with getConnectedComponents (:430-453) ---------------
  int merges = 0;
  if (nplanes > 0) {
    // adjacency pairs from rows 0..nv-2 x cols 0..nh-2 (right and down neighbour).  A pair is skipped when the
    // previous cell along the same border already produced it (pure de-duplication, the set is unchanged).
    int npairs = 0;
    const int gw = nh - 1, gh = nv - 1;
    const int total = gw > 0 && gh > 0 ? gw * gh : 0;
    for (int base = 0; base < total; base += 64) {
      const int idx = base + lane;
      bool hp = false, vp = false;
      int a = 0, right = 0, down = 0;
      if (idx < total) {
        const int r = idx / gw, c = idx - r * gw;
        a = cell_labels[r * nh + c];
        right = cell_labels[r * nh + c + 1];
        down = cell_labels[(r + 1) * nh + c];
        hp = a > 0 && right > 0 && a != right;
        vp = a > 0 && down > 0 && a != down;
        if (hp && r > 0 && cell_labels[(r - 1) * nh + c] == a && cell_labels[(r - 1) * nh + c + 1] == right) hp = false;
        if (vp && c > 0 && cell_labels[r * nh + c - 1] == a && cell_labels[(r + 1) * nh + c - 1] == down) vp = false;
      }
      unsigned long long m1 = __ballot(hp);
      if (hp) pairs[npairs + __popcll(m1 & lt_mask)] = make_uint2(static_cast<uint32_t>(a - 1), static_cast<uint32_t>(right - 1));
      npairs += __popcll(m1);
      m1 = __ballot(vp);
      if (vp) pairs[npairs + __popcll(m1 & lt_mask)] = make_uint2(static_cast<uint32_t>(a - 1), static_cast<uint32_t>(down - 1));
      npairs += __popcll(m1);
    }
    __syncthreads();
    uint32_t* bits = queue;  // the BFS queue is dead from here on; C words >= nplanes bits
    const int words = (nplanes + 31) >> 5;
    for (int row = 0; row < nplanes; ++row) {
      for (int w = lane; w < words; w += 64) bits[w] = 0;
      __syncthreads();
      for (int i = lane; i < npairs; i += 64) {
        const uint2 p = pairs[i];
        if (static_cast<int>(p.x) == row) atomicOr(&bits[p.y >> 5], 1u << (p.y & 31));
        if (static_cast<int>(p.y) == row) atomicOr(&bits[p.x >> 5], 1u << (p.x & 31));
      }
      __syncthreads();
      const int pid = merge[row];
      const dpx_plane& pp_ = planes[pid];
      const float pnx = pp_.normal[0], pny = pp_.normal[1], pnz = pp_.normal[2], pd = pp_.d;
      bool expanded = false;
      for (int cbase = row + 1; cbase < nplanes; cbase += 64) {
        const int col = cbase + lane;
        bool acc = false;
        if (col < nplanes && ((bits[col >> 5] >> (col & 31)) & 1u)) {
          const dpx_plane& pc = planes[col];
          const float cosv = dot3(pnx, pny, pnz, pc.normal[0], pc.normal[1], pc.normal[2]);
          const float tf = dot3(pnx, pny, pnz, pc.mean[0], pc.mean[1], pc.mean[2]) + pd;  // float add (:410-411)
          const double dist = static_cast<double>(tf) * static_cast<double>(tf);
          acc = static_cast<double>(cosv) > static_cast<double>(prm.min_cos) && dist < static_cast<double>(prm.max_merge_dist);
        }
        unsigned long long mask = __ballot(acc);
        while (mask) {
          const int l = __builtin_ctzll(mask);
          mask &= mask - 1;
          const int colx = cbase + l;
          float* pf = reinterpret_cast<float*>(&planes[pid]);
          const float* cfp = reinterpret_cast<const float*>(&planes[colx]);
          if (lane < 9) pf[lane] = pf[lane] + cfp[lane];
          if (lane == 9) planes[pid].n_pts += planes[colx].n_pts;
          if (lane == 10) merge[colx] = pid;
          expanded = true;
          ++merges;
        }
      }
      if (expanded) {
        __syncthreads();
        if (lane == 0) {
          dpx_plane& pl = planes[pid];
          PlaneFit fit;
          fit_plane(pl.sum, pl.pp, pl.n_pts, fit);
          pl.mean[0] = fit.mean[0]; pl.mean[1] = fit.mean[1]; pl.mean[2] = fit.mean[2];
          pl.normal[0] = fit.normal[0]; pl.normal[1] = fit.normal[1]; pl.normal[2] = fit.normal[2];
          pl.d = fit.d; pl.mse = fit.mse; pl.score = fit.score;
        }
      }
      __syncthreads();
    }
    // ---------------- P6: merged label per cell (toImageLabels' lookup, :464-465) ------------------------
    for (int c = lane; c < C; c += 64) {
      const int L = cell_labels[c];
      cell_labels[c] = L == 0 ? 0 : merge[L - 1] + 1;
    }
    for (int i = lane; i < nplanes; i += 64) planes[i].merge_label = merge[i];
  }
  if (lane == 0) {
    dpx_frame_stats st;
    st.planar_cells = planar_total;
    st.grows = ngrow;
    st.activations = activations;
    st.segments = nplanes;
    st.merges = merges;
    st.overflow = 0;
    rb.stats[frame] = st;
  }
}

}  // namespace dpx
