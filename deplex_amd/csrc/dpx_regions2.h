// dpx_regions2.h — K3+K4+K5: cell graph, histogram seeding, FIFO region growing, segment fits, plane merging; one
// workgroup (DPX_RT threads) per frame.
//
// Same algorithm and the same bit-exact results as dpx_regions.h (the first-generation kernel, kept as the fallback
// for frames beyond every tier here and as a cross-check); what changes is where the time goes:
//   * the per-cell state of the sequential phase lives in LDS as far as the frame allows (tiers below: flag byte,
//     queue, queue positions, seed lists, bin), so a grow iteration waits on LDS, not on HBM/L2;
//   * the LDS flag byte of a cell holds its IN-edges (which neighbours may activate it), so a BFS lane needs only the
//     target cell's byte, and duplicate resolution is PULL based: the lane reads the queue positions of the target's
//     four neighbours and loses to any in-neighbour that sits earlier in the current FIFO batch — 2 dependent LDS
//     round trips per BFS step instead of ~8 for the claim / re-claim protocol of the first generation;
//   * seed lists are sorted once per frame by (bin, MSE, cell id): the reference's "first minimum-MSE cell of the most
//     frequent bin" is the first live entry of the bin's list; a grown cell leaves the histogram at activation time
//     (LDS atomic) and the arg-max runs over a register-resident list of the non-empty bins;
//   * seeds whose neighbours cannot be activated are committed without a BFS step; in tier 0 large regions grow with
//     workgroup-wide BFS batches;
//   * large frames (tiers 1 and 0) do not take one grow after the other: the bins that stand at the histogram's maximum
//     fire in ascending order and do not influence each other unless their grows touch, so up to 64 of them are grown
//     speculatively at once, one lane each, checked against each other with claim tables and committed together
//     ("sweep rounds", see the comment in the grow phase: 18 776 grows of a noisy 3840x2160 frame in ~2 500 rounds; a
//     round gathers the bins at the level from all 64-bin chunks of the bin list, and the non-seed cells of its grows
//     are claimed and committed as flattened items, one lane each);
//   * tier 0 also exists as three launches with a ONE-wave grow launch that keeps 29.5 KB of LDS for a 1280x960 frame:
//     what the slots of a pipeline take, because then five frames' grow waves share a CU;
//   * wave reductions use DPP row shifts / broadcasts (gfx9) instead of ds_bpermute butterflies;
//   * the parallel phases (state load, sort, float accumulation of the grown segments, f64 segment fits, label
//     writes, adjacency build) use all eight waves; the inherently sequential sequence of grows runs on wave 0 — as a
//     one-wave launch of its own in the phase-split form.
// Reference: plane_extractor.cpp:297-453, normals_histogram.cpp:51-72 (see dpx_regions.h for the parity notes).
#pragma once
#include <type_traits>

#include "dpx_binsort.h"
#include "dpx_regions.h"

namespace dpx {


// DPP controls (gfx9): row_shr:n = 0x110 + n, row_bcast:15 = 0x142, row_bcast:31 = 0x143
#define DPX_DPP(x, ctrl, rmask, bmask) \
  static_cast<uint32_t>(__builtin_amdgcn_update_dpp(static_cast<int>(x), static_cast<int>(x), ctrl, rmask, bmask, false))

// max over the 64 lanes, returned in every lane (via readlane 63)
__device__ __forceinline__ uint32_t wave_max_u32(uint32_t x) {
  uint32_t t;
  t = DPX_DPP(x, 0x111, 0xf, 0xf); x = x > t ? x : t;
  t = DPX_DPP(x, 0x112, 0xf, 0xf); x = x > t ? x : t;
  t = DPX_DPP(x, 0x114, 0xf, 0xf); x = x > t ? x : t;
  t = DPX_DPP(x, 0x118, 0xf, 0xf); x = x > t ? x : t;
  t = DPX_DPP(x, 0x142, 0xa, 0xf); x = x > t ? x : t;
  t = DPX_DPP(x, 0x143, 0xc, 0xf); x = x > t ? x : t;
  return static_cast<uint32_t>(__builtin_amdgcn_readlane(static_cast<int>(x), 63));
}
__device__ __forceinline__ uint32_t wave_min_u32(uint32_t x) { return ~wave_max_u32(~x); }
// inclusive prefix sum over the 64 lanes (DPP row shifts / broadcasts)
__device__ __forceinline__ int wave_incl_scan_i32(int x) {
  x += __builtin_amdgcn_update_dpp(0, x, 0x111, 0xf, 0xf, false);
  x += __builtin_amdgcn_update_dpp(0, x, 0x112, 0xf, 0xf, false);
  x += __builtin_amdgcn_update_dpp(0, x, 0x114, 0xf, 0xf, false);
  x += __builtin_amdgcn_update_dpp(0, x, 0x118, 0xf, 0xf, false);
  x += __builtin_amdgcn_update_dpp(0, x, 0x142, 0xa, 0xf, false);
  x += __builtin_amdgcn_update_dpp(0, x, 0x143, 0xc, 0xf, false);
  return x;
}

__device__ __forceinline__ void wave_sync() {
  __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
  __builtin_amdgcn_wave_barrier();
}

// float -> uint32 preserving < order (zeros canonicalised to +0 by the caller)
__device__ __forceinline__ uint32_t ordered_f32(float f) {
  const uint32_t u = __float_as_uint(f);
  return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}

#ifndef DPX_RT
#define DPX_RT 512
#endif
constexpr int kRT = DPX_RT;        // threads per frame workgroup
constexpr int kRW = DPX_RT / 64;   // waves
// Threads of the tail launch (PHASE 3) of the phase-split form.  256 = one wave per SIMD: a tail workgroup then needs
// 112 vector registers per SIMD instead of 224 and can start on a CU that still runs three plane-fit waves per SIMD of
// another group (its parallel phases get slower — lifetime 87 k -> 110 k cycles on an empty chip — but with groups in
// flight the launch no longer waits for half-empty CUs: +7 % pipeline throughput, -1.5 % in a 20-step run).
#ifndef DPX_TAIL_RT
#define DPX_TAIL_RT (DPX_RT < 256 ? DPX_RT : 256)
#endif
constexpr int kTailRT = DPX_TAIL_RT;
// Threads of the lean tier-2 preparation launch (k_region_prep).  256 for the same reason as the tail: one wave per SIMD
// (64 registers) finds room next to four plane-fit waves; its lifetime on an empty chip grows from 103 k to 136 k cycles,
// the pipeline gains 6 % (and the 20-step run 1.5 %).  128 threads: no further gain, short runs lose 5 %.
#ifndef DPX_PREP_RT
#define DPX_PREP_RT (DPX_RT < 256 ? DPX_RT : 256)
#endif
constexpr int kPrepRT = DPX_PREP_RT;
static_assert(kTailRT % 64 == 0 && kTailRT <= kRT && kTailRT >= 64, "tail workgroup: whole waves, at most the preparation's");
// PHASE 5 = the tail of the core-based path (dpx_regions3.h): segment sums and fits come from chip-wide launches, what is
// left per frame — ordered ids, labels_map_, adjacency, the merge walk, final labels — is cell- and plane-parallel work around
// one sequential walk, so the workgroup is as wide as the register file allows
constexpr int kTail5RT = 1024;
// entries of the per-wave staging buffer of the accumulation phase: 32 where the buffer reuses dead image arrays
// (tier 2), 64 where it lies behind the image anyway (large frames: fewer, longer chunks per segment)
__host__ __device__ constexpr int stage_entries(int tier) { return tier == 2 ? 32 : 64; }
// floats of the staging buffer; the merge phase needs at least 2304 words of it
__host__ __device__ constexpr int stage_floats(int tier) { return kRW * stage_entries(tier) * 9 > 2304 ? kRW * stage_entries(tier) * 9 : 2304; }

// LDS image of a frame's region-growing state (byte offsets), shared by the device code and the host's size
// computation.  Everything up to image_end is handed from launch to launch in split mode.  The float staging buffer of
// the tail phases (accumulation, merge bit matrix) reuses the block hist..binof, which is dead by then, when that
// block is large enough, and lies behind the image otherwise; the sort keys of the tier-2 preparation always lie
// behind the image.
struct R2Layout {
  uint32_t hist, cursor, bin_start, nzlist, flags, binof, qpos, queue, order, image_end, stage, tail_end, grow_x, grow_end;
};
// Scratch of the grow phase (P1), behind the image and NOT part of it (the phase initialises it itself; it overlays the
// tail phases' staging buffer / the fused preparation's sort keys, which are dead while P1 runs): cand[nb] = the verified
// first live entry of every bin's seed list, bclaim[nb] = lowest lane of a sweep round that takes a cell of the bin
// (exact: a bin that lost a cell is SKIPPED, so a hash collision must not fake one), and the hashed claim table of the
// round's cells (a collision there only ends a round early).
__host__ __device__ constexpr int claim_bits(int) { return 12; }
// Tier 0 keeps the BFS queue in global memory; the last 2^kRingBits entries are mirrored in an LDS ring, which is where
// the BFS reads its batch from (a batch is at most 128 entries behind the tail unless the frontier is wider than the ring).
constexpr int kRingBits = 13;
// the bins of a sweep round's events when they come from several 64-bin chunks (64 x uint16, behind the claim table)
// + the staging of a round's items: 64 x 16-byte records, 64 results, 64 source lanes
constexpr uint32_t kEvBytes = 128 + 16 + 1024 + 256 + 64;
// one-wave tier-0 grow launch (split form): small claim table and ring, so that several frames' grow waves share a CU
#ifndef DPX_LEAN_CLAIM_BITS
#define DPX_LEAN_CLAIM_BITS 10
#endif
#ifndef DPX_LEAN_RING_BITS
#define DPX_LEAN_RING_BITS 10
#endif
constexpr int kLeanClaimBits = DPX_LEAN_CLAIM_BITS, kLeanRingBits = DPX_LEAN_RING_BITS;
// The BFS and the seed scan read the flag byte / queue position of a cell's four neighbours WITHOUT clamping the index
// (an index outside the grid is masked by the in-edge bit of that non-existent neighbour): `nh` guard bytes in front
// of and behind the flag array (and the arrays that follow it) keep every such read inside the allocation.
__host__ __device__ inline R2Layout r2_layout(int tier, int C, int nb, int nh) {
  R2Layout L{};
  uint32_t o = 192;  // header words
  L.hist = o; o += 4u * nb;
  L.cursor = o; o += 4u * nb;
  L.bin_start = o; o += static_cast<uint32_t>(nb + 2) * (tier == 0 ? 4u : 2u);
  o = (o + 3u) & ~3u;
  L.nzlist = o; o += 2u * nb;
  o = (o + 3u) & ~3u;
  if (o < static_cast<uint32_t>(nh)) o = (static_cast<uint32_t>(nh) + 3u) & ~3u;  // flags[c - nh] stays in front of flags
  L.flags = o; o += static_cast<uint32_t>(C);
  if (tier == 0) o += static_cast<uint32_t>(nh);  // flags[c + nh]: tiers 1/2 have 2*C bytes of qpos behind the flags
  o = (o + 3u) & ~3u;
  if (tier >= 1) { L.binof = o; o += 2u * C; }  // (bin + 1 of every cell: read by each BFS activation)
  o = (o + 3u) & ~3u;
  const uint32_t dead_end = o;  // hist..here is not read after the grow phase
  if (tier >= 1) {
    L.qpos = o; o += 2u * C;
    L.queue = o; o += 2u * C;
    L.order = o; o += 2u * C;
  }
  L.image_end = (o + 15u) & ~15u;
  const uint32_t stage_bytes = static_cast<uint32_t>(stage_floats(tier)) * 4u;
  if (dead_end - L.hist >= stage_bytes) {
    L.stage = L.hist;
    L.tail_end = L.image_end;
  } else {
    L.stage = L.image_end;
    L.tail_end = L.image_end + stage_bytes;
  }
  L.grow_x = L.image_end;
  L.grow_end = tier == 2 ? L.grow_x  // (tier 2 grows event by event: no scratch)
                         : L.grow_x + 8u * static_cast<uint32_t>(nb) + (4u << claim_bits(tier)) + kEvBytes + (tier == 0 ? (4u << kRingBits) + 8u * static_cast<uint32_t>(nb) : 0u);
  return L;
}
// LDS of the one-wave tier-0 grow launch of the split form
__host__ __device__ inline uint32_t r2_lean_grow_end(const R2Layout& L, int nb) {
  return L.grow_x + 8u * static_cast<uint32_t>(nb) + (4u << kLeanClaimBits) + kEvBytes + (4u << kLeanRingBits);
}

struct Regions2Extra {
  const float* mse_c;   // [B][C] compact per-cell mse (k_cellgraph output; read by dpx_binsort.h, not by k_regions2)
  uint32_t* cand_grow;  // [B][pcap][4] per candidate segment: grow index, first and one-past-last queue position, unused
  unsigned long long* dbg;  // diagnostic build only: [B][16] shader-clock stamps / sums
  float4* lds_img;          // split mode: [B][img_bytes] image of the workgroup's LDS handed from phase to phase
  int img_bytes;            // multiple of 16
  const uint32_t* bin_off;  // sorted tiers: [B][nb + 1] offsets of each bin's list inside rb.order (dpx_binsort.h)
  uint32_t* g_qpos;         // tier 0: [B][C] queue position of each cell
  int sort_pad;             // tier 2: power of two >= C, slots of the in-LDS seed-list sort
  int csr_chunk;            // merge phase: rows with more adjacency entries than this (<= 64) use the bit-row walk
  unsigned long long* res;  // diagnostic (DPX_RESIDENCY=1): [B][4] shader-clock lifetime of the frame's workgroup in launch 1, 2, 3
  int32_t* labels_out[kMaxGroup];   // non-null: the tail also expands the cell labels to the pixel label map (fused K6);
  int frames_per_out;       // frame f writes to labels_out[f / frames_per_out] (grouped submissions have one buffer each)
  const uint32_t* biglist;  // PHASE 5: [B][pcap] the candidates of 1 024 cells and more, listed from the end (dpx_regions3.h)
  const float* candfit;     // PHASE 5: [B][pcap][12] the candidate segments' fits (dpx_regions3.h k_seg_fit): mean, normal, d, mse, score, ok
};

constexpr int kDbgSlots = 32;  // words per frame of the diagnostic instantiation's output
#define DPX_STAMP() (DBG ? static_cast<unsigned long long>(__builtin_amdgcn_s_memtime()) : 0ull)

// DBG = true is a separate diagnostic instantiation (phase timing through s_memtime); production uses DBG = false.
//
// PHASE splits the kernel into three launches that hand the workgroup's LDS image over through global memory:
//   1 = preparation (P0, kRT threads), 2 = seeds + region growing (P1, ONE wave per frame), 3 = the parallel tail
//   (P2..P6, kRT threads); 0 = everything in one launch.  Why: the sequential phase takes ~2/3 of the kernel's time but
//   uses one wave; in a single launch the other waves of the workgroup sit at a barrier holding ~100 VGPRs each, and with
//   many batches in flight the resident region workgroups leave no registers / wave slots for the plane-fit kernel of
//   the next batches (measured: its marginal cost doubles).  The one-wave launch needs ~1/8 of the registers and slots
//   and, in tier 2, 32.6 KB of LDS instead of the preparation's 65 KB.
#ifndef DPX_REGIONS_PRIO
#define DPX_REGIONS_PRIO 3
#endif
// TIER: where the per-cell state of the sequential phase lives
//   2 = everything in LDS (9 B/cell + 8 B per sort slot while preparing), cell graph fused in, seed lists sorted in LDS;
//   1 = flags, queue, queue positions and the seed lists in LDS (7 B/cell); bins via the cell-graph words;
//   0 = only the flag byte in LDS; queue, queue positions and sorted seed lists in global memory (L2) with 32-bit
//       indices — frames up to ~150 000 cells (3840x2160 at P = 10 has 82 944).
// Tiers 1 and 0 take their seed lists from dpx_binsort.h (rb.order + rx.bin_off).
//   PHASE 4 (tier 2) = the lean preparation (region_prep_body) and the grow phase in ONE launch: the waves other than wave 0
//   end after the preparation.  A frame's grow then starts as soon as its own preparation is done instead of after the
//   slowest preparation workgroup of the whole group, and the image does not travel through global memory in between.
template <bool FUSED>
__device__ __forceinline__ void region_prep_body(const Params& prm, const RegionBufs& rb, const Regions2Extra& rx, unsigned char* smem,
                                                 const int frame, const unsigned long long t_res);
template <int TIER, bool DBG, int PHASE>
// (the fused preparation + grow launch is held at the 72 vector registers its preparation part needs: 73 would be
// allocated as 80, and the launch shares its CUs with the plane-fit waves of the other groups)
// (second launch bound: 7 workgroups of kPrepRT = 256 threads per CU = 7 waves per SIMD = at most 72 registers)
__global__ __launch_bounds__(PHASE == 2 ? 64 : (PHASE == 3 ? kTailRT : (PHASE == 4 ? kPrepRT : (PHASE == 5 ? kTail5RT : kRT))), PHASE == 4 ? 7 : 1) void k_regions2(Params prm, RegionBufs rb, Regions2Extra rx) {
  // (inside the kernel kRT / kRW are THIS launch's workgroup size: the tail may run with fewer waves)
  constexpr int kRT = PHASE == 3 ? dpx::kTailRT : (PHASE == 4 ? dpx::kPrepRT : (PHASE == 2 ? 64 : (PHASE == 5 ? dpx::kTail5RT : dpx::kRT)));
  constexpr int kRW = kRT / 64;
  constexpr bool FULL = TIER == 2;
  constexpr bool HASBIN = TIER >= 1;  // bin of every cell in LDS (tier 0 reads the cell-graph words in global memory)
  constexpr bool GST = TIER == 0;
  using idx_t = typename std::conditional<GST, uint32_t, uint16_t>::type;
  constexpr uint32_t kNoPos = GST ? 0xFFFFFFFFu : 0xFFFFu;
  // (tier 0 in split form: its label map and queues are in global memory anyway; the grow launch is ONE wave, so large
  // regions are not widened over a workgroup there — the form is for streams of mid-size frames, see the host side)
  static_assert(PHASE != 4 || (TIER == 2 && !DBG), "the fused preparation + grow launch exists for tier 2");
  WaveTrace trace_(PHASE == 1 ? TK_PREP : ((PHASE == 2 || PHASE == 4) ? TK_GROW : TK_TAIL));
  static_assert(PHASE != 5 || TIER == 0, "the core path's tail keeps its queue and label map in global memory");
#if DPX_REGIONS_PRIO
  __builtin_amdgcn_s_setprio(DPX_REGIONS_PRIO);  // latency-bound waves: win issue arbitration against co-resident streaming kernels
#endif
  unsigned long long t_start = DPX_STAMP(), t_p0 = 0, t_p1 = 0, t_p2 = 0, t_p3 = 0, t_p4 = 0, t_p5 = 0;
  const unsigned long long t_res = (PHASE > 0 && rx.res) ? static_cast<unsigned long long>(__builtin_amdgcn_s_memtime()) : 0ull;
  const unsigned long long t_real = ((PHASE == 2 || PHASE == 4) && rx.res) ? static_cast<unsigned long long>(__builtin_amdgcn_s_memrealtime()) : 0ull;
  unsigned long long a_argmax = 0, a_seed = 0, a_bfs = 0, n_steps = 0, n_spec = 0, n_rounds = 0, n_scan1 = 0, n_scan64 = 0, a_scan = 0, a_spec = 0, n_pops = 0, n_stop_inc = 0, n_stop_bin = 0, n_stop_cell = 0, a_claims = 0, a_commit = 0, n_fast = 0, a_fast = 0, n_gather = 0, n_itemover = 0, n_cont = 0, a_walk = 0;
  const int frame = blockIdx.x;
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int C = prm.C, nh = prm.nh, nv = prm.nv;
  const int nb = prm.hist_B * prm.hist_B;
  const unsigned long long lt_mask = (1ull << lane) - 1ull;

  extern __shared__ __align__(16) unsigned char smem[];
  const R2Layout lay = r2_layout(TIER, C, nb, nh);
  int* hdr = reinterpret_cast<int*>(smem);  // 48 ints: 0 planar count, 1 tail, 2 ngrow, 3 ncand, 5 nnz bins, 16.. wave counts
  int* hist = reinterpret_cast<int*>(smem + lay.hist);
  int* cursor = reinterpret_cast<int*>(smem + lay.cursor);  // END of bin b's cell list
  idx_t* bin_start = reinterpret_cast<idx_t*>(smem + lay.bin_start);  // nb + 1 (+1 pad): head of bin b's cell list
  uint16_t* nzlist = reinterpret_cast<uint16_t*>(smem + lay.nzlist);   // non-empty bins, ascending
  uint8_t* flags = smem + lay.flags;
  idx_t *qpos, *queue, *order;
  if constexpr (GST) {
    qpos = rx.g_qpos + static_cast<size_t>(frame) * C;
    queue = rb.g_queue + static_cast<size_t>(frame) * C;
    order = rb.order + static_cast<size_t>(frame) * C;
  } else {
    qpos = reinterpret_cast<idx_t*>(smem + lay.qpos);
    queue = reinterpret_cast<idx_t*>(smem + lay.queue);
    order = reinterpret_cast<idx_t*>(smem + lay.order);
  }
  uint16_t* binof = reinterpret_cast<uint16_t*>(smem + lay.binof);  // HASBIN: C (bin + 1)
  // phase-local: float staging of the tail phases (P2.., also the merge phase's bit matrix; see r2_layout) and, in
  // the tier-2 preparation, the 64-bit sort keys behind the image
  float* stage = reinterpret_cast<float*>(smem + lay.stage);
  unsigned long long* skeys = reinterpret_cast<unsigned long long*>(smem + lay.image_end);

  const dpx_cell* cells = rb.cells + static_cast<size_t>(frame) * C;
  const uint32_t* meta = rb.meta + static_cast<size_t>(frame) * C;
  uint32_t* grow_start = rb.grow_start + static_cast<size_t>(frame) * (C + 1);
  uint4* cand_grow = reinterpret_cast<uint4*>(rx.cand_grow) + static_cast<size_t>(frame) * rb.pcap;
  float* cand = rb.cand + static_cast<size_t>(frame) * rb.pcap * 12;
  dpx_plane* planes = rb.planes + static_cast<size_t>(frame) * rb.pcap;
  int32_t* cell_labels = rb.cell_labels + static_cast<size_t>(frame) * C;
  uint2* pairs = rb.pairs + static_cast<size_t>(frame) * 2 * C;
  int32_t* merge = rb.merge + static_cast<size_t>(frame) * rb.pcap;

  float4* lds_v = reinterpret_cast<float4*>(smem);
  float4* img = rx.lds_img + static_cast<size_t>(frame) * (rx.img_bytes >> 4);
  const int img_vec = rx.img_bytes >> 4;
  if (PHASE == 2 || PHASE == 3 || PHASE == 5) {
    // continue from the previous launch's LDS image: gfx950 LDS-DMA (global_load_lds_dwordx4: lane l of a wave delivers
    // bytes [16 l, 16 l + 16) of a 1 KB chunk straight into LDS, no staging registers).  All chunks of a wave are in
    // flight together and awaited once — the plain copy loop compiles to load, wait, store per KB, i.e. 32 serialized
    // memory round trips in front of the one-wave grow launch.
    // Not everything travels: the tail reads only the header and the BFS queue — every other array of the image is dead
    // by then or rebuilt (labels_map_, staging, plane views), so the grow launch writes back, and the tail fetches, 6 KB
    // instead of 32 KB per 640x480 frame.
    const int nthreads = static_cast<int>(blockDim.x);
    const int hdr_v = 192 >> 4;
    const int q_v0 = static_cast<int>(lay.queue >> 4), q_v1 = static_cast<int>((lay.queue + 2u * static_cast<uint32_t>(C) + 15u) >> 4);
    auto dma = [&](int v0, int v1) {  // vectors [v0, v1) of the image, one 64-vector chunk per wave and pass
      for (int base = v0 + (tid - lane); base < v1; base += nthreads)  // (tid - lane: first vector of this wave's chunk)
        if (base + lane < v1)
          __builtin_amdgcn_global_load_lds((dpx_glb_ptr)(img + base + lane), (dpx_lds_ptr)(lds_v + base), 16, 0, 0);
    };
    if ((PHASE == 3 || PHASE == 5) && !GST) {
      dma(0, hdr_v);
      dma(q_v0, q_v1 < img_vec ? q_v1 : img_vec);
    } else {
      dma(0, img_vec);
    }
    __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0)
    __syncthreads();
  }
  if constexpr (PHASE == 4) {
    region_prep_body<true>(prm, rb, rx, smem, frame, t_res);
    if (wid != 0) return;  // (the grow phase is wave 0's; the others give their registers and wave slots back)
  }
  if (PHASE <= 1) {
#include "dpx_regions2_prep.inc"
  }  // PHASE <= 1
  if (PHASE == 1) {
    for (int i = tid; i < img_vec; i += static_cast<int>(blockDim.x)) img[i] = lds_v[i];
    if (rx.res && tid == 0) rx.res[frame * 4 + 0] = static_cast<unsigned long long>(__builtin_amdgcn_s_memtime()) - t_res;
    return;
  }
  const int planar_total = hdr[0];
  t_p0 = DPX_STAMP();

#include "dpx_regions2_grow.inc"
  __syncthreads();
  if (PHASE == 2 || PHASE == 4) {
    if (!GST) {  // (what the tail reads: header and queue — see the image load above)
      const int q_v0 = static_cast<int>(lay.queue >> 4), q_v1 = static_cast<int>((lay.queue + 2u * static_cast<uint32_t>(C) + 15u) >> 4);
      const int nthr = PHASE == 4 ? 64 : static_cast<int>(blockDim.x);  // (fused launch: only wave 0 is left)
      for (int i = tid; i < (192 >> 4); i += nthr) img[i] = lds_v[i];
      for (int i = q_v0 + tid; i < q_v1 && i < img_vec; i += nthr) img[i] = lds_v[i];
    } else {
      for (int i = tid; i < img_vec; i += static_cast<int>(blockDim.x)) img[i] = lds_v[i];
    }
    if (rx.res && tid == 0) {
      rx.res[frame * 4 + 1] = static_cast<unsigned long long>(__builtin_amdgcn_s_memtime()) - t_res;
      rx.res[frame * 4 + 3] = static_cast<unsigned long long>(__builtin_amdgcn_s_memrealtime()) - t_real;  // 100 MHz ticks
    }
    return;
  }
  const int activations = hdr[1], ngrow = hdr[2], ncand = hdr[3];
  t_p1 = DPX_STAMP();

  // `order` is dead after the seed loop: it becomes the uint16 label map of P4/P5 (labels_map_)
  idx_t* lab16 = order;
  for (int c = tid; c < C; c += kRT) lab16[c] = 0;
#include "dpx_regions2_tail.inc"
#include "dpx_regions2_merge.inc"
  if (tid == 0) {
    dpx_frame_stats st;
    st.planar_cells = planar_total;
    st.grows = ngrow;
    st.activations = activations;
    st.segments = nplanes;
    st.merges = merges;
    st.overflow = 0;
    rb.stats[frame] = st;
    if ((PHASE == 3 || PHASE == 5) && rx.res) rx.res[frame * 4 + 2] = static_cast<unsigned long long>(__builtin_amdgcn_s_memtime()) - t_res;
    if (DBG && rx.dbg) {
      unsigned long long* d = rx.dbg + static_cast<size_t>(frame) * kDbgSlots;
      const unsigned long long t_end = DPX_STAMP();
      d[0] = t_p0 - t_start; d[1] = t_p1 - t_p0; d[2] = t_p2 - t_p1; d[3] = t_p3 - t_p2; d[4] = t_p4 - t_p3;
      d[5] = t_p5 - t_p4; d[6] = t_end - t_p5; d[7] = t_end - t_start;
      d[8] = a_argmax; d[9] = a_seed; d[10] = a_bfs; d[11] = n_rounds | (n_spec << 32); d[12] = n_steps; d[13] = ngrow; d[14] = ncand; d[15] = activations;
      d[16] = a_scan; d[17] = a_spec; d[18] = n_scan1; d[19] = n_scan64; d[20] = n_pops; d[21] = n_stop_inc; d[22] = n_stop_bin; d[23] = n_stop_cell; d[24] = a_claims; d[25] = a_commit; d[26] = n_fast; d[27] = a_fast; d[28] = n_gather; d[29] = n_itemover; d[30] = n_cont; d[31] = a_walk;
    }
  }
}
#undef DPX_STAMP

}  // namespace dpx
#include "dpx_regions2_prep.h"
