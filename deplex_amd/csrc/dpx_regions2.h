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
    // instead of 32 KB per 640x480 frame.  (Letting the grow launch fill the queue positions itself instead of
    // fetching 2 C bytes of 0xFF made its wave 9 k cycles slower, for reasons not understood; not done.)
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
  // ---------------- P0: state into LDS, histogram, bin-grouped order ------------------------------------
  for (int b = tid; b < nb; b += kRT) hist[b] = 0;
  if (tid < 48) hdr[tid] = 0;
  __syncthreads();
  {
    int planar_cnt = 0;
    for (int c0 = 0; c0 < C; c0 += kRT) {  // (uniform trip count: the key compaction below uses wave ballots)
      const int c = c0 + tid < C ? c0 + tid : C - 1;
      const bool live = c0 + tid < C;
      uint32_t fl, bin1;
      unsigned long long fkey = ~0ull;
      if (FULL) {
        // fused K3 (k_cellgraph): planar bit, histogram bin and the four IN-edges of the growSeed predicate
        // (plane_extractor.cpp:377-383) straight from the plane-fit records; no meta/mse round trip through HBM
        const float4* rc = reinterpret_cast<const float4*>(cells + c);
        const float4 q2 = rc[2], q3 = rc[3];  // {pp5, n0, n1, n2}, {d, mse, tol, info}
        const uint32_t info = __float_as_uint(q3.w);
        fl = (info & CELL_PLANAR) ? F_UNASSIGNED : 0u;
        bin1 = (info & CELL_PLANAR) ? (info >> 8) : 0u;
        if (fl) {
          const float4 q0 = rc[0];
          const float nf = static_cast<float>(prm.n);
          const float smx = q0.x / nf, smy = q0.y / nf, smz = q0.z / nf;  // mean_ = coord_sum_ / nr_pts_
          const double tol_self = static_cast<double>(q3.z);
          const int cr = c / nh, cc = c - cr * nh;
          const int nbr[4] = {cr >= 1 ? c - nh : -1, cr + 1 < nv ? c + nh : -1, cc >= 1 ? c - 1 : -1, cc + 1 < nh ? c + 1 : -1};
#pragma unroll
          for (int dir = 0; dir < 4; ++dir) {
            const int v = nbr[dir];
            if (v < 0) continue;
            const float4* rv = reinterpret_cast<const float4*>(cells + v);
            const float4 v2 = rv[2], v3 = rv[3];
            if (!(__float_as_uint(v3.w) & CELL_PLANAR)) continue;
            const float cosf_ = dot3(q2.y, q2.z, q2.w, v2.y, v2.z, v2.w);  // same products/sum order as n_v . n_c
            if (!(static_cast<double>(cosf_) >= static_cast<double>(prm.min_cos))) continue;
            const double ti = static_cast<double>(dot3(v2.y, v2.z, v2.w, smx, smy, smz)) + static_cast<double>(v3.x);
            if (ti * ti <= tol_self) fl |= (F_EDGE0 << dir);
          }
        }
        binof[c] = static_cast<uint16_t>(bin1);
        // sort key of the seed lists (dpx_binsort.h): bin, then MSE, then cell id; cells without a bin sort last
        const int kb = static_cast<int>(bin1) - 1;
        fkey = (fl && kb >= 0 && kb < nb)
                   ? (static_cast<unsigned long long>(kb) << kKeyBinShift) |
                         (static_cast<unsigned long long>(sort_mse_key(q3.y)) << kKeyCellBits) | static_cast<unsigned long long>(c)
                   : ~0ull;
      } else {
        const uint32_t m = meta[c];
        // LDS flag byte: bit 0 unassigned, bits 2..5 IN-edges (neighbour U/D/L/R of this cell may activate it)
        fl = (m & F_UNASSIGNED) | (((m >> 8) & 0xFu) << 2);
        bin1 = m >> META_BIN_SHIFT;
        if constexpr (HASBIN) binof[c] = static_cast<uint16_t>(bin1);
      }
      if (live) {
        flags[c] = static_cast<uint8_t>(fl);
        qpos[c] = static_cast<idx_t>(kNoPos);
        if (fl & F_UNASSIGNED) {
          ++planar_cnt;
          const int b = static_cast<int>(bin1) - 1;
          if (FULL && b >= 0 && b < nb) atomicAdd(&hist[b], 1);
        }
      }
      if constexpr (FULL) {  // compact the valid keys (any order: they are sorted next), one LDS atomic per wave
        const bool has = live && fkey != ~0ull;
        const unsigned long long hm = __ballot(has);
        int base = 0;
        if (lane == 0 && hm) base = atomicAdd(&hdr[6], __popcll(hm));
        base = __shfl(base, 0);
        if (has) skeys[base + __popcll(hm & lt_mask)] = fkey;
      }
    }
    if constexpr (!FULL) {  // bin counts and list bounds come from the sorted seed lists
      const uint32_t* bo = rx.bin_off + static_cast<size_t>(frame) * (nb + 1);
      for (int b = tid; b < nb; b += kRT) {
        const uint32_t s0 = bo[b], s1 = bo[b + 1];
        hist[b] = static_cast<int>(s1 - s0);
        bin_start[b] = static_cast<idx_t>(s0);
        cursor[b] = static_cast<int>(s1);
      }
      if constexpr (!GST) {
        const uint32_t* go = rb.order + static_cast<size_t>(frame) * C;
        const int listed = static_cast<int>(bo[nb]);
        for (int i = tid; i < listed; i += kRT) order[i] = static_cast<idx_t>(go[i]);
      }
    }
    planar_cnt = wave_sum(planar_cnt);
    if (lane == 0) atomicAdd(&hdr[0], planar_cnt);
  }
  __syncthreads();
  if (wid == 0) {
    int carry = 0;
    for (int base = 0; base < nb; base += 64) {
      const int b = base + lane;
      const int v = b < nb ? hist[b] : 0;
      const int incl = wave_incl_scan(v, lane);
      if (FULL && b < nb) {
        bin_start[b] = static_cast<idx_t>(carry + incl - v);
        cursor[b] = carry + incl;
      }
      carry += __shfl(incl, 63);
      // compact list of non-empty bins, ascending
      const unsigned long long nzm = __ballot(v > 0);
      if (v > 0) nzlist[hdr[5] + __popcll(nzm & lt_mask)] = static_cast<uint16_t>(b);
      if (lane == 0) hdr[5] += __popcll(nzm);
      wave_sync();
    }
    if (lane == 0) bin_start[nb] = static_cast<idx_t>(carry);
  }
  __syncthreads();
  if constexpr (FULL) {
    // seed lists: bitonic sort of the keys in LDS (all threads), then the cell ids in sorted order.  Sorted by bin
    // first, the list of bin b occupies exactly [bin_start[b], cursor[b]).
    const int nkeys = hdr[6];
    int np2 = 2;
    while (np2 < nkeys) np2 <<= 1;  // <= rx.sort_pad
    for (int i = nkeys + tid; i < np2; i += kRT) skeys[i] = ~0ull;
    __syncthreads();
    // Strides >= 64 are workgroup-wide compare-exchange passes; the strides below 64 of a stage stay inside aligned
    // 128-key chunks, so a wave finishes them chunk by chunk on its own (one pair per lane, no workgroup barrier).
    for (int k = 2; k <= np2; k <<= 1) {
      for (int j = k >> 1; j >= 64; j >>= 1) {
        for (int pi = tid; pi < (np2 >> 1); pi += kRT) {
          const int lo = ((pi & ~(j - 1)) << 1) | (pi & (j - 1));
          const int hi = lo | j;
          const unsigned long long a = skeys[lo], b = skeys[hi];
          if ((a > b) == ((lo & k) == 0)) {
            skeys[lo] = b;
            skeys[hi] = a;
          }
        }
        __syncthreads();
      }
      const int j0 = (k >> 1) < 32 ? (k >> 1) : 32;
      for (int cb = wid * 128; cb < np2; cb += kRW * 128) {
        for (int j = j0; j >= 1; j >>= 1) {
          const int lo = cb + (((lane & ~(j - 1)) << 1) | (lane & (j - 1)));
          const int hi = lo | j;
          if (hi < np2) {  // (np2 < 128 only for tiny frames)
            const unsigned long long a = skeys[lo], b = skeys[hi];
            if ((a > b) == ((lo & k) == 0)) {
              skeys[lo] = b;
              skeys[hi] = a;
            }
          }
          wave_sync();
        }
      }
      __syncthreads();
    }
    for (int i = tid; i < nkeys; i += kRT) order[i] = static_cast<idx_t>(skeys[i] & ((1u << kKeyCellBits) - 1u));
    __syncthreads();
  }
  }  // PHASE <= 1
  if (PHASE == 1) {
    for (int i = tid; i < img_vec; i += static_cast<int>(blockDim.x)) img[i] = lds_v[i];
    if (rx.res && tid == 0) rx.res[frame * 4 + 0] = static_cast<unsigned long long>(__builtin_amdgcn_s_memtime()) - t_res;
    return;
  }
  const int planar_total = hdr[0];
  t_p0 = DPX_STAMP();

  // ---------------- P1: seeds + region growing, wave 0 only ---------------------------------------------
  // cursor[b] holds the END of bin b's sorted cell list; bin_start[b] the position of its first possibly live entry.
  if ((PHASE == 0 || PHASE == 2 || PHASE == 4) && (GST || wid == 0)) {
    // Tier 0 keeps all waves in this loop: wave 0 runs the grows exactly like the other tiers, the rest wait at the
    // per-grow barrier and join when a BFS queue gets wide (>= kWideSlots pending cells): then every wave takes 16
    // consecutive queue slots of a batch, appends in (wave, lane) order through per-wave counts, and the FIFO order
    // stays the reference's.  All control decisions are read from LDS after a barrier, so they are workgroup-uniform.
    constexpr int kWideSlots = PHASE == 2 ? (1 << 30) : 64;  // (one-wave grow launch: never)
    int remaining = planar_total;
    int tail = 0, ngrow = 0, ncand = 0;
    const int nnz = hdr[5];
    // up to four non-empty bins per lane (bins nzlist[lane + 64 q], ascending in q then lane), kept in registers for the
    // whole frame; chunks q >= 4 of frames with more than 256 non-empty bins are read from the list
    int mybin[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) mybin[q] = (lane + 64 * q < nnz) ? static_cast<int>(nzlist[lane + 64 * q]) : -1;
    const int nsets = (nnz + 63) >> 6;
    const bool reg_bins = nnz <= 256;  // (tier-2 path: otherwise the arg-max walks the list in LDS)
    const int reg_sets = nsets;
    // grow-phase scratch (r2_layout): the cached seed of every bin and the claim table of a sweep round
    constexpr uint32_t kCandNone = 0xFFFFFFFFu, kCandBig = 0x80000000u;
    constexpr bool LEAN = GST && PHASE == 2;  // one-wave tier-0 grow launch: small scratch, several frames per CU
    constexpr int kClaimBits = LEAN ? kLeanClaimBits : claim_bits(TIER);
    constexpr int kRingBitsL = LEAN ? kLeanRingBits : kRingBits;
    constexpr int kSpecMax = 8;  // cells of a speculative grow: one relative cell code per byte of a 64-bit register pair
    // Rounds with fewer events take the first one on its own (see there): a sweep round costs about what four single
    // events cost, and on average 60 % of a round's events are committed before it stops.
#ifndef DPX_SWEEP_MIN
#define DPX_SWEEP_MIN 6
#endif
    constexpr int kSweepMinEvents = DPX_SWEEP_MIN;
    uint32_t* candc = reinterpret_cast<uint32_t*>(smem + lay.grow_x);
    uint32_t* bclaim = candc + nb;
    uint32_t* ctab = bclaim + nb;
    uint16_t* evb = reinterpret_cast<uint16_t*>(ctab + (1 << kClaimBits));
    // round staging behind the event bins: a 16-byte record per lane (list, seed, first item), per-lane results, the
    // source lane of every item
    uint4* rrec = reinterpret_cast<uint4*>(smem + ((static_cast<uint32_t>(reinterpret_cast<unsigned char*>(evb + 64) - smem) + 15u) & ~15u));
    uint32_t* rres = reinterpret_cast<uint32_t*>(rrec + 64);
    uint8_t* rsrc = reinterpret_cast<uint8_t*>(rres + 64);
    uint32_t* ring = ctab + (1 << kClaimBits) + (kEvBytes >> 2);  // tier 0 only (kEvBytes is a multiple of 4)
    // Tier 0, single launch: wave 1 — idle otherwise while wave 0 grows — keeps looking for the first live entry of every
    // bin's list and publishes (list position << 32 | cell) here; wave 0 starts its own search from there.  Sound because
    // an entry that is dead stays dead: "everything in front of this position is dead" never stops being true, and wave 0
    // checks the published cell's flag itself before taking it.  One 8-byte word per bin, written by wave 1 only.
    constexpr bool HELP = GST && PHASE == 0;
    constexpr uint32_t kHeadNone = 0xFFFFFFFFu;
    unsigned long long* headpc = reinterpret_cast<unsigned long long*>(ring + (1 << kRingBitsL));
    // (one 8-byte LDS access each — atomics, so that the compiler cannot split them: a position from one publication with
    // the cell of another would let wave 0 step over an entry nobody has looked at)
    auto head_load = [](unsigned long long* a) { return __hip_atomic_load(a, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); };
    auto head_store = [](unsigned long long* a, unsigned long long v) { __hip_atomic_store(a, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); };
    constexpr uint32_t kRingMask = (1u << kRingBitsL) - 1u;
    auto claim_slot = [&](uint32_t k) { return (k * 2654435761u) >> (32 - kClaimBits); };
    if (TIER != 2 && wid == 0) {
      for (int i = lane; i < nb; i += 64) {
        candc[i] = kCandNone;
        bclaim[i] = 0xFFFFFFFFu;
      }
      for (int i = lane; i < (1 << kClaimBits); i += 64) ctab[i] = 0xFFFFFFFFu;
      if constexpr (HELP)
        for (int i = lane; i < nb; i += 64) headpc[i] = static_cast<unsigned long long>(kHeadNone);  // position 0, no cell
      wave_sync();
    }
    if constexpr (HELP) __syncthreads();  // (every wave of a tier-0 workgroup is here: wave 1 reads what wave 0 just set up)
    // BFS lane constants: lane = (slot, dir); dir 0..3 = U, D, L, R
    const int slot = lane >> 2, dir = lane & 3;
    const int delta = dir == 0 ? -nh : (dir == 1 ? nh : (dir == 2 ? -1 : 1));
    const uint32_t opp = static_cast<uint32_t>(dir ^ 1);
    const uint32_t notme = ~(1u << opp) & 0xFu;
    int guard = 0;  // grows started (wave 0's count; the reference's loop ends by itself, this is a safety net)
    int cont_b = -1, cont_level = 0;  // the event list of the last arg-max, see "An unfinished sweep" in the round
    int head = 0, gstart = 0;
    bool in_grow = false;
    unsigned long long ta = 0, tb = 0, tc = 0;
    // segment bookkeeping of a finished grow (wave 0)
    auto finish_grow = [&]() {
      const unsigned long long td = DPX_STAMP();
      const int g = tail - gstart;
      const bool is_cand = !(static_cast<unsigned long long>(g) < static_cast<unsigned long long>(static_cast<long long>(prm.min_cells))) &&
                           ncand < rb.pcap;
      if (lane == 0) {
        grow_start[ngrow] = static_cast<uint32_t>(gstart);
        // (the tail reads the segment's queue range from here with one load instead of chasing grow_start twice)
        if (is_cand) cand_grow[ncand] = make_uint4(static_cast<uint32_t>(ngrow), static_cast<uint32_t>(gstart), static_cast<uint32_t>(tail), 0u);
      }
      ncand += is_cand;
      ++ngrow;
      remaining -= g;
      if (DBG) {
        a_argmax += tb - ta; a_seed += tc - tb; a_bfs += td - tc;
      }
    };
    for (int round = 0;; ++round) {
      int st = 2;  // 2: no further grow, 0: wave 0 finished this grow's BFS (or committed it), 1: wide BFS wanted
      if (HELP && wid == 1) {
        // The helper: chunk after chunk of the bin list, one bin per lane — exactly wave 0's seed search (four entries per
        // lane, then up to two 64-entry steps for lanes that found nothing) — until wave 0 announces the round's barrier
        // in hdr[42].  It reads bin_start / hist / flags while wave 0 changes them: a stale value only makes it look at
        // entries that are dead by now, never skip a live one.
        volatile int* req = hdr + 42;
        for (int spin = 0; *req != round + 1 && spin < (1 << 16); ++spin) {  // (bounded: a helper that stops helping is harmless)
          for (int q = 0; q < nsets && *req != round + 1; ++q) {
            const int t = lane + 64 * q;
            const int hb = t < nnz ? static_cast<int>(nzlist[t]) : -1;
            const bool have = hb >= 0 && hist[hb >= 0 ? hb : 0] > 0;
            const int hbs = have ? hb : 0;
            const unsigned long long pc = head_load(&headpc[hbs]);
            const int ppos = static_cast<int>(pc >> 32);
            const uint32_t pcell = static_cast<uint32_t>(pc);
            const int bst = static_cast<int>(bin_start[hbs]);
            const int hend = cursor[hbs];
            int hpos = ppos > bst ? ppos : bst;
            const bool cached = pcell != kHeadNone && ppos == hpos;
            bool hneed = have && hpos < hend && !(cached && (flags[pcell] & F_UNASSIGNED) != 0u);
            if (hneed && cached) ++hpos;  // the published head died
            uint32_t hfound = kHeadNone;
            if (hneed && hpos < hend) {
              const int i1 = hpos + 1 < hend ? hpos + 1 : hend - 1, i2 = hpos + 2 < hend ? hpos + 2 : hend - 1, i3 = hpos + 3 < hend ? hpos + 3 : hend - 1;
              const uint32_t e0 = order[hpos], e1 = order[i1], e2 = order[i2], e3 = order[i3];
              const uint32_t f0 = flags[e0], f1 = flags[e1], f2 = flags[e2], f3 = flags[e3];
              if (f0 & F_UNASSIGNED) hfound = e0;
              else if ((f1 & F_UNASSIGNED) && hpos + 1 < hend) { hfound = e1; hpos += 1; }
              else if ((f2 & F_UNASSIGNED) && hpos + 2 < hend) { hfound = e2; hpos += 2; }
              else if ((f3 & F_UNASSIGNED) && hpos + 3 < hend) { hfound = e3; hpos += 3; }
              else hpos = hpos + 4 < hend ? hpos + 4 : hend;
            }
            if (hneed) head_store(&headpc[hbs], (static_cast<unsigned long long>(static_cast<uint32_t>(hpos)) << 32) | hfound);
            unsigned long long hm = __ballot(hneed && hfound == kHeadNone && hpos < hend);
            for (int it = 0; it < 2 && hm; ++it) {
              const int l = static_cast<int>(__builtin_ctzll(hm));
              hm &= hm - 1ull;
              const int p = __builtin_amdgcn_readlane(hpos, l), e = __builtin_amdgcn_readlane(hend, l), lb = __builtin_amdgcn_readlane(hbs, l);
              const int i = p + lane;
              const uint32_t ce = order[i < e ? i : e - 1];
              const bool alive = i < e && (flags[ce] & F_UNASSIGNED) != 0u;
              const unsigned long long am = __ballot(alive);
              int np = p + 64 < e ? p + 64 : e;
              uint32_t nf = kHeadNone;
              if (am) {
                const int first = static_cast<int>(__builtin_ctzll(am));
                np = p + first;
                nf = static_cast<uint32_t>(__builtin_amdgcn_readlane(static_cast<int>(ce), first));
              }
              if (lane == 0) head_store(&headpc[lb], (static_cast<unsigned long long>(static_cast<uint32_t>(np)) << 32) | nf);
            }
          }
          __builtin_amdgcn_s_sleep(8);
        }
      }
      if (!GST || wid == 0) {
        // wave 0 runs grow after grow on its own and comes to the barrier below only to ask for a wide BFS or to end
        for (;; ++guard) {
        st = 2;
        in_grow = false;
        do {
          if (!(remaining > 0 && guard <= C)) break;
          ta = DPX_STAMP();
          // ---- one SWEEP ROUND ------------------------------------------------------------------------------------
          // The reference takes the most frequent bin (lowest index among maxima), grows from the first live entry of its
          // sorted list, removes the grown cells from the histogram and repeats (plane_extractor.cpp:297-347,
          // normals_histogram.cpp:51-72).  While the maximum count is L, the bins that hold exactly L cells fire one after
          // the other in ascending bin order ("sweep"): a bin that has fired is at L - 1 and cannot fire again before all
          // others at L have, and a bin only ever loses cells.  A bin whose turn comes with its count still at L has lost
          // no cell since the sweep began, so its seed is the first live entry of its list as of the START of the sweep;
          // and its grow is the grow computed on the mask as of the start of the sweep unless an earlier grow of the sweep
          // activated one of its cells (growSeed's accept test reads only static per-cell data besides the unassigned
          // mask, so removing cells OUTSIDE a grow's result changes neither the result nor its order).
          // Hence a round: the first (up to) 64 bins at level L in ascending order, one lane each.
          // Every lane grows its seed on the current mask by itself, up to kSpecMax cells (on noisy large frames two thirds
          // of all grows are a single cell and 98 % have fewer than 16); all cells and the bins of the non-seed cells are
          // claimed in a small hash table with atomic min(lane).  Lane t is CLEAN if its grow is complete, no lower lane
          // claimed one of its cells (or collides with them in the table: conservative) and no lower lane claimed a cell
          // of its bin (its count would not be L at its turn).  The clean prefix of lanes is exactly what the sequential
          // loop would have done, in this order, and is committed at once: queue ranges by prefix sum, flags, histogram,
          // grow / candidate-segment records.  The first lane that is not clean waits for the next round; if it is the
          // first event of a round it can only be an incomplete (large) grow, which takes the wave-wide BFS below.
          uint32_t seed = 0xFFFFFFFFu;
          int bb = 0;
          if constexpr (TIER == 2) {
            // Small frames (<= ~8 k cells: 640x480) keep the plain sequence — arg-max, first live entry of the bin's sorted
            // list with its isolated-seed test, grow — one event per iteration: frames with a few large planes (the TUM
            // stream of the bench) hardly ever have several bins at the same level, a sweep round costs about what four
            // to six single events cost, and the one-wave grow launch of the pipeline must stay small in registers.
            // most frequent bin, lowest index among maxima: key = count << kBinBits | (kBinMask - bin)
            constexpr int kBinBits = GST ? 13 : 16;  // tier 0: counts up to 2^19, bins < 2^13
            constexpr uint32_t kBinMask = (1u << kBinBits) - 1u;
            uint32_t key = 0;
            if (reg_bins) {
              if (mybin[0] >= 0) key = (static_cast<uint32_t>(hist[mybin[0]]) << kBinBits) | (kBinMask - static_cast<uint32_t>(mybin[0]));
              if (reg_sets > 1) {  // uniform
      #pragma unroll
                for (int q = 1; q < 4; ++q) {
                  const uint32_t k = mybin[q] >= 0 ? (static_cast<uint32_t>(hist[mybin[q]]) << kBinBits) | (kBinMask - static_cast<uint32_t>(mybin[q])) : 0u;
                  key = k > key ? k : key;
                }
              }
            } else {
              for (int t = lane; t < nnz; t += 64) {
                const int b = nzlist[t];
                const uint32_t k = (static_cast<uint32_t>(hist[b]) << kBinBits) | (kBinMask - static_cast<uint32_t>(b));
                key = k > key ? k : key;
              }
            }
            key = wave_max_u32(key);
            const int bv = static_cast<int>(key >> kBinBits);
            bb = static_cast<int>(kBinMask - (key & kBinMask));
            if (bv <= 0) break;
            if (static_cast<unsigned long long>(bv) < static_cast<unsigned long long>(static_cast<long long>(prm.min_cand))) break;
            tb = DPX_STAMP();
            {
              // the bin's list is sorted by (mse, cell id): the seed is its first live entry.  bin_start[bb] only moves
              // forward (a grown cell never comes back), so every list entry is skipped at most once per frame.
              //
              // Isolated seeds: on noisy frames two thirds of all grows activate nothing but their seed (no neighbour is both
              // unassigned and reachable from it).  The chunk scan reads the seed's four neighbour flags along with its own,
              // and such a grow is committed right here — one queue slot, one grow — without a BFS step.  (Committing several
              // in a row does not pay: the arg-max keeps the fullest bins level, so it changes bin after almost every grow.)
              int s0 = static_cast<int>(bin_start[bb]);
              const int e = cursor[bb];
              const bool iso_ok = prm.min_cells > 1 || prm.min_cells < 0;  // a one-cell grow is never a segment candidate
              bool committed = false;
              while (s0 < e) {
                const int i = s0 + lane;
                uint32_t c = 0, f = 0;
                bool alive = false, iso = false;
                if (i < e) {
                  c = order[i];
                  const int cc = static_cast<int>(c);
                  // (not clamped: the flag bytes are in LDS in every tier, a read outside the grid returns bytes of the
                  // neighbouring arrays and can only make the cell look connected — see below)
                  const int wU = cc - nh, wD = cc + nh, wL = cc - 1, wR = cc + 1;
                  f = flags[cc];
                  const uint32_t fU = flags[wU], fD = flags[wD], fL = flags[wL], fR = flags[wR];
                  alive = (f & F_UNASSIGNED) != 0;
                  // a neighbour that is unassigned and whose in-edge bit for this cell's side is set can be activated from
                  // here; non-existent neighbours can only make a cell look connected, never isolated
                  const uint32_t out = ((fU & (fU >> 3)) | (fD & (fD >> 2)) | (fL & (fL >> 5)) | (fR & (fR >> 4))) & 1u;
                  iso = alive && out == 0u && iso_ok;
                }
                const unsigned long long am = __ballot(alive);
                if (am) {
                  const int first = __builtin_ctzll(am);
                  seed = static_cast<uint32_t>(__builtin_amdgcn_readlane(static_cast<int>(c), first));
                  committed = ((__ballot(iso) >> first) & 1ull) != 0ull;
                  if (committed && lane == first) {
                    queue[tail] = static_cast<idx_t>(c);
                    qpos[c] = static_cast<idx_t>(tail);
                    flags[c] = static_cast<uint8_t>(f & ~F_UNASSIGNED);
                    grow_start[ngrow] = static_cast<uint32_t>(tail);
                    atomicSub(&hist[bb], 1);
                  }
                  s0 += first + 1;
                  break;
                }
                s0 += 64;
              }
              if (lane == 0) bin_start[bb] = static_cast<idx_t>(s0 < e ? s0 : e);
              if (committed) {
                ++tail;
                ++ngrow;
                --remaining;
                wave_sync();
                if (DBG) {
                  const unsigned long long tc2 = DPX_STAMP();
                  a_argmax += tb - ta; a_seed += tc2 - tb;
                }
                st = 0;
                break;
              }
            }
            if (seed == 0xFFFFFFFFu) break;
          } else {
            if (DBG) ++n_rounds;
            // 0. An unfinished sweep goes on without an arg-max: the events of the previous round (cont_b, one bin per lane,
            // at level cont_level) that neither fired nor lost a cell are still at that level — exactly the bins whose count
            // still equals it, since a bin only ever loses cells — no bin can have risen above it, and every bin at that
            // level that the previous list did not hold (it holds at most 64) has a higher index than all it holds.  So the
            // remaining listed bins, in lane order, are the next events; only when none is left does the maximum have to be
            // found again.
            int b = -1, bv = 0, nev = 0;
            bool part = false;
            unsigned long long partmask = 0ull;
            if (cont_level > 0) {  // (uniform)
              part = cont_b >= 0 && hist[cont_b >= 0 ? cont_b : 0] == cont_level;
              partmask = __ballot(part);
              if (partmask) {
                b = part ? cont_b : -1;
                bv = cont_level;
                nev = static_cast<int>(__popcll(partmask));
                if (DBG) ++n_cont;
              }
            }
            if (!partmask) {
            // 1. level of the most frequent bin, and the first 64-bin chunk that holds a bin at that level
            uint32_t key = 0;
            uint32_t cntq[4] = {0u, 0u, 0u, 0u};  // counts of this lane's bins (chunks 0..3)
  #pragma unroll
            for (int q = 0; q < 4; ++q)
              if (q < nsets) {  // (uniform)
                cntq[q] = mybin[q] >= 0 ? static_cast<uint32_t>(hist[mybin[q]]) : 0u;
                const uint32_t k = (cntq[q] << 8) | static_cast<uint32_t>(255 - q);
                key = k > key ? k : key;
              }
            for (int q = 4; q < nsets; ++q) {
              const int t = lane + 64 * q;
              const uint32_t cnt = t < nnz ? static_cast<uint32_t>(hist[nzlist[t]]) : 0u;
              const uint32_t k = (cnt << 8) | static_cast<uint32_t>(255 - (q < 255 ? q : 255));
              key = k > key ? k : key;
            }
            key = wave_max_u32(key);
            bv = static_cast<int>(key >> 8);
            int qc = 255 - static_cast<int>(key & 255u);
            if (bv <= 0) break;
            if (static_cast<unsigned long long>(bv) < static_cast<unsigned long long>(static_cast<long long>(prm.min_cand))) break;
            if (qc >= 255)  // (more than 255 chunks: find the chunk by scanning; never with the default 400 bins)
              for (qc = 255; qc < nsets; ++qc) {
                const int t = lane + 64 * qc;
                if (__ballot(t < nnz && hist[nzlist[t]] == bv)) break;
              }
            // 2. the events of this round: bins at level bv, ascending = lane order; those of chunk qc first
            if (qc < 4) {
              b = mybin[0];
              b = qc == 1 ? mybin[1] : b;
              b = qc == 2 ? mybin[2] : b;
              b = qc == 3 ? mybin[3] : b;
              uint32_t cs_ = cntq[0];
              cs_ = qc == 1 ? cntq[1] : cs_;
              cs_ = qc == 2 ? cntq[2] : cs_;
              cs_ = qc == 3 ? cntq[3] : cs_;
              part = b >= 0 && cs_ == static_cast<uint32_t>(bv);
            } else {
              const int t = lane + 64 * qc;
              b = t < nnz ? static_cast<int>(nzlist[t]) : -1;
              part = b >= 0 && hist[b >= 0 ? b : 0] == bv;
            }
            partmask = __ballot(part);
            nev = static_cast<int>(__popcll(partmask));
            // The sweep does not end with the chunk: the bins at level bv of the LATER chunks fire right after these, in the
            // same ascending order.  They are appended (through a 64-entry staging array: event e = lane e) until the round
            // is full — a frame with 300 non-empty bins spreads the 20 events of a typical sweep over five chunks, which
            // would otherwise be five rounds of four events, i.e. twenty single events.
            if (qc + 1 < nsets && nev < 64) {  // (uniform)
              int total = nev;
              bool staged = false;
              auto append = [&](int bq, uint32_t cq) {
                const bool pq = bq >= 0 && cq == static_cast<uint32_t>(bv);
                const unsigned long long mq = __ballot(pq);
                if (mq) {
                  if (!staged) {
                    if (part) evb[__popcll(partmask & lt_mask)] = static_cast<uint16_t>(b);
                    staged = true;
                  }
                  const int at = total + static_cast<int>(__popcll(mq & lt_mask));
                  if (pq && at < 64) evb[at] = static_cast<uint16_t>(bq);
                  total += static_cast<int>(__popcll(mq));
                }
              };
  #pragma unroll
              for (int q = 1; q < 4; ++q)
                if (q > qc && q < nsets && total < 64) append(mybin[q], cntq[q]);  // (uniform)
              for (int q = qc + 1 > 4 ? qc + 1 : 4; q < nsets && total < 64; ++q) {
                const int t = lane + 64 * q;
                const int bq = t < nnz ? static_cast<int>(nzlist[t]) : -1;
                append(bq, bq >= 0 ? static_cast<uint32_t>(hist[bq]) : 0u);
              }
              if (staged) {
                if (DBG) ++n_gather;
                asm volatile("" ::: "memory");  // (LDS executes a wave's operations in order)
                __builtin_amdgcn_wave_barrier();
                nev = total < 64 ? total : 64;
                part = lane < nev;
                b = part ? static_cast<int>(evb[lane]) : -1;
                partmask = nev == 64 ? ~0ull : ((1ull << nev) - 1ull);
              }
            }
            cont_b = part ? b : -1;
            cont_level = bv;
            }  // (arg-max)
            const int bs = b >= 0 ? b : 0;
            tb = DPX_STAMP();
            if (nev < kSweepMinEvents) {
              // A few events only (frames with a few large planes spend most rounds here: the fullest bins differ in count).
              // Too little to run in parallel for what a sweep round costs, so only the first event is taken, without
              // speculation: the wave looks at the bin's list 64 entries at a time, each lane reading its entry's flag byte
              // and the four neighbours' — the first live entry is the seed, and if no neighbour can be activated from it
              // the grow is that single cell and is committed on the spot.
              if (DBG) ++n_fast;
              const int l1 = static_cast<int>(__builtin_ctzll(partmask));
              bb = __builtin_amdgcn_readlane(b, l1);
              int s0 = static_cast<int>(bin_start[bb]);  // (same address in every lane: one LDS broadcast)
              if constexpr (HELP) {
                const int hp = static_cast<int>(head_load(&headpc[bb]) >> 32);  // (the helper's position: everything in front of it is dead)
                s0 = hp > s0 ? hp : s0;
              }
              const int e1 = cursor[bb];
              bool iso1 = false;
              while (s0 < e1) {
                const int i = s0 + lane;
                uint32_t ce = 0, f = 0;
                bool alive = false, iso = false;
                if (i < e1) {
                  ce = order[i];
                  const int cc = static_cast<int>(ce);
                  f = flags[cc];
                  const uint32_t fU = flags[cc - nh], fD = flags[cc + nh], fL = flags[cc - 1], fR = flags[cc + 1];
                  alive = (f & F_UNASSIGNED) != 0u;
                  // (a read outside the grid returns bytes of the neighbouring arrays: it can only make the cell look connected)
                  iso = alive && (((fU & (fU >> 3)) | (fD & (fD >> 2)) | (fL & (fL >> 5)) | (fR & (fR >> 4))) & 1u) == 0u;
                }
                const unsigned long long am = __ballot(alive);
                if (am) {
                  const int first = static_cast<int>(__builtin_ctzll(am));
                  seed = static_cast<uint32_t>(__builtin_amdgcn_readlane(static_cast<int>(ce), first));
                  iso1 = ((__ballot(iso) >> first) & 1ull) != 0ull;
                  s0 += first + 1;
                  break;
                }
                s0 += 64;
              }
              if (seed == 0xFFFFFFFFu) break;  // (cannot happen: the bin's count is the number of live entries of its list)
              if (lane == 0) {
                bin_start[bb] = static_cast<idx_t>(s0);
                candc[bb] = kCandNone;
              }
              // a one-cell grow is a segment candidate only for min_cells <= 1: those take the general path below
              if (iso1 && (prm.min_cells > 1 || prm.min_cells < 0)) {
                if (lane == 0) {
                  queue[tail] = static_cast<idx_t>(seed);
                  const int sc = static_cast<int>(seed);
                  atomicAnd(reinterpret_cast<uint32_t*>(flags + (sc & ~3)), ~(F_UNASSIGNED << ((sc & 3) * 8)));
                  atomicSub(&hist[bb], 1);
                  grow_start[ngrow] = static_cast<uint32_t>(tail);
                }
                tail = __builtin_amdgcn_readfirstlane(tail + 1);
                ++ngrow;
                --remaining;
                asm volatile("" ::: "memory");
                __builtin_amdgcn_wave_barrier();
                if (DBG) {
                  const unsigned long long tc2 = DPX_STAMP();
                  a_argmax += tb - ta; a_seed += tc2 - tb; a_fast += tc2 - tb;
                  ++n_spec;
                }
                st = 0;
                break;
              }
              asm volatile("" ::: "memory");
              __builtin_amdgcn_wave_barrier();
              if (DBG) a_fast += DPX_STAMP() - tb;
            } else {
            int pos = static_cast<int>(bin_start[bs]);
            const int lend = cursor[bs];
            const uint32_t cw = candc[bs];
            // 3. seeds: the first live entry of each bin's list (sorted by mse, then cell id).  candc[] caches the verified
            // entry at bin_start[] until it is used or dies; bin_start[] only moves forward.
            uint32_t c = cw == kCandNone ? 0u : (cw & ~kCandBig);
            bool big = cw != kCandNone && (cw & kCandBig) != 0u;  // this seed's grow exceeded kSpecMax cells before
            bool need = part && (cw == kCandNone || (flags[c] & F_UNASSIGNED) == 0u);
            if (need) {
              if (cw != kCandNone) ++pos;  // the cached entry died (activated by another bin's grow)
              big = false;
              if constexpr (HELP) {  // what the helper has found for this bin (see headpc)
                const unsigned long long pc = head_load(&headpc[bs]);
                const int hp = static_cast<int>(pc >> 32);
                const uint32_t hc = static_cast<uint32_t>(pc);
                if (hp >= pos) {
                  pos = hp;  // everything in front of hp is dead
                  if (hc != kHeadNone) {
                    if ((flags[hc] & F_UNASSIGNED) != 0u) {
                      c = hc;
                      need = false;
                    } else {
                      ++pos;
                    }
                  }
                }
              }
            }
            bool broken = false;
            const unsigned long long t_s0 = DPX_STAMP();
            if (__ballot(need)) {
              // one step per lane on its own (four entries: most dead runs are short) ...
              if (need) {
                if (pos >= lend) {  // cannot happen (a bin's count is the number of live entries of its list): safety net
                  broken = true;
                  need = false;
                } else {
                  const int i1 = pos + 1 < lend ? pos + 1 : lend - 1, i2 = pos + 2 < lend ? pos + 2 : lend - 1, i3 = pos + 3 < lend ? pos + 3 : lend - 1;
                  const uint32_t e0 = order[pos], e1 = order[i1], e2 = order[i2], e3 = order[i3];
                  const uint32_t f0 = flags[e0], f1 = flags[e1], f2 = flags[e2], f3 = flags[e3];
                  if (f0 & F_UNASSIGNED) { c = e0; need = false; }
                  else if ((f1 & F_UNASSIGNED) && pos + 1 < lend) { c = e1; pos += 1; need = false; }
                  else if ((f2 & F_UNASSIGNED) && pos + 2 < lend) { c = e2; pos += 2; need = false; }
                  else if ((f3 & F_UNASSIGNED) && pos + 3 < lend) { c = e3; pos += 3; need = false; }
                  else pos += 4;
                }
              }
              if (DBG) ++n_scan1;
              // ... then the wave together, bin after bin, 64 entries per step: behind a large grow the front of a list (its
              // lowest-MSE cells) is dead for hundreds or thousands of entries
              unsigned long long nm = __ballot(need);
              while (nm) {
                const int l = static_cast<int>(__builtin_ctzll(nm));
                nm &= nm - 1ull;
                int p = __builtin_amdgcn_readlane(pos, l);
                const int e = __builtin_amdgcn_readlane(lend, l);
                uint32_t found = 0xFFFFFFFFu;
                while (p < e) {
                  const int i = p + lane;
                  uint32_t ce = 0;
                  bool alive = false;
                  if (i < e) {
                    ce = order[i];
                    alive = (flags[ce] & F_UNASSIGNED) != 0u;
                  }
                  const unsigned long long am = __ballot(alive);
                  if (DBG) ++n_scan64;
                  if (am) {
                    const int first = static_cast<int>(__builtin_ctzll(am));
                    found = static_cast<uint32_t>(__builtin_amdgcn_readlane(static_cast<int>(ce), first));
                    p += first;
                    break;
                  }
                  p += 64;
                }
                if (lane == l) {
                  need = false;
                  pos = p;
                  if (found == 0xFFFFFFFFu) broken = true;
                  else c = found;
                }
              }
            }
            if (DBG) a_scan += DPX_STAMP() - t_s0;
            if (__ballot(broken)) break;
            // 4. speculative grow of every event on the current mask, one lane each: exact FIFO BFS (queue order, then U, D, L,
            // R: plane_extractor.cpp:349-392).  A cell is a relative code ((dr + 8) << 4 | (dc + 8)) in one byte, the grow a
            // list of <= 8 bytes in two registers (entry 0 = the seed, 0x88); a BFS of <= 8 cells stays within 7 rows / columns
            // of its seed, and an empty byte (0) matches no code.  In-edge bits make the L / R steps across a row end and the
            // bounds test the U / D steps across the grid's first / last row fail, as in the wave-wide BFS.  A lone wave is
            // bound by its instruction count, so the accept tests come first and the list work runs only where a lane accepts.
            static_assert(kSpecMax == 8, "the list of a speculative grow is the eight bytes of a 64-bit register pair");
            unsigned long long ll = 0x88ull;
            int n = 1, hpop = 0;
            bool complete = !big;
            bool act = part && !big;
            const int cs = part ? static_cast<int>(c) : 0;
            // How far speculation pays depends on how many events can be committed together: a pop costs about as much as a
            // step of the wave-wide BFS (which takes 16 cells).
#ifndef DPX_MAXPOPS
#define DPX_MAXPOPS kSpecMax
#endif
#ifndef DPX_MAXPOPS_SMALL
#define DPX_MAXPOPS_SMALL 4
#endif
            const int maxpops = nev >= 8 ? DPX_MAXPOPS : DPX_MAXPOPS_SMALL;
            const unsigned long long t_sp0 = DPX_STAMP();
            while (__ballot(act)) {
              if (DBG) ++n_pops;
              const uint32_t code = act ? (static_cast<uint32_t>(ll >> (8 * hpop)) & 0xFFu) : 0x88u;
              const int cr = static_cast<int>(code >> 4), cc = static_cast<int>(code & 15u);
              const int u = cs + (cr - 8) * nh + (cc - 8);
              // (unclamped reads: u is a grid cell, its neighbours' bytes lie inside the flag array or its nh guard bytes)
              const uint32_t fU = flags[u - nh], fD = flags[u + nh], fL = flags[u - 1], fR = flags[u + 1];
              // v accepts u iff v is unassigned and its in-edge from u's side is set (u is v's D / U / R / L neighbour)
              const bool okU = act && u >= nh && ((fU & (fU >> 3)) & 1u) != 0u;
              const bool okD = act && u + nh < C && ((fD & (fD >> 2)) & 1u) != 0u;
              const bool okL = act && u >= 1 && ((fL & (fL >> 5)) & 1u) != 0u;
              const bool okR = act && u + 1 < C && ((fR & (fR >> 4)) & 1u) != 0u;
              // A neighbour's code is code -+ 16 (U / D) or -+ 1 (L / R).  A list entry has both nibbles in 1..15 (at most
              // seven rows / columns from the seed); a neighbour eight away — possible only when the list is full — has a
              // zero nibble in its low byte, matches no entry and no empty byte, and ends the speculation like any other
              // ninth cell.
              auto visit = [&](bool ok, uint32_t vcode) {
                const uint32_t rep = __builtin_amdgcn_perm(0u, vcode, 0u);  // the low byte in all four bytes
                const uint32_t x0 = static_cast<uint32_t>(ll) ^ rep, x1 = static_cast<uint32_t>(ll >> 32) ^ rep;
                const bool dup = ((((x0 - 0x01010101u) & ~x0) | ((x1 - 0x01010101u) & ~x1)) & 0x80808080u) != 0u;
                if (ok && act && !dup) {
                  if (n == kSpecMax) {
                    complete = false;
                    act = false;
                  } else {
                    ll |= static_cast<unsigned long long>(vcode & 0xFFu) << (8 * n);
                    ++n;
                  }
                }
              };
              if (__ballot(okU)) visit(okU, code - 16u);
              if (__ballot(okD)) visit(okD, code + 16u);
              if (__ballot(okL)) visit(okL, code - 1u);
              if (__ballot(okR)) visit(okR, code + 1u);
              ++hpop;
              if (hpop >= n) act = false;
              if (act && hpop >= maxpops) {  // (see maxpops)
                complete = false;
                act = false;
              }
              // a seed whose four neighbours all join sits inside a plane: leave it to the wave-wide BFS right away (any
              // give-up rule is exact — an incomplete grow is simply not committed from here)
              if (act && hpop == 1 && n == 5) {
                complete = false;
                act = false;
              }
            }
            const uint32_t l0 = static_cast<uint32_t>(ll), l1 = static_cast<uint32_t>(ll >> 32);
            if (DBG) a_spec += DPX_STAMP() - t_sp0;
            const unsigned long long t_cl0 = DPX_STAMP();
            // 5. which events fire.  Rounds in which every grow is a single cell need no claims: the seeds are distinct
            // cells of distinct bins.  Otherwise every complete lane claims its cells, and the bins of its non-seed cells,
            // with min(lane); every lane then reads the lowest claimant of its bin (lowb) and of its cells (lowc).
            // The non-seed cells of all lanes are handled as ITEMS, one wave lane each (a round has 15-20 of them on a noisy
            // frame, while a loop over list positions runs seven times for every lane): item k = (source lane, list index)
            // through an exclusive prefix sum of the list lengths, the source lane scattered into a byte array; a lane's
            // list and seed travel through a 16-byte record.  Lists beyond the 64th item are given up (any give-up is exact).
            bool spec = part && complete;
            int cnt = spec ? n - 1 : 0;
            int incl_items = wave_incl_scan_i32(cnt);
            if (DBG) n_itemover += __popcll(__ballot(cnt > 0 && incl_items > 64));
            if (cnt > 0 && incl_items > 64) {
              complete = false;
              spec = false;
              cnt = 0;
            }
            const unsigned long long multi = __ballot(cnt > 0);
            uint32_t lowb = 0xFFFFFFFFu, lowc = 0xFFFFFFFFu;
            bool it_ok = false;
            int it_src = 0, it_cell = 0, it_bin = -1, it_i = 0;
            if (multi) {
              const int off = incl_items - cnt;
              const int nitems = static_cast<int>(wave_max_u32(cnt > 0 ? static_cast<uint32_t>(incl_items) : 0u));
              rrec[lane] = make_uint4(l0, l1, static_cast<uint32_t>(cs), static_cast<uint32_t>(off));
              rres[lane] = 0xFFFFFFFFu;
  #pragma unroll
              for (int i = 1; i < kSpecMax; ++i)
                if (i <= cnt) rsrc[off + i - 1] = static_cast<uint8_t>(lane);
              asm volatile("" ::: "memory");  // (LDS executes a wave's operations in order; only the compiler must keep it)
              __builtin_amdgcn_wave_barrier();
              it_ok = lane < nitems;
              if (it_ok) {
                it_src = static_cast<int>(rsrc[lane]);
                const uint4 r = rrec[it_src];
                it_i = lane - static_cast<int>(r.w) + 1;
                const unsigned long long ll = (static_cast<unsigned long long>(r.y) << 32) | r.x;
                const uint32_t code = static_cast<uint32_t>(ll >> (8 * it_i)) & 0xFFu;
                it_cell = static_cast<int>(r.z) + (static_cast<int>(code >> 4) - 8) * nh + (static_cast<int>(code & 15u) - 8);
                const int bn = HASBIN ? static_cast<int>(binof[it_cell]) : static_cast<int>(meta[it_cell] >> META_BIN_SHIFT);
                it_bin = (bn - 1 < nb) ? bn - 1 : -1;
              }
              if (spec) atomicMin(&ctab[claim_slot(static_cast<uint32_t>(cs))], static_cast<uint32_t>(lane));
              if (it_ok) {
                atomicMin(&ctab[claim_slot(static_cast<uint32_t>(it_cell))], static_cast<uint32_t>(it_src));
                if (it_bin >= 0) atomicMin(&bclaim[it_bin], static_cast<uint32_t>(it_src));
              }
              asm volatile("" ::: "memory");
              __builtin_amdgcn_wave_barrier();
              if (part) {  // (incomplete grows read too: whether their bin / seed is still theirs decides skip or wide BFS)
                lowb = bclaim[b];
                lowc = ctab[claim_slot(static_cast<uint32_t>(cs))];
              }
              if (it_ok) atomicMin(&rres[it_src], ctab[claim_slot(static_cast<uint32_t>(it_cell))]);
              asm volatile("" ::: "memory");
              __builtin_amdgcn_wave_barrier();
              if (cnt > 0) {
                const uint32_t m = rres[lane];
                lowc = m < lowc ? m : lowc;
              }
              // leave the tables clean for the next round
              if (spec) ctab[claim_slot(static_cast<uint32_t>(cs))] = 0xFFFFFFFFu;
              if (it_ok) {
                ctab[claim_slot(static_cast<uint32_t>(it_cell))] = 0xFFFFFFFFu;
                if (it_bin >= 0) bclaim[it_bin] = 0xFFFFFFFFu;
              }
            }
            // Lanes in order; only the flagged ones need a look.  A lane whose bin lost a cell to a lower lane THAT FIRES is no
            // longer at level L when its turn comes: it does not fire in this sweep and is skipped.  The first lane that
            // fires with a grow that is large, or that shares a cell with a lower lane, ends the prefix; so does any lane whose
            // lowest claimant was itself skipped (the table keeps only the minimum: what the other claimants did is unknown).
            const unsigned long long t_cm0 = DPX_STAMP();
            if (DBG) a_claims += t_cm0 - t_cl0;
            const bool flag = part && (!complete || lowb < static_cast<uint32_t>(lane) || lowc < static_cast<uint32_t>(lane));
            unsigned long long fl = __ballot(flag), skipped = 0ull;
            int stop = 64;
            bool stop_wide = false;
            while (fl) {
              const int t = static_cast<int>(__builtin_ctzll(fl));
              fl &= fl - 1ull;
              const uint32_t mb = static_cast<uint32_t>(__builtin_amdgcn_readlane(static_cast<int>(lowb), t));
              const uint32_t mc = static_cast<uint32_t>(__builtin_amdgcn_readlane(static_cast<int>(lowc), t));
              const bool inc = __builtin_amdgcn_readlane(static_cast<int>(complete), t) == 0;
              if (mb < static_cast<uint32_t>(t)) {
                if (((skipped >> mb) & 1ull) == 0ull) {
                  skipped |= 1ull << t;
                  if (DBG) ++n_stop_bin;
                  continue;
                }
                stop = t;
                break;
              }
              stop = t;
              stop_wide = !(mc < static_cast<uint32_t>(t)) && inc;
              if (DBG) { if (stop_wide) ++n_stop_inc; else ++n_stop_cell; }
              break;
            }
            // 6. commit the lanes that fire before the stop
            const unsigned long long cm = (stop >= 64 ? partmask : (partmask & ((1ull << stop) - 1ull))) & ~skipped;
            const bool cmt = ((cm >> lane) & 1ull) != 0ull;
            if (part) {  // list head and seed cache of every participating bin
              const bool used = cmt || (stop_wide && lane == stop);
              bin_start[b] = static_cast<idx_t>(used ? pos + 1 : pos);
              candc[b] = used ? kCandNone : (c | ((complete && !big) ? 0u : kCandBig));
            }
            if (cm) {
              const int grank = static_cast<int>(__popcll(cm & lt_mask));
              const int ncm = static_cast<int>(__popcll(cm));
              int base, total;
              if (!(multi & cm)) {  // (uniform) one cell each
                base = tail + grank;
                total = ncm;
              } else {
                const int myn = cmt ? n : 0;
                int incl = myn;  // inclusive wave scan with DPP row shifts / broadcasts
                incl += __builtin_amdgcn_update_dpp(0, incl, 0x111, 0xf, 0xf, false);
                incl += __builtin_amdgcn_update_dpp(0, incl, 0x112, 0xf, 0xf, false);
                incl += __builtin_amdgcn_update_dpp(0, incl, 0x114, 0xf, 0xf, false);
                incl += __builtin_amdgcn_update_dpp(0, incl, 0x118, 0xf, 0xf, false);
                incl += __builtin_amdgcn_update_dpp(0, incl, 0x142, 0xa, 0xf, false);
                incl += __builtin_amdgcn_update_dpp(0, incl, 0x143, 0xc, 0xf, false);
                total = __builtin_amdgcn_readlane(incl, 63);
                base = tail + incl - myn;
              }
              const bool want_cand = cmt && !(static_cast<unsigned long long>(n) < static_cast<unsigned long long>(static_cast<long long>(prm.min_cells)));
              const unsigned long long wantmask = __ballot(want_cand);
              if (cmt) {
                queue[base] = static_cast<idx_t>(cs);
                // (no queue position: the BFS's pull test only needs the positions of the CURRENT batch, and a stale or
                // absent one can never fall into it)
                atomicAnd(reinterpret_cast<uint32_t*>(flags + (cs & ~3)), ~(F_UNASSIGNED << ((cs & 3) * 8)));
                atomicSub(&hist[b], 1);  // hist.removePoint
                grow_start[ngrow + grank] = static_cast<uint32_t>(base);
              }
              if (multi & cm) {  // the non-seed cells, item by item: queue slot, flag bit, histogram
                rres[lane] = static_cast<uint32_t>(base);
                asm volatile("" ::: "memory");
                __builtin_amdgcn_wave_barrier();
                if (it_ok && ((cm >> it_src) & 1ull) != 0ull) {
                  const int v = it_cell;
                  queue[static_cast<int>(rres[it_src]) + it_i] = static_cast<idx_t>(v);
                  atomicAnd(reinterpret_cast<uint32_t*>(flags + (v & ~3)), ~(F_UNASSIGNED << ((v & 3) * 8)));
                  if (it_bin >= 0) atomicSub(&hist[it_bin], 1);
                }
              }
              if (wantmask) {
                const int crank = static_cast<int>(__popcll(wantmask & lt_mask));
                if (want_cand && ncand + crank < rb.pcap)
                  cand_grow[ncand + crank] = make_uint4(static_cast<uint32_t>(ngrow + grank), static_cast<uint32_t>(base), static_cast<uint32_t>(base + n), 0u);
                const int nwant = static_cast<int>(__popcll(wantmask));
                ncand += nwant < rb.pcap - ncand ? nwant : (rb.pcap - ncand > 0 ? rb.pcap - ncand : 0);
              }
              tail = __builtin_amdgcn_readfirstlane(tail + total);
              ngrow += ncm;
              remaining -= total;
              if (DBG) n_spec += ncm;
            }
            // (nothing written here is read back through global memory before the tail phases' barriers: no fence)
            asm volatile("" ::: "memory");
            __builtin_amdgcn_wave_barrier();
            if (DBG) a_commit += DPX_STAMP() - t_cm0;
            if (!stop_wide) {
              if (DBG) {
                const unsigned long long tc2 = DPX_STAMP();
                a_argmax += tb - ta; a_seed += tc2 - tb;
              }
              st = 0;
              break;
            }
            // the stopping event is a large grow whose bin and seed no lower lane touched: it fires now, wave-wide
            seed = static_cast<uint32_t>(__builtin_amdgcn_readlane(static_cast<int>(c), stop));
            bb = __builtin_amdgcn_readlane(b, stop);
            }  // (several events)
          }
          asm volatile("" ::: "memory");
          __builtin_amdgcn_wave_barrier();

          tc = DPX_STAMP();
          gstart = tail;
          if (lane == 0) {
            queue[tail] = static_cast<idx_t>(seed);
            if constexpr (GST) ring[static_cast<uint32_t>(tail) & kRingMask] = seed;
            qpos[seed] = static_cast<idx_t>(tail);
            flags[seed] = static_cast<uint8_t>(flags[seed] & ~F_UNASSIGNED);
            atomicSub(&hist[bb], 1);  // hist.removePoint(seed); the seed's bin is bb
          }
          tail = __builtin_amdgcn_readfirstlane(tail + 1);
          head = __builtin_amdgcn_readfirstlane(gstart);
          in_grow = true;
          st = 0;
          wave_sync();
          while (head < tail) {
            if constexpr (GST) {
              if (tail - head >= kWideSlots) {
                st = 1;
                break;
              }
            }
            if constexpr (TIER == 1) {
              if (tail - head == 1) {  // (uniform)
                // A frontier of ONE cell — 43 % of the steps on a noisy large frame, every grow's first: its four neighbours
                // are distinct cells, so there is nothing to deduplicate and no queue position to read.  Lanes 0..3 = U, D,
                // L, R.  (Tier 1 only: in tier 0 the bin lookup is a global round trip that the longer general step hides
                // and this one does not — measured, 14.1 -> 14.4 ms on 4K; tier 2 keeps its one-wave launch small.)
                const bool active1 = slot == 0;
                const int u1 = static_cast<int>(queue[head]);
                const int vraw1 = u1 + delta;
                const bool inrange1 = active1 && vraw1 >= 0 && vraw1 < C;
                const int v1 = vraw1 < 0 ? 0 : (vraw1 >= C ? C - 1 : vraw1);
                const uint32_t fv1 = flags[v1];
                int bin1 = static_cast<int>(binof[v1]) - 1;
                asm volatile("" : "+v"(bin1));  // (fetched with the flag byte, not after the decision)
                const bool win1 = inrange1 && (fv1 & (((fv1 >> 2) & 0xFu) >> opp) & 1u) != 0u;
                const unsigned long long mask1 = __ballot(win1);
                if (win1) {
                  const int pos = tail + __popcll(mask1 & lt_mask);
                  queue[pos] = static_cast<idx_t>(v1);
                  qpos[v1] = static_cast<idx_t>(pos);
                  flags[v1] = static_cast<uint8_t>(fv1 & ~F_UNASSIGNED);
                  if (bin1 >= 0 && bin1 < nb) atomicSub(&hist[bin1], 1);
                }
                tail = __builtin_amdgcn_readfirstlane(tail + static_cast<int>(__popcll(mask1)));
                head = __builtin_amdgcn_readfirstlane(head + 1);
                asm volatile("" ::: "memory");
                __builtin_amdgcn_wave_barrier();
                if (DBG) ++n_steps;
                continue;
              }
            }
            // Branch-free step: lane = (slot, dir).  All reads are unconditional (clamped addresses), the decision is
            // pure bit arithmetic: a single wave issues one instruction per ~4-8 cycles, so instruction count is latency.
            const int nslots = (tail - head) < 16 ? (tail - head) : 16;
            const bool active = slot < nslots;
            const int qi = head + (active ? slot : 0);
            const uint32_t uh = static_cast<uint32_t>(head), us = static_cast<uint32_t>(slot);
            // (tier 0: the batch is in the LDS ring unless the frontier is wider than the ring — then from global memory,
            // which the fence at the end of the previous step made visible)
            const int u = GST ? ((tail - head) <= (1 << kRingBitsL) - 32 ? static_cast<int>(ring[static_cast<uint32_t>(qi) & kRingMask]) : static_cast<int>(queue[qi]))
                              : static_cast<int>(queue[qi]);
            const int vraw = u + delta;
            const bool inrange = active && (vraw >= 0) && (vraw < C);
            const int v = vraw < 0 ? 0 : (vraw >= C ? C - 1 : vraw);
            const uint32_t fv = flags[v];
            // positions of v's four neighbours in the queue.  In the LDS tiers the reads are not clamped: an index outside
            // the grid lands in the neighbouring LDS arrays, and whatever it returns is masked by v's in-edge bit for that
            // (non-existent) neighbour being 0.  Tier 0 reads global memory and clamps.
            const int wU = GST ? (v - nh > 0 ? v - nh : 0) : v - nh, wD = GST ? (v + nh < C ? v + nh : C - 1) : v + nh;
            const int wL = GST ? (v - 1 > 0 ? v - 1 : 0) : v - 1, wR = GST ? (v + 1 < C ? v + 1 : C - 1) : v + 1;
            const uint32_t qU = qpos[wU], qD = qpos[wD], qL = qpos[wL], qR = qpos[wR];
            // v's bin, needed only if v is activated, is fetched with the reads above instead of after the decision
            int bin_v = (HASBIN ? static_cast<int>(binof[v]) : static_cast<int>(meta[v] >> META_BIN_SHIFT)) - 1;
            asm volatile("" : "+v"(bin_v));  // keep the load up here with the others (it would be sunk into the branch)
            // bit k set: v's neighbour k sits EARLIER in this FIFO batch (queue position in [batch head, batch head + slot))
            const uint32_t earlier = (((qU - uh) < us) ? 1u : 0u) | (((qD - uh) < us) ? 2u : 0u) |
                                     (((qL - uh) < us) ? 4u : 0u) | (((qR - uh) < us) ? 8u : 0u);
            const uint32_t inb = (fv >> 2) & 0xFu;  // in-edge bits of v
            // u is v's neighbour number dir^1: v accepts u iff that in-edge is set; u loses to any OTHER in-neighbour
            // that is earlier in the batch.  Integer logic only (no short-circuit): keeps the reads above in one
            // batch instead of three dependent round trips.
            const uint32_t okbits = (inrange ? 1u : 0u) & fv & (inb >> opp) & (((inb & earlier & notme) == 0u) ? 1u : 0u);
            const bool win = okbits != 0u;
            const unsigned long long mask = __ballot(win);
            if (win) {
              const int pos = tail + __popcll(mask & lt_mask);
              queue[pos] = static_cast<idx_t>(v);
              if constexpr (GST) ring[static_cast<uint32_t>(pos) & kRingMask] = static_cast<uint32_t>(v);
              qpos[v] = static_cast<idx_t>(pos);
              flags[v] = static_cast<uint8_t>(fv & ~F_UNASSIGNED);
              if (bin_v >= 0 && bin_v < nb) atomicSub(&hist[bin_v], 1);  // hist.removePoint at activation time
            }
            // head and tail are wave-uniform: keep them in scalar registers (the compiler cannot prove it)
            tail = __builtin_amdgcn_readfirstlane(tail + static_cast<int>(__popcll(mask)));
            head = __builtin_amdgcn_readfirstlane(head + nslots);
            if constexpr (GST) {
              wave_sync();  // queue / queue positions in global memory: the next step's loads must see these stores
            } else {
              // LDS executes a wave's operations in order, so the next step's reads see this step's writes without
              // waiting for them to retire; only the compiler has to keep the order
              asm volatile("" ::: "memory");
              __builtin_amdgcn_wave_barrier();
            }
            if (DBG) ++n_steps;
          }
        } while (false);
        if (st != 0) break;
        if (in_grow) finish_grow();
        }
      }
      if constexpr (GST) {
        int* ctl = hdr + 8 + 4 * (round & 1);  // two sets: a wave may still be reading the previous round's
        if (HELP && tid == 0) *static_cast<volatile int*>(hdr + 42) = round + 1;  // (the helper leaves its loop and comes to the barrier)
        if (tid == 0) {
          ctl[0] = st;
          ctl[1] = head;
          ctl[2] = tail;
        }
        __syncthreads();
        st = ctl[0];
        if (st == 2) break;
        if (st == 1) {
          int H = ctl[1], T = ctl[2];
          int* wc = hdr + 24;  // per-wave winner counts of the current batch
          while (H < T) {
            const int ns = (T - H) < 16 * kRW ? (T - H) : 16 * kRW;
            const int gs = wid * 16 + slot;  // slot of this lane's queue entry within the batch
            const bool active = gs < ns;
            const int qi = H + (active ? gs : 0);
            const uint32_t uh = static_cast<uint32_t>(H), us = static_cast<uint32_t>(gs);
            const int u = (T - H) <= (1 << kRingBitsL) - 256 ? static_cast<int>(ring[static_cast<uint32_t>(qi) & kRingMask]) : static_cast<int>(queue[qi]);  // (uniform)
            const int vraw = u + delta;
            const bool inrange = active && (vraw >= 0) && (vraw < C);
            const int v = vraw < 0 ? 0 : (vraw >= C ? C - 1 : vraw);
            const uint32_t fv = flags[v];
            // positions of v's four neighbours in the queue.  In the LDS tiers the reads are not clamped: an index outside
            // the grid lands in the neighbouring LDS arrays, and whatever it returns is masked by v's in-edge bit for that
            // (non-existent) neighbour being 0.  Tier 0 reads global memory and clamps.
            const int wU = GST ? (v - nh > 0 ? v - nh : 0) : v - nh, wD = GST ? (v + nh < C ? v + nh : C - 1) : v + nh;
            const int wL = GST ? (v - 1 > 0 ? v - 1 : 0) : v - 1, wR = GST ? (v + 1 < C ? v + 1 : C - 1) : v + 1;
            const uint32_t qU = qpos[wU], qD = qpos[wD], qL = qpos[wL], qR = qpos[wR];
            // v's bin, needed only if v is activated, is fetched with the reads above instead of after the decision
            int bin_v = (HASBIN ? static_cast<int>(binof[v]) : static_cast<int>(meta[v] >> META_BIN_SHIFT)) - 1;
            asm volatile("" : "+v"(bin_v));  // keep the load up here with the others (it would be sunk into the branch)
            // bit k set: v's neighbour k sits EARLIER in this FIFO batch (queue position in [batch head, batch head + slot))
            const uint32_t earlier = (((qU - uh) < us) ? 1u : 0u) | (((qD - uh) < us) ? 2u : 0u) |
                                     (((qL - uh) < us) ? 4u : 0u) | (((qR - uh) < us) ? 8u : 0u);
            const uint32_t inb = (fv >> 2) & 0xFu;  // in-edge bits of v
            // u is v's neighbour number dir^1: v accepts u iff that in-edge is set; u loses to any OTHER in-neighbour
            // that is earlier in the batch.  Integer logic only (no short-circuit): keeps the reads above in one
            // batch instead of three dependent round trips.
            const uint32_t okbits = (inrange ? 1u : 0u) & fv & (inb >> opp) & (((inb & earlier & notme) == 0u) ? 1u : 0u);
            const bool win = okbits != 0u;
            const unsigned long long mask = __ballot(win);
            if (lane == 0) wc[wid] = __popcll(mask);
            __syncthreads();
            int off = 0, tot = 0;
#pragma unroll
            for (int w = 0; w < kRW; ++w) {
              const int cw = wc[w];
              off += w < wid ? cw : 0;
              tot += cw;
            }
            if (win) {
              const int pos = T + off + __popcll(mask & lt_mask);
              queue[pos] = static_cast<idx_t>(v);
              if constexpr (GST) ring[static_cast<uint32_t>(pos) & kRingMask] = static_cast<uint32_t>(v);
              qpos[v] = static_cast<idx_t>(pos);
              flags[v] = static_cast<uint8_t>(fv & ~F_UNASSIGNED);
              if (bin_v >= 0 && bin_v < nb) atomicSub(&hist[bin_v], 1);
            }
            T += tot;
            H += ns;
            if (DBG) ++n_steps;
            __syncthreads();  // this batch's queue / position / flag writes are visible before the next batch reads
          }
          tail = T;
        }
      } else {
        if (st == 2) break;
      }
      if (wid == 0) {  // (tier 0 only gets here: the wide BFS of wave 0's current grow just finished)
        finish_grow();
        ++guard;
      }
    }
    if (wid == 0 && lane == 0) {
      grow_start[ngrow] = static_cast<uint32_t>(tail);
      hdr[1] = tail;
      hdr[2] = ngrow;
      hdr[3] = ncand;
    }
  }
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
  // ---------------- P2: exact float accumulation of every candidate segment (wave per segment) ------------
  if constexpr (PHASE != 5) {
    constexpr int kStage = stage_entries(TIER);
    float* st = stage + wid * kStage * 9;
    const float4* cf4 = reinterpret_cast<const float4*>(cells);
    // (segments are claimed from a counter, hdr[40], rather than dealt out round robin: their sizes differ by two orders
    // of magnitude.  Fetching the next segment's first records while this one's chains run was tried and gained nothing.)
    for (;;) {
      int ci = 0;
      if (lane == 0) ci = atomicAdd(&hdr[40], 1);
      ci = __builtin_amdgcn_readfirstlane(ci);
      if (ci >= ncand) break;
      const uint4 cg = cand_grow[ci];
      const int gi = static_cast<int>(cg.x), gs = static_cast<int>(cg.y), ge = static_cast<int>(cg.z);
      const int comp = lane < 9 ? lane : 0;
      float acc = reinterpret_cast<const float*>(cells)[static_cast<size_t>(queue[gs]) * 16 + comp];  // the seed, counted twice
      // the records of the NEXT 32 cells are fetched while the 9 chains run over the current 32 (the chains are
      // sequential by definition, the gathers are not)
      float4 pa = make_float4(0.f, 0.f, 0.f, 0.f), pb = pa, pc = pa;
      if (gs + lane < ge && lane < kStage) {
        const size_t v = queue[gs + lane];
        pa = cf4[v * 4 + 0]; pb = cf4[v * 4 + 1]; pc = cf4[v * 4 + 2];
      }
      for (int base = gs; base < ge; base += kStage) {
        const int cnt = (ge - base) < kStage ? (ge - base) : kStage;
        if (lane < cnt) {  // staging is component-major: st[comp][entry], so a chain reads its 32 values as 8 x 16 bytes
          float* d = st + lane;
          d[0 * kStage] = pa.x; d[1 * kStage] = pa.y; d[2 * kStage] = pa.z; d[3 * kStage] = pa.w; d[4 * kStage] = pb.x;
          d[5 * kStage] = pb.y; d[6 * kStage] = pb.z; d[7 * kStage] = pb.w; d[8 * kStage] = pc.x;
        }
        wave_sync();
        {
          const int nx = base + kStage + lane;
          if (nx < ge && lane < kStage) {
            const size_t v = queue[nx];
            pa = cf4[v * 4 + 0]; pb = cf4[v * 4 + 1]; pc = cf4[v * 4 + 2];
          }
        }
        if (lane < 9) {
          const float4* row = reinterpret_cast<const float4*>(st + comp * kStage);
          float t[kStage];
#pragma unroll
          for (int q = 0; q < kStage / 4; ++q) {
            const float4 x = row[q];
            t[4 * q] = x.x; t[4 * q + 1] = x.y; t[4 * q + 2] = x.z; t[4 * q + 3] = x.w;
          }
          if (cnt == kStage) {
#pragma unroll
            for (int q = 0; q < kStage; ++q) acc = acc + t[q];
          } else {
#pragma unroll
            for (int q = 0; q < kStage; ++q)
              if (q < cnt) acc = acc + t[q];
          }
        }
        wave_sync();
      }
      float* co = cand + static_cast<size_t>(ci) * 12;
      if (lane < 9) co[lane] = acc;
      if (lane == 9) co[9] = __int_as_float((ge - gs + 1) * prm.n);
      if (lane == 10) co[10] = __int_as_float(gi);
    }
  }
  __syncthreads();
  t_p2 = DPX_STAMP();

  // ---------------- P3: plane fit per candidate, thread-parallel; ordered ids ----------------------------
  int nplanes = 0;
  for (int base = 0; base < ncand; base += kRT) {
    const int ci = base + tid;
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
      if constexpr (PHASE == 5) {
        const float* cf = rx.candfit + (static_cast<size_t>(frame) * rb.pcap + ci) * 12;
        fit.mean[0] = cf[0]; fit.mean[1] = cf[1]; fit.mean[2] = cf[2];
        fit.normal[0] = cf[3]; fit.normal[1] = cf[4]; fit.normal[2] = cf[5];
        fit.d = cf[6]; fit.mse = cf[7]; fit.score = cf[8];
        ok = __float_as_int(cf[9]) != 0;
      } else {
        fit_plane(s, pp, npts, fit);
        ok = fit.score > prm.min_score;
      }
    }
    const unsigned long long mask = __ballot(ok);
    if (lane == 0) hdr[16 + wid] = __popcll(mask);
    __syncthreads();
    int off = 0, total = 0;
    for (int w = 0; w < kRW; ++w) {
      const int cw = hdr[16 + w];
      off += w < wid ? cw : 0;
      total += cw;
    }
    const int id = nplanes + off + __popcll(mask & lt_mask);
    if (ok) {
      dpx_plane& pl = planes[id];
      pl.sum[0] = s[0]; pl.sum[1] = s[1]; pl.sum[2] = s[2];
      pl.pp[0] = pp[0]; pl.pp[1] = pp[1]; pl.pp[2] = pp[2]; pl.pp[3] = pp[3]; pl.pp[4] = pp[4]; pl.pp[5] = pp[5];
      pl.mean[0] = fit.mean[0]; pl.mean[1] = fit.mean[1]; pl.mean[2] = fit.mean[2];
      pl.normal[0] = fit.normal[0]; pl.normal[1] = fit.normal[1]; pl.normal[2] = fit.normal[2];
      pl.d = fit.d; pl.mse = fit.mse; pl.score = fit.score;
      pl.n_pts = npts;
      pl.n_cells = static_cast<int>(cand_grow[ci].z - cand_grow[ci].y);
      pl.merge_label = id;
      merge[id] = id;
    }
    if (ci < ncand) co[11] = __int_as_float(ok ? id + 1 : 0);
    nplanes += total;
    __syncthreads();
  }

  t_p3 = DPX_STAMP();
  // ---------------- P4: labels_map_ ------------------------------------------------------------------------
  for (int ci = wid; ci < ncand; ci += kRW) {
    const float* co = cand + static_cast<size_t>(ci) * 12;
    const int id1 = __float_as_int(co[11]);
    if (id1 == 0) continue;
    const uint4 cg = cand_grow[ci];
    const int gs = static_cast<int>(cg.y), ge = static_cast<int>(cg.z);
    for (int i = gs + lane; i < ge; i += 64) lab16[queue[i]] = static_cast<idx_t>(id1);
  }
  __syncthreads();
  t_p4 = DPX_STAMP();

  // ---------------- P5: findMergedLabels ---------------------------------------------------------------------
  // getConnectedComponents (plane_extractor.cpp:430-453) looks at rows 0..nv-2 x cols 0..nh-2 of labels_map_, right and
  // down neighbour, symmetrised; findMergedLabels (:394-428) then walks the planes in id order and tests, against the
  // current parameters of merge[row], every adjacent plane with a larger id, folding accepted ones in ascending order.
  // Small plane counts (up to a few hundred): adjacency bit matrix, a 7-float view of every plane and the merge labels
  // in LDS, wave 0 only.  Larger counts: the workgroup builds a CSR adjacency (lower id -> higher ids) in global
  // scratch with a counting sort, then wave 0 walks only the non-empty rows.
  int merges = 0;
  const int wpr = (nplanes + 31) >> 5;  // words per adjacency bit row
  constexpr int kMatWords = 2304;       // stage holds at least 2304 words
  // Plane views: tiers 2/1 keep them in qpos|queue (dead, 4*C bytes); tier 0 in the LDS flag array (dead, C bytes),
  // behind the part of it that the staging buffer (the bit matrix below) overlays.
  uint32_t pv_off = lay.flags;
  {
    const uint32_t stage_end = lay.stage + static_cast<uint32_t>(stage_floats(TIER)) * 4u;
    if (lay.stage < lay.flags + static_cast<uint32_t>(C) && stage_end > pv_off) pv_off = (stage_end + 3u) & ~3u;
  }
  const int pv_room = GST ? (pv_off < lay.flags + static_cast<uint32_t>(C) ? static_cast<int>(lay.flags + C - pv_off) : 0) : 4 * C;
  const bool fast = nplanes * 32 <= pv_room && nplanes * wpr <= kMatWords;
  // Larger plane counts (CSR walk below): the 7-float views and the merge labels still go to LDS when they fit — the
  // walk then reads a row's and its columns' parameters from LDS instead of gathering plane records from global memory,
  // four dependent round trips per row less (1 716 planes on a noisy 4K frame: 48 KB of the dead flag array).  In the
  // LDS tiers the bit row of the wide-row walk follows the views.
  const bool views = nplanes > 0 && !fast && nplanes * 32 + (GST ? 0 : wpr * 4) <= pv_room;
  // The merge walk below is one wave's work, and so is what follows it for a frame of a few thousand cells: when the
  // walk does not need the workgroup (no CSR build, no fused label expansion), waves 1.. end here and give their
  // registers and wave slots back to the chip; wave 0 finishes the frame alone, without further workgroup barriers.
  const bool solo = !DBG && !rx.labels_out[0] && (nplanes == 0 || fast);
  const int gw = nh - 1, gh = nv - 1;
  const int gtotal = gw > 0 && gh > 0 ? gw * gh : 0;
  uint32_t* mat = reinterpret_cast<uint32_t*>(stage);
  float* pview = GST ? reinterpret_cast<float*>(smem + pv_off) : reinterpret_cast<float*>(qpos);  // dead arrays
  int32_t* lmerge = reinterpret_cast<int32_t*>(pview + 7 * nplanes);
  if (nplanes > 0 && fast) {
    // adjacency bit matrix and plane views: the whole workgroup builds them (a scan of the cell grid), before waves 1..
    // leave
    for (int w = tid; w < nplanes * wpr; w += kRT) mat[w] = 0;
    for (int i = tid; i < nplanes; i += kRT) {
      const dpx_plane& pl = planes[i];
      float* pv = pview + i * 7;
      pv[0] = pl.normal[0]; pv[1] = pl.normal[1]; pv[2] = pl.normal[2]; pv[3] = pl.d;
      pv[4] = pl.mean[0]; pv[5] = pl.mean[1]; pv[6] = pl.mean[2];
      lmerge[i] = i;
    }
    __syncthreads();
    for (int idx = tid; idx < gtotal; idx += kRT) {
      const int r = idx / gw, c = idx - r * gw;
      const int a = lab16[r * nh + c];
      const int right = lab16[r * nh + c + 1];
      const int down = lab16[(r + 1) * nh + c];
      if (a > 0 && right > 0 && a != right) {
        atomicOr(&mat[(a - 1) * wpr + ((right - 1) >> 5)], 1u << ((right - 1) & 31));
        atomicOr(&mat[(right - 1) * wpr + ((a - 1) >> 5)], 1u << ((a - 1) & 31));
      }
      if (a > 0 && down > 0 && a != down) {
        atomicOr(&mat[(a - 1) * wpr + ((down - 1) >> 5)], 1u << ((down - 1) & 31));
        atomicOr(&mat[(down - 1) * wpr + ((a - 1) >> 5)], 1u << ((a - 1) & 31));
      }
    }
    __syncthreads();
  }
  if (solo && wid != 0) return;
  const int tstride = solo ? 64 : kRT;
  // merge test of the row plane (pnx.., pd) against a column plane (plane_extractor.cpp:405-412)
  auto merge_test = [&](float pnx, float pny, float pnz, float pd, float cnx, float cny, float cnz, float cmx, float cmy, float cmz) {
    const float cosv = dot3(pnx, pny, pnz, cnx, cny, cnz);
    const float tf = dot3(pnx, pny, pnz, cmx, cmy, cmz) + pd;  // float add (plane_extractor.cpp:410-411)
    const double dist = static_cast<double>(tf) * static_cast<double>(tf);
    return static_cast<double>(cosv) > static_cast<double>(prm.min_cos) && dist < static_cast<double>(prm.max_merge_dist);
  };
  if (nplanes > 0 && fast && wid == 0) {
    // rows in ascending order; a row whose plane has no adjacent plane with a larger id does nothing and is skipped (the
    // bit matrix does not change during the walk)
    for (int rbase = 0; rbase < nplanes; rbase += 64) {
     unsigned long long todo;
     {
       const int rr = rbase + lane;
       bool any = false;
       if (rr < nplanes)
         for (int w = rr >> 5; w < wpr; ++w) {
           uint32_t bits = mat[rr * wpr + w];
           if (w == (rr >> 5)) bits &= ~((2u << (rr & 31)) - 1u);  // columns above rr only
           any = any || bits != 0u;
         }
       todo = __ballot(any);
     }
     while (todo) {
      const int row = rbase + static_cast<int>(__builtin_ctzll(todo));
      todo &= todo - 1;
      const uint32_t* rowbits = mat + row * wpr;
      const int pid = lmerge[row];
      const float* pv = pview + pid * 7;
      const float pnx = pv[0], pny = pv[1], pnz = pv[2], pd = pv[3];
      bool expanded = false;
      for (int cbase = row + 1; cbase < nplanes; cbase += 64) {
        const int col = cbase + lane;
        bool acc = false;
        if (col < nplanes && ((rowbits[col >> 5] >> (col & 31)) & 1u)) {
          const float* cv = pview + col * 7;
          acc = merge_test(pnx, pny, pnz, pd, cv[0], cv[1], cv[2], cv[4], cv[5], cv[6]);
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
          if (lane == 10) {
            merge[colx] = pid;
            lmerge[colx] = pid;
          }
          expanded = true;
          ++merges;
        }
      }
      if (expanded) {
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
        wave_sync();
        if (lane == 0) {
          dpx_plane& pl = planes[pid];
          PlaneFit fit;
          fit_plane(pl.sum, pl.pp, pl.n_pts, fit);
          pl.mean[0] = fit.mean[0]; pl.mean[1] = fit.mean[1]; pl.mean[2] = fit.mean[2];
          pl.normal[0] = fit.normal[0]; pl.normal[1] = fit.normal[1]; pl.normal[2] = fit.normal[2];
          pl.d = fit.d; pl.mse = fit.mse; pl.score = fit.score;
          float* pw = pview + pid * 7;
          pw[0] = fit.normal[0]; pw[1] = fit.normal[1]; pw[2] = fit.normal[2]; pw[3] = fit.d;
          pw[4] = fit.mean[0]; pw[5] = fit.mean[1]; pw[6] = fit.mean[2];
        }
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
      }
      wave_sync();
     }
    }
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
  }
  if (nplanes > 0 && !fast) {
    // ---- CSR adjacency in the pair scratch (4*C words): cnt[C] | off[C + 2] | cols[<= 2*gtotal]
    uint32_t* adj = reinterpret_cast<uint32_t*>(pairs);
    uint32_t* cnt = adj;
    uint32_t* off = adj + C;
    uint32_t* cols = adj + 2 * static_cast<size_t>(C) + 2;
    uint32_t* part = reinterpret_cast<uint32_t*>(stage);  // kRT partial sums
    for (int i = tid; i < nplanes; i += kRT) cnt[i] = 0;
    __syncthreads();
    // a label pair that the grid row above (horizontal pairs) / the grid column to the left (vertical pairs) already
    // produced is skipped; the remaining duplicates are harmless (the row walk takes each column once)
    auto grid_pairs = [&](int idx, int& h_lo, int& h_hi, int& v_lo, int& v_hi) {
      const int r = idx / gw, c = idx - r * gw;
      const int a = lab16[r * nh + c];
      const int right = lab16[r * nh + c + 1];
      const int down = lab16[(r + 1) * nh + c];
      bool hp = a > 0 && right > 0 && a != right;
      bool vp = a > 0 && down > 0 && a != down;
      if (hp && r > 0 && static_cast<int>(lab16[(r - 1) * nh + c]) == a && static_cast<int>(lab16[(r - 1) * nh + c + 1]) == right) hp = false;
      if (vp && c > 0 && static_cast<int>(lab16[r * nh + c - 1]) == a && static_cast<int>(lab16[(r + 1) * nh + c - 1]) == down) vp = false;
      h_lo = hp ? (a < right ? a : right) - 1 : -1;
      h_hi = (a < right ? right : a) - 1;
      v_lo = vp ? (a < down ? a : down) - 1 : -1;
      v_hi = (a < down ? down : a) - 1;
    };
    for (int idx = tid; idx < gtotal; idx += kRT) {
      int hl, hh, vl, vh;
      grid_pairs(idx, hl, hh, vl, vh);
      if (hl >= 0) atomicAdd(&cnt[hl], 1u);
      if (vl >= 0) atomicAdd(&cnt[vl], 1u);
    }
    __syncthreads();
    {  // exclusive scan of cnt -> off; cnt becomes the fill cursor
      const int per = (nplanes + kRT - 1) / kRT;
      const int i0 = tid * per, i1 = (i0 + per) < nplanes ? (i0 + per) : nplanes;
      uint32_t sum = 0;
      for (int i = i0; i < i1; ++i) sum += cnt[i];
      part[tid] = sum;
      __syncthreads();
      uint32_t base = 0;
      for (int t = 0; t < tid; ++t) base += part[t];
      for (int i = i0; i < i1; ++i) {
        const uint32_t v = cnt[i];
        off[i] = base;
        cnt[i] = 0;
        base += v;
      }
      if (tid == kRT - 1) off[nplanes] = base;
    }
    __syncthreads();
    for (int idx = tid; idx < gtotal; idx += kRT) {
      int hl, hh, vl, vh;
      grid_pairs(idx, hl, hh, vl, vh);
      if (hl >= 0) cols[off[hl] + atomicAdd(&cnt[hl], 1u)] = static_cast<uint32_t>(hh);
      if (vl >= 0) cols[off[vl] + atomicAdd(&cnt[vl], 1u)] = static_cast<uint32_t>(vh);
    }
    if (views)
      for (int i = tid; i < nplanes; i += kRT) {
        const dpx_plane& pl = planes[i];
        float* pv = pview + i * 7;
        pv[0] = pl.normal[0]; pv[1] = pl.normal[1]; pv[2] = pl.normal[2]; pv[3] = pl.d;
        pv[4] = pl.mean[0]; pv[5] = pl.mean[1]; pv[6] = pl.mean[2];
        lmerge[i] = i;
      }
    __syncthreads();
    const unsigned long long t_walk0 = DPX_STAMP();
    if (wid == 0) {
      // bit row for planes with more than 64 adjacency entries (tier 0: global scratch; LDS tiers: behind the views, or
      // the dead queue-position array when there are none)
      uint32_t* slowbits = (views && !GST) ? reinterpret_cast<uint32_t*>(lmerge + nplanes) : reinterpret_cast<uint32_t*>(qpos);
      for (int rbase = 0; rbase < nplanes; rbase += 64) {
        const int rr = rbase + lane;
        const uint32_t o0 = rr < nplanes ? off[rr] : 0u, o1 = rr < nplanes ? off[rr + 1] : 0u;
        unsigned long long todo = __ballot(o1 > o0);
        bool one_row = false;  // the next row goes through the one-row path (it has a column to fold)
        while (todo) {
          // Four rows at a time (plane views in LDS, rows of at most 16 columns): sixteen lanes per row test the row's
          // columns against the row's CURRENT plane.  Nothing changes while no column is accepted, so if none of the four
          // rows accepts one — 85 % of the rows of a noisy frame — all four are done; otherwise the rows in front of the
          // first accepting one are done and that one is walked again on its own below (folds, refit), the later ones
          // after it.  Same tests on the same state as the sequential walk, row after row.
          if (views && !one_row) {
            unsigned long long t2 = todo;
            int r4[4];
  #pragma unroll
            for (int j = 0; j < 4; ++j) {
              r4[j] = t2 ? static_cast<int>(__builtin_ctzll(t2)) : -1;
              t2 &= t2 - 1ull;  // (0 stays 0)
            }
            const int g = lane >> 4, l16 = lane & 15;
            const int rsel = g == 0 ? r4[0] : (g == 1 ? r4[1] : (g == 2 ? r4[2] : r4[3]));
            const int src = rsel >= 0 ? rsel : 0;
            const int gs0 = __shfl(static_cast<int>(o0), src), ge0 = __shfl(static_cast<int>(o1), src);
            const int gdeg = rsel >= 0 ? ge0 - gs0 : 0;
            if (!__ballot(gdeg > 16)) {
              const int grow = rbase + src;
              const int gpid = lmerge[grow];
              const float* gv = pview + gpid * 7;
              const float gnx = gv[0], gny = gv[1], gnz = gv[2], gd = gv[3];
              bool gacc = false;
              if (l16 < gdeg) {
                const uint32_t gcol = cols[gs0 + l16];
                const float* cv = pview + gcol * 7;
                gacc = merge_test(gnx, gny, gnz, gd, cv[0], cv[1], cv[2], cv[4], cv[5], cv[6]);
              }
              const unsigned long long gm = __ballot(gacc);
              if (!gm) {
                todo = t2;  // all (up to) four rows had nothing to fold
                continue;
              }
              const int gstar = static_cast<int>(__builtin_ctzll(gm)) >> 4;
  #pragma unroll
              for (int j = 0; j < 4; ++j)
                if (j < gstar) todo &= ~(1ull << r4[j]);  // (r4[j] >= 0 for j < gstar: a group in front of an accepting one is a row)
              one_row = true;
              continue;
            }
          }
          one_row = false;
          const int rl = __builtin_ctzll(todo);
          todo &= todo - 1;
          const int row = rbase + rl;
          const int s0 = __builtin_amdgcn_readlane(static_cast<int>(o0), rl), e0 = __builtin_amdgcn_readlane(static_cast<int>(o1), rl);
          const int deg = e0 - s0;
          const int pid = views ? lmerge[row] : merge[row];
          float pnx, pny, pnz, pd;
          if (views) {
            const float* pv = pview + pid * 7;
            pnx = pv[0]; pny = pv[1]; pnz = pv[2]; pd = pv[3];
          } else {
            const dpx_plane& pp_ = planes[pid];
            pnx = pp_.normal[0]; pny = pp_.normal[1]; pnz = pp_.normal[2]; pd = pp_.d;
          }
          // a column plane's normal and mean for the merge test
          auto col_test = [&](uint32_t col) {
            if (views) {
              const float* cv = pview + col * 7;
              return merge_test(pnx, pny, pnz, pd, cv[0], cv[1], cv[2], cv[4], cv[5], cv[6]);
            }
            const dpx_plane& pc = planes[col];
            return merge_test(pnx, pny, pnz, pd, pc.normal[0], pc.normal[1], pc.normal[2], pc.mean[0], pc.mean[1], pc.mean[2]);
          };
          bool expanded = false;
          auto fold = [&](int colx) {  // plane_extractor.cpp:413-417: accumulate the column's sums into the row's plane
            float* pf = reinterpret_cast<float*>(&planes[pid]);
            const float* cfp = reinterpret_cast<const float*>(&planes[colx]);
            if (lane < 9) pf[lane] = pf[lane] + cfp[lane];
            if (lane == 9) planes[pid].n_pts += planes[colx].n_pts;
            if (lane == 10) {
              merge[colx] = pid;
              if (views) lmerge[colx] = pid;
            }
            expanded = true;
            ++merges;
          };
          if (deg <= rx.csr_chunk) {
            const uint32_t col = lane < deg ? cols[s0 + lane] : 0xFFFFFFFFu;
            bool acc = false;
            if (lane < deg) acc = col_test(col);
            // accepted columns in ascending order; duplicates of a column are retired together
            while (__ballot(acc)) {
              const uint32_t cmin = wave_min_u32(acc ? col : 0xFFFFFFFFu);
              if (col == cmin) acc = false;
              fold(static_cast<int>(cmin));
            }
          } else {
            for (int w = lane; w < wpr; w += 64) slowbits[w] = 0;
            wave_sync();
            for (int i = s0 + lane; i < e0; i += 64) {
              const uint32_t c = cols[i];
              atomicOr(&slowbits[c >> 5], 1u << (c & 31));
            }
            __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
            wave_sync();
            for (int cbase = row + 1; cbase < nplanes; cbase += 64) {
              const int col = cbase + lane;
              bool acc = false;
              if (col < nplanes && ((slowbits[col >> 5] >> (col & 31)) & 1u)) acc = col_test(static_cast<uint32_t>(col));
              unsigned long long mask = __ballot(acc);
              while (mask) {
                const int l = __builtin_ctzll(mask);
                mask &= mask - 1;
                fold(cbase + l);
              }
            }
          }
          if (expanded) {
            __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
            wave_sync();
            if (lane == 0) {
              dpx_plane& pl = planes[pid];
              PlaneFit fit;
              fit_plane(pl.sum, pl.pp, pl.n_pts, fit);
              pl.mean[0] = fit.mean[0]; pl.mean[1] = fit.mean[1]; pl.mean[2] = fit.mean[2];
              pl.normal[0] = fit.normal[0]; pl.normal[1] = fit.normal[1]; pl.normal[2] = fit.normal[2];
              pl.d = fit.d; pl.mse = fit.mse; pl.score = fit.score;
              if (views) {
                float* pw = pview + pid * 7;
                pw[0] = fit.normal[0]; pw[1] = fit.normal[1]; pw[2] = fit.normal[2]; pw[3] = fit.d;
                pw[4] = fit.mean[0]; pw[5] = fit.mean[1]; pw[6] = fit.mean[2];
              }
            }
          }
          __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
          wave_sync();
        }
      }
      __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
      if (DBG) a_walk = DPX_STAMP() - t_walk0;
    }
  }
  if (!solo) __syncthreads();
  t_p5 = DPX_STAMP();
  // ---------------- P6: merged label per cell (toImageLabels' lookup, plane_extractor.cpp:464-465); records --------
  // (merge labels from their LDS copy where the merge ran on the bit matrix: one wave walks all cells here, and a
  // global gather per iteration would be a serialized memory round trip each)
  const bool lds_merge = nplanes > 0 && (fast || views);
  for (int c = tid; c < C; c += tstride) {
    const int L = lab16[c];
    const int fin = (L == 0 || nplanes == 0) ? 0 : (lds_merge ? lmerge[L - 1] : merge[L - 1]) + 1;
    cell_labels[c] = fin;
    lab16[c] = static_cast<idx_t>(fin);  // final labels stay in LDS for the fused expansion below
  }
  if (nplanes > 0)
    for (int i = tid; i < nplanes; i += tstride) planes[i].merge_label = lds_merge ? lmerge[i] : merge[i];
  if (rx.labels_out[0]) {
    // Fused toImageLabels (plane_extractor.cpp:455-470), W % 4 == 0: this workgroup writes its frame's label map with
    // 16-byte stores straight from the LDS-resident cell labels.  Done here (on the few CUs that run region growing,
    // in the background of the other batches' kernels) it removes a chip-wide launch from every batch.
    __syncthreads();
    const int sub = frame / rx.frames_per_out;
    int32_t* outp = rx.labels_out[sub] + static_cast<size_t>(frame - sub * rx.frames_per_out) * prm.N;
    const int W4 = prm.W >> 2, P = prm.P;
    const int total = prm.H * W4;
    for (int idx = tid; idx < total; idx += kRT) {
      const int row = idx / W4, q = idx - row * W4;
      const int col = q << 2;
      const int cr = row / P;
      int cc = col / P;
      int rem = col - cc * P;
      int4 v;
      int32_t* vp = reinterpret_cast<int32_t*>(&v);
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        vp[i] = (cr < nv && cc < nh) ? static_cast<int32_t>(lab16[cr * nh + cc]) : 0;
        if (++rem == P) {
          rem = 0;
          ++cc;
        }
      }
      *reinterpret_cast<int4*>(outp + static_cast<size_t>(row) * prm.W + col) = v;
    }
  }
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
