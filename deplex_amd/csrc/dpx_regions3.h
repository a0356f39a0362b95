// dpx_regions3.h — region growing for large frames through CORES (round 4).
//
// The reference's loop (plane_extractor.cpp:297-347) takes the most frequent histogram bin, grows a region from the
// bin's minimum-MSE cell with a FIFO BFS over the unassigned cells (growSeed, :349-392), removes the grown cells from
// the histogram and repeats.  growSeed's accept test reads only static per-cell data besides the unassigned mask, so the
// cells a grow activates are exactly the cells REACHABLE from its seed in the static directed graph G (edge u -> v iff v
// accepts u) restricted to the unassigned cells; only their ORDER depends on the FIFO discipline, and the order is read by
// nothing but the float accumulation of segments that survive the size test (:318-323, :329).
//
// Almost every edge of G is symmetric (both directions pass: the cosine is symmetric, the distance test nearly so).  A CORE
// is a connected component of the symmetric edges.  A core is strongly connected, so a grow takes cores whole: the cells
// of a grow from seed s are the cores reachable from core(s) over the remaining one-way edges among the cores still alive
// — and with no one-way edge leaving core(s), simply core(s), whatever happened before.  (Removing a reachable set never
// splits a core: the unassigned set stays a union of whole cores, by induction.)  Hence:
//   k_core_*    cores by parallel union-find over the symmetric edges: 32 x 32-cell tiles in LDS (k_core_tiles), then the
//               edges across tile borders in memory (k_core_link), roots per cell and the (core, bin) pairs (k_core_flatten);
//   k_pair_*, k_core_alloc, k_edge_scatter, k_bin_sort    pairs by core, one-way edges by core, seed lists by bin;
//   k_regsim    the SEQUENCE of grows, one wave per frame (+ a helper wave that keeps the seed lists' windows in LDS filled):
//               which bin fires, its seed, the cores it takes, the histogram update — no BFS.  Up to 64 bins of one
//               histogram level fire per round ("sweep rounds", as in dpx_regions2.h, where the argument for their order is
//               spelled out), validated through per-bin claims;
//   k_cellgrow  per cell: the grow that took it + its in-edge bits in one word;
//   k_order     the FIFO activation order of every grow that is a segment candidate, all grows in parallel (wave per
//               grow; grows of up to 8 cells one LANE each), written to the queue ranges the sequence assigned;
//   k_seg_accum, k_seg_fit    segment sums in queue order and f64 fits, chip-wide;
//   the rest of the tail (ordered ids, labels_map_, plane merging, final labels) is k_regions2<0, ., 5>.
// Bit-exact with the cell-level kernels of dpx_regions2.h / the oracle on every record (tests, tools/fuzz_parity.py).
#pragma once
#include "dpx_regions2.h"

namespace dpx {

constexpr uint32_t kCoreOut = 0x80000000u;     // csize[root]: a one-way edge leaves this core
constexpr uint32_t kCoreSizeMask = 0x000FFFFFu;
constexpr int kCellBits = kKeyCellBits;        // 19: C < 2^19
constexpr uint32_t kCellMask = (1u << kCellBits) - 1u;
constexpr int kPairBinBits = 13;               // bins < 2^13 (dpx_binsort.h)
constexpr uint32_t kPairBinMask = (1u << kPairBinBits) - 1u;
constexpr uint32_t kPairEmpty = 0xFFFFFFFFu;
// Words between two frames' counters (cursor[], order_task[]): a 128-byte line each.  Atomics on one LINE are served one after the
// other at the memory side (tools/ubench/atomic_same.hip: 11.4 ns on one word, 5.7 ns over the 32 words of a line, 0.1 ns on
// lines of their own): with four words per frame, eight frames' counters shared a line.
constexpr int kCtr = 32;
constexpr int kItemMax = 8;  // cores of up to this many PAIRS are handled inside a sweep round (one item per pair)

// What the sequence of grows needs to know about a frame is proportional to its (core, bin) PAIRS, not its cells:
//   * the seed of a bin is its minimum-(MSE, cell id) live cell (plane_extractor.cpp:309-316), and a core's cells die
//     together — so of the cells a core has in a bin only the best one (the pair's LEADER) can ever be a seed, and the
//     seed lists hold one entry per pair;
//   * a grow takes cores whole, so what it does to the histogram is the (bin, count) list of its cores' pairs, and whether
//     a cell is still unassigned is one bit per core.
struct CoreBufs {
  const uint32_t* meta;   // [B][C]   k_cellgraph
  const float* mse_c;     // [B][C]
  uint32_t* parent;       // [B][C]   union-find forest; after k_core_flatten: the core's root (its lowest cell id)
  uint32_t* csize;        // [B][C]   at the root: cells of the core
  uint32_t* npairs;       // [B][C]   at the root: (core, bin) pairs
  uint32_t* pstart;       // [B][C]   at the root: first slot of the core in corepairs = the core's ID
  uint32_t* pfill;        // [B][C]
  uint32_t* noe;          // [B][C]   at the root: one-way edges to other cores (duplicates included)
  uint32_t* oestart;      // [B][C]
  uint32_t* oefill;       // [B][C]
  uint32_t* hkey;         // [B][T]   hash table of the pairs: root << 13 | bin (empty between calls: k_pair_scatter resets what it reads)
  uint32_t* hslots;       // [B][C]   the occupied slots, in insertion order (cursor[3] of them)
  uint32_t* hcnt;         // [B][T]   cells of the pair
  unsigned long long* hbest;  // [B][T] min(mse key << 19 | cell): the pair's leader
  uint32_t* corepairs;    // [B][C]   pairs grouped by core: bin | count << 13
  uint4* oedges;          // [B][4C]  one-way edges grouped by source core: the TARGET core as {ID | one-way-out << 31,
                          //          pairs, cells, first pair} (k_edge_scatter)
  uint2* crec;            // [B][C]   by core ID: {first one-way edge, one-way edges}
  unsigned long long* keys;  // [B][Cpad] sort keys of the seed lists (dpx_binsort.h), one per pair
  uint32_t* histg;        // [B][nb]  cells per histogram bin
  uint32_t* pbcnt;        // [B][nb]  pairs per bin, then the scatter cursor of the bin's segment
  uint32_t* bin_off;      // [B][nb + 1] first slot of every bin's segment in keys / lists
  uint4* lists;           // [B][C][2] the seed lists (k_bin_sort)
  uint32_t* cgrow;        // [B][C]   indexed by core ID: grow index + 1 (k_regsim)
  uint32_t* cursor;       // [B][kCtr] (4 used) allocation cursors of corepairs / oedges, planar cells of the frame
  uint32_t* cellq;        // [B][C]   k_cellgrow: (grow + 1) << 4 | in-edge bits
  int* order_task;        // [B][kCtr] (4 used) task counters of k_order and k_seg_accum
  int T, logT, Cpad;
};

// Words that several XCDs update with atomics inside one launch (the union-find forest, counters, the pair table) are
// never written with plain stores: the per-XCD L2s are not coherent with each other, atomics execute at the memory side,
// and a plain store stays in the storing XCD's L2 as a dirty line whose later write-back would put the old value over an
// atomic's result (seen as a lost union: one frame in ~100 under load).  Write-through (sc1) stores leave no such line.
__device__ __forceinline__ uint32_t cc_load(const uint32_t* p) {
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void cc_store(uint32_t* p, uint32_t v) {
  __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void cc_store64(unsigned long long* p, unsigned long long v) {
  __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// Lock-free union-find with root hooking.  A root is hooked under a lower id with compare-and-swap — it succeeds only
// while the node still is a root — and every other write is a compression (atomicMin towards an ancestor).  So a parent
// pointer, once it leaves its node, only ever moves up that node's own ancestor chain: every value it ever held stays an
// ancestor.  That is what makes the sc1 loads below safe although the per-XCD L2s are not coherent with each other and may
// serve a value another XCD has since replaced — a stale parent is merely an older ancestor, and the compare-and-swap itself
// acts on memory.  (A first version linked with atomicMin on whatever node the find returned and re-united the parent it
// overwrote: one 4K frame in ~100 came out with a core split in two.  Overwriting a non-root's link makes "was an ancestor"
// false, and two such repairs can each take the other's dead link for proof that the trees are already joined.)
__device__ __forceinline__ uint32_t cc_find(uint32_t* p, uint32_t x) {
  uint32_t r = x;
  for (;;) {
    const uint32_t q = cc_load(p + r);
    if (q == r) break;
    r = q;
  }
  while (x != r) {  // compression
    const uint32_t nx = cc_load(p + x);
    if (nx > r) atomicMin(p + x, r);
    x = nx < x ? nx : r;
  }
  return r;
}
__device__ __forceinline__ void cc_union(uint32_t* p, uint32_t a, uint32_t b) {
  for (;;) {
    a = cc_find(p, a);
    b = cc_find(p, b);
    if (a == b) return;
    if (a < b) {
      const uint32_t t = a;
      a = b;
      b = t;
    }
    const uint32_t old = atomicCAS(p + a, a, b);  // hook the higher root under the lower node
    if (old == a) return;
    a = old;  // a was not a root (any more): go on from its parent
  }
}
// slot of pair key K in the frame's table: found or inserted
__device__ __forceinline__ uint32_t pair_slot(uint32_t* hkey, int logT, uint32_t K, bool* isnew) {
  const uint32_t mask = (1u << logT) - 1u;
  uint32_t h = (K * 2654435761u) >> (32 - logT);
  for (;;) {
    const uint32_t k = cc_load(hkey + h);
    if (k == K) return h;
    if (k == kPairEmpty) {
      const uint32_t old = atomicCAS(hkey + h, kPairEmpty, K);
      if (old == kPairEmpty) {  // a new pair (the caller lists it, so that the later launches walk the pairs, not the table)
        *isnew = true;
        return h;
      }
      if (old == K) return h;
    }
    h = (h + 1u) & mask;
  }
}

// ---- cores, first level: 32 x 32-cell tiles joined in LDS ---------------------------------------------------------------
// A workgroup takes one tile, joins its cells along the symmetric edges with the same union-find in LDS (hooking by
// compare-and-swap, parents always lower than children — local index order is cell index order inside a tile), and writes
// every cell's tile root into the forest.  What is left for memory are the edges across tile borders (k_core_link): 1/16 of
// the unions, on trees that are at most (tiles of the frame) deep instead of (wave runs of the frame).
constexpr int kTile = 32;
__device__ __forceinline__ uint32_t lds_find(volatile uint32_t* lp, uint32_t x) {
  for (;;) {
    const uint32_t q = lp[x];
    if (q == x) return x;
    x = q;
  }
}
// grid (tiles across * tiles down, B), 1024 threads; also zeroes the counters of the later launches
__global__ __launch_bounds__(kTile * kTile) void k_core_tiles(Params prm, CoreBufs cb) {
  __shared__ uint32_t lp[kTile * kTile];
  const int frame = blockIdx.y;
  const int tiles_x = (prm.nh + kTile - 1) / kTile;
  const int bx = static_cast<int>(blockIdx.x) % tiles_x, by = static_cast<int>(blockIdx.x) / tiles_x;
  const int t = threadIdx.x, tx = t & (kTile - 1), ty = t >> 5;
  const int X = bx * kTile + tx, Y = by * kTile + ty;
  const bool inside = X < prm.nh && Y < prm.nv;
  const int c = Y * prm.nh + X;
  const size_t fo = static_cast<size_t>(frame) * prm.C;
  const uint32_t m = inside ? cb.meta[fo + c] : 0u;
  const bool planar = (m & F_UNASSIGNED) != 0u;
  const bool sym_r = planar && tx + 1 < kTile && ((m >> 5) & (m >> 11) & 1u) != 0u;  // out- and in-edge R, inside the tile
  const bool sym_d = planar && ty + 1 < kTile && ((m >> 3) & (m >> 9) & 1u) != 0u;   // out- and in-edge D, inside the tile
  {
    // horizontal runs for free: a wave holds two tile rows; a cell's first parent is the first cell of its run of joined
    // right-edges (bit tricks on one ballot; sym_r is never set in a row's last column, so runs end with their row)
    const int lane = t & 63;
    const unsigned long long E = __ballot(sym_r);
    const unsigned long long gaps = ~E & ((1ull << lane) - 1ull);
    const int run_start = gaps ? 64 - static_cast<int>(__builtin_clzll(gaps)) : 0;
    lp[t] = static_cast<uint32_t>(t - lane + run_start);
  }
  __syncthreads();
  if (sym_d) {  // (most of these find both ends joined already: LDS round trips, no memory)
    uint32_t a = static_cast<uint32_t>(t), b = static_cast<uint32_t>(t + kTile);
    for (;;) {
      a = lds_find(lp, a);
      b = lds_find(lp, b);
      if (a == b) break;
      if (a < b) {
        const uint32_t x = a;
        a = b;
        b = x;
      }
      const uint32_t old = atomicCAS(&lp[a], a, b);
      if (old == a) break;
      a = old;
    }
  }
  __syncthreads();
  if (inside) {
    const uint32_t lr = planar ? lds_find(lp, static_cast<uint32_t>(t)) : static_cast<uint32_t>(t);
    cc_store(cb.parent + fo + c, static_cast<uint32_t>((by * kTile + static_cast<int>(lr >> 5)) * prm.nh + bx * kTile + static_cast<int>(lr & (kTile - 1))));
    // the per-cell counters of the preparation start at zero (write-through, like every store that a later launch's
    // atomics must find in memory)
    cc_store(cb.csize + fo + c, 0u);
    cc_store(cb.npairs + fo + c, 0u);
    cc_store(cb.pfill + fo + c, 0u);
    cc_store(cb.noe + fo + c, 0u);
    cc_store(cb.oefill + fo + c, 0u);
    cb.cgrow[fo + c] = 0u;  // (written by one wave of one later launch, read by the launch after it: plain)
  }
  if (blockIdx.x == 0) {  // and the frame's per-bin counters and cursors
    const int nb = prm.hist_B * prm.hist_B;
    for (int i = t; i < nb; i += kTile * kTile) {
      cc_store(cb.histg + static_cast<size_t>(frame) * nb + i, 0u);
      cc_store(cb.pbcnt + static_cast<size_t>(frame) * nb + i, 0u);
    }
    if (t < 4) {
      cc_store(cb.cursor + kCtr * frame + t, 0u);
      __hip_atomic_store(cb.order_task + kCtr * frame + t, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
}

// The unions across tile borders.  grid (tiles, B), 64 threads: lanes 0..31 the tile's bottom row (down-edges), lanes 32..63
// its right column (right-edges).  An edge is left out when the previous cell along the border has the same tile root and
// an edge too, and the two cells across have the same tile root as each other: that union is implied by the previous one.
__global__ __launch_bounds__(64) void k_core_link(Params prm, CoreBufs cb) {
  const int frame = blockIdx.y;
  const int tiles_x = (prm.nh + kTile - 1) / kTile;
  const int bx = static_cast<int>(blockIdx.x) % tiles_x, by = static_cast<int>(blockIdx.x) / tiles_x;
  const int lane = threadIdx.x, k = lane & (kTile - 1);
  const bool down = lane < kTile;
  const int X = bx * kTile + (down ? k : kTile - 1), Y = by * kTile + (down ? kTile - 1 : k);
  const size_t fo = static_cast<size_t>(frame) * prm.C;
  const bool inside = X < prm.nh && Y < prm.nv;
  const int c = Y * prm.nh + X;
  const int step = down ? prm.nh : 1;   // to the cell across the border
  const int prev = down ? 1 : prm.nh;   // back along the border
  const uint32_t m = inside ? cb.meta[fo + c] : 0u;
  const bool edge = (m & F_UNASSIGNED) != 0u && (down ? ((m >> 3) & (m >> 9) & 1u) : ((m >> 5) & (m >> 11) & 1u)) != 0u;
  uint32_t* p = cb.parent + fo;
  // (the tile roots are the previous launch's plain results: ordinary loads)
  const uint32_t mine = edge ? cb.parent[fo + c] : 0xFFFFFFFFu, across = edge ? cb.parent[fo + c + step] : 0xFFFFFFFFu;
  const uint32_t mine_prev = static_cast<uint32_t>(__shfl_up(static_cast<int>(mine), 1)), across_prev = static_cast<uint32_t>(__shfl_up(static_cast<int>(across), 1));
  (void)prev;
  const bool implied = edge && k > 0 && mine_prev == mine && across_prev == across;  // (lane - 1 is the previous cell of the same border: k > 0)
  if (edge && !implied) cc_union(p, static_cast<uint32_t>(c), static_cast<uint32_t>(c + step));
}

// Same-address atomics are served one after the other at the memory side (a few hundred ns each on this part, whether or
// not the wave waits for the result), so a counter that every wave of a frame bumps costs (waves per frame) x that — 1 300
// waves at 3840x2160, and a plane that covers most of a frame has every wave add to the same core and the same pair.  The
// preparation launches therefore run 1024-thread workgroups and combine in LDS first: one atomic per workgroup and address.
constexpr int kPrepT = 1024;
// groups of a wave's lanes with equal keys: the group's first lane and its size, for every lane with `valid`
__device__ __forceinline__ void wave_groups(uint32_t key, bool valid, int lane, int& lead, int& cnt) {
  lead = lane;
  cnt = 0;
  unsigned long long todo = __ballot(valid);
  while (todo) {
    const int l = static_cast<int>(__builtin_ctzll(todo));
    const uint32_t k0 = static_cast<uint32_t>(__builtin_amdgcn_readlane(static_cast<int>(key), l));
    const unsigned long long same = __ballot(valid && key == k0);
    if (valid && key == k0) {
      lead = l;
      cnt = __popcll(same);
    }
    todo &= ~same;
  }
}
constexpr int kAggN = 128;   // LDS slots for the cores / pairs that whole groups of a wave's lanes belong to
constexpr int kAggMin = 16;  // lanes of a wave in one core (pair) from which the group goes through LDS
// root of every cell; per core: cells, one-way edges out; per (core, bin) pair: cells and leader.  grid (ceil(C / 1024), B)
__global__ __launch_bounds__(kPrepT) void k_core_flatten(Params prm, CoreBufs cb) {
  __shared__ uint32_t s_ckey[kAggN], s_cval[kAggN], s_pkey[kAggN], s_pcnt[kAggN];
  __shared__ unsigned long long s_pbest[kAggN];
  __shared__ uint32_t s_misc[4];  // planar cells, new table slots of the workgroup, their first list position
  const int frame = blockIdx.y;
  const int tid = threadIdx.x;
  const int c = static_cast<int>(blockIdx.x) * kPrepT + tid;
  const size_t fo = static_cast<size_t>(frame) * prm.C;
  const int lane = tid & 63;
  const unsigned long long lt_mask = (1ull << lane) - 1ull;
  if (tid < kAggN) {
    s_ckey[tid] = kPairEmpty;
    s_cval[tid] = 0u;
    s_pkey[tid] = kPairEmpty;
    s_pcnt[tid] = 0u;
    s_pbest[tid] = ~0ull;
  }
  if (tid < 4) s_misc[tid] = 0u;
  __syncthreads();
  const uint32_t m = c < prm.C ? cb.meta[fo + c] : 0u;
  const bool planar = (m & F_UNASSIGNED) != 0u;
  uint32_t* p = cb.parent + fo;
  uint32_t r = 0u, K = kPairEmpty;
  int nout = 0;
  // Cells of one tile component share their entry (the tile root, k_core_tiles): one lane per distinct entry of the wave walks
  // the forest from there (and compresses its path), the others take the root from that lane and put it into their own entry
  // with a write-through store: no pointer chase and no atomic for most cells of a plane, and the stores of a wave's 64
  // consecutive entries travel together.  (Nothing hooks roots during this launch, and a compression that meets the store
  // can only write an ancestor: either order leaves the root, which is the minimum.)
  {
    const uint32_t tr = planar ? cb.parent[fo + c] : 0u;  // (the entry as the earlier launches left it: at least the tile root)
    int lead_t, n_t;
    wave_groups(tr, planar, lane, lead_t, n_t);
    if (planar && lane == lead_t) r = cc_find(p, tr);
    r = static_cast<uint32_t>(__shfl(static_cast<int>(r), lead_t));
    if (planar && tr != r) cc_store(p + c, r);
  }
  if (planar) {
    const int bn = static_cast<int>(m >> META_BIN_SHIFT);
    if (bn >= 1 && bn - 1 < prm.hist_B * prm.hist_B) K = (r << kPairBinBits) | static_cast<uint32_t>(bn - 1);
    const uint32_t ow = ((m >> 2) & ~(m >> 8)) & 0xFu;  // one-way out-edges
    if (ow) {
#pragma unroll
      for (int dir = 0; dir < 4; ++dir)
        if ((ow >> dir) & 1u) {
          const int v = c + (dir == 0 ? -prm.nh : (dir == 1 ? prm.nh : (dir == 2 ? -1 : 1)));
          if (cc_find(p, static_cast<uint32_t>(v)) != r) ++nout;
        }
    }
  }
  if (nout) atomicAdd(cb.noe + fo + r, static_cast<uint32_t>(nout));
  // neighbouring cells mostly share their core and bin: one operation per distinct core / (core, bin) of the wave, in LDS
  // where the group is large (a plane's interior: the other waves of the workgroup have the same), in memory otherwise.  The
  // groups are found first (registers only); then every group's first lane works at the same time — a loop with an atomic
  // per group would wait for up to 64 round trips to memory in a row where no two cells share a core.
  const unsigned long long pmask = __ballot(planar);
  if (lane == 0 && pmask) atomicAdd(&s_misc[0], static_cast<uint32_t>(__popcll(pmask)));
  int lead_r, n_r;
  wave_groups(r, planar, lane, lead_r, n_r);
  if (planar && lane == lead_r) {
    bool in_lds = false;
    if (n_r >= kAggMin) {
      const uint32_t hs = (r * 2654435761u) >> 25;  // 7 bits: kAggN slots
      const uint32_t old = atomicCAS(&s_ckey[hs], kPairEmpty, r);
      if (old == kPairEmpty || old == r) {
        atomicAdd(&s_cval[hs], static_cast<uint32_t>(n_r));
        in_lds = true;
      }
    }
    if (!in_lds) atomicAdd(cb.csize + fo + r, static_cast<uint32_t>(n_r));
  }
  uint32_t* hkey = cb.hkey + static_cast<size_t>(frame) * cb.T;
  const bool has_k = K != kPairEmpty;
  const unsigned long long val = has_k ? (static_cast<unsigned long long>(sort_mse_key(cb.mse_c[fo + c])) << kCellBits) | static_cast<unsigned long long>(c)
                                       : ~0ull;
  int lead_k, n_k;
  wave_groups(K, has_k, lane, lead_k, n_k);
  const bool big_k = has_k && n_k >= kAggMin;
  // a large group's leader (minimum of mse key | cell) as one value, at the group's first lane (at most four such groups)
  unsigned long long vmin = val;
  for (unsigned long long bl = __ballot(big_k && lane == lead_k); bl; bl &= bl - 1ull) {
    const int l = static_cast<int>(__builtin_ctzll(bl));
    unsigned long long v = (has_k && lead_k == l) ? val : ~0ull;
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
      const unsigned long long o = __shfl_xor(v, off);
      v = o < v ? o : v;
    }
    if (lane == l) vmin = v;
  }
  bool mynew = false;
  uint32_t newslot = 0u;
  if (has_k && lane == lead_k) {
    bool in_lds = false;
    if (big_k) {
      const uint32_t hs = (K * 2654435761u) >> 25;
      const uint32_t old = atomicCAS(&s_pkey[hs], kPairEmpty, K);
      if (old == kPairEmpty || old == K) {
        atomicAdd(&s_pcnt[hs], static_cast<uint32_t>(n_k));
        atomicMin(&s_pbest[hs], vmin);
        in_lds = true;
      }
    }
    if (!in_lds) {
      newslot = pair_slot(hkey, cb.logT, K, &mynew);
      atomicAdd(cb.hcnt + static_cast<size_t>(frame) * cb.T + newslot, static_cast<uint32_t>(n_k));
      if (big_k) {
        unsigned long long* hb = cb.hbest + static_cast<size_t>(frame) * cb.T + newslot;
        if (vmin < __hip_atomic_load(hb, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMin(hb, vmin);
      }
    }
  }
  {
    const uint32_t gslot = static_cast<uint32_t>(__shfl(static_cast<int>(newslot), has_k ? lead_k : lane));
    if (has_k && !big_k) {
      unsigned long long* hb = cb.hbest + static_cast<size_t>(frame) * cb.T + gslot;
      if (val < __hip_atomic_load(hb, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMin(hb, val);
    }
  }
  __syncthreads();
  // what the workgroup combined goes to memory: one atomic per core / pair
  bool mynew2 = false;
  uint32_t newslot2 = 0u;
  if (tid < kAggN) {
    if (s_ckey[tid] != kPairEmpty) atomicAdd(cb.csize + fo + s_ckey[tid], s_cval[tid]);
    if (s_pkey[tid] != kPairEmpty) {
      newslot2 = pair_slot(hkey, cb.logT, s_pkey[tid], &mynew2);
      atomicAdd(cb.hcnt + static_cast<size_t>(frame) * cb.T + newslot2, s_pcnt[tid]);
      unsigned long long* hb = cb.hbest + static_cast<size_t>(frame) * cb.T + newslot2;
      const unsigned long long v = s_pbest[tid];
      if (v < __hip_atomic_load(hb, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMin(hb, v);
    }
  }
  // the slots this workgroup has newly occupied go to the frame's slot list: one atomic per workgroup
  const unsigned long long nm1 = __ballot(mynew), nm2 = __ballot(mynew2);
  const int n1 = __popcll(nm1), n2 = __popcll(nm2);
  uint32_t woff = 0u;
  if (lane == 0 && n1 + n2) woff = atomicAdd(&s_misc[1], static_cast<uint32_t>(n1 + n2));
  woff = static_cast<uint32_t>(__builtin_amdgcn_readfirstlane(static_cast<int>(woff)));
  __syncthreads();
  if (tid == 0) {
    if (s_misc[0]) atomicAdd(cb.cursor + kCtr * frame + 2, s_misc[0]);
    if (s_misc[1]) s_misc[2] = atomicAdd(cb.cursor + kCtr * frame + 3, s_misc[1]);
  }
  __syncthreads();
  const uint32_t base = s_misc[2] + woff;
  if (mynew) cb.hslots[fo + base + static_cast<uint32_t>(__popcll(nm1 & lt_mask))] = newslot;
  if (mynew2) cb.hslots[fo + base + static_cast<uint32_t>(n1 + __popcll(nm2 & lt_mask))] = newslot2;
}

// grid (ceil(C / 1024), B) over the list of occupied slots, 4 * bins bytes of LDS: every pair counts for its core and its bin
__global__ __launch_bounds__(kPrepT) void k_pair_count(Params prm, CoreBufs cb) {
  extern __shared__ uint32_t s_bins[];
  const int frame = blockIdx.y;
  const int npair = static_cast<int>(cb.cursor[kCtr * frame + 3]);
  if (static_cast<int>(blockIdx.x) * kPrepT >= npair) return;  // (the whole workgroup)
  const int nb = prm.hist_B * prm.hist_B;
  for (int b = threadIdx.x; b < nb; b += kPrepT) s_bins[b] = 0u;
  __syncthreads();
  const int li = static_cast<int>(blockIdx.x) * kPrepT + threadIdx.x;
  if (li < npair) {
    const uint32_t K = cb.hkey[static_cast<size_t>(frame) * cb.T + cb.hslots[static_cast<size_t>(frame) * prm.C + li]];
    atomicAdd(cb.npairs + static_cast<size_t>(frame) * prm.C + (K >> kPairBinBits), 1u);
    atomicAdd(&s_bins[K & kPairBinMask], 1u);
  }
  __syncthreads();
  for (int b = threadIdx.x; b < nb; b += kPrepT)
    if (s_bins[b]) atomicAdd(cb.pbcnt + static_cast<size_t>(frame) * nb + b, s_bins[b]);
}

// grid (ceil(C / 1024), B): room for every core's pairs and one-way edges; one atomic per workgroup and counter
__global__ __launch_bounds__(kPrepT) void k_core_alloc(Params prm, CoreBufs cb) {
  __shared__ uint32_t s_tot[2 * (kPrepT / 64)];
  __shared__ uint32_t s_base[2];
  const int frame = blockIdx.y;
  const int c = static_cast<int>(blockIdx.x) * kPrepT + threadIdx.x;
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const size_t fo = static_cast<size_t>(frame) * prm.C;
  const bool root = c < prm.C && cb.csize[fo + c] != 0u;  // (only roots of planar cores have a size)
  // the workgroup's cores take consecutive ranges: prefix sums over the lanes, then over the waves
  const uint32_t np = root ? cb.npairs[fo + c] : 0u, ne = root ? cb.noe[fo + c] : 0u;
  uint32_t sp = np, se = ne;
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    const uint32_t a = static_cast<uint32_t>(__shfl_up(static_cast<int>(sp), off)), b = static_cast<uint32_t>(__shfl_up(static_cast<int>(se), off));
    if (lane >= off) {
      sp += a;
      se += b;
    }
  }
  if (lane == 63) {
    s_tot[2 * wid] = sp;
    s_tot[2 * wid + 1] = se;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    uint32_t tp = 0u, te = 0u;
    for (int w = 0; w < kPrepT / 64; ++w) {
      const uint32_t a = s_tot[2 * w], b = s_tot[2 * w + 1];
      s_tot[2 * w] = tp;
      s_tot[2 * w + 1] = te;
      tp += a;
      te += b;
    }
    s_base[0] = tp ? atomicAdd(cb.cursor + kCtr * frame, tp) : 0u;
    s_base[1] = te ? atomicAdd(cb.cursor + kCtr * frame + 1, te) : 0u;
  }
  __syncthreads();
  if (!root) return;
  const uint32_t ps = s_base[0] + s_tot[2 * wid] + sp - np, es = ne ? s_base[1] + s_tot[2 * wid + 1] + se - ne : 0u;
  cb.pstart[fo + c] = ps;
  cb.oestart[fo + c] = es;
  cb.crec[fo + ps] = make_uint2(es, ne);
}

// grid (ceil(C / 1024), B) over the list of occupied slots, 8 * bins bytes of LDS: pairs into their core's list + their sort
// key into its bin's segment.  A workgroup's pairs of one bin take consecutive slots of the segment: ranks from an LDS counter
// per bin, then one atomic per (workgroup, bin) on the frame's cursor — a hot bin holds thousands of pairs.
__global__ __launch_bounds__(kPrepT) void k_pair_scatter(Params prm, CoreBufs cb) {
  extern __shared__ uint32_t s_bins[];  // [bins] pairs of the workgroup -> first slot; [bins] their cells
  const int frame = blockIdx.y;
  const int npair = static_cast<int>(cb.cursor[kCtr * frame + 3]);
  if (static_cast<int>(blockIdx.x) * kPrepT >= npair) return;  // (the whole workgroup)
  const int nb = prm.hist_B * prm.hist_B;
  uint32_t* s_cnt = s_bins;
  uint32_t* s_cells = s_bins + nb;
  for (int b = threadIdx.x; b < 2 * nb; b += kPrepT) s_bins[b] = 0u;
  __syncthreads();
  const int i = static_cast<int>(blockIdx.x) * kPrepT + threadIdx.x;
  const size_t fo = static_cast<size_t>(frame) * prm.C;
  const size_t bo = static_cast<size_t>(frame) * nb;
  const bool act = i < npair;
  size_t to = 0;
  uint32_t root = 0u, bin = 0u, cnt = 0u, rank = 0u;
  if (act) {
    to = static_cast<size_t>(frame) * cb.T + cb.hslots[fo + i];
    const uint32_t K = cb.hkey[to];
    root = K >> kPairBinBits;
    bin = K & kPairBinMask;
    cnt = cb.hcnt[to];
    rank = atomicAdd(&s_cnt[bin], 1u);
    atomicAdd(&s_cells[bin], cnt);
  }
  __syncthreads();
  for (int b = threadIdx.x; b < nb; b += kPrepT) {
    const uint32_t n = s_cnt[b];
    if (n) {
      s_cnt[b] = atomicAdd(cb.pbcnt + bo + b, n);
      atomicAdd(cb.histg + bo + b, s_cells[b]);
    }
  }
  __syncthreads();
  if (act) {
    const uint32_t j = cb.pstart[fo + root] + atomicAdd(cb.pfill + fo + root, 1u);
    cb.corepairs[fo + j] = bin | (cnt << kPairBinBits);
    const uint32_t at = cb.bin_off[bo + frame + bin] + s_cnt[bin] + rank;
    cb.keys[static_cast<size_t>(frame) * cb.Cpad + at] = cb.hbest[to];  // mse key << 19 | leader cell (k_bin_sort orders the segment)
    // the slot is empty again for the next call (write-through: the next call's atomics must find these values in memory)
    cc_store(cb.hkey + to, kPairEmpty);
    cc_store(cb.hcnt + to, 0u);
    cc_store64(cb.hbest + to, ~0ull);
  }
}

// grid (ceil(C / 256), B), after k_pair_scatter: every one-way edge that leaves its core goes to its source core's list as
// the TARGET core's {ID | one-way-out << 31, pairs, cells, first pair}: a grow that follows the edge to a one-bin core
// without further one-way edges — the usual case — needs nothing else
__global__ __launch_bounds__(256) void k_edge_scatter(Params prm, CoreBufs cb) {
  const int frame = blockIdx.y;
  const int i = static_cast<int>(blockIdx.x) * 256 + threadIdx.x;
  if (i >= prm.C) return;
  const size_t fo = static_cast<size_t>(frame) * prm.C;
  const uint32_t m = cb.meta[fo + i];
  const uint32_t ow = (m & F_UNASSIGNED) ? (((m >> 2) & ~(m >> 8)) & 0xFu) : 0u;
  if (!ow) return;
  const uint32_t r = cb.parent[fo + i];
#pragma unroll
  for (int dir = 0; dir < 4; ++dir)
    if ((ow >> dir) & 1u) {
      const int v = i + (dir == 0 ? -prm.nh : (dir == 1 ? prm.nh : (dir == 2 ? -1 : 1)));
      const uint32_t rv = cb.parent[fo + v];
      if (rv != r) {
        const uint32_t ps = cb.pstart[fo + rv];
        cb.oedges[4 * fo + cb.oestart[fo + r] + atomicAdd(cb.oefill + fo + r, 1u)] =
            make_uint4(ps | (cb.noe[fo + rv] ? kCoreOut : 0u), cb.npairs[fo + rv], cb.csize[fo + rv], cb.corepairs[fo + ps]);
      }
    }
}

// grid (B), 64 threads: pairs per bin -> first slot of every bin's segment (bin_off[nb] = pairs of the frame); the counts
// become the scatter cursors
__global__ __launch_bounds__(64) void k_bin_scan(Params prm, CoreBufs cb) {
  const int frame = blockIdx.x, lane = threadIdx.x;
  const int nb = prm.hist_B * prm.hist_B;
  uint32_t* cnt = cb.pbcnt + static_cast<size_t>(frame) * nb;
  uint32_t* bo = cb.bin_off + static_cast<size_t>(frame) * (nb + 1);
  int carry = 0;
  for (int base = 0; base < nb; base += 64) {
    const int b = base + lane;
    const int v = b < nb ? static_cast<int>(cnt[b]) : 0;
    const int incl = wave_incl_scan_i32(v);
    if (b < nb) {
      bo[b] = static_cast<uint32_t>(carry + incl - v);
      cc_store(cnt + b, 0u);
    }
    carry += __builtin_amdgcn_readlane(incl, 63);
  }
  if (lane == 0) bo[nb] = static_cast<uint32_t>(carry);
}

// grid (G, B), kBinSortT threads, bins dealt round robin: one bin's segment of pair keys (mse key << 19 | leader cell) sorted ascending — the
// reference's "first minimum-MSE cell, lowest id among equals" (plane_extractor.cpp:309-316) is then the first live entry
// — and written out as the seed list: one 32-byte entry per pair, {leader cell | one-way-out << 31, core ID, pairs of the
// core, cells of the core} and the first four (bin | count << 13) pairs of the core, so that the sequence launch needs no
// further load for a core of up to four bins (a core with one-way edges out: two pairs, first edge, number of edges).  Segments of up to kBinSortMax keys are sorted in LDS (all-ascending bitonic
// network on exactly n keys, as in dpx_regions2.h's preparation), longer ones in place in global memory by the same
// workgroup.
constexpr int kBinSortT = 256, kBinSortMax = 1024;       // the core path's form: pairs per bin are few (574 in the top bin of the noisiest 4K test frame); 8 KB of LDS, so that all of a batch's workgroups are resident at once
constexpr int kCellSortT = 1024, kCellSortMax = 16384;   // the cell-level form: a clean frame has most of its cells in one bin
constexpr int kCellSortSmall = 1024;                     // segments up to here: the launch of many small workgroups
constexpr int kBinListMax = 1024;                        // bins of more than 64 keys a launch can list (LDS)
// CELLS = the cell-level tiers' form: the sorted segment's cell ids go to `order` (cb.keys, cb.bin_off, cb.Cpad are all it reads).
// `cap` = keys the launch's dynamic LDS holds (8 bytes each).
// The launch takes the segments of more than `above` keys (0: all; then segments of up to 64 keys are a wave's work).
template <bool CELLS>
__global__ __launch_bounds__(CELLS ? kCellSortT : kBinSortT) void k_bin_sort(Params prm, CoreBufs cb, uint32_t* __restrict__ order, int cap, int above) {
  const int kBinSortT = static_cast<int>(blockDim.x);  // (this launch's workgroup size: up to the bound above)
  const int kBinSortMax = cap;
  const int frame = blockIdx.y, tid = threadIdx.x;
  const int nb = prm.hist_B * prm.hist_B;
  const uint32_t* bo = cb.bin_off + static_cast<size_t>(frame) * (nb + 1);
  extern __shared__ __align__(16) unsigned char sk_raw[];
  unsigned long long* sk = reinterpret_cast<unsigned long long*>(sk_raw);
  const size_t fo = static_cast<size_t>(frame) * prm.C;
  // a list entry of the core path from its sorted key (CELLS: the cell id into `order`)
  auto emit = [&](int at, unsigned long long key) {
    const uint32_t cell = static_cast<uint32_t>(key & kCellMask);
    if constexpr (CELLS) {
      order[fo + at] = cell;
    } else {
      const uint32_t r = cb.parent[fo + cell];
      const uint32_t ps = cb.pstart[fo + r], np = cb.npairs[fo + r];
      uint4 pr = make_uint4(0u, 0u, 0u, 0u);
      const uint32_t* cp = cb.corepairs + fo + ps;
      pr.x = cp[0];
      if (np > 1) pr.y = cp[1];
      if (np > 2) pr.z = cp[2];
      if (np > 3) pr.w = cp[3];
      const uint32_t ne = cb.noe[fo + r];
      if (ne) {  // a core with one-way edges out: two pairs + where its edges are
        pr.z = cb.oestart[fo + r];
        pr.w = ne;
      }
      cb.lists[2 * (fo + at)] = make_uint4(cell | (ne ? kCoreOut : 0u), ps, np, cb.csize[fo + r]);
      cb.lists[2 * (fo + at) + 1] = pr;
    }
  };
  // Bins of at most 64 keys — most bins of a noisy frame — are a WAVE's work: the keys in registers, a bitonic network
  // over the lanes, no LDS and no workgroup barrier; the waves of the launch take such bins independently.
  if (above == 0) {
    const int lane = tid & 63, wid = tid >> 6, nwaves = kBinSortT >> 6;
    for (int bin = static_cast<int>(blockIdx.x) * nwaves + wid; bin < nb; bin += static_cast<int>(gridDim.x) * nwaves) {
      const int s0 = static_cast<int>(bo[bin]), n = static_cast<int>(bo[bin + 1]) - s0;
      if (n <= 0 || n > 64) continue;
      unsigned long long v = lane < n ? cb.keys[static_cast<size_t>(frame) * cb.Cpad + s0 + lane] : ~0ull;
      if (n > 1) {
#pragma unroll
        for (int k = 2; k <= 64; k <<= 1) {
#pragma unroll
          for (int j = k >> 1; j >= 1; j >>= 1) {
            const unsigned long long o = __shfl_xor(v, j);
            const bool up = (lane & k) == 0;             // this block sorts ascending
            const bool low = (lane & j) == 0;            // this lane keeps the smaller of the pair (in an ascending block)
            const bool take_min = up == low;
            v = take_min ? (o < v ? o : v) : (o > v ? o : v);
          }
        }
      }
      if (lane < n) emit(s0 + lane, v);
    }
  }
  // the larger bins: the workgroup, keys in LDS.  Which bins those are is found by all threads at once (a loop over the
  // frame's bins with two dependent loads each, just to skip most of them, was most of this launch's time)
  __shared__ int s_big[kBinListMax];
  __shared__ int s_nbig;
  if (tid == 0) s_nbig = 0;
  __syncthreads();
  for (int b = tid; b < nb; b += kBinSortT) {
    const int n = static_cast<int>(bo[b + 1]) - static_cast<int>(bo[b]);
    // (CELLS: the segments beyond this launch's LDS are the next launch's)
    if (static_cast<unsigned>(b) % gridDim.x == blockIdx.x && n > 64 && n > above && !(CELLS && above == 0 && n > kBinSortMax)) {  // (this workgroup's bins: every gridDim.x-th)
      const int k = atomicAdd(&s_nbig, 1);
      if (k < kBinListMax) s_big[k] = b;
    }
  }
  __syncthreads();
  const int nbig_bins = s_nbig;
  // (more listed bins than the list holds cannot happen for C < 2^19 cells and > 64 keys each x kBinListMax; if it did, the
  // plain loop over the workgroup's bins takes over)
  const bool listed = nbig_bins <= kBinListMax;
  for (int it = 0; it < (listed ? nbig_bins : (nb - static_cast<int>(blockIdx.x) + static_cast<int>(gridDim.x) - 1) / static_cast<int>(gridDim.x)); ++it) {
  const int bin = listed ? s_big[it] : static_cast<int>(blockIdx.x) + it * static_cast<int>(gridDim.x);
  const int s0 = static_cast<int>(bo[bin]), n = static_cast<int>(bo[bin + 1]) - s0;
  if (n <= 64 || n <= above || (CELLS && above == 0 && n > kBinSortMax)) continue;
  __syncthreads();  // (sk is reused)
  unsigned long long* g = cb.keys + static_cast<size_t>(frame) * cb.Cpad + s0;
  const bool in_lds = n <= kBinSortMax;
  if (in_lds) {
    for (int i = tid; i < n; i += kBinSortT) sk[i] = g[i];
    __syncthreads();
  }
  if (n > 1) {
    auto ld = [&](int i) { return in_lds ? sk[i] : __hip_atomic_load(g + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); };
    auto st = [&](int i, unsigned long long v) {
      if (in_lds) sk[i] = v;
      else __hip_atomic_store(g + i, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    };
    auto cex = [&](int lo, int hi) {
      if (hi < n) {
        const unsigned long long a = ld(lo), b = ld(hi);
        if (a > b) {
          st(lo, b);
          st(hi, a);
        }
      }
    };
    auto sync = [&]() {
      if (!in_lds) __threadfence();
      __syncthreads();
    };
    const int half = (n + 1) >> 1;
    for (int k = 2; (k >> 1) < n; k <<= 1) {
      const int hk = k >> 1;
      for (int pi = tid; pi < ((n + k - 1) / k) * hk; pi += kBinSortT) {  // mirror step: i against the block's k-1-i
        const int blk = pi / hk, off = pi - blk * hk;
        cex(blk * k + off, blk * k + (k - 1 - off));
      }
      sync();
      for (int j = k >> 2; j >= 1; j >>= 1) {  // half-cleaners: i against i + j
        for (int pi = tid; pi < half + j; pi += kBinSortT) cex(((pi & ~(j - 1)) << 1) | (pi & (j - 1)), (((pi & ~(j - 1)) << 1) | (pi & (j - 1))) | j);
        sync();
      }
    }
  }
  for (int i = tid; i < n; i += kBinSortT) emit(s0 + i, in_lds ? sk[i] : __hip_atomic_load(g + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
  }
}

// ---- the same counting sort for the cell-level tiers of dpx_regions2.h (seed lists of CELLS: bin, then MSE, then cell id) ----
// k_cell_bincount: cells per bin; k_bin_scan: segment offsets; k_cell_scatter: every planar cell's (mse key << 19 | cell) into
// its bin's segment; k_bin_sort_cells: each segment sorted, cell ids written to order[B][C].  Replaces the frame-wide bitonic
// network of dpx_binsort.h (O(log^2 C) chip-wide passes over Cpad slots for keys whose top field is one of a few hundred bins).
// Both launches: grid (ceil(C / 1024), B), 1024 threads, LDS per-bin counters — a clean frame has 10 000 cells in ONE bin, and
// one atomic per cell (or per wave) on that bin's counter is served one at a time at the memory side.
__global__ __launch_bounds__(kPrepT) void k_cell_bincount(Params prm, const uint32_t* __restrict__ meta, uint32_t* __restrict__ pbcnt) {
  extern __shared__ uint32_t s_bins[];
  const int frame = blockIdx.y;
  const int c = static_cast<int>(blockIdx.x) * kPrepT + threadIdx.x;
  const int nb = prm.hist_B * prm.hist_B;
  for (int b = threadIdx.x; b < nb; b += kPrepT) s_bins[b] = 0u;
  __syncthreads();
  if (c < prm.C) {
    const uint32_t m = meta[static_cast<size_t>(frame) * prm.C + c];
    const int bn = static_cast<int>(m >> META_BIN_SHIFT);
    if ((m & F_UNASSIGNED) && bn >= 1 && bn - 1 < nb) atomicAdd(&s_bins[bn - 1], 1u);
  }
  __syncthreads();
  for (int b = threadIdx.x; b < nb; b += kPrepT)
    if (s_bins[b]) atomicAdd(pbcnt + static_cast<size_t>(frame) * nb + b, s_bins[b]);
}
__global__ __launch_bounds__(kPrepT) void k_cell_scatter(Params prm, const uint32_t* __restrict__ meta, const float* __restrict__ mse_c,
                                                         uint32_t* __restrict__ pbcnt, const uint32_t* __restrict__ bin_off,
                                                         unsigned long long* __restrict__ keys, int Cpad) {
  extern __shared__ uint32_t s_bins[];  // cells of the workgroup per bin -> first slot of theirs in the bin's segment
  const int frame = blockIdx.y;
  const int c = static_cast<int>(blockIdx.x) * kPrepT + threadIdx.x;
  const int nb = prm.hist_B * prm.hist_B;
  const size_t fo = static_cast<size_t>(frame) * prm.C;
  for (int b = threadIdx.x; b < nb; b += kPrepT) s_bins[b] = 0u;
  __syncthreads();
  int bin = -1;
  uint32_t rank = 0u;
  if (c < prm.C) {
    const uint32_t m = meta[fo + c];
    const int bn = static_cast<int>(m >> META_BIN_SHIFT);
    if ((m & F_UNASSIGNED) && bn >= 1 && bn - 1 < nb) {
      bin = bn - 1;
      rank = atomicAdd(&s_bins[bin], 1u);
    }
  }
  __syncthreads();
  for (int b = threadIdx.x; b < nb; b += kPrepT)
    if (s_bins[b]) s_bins[b] = atomicAdd(pbcnt + static_cast<size_t>(frame) * nb + b, s_bins[b]);
  __syncthreads();
  if (bin >= 0) {
    const uint32_t at = bin_off[static_cast<size_t>(frame) * (nb + 1) + bin] + s_bins[bin] + rank;
    keys[static_cast<size_t>(frame) * Cpad + at] = (static_cast<unsigned long long>(sort_mse_key(mse_c[fo + c])) << kCellBits) | static_cast<unsigned long long>(c);
  }
}

// ---- the sequence of grows --------------------------------------------------------------------------------------------
struct SimBufs {
  const uint4* lists;                 // [B][C][2]  seed lists (k_sort_lists_pairs)
  const uint32_t* bin_off;            // [B][nb + 1]
  const uint32_t* histg;              // [B][nb]
  const uint32_t* corepairs;          // [B][C]
  const uint4* oedges;                // [B][4C]
  const uint2* crec;                  // [B][C]
  const uint32_t* cursor;             // [B][kCtr]: pairs, one-way edges, planar cells of the frame
  uint32_t* cgrow;                    // [B][C]  by core ID: grow + 1
  uint32_t* worklist;                 // [B][3C] irregular grows: IDs of the cores taken [C], cores still to expand [C][2]
  uint4* cand_grow;                   // [B][pcap]  grow index, queue range [y, z), seed | kCandWide
  uint32_t* biglist;                  // [B][pcap]  candidate indices of the grows k_order gives a whole wave
  int* hdr_out;                       // [B][48] the header the tail launch starts from (k_regions2 hdr[] layout) + counts
  unsigned long long* dbg;            // diagnostic: [B][kDbgSlots]
  int pcap;
};
constexpr uint32_t kCandWide = 0x80000000u;  // cand_grow.w: the grow is ordered by a whole wave (it is in biglist)
// hdr_out words: 0 planar cells, 1 activations (queue tail), 2 grows, 3 candidates, 4 entries of biglist, 5 of them huge
// (a whole workgroup's work in k_order: they are listed from the END of biglist backwards)
constexpr int kHdrBig = 4, kHdrHuge = 5;
constexpr int kHugeMin = 1024;
#ifndef DPX_S3_WIN
#define DPX_S3_WIN 8
#endif
constexpr int kWin = DPX_S3_WIN;  // seed-list entries per bin kept in LDS
#ifndef DPX_S3_SCAN
#define DPX_S3_SCAN 4
#endif
constexpr int kScan = DPX_S3_SCAN;  // of which the sequence reads so many at a time
constexpr int kEdgePre = 4;  // one-way edges of an event's core fetched ahead of its turn
constexpr int kWideLds = 64; // cores of an irregular grow listed in LDS (more: in global memory)

constexpr int kReqRing = 256;  // refill requests in flight from the sequence's wave to its helper (bins, 16 bits each)
struct S3Layout {
  uint32_t hist, cursor, lo, hi, bclaim, nzlist, evb, rrec, rsrc, idata, edata, tk, td, req, win, cbits, end;
};
__host__ __device__ inline S3Layout s3_layout(int nb, int C) {
  S3Layout L{};
  uint32_t o = 256;  // header words
  L.hist = o; o += 4u * nb;
  L.cursor = o; o += 4u * nb;
  L.lo = o; o += 4u * nb;
  L.hi = o; o += 4u * nb;
  L.bclaim = o; o += 4u * nb;
  L.nzlist = o; o += 2u * nb;
  o = (o + 15u) & ~15u;
  L.evb = o; o += 128;
  L.rrec = o; o += 64 * 8;
  L.rsrc = o; o += 64;
  L.idata = o; o += 64 * 4;
  L.edata = o; o += 64 * kEdgePre * 16;
  L.tk = o; o += kWideLds * 4;
  L.td = o; o += kWideLds * 8;
  L.req = o; o += 2u * kReqRing;
  L.win = o; o += 32u * kWin * static_cast<uint32_t>(nb);
  L.cbits = o; o += 4u * static_cast<uint32_t>((C + 31) / 32 + 1);
  L.end = (o + 15u) & ~15u;
  return L;
}

template <bool DBG>
__global__ __launch_bounds__(128) void k_regsim(Params prm, SimBufs sb) {
  WaveTrace trace_(TK_GROW);
#if DPX_REGIONS_PRIO
  __builtin_amdgcn_s_setprio(DPX_REGIONS_PRIO);
#endif
  const unsigned long long t_start = DBG ? static_cast<unsigned long long>(__builtin_amdgcn_s_memtime()) : 0ull;
  unsigned long long n_rounds = 0, n_wide = 0, n_irreg = 0, n_refill = 0, n_multi = 0, n_skip = 0, t_setup = 0;
  unsigned long long a_ev = 0, a_seed = 0, a_items = 0, a_walk = 0, a_commit = 0, a_wide = 0, ts0 = 0, ts1 = 0;
#define S3_STAMP(acc) do { if (DBG) { ts1 = static_cast<unsigned long long>(__builtin_amdgcn_s_memtime()); acc += ts1 - ts0; ts0 = ts1; } } while (0)
  const int frame = blockIdx.x, lane = threadIdx.x & 63, wid = threadIdx.x >> 6;  // wave 0: the sequence; wave 1: the windows' helper
  const int C = prm.C;
  const int nb = prm.hist_B * prm.hist_B;
  const unsigned long long lt_mask = (1ull << lane) - 1ull;
  extern __shared__ __align__(16) unsigned char smem[];
  const S3Layout lay = s3_layout(nb, C);
  int* hdr = reinterpret_cast<int*>(smem);
  int* hist = reinterpret_cast<int*>(smem + lay.hist);
  int* cursor = reinterpret_cast<int*>(smem + lay.cursor);  // END of bin b's list
  int* lo = reinterpret_cast<int*>(smem + lay.lo);          // first possibly live entry of bin b's list
  int* hi = reinterpret_cast<int*>(smem + lay.hi);          // entries [lo, hi) are in the bin's window (slot = index & 3)
  uint32_t* bclaim = reinterpret_cast<uint32_t*>(smem + lay.bclaim);
  uint16_t* nzlist = reinterpret_cast<uint16_t*>(smem + lay.nzlist);
  uint16_t* evb = reinterpret_cast<uint16_t*>(smem + lay.evb);
  uint2* rrec = reinterpret_cast<uint2*>(smem + lay.rrec);
  uint8_t* rsrc = smem + lay.rsrc;
  uint4* win = reinterpret_cast<uint4*>(smem + lay.win);          // [bin][kWin][2]
  uint32_t* idata = reinterpret_cast<uint32_t*>(smem + lay.idata);  // a round's items: bin | count << 13
  uint4* edata = reinterpret_cast<uint4*>(smem + lay.edata);        // [lane][kEdgePre] first one-way edges of the lane's core
  uint32_t* tkl = reinterpret_cast<uint32_t*>(smem + lay.tk);       // an irregular grow's first cores: IDs taken,
  uint2* tdl = reinterpret_cast<uint2*>(smem + lay.td);             // cores still to expand
  uint32_t* cbits = reinterpret_cast<uint32_t*>(smem + lay.cbits);  // bit (core ID): the core is unassigned
  uint16_t* req = reinterpret_cast<uint16_t*>(smem + lay.req);      // ring of bins whose window wants entries (hdr[11] head, hdr[12] tail)

  const size_t fo = static_cast<size_t>(frame) * C;
  const uint4* lists = sb.lists + 2 * fo;
  const uint32_t* corepairs = sb.corepairs + fo;
  uint32_t* cgrow = sb.cgrow + fo;
  uint4* cand_grow = sb.cand_grow + static_cast<size_t>(frame) * sb.pcap;
  uint32_t* biglist = sb.biglist + static_cast<size_t>(frame) * sb.pcap;

  // ---- state: histogram, list bounds, every core alive, non-empty bins (wave 0; the helper waits at the barrier below)
  if (wid == 0) {
  hdr[lane] = 0;
  {
    const uint32_t* bo = sb.bin_off + static_cast<size_t>(frame) * (nb + 1);
    const uint32_t* hg = sb.histg + static_cast<size_t>(frame) * nb;
    for (int b = lane; b < nb; b += 64) {
      const uint32_t s0 = bo[b], s1 = bo[b + 1];
      const int hv = static_cast<int>(hg[b]);
      hist[b] = hv;
      lo[b] = static_cast<int>(s0);
      const int n0 = static_cast<int>(s1 - s0) < kWin ? static_cast<int>(s1 - s0) : kWin;
      hi[b] = static_cast<int>(s0) + n0;  // every window starts filled (the loads of all bins are in flight together)
#pragma unroll
      for (int k = 0; k < kWin; ++k)
        if (k < n0) {
          const int sl = (b * kWin + ((static_cast<int>(s0) + k) & (kWin - 1))) * 2;
          win[sl] = lists[2 * (static_cast<int>(s0) + k)];
          win[sl + 1] = lists[2 * (static_cast<int>(s0) + k) + 1];
        }
      cursor[b] = static_cast<int>(s1);
      bclaim[b] = 0xFFFFFFFFu;
    }
    const int npair_total = static_cast<int>(sb.cursor[kCtr * frame]);
    for (int w = lane; w < (npair_total + 31) / 32 + 1; w += 64) cbits[w] = 0xFFFFFFFFu;
  }
  wave_sync();
  {
    int nz = 0;
    for (int base = 0; base < nb; base += 64) {
      const int b = base + lane;
      const int v = b < nb ? hist[b] : 0;
      const unsigned long long nzm = __ballot(v > 0);
      if (v > 0) nzlist[nz + __popcll(nzm & lt_mask)] = static_cast<uint16_t>(b);
      nz += static_cast<int>(__popcll(nzm));
    }
    if (lane == 0) hdr[5] = nz;
  }
  wave_sync();
  }
  __syncthreads();
  // ---- wave 1, the windows' helper.  Every bin's list has a 4-entry window in LDS; wave 0 reads its seeds there and never
  // touches the lists in global memory (a cold entry is ~3 000 cycles away).  When it leaves a window with at most one
  // unread entry it posts the bin in a ring; the helper takes the posted bins, one per lane, loads the next entries into the
  // window's free slots and then moves the window's end.  Single writer each: wave 0 writes lo[] (entries in front of it are
  // read or dead) and the ring's head, the helper writes hi[], the slots behind it and the ring's tail; LDS executes a wave's
  // operations in order, so whoever sees a new end sees the entries.  (A bin posted twice is loaded twice with the same
  // values.)
  if (wid == 1) {
    volatile int* vlo = lo;
    volatile int* vhi = hi;
    volatile int* done = hdr + 10;
    volatile int* vhead = hdr + 11;
    volatile int* vtail = hdr + 12;
    int tail = 0;
    for (int it = 0; it < (1 << 28) && !*done; ++it) {  // (bounded: a helper that stops helping cannot hang the launch)
      int n = *vhead - tail;
      if (n <= 0) {
        __builtin_amdgcn_s_sleep(1);
        continue;
      }
      if (n > 64) n = 64;
      const unsigned long long hb0 = DBG ? static_cast<unsigned long long>(__builtin_amdgcn_s_memtime()) : 0ull;
      if (lane < n) {
        const int b = static_cast<volatile uint16_t*>(req)[(tail + lane) & (kReqRing - 1)];
        const int l = vlo[b], h = vhi[b], e = cursor[b];
        const int room = kWin - (h - l);
        const int m = e - h < room ? e - h : room;
        if (m > 0) {
          uint4 a[2 * kWin];
#pragma unroll
          for (int k = 0; k < kWin; ++k)
            if (k < m) {
              a[2 * k] = lists[2 * (h + k)];
              a[2 * k + 1] = lists[2 * (h + k) + 1];
            }
#pragma unroll
          for (int k = 0; k < kWin; ++k)
            if (k < m) {
              const int sl = (b * kWin + ((h + k) & (kWin - 1))) * 2;
              win[sl] = a[2 * k];
              win[sl + 1] = a[2 * k + 1];
            }
          asm volatile("" ::: "memory");
          vhi[b] = h + m;
        }
      }
      tail += n;
      asm volatile("" ::: "memory");
      if (lane == 0) *vtail = tail;
      if (DBG && lane == 0) {
        hdr[13] += 1;
        hdr[14] += static_cast<int>(static_cast<unsigned long long>(__builtin_amdgcn_s_memtime()) - hb0);
        hdr[15] += n;
      }
    }
    return;
  }
  int rq_head = 0;
  // posts the bins of the lanes that ask (the helper's ring never holds more than kReqRing)
  auto post = [&](bool want, int b) {
    const unsigned long long m = __ballot(want);
    if (!m) return;
    const int cnt = __popcll(m);
    for (int g = 0; g < (1 << 20) && rq_head + cnt - *static_cast<volatile int*>(hdr + 12) > kReqRing; ++g) __builtin_amdgcn_s_sleep(1);
    if (want) req[(rq_head + __popcll(m & ((1ull << lane) - 1ull))) & (kReqRing - 1)] = static_cast<uint16_t>(b);
    rq_head += cnt;
    asm volatile("" ::: "memory");
    __builtin_amdgcn_wave_barrier();
    if (lane == 0) *static_cast<volatile int*>(hdr + 11) = rq_head;
  };
  const int planar_total = static_cast<int>(sb.cursor[kCtr * frame + 2]);
  auto alive = [&](uint32_t id) { return ((cbits[id >> 5] >> (id & 31u)) & 1u) != 0u; };
  if (DBG) t_setup = static_cast<unsigned long long>(__builtin_amdgcn_s_memtime()) - t_start;

  int remaining = planar_total;
  int tail = 0, ngrow = 0, ncand = 0, nbig = 0, nhuge = 0;
  const int nnz = hdr[5];
  int mybin[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) mybin[q] = (lane + 64 * q < nnz) ? static_cast<int>(nzlist[lane + 64 * q]) : -1;
  const int nsets = (nnz + 63) >> 6;
  int cont_b = -1, cont_level = 0;
  for (int guard = 0; remaining > 0 && guard <= C; ++guard) {
    if (DBG) { ++n_rounds; ts0 = static_cast<unsigned long long>(__builtin_amdgcn_s_memtime()); }
    // ---- the events of this round: the bins at the histogram's maximum, ascending = lane order (dpx_regions2.h, "sweep
    // rounds": a bin whose turn comes with its count still at the level has lost no cell since the sweep began)
    int b = -1, bv = 0, nev = 0;
    bool part = false;
    unsigned long long partmask = 0ull;
    if (cont_level > 0) {  // an unfinished sweep goes on without an arg-max: the listed bins still at the level, in lane order
      part = cont_b >= 0 && hist[cont_b >= 0 ? cont_b : 0] == cont_level;
      partmask = __ballot(part);
      if (partmask) {
        b = part ? cont_b : -1;
        bv = cont_level;
        nev = static_cast<int>(__popcll(partmask));
      }
    }
    if (!partmask) {
      uint32_t key = 0;
      uint32_t cntq[4] = {0u, 0u, 0u, 0u};
#pragma unroll
      for (int q = 0; q < 4; ++q)
        if (q < nsets) {
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
      // seed_candidates.size() < min_region_growing_candidate_size (size_t comparison, plane_extractor.cpp:305)
      if (static_cast<unsigned long long>(bv) < static_cast<unsigned long long>(static_cast<long long>(prm.min_cand))) break;
      if (qc >= 255)
        for (qc = 255; qc < nsets; ++qc) {
          const int t = lane + 64 * qc;
          if (__ballot(t < nnz && hist[nzlist[t]] == bv)) break;
        }
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
      if (qc + 1 < nsets && nev < 64) {  // the bins at the level of the later chunks fire right after these
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
          if (q > qc && q < nsets && total < 64) append(mybin[q], cntq[q]);
        for (int q = qc + 1 > 4 ? qc + 1 : 4; q < nsets && total < 64; ++q) {
          const int t = lane + 64 * q;
          const int bq = t < nnz ? static_cast<int>(nzlist[t]) : -1;
          append(bq, bq >= 0 ? static_cast<uint32_t>(hist[bq]) : 0u);
        }
        if (staged) {
          asm volatile("" ::: "memory");
          __builtin_amdgcn_wave_barrier();
          nev = total < 64 ? total : 64;
          part = lane < nev;
          b = part ? static_cast<int>(evb[lane]) : -1;
          partmask = nev == 64 ? ~0ull : ((1ull << nev) - 1ull);
        }
      }
      cont_b = part ? b : -1;
      cont_level = bv;
    }
    const int bs = b >= 0 ? b : 0;
    S3_STAMP(a_ev);
    // ---- seeds: the first live entry of each bin's list (one entry per (core, bin) pair, sorted by the leader's mse, then
    // cell id: plane_extractor.cpp:309-316).  The next few entries of every list sit in an LDS window; a list is read from
    // global memory only when its window runs empty.
    int pos = lo[bs];
    int whi = static_cast<volatile int*>(hi)[bs];
    const int lend = cursor[bs];
    uint4 e = make_uint4(0u, 0u, 0u, 0u);
    bool need = part, broken = false;
    for (int spins = 0;; ++spins) {
      bool ask = false;  // the lane has just read its window to the end
      if (need && (spins == 0 || whi > pos)) {
        // the window's (up to) four entries and their cores' bits in two LDS round trips; the first live one is the seed
        uint4 x[kScan];
        bool ok[kScan];
#pragma unroll
        for (int k = 0; k < kScan; ++k) x[k] = win[(bs * kWin + ((pos + k) & (kWin - 1))) * 2];
#pragma unroll
        for (int k = 0; k < kScan; ++k) ok[k] = pos + k < whi && alive(x[k].y);
        int adv = whi - pos < kScan ? whi - pos : kScan;  // (no live entry: these are used up)
#pragma unroll
        for (int k = kScan - 1; k >= 0; --k)
          if (ok[k]) {
            e = x[k];
            adv = k;
          }
        bool any_ok = false;
#pragma unroll
        for (int k = 0; k < kScan; ++k) any_ok = any_ok || ok[k];
        need = !any_ok;
        pos += adv;
        ask = need && pos >= whi;
      }
      if (!__ballot(need)) break;
      // (pos == whi: every entry the window held is read or dead) — say so, and wait for the helper to move the window's end
      if (need && (pos >= lend || spins > (1 << 22))) {  // cannot happen (a bin's count is the number of its live cells, every one behind a live leader)
        broken = true;
        need = ask = false;
      }
      if (ask) static_cast<volatile int*>(lo)[bs] = pos;
      if (DBG && __ballot(ask)) ++n_refill;
      asm volatile("" ::: "memory");
      __builtin_amdgcn_wave_barrier();
      post(ask, bs);
      if (need) whi = static_cast<volatile int*>(hi)[bs];
      if (!__ballot(need && whi > pos)) __builtin_amdgcn_s_sleep(1);
    }
    if (__ballot(broken)) break;
    // what the scan stepped over is dead: the helper may already put the next entries there (the seed's own slot stays)
    if (part) static_cast<volatile int*>(lo)[bs] = pos;
    asm volatile("" ::: "memory");
    __builtin_amdgcn_wave_barrier();
    post(part && whi < lend && whi - pos < kWin, bs);
    S3_STAMP(a_seed);
    const uint32_t cell = e.x & kCellMask;
    const bool cout_ = (e.x & kCoreOut) != 0u;
    // An event whose core has one-way edges out fires alone, later in this round; its first edges are fetched now, for all
    // such lanes at once, and parked in LDS before the walk (the load's latency hides behind the claims)
    uint4 epre[kEdgePre];
#pragma unroll
    for (int k = 0; k < kEdgePre; ++k) epre[k] = make_uint4(0u, 0u, 0u, 0u);
    const unsigned long long outmask = __ballot(part && cout_);
    if (part && cout_) {
      const uint4 eb = win[(bs * kWin + (pos & (kWin - 1))) * 2 + 1];
#pragma unroll
      for (int k = 0; k < kEdgePre; ++k)
        if (static_cast<uint32_t>(k) < eb.w) epre[k] = sb.oedges[4 * fo + eb.z + k];
    }
    const uint32_t cid = e.y;
    const int np = part ? static_cast<int>(e.z) : 0;
    const int csz = part ? static_cast<int>(e.w) : 0;
    // ---- which events fire.  A grow from a core without a one-way edge out takes exactly that core, so events interact
    // only through the histogram: a bin that loses a cell to a lower lane THAT FIRES is no longer at the level when its
    // turn comes and is skipped (that covers a lower lane taking this lane's seed too — the seed's bin is this lane's).
    // Every pair of a multi-bin core is an ITEM, one wave lane each: it claims its bin with min(source lane) and on commit
    // takes its cells out of the histogram.  Cores of more than kItemMax bins and cores with a one-way edge out stop the
    // round and fire alone ("wide").
    bool complete = part && np <= kItemMax && !cout_;
    int cnt = (complete && np > 1) ? np : 0;
    const int incl_items = wave_incl_scan_i32(cnt);
    if (cnt > 0 && incl_items > 64) {  // (any give-up is exact: the event waits for a later round or fires alone)
      complete = false;
      cnt = 0;
    }
    const unsigned long long multi = __ballot(cnt > 0);
    uint32_t lowb = 0xFFFFFFFFu;
    bool it_ok = false;
    int it_src = 0, it_bin = -1, it_cnt = 0;
    if (multi) {
      if (DBG) ++n_multi;
      const int off = incl_items - cnt;
      const int nitems = static_cast<int>(wave_max_u32(cnt > 0 ? static_cast<uint32_t>(incl_items) : 0u));
      rrec[lane] = make_uint2(cid, static_cast<uint32_t>(off));
      if (cnt > 0) {  // the core's first four pairs travel with its list entry
        const uint4 pr = win[(bs * kWin + (pos & (kWin - 1))) * 2 + 1];
        idata[off] = pr.x;
        idata[off + 1] = pr.y;
        if (cnt > 2) idata[off + 2] = pr.z;
        if (cnt > 3) idata[off + 3] = pr.w;
      }
#pragma unroll
      for (int i = 0; i < kItemMax; ++i)
        if (i < cnt) rsrc[off + i] = static_cast<uint8_t>(lane);
      asm volatile("" ::: "memory");  // (LDS executes a wave's operations in order; only the compiler must keep it)
      __builtin_amdgcn_wave_barrier();
      it_ok = lane < nitems;
      if (it_ok) {
        it_src = static_cast<int>(rsrc[lane]);
        const uint2 r = rrec[it_src];
        const int j = lane - static_cast<int>(r.y);
        const uint32_t pr = j < 4 ? idata[lane] : corepairs[r.x + static_cast<uint32_t>(j)];
        it_bin = static_cast<int>(pr & kPairBinMask);
        it_cnt = static_cast<int>(pr >> kPairBinBits);
        atomicMin(&bclaim[it_bin], static_cast<uint32_t>(it_src));
      }
      asm volatile("" ::: "memory");
      __builtin_amdgcn_wave_barrier();
      if (part) lowb = bclaim[bs];
      asm volatile("" ::: "memory");
      __builtin_amdgcn_wave_barrier();
      if (it_ok) bclaim[it_bin] = 0xFFFFFFFFu;  // clean for the next round
    }
    S3_STAMP(a_items);
    if (outmask) {
      if (part && cout_) {
#pragma unroll
        for (int k = 0; k < kEdgePre; ++k) edata[lane * kEdgePre + k] = epre[k];
      }
      asm volatile("" ::: "memory");
      __builtin_amdgcn_wave_barrier();
    }
    // A round is walked in SEGMENTS: the lanes up to the first event that must fire alone are committed together, that event
    // fires, the lanes behind it whose bin is still at the level go on (the others have lost a cell: skipped), and so on.
    unsigned long long pending = partmask, skipped = 0ull, usedmask = 0ull;
    for (;;) {
    const bool flag = part && ((pending >> lane) & 1ull) != 0ull && (!complete || lowb < static_cast<uint32_t>(lane));
    unsigned long long fl = __ballot(flag);
    int stop = 64;
    bool stop_wide = false;
    while (fl) {  // lanes in order; only the flagged ones need a look
      const int t = static_cast<int>(__builtin_ctzll(fl));
      fl &= fl - 1ull;
      const uint32_t mb = static_cast<uint32_t>(__builtin_amdgcn_readlane(static_cast<int>(lowb), t));
      if (mb < static_cast<uint32_t>(t)) {
        if (((skipped >> mb) & 1ull) == 0ull) {  // its bin lost a cell to a lane that fires: not at the level any more
          skipped |= 1ull << t;
          if (DBG) ++n_skip;
          continue;
        }
        stop = t;  // the lowest claimant was itself skipped — what the other claimants did is unknown: next round
        break;
      }
      stop = t;  // a grow that is not handled in a round and whose bin nobody touched: it fires now, alone
      stop_wide = true;
      break;
    }
    S3_STAMP(a_walk);
    // ---- commit the lanes that fire before the stop
    const unsigned long long cm = (stop >= 64 ? pending : (pending & ((1ull << stop) - 1ull))) & ~skipped;
    const bool cmt = ((cm >> lane) & 1ull) != 0ull;
    usedmask |= cm;
    if (cm) {
      const int grank = static_cast<int>(__popcll(cm & lt_mask));
      const int ncm = static_cast<int>(__popcll(cm));
      const int myn = cmt ? csz : 0;
      const int incl = wave_incl_scan_i32(myn);
      const int total = __builtin_amdgcn_readlane(incl, 63);
      const int base = tail + incl - myn;
      if (cmt) {
        atomicAnd(&cbits[cid >> 5], ~(1u << (cid & 31u)));  // the core is assigned
        if (np == 1) atomicSub(&hist[bs], csz);              // hist.removePoint for each of its cells
      }
      if ((multi & cm) && it_ok && ((cm >> it_src) & 1ull) != 0ull) atomicSub(&hist[it_bin], it_cnt);
      // cells_to_merge.size() < min_region_growing_cells_activated (size_t comparison, plane_extractor.cpp:329)
      const bool want_cand = cmt && !(static_cast<unsigned long long>(csz) < static_cast<unsigned long long>(static_cast<long long>(prm.min_cells)));
      const unsigned long long wantmask = __ballot(want_cand);
      if (wantmask) {
        const int crank = static_cast<int>(__popcll(wantmask & lt_mask));
        const bool have = want_cand && ncand + crank < sb.pcap;
        const bool bigc = have && csz > kItemMax;  // more cells than a lane orders on its own: a wave's work in k_order,
        const bool hugec = bigc && csz >= kHugeMin;  // or a workgroup's
        const unsigned long long bigmask = __ballot(bigc && !hugec), hugemask = __ballot(hugec);
        if (have) {
          cand_grow[ncand + crank] = make_uint4(static_cast<uint32_t>(ngrow + grank), static_cast<uint32_t>(base), static_cast<uint32_t>(base + csz), cell | (bigc ? kCandWide : 0u));
          cgrow[cid] = static_cast<uint32_t>(ngrow + grank + 1);
          if (bigc && !hugec) biglist[nbig + static_cast<int>(__popcll(bigmask & lt_mask))] = static_cast<uint32_t>(ncand + crank);
          if (hugec) biglist[sb.pcap - 1 - nhuge - static_cast<int>(__popcll(hugemask & lt_mask))] = static_cast<uint32_t>(ncand + crank);
        }
        nbig += static_cast<int>(__popcll(bigmask));
        nhuge += static_cast<int>(__popcll(hugemask));
        const int nwant = static_cast<int>(__popcll(wantmask));
        ncand += nwant < sb.pcap - ncand ? nwant : (sb.pcap - ncand > 0 ? sb.pcap - ncand : 0);
      }
      tail = __builtin_amdgcn_readfirstlane(tail + total);
      ngrow += ncm;
      remaining -= total;
    }
    asm volatile("" ::: "memory");
    __builtin_amdgcn_wave_barrier();
    S3_STAMP(a_commit);
    if (!stop_wide) break;
    // ---- the stopping event fires alone: the pairs of its core(s) leave the histogram
    if (DBG) ++n_wide;
    usedmask |= 1ull << stop;
    const uint32_t wcell = static_cast<uint32_t>(__builtin_amdgcn_readlane(static_cast<int>(cell), stop));
    const uint32_t wcid = static_cast<uint32_t>(__builtin_amdgcn_readlane(static_cast<int>(cid), stop));
    const int wnp = __builtin_amdgcn_readlane(np, stop);
    const int wsz = __builtin_amdgcn_readlane(csz, stop);
    const bool wout = ((__ballot(cout_) >> stop) & 1ull) != 0ull;
    uint4 wB = make_uint4(0u, 0u, 0u, 0u);  // the second half of its list entry
    if (lane == stop) wB = win[(bs * kWin + (pos & (kWin - 1))) * 2 + 1];
    wB.x = static_cast<uint32_t>(__builtin_amdgcn_readlane(static_cast<int>(wB.x), stop));
    wB.y = static_cast<uint32_t>(__builtin_amdgcn_readlane(static_cast<int>(wB.y), stop));
    wB.z = static_cast<uint32_t>(__builtin_amdgcn_readlane(static_cast<int>(wB.z), stop));
    wB.w = static_cast<uint32_t>(__builtin_amdgcn_readlane(static_cast<int>(wB.w), stop));
    // pairs [from, n) of a core out of corepairs, 64 per step
    auto take_pairs = [&](uint32_t pst, int from, int n) {
      for (int i = from + lane; i < n; i += 64) {
        const uint32_t pr = corepairs[pst + static_cast<uint32_t>(i)];
        atomicSub(&hist[pr & kPairBinMask], static_cast<int>(pr >> kPairBinBits));
      }
    };
    if (lane == 0) atomicAnd(&cbits[wcid >> 5], ~(1u << (wcid & 31u)));
    {
      const int ninl = wout ? (wnp < 2 ? wnp : 2) : (wnp < 4 ? wnp : 4);  // pairs that came with the list entry
      const uint32_t pr = lane == 0 ? wB.x : (lane == 1 ? wB.y : (lane == 2 ? wB.z : wB.w));
      if (lane < ninl) atomicSub(&hist[pr & kPairBinMask], static_cast<int>(pr >> kPairBinBits));
      take_pairs(wcid, ninl, wnp);
    }
    int gsize = wsz;
    int ntaken = 1;
    if (wout) {
      if (DBG) ++n_irreg;
      // The cores reachable from the seed's core over one-way edges among the cores still unassigned, in any order (the SET
      // is what the sequence needs; k_order finds the activation order).  An edge carries its target core's ID, pair count,
      // size and first pair, so a target of one bin without edges of its own is done on the spot; the others wait in `todo`.
      uint32_t* taken = sb.worklist + 3 * fo;   // IDs of the grow's cores (for the marks k_cellgrow reads)
      uint32_t* todo = taken + C;               // {ID | one-way-out << 31, pairs}
      if (lane == 0) {
        hdr[8] = 0;   // todo entries
        hdr[9] = 1;   // cores taken
        tkl[0] = wcid;
      }
      asm volatile("" ::: "memory");
      __builtin_amdgcn_wave_barrier();
      int lsize = 0;
      auto visit = [&](const uint4 ed) {
        {
          const uint32_t id = ed.x & ~kCoreOut;
          // a core is unassigned as long as its bit is set; whoever clears it has found the core
          const uint32_t bit = 1u << (id & 31u);
          const uint32_t old = atomicAnd(&cbits[id >> 5], ~bit);
          if (old & bit) {
            lsize += static_cast<int>(ed.z);
            const int tk = atomicAdd(&hdr[9], 1);
            if (tk < kWideLds) tkl[tk] = id;
            else taken[tk] = id;
            if (ed.y == 1u) atomicSub(&hist[ed.w & kPairBinMask], static_cast<int>(ed.w >> kPairBinBits));
            if (ed.y != 1u || (ed.x & kCoreOut)) {
              const int at = atomicAdd(&hdr[8], 1);
              if (at < kWideLds) {
                tdl[at] = make_uint2(ed.x, ed.y);
              } else {
                todo[2 * at] = ed.x;
                todo[2 * at + 1] = ed.y;
              }
            }
          }
        }
      };
      auto expand = [&](uint32_t es, uint32_t from, uint32_t ne) {
        for (uint32_t i = from + lane; i < ne; i += 64) visit(sb.oedges[4 * fo + es + i]);
      };
      if (static_cast<uint32_t>(lane) < wB.w && lane < kEdgePre) visit(edata[stop * kEdgePre + lane]);  // (fetched ahead)
      expand(wB.z, kEdgePre, wB.w);
      // (lists beyond kWideLds entries are in global memory: their stores must be visible before they are read back)
      auto list_sync = [&]() {
        asm volatile("" ::: "memory");
        __builtin_amdgcn_wave_barrier();
        if (*static_cast<volatile int*>(&hdr[8]) > kWideLds || *static_cast<volatile int*>(&hdr[9]) > kWideLds) wave_sync();
      };
      list_sync();
      for (int ti = 0;; ++ti) {
        const int ntodo = *static_cast<volatile int*>(&hdr[8]);
        if (ti >= ntodo) break;
        uint32_t tx;
        int tnp;
        if (ti < kWideLds) {
          const uint2 t2 = tdl[ti];
          tx = t2.x;
          tnp = static_cast<int>(t2.y);
        } else {
          tx = __hip_atomic_load(todo + 2 * ti, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          tnp = static_cast<int>(__hip_atomic_load(todo + 2 * ti + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
        }
        const uint32_t tid_ = tx & ~kCoreOut;
        if (tnp != 1) take_pairs(tid_, 0, tnp);
        if (tx & kCoreOut) {
          const uint2 rec = sb.crec[fo + tid_];
          expand(rec.x, 0u, rec.y);
        }
        list_sync();
      }
      gsize += __builtin_amdgcn_readlane(wave_incl_scan_i32(lsize), 63);  // (DPP: the ds_bpermute butterfly of wave_sum costs ~800 cycles here)
      ntaken = hdr[9];
    }
    const bool is_cand = !(static_cast<unsigned long long>(gsize) < static_cast<unsigned long long>(static_cast<long long>(prm.min_cells))) && ncand < sb.pcap;
    if (is_cand) {
      if (!wout) {
        if (lane == 0) cgrow[wcid] = static_cast<uint32_t>(ngrow + 1);
      } else {
        const uint32_t* taken = sb.worklist + 3 * fo;
        for (int i = lane; i < ntaken; i += 64)
          cgrow[i < kWideLds ? tkl[i] : __hip_atomic_load(taken + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)] = static_cast<uint32_t>(ngrow + 1);
      }
      if (lane == 0) {
        cand_grow[ncand] = make_uint4(static_cast<uint32_t>(ngrow), static_cast<uint32_t>(tail), static_cast<uint32_t>(tail + gsize), wcell | kCandWide);
        if (gsize >= kHugeMin) biglist[sb.pcap - 1 - nhuge] = static_cast<uint32_t>(ncand);
        else biglist[nbig] = static_cast<uint32_t>(ncand);
      }
      ++ncand;
      if (gsize >= kHugeMin) ++nhuge;
      else ++nbig;
    }
    tail = __builtin_amdgcn_readfirstlane(tail + gsize);
    ++ngrow;
    remaining -= gsize;
    asm volatile("" ::: "memory");
    __builtin_amdgcn_wave_barrier();
    // the lanes behind it: a bin still at the level has lost no cell — its event stands as computed; the others are out
    {
      const bool still = part && lane > stop && ((pending >> lane) & 1ull) != 0ull && ((skipped >> lane) & 1ull) == 0ull && hist[bs] == bv;
      const unsigned long long stillmask = __ballot(still);
      const unsigned long long above = stop >= 63 ? 0ull : (~0ull << (stop + 1));
      skipped |= partmask & above & ~stillmask;
      pending = stillmask;
    }
    S3_STAMP(a_wide);
    if (!pending) break;
    }  // (segments)
    if (part) {  // list heads of the round's bins (the helper sees a window running low and refills it)
      const bool used = ((usedmask >> lane) & 1ull) != 0ull;
      static_cast<volatile int*>(lo)[bs] = used ? pos + 1 : pos;
    }
    asm volatile("" ::: "memory");
    __builtin_amdgcn_wave_barrier();
    post(part && whi < lend && ((usedmask >> lane) & 1ull) != 0ull, bs);
    asm volatile("" ::: "memory");
    __builtin_amdgcn_wave_barrier();
  }
  if (lane == 0) *static_cast<volatile int*>(hdr + 10) = 1;  // the helper leaves its loop
  // the header the tail launch starts from (k_regions2 PHASE 3 reads hdr[0..3], its segment counter hdr[40] must be 0)
  int* ho = sb.hdr_out + static_cast<size_t>(frame) * 48;
  if (lane < 48) {
    int v = 0;
    v = lane == 0 ? planar_total : v;
    v = lane == 1 ? tail : v;
    v = lane == 2 ? ngrow : v;
    v = lane == 3 ? ncand : v;
    v = lane == kHdrBig ? nbig : v;
    v = lane == kHdrHuge ? nhuge : v;
    ho[lane] = v;
  }
  if (DBG && sb.dbg && lane == 0) {
    unsigned long long* d = sb.dbg + static_cast<size_t>(frame) * kDbgSlots;
    d[0] = t_setup;
    d[1] = static_cast<unsigned long long>(__builtin_amdgcn_s_memtime()) - t_start;
    d[11] = n_rounds; d[13] = static_cast<unsigned long long>(ngrow); d[14] = static_cast<unsigned long long>(ncand);
    d[15] = static_cast<unsigned long long>(tail); d[19] = n_refill; d[24] = static_cast<unsigned long long>(hdr[13]); d[25] = static_cast<unsigned long long>(static_cast<uint32_t>(hdr[14])); d[26] = static_cast<unsigned long long>(hdr[15]); d[21] = n_wide; d[22] = n_skip; d[23] = n_irreg; d[28] = n_multi;
    d[3] = a_ev; d[4] = a_seed; d[5] = a_items; d[6] = a_walk; d[7] = a_commit; d[8] = a_wide;
    d[29] = static_cast<unsigned long long>(sb.cursor[kCtr * frame]); d[30] = static_cast<unsigned long long>(nbig) | (static_cast<unsigned long long>(nhuge) << 32);
  }
}

#undef S3_STAMP
// ---- per cell: the grow that took it (candidates only: other grows leave no trace) and its in-edge bits ------------------
__global__ __launch_bounds__(256) void k_cellgrow(Params prm, CoreBufs cb) {
  const int frame = blockIdx.y;
  const int c = static_cast<int>(blockIdx.x) * 256 + threadIdx.x;
  if (c >= prm.C) return;
  const size_t fo = static_cast<size_t>(frame) * prm.C;
  const uint32_t m = cb.meta[fo + c];
  uint32_t q = 0u;
  if (m & F_UNASSIGNED) {
    const uint32_t g1 = cb.cgrow[fo + cb.pstart[fo + cb.parent[fo + c]]];
    q = (g1 << 4) | ((m >> 8) & 0xFu);
  }
  cb.cellq[fo + c] = q;
}

// ---- FIFO activation order of the candidate grows (plane_extractor.cpp:349-392 restricted to the grow's own cells) -------
// A grow's cells are known (cellq: grow + 1 in the high bits); what is computed here is the order in which growSeed
// activates them: queue order, then U, D, L, R, a cell going to the first (slot, direction) that reaches it.  Grows are
// disjoint, so all of them run at once: a task = one "wide" candidate (a whole wave: 16 queue slots x 4 directions per step,
// duplicates resolved through a small per-wave claim table) or 64 consecutive candidates of up to kItemMax cells (one LANE
// each: the grow is a list of relative cell codes in a 64-bit register, as in the sweep rounds of dpx_regions2.h).
struct OrderBufs {
  uint32_t* cellq;          // [B][C]
  const uint4* cand_grow;   // [B][pcap]
  const uint32_t* biglist;  // [B][pcap]
  const int* hdr;           // [B][48]
  uint32_t* queue;          // [B][C]
  int* task;                // [B][kCtr] task counters: waves' tasks, workgroups' tasks (+ k_seg_accum's two) (zeroed by k_core_tiles)
  unsigned long long* dbg;
  int pcap;
};
constexpr int kOrderWaves = 8;
constexpr int kOrderClaim = 256;  // claim-table slots per wave
// A HUGE grow (>= kHugeMin cells) takes the whole workgroup: 16 queue slots per wave and step, the grow's cells as one byte
// each in LDS (bit 0: belongs to the grow and is not activated yet, bits 2..5: in-edges U, D, L, R), the last kHugeRing
// queue entries in an LDS ring, and duplicates resolved by PULL as in dpx_regions2.h — a lane loses to any in-neighbour of
// its target that sits earlier in the batch — with the batch's (cell -> slot) map in a small LDS hash table instead of a
// queue-position array: no global memory inside a step.
constexpr int kHugeRingBits = 12, kHugeRing = 1 << kHugeRingBits;
constexpr int kHugeHash = 2048;  // slots of the batch map (a batch holds up to 16 * kHugeWaves = 256 cells)
__host__ __device__ inline size_t order_lds_bytes(int C) { return (static_cast<size_t>(C) + 15u) / 16u * 16u + 4u * kHugeRing + 4u * kHugeHash + 128u; }  // (+ control words: task, per-wave counts)

// HUGE = the launch that orders the huge grows (dynamic LDS: order_lds_bytes); the other launch takes the waves' tasks
constexpr int kHugeWaves = 16;  // waves of the launch that orders huge grows: 256 queue slots per step
template <bool DBG, bool HUGE>
__global__ __launch_bounds__((HUGE ? kHugeWaves : kOrderWaves) * 64) void k_order(Params prm, OrderBufs ob) {
  constexpr int kOrderWaves = HUGE ? kHugeWaves : dpx::kOrderWaves;  // (this launch's)
  const unsigned long long t_start = DBG ? static_cast<unsigned long long>(__builtin_amdgcn_s_memtime()) : 0ull;
  const int frame = blockIdx.y;
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int C = prm.C, nh = prm.nh;
  const unsigned long long lt_mask = (1ull << lane) - 1ull;
  __shared__ uint32_t s_claim[HUGE ? 1 : kOrderWaves][kOrderClaim];
  uint32_t* ctab = s_claim[HUGE ? 0 : wid];
  if (!HUGE) {
    for (int i = lane; i < kOrderClaim; i += 64) ctab[i] = 0xFFFFFFFFu;
    wave_sync();
  }
  const size_t fo = static_cast<size_t>(frame) * C;
  uint32_t* cellq = ob.cellq + fo;
  uint32_t* queue = ob.queue + fo;
  const uint4* cand_grow = ob.cand_grow + static_cast<size_t>(frame) * ob.pcap;
  const uint32_t* biglist = ob.biglist + static_cast<size_t>(frame) * ob.pcap;
  const int* hdr = ob.hdr + static_cast<size_t>(frame) * 48;
  const int ncand = hdr[3], nbig = hdr[kHdrBig], nhuge = hdr[kHdrHuge];
  const int ntasks = nbig + ((ncand + 63) >> 6);
  const int slot = lane >> 2, dir = lane & 3;
  const int delta = dir == 0 ? -nh : (dir == 1 ? nh : (dir == 2 ? -1 : 1));
  const uint32_t opp = static_cast<uint32_t>(dir ^ 1);
  // ---- huge grows, the whole workgroup on one at a time
  if constexpr (HUGE) {
    if (nhuge == 0) return;
    extern __shared__ __align__(16) unsigned char dsm[];
    uint8_t* st = dsm;
    uint32_t* ring = reinterpret_cast<uint32_t*>(dsm + (static_cast<size_t>(C) + 15u) / 16u * 16u);
    uint32_t* bmap = ring + kHugeRing;
    int* ctl = reinterpret_cast<int*>(bmap + kHugeHash);  // 0 task, 8.. per-wave winner counts
    constexpr uint32_t kRingMask = kHugeRing - 1u;
    const uint32_t notme = ~(1u << opp) & 0xFu;
    for (int guard = 0; guard <= nhuge; ++guard) {
      if (tid == 0) ctl[0] = atomicAdd(ob.task + kCtr * frame + 1, 1);
      for (int i = tid; i < kHugeHash; i += kOrderWaves * 64) bmap[i] = 0xFFFFFFFFu;
      __syncthreads();
      const int t = ctl[0];
      if (t >= nhuge) break;
      const uint4 cg = cand_grow[biglist[ob.pcap - 1 - t]];
      const uint32_t g1 = cg.x + 1u;
      const int gs = static_cast<int>(cg.y);
      const uint32_t seed = cg.w & kCellMask;
      for (int c = tid; c < C; c += kOrderWaves * 64) {
        const uint32_t q = cellq[c];
        st[c] = static_cast<uint8_t>((((q >> 4) == g1 && static_cast<uint32_t>(c) != seed) ? 1u : 0u) | ((q & 0xFu) << 2));
      }
      if (tid == 0) {
        queue[gs] = seed;
        ring[static_cast<uint32_t>(gs) & kRingMask] = seed;
      }
      __syncthreads();
      int H = gs, T = gs + 1;
      while (H < T) {
        const int ns = (T - H) < 16 * kOrderWaves ? (T - H) : 16 * kOrderWaves;
        const int gsl = wid * 16 + slot;  // slot of this lane's queue entry within the batch
        const bool active = gsl < ns;
        // (the frontier of a BFS over a grid is far narrower than the ring; should it ever not be, the batch comes from the
        // queue in global memory, written through below)
        const bool from_ring = (T - H) <= kHugeRing - 16 * kOrderWaves;
        const int qi = H + (active ? gsl : 0);
        const int u = from_ring ? static_cast<int>(ring[static_cast<uint32_t>(qi) & kRingMask])
                                : static_cast<int>(__hip_atomic_load(queue + qi, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
        uint32_t myh = 0xFFFFFFFFu;
        if (active && dir == 0) {  // the batch's cells into the map: cell << 8 | slot
          uint32_t h = (static_cast<uint32_t>(u) * 2654435761u) >> 21;
          const uint32_t key = (static_cast<uint32_t>(u) << 8) | static_cast<uint32_t>(gsl);
          while (atomicCAS(&bmap[h], 0xFFFFFFFFu, key) != 0xFFFFFFFFu) h = (h + 1u) & (kHugeHash - 1u);
          myh = h;
        }
        __syncthreads();
        const int vraw = u + delta;
        const bool inrange = active && vraw >= 0 && vraw < C;
        const int v = vraw < 0 ? 0 : (vraw >= C ? C - 1 : vraw);
        const uint32_t fv = st[v];
        const uint32_t inb = (fv >> 2) & 0xFu;  // in-edge bits of v
        bool win = inrange && (fv & 1u) != 0u && ((inb >> opp) & 1u) != 0u;
        if (win) {
          // u loses to any OTHER in-neighbour of v that sits earlier in this batch
          const uint32_t others = inb & notme;
#pragma unroll
          for (int e = 0; e < 4; ++e)
            if (win && ((others >> e) & 1u)) {
              const int w = v + (e == 0 ? -nh : (e == 1 ? nh : (e == 2 ? -1 : 1)));
              uint32_t h = (static_cast<uint32_t>(w) * 2654435761u) >> 21;
              for (;;) {
                const uint32_t k = bmap[h];
                if (k == 0xFFFFFFFFu) break;
                if ((k >> 8) == static_cast<uint32_t>(w)) {
                  if (static_cast<int>(k & 0xFFu) < gsl) win = false;
                  break;
                }
                h = (h + 1u) & (kHugeHash - 1u);
              }
            }
        }
        const unsigned long long mask = __ballot(win);
        if (lane == 0) ctl[8 + wid] = static_cast<int>(__popcll(mask));
        __syncthreads();
        int off = 0, tot = 0;
#pragma unroll
        for (int w = 0; w < kOrderWaves; ++w) {
          const int cw = ctl[8 + w];
          off += w < wid ? cw : 0;
          tot += cw;
        }
        if (win) {
          const int pos = T + off + static_cast<int>(__popcll(mask & lt_mask));
          if (from_ring) queue[pos] = static_cast<uint32_t>(v);
          else __hip_atomic_store(queue + pos, static_cast<uint32_t>(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          ring[static_cast<uint32_t>(pos) & kRingMask] = static_cast<uint32_t>(v);
          st[v] = static_cast<uint8_t>(fv & ~1u);
        }
        if (myh != 0xFFFFFFFFu) bmap[myh] = 0xFFFFFFFFu;
        T += tot;
        H += ns;
        if (!from_ring) __threadfence();
        __syncthreads();
      }
    }
    return;
  }
  for (int guard = 0; guard <= ntasks; ++guard) {  // (every task loop is bounded: a wave takes each task at most once)
    int t = 0;
    if (lane == 0) t = atomicAdd(ob.task + kCtr * frame, 1);
    t = __builtin_amdgcn_readfirstlane(t);
    if (t >= ntasks) break;
    if (t < nbig) {
      // ---- one wide grow: wave-wide FIFO BFS over its own cells
      const uint4 cg = cand_grow[biglist[t]];
      const uint32_t g1 = cg.x + 1u;
      const int gs = static_cast<int>(cg.y);
      const uint32_t seed = cg.w & kCellMask;
      if (lane == 0) {
        queue[gs] = seed;
        cellq[seed] = 0u;  // a cell leaves its grow's membership when it is activated
      }
      int head = gs, tail = gs + 1;
      wave_sync();
      while (head < tail) {
        const int nslots = (tail - head) < 16 ? (tail - head) : 16;
        const bool active = slot < nslots;
        const int u = static_cast<int>(__hip_atomic_load(queue + head + (active ? slot : 0), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
        const int vraw = u + delta;
        const bool inrange = active && vraw >= 0 && vraw < C;
        const int v = vraw < 0 ? 0 : (vraw >= C ? C - 1 : vraw);
        const uint32_t w = __hip_atomic_load(cellq + v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        // v accepts u iff v belongs to this grow, is not activated yet, and its in-edge from u's side is set (an L / R
        // step across a row end and a step outside the grid have no such edge)
        bool cnd = inrange && (w >> 4) == g1 && ((w >> opp) & 1u) != 0u;
        // the lowest lane that reaches a cell wins it: min(lane) per cell in the claim table, linear probing
        bool win = false;
        uint32_t h = (static_cast<uint32_t>(v) * 2654435761u) >> 24;  // 8 bits
        const uint32_t mykey = (static_cast<uint32_t>(v) << 6) | static_cast<uint32_t>(lane);
        const uint32_t h0 = h;
        int ntouch = 0;
        bool pending = cnd;
        while (__ballot(pending)) {
          if (pending) atomicMin(&ctab[h], mykey);
          asm volatile("" ::: "memory");
          __builtin_amdgcn_wave_barrier();
          if (pending) {
            const uint32_t k = ctab[h];
            ++ntouch;
            if ((k >> 6) == static_cast<uint32_t>(v)) {  // this slot is v's: its lowest lane wins
              win = (k & 63u) == static_cast<uint32_t>(lane);
              pending = false;
            } else {
              h = (h + 1u) & (kOrderClaim - 1u);  // another cell's: next slot
            }
          }
          asm volatile("" ::: "memory");
          __builtin_amdgcn_wave_barrier();
        }
        const unsigned long long mask = __ballot(win);
        if (win) {
          const int pos = tail + static_cast<int>(__popcll(mask & lt_mask));
          queue[pos] = static_cast<uint32_t>(v);
          cellq[v] = 0u;
        }
        // leave the table clean: every slot this lane probed (all lanes are done with the table by now)
        for (int i = 0; i < ntouch; ++i) ctab[(h0 + static_cast<uint32_t>(i)) & (kOrderClaim - 1u)] = 0xFFFFFFFFu;
        tail = __builtin_amdgcn_readfirstlane(tail + static_cast<int>(__popcll(mask)));
        head = __builtin_amdgcn_readfirstlane(head + nslots);
        wave_sync();  // queue / cellq in global memory: the next step's loads must see these stores
      }
    } else {
      // ---- 64 candidates, one lane each: grows of up to kItemMax cells as lists of relative cell codes
      const int ci = (t - nbig) * 64 + lane;
      uint4 cg = make_uint4(0u, 0u, 0u, kCandWide);
      if (ci < ncand) cg = cand_grow[ci];
      bool act = (cg.w & kCandWide) == 0u;
      const bool mine = act;
      const uint32_t g1 = cg.x + 1u;
      const int cs = static_cast<int>(cg.w & kCellMask);
      unsigned long long ll = 0x88ull;  // entry 0 = the seed, code (8, 8)
      int n = 1, hpop = 0;
      while (__ballot(act)) {
        const uint32_t code = act ? (static_cast<uint32_t>(ll >> (8 * hpop)) & 0xFFu) : 0x88u;
        const int cr = static_cast<int>(code >> 4), cc = static_cast<int>(code & 15u);
        const int u = cs + (cr - 8) * nh + (cc - 8);
        auto nbq = [&](int v) { return (act && v >= 0 && v < C) ? cellq[v] : 0u; };
        const uint32_t qU = nbq(u - nh), qD = nbq(u + nh), qL = nbq(u - 1), qR = nbq(u + 1);
        // v accepts u iff it belongs to this grow and its in-edge from u's side is set (u is v's D / U / R / L neighbour)
        const bool okU = (qU >> 4) == g1 && ((qU >> 1) & 1u) != 0u;
        const bool okD = (qD >> 4) == g1 && ((qD >> 0) & 1u) != 0u;
        const bool okL = (qL >> 4) == g1 && ((qL >> 3) & 1u) != 0u;
        const bool okR = (qR >> 4) == g1 && ((qR >> 2) & 1u) != 0u;
        auto visit = [&](bool ok, uint32_t vcode) {
          const uint32_t rep = __builtin_amdgcn_perm(0u, vcode, 0u);  // the low byte in all four bytes
          const uint32_t x0 = static_cast<uint32_t>(ll) ^ rep, x1 = static_cast<uint32_t>(ll >> 32) ^ rep;
          const bool dup = ((((x0 - 0x01010101u) & ~x0) | ((x1 - 0x01010101u) & ~x1)) & 0x80808080u) != 0u;
          if (ok && act && !dup && n < 8) {
            ll |= static_cast<unsigned long long>(vcode & 0xFFu) << (8 * n);
            ++n;
          }
        };
        visit(okU, code - 16u);
        visit(okD, code + 16u);
        visit(okL, code - 1u);
        visit(okR, code + 1u);
        ++hpop;
        if (hpop >= n) act = false;
      }
      if (mine) {
        const int gs = static_cast<int>(cg.y), ge = static_cast<int>(cg.z);
#pragma unroll
        for (int i = 0; i < 8; ++i)
          if (i < n && gs + i < ge) {
            const uint32_t code = static_cast<uint32_t>(ll >> (8 * i)) & 0xFFu;
            queue[gs + i] = static_cast<uint32_t>(cs + (static_cast<int>(code >> 4) - 8) * nh + (static_cast<int>(code & 15u) - 8));
          }
      }
    }
  }
  if (DBG && ob.dbg && lane == 0)
    atomicMax(ob.dbg + static_cast<size_t>(frame) * kDbgSlots + 2, static_cast<unsigned long long>(__builtin_amdgcn_s_memtime()) - t_start);
}

// ---- segment sums and fits, chip-wide (the tail's P2 and P3 of dpx_regions2.h taken out of the frame's workgroup) ----------
// A candidate segment's nine sums are nine strictly sequential float chains in queue order, the seed counted twice
// (plane_extractor.cpp:318-323); different segments are independent.  k_seg_accum<false>: a wave per segment, pulled from a
// counter (64-entry chunks staged through LDS, the next chunk's records in flight while the chains run).  k_seg_accum<true>:
// a HUGE segment (>= kHugeMin cells — a clean 4K frame has one plane of 70 000) takes a workgroup: three producer waves gather
// the records of 64-cell chunks into a ring of LDS buffers, twelve chunks per round, while wave 0 does nothing but read them
// back and add — about 6 cycles per cell instead of 30.
struct TailBufs {
  const dpx_cell* cells;   // [B][C]
  const uint32_t* queue;   // [B][C]
  const uint4* cand_grow;  // [B][pcap]
  const uint32_t* biglist; // [B][pcap]
  const int* hdr;          // [B][48]
  float* cand;             // [B][pcap][12] sums(9), n_pts, grow index, (accepted id: the tail launch)
  float* candfit;          // [B][pcap][12] mean(3), normal(3), d, mse, score, ok
  int* task;               // [B][kCtr]: 2 = segments of k_seg_accum<false>, 3 = of k_seg_accum<true>
  int pcap;
};
constexpr int kAccWaves = 4, kAccStage = 64, kAccChunk = 4;
constexpr int kAccRound = 12, kAccRing = 2 * kAccRound;  // chunks per round of the huge-segment form, buffers in its ring

template <bool HUGE>
__global__ __launch_bounds__(kAccWaves * 64) void k_seg_accum(Params prm, TailBufs tb) {
  const int frame = blockIdx.y;
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int C = prm.C;
  const size_t fo = static_cast<size_t>(frame) * C;
  const dpx_cell* cells = tb.cells + fo;
  const float4* cf4 = reinterpret_cast<const float4*>(cells);
  const uint32_t* queue = tb.queue + fo;
  const uint4* cand_grow = tb.cand_grow + static_cast<size_t>(frame) * tb.pcap;
  float* cand = tb.cand + static_cast<size_t>(frame) * tb.pcap * 12;
  const int* hdr = tb.hdr + static_cast<size_t>(frame) * 48;
  const int ncand = hdr[3], nhuge = hdr[kHdrHuge];
  constexpr int kStage = kAccStage;
  if constexpr (!HUGE) {
    __shared__ __align__(16) float s_stage[kAccWaves][kStage * 9];  // (read back 16 bytes at a time)
    float* st = s_stage[wid];
    for (int guard = 0; guard <= ncand; ++guard) {  // (a wave can take at most every segment once)
      // (kAccChunk segments per pull: the frame's counter serves its atomics one after the other, ~100 ns each)
      int c0 = 0;
      if (lane == 0) c0 = atomicAdd(tb.task + kCtr * frame + 2, kAccChunk);
      c0 = __builtin_amdgcn_readfirstlane(c0);
      if (c0 >= ncand) break;
      const int c1 = c0 + kAccChunk < ncand ? c0 + kAccChunk : ncand;
      for (int ci = c0; ci < c1; ++ci) {
      const uint4 cg = cand_grow[ci];
      const int gi = static_cast<int>(cg.x), gs = static_cast<int>(cg.y), ge = static_cast<int>(cg.z);
      if (__builtin_amdgcn_readfirstlane(ge - gs) < kHugeMin) {  // (the huge ones are the other launch's)
      const int comp = lane < 9 ? lane : 0;
      float acc = reinterpret_cast<const float*>(cells)[static_cast<size_t>(queue[gs]) * 16 + comp];  // the seed, counted twice
      float4 pa = make_float4(0.f, 0.f, 0.f, 0.f), pb = pa, pc = pa;
      if (gs + lane < ge) {
        const size_t v = queue[gs + lane];
        pa = cf4[v * 4 + 0]; pb = cf4[v * 4 + 1]; pc = cf4[v * 4 + 2];
      }
      for (int base = gs; base < ge; base += kStage) {
        const int cnt = (ge - base) < kStage ? (ge - base) : kStage;
        if (lane < cnt) {  // staging is component-major: st[comp][entry]
          float* d = st + lane;
          d[0 * kStage] = pa.x; d[1 * kStage] = pa.y; d[2 * kStage] = pa.z; d[3 * kStage] = pa.w; d[4 * kStage] = pb.x;
          d[5 * kStage] = pb.y; d[6 * kStage] = pb.z; d[7 * kStage] = pb.w; d[8 * kStage] = pc.x;
        }
        wave_sync();
        {
          const int nx = base + kStage + lane;
          if (nx < ge) {
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
    }
  } else {
    if (nhuge == 0) return;
    __shared__ __align__(16) float s_ring[kAccRing][kStage * 9];
    __shared__ int s_task;
    const uint32_t* biglist = tb.biglist + static_cast<size_t>(frame) * tb.pcap;
    for (int guard = 0; guard <= nhuge; ++guard) {
      __syncthreads();
      if (tid == 0) s_task = atomicAdd(tb.task + kCtr * frame + 3, 1);
      __syncthreads();
      const int t = s_task;
      if (t >= nhuge) break;
      const int ci = static_cast<int>(biglist[tb.pcap - 1 - t]);
      const uint4 cg = cand_grow[ci];
      const int gi = static_cast<int>(cg.x), gs = static_cast<int>(cg.y), ge = static_cast<int>(cg.z);
      const int nchunks = (ge - gs + kStage - 1) / kStage;
      const int comp = lane < 9 ? lane : 0;
      float acc = 0.f;
      if (wid == 0) acc = reinterpret_cast<const float*>(cells)[static_cast<size_t>(queue[gs]) * 16 + comp];  // the seed, counted twice
      // round r: the producers fill chunks [12 r, 12 r + 12) while wave 0 adds chunks [12 (r - 1), 12 r)
      const int nrounds = (nchunks + kAccRound - 1) / kAccRound;
      for (int r = 0; r <= nrounds; ++r) {
        if (wid > 0 && r < nrounds) {
          // this wave's four chunks of the round: all queue loads, then all record loads, then the LDS writes
          float4 pa[4], pb[4], pc[4];
          int ck[4];
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            ck[j] = r * kAccRound + (wid - 1) * 4 + j;
            const int e = gs + ck[j] * kStage + lane;
            const bool in = ck[j] < nchunks && e < ge;
            const size_t v = in ? queue[e] : 0u;
            pa[j] = make_float4(0.f, 0.f, 0.f, 0.f); pb[j] = pa[j]; pc[j] = pa[j];
            if (in) { pa[j] = cf4[v * 4 + 0]; pb[j] = cf4[v * 4 + 1]; pc[j] = cf4[v * 4 + 2]; }
          }
#pragma unroll
          for (int j = 0; j < 4; ++j)
            if (ck[j] < nchunks) {
              float* d = s_ring[ck[j] % kAccRing] + lane;
              d[0 * kStage] = pa[j].x; d[1 * kStage] = pa[j].y; d[2 * kStage] = pa[j].z; d[3 * kStage] = pa[j].w; d[4 * kStage] = pb[j].x;
              d[5 * kStage] = pb[j].y; d[6 * kStage] = pb[j].z; d[7 * kStage] = pb[j].w; d[8 * kStage] = pc[j].x;
            }
        }
        if (wid == 0 && r > 0 && lane < 9) {
          for (int k = (r - 1) * kAccRound; k < r * kAccRound && k < nchunks; ++k) {
            const int cnt = (ge - gs - k * kStage) < kStage ? (ge - gs - k * kStage) : kStage;
            const float4* row = reinterpret_cast<const float4*>(s_ring[k % kAccRing] + comp * kStage);
            float tv[kStage];
#pragma unroll
            for (int q = 0; q < kStage / 4; ++q) {
              const float4 x = row[q];
              tv[4 * q] = x.x; tv[4 * q + 1] = x.y; tv[4 * q + 2] = x.z; tv[4 * q + 3] = x.w;
            }
            if (cnt == kStage) {
#pragma unroll
              for (int q = 0; q < kStage; ++q) acc = acc + tv[q];
            } else {
#pragma unroll
              for (int q = 0; q < kStage; ++q)
                if (q < cnt) acc = acc + tv[q];
            }
          }
        }
        __syncthreads();
      }
      float* co = cand + static_cast<size_t>(ci) * 12;
      if (wid == 0) {
        if (lane < 9) co[lane] = acc;
        if (lane == 9) co[9] = __int_as_float((ge - gs + 1) * prm.n);
        if (lane == 10) co[10] = __int_as_float(gi);
      }
    }
  }
}

// grid (ceil(pcap / 256), B): CellSegmentStat::fitPlane of every candidate segment (plane_extractor.cpp:333-337), one thread each
__global__ __launch_bounds__(256) void k_seg_fit(Params prm, TailBufs tb) {
  const int frame = blockIdx.y;
  const int ci = static_cast<int>(blockIdx.x) * 256 + threadIdx.x;
  const int ncand = tb.hdr[static_cast<size_t>(frame) * 48 + 3];
  if (ci >= ncand) return;
  const float* co = tb.cand + (static_cast<size_t>(frame) * tb.pcap + ci) * 12;
  float s[3] = {co[0], co[1], co[2]}, pp[6] = {co[3], co[4], co[5], co[6], co[7], co[8]};
  PlaneFit fit;
  fit_plane(s, pp, __float_as_int(co[9]), fit);
  float* cf = tb.candfit + (static_cast<size_t>(frame) * tb.pcap + ci) * 12;
  cf[0] = fit.mean[0]; cf[1] = fit.mean[1]; cf[2] = fit.mean[2];
  cf[3] = fit.normal[0]; cf[4] = fit.normal[1]; cf[5] = fit.normal[2];
  cf[6] = fit.d; cf[7] = fit.mse; cf[8] = fit.score;
  cf[9] = __int_as_float(fit.score > prm.min_score ? 1 : 0);
}

}  // namespace dpx