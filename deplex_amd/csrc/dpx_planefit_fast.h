// dpx_planefit_fast.h — the tuned depth-input plane-fit kernel (K1+K2) for even patch sizes on images whose
// width is a multiple of the patch (every BASELINE config).
//
// Mapping: one lane = one cell, one wavefront = 64 horizontally adjacent cells of one cell row.  A cell's row of
// P uint16 depths is P*2 contiguous bytes and the next lane's cell starts right after it, so the per-lane row
// loads (dwordx4 + dword for P = 10) of a wavefront cover one contiguous span of the image row: fully coalesced
// HBM reads with no staging; every 128-byte line is fetched once.  The lane then owns all P*P points of its cell
// in registers and runs the reference's exact float32 chains:
//   sum_j     8 residue-class chains + pairwise fold + horizontal add   (Eigen packet redux order)
//   sum p p^T 6 strictly sequential mul-then-add chains                  (Eigen GEMM scalar tail)
// followed by the 3x3 eigen plane fit in f64 (dpx_fit.h) — all in registers, one 64-byte record out.
//
// Back-projection x = ((col - cx) * z) / fx needs two IEEE float divisions per pixel, which cost ~18 VALU issue
// slots each on gfx950 (measured, tools/ubench/valu.hip).  Dividing by the constant fx is done instead as
//   q = a * r;  e = fma(-q, fx, a);  x = fma(e, r, q)        with r = RN(1 / fx)
// (3 issue slots).  That sequence is NOT assumed to be correctly rounded: dpx_verify_fastdiv (k_verify_fastdiv
// below) checks it against the IEEE quotient for EVERY numerator the frame geometry can produce
// ((col - cx) * z for all columns and all 65536 depths, same for rows) when intrinsics are first seen; if any
// quotient differs the kernel is instantiated with IEEE division instead (FASTDIV = false).  The only remaining
// difference is the sign of a zero quotient (z == 0 with a negative numerator gives +0 instead of -0); it can
// influence a result only when every x (or y) of a cell is a zero, i.e. the whole cell has z == 0, and is
// patched in closed form there (sum of signed zeros = -0 iff all terms are -0).
#pragma once
#include <utility>

#include "dpx_kernels.h"

namespace dpx {

// Pixels per scheduling group of the moment loop (see CellAcc::one).  1 with two-row buffers (DPX_PF_ROWBUF) brings the
// P = 10 kernel to 112 vector registers: four plane-fit waves then leave 64 registers per SIMD free, enough for the
// label-expansion waves (16) and a grow wave (48) of the other groups in flight to start on a CU that is full of plane-fit
// work — measured +3 % in the pipeline (the kernel on its own is unchanged; 121 registers = 128 allocated left none).
#ifndef DPX_PF_GROUP
#define DPX_PF_GROUP 1
#endif
#ifndef DPX_PF_WAVES
#define DPX_PF_WAVES 4
#endif
// Diagnostic hook (tools/pfbench.hip defines it to record s_memtime stamps per wave); nothing in the product build.
#ifndef DPX_PF_STAMP
#define DPX_PF_STAMP(slot)
#endif
typedef uint32_t u32x4a __attribute__((ext_vector_type(4), aligned(4)));
typedef float f32x4a8 __attribute__((ext_vector_type(4), aligned(8)));
typedef uint32_t u32x2a __attribute__((ext_vector_type(2), aligned(4)));

template <bool FASTDIV>
__device__ __forceinline__ float div_const(float a, float b, float r) {
  if (FASTDIV) {
    const float q = a * r;
    const float e = __builtin_fmaf(-q, b, a);
    return __builtin_fmaf(e, r, q);
  }
  return a / b;
}

// Exhaustive check of div_const<true> against IEEE division for numerators (idx - c) * z, idx in [0, count),
// z in [0, 65535].  mismatches[0] counts quotients that differ (as values; +0 == -0).
__global__ __launch_bounds__(256) void k_verify_fastdiv(int count, float c, float b, float r, unsigned int* mismatches) {
  const long long gid = static_cast<long long>(blockIdx.x) * 256 + threadIdx.x;
  const long long total = static_cast<long long>(count) * 65536;
  if (gid >= total) return;
  const int idx = static_cast<int>(gid >> 16);
  const float z = static_cast<float>(static_cast<int>(gid & 0xFFFF));
  const float a = (static_cast<float>(idx) - c) * z;
  const float fast = div_const<true>(a, b, r);
  const float ieee = a / b;
  const bool same = (fast == ieee) || (fast != fast && ieee != ieee);
  if (!same) atomicAdd(mismatches, 1u);
}

// Chain accumulator for the 8 residue classes, initialised to -0.0f so that the first element
// can be ADDED ((-0) + v == v bit for bit, signed zeros included) and no first-touch special case is needed.
struct Chains8 {
  float c[8];
  __device__ __forceinline__ void init() {
#pragma unroll
    for (int m = 0; m < 8; ++m) c[m] = -0.0f;
  }
  __device__ __forceinline__ void fold() {
    c[0] = c[0] + c[4];
    c[1] = c[1] + c[5];
    c[2] = c[2] + c[6];
    c[3] = c[3] + c[7];
  }
};

// ZI (depth input with n * 65535 < 2^24, i.e. P <= 16): every z is an integer below 2^16 and every partial sum of z
// stays below 2^24, so all float additions of the z column are exact and the column sum is the same number in any
// order — it is accumulated as an integer, two pixels per instruction (v_sad_u16 on the raw input dword), and
// converted once.  The count of valid (non-zero) pixels is kept per 16-bit half (v_pk_min_u16 + v_pk_add_u16).
typedef unsigned short dpx_u16x2 __attribute__((ext_vector_type(2)));
template <int P, bool FASTDIV, bool ZI = false>
struct CellAcc {
  static constexpr int n = P * P;
  static constexpr int A2 = (n / 8) * 8;
  static_assert(!ZI || n <= 256, "integer z sums need n * 65535 < 2^24");
  Chains8 sx, sy, sz;
  uint32_t zsum;      // ZI: sum of the raw depths
  dpx_u16x2 vcnt2;    // ZI: valid pixels seen in the low / high halves of the input dwords
  float xx, xy, xz, yy, yz, zz;
  float fx0, fy0, fz0, lx, ly, lz;
  float hprev, vprev;
  uint32_t disc;  // discontinuities: centre row in bits 0..15, centre column in bits 16..31 (one register: none to spare)
  int valid_pts;
  bool folded;

  __device__ __forceinline__ void init() {
    sx.init(); sy.init(); sz.init();
    xx = xy = xz = yy = yz = zz = 0.f;
    fx0 = fy0 = fz0 = lx = ly = lz = 0.f;
    hprev = vprev = 0.f;
    disc = 0u;
    valid_pts = 0;
    folded = false;
    zsum = 0u;
    vcnt2 = dpx_u16x2{0, 0};
  }
  // ZI: both pixels of one input dword
  __device__ __forceinline__ void dword(uint32_t w) {
    zsum = __builtin_amdgcn_sad_u16(w, 0u, zsum);
    // min(depth, 1) per half, added to the packed counters (the compiler expands the vector form of the minimum into
    // compares and selects).  Volatile: left to itself the scheduler computes the minima of several rows' dwords at
    // once, as soon as their loads are back, and keeps them in registers until the additions (30 registers at the peak).
    uint32_t m, vc;
    __builtin_memcpy(&vc, &vcnt2, 4);
    asm volatile("v_pk_min_u16 %1, %2, %3\n\tv_pk_add_u16 %0, %0, %1" : "+v"(vc), "=&v"(m) : "v"(w), "s"(0x00010001u));
    __builtin_memcpy(&vcnt2, &vc, 4);
  }
  __device__ __forceinline__ bool continuous(int max_disc) const {  // (at most P discontinuities per scan: no carry)
    return (static_cast<int>(disc & 0xFFFFu) < max_disc) && (static_cast<int>(disc >> 16) < max_disc);
  }
  __device__ __forceinline__ int valid_count() const { return ZI ? static_cast<int>(vcnt2.x) + static_cast<int>(vcnt2.y) : valid_pts; }
  __device__ __forceinline__ float z_sum() const {
    return ZI ? static_cast<float>(zsum) : (sz.c[0] + sz.c[2]) + (sz.c[1] + sz.c[3]);
  }

  // One pixel.  KM = k & 7 and the position flags are compile-time; the row index may be a runtime value.
  template <int J, int KM, bool TAIL, int TAILIDX, bool FIRST, bool LAST, bool HROW>
  __device__ __forceinline__ void pixel(uint32_t zi, float cx_j, float cy_i, const Params& prm, float rx, float ry) {
    float z = static_cast<float>(zi);
    // hide that z came from an integer: otherwise z * z is rewritten as convert(zi * zi) — two instructions for one
    asm volatile("" : "+v"(z));
    const float x = div_const<FASTDIV>(cx_j * z, prm.fx, rx);
    const float y = div_const<FASTDIV>(cy_i * z, prm.fy, ry);
    if (!ZI) valid_pts += zi != 0u;
    point<J, KM, TAIL, TAILIDX, FIRST, LAST, HROW>(x, y, z, prm);
  }

  // One point of the organised cloud (also the tail of pixel()): continuity scans, moments.
  template <int J, int KM, bool TAIL, int TAILIDX, bool FIRST, bool LAST, bool HROW>
  __device__ __forceinline__ void point(float x, float y, float z, const Params& prm) {
    if (FIRST) { fx0 = x; fy0 = y; fz0 = z; }
    if (LAST) { lx = x; ly = y; lz = z; }
    if (HROW) {  // centre row, in j order (cell_segment.cpp:62-76)
      if (J == 0) hprev = z;
      if (z > 0 && __builtin_fabsf(z - hprev) < prm.disc_thr) hprev = z;
      else if (z > 0) disc += 1u;
    }
    if (J == P / 2) {  // centre column, in i order (cell_segment.cpp:78-91); vprev is preset to the row-0 value
      if (z > 0 && __builtin_fabsf(z - vprev) < prm.disc_thr) vprev = z;
      else if (z > 0) disc += 0x10000u;
    }
    if (!TAIL) {
      sx.c[KM] = sx.c[KM] + x;
      sy.c[KM] = sy.c[KM] + y;
      if (!ZI) sz.c[KM] = sz.c[KM] + z;
    } else {  // trailing 4-packet after the 8-wide part: fold first, then add to chains 0..3
      if (TAILIDX == 0 && A2 > 0) { sx.fold(); sy.fold(); if (!ZI) sz.fold(); }
      sx.c[TAILIDX] = sx.c[TAILIDX] + x;
      sy.c[TAILIDX] = sy.c[TAILIDX] + y;
      if (!ZI) sz.c[TAILIDX] = sz.c[TAILIDX] + z;
    }
    if (n + 6 >= 20 || !FIRST) {
      xx = x * x + xx; xy = x * y + xy; xz = x * z + xz; yy = y * y + yy; yz = y * z + yz; zz = z * z + zz;
    } else {
      xx = x * x; xy = x * y; xz = x * z; yy = y * y; yz = y * z; zz = z * z;
    }
  }

  // One cell row with compile-time row residue RM = i & 3 (k & 7 depends on i only through i & 3 for even P).
  // STATIC_I >= 0 marks rows whose index is known at compile time (needed for first/last/tail handling).
  template <int RM, int STATIC_I>
  __device__ __forceinline__ void row(const uint32_t (&w)[P / 2], const float (&cxj)[P], float cy_i, bool is_hrow,
                                      const Params& prm, float rx, float ry) {
    row_impl<RM, STATIC_I>(w, cxj, cy_i, is_hrow, prm, rx, ry, std::make_integer_sequence<int, P>{});
  }
  template <int RM, int STATIC_I, int... Js>
  __device__ __forceinline__ void row_impl(const uint32_t (&w)[P / 2], const float (&cxj)[P], float cy_i, bool is_hrow,
                                           const Params& prm, float rx, float ry, std::integer_sequence<int, Js...>) {
    (one<RM, STATIC_I, Js>(w, cxj, cy_i, is_hrow, prm, rx, ry), ...);
  }
  template <int RM, int STATIC_I, int J>
  __device__ __forceinline__ void one(const uint32_t (&w)[P / 2], const float (&cxj)[P], float cy_i, bool is_hrow,
                                      const Params& prm, float rx, float ry) {
    constexpr int KM = (RM * P + J) & 7;
    constexpr int K = STATIC_I >= 0 ? STATIC_I * P + J : -1;
    constexpr bool TAIL = K >= 0 && K >= A2;
    constexpr int TAILIDX = TAIL ? K - A2 : 0;
    constexpr bool FIRST = K == 0;
    constexpr bool LAST = K == n - 1;
    const uint32_t zi = (J & 1) ? (w[J >> 1] >> 16) : (w[J >> 1] & 0xFFFFu);
    if (ZI && (J & 1) == 0) dword(w[J >> 1]);
    if (STATIC_I >= 0) {
      pixel<J, KM, TAIL, TAILIDX, FIRST, LAST, STATIC_I == P / 2>(zi, cxj[J], cy_i, prm, rx, ry);
    } else if (is_hrow) {
      pixel<J, KM, false, 0, false, false, true>(zi, cxj[J], cy_i, prm, rx, ry);
    } else {
      pixel<J, KM, false, 0, false, false, false>(zi, cxj[J], cy_i, prm, rx, ry);
    }
    // Order barriers: the whole unrolled cell is one basic block, and without them instruction selection separates
    // the work into phases (all converts/products first, the sequential chain adds last) and spills hundreds of
    // values.  The empty asm pins every accumulator touched by the last DPX_PF_GROUP pixels at this point of the
    // instruction stream; inside a group the scheduler is free to interleave the pixels' independent work (ILP).
    if ((J % DPX_PF_GROUP) == DPX_PF_GROUP - 1 || J == P - 1) pin();
  }

  __device__ __forceinline__ void pin() {
    {
      asm volatile("" : "+v"(sx.c[0]), "+v"(sx.c[1]), "+v"(sx.c[2]), "+v"(sx.c[3]), "+v"(sx.c[4]), "+v"(sx.c[5]), "+v"(sx.c[6]), "+v"(sx.c[7]));
      asm volatile("" : "+v"(sy.c[0]), "+v"(sy.c[1]), "+v"(sy.c[2]), "+v"(sy.c[3]), "+v"(sy.c[4]), "+v"(sy.c[5]), "+v"(sy.c[6]), "+v"(sy.c[7]));
      if (!ZI) asm volatile("" : "+v"(sz.c[0]), "+v"(sz.c[1]), "+v"(sz.c[2]), "+v"(sz.c[3]), "+v"(sz.c[4]), "+v"(sz.c[5]), "+v"(sz.c[6]), "+v"(sz.c[7]));
      asm volatile("" : "+v"(xx), "+v"(xy), "+v"(xz), "+v"(yy), "+v"(yz), "+v"(zz));
    }
  }

  // ---- organised-cloud input: row i of the cell is 3*P floats (x y z per point), all indices static
  template <int I>
  __device__ __forceinline__ void cloud_row(const float (&w)[3 * P], const Params& prm) {
    cloud_row_impl<I>(w, prm, std::make_integer_sequence<int, P>{});
  }
  template <int I, int... Js>
  __device__ __forceinline__ void cloud_row_impl(const float (&w)[3 * P], const Params& prm, std::integer_sequence<int, Js...>) {
    (cloud_one<I, Js>(w, prm), ...);
  }
  template <int I, int J>
  __device__ __forceinline__ void cloud_one(const float (&w)[3 * P], const Params& prm) {
    constexpr int K = I * P + J;
    constexpr int KM = K & 7;
    constexpr bool TAIL = K >= A2;
    constexpr int TAILIDX = TAIL ? K - A2 : 0;
    const float x = w[3 * J], y = w[3 * J + 1], z = w[3 * J + 2];
    valid_pts += z > 0;
    point<J, KM, TAIL, TAILIDX, K == 0, K == n - 1, I == P / 2>(x, y, z, prm);
    if ((J % DPX_PF_GROUP) == DPX_PF_GROUP - 1 || J == P - 1) pin();
  }

  // organised-cloud row with compile-time row residue RM = i & 3; STATIC_I >= 0 as in row<>
  template <int RM, int STATIC_I>
  __device__ __forceinline__ void cloud_row_rm(const float (&w)[3 * P], bool is_hrow, const Params& prm) {
    cloud_row_rm_impl<RM, STATIC_I>(w, is_hrow, prm, std::make_integer_sequence<int, P>{});
  }
  template <int RM, int STATIC_I, int... Js>
  __device__ __forceinline__ void cloud_row_rm_impl(const float (&w)[3 * P], bool is_hrow, const Params& prm, std::integer_sequence<int, Js...>) {
    (cloud_one_rm<RM, STATIC_I, Js>(w, is_hrow, prm), ...);
  }
  template <int RM, int STATIC_I, int J>
  __device__ __forceinline__ void cloud_one_rm(const float (&w)[3 * P], bool is_hrow, const Params& prm) {
    constexpr int KM = (RM * P + J) & 7;
    constexpr int K = STATIC_I >= 0 ? STATIC_I * P + J : -1;
    constexpr bool TAIL = K >= 0 && K >= A2;
    constexpr int TAILIDX = TAIL ? K - A2 : 0;
    const float x = w[3 * J], y = w[3 * J + 1], z = w[3 * J + 2];
    valid_pts += z > 0;
    if (STATIC_I >= 0) {
      point<J, KM, TAIL, TAILIDX, K == 0, K == n - 1, STATIC_I == P / 2>(x, y, z, prm);
    } else if (is_hrow) {
      point<J, KM, false, 0, false, false, true>(x, y, z, prm);
    } else {
      point<J, KM, false, 0, false, false, false>(x, y, z, prm);
    }
    if ((J % DPX_PF_GROUP) == DPX_PF_GROUP - 1 || J == P - 1) pin();
  }
};

template <int P>
__device__ __forceinline__ void load_cell_row(const uint16_t* rowp, uint32_t (&w)[P / 2]) {
  constexpr int DW = P / 2;
  const uint32_t* rp = reinterpret_cast<const uint32_t*>(rowp);
  int d = 0;
#pragma unroll
#ifdef DPX_PF_NT_LOADS  // experiment: depth pixels are read once — keep them out of the way of the data the region kernels re-read
  for (; d + 4 <= DW; d += 4) {
    const u32x4a v = __builtin_nontemporal_load(reinterpret_cast<const u32x4a*>(rp + d));
    w[d] = v.x; w[d + 1] = v.y; w[d + 2] = v.z; w[d + 3] = v.w;
  }
  if (d + 2 <= DW) {
    const u32x2a v = __builtin_nontemporal_load(reinterpret_cast<const u32x2a*>(rp + d));
    w[d] = v.x; w[d + 1] = v.y;
    d += 2;
  }
  if (d < DW) w[d] = __builtin_nontemporal_load(rp + d);
#else
  for (; d + 4 <= DW; d += 4) {
    const u32x4a v = *reinterpret_cast<const u32x4a*>(rp + d);
    w[d] = v.x; w[d + 1] = v.y; w[d + 2] = v.z; w[d + 3] = v.w;
  }
  if (d + 2 <= DW) {
    const u32x2a v = *reinterpret_cast<const u32x2a*>(rp + d);
    w[d] = v.x; w[d + 1] = v.y;
    d += 2;
  }
  if (d < DW) w[d] = rp[d];
#endif
}

// One launch fits the frames of up to kMaxGroup submissions (a dpx_pipeline group): frame f of the launch belongs to
// submission f / per_sub, which has its own input buffer and its own intrinsics (and verified reciprocals).  The cell
// records of all submissions are contiguous (frame-major), as the region kernels expect.
struct PfGroup {
  const void* in[kMaxGroup];
  float fx[kMaxGroup], fy[kMaxGroup], cx[kMaxGroup], cy[kMaxGroup], rx[kMaxGroup], ry[kMaxGroup];
  int per_sub;
};

// DIAG != 0 are timing-only instantiations (results are garbage): 1 = skip the plane fit / record maths (moments only),
// 2 = loads + record store only.  Selected with DPX_PF_DIAG for tools/pf_breakdown.py; never used by the product path.
#ifdef DPX_PF_VGPR
#define DPX_PF_VGPR_ATTR __attribute__((amdgpu_num_vgpr(DPX_PF_VGPR)))
#else
#define DPX_PF_VGPR_ATTR
#endif
// Rows two at a time, every row index static: rows R, R+1 are in `a`, rows R+2, R+3 are fetched into `b` before `a` is
// evaluated, then the roles swap (no copies).  Two-row buffers instead of four-row ones keep the kernel at <= 120
// vector registers (see k_planefit_depth).
template <int P, int R>
struct PairRows {
  template <class ACC>
  static __device__ __forceinline__ void run(ACC& acc, const uint16_t* src, long long row_stride, uint32_t (&a)[2][P / 2],
                                             uint32_t (&b)[2][P / 2], const float (&cxj)[P], int row0, const Params& prm,
                                             float rx, float ry) {
    if (R + 2 < P) load_cell_row<P>(src + static_cast<long long>(R + 2) * row_stride, b[0]);
    if (R + 3 < P) load_cell_row<P>(src + static_cast<long long>(R + 3) * row_stride, b[1]);
    acc.template row<R & 3, R>(a[0], cxj, static_cast<float>(row0 + R) - prm.cy, false, prm, rx, ry);
    if (R + 1 < P) acc.template row<(R + 1) & 3, (R + 1 < P ? R + 1 : 0)>(a[1], cxj, static_cast<float>(row0 + R + 1) - prm.cy, false, prm, rx, ry);
    asm volatile("" ::: "memory");  // keep the following rows' loads where they are (one pair ahead, not all at the top)
    if (R + 2 < P) PairRows<P, (R + 2 < P ? R + 2 : P)>::run(acc, src, row_stride, b, a, cxj, row0, prm, rx, ry);
  }
};
template <int P>
struct PairRows<P, P> {
  template <class ACC>
  static __device__ __forceinline__ void run(ACC&, const uint16_t*, long long, uint32_t (&)[2][P / 2], uint32_t (&)[2][P / 2],
                                             const float (&)[P], int, const Params&, float, float) {}
};

// The same with one-row buffers: row R is in `a`, row R+1 is fetched into `b` before `a` is evaluated.
template <int P, int R>
struct SingleRows {
  template <class ACC>
  static __device__ __forceinline__ void run(ACC& acc, const uint16_t* src, long long row_stride, uint32_t (&a)[P / 2],
                                             uint32_t (&b)[P / 2], const float (&cxj)[P], int row0, const Params& prm, float rx,
                                             float ry) {
    if (R + 1 < P) load_cell_row<P>(src + static_cast<long long>(R + 1) * row_stride, b);
    acc.template row<R & 3, R>(a, cxj, static_cast<float>(row0 + R) - prm.cy, false, prm, rx, ry);
    asm volatile("" ::: "memory");
    if (R + 1 < P) SingleRows<P, (R + 1 < P ? R + 1 : P)>::run(acc, src, row_stride, b, a, cxj, row0, prm, rx, ry);
  }
};
template <int P>
struct SingleRows<P, P> {
  template <class ACC>
  static __device__ __forceinline__ void run(ACC&, const uint16_t*, long long, uint32_t (&)[P / 2], uint32_t (&)[P / 2],
                                             const float (&)[P], int, const Params&, float, float) {}
};

#ifndef DPX_PF_ROWBUF
#define DPX_PF_ROWBUF 2
#endif
template <int P, bool FASTDIV, int DIAG>
__device__ __forceinline__ void pf_depth_task(Params prm, const PfGroup& grp, dpx_cell* __restrict__ cells, const long long wave, const int lane) {
  static_assert(P % 2 == 0 && P >= 2, "even patch sizes only");
  constexpr int n = P * P;
  constexpr int DW = P / 2;  // dwords per cell row
  constexpr int A2 = (n / 8) * 8;
  // rows handled by the rolled 4-row loop: whole groups of rows that lie entirely inside the 8-wide part, never row 0
  // (first-point bookkeeping) — row 0..3 are peeled as the static prologue group.
  // rows [0, FULL) are whole 4-row groups below A2 that do not contain the last row (its last point is needed)
  constexpr int FULL_A = ((A2 / P) / 4) * 4, FULL_B = ((P - 1) / 4) * 4;
  constexpr int FULL = FULL_A < FULL_B ? FULL_A : FULL_B;
  // frame, cell row and strip are wave-uniform: kept in scalar registers (the vector registers are all spoken for)
  const int strips = (prm.nh + 63) >> 6;
  const int per_frame = prm.nv * strips;
  const int frame = __builtin_amdgcn_readfirstlane(static_cast<int>(wave / per_frame));
  const int rem = __builtin_amdgcn_readfirstlane(static_cast<int>(wave - static_cast<long long>(frame) * per_frame));
  const int cr = __builtin_amdgcn_readfirstlane(rem / strips);
  const int cc0 = __builtin_amdgcn_readfirstlane((rem - cr * strips) * 64);
  const int cc = cc0 + lane;
  if (cc >= prm.nh) return;
  // this wave's submission (wave-uniform: scalar loads from the argument block)
  const int sub = __builtin_amdgcn_readfirstlane(frame / grp.per_sub);
  prm.fx = grp.fx[sub]; prm.fy = grp.fy[sub]; prm.cx = grp.cx[sub]; prm.cy = grp.cy[sub];
  const float rx = grp.rx[sub], ry = grp.ry[sub];
  const uint16_t* depth = static_cast<const uint16_t*>(grp.in[sub]);

  const int row0 = cr * P, col0 = cc * P;
  const uint16_t* src = depth + static_cast<long long>(frame - sub * grp.per_sub) * prm.N + static_cast<long long>(row0) * prm.W + col0;
  DPX_PF_STAMP(0)

  float cxj[P];
#pragma unroll
  for (int j = 0; j < P; ++j) cxj[j] = static_cast<float>(col0 + j) - prm.cx;

  CellAcc<P, FASTDIV, (P * P <= 256)> acc;
  acc.init();

  constexpr bool kPairs = DPX_PF_ROWBUF == 2 && DIAG != 2 && P >= 2;
  constexpr bool kSingles = DPX_PF_ROWBUF == 1 && DIAG != 2;
  if constexpr (kSingles) {
    uint32_t a[DW], b[DW];
    load_cell_row<P>(src, a);
    {
      const uint32_t wv = a[(P / 2) / 2];
      acc.vprev = static_cast<float>(((P / 2) & 1) ? (wv >> 16) : (wv & 0xFFFFu));  // k = P/2 -> (i = 0, j = P/2)
    }
    SingleRows<P, 0>::run(acc, src, static_cast<long long>(prm.W), a, b, cxj, row0, prm, rx, ry);
  } else if constexpr (kPairs) {
    uint32_t a[2][DW], b[2][DW];
    load_cell_row<P>(src, a[0]);
    if (1 < P) load_cell_row<P>(src + static_cast<long long>(prm.W), a[1]);
    {
      const uint32_t wv = a[0][(P / 2) / 2];
      acc.vprev = static_cast<float>(((P / 2) & 1) ? (wv >> 16) : (wv & 0xFFFFu));  // k = P/2 -> (i = 0, j = P/2)
    }
    PairRows<P, 0>::run(acc, src, static_cast<long long>(prm.W), a, b, cxj, row0, prm, rx, ry);
  } else {
  // group buffers: `cur` holds the 4 rows being processed, `nxt` the 4 rows in flight behind them
  uint32_t cur[4][DW], nxt[4][DW];
#pragma unroll
  for (int r = 0; r < 4; ++r)
    if (r < P) load_cell_row<P>(src + static_cast<long long>(r) * prm.W, cur[r]);
  {
    const uint32_t wv = cur[0][(P / 2) / 2];
    acc.vprev = static_cast<float>(((P / 2) & 1) ? (wv >> 16) : (wv & 0xFFFFu));  // k = P/2 -> (i = 0, j = P/2)
  }

  if (DIAG == 2) {
    // loads + store only: fold the raw words so that the loads stay live
    uint32_t x = 0;
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
      for (int d = 0; d < DW; ++d) x ^= cur[r][d];
    for (int g = 4; g < P; g += 4) {
#pragma unroll
      for (int r = 0; r < 4; ++r)
        if (g + r < P) {
          load_cell_row<P>(src + static_cast<long long>(g + r) * prm.W, nxt[r]);
#pragma unroll
          for (int d = 0; d < DW; ++d) x ^= nxt[r][d];
        }
    }
    acc.xx = __uint_as_float(x);
  } else if (FULL >= 4) {
    // ---- group 0 (rows 0..3), static row indices
#pragma unroll
    for (int r = 0; r < 4; ++r)
      if (4 + r < P) load_cell_row<P>(src + static_cast<long long>(4 + r) * prm.W, nxt[r]);
    acc.template row<0, 0>(cur[0], cxj, static_cast<float>(row0 + 0) - prm.cy, false, prm, rx, ry);
    acc.template row<1, 1>(cur[1], cxj, static_cast<float>(row0 + 1) - prm.cy, false, prm, rx, ry);
    acc.template row<2, 2>(cur[2], cxj, static_cast<float>(row0 + 2) - prm.cy, false, prm, rx, ry);
    acc.template row<3, 3>(cur[3], cxj, static_cast<float>(row0 + 3) - prm.cy, false, prm, rx, ry);
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
      for (int d = 0; d < DW; ++d) cur[r][d] = nxt[r][d];
    // ---- groups 1.. (rows 4..FULL-1), rolled: the row index is a runtime value, k & 7 is still static
#pragma nounroll
    for (int g = 4; g < FULL; g += 4) {
#pragma unroll
      for (int r = 0; r < 4; ++r)
        if (g + 4 + r < P) load_cell_row<P>(src + static_cast<long long>(g + 4 + r) * prm.W, nxt[r]);
      acc.template row<0, -1>(cur[0], cxj, static_cast<float>(row0 + g + 0) - prm.cy, g + 0 == P / 2, prm, rx, ry);
      acc.template row<1, -1>(cur[1], cxj, static_cast<float>(row0 + g + 1) - prm.cy, g + 1 == P / 2, prm, rx, ry);
      acc.template row<2, -1>(cur[2], cxj, static_cast<float>(row0 + g + 2) - prm.cy, g + 2 == P / 2, prm, rx, ry);
      acc.template row<3, -1>(cur[3], cxj, static_cast<float>(row0 + g + 3) - prm.cy, g + 3 == P / 2, prm, rx, ry);
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int d = 0; d < DW; ++d) cur[r][d] = nxt[r][d];
    }
  }
  // ---- remaining rows FULL..P-1 (at most 3 + the rows that contain the trailing packet), static indices
  if (DIAG != 2) {
    constexpr int R0 = FULL >= 4 ? FULL : 0;
    if (R0 + 0 < P) acc.template row<(R0 + 0) & 3, R0 + 0 < P ? R0 + 0 : 0>(cur[0], cxj, static_cast<float>(row0 + R0 + 0) - prm.cy, false, prm, rx, ry);
    if (R0 + 1 < P) acc.template row<(R0 + 1) & 3, R0 + 1 < P ? R0 + 1 : 0>(cur[1], cxj, static_cast<float>(row0 + R0 + 1) - prm.cy, false, prm, rx, ry);
    if (R0 + 2 < P) acc.template row<(R0 + 2) & 3, R0 + 2 < P ? R0 + 2 : 0>(cur[2], cxj, static_cast<float>(row0 + R0 + 2) - prm.cy, false, prm, rx, ry);
    if (R0 + 3 < P) acc.template row<(R0 + 3) & 3, R0 + 3 < P ? R0 + 3 : 0>(cur[3], cxj, static_cast<float>(row0 + R0 + 3) - prm.cy, false, prm, rx, ry);
    static_assert(P - R0 <= 4, "at most one partial group after the rolled loop");
  }
  }  // !kPairs

  if (A2 > 0 && n == A2) { acc.sx.fold(); acc.sy.fold(); acc.sz.fold(); }
  DPX_PF_STAMP(1)
  float sum[3], pp[6];
  sum[0] = (acc.sx.c[0] + acc.sx.c[2]) + (acc.sx.c[1] + acc.sx.c[3]);
  sum[1] = (acc.sy.c[0] + acc.sy.c[2]) + (acc.sy.c[1] + acc.sy.c[3]);
  sum[2] = acc.z_sum();
  const int valid_pts = acc.valid_count();
  if (FASTDIV && valid_pts == 0) {  // every x and y is a signed zero: restore the IEEE sign of the sums
    // (a sum of signed zeros is -0 iff every term is -0, i.e. iff (col - cx) * 0 / fx is -0 for every column of the cell;
    // evaluated here, in the rare branch, rather than carried through the kernel)
    const int lane_z = static_cast<int>(__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)));
    bool all_neg_x = true, all_neg_y = true;
    for (int j = 0; j < P; ++j)
      all_neg_x = all_neg_x && (__builtin_signbit(static_cast<float>((cc0 + lane_z) * P + j) - prm.cx) != __builtin_signbit(prm.fx));
    for (int i = 0; i < P; ++i)
      all_neg_y = all_neg_y && (__builtin_signbit(static_cast<float>(row0 + i) - prm.cy) != __builtin_signbit(prm.fy));
    sum[0] = all_neg_x ? -0.0f : 0.0f;
    sum[1] = all_neg_y ? -0.0f : 0.0f;
  }
  pp[0] = acc.xx; pp[1] = acc.xy; pp[2] = acc.xz; pp[3] = acc.yy; pp[4] = acc.yz; pp[5] = acc.zz;
  const bool cont = acc.continuous(prm.max_disc);
  const bool valid = prm.valid_thr >= 0 && static_cast<long long>(valid_pts) >= prm.valid_thr && cont;

  dpx_cell rec;
  if (DIAG == 0) {
    if (valid) {
      finish_cell(prm, n, sum, pp, acc.fx0 - acc.lx, acc.fy0 - acc.ly, acc.fz0 - acc.lz, true, rec);
    } else {  // the reference does not compute statistics for invalid cells either (cell_segment.cpp:30)
      rec.sum[0] = rec.sum[1] = rec.sum[2] = 0.f;
      rec.pp[0] = rec.pp[1] = rec.pp[2] = rec.pp[3] = rec.pp[4] = rec.pp[5] = 0.f;
      rec.normal[0] = rec.normal[1] = rec.normal[2] = 0.f;
      rec.d = 0.f;
      rec.mse = 3.402823466e+38f;
      rec.tol = 0.f;
      rec.info = 0u;
    }
  } else {
    rec.sum[0] = sum[0]; rec.sum[1] = sum[1]; rec.sum[2] = sum[2];
    rec.pp[0] = pp[0]; rec.pp[1] = pp[1]; rec.pp[2] = pp[2]; rec.pp[3] = pp[3]; rec.pp[4] = pp[4]; rec.pp[5] = pp[5];
    rec.normal[0] = acc.fx0 - acc.lx; rec.normal[1] = acc.fy0 - acc.ly; rec.normal[2] = acc.fz0 - acc.lz;
    rec.d = 0.f; rec.mse = 0.f; rec.tol = 0.f; rec.info = valid ? 1u : 0u;
  }
  // record address = scalar base of the strip + this lane's index, re-derived here (v_mbcnt) instead of being carried
  // in a vector register pair through the whole kernel
  const int lane_now = static_cast<int>(__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)));
  float4* dst = reinterpret_cast<float4*>(cells + static_cast<long long>(frame) * prm.C + static_cast<long long>(cr) * prm.nh + cc0) + 4 * lane_now;
  const float4* s4 = reinterpret_cast<const float4*>(&rec);
  dst[0] = s4[0];
  dst[1] = s4[1];
  dst[2] = s4[2];
  dst[3] = s4[3];
  DPX_PF_STAMP(2)
}

// One wave per (frame, cell row, 64-cell strip) task.  (A persistent grid walking a task list with the grid stride was
// tried, to keep part of the register file free for the other groups' region workgroups: it made no difference in the
// pipeline, and the loop around the body cost 119 scalar and 36 vector register spills.)
template <int P, bool FASTDIV, int DIAG = 0>
__global__ __launch_bounds__(512, DPX_PF_WAVES) DPX_PF_VGPR_ATTR void k_planefit_depth(Params prm, PfGroup grp, dpx_cell* __restrict__ cells) {
  WaveTrace trace_(TK_PLANEFIT);
#ifdef DPX_PF_PAD128
  asm volatile("" ::: "v127");  // A/B switch: make the kernel allocate the full 128 registers
#endif
  const long long wave = static_cast<long long>(blockIdx.x) * (blockDim.x >> 6) + (threadIdx.x >> 6);
  if (wave >= static_cast<long long>(prm.batch) * prm.nv * ((prm.nh + 63) >> 6)) return;
  pf_depth_task<P, FASTDIV, DIAG>(prm, grp, cells, wave, threadIdx.x & 63);
}

// ---- organised-cloud input (the reference API's own input, plane_extractor.h:48): 12 B/point, no back-projection --
// Same lane = cell mapping; a cell row is 3*P contiguous floats.  One row is in flight behind the row being
// processed (120 B per lane for P = 10).  Memory-bound: 17 VALU ops per 12 input bytes.
template <int P>
__device__ __forceinline__ void load_cloud_row(const float* rowp, float (&w)[3 * P]) {
  constexpr int DWR = 3 * P;
  const uint32_t* rp = reinterpret_cast<const uint32_t*>(rowp);
  int d = 0;
#pragma unroll
  for (; d + 4 <= DWR; d += 4) {
    const u32x4a v = *reinterpret_cast<const u32x4a*>(rp + d);
    w[d] = __uint_as_float(v.x); w[d + 1] = __uint_as_float(v.y); w[d + 2] = __uint_as_float(v.z); w[d + 3] = __uint_as_float(v.w);
  }
  if (d + 2 <= DWR) {
    const u32x2a v = *reinterpret_cast<const u32x2a*>(rp + d);
    w[d] = __uint_as_float(v.x); w[d + 1] = __uint_as_float(v.y);
    d += 2;
  }
  if (d < DWR) w[d] = __uint_as_float(rp[d]);
}

template <int P, int I>
struct CloudRows {
  template <class ACC>
  static __device__ __forceinline__ void run(ACC& acc, const float* src, long long row_stride, float (&a)[3 * P], float (&b)[3 * P],
                                             const Params& prm) {
    // row I is in `a`; fetch row I+1 into `b`, process row I, then continue with the roles swapped
    if (I + 1 < P) load_cloud_row<P>(src + static_cast<long long>(I + 1) * row_stride, b);
    if (I == 0) acc.vprev = a[3 * (P / 2) + 2];  // k = P/2 -> (i = 0, j = P/2)
    acc.template cloud_row<I>(a, prm);
    asm volatile("" ::: "memory");  // keep the next row's loads where they are (one row ahead, not all at the top)
    if (I + 1 < P) CloudRows<P, (I + 1 < P ? I + 1 : P)>::run(acc, src, row_stride, b, a, prm);
  }
};
template <int P>
struct CloudRows<P, P> {
  template <class ACC>
  static __device__ __forceinline__ void run(ACC&, const float*, long long, float (&)[3 * P], float (&)[3 * P], const Params&) {}
};

template <int P>
__global__ __launch_bounds__(256, 3) void k_planefit_cloud(Params prm, const float* __restrict__ xyz,
                                                                       dpx_cell* __restrict__ cells) {
  static_assert(P % 2 == 0 && P >= 2, "even patch sizes only");
  constexpr int n = P * P;
  constexpr int A2 = (n / 8) * 8;
  const int strips = (prm.nh + 63) >> 6;
  const long long wave = static_cast<long long>(blockIdx.x) * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  const long long per_frame = static_cast<long long>(prm.nv) * strips;
  if (wave >= per_frame * prm.batch) return;
  const int frame = static_cast<int>(wave / per_frame);
  const int rem = static_cast<int>(wave - frame * per_frame);
  const int cr = rem / strips;
  const int cc = (rem - cr * strips) * 64 + lane;
  if (cc >= prm.nh) return;
  const long long row_stride = 3ll * prm.W;
  const float* src = xyz + 3ll * (static_cast<long long>(frame) * prm.N + static_cast<long long>(cr) * P * prm.W + static_cast<long long>(cc) * P);

  CellAcc<P, false> acc;
  acc.init();
  float a[3 * P], b[3 * P];
  load_cloud_row<P>(src, a);
  CloudRows<P, 0>::run(acc, src, row_stride, a, b, prm);

  if (A2 > 0 && n == A2) { acc.sx.fold(); acc.sy.fold(); acc.sz.fold(); }
  float sum[3], pp[6];
  sum[0] = (acc.sx.c[0] + acc.sx.c[2]) + (acc.sx.c[1] + acc.sx.c[3]);
  sum[1] = (acc.sy.c[0] + acc.sy.c[2]) + (acc.sy.c[1] + acc.sy.c[3]);
  sum[2] = (acc.sz.c[0] + acc.sz.c[2]) + (acc.sz.c[1] + acc.sz.c[3]);
  pp[0] = acc.xx; pp[1] = acc.xy; pp[2] = acc.xz; pp[3] = acc.yy; pp[4] = acc.yz; pp[5] = acc.zz;
  const bool cont = acc.continuous(prm.max_disc);
  const bool valid = prm.valid_thr >= 0 && static_cast<long long>(acc.valid_pts) >= prm.valid_thr && cont;
  dpx_cell rec;
  if (valid) {
    finish_cell(prm, n, sum, pp, acc.fx0 - acc.lx, acc.fy0 - acc.ly, acc.fz0 - acc.lz, true, rec);
  } else {
    rec.sum[0] = rec.sum[1] = rec.sum[2] = 0.f;
    rec.pp[0] = rec.pp[1] = rec.pp[2] = rec.pp[3] = rec.pp[4] = rec.pp[5] = 0.f;
    rec.normal[0] = rec.normal[1] = rec.normal[2] = 0.f;
    rec.d = 0.f;
    rec.mse = 3.402823466e+38f;
    rec.tol = 0.f;
    rec.info = 0u;
  }
  float4* dst = reinterpret_cast<float4*>(cells + static_cast<long long>(frame) * prm.C + static_cast<long long>(cr) * prm.nh + cc);
  const float4* s4 = reinterpret_cast<const float4*>(&rec);
  dst[0] = s4[0];
  dst[1] = s4[1];
  dst[2] = s4[2];
  dst[3] = s4[3];
}

// ---- organised-cloud input, LDS-staged (W % 4 == 0, 16-byte aligned cloud) ----------------------------------------
// The per-lane loads of k_planefit_cloud put 64 different cache lines under every load instruction (lane stride 12*P
// bytes).  Here a wave moves its 64-cell row segment (64 * 12 * P contiguous bytes) straight from global memory into
// a wave-private LDS buffer with the gfx950 LDS-DMA loads (global_load_lds_dwordx4: lane l delivers bytes
// [16 l, 16 l + 16) of every KB, no staging registers), and every lane reads its own cell row back (lane stride 12*P
// bytes: at most 2-way bank conflicts).  The DMA of row i + 1 is in flight while row i is summed.  Everything is
// branch-free: chunks past the end of a partial strip read a clamped address and land in the padding of the buffer.
// Rows run as rolled 4-row groups like k_planefit_depth.  Same arithmetic, same results as k_planefit_cloud.
#ifndef DPX_CLOUD_WAVES
#define DPX_CLOUD_WAVES 2
#endif

template <int P>
struct CloudStage {
  static constexpr int DWR = 3 * P;                     // dwords per cell row
  static constexpr int NCH = (DWR * 64 / 4 + 63) / 64;  // 16-byte chunks per lane and row segment
  static_assert((DWR * 2) % 4 == 0, "segments of an even number of cells end on a 16-byte boundary");

  // asynchronous: completion is observed with s_waitcnt vmcnt(0)
  static __device__ __forceinline__ void dma(const float* seg, int last_chunk_dw, int lane, float* lds) {
#pragma unroll
    for (int k = 0; k < NCH; ++k) {
      const int d = (k * 64 + lane) * 4;
      __builtin_amdgcn_global_load_lds((dpx_glb_ptr)(seg + (d < last_chunk_dw ? d : last_chunk_dw)), (dpx_lds_ptr)(lds + k * 256), 16, 0, 0);
    }
  }
  static __device__ __forceinline__ void fetch(const float* lds, int lane, float (&w)[DWR]) {
    const float* rp = lds + lane * DWR;  // 8-byte aligned (DWR even)
    int d = 0;
#pragma unroll
    for (; d + 4 <= DWR; d += 4) {
      const f32x4a8 v = *reinterpret_cast<const f32x4a8*>(rp + d);
      w[d] = v.x; w[d + 1] = v.y; w[d + 2] = v.z; w[d + 3] = v.w;
    }
    if (d + 2 <= DWR) {
      const float2 v = *reinterpret_cast<const float2*>(rp + d);
      w[d] = v.x; w[d + 1] = v.y;
    }
  }
};

// one row: wait for its DMA (the next row's may stay in flight), read this lane's cell row, start the DMA of the row
// after next into the same buffer, sum the row.  Two buffers per wave: rows alternate between them.
// s_waitcnt immediates (gfx9 encoding): 0x0F70 | n = vmcnt(n) for n < 16, 0xC07F = lgkmcnt(0).
#define DPX_CLOUD_STEP(RM, SI, IROW, HASNEXT, HASNEXT2)                                                        \
  {                                                                                                            \
    if (HASNEXT) __builtin_amdgcn_s_waitcnt(0x0F70 | CloudStage<P>::NCH);                                      \
    else __builtin_amdgcn_s_waitcnt(0x0F70);                                                                   \
    __builtin_amdgcn_wave_barrier();                                                                           \
    float* buf_ = lds + ((RM) & 1) * kWaveFloats;                                                              \
    CloudStage<P>::fetch(buf_, lane, a);                                                                       \
    __builtin_amdgcn_s_waitcnt(0xC07F); /* every lane's reads are done before the buffer is overwritten */     \
    __builtin_amdgcn_wave_barrier();                                                                           \
    if (HASNEXT2) CloudStage<P>::dma(seg0 + static_cast<long long>((IROW) + 2) * row_stride, last_dw, lane, buf_); \
    if ((SI) == 0) acc.vprev = a[3 * (P / 2) + 2]; /* k = P/2 -> (i = 0, j = P/2) */                           \
    acc.template cloud_row_rm<RM, SI>(a, (IROW) == P / 2, prm);                                                \
  }

template <int P>
__global__ __launch_bounds__(256, DPX_CLOUD_WAVES) void k_planefit_cloud_lds(Params prm, PfGroup grp, dpx_cell* __restrict__ cells) {
  static_assert(P % 2 == 0 && P >= 2, "even patch sizes only");
  constexpr int n = P * P;
  constexpr int A2 = (n / 8) * 8;
  constexpr int FULL_A = ((A2 / P) / 4) * 4, FULL_B = ((P - 1) / 4) * 4;
  constexpr int FULL = FULL_A < FULL_B ? FULL_A : FULL_B;  // rows of the rolled groups, as in k_planefit_depth
  constexpr int kWaveFloats = CloudStage<P>::NCH * 64 * 4;  // segment buffer of one wave, padded to whole chunks
  static_assert(CloudStage<P>::NCH < 16, "vmcnt immediate");
  __shared__ __align__(16) float stage_lds[4 * 2 * kWaveFloats];
  const int strips = (prm.nh + 63) >> 6;
  const int wid = threadIdx.x >> 6;
  const long long wave = static_cast<long long>(blockIdx.x) * 4 + wid;
  const int lane = threadIdx.x & 63;
  const long long per_frame = static_cast<long long>(prm.nv) * strips;
  if (wave >= per_frame * prm.batch) return;  // wave-uniform: no workgroup barrier is used below
  const int frame = static_cast<int>(wave / per_frame);
  const int rem = static_cast<int>(wave - frame * per_frame);
  const int cr = rem / strips;
  const int cc0 = (rem - cr * strips) * 64;
  const int ncell = (prm.nh - cc0) < 64 ? (prm.nh - cc0) : 64;
  const int last_dw = ncell * 3 * P - 4;  // first dword of the segment's last whole chunk
  const long long row_stride = 3ll * prm.W;
  // one launch covers the submissions of a pipeline group (PfGroup): this wave's cloud
  const int sub = __builtin_amdgcn_readfirstlane(frame / grp.per_sub);
  const float* xyz = static_cast<const float*>(grp.in[sub]);
  const float* seg0 = xyz + 3ll * (static_cast<long long>(frame - sub * grp.per_sub) * prm.N + static_cast<long long>(cr) * P * prm.W + static_cast<long long>(cc0) * P);
  float* lds = stage_lds + wid * 2 * kWaveFloats;

  CellAcc<P, false> acc;
  acc.init();
  float a[3 * P];
  CloudStage<P>::dma(seg0, last_dw, lane, lds);
  if (P > 1) CloudStage<P>::dma(seg0 + row_stride, last_dw, lane, lds + kWaveFloats);
  if (FULL >= 4) {
    DPX_CLOUD_STEP(0, 0, 0, true, 2 < P)
    DPX_CLOUD_STEP(1, 1, 1, true, 3 < P)
    DPX_CLOUD_STEP(2, 2, 2, true, 4 < P)
    DPX_CLOUD_STEP(3, 3, 3, 4 < P, 5 < P)
#pragma nounroll
    for (int g = 4; g < FULL; g += 4) {  // rows below FULL <= P - 2: both following rows exist
      DPX_CLOUD_STEP(0, -1, g + 0, true, true)
      DPX_CLOUD_STEP(1, -1, g + 1, true, true)
      DPX_CLOUD_STEP(2, -1, g + 2, true, true)
      DPX_CLOUD_STEP(3, -1, g + 3, true, true)
    }
  }
  {
    constexpr int R0 = FULL >= 4 ? FULL : 0;
    static_assert(P - R0 <= 4, "at most one partial group after the rolled loop");
    if (R0 + 0 < P) DPX_CLOUD_STEP((R0 + 0) & 3, (R0 + 0 < P ? R0 + 0 : 0), R0 + 0, R0 + 1 < P, R0 + 2 < P)
    if (R0 + 1 < P) DPX_CLOUD_STEP((R0 + 1) & 3, (R0 + 1 < P ? R0 + 1 : 0), R0 + 1, R0 + 2 < P, R0 + 3 < P)
    if (R0 + 2 < P) DPX_CLOUD_STEP((R0 + 2) & 3, (R0 + 2 < P ? R0 + 2 : 0), R0 + 2, R0 + 3 < P, R0 + 4 < P)
    if (R0 + 3 < P) DPX_CLOUD_STEP((R0 + 3) & 3, (R0 + 3 < P ? R0 + 3 : 0), R0 + 3, R0 + 4 < P, R0 + 5 < P)
  }
  if (lane >= ncell) return;  // helper lanes of a partial strip only moved data

  const int cc = cc0 + lane;
  if (A2 > 0 && n == A2) { acc.sx.fold(); acc.sy.fold(); acc.sz.fold(); }
  float sum[3], pp[6];
  sum[0] = (acc.sx.c[0] + acc.sx.c[2]) + (acc.sx.c[1] + acc.sx.c[3]);
  sum[1] = (acc.sy.c[0] + acc.sy.c[2]) + (acc.sy.c[1] + acc.sy.c[3]);
  sum[2] = (acc.sz.c[0] + acc.sz.c[2]) + (acc.sz.c[1] + acc.sz.c[3]);
  pp[0] = acc.xx; pp[1] = acc.xy; pp[2] = acc.xz; pp[3] = acc.yy; pp[4] = acc.yz; pp[5] = acc.zz;
  const bool cont = acc.continuous(prm.max_disc);
  const bool valid = prm.valid_thr >= 0 && static_cast<long long>(acc.valid_pts) >= prm.valid_thr && cont;
  dpx_cell rec;
  if (valid) {
    finish_cell(prm, n, sum, pp, acc.fx0 - acc.lx, acc.fy0 - acc.ly, acc.fz0 - acc.lz, true, rec);
  } else {
    rec.sum[0] = rec.sum[1] = rec.sum[2] = 0.f;
    rec.pp[0] = rec.pp[1] = rec.pp[2] = rec.pp[3] = rec.pp[4] = rec.pp[5] = 0.f;
    rec.normal[0] = rec.normal[1] = rec.normal[2] = 0.f;
    rec.d = 0.f;
    rec.mse = 3.402823466e+38f;
    rec.tol = 0.f;
    rec.info = 0u;
  }
  float4* dst = reinterpret_cast<float4*>(cells + static_cast<long long>(frame) * prm.C + static_cast<long long>(cr) * prm.nh + cc);
  const float4* s4 = reinterpret_cast<const float4*>(&rec);
  dst[0] = s4[0];
  dst[1] = s4[1];
  dst[2] = s4[2];
  dst[3] = s4[3];
}
#undef DPX_CLOUD_STEP

}  // namespace dpx
