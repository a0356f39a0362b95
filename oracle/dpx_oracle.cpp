// dpx_oracle.cpp — CPU oracle (TEST INFRASTRUCTURE; see dpx_oracle.h).
//
// Scalar restatement of prime-slam/deplex's depth-plane extraction, following the reference file by
// file (citations are relative to /root/reference).  No Eigen: every Eigen expression the reference
// uses is replaced by the scalar evaluation order Eigen 3.4 produces in the reference's canonical
// build (x86-64 SSE2, no FMA).  Build with -O2 -ffp-contract=off, no -march.
#include "dpx_oracle.h"

#include <algorithm>
#include <atomic>
#include <cfloat>
#include <climits>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <queue>
#include <random>
#include <set>
#include <string>
#include <thread>
#include <vector>

#include "../deplex_amd/csrc/dpx_math.h"  // only for the "devmath" flavour (orc_set_math(1))

namespace {

std::atomic<int> g_math{0};

// ---- libm switch -----------------------------------------------------------------------------
inline double m_sqrt(double x) { return std::sqrt(x); }  // IEEE in both flavours
inline double m_atan2(double y, double x) { return g_math.load(std::memory_order_relaxed) ? dpxm::datan2(y, x) : std::atan2(y, x); }
inline double m_acos(double x) { return g_math.load(std::memory_order_relaxed) ? dpxm::dacos(x) : std::acos(x); }
inline void m_sincos(double x, double* s, double* c) {
  if (g_math.load(std::memory_order_relaxed)) {
    dpxm::sincos_small(x, s, c);
  } else {
    *c = std::cos(x);
    *s = std::sin(x);
  }
}

// ---- 3x3 symmetric eigensolver (Kopp, arXiv:physics/0610206; hybrid variant) -------------------
// Restates libs/dsyev/src/dsyevc3.c:46-80 (Cardano), dsyevh3.c:57-134 (cross-product vectors with QL
// fallback), dsytrd3.h:27-102 (Householder) and dsyevq3.c:49-134 (implicit-shift QL), keeping every
// operation in the reference's order so results are bit-identical (checked against oracle/_ref).
inline double sq(double x) { return x * x; }

void cardano_eigenvalues(const double A[3][3], double w[3]) {
  // A = [a d f; d b e; f e c]
  const double a = A[0][0], b = A[1][1], c = A[2][2], d = A[0][1], e = A[1][2], f = A[0][2];
  const double de = d * e, dd = sq(d), ee = sq(e), ff = sq(f);
  const double m = a + b + c;
  const double c1 = (a * b + a * c + b * c) - (dd + ee + ff);
  const double c0 = c * dd + a * ee + b * ff - a * b * c - 2.0 * f * de;
  const double p = sq(m) - 3.0 * c1;
  const double q = m * (p - (3.0 / 2.0) * c1) - (27.0 / 2.0) * c0;
  const double sqrt_p = m_sqrt(std::fabs(p));
  double phi = 27.0 * (0.25 * sq(c1) * (p - c1) + c0 * (q + 27.0 / 4.0 * c0));
  phi = (1.0 / 3.0) * m_atan2(m_sqrt(std::fabs(phi)), q);
  double sn, cs;
  m_sincos(phi, &sn, &cs);
  const double cc = sqrt_p * cs;
  const double ss = (1.0 / 1.73205080756887729352744634151) * sqrt_p * sn;
  w[1] = (1.0 / 3.0) * (m - cc);
  w[2] = w[1] + ss;
  w[0] = w[1] + cc;
  w[1] -= ss;
}

void householder_tridiag(const double A[3][3], double Q[3][3], double dg[3], double off[2]) {
  double u[3], q[3];
  for (int i = 0; i < 3; ++i) {
    Q[i][i] = 1.0;
    for (int j = 0; j < i; ++j) Q[i][j] = Q[j][i] = 0.0;
  }
  const double h = sq(A[0][1]) + sq(A[0][2]);
  const double g = (A[0][1] > 0) ? -m_sqrt(h) : m_sqrt(h);
  off[0] = g;
  double f = g * A[0][1];
  u[1] = A[0][1] - g;
  u[2] = A[0][2];
  double omega = h - f;
  if (omega > 0.0) {
    omega = 1.0 / omega;
    double K = 0.0;
    for (int i = 1; i < 3; ++i) {
      f = A[1][i] * u[1] + A[i][2] * u[2];
      q[i] = omega * f;
      K += u[i] * f;
    }
    K *= 0.5 * sq(omega);
    for (int i = 1; i < 3; ++i) q[i] = q[i] - K * u[i];
    dg[0] = A[0][0];
    dg[1] = A[1][1] - 2.0 * q[1] * u[1];
    dg[2] = A[2][2] - 2.0 * q[2] * u[2];
    for (int j = 1; j < 3; ++j) {
      f = omega * u[j];
      for (int i = 1; i < 3; ++i) Q[i][j] = Q[i][j] - f * u[i];
    }
    off[1] = A[1][2] - q[1] * u[2] - u[1] * q[2];
  } else {
    for (int i = 0; i < 3; ++i) dg[i] = A[i][i];
    off[1] = A[1][2];
  }
}

int ql_eigensystem(const double A[3][3], double Q[3][3], double w[3]) {
  double e[3];
  householder_tridiag(A, Q, w, e);
  for (int l = 0; l < 2; ++l) {
    int iter = 0;
    for (;;) {
      int m;
      double g;
      for (m = l; m <= 1; ++m) {
        g = std::fabs(w[m]) + std::fabs(w[m + 1]);
        if (std::fabs(e[m]) + g == g) break;
      }
      if (m == l) break;
      if (iter++ >= 30) return -1;
      g = (w[l + 1] - w[l]) / (e[l] + e[l]);
      double r = m_sqrt(sq(g) + 1.0);
      if (g > 0)
        g = w[m] - w[l] + e[l] / (g + r);
      else
        g = w[m] - w[l] + e[l] / (g - r);
      double s = 1.0, c = 1.0, p = 0.0;
      for (int i = m - 1; i >= l; --i) {
        const double f = s * e[i];
        const double b = c * e[i];
        if (std::fabs(f) > std::fabs(g)) {
          c = g / f;
          r = m_sqrt(sq(c) + 1.0);
          e[i + 1] = f * r;
          s = 1.0 / r;
          c *= s;
        } else {
          s = f / g;
          r = m_sqrt(sq(s) + 1.0);
          e[i + 1] = g * r;
          c = 1.0 / r;
          s *= c;
        }
        g = w[i + 1] - p;
        r = (w[i] - g) * s + 2.0 * c * b;
        p = s * r;
        w[i + 1] = g + p;
        g = c * r - b;
        for (int k = 0; k < 3; ++k) {
          const double t = Q[k][i + 1];
          Q[k][i + 1] = s * Q[k][i] + c * t;
          Q[k][i] = c * Q[k][i] - s * t;
        }
      }
      w[l] -= p;
      e[l] = g;
      e[m] = 0.0;
    }
  }
  return 0;
}

// returns 0 analytic, 1 QL, -1 QL without convergence
int hybrid_eigensystem(const double A[3][3], double Q[3][3], double w[3]) {
  cardano_eigenvalues(A, w);
  double t = std::fabs(w[0]), u;
  if ((u = std::fabs(w[1])) > t) t = u;
  if ((u = std::fabs(w[2])) > t) t = u;
  u = (t < 1.0) ? t : sq(t);
  const double error = 256.0 * DBL_EPSILON * sq(u);

  Q[0][1] = A[0][1] * A[1][2] - A[0][2] * A[1][1];
  Q[1][1] = A[0][2] * A[0][1] - A[1][2] * A[0][0];
  Q[2][1] = sq(A[0][1]);

  Q[0][0] = Q[0][1] + A[0][2] * w[0];
  Q[1][0] = Q[1][1] + A[1][2] * w[0];
  Q[2][0] = (A[0][0] - w[0]) * (A[1][1] - w[0]) - Q[2][1];
  double norm = sq(Q[0][0]) + sq(Q[1][0]) + sq(Q[2][0]);
  if (norm <= error) return ql_eigensystem(A, Q, w) == 0 ? 1 : -1;
  norm = m_sqrt(1.0 / norm);
  for (int j = 0; j < 3; ++j) Q[j][0] = Q[j][0] * norm;

  Q[0][1] = Q[0][1] + A[0][2] * w[1];
  Q[1][1] = Q[1][1] + A[1][2] * w[1];
  Q[2][1] = (A[0][0] - w[1]) * (A[1][1] - w[1]) - Q[2][1];
  norm = sq(Q[0][1]) + sq(Q[1][1]) + sq(Q[2][1]);
  if (norm <= error) return ql_eigensystem(A, Q, w) == 0 ? 1 : -1;
  norm = m_sqrt(1.0 / norm);
  for (int j = 0; j < 3; ++j) Q[j][1] = Q[j][1] * norm;

  Q[0][2] = Q[1][0] * Q[2][1] - Q[2][0] * Q[1][1];
  Q[1][2] = Q[2][0] * Q[0][1] - Q[0][0] * Q[2][1];
  Q[2][2] = Q[0][0] * Q[1][1] - Q[1][0] * Q[0][1];
  return 0;
}

// ---- CellSegmentStat (cell_segment_stat.h:62-69, cell_segment_stat.cpp) -------------------------
struct Stat {
  float d = 0.f, score = 0.f, mse = FLT_MAX;  // default ctor cell_segment_stat.cpp:27
  int32_t n = 0;
  float sum[3] = {0, 0, 0};
  float var[3][3] = {{0, 0, 0}, {0, 0, 0}, {0, 0, 0}};
  float mean[3] = {0, 0, 0};
  float normal[3] = {0, 0, 0};
  int used_ql = 0;
};

// Eigen fixed-size-3 float reduction order: x0 + (x1 + x2)   
inline float dot3(const float a[3], const float b[3]) { return a[0] * b[0] + (a[1] * b[1] + a[2] * b[2]); }

struct Counters {
  int eig_solves = 0, ql = 0;
};

// cell_segment_stat.cpp:55-81
void fit_plane(Stat& s, Counters* cnt) {
  const float nf = static_cast<float>(s.n);
  double A[3][3], Q[3][3], w[3];
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) {
      const float cov = s.var[i][j] - (s.sum[i] * s.sum[j]) / nf;  // :56
      A[i][j] = cov;
    }
  const int rc = hybrid_eigensystem(A, Q, w);
  s.used_ql = rc != 0;
  if (cnt) {
    cnt->eig_solves++;
    cnt->ql += rc != 0;
  }
  // std::min_element / max_element: first minimum, first maximum (:67-68)
  int mn = 0, mx = 0;
  for (int i = 1; i < 3; ++i) {
    if (w[i] < w[mn]) mn = i;
    if (w[mx] < w[i]) mx = i;
  }
  float v[3];
  for (int i = 0; i < 3; ++i) v[i] = static_cast<float>(Q[i][mn]);
  float d = -dot3(s.mean, v);  // :74
  for (int i = 0; i < 3; ++i) s.normal[i] = d > 0 ? v[i] : -v[i];
  s.d = d > 0 ? d : -d;
  s.mse = static_cast<float>(w[mn] / s.n);                      // :79
  s.score = static_cast<float>(w[mx] / ((w[0] + w[1]) + w[2]));  // :80, Map<Vector3d>.sum() = (w0+w1)+w2
}

// Eigen linear-vectorised sum of a contiguous float column whose first element sits `misalign`
// floats past a 16-byte boundary (redux_impl<…, LinearVectorizedTraversal, NoUnrolling>, packet = 4).
float eigen_column_sum(const float* x, int64_t n, int64_t first_aligned) {
  const int64_t a = std::min<int64_t>(first_aligned, n);
  const int64_t A = ((n - a) / 4) * 4;
  const int64_t A2 = ((n - a) / 8) * 8;
  if (A == 0) {
    float r = x[0];
    for (int64_t i = 1; i < n; ++i) r = r + x[i];
    return r;
  }
  float p0[4], p1[4];
  for (int l = 0; l < 4; ++l) p0[l] = x[a + l];
  if (A > 4) {
    for (int l = 0; l < 4; ++l) p1[l] = x[a + 4 + l];
    for (int64_t i = a + 8; i < a + A2; i += 8)
      for (int l = 0; l < 4; ++l) {
        p0[l] = p0[l] + x[i + l];
        p1[l] = p1[l] + x[i + 4 + l];
      }
    for (int l = 0; l < 4; ++l) p0[l] = p0[l] + p1[l];
    if (A > A2)
      for (int l = 0; l < 4; ++l) p0[l] = p0[l] + x[a + A2 + l];
  }
  float r = (p0[0] + p0[2]) + (p0[1] + p0[3]);  // SSE predux
  for (int64_t i = 0; i < a; ++i) r = r + x[i];
  for (int64_t i = a + A; i < n; ++i) r = r + x[i];
  return r;
}

// CellSegmentStat(MatrixX3f const&) (cell_segment_stat.cpp:29-35).  pts: n x 3, point-major.
void stat_from_points(Stat& s, const float* pts, int64_t n, Counters* cnt) {
  s.n = static_cast<int32_t>(n);
  std::vector<float> col(n);
  for (int j = 0; j < 3; ++j) {
    for (int64_t k = 0; k < n; ++k) col[k] = pts[3 * k + j];
    // column j of a 16-byte-aligned column-major n x 3 float matrix starts j*n floats in
    const int64_t first_aligned = (4 - ((j * n) % 4)) % 4;
    s.sum[j] = eigen_column_sum(col.data(), n, first_aligned);
  }
  // points^T * points: Eigen GEMM scalar tail, acc = p_i*p_j + acc in k order for
  // n + 6 >= 20; the lazy coefficient product for smaller n (Eigen's GEMM threshold) sums the same
  // products through its own redux, which for a dynamic inner size is also left-to-right.
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) {
      float acc = 0.f;
      if (n + 6 >= 20) {
        for (int64_t k = 0; k < n; ++k) acc = pts[3 * k + i] * pts[3 * k + j] + acc;
      } else if (n > 0) {
        acc = pts[i] * pts[j];
        for (int64_t k = 1; k < n; ++k) acc = acc + pts[3 * k + i] * pts[3 * k + j];
      }
      s.var[i][j] = acc;
    }
  const float nf = static_cast<float>(s.n);
  for (int j = 0; j < 3; ++j) s.mean[j] = s.sum[j] / nf;
  fit_plane(s, cnt);
}

// operator+= (cell_segment_stat.cpp:37-43)
void stat_add(Stat& a, const Stat& b) {
  a.n += b.n;
  for (int j = 0; j < 3; ++j) a.sum[j] += b.sum[j];
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) a.var[i][j] += b.var[i][j];
  const float nf = static_cast<float>(a.n);
  for (int j = 0; j < 3; ++j) a.mean[j] = a.sum[j] / nf;
}

// ---- CellSegment (cell_segment.cpp) -----------------------------------------------------------
struct Cell {
  Stat stat;
  bool valid = false;
  bool planar = false;
  float tol = 0.f;
};

// cell_segment.cpp:62-91, one scan
bool scan_continuous(const float* pts, int64_t first, int64_t last_excl, int64_t step, float thr, int32_t max_disc) {
  float prev = pts[3 * first + 2];
  int32_t disc = 0;
  for (int64_t i = first; i < last_excl; i += step) {
    const float cur = pts[3 * i + 2];
    if (cur > 0 && std::fabs(cur - prev) < thr)
      prev = cur;
    else if (cur > 0)
      ++disc;
  }
  return disc < max_disc;
}

void build_cell(Cell& c, const float* pts, const orc_config& cfg, Counters* cnt) {
  const int32_t P = cfg.patch_size;
  const int64_t n = static_cast<int64_t>(P) * P;
  // cell_segment.cpp:23: cell_points.size() (= 3n coefficients) / min_pts_per_cell
  const long thr_signed = static_cast<long>(3 * n) / cfg.min_pts_per_cell;
  const size_t valid_thr = static_cast<size_t>(thr_signed);
  size_t valid_pts = 0;
  for (int64_t k = 0; k < n; ++k) valid_pts += pts[3 * k + 2] > 0;
  bool ok = valid_pts >= valid_thr;
  if (ok) {
    const int64_t middle = n / 2;  // cell_width*cell_height/2
    ok = scan_continuous(pts, middle, middle + P, 1, cfg.depth_discontinuity_threshold,
                         cfg.max_number_depth_discontinuity) &&
         scan_continuous(pts, P / 2, n, P, cfg.depth_discontinuity_threshold, cfg.max_number_depth_discontinuity);
  }
  c.valid = ok;
  if (!ok) return;
  stat_from_points(c.stat, pts, n, cnt);
  // hasSmallPlaneError (:99-102)
  const float mz = c.stat.mean[2];
  const float planar_thr = cfg.depth_sigma_coeff * (mz * mz) + cfg.depth_sigma_margin;
  c.planar = static_cast<double>(c.stat.mse) <= static_cast<double>(planar_thr) * static_cast<double>(planar_thr);
  // calculateMergeTolerance (:104-110), min_merge_dist = 20.0
  const float cosm = cfg.min_cos_angle_merge;
  const float sin_merge = std::sqrt(1.0f - cosm * cosm);
  const float* p0 = pts;
  const float* pl = pts + 3 * (n - 1);
  const float dx = p0[0] - pl[0], dy = p0[1] - pl[1], dz = p0[2] - pl[2];
  const float diam = std::sqrt(dx * dx + (dy * dy + dz * dz));
  const float trunc = std::min(std::max(diam * sin_merge, 20.0f), cfg.max_merge_dist);
  c.tol = trunc * trunc;
}

// ---- NormalsHistogram (normals_histogram.cpp) -----------------------------------------------------
struct Histogram {
  int32_t B = 0;
  std::vector<int32_t> hist, bins;

  void init(int32_t nbins, const std::vector<Cell>& cells) {
    B = nbins;
    hist.assign(static_cast<size_t>(B) * B, 0);
    bins.assign(cells.size(), -1);
    for (size_t i = 0; i < cells.size(); ++i) {
      if (!cells[i].planar) continue;  // initializeHistogram leaves non-planar rows zero (plane_extractor.cpp:286-291)
      const float* nrm = cells[i].stat.normal;
      // row.isZero(): all |x| <= 1e-5
      if (std::fabs(nrm[0]) <= 1e-5f && std::fabs(nrm[1]) <= 1e-5f && std::fabs(nrm[2]) <= 1e-5f) continue;
      const double nx = nrm[0], ny = nrm[1];
      const double proj = m_sqrt(nx * nx + ny * ny);
      const double polar = m_acos(static_cast<double>(-nrm[2]));
      const double azimuth = m_atan2(nx / proj, ny / proj);
      const double min_X = 0, max_X = M_PI, min_Y = -M_PI, max_Y = M_PI;
      const int32_t Xq = static_cast<int32_t>((B - 1) * (polar - min_X) / (max_X - min_X));
      int32_t Yq = 0;
      if (Xq > 0) Yq = static_cast<int32_t>((B - 1) * (azimuth - min_Y) / (max_Y - min_Y));
      const int32_t bin = Yq * B + Xq;
      bins[i] = bin;
      ++hist[bin];
    }
  }
  std::vector<int32_t> most_frequent() const {
    std::vector<int32_t> ids;
    if (hist.empty()) return ids;
    const auto it = std::max_element(hist.begin(), hist.end());
    const int32_t mx = *it;
    const int32_t bin = static_cast<int32_t>(it - hist.begin());
    if (mx > 0)
      for (int32_t i = 0; i < static_cast<int32_t>(bins.size()); ++i)
        if (bins[i] == bin) ids.push_back(i);
    return ids;
  }
  void remove(int32_t id) {
    --hist[bins[id]];
    bins[id] = -1;
  }
};

// ---- RANSAC refinement (plane_extractor.cpp:472-509, libs/rtl/include/rtl/{Plane,RANSAC}.hpp) ------------
// The RNG (std::mt19937, default seed) and std::uniform_int_distribution<int> are the standard library's own,
// exactly as the reference uses them: the draw sequence is that of the libstdc++ this file is compiled with
// (GCC 11: Lemire's multiply-shift rejection for a 32-bit engine, bits/uniform_int_dist.h).
struct PlaneModel {
  float p[4];
};

// PlaneEstimator::ComputeModel (Plane.hpp:13-44): three points in ascending index order, all float32
PlaneModel ransac_model(const float* P0, const float* P1, const float* P2) {
  const float x0 = P0[0], x1 = P1[0], x2 = P2[0];
  const float y0 = P0[1], y1 = P1[1], y2 = P2[1];
  const float z0 = P0[2], z1 = P1[2], z2 = P2[2];
  const float den = x0 * y1 - x1 * y0 - x0 * y2 + x2 * y0 + x1 * y2 - x2 * y1;
  const float a = (z0 * (y1 - y2)) / den - (z1 * (y0 - y2)) / den + (z2 * (y0 - y1)) / den;
  const float b = (z1 * (x0 - x2)) / den - (z0 * (x1 - x2)) / den - (z2 * (x0 - x1)) / den;
  const float d = (z2 * (x0 * y1 - x1 * y0)) / den - (z1 * (x0 * y2 - x2 * y0)) / den + (z0 * (x1 * y2 - x2 * y1)) / den;
  const float c = -1.0f;
  const float l = static_cast<float>(std::sqrt(static_cast<double>(a * a + b * b + c * c)));
  PlaneModel m;
  m.p[0] = a / l;
  m.p[1] = b / l;
  m.p[2] = c / l;
  m.p[3] = d / l;
  return m;
}

// PlaneEstimator::ComputeError (Plane.hpp:46-49): float expression widened to double
inline double ransac_error(const PlaneModel& m, const float* pt) {
  return static_cast<double>(m.p[0] * pt[0] + m.p[1] * pt[1] + m.p[2] * pt[2] + m.p[3]);
}

struct RefineCounters {
  int labels = 0;
  long long iterations = 0;
  long long zeroed = 0;
};

void refine_labels(const orc_config& cfg, const float* xyz, int64_t N, int32_t* labels, RefineCounters* rc) {
  int32_t maxl = 0;
  for (int64_t i = 0; i < N; ++i) maxl = std::max(maxl, labels[i]);  // labels->maxCoeff() (>= 0 here)
  std::vector<std::vector<int32_t>> idx(static_cast<size_t>(maxl));
  for (int64_t i = 0; i < N; ++i)
    if (labels[i] != 0) idx[labels[i] - 1].push_back(static_cast<int32_t>(i));
  std::mt19937 gen;  // one generator per process() call, default seed 5489 (RANSAC.hpp:111)
  const int max_iter = cfg.ransac_max_iterations;
  const double thr = cfg.ransac_threshold;
  const double ratio = cfg.ransac_inliers_ratio;
  for (int32_t l = 0; l < maxl; ++l) {
    const int n = static_cast<int>(idx[l].size());
    if (n == 0) continue;
    if (rc) rc->labels++;
    std::vector<float> pts(static_cast<size_t>(3) * n);
    for (int r = 0; r < n; ++r) std::memcpy(&pts[3 * r], &xyz[3 * static_cast<int64_t>(idx[l][r])], 3 * sizeof(float));
    // FindBest (RANSAC.hpp:25-51)
    std::uniform_int_distribution<int> uni(0, n - 1);
    PlaneModel best{{0.f, 0.f, 0.f, 0.f}};
    double bestloss = HUGE_VAL;
    int iteration = 0;
    for (;;) {
      // IsContinued(iteration, N - bestloss, N): the first call converts -inf to int (INT_MIN on x86-64)
      const int inliers_count = std::isinf(bestloss) ? INT_MIN : static_cast<int>(n - bestloss);
      if (!(iteration < max_iter && inliers_count < ratio * n)) break;
      iteration++;
      std::set<int> samples;
      while (static_cast<int>(samples.size()) < 3) samples.insert(uni(gen));
      auto it = samples.begin();
      const int i0 = *it++, i1 = *it++, i2 = *it++;
      const PlaneModel m = ransac_model(&pts[3 * i0], &pts[3 * i1], &pts[3 * i2]);
      double loss = 0;
      for (int r = 0; r < n; ++r) loss += (std::fabs(ransac_error(m, &pts[3 * r])) >= thr);
      if (loss < bestloss) {
        best = m;
        bestloss = loss;
      }
    }
    if (rc) rc->iterations += iteration;
    // FindInliers (:53-62) + the relabel loop (plane_extractor.cpp:500-507): stops after the last inlier
    std::vector<int> inliers;
    for (int r = 0; r < n; ++r)
      if (std::fabs(ransac_error(best, &pts[3 * r])) < thr) inliers.push_back(r);
    size_t cur = 0;
    for (int r = 0; r < n && cur < inliers.size(); ++r) {
      if (inliers[cur] != r) {
        labels[idx[l][r]] = 0;
        if (rc) rc->zeroed++;
      } else {
        ++cur;
      }
    }
  }
}

// ---- PlaneExtractor::Impl (plane_extractor.cpp) -------------------------------------------------
struct Segment {
  Stat stat;
  int32_t n_cells = 0;
};

struct Extractor {
  orc_config cfg;
  int32_t H, W, nh, nv;
  std::vector<std::vector<int32_t>> nbr;
  std::vector<int32_t> labels_map;  // nv x nh

  // plane_extractor.cpp:153-176
  int construct(const orc_config& c, int32_t h, int32_t w, std::string* err) {
    cfg = c;
    H = h;
    W = w;
    nh = w / std::max(c.patch_size, 1);
    nv = h / std::max(c.patch_size, 1);
    cfg.patch_size = std::min(cfg.patch_size, std::min(h, w));
    if (c.patch_size == 0) {
      if (err) *err = "Error! Invalid config parameter: patchSize(" + std::to_string(c.patch_size) + "). patchSize has to be positive.";
      return -1;
    }
    if (c.patch_size < 0 || c.min_pts_per_cell == 0 || c.histogram_bins_per_coord < 1 || h < 0 || w < 0) {
      if (err) *err = "configuration undefined in the reference (negative patchSize, minPtsPerCell==0 or histogramBinsPerCoord<1)";
      return -3;
    }
    labels_map.assign(static_cast<size_t>(nh) * nv, 0);
    nbr.assign(static_cast<size_t>(nh) * nv, {});
    for (int32_t i = 0; i < nh * nv; ++i) {
      const int32_t x = i / nh, y = i % nh;
      if (x >= 1) nbr[i].push_back(i - nh);
      if (x + 1 < nv) nbr[i].push_back(i + nh);
      if (y >= 1) nbr[i].push_back(i - 1);
      if (y + 1 < nh) nbr[i].push_back(i + 1);
    }
    return 0;
  }

  // growSeed (:349-392)
  std::vector<int32_t> grow(int32_t seed, const std::vector<char>& unassigned, const std::vector<Cell>& cells) const {
    std::vector<char> active(unassigned.size(), 0);
    std::vector<int32_t> out;
    if (!unassigned[seed]) return out;
    std::queue<int32_t> q;
    q.push(seed);
    active[seed] = 1;
    out.push_back(seed);
    while (!q.empty()) {
      const int32_t cur = q.front();
      q.pop();
      const double d_cur = cells[cur].stat.d;
      const float* n_cur = cells[cur].stat.normal;
      for (int32_t nb : nbr[cur]) {
        if (!unassigned[nb] || active[nb]) continue;
        const double cos_angle = dot3(n_cur, cells[nb].stat.normal);
        const double t = static_cast<double>(dot3(n_cur, cells[nb].stat.mean)) + d_cur;
        const double merge_dist = t * t;
        if (cos_angle >= cfg.min_cos_angle_merge && merge_dist <= cells[nb].tol) {
          active[nb] = 1;
          out.push_back(nb);
          q.push(nb);
        }
      }
    }
    return out;
  }

  int process(const float* xyz, int64_t n_points, int32_t* labels, orc_cell* cells_out, orc_plane* planes_out,
              int32_t planes_cap, int32_t* cell_labels_out, orc_stats* st, std::string* err) {
    if (n_points != static_cast<int64_t>(W) * H) {  // :188-194
      if (err)
        *err = "Error! Number of points doesn't match image shape: " + std::to_string(n_points) + " != " + std::to_string(H) + " x " + std::to_string(W);
      return -2;
    }
    Counters cnt;
    const int32_t P = cfg.patch_size;
    const int32_t C = nh * nv;
    const int64_t n = static_cast<int64_t>(P) * P;
    // 1. CellGrid: re-tile (cell_grid.cpp:69-83; row stride nh*P) and build every cell (:37-43)
    std::vector<Cell> cells(C);
    {
      std::vector<float> tile(static_cast<size_t>(3 * n));
      const int64_t stride_w = static_cast<int64_t>(nh) * P;
      for (int32_t c = 0; c < C; ++c) {
        const int64_t base = static_cast<int64_t>(c / nh) * stride_w * P + (static_cast<int64_t>(c) * P) % stride_w;
        for (int32_t i = 0; i < P; ++i)
          std::memcpy(&tile[static_cast<size_t>(3) * i * P], &xyz[3 * (i * stride_w + base)], sizeof(float) * 3 * P);
        build_cell(cells[c], tile.data(), cfg, &cnt);
      }
    }
    // 2. histogram (:285-295)
    Histogram hist;
    hist.init(cfg.histogram_bins_per_coord, cells);
    // 3. createPlaneSegments (:297-347)
    std::vector<Segment> segs;
    std::vector<char> unassigned(C);
    int32_t remaining = 0;
    for (int32_t c = 0; c < C; ++c) {
      unassigned[c] = cells[c].planar;
      remaining += cells[c].planar;
    }
    orc_stats s{};
    s.nh = nh;
    s.nv = nv;
    s.planar_cells = remaining;
    while (remaining > 0) {
      const std::vector<int32_t> cand = hist.most_frequent();
      if (cand.size() < static_cast<size_t>(cfg.min_region_growing_candidate_size)) break;
      int32_t seed = -1;
      double min_mse = INT_MAX;
      for (int32_t c : cand)
        if (cells[c].stat.mse < min_mse) {
          seed = c;
          min_mse = cells[c].stat.mse;
        }
      if (seed < 0) seed = cand[0];  // reference reads an uninitialised int here; unreachable for finite data
      Stat acc = cells[seed].stat;  // copy of the seed: the seed is counted twice (:318,322-323)
      const int32_t seed_bin = hist.bins[seed];
      const std::vector<int32_t> grown = grow(seed, unassigned, cells);
      for (int32_t v : grown) {
        stat_add(acc, cells[v].stat);
        hist.remove(v);
        unassigned[v] = 0;
        --remaining;
      }
      if (const char* tp = std::getenv("ORC_GROW_TRACE")) {  // analysis aid (tools/grow_trace.py): bin, seed, size per grow
        if (FILE* tf = std::fopen(tp, "a")) {
          std::fprintf(tf, "%d %d %zu\n", seed_bin, seed, grown.size());
          std::fclose(tf);
        }
      }
      s.grows++;
      s.activations += static_cast<int32_t>(grown.size());
      s.max_grow = std::max<int32_t>(s.max_grow, static_cast<int32_t>(grown.size()));
      if (grown.size() < static_cast<size_t>(cfg.min_region_growing_cells_activated)) continue;
      fit_plane(acc, &cnt);
      if (acc.score > cfg.min_region_planarity_score) {
        Segment sg;
        sg.stat = acc;
        sg.n_cells = static_cast<int32_t>(grown.size());
        segs.push_back(sg);
        const int32_t id = static_cast<int32_t>(segs.size());
        for (int32_t v : grown) labels_map[v] = id;
      }
    }
    s.segments = static_cast<int32_t>(segs.size());
    if (cell_labels_out) std::memcpy(cell_labels_out, labels_map.data(), sizeof(int32_t) * labels_map.size());
    const int64_t N = static_cast<int64_t>(W) * H;
    std::vector<int32_t> merge(segs.size());
    if (segs.empty()) {
      std::fill(labels, labels + N, 0);  // :230-232 (labels_map_ is still all zero)
    } else {
      // 5. findMergedLabels (:394-426) with getConnectedComponents (:430-453)
      const size_t np = segs.size();
      std::vector<std::vector<char>> assoc(np, std::vector<char>(np, 0));
      for (int32_t r = 0; r < nv - 1; ++r)
        for (int32_t c = 0; c < nh - 1; ++c) {
          const int32_t id = labels_map[static_cast<size_t>(r) * nh + c];
          if (id > 0) {
            const int32_t right = labels_map[static_cast<size_t>(r) * nh + c + 1];
            const int32_t down = labels_map[static_cast<size_t>(r + 1) * nh + c];
            if (right > 0 && id != right) assoc[id - 1][right - 1] = 1;
            if (down > 0 && id != down) assoc[id - 1][down - 1] = 1;
          }
        }
      for (size_t r = 0; r < np; ++r)
        for (size_t c = 0; c < np; ++c) assoc[r][c] = assoc[r][c] || assoc[c][r];
      for (size_t i = 0; i < np; ++i) merge[i] = static_cast<int32_t>(i);
      for (size_t row = 0; row < np; ++row) {
        const int32_t pid = merge[row];
        bool expanded = false;
        for (size_t col = row + 1; col < np; ++col) {
          if (!assoc[row][col]) continue;
          const double cos_angle = dot3(segs[pid].stat.normal, segs[col].stat.normal);
          const float tf = dot3(segs[pid].stat.normal, segs[col].stat.mean) + segs[pid].stat.d;  // float add (:410-411)
          const double distance = static_cast<double>(tf) * static_cast<double>(tf);
          if (cos_angle > cfg.min_cos_angle_merge && distance < cfg.max_merge_dist) {
            stat_add(segs[pid].stat, segs[col].stat);
            merge[col] = pid;
            expanded = true;
            s.merges++;
          }
        }
        if (expanded) fit_plane(segs[pid].stat, &cnt);
      }
      // 6. toImageLabels (:455-470).  Pixels outside the nv*P x nh*P grid are undefined behaviour in the
      // reference (out-of-bounds labels_map_ read); defined as 0 here.
      for (int32_t row = 0; row < H; ++row)
        for (int32_t col = 0; col < W; ++col) {
          const int32_t cr = row / P, cc = col / P;
          int32_t lab = 0;
          if (cr < nv && cc < nh) {
            const int32_t L = labels_map[static_cast<size_t>(cr) * nh + cc];
            lab = L == 0 ? 0 : merge[L - 1] + 1;
          }
          labels[static_cast<int64_t>(row) * W + col] = lab;
        }
    }
    // 7. refineLabels (:265-279), only when there are segments (the early return at :230-232 skips it)
    if (cfg.ransac_refinement && !segs.empty()) refine_labels(cfg, xyz, N, labels, nullptr);
    // 8. cleanArtifacts (:281)
    std::fill(labels_map.begin(), labels_map.end(), 0);

    s.eig_solves = cnt.eig_solves;
    s.ql_fallbacks = cnt.ql;
    if (st) *st = s;
    if (cells_out) {
      for (int32_t c = 0; c < C; ++c) {
        orc_cell& o = cells_out[c];
        std::memset(&o, 0, sizeof(o));
        const Cell& ce = cells[c];
        o.valid = ce.valid;
        o.planar = ce.planar;
        o.bin = -1;
        o.mse = ce.stat.mse;
        o.score = ce.stat.score;
        if (ce.valid) {
          for (int j = 0; j < 3; ++j) {
            o.sum[j] = ce.stat.sum[j];
            o.mean[j] = ce.stat.mean[j];
            o.normal[j] = ce.stat.normal[j];
          }
          o.pp[0] = ce.stat.var[0][0];
          o.pp[1] = ce.stat.var[0][1];
          o.pp[2] = ce.stat.var[0][2];
          o.pp[3] = ce.stat.var[1][1];
          o.pp[4] = ce.stat.var[1][2];
          o.pp[5] = ce.stat.var[2][2];
          o.d = ce.stat.d;
          o.tol = ce.tol;
          o.used_ql = ce.stat.used_ql;
        }
      }
      // bins as initially assigned (before any removal)
      Histogram h0;
      h0.init(cfg.histogram_bins_per_coord, cells);
      for (int32_t c = 0; c < C; ++c) cells_out[c].bin = h0.bins[c];
    }
    if (planes_out) {
      for (size_t i = 0; i < segs.size() && static_cast<int32_t>(i) < planes_cap; ++i) {
        orc_plane& o = planes_out[i];
        const Stat& t = segs[i].stat;
        for (int j = 0; j < 3; ++j) {
          o.sum[j] = t.sum[j];
          o.mean[j] = t.mean[j];
          o.normal[j] = t.normal[j];
        }
        o.pp[0] = t.var[0][0];
        o.pp[1] = t.var[0][1];
        o.pp[2] = t.var[0][2];
        o.pp[3] = t.var[1][1];
        o.pp[4] = t.var[1][2];
        o.pp[5] = t.var[2][2];
        o.d = t.d;
        o.mse = t.mse;
        o.score = t.score;
        o.n_pts = t.n;
        o.n_cells = segs[i].n_cells;
        o.merge_label = merge[i];
      }
    }
    return 0;
  }
};

void copy_err(const std::string& e, char* err, int32_t cap) {
  if (!err || cap <= 0) return;
  std::snprintf(err, static_cast<size_t>(cap), "%s", e.c_str());
}

}  // namespace

extern "C" {

void orc_set_math(int flavour) { g_math.store(flavour ? 1 : 0); }
int orc_get_math(void) { return g_math.load(); }

void orc_default_config(orc_config* c) {
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

void orc_backproject(const uint16_t* depth, int32_t height, int32_t width, float fx, float fy, float cx, float cy,
                     float* xyz) {
  // depth_image.cpp:64-73: z = float(u16); x = (col - cx) * z / fx; y = (row - cy) * z / fy (float32)
  for (int32_t r = 0; r < height; ++r)
    for (int32_t c = 0; c < width; ++c) {
      const int64_t i = static_cast<int64_t>(r) * width + c;
      const float z = static_cast<float>(depth[i]);
      xyz[3 * i + 2] = z;
      xyz[3 * i + 0] = ((static_cast<float>(c) - cx) * z) / fx;
      xyz[3 * i + 1] = ((static_cast<float>(r) - cy) * z) / fy;
    }
}

int orc_process(const orc_config* cfg, int32_t height, int32_t width, const float* xyz, int64_t n_points,
                int32_t* labels, orc_cell* cells_out, orc_plane* planes_out, int32_t planes_cap,
                int32_t* cell_labels_out, orc_stats* stats, char* err, int32_t err_cap) {
  Extractor ex;
  std::string e;
  int rc = ex.construct(*cfg, height, width, &e);
  if (rc == 0) rc = ex.process(xyz, n_points, labels, cells_out, planes_out, planes_cap, cell_labels_out, stats, &e);
  if (rc != 0) copy_err(e, err, err_cap);
  return rc;
}

int orc_process_depth(const orc_config* cfg, int32_t height, int32_t width, const uint16_t* depth, float fx, float fy,
                      float cx, float cy, int32_t* labels, orc_cell* cells_out, orc_plane* planes_out,
                      int32_t planes_cap, int32_t* cell_labels_out, orc_stats* stats, char* err, int32_t err_cap) {
  std::vector<float> xyz(static_cast<size_t>(3) * height * width);
  orc_backproject(depth, height, width, fx, fy, cx, cy, xyz.data());
  return orc_process(cfg, height, width, xyz.data(), static_cast<int64_t>(height) * width, labels, cells_out,
                     planes_out, planes_cap, cell_labels_out, stats, err, err_cap);
}

int orc_process_depth_batch(const orc_config* cfg, int32_t height, int32_t width, const uint16_t* depth, int32_t batch,
                            float fx, float fy, float cx, float cy, int32_t* labels, int32_t threads) {
  const int64_t N = static_cast<int64_t>(height) * width;
  std::atomic<int32_t> next{0};
  std::atomic<int> status{0};
  auto work = [&]() {
    Extractor ex;
    if (ex.construct(*cfg, height, width, nullptr) != 0) {
      status = -1;
      return;
    }
    std::vector<float> xyz(static_cast<size_t>(3) * N);
    for (;;) {
      const int32_t f = next.fetch_add(1);
      if (f >= batch) break;
      orc_backproject(depth + f * N, height, width, fx, fy, cx, cy, xyz.data());
      if (ex.process(xyz.data(), N, labels + f * N, nullptr, nullptr, 0, nullptr, nullptr, nullptr) != 0) status = -2;
    }
  };
  threads = std::max(1, threads);
  if (threads == 1) {
    work();
  } else {
    std::vector<std::thread> pool;
    for (int t = 0; t < threads; ++t) pool.emplace_back(work);
    for (auto& t : pool) t.join();
  }
  return status.load();
}

int orc_eig3(const double* A, double* Q, double* w) {
  double a[3][3], q[3][3];
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) {
      a[i][j] = A[3 * i + j];
      q[i][j] = 0.0;
    }
  const int rc = hybrid_eigensystem(a, q, w);
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) Q[3 * i + j] = q[i][j];
  return rc;
}

float orc_points_mse(const float* xyz, int64_t n) {
  Stat s;
  stat_from_points(s, xyz, n, nullptr);
  return s.mse;
}

}  // extern "C"
