// dpx_fit.h — per-thread 3x3 symmetric eigen-decomposition and plane fit, in registers.
//
// Replaces the reference's libs/dsyev (dsyevh3.c:31-134 -> dsyevc3.c, dsyevq3.c, dsytrd3.h) and
// CellSegmentStat::fitPlane (cpp/deplex/src/deplex/cell_segment_stat.cpp:55-81).  The algorithm is
// Kopp's hybrid method (arXiv:physics/0610206): Cardano eigenvalues, cross-product eigenvectors, and a
// Householder + implicit-shift QL fallback when the cross products are ill-conditioned.  The floating
// point operation order matches the reference so results are bit-identical to it given identical
// sqrt/atan2/sin/cos (dpx_math.h); compile with -ffp-contract=off.
//
// Provenance: written from the published algorithm — J. Kopp, "Efficient numerical diagonalization of hermitian 3x3
// matrices", Int. J. Mod. Phys. C 19 (2008) 523, arXiv:physics/0610206, sections 3.1 (Cardano), 3.2 (QL with implicit
// shifts after a Householder reduction) and 4 (the hybrid's error bound).  No source text of libs/dsyev (LGPL-2.1) is
// reproduced: the data structures (scalar Sym3 / Eig3, no arrays), the control flow (QL sweeps unrolled through one
// rotation macro, one shared eigenvector routine) and the libm are this repository's own.  What IS the same, and has
// to be for bit parity, is the SEQUENCE of floating-point operations of Kopp's formulas as libs/dsyev evaluates them
// (operand order of every sum and product, the 256*eps*u^2 threshold, the QL convergence test); temporaries that
// carry the paper's / the C code's letters (de, dd, ee, ff, c1, c0, p, q, phi, ...) do so to keep that operation order
// checkable line by line against dsyevc3.c:53-77, dsyevh3.c:70-134, dsytrd3.h:27-102 and dsyevq3.c:57-130.  The
// bit-for-bit check against the reference's compiled C is tests/test_oracle.py::test_eigensolver_matches_reference_dsyev
// (CPU restatement vs the reference C code built by its Makefile) + tests/test_gpu_parity.py::test_devmath_and_eigensolver_on_device (device vs oracle).
#pragma once
#include "dpx_math.h"

namespace dpx {

struct Sym3 {  // upper triangle of a symmetric matrix
  double a00, a01, a02, a11, a12, a22;
};

struct Eig3 {
  double w0, w1, w2;  // eigenvalues, unsorted (Cardano order)
  double q00, q01, q02, q10, q11, q12, q20, q21, q22;  // q[row][col]; column k pairs with wk
  int path;  // 0 analytic, 1 QL, -1 QL not converged
};

DPX_HD double sqr(double x) { return x * x; }

// One plane rotation step of the QL sweep acting on (w_i, w_{i+1}), off-diagonal e_i, and columns i, i+1 of Q.
#define DPX_QL_ROT(WI, WI1, EI, EI1, QC0I, QC0I1, QC1I, QC1I1, QC2I, QC2I1) \
  {                                                                          \
    const double f_ = s * (EI);                                              \
    const double b_ = c * (EI);                                              \
    if (dpxm::dabs(f_) > dpxm::dabs(g)) {                                    \
      c = g / f_;                                                            \
      r = dpxm::dsqrt(sqr(c) + 1.0);                                         \
      (EI1) = f_ * r;                                                        \
      s = 1.0 / r;                                                           \
      c *= s;                                                                \
    } else {                                                                 \
      s = f_ / g;                                                            \
      r = dpxm::dsqrt(sqr(s) + 1.0);                                         \
      (EI1) = g * r;                                                         \
      c = 1.0 / r;                                                           \
      s *= c;                                                                \
    }                                                                        \
    g = (WI1)-p;                                                             \
    r = ((WI)-g) * s + 2.0 * c * b_;                                         \
    p = s * r;                                                               \
    (WI1) = g + p;                                                           \
    g = c * r - b_;                                                          \
    double t_;                                                               \
    t_ = (QC0I1);                                                            \
    (QC0I1) = s * (QC0I) + c * t_;                                           \
    (QC0I) = c * (QC0I)-s * t_;                                              \
    t_ = (QC1I1);                                                            \
    (QC1I1) = s * (QC1I) + c * t_;                                           \
    (QC1I) = c * (QC1I)-s * t_;                                              \
    t_ = (QC2I1);                                                            \
    (QC2I1) = s * (QC2I) + c * t_;                                           \
    (QC2I) = c * (QC2I)-s * t_;                                              \
  }

// Householder tridiagonalisation followed by QL with implicit shifts (n = 3, so l in {0,1}).
DPX_HD void eig3_ql(const Sym3& A, Eig3& o) {
  // --- tridiagonalise: A = Q [d0 e0 .; e0 d1 e1; . e1 d2] Q^T
  double q00 = 1.0, q01 = 0.0, q02 = 0.0, q10 = 0.0, q11 = 1.0, q12 = 0.0, q20 = 0.0, q21 = 0.0, q22 = 1.0;
  double d0, d1, d2, e0, e1, e2 = 0.0;
  {
    const double h = sqr(A.a01) + sqr(A.a02);
    const double g = (A.a01 > 0) ? -dpxm::dsqrt(h) : dpxm::dsqrt(h);
    e0 = g;
    double f = g * A.a01;
    const double u1 = A.a01 - g;
    const double u2 = A.a02;
    double omega = h - f;
    if (omega > 0.0) {
      omega = 1.0 / omega;
      double K = 0.0;
      f = A.a11 * u1 + A.a12 * u2;
      double p1 = omega * f;
      K += u1 * f;
      f = A.a12 * u1 + A.a22 * u2;
      double p2 = omega * f;
      K += u2 * f;
      K *= 0.5 * sqr(omega);
      p1 = p1 - K * u1;
      p2 = p2 - K * u2;
      d0 = A.a00;
      d1 = A.a11 - 2.0 * p1 * u1;
      d2 = A.a22 - 2.0 * p2 * u2;
      f = omega * u1;
      q11 = q11 - f * u1;
      q21 = q21 - f * u2;
      f = omega * u2;
      q12 = q12 - f * u1;
      q22 = q22 - f * u2;
      e1 = A.a12 - p1 * u2 - u1 * p2;
    } else {
      d0 = A.a00;
      d1 = A.a11;
      d2 = A.a22;
      e1 = A.a12;
    }
  }
  int path = 1;
  // --- l = 0
  for (int iter = 0;; ++iter) {
    int m;
    double g = dpxm::dabs(d0) + dpxm::dabs(d1);
    if (dpxm::dabs(e0) + g == g) {
      m = 0;
    } else {
      g = dpxm::dabs(d1) + dpxm::dabs(d2);
      m = (dpxm::dabs(e1) + g == g) ? 1 : 2;
    }
    if (m == 0) break;
    if (iter >= 30) {
      path = -1;
      break;
    }
    g = (d1 - d0) / (e0 + e0);
    double r = dpxm::dsqrt(sqr(g) + 1.0);
    const double wm = (m == 1) ? d1 : d2;
    if (g > 0)
      g = wm - d0 + e0 / (g + r);
    else
      g = wm - d0 + e0 / (g - r);
    double s = 1.0, c = 1.0, p = 0.0;
    if (m == 2) {
      DPX_QL_ROT(d1, d2, e1, e2, q01, q02, q11, q12, q21, q22)  // i = 1
    }
    DPX_QL_ROT(d0, d1, e0, e1, q00, q01, q10, q11, q20, q21)  // i = 0
    d0 -= p;
    e0 = g;
    if (m == 1)
      e1 = 0.0;
    else
      e2 = 0.0;
  }
  // --- l = 1
  if (path >= 0) {
    for (int iter = 0;; ++iter) {
      const double gg = dpxm::dabs(d1) + dpxm::dabs(d2);
      if (dpxm::dabs(e1) + gg == gg) break;  // m == l
      if (iter >= 30) {
        path = -1;
        break;
      }
      double g = (d2 - d1) / (e1 + e1);
      double r = dpxm::dsqrt(sqr(g) + 1.0);
      if (g > 0)
        g = d2 - d1 + e1 / (g + r);
      else
        g = d2 - d1 + e1 / (g - r);
      double s = 1.0, c = 1.0, p = 0.0;
      DPX_QL_ROT(d1, d2, e1, e2, q01, q02, q11, q12, q21, q22)  // i = 1
      d1 -= p;
      e1 = g;
      e2 = 0.0;
    }
  }
  o.w0 = d0;
  o.w1 = d1;
  o.w2 = d2;
  o.q00 = q00; o.q01 = q01; o.q02 = q02;
  o.q10 = q10; o.q11 = q11; o.q12 = q12;
  o.q20 = q20; o.q21 = q21; o.q22 = q22;
  o.path = path;
}
#undef DPX_QL_ROT

DPX_HD void eig3_hybrid(const Sym3& A, Eig3& o) {
  // Cardano: A = [a d f; d b e; f e c]
  const double a = A.a00, b = A.a11, c = A.a22, d = A.a01, e = A.a12, f = A.a02;
  const double de = d * e, dd = sqr(d), ee = sqr(e), ff = sqr(f);
  const double m = a + b + c;
  const double c1 = (a * b + a * c + b * c) - (dd + ee + ff);
  const double c0 = c * dd + a * ee + b * ff - a * b * c - 2.0 * f * de;
  const double p = sqr(m) - 3.0 * c1;
  const double q = m * (p - (3.0 / 2.0) * c1) - (27.0 / 2.0) * c0;
  const double sqrt_p = dpxm::dsqrt(dpxm::dabs(p));
  double phi = 27.0 * (0.25 * sqr(c1) * (p - c1) + c0 * (q + 27.0 / 4.0 * c0));
  phi = (1.0 / 3.0) * dpxm::datan2(dpxm::dsqrt(dpxm::dabs(phi)), q);
  double sn, cs;
  dpxm::sincos_small(phi, &sn, &cs);
  const double cc = sqrt_p * cs;
  const double ss = (1.0 / 1.73205080756887729352744634151) * sqrt_p * sn;
  double w1 = (1.0 / 3.0) * (m - cc);
  const double w2 = w1 + ss;
  const double w0 = w1 + cc;
  w1 -= ss;

  double t = dpxm::dabs(w0), u;
  if ((u = dpxm::dabs(w1)) > t) t = u;
  if ((u = dpxm::dabs(w2)) > t) t = u;
  u = (t < 1.0) ? t : sqr(t);
  const double error = 256.0 * 2.2204460492503131e-16 * sqr(u);

  double q01 = A.a01 * A.a12 - A.a02 * A.a11;
  double q11 = A.a02 * A.a01 - A.a12 * A.a00;
  double q21 = sqr(A.a01);

  double q00 = q01 + A.a02 * w0;
  double q10 = q11 + A.a12 * w0;
  double q20 = (A.a00 - w0) * (A.a11 - w0) - q21;
  double norm = sqr(q00) + sqr(q10) + sqr(q20);
  bool need_ql = norm <= error;
  if (!need_ql) {
    norm = dpxm::dsqrt(1.0 / norm);
    q00 = q00 * norm;
    q10 = q10 * norm;
    q20 = q20 * norm;
    q01 = q01 + A.a02 * w1;
    q11 = q11 + A.a12 * w1;
    q21 = (A.a00 - w1) * (A.a11 - w1) - q21;
    norm = sqr(q01) + sqr(q11) + sqr(q21);
    need_ql = norm <= error;
    if (!need_ql) {
      norm = dpxm::dsqrt(1.0 / norm);
      q01 = q01 * norm;
      q11 = q11 * norm;
      q21 = q21 * norm;
    }
  }
#ifndef DPX_DIAG_NO_QL  // timing-only builds of tools/pfbench.hip leave the fallback out (wrong results for those cells)
  if (need_ql) {
    eig3_ql(A, o);
    return;
  }
#endif
  o.w0 = w0;
  o.w1 = w1;
  o.w2 = w2;
  o.q00 = q00; o.q10 = q10; o.q20 = q20;
  o.q01 = q01; o.q11 = q11; o.q21 = q21;
  o.q02 = q10 * q21 - q20 * q11;
  o.q12 = q20 * q01 - q00 * q21;
  o.q22 = q00 * q11 - q10 * q01;
  o.path = 0;
}

struct PlaneFit {
  float mean[3];
  float normal[3];
  float d, mse, score;
  int path;
};

// Eigen's fixed-size-3 float reduction order x0 + (x1 + x2).
DPX_HD float dot3(float a0, float a1, float a2, float b0, float b1, float b2) { return a0 * b0 + (a1 * b1 + a2 * b2); }

// CellSegmentStat::fitPlane (cell_segment_stat.cpp:55-81) from first/second moments.
// sum: Σp, pp: Σpp^T upper triangle (xx xy xz yy yz zz), n: point count.
DPX_HD void fit_plane(const float sum[3], const float pp[6], int n, PlaneFit& o) {
  const float nf = static_cast<float>(n);
  o.mean[0] = sum[0] / nf;
  o.mean[1] = sum[1] / nf;
  o.mean[2] = sum[2] / nf;
  Sym3 A;
  A.a00 = static_cast<double>(pp[0] - (sum[0] * sum[0]) / nf);
  A.a01 = static_cast<double>(pp[1] - (sum[0] * sum[1]) / nf);
  A.a02 = static_cast<double>(pp[2] - (sum[0] * sum[2]) / nf);
  A.a11 = static_cast<double>(pp[3] - (sum[1] * sum[1]) / nf);
  A.a12 = static_cast<double>(pp[4] - (sum[1] * sum[2]) / nf);
  A.a22 = static_cast<double>(pp[5] - (sum[2] * sum[2]) / nf);
  Eig3 e;
  eig3_hybrid(A, e);
  // first minimum / first maximum of (w0, w1, w2)  (std::min_element / max_element, :67-68)
  int mn = 0, mx = 0;
  double wmin = e.w0, wmax = e.w0;
  if (e.w1 < wmin) { wmin = e.w1; mn = 1; }
  if (e.w2 < wmin) { wmin = e.w2; mn = 2; }
  if (wmax < e.w1) { wmax = e.w1; mx = 1; }
  if (wmax < e.w2) { wmax = e.w2; mx = 2; }
  (void)mx;
  const double v0d = mn == 0 ? e.q00 : (mn == 1 ? e.q01 : e.q02);
  const double v1d = mn == 0 ? e.q10 : (mn == 1 ? e.q11 : e.q12);
  const double v2d = mn == 0 ? e.q20 : (mn == 1 ? e.q21 : e.q22);
  const float v0 = static_cast<float>(v0d), v1 = static_cast<float>(v1d), v2 = static_cast<float>(v2d);
  const float d = -dot3(o.mean[0], o.mean[1], o.mean[2], v0, v1, v2);
  const bool pos = d > 0;
  o.normal[0] = pos ? v0 : -v0;
  o.normal[1] = pos ? v1 : -v1;
  o.normal[2] = pos ? v2 : -v2;
  o.d = pos ? d : -d;
  o.mse = static_cast<float>(wmin / static_cast<double>(n));
  o.score = static_cast<float>(wmax / ((e.w0 + e.w1) + e.w2));
  o.path = e.path;
}

}  // namespace dpx
