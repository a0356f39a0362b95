// dpx_math.h — double-precision elementary functions with ONE definition for host and device.
//
// Why this exists: the reference's plane fit calls glibc `atan2/cos/sin/sqrt` (libs/dsyev/src/dsyevc3.c:62-72)
// and its normal histogram calls glibc `acos/atan2/sqrt` (cpp/deplex/src/deplex/normals_histogram.cpp:35-37).
// On the GPU those names resolve to OCML, whose results are not bit-identical to glibc's. Every function
// here is plain IEEE-754 binary64 arithmetic (+ - * / sqrt, no FMA, no table lookups that depend on the
// target), so the same source compiled with `-ffp-contract=off` by g++ (oracle, "devmath" flavour) and by
// hipcc for gfx950 returns the same bits. Accuracy is < 1 ulp (measured against mpmath in
// tests/test_devmath.py), i.e. the same class as glibc's; disagreements with glibc are counted there.
//
// Nothing in this header is taken from the reference; constants are Taylor coefficients and atan(i/16)
// values generated with mpmath (tools/gen_math_constants.py).
#pragma once

#if defined(__HIPCC__)
#define DPX_HD __host__ __device__ inline
#else
#define DPX_HD inline
#endif

namespace dpxm {

constexpr double kPiHi = 0x1.921fb54442d18p+1;    // 3.141592653589793
constexpr double kPiLo = 0x1.1a62633145c07p-53;   // 1.2246467991473532e-16
constexpr double kPio2Hi = 0x1.921fb54442d18p+0;  // 1.5707963267948966
constexpr double kPio2Lo = 0x1.1a62633145c07p-54;
constexpr double kPio4Hi = 0x1.921fb54442d18p-1;
constexpr double kPio4Lo = 0x1.1a62633145c07p-55;

DPX_HD double dabs(double x) { return __builtin_fabs(x); }
DPX_HD bool disnan(double x) { return x != x; }
DPX_HD bool dsignbit(double x) { return __builtin_signbit(x); }
// IEEE correctly rounded on both targets (x86 sqrtsd; AMDGPU lowers llvm.sqrt.f64 to a correctly
// rounded sequence) — checked against host results in tests/test_gpu_parity.py::test_devmath_on_device.
DPX_HD double dsqrt(double x) { return __builtin_sqrt(x); }

// sin(x + t) for |x| <= pi/4, t a tail with |t| << |x|.  Taylor through x^19.
DPX_HD double ksin(double x, double t) {
  const double S1 = -0x1.5555555555555p-3, S2 = 0x1.1111111111111p-7, S3 = -0x1.a01a01a01a01ap-13,
               S4 = 0x1.71de3a556c734p-19, S5 = -0x1.ae64567f544e4p-26, S6 = 0x1.6124613a86d09p-33,
               S7 = -0x1.ae7f3e733b81fp-41, S8 = 0x1.952c77030ad4ap-49, S9 = -0x1.2f49b46814157p-57;
  double z = x * x;
  double v = z * x;
  double r = S2 + z * (S3 + z * (S4 + z * (S5 + z * (S6 + z * (S7 + z * (S8 + z * S9))))));
  // x + [ v*S1 + t + (v*z*r - 0.5*z*t) ]
  return x - ((z * (0.5 * t - v * r) - t) - v * S1);
}

// cos(x + t) for |x| <= pi/4.  Taylor through x^20, evaluated as (1 - q) - ((z/2 - q) - (z*z*R - x*t))
// with q ~ x/4 so that the leading subtraction is exact.
DPX_HD double kcos(double x, double t) {
  const double C2 = 0x1.5555555555555p-5, C3 = -0x1.6c16c16c16c17p-10, C4 = 0x1.a01a01a01a01ap-16,
               C5 = -0x1.27e4fb7789f5cp-22, C6 = 0x1.1eed8eff8d898p-29, C7 = -0x1.93974a8c07c9dp-37,
               C8 = 0x1.ae7f3e733b81fp-45, C9 = -0x1.6827863b97d97p-53, C10 = 0x1.e542ba4020225p-62;
  double z = x * x;
  double r = z * (C2 + z * (C3 + z * (C4 + z * (C5 + z * (C6 + z * (C7 + z * (C8 + z * (C9 + z * C10))))))));
  double ax = dabs(x);
  if (ax < 0.3) return 1.0 - (0.5 * z - (z * r - x * t));
  double q = ax > 0.78125 ? 0.28125 : 0.25 * ax;
  // keep only the leading bits of q so that 1-q and z/2-q are exact enough
  q = (q + 0x1.0p+30 * q) - 0x1.0p+30 * q;  // rounds q to ~23 significant bits
  double hz = 0.5 * z - q;
  double a = 1.0 - q;
  return a - (hz - (z * r - x * t));
}

// sin and cos for 0 <= x <= ~1.2 (the Cardano angle is atan2(..)/3 in [0, pi/3]).  Outside that range
// (only reachable with non-finite input) NaN is returned for both.
DPX_HD void sincos_small(double x, double* s, double* c) {
  if (!(x >= -kPio4Hi && x <= 1.25)) {
    *s = __builtin_nan("");
    *c = *s;
    return;
  }
  // Above pi/4 the complement y = pi/2 - x (head + tail) is evaluated with the roles of sin and cos swapped.  Both
  // cases run through ONE evaluation of each kernel with selected arguments (on a GPU the lanes of a wave fall on
  // both sides of pi/4 and two separate branches would both be executed); the operations on each value are unchanged.
  const bool small = x <= kPio4Hi;
  const double h = kPio2Hi - x;  // exact (Sterbenz) where it is used
  const double yc = h + kPio2Lo;
  const double tc = (h - yc) + kPio2Lo;
  const double y = small ? x : yc, t = small ? 0.0 : tc;
  const double sv = ksin(y, t), cv = kcos(y, t);
  *s = small ? sv : cv;
  *c = small ? cv : sv;
}

// atan(a) for 0 <= a <= 1, returned as head + tail (tail only approximate).
DPX_HD double atan01(double a, double* tail) {
  const double T[17][2] = {{0x0.0p+0, 0x0.0p+0},
                           {0x1.ff55bb72cfdeap-5, -0x1.c934d86d23f1dp-60},
                           {0x1.fd5ba9aac2f6ep-4, -0x1.cd37686760c17p-59},
                           {0x1.7b97b4bce5b02p-3, 0x1.347b0b4f881cap-58},
                           {0x1.f5b75f92c80ddp-3, 0x1.8ab6e3cf7afbdp-57},
                           {0x1.362773707ebccp-2, -0x1.963a544b672d8p-57},
                           {0x1.6f61941e4def1p-2, -0x1.c63aae6f6e918p-56},
                           {0x1.a64eec3cc23fdp-2, -0x1.24dec1b50b7ffp-56},
                           {0x1.dac670561bb4fp-2, 0x1.a2b7f222f65e2p-56},
                           {0x1.0657e94db30d0p-1, -0x1.d5b495f6349e6p-56},
                           {0x1.1e00babdefeb4p-1, -0x1.928df287a668fp-58},
                           {0x1.345f01cce37bbp-1, 0x1.1021137c71102p-55},
                           {0x1.4978fa3269ee1p-1, 0x1.2419a87f2a458p-56},
                           {0x1.5d58987169b18p-1, 0x1.0028e4bc5e7cap-57},
                           {0x1.700a7c5784634p-1, -0x1.8c34d25aadef6p-56},
                           {0x1.819d0b7158a4dp-1, -0x1.bf76229d3b917p-56},
                           {0x1.921fb54442d18p-1, 0x1.1a62633145c07p-55}};
  int i = (int)(a * 16.0 + 0.5);  // nearest table point, 0..16
  double c = (double)i * 0.0625;
  double r = (i == 0) ? a : (a - c) / (1.0 + a * c);  // |r| <= 1/32 (+rounding)
  double z = r * r;
  // atan(r) = r - r^3/3 + r^5/5 - ... through r^15
  double p = z * (-0x1.5555555555555p-2 +
                  z * (0x1.999999999999ap-3 +
                       z * (-0x1.2492492492492p-3 +
                            z * (0x1.c71c71c71c71cp-4 +
                                 z * (-0x1.745d1745d1746p-4 + z * (0x1.3b13b13b13b14p-4 + z * -0x1.1111111111111p-4))))));
  double small = r + (r * p + T[i][1]);
  double hi = T[i][0] + small;
  *tail = (T[i][0] - hi) + small;
  return hi;
}

// atan2 with the C99 special cases that can arise here (zeros, infinities, NaN).
DPX_HD double datan2(double y, double x) {
  if (disnan(x) || disnan(y)) return x + y;
  double ay = dabs(y), ax = dabs(x);
  double z, zt = 0.0;
  const double inf = __builtin_inf();
  if (ay == 0.0) {
    z = 0.0;  // result 0 or pi below
    if (dsignbit(x)) {
      z = kPiHi;
    }
    return dsignbit(y) ? -z : z;
  }
  if (ax == 0.0) {
    return dsignbit(y) ? -kPio2Hi : kPio2Hi;
  }
  if (ax == inf || ay == inf) {
    if (ax == inf && ay == inf)
      z = dsignbit(x) ? 3.0 * kPio4Hi : kPio4Hi;
    else if (ay == inf)
      z = kPio2Hi;
    else
      z = dsignbit(x) ? kPiHi : 0.0;
    return dsignbit(y) ? -z : z;
  }
  bool swap = ay > ax;
  const double num = swap ? ax : ay, den = swap ? ay : ax;  // (one division, operands selected)
  double a = num / den;
  z = atan01(a, &zt);
  if (swap) {  // pi/2 - z
    double h = kPio2Hi - z;
    double t = ((kPio2Hi - h) - z) + (kPio2Lo - zt);
    z = h + t;
    zt = (h - z) + t;
  }
  if (dsignbit(x)) {  // pi - z
    double h = kPiHi - z;
    double t = ((kPiHi - h) - z) + (kPiLo - zt);
    z = h + t;
  }
  return dsignbit(y) ? -z : z;
}

// acos(x) = atan2(sqrt((1-x)(1+x)), x); NaN for |x| > 1.
DPX_HD double dacos(double x) {
  if (disnan(x)) return x;
  if (x > 1.0 || x < -1.0) return __builtin_nan("");
  return datan2(dsqrt((1.0 - x) * (1.0 + x)), x);
}

}  // namespace dpxm
