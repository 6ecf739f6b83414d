/*
 * d2d_sincos.h — restatement of the libm sin() / cos() that the reference's Oxford gaze policy calls.
 *
 * Reference call site: yaw_planner.py:71 `vec_yaw = [math.cos(math.radians(yaw)), -math.sin(math.radians(yaw))]`
 * (math.cos / math.sin -> host libm).  The view-cone test `arccos(q) <= half_fov` that follows decides
 * cells that lie exactly on the cone's edge (drone on an integer position, yaw a multiple of 45 degrees:
 * the reference's own start pose), so the direction vector has to carry libm's roundings, not just be
 * accurate.  glibc 2.35's double sin / cos are the IBM Accurate Mathematical Library routines
 * (sysdeps/ieee754/dbl-64/s_sin.c) with the multi-precision slow paths removed: < 1 ulp, not correctly
 * rounded.  x86-64 libm dispatches them through an ifunc; on every CPU with FMA + AVX2 it resolves to
 * the variant built with -mfma -mavx2, where the compiler contracted a fixed set of multiply-adds.  The
 * sequence below is that variant's published algorithm, operation for operation and fused where it is
 * fused there:
 *
 *   do_sin(x, dx)   |x| < 0.126: odd Taylor polynomial s1..s5 with the dx correction
 *                   else x = x_k + r (x_k = k / 128 by the `big` trick), sin x_k cos r + cos x_k sin r
 *                   from the table row (sn, ssn, cs, ccs) and short polynomials in r
 *   do_cos(x, dx)   the same split, cos x_k cos r - sin x_k sin r
 *   sin x           |x| < 2^-26: x ; < 0.855469: do_sin(x, 0) ; < 2.426265: do_cos(hp0 - |x|, hp1) ;
 *                   < 105414350: n, a + da = x - n pi/2 (hpinv / toint trick, pi/2 = mp1 + mp2 + pp3 + pp4),
 *                   then do_sin or do_cos by n & 1, negated by n & 2
 *   cos x           |x| < 2^-27: 1 ; < 0.855469: do_cos(x, 0) ; < 2.426265: a + da = hp0 - |x| + hp1, do_sin ;
 *                   < 105414350: the same reduction with n + 1
 *   |x| >= 105414350 is not needed (arguments are radians(yaw), yaw in [0, 360)) and returns NaN.
 *
 * Must be compiled with -ffp-contract=off: every '*' '+' '-' below is one IEEE-754 binary64 operation,
 * every D2D_FMA one fused multiply-add.  tests/test_sincos.py checks the host build of this file against
 * libm bit for bit on > 10^7 arguments, tests/test_gpu_parity.py the device build.
 */
#ifndef D2D_SINCOS_H
#define D2D_SINCOS_H

#ifndef D2D_SINCOS_QUAL
#define D2D_SINCOS_QUAL static inline
#endif
#ifndef D2D_SINCOS_TBL_QUAL
#define D2D_SINCOS_TBL_QUAL static const
#endif
#ifndef D2D_FMA
#define D2D_FMA(a, b, c) __builtin_fma((a), (b), (c))
#endif

#include "d2d_sincos_tbl.h"

#define D2D_SC_BIG 0x1.8p+45
#define D2D_SC_SN3 (-0x1.5555555555515p-3)
#define D2D_SC_SN5 0x1.11110e829872fp-7
#define D2D_SC_CS2 0x1p-1
#define D2D_SC_CS4 (-0x1.5555555555535p-5)
#define D2D_SC_CS6 0x1.6c16bedd9e239p-10

D2D_SINCOS_QUAL int d2d_sc_row(double u) {
  long long b;
  __builtin_memcpy(&b, &u, 8);
  return (int)(b & 0xffffffffll);
}

D2D_SINCOS_QUAL double d2d_taylor_sin(double a, double da) {
  const double s1 = -0x1.5555555555555p-3, s2 = 0x1.1111111110ecep-7, s3 = -0x1.a01a019db08b8p-13,
               s4 = 0x1.71de27b9a7ed9p-19, s5 = -0x1.addffc2fcdf59p-26;
  const double xx = a * a;
  double p = D2D_FMA(xx, s5, s4);
  p = D2D_FMA(xx, p, s3);
  p = D2D_FMA(xx, p, s2);
  p = D2D_FMA(xx, p, s1);
  const double t = D2D_FMA(xx, D2D_FMA(p, a, -(0.5 * da)), da);
  return a + t;
}

D2D_SINCOS_QUAL double d2d_do_sin(double x, double dx) {
  const double w = __builtin_fabs(x);
  if (w < 0x1.020c49ba5e354p-3) return d2d_taylor_sin(x, dx);
  if (x <= 0.0) dx = -dx;
  const double u = w + D2D_SC_BIG;
  const double r = w - (u - D2D_SC_BIG);
  const int k = d2d_sc_row(u);
  const double sn = d2d_sincos_tbl[k][0], ssn = d2d_sincos_tbl[k][1], cs = d2d_sincos_tbl[k][2], ccs = d2d_sincos_tbl[k][3];
  const double xx = r * r;
  const double s = r + D2D_FMA(r * xx, D2D_FMA(xx, D2D_SC_SN5, D2D_SC_SN3), dx);
  const double c = D2D_FMA(r, dx, xx * D2D_FMA(xx, D2D_FMA(xx, D2D_SC_CS6, D2D_SC_CS4), D2D_SC_CS2));
  const double cor = D2D_FMA(s, cs, D2D_FMA(-c, sn, D2D_FMA(s, ccs, ssn)));
  return __builtin_copysign(sn + cor, x);
}

D2D_SINCOS_QUAL double d2d_do_cos(double x, double dx) {
  const double w = __builtin_fabs(x);
  if (x < 0.0) dx = -dx;
  const double u = w + D2D_SC_BIG;
  const double r = (w - (u - D2D_SC_BIG)) + dx;
  const int k = d2d_sc_row(u);
  const double sn = d2d_sincos_tbl[k][0], ssn = d2d_sincos_tbl[k][1], cs = d2d_sincos_tbl[k][2], ccs = d2d_sincos_tbl[k][3];
  const double xx = r * r;
  const double s = D2D_FMA(r * xx, D2D_FMA(xx, D2D_SC_SN5, D2D_SC_SN3), r);
  const double c = xx * D2D_FMA(xx, D2D_FMA(xx, D2D_SC_CS6, D2D_SC_CS4), D2D_SC_CS2);
  const double cor = D2D_FMA(-s, sn, D2D_FMA(-c, cs, D2D_FMA(-s, ssn, ccs)));
  return cs + cor;
}

/* n, a + da = x - n pi/2 for 2.426265 <= |x| < 105414350 */
D2D_SINCOS_QUAL int d2d_reduce_sincos(double x, double *a, double *da) {
  const double hpinv = 0x1.45f306dc9c883p-1, toint = 0x1.8p+52;
  const double mp1 = 0x1.921fb58p+0, mp2 = -0x1.dde973cp-27, pp3 = -0x1.cb3b398p-55, pp4 = -0x1.d747f23e32ed7p-83;
  const double t = D2D_FMA(x, hpinv, toint);
  const double xn = t - toint;
  const double y = D2D_FMA(-xn, mp2, D2D_FMA(-xn, mp1, x));
  const double t2 = D2D_FMA(-xn, pp3, y);
  double db = D2D_FMA(-xn, pp3, y - t2);
  const double b = D2D_FMA(-xn, pp4, t2);
  db = db + D2D_FMA(-xn, pp4, t2 - b);
  *a = b;
  *da = db;
  return d2d_sc_row(t) & 3;
}

D2D_SINCOS_QUAL double d2d_do_sincos(double a, double da, int n) {
  const double r = (n & 1) ? d2d_do_cos(a, da) : d2d_do_sin(a, da);
  return (n & 2) ? -r : r;
}

D2D_SINCOS_QUAL double d2d_sin(double x) {
  const double hp0 = 0x1.921fb54442d18p+0, hp1 = 0x1.1a62633145c07p-54;
  const double w = __builtin_fabs(x);
  if (w < 0x1p-26) return x;
  if (w < 0x1.b6p-1) return d2d_do_sin(x, 0.0);                                /* 0.85546875 */
  if (w < 0x1.368fdp+1) return __builtin_copysign(d2d_do_cos(hp0 - w, hp1), x); /* 2.426265 */
  if (!(w < 0x1.921fbp+26)) return __builtin_nan("");                         /* 105414350 */
  double a, da;
  const int n = d2d_reduce_sincos(x, &a, &da);
  return d2d_do_sincos(a, da, n);
}

D2D_SINCOS_QUAL double d2d_cos(double x) {
  const double hp0 = 0x1.921fb54442d18p+0, hp1 = 0x1.1a62633145c07p-54;
  const double w = __builtin_fabs(x);
  if (w < 0x1p-27) return 1.0;
  if (w < 0x1.b6p-1) return d2d_do_cos(x, 0.0);
  if (w < 0x1.368fdp+1) {
    const double y = hp0 - w;
    const double a = y + hp1;
    const double da = (y - a) + hp1;
    return d2d_do_sin(a, da);
  }
  if (!(w < 0x1.921fbp+26)) return __builtin_nan("");
  double a, da;
  const int n = d2d_reduce_sincos(x, &a, &da);
  return d2d_do_sincos(a, da, n + 1);
}

/* sin x (is_cos == 0) or cos x (is_cos != 0) in ONE pass through the routines above: d2d_sin and d2d_cos both end in
 * do_sin or do_cos of a reduced argument (a, da) -- which of the two, with which (a, da) and which sign fix-up, is what differs
 * between them and between the ranges of |x|.  A SIMD caller that needs both the sine and the cosine of its angles gives the two to
 * different lanes and pays for one reduction, one do_sin and one do_cos instead of two of each (the gaze stage: lanes 0..7 the
 * cosines, 8..15 the sines of its eight view directions).  Every lane performs exactly the operations d2d_sin / d2d_cos perform for
 * its argument: the results are bit-identical (tests/test_sincos.py and the device hook run THIS function against libm). */
D2D_SINCOS_QUAL double d2d_sin_or_cos(double x, int is_cos) {
  const double hp0 = 0x1.921fb54442d18p+0, hp1 = 0x1.1a62633145c07p-54;
  const double w = __builtin_fabs(x);
  if (w < (is_cos ? 0x1p-27 : 0x1p-26)) return is_cos ? 1.0 : x;
  if (!(w < 0x1.921fbp+26)) return __builtin_nan(""); /* 105414350 */
  /* what to evaluate: do_cos (use_cos) or do_sin of (a, da); then copysign(result, x) (sign_x) or negation (neg) */
  double a, da;
  int use_cos, sign_x = 0, neg = 0;
  if (w < 0x1.b6p-1) { /* 0.85546875 */
    a = x;
    da = 0.0;
    use_cos = is_cos;
  } else if (w < 0x1.368fdp+1) { /* 2.426265 */
    if (is_cos) {
      const double y = hp0 - w;
      a = y + hp1;
      da = (y - a) + hp1;
      use_cos = 0;
    } else {
      a = hp0 - w;
      da = hp1;
      use_cos = 1;
      sign_x = 1;
    }
  } else {
    const int n = d2d_reduce_sincos(x, &a, &da) + (is_cos ? 1 : 0);
    use_cos = n & 1;
    neg = n & 2;
  }
  double r = use_cos ? d2d_do_cos(a, da) : d2d_do_sin(a, da);
  if (sign_x) r = __builtin_copysign(r, x);
  return neg ? -r : r;
}

#endif /* D2D_SINCOS_H */
