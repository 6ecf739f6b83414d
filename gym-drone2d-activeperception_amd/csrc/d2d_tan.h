/*
 * d2d_tan.h — restatement of the libm tan() that the reference's ray-slope computation calls.
 *
 * Reference call site: utils.py:640 `slope = tan(ray_angle)` (math.tan -> host libm).  The host libm of
 * the reference's runtime here is glibc 2.35 (Ubuntu 2.35-0ubuntu3.x), whose double tan is the IBM
 * Accurate Mathematical Library routine (sysdeps/ieee754/dbl-64/s_tan.c) with the multi-precision
 * slow paths removed: the first-stage result is returned, so the function is NOT correctly rounded
 * (it differs from a correctly rounded tan by 1 ulp on ~0.26 % of arguments).  Ray cells are decided by
 * floor(x / scale) of iterated sums of 9 * slope, so for bit-exact occupancy grids the device must
 * reproduce those roundings, not just be accurate.  x86-64 libm dispatches tan through an ifunc; on
 * every CPU with FMA + AVX2 (any host an MI355X sits in) it resolves to the variant built with
 * -mfma -mavx2, where the compiler contracted a fixed set of multiply-adds.  The sequence below is that
 * variant's published algorithm, operation for operation and fused where it is fused there:
 *
 *   (I)   |x| <= 1.259e-8            tan x = x
 *   (II)  |x| <= 0.0608              odd polynomial d3..d11
 *   (III) |x| <= 0.787               table: x = x_i + z, tan = f_i + pz (g_i + f_i) / (g_i - pz)
 *   (IV)  |x| <= 25                  reduce by pi/2 = mp1 + mp2 + mp3 (x * hpinv + toint trick),
 *                                     a + da = x - n pi/2; then (II) or (III) on |a| with da folded in;
 *                                     odd n: -cot by a double-length reciprocal (II) or the table's
 *                                     cot column (III).
 *   |x| > 25 is not needed (ray angles lie in [0, 2 pi)) and returns NaN.
 *
 * Must be compiled with -ffp-contract=off: every '*' '+' '-' '/' below is one IEEE-754 binary64
 * operation, every D2D_FMA one fused multiply-add.  tests/test_tan.py checks the host build of this
 * file against libm tan bit for bit on >10^7 arguments, tests/test_gpu_parity.py the device build.
 */
#ifndef D2D_TAN_H
#define D2D_TAN_H

#ifndef D2D_TAN_QUAL
#define D2D_TAN_QUAL static inline
#endif
#ifndef D2D_TAN_TBL_QUAL
#define D2D_TAN_TBL_QUAL static const
#endif
#define D2D_FMA(a, b, c) __builtin_fma((a), (b), (c))

#include "d2d_tan_tbl.h"

D2D_TAN_QUAL double d2d_tan(double x) {
  const double g1 = 0x1.b096cp-27, g2 = 0x1.f212dp-5, g3 = 0x1.92f1ap-1, g4 = 25.0;
  const double d3 = 0x1.5555555555555p-2, d5 = 0x1.11111111107c6p-3, d7 = 0x1.ba1ba1cdb8745p-5,
               d9 = 0x1.664ed49cfc666p-6, d11 = 0x1.2385a3cf2e4eap-7;
  const double e0 = 0x1.5555555554dbdp-2, e1 = 0x1.11112e0a6b45fp-3;
  const double hpinv = 0x1.45f306dc9c883p-1, toint = 0x1.8p+52;
  const double mp1 = 0x1.921fb58p+0, mp2 = -0x1.dde973cp-27, mp3 = -0x1.cb3b399d747f2p-55;

  const double w = __builtin_fabs(x);
  if (w <= g1) return x;
  if (w <= g2) {
    const double x2 = x * x;
    double t = D2D_FMA(x2, d11, d9);
    t = D2D_FMA(x2, t, d7);
    t = D2D_FMA(x2, t, d5);
    t = D2D_FMA(x2, t, d3);
    return D2D_FMA(x * x2, t, x);
  }
  if (w <= g3) {
    const int i = (int)D2D_FMA(w, 256.0, -15.5);
    const double z = w - d2d_tan_tbl[i][0];
    const double z2 = z * z;
    const double pz = D2D_FMA(z * z2, D2D_FMA(z2, e1, e0), z);
    const double fi = d2d_tan_tbl[i][1], gi = d2d_tan_tbl[i][2];
    const double t2 = ((fi + gi) * pz) / (gi - pz);
    const double y = t2 + fi;
    return (x < 0.0) ? -y : y;
  }
  if (!(w <= g4)) return __builtin_nan("");

  /* (IV) reduction */
  const double t = D2D_FMA(x, hpinv, toint);
  const double xn = t - toint;
  long long tb;
  __builtin_memcpy(&tb, &t, 8);
  const int n = (int)(tb & 1);
  const double t1 = D2D_FMA(-xn, mp2, D2D_FMA(-xn, mp1, x));
  const double a = D2D_FMA(-xn, mp3, t1);
  const double da = D2D_FMA(-xn, mp3, t1 - a);
  const double sy = (a < 0.0) ? -1.0 : 1.0;
  const double ya = (a < 0.0) ? -a : a;
  const double yya = (a < 0.0) ? -da : da;

  if (ya <= g2) {
    const double a2 = a * a;
    double p = D2D_FMA(a2, d11, d9);
    p = D2D_FMA(a2, p, d7);
    p = D2D_FMA(a2, p, d5);
    p = D2D_FMA(a2, p, d3);
    const double t2 = D2D_FMA(a * a2, p, da);
    const double y = a + t2;
    if (n == 0) return y;
    /* -cot: (b, db) = a + t2 as a double-length sum, then 1 / (b + db) in double length */
    const double db = (__builtin_fabs(a) > __builtin_fabs(t2)) ? (a - y) + t2 : (t2 - y) + a;
    const double c = 1.0 / y;
    const double u3 = c * y;
    const double u4 = D2D_FMA(c, y, -u3);
    double r = ((1.0 - u3) - u4) + 0.0;
    r = D2D_FMA(-db, c, r);
    const double dc = r / y;
    const double s = c + dc;
    const double q = ((c - s) + dc) + s;
    return -q;
  }
  {
    const int i = (int)D2D_FMA(ya, 256.0, -15.5);
    const double z = (ya - d2d_tan_tbl[i][0]) + yya;
    const double z2 = z * z;
    const double pz = D2D_FMA(z * z2, D2D_FMA(z2, e1, e0), z);
    const double fi = d2d_tan_tbl[i][1], gi = d2d_tan_tbl[i][2];
    const double num = (fi + gi) * pz;
    if (n == 0) {
      const double y = num / (gi - pz) + fi;
      return y * sy;
    }
    const double y = gi - num / (pz + fi);
    return y * (-sy);
  }
}

#endif /* D2D_TAN_H */
