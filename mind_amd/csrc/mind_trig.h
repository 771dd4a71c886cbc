/* sin / cos / tan of the bicycle model (trajectory_tree.py:168-175 and its Jacobian), shared by k_ilqr (ilqr_kernels.hip) and the C
 * oracle (oracle/ilqr_ref.c): the same IEEE-754 double operations in the same order on both sides => the same bits, whatever the
 * platform's math library does.  The device library's sincos + tan cost 408 + 512 cycles of a 1.5 k-cycle state-chain node
 * (tools/micro/lat_f64.hip); this evaluation costs about 45 float64 instructions for the pair.
 *
 * Method: k = rint(x * 2/pi); r + rl = x - k * pi/2 with pi/2 = P1 + P2 + P3 (P1 has 33 significant bits: k * P1 is exact for
 * |k| < 2^20, the subtraction x - k P1 is exact, the rounding error of the second step is carried in rl); sin and cos of
 * |r| <= pi/4 from their Taylor polynomials up to r^17 / r^18 (truncation < 1e-19 relative) evaluated so that the last operation
 * adds a small correction to an exactly known leading term; quadrant selection by k mod 4.  Measured against the x87 long-double
 * functions (tests/test_trig.py): sin, cos <= 0.75 ulp, tan (one division + one correction step) <= 0.9 ulp for
 * |x| < 1e5.  Every product-sum that is meant to be fused is written fma(); both translation units are compiled with
 * floating-point contraction off, so nothing else fuses. */
#ifndef MIND_TRIG_H
#define MIND_TRIG_H
#ifdef __HIPCC__
#define MT_FN __device__ __forceinline__
#define MT_ALL_ZERO(k) __all((k) == 0.0)          /* wave-uniform: arguments within pi/4 in every lane skip the reduction */
#else
#include <math.h>
#include <stdbool.h>
#define MT_FN static inline
#define MT_ALL_ZERO(k) 0
#endif

#define MT_2_PI 0x1.45f306dc9c883p-1      /* 2 / pi */
#define MT_P1 0x1.921fb54400000p+0        /* pi / 2, first 33 bits */
#define MT_P2 0x1.0b4611a626331p-34       /* pi / 2 - P1 */
#define MT_P3 0x1.1701b839a2520p-88       /* pi / 2 - P1 - P2 */

/* sin(r + rl) = sh + sl, cos(r + rl) = ch + cl for the reduced argument of x (sh, ch the rounded values, sl, cl what the last
 * addition dropped: the quotient below needs them), q = quadrant (k mod 4) */
MT_FN void mt_kernel(double x, double *sh, double *sl, double *ch, double *cl, int *q) {
  const double k = rint(x * MT_2_PI);
  double r = x, rl = 0.0;
  if (!MT_ALL_ZERO(k)) {                          /* k = 0 gives r = x, rl = +0 below as well: skipping it changes no bit */
    const double r1 = fma(-k, MT_P1, x);          /* exact */
    const double t = k * MT_P2;
    const double tl = fma(k, MT_P2, -t);          /* k P2 = t + tl exactly */
    r = r1 - t;
    const double bb = r - r1;
    const double e = (r1 - (r - bb)) - (t + bb);  /* r1 - t = r + e exactly */
    rl = (e - tl) - k * MT_P3;
  }
  const double z = r * r;
  const double zl = fma(r, r, -z);                /* r^2 = z + zl exactly */
  double sp = 0x1.952c77030ad4ap-49;              /* 1/17! */
  sp = fma(sp, z, -0x1.ae7f3e733b81fp-41);        /* -1/15! */
  sp = fma(sp, z, 0x1.6124613a86d09p-33);         /* 1/13! */
  sp = fma(sp, z, -0x1.ae64567f544e4p-26);        /* -1/11! */
  sp = fma(sp, z, 0x1.71de3a556c734p-19);         /* 1/9! */
  sp = fma(sp, z, -0x1.a01a01a01a01ap-13);        /* -1/7! */
  sp = fma(sp, z, 0x1.1111111111111p-7);          /* 1/5! */
  sp = fma(sp, z, -0x1.5555555555555p-3);         /* -1/3! */
  double cp = -0x1.6827863b97d97p-53;             /* -1/18! */
  cp = fma(cp, z, 0x1.ae7f3e733b81fp-45);         /* 1/16! */
  cp = fma(cp, z, -0x1.93974a8c07c9dp-37);        /* -1/14! */
  cp = fma(cp, z, 0x1.1eed8eff8d898p-29);         /* 1/12! */
  cp = fma(cp, z, -0x1.27e4fb7789f5cp-22);        /* -1/10! */
  cp = fma(cp, z, 0x1.a01a01a01a01ap-16);         /* 1/8! */
  cp = fma(cp, z, -0x1.6c16c16c16c17p-10);        /* -1/6! */
  cp = fma(cp, z, 0x1.5555555555555p-5);          /* 1/4! */
  const double hz = 0.5 * z;
  const double w = 1.0 - hz;
  const double ew = (1.0 - w) - hz;               /* 1 - z/2 = w + ew exactly */
  const double cs_ = fma(z * r, sp, rl * w);      /* sin(r + rl) = r + r^3 S(z) + rl cos(r) */
  const double cc_ = (ew - 0.5 * zl) + fma(z * z, cp, -(r * rl));      /* cos(r + rl) = 1 - r^2/2 + z^2 C(z) - rl sin(r) */
  const double s = r + cs_, c = w + cc_;
  *sh = s; *sl = (r - s) + cs_;                   /* |r| >= |cs_|, |w| >= |cc_|: the dropped parts are exact */
  *ch = c; *cl = (w - c) + cc_;
  *q = (int)(k - 4.0 * floor(0.25 * k));         /* k mod 4 in floating point: a conversion of a huge k would differ between host and device */
}

MT_FN void mind_sincos(double x, double *sn, double *cs) {
  double s, sl, c, cl;
  int q;
  mt_kernel(x, &s, &sl, &c, &cl, &q);
  const double a = (q & 1) ? c : s, b = (q & 1) ? s : c;
  *sn = (q & 2) ? -a : a;
  *cs = ((q + 1) & 2) ? -b : b;
}

/* tan(x) and cos(x) from one evaluation (the Jacobian needs both: trajectory_tree.py:160-166).  tan = n / d with n + nl, d + dl the
 * unrounded sine and cosine (swapped and negated in the odd quadrants): one division, then one Newton step on the quotient that also
 * takes the dropped parts in */
MT_FN void mind_tan_cos(double x, double *tn, double *cs) {
  double s, sl, c, cl;
  int q;
  mt_kernel(x, &s, &sl, &c, &cl, &q);
  const bool odd = (q & 1) != 0;
  const double n = odd ? -c : s, nl = odd ? -cl : sl, d = odd ? s : c, dl = odd ? sl : cl;
  const double rd = 1.0 / d;
  const double t0 = n * rd;
  *tn = t0 + (fma(-t0, d, n) + (nl - t0 * dl)) * rd;
  const double b = odd ? s : c;
  *cs = ((q + 1) & 2) ? -b : b;
}

MT_FN double mind_tan(double x) {
  double t, c;
  mind_tan_cos(x, &t, &c);
  return t;
}
#endif
