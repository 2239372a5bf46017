/*
 * fb_detmath.h -- scalar helpers whose results must be bit-identical on the host
 * (gcc) and on gfx950 (hipcc).  Only IEEE +,-,*,/ and rint/floor are used and both
 * sides are compiled with -ffp-contract=off, so every function here is a pure
 * function of its input bits.
 *
 * The same text lives in oracle/ (CPU restatement, test infrastructure) and in
 * fishbirdeyevisualslam_amd/csrc/ (product); neither includes the other.
 *
 * OpenCV semantics restated here (OpenCV is NOT vendored in the reference, so this
 * is "parity unpinned" -- see DESIGN.md):
 *   cvRound      -> round half to even          (call sites ORBextractor.cc:81,119-120,442,460,1112)
 *   cv::fastAtan2-> 7th-order odd polynomial, degrees (call site ORBextractor.cc:103)
 *   cos/sin      -> ORBextractor.cc:112-113 calls libm cosf/sinf; restated as a
 *                   double-precision polynomial rounded to float (correctly rounded
 *                   in all but ~1e-8 of cases; deterministic on both sides).
 */
#ifndef FB_DETMATH_H_
#define FB_DETMATH_H_

#ifndef FB_HD
#define FB_HD
#endif

#include <math.h>

FB_HD static inline int fb_cvround(float v) { return (int)rintf(v); }
FB_HD static inline int fb_cvround_d(double v) { return (int)rint(v); }
FB_HD static inline int fb_cvfloor(float v) { return (int)floorf(v); }
FB_HD static inline int fb_cvceil(float v) { return (int)ceilf(v); }

/* cv::fastAtan2(y, x) in degrees, [0,360) */
FB_HD static inline float fb_fast_atan2(float y, float x) {
  const float p1 = 0x1.ca44dep+5f;  /* 0.9997878412794807f*(float)(180/CV_PI) */
  const float p3 = -0x1.2aaddcp+4f; /* -0.3258083974640975f*... */
  const float p5 = 0x1.1d3f7ep+3f;  /* 0.1555786518463281f*... */
  const float p7 = -0x1.4515b2p+1f; /* -0.04432655554792128f*... */
  const float eps = 0x1p-52f;       /* (float)DBL_EPSILON */
  float ax = fabsf(x), ay = fabsf(y);
  float a, c, c2;
  if (ax >= ay) {
    c = ay / (ax + eps);
    c2 = c * c;
    a = (((p7 * c2 + p5) * c2 + p3) * c2 + p1) * c;
  } else {
    c = ax / (ay + eps);
    c2 = c * c;
    a = 90.f - (((p7 * c2 + p5) * c2 + p3) * c2 + p1) * c;
  }
  if (x < 0) a = 180.f - a;
  if (y < 0) a = 360.f - a;
  return a;
}

/* sin and cos of a float angle in radians (|x| < ~10), computed in double with
 * plain mul/add only, rounded to float. */
FB_HD static inline void fb_sincos_f(float xf, float *s_out, float *c_out) {
  const double x = (double)xf;
  const double two_over_pi = 0x1.45f306dc9c883p-1;
  const double pio2_hi = 0x1.921fb54442d18p+0;
  const double pio2_lo = 0x1.1a62633145c07p-54;
  const double kd = rint(x * two_over_pi);
  const int k = (int)kd;
  double r = x - kd * pio2_hi;
  r = r - kd * pio2_lo;
  const double r2 = r * r;
  /* Taylor, |r| <= pi/4 (+tiny): sin to r^17, cos to r^18 */
  double ps = -1.0 / 355687428096000.0;            /* -1/17! */
  ps = ps * r2 + 1.0 / 1307674368000.0;            /* 1/15! */
  ps = ps * r2 - 1.0 / 6227020800.0;               /* -1/13! */
  ps = ps * r2 + 1.0 / 39916800.0;                 /* 1/11! */
  ps = ps * r2 - 1.0 / 362880.0;                   /* -1/9! */
  ps = ps * r2 + 1.0 / 5040.0;                     /* 1/7! */
  ps = ps * r2 - 1.0 / 120.0;                      /* -1/5! */
  ps = ps * r2 + 1.0 / 6.0;                        /* 1/3!, sign folded below */
  const double sr = r - (r * r2) * ps;
  double pc = 1.0 / 6402373705728000.0;            /* 1/18! */
  pc = pc * r2 - 1.0 / 20922789888000.0;           /* -1/16! */
  pc = pc * r2 + 1.0 / 87178291200.0;              /* 1/14! */
  pc = pc * r2 - 1.0 / 479001600.0;                /* -1/12! */
  pc = pc * r2 + 1.0 / 3628800.0;                  /* 1/10! */
  pc = pc * r2 - 1.0 / 40320.0;                    /* -1/8! */
  pc = pc * r2 + 1.0 / 720.0;                      /* 1/6! */
  pc = pc * r2 - 1.0 / 24.0;                       /* -1/4! */
  pc = pc * r2 + 0.5;                              /* 1/2! */
  const double cr = 1.0 - r2 * pc;
  double s, c;
  switch (k & 3) {
    case 0: s = sr; c = cr; break;
    case 1: s = cr; c = -sr; break;
    case 2: s = -sr; c = -cr; break;
    default: s = -cr; c = sr; break;
  }
  *s_out = (float)s;
  *c_out = (float)c;
}

#endif /* FB_DETMATH_H_ */
