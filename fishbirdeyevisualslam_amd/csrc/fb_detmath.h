/*
 * fb_detmath.h -- scalar helpers whose results must be bit-identical on the host
 * (gcc) and on gfx950 (hipcc).  Only IEEE +,-,*,/ and rint/floor are used and both
 * sides are compiled with -ffp-contract=off, so every function here is a pure
 * function of its input bits.
 *
 * Product copy.  The oracle has its OWN, independently written oracle/fb_detmath.h (libm in double rounded to float,
 * OpenCV's integer rounding formulas): the two can disagree, and tests/test_detmath_independent.py compares them.
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

/* natural logarithm of a positive finite float, computed in double with +,-,*,/ only and rounded to float
 * (MapPoint::PredictScale calls libm logf, MapPoint.cc:393,410; restated like fb_sincos_f so that the CPU and the
 * GPU agree on every bit).  x = m * 2^e, m in [sqrt(1/2), sqrt(2)): ln x = e ln2 + 2 atanh((m-1)/(m+1)). */
FB_HD static inline float fb_log_f(float xf) {
  double x = (double)xf;
  int e = 0;
  /* exact scaling by powers of two */
  for (int it = 0; it < 400 && x >= 1.4142135623730951; it++) { x *= 0.5; e++; }
  for (int it = 0; it < 400 && x < 0.7071067811865476; it++) { x *= 2.0; e--; }
  const double z = (x - 1.0) / (x + 1.0), z2 = z * z;
  double p = 1.0 / 23.0;
  p = p * z2 + 1.0 / 21.0;
  p = p * z2 + 1.0 / 19.0;
  p = p * z2 + 1.0 / 17.0;
  p = p * z2 + 1.0 / 15.0;
  p = p * z2 + 1.0 / 13.0;
  p = p * z2 + 1.0 / 11.0;
  p = p * z2 + 1.0 / 9.0;
  p = p * z2 + 1.0 / 7.0;
  p = p * z2 + 1.0 / 5.0;
  p = p * z2 + 1.0 / 3.0;
  p = p * z2 + 1.0;
  const double ln2_hi = 0x1.62e42fefa39efp-1;
  return (float)((double)e * ln2_hi + 2.0 * z * p);
}

/* tan of a double in (-pi/2, pi/2) from the deterministic sin/cos kernels above (cv::fisheye::undistortPoints
 * calls std::tan); only used through fb_fisheye_undistort. */
FB_HD static inline double fb_tan_d(double x) {
  /* reduce with the same scheme as fb_sincos_f but keep doubles */
  const double two_over_pi = 0x1.45f306dc9c883p-1;
  const double pio2_hi = 0x1.921fb54442d18p+0;
  const double pio2_lo = 0x1.1a62633145c07p-54;
  const double kd = rint(x * two_over_pi);
  const int k = (int)kd;
  double r = x - kd * pio2_hi;
  r = r - kd * pio2_lo;
  const double r2 = r * r;
  double ps = -1.0 / 355687428096000.0;
  ps = ps * r2 + 1.0 / 1307674368000.0;
  ps = ps * r2 - 1.0 / 6227020800.0;
  ps = ps * r2 + 1.0 / 39916800.0;
  ps = ps * r2 - 1.0 / 362880.0;
  ps = ps * r2 + 1.0 / 5040.0;
  ps = ps * r2 - 1.0 / 120.0;
  ps = ps * r2 + 1.0 / 6.0;
  const double sr = r - (r * r2) * ps;
  double pc = 1.0 / 6402373705728000.0;
  pc = pc * r2 - 1.0 / 20922789888000.0;
  pc = pc * r2 + 1.0 / 87178291200.0;
  pc = pc * r2 - 1.0 / 479001600.0;
  pc = pc * r2 + 1.0 / 3628800.0;
  pc = pc * r2 - 1.0 / 40320.0;
  pc = pc * r2 + 1.0 / 720.0;
  pc = pc * r2 - 1.0 / 24.0;
  pc = pc * r2 + 0.5;
  const double cr = 1.0 - r2 * pc;
  return (k & 1) ? -cr / sr : sr / cr;
}

/* cv::fisheye::undistortPoints(src, dst, K, D, R = I, P = K) for one point, OpenCV 3.0-3.3 semantics
 * (10 fixed-point iterations), call sites Frame.cc:657,754.  K = (fx, fy, cx, cy) and D = k1..k4 as float. */
FB_HD static inline void fb_fisheye_undistort(float px, float py, const float K4[4], const float D4[4], float *ox, float *oy) {
  const double fx = K4[0], fy = K4[1], cx = K4[2], cy = K4[3];
  const double k0 = D4[0], k1 = D4[1], k2 = D4[2], k3 = D4[3];
  const double pwx = ((double)px - cx) / fx, pwy = ((double)py - cy) / fy;
  double scale = 1.0;
  const double theta_d = sqrt(pwx * pwx + pwy * pwy);
  if (theta_d > 1e-8) {
    double theta = theta_d;
    for (int j = 0; j < 10; j++) {
      const double theta2 = theta * theta, theta4 = theta2 * theta2, theta6 = theta4 * theta2, theta8 = theta6 * theta2;
      theta = theta_d / (1 + k0 * theta2 + k1 * theta4 + k2 * theta6 + k3 * theta8);
    }
    scale = fb_tan_d(theta) / theta_d;
  }
  const double pux = pwx * scale, puy = pwy * scale;
  /* pr = K * (pu, 1): pr.z = 1 */
  const double prx = fx * pux + 0.0 * puy + cx * 1.0, pry = 0.0 * pux + fy * puy + cy * 1.0, prz = 0.0 * pux + 0.0 * puy + 1.0;
  *ox = (float)(prx / prz);
  *oy = (float)(pry / prz);
}

#endif /* FB_DETMATH_H_ */
