/*
 * oracle/fb_detmath.h -- the ORACLE's own statement of the scalar OpenCV / libm semantics on the hot path.
 * TEST INFRASTRUCTURE.  Written independently of fishbirdeyevisualslam_amd/csrc/fb_detmath.h (the product's
 * bit-reproducible kernels): different formulations on purpose, so that an error in either shows up as a parity
 * failure instead of hiding in shared text.  tests/test_detmath_independent.py compiles both into one program and
 * compares them densely (exhaustively for sin/cos over [0, 2 pi]).
 *
 *   cvRound / cvFloor / cvCeil  OpenCV's own formulas: lrint (current rounding mode = nearest-even) and the
 *                               "truncate, then correct" integer forms (call sites ORBextractor.cc:81,119-120,442,460,1112)
 *   cv::fastAtan2               the published polynomial with its decimal coefficients times (float)(180/CV_PI)
 *                               (call site ORBextractor.cc:103)
 *   cosf / sinf / logf / tan    what the reference actually calls is libm (ORBextractor.cc:112-113, MapPoint.cc:393,410,
 *                               cv::fisheye::undistortPoints): here the DOUBLE libm function rounded once to float -- correctly
 *                               rounded except where the double result lies within ~1e-16 relative of a float rounding
 *                               boundary.  The product uses its own polynomial kernels; the two agree on every float angle in
 *                               [0, 2 pi] (exhaustive run recorded in DESIGN.md section 2).
 * OpenCV is not vendored in the reference: all of this is "parity unpinned" (DESIGN.md section 2).
 */
#ifndef FB_ORACLE_DETMATH_H_
#define FB_ORACLE_DETMATH_H_

#ifndef FB_HD
#define FB_HD
#endif

#include <math.h>

FB_HD static inline int fb_cvround(float v) { return (int)lrintf(v); }
FB_HD static inline int fb_cvround_d(double v) { return (int)lrint(v); }
FB_HD static inline int fb_cvfloor(float v) { const int i = (int)v; return i - (i > v); }
FB_HD static inline int fb_cvceil(float v) { const int i = (int)v; return i + (i < v); }

/* cv::fastAtan2(y, x), degrees in [0, 360) */
FB_HD static inline float fb_fast_atan2(float y, float x) {
  const float deg = (float)(180.0 / 3.1415926535897932384626433832795);
  const float p1 = 0.9997878412794807f * deg, p3 = -0.3258083974640975f * deg;
  const float p5 = 0.1555786518463281f * deg, p7 = -0.04432655554792128f * deg;
  const float tiny = (float)2.2204460492503131e-16; /* (float)DBL_EPSILON */
  const float ax = x < 0 ? -x : x, ay = y < 0 ? -y : y;
  float a;
  if (ax >= ay) {
    const float c = ay / (ax + tiny), c2 = c * c;
    a = (((p7 * c2 + p5) * c2 + p3) * c2 + p1) * c;
  } else {
    const float c = ax / (ay + tiny), c2 = c * c;
    a = 90.f - (((p7 * c2 + p5) * c2 + p3) * c2 + p1) * c;
  }
  if (x < 0) a = 180.f - a;
  if (y < 0) a = 360.f - a;
  return a;
}

FB_HD static inline void fb_sincos_f(float x, float *s_out, float *c_out) {
  *s_out = (float)sin((double)x);
  *c_out = (float)cos((double)x);
}

FB_HD static inline float fb_log_f(float x) { return (float)log((double)x); }

FB_HD static inline double fb_tan_d(double x) { return tan(x); }

/* cv::fisheye::undistortPoints(src, dst, K, D, R = I, P = K), one point, OpenCV 3.0-3.3: ten fixed-point iterations of
 * theta <- theta_d / (1 + k1 theta^2 + k2 theta^4 + k3 theta^6 + k4 theta^8) (call sites Frame.cc:657,754). */
FB_HD static inline void fb_fisheye_undistort(float px, float py, const float K4[4], const float D4[4], float *ox, float *oy) {
  const double f[2] = {K4[0], K4[1]}, c[2] = {K4[2], K4[3]};
  const double k[4] = {D4[0], D4[1], D4[2], D4[3]};
  const double pw[2] = {((double)px - c[0]) / f[0], ((double)py - c[1]) / f[1]};
  double scale = 1.0;
  const double theta_d = sqrt(pw[0] * pw[0] + pw[1] * pw[1]);
  if (theta_d > 1e-8) {
    double theta = theta_d;
    for (int j = 0; j < 10; j++) {
      const double theta2 = theta * theta, theta4 = theta2 * theta2, theta6 = theta4 * theta2, theta8 = theta6 * theta2;
      theta = theta_d / (1 + k[0] * theta2 + k[1] * theta4 + k[2] * theta6 + k[3] * theta8);
    }
    scale = tan(theta) / theta_d;
  }
  const double pu[2] = {pw[0] * scale, pw[1] * scale};
  /* P = K as a 3x3 product with (pu, 1), then the perspective division by pr.z = 1 */
  const double pr[3] = {f[0] * pu[0] + 0.0 * pu[1] + c[0] * 1.0, 0.0 * pu[0] + f[1] * pu[1] + c[1] * 1.0, 0.0 * pu[0] + 0.0 * pu[1] + 1.0};
  *ox = (float)(pr[0] / pr[2]);
  *oy = (float)(pr[1] / pr[2]);
}

#endif /* FB_ORACLE_DETMATH_H_ */
