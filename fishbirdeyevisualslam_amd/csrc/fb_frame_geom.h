// fb_frame_geom.h -- device helpers shared by the projection matchers and the Frame geometry kernels.
// Reference semantics (file:line) are quoted per function; OpenCV's accumulation widths are restated as in
// DESIGN.md ("parity unpinned": OpenCV is not vendored in the reference).
#pragma once
#include "fb_common.h"

namespace fb {

// MapPoint::PredictScale(currentDist, Frame*), src/MapPoint.cc:401-416 (libm logf -> fb_log_f)
__device__ __forceinline__ int predict_scale(float maxDistance, float currentDist, float logScaleFactor, int nLevels) {
  const float ratio = maxDistance / currentDist;
  int nScale;
  if (!(ratio > 0.0f) || isinf(ratio)) nScale = 0;
  else nScale = (int)ceilf(fb_log_f(ratio) / logScaleFactor);
  if (nScale < 0) nScale = 0;
  else if (nScale >= nLevels) nScale = nLevels - 1;
  return nScale;
}

// Ow = -Rcw.t()*tcw (ORBmatcher.cc:1479, Frame.cc:432): gemm general path, double accumulation
__device__ __forceinline__ void camera_centre(const float *T, float *Ow) {
#pragma unroll
  for (int r = 0; r < 3; r++) {
    double s = 0.0;
#pragma unroll
    for (int k = 0; k < 3; k++) s += (double)T[k * 4 + r] * (double)T[k * 4 + 3];
    Ow[r] = (float)(-s);
  }
}

// cv::norm of a 3x1 CV_32F (double sum of squares)
__device__ __forceinline__ float norm3(float a, float b, float c) {
  double s = 0.0;
  s += (double)a * (double)a;
  s += (double)b * (double)b;
  s += (double)c * (double)c;
  return (float)sqrt(s);
}

// Frame::isInFrustum(pMP, viewingCosLimit), src/Frame.cc:435-491, for one map point: returns false when the point is not
// in view, otherwise the MapPoint track members it fills (:483-488).  T = mRcw | mtcw rows, Ow = mOw.
struct FrustumOut { float u, v, invz, view_cos; int level; };
__device__ __forceinline__ bool in_frustum(const float *T, const float *Ow, const fb_camera &cam, float P0, float P1, float P2,
                                           float n0, float n1, float n2, float maxD, float minD, float viewing_cos_limit,
                                           float log_scale_factor, int n_levels, FrustumOut &o) {
  const float PcX = ((T[0] * P0 + T[1] * P1) + T[2] * P2) + T[3];
  const float PcY = ((T[4] * P0 + T[5] * P1) + T[6] * P2) + T[7];
  const float PcZ = ((T[8] * P0 + T[9] * P1) + T[10] * P2) + T[11];
  if (PcZ < 0.0f) return false;
  const float invz = 1.0f / PcZ;
  const float u = cam.fx * PcX * invz + cam.cx;
  const float v = cam.fy * PcY * invz + cam.cy;
  if (u < cam.min_x || u > cam.max_x) return false;
  if (v < cam.min_y || v > cam.max_y) return false;
  const float maxDistance = 1.2f * maxD, minDistance = 0.8f * minD;
  const float PO0 = P0 - Ow[0], PO1 = P1 - Ow[1], PO2 = P2 - Ow[2];
  const float dist = norm3(PO0, PO1, PO2);
  if (dist < minDistance || dist > maxDistance) return false;
  double dot = 0.0;  // cv::Mat::dot accumulates CV_32F products in double
  dot += (double)PO0 * (double)n0;
  dot += (double)PO1 * (double)n1;
  dot += (double)PO2 * (double)n2;
  const float viewCos = (float)(dot / (double)dist);
  if (viewCos < viewing_cos_limit) return false;
  o.u = u; o.v = v; o.invz = invz; o.view_cos = viewCos;
  o.level = predict_scale(maxD, dist, log_scale_factor, n_levels);
  return true;
}

// Converter::invT(Tcw) rows 0..2 (Converter.cc:176-187), as Tracking::FilterBirdOutlierInFront uses it (Tracking.cc:1832)
__device__ __forceinline__ void inv_T(const float *T1, float *Twc) {
  for (int r = 0; r < 3; r++) {
    for (int c = 0; c < 3; c++) Twc[r * 4 + c] = T1[c * 4 + r];
    Twc[r * 4 + 3] = -((T1[0 * 4 + r] * T1[3] + T1[1 * 4 + r] * T1[7]) + T1[2 * 4 + r] * T1[11]);
  }
}

// geometric test of Tracking::FilterBirdOutlierInFront (Tracking.cc:1866-1886): ptwC = Twc1 * pt1c, disC = |Tcw2 * ptwC - pt2c|
__device__ __forceinline__ bool bird_filter_test(const float *Twc1, const float *T2, const float *p1, const float *p2,
                                                 float window_size, float *ptw) {
  float pc2[3];
#pragma unroll
  for (int r = 0; r < 3; r++) ptw[r] = ((Twc1[r * 4] * p1[0] + Twc1[r * 4 + 1] * p1[1]) + Twc1[r * 4 + 2] * p1[2]) + Twc1[r * 4 + 3];
#pragma unroll
  for (int r = 0; r < 3; r++) pc2[r] = ((T2[r * 4] * ptw[0] + T2[r * 4 + 1] * ptw[1]) + T2[r * 4 + 2] * ptw[2]) + T2[r * 4 + 3];
  const float d0 = pc2[0] - p2[0], d1 = pc2[1] - p2[1], d2 = pc2[2] - p2[2];
  const double disC = sqrt((double)d0 * d0 + (double)d1 * d1 + (double)d2 * d2);
  return disC < (double)window_size;
}

}  // namespace fb
