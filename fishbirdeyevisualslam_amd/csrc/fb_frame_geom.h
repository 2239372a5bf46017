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

}  // namespace fb
