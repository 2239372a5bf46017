/*
 * frame_oracle.cpp -- CPU restatement of the bird-frame steps between extraction and matching
 * (TEST INFRASTRUCTURE ONLY; see orb_oracle.cpp header for who may call it).
 *
 * Follows /root/reference/src/Frame.cc:671-684 (GuidenceKeyBirdPts), :686-715 (genEdgesPC), :717-739 (nearEdges),
 * :365-373 with src/Converter.cc:284-292,312-318 (BirdPixel2BaseXY, BaseXY2CamXYZ).  All of it is plain C++ in the
 * reference (no OpenCV arithmetic beyond Mat::at and one 3x3 float product), so nothing here is a guess; the mask
 * predicate is the KeyPointsFilter::runByPixelsMask rule applied to the level-0 position (part of the E9 substitution).
 */
#include <cstddef>
#include <cstdint>
#include <cstring>

#include "../include/fishbird.h"

namespace {

inline uint8_t at(const uint8_t *img, int rows, int cols, int pitch, size_t row, size_t col) {
  // Mat::at<uchar>(row, col); outside the image the reference reads out of bounds -- counted as free (0) here
  if (row >= (size_t)rows || col >= (size_t)cols) return 0;
  return img[row * (size_t)pitch + col];
}

bool near_edges(const uint8_t *icp, int rows, int cols, int pitch, const fb_keypoint &kpt) {  // Frame.cc:717-739
  const int r = 10;
  const float pt1x = (kpt.x - r) > 0 ? (kpt.x - r) : 0;
  const float pt1y = (kpt.y - r) > 0 ? (kpt.y - r) : 0;
  const float pt2x = (kpt.x + r) < cols ? (kpt.x + r) : cols;
  const float pt2y = (kpt.y + r) < rows ? (kpt.y + r) : rows;
  for (size_t row = pt1x; row < pt2x; row++)      // sic: x walks the rows
    for (size_t col = pt1y; col < pt2y; col++) {
      if (at(icp, rows, cols, pitch, row, col) < 10) continue;  // free
      return true;                                               // edge (< 150) or freespace
    }
  return false;
}

}  // namespace

extern "C" int orc_bird_guidance(const fb_bird_guidance_args *A) {
  for (int b = 0; b < A->batch; b++) {
    const uint8_t *icp = A->contour + (size_t)b * A->rows * A->pitch;
    const uint8_t *mask = A->mask ? A->mask + (size_t)b * A->rows * A->pitch : nullptr;
    const size_t ko = (size_t)b * A->kp_stride;
    if (A->edge_cap > 0) {  // genEdgesPC
      int ns = 0, nf = 0;
      for (size_t row = 0; row < (size_t)A->rows; row++)
        for (size_t col = 0; col < (size_t)A->cols; col++) {
          const uint8_t v = icp[row * A->pitch + col];
          if (v < 10) continue;
          const int label = v < 150 ? 0 : 1;
          float *dst = label ? A->edge_free : A->edge_sign;
          int &n = label ? nf : ns;
          if (n < A->edge_cap) { dst[((size_t)b * A->edge_cap + n) * 2] = (float)col; dst[((size_t)b * A->edge_cap + n) * 2 + 1] = (float)row; }
          n++;
        }
      A->n_edge_sign[b] = ns;
      A->n_edge_free[b] = nf;
    }
    int n = 0;
    for (int i = 0; i < A->n_in[b]; i++) {
      const fb_keypoint kpt = A->kps_in[ko + i];
      bool ok = true;
      if (mask) {  // KeyPointsFilter::runByPixelsMask
        const int my = (int)(kpt.y + 0.5f), mx = (int)(kpt.x + 0.5f);
        ok = my >= 0 && my < A->rows && mx >= 0 && mx < A->cols && mask[(size_t)my * A->pitch + mx] != 0;
      }
      ok = ok && near_edges(icp, A->rows, A->cols, A->pitch, kpt);
      if (A->keep) A->keep[ko + i] = ok ? 1 : 0;
      if (!ok) continue;
      A->kps_out[ko + n] = kpt;
      if (A->desc_in) std::memcpy(A->desc_out + (ko + n) * 32, A->desc_in + (ko + i) * 32, 32);
      n++;
    }
    A->n_out[b] = n;
  }
  return FB_OK;
}

// Frame.cc:365-373: mvKeysBirdCamXYZ[k] = BaseXY2CamXYZ(BirdPixel2BaseXY(mvKeysBird[k]))
extern "C" int orc_bird_keys_to_cam(const fb_keypoint *kps, const int32_t *n, int batch, int kp_stride, int bird_cols, int bird_rows,
                                    double pixel2meter, double rear_axle_to_center, const float *Tcb12, float *cam_xyz) {
  for (int b = 0; b < batch; b++)
    for (int i = 0; i < n[b]; i++) {
      const fb_keypoint &kp = kps[(size_t)b * kp_stride + i];
      float p[3];  // Converter.cc:284-292: int/2 - float in float, times double, stored to float
      p[0] = (float)((bird_rows / 2 - kp.y) * pixel2meter + rear_axle_to_center);
      p[1] = (float)((bird_cols / 2 - kp.x) * pixel2meter);
      p[2] = 0;
      float *dst = cam_xyz + ((size_t)b * kp_stride + i) * 3;
      for (int r = 0; r < 3; r++)  // Converter.cc:312-318: 3x3 float Mat product, then + the translation column
        dst[r] = ((Tcb12[r * 4] * p[0] + Tcb12[r * 4 + 1] * p[1]) + Tcb12[r * 4 + 2] * p[2]) + Tcb12[r * 4 + 3];
    }
  return FB_OK;
}
