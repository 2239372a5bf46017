// frame.hip -- Frame geometry that sits either side of the matchers, on gfx950.
//
// Replaces (reference file:line):
//   Frame::isInFrustum            src/Frame.cc:435-491 (loop: Tracking::SearchLocalPoints, src/Tracking.cc:1071-1091)
//   Frame::UndistortKeyPoints     src/Frame.cc:636-669 (cv::fisheye::undistortPoints, R = I, P = K)
//   Frame::ComputeImageBounds     src/Frame.cc:741-795
//
// Both kernels are one-lane-per-element streaming maps (28-44 B in, 1-21 B out per element): HBM-bound by
// construction, no LDS needed.  Per-problem constants (pose, camera centre) sit in SGPRs via uniform loads.
#include "fb_common.h"
#include "fb_frame_geom.h"

namespace {

constexpr int FRAME_THREADS = 256;

__global__ __launch_bounds__(FRAME_THREADS) void k_in_frustum(fb_frustum_args A) {
  const int b = blockIdx.y;
  const int i = blockIdx.x * FRAME_THREADS + threadIdx.x;
  if (i >= A.n_mp[b]) return;
  const size_t e = (size_t)b * A.mp_stride + i;
  A.in_view[e] = 0;
  if (A.mp_valid && !A.mp_valid[e]) return;
  const float *T = A.Tcw + (size_t)b * 12, *Ow = A.Ow + (size_t)b * 3;
  const float P0 = A.mp_xw[e * 3], P1 = A.mp_xw[e * 3 + 1], P2 = A.mp_xw[e * 3 + 2];
  const float PcX = ((T[0] * P0 + T[1] * P1) + T[2] * P2) + T[3];
  const float PcY = ((T[4] * P0 + T[5] * P1) + T[6] * P2) + T[7];
  const float PcZ = ((T[8] * P0 + T[9] * P1) + T[10] * P2) + T[11];
  if (PcZ < 0.0f) return;
  const float invz = 1.0f / PcZ;
  const float u = A.cam.fx * PcX * invz + A.cam.cx;
  const float v = A.cam.fy * PcY * invz + A.cam.cy;
  if (u < A.cam.min_x || u > A.cam.max_x) return;
  if (v < A.cam.min_y || v > A.cam.max_y) return;
  const float maxD = A.mp_max_dist[e];
  const float maxDistance = 1.2f * maxD, minDistance = 0.8f * A.mp_min_dist[e];
  const float PO0 = P0 - Ow[0], PO1 = P1 - Ow[1], PO2 = P2 - Ow[2];
  const float dist = fb::norm3(PO0, PO1, PO2);
  if (dist < minDistance || dist > maxDistance) return;
  double dot = 0.0;  // cv::Mat::dot accumulates CV_32F products in double
  dot += (double)PO0 * (double)A.mp_normal[e * 3];
  dot += (double)PO1 * (double)A.mp_normal[e * 3 + 1];
  dot += (double)PO2 * (double)A.mp_normal[e * 3 + 2];
  const float viewCos = (float)(dot / (double)dist);
  if (viewCos < A.viewing_cos_limit) return;
  const int lvl = fb::predict_scale(maxD, dist, A.log_scale_factor, A.n_levels);
  A.in_view[e] = 1;
  A.proj[e * 2] = u;
  A.proj[e * 2 + 1] = v;
  if (A.proj_xr) A.proj_xr[e] = u - A.mbf * invz;
  A.level[e] = lvl;
  A.view_cos[e] = viewCos;
}

// Tracking::FilterBirdOutlierInFront geometric test: one workgroup per frame pair.  Pass 1 evaluates every match and
// lets the passing ones compete for their train slot with atomicMin on the match index (the serial rule "the first
// passing match takes the slot"); pass 2 writes the flags.
__global__ __launch_bounds__(256) void k_bird_filter(fb_bird_filter_args A) {
  extern __shared__ int s_first[];  // [kp2_stride] smallest passing match index per train slot
  const int b = blockIdx.x, tid = threadIdx.x, nt = blockDim.x;
  const size_t mo = (size_t)b * A.match_stride, o1 = (size_t)b * A.kp1_stride, o2 = (size_t)b * A.kp2_stride;
  const int nm = A.n_matches[b];
  __shared__ float s_Twc1[12], s_T2[12];
  if (tid == 0) {
    const float *T1 = A.Tcw1 + (size_t)b * 12;
    for (int r = 0; r < 3; r++) {
      for (int c = 0; c < 3; c++) s_Twc1[r * 4 + c] = T1[c * 4 + r];
      s_Twc1[r * 4 + 3] = -((T1[0 * 4 + r] * T1[3] + T1[1 * 4 + r] * T1[7]) + T1[2 * 4 + r] * T1[11]);  // Converter::invT
    }
    for (int i = 0; i < 12; i++) s_T2[i] = A.Tcw2[(size_t)b * 12 + i];
  }
  for (int i = tid; i < A.kp2_stride; i += nt) s_first[i] = 0x7fffffff;
  __syncthreads();
  for (int i = tid; i < nm; i += nt) {
    const int qi = A.query_idx[mo + i], ti = A.train_idx[mo + i];
    bool pass = false;
    float ptw[3] = {0.f, 0.f, 0.f};
    if (!A.occupied2[o2 + ti]) {
      const float *p1 = A.cam_xyz1 + (o1 + qi) * 3;
      float pc2[3];
#pragma unroll
      for (int r = 0; r < 3; r++) ptw[r] = ((s_Twc1[r * 4] * p1[0] + s_Twc1[r * 4 + 1] * p1[1]) + s_Twc1[r * 4 + 2] * p1[2]) + s_Twc1[r * 4 + 3];
#pragma unroll
      for (int r = 0; r < 3; r++) pc2[r] = ((s_T2[r * 4] * ptw[0] + s_T2[r * 4 + 1] * ptw[1]) + s_T2[r * 4 + 2] * ptw[2]) + s_T2[r * 4 + 3];
      const float *p2 = A.cam_xyz2 + (o2 + ti) * 3;
      const float d0 = pc2[0] - p2[0], d1 = pc2[1] - p2[1], d2 = pc2[2] - p2[2];
      const double disC = sqrt((double)d0 * d0 + (double)d1 * d1 + (double)d2 * d2);
      pass = disC < (double)A.window_size;
    }
    if (pass) {
      atomicMin(&s_first[ti], i);
#pragma unroll
      for (int k = 0; k < 3; k++) A.pt_world[(mo + i) * 3 + k] = ptw[k];
    }
    A.keep[mo + i] = pass ? 2 : 0;  // 2 = passed the test, resolved below
  }
  __syncthreads();
  for (int i = tid; i < nm; i += nt)
    if (A.keep[mo + i] == 2) A.keep[mo + i] = (s_first[A.train_idx[mo + i]] == i) ? 1 : 0;
}

struct CamKD { float K[4], D[4]; };

__global__ __launch_bounds__(FRAME_THREADS) void k_undistort(const fb_keypoint *kps, const int32_t *n, int stride, CamKD C,
                                                             fb_keypoint *out) {
  const int b = blockIdx.y;
  const int i = blockIdx.x * FRAME_THREADS + threadIdx.x;
  if (i >= n[b]) return;
  const size_t e = (size_t)b * stride + i;
  fb_keypoint kp = kps[e];
  if (C.D[0] != 0.0f) fb_fisheye_undistort(kp.x, kp.y, C.K, C.D, &kp.x, &kp.y);
  out[e] = kp;
}

}  // namespace

extern "C" {

int fb_in_frustum_dev(const fb_frustum_args *A, void *stream) {
  FB_TRY(fb::check_device());
  FB_ARG(A && A->batch >= 0 && A->mp_stride >= 0 && A->n_levels > 0);
  if (A->batch == 0 || A->mp_stride == 0) return FB_OK;
  FB_ARG(A->Tcw && A->Ow && A->n_mp && A->mp_xw && A->mp_normal && A->mp_max_dist && A->mp_min_dist);
  FB_ARG(A->in_view && A->proj && A->level && A->view_cos);
  FB_ARG(A->batch <= 65535);
  fb::ProfScope prof_(fb::P_FRUSTUM, fb::as_stream(stream));
  const dim3 grid((A->mp_stride + FRAME_THREADS - 1) / FRAME_THREADS, A->batch);
  k_in_frustum<<<grid, FRAME_THREADS, 0, fb::as_stream(stream)>>>(*A);
  FB_HIP(hipGetLastError());
  return FB_OK;
}

int fb_in_frustum(const fb_frustum_args *H) {
  FB_TRY(fb::check_device());
  FB_ARG(H && H->batch >= 0 && H->mp_stride >= 0);
  fb_frustum_args D = *H;
  const size_t B = H->batch, ms = H->mp_stride;
  if (B == 0 || ms == 0) return FB_OK;
  fb::DevBuf b0, b1, b2, b3, b4, b5, b6, b7, o0, o1, o2, o3, o4;
  FB_TRY(b0.upload(H->Tcw, B * 48)); D.Tcw = b0.as<float>();
  FB_TRY(b1.upload(H->Ow, B * 12)); D.Ow = b1.as<float>();
  FB_TRY(b2.upload(H->n_mp, B * 4)); D.n_mp = b2.as<int32_t>();
  if (H->mp_valid) { FB_TRY(b3.upload(H->mp_valid, B * ms)); D.mp_valid = b3.as<uint8_t>(); }
  FB_TRY(b4.upload(H->mp_xw, B * ms * 12)); D.mp_xw = b4.as<float>();
  FB_TRY(b5.upload(H->mp_normal, B * ms * 12)); D.mp_normal = b5.as<float>();
  FB_TRY(b6.upload(H->mp_max_dist, B * ms * 4)); D.mp_max_dist = b6.as<float>();
  FB_TRY(b7.upload(H->mp_min_dist, B * ms * 4)); D.mp_min_dist = b7.as<float>();
  // outputs are in/out (entries that are not in view keep their previous content)
  FB_TRY(o0.upload(H->in_view, B * ms)); D.in_view = o0.as<uint8_t>();
  FB_TRY(o1.upload(H->proj, B * ms * 8)); D.proj = o1.as<float>();
  if (H->proj_xr) { FB_TRY(o2.upload(H->proj_xr, B * ms * 4)); D.proj_xr = o2.as<float>(); }
  FB_TRY(o3.upload(H->level, B * ms * 4)); D.level = o3.as<int32_t>();
  FB_TRY(o4.upload(H->view_cos, B * ms * 4)); D.view_cos = o4.as<float>();
  FB_TRY(fb_in_frustum_dev(&D, nullptr));
  FB_HIP(hipDeviceSynchronize());
  FB_TRY(o0.download(H->in_view, B * ms));
  FB_TRY(o1.download(H->proj, B * ms * 8));
  if (H->proj_xr) FB_TRY(o2.download(H->proj_xr, B * ms * 4));
  FB_TRY(o3.download(H->level, B * ms * 4));
  return o4.download(H->view_cos, B * ms * 4);
}

int fb_bird_filter_matches_dev(const fb_bird_filter_args *A, void *stream) {
  FB_TRY(fb::check_device());
  FB_ARG(A && A->batch >= 0 && A->match_stride >= 0 && A->kp1_stride > 0 && A->kp2_stride > 0 && A->kp2_stride <= 40000);
  if (A->batch == 0 || A->match_stride == 0) return FB_OK;
  FB_ARG(A->n_matches && A->query_idx && A->train_idx && A->cam_xyz1 && A->cam_xyz2 && A->Tcw1 && A->Tcw2 && A->occupied2 && A->keep && A->pt_world);
  const size_t lds = (size_t)A->kp2_stride * 4;
  FB_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(k_bird_filter), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  fb::ProfScope prof_(fb::P_FRUSTUM, fb::as_stream(stream));
  k_bird_filter<<<A->batch, 256, lds, fb::as_stream(stream)>>>(*A);
  FB_HIP(hipGetLastError());
  return FB_OK;
}

int fb_bird_filter_matches(const fb_bird_filter_args *H) {
  FB_TRY(fb::check_device());
  FB_ARG(H && H->batch >= 0 && H->match_stride >= 0);
  fb_bird_filter_args D = *H;
  const size_t B = H->batch, ms = H->match_stride, s1 = H->kp1_stride, s2 = H->kp2_stride;
  if (B == 0 || ms == 0) return FB_OK;
  fb::DevBuf b0, b1, b2, b3, b4, b5, b6, b7, o0, o1;
  FB_TRY(b0.upload(H->n_matches, B * 4)); D.n_matches = b0.as<int32_t>();
  FB_TRY(b1.upload(H->query_idx, B * ms * 4)); D.query_idx = b1.as<int32_t>();
  FB_TRY(b2.upload(H->train_idx, B * ms * 4)); D.train_idx = b2.as<int32_t>();
  FB_TRY(b3.upload(H->cam_xyz1, B * s1 * 12)); D.cam_xyz1 = b3.as<float>();
  FB_TRY(b4.upload(H->cam_xyz2, B * s2 * 12)); D.cam_xyz2 = b4.as<float>();
  FB_TRY(b5.upload(H->Tcw1, B * 48)); D.Tcw1 = b5.as<float>();
  FB_TRY(b6.upload(H->Tcw2, B * 48)); D.Tcw2 = b6.as<float>();
  FB_TRY(b7.upload(H->occupied2, B * s2)); D.occupied2 = b7.as<uint8_t>();
  FB_TRY(o0.upload(H->keep, B * ms)); D.keep = o0.as<uint8_t>();
  FB_TRY(o1.upload(H->pt_world, B * ms * 12)); D.pt_world = o1.as<float>();
  FB_TRY(fb_bird_filter_matches_dev(&D, nullptr));
  FB_HIP(hipDeviceSynchronize());
  FB_TRY(o0.download(H->keep, B * ms));
  return o1.download(H->pt_world, B * ms * 12);
}

int fb_undistort_keypoints_dev(const fb_keypoint *d_kps, const int32_t *d_n, int batch, int kp_stride, const float *K4,
                               const float *D4, fb_keypoint *d_kps_un, void *stream) {
  FB_TRY(fb::check_device());
  FB_ARG(batch >= 0 && kp_stride >= 0 && K4 && D4 && batch <= 65535);
  if (batch == 0 || kp_stride == 0) return FB_OK;
  FB_ARG(d_kps && d_n && d_kps_un);
  CamKD C;
  for (int i = 0; i < 4; i++) { C.K[i] = K4[i]; C.D[i] = D4[i]; }
  fb::ProfScope prof_(fb::P_UNDISTORT, fb::as_stream(stream));
  const dim3 grid((kp_stride + FRAME_THREADS - 1) / FRAME_THREADS, batch);
  k_undistort<<<grid, FRAME_THREADS, 0, fb::as_stream(stream)>>>(d_kps, d_n, kp_stride, C, d_kps_un);
  FB_HIP(hipGetLastError());
  return FB_OK;
}

int fb_undistort_keypoints(const fb_keypoint *kps, int n, const float *K4, const float *D4, fb_keypoint *kps_un) {
  FB_TRY(fb::check_device());
  FB_ARG(n >= 0 && K4 && D4);
  if (n == 0) return FB_OK;
  FB_ARG(kps && kps_un);
  fb::DevBuf k, c, o;
  const int32_t n32 = n;
  FB_TRY(k.upload(kps, (size_t)n * sizeof(fb_keypoint)));
  FB_TRY(c.upload(&n32, 4));
  FB_TRY(o.alloc((size_t)n * sizeof(fb_keypoint)));
  FB_TRY(fb_undistort_keypoints_dev(k.as<fb_keypoint>(), c.as<int32_t>(), 1, n, K4, D4, o.as<fb_keypoint>(), nullptr));
  FB_HIP(hipDeviceSynchronize());
  return o.download(kps_un, (size_t)n * sizeof(fb_keypoint));
}

int fb_image_bounds(int cols, int rows, const float *K4, const float *D4, float *bounds) {
  FB_ARG(cols > 0 && rows > 0 && K4 && D4 && bounds);
  if (D4[0] == 0.0f) {  // Frame.cc:787-793
    bounds[0] = 0.0f; bounds[1] = (float)cols; bounds[2] = 0.0f; bounds[3] = (float)rows;
    return FB_OK;
  }
  fb_keypoint c[4] = {}, u[4];
  c[1].x = (float)cols; c[2].y = (float)rows; c[3].x = (float)cols; c[3].y = (float)rows;
  FB_TRY(fb_undistort_keypoints(c, 4, K4, D4, u));
  // sic: the running maxima start from numeric_limits<float>::min(), the smallest POSITIVE float (Frame.cc:758-761)
  float mnMinX = 3.402823466e+38f, mnMaxX = 1.175494351e-38f, mnMinY = 3.402823466e+38f, mnMaxY = 1.175494351e-38f;
  for (int i = 0; i < 4; i++) {
    if (u[i].x < mnMinX) mnMinX = u[i].x;
    if (u[i].x > mnMaxX) mnMaxX = u[i].x;
    if (u[i].y < mnMinY) mnMinY = u[i].y;
    if (u[i].y > mnMaxY) mnMaxY = u[i].y;
  }
  bounds[0] = mnMinX; bounds[1] = mnMaxX; bounds[2] = mnMinY; bounds[3] = mnMaxY;
  return FB_OK;
}

}  // extern "C"
