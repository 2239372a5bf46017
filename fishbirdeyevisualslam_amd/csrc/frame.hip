// frame.hip -- Frame geometry that sits either side of the matchers, on gfx950.
//
// Replaces (reference file:line):
//   Frame::isInFrustum            src/Frame.cc:435-491 (loop: Tracking::SearchLocalPoints, src/Tracking.cc:1071-1091)
//   Frame::UndistortKeyPoints     src/Frame.cc:636-669 (cv::fisheye::undistortPoints, R = I, P = K)
//   Frame::ComputeImageBounds     src/Frame.cc:741-795
//   Frame::GuidenceKeyBirdPts     src/Frame.cc:671-684 with nearEdges :717-739 and genEdgesPC :686-715 (bird key point filter)
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
  fb::FrustumOut o;
  if (!fb::in_frustum(T, Ow, A.cam, A.mp_xw[e * 3], A.mp_xw[e * 3 + 1], A.mp_xw[e * 3 + 2], A.mp_normal[e * 3], A.mp_normal[e * 3 + 1],
                      A.mp_normal[e * 3 + 2], A.mp_max_dist[e], A.mp_min_dist[e], A.viewing_cos_limit, A.log_scale_factor, A.n_levels, o))
    return;
  const float u = o.u, v = o.v, invz = o.invz, viewCos = o.view_cos;
  const int lvl = o.level;
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
    fb::inv_T(A.Tcw1 + (size_t)b * 12, s_Twc1);
    for (int i = 0; i < 12; i++) s_T2[i] = A.Tcw2[(size_t)b * 12 + i];
  }
  for (int i = tid; i < A.kp2_stride; i += nt) s_first[i] = 0x7fffffff;
  __syncthreads();
  for (int i = tid; i < nm; i += nt) {
    const int qi = A.query_idx[mo + i], ti = A.train_idx[mo + i];
    bool pass = false;
    float ptw[3] = {0.f, 0.f, 0.f};
    if (!A.occupied2[o2 + ti]) {
      pass = fb::bird_filter_test(s_Twc1, s_T2, A.cam_xyz1 + (o1 + qi) * 3, A.cam_xyz2 + (o2 + ti) * 3, A.window_size, ptw);
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


// ---- Frame::GuidenceKeyBirdPts (Frame.cc:671-739) ------------------------------------------------------------------
constexpr int GUIDE_THREADS = 1024;

__device__ __forceinline__ int wave_incl_scan_i(int v) {
  const int lane = threadIdx.x & 63;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const int t = __shfl_up(v, o, 64);
    if (lane >= o) v += t;
  }
  return v;
}
// exclusive scan of one int per thread over the 1024-thread block (s_w: 16 ints); *total = block sum
__device__ __forceinline__ int block_excl_scan1024(int v, int *s_w, int *total) {
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int inc = wave_incl_scan_i(v);
  __syncthreads();
  if (lane == 63) s_w[wv] = inc;
  __syncthreads();
  int base = 0, tot = 0;
#pragma unroll
  for (int w = 0; w < GUIDE_THREADS / 64; w++) { const int x = s_w[w]; if (w < wv) base += x; tot += x; }
  *total = tot;
  return base + inc - v;
}

// One workgroup per bird image.  Phase A: 32 lanes per key point, one lane per contour ROW of its box (the box has at
// most 21 rows: trunc(x-10) .. ceil(x+10)-1), each lane walks its <= 21 contiguous bytes; the verdict is the OR over the
// half wave.  Phase B: stable compaction (push_back order) with one block scan; descriptors move as two 16-byte words.
__global__ __launch_bounds__(GUIDE_THREADS) void k_bird_guidance(fb_bird_guidance_args A) {
  extern __shared__ uint8_t s_keep[];  // [kp_stride]
  __shared__ int s_w[GUIDE_THREADS / 64];
  const int b = blockIdx.x, tid = threadIdx.x;
  const int n = min(max(A.n_in[b], 0), A.kp_stride);  // the LDS flag array and the output rows hold kp_stride entries
  const size_t ko = (size_t)b * A.kp_stride;
  const uint8_t *icp = A.contour + (size_t)b * A.rows * A.pitch;
  const uint8_t *mask = A.mask ? A.mask + (size_t)b * A.rows * A.pitch : nullptr;
  const int sub = tid & 31;
  for (int base = 0; base < n; base += GUIDE_THREADS / 32) {
    const int i = base + (tid >> 5);
    bool hit = false, ok = true;
    if (i < n) {
      const fb_keypoint kpt = A.kps_in[ko + i];
      const int r = 10;
      const float pt1x = (kpt.x - r) > 0 ? (kpt.x - r) : 0;
      const float pt1y = (kpt.y - r) > 0 ? (kpt.y - r) : 0;
      const float pt2x = (kpt.x + r) < A.cols ? (kpt.x + r) : A.cols;
      const float pt2y = (kpt.y + r) < A.rows ? (kpt.y + r) : A.rows;
      const size_t row = (size_t)pt1x + sub;       // size_t row = pt1x; row < pt2x; row++
      if ((float)row < pt2x && row < (size_t)A.rows) {
        const uint8_t *src = icp + row * (size_t)A.pitch;
        for (size_t col = (size_t)pt1y; (float)col < pt2y && col < (size_t)A.cols; col++)
          if (src[col] >= 10) { hit = true; break; }
      }
      if (mask) {
        const int my = (int)(kpt.y + 0.5f), mx = (int)(kpt.x + 0.5f);
        ok = my >= 0 && my < A.rows && mx >= 0 && mx < A.cols && mask[(size_t)my * A.pitch + mx] != 0;
      }
    }
    const unsigned long long bal = __ballot(hit);
    const unsigned int mine = (unsigned int)(bal >> (tid & 32));
    if (sub == 0 && i < n) {
      const uint8_t k = (ok && mine != 0u) ? 1 : 0;
      s_keep[i] = k;
      if (A.keep) A.keep[ko + i] = k;
    }
  }
  __syncthreads();
  const int per = (n + GUIDE_THREADS - 1) / GUIDE_THREADS;
  const int i0 = tid * per, i1 = min(n, i0 + per);
  int cnt = 0;
  for (int i = i0; i < i1; i++) cnt += s_keep[i];
  int total;
  int o = block_excl_scan1024(cnt, s_w, &total);
  for (int i = i0; i < i1; i++) {
    if (!s_keep[i]) continue;
    A.kps_out[ko + o] = A.kps_in[ko + i];
    if (A.desc_in) {
      const uint4 *src = reinterpret_cast<const uint4 *>(A.desc_in + (ko + i) * 32);
      uint4 *dst = reinterpret_cast<uint4 *>(A.desc_out + (ko + o) * 32);
      dst[0] = src[0]; dst[1] = src[1];
    }
    o++;
  }
  if (tid == 0) A.n_out[b] = total;
}

// The same in two launches, taken when the caller passes a `keep` array (it doubles as the flag store): the verdicts by
// 8 key points per 256-thread workgroup -- thousands of independent row walks in flight instead of 32 at a time inside
// one workgroup, whose serialised byte loads made the one-kernel version 0.29 ms per frame at batch 1 (rocprof, round 3)
// -- and the stable compaction by one workgroup per image.
__global__ __launch_bounds__(256) void k_bird_flags(fb_bird_guidance_args A, int aligned4) {
  const int b = blockIdx.y, tid = threadIdx.x;
  const int n = min(max(A.n_in[b], 0), A.kp_stride);
  const int i = blockIdx.x * 8 + (tid >> 5), sub = tid & 31;
  if (blockIdx.x * 8 >= n) return;
  const size_t ko = (size_t)b * A.kp_stride;
  const uint8_t *icp = A.contour + (size_t)b * A.rows * A.pitch;
  const uint8_t *mask = A.mask ? A.mask + (size_t)b * A.rows * A.pitch : nullptr;
  bool hit = false, ok = true;
  if (i < n) {
    const fb_keypoint kpt = A.kps_in[ko + i];
    const int r = 10;
    const float pt1x = (kpt.x - r) > 0 ? (kpt.x - r) : 0;
    const float pt1y = (kpt.y - r) > 0 ? (kpt.y - r) : 0;
    const float pt2x = (kpt.x + r) < A.cols ? (kpt.x + r) : A.cols;
    const float pt2y = (kpt.y + r) < A.rows ? (kpt.y + r) : A.rows;
    const size_t row = (size_t)pt1x + sub;       // size_t row = pt1x; row < pt2x; row++
    if ((float)row < pt2x && row < (size_t)A.rows) {
      const uint8_t *src = icp + row * (size_t)A.pitch;
      // columns [c0, c1): size_t col = pt1y; col < pt2y && col < cols (at most 21 of them)
      const size_t c0 = (size_t)pt1y;
      size_t c1 = (size_t)pt2y;
      if ((float)c1 < pt2y) c1++;                 // first integer >= pt2y
      if (c1 > (size_t)A.cols) c1 = (size_t)A.cols;
      if (aligned4) {
        // the <= 21 bytes as <= 6 aligned dwords requested together (a byte loop costs one dependent round trip per
        // column); a byte b is >= 10  <=>  ((b & 0x7f) + 0x76) | b  has bit 7 set
        const size_t d0 = c0 >> 2;
        uint32_t any = 0;
#pragma unroll
        for (int k = 0; k < 6; k++) {
          const size_t d = d0 + k;
          if (d * 4 >= c1) break;
          const uint32_t v = reinterpret_cast<const uint32_t *>(src)[d];
          uint32_t m = (((v & 0x7f7f7f7fu) + 0x76767676u) | v) & 0x80808080u;
          const long long lo = (long long)c0 - (long long)(d * 4), hi = (long long)c1 - (long long)(d * 4);  // valid bytes [lo, hi)
          if (lo > 0) m &= 0xffffffffu << (8 * (int)lo);
          if (hi < 4) m &= 0xffffffffu >> (8 * (int)(4 - hi));
          any |= m;
        }
        hit = any != 0u;
      } else {
        for (size_t col = c0; col < c1; col++)
          if (src[col] >= 10) { hit = true; break; }
      }
    }
    if (mask) {
      const int my = (int)(kpt.y + 0.5f), mx = (int)(kpt.x + 0.5f);
      ok = my >= 0 && my < A.rows && mx >= 0 && mx < A.cols && mask[(size_t)my * A.pitch + mx] != 0;
    }
  }
  const unsigned long long bal = __ballot(hit);
  const unsigned int mine = (unsigned int)(bal >> (tid & 32));
  if (sub == 0 && i < n) A.keep[ko + i] = (ok && mine != 0u) ? 1 : 0;
}

__global__ __launch_bounds__(GUIDE_THREADS) void k_bird_compact(fb_bird_guidance_args A) {
  __shared__ int s_w[GUIDE_THREADS / 64];
  const int b = blockIdx.x, tid = threadIdx.x;
  const int n = min(max(A.n_in[b], 0), A.kp_stride);
  const size_t ko = (size_t)b * A.kp_stride;
  const int per = (n + GUIDE_THREADS - 1) / GUIDE_THREADS;
  const int i0 = min(n, tid * per), i1 = min(n, i0 + per);
  int cnt = 0;
  for (int i = i0; i < i1; i++) cnt += A.keep[ko + i];
  int total;
  int o = block_excl_scan1024(cnt, s_w, &total);
  for (int i = i0; i < i1; i++) {
    if (!A.keep[ko + i]) continue;
    A.kps_out[ko + o] = A.kps_in[ko + i];
    if (A.desc_in) {
      const uint4 *src = reinterpret_cast<const uint4 *>(A.desc_in + (ko + i) * 32);
      uint4 *dst = reinterpret_cast<uint4 *>(A.desc_out + (ko + o) * 32);
      dst[0] = src[0]; dst[1] = src[1];
    }
    o++;
  }
  if (tid == 0) A.n_out[b] = total;
}

// genEdgesPC: raster-order compaction of the contour pixels into the "sign" ([10,150)) and "free" (>= 150) point lists.
// A thread owns a contiguous run of the raster, two block scans give its output offsets.
__global__ __launch_bounds__(GUIDE_THREADS) void k_bird_edges(fb_bird_guidance_args A) {
  __shared__ int s_w[GUIDE_THREADS / 64];
  const int b = blockIdx.x, tid = threadIdx.x;
  const uint8_t *icp = A.contour + (size_t)b * A.rows * A.pitch;
  const int total = A.rows * A.cols;
  const int per = (total + GUIDE_THREADS - 1) / GUIDE_THREADS;
  const int p0 = min(total, tid * per), p1 = min(total, p0 + per);
  int ns = 0, nf = 0;
  {
    int row = p0 / A.cols, col = p0 - row * A.cols;
    for (int p = p0; p < p1; p++) {
      const uint8_t v = icp[(size_t)row * A.pitch + col];
      ns += v >= 10 && v < 150;
      nf += v >= 150;
      if (++col == A.cols) { col = 0; row++; }
    }
  }
  int ts, tf;
  int os = block_excl_scan1024(ns, s_w, &ts);
  int of = block_excl_scan1024(nf, s_w, &tf);
  float *es = A.edge_sign + (size_t)b * A.edge_cap * 2, *ef = A.edge_free + (size_t)b * A.edge_cap * 2;
  {
    int row = p0 / A.cols, col = p0 - row * A.cols;
    for (int p = p0; p < p1; p++) {
      const uint8_t v = icp[(size_t)row * A.pitch + col];
      if (v >= 150) { if (of < A.edge_cap) { ef[of * 2] = (float)col; ef[of * 2 + 1] = (float)row; } of++; }
      else if (v >= 10) { if (os < A.edge_cap) { es[os * 2] = (float)col; es[os * 2 + 1] = (float)row; } os++; }
      if (++col == A.cols) { col = 0; row++; }
    }
  }
  if (tid == 0) { A.n_edge_sign[b] = ts; A.n_edge_free[b] = tf; }
}

}  // namespace

extern "C" {

int fb_bird_guidance_dev(const fb_bird_guidance_args *A, void *stream) {
  FB_TRY(fb::check_device());
  FB_ARG(A && A->batch >= 0 && A->kp_stride >= 0 && A->cols > 0 && A->rows > 0 && A->pitch >= A->cols && A->edge_cap >= 0);
  if (A->batch == 0) return FB_OK;
  FB_ARG(A->contour && A->n_in && A->n_out && (A->kp_stride == 0 || (A->kps_in && A->kps_out)));
  FB_ARG(!A->desc_in || (A->desc_out && ((uintptr_t)A->desc_in % 16 == 0) && ((uintptr_t)A->desc_out % 16 == 0)));
  FB_ARG(A->kps_out != A->kps_in && (!A->desc_in || A->desc_out != A->desc_in));
  FB_ARG(A->edge_cap == 0 || (A->n_edge_sign && A->n_edge_free && A->edge_sign && A->edge_free));
  FB_ARG((size_t)A->rows * A->cols < (size_t)INT_MAX);
  const size_t lds = (size_t)A->kp_stride + 16;
  if (lds > 150 * 1024) { fb::set_error("fb_bird_guidance: kp_stride %d beyond the LDS flag array", A->kp_stride); return FB_ERR_CAPACITY; }
  FB_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(k_bird_guidance), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  fb::ProfScope prof_(fb::P_BIRDCAM, fb::as_stream(stream));
  if (A->keep && A->kp_stride > 0 && A->batch <= 65535) {
    const int aligned4 = (A->pitch % 4 == 0) && ((uintptr_t)A->contour % 4 == 0) ? 1 : 0;
    k_bird_flags<<<dim3((A->kp_stride + 7) / 8, A->batch), 256, 0, fb::as_stream(stream)>>>(*A, aligned4);
    k_bird_compact<<<A->batch, GUIDE_THREADS, 0, fb::as_stream(stream)>>>(*A);
  } else {
    k_bird_guidance<<<A->batch, GUIDE_THREADS, lds, fb::as_stream(stream)>>>(*A);
  }
  FB_HIP(hipGetLastError());
  if (A->edge_cap > 0) {
    k_bird_edges<<<A->batch, GUIDE_THREADS, 0, fb::as_stream(stream)>>>(*A);
    FB_HIP(hipGetLastError());
  }
  return FB_OK;
}

int fb_bird_guidance(const fb_bird_guidance_args *H) {
  FB_TRY(fb::check_device());
  FB_ARG(H && H->batch >= 0 && H->kp_stride >= 0 && H->cols > 0 && H->rows > 0 && H->pitch >= H->cols && H->edge_cap >= 0);
  if (H->batch == 0) return FB_OK;
  FB_ARG(H->contour && H->n_in && H->n_out && (H->kp_stride == 0 || (H->kps_in && H->kps_out)));
  FB_ARG(!H->desc_in || H->desc_out);
  for (int b = 0; b < H->batch; b++) FB_ARG(H->n_in[b] >= 0 && H->n_in[b] <= H->kp_stride);
  fb_bird_guidance_args D = *H;
  const size_t B = H->batch, ks = H->kp_stride, img = (size_t)H->rows * H->pitch, ec = H->edge_cap;
  fb::DevBuf c, m, ni, ki, di, no, ko, dout, kp, ns, nf, es, ef;
  FB_TRY(c.upload(H->contour, B * img)); D.contour = c.as<uint8_t>();
  if (H->mask) { FB_TRY(m.upload(H->mask, B * img)); D.mask = m.as<uint8_t>(); }
  FB_TRY(ni.upload(H->n_in, B * 4)); D.n_in = ni.as<int32_t>();
  FB_TRY(ki.upload(H->kps_in, B * ks * sizeof(fb_keypoint))); D.kps_in = ki.as<fb_keypoint>();
  if (H->desc_in) { FB_TRY(di.upload(H->desc_in, B * ks * 32)); D.desc_in = di.as<uint8_t>(); }
  FB_TRY(no.alloc(B * 4)); D.n_out = no.as<int32_t>();
  // outputs are in/out: entries past n_out keep their previous content
  FB_TRY(ko.upload(H->kps_out, B * ks * sizeof(fb_keypoint))); D.kps_out = ko.as<fb_keypoint>();
  if (H->desc_in) { FB_TRY(dout.upload(H->desc_out, B * ks * 32)); D.desc_out = dout.as<uint8_t>(); }
  if (H->keep) { FB_TRY(kp.upload(H->keep, B * ks)); D.keep = kp.as<uint8_t>(); }
  if (ec) {
    FB_ARG(H->n_edge_sign && H->n_edge_free && H->edge_sign && H->edge_free);
    FB_TRY(ns.alloc(B * 4)); D.n_edge_sign = ns.as<int32_t>();
    FB_TRY(nf.alloc(B * 4)); D.n_edge_free = nf.as<int32_t>();
    FB_TRY(es.upload(H->edge_sign, B * ec * 8)); D.edge_sign = es.as<float>();
    FB_TRY(ef.upload(H->edge_free, B * ec * 8)); D.edge_free = ef.as<float>();
  }
  FB_TRY(fb_bird_guidance_dev(&D, nullptr));
  FB_HIP(hipDeviceSynchronize());
  FB_TRY(no.download(H->n_out, B * 4));
  FB_TRY(ko.download(H->kps_out, B * ks * sizeof(fb_keypoint)));
  if (H->desc_in) FB_TRY(dout.download(H->desc_out, B * ks * 32));
  if (H->keep) FB_TRY(kp.download(H->keep, B * ks));
  if (ec) {
    FB_TRY(ns.download(H->n_edge_sign, B * 4));
    FB_TRY(nf.download(H->n_edge_free, B * 4));
    FB_TRY(es.download(H->edge_sign, B * ec * 8));
    FB_TRY(ef.download(H->edge_free, B * ec * 8));
  }
  return FB_OK;
}

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
  FB_ARG(H->in_view && H->proj && H->level && H->view_cos);
  fb::Stager st;  // one staged upload / download
  st.in((void **)&D.Tcw, H->Tcw, B * 48); st.in((void **)&D.Ow, H->Ow, B * 12); st.in((void **)&D.n_mp, H->n_mp, B * 4);
  st.in((void **)&D.mp_valid, H->mp_valid, B * ms); st.in((void **)&D.mp_xw, H->mp_xw, B * ms * 12);
  st.in((void **)&D.mp_normal, H->mp_normal, B * ms * 12); st.in((void **)&D.mp_max_dist, H->mp_max_dist, B * ms * 4);
  st.in((void **)&D.mp_min_dist, H->mp_min_dist, B * ms * 4);
  // outputs are in/out (entries that are not in view keep their previous content)
  st.out((void **)&D.in_view, H->in_view, B * ms, true); st.out((void **)&D.proj, H->proj, B * ms * 8, true);
  st.out((void **)&D.proj_xr, H->proj_xr, B * ms * 4, true); st.out((void **)&D.level, H->level, B * ms * 4, true);
  st.out((void **)&D.view_cos, H->view_cos, B * ms * 4, true);
  FB_TRY(st.commit(nullptr));
  FB_TRY(fb_in_frustum_dev(&D, nullptr));
  return st.fetch(nullptr);
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
  FB_ARG(H->keep && H->pt_world);
  fb::Stager st;
  st.in((void **)&D.n_matches, H->n_matches, B * 4); st.in((void **)&D.query_idx, H->query_idx, B * ms * 4);
  st.in((void **)&D.train_idx, H->train_idx, B * ms * 4); st.in((void **)&D.cam_xyz1, H->cam_xyz1, B * s1 * 12);
  st.in((void **)&D.cam_xyz2, H->cam_xyz2, B * s2 * 12); st.in((void **)&D.Tcw1, H->Tcw1, B * 48); st.in((void **)&D.Tcw2, H->Tcw2, B * 48);
  st.in((void **)&D.occupied2, H->occupied2, B * s2);
  st.out((void **)&D.keep, H->keep, B * ms, true); st.out((void **)&D.pt_world, H->pt_world, B * ms * 12, true);
  FB_TRY(st.commit(nullptr));
  FB_TRY(fb_bird_filter_matches_dev(&D, nullptr));
  return st.fetch(nullptr);
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
