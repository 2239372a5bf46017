/*
 * track_oracle.cpp -- CPU restatement of the per-frame tracking chain (TEST INFRASTRUCTURE ONLY; see orb_oracle.cpp
 * header for who may call it).  Parity unpinned: the reference holds no fixture for this path, and the OpenCV pieces
 * inside the steps are restated.
 *
 * A host-side Frame (orc_frame) with the members the chain touches, and the sequence the reference runs per tracked
 * frame, serially and in the reference's loop order:
 *   Frame::Frame(...)                  /root/reference/src/Frame.cc:262-379
 *   Tracking::TrackWithMotionModel     /root/reference/src/Tracking.cc:1312-1385
 *   Tracking::TrackLocalMap            :1387-1441 with GetPerFrameMatchedBirdPoints :2724-2733,
 *                                      FilterBirdOutlierInFront :1825-1914, SearchLocalPoints :1947-1997,
 *                                      GetLocalMapForBird :1999-2012
 *   end of Tracking::Track             :690-701 (clean VO matches), :721-725 (drop outliers)
 *   Tracking::TrackUsingBird           :2014-2061
 *   Tracking::TrackReferenceKeyFrame   :1180-1244 with Frame::ComputeBoW (Frame.cc:628-635); the reference key frame is a
 *                                      frame object (a KeyFrame is built from one, KeyFrame.cc:32-91)
 * mvpMapPoints / mvpMapPointsBird are indices into the caller's map tables (-1 = NULL), like the product's fb_frame.
 * The matchers / extractor / optimiser called in between are the other files of this oracle.
 */
#include <chrono>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <vector>

#include "../include/fishbird.h"

extern "C" {
int orc_orb_tables(const fb_orb_params *p, fb_orb_tables *out);
int orc_orb_extract(const fb_orb_params *p, const uint8_t *img, int w, int h, int stride, fb_keypoint *kps, uint8_t *desc, int32_t *n);
int orc_undistort_keypoints(const fb_keypoint *kps, int n, const float *K4, const float *D4, fb_keypoint *out);
int orc_image_bounds(int cols, int rows, const float *K4, const float *D4, float *bounds);
int orc_bird_guidance(const fb_bird_guidance_args *A);
int orc_bird_keys_to_cam(const fb_keypoint *kps, const int32_t *n, int batch, int kp_stride, int bird_cols, int bird_rows,
                         double pixel2meter, double rear_axle_to_center, const float *Tcb12, float *cam_xyz);
int orc_grid_build(const fb_keypoint *kps, const int32_t *n, int batch, int kp_stride, const fb_grid_geom *g, int32_t *cs, int32_t *ci);
int orc_match_bird_mappoints(const fb_bird_mp_args *A);
int orc_match_projection_frame(const fb_proj_frame_args *A);
int orc_match_projection_points(const fb_proj_points_args *A);
int orc_match_birdview(const fb_birdview_args *A);
int orc_in_frustum(const fb_frustum_args *A);
int orc_pose_opt(const fb_pose_opt_args *A);
int orc_bow_transform(const fb_vocabulary *V, const fb_bow_transform_args *A);
int orc_match_bow(const fb_bow_args *A);
}

struct orc_frame {
  fb_frame_params P;
  int cap, B;
  fb_orb_tables tab;
  fb_grid_geom gF, gB;
  fb_camera cam;
  float logScale;
  std::vector<int32_t> n, nb, mp, mpb, cs, ci, bcs, bci, counts;
  std::vector<fb_keypoint> kps, kps_un, bkps;
  std::vector<uint8_t> desc, bdesc, outlier, boutlier;
  std::vector<float> bcam, Tcw;
  // mBowVec / mFeatVec (Frame.h:128-129), empty until ComputeBoW
  int min_inliers = 30;   // threshold of the last end-of-Track clean-up
  bool bow_done = false;
  std::vector<int32_t> bow_nw, fv_nn, fv_start, fv_items;
  std::vector<uint32_t> bow_ids, fv_ids;
  std::vector<double> bow_vals;
  double stage_s[8];  // seconds inside each stage of the last orc_frame_extract / orc_frame_track call
};

namespace {

double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

int *cnt(orc_frame *f, int slot) { return f->counts.data() + (size_t)slot * f->B; }

// Edge construction of PoseOptimizationWithBird (Optimizer.cc:525-602) from the frame's members, then the optimiser
void pose_optimization(orc_frame *f, const fb_map_points *map, const fb_map_points_bird *mpb, int mode, float wB, float wF, int slot,
                       bool front = true) {
  const size_t B = f->B, cap = f->cap;
  std::vector<float> fxw(B * cap * 3), fobs(B * cap * 2), finf(B * cap), bxw(B * cap * 3), bxc(B * cap * 3), binf(B * cap);
  std::vector<uint8_t> fv(B * cap, 0), bv(B * cap, 0);
  for (size_t b = 0; b < B; b++) {
    for (int i = 0; i < f->n[b]; i++) {
      const size_t o = b * cap + i;
      const int id = front ? f->mp[o] : -1;  // BirdOptimization builds no front edges
      if (id < 0) continue;
      const fb_keypoint &kpUn = f->kps_un[o];
      fv[o] = 1;
      for (int k = 0; k < 3; k++) fxw[o * 3 + k] = map->xw[(b * map->stride + id) * 3 + k];
      fobs[o * 2] = kpUn.x; fobs[o * 2 + 1] = kpUn.y;
      finf[o] = f->tab.inv_level_sigma2[kpUn.octave];
    }
    for (int i = 0; i < f->nb[b]; i++) {
      const size_t o = b * cap + i;
      const int id = f->mpb[o];
      if (id < 0) continue;
      bv[o] = 1;
      for (int k = 0; k < 3; k++) { bxw[o * 3 + k] = mpb->xw[(b * mpb->stride + id) * 3 + k]; bxc[o * 3 + k] = f->bcam[o * 3 + k]; }
      binf[o] = f->tab.inv_level_sigma2[f->bkps[o].octave];
    }
  }
  fb_pose_opt_args A;
  std::memset(&A, 0, sizeof(A));
  A.batch = f->B; A.mode = mode; A.front_stride = f->cap; A.bird_stride = f->cap;
  A.fx = f->P.K[0]; A.fy = f->P.K[1]; A.cx = f->P.K[2]; A.cy = f->P.K[3]; A.wF = wF; A.wB = wB;
  A.n_front = f->n.data(); A.front_xw = fxw.data(); A.front_obs = fobs.data(); A.front_inv_sigma2 = finf.data(); A.front_valid = fv.data();
  A.n_bird = f->nb.data(); A.bird_xw = bxw.data(); A.bird_xc = bxc.data(); A.bird_inv_sigma2 = binf.data(); A.bird_valid = bv.data();
  A.bird_outlier = f->boutlier.data(); A.Tcw = f->Tcw.data(); A.front_outlier = f->outlier.data(); A.ninliers = cnt(f, slot);
  orc_pose_opt(&A);
}

}  // namespace

extern "C" {

int orc_frame_create(const fb_frame_params *p, orc_frame **out) {
  orc_frame *f = new orc_frame();
  f->P = *p;
  f->B = p->batch;
  f->cap = p->orb.nfeatures + 8 * p->orb.nlevels;
  orc_orb_tables(&p->orb, &f->tab);
  f->logScale = (float)std::log((double)p->orb.scale_factor);  // Frame.cc:301
  float bounds[4];
  orc_image_bounds(p->front_width, p->front_height, p->K, p->D, bounds);  // Frame.cc:271-277
  f->gF.min_x = bounds[0]; f->gF.min_y = bounds[2];
  f->gF.inv_w = 64.0f / (bounds[1] - bounds[0]);
  f->gF.inv_h = 48.0f / (bounds[3] - bounds[2]);
  f->gF.cols = 64; f->gF.rows = 48;
  f->gB.min_x = 0.f; f->gB.min_y = 0.f;
  f->gB.inv_w = 32.0f / (float)p->bird_width;  // Frame.cc:282-283
  f->gB.inv_h = 32.0f / (float)p->bird_height;
  f->gB.cols = 32; f->gB.rows = 32;
  f->cam.fx = p->K[0]; f->cam.fy = p->K[1]; f->cam.cx = p->K[2]; f->cam.cy = p->K[3];
  f->cam.min_x = bounds[0]; f->cam.max_x = bounds[1]; f->cam.min_y = bounds[2]; f->cam.max_y = bounds[3];
  const size_t B = f->B, cap = f->cap;
  f->n.assign(B, 0); f->nb.assign(B, 0);
  f->mp.assign(B * cap, -1); f->mpb.assign(B * cap, -1);
  f->cs.assign(B * (64 * 48 + 1), 0); f->ci.assign(B * cap, 0); f->bcs.assign(B * (32 * 32 + 1), 0); f->bci.assign(B * cap, 0);
  f->counts.assign(B * FB_CNT_COUNT, 0);
  f->kps.assign(B * cap, fb_keypoint{}); f->kps_un.assign(B * cap, fb_keypoint{}); f->bkps.assign(B * cap, fb_keypoint{});
  f->desc.assign(B * cap * 32, 0); f->bdesc.assign(B * cap * 32, 0); f->outlier.assign(B * cap, 0); f->boutlier.assign(B * cap, 1);
  f->bcam.assign(B * cap * 3, 0.f); f->Tcw.assign(B * 12, 0.f);
  std::memset(f->stage_s, 0, sizeof(f->stage_s));
  *out = f;
  return FB_OK;
}

void orc_frame_destroy(orc_frame *f) { delete f; }

// Frame::Frame (Frame.cc:262-379), host images packed batch after batch
int orc_frame_extract(orc_frame *f, const uint8_t *front, int front_stride, const uint8_t *bird, int bird_stride,
                      const uint8_t *contour, const uint8_t *mask) {
  const size_t B = f->B, cap = f->cap;
  const size_t fbytes = (size_t)front_stride * f->P.front_height, bbytes = (size_t)bird_stride * f->P.bird_height;
  double t0 = now();
  for (size_t b = 0; b < B; b++)  // ExtractORB(0, imGray), :310
    orc_orb_extract(&f->P.orb, front + b * fbytes, f->P.front_width, f->P.front_height, front_stride, f->kps.data() + b * cap,
                    f->desc.data() + b * cap * 32, &f->n[b]);
  double t1 = now();
  f->stage_s[0] = t1 - t0;
  for (size_t b = 0; b < B; b++)  // UndistortKeyPoints, :320
    orc_undistort_keypoints(f->kps.data() + b * cap, f->n[b], f->P.K, f->P.D, f->kps_un.data() + b * cap);
  std::fill(f->mp.begin(), f->mp.end(), -1);           // :327
  std::fill(f->outlier.begin(), f->outlier.end(), 0);   // :328
  t0 = now();
  std::vector<fb_keypoint> pre(B * cap);
  std::vector<uint8_t> pred(B * cap * 32);
  std::vector<int32_t> npre(B, 0);
  fb_orb_params pb = f->P.orb;
  if (f->P.bird_nfeatures > 0) pb.nfeatures = f->P.bird_nfeatures;
  for (size_t b = 0; b < B; b++)  // the E9 substitution: ORBextractor on the bird image (:337-339, :355)
    orc_orb_extract(&pb, bird + b * bbytes, f->P.bird_width, f->P.bird_height, bird_stride, pre.data() + b * cap,
                    pred.data() + b * cap * 32, &npre[b]);
  t1 = now();
  f->stage_s[1] = t1 - t0;
  if (contour) {  // GuidenceKeyBirdPts(preKeysBird), :342 (+ the detect mask, :339)
    fb_bird_guidance_args G;
    std::memset(&G, 0, sizeof(G));
    G.batch = f->B; G.kp_stride = f->cap; G.cols = f->P.bird_width; G.rows = f->P.bird_height; G.pitch = bird_stride;
    G.contour = contour; G.mask = mask; G.n_in = npre.data(); G.kps_in = pre.data(); G.desc_in = pred.data();
    G.n_out = f->nb.data(); G.kps_out = f->bkps.data(); G.desc_out = f->bdesc.data();
    orc_bird_guidance(&G);
  } else {
    f->nb = npre;
    f->bkps = pre;
    f->bdesc = pred;
  }
  std::fill(f->mpb.begin(), f->mpb.end(), -1);            // :355
  std::fill(f->boutlier.begin(), f->boutlier.end(), 1);    // :356  mvBirdOutlier = vector<bool>(Nbird, true)
  orc_bird_keys_to_cam(f->bkps.data(), f->nb.data(), f->B, f->cap, f->P.bird_width, f->P.bird_height, f->P.pixel2meter,
                       f->P.rear_axle_to_center, f->P.Tcb, f->bcam.data());  // :365-373
  orc_grid_build(f->kps_un.data(), f->n.data(), f->B, f->cap, &f->gF, f->cs.data(), f->ci.data());  // AssignFeaturesToGrid, :376
  orc_grid_build(f->bkps.data(), f->nb.data(), f->B, f->cap, &f->gB, f->bcs.data(), f->bci.data());
  std::fill(f->counts.begin(), f->counts.end(), 0);
  f->bow_done = false;
  return FB_OK;
}

// Frame(const Frame &) (Frame.cc:49-82) / the member copies of KeyFrame::KeyFrame(Frame &F, ...) (KeyFrame.cc:32-91)
int orc_frame_copy(orc_frame *dst, const orc_frame *src) {
  *dst = *src;
  return FB_OK;
}

int orc_frame_set_pose(orc_frame *f, const float *Tcw) {
  std::memcpy(f->Tcw.data(), Tcw, (size_t)f->B * 48);
  return FB_OK;
}

int orc_frame_set_map_points(orc_frame *f, const int32_t *mp, const int32_t *mpb) {
  const size_t B = f->B, cap = f->cap;
  for (size_t b = 0; b < B; b++)
    for (size_t i = 0; i < cap; i++) {
      if (mp) { f->mp[b * cap + i] = (int)i < f->n[b] ? mp[b * cap + i] : -1; f->outlier[b * cap + i] = 0; }
      if (mpb) f->mpb[b * cap + i] = (int)i < f->nb[b] ? mpb[b * cap + i] : -1;
    }
  return FB_OK;
}

}  // extern "C"

namespace {

// mCurrentFrame.SetPose(detlaT * src.mTcw), Tracking.cc:1320 / :1186 (4x4 * 4x4 CV_32F, small-matrix gemm path)
void set_predicted_pose(orc_frame *cur, const orc_frame *src, const float *delta) {
  const size_t B = cur->B;
  for (size_t b = 0; b < B; b++) {
    const float *D = delta + b * 12, *L = src->Tcw.data() + b * 12;
    float *C = cur->Tcw.data() + b * 12;
    for (int r = 0; r < 3; r++)
      for (int c = 0; c < 4; c++) {
        float s = (D[r * 4 + 0] * L[0 * 4 + c] + D[r * 4 + 1] * L[1 * 4 + c]) + D[r * 4 + 2] * L[2 * 4 + c];
        s = s + D[r * 4 + 3] * (c == 3 ? 1.0f : 0.0f);
        C[r * 4 + c] = s;
      }
  }
}

// GetLocalMapForBird, Tracking.cc:1999-2012
void local_map_for_bird(orc_frame *cur, const fb_track_args *T, const fb_map_points_bird *mpb) {
  const size_t B = cur->B, cap = cur->cap;
  {
    const int lcap = T->d_local_mpb ? cur->P.local_mpb_cap : mpb->stride;
    std::vector<uint8_t> valid(B * lcap, 0), rdesc(B * (size_t)lcap * 32, 0);
    std::vector<float> rxw(B * (size_t)lcap * 3, 0.f);
    std::vector<int32_t> nref(B, 0), match(B * cap, -1);
    for (size_t b = 0; b < B; b++) {
      const int nl = T->d_local_mpb ? T->d_n_local_mpb[b] : mpb->n[b];
      nref[b] = nl > 10 ? nl : 0;  // if (vlocalMPB.size() > 10)
      for (int j = 0; j < nl; j++) {
        const int id = T->d_local_mpb ? T->d_local_mpb[b * lcap + j] : j;
        if (id < 0 || id >= mpb->n[b]) continue;
        valid[b * lcap + j] = 1;
        std::memcpy(&rxw[(b * lcap + j) * 3], mpb->xw + (b * mpb->stride + id) * 3, 12);
        std::memcpy(&rdesc[(b * lcap + j) * 32], mpb->desc + (b * mpb->stride + id) * 32, 32);
      }
    }
    fb_bird_mp_args A;
    std::memset(&A, 0, sizeof(A));
    A.batch = cur->B; A.cur_stride = cur->cap; A.ref_stride = lcap;
    A.n_cur = cur->nb.data(); A.cur_kps = cur->bkps.data(); A.cur_desc = cur->bdesc.data(); A.cur_cam_xyz = cur->bcam.data();
    A.cur_cell_start = cur->bcs.data(); A.cur_cell_items = cur->bci.data(); A.cur_Tcw = cur->Tcw.data();
    A.n_ref = nref.data(); A.ref_valid = valid.data(); A.ref_xw = rxw.data(); A.ref_desc = rdesc.data();
    std::memcpy(A.Tbc, cur->P.Tbc, sizeof(A.Tbc));
    A.bird_cols = cur->P.bird_width; A.bird_rows = cur->P.bird_height; A.meter2pixel = cur->P.meter2pixel;
    A.rear_axle_to_center = cur->P.rear_axle_to_center; A.grid = cur->gB; A.window_size = 10; A.filter_size = 0.05f;
    A.matcher.nnratio = 0.9f; A.matcher.check_orientation = 1;
    A.match_cur_to_ref = match.data(); A.ninliers = cnt(cur, FB_CNT_BIRD_KF_MATCHES);
    orc_match_bird_mappoints(&A);
    for (size_t b = 0; b < B; b++)  // CurF.mvpMapPointsBird[match] = pMPBird, ORBmatcher.cc:1891
      for (int i = 0; i < cur->nb[b]; i++) {
        const int m = match[b * cap + i];
        if (m >= 0) cur->mpb[b * cap + i] = T->d_local_mpb ? T->d_local_mpb[b * lcap + m] : m;
      }
  }
}

// fill(mvpMapPoints, NULL); SearchByProjection(cur, last, 15, mono), Tracking.cc:1330-1339
void search_by_projection_last(orc_frame *cur, const orc_frame *last, const fb_map_points *map) {
  const size_t B = cur->B, cap = cur->cap;
  {
    std::fill(cur->mp.begin(), cur->mp.end(), -1);
    std::vector<uint8_t> valid(B * cap, 0), obs(B * cap, 0), ldesc(B * cap * 32, 0);
    std::vector<float> lxw(B * cap * 3, 0.f), lang(B * cap, 0.f);
    std::vector<int32_t> loct(B * cap, 0), match(B * cap, -1);
    for (size_t b = 0; b < B; b++)
      for (int i = 0; i < last->n[b]; i++) {
        const size_t o = b * cap + i;
        const int id = last->mp[o];
        if (id < 0 || last->outlier[o]) continue;  // pMP && !LastFrame.mvbOutlier[i], ORBmatcher.cc:1357-1359
        valid[o] = 1;
        obs[o] = map->obs_pos[b * map->stride + id];
        std::memcpy(&lxw[o * 3], map->xw + (b * map->stride + id) * 3, 12);
        std::memcpy(&ldesc[o * 32], map->desc + (b * map->stride + id) * 32, 32);
        loct[o] = last->kps[o].octave;
        lang[o] = last->kps_un[o].angle;
      }
    fb_proj_frame_args A;
    std::memset(&A, 0, sizeof(A));
    A.batch = cur->B; A.cur_stride = cur->cap; A.last_stride = cur->cap;
    A.n_cur = cur->n.data(); A.cur_kps = cur->kps_un.data(); A.cur_desc = cur->desc.data();
    A.cur_cell_start = cur->cs.data(); A.cur_cell_items = cur->ci.data(); A.cur_blocked = nullptr; A.cur_Tcw = cur->Tcw.data();
    A.n_last = last->n.data(); A.last_valid = valid.data(); A.last_obs_pos = obs.data(); A.last_xw = lxw.data();
    A.last_desc = ldesc.data(); A.last_octave = loct.data(); A.last_angle = lang.data();
    A.cam = cur->cam; A.grid = cur->gF;
    for (int i = 0; i < FB_MAX_LEVELS; i++) A.scale_factors[i] = cur->tab.scale_factor[i];
    A.th = 15.0f; A.matcher.nnratio = 0.9f; A.matcher.check_orientation = 1;
    A.match_cur_to_last = match.data(); A.nmatches = cnt(cur, FB_CNT_PROJ_MATCHES);
    orc_match_projection_frame(&A);
    {  // if (nmatches < 20) { fill(mvpMapPoints, NULL); nmatches = SearchByProjection(cur, last, 2 * th, mono); }, :1342-1349
      std::vector<int32_t> n2(B, 0), match2(B * cap, -1), cnt2(B, 0);
      bool any = false;
      for (size_t b = 0; b < B; b++)
        if (cnt(cur, FB_CNT_PROJ_MATCHES)[b] < 20) { n2[b] = last->n[b]; any = true; cnt(cur, FB_CNT_PROJ_RETRIED)[b] = 1; }   // the other sequences: no queries
      if (any) {
        A.n_last = n2.data(); A.th = 30.0f; A.match_cur_to_last = match2.data(); A.nmatches = cnt2.data();
        orc_match_projection_frame(&A);
        for (size_t b = 0; b < B; b++)
          if (cnt(cur, FB_CNT_PROJ_MATCHES)[b] < 20) {
            cnt(cur, FB_CNT_PROJ_MATCHES)[b] = cnt2[b];
            std::memcpy(&match[b * cap], &match2[b * cap], cap * 4);
          }
      }
    }
    for (size_t b = 0; b < B; b++)  // CurrentFrame.mvpMapPoints[bestIdx2] = pMP, ORBmatcher.cc:1430
      for (int i = 0; i < cur->n[b]; i++) {
        const int m = match[b * cap + i];
        cur->mp[b * cap + i] = m >= 0 ? last->mp[b * cap + m] : -1;
      }
  }
}

// the "Discard outliers" loop of TrackWithMotionModel (Tracking.cc:1358-1376) and TrackReferenceKeyFrame (:1221-1241);
// nmatches starts from the matcher's return value (src_slot).  gate: sequences whose gate_slot counter is below gate_min
// returned before the loop (TrackReferenceKeyFrame :1209-1210)
void discard_outliers(orc_frame *cur, const fb_map_points *map, int src_slot, bool gate = false, int gate_slot = 0, int gate_min = 0) {
  const size_t B = cur->B, cap = cur->cap;
  for (size_t b = 0; b < B; b++) {
    if (gate && cnt(cur, gate_slot)[b] < gate_min) continue;
    int nmatches = cnt(cur, src_slot)[b], nmatchesMap = 0;
    for (int i = 0; i < cur->n[b]; i++) {
      const size_t o = b * cap + i;
      if (cur->mp[o] < 0) continue;
      if (cur->outlier[o]) { cur->mp[o] = -1; cur->outlier[o] = 0; nmatches--; }
      else if (map->obs_pos[b * map->stride + cur->mp[o]]) nmatchesMap++;
    }
    cnt(cur, FB_CNT_MATCHES)[b] = nmatches;
    cnt(cur, FB_CNT_MATCHES_MAP)[b] = nmatchesMap;
  }
}

// GetPerFrameMatchedBirdPoints (Tracking.cc:2724-2733) with tmpRefFrame = last; only_below > 0: only for the sequences
// whose frame holds fewer bird map points than that (TrackReferenceKeyFrame, :1196-1200)
void per_frame_matched_bird_points(orc_frame *cur, orc_frame *last, fb_map_points_bird *mpb, int only_below) {
  const size_t B = cur->B, cap = cur->cap;
  {
    std::vector<int32_t> m12(B * cap, -1), mdist(B * cap, 0), ndm(B, 0);
    fb_birdview_args A;
    std::memset(&A, 0, sizeof(A));
    A.batch = cur->B; A.cur_stride = cur->cap; A.ref_stride = cur->cap;
    A.n_cur = cur->nb.data(); A.cur_kps = cur->bkps.data(); A.cur_desc = cur->bdesc.data();
    A.cur_cell_start = cur->bcs.data(); A.cur_cell_items = cur->bci.data();
    A.n_ref = last->nb.data(); A.ref_kps = last->bkps.data(); A.ref_desc = last->bdesc.data();
    A.grid = cur->gB; A.window_size = 10; A.matcher.nnratio = 0.9f; A.matcher.check_orientation = 1;
    A.match_ref_to_cur = m12.data(); A.match_dist = mdist.data(); A.nmatches = cnt(cur, FB_CNT_BIRDVIEW_MATCHES); A.n_dmatches = ndm.data();
    orc_match_birdview(&A);
    // FilterBirdOutlierInFront(tmpRefFrame, &mCurrentFrame, vDMatches12, 0.05), :1825-1914
    for (size_t b = 0; b < B; b++) {
      if (only_below > 0) {  // int numPt = GetBirdMapPointsNum(); if (numPt < 10), Tracking.cc:1196-1200 (Frame.cc:1081-1092)
        int numPt = 0;
        for (int i = 0; i < cur->nb[b]; i++) numPt += cur->mpb[b * cap + i] >= 0;
        cnt(cur, FB_CNT_BIRD_POINTS)[b] = numPt;
        if (numPt >= only_below) { cnt(cur, FB_CNT_BIRDVIEW_MATCHES)[b] = 0; continue; }  // the matcher did not run for this sequence
      }
      const float *T1 = last->Tcw.data() + b * 12, *T2 = cur->Tcw.data() + b * 12;
      float Twc1[12];  // Converter::invT(Tcw1), Converter.cc:176-187
      for (int r = 0; r < 3; r++) {
        for (int c = 0; c < 3; c++) Twc1[r * 4 + c] = T1[c * 4 + r];
        Twc1[r * 4 + 3] = -((T1[0 * 4 + r] * T1[3] + T1[1 * 4 + r] * T1[7]) + T1[2 * 4 + r] * T1[11]);
      }
      int inlier = 0, buildNew = 0;
      std::vector<uint8_t> taken(cap, 0);  // a slot a passing match took stays taken even when the bird table is full
      for (int i1 = 0; i1 < last->nb[b]; i1++) {        // vDMatches12: DMatch(i1, vnMatches12[i1]) iff vnMatches12[i1] > 0
        const int t = m12[b * cap + i1];
        if (!(t > 0)) continue;
        if (cur->mpb[b * cap + t] >= 0 || taken[t]) continue;  // tmpMP, :1861-1863
        const float *p1 = last->bcam.data() + (b * cap + i1) * 3, *p2 = cur->bcam.data() + (b * cap + t) * 3;
        float ptw[3], pc2[3];
        for (int r = 0; r < 3; r++) ptw[r] = ((Twc1[r * 4] * p1[0] + Twc1[r * 4 + 1] * p1[1]) + Twc1[r * 4 + 2] * p1[2]) + Twc1[r * 4 + 3];
        for (int r = 0; r < 3; r++) pc2[r] = ((T2[r * 4] * ptw[0] + T2[r * 4 + 1] * ptw[1]) + T2[r * 4 + 2] * ptw[2]) + T2[r * 4 + 3];
        const float d[3] = {pc2[0] - p2[0], pc2[1] - p2[1], pc2[2] - p2[2]};
        const double disC = std::sqrt((double)d[0] * d[0] + (double)d[1] * d[1] + (double)d[2] * d[2]);  // cv::norm, :1874
        if (disC < 0.05f) {
          cur->boutlier[b * cap + t] = 0;                 // :1887
          taken[t] = 1;
          inlier++;
          const int refId = last->mpb[b * cap + i1];
          if (refId >= 0) {
            cur->mpb[b * cap + t] = refId;                // :1893
          } else {
            const int id = mpb->n[b];
            if (id < mpb->stride) {                       // new MapPointBird(ptwC, MatchedFrame2, mpMap, trainIdx), :1897
              std::memcpy(mpb->xw + (b * mpb->stride + id) * 3, ptw, 12);
              std::memcpy(mpb->desc + (b * mpb->stride + id) * 32, cur->bdesc.data() + (b * cap + t) * 32, 32);
              mpb->n[b] = id + 1;
              cur->mpb[b * cap + t] = id;
              last->mpb[b * cap + i1] = id;
            }
            buildNew++;
          }
        }
      }
      cnt(cur, FB_CNT_BIRD_INLIERS)[b] = inlier;
      cnt(cur, FB_CNT_BIRD_NEW)[b] = buildNew;
      int numPt = 0;  // GetBirdMapPointsNum() as BirdNeedKF reads it afterwards (Tracking.cc:2067)
      for (int i = 0; i < cur->nb[b]; i++) numPt += cur->mpb[b * cap + i] >= 0;
      cnt(cur, FB_CNT_BIRD_POINTS_FINAL)[b] = numPt;
    }
  }
}

// SearchLocalPoints, Tracking.cc:1947-1997
void search_local_points(orc_frame *cur, const fb_track_args *T, const fb_map_points *map) {
  const size_t B = cur->B, cap = cur->cap;
  {
    const int lcap = T->d_local_mp ? cur->P.local_mp_cap : map->stride;
    std::vector<uint8_t> seen(B * (size_t)map->stride, 0), blocked(B * cap, 0), lvalid(B * (size_t)lcap, 0), inview(B * (size_t)lcap, 0),
        lobs(B * (size_t)lcap, 0), ldesc(B * (size_t)lcap * 32, 0);
    std::vector<float> lxw(B * (size_t)lcap * 3, 0.f), lnrm(B * (size_t)lcap * 3, 0.f), lmax(B * (size_t)lcap, 0.f), lmin(B * (size_t)lcap, 0.f),
        proj(B * (size_t)lcap * 2, 0.f), vcos(B * (size_t)lcap, 0.f), Ow(B * 3, 0.f);
    std::vector<int32_t> level(B * (size_t)lcap, 0), nl(B, 0), match(B * cap, -1);
    for (size_t b = 0; b < B; b++) {
      for (int i = 0; i < cur->n[b]; i++) {  // :1950-1966
        const size_t o = b * cap + i;
        const int id = cur->mp[o];
        if (id < 0) continue;
        if (map->bad[b * map->stride + id]) cur->mp[o] = -1;
        else { seen[b * map->stride + id] = 1; blocked[o] = map->obs_pos[b * map->stride + id]; }
      }
      nl[b] = T->d_local_mp ? T->d_n_local_mp[b] : map->n[b];
      for (int j = 0; j < nl[b]; j++) {      // :1971-1984
        const int id = T->d_local_mp ? T->d_local_mp[b * lcap + j] : j;
        if (id < 0 || id >= map->n[b]) continue;
        const size_t m = b * map->stride + id, o = b * lcap + j;
        if (seen[m] || map->bad[m]) continue;
        lvalid[o] = 1;
        std::memcpy(&lxw[o * 3], map->xw + m * 3, 12);
        std::memcpy(&lnrm[o * 3], map->normal + m * 3, 12);
        lmax[o] = map->max_dist[m]; lmin[o] = map->min_dist[m];
        lobs[o] = map->obs_pos[m];
        std::memcpy(&ldesc[o * 32], map->desc + m * 32, 32);
      }
      const float *Tc = cur->Tcw.data() + b * 12;  // mOw = -mRcw.t() * mtcw (Frame.cc:432): general gemm path, double accumulation
      for (int r = 0; r < 3; r++) {
        double s = 0.0;
        for (int k = 0; k < 3; k++) s += (double)Tc[k * 4 + r] * (double)Tc[k * 4 + 3];
        Ow[b * 3 + r] = (float)(-s);
      }
    }
    fb_frustum_args F;
    std::memset(&F, 0, sizeof(F));
    F.batch = cur->B; F.mp_stride = lcap; F.Tcw = cur->Tcw.data(); F.Ow = Ow.data(); F.n_mp = nl.data(); F.mp_valid = lvalid.data();
    F.mp_xw = lxw.data(); F.mp_normal = lnrm.data(); F.mp_max_dist = lmax.data(); F.mp_min_dist = lmin.data(); F.cam = cur->cam;
    F.mbf = 0.f; F.viewing_cos_limit = 0.5f; F.log_scale_factor = cur->logScale; F.n_levels = cur->P.orb.nlevels;
    F.in_view = inview.data(); F.proj = proj.data(); F.proj_xr = nullptr; F.level = level.data(); F.view_cos = vcos.data();
    orc_in_frustum(&F);
    for (size_t b = 0; b < B; b++) {
      int nToMatch = 0;
      for (int j = 0; j < nl[b]; j++) nToMatch += inview[b * lcap + j];
      cnt(cur, FB_CNT_TO_MATCH)[b] = nToMatch;
    }
    fb_proj_points_args A;
    std::memset(&A, 0, sizeof(A));
    A.batch = cur->B; A.cur_stride = cur->cap; A.mp_stride = lcap;
    A.n_cur = cur->n.data(); A.cur_kps = cur->kps_un.data(); A.cur_desc = cur->desc.data();
    A.cur_cell_start = cur->cs.data(); A.cur_cell_items = cur->ci.data(); A.cur_blocked = blocked.data();
    A.n_mp = nl.data(); A.mp_track = inview.data(); A.mp_obs_pos = lobs.data(); A.mp_proj = proj.data(); A.mp_level = level.data();
    A.mp_view_cos = vcos.data(); A.mp_desc = ldesc.data(); A.grid = cur->gF;
    for (int i = 0; i < FB_MAX_LEVELS; i++) A.scale_factors[i] = cur->tab.scale_factor[i];
    A.th = 1.0f; A.matcher.nnratio = 0.8f; A.matcher.check_orientation = 1;
    A.match_cur_to_mp = match.data(); A.nmatches = cnt(cur, FB_CNT_LOCAL_MATCHES);
    orc_match_projection_points(&A);  // (with nothing in view it matches nothing: the if (nToMatch > 0) of :1986)
    for (size_t b = 0; b < B; b++)    // F.mvpMapPoints[bestIdx] = pMP, ORBmatcher.cc:124
      for (int i = 0; i < cur->n[b]; i++) {
        const int m = match[b * cap + i];
        if (m >= 0) cur->mp[b * cap + i] = T->d_local_mp ? T->d_local_mp[b * lcap + m] : m;
      }
  }
}

// mnMatchesInliers (Tracking.cc:1411-1424), clean VO matches (:690-701), drop outliers (:721-725)
void finish_frame(orc_frame *cur, const fb_map_points *map, bool keep_outliers = false, int min_inliers = 30) {
  cur->min_inliers = min_inliers;
  const size_t B = cur->B, cap = cur->cap;
  for (size_t b = 0; b < B; b++) {
    int inl = 0;
    for (int i = 0; i < cur->n[b]; i++) {
      const size_t o = b * cap + i;
      if (cur->mp[o] < 0) continue;
      if (!cur->outlier[o] && map->obs_pos[b * map->stride + cur->mp[o]]) inl++;
    }
    cnt(cur, FB_CNT_MATCHES_INLIERS)[b] = inl;
    if (inl < min_inliers) continue;  // TrackLocalMap returned false (:1435-1438): bOK = false, the block of :681-726 is skipped
    for (int i = 0; i < cur->n[b]; i++) {
      const size_t o = b * cap + i;
      if (cur->mp[o] >= 0 && !map->obs_pos[b * map->stride + cur->mp[o]]) { cur->outlier[o] = 0; cur->mp[o] = -1; }
    }
    if (keep_outliers) continue;  // the caller creates its key frame first (:716-718) and drops the outliers afterwards
    for (int i = 0; i < cur->n[b]; i++) {
      const size_t o = b * cap + i;
      if (cur->mp[o] >= 0 && cur->outlier[o]) cur->mp[o] = -1;
    }
  }
}

// TrackLocalMap (Tracking.cc:1387-1441) + the end of Track
void track_local_map(orc_frame *cur, orc_frame *last, const fb_track_args *T, fb_map_points_bird *mpb, bool gated) {
  const fb_map_points *map = &T->map;
  // gated: `if (bOK) bOK = TrackLocalMap()` (Tracking.cc:642) per sequence -- a sequence whose first stage returned false
  // (nmatchesMap < 10) is not touched.  Done here by running the batch and putting such sequences back as they were.
  const size_t B = cur->B, cap = cur->cap;
  std::vector<uint8_t> skip(B, 0);
  orc_frame keepCur, keepLast;
  std::vector<int32_t> keepN(B, 0);
  bool anySkip = false;
  if (gated)
    for (size_t b = 0; b < B; b++)
      if (cnt(cur, FB_CNT_MATCHES_MAP)[b] < 10) { skip[b] = 1; anySkip = true; }
  if (anySkip) {
    keepCur = *cur; keepLast = *last;
    for (size_t b = 0; b < B; b++) keepN[b] = mpb->n[b];
  }
  double t0 = now();
  per_frame_matched_bird_points(cur, last, mpb, 0);                                              // :1392
  double t1 = now();
  cur->stage_s[5] = t1 - t0;
  t0 = t1;
  search_local_points(cur, T, map);                                                              // :1396
  t1 = now();
  cur->stage_s[6] = t1 - t0;
  t0 = t1;
  pose_optimization(cur, map, mpb, FB_POSE_FRONT_BIRD, T->wB, T->wF, FB_CNT_POSE2_INLIERS);      // :1400
  t1 = now();
  cur->stage_s[7] = t1 - t0;
  finish_frame(cur, map, T->defer_outlier_drop != 0, T->min_inliers > 0 ? T->min_inliers : 30);
  if (anySkip)
    for (size_t b = 0; b < B; b++) {
      if (!skip[b]) continue;
      std::copy(keepCur.mp.begin() + b * cap, keepCur.mp.begin() + (b + 1) * cap, cur->mp.begin() + b * cap);
      std::copy(keepCur.mpb.begin() + b * cap, keepCur.mpb.begin() + (b + 1) * cap, cur->mpb.begin() + b * cap);
      std::copy(keepCur.outlier.begin() + b * cap, keepCur.outlier.begin() + (b + 1) * cap, cur->outlier.begin() + b * cap);
      std::copy(keepCur.boutlier.begin() + b * cap, keepCur.boutlier.begin() + (b + 1) * cap, cur->boutlier.begin() + b * cap);
      std::copy(keepCur.Tcw.begin() + b * 12, keepCur.Tcw.begin() + (b + 1) * 12, cur->Tcw.begin() + b * 12);
      std::copy(keepLast.mpb.begin() + b * cap, keepLast.mpb.begin() + (b + 1) * cap, last->mpb.begin() + b * cap);
      for (int slot = 0; slot < FB_CNT_COUNT; slot++) cnt(cur, slot)[b] = keepCur.counts[(size_t)slot * B + b];
      mpb->n[b] = keepN[b];  // the MapPointBirds the skipped TrackLocalMap appended do not exist
    }
}

}  // namespace

extern "C" {

// Tracking::Track for a frame in state OK: TrackWithMotionModel + TrackLocalMap + the clean-up (host pointers in T)
int orc_frame_track_motion_model(orc_frame *cur, orc_frame *last, const fb_track_args *T) {
  const fb_map_points *map = &T->map;
  fb_map_points_bird mpbv = T->mpb;
  std::memset(cur->stage_s, 0, sizeof(cur->stage_s));
  double t0 = now();
  set_predicted_pose(cur, last, T->d_delta);                                                     // :1314-1320
  local_map_for_bird(cur, T, &mpbv);                                                             // :1322-1323
  double t1 = now();
  cur->stage_s[2] = t1 - t0;
  t0 = t1;
  search_by_projection_last(cur, last, map);                                                     // :1330-1339
  t1 = now();
  cur->stage_s[3] = t1 - t0;
  t0 = t1;
  {  // if (nmatches < 20) return false, :1351-1352: those sequences keep the matches, the predicted pose and their flags
    std::vector<int32_t> n_keep = cur->n, nb_keep = cur->nb;
    for (size_t b = 0; b < (size_t)cur->B; b++)
      if (cnt(cur, FB_CNT_PROJ_MATCHES)[b] < 20) { cur->n[b] = 0; cur->nb[b] = 0; }  // no edges: the optimiser leaves the sequence alone
    pose_optimization(cur, map, &mpbv, FB_POSE_FRONT_BIRD, T->wB, T->wF, FB_CNT_POSE1_INLIERS);  // :1353
    cur->n = n_keep; cur->nb = nb_keep;
  }
  t1 = now();
  cur->stage_s[4] = t1 - t0;
  discard_outliers(cur, map, FB_CNT_PROJ_MATCHES, true, FB_CNT_PROJ_MATCHES, 20);                // :1358-1376
  return FB_OK;
}

int orc_frame_track_local_map(orc_frame *cur, orc_frame *last, const fb_track_args *T) {
  fb_map_points_bird mpbv = T->mpb;
  track_local_map(cur, last, T, &mpbv, T->gate_local_map != 0);
  return FB_OK;
}

// Tracking.cc:721-725 on its own (fb_track_args.defer_outlier_drop), for the sequences whose clean-up ran
int orc_frame_drop_outliers(orc_frame *f) {
  const size_t B = f->B, cap = f->cap;
  for (size_t b = 0; b < B; b++) {
    if (cnt(f, FB_CNT_MATCHES_INLIERS)[b] < f->min_inliers) continue;
    for (int i = 0; i < f->n[b]; i++) {
      const size_t o = b * cap + i;
      if (f->mp[o] >= 0 && f->outlier[o]) f->mp[o] = -1;
    }
  }
  return FB_OK;
}

// Frame::ComputeBoW (Frame.cc:628-635): if (mBowVec.empty()) transform(descriptors, mBowVec, mFeatVec, 4)
int orc_frame_compute_bow(orc_frame *f, const fb_vocabulary *voc) {
  if (f->bow_done) return FB_OK;
  const size_t B = f->B, cap = f->cap;
  f->bow_nw.assign(B, 0); f->fv_nn.assign(B, 0); f->bow_ids.assign(B * cap, 0); f->bow_vals.assign(B * cap, 0.0);
  f->fv_ids.assign(B * cap, 0); f->fv_start.assign(B * (cap + 1), 0); f->fv_items.assign(B * cap, 0);
  fb_bow_transform_args A;
  std::memset(&A, 0, sizeof(A));
  A.batch = f->B; A.f_stride = f->cap; A.n_f = f->n.data(); A.desc = f->desc.data(); A.levelsup = 4;
  A.n_words = f->bow_nw.data(); A.bow_ids = f->bow_ids.data(); A.bow_vals = f->bow_vals.data();
  A.fv_n_nodes = f->fv_nn.data(); A.fv_node_ids = f->fv_ids.data(); A.fv_node_start = f->fv_start.data(); A.fv_items = f->fv_items.data();
  const int rc = orc_bow_transform(voc, &A);
  f->bow_done = rc == FB_OK;
  return rc;
}

// Tracking::TrackReferenceKeyFrame (Tracking.cc:1180-1244), bLooseCouple = true (System.cc:32), bHaveBird.
// kf = mpReferenceKF (its BoW computed, KeyFrame.cc:93-102), last = tmpRefFrame, T->d_delta = detlaT of :1185.
int orc_frame_track_reference(orc_frame *cur, orc_frame *kf, orc_frame *last, const fb_vocabulary *voc, const fb_track_args *T) {
  const size_t B = cur->B, cap = cur->cap;
  const fb_map_points *map = &T->map;
  fb_map_points_bird mpbv = T->mpb;
  if (!kf->bow_done) return FB_ERR_ARG;
  set_predicted_pose(cur, kf, T->d_delta);                       // :1185-1186
  local_map_for_bird(cur, T, &mpbv);                             // :1193-1194
  per_frame_matched_bird_points(cur, last, &mpbv, 10);           // :1196-1200
  orc_frame_compute_bow(cur, voc);                               // :1203
  std::vector<uint8_t> has_mp(B * cap, 0);
  std::vector<int32_t> match(B * cap, -1);
  for (size_t b = 0; b < B; b++)
    for (int i = 0; i < kf->n[b]; i++) {                         // pMP && !pMP->isBad(), ORBmatcher.cc:196-203
      const int id = kf->mp[b * cap + i];
      has_mp[b * cap + i] = id >= 0 && !map->bad[b * map->stride + id];
    }
  fb_bow_args A;
  std::memset(&A, 0, sizeof(A));
  A.batch = cur->B; A.kf_stride = cur->cap; A.f_stride = cur->cap;
  A.n_kf = kf->n.data(); A.kf_kps = kf->kps_un.data(); A.kf_desc = kf->desc.data(); A.kf_has_mp = has_mp.data();
  A.kf_fv.node_stride = cur->cap; A.kf_fv.item_stride = cur->cap; A.kf_fv.n_nodes = kf->fv_nn.data(); A.kf_fv.node_ids = kf->fv_ids.data();
  A.kf_fv.node_start = kf->fv_start.data(); A.kf_fv.items = kf->fv_items.data();
  A.n_f = cur->n.data(); A.f_kps = cur->kps.data(); A.f_desc = cur->desc.data();   // F.mvKeys (ORBmatcher.cc:258)
  A.f_fv.node_stride = cur->cap; A.f_fv.item_stride = cur->cap; A.f_fv.n_nodes = cur->fv_nn.data(); A.f_fv.node_ids = cur->fv_ids.data();
  A.f_fv.node_start = cur->fv_start.data(); A.f_fv.items = cur->fv_items.data();
  A.matcher.nnratio = 0.7f; A.matcher.check_orientation = 1;     // ORBmatcher matcher(0.7,true), :1207
  A.match_f_to_kf = match.data(); A.nmatches = cnt(cur, FB_CNT_BOW_MATCHES);
  orc_match_bow(&A);                                             // :1210
  // per sequence from here on: if (nmatches < 15) return false, :1212-1213
  for (size_t b = 0; b < B; b++) {
    if (cnt(cur, FB_CNT_BOW_MATCHES)[b] < 15) continue;
    for (size_t i = 0; i < cap; i++) {                           // mCurrentFrame.mvpMapPoints = vpMapPointMatches, :1215
      const int m = (int)i < cur->n[b] ? match[b * cap + i] : -1;
      cur->mp[b * cap + i] = m >= 0 ? kf->mp[b * cap + m] : -1;
    }
  }
  {  // PoseOptimizationWithBird for the sequences that went on (:1217-1220): the others keep pose, flags and counter
    std::vector<int32_t> n_keep = cur->n, nb_keep = cur->nb;
    for (size_t b = 0; b < B; b++)
      if (cnt(cur, FB_CNT_BOW_MATCHES)[b] < 15) { cur->n[b] = 0; cur->nb[b] = 0; }  // no edges: the optimiser leaves the sequence alone
    pose_optimization(cur, map, &mpbv, FB_POSE_FRONT_BIRD, T->wB, T->wF, FB_CNT_POSE1_INLIERS);
    cur->n = n_keep; cur->nb = nb_keep;
  }
  discard_outliers(cur, map, FB_CNT_BOW_MATCHES, true, FB_CNT_BOW_MATCHES, 15);   // :1222-1241
  return FB_OK;
}

// Tracking::TrackUsingBird (Tracking.cc:2014-2061); src = mpReferenceKF or tmpRefFrame, last = tmpRefFrame
int orc_frame_track_using_bird(orc_frame *cur, orc_frame *src, orc_frame *last, const fb_track_args *T) {
  const fb_map_points *map = &T->map;
  fb_map_points_bird mpbv = T->mpb;
  set_predicted_pose(cur, src, T->d_delta);                      // :2016-2034
  local_map_for_bird(cur, T, &mpbv);                             // :2036
  per_frame_matched_bird_points(cur, last, &mpbv, 11);           // :2038-2053: else branch (numPt <= 10) matches first
  pose_optimization(cur, map, &mpbv, FB_POSE_BIRD, 1.0f, 1.0f, FB_CNT_POSE1_INLIERS, false);   // BirdOptimization(&mCurrentFrame, 1.0)
  per_frame_matched_bird_points(cur, last, &mpbv, 0);            // :2056
  return FB_OK;
}

int orc_frame_track(orc_frame *cur, orc_frame *last, const fb_track_args *T) {
  orc_frame_track_motion_model(cur, last, T);
  fb_map_points_bird mpbv = T->mpb;
  track_local_map(cur, last, T, &mpbv, true);   // if (bOK) bOK = TrackLocalMap(), per sequence
  return FB_OK;
}

int orc_frame_view(orc_frame *f, fb_frame_view *v) {
  v->batch = f->B; v->kp_stride = f->cap;
  v->n = f->n.data(); v->kps = f->kps.data(); v->kps_un = f->kps_un.data(); v->desc = f->desc.data();
  v->map_point = f->mp.data(); v->outlier = f->outlier.data();
  v->n_bird = f->nb.data(); v->kps_bird = f->bkps.data(); v->desc_bird = f->bdesc.data(); v->bird_cam_xyz = f->bcam.data();
  v->map_point_bird = f->mpb.data(); v->bird_outlier = f->boutlier.data(); v->Tcw = f->Tcw.data(); v->counts = f->counts.data();
  return FB_OK;
}

int orc_frame_stage_seconds(orc_frame *f, double *out8) {
  std::memcpy(out8, f->stage_s, sizeof(f->stage_s));
  return FB_OK;
}

}  // extern "C"
