// fishbird_host.hpp -- C++ host side above the C-ABI (include/fishbird.h), mirroring the reference's
// operator interface for the hot path: same class names, argument meaning and return values as
//   ORBextractor  (include/ORBextractor.h:51-85)
//   ORBmatcher    (include/ORBmatcher.h:41-88)      SearchByProjection x2, BirdMapPointMatch, BirdviewMatch
//   Optimizer     (include/Optimizer.h:40-68)       PoseOptimization, PoseOptimizationWithBird, BirdOptimization,
//                                                   LocalBundleAdjustment[WithOdom] (flat graph, or KeyFrame*/Map* as
//                                                   in the reference: fishbird_map.hpp collects the graph),
//                                                   BundleAdjustmentWithOdom / GlobalBundleAdjustemntWithOdom
// on plain-old-data frames (the reference's Frame/KeyFrame/MapPoint own OpenCV and graph state that stays in the
// host application).  Header only; link with -lfishbird_hip.  Errors throw std::runtime_error(fb_last_error()).
#ifndef FISHBIRD_HOST_HPP_
#define FISHBIRD_HOST_HPP_

#include <cmath>
#include <cstdint>
#include <cstring>
#include <set>
#include <stdexcept>
#include <vector>

#include "../../include/fishbird.h"
#include "fishbird_map.hpp"

namespace fishbird {

inline void check(int rc) {
  if (rc != FB_OK) throw std::runtime_error(fb_last_error());
}

// ---- the part of Frame (include/Frame.h) the hot path reads and writes -------------------------------------------
struct MapPointRef {            // what the path needs from a MapPoint / MapPointBird
  bool valid = false;           // pointer != NULL (&& !isBad())
  bool hasObservations = true;  // Observations() > 0
  float Xw[3] = {0, 0, 0};      // GetWorldPos()
  uint8_t descriptor[32] = {0}; // GetDescriptor()
};

struct LocalMapPoint {          // a MapPoint of the local map: what isInFrustum reads and the track members it writes
  MapPointRef ref;
  float normal[3] = {0, 0, 0};  // GetNormal()
  float mfMaxDistance = 0, mfMinDistance = 0;
  bool mbTrackInView = false;   // MapPoint.h track members (written by Frame::isInFrustum)
  float mTrackProjX = 0, mTrackProjY = 0, mTrackProjXR = 0, mTrackViewCos = 0;
  int mnTrackScaleLevel = 0;
};

struct Frame {
  // intrinsics and static grid data (Frame.h statics)
  float fx = 0, fy = 0, cx = 0, cy = 0;
  float mnMinX = 0, mnMinY = 0, mnMaxX = 0, mnMaxY = 0;
  int birdviewCols = 0, birdviewRows = 0;
  std::vector<float> mvScaleFactors, mvInvLevelSigma2;
  float mTcw[12] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0};  // rows 0..2 of the 4x4
  // front
  std::vector<fb_keypoint> mvKeysUn;
  std::vector<uint8_t> mDescriptors;          // N x 32
  BowVector mBowVec;                          // Frame::ComputeBoW (ORBVocabulary::ComputeBoW below)
  FeatureVector mFeatVec;
  std::vector<int32_t> mvpMapPoints;          // index into the caller's map-point table, -1 = NULL
  std::vector<uint8_t> mvpMapPointHasObs;     // Observations()>0 of the point currently in slot i
  std::vector<uint8_t> mvbOutlier;
  // bird
  std::vector<fb_keypoint> mvKeysBird;
  std::vector<uint8_t> mDescriptorsBird;
  std::vector<float> mvKeysBirdCamXYZ;        // NB x 3
  std::vector<int32_t> mvpMapPointsBird;
  std::vector<uint8_t> mvBirdOutlier;
  // grids (Frame::AssignFeaturesToGrid), CSR with cell id = ix*rows+iy
  std::vector<int32_t> gridStart, gridItems, gridBirdStart, gridBirdItems;

  int N() const { return (int)mvKeysUn.size(); }
  int Nbird() const { return (int)mvKeysBird.size(); }
  fb_grid_geom frontGrid() const { return {mnMinX, mnMinY, 64.f / (mnMaxX - mnMinX), 48.f / (mnMaxY - mnMinY), 64, 48}; }
  fb_grid_geom birdGrid() const { return {0.f, 0.f, 32.f / (float)birdviewCols, 32.f / (float)birdviewRows, 32, 32}; }

  // Frame::AssignFeaturesToGrid (Frame.cc:381-411) on the host (tiny); PosInGrid rounding as Frame.cc:548-570
  static void assignToGrid(const std::vector<fb_keypoint> &k, const fb_grid_geom &g, std::vector<int32_t> &start,
                           std::vector<int32_t> &items) {
    const int ncell = g.cols * g.rows;
    std::vector<std::vector<int32_t>> cells(ncell);
    for (size_t i = 0; i < k.size(); i++) {
      const int px = (int)std::round((k[i].x - g.min_x) * g.inv_w), py = (int)std::round((k[i].y - g.min_y) * g.inv_h);
      if (px < 0 || px >= g.cols || py < 0 || py >= g.rows) continue;
      cells[px * g.rows + py].push_back((int32_t)i);
    }
    start.assign(ncell + 1, 0);
    items.assign(k.size() ? k.size() : 1, 0);
    int off = 0;
    for (int c = 0; c < ncell; c++) { start[c] = off; for (int32_t i : cells[c]) items[off++] = i; }
    start[ncell] = off;
  }
  void AssignFeaturesToGrid() {
    assignToGrid(mvKeysUn, frontGrid(), gridStart, gridItems);
    if (birdviewCols > 0) assignToGrid(mvKeysBird, birdGrid(), gridBirdStart, gridBirdItems);
  }

  // Frame::UndistortKeyPoints (Frame.cc:636-669): mvKeys -> mvKeysUn through cv::fisheye::undistortPoints(K, D, R=I, P=K);
  // DistCoef[0] == 0 copies.  K4 = fx, fy, cx, cy; D4 = k1..k4.
  void UndistortKeyPoints(const std::vector<fb_keypoint> &mvKeys, const float D4[4]) {
    mvKeysUn.resize(mvKeys.size());
    if (mvKeys.empty()) return;
    const float K4[4] = {fx, fy, cx, cy};
    check(fb_undistort_keypoints(mvKeys.data(), (int)mvKeys.size(), K4, D4, mvKeysUn.data()));
  }
  // Frame::ComputeImageBounds (Frame.cc:741-795), including its numeric_limits<float>::min() quirk: sets mnMinX..mnMaxY
  void ComputeImageBounds(int cols, int rows, const float D4[4]) {
    const float K4[4] = {fx, fy, cx, cy};
    float bnd[4];
    check(fb_image_bounds(cols, rows, K4, D4, bnd));
    mnMinX = bnd[0]; mnMaxX = bnd[1]; mnMinY = bnd[2]; mnMaxY = bnd[3];
  }

  // Frame::isInFrustum(pMP, viewingCosLimit) (Frame.cc:435-491) over the whole local map in one call, the loop of
  // Tracking::SearchLocalPoints (Tracking.cc:1071-1091).  Writes the track members of every point; returns how many
  // are in view.  mOw = -Rcw^T tcw as Frame::UpdatePoseMatrices computes it (float Mat product, double accumulation).
  int isInFrustum(std::vector<LocalMapPoint> &points, float viewingCosLimit, float mbf = 0.f) const {
    const int32_t n = (int32_t)points.size();
    if (n == 0) return 0;
    std::vector<uint8_t> valid(n), inView(n, 0);
    std::vector<float> xw((size_t)n * 3), nrm((size_t)n * 3), mx(n), mn(n), proj((size_t)n * 2, 0.f), xr(n, 0.f), vc(n, 0.f);
    std::vector<int32_t> lvl(n, 0);
    for (int i = 0; i < n; i++) {
      valid[i] = points[i].ref.valid;
      std::memcpy(&xw[3 * (size_t)i], points[i].ref.Xw, 12);
      std::memcpy(&nrm[3 * (size_t)i], points[i].normal, 12);
      mx[i] = points[i].mfMaxDistance; mn[i] = points[i].mfMinDistance;
    }
    float Ow[3];
    for (int r = 0; r < 3; r++) {
      double s = 0;
      for (int k = 0; k < 3; k++) s += (double)mTcw[k * 4 + r] * (double)mTcw[k * 4 + 3];
      Ow[r] = (float)(-s);
    }
    fb_frustum_args a{};
    a.batch = 1; a.mp_stride = n; a.Tcw = mTcw; a.Ow = Ow; a.n_mp = &n; a.mp_valid = valid.data(); a.mp_xw = xw.data();
    a.mp_normal = nrm.data(); a.mp_max_dist = mx.data(); a.mp_min_dist = mn.data();
    a.cam = {fx, fy, cx, cy, mnMinX, mnMinY, mnMaxX, mnMaxY};
    a.mbf = mbf; a.viewing_cos_limit = viewingCosLimit;
    a.log_scale_factor = mvScaleFactors.size() > 1 ? std::log(mvScaleFactors[1]) : 1.f;
    a.n_levels = (int32_t)mvScaleFactors.size();
    a.in_view = inView.data(); a.proj = proj.data(); a.proj_xr = xr.data(); a.level = lvl.data(); a.view_cos = vc.data();
    check(fb_in_frustum(&a));
    int cnt = 0;
    for (int i = 0; i < n; i++) {
      points[i].mbTrackInView = inView[i] != 0;
      if (!inView[i]) continue;
      cnt++;
      points[i].mTrackProjX = proj[2 * (size_t)i]; points[i].mTrackProjY = proj[2 * (size_t)i + 1];
      points[i].mTrackProjXR = xr[i]; points[i].mnTrackScaleLevel = lvl[i]; points[i].mTrackViewCos = vc[i];
    }
    return cnt;
  }
};

// ---- ORBextractor ------------------------------------------------------------------------------------------------
class ORBextractor {
 public:
  ORBextractor(int nfeatures, float scaleFactor, int nlevels, int iniThFAST, int minThFAST)
      : p_{nfeatures, scaleFactor, nlevels, iniThFAST, minThFAST} {
    check(fb_orb_create(&p_, &h_));
    check(fb_orb_get_tables(h_, &t_));
  }
  ~ORBextractor() { fb_orb_destroy(h_); }
  ORBextractor(const ORBextractor &) = delete;
  ORBextractor &operator=(const ORBextractor &) = delete;

  // operator()(image, mask, keypoints, descriptors); the mask is ignored like in the reference (ORBextractor.h:58)
  void operator()(const uint8_t *image, int width, int height, int stride, std::vector<fb_keypoint> &keypoints,
                  std::vector<uint8_t> &descriptors) {
    keypoints.clear();
    descriptors.clear();
    if (!image || width <= 0 || height <= 0) return;  // _image.empty()
    const int cap = fb_orb_capacity(&p_);
    keypoints.resize(cap);
    descriptors.resize((size_t)cap * 32);
    int32_t n = 0;
    check(fb_orb_extract(h_, image, width, height, stride, keypoints.data(), descriptors.data(), &n));
    keypoints.resize(n);
    descriptors.resize((size_t)n * 32);
  }
  int GetLevels() const { return p_.nlevels; }
  float GetScaleFactor() const { return p_.scale_factor; }
  std::vector<float> GetScaleFactors() const { return {t_.scale_factor, t_.scale_factor + p_.nlevels}; }
  std::vector<float> GetInverseScaleFactors() const { return {t_.inv_scale_factor, t_.inv_scale_factor + p_.nlevels}; }
  std::vector<float> GetScaleSigmaSquares() const { return {t_.level_sigma2, t_.level_sigma2 + p_.nlevels}; }
  std::vector<float> GetInverseScaleSigmaSquares() const { return {t_.inv_level_sigma2, t_.inv_level_sigma2 + p_.nlevels}; }

 private:
  fb_orb_params p_;
  fb_orb *h_ = nullptr;
  fb_orb_tables t_;
};

// ---- ORBVocabulary (DBoW2::TemplatedVocabulary<FORB::TDescriptor, FORB>, TemplatedVocabulary.h) ------------------
// The tree as flat arrays (fb_vocabulary).  transform() is Frame::ComputeBoW / KeyFrame::ComputeBoW (Frame.cc:628-635):
// mBowVec as word id -> value, mFeatVec as NodeId -> feature indices.
class ORBVocabulary {
 public:
  int L = 0;
  std::vector<int32_t> child_start{0}, children, word_ids;
  std::vector<uint8_t> descriptors;
  std::vector<double> weights;

  void transform(const std::vector<uint8_t> &features, BowVector &v, FeatureVector &fv, int levelsup) const {
    v.clear();
    fv.clear();
    const int32_t n = (int32_t)(features.size() / 32);
    if (n == 0) return;
    fb_vocabulary V{};
    V.n_nodes = (int32_t)word_ids.size(); V.L = L; V.child_start = child_start.data(); V.children = children.data();
    V.descriptors = descriptors.data(); V.weights = weights.data(); V.word_ids = word_ids.data();
    std::vector<uint32_t> bid(n), nid(n);
    std::vector<double> bval(n);
    std::vector<int32_t> nstart(n + 1), items(n);
    int32_t nw = 0, nn = 0;
    fb_bow_transform_args a{};
    a.batch = 1; a.f_stride = n; a.n_f = &n; a.desc = features.data(); a.levelsup = levelsup;
    a.n_words = &nw; a.bow_ids = bid.data(); a.bow_vals = bval.data();
    a.fv_n_nodes = &nn; a.fv_node_ids = nid.data(); a.fv_node_start = nstart.data(); a.fv_items = items.data();
    check(fb_bow_transform(&V, &a));
    for (int i = 0; i < nw; i++) v[bid[i]] = bval[i];
    for (int k = 0; k < nn; k++) {
      std::vector<unsigned> &dst = fv[nid[k]];
      for (int j = nstart[k]; j < nstart[k + 1]; j++) dst.push_back((unsigned)items[j]);
    }
  }
  // Frame::ComputeBoW / KeyFrame::ComputeBoW (Frame.cc:628-635, KeyFrame.cc:76-86): only when mBowVec is still empty
  void ComputeBoW(Frame &F) const { if (F.mBowVec.empty()) transform(F.mDescriptors, F.mBowVec, F.mFeatVec, 4); }
  void ComputeBoW(KeyFrame &K) const { if (K.mBowVec.empty()) transform(K.mDescriptors, K.mBowVec, K.mFeatVec, 4); }
};

// ---- ORBmatcher --------------------------------------------------------------------------------------------------
class ORBmatcher {
 public:
  explicit ORBmatcher(float nnratio = 0.6f, bool checkOri = true) : m_{nnratio, checkOri ? 1 : 0} {}

  static int DescriptorDistance(const uint8_t *a, const uint8_t *b) {
    int32_t d = 0;
    check(fb_descriptor_distance(a, b, 1, &d));
    return d;
  }

  // SearchByProjection(CurrentFrame, LastFrame, th, bMono=true), ORBmatcher.cc:1329-1471.
  // lastPoints[i] describes LastFrame.mvpMapPoints[i] (valid = pointer && !mvbOutlier[i]).
  int SearchByProjection(Frame &cur, const Frame &last, const std::vector<MapPointRef> &lastPoints, float th) const {
    const int32_t N = cur.N(), NL = last.N();
    std::vector<uint8_t> valid(NL), obs(NL), desc((size_t)NL * 32), blocked(N, 0);
    std::vector<float> xw((size_t)NL * 3), angle(NL);
    std::vector<int32_t> oct(NL), match(N > 0 ? N : 1);
    for (int i = 0; i < NL; i++) {
      valid[i] = lastPoints[i].valid && !(i < (int)last.mvbOutlier.size() && last.mvbOutlier[i]);
      obs[i] = lastPoints[i].hasObservations;
      std::memcpy(&xw[3 * i], lastPoints[i].Xw, 12);
      std::memcpy(&desc[32 * (size_t)i], lastPoints[i].descriptor, 32);
      oct[i] = last.mvKeysUn[i].octave;
      angle[i] = last.mvKeysUn[i].angle;
    }
    for (int i = 0; i < N; i++) blocked[i] = cur.mvpMapPoints[i] >= 0 && cur.mvpMapPointHasObs[i];
    fb_proj_frame_args a{};
    a.batch = 1; a.cur_stride = N; a.last_stride = NL;
    a.n_cur = &N; a.cur_kps = cur.mvKeysUn.data(); a.cur_desc = cur.mDescriptors.data();
    a.cur_cell_start = cur.gridStart.data(); a.cur_cell_items = cur.gridItems.data(); a.cur_blocked = blocked.data();
    a.cur_Tcw = cur.mTcw; a.n_last = &NL; a.last_valid = valid.data(); a.last_obs_pos = obs.data(); a.last_xw = xw.data();
    a.last_desc = desc.data(); a.last_octave = oct.data(); a.last_angle = angle.data();
    a.cam = {cur.fx, cur.fy, cur.cx, cur.cy, cur.mnMinX, cur.mnMinY, cur.mnMaxX, cur.mnMaxY};
    a.grid = cur.frontGrid();
    for (size_t i = 0; i < cur.mvScaleFactors.size() && i < FB_MAX_LEVELS; i++) a.scale_factors[i] = cur.mvScaleFactors[i];
    a.th = th; a.matcher = m_;
    int32_t nmatches = 0;
    a.match_cur_to_last = match.data(); a.nmatches = &nmatches;
    check(fb_match_projection_frame(&a));
    for (int i = 0; i < N; i++)
      if (match[i] >= 0) { cur.mvpMapPoints[i] = last.mvpMapPoints[match[i]]; cur.mvpMapPointHasObs[i] = obs[match[i]]; }
    // culled assignments (rotation histogram) are already -1 in match[]; reproduce the NULL writes
    return nmatches;
  }

  // SearchByProjection(Frame &F, const vector<MapPoint*> &vpMapPoints, th), ORBmatcher.cc:46-130 (TrackLocalMap).
  // points[i] carries the MapPoint's track members as Frame::isInFrustum left them (below); returns nmatches and
  // writes F.mvpMapPoints[idx] = index of the matched point.
  int SearchByProjection(Frame &F, const std::vector<LocalMapPoint> &points, float th = 1.f) const {
    const int32_t N = F.N(), NM = (int32_t)points.size();
    std::vector<uint8_t> track(NM > 0 ? NM : 1), obs(NM > 0 ? NM : 1), desc((size_t)(NM > 0 ? NM : 1) * 32), blocked(N > 0 ? N : 1, 0);
    std::vector<float> proj((size_t)(NM > 0 ? NM : 1) * 2), vcos(NM > 0 ? NM : 1);
    std::vector<int32_t> level(NM > 0 ? NM : 1), match(N > 0 ? N : 1);
    for (int i = 0; i < NM; i++) {
      track[i] = points[i].mbTrackInView && points[i].ref.valid;
      obs[i] = points[i].ref.hasObservations;
      proj[2 * i] = points[i].mTrackProjX; proj[2 * i + 1] = points[i].mTrackProjY;
      level[i] = points[i].mnTrackScaleLevel; vcos[i] = points[i].mTrackViewCos;
      std::memcpy(&desc[32 * (size_t)i], points[i].ref.descriptor, 32);
    }
    for (int i = 0; i < N; i++) blocked[i] = F.mvpMapPoints[i] >= 0 && F.mvpMapPointHasObs[i];
    fb_proj_points_args a{};
    a.batch = 1; a.cur_stride = N; a.mp_stride = NM > 0 ? NM : 1;
    a.n_cur = &N; a.cur_kps = F.mvKeysUn.data(); a.cur_desc = F.mDescriptors.data();
    a.cur_cell_start = F.gridStart.data(); a.cur_cell_items = F.gridItems.data(); a.cur_blocked = blocked.data();
    a.n_mp = &NM; a.mp_track = track.data(); a.mp_obs_pos = obs.data(); a.mp_proj = proj.data(); a.mp_level = level.data();
    a.mp_view_cos = vcos.data(); a.mp_desc = desc.data();
    a.grid = F.frontGrid();
    for (size_t i = 0; i < F.mvScaleFactors.size() && i < FB_MAX_LEVELS; i++) a.scale_factors[i] = F.mvScaleFactors[i];
    a.th = th; a.matcher = m_;
    int32_t nmatches = 0;
    a.match_cur_to_mp = match.data(); a.nmatches = &nmatches;
    check(fb_match_projection_points(&a));
    for (int i = 0; i < N; i++)
      if (match[i] >= 0) { F.mvpMapPoints[i] = match[i]; F.mvpMapPointHasObs[i] = obs[match[i]]; }
    return nmatches;
  }

  // BirdMapPointMatch(CurF, vRefMapPointsBird, windowSize, filterSize), ORBmatcher.cc:1763-1902
  int BirdMapPointMatch(Frame &cur, const std::vector<MapPointRef> &ref, const float Tbc[12], int windowSize,
                        float filterSize, double meter2pixel = 25.1, double rearAxleToCenter = 1.393) const {
    const int32_t NB = cur.Nbird(), NR = (int32_t)ref.size();
    std::vector<uint8_t> valid(NR), desc((size_t)NR * 32);
    std::vector<float> xw((size_t)NR * 3);
    for (int i = 0; i < NR; i++) {
      valid[i] = ref[i].valid;
      std::memcpy(&xw[3 * i], ref[i].Xw, 12);
      std::memcpy(&desc[32 * (size_t)i], ref[i].descriptor, 32);
    }
    fb_bird_mp_args a{};
    a.batch = 1; a.cur_stride = NB; a.ref_stride = NR;
    a.n_cur = &NB; a.cur_kps = cur.mvKeysBird.data(); a.cur_desc = cur.mDescriptorsBird.data();
    a.cur_cam_xyz = cur.mvKeysBirdCamXYZ.data(); a.cur_cell_start = cur.gridBirdStart.data();
    a.cur_cell_items = cur.gridBirdItems.data(); a.cur_Tcw = cur.mTcw;
    a.n_ref = &NR; a.ref_valid = valid.data(); a.ref_xw = xw.data(); a.ref_desc = desc.data();
    std::memcpy(a.Tbc, Tbc, sizeof(a.Tbc));
    a.bird_cols = cur.birdviewCols; a.bird_rows = cur.birdviewRows; a.meter2pixel = meter2pixel;
    a.rear_axle_to_center = rearAxleToCenter; a.grid = cur.birdGrid(); a.window_size = windowSize;
    a.filter_size = filterSize; a.matcher = m_;
    int32_t ninl = 0;
    a.match_cur_to_ref = cur.mvpMapPointsBird.data(); a.ninliers = &ninl;
    check(fb_match_bird_mappoints(&a));
    return ninl;
  }

  // ================= key-frame side (LocalMapping / LoopClosing / relocalisation / initialisation) =================
  // SearchByBoW(pKF, F, vpMapPointMatches), ORBmatcher.cc:160-289
  int SearchByBoW(KeyFrame *pKF, Frame &F, std::vector<MapPoint *> &vpMapPointMatches) const {
    const std::vector<MapPoint *> vpMapPointsKF = pKF->GetMapPointMatches();
    const int32_t NK = pKF->N(), NF = F.N();
    vpMapPointMatches.assign(NF, nullptr);
    if (NK == 0 || NF == 0) return 0;
    std::vector<uint8_t> has(NK);
    for (int i = 0; i < NK; i++) has[i] = vpMapPointsKF[i] && !vpMapPointsKF[i]->isBad();
    FvFlat fk(pKF->mFeatVec), ff(F.mFeatVec);
    std::vector<int32_t> match(NF, -1);
    int32_t n = 0;
    fb_bow_args a{};
    a.batch = 1; a.kf_stride = NK; a.f_stride = NF;
    a.n_kf = &NK; a.kf_kps = pKF->mvKeysUn.data(); a.kf_desc = pKF->mDescriptors.data(); a.kf_has_mp = has.data(); a.kf_fv = fk.view();
    a.n_f = &NF; a.f_kps = F.mvKeysUn.data(); a.f_desc = F.mDescriptors.data(); a.f_fv = ff.view();
    a.matcher = m_; a.match_f_to_kf = match.data(); a.nmatches = &n;
    check(fb_match_bow(&a));
    for (int i = 0; i < NF; i++) if (match[i] >= 0) vpMapPointMatches[i] = vpMapPointsKF[match[i]];
    return n;
  }

  // SearchByBoW(pKF1, pKF2, vpMatches12), ORBmatcher.cc:523-656
  int SearchByBoW(KeyFrame *pKF1, KeyFrame *pKF2, std::vector<MapPoint *> &vpMatches12) const {
    const std::vector<MapPoint *> mp1 = pKF1->GetMapPointMatches(), mp2 = pKF2->GetMapPointMatches();
    const int32_t N1 = pKF1->N(), N2 = pKF2->N();
    vpMatches12.assign(N1, nullptr);
    if (N1 == 0 || N2 == 0) return 0;
    std::vector<uint8_t> h1(N1), h2(N2);
    for (int i = 0; i < N1; i++) h1[i] = mp1[i] && !mp1[i]->isBad();
    for (int i = 0; i < N2; i++) h2[i] = mp2[i] && !mp2[i]->isBad();
    FvFlat f1(pKF1->mFeatVec), f2(pKF2->mFeatVec);
    std::vector<int32_t> m12(N1, -1);
    int32_t n = 0;
    fb_bow_kf_args a{};
    a.batch = 1; a.kf1_stride = N1; a.kf2_stride = N2;
    a.n1 = &N1; a.kps1 = pKF1->mvKeysUn.data(); a.desc1 = pKF1->mDescriptors.data(); a.has_mp1 = h1.data(); a.fv1 = f1.view();
    a.n2 = &N2; a.kps2 = pKF2->mvKeysUn.data(); a.desc2 = pKF2->mDescriptors.data(); a.has_mp2 = h2.data(); a.fv2 = f2.view();
    a.matcher = m_; a.matches12 = m12.data(); a.nmatches = &n;
    check(fb_match_bow_kf(&a));
    for (int i = 0; i < N1; i++) if (m12[i] >= 0) vpMatches12[i] = mp2[m12[i]];
    return n;
  }

  // SearchForTriangulation(pKF1, pKF2, F12, vMatchedPairs, bOnlyStereo), ORBmatcher.cc:658-824.  F12 row-major 3x3.
  int SearchForTriangulation(KeyFrame *pKF1, KeyFrame *pKF2, const float F12[9], std::vector<std::pair<size_t, size_t>> &vMatchedPairs,
                             bool bOnlyStereo) const {
    vMatchedPairs.clear();
    const int32_t N1 = pKF1->N(), N2 = pKF2->N();
    if (bOnlyStereo || N1 == 0 || N2 == 0) return 0;  // monocular key frames: every candidate fails the stereo test (:701-703)
    std::vector<uint8_t> h1(N1), h2(N2);
    for (int i = 0; i < N1; i++) h1[i] = pKF1->GetMapPoint(i) != nullptr;
    for (int i = 0; i < N2; i++) h2[i] = pKF2->GetMapPoint(i) != nullptr;
    FvFlat f1(pKF1->mFeatVec), f2(pKF2->mFeatVec);
    float R2w[9], t2w[3];
    for (int r = 0; r < 3; r++) { for (int c = 0; c < 3; c++) R2w[r * 3 + c] = pKF2->Tcw[r * 4 + c]; t2w[r] = pKF2->Tcw[r * 4 + 3]; }
    std::vector<int32_t> m12(N1, -1);
    int32_t n = 0;
    fb_triangulation_args a{};
    a.batch = 1; a.kf1_stride = N1; a.kf2_stride = N2;
    a.n1 = &N1; a.kps1 = pKF1->mvKeysUn.data(); a.desc1 = pKF1->mDescriptors.data(); a.has_mp1 = h1.data(); a.fv1 = f1.view();
    a.n2 = &N2; a.kps2 = pKF2->mvKeysUn.data(); a.desc2 = pKF2->mDescriptors.data(); a.has_mp2 = h2.data(); a.fv2 = f2.view();
    a.F12 = F12; a.Cw1 = pKF1->GetCameraCenter(); a.R2w = R2w; a.t2w = t2w;
    a.fx = pKF2->fx; a.fy = pKF2->fy; a.cx = pKF2->cx; a.cy = pKF2->cy;
    for (size_t i = 0; i < pKF2->mvScaleFactors.size() && i < FB_MAX_LEVELS; i++) a.scale_factors[i] = pKF2->mvScaleFactors[i];
    for (size_t i = 0; i < pKF2->mvLevelSigma2.size() && i < FB_MAX_LEVELS; i++) a.level_sigma2[i] = pKF2->mvLevelSigma2[i];
    a.matcher = m_; a.matches12 = m12.data(); a.nmatches = &n;
    check(fb_match_triangulation(&a));
    for (int i = 0; i < N1; i++) if (m12[i] >= 0) vMatchedPairs.push_back(std::make_pair((size_t)i, (size_t)m12[i]));
    return n;
  }

  // Fuse(pKF, vpMapPoints, th), ORBmatcher.cc:826-976: batched search, then the map mutation in list order
  int Fuse(KeyFrame *pKF, const std::vector<MapPoint *> &vpMapPoints, float th = 3.f) const {
    const int32_t nMPs = (int32_t)vpMapPoints.size();
    if (nMPs == 0 || pKF->N() == 0) return 0;
    KfFlat kf(pKF);
    MpFlat mp(vpMapPoints, [&](MapPoint *p) { return p && !p->isBad() && !p->IsInKeyFrame(pKF); });
    std::vector<int32_t> best(nMPs, -1);
    fb_fuse_args a{};
    a.batch = 1; a.kf = kf.t; a.mp = mp.view(); a.pose = pKF->Tcw; a.Ow = pKF->GetCameraCenter(); a.th = th; a.best_idx = best.data();
    check(fb_fuse_search(&a));
    int nFused = 0;
    for (int i = 0; i < nMPs; i++) {
      MapPoint *pMP = vpMapPoints[i];
      // re-evaluated in order: an earlier Replace / AddObservation of this loop can have changed it (:846-850)
      if (!pMP || pMP->isBad() || pMP->IsInKeyFrame(pKF) || best[i] < 0) continue;
      MapPoint *pMPinKF = pKF->GetMapPoint(best[i]);
      if (pMPinKF) {
        if (!pMPinKF->isBad()) {
          if (pMPinKF->Observations() > pMP->Observations()) pMP->Replace(pMPinKF);
          else pMPinKF->Replace(pMP);
        }
      } else {
        pMP->AddObservation(pKF, best[i]);
        pKF->AddMapPoint(pMP, best[i]);
      }
      nFused++;
    }
    return nFused;
  }

  // Fuse(pKF, Scw, vpPoints, th, vpReplacePoint), ORBmatcher.cc:978-1101.  Scw = rows 0..2 of the 4x4.
  int Fuse(KeyFrame *pKF, const float Scw[12], const std::vector<MapPoint *> &vpPoints, float th,
           std::vector<MapPoint *> &vpReplacePoint) const {
    const int32_t nPoints = (int32_t)vpPoints.size();
    if (nPoints == 0 || pKF->N() == 0) return 0;
    const std::set<MapPoint *> spAlreadyFound = pKF->GetMapPoints();
    KfFlat kf(pKF);
    MpFlat mp(vpPoints, [&](MapPoint *p) { return p && !p->isBad() && !spAlreadyFound.count(p); });
    std::vector<int32_t> best(nPoints, -1);
    fb_fuse_args a{};
    a.batch = 1; a.kf = kf.t; a.mp = mp.view(); a.pose = Scw; a.Ow = nullptr; a.th = th; a.best_idx = best.data();
    check(fb_fuse_sim3_search(&a));
    int nFused = 0;
    for (int i = 0; i < nPoints; i++) {
      if (best[i] < 0) continue;
      MapPoint *pMP = vpPoints[i];
      MapPoint *pMPinKF = pKF->GetMapPoint(best[i]);
      if (pMPinKF) {
        if (!pMPinKF->isBad()) vpReplacePoint[i] = pMPinKF;
      } else {
        pMP->AddObservation(pKF, best[i]);
        pKF->AddMapPoint(pMP, best[i]);
      }
      nFused++;
    }
    return nFused;
  }

  // SearchByProjection(pKF, Scw, vpPoints, vpMatched, th), ORBmatcher.cc:291-404
  int SearchByProjection(KeyFrame *pKF, const float Scw[12], const std::vector<MapPoint *> &vpPoints, std::vector<MapPoint *> &vpMatched,
                         int th) const {
    const int32_t N = pKF->N();
    if (vpPoints.empty() || N == 0) return 0;
    std::set<MapPoint *> spAlreadyFound(vpMatched.begin(), vpMatched.end());
    spAlreadyFound.erase(nullptr);
    KfFlat kf(pKF);
    MpFlat mp(vpPoints, [&](MapPoint *p) { return p && !p->isBad() && !spAlreadyFound.count(p); });
    std::vector<uint8_t> matched(N);
    for (int i = 0; i < N; i++) matched[i] = vpMatched[i] != nullptr;
    std::vector<int32_t> out(N, -1);
    int32_t n = 0;
    fb_proj_sim3_args a{};
    a.batch = 1; a.kf = kf.t; a.mp = mp.view(); a.Scw = Scw; a.kf_matched = matched.data(); a.th = th;
    a.match_kf_to_mp = out.data(); a.nmatches = &n;
    check(fb_match_projection_sim3(&a));
    for (int i = 0; i < N; i++) if (out[i] >= 0) vpMatched[i] = vpPoints[out[i]];
    return n;
  }

  // SearchBySim3(pKF1, pKF2, vpMatches12, s12, R12, t12, th), ORBmatcher.cc:1103-1327.  R12 row-major 3x3.
  int SearchBySim3(KeyFrame *pKF1, KeyFrame *pKF2, std::vector<MapPoint *> &vpMatches12, float s12, const float R12[9],
                   const float t12[3], float th) const {
    const std::vector<MapPoint *> vp1 = pKF1->GetMapPointMatches(), vp2 = pKF2->GetMapPointMatches();
    const int N1 = (int)vp1.size(), N2 = (int)vp2.size();
    if (N1 == 0 || N2 == 0) return 0;
    std::vector<bool> am1(N1, false), am2(N2, false);
    for (int i = 0; i < N1; i++) {
      MapPoint *pMP = vpMatches12[i];
      if (!pMP) continue;
      am1[i] = true;
      const int idx2 = pMP->GetIndexInKeyFrame(pKF2);
      if (idx2 >= 0 && idx2 < N2) am2[idx2] = true;
    }
    KfFlat k1(pKF1), k2(pKF2);
    int i1 = 0, i2 = 0;
    MpFlat m1(vp1, [&](MapPoint *p) { const bool ok = p && !am1[i1] && !p->isBad(); i1++; return ok; });
    MpFlat m2(vp2, [&](MapPoint *p) { const bool ok = p && !am2[i2] && !p->isBad(); i2++; return ok; });
    std::vector<int32_t> m12(N1, -1);
    int32_t n = 0;
    fb_sim3_args a{};
    a.batch = 1; a.kf1 = k1.t; a.kf2 = k2.t; a.mp1 = m1.view(); a.mp2 = m2.view();
    a.T1w = pKF1->Tcw; a.T2w = pKF2->Tcw; a.s12 = &s12; a.R12 = R12; a.t12 = t12; a.th = th;
    a.matches12 = m12.data(); a.nfound = &n;
    check(fb_match_sim3(&a));
    for (int i = 0; i < N1; i++) if (m12[i] >= 0) vpMatches12[i] = vp2[m12[i]];
    return n;
  }

  // SearchForInitialization(F1, F2, vbPrevMatched, vnMatches12, windowSize), ORBmatcher.cc:406-521.
  // vbPrevMatched: N1 x (x, y), updated in place with the matched F2 positions (:514-517).
  int SearchForInitialization(Frame &F1, Frame &F2, std::vector<float> &vbPrevMatched, std::vector<int> &vnMatches12,
                              int windowSize = 10) const {
    const int32_t N1 = F1.N(), N2 = F2.N();
    vnMatches12.assign(N1, -1);
    if (N1 == 0 || N2 == 0) return 0;
    std::vector<int32_t> m12(N1, -1);
    int32_t n = 0;
    fb_init_match_args a{};
    a.batch = 1; a.f1_stride = N1; a.f2_stride = N2;
    a.n1 = &N1; a.kps1 = F1.mvKeysUn.data(); a.desc1 = F1.mDescriptors.data();
    a.n2 = &N2; a.kps2 = F2.mvKeysUn.data(); a.desc2 = F2.mDescriptors.data();
    a.f2_cell_start = F2.gridStart.data(); a.f2_cell_items = F2.gridItems.data(); a.grid = F2.frontGrid();
    a.window_size = windowSize; a.matcher = m_;
    a.prev_matched = vbPrevMatched.data(); a.matches12 = m12.data(); a.nmatches = &n;
    check(fb_match_initialization(&a));
    for (int i = 0; i < N1; i++) vnMatches12[i] = m12[i];
    return n;
  }

  // SearchByProjection(CurrentFrame, pKF, sAlreadyFound, th, ORBdist), ORBmatcher.cc:1473-1600 (relocalisation).
  // curMapPoints plays CurrentFrame.mvpMapPoints (pointer form).
  int SearchByProjection(Frame &CurrentFrame, std::vector<MapPoint *> &curMapPoints, KeyFrame *pKF,
                         const std::set<MapPoint *> &sAlreadyFound, float th, int ORBdist) const {
    const std::vector<MapPoint *> vpMPs = pKF->GetMapPointMatches();
    const int32_t N = CurrentFrame.N(), NK = (int32_t)vpMPs.size();
    if (N == 0 || NK == 0) return 0;
    std::vector<uint8_t> blocked(N), valid(NK), desc((size_t)NK * 32);
    std::vector<float> xw((size_t)NK * 3), mx(NK), mn(NK), ang(NK);
    for (int i = 0; i < N; i++) blocked[i] = curMapPoints[i] != nullptr;
    for (int i = 0; i < NK; i++) {
      MapPoint *p = vpMPs[i];
      valid[i] = p && !p->isBad() && !sAlreadyFound.count(p);
      ang[i] = pKF->mvKeysUn[i].angle;
      if (!p) continue;
      std::memcpy(&xw[3 * (size_t)i], p->mWorldPos, 12);
      std::memcpy(&desc[32 * (size_t)i], p->mDescriptor, 32);
      mx[i] = p->mfMaxDistance; mn[i] = p->mfMinDistance;
    }
    std::vector<int32_t> match(N, -1);
    int32_t n = 0;
    fb_proj_kf_args a{};
    a.batch = 1; a.cur_stride = N; a.kf_stride = NK;
    a.n_cur = &N; a.cur_kps = CurrentFrame.mvKeysUn.data(); a.cur_desc = CurrentFrame.mDescriptors.data();
    a.cur_cell_start = CurrentFrame.gridStart.data(); a.cur_cell_items = CurrentFrame.gridItems.data(); a.cur_blocked = blocked.data();
    a.cur_Tcw = CurrentFrame.mTcw; a.n_kf = &NK; a.kf_valid = valid.data(); a.kf_xw = xw.data(); a.kf_desc = desc.data();
    a.kf_max_dist = mx.data(); a.kf_min_dist = mn.data(); a.kf_angle = ang.data();
    a.cam = {CurrentFrame.fx, CurrentFrame.fy, CurrentFrame.cx, CurrentFrame.cy, CurrentFrame.mnMinX, CurrentFrame.mnMinY,
             CurrentFrame.mnMaxX, CurrentFrame.mnMaxY};
    a.grid = CurrentFrame.frontGrid();
    for (size_t i = 0; i < CurrentFrame.mvScaleFactors.size() && i < FB_MAX_LEVELS; i++) a.scale_factors[i] = CurrentFrame.mvScaleFactors[i];
    a.log_scale_factor = CurrentFrame.mvScaleFactors.size() > 1 ? std::log(CurrentFrame.mvScaleFactors[1]) : 1.f;
    a.n_levels = (int32_t)CurrentFrame.mvScaleFactors.size();
    a.th = th; a.orb_dist = ORBdist; a.matcher = m_; a.match_cur_to_kf = match.data(); a.nmatches = &n;
    check(fb_match_projection_keyframe(&a));
    for (int i = 0; i < N; i++) if (match[i] >= 0) curMapPoints[i] = vpMPs[match[i]];
    return n;
  }

  // BirdviewMatch(CurF, vRefKeysBird, DescriptorsBird, vRefMapPointsBird, vDMatches12, isProject = 0, windowSize),
  // ORBmatcher.cc:1602-1760 (the live form, Tracking.cc:2728).  DMatch = {queryIdx, trainIdx, distance}.
  struct DMatch { int queryIdx, trainIdx; float distance; };
  int BirdviewMatch(Frame &CurF, const std::vector<fb_keypoint> &vRefKeysBird, const std::vector<uint8_t> &DescriptorsBird,
                    std::vector<DMatch> &vDMatches12, int windowSize = 10) const {
    vDMatches12.clear();
    const int32_t NC = CurF.Nbird(), NR = (int32_t)vRefKeysBird.size();
    if (NC == 0 || NR == 0) return 0;
    std::vector<int32_t> m(NR, -1), dist(NR, 0);
    int32_t n = 0, nd = 0;
    fb_birdview_args a{};
    a.batch = 1; a.cur_stride = NC; a.ref_stride = NR;
    a.n_cur = &NC; a.cur_kps = CurF.mvKeysBird.data(); a.cur_desc = CurF.mDescriptorsBird.data();
    a.cur_cell_start = CurF.gridBirdStart.data(); a.cur_cell_items = CurF.gridBirdItems.data();
    a.n_ref = &NR; a.ref_kps = vRefKeysBird.data(); a.ref_desc = DescriptorsBird.data();
    a.grid = CurF.birdGrid(); a.window_size = windowSize; a.matcher = m_;
    a.match_ref_to_cur = m.data(); a.match_dist = dist.data(); a.nmatches = &n; a.n_dmatches = &nd;
    check(fb_match_birdview(&a));
    for (int i = 0; i < NR; i++) if (m[i] > 0) vDMatches12.push_back({i, m[i], (float)dist[i]});  // sic: index 0 is dropped (:1755)
    return n;
  }

 private:
  // flat views of the host containers for the C-ABI
  struct FvFlat {
    std::vector<uint32_t> ids;
    std::vector<int32_t> start, items;
    int32_t n = 0;
    explicit FvFlat(const FeatureVector &fv) {
      start.push_back(0);
      for (FeatureVector::const_iterator it = fv.begin(); it != fv.end(); ++it) {
        ids.push_back(it->first);
        for (size_t k = 0; k < it->second.size(); k++) items.push_back((int32_t)it->second[k]);
        start.push_back((int32_t)items.size());
      }
      n = (int32_t)ids.size();
      if (ids.empty()) { ids.push_back(0); start.push_back(0); }
      if (items.empty()) items.push_back(0);
    }
    fb_feature_vector view() const { return {(int32_t)ids.size(), (int32_t)items.size(), &n, ids.data(), start.data(), items.data()}; }
  };
  struct KfFlat {
    int32_t n;
    fb_kf_target t;
    explicit KfFlat(KeyFrame *pKF) : n(pKF->N()) {
      std::memset(&t, 0, sizeof(t));
      if (pKF->gridStart.empty()) pKF->AssignFeaturesToGrid();
      t.kf_stride = n; t.n_kf = &n; t.kf_kps = pKF->mvKeysUn.data(); t.kf_desc = pKF->mDescriptors.data();
      t.kf_cell_start = pKF->gridStart.data(); t.kf_cell_items = pKF->gridItems.data();
      t.cam = pKF->camera(); t.grid = pKF->gridGeom();
      for (size_t i = 0; i < pKF->mvScaleFactors.size() && i < FB_MAX_LEVELS; i++) t.scale_factors[i] = pKF->mvScaleFactors[i];
      for (size_t i = 0; i < pKF->mvInvLevelSigma2.size() && i < FB_MAX_LEVELS; i++) t.inv_level_sigma2[i] = pKF->mvInvLevelSigma2[i];
      t.log_scale_factor = pKF->mfLogScaleFactor; t.n_levels = pKF->mnScaleLevels;
    }
    KfFlat(const KfFlat &) = delete;
  };
  struct MpFlat {
    int32_t n;
    std::vector<uint8_t> valid, desc;
    std::vector<float> xw, normal, mx, mn;
    template <typename Pred> MpFlat(const std::vector<MapPoint *> &v, Pred ok) : n((int32_t)v.size()) {
      const size_t m = v.empty() ? 1 : v.size();
      valid.assign(m, 0); desc.assign(m * 32, 0); xw.assign(m * 3, 0.f); normal.assign(m * 3, 0.f); mx.assign(m, 0.f); mn.assign(m, 0.f);
      for (size_t i = 0; i < v.size(); i++) {
        MapPoint *p = v[i];
        valid[i] = ok(p) ? 1 : 0;
        if (!p) continue;
        std::memcpy(&xw[3 * i], p->mWorldPos, 12);
        std::memcpy(&normal[3 * i], p->mNormalVector, 12);
        std::memcpy(&desc[32 * i], p->mDescriptor, 32);
        mx[i] = p->mfMaxDistance; mn[i] = p->mfMinDistance;
      }
    }
    MpFlat(const MpFlat &) = delete;
    fb_mp_list view() const {
      return {(int32_t)valid.size(), &n, valid.data(), xw.data(), normal.data(), mx.data(), mn.data(), desc.data()};
    }
  };

  fb_matcher_params m_;
};

// ---- Optimizer ---------------------------------------------------------------------------------------------------
class Optimizer {
 public:
  // points[i] / birdPoints[i] describe pFrame->mvpMapPoints[i] / mvpMapPointsBird[i]
  static int PoseOptimization(Frame *f, const std::vector<MapPointRef> &points) {
    return run(f, FB_POSE_FRONT, points, {}, 1.f, 1.f);
  }
  static int PoseOptimizationWithBird(Frame *f, const std::vector<MapPointRef> &points,
                                      const std::vector<MapPointRef> &birdPoints, float wB = 1.f, float wF = 1.f) {
    return run(f, FB_POSE_FRONT_BIRD, points, birdPoints, wB, wF);
  }
  static int BirdOptimization(Frame *f, const std::vector<MapPointRef> &birdPoints, float wB = 1.f) {
    return run(f, FB_POSE_BIRD, {}, birdPoints, wB, 1.f);
  }
  // LocalBundleAdjustment / LocalBundleAdjustmentWithOdom on an already flattened graph
  static void LocalBundleAdjustment(fb_local_ba_args &graph, bool *pbStopFlag) {
    graph.with_odom = 0;
    graph.stop_flag = reinterpret_cast<volatile uint8_t *>(pbStopFlag);
    check(fb_local_ba(&graph));
  }
  static void LocalBundleAdjustmentWithOdom(fb_local_ba_args &graph, bool *pbStopFlag, float wF = 1.f, float wB = 1.f,
                                            float wP = 3.f) {
    graph.with_odom = 1;
    graph.wF = wF; graph.wB = wB; graph.wP = wP;
    graph.stop_flag = reinterpret_cast<volatile uint8_t *>(pbStopFlag);
    check(fb_local_ba(&graph));
  }

  // ---- the reference signatures on the map stand-ins of fishbird_map.hpp --------------------------------------------
  // LocalBundleAdjustment(KeyFrame *pKF, bool *pbStopFlag, Map *pMap), Optimizer.cc:838-1165
  static void LocalBundleAdjustment(KeyFrame *pKF, bool *pbStopFlag, Map * /*pMap*/) {
    BAGraph G;
    collectLocalGraph(pKF, false, G);
    if (pbStopFlag && *pbStopFlag) return;  // :1042-1044
    fb_local_ba_args a = G.args(pKF, 0, 1.f, 1.f, 3.f, pbStopFlag);
    check(fb_local_ba(&a));
    writeBackLocal(G, false);
  }
  // LocalBundleAdjustmentWithOdom(pKF, pbStopFlag, pMap, wF, wB, wP), Optimizer.cc:2137-2670
  static void LocalBundleAdjustmentWithOdom(KeyFrame *pKF, bool *pbStopFlag, Map * /*pMap*/, float wF = 1.f, float wB = 1.f,
                                            float wP = 3.f) {
    BAGraph G;
    collectLocalGraph(pKF, switches().bHaveBird, G);
    if (switches().bTightCouple) G.addOdometryChain(wP);
    if (pbStopFlag && *pbStopFlag) return;  // :2497-2499
    fb_local_ba_args a = G.args(pKF, 1, wF, wB, wP, pbStopFlag);
    check(fb_local_ba(&a));
    writeBackLocal(G, switches().bHaveBird);
  }
  // BundleAdjustmentWithOdom(vpKFs, vpMP, vpMPB, nIterations, pbStopFlag, nLoopKF, bRobust, wF, wB, wP), Optimizer.cc:1787-2135
  static void BundleAdjustmentWithOdom(const std::vector<KeyFrame *> &vpKFs, const std::vector<MapPoint *> &vpMP,
                                       const std::vector<MapPointBird *> &vpMPB, int nIterations = 5, bool *pbStopFlag = nullptr,
                                       unsigned long nLoopKF = 0, bool bRobust = true, float wF = 1.f, float wB = 1.f,
                                       float wP = 3.f) {
    BAGraph G;
    long maxKFid = 0;
    const KeyFrame *first = nullptr;
    for (size_t i = 0; i < vpKFs.size(); i++) {
      KeyFrame *pKF = vpKFs[i];
      if (pKF->isBad()) continue;
      G.addKeyFrame(pKF, pKF->mnId == 0);
      if (!first) first = pKF;
      if ((long)pKF->mnId > maxKFid) maxKFid = (long)pKF->mnId;
    }
    G.nLocal = G.kfs.size();
    if (!first) return;
    std::vector<bool> vbNotIncludedMP(vpMP.size(), true), vbNotIncludedMPBird(vpMPB.size(), true);
    for (size_t i = 0; i < vpMP.size(); i++) {
      if (vpMP[i]->isBad()) continue;
      if (G.addMapPoint(vpMP[i], maxKFid) == 0) G.popMapPoint();
      else vbNotIncludedMP[i] = false;
    }
    if (switches().bHaveBird)
      for (size_t i = 0; i < vpMPB.size(); i++) {
        if (vpMPB[i]->isBad()) continue;
        if (G.addMapPointBird(vpMPB[i]) == 0) G.popMapPointBird();
        else vbNotIncludedMPBird[i] = false;
      }
    fb_local_ba_args a = G.args(first, 1, wF, wB, wP, pbStopFlag);
    check(fb_global_ba(&a, nIterations, bRobust ? 1 : 0));
    for (size_t k = 0; k < G.kfs.size(); k++) {
      KeyFrame *pKF = G.kfs[k];
      if (nLoopKF == 0) pKF->SetPose(&G.kfTcw[12 * k]);
      else { std::memcpy(pKF->mTcwGBA, &G.kfTcw[12 * k], 48); pKF->mnBAGlobalForKF = nLoopKF; }
    }
    for (size_t j = 0; j < G.mps.size(); j++) {
      MapPoint *pMP = G.mps[j];
      if (nLoopKF == 0) { pMP->SetWorldPos(&G.mpXw[3 * j]); pMP->UpdateNormalAndDepth(); }
      else { std::memcpy(pMP->mPosGBA, &G.mpXw[3 * j], 12); pMP->mnBAGlobalForKF = nLoopKF; }
    }
    for (size_t j = 0; j < G.mpbs.size(); j++) {
      MapPointBird *pMPB = G.mpbs[j];
      if (nLoopKF == 0) pMPB->SetWorldPos(&G.mpbXw[3 * j]);
      else { std::memcpy(pMPB->mPosGBA, &G.mpbXw[3 * j], 12); pMPB->mnBAGlobalForKF = nLoopKF; }
    }
  }
  // GlobalBundleAdjustemntWithOdom(pMap, nIterations, pbStopFlag, nLoopKF, bRobust), Optimizer.cc:1778-1785 (sic)
  static void GlobalBundleAdjustemntWithOdom(Map *pMap, int nIterations = 5, bool *pbStopFlag = nullptr,
                                             unsigned long nLoopKF = 0, bool bRobust = true) {
    BundleAdjustmentWithOdom(pMap->GetAllKeyFrames(), pMap->GetAllMapPoints(), pMap->GetAllMapPointsBird(), nIterations,
                             pbStopFlag, nLoopKF, bRobust);
  }

 private:
  static int run(Frame *f, int mode, const std::vector<MapPointRef> &points, const std::vector<MapPointRef> &birdPoints,
                 float wB, float wF) {
    const int32_t N = mode == FB_POSE_BIRD ? 0 : f->N(), NB = mode == FB_POSE_FRONT ? 0 : f->Nbird();
    std::vector<float> fxw((size_t)N * 3 + 3), fobs((size_t)N * 2 + 2), finf(N + 1), bxw((size_t)NB * 3 + 3), binf(NB + 1);
    std::vector<uint8_t> fvalid(N + 1), bvalid(NB + 1);
    for (int i = 0; i < N; i++) {
      fvalid[i] = points[i].valid;
      std::memcpy(&fxw[3 * i], points[i].Xw, 12);
      fobs[2 * i] = f->mvKeysUn[i].x; fobs[2 * i + 1] = f->mvKeysUn[i].y;
      finf[i] = f->mvInvLevelSigma2[f->mvKeysUn[i].octave];
    }
    for (int i = 0; i < NB; i++) {
      bvalid[i] = birdPoints[i].valid;
      std::memcpy(&bxw[3 * i], birdPoints[i].Xw, 12);
      binf[i] = f->mvInvLevelSigma2[f->mvKeysBird[i].octave];
    }
    f->mvbOutlier.resize(N + (N == 0), 0);
    f->mvBirdOutlier.resize(NB + (NB == 0), 0);
    fb_pose_opt_args a{};
    a.batch = 1; a.mode = mode; a.front_stride = N; a.bird_stride = NB;
    a.fx = f->fx; a.fy = f->fy; a.cx = f->cx; a.cy = f->cy; a.wF = wF; a.wB = wB;
    a.n_front = &N; a.front_xw = fxw.data(); a.front_obs = fobs.data(); a.front_inv_sigma2 = finf.data();
    a.front_valid = fvalid.data();
    a.n_bird = &NB; a.bird_xw = bxw.data(); a.bird_xc = f->mvKeysBirdCamXYZ.empty() ? bxw.data() : f->mvKeysBirdCamXYZ.data();
    a.bird_inv_sigma2 = binf.data(); a.bird_valid = bvalid.data(); a.bird_outlier = f->mvBirdOutlier.data();
    a.Tcw = f->mTcw; a.front_outlier = f->mvbOutlier.data();
    int32_t ninl = 0;
    a.ninliers = &ninl;
    check(fb_pose_opt(&a));
    return ninl;
  }
};

}  // namespace fishbird
#endif
