// fishbird_host.hpp -- C++ host side above the C-ABI (include/fishbird.h), mirroring the reference's
// operator interface for the hot path: same class names, argument meaning and return values as
//   ORBextractor  (include/ORBextractor.h:51-85)
//   ORBmatcher    (include/ORBmatcher.h:41-88)      SearchByProjection x2, SearchByBoW(KF, F), SearchForInitialization,
//                                                   BirdMapPointMatch, BirdviewMatch
//   Optimizer     (include/Optimizer.h:40-68)       PoseOptimization, PoseOptimizationWithBird, BirdOptimization,
//                                                   LocalBundleAdjustment[WithOdom] on a flattened graph
// on plain-old-data frames.  The key-frame / map side entry points (Fuse, SearchBySim3, ... and the graph collection of
// the BA functions) walk KeyFrame / MapPoint / Map objects, which are out of scope: INTEGRATION.md shows
// the gather -> C-ABI call -> scatter shim a maintainer adds inside the reference's own classes (the reference's Frame/KeyFrame/MapPoint own OpenCV and graph state that stays in the
// host application).  Header only; link with -lfishbird_hip.  Errors throw std::runtime_error(fb_last_error()).
#ifndef FISHBIRD_HOST_HPP_
#define FISHBIRD_HOST_HPP_

#include <cmath>
#include <cstdint>
#include <cstring>
#include <map>
#include <stdexcept>
#include <vector>

#include "../../include/fishbird.h"

namespace fishbird {

// DBoW2::BowVector / FeatureVector (Thirdparty/DBoW2/DBoW2/BowVector.h:54, FeatureVector.h:22): ordered maps
typedef std::map<unsigned, double> BowVector;
typedef std::map<unsigned, std::vector<unsigned>> FeatureVector;

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
  fb_orb *handle() const { return h_; }              // for the device-resident Frame (DeviceFrame below)
  const fb_orb_params &params() const { return p_; }

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
  // Frame::ComputeBoW (Frame.cc:628-635): only when mBowVec is still empty
  void ComputeBoW(Frame &F) const { if (F.mBowVec.empty()) transform(F.mDescriptors, F.mBowVec, F.mFeatVec, 4); }
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

  // SearchByBoW(pKF, F, vpMapPointMatches), ORBmatcher.cc:160-289, on POD frames: KF plays the key frame (its mFeatVec and
  // key points), kfHasMapPoint[i] = vpMapPointsKF[i] && !isBad().  matchFtoKF[iF] = key-frame feature whose MapPoint lands
  // in vpMapPointMatches[iF] (-1 = NULL); returns nmatches.
  int SearchByBoW(const Frame &KF, const std::vector<uint8_t> &kfHasMapPoint, Frame &F, std::vector<int32_t> &matchFtoKF) const {
    const int32_t NK = KF.N(), NF = F.N();
    matchFtoKF.assign(NF, -1);
    if (NK == 0 || NF == 0) return 0;
    FvFlat fk(KF.mFeatVec), ff(F.mFeatVec);
    int32_t n = 0;
    fb_bow_args a{};
    a.batch = 1; a.kf_stride = NK; a.f_stride = NF;
    a.n_kf = &NK; a.kf_kps = KF.mvKeysUn.data(); a.kf_desc = KF.mDescriptors.data(); a.kf_has_mp = kfHasMapPoint.data(); a.kf_fv = fk.view();
    a.n_f = &NF; a.f_kps = F.mvKeysUn.data(); a.f_desc = F.mDescriptors.data(); a.f_fv = ff.view();
    a.matcher = m_; a.match_f_to_kf = matchFtoKF.data(); a.nmatches = &n;
    check(fb_match_bow(&a));
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
    // only the edge family the mode optimises is (re)sized: BirdOptimization (Optimizer.cc:708-835) never touches
    // mvbOutlier, PoseOptimization (:246-475) never touches mvBirdOutlier, and Tracking indexes both right after
    if (mode != FB_POSE_BIRD) f->mvbOutlier.resize(N, 0);
    if (mode != FB_POSE_FRONT) f->mvBirdOutlier.resize(NB, 0);
    uint8_t unusedFamily[4] = {0, 0, 0, 0};
    fb_pose_opt_args a{};
    a.batch = 1; a.mode = mode; a.front_stride = N; a.bird_stride = NB;
    a.fx = f->fx; a.fy = f->fy; a.cx = f->cx; a.cy = f->cy; a.wF = wF; a.wB = wB;
    a.n_front = &N; a.front_xw = fxw.data(); a.front_obs = fobs.data(); a.front_inv_sigma2 = finf.data();
    a.front_valid = fvalid.data();
    a.n_bird = &NB; a.bird_xw = bxw.data(); a.bird_xc = f->mvKeysBirdCamXYZ.empty() ? bxw.data() : f->mvKeysBirdCamXYZ.data();
    a.bird_inv_sigma2 = binf.data(); a.bird_valid = bvalid.data();
    a.bird_outlier = (mode == FB_POSE_FRONT || NB == 0) ? unusedFamily : f->mvBirdOutlier.data();
    a.Tcw = f->mTcw; a.front_outlier = (mode == FB_POSE_BIRD || N == 0) ? unusedFamily : f->mvbOutlier.data();
    int32_t ninl = 0;
    a.ninliers = &ninl;
    check(fb_pose_opt(&a));
    return ninl;
  }
};

// ---- the device-resident Frame and the per-frame chain ------------------------------------------------------------
// DeviceFrame = the reference's Frame with its per-key-point members in HBM (fb_frame, include/fishbird.h); the methods
// below carry the names of the reference functions they stand for and only ENQUEUE on `stream`.  TrackedFrame() is what
// Tracking::Track runs for a frame in state OK (TrackWithMotionModel + TrackLocalMap, Tracking.cc:1312-1441) followed by the
// one read-back the state machine needs; retries / LOST handling stay with the caller.
struct DeviceMap {                 // the caller's MapPoint / MapPointBird tables on the device + the local-map index lists
  fb_map_points points{};
  fb_map_points_bird birdPoints{};
  const int32_t *localPoints = nullptr, *nLocalPoints = nullptr;      // mvpLocalMapPoints (nullptr = the whole table)
  const int32_t *localBirdPoints = nullptr, *nLocalBirdPoints = nullptr;  // Map::GetLocalMapPointsBird()
};

class DeviceFrame {
 public:
  explicit DeviceFrame(const fb_frame_params &p) : p_(p) { check(fb_frame_create(&p_, &h_)); }
  ~DeviceFrame() { fb_frame_destroy(h_); }
  DeviceFrame(const DeviceFrame &) = delete;
  DeviceFrame &operator=(const DeviceFrame &) = delete;
  fb_frame *handle() const { return h_; }
  int batch() const { return p_.batch; }

  // Frame::Frame(imGray, BirdGray, ..., birdviewmask, ..., birdviewContourICP, ..., extractor, ...) from host images
  void Construct(ORBextractor &front, ORBextractor &bird, const uint8_t *imGray, int stride, const uint8_t *birdGray, int birdStride,
                 const uint8_t *contourICP, const uint8_t *mask, void *stream = nullptr) {
    check(fb_frame_extract(h_, front.handle(), bird.handle(), imGray, stride, birdGray, birdStride, contourICP, mask, stream));
  }
  void SetPose(const float *d_Tcw, void *stream = nullptr) { check(fb_frame_set_pose_dev(h_, d_Tcw, stream)); }

  struct TrackResult {             // what Tracking reads after a frame (one synchronisation)
    std::vector<int32_t> counts;   // [FB_CNT_COUNT][batch]
    std::vector<float> Tcw;        // [batch][12]
    int count(int slot, int b = 0) const { return counts[(size_t)slot * (counts.size() / FB_CNT_COUNT) + b]; }
    bool trackedWithMotionModel(int b = 0) const { return count(FB_CNT_PROJ_MATCHES, b) >= 20 && count(FB_CNT_MATCHES_MAP, b) >= 10; }  // Tracking.cc:1351,1384
    bool trackedLocalMap(int b = 0) const { return count(FB_CNT_MATCHES_INLIERS, b) >= 30; }                                              // :1438
    // Tracking::BirdNeedKF (Tracking.cc:2063-2083) on the counters of a frame tracked with TrackUsingBird
    bool birdNeedKF(int b = 0) const {
      const int kf = count(FB_CNT_BIRD_KF_MATCHES, b), numPt = count(FB_CNT_BIRD_POINTS_FINAL, b);
      return kf < 0.7 * numPt || (kf < 10 && numPt > 10);
    }
    bool trackedReferenceKeyFrame(int b = 0) const { return count(FB_CNT_BOW_MATCHES, b) >= 15 && count(FB_CNT_MATCHES_MAP, b) >= 10; }   // :1212,1243
  };
  // Tracking::Track, state OK: this frame against `last`; d_deltaT = rows 0..2 of detlaT (Tracking.cc:1316) on the device
  TrackResult TrackedFrame(DeviceFrame &last, const DeviceMap &map, const float *d_deltaT, float wB = 1.f, float wF = 1.f, void *stream = nullptr) {
    const fb_track_args T = Args(map, d_deltaT, wB, wF);
    check(fb_frame_track_dev(h_, last.h_, &T, stream));
    return Result(stream);
  }

  // the pieces of Tracking::Track a state machine chooses between (Tracking.cc:529-540, 640-643), each one enqueue + one read-back:
  //   bOK = TrackWithMotionModel(); if (!bOK) bOK = TrackReferenceKeyFrame(); if (bOK) bOK = TrackLocalMap();
  TrackResult TrackWithMotionModel(DeviceFrame &last, const DeviceMap &map, const float *d_deltaT, float wB = 1.f, float wF = 1.f, void *stream = nullptr) {
    const fb_track_args T = Args(map, d_deltaT, wB, wF);
    check(fb_frame_track_motion_model_dev(h_, last.h_, &T, stream));
    return Result(stream);
  }
  // mpReferenceKF = the frame handle the key frame was made from (CopyFrom + ComputeBoW); d_deltaT = detlaT of Tracking.cc:1185
  TrackResult TrackReferenceKeyFrame(DeviceFrame &referenceKF, DeviceFrame &tmpRefFrame, const fb_vocabulary &d_voc, const DeviceMap &map,
                                     const float *d_deltaT, float wB = 1.f, float wF = 1.f, void *stream = nullptr) {
    const fb_track_args T = Args(map, d_deltaT, wB, wF);
    check(fb_frame_track_reference_dev(h_, referenceKF.h_, tmpRefFrame.h_, &d_voc, &T, stream));
    return Result(stream);
  }
  // Tracking::TrackUsingBird (Tracking.cc:2014-2061): poseSource = mpReferenceKF's handle (IsbirdWithRefKF == 1) or tmpRefFrame
  TrackResult TrackUsingBird(DeviceFrame &poseSource, DeviceFrame &tmpRefFrame, const DeviceMap &map, const float *d_deltaT, void *stream = nullptr) {
    const fb_track_args T = Args(map, d_deltaT, 1.f, 1.f);
    check(fb_frame_track_using_bird_dev(h_, poseSource.h_, tmpRefFrame.h_, &T, stream));
    return Result(stream);
  }
  TrackResult TrackLocalMap(DeviceFrame &tmpRefFrame, const DeviceMap &map, float wB = 1.f, float wF = 1.f, void *stream = nullptr) {
    const fb_track_args T = Args(map, nullptr, wB, wF);
    check(fb_frame_track_local_map_dev(h_, tmpRefFrame.h_, &T, stream));
    return Result(stream);
  }
  void DropOutliers(void *stream = nullptr) { check(fb_frame_drop_outliers_dev(h_, stream)); }  // Tracking.cc:721-725, after the key-frame decision
  void ComputeBoW(const fb_vocabulary &d_voc, void *stream = nullptr) { check(fb_frame_compute_bow_dev(h_, &d_voc, stream)); }  // Frame.cc:628-635
  void CopyFrom(const DeviceFrame &f, void *stream = nullptr) { check(fb_frame_copy_dev(h_, f.h_, stream)); }  // Frame(const Frame&), KeyFrame(Frame&, ...)

  TrackResult Result(void *stream = nullptr) {
    TrackResult r;
    r.counts.resize((size_t)FB_CNT_COUNT * p_.batch);
    r.Tcw.resize((size_t)12 * p_.batch);
    check(fb_frame_counts(h_, r.counts.data(), r.Tcw.data(), stream));
    return r;
  }

 private:
  static fb_track_args Args(const DeviceMap &map, const float *d_deltaT, float wB, float wF) {
    fb_track_args T{};
    T.map = map.points; T.mpb = map.birdPoints; T.d_delta = d_deltaT;
    T.d_local_mp = map.localPoints; T.d_n_local_mp = map.nLocalPoints; T.d_local_mpb = map.localBirdPoints; T.d_n_local_mpb = map.nLocalBirdPoints;
    T.wB = wB; T.wF = wF;
    T.gate_local_map = 1;       // every step follows its sequence's own bOK
    T.defer_outlier_drop = 0;   // set to 1 by a caller that copies its key frame before DropOutliers()
    return T;
  }
  fb_frame_params p_;
  fb_frame *h_ = nullptr;
};

// the one-call-per-reference-function forms on device frames (results stay in the frames; counts via fb_frame_counts)
inline void SearchByProjection(const ORBmatcher &, DeviceFrame &cur, const DeviceFrame &last, const DeviceMap &map, float th, float nnratio = 0.9f,
                               bool checkOri = true, void *stream = nullptr) {  // ORBmatcher::SearchByProjection(Frame&, const Frame&, th, mono)
  const fb_matcher_params m{nnratio, checkOri ? 1 : 0};
  check(fb_frame_search_by_projection_dev(cur.handle(), last.handle(), &map.points, th, &m, stream));
}
inline void SearchByBoW(const ORBmatcher &, const DeviceFrame &kf, DeviceFrame &cur, const DeviceMap &map, float nnratio = 0.7f, bool checkOri = true,
                        int minMatches = 15, void *stream = nullptr) {  // ORBmatcher::SearchByBoW(KeyFrame*, Frame&, vpMapPointMatches) + Tracking.cc:1212-1215
  const fb_matcher_params m{nnratio, checkOri ? 1 : 0};
  check(fb_frame_search_by_bow_dev(cur.handle(), kf.handle(), &map.points, &m, minMatches, stream));
}
inline void PoseOptimizationWithBird(DeviceFrame &f, const DeviceMap &map, float wB = 1.f, float wF = 1.f, int which = 0, void *stream = nullptr) {
  check(fb_frame_pose_optimization_dev(f.handle(), &map.points, &map.birdPoints, FB_POSE_FRONT_BIRD, wB, wF, which, stream));
}

}  // namespace fishbird
#endif
