// fishbird_map.hpp -- the map side of the optimiser boundary: plain C++ stand-ins for the parts of KeyFrame,
// MapPoint, MapPointBird and Map that Optimizer::LocalBundleAdjustment[WithOdom] and BundleAdjustmentWithOdom read
// and write, and the graph collection / write-back code of those functions (everything of Optimizer.cc:838-1165,
// 1778-2135 and 2137-2670 that is NOT g2o).  The optimisation itself is fb_local_ba / fb_global_ba of the C-ABI.
//
// Differences from the reference that a maintainer must know:
//  * observations are std::map<KeyFrame*, size_t> ordered by mnId here; the reference orders them by pointer value
//    (allocation order).  The order only decides the insertion order of fixed cameras and of the edges of one
//    landmark, i.e. floating-point summation order.
//  * there are no mutexes: the caller serialises access the way pMap->mMutexMapUpdate does (Optimizer.cc:1133,2613).
#ifndef FISHBIRD_MAP_HPP_
#define FISHBIRD_MAP_HPP_

#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <list>
#include <map>
#include <set>
#include <stdexcept>
#include <vector>

#include "../../include/fishbird.h"

namespace fishbird {

struct KeyFrame;
struct Map;
struct KFIdLess {
  inline bool operator()(const KeyFrame *a, const KeyFrame *b) const;
};
typedef std::map<KeyFrame *, size_t, KFIdLess> ObservationMap;
// DBoW2::FeatureVector (Thirdparty/DBoW2/DBoW2/FeatureVector.h:23): NodeId -> feature indices in addFeature order
typedef std::map<uint32_t, std::vector<unsigned>> FeatureVector;
// DBoW2::BowVector (BowVector.h): WordId -> value
typedef std::map<uint32_t, double> BowVector;

// Frame::bHaveBird / bTightCouple / bTightCouple2 (static switches read by the optimiser, Optimizer.cc:2190,2417,2452)
struct OptimizerSwitches {
  bool bHaveBird = true, bTightCouple = true, bTightCouple2 = true;
};
inline OptimizerSwitches &switches() {
  static OptimizerSwitches s;
  return s;
}

// Frame::CalculateExtrinsics (Frame.cc:1015-1037): camera <-> base footprint, float 3x4 (rows 0..2)
struct Extrinsics {
  float Tbc[12], Tcb[12];
  Extrinsics() {
    const float tbc[3] = {3.747f, 0.040f, 0.736f};
    double qx = 0.631, qy = -0.623, qz = 0.325, qw = -0.330;
    const double qn = std::sqrt(qx * qx + qy * qy + qz * qz + qw * qw);
    qx /= qn; qy /= qn; qz /= qn; qw /= qn;
    const float R[9] = {(float)(1 - 2 * (qy * qy + qz * qz)), (float)(2 * (qx * qy - qw * qz)), (float)(2 * (qx * qz + qw * qy)),
                        (float)(2 * (qx * qy + qw * qz)), (float)(1 - 2 * (qx * qx + qz * qz)), (float)(2 * (qy * qz - qw * qx)),
                        (float)(2 * (qx * qz - qw * qy)), (float)(2 * (qy * qz + qw * qx)), (float)(1 - 2 * (qx * qx + qy * qy))};
    for (int r = 0; r < 3; r++) {
      for (int c = 0; c < 3; c++) { Tbc[r * 4 + c] = R[r * 3 + c]; Tcb[r * 4 + c] = R[c * 3 + r]; }
      Tbc[r * 4 + 3] = tbc[r];
    }
    for (int r = 0; r < 3; r++) {  // tcb = -Rcb*tbc on a materialised Rcb: cv::gemm small-matrix path, float sums
      const float s = Tcb[r * 4 + 0] * tbc[0] + Tcb[r * 4 + 1] * tbc[1] + Tcb[r * 4 + 2] * tbc[2];
      Tcb[r * 4 + 3] = -s;
    }
  }
};
inline const Extrinsics &extrinsics() {
  static const Extrinsics e;
  return e;
}

// Frame::GetTransformFromOdometer (Frame.cc:1049-1067): planar odometer poses (x, y, theta) -> Tcb*T12b*Tbc, float
inline void GetTransformFromOdometer(const double p1[3], const double p2[3], float T12c[12]) {
  const double x1 = p1[0], y1 = p1[1], th1 = p1[2], x2 = p2[0], y2 = p2[1], th2 = p2[2];
  const double th12 = th2 - th1;
  const double x12 = (x2 - x1) * std::cos(th1) + (y2 - y1) * std::sin(th1);
  const double y12 = (y2 - y1) * std::cos(th1) - (x2 - x1) * std::sin(th1);
  const float Tb[16] = {(float)std::cos(th12), (float)-std::sin(th12), 0, (float)x12,
                        (float)std::sin(th12), (float)std::cos(th12), 0, (float)y12, 0, 0, 1, 0, 0, 0, 0, 1};
  const Extrinsics &E = extrinsics();
  float A[16], Bm[16], M[16], R[16];
  std::memcpy(A, E.Tcb, 48); A[12] = A[13] = A[14] = 0; A[15] = 1;
  std::memcpy(Bm, E.Tbc, 48); Bm[12] = Bm[13] = Bm[14] = 0; Bm[15] = 1;
  auto mul = [](const float *a, const float *b, float *d) {  // 4x4 float product, float sums (cv::gemm small path)
    for (int r = 0; r < 4; r++)
      for (int c = 0; c < 4; c++) d[r * 4 + c] = a[r * 4] * b[c] + a[r * 4 + 1] * b[4 + c] + a[r * 4 + 2] * b[8 + c] + a[r * 4 + 3] * b[12 + c];
  };
  mul(A, Tb, M);
  mul(M, Bm, R);
  std::memcpy(T12c, R, 48);
}

// ---- MapPoint (include/MapPoint.h, src/MapPoint.cc) ---------------------------------------------------------------
struct MapPoint {
  unsigned long mnId = 0;
  float mWorldPos[3] = {0, 0, 0};
  float mNormalVector[3] = {0, 0, 0};
  float mfMinDistance = 0, mfMaxDistance = 0;
  KeyFrame *mpRefKF = nullptr;
  Map *mpMap = nullptr;
  ObservationMap mObservations;
  int nObs = 0;
  bool mbBad = false;
  unsigned long mnBALocalForKF = 0;   // MapPoint.cc:35 initialises these to 0
  float mPosGBA[3] = {0, 0, 0};
  unsigned long mnBAGlobalForKF = 0;
  uint8_t mDescriptor[32] = {0};      // GetDescriptor()
  int mnVisible = 1, mnFound = 1;
  MapPoint *mpReplaced = nullptr;

  bool isBad() const { return mbBad; }
  int Observations() const { return nObs; }
  ObservationMap GetObservations() const { return mObservations; }
  int GetIndexInKeyFrame(KeyFrame *pKF) const {
    ObservationMap::const_iterator it = mObservations.find(pKF);
    return it == mObservations.end() ? -1 : (int)it->second;
  }
  bool IsInKeyFrame(KeyFrame *pKF) const { return mObservations.count(pKF) != 0; }
  void IncreaseVisible(int n = 1) { mnVisible += n; }
  void IncreaseFound(int n = 1) { mnFound += n; }
  void SetWorldPos(const float p[3]) { std::memcpy(mWorldPos, p, 12); }
  inline void Replace(MapPoint *pMP);                          // MapPoint.cc:172-217
  inline void ComputeDistinctiveDescriptors();                 // MapPoint.cc:242-307, through fb_distinctive_descriptors
  inline void AddObservation(KeyFrame *pKF, size_t idx);      // MapPoint.cc:98-110
  inline void EraseObservation(KeyFrame *pKF);                 // MapPoint.cc:112-138
  inline void SetBadFlag();                                    // MapPoint.cc:152-170
  inline void UpdateNormalAndDepth();                          // MapPoint.cc:330-372
};

// ---- MapPointBird (src/MapPointBird.cc) ---------------------------------------------------------------------------
struct MapPointBird {
  unsigned long mnId = 0;
  float mWorldPos[3] = {0, 0, 0};
  KeyFrame *mpRefKF = nullptr;
  ObservationMap mObservations;
  int nObs = 0;
  bool mbBad = false;
  unsigned long mnBALocalForKF = 0;
  float mPosGBA[3] = {0, 0, 0};
  unsigned long mnBAGlobalForKF = 0;

  bool isBad() const { return mbBad; }
  ObservationMap GetObservations() const { return mObservations; }
  int GetIndexInKeyFrame(KeyFrame *pKF) const {
    ObservationMap::const_iterator it = mObservations.find(pKF);
    return it == mObservations.end() ? -1 : (int)it->second;
  }
  void SetWorldPos(const float p[3]) { std::memcpy(mWorldPos, p, 12); }
  void AddObservation(KeyFrame *pKF, size_t idx) {
    if (mObservations.count(pKF)) return;
    mObservations[pKF] = idx;
    nObs++;
  }
  void EraseObservation(KeyFrame *pKF) {  // MapPointBird.cc:44-56 (no bad flag on few observations)
    if (!mObservations.count(pKF)) return;
    nObs--;
    mObservations.erase(pKF);
    if (mpRefKF == pKF) mpRefKF = mObservations.empty() ? nullptr : mObservations.begin()->first;
  }
};

// ---- KeyFrame (include/KeyFrame.h) --------------------------------------------------------------------------------
struct KeyFrame {
  unsigned long mnId = 0;
  bool isInit = false;  // fixed in the local BA (Optimizer.cc:915,2254)
  bool mbBad = false;
  float fx = 0, fy = 0, cx = 0, cy = 0;
  float Tcw[12] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0};
  float Ow[3] = {0, 0, 0};
  double mGtPose[3] = {0, 0, 0};  // odometer pose (x, y, theta)
  int mnScaleLevels = 0;
  std::vector<float> mvScaleFactors, mvInvLevelSigma2;
  std::vector<fb_keypoint> mvKeysUn;
  std::vector<MapPoint *> mvpMapPoints;
  std::vector<fb_keypoint> mvKeysBird;
  std::vector<float> mvKeysBirdCamXYZ;  // N x 3
  std::vector<MapPointBird *> mvpMapPointsBird;
  std::vector<KeyFrame *> mvpOrderedConnectedKeyFrames;  // maintained by the caller (KeyFrame::UpdateBestCovisibles)
  unsigned long mnBALocalForKF = 0, mnBAFixedForKF = 0, mnBAGlobalForKF = 0;
  float mTcwGBA[12] = {0};
  // matcher side (KeyFrame.h:170-213): descriptors, BoW feature vector, the feature grid copied from the Frame
  std::vector<uint8_t> mDescriptors;  // N x 32
  BowVector mBowVec;
  FeatureVector mFeatVec;
  std::vector<float> mvLevelSigma2;
  float mfLogScaleFactor = 0;
  int mnMinX = 0, mnMinY = 0, mnMaxX = 0, mnMaxY = 0;  // const int in the reference (truncated Frame bounds)
  int mnGridCols = 64, mnGridRows = 48;
  float mfGridElementWidthInv = 0, mfGridElementHeightInv = 0;
  float mnFrameMinX = 0, mnFrameMinY = 0;  // the Frame's float bounds the grid was built with (Frame.cc:381-411)
  std::vector<int32_t> gridStart, gridItems;  // mGrid as CSR, cell id = ix*rows+iy

  int N() const { return (int)mvKeysUn.size(); }
  MapPoint *GetMapPoint(size_t idx) const { return mvpMapPoints[idx]; }
  inline std::set<MapPoint *> GetMapPoints() const;  // KeyFrame.cc:337-350: the non-NULL, non-bad ones
  void AddMapPoint(MapPoint *pMP, size_t idx) { mvpMapPoints[idx] = pMP; }
  void ReplaceMapPointMatch(size_t idx, MapPoint *pMP) { mvpMapPoints[idx] = pMP; }
  fb_grid_geom gridGeom() const {
    return {mnFrameMinX, mnFrameMinY, mfGridElementWidthInv, mfGridElementHeightInv, mnGridCols, mnGridRows};
  }
  fb_camera camera() const { return {fx, fy, cx, cy, (float)mnMinX, (float)mnMinY, (float)mnMaxX, (float)mnMaxY}; }
  // Frame::AssignFeaturesToGrid (Frame.cc:381-411; the KeyFrame copies F.mGrid, KeyFrame.cc:51-57)
  void AssignFeaturesToGrid() {
    const fb_grid_geom g = gridGeom();
    const int ncell = g.cols * g.rows;
    std::vector<std::vector<int32_t>> cells(ncell);
    for (size_t i = 0; i < mvKeysUn.size(); i++) {
      const int px = (int)std::round((mvKeysUn[i].x - g.min_x) * g.inv_w), py = (int)std::round((mvKeysUn[i].y - g.min_y) * g.inv_h);
      if (px < 0 || px >= g.cols || py < 0 || py >= g.rows) continue;
      cells[px * g.rows + py].push_back((int32_t)i);
    }
    gridStart.assign(ncell + 1, 0);
    gridItems.assign(mvKeysUn.size() ? mvKeysUn.size() : 1, 0);
    int off = 0;
    for (int c = 0; c < ncell; c++) { gridStart[c] = off; for (int32_t i : cells[c]) gridItems[off++] = i; }
    gridStart[ncell] = off;
  }

  bool isBad() const { return mbBad; }
  const float *GetPose() const { return Tcw; }
  const float *GetCameraCenter() const { return Ow; }
  std::vector<KeyFrame *> GetVectorCovisibleKeyFrames() const { return mvpOrderedConnectedKeyFrames; }
  std::vector<MapPoint *> GetMapPointMatches() const { return mvpMapPoints; }
  std::vector<MapPointBird *> GetMapPointBirdMatches() const { return mvpMapPointsBird; }
  // KeyFrame::SetPose (KeyFrame.cc:104-118): Ow = -Rwc*tcw on the materialised transpose (float sums)
  void SetPose(const float T[12]) {
    std::memcpy(Tcw, T, 48);
    for (int r = 0; r < 3; r++) {
      const float s = Tcw[0 * 4 + r] * Tcw[3] + Tcw[1 * 4 + r] * Tcw[7] + Tcw[2 * 4 + r] * Tcw[11];
      Ow[r] = -s;
    }
  }
  void EraseMapPointMatch(size_t idx) { mvpMapPoints[idx] = nullptr; }
  void EraseMapPointMatch(MapPoint *pMP) {  // KeyFrame.cc:318-323
    const int idx = pMP->GetIndexInKeyFrame(this);
    if (idx >= 0) mvpMapPoints[idx] = nullptr;
  }
  void EraseMapPointBirdMatch(MapPointBird *pMPB) {  // KeyFrame.cc:325-330
    const int idx = pMPB->GetIndexInKeyFrame(this);
    if (idx >= 0) mvpMapPointsBird[idx] = nullptr;
  }
};
inline std::set<MapPoint *> KeyFrame::GetMapPoints() const {
  std::set<MapPoint *> s;
  for (size_t i = 0; i < mvpMapPoints.size(); i++)
    if (mvpMapPoints[i] && !mvpMapPoints[i]->isBad()) s.insert(mvpMapPoints[i]);
  return s;
}
inline bool KFIdLess::operator()(const KeyFrame *a, const KeyFrame *b) const { return a->mnId < b->mnId; }

// ---- Map (include/Map.h) ------------------------------------------------------------------------------------------
struct Map {
  std::vector<KeyFrame *> mspKeyFrames;
  std::vector<MapPoint *> mspMapPoints;
  std::vector<MapPointBird *> mspMapPointsBird;
  std::vector<KeyFrame *> GetAllKeyFrames() const { return mspKeyFrames; }
  std::vector<MapPoint *> GetAllMapPoints() const { return mspMapPoints; }
  std::vector<MapPointBird *> GetAllMapPointsBird() const { return mspMapPointsBird; }
  void EraseMapPoint(MapPoint *pMP) {  // Map.cc: the set forgets the point, the object stays alive
    mspMapPoints.erase(std::remove(mspMapPoints.begin(), mspMapPoints.end(), pMP), mspMapPoints.end());
  }
};

inline void MapPoint::AddObservation(KeyFrame *pKF, size_t idx) {
  if (mObservations.count(pKF)) return;
  mObservations[pKF] = idx;
  nObs++;  // monocular observation (mvuRight < 0)
}
inline void MapPoint::EraseObservation(KeyFrame *pKF) {
  bool bBad = false;
  if (mObservations.count(pKF)) {
    nObs--;
    mObservations.erase(pKF);
    if (mpRefKF == pKF) mpRefKF = mObservations.empty() ? nullptr : mObservations.begin()->first;
    if (nObs <= 2) bBad = true;  // "If only 2 observations or less, discard point"
  }
  if (bBad) SetBadFlag();
}
inline void MapPoint::SetBadFlag() {
  ObservationMap obs;
  mbBad = true;
  obs.swap(mObservations);
  for (ObservationMap::iterator it = obs.begin(); it != obs.end(); ++it) it->first->EraseMapPointMatch(it->second);
  if (mpMap) mpMap->EraseMapPoint(this);
}
inline void MapPoint::Replace(MapPoint *pMP) {
  if (pMP->mnId == this->mnId) return;
  ObservationMap obs;
  obs.swap(mObservations);
  mbBad = true;
  const int nvisible = mnVisible, nfound = mnFound;
  mpReplaced = pMP;
  for (ObservationMap::iterator mit = obs.begin(); mit != obs.end(); ++mit) {
    KeyFrame *pKF = mit->first;
    if (!pMP->IsInKeyFrame(pKF)) {
      pKF->ReplaceMapPointMatch(mit->second, pMP);
      pMP->AddObservation(pKF, mit->second);
    } else {
      pKF->EraseMapPointMatch(mit->second);
    }
  }
  pMP->IncreaseFound(nfound);
  pMP->IncreaseVisible(nvisible);
  pMP->ComputeDistinctiveDescriptors();
  if (mpMap) mpMap->EraseMapPoint(this);
}
// The descriptor with the least median distance to the other observations (MapPoint.cc:242-307).  One point per call
// here as in the reference; a caller that touches many points batches them into one fb_distinctive_descriptors call.
inline void MapPoint::ComputeDistinctiveDescriptors() {
  if (mbBad || mObservations.empty()) return;
  std::vector<uint8_t> desc;
  for (ObservationMap::iterator mit = mObservations.begin(); mit != mObservations.end(); ++mit) {
    KeyFrame *pKF = mit->first;
    if (pKF->isBad()) continue;
    const uint8_t *d = &pKF->mDescriptors[32 * mit->second];
    desc.insert(desc.end(), d, d + 32);
  }
  if (desc.empty()) return;
  const int32_t start[2] = {0, (int32_t)(desc.size() / 32)};
  int32_t best = -1;
  if (fb_distinctive_descriptors(start, desc.data(), 1, &best) != FB_OK) throw std::runtime_error(fb_last_error());
  if (best >= 0) std::memcpy(mDescriptor, &desc[32 * (size_t)best], 32);
}
inline void MapPoint::UpdateNormalAndDepth() {
  if (mbBad || mObservations.empty() || !mpRefKF) return;
  // cv::norm accumulates in double; Mat / scalar multiplies by the float cast of the reciprocal... of a double: the
  // expression normali/cv::norm(normali) is MatExpr(normali, alpha = 1/norm) evaluated as float(x * alpha)
  float normal[3] = {0, 0, 0};
  int n = 0;
  for (ObservationMap::iterator it = mObservations.begin(); it != mObservations.end(); ++it) {
    const float *Owi = it->first->GetCameraCenter();
    const float d[3] = {mWorldPos[0] - Owi[0], mWorldPos[1] - Owi[1], mWorldPos[2] - Owi[2]};
    const double nrm = std::sqrt((double)d[0] * d[0] + (double)d[1] * d[1] + (double)d[2] * d[2]);
    const double alpha = 1.0 / nrm;
    for (int k = 0; k < 3; k++) normal[k] = normal[k] + (float)(d[k] * alpha);
    n++;
  }
  const float *Or = mpRefKF->GetCameraCenter();
  const float pc[3] = {mWorldPos[0] - Or[0], mWorldPos[1] - Or[1], mWorldPos[2] - Or[2]};
  const float dist = (float)std::sqrt((double)pc[0] * pc[0] + (double)pc[1] * pc[1] + (double)pc[2] * pc[2]);
  const int level = mpRefKF->mvKeysUn[mObservations[mpRefKF]].octave;
  const float levelScaleFactor = mpRefKF->mvScaleFactors[level];
  const int nLevels = mpRefKF->mnScaleLevels;
  mfMaxDistance = dist * levelScaleFactor;
  mfMinDistance = mfMaxDistance / mpRefKF->mvScaleFactors[nLevels - 1];
  for (int k = 0; k < 3; k++) normal[k] = (float)(normal[k] * (1.0 / n));
  std::memcpy(mNormalVector, normal, 12);
}

// ---- the flattened graph handed to the C-ABI + the back references needed for the write-back ----------------------
struct BAGraph {
  std::vector<KeyFrame *> kfs;          // vertex order: local key frames, then fixed cameras
  std::vector<uint8_t> kfFixed;
  std::vector<float> kfTcw;
  std::map<KeyFrame *, int32_t, KFIdLess> kfIndex;
  size_t nLocal = 0;
  std::vector<MapPoint *> mps;
  std::vector<float> mpXw;
  std::vector<MapPointBird *> mpbs;
  std::vector<float> mpbXw;
  std::vector<int32_t> obsKf, obsMp, bobsKf, bobsMpb, odomI, odomJ;
  std::vector<float> obsUv, obsInf, bobsXc, bobsInf, odomT;
  std::vector<double> odomInfo;
  std::vector<KeyFrame *> vpEdgeKFMono, vpEdgeKFBird;
  std::vector<MapPoint *> vpMapPointEdgeMono;
  std::vector<MapPointBird *> vpMapPointEdgeBird;
  std::vector<uint8_t> obsOutlier, bobsOutlier;

  int32_t addKeyFrame(KeyFrame *pKF, bool fixed) {
    const int32_t id = (int32_t)kfs.size();
    kfs.push_back(pKF);
    kfFixed.push_back(fixed ? 1 : 0);
    kfTcw.insert(kfTcw.end(), pKF->Tcw, pKF->Tcw + 12);
    kfIndex[pKF] = id;
    return id;
  }
  // one landmark vertex and its edges (Optimizer.cc:961-1037 / 2325-2372 / 1838-1931); returns the edge count
  int addMapPoint(MapPoint *pMP, long maxKFid = -1) {
    const int32_t id = (int32_t)mps.size();
    int nEdges = 0;
    const ObservationMap observations = pMP->GetObservations();
    for (ObservationMap::const_iterator mit = observations.begin(); mit != observations.end(); ++mit) {
      KeyFrame *pKFi = mit->first;
      if (pKFi->isBad() || (maxKFid >= 0 && (long)pKFi->mnId > maxKFid)) continue;
      std::map<KeyFrame *, int32_t, KFIdLess>::const_iterator v = kfIndex.find(pKFi);
      if (v == kfIndex.end()) continue;  // optimizer.vertex(mnId) == NULL: g2o refuses the edge
      const fb_keypoint &kpUn = pKFi->mvKeysUn[mit->second];
      obsKf.push_back(v->second);
      obsMp.push_back(id);
      obsUv.push_back(kpUn.x);
      obsUv.push_back(kpUn.y);
      obsInf.push_back(pKFi->mvInvLevelSigma2[kpUn.octave]);
      vpEdgeKFMono.push_back(pKFi);
      vpMapPointEdgeMono.push_back(pMP);
      nEdges++;
    }
    mps.push_back(pMP);
    mpXw.insert(mpXw.end(), pMP->mWorldPos, pMP->mWorldPos + 3);
    return nEdges;
  }
  void popMapPoint() {  // optimizer.removeVertex(vPoint) of a landmark without edges (Optimizer.cc:1922-1926)
    mps.pop_back();
    mpXw.resize(mpXw.size() - 3);
  }
  int addMapPointBird(MapPointBird *pMPB) {  // Optimizer.cc:2377-2414 / 1937-1999
    const int32_t id = (int32_t)mpbs.size();
    int nEdges = 0;
    const ObservationMap observations = pMPB->GetObservations();
    for (ObservationMap::const_iterator mit = observations.begin(); mit != observations.end(); ++mit) {
      KeyFrame *pKFi = mit->first;
      if (pKFi->isBad()) continue;
      std::map<KeyFrame *, int32_t, KFIdLess>::const_iterator v = kfIndex.find(pKFi);
      if (v == kfIndex.end()) continue;
      const float *pt = &pKFi->mvKeysBirdCamXYZ[3 * mit->second];
      bobsKf.push_back(v->second);
      bobsMpb.push_back(id);
      bobsXc.insert(bobsXc.end(), pt, pt + 3);
      bobsInf.push_back(pKFi->mvInvLevelSigma2[pKFi->mvKeysBird[mit->second].octave]);
      vpEdgeKFBird.push_back(pKFi);
      vpMapPointEdgeBird.push_back(pMPB);
      nEdges++;
    }
    mpbs.push_back(pMPB);
    mpbXw.insert(mpbXw.end(), pMPB->mWorldPos, pMPB->mWorldPos + 3);
    return nEdges;
  }
  void popMapPointBird() {
    mpbs.pop_back();
    mpbXw.resize(mpbXw.size() - 3);
  }
  void addOdom(KeyFrame *a, KeyFrame *b, double info) {
    float T[12];
    GetTransformFromOdometer(a->mGtPose, b->mGtPose, T);
    odomI.push_back(kfIndex[a]);
    odomJ.push_back(kfIndex[b]);
    odomT.insert(odomT.end(), T, T + 12);
    odomInfo.push_back(info);
  }
  // PoseGraph constraints over the local key frames sorted by mnId (Optimizer.cc:2417-2495)
  void addOdometryChain(float wP) {
    std::vector<KeyFrame *> v(kfs.begin(), kfs.begin() + nLocal);
    std::sort(v.begin(), v.end(), KFIdLess());
    for (size_t i = 0; i + 1 < v.size(); i++) {
      addOdom(v[i], v[i + 1], 1e4 * (double)wP);
      if (switches().bTightCouple2 && i + 2 < v.size()) {
        addOdom(v[i], v[i + 2], 2e3);
        if (i + 3 < v.size()) addOdom(v[i], v[i + 3], 1e3 * (double)wP);
      }
    }
  }
  fb_local_ba_args args(const KeyFrame *intrinsicsOf, int withOdom, float wF, float wB, float wP, bool *pbStopFlag) {
    fb_local_ba_args a;
    std::memset(&a, 0, sizeof(a));
    obsOutlier.assign(obsKf.size() + 1, 0);
    bobsOutlier.assign(bobsKf.size() + 1, 0);
    a.with_odom = withOdom;
    a.fx = intrinsicsOf->fx; a.fy = intrinsicsOf->fy; a.cx = intrinsicsOf->cx; a.cy = intrinsicsOf->cy;
    a.wF = wF; a.wB = wB; a.wP = wP;
    a.n_kf = (int32_t)kfs.size(); a.kf_Tcw = kfTcw.data(); a.kf_fixed = kfFixed.data();
    a.n_mp = (int32_t)mps.size(); a.mp_xw = mpXw.data();
    a.n_mpb = (int32_t)mpbs.size(); a.mpb_xw = mpbXw.data();
    a.n_obs = (int32_t)obsKf.size(); a.obs_kf = obsKf.data(); a.obs_mp = obsMp.data(); a.obs_uv = obsUv.data();
    a.obs_inv_sigma2 = obsInf.data();
    a.n_bobs = (int32_t)bobsKf.size(); a.bobs_kf = bobsKf.data(); a.bobs_mpb = bobsMpb.data(); a.bobs_xc = bobsXc.data();
    a.bobs_inv_sigma2 = bobsInf.data();
    a.n_odom = (int32_t)odomI.size(); a.odom_kf_i = odomI.data(); a.odom_kf_j = odomJ.data(); a.odom_Tij = odomT.data();
    a.odom_info = odomInfo.data();
    a.stop_flag = reinterpret_cast<const volatile uint8_t *>(pbStopFlag);
    a.obs_outlier = obsOutlier.data(); a.bobs_outlier = bobsOutlier.data();
    return a;
  }
};

// Local key frames, local points, fixed cameras (Optimizer.cc:841-889 / 2140-2227)
inline void collectLocalGraph(KeyFrame *pKF, bool withBird, BAGraph &G) {
  std::list<KeyFrame *> lLocalKeyFrames;
  lLocalKeyFrames.push_back(pKF);
  pKF->mnBALocalForKF = pKF->mnId;
  const std::vector<KeyFrame *> vNeighKFs = pKF->GetVectorCovisibleKeyFrames();
  for (size_t i = 0; i < vNeighKFs.size(); i++) {
    KeyFrame *pKFi = vNeighKFs[i];
    pKFi->mnBALocalForKF = pKF->mnId;
    if (!pKFi->isBad()) lLocalKeyFrames.push_back(pKFi);
  }
  std::list<MapPoint *> lLocalMapPoints;
  for (std::list<KeyFrame *>::iterator lit = lLocalKeyFrames.begin(); lit != lLocalKeyFrames.end(); ++lit) {
    const std::vector<MapPoint *> vpMPs = (*lit)->GetMapPointMatches();
    for (size_t i = 0; i < vpMPs.size(); i++) {
      MapPoint *pMP = vpMPs[i];
      if (pMP && !pMP->isBad() && pMP->mnBALocalForKF != pKF->mnId) {
        lLocalMapPoints.push_back(pMP);
        pMP->mnBALocalForKF = pKF->mnId;
      }
    }
  }
  std::list<KeyFrame *> lFixedCameras;
  for (std::list<MapPoint *>::iterator lit = lLocalMapPoints.begin(); lit != lLocalMapPoints.end(); ++lit) {
    const ObservationMap observations = (*lit)->GetObservations();
    for (ObservationMap::const_iterator mit = observations.begin(); mit != observations.end(); ++mit) {
      KeyFrame *pKFi = mit->first;
      if (pKFi->mnBALocalForKF != pKF->mnId && pKFi->mnBAFixedForKF != pKF->mnId) {
        pKFi->mnBAFixedForKF = pKF->mnId;
        if (!pKFi->isBad()) lFixedCameras.push_back(pKFi);
      }
    }
  }
  std::list<MapPointBird *> lLocalMapPointsBirds;
  if (withBird) {
    for (std::list<KeyFrame *>::iterator lit = lLocalKeyFrames.begin(); lit != lLocalKeyFrames.end(); ++lit) {
      const std::vector<MapPointBird *> vpMPBs = (*lit)->GetMapPointBirdMatches();
      for (size_t i = 0; i < vpMPBs.size(); i++) {
        MapPointBird *pMPB = vpMPBs[i];
        if (pMPB && !pMPB->isBad() && pMPB->mnBALocalForKF != pKF->mnId) {
          lLocalMapPointsBirds.push_back(pMPB);
          pMPB->mnBALocalForKF = pKF->mnId;
        }
      }
    }
    for (std::list<MapPointBird *>::iterator lit = lLocalMapPointsBirds.begin(); lit != lLocalMapPointsBirds.end(); ++lit) {
      const ObservationMap observations = (*lit)->GetObservations();
      for (ObservationMap::const_iterator mit = observations.begin(); mit != observations.end(); ++mit) {
        KeyFrame *pKFi = mit->first;
        if (pKFi->mnBALocalForKF != pKF->mnId && pKFi->mnBAFixedForKF != pKF->mnId) {
          pKFi->mnBAFixedForKF = pKF->mnId;
          if (!pKFi->isBad()) lFixedCameras.push_back(pKFi);
        }
      }
    }
  }
  for (std::list<KeyFrame *>::iterator lit = lLocalKeyFrames.begin(); lit != lLocalKeyFrames.end(); ++lit)
    G.addKeyFrame(*lit, (*lit)->isInit);
  G.nLocal = G.kfs.size();
  for (std::list<KeyFrame *>::iterator lit = lFixedCameras.begin(); lit != lFixedCameras.end(); ++lit) G.addKeyFrame(*lit, true);
  for (std::list<MapPoint *>::iterator lit = lLocalMapPoints.begin(); lit != lLocalMapPoints.end(); ++lit) G.addMapPoint(*lit);
  for (std::list<MapPointBird *>::iterator lit = lLocalMapPointsBirds.begin(); lit != lLocalMapPointsBirds.end(); ++lit)
    G.addMapPointBird(*lit);
}

// Erase the outlier observations and recover the optimised data (Optimizer.cc:1096-1164 / 2575-2668)
inline void writeBackLocal(BAGraph &G, bool withBird) {
  std::vector<std::pair<KeyFrame *, MapPoint *>> vToErase;
  for (size_t i = 0; i < G.vpEdgeKFMono.size(); i++) {
    MapPoint *pMP = G.vpMapPointEdgeMono[i];
    if (pMP->isBad()) continue;
    if (G.obsOutlier[i]) vToErase.push_back(std::make_pair(G.vpEdgeKFMono[i], pMP));
  }
  std::vector<std::pair<KeyFrame *, MapPointBird *>> vToEraseBird;
  if (withBird)
    for (size_t i = 0; i < G.vpEdgeKFBird.size(); i++)
      if (G.bobsOutlier[i]) vToEraseBird.push_back(std::make_pair(G.vpEdgeKFBird[i], G.vpMapPointEdgeBird[i]));
  for (size_t i = 0; i < vToErase.size(); i++) {
    vToErase[i].first->EraseMapPointMatch(vToErase[i].second);
    vToErase[i].second->EraseObservation(vToErase[i].first);
  }
  for (size_t i = 0; i < vToEraseBird.size(); i++) {
    vToEraseBird[i].first->EraseMapPointBirdMatch(vToEraseBird[i].second);
    vToEraseBird[i].second->EraseObservation(vToEraseBird[i].first);
  }
  for (size_t k = 0; k < G.nLocal; k++) G.kfs[k]->SetPose(&G.kfTcw[12 * k]);
  for (size_t j = 0; j < G.mps.size(); j++) {
    G.mps[j]->SetWorldPos(&G.mpXw[3 * j]);
    G.mps[j]->UpdateNormalAndDepth();
  }
  for (size_t j = 0; j < G.mpbs.size(); j++) G.mpbs[j]->SetWorldPos(&G.mpbXw[3 * j]);
}

}  // namespace fishbird
#endif
