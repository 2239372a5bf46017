// track.hip -- the device-resident Frame (fb_frame) and the per-frame tracking chain on gfx950.
//
// Replaces, as ONE device-resident sequence whose steps hand their results over in HBM (reference file:line):
//   Frame::Frame(...)                      src/Frame.cc:262-379   (extract, undistort, bird filter, cam XYZ, grids, member init)
//   Frame::SetPose / UpdatePoseMatrices    src/Frame.cc:421-433
//   Tracking::TrackWithMotionModel         src/Tracking.cc:1312-1385  (pose prediction, M9, M3, pose optimisation, outlier discard)
//   Tracking::TrackLocalMap                src/Tracking.cc:1387-1441  (M8 + FilterBirdOutlierInFront :1825-1914,
//                                                                      SearchLocalPoints :1947-1997, pose optimisation, inlier count)
//   end of Tracking::Track                 src/Tracking.cc:690-701, 721-725 (clean VO matches, drop outliers)
//
// The matchers, the extractor and the pose optimiser are the kernels of match.hip / orb.hip / pose.hip, reached through
// their *_dev entry points; this file adds the Frame state and the glue the reference does with pointer walks between
// them (mvpMapPoints bookkeeping, edge construction, MapPointBird creation).  Glue kernels are streaming maps over
// <= ~2000 slots per frame: one lane per slot, grid (slots / 256, batch), or one workgroup per sequence where a serial
// rule needs a block-wide scan / reduction.  They are latency-, not bandwidth-bound; what matters is the launch count
// per frame (17 + the extractor's), so commits ride in the kernel that follows them.
#include "fb_common.h"
#include "fb_frame_geom.h"

#include <cmath>
#include <new>

namespace {

constexpr int TT = 256;     // lane-per-slot kernels
constexpr int WG = 1024;    // workgroup-per-sequence kernels

struct FrameDev {  // device pointers of one frame, passed to kernels by value
  int cap;
  int32_t *n; fb_keypoint *kps, *kps_un; uint8_t *desc; int32_t *mp; uint8_t *outlier;
  int32_t *nb; fb_keypoint *bkps; uint8_t *bdesc; float *bcam; int32_t *mpb; uint8_t *boutlier;
  float *Tcw;
};

// mvpMapPoints = NULL, mvbOutlier = false (Frame.cc:327-328); mvpMapPointsBird = NULL, mvBirdOutlier = TRUE (:355-356)
__global__ __launch_bounds__(TT) void k_frame_reset(FrameDev F, int32_t *counts, int B) {
  const int b = blockIdx.y, i = blockIdx.x * TT + threadIdx.x;
  if (blockIdx.x == 0 && threadIdx.x < FB_CNT_COUNT) counts[threadIdx.x * B + b] = 0;
  if (i >= F.cap) return;
  const size_t o = (size_t)b * F.cap + i;
  F.mp[o] = -1; F.outlier[o] = 0; F.mpb[o] = -1; F.boutlier[o] = 1;
}

// mCurrentFrame.SetPose(detlaT * mLastFrame.mTcw) (Tracking.cc:1320): cv::Mat 4x4 * 4x4 CV_32F = gemm's small-matrix
// path, every element ((a0*b0 + a1*b1) + a2*b2) + a3*b3 in float; the last row of both is (0 0 0 1)
__global__ void k_predict_pose(const float *delta, const float *Tlast, float *Tcur, int B) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  const float *D = delta + (size_t)b * 12, *L = Tlast + (size_t)b * 12;
  float *C = Tcur + (size_t)b * 12;
  for (int r = 0; r < 3; r++)
    for (int c = 0; c < 4; c++) {
      float s = (D[r * 4 + 0] * L[0 * 4 + c] + D[r * 4 + 1] * L[1 * 4 + c]) + D[r * 4 + 2] * L[2 * 4 + c];
      s = s + D[r * 4 + 3] * (c == 3 ? 1.0f : 0.0f);
      C[r * 4 + c] = s;
    }
}

struct MapDev { int stride; const int32_t *n; const uint8_t *bad, *obs_pos; const float *xw, *normal, *max_dist, *min_dist; const uint8_t *desc; };
struct BirdMapDev { int stride; int32_t *n; float *xw; uint8_t *desc; };

struct M3Scratch { uint8_t *valid, *obs; float *xw; uint8_t *desc; int32_t *oct; float *ang; };

// SearchByProjection(cur, last): the per-slot view of the last frame the matcher reads (ORBmatcher.cc:1355-1392:
// pMP = LastFrame.mvpMapPoints[i], !LastFrame.mvbOutlier[i], GetWorldPos, GetDescriptor, mvKeys[i].octave,
// mvKeysUn[i].angle, Observations() > 0) + the fill(mvpMapPoints, NULL) both call sites do first (Tracking.cc:1330,1344)
__global__ __launch_bounds__(TT) void k_m3_prepare(FrameDev cur, FrameDev last, MapDev map, M3Scratch S) {
  const int b = blockIdx.y, i = blockIdx.x * TT + threadIdx.x;
  if (i >= cur.cap) return;
  const size_t o = (size_t)b * cur.cap + i;
  cur.mp[o] = -1;
  uint8_t v = 0;
  if (i < last.n[b]) {
    const int id = last.mp[o];
    if (id >= 0 && !last.outlier[o]) {
      v = 1;
      const size_t m = (size_t)b * map.stride + id;
      S.obs[o] = map.obs_pos[m];
      S.xw[o * 3] = map.xw[m * 3]; S.xw[o * 3 + 1] = map.xw[m * 3 + 1]; S.xw[o * 3 + 2] = map.xw[m * 3 + 2];
      const uint4 *src = reinterpret_cast<const uint4 *>(map.desc + m * 32);
      uint4 *dst = reinterpret_cast<uint4 *>(S.desc + o * 32);
      dst[0] = src[0]; dst[1] = src[1];
      S.oct[o] = last.kps[o].octave;
      S.ang[o] = last.kps_un[o].angle;
    }
  }
  S.valid[o] = v;
}

// SearchByBoW(pKF, F, ...): vpMapPointsKF[i] && !isBad() per key-frame slot (ORBmatcher.cc:196-203)
__global__ __launch_bounds__(TT) void k_bow_prepare(FrameDev kf, MapDev map, uint8_t *has_mp) {
  const int b = blockIdx.y, i = blockIdx.x * TT + threadIdx.x;
  if (i >= kf.cap) return;
  const size_t o = (size_t)b * kf.cap + i;
  uint8_t v = 0;
  if (i < kf.n[b]) {
    const int id = kf.mp[o];
    v = id >= 0 && !map.bad[(size_t)b * map.stride + id];
  }
  has_mp[o] = v;
}

// dense view of a vlocalMPB list for BirdMapPointMatch (ref_valid / ref_xw / ref_desc by list position)
__global__ __launch_bounds__(TT) void k_m9_prepare(BirdMapDev mpb, const int32_t *local, const int32_t *n_local, int lcap,
                                                   uint8_t *valid, float *xw, uint8_t *desc, int32_t *n_eff, int32_t *match, int cap) {
  const int b = blockIdx.y, j = blockIdx.x * TT + threadIdx.x;
  if (j < cap) match[(size_t)b * cap + j] = -1;
  const int nl = min(max(n_local[b], 0), lcap);
  if (j == 0) n_eff[b] = nl > 10 ? nl : 0;  // Tracking.cc:2004: if (vlocalMPB.size() > 10)
  if (j >= nl) return;
  const size_t o = (size_t)b * lcap + j;
  const int id = local[o];
  const bool ok = id >= 0 && id < mpb.n[b];
  valid[o] = ok ? 1 : 0;
  if (!ok) return;
  const size_t m = (size_t)b * mpb.stride + id;
  xw[o * 3] = mpb.xw[m * 3]; xw[o * 3 + 1] = mpb.xw[m * 3 + 1]; xw[o * 3 + 2] = mpb.xw[m * 3 + 2];
  const uint4 *src = reinterpret_cast<const uint4 *>(mpb.desc + m * 32);
  uint4 *dst = reinterpret_cast<uint4 *>(desc + o * 32);
  dst[0] = src[0]; dst[1] = src[1];
}
// whole table as the list: only the count rule and the match pre-fill
__global__ __launch_bounds__(TT) void k_m9_prepare_all(BirdMapDev mpb, int32_t *n_eff, int32_t *match, int cap) {
  const int b = blockIdx.y, j = blockIdx.x * TT + threadIdx.x;
  if (j < cap) match[(size_t)b * cap + j] = -1;
  if (j == 0) { const int nl = min(max(mpb.n[b], 0), mpb.stride); n_eff[b] = nl > 10 ? nl : 0; }
}

struct EdgeOut { float *fxw, *fobs, *finf; uint8_t *fvalid; float *bxw, *bxc, *binf; uint8_t *bvalid; };
struct SigmaTab { float inv_sigma2[FB_MAX_LEVELS]; };

// Edge construction loops of PoseOptimizationWithBird (Optimizer.cc:525-571 front, :575-602 bird) straight from the
// frame's members, after committing the matcher result that precedes the optimisation:
//   commit 1 (after M3): mvpMapPoints[i] = LastFrame.mvpMapPoints[match[i]]           (ORBmatcher.cc:1430)
//                        mvpMapPointsBird[i] = vlocalMPB[match_bird[i]]                (ORBmatcher.cc:1891)
//   commit 2 (after M2): mvpMapPoints[i] = mvpLocalMapPoints[match[i]] where matched  (ORBmatcher.cc:124)
struct Commit {  // the matcher result a frame has not folded into mvpMapPoints / mvpMapPointsBird yet
  int kind;                 // 0 none, 1 after M3 (+ M9), 2 after M2, 3 after SearchByBoW (replace every slot, gated)
  const int32_t *match;     // kind 1: slot of the last frame; kind 2: position in mvpLocalMapPoints
  const int32_t *src_mp;    // kind 1: LastFrame.mvpMapPoints; kind 2: the local list (NULL = the table itself)
  int src_stride;
  const int32_t *match_bird;  // kind 1: position in vlocalMPB, or NULL
  const int32_t *local_mpb;   // the vlocalMPB list (NULL = the table itself)
  int lcap_mpb;
  const int32_t *gate;        // kind 3: the matcher's count row; a sequence below gate_min "returned false" (Tracking.cc:1212):
  int gate_min;               //         no commit, no edges (the optimiser then leaves pose and flags alone)
  int no_front;               // BirdOptimization (Optimizer.cc:708-835) builds bird edges only
};
__device__ __forceinline__ bool commit_gate_open(const Commit &C, int b) { return !C.gate || C.gate[b] >= C.gate_min; }
__device__ __forceinline__ void commit_slot(const FrameDev &F, const Commit &C, int b, int i, size_t o, int &id, int &idb) {
  id = -1; idb = -1;
  if (i < F.n[b]) {
    id = F.mp[o];
    if (C.kind == 1 && C.match) {
      const int m = C.match[o];
      id = m >= 0 ? C.src_mp[(size_t)b * C.src_stride + m] : -1;
      F.mp[o] = id;
    } else if (C.kind == 2) {
      const int m = C.match[o];
      if (m >= 0) { id = C.src_mp ? C.src_mp[(size_t)b * C.src_stride + m] : m; F.mp[o] = id; }
    } else if (C.kind == 3 && commit_gate_open(C, b)) {  // mCurrentFrame.mvpMapPoints = vpMapPointMatches (Tracking.cc:1215)
      const int m = C.match[o];
      id = m >= 0 ? C.src_mp[(size_t)b * C.src_stride + m] : -1;
      F.mp[o] = id;
    }
  } else if (C.kind == 3 && commit_gate_open(C, b)) {
    F.mp[o] = -1;
  }
  if (i < F.nb[b]) {
    idb = F.mpb[o];
    if (C.kind == 1 && C.match_bird) {
      const int m = C.match_bird[o];
      if (m >= 0) { idb = C.local_mpb ? C.local_mpb[(size_t)b * C.lcap_mpb + m] : m; F.mpb[o] = idb; }
    }
  }
}

__global__ __launch_bounds__(TT) void k_commit(FrameDev F, Commit C) {
  const int b = blockIdx.y, i = blockIdx.x * TT + threadIdx.x;
  if (i >= F.cap) return;
  int id, idb;
  commit_slot(F, C, b, i, (size_t)b * F.cap + i, id, idb);
}

__global__ __launch_bounds__(TT) void k_edges(FrameDev F, MapDev map, BirdMapDev mpb, SigmaTab G, EdgeOut E, Commit C) {
  const int b = blockIdx.y, i = blockIdx.x * TT + threadIdx.x;
  if (i >= F.cap) return;
  const size_t o = (size_t)b * F.cap + i;
  int id, idb;
  commit_slot(F, C, b, i, o, id, idb);
  if (!commit_gate_open(C, b)) { id = -1; idb = -1; }
  if (C.no_front) id = -1;
  if (id >= 0) {
    const fb_keypoint kp = F.kps_un[o];
    const float *X = map.xw + ((size_t)b * map.stride + id) * 3;
    E.fxw[o * 3] = X[0]; E.fxw[o * 3 + 1] = X[1]; E.fxw[o * 3 + 2] = X[2];
    E.fobs[o * 2] = kp.x; E.fobs[o * 2 + 1] = kp.y;
    E.finf[o] = G.inv_sigma2[kp.octave];
    E.fvalid[o] = 1;
  } else {
    E.fvalid[o] = 0;
  }
  if (idb >= 0) {
    const float *X = mpb.xw + ((size_t)b * mpb.stride + idb) * 3;
    E.bxw[o * 3] = X[0]; E.bxw[o * 3 + 1] = X[1]; E.bxw[o * 3 + 2] = X[2];
    E.bxc[o * 3] = F.bcam[o * 3]; E.bxc[o * 3 + 1] = F.bcam[o * 3 + 1]; E.bxc[o * 3 + 2] = F.bcam[o * 3 + 2];
    E.binf[o] = G.inv_sigma2[F.bkps[o].octave];
    E.bvalid[o] = 1;
  } else {
    E.bvalid[o] = 0;
  }
}

__device__ __forceinline__ int block_sum(int v, int *s_w) {  // sum over a WG-thread block; every thread gets it
  for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o, 64);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) s_w[threadIdx.x >> 6] = v;
  __syncthreads();
  int t = 0;
  for (int w = 0; w < WG / 64; w++) t += s_w[w];
  return t;
}
__device__ __forceinline__ int wave_incl_scan(int v) {
  const int lane = threadIdx.x & 63;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) { const int t = __shfl_up(v, o, 64); if (lane >= o) v += t; }
  return v;
}
__device__ __forceinline__ int block_excl_scan(int v, int *s_w, int *total) {  // WG threads
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int inc = wave_incl_scan(v);
  __syncthreads();
  if (lane == 63) s_w[wv] = inc;
  __syncthreads();
  int base = 0, tot = 0;
#pragma unroll
  for (int w = 0; w < WG / 64; w++) { const int x = s_w[w]; if (w < wv) base += x; tot += x; }
  *total = tot;
  return base + inc - v;
}

// "Discard outliers" of TrackWithMotionModel (Tracking.cc:1358-1376): one workgroup per sequence
// and of TrackReferenceKeyFrame (:1222-1241); src_slot = the matcher count nmatches starts from; gate_min > 0: a sequence
// whose src_slot count is below it returned before the loop (:1212-1213)
__global__ __launch_bounds__(WG) void k_discard(FrameDev F, MapDev map, int32_t *counts, int B, int src_slot, int gate_min) {
  __shared__ int s_w[WG / 64];
  const int b = blockIdx.x, n = min(F.n[b], F.cap);
  if (gate_min > 0 && counts[src_slot * B + b] < gate_min) return;
  int dropped = 0, inmap = 0;
  for (int i = threadIdx.x; i < n; i += WG) {
    const size_t o = (size_t)b * F.cap + i;
    const int id = F.mp[o];
    if (id < 0) continue;
    if (F.outlier[o]) { F.mp[o] = -1; F.outlier[o] = 0; dropped++; }
    else if (map.obs_pos[(size_t)b * map.stride + id]) inmap++;
  }
  dropped = block_sum(dropped, s_w);
  inmap = block_sum(inmap, s_w);
  if (threadIdx.x == 0) {
    counts[FB_CNT_MATCHES * B + b] = counts[src_slot * B + b] - dropped;
    counts[FB_CNT_MATCHES_MAP * B + b] = inmap;
  }
}

// FilterBirdOutlierInFront (Tracking.cc:1825-1914) on the result of BirdviewMatch, one workgroup per sequence.
// vDMatches12 = (i1, vnMatches12[i1]) for ascending i1 with vnMatches12[i1] > 0 (sic, ORBmatcher.cc:1755).  Walking it in
// order: skip when the train slot holds a point (on entry or through an earlier PASSING match), test, on a pass take the
// slot.  => kept(i1) = passes(i1) && i1 is the smallest passing query of its train slot && the slot was free on entry.
// New MapPointBirds get ids in list order (mnId = nNextId++, here: the next table row).
// only_below > 0 (TrackReferenceKeyFrame, Tracking.cc:1196-1200): numPt = GetBirdMapPointsNum() (Frame.cc:1081-1092) is
// counted first and a sequence holding that many bird points or more skips the step (its BirdviewMatch count reads 0).
__global__ __launch_bounds__(WG) void k_bird_commit(FrameDev cur, FrameDev ref, BirdMapDev mpb, const int32_t *m12,
                                                    float window, int32_t *counts, int B, int only_below, const int32_t *gate_row,
                                                    int gate_min, const int32_t *m8_count) {
  extern __shared__ int s_first[];  // [cap] smallest passing query per train slot
  __shared__ int s_w[WG / 64];
  __shared__ float s_Twc1[12], s_T2[12];
  const int b = blockIdx.x, tid = threadIdx.x, cap = cur.cap;
  const size_t fo = (size_t)b * cap;
  const int nref = min(ref.nb[b], cap);
  if (gate_row && gate_row[b] < gate_min) return;  // TrackLocalMap does not run for this sequence (Tracking.cc:642: if (bOK))
  if (tid == 0) counts[FB_CNT_BIRDVIEW_MATCHES * B + b] = m8_count[b];  // (the matcher counted into scratch: a gated sequence keeps its counter)
  if (only_below > 0) {
    int have = 0;
    const int ncur = min(cur.nb[b], cap);
    for (int i = tid; i < ncur; i += WG) have += cur.mpb[fo + i] >= 0;
    have = block_sum(have, s_w);
    if (tid == 0) counts[FB_CNT_BIRD_POINTS * B + b] = have;
    if (have >= only_below) {
      if (tid == 0) counts[FB_CNT_BIRDVIEW_MATCHES * B + b] = 0;
      return;
    }
    __syncthreads();
  }
  if (tid == 0) {
    fb::inv_T(ref.Tcw + (size_t)b * 12, s_Twc1);
    for (int i = 0; i < 12; i++) s_T2[i] = cur.Tcw[(size_t)b * 12 + i];
  }
  for (int i = tid; i < cap; i += WG) s_first[i] = 0x7fffffff;
  __syncthreads();
  const int per = (nref + WG - 1) / WG;   // a thread owns a contiguous run of queries: ids in list order from one scan
  const int i0 = min(nref, tid * per), i1e = min(nref, i0 + per);
  for (int i1 = i0; i1 < i1e; i1++) {
    const int t = m12[fo + i1];
    if (!(t > 0)) continue;
    if (cur.mpb[fo + t] >= 0) continue;
    float ptw[3];
    if (fb::bird_filter_test(s_Twc1, s_T2, ref.bcam + (fo + i1) * 3, cur.bcam + (fo + t) * 3, window, ptw)) atomicMin(&s_first[t], i1);
  }
  __syncthreads();
  int nkept = 0, nnew = 0;
  for (int i1 = i0; i1 < i1e; i1++) {
    const int t = m12[fo + i1];
    if (!(t > 0) || s_first[t] != i1) continue;
    nkept++;
    if (ref.mpb[fo + i1] < 0) nnew++;
  }
  int totalNew;
  int idBase = block_excl_scan(nnew, s_w, &totalNew);
  const int totalKept = block_sum(nkept, s_w);
  const int n0 = min(max(mpb.n[b], 0), mpb.stride);
  for (int i1 = i0; i1 < i1e; i1++) {
    const int t = m12[fo + i1];
    if (!(t > 0) || s_first[t] != i1) continue;
    cur.boutlier[fo + t] = 0;
    int id = ref.mpb[fo + i1];
    if (id < 0) {
      id = n0 + idBase++;
      if (id < mpb.stride) {  // new MapPointBird(ptwC, MatchedFrame2, mpMap, trainIdx) (Tracking.cc:1897, MapPointBird.cc:18-28)
        float ptw[3];
        fb::bird_filter_test(s_Twc1, s_T2, ref.bcam + (fo + i1) * 3, cur.bcam + (fo + t) * 3, window, ptw);
        const size_t m = (size_t)b * mpb.stride + id;
        mpb.xw[m * 3] = ptw[0]; mpb.xw[m * 3 + 1] = ptw[1]; mpb.xw[m * 3 + 2] = ptw[2];
        const uint4 *src = reinterpret_cast<const uint4 *>(cur.bdesc + (fo + t) * 32);
        uint4 *dst = reinterpret_cast<uint4 *>(mpb.desc + m * 32);
        dst[0] = src[0]; dst[1] = src[1];
        ref.mpb[fo + i1] = id;
      } else {
        id = -1;  // table full: the point is not created (documented capacity)
      }
    }
    cur.mpb[fo + t] = id;
  }
  if (tid == 0) {
    mpb.n[b] = min(n0 + totalNew, mpb.stride);
    counts[FB_CNT_BIRD_INLIERS * B + b] = totalKept;
    counts[FB_CNT_BIRD_NEW * B + b] = totalNew;
  }
  __syncthreads();  // (this workgroup's own writes to cur.mpb)
  {
    int have = 0;
    const int ncur = min(cur.nb[b], cap);
    for (int i = tid; i < ncur; i += WG) have += cur.mpb[fo + i] >= 0;
    have = block_sum(have, s_w);
    if (tid == 0) counts[FB_CNT_BIRD_POINTS_FINAL * B + b] = have;
  }
}

struct LocalScratch { uint8_t *seen, *blocked, *inview, *obs; float *proj; int32_t *level; float *cosv; uint8_t *desc; int32_t *n_eff; };
struct FrustumK { fb_camera cam; float log_scale_factor; int n_levels; };

// SearchLocalPoints (Tracking.cc:1947-1984), one workgroup per sequence:
//   1. points the frame holds: bad ones are dropped from the frame, the others are marked seen (mnLastFrameSeen = mnId)
//   2. every other good point of mvpLocalMapPoints: isInFrustum(pMP, 0.5) -> the dense track members M2 reads
__global__ __launch_bounds__(WG) void k_local_points(FrameDev F, MapDev map, const int32_t *local, const int32_t *n_local, int lcap,
                                                     FrustumK K, LocalScratch S, int32_t *counts, int B, const int32_t *gate_row, int gate_min) {
  __shared__ int s_w[WG / 64];
  __shared__ float s_T[12], s_Ow[3];
  const int b = blockIdx.x, tid = threadIdx.x;
  if (gate_row && gate_row[b] < gate_min) {  // no TrackLocalMap for this sequence: M2 gets an empty list
    if (tid == 0) S.n_eff[b] = 0;
    return;
  }
  const size_t fo = (size_t)b * F.cap, mo = (size_t)b * map.stride, lo = (size_t)b * lcap;
  const int nmap = min(max(map.n[b], 0), map.stride);
  if (tid < 12) s_T[tid] = F.Tcw[(size_t)b * 12 + tid];
  if (tid == 64) fb::camera_centre(F.Tcw + (size_t)b * 12, s_Ow);
  for (int i = tid; i < nmap; i += WG) S.seen[mo + i] = 0;
  __syncthreads();
  const int n = min(F.n[b], F.cap);
  for (int i = tid; i < n; i += WG) {
    const int id = F.mp[fo + i];
    uint8_t blk = 0;
    if (id >= 0) {
      if (map.bad[mo + id]) F.mp[fo + i] = -1;
      else { S.seen[mo + id] = 1; blk = map.obs_pos[mo + id]; }
    }
    S.blocked[fo + i] = blk;  // M2 skips a slot whose point has observations (ORBmatcher.cc:88-90)
  }
  __syncthreads();
  const int nl = local ? min(max(n_local[b], 0), lcap) : min(nmap, lcap);
  int toMatch = 0;
  for (int j = tid; j < nl; j += WG) {
    const int id = local ? local[lo + j] : j;
    uint8_t v = 0;
    if (id >= 0 && id < nmap && !S.seen[mo + id] && !map.bad[mo + id]) {
      const size_t m = mo + id;
      fb::FrustumOut o;
      if (fb::in_frustum(s_T, s_Ow, K.cam, map.xw[m * 3], map.xw[m * 3 + 1], map.xw[m * 3 + 2], map.normal[m * 3], map.normal[m * 3 + 1],
                         map.normal[m * 3 + 2], map.max_dist[m], map.min_dist[m], 0.5f, K.log_scale_factor, K.n_levels, o)) {
        v = 1;
        toMatch++;
        S.proj[(lo + j) * 2] = o.u; S.proj[(lo + j) * 2 + 1] = o.v;
        S.level[lo + j] = o.level;
        S.cosv[lo + j] = o.view_cos;
        S.obs[lo + j] = map.obs_pos[m];
        const uint4 *src = reinterpret_cast<const uint4 *>(map.desc + m * 32);
        uint4 *dst = reinterpret_cast<uint4 *>(S.desc + (lo + j) * 32);
        dst[0] = src[0]; dst[1] = src[1];
      }
    }
    S.inview[lo + j] = v;
  }
  toMatch = block_sum(toMatch, s_w);
  if (tid == 0) { S.n_eff[b] = nl; counts[FB_CNT_TO_MATCH * B + b] = toMatch; }
}

// End of a tracked frame: mnMatchesInliers (Tracking.cc:1411-1424), clean VO matches (:690-701), drop outliers (:721-725)
// gate_row / gate_min: sequences whose first stage failed did not run TrackLocalMap (nothing to count, nothing to clean);
// min_inliers: TrackLocalMap returns mnMatchesInliers >= 30 (Tracking.cc:1438) and the clean-up sits inside if (bOK) (:681);
// keep_outliers: the final "mvpMapPoints[i] = NULL for outliers" (:721-725) is left to fb_frame_drop_outliers_dev, for a host
// that creates its key frame in between (:716-718: the outliers pass to the new key frame)
__global__ __launch_bounds__(WG) void k_finish(FrameDev F, MapDev map, int32_t *counts, int B, const int32_t *gate_row, int gate_min,
                                               int min_inliers, int keep_outliers) {
  __shared__ int s_w[WG / 64];
  const int b = blockIdx.x, n = min(F.n[b], F.cap);
  if (gate_row && gate_row[b] < gate_min) return;
  int inl = 0;
  for (int i = threadIdx.x; i < n; i += WG) {
    const size_t o = (size_t)b * F.cap + i;
    const int id = F.mp[o];
    if (id < 0) continue;
    if (!F.outlier[o] && map.obs_pos[(size_t)b * map.stride + id] != 0) inl++;
  }
  inl = block_sum(inl, s_w);
  if (threadIdx.x == 0) counts[FB_CNT_MATCHES_INLIERS * B + b] = inl;
  if (inl < min_inliers) return;   // bOK = false: mState = LOST, the frame keeps its members (Tracking.cc:668-675)
  for (int i = threadIdx.x; i < n; i += WG) {
    const size_t o = (size_t)b * F.cap + i;
    const int id = F.mp[o];
    if (id < 0) continue;
    const bool obs = map.obs_pos[(size_t)b * map.stride + id] != 0;
    if (!obs) { F.outlier[o] = 0; F.mp[o] = -1; }                     // Observations() < 1
    else if (F.outlier[o] && !keep_outliers) F.mp[o] = -1;            // mvbOutlier stays set, as in the reference
  }
}
// Tracking.cc:721-725 on its own (after a key-frame decision), for the sequences whose clean-up ran
__global__ __launch_bounds__(WG) void k_drop_outliers(FrameDev F, const int32_t *counts, int B, int min_inliers) {
  const int b = blockIdx.x, n = min(F.n[b], F.cap);
  if (counts[FB_CNT_MATCHES_INLIERS * B + b] < min_inliers) return;
  for (int i = threadIdx.x; i < n; i += WG) {
    const size_t o = (size_t)b * F.cap + i;
    if (F.mp[o] >= 0 && F.outlier[o]) F.mp[o] = -1;
  }
}

__global__ __launch_bounds__(TT) void k_set_map_points(FrameDev F, const int32_t *mp, const int32_t *mpb) {
  const int b = blockIdx.y, i = blockIdx.x * TT + threadIdx.x;
  if (i >= F.cap) return;
  const size_t o = (size_t)b * F.cap + i;
  if (mp) { F.mp[o] = i < F.n[b] ? mp[o] : -1; F.outlier[o] = 0; }
  if (mpb) F.mpb[o] = i < F.nb[b] ? mpb[o] : -1;
}

__global__ __launch_bounds__(TT) void k_clear_mp(FrameDev F) {
  const int b = blockIdx.y, i = blockIdx.x * TT + threadIdx.x;
  if (i < F.cap) F.mp[(size_t)b * F.cap + i] = -1;
}

}  // namespace

struct fb_frame {
  fb_frame_params P;
  int cap = 0, B = 0;
  fb_orb_tables tab;
  fb_grid_geom gF, gB;
  fb_camera cam;
  float logScale = 0.f;
  SigmaTab sig;
  // frame members
  fb::DevBuf n, kps, kps_un, desc, cs, ci, mp, outlier;
  fb::DevBuf nb, bkps, bdesc, bcam, bcs, bci, mpb, boutlier, nb_pre, bkps_pre, bdesc_pre;
  fb::DevBuf Tcw, counts;
  // scratch
  fb::DevBuf m3_valid, m3_obs, m3_xw, m3_desc, m3_oct, m3_ang, m_front;
  fb::DevBuf m9_valid, m9_xw, m9_desc, m9_n, m_bird, ones;
  fb::DevBuf e_fxw, e_fobs, e_finf, e_fvalid, e_bxw, e_bxc, e_binf, e_bvalid;
  fb::DevBuf m8_m12, m8_dist, m8_n, m8_nd;
  fb::DevBuf l_seen, l_blocked, l_inview, l_obs, l_proj, l_level, l_cos, l_desc, l_n, m_local, m2_ws;
  // mBowVec / mFeatVec (Frame.h:128-129), allocated by the first fb_frame_compute_bow_dev
  fb::DevBuf bow_nw, bow_ids, bow_vals, fv_nn, fv_ids, fv_start, fv_items;
  bool bowDone = false;   // !mBowVec.empty() (in the order the calls were enqueued)
  int minInliers = 30;    // the threshold the last end-of-Track clean-up used (fb_frame_drop_outliers_dev follows it)
  // images from host callers
  fb::DevBuf img_f, img_b, img_c, img_m;
  uint8_t *pin = nullptr;
  size_t pinBytes = 0;
  int32_t *pinCounts = nullptr;
  hipStream_t sBird = nullptr;
  hipEvent_t evFork = nullptr, evJoin = nullptr, evCopy = nullptr;
  ~fb_frame() {
    if (evFork) (void)hipEventDestroy(evFork);
    if (evJoin) (void)hipEventDestroy(evJoin);
    if (evCopy) (void)hipEventDestroy(evCopy);
    if (sBird) (void)hipStreamDestroy(sBird);
    if (pin) (void)hipHostFree(pin);
    if (pinCounts) (void)hipHostFree(pinCounts);
  }
  FrameDev dev() const {
    FrameDev F;
    F.cap = cap;
    F.n = n.as<int32_t>(); F.kps = kps.as<fb_keypoint>(); F.kps_un = kps_un.as<fb_keypoint>(); F.desc = desc.as<uint8_t>();
    F.mp = mp.as<int32_t>(); F.outlier = outlier.as<uint8_t>();
    F.nb = nb.as<int32_t>(); F.bkps = bkps.as<fb_keypoint>(); F.bdesc = bdesc.as<uint8_t>(); F.bcam = bcam.as<float>();
    F.mpb = mpb.as<int32_t>(); F.boutlier = boutlier.as<uint8_t>();
    F.Tcw = Tcw.as<float>();
    return F;
  }
  int32_t *cnt(int slot) const { return counts.as<int32_t>() + (size_t)slot * B; }
};

namespace {

MapDev map_dev(const fb_map_points *m) {
  MapDev M;
  M.stride = m->stride; M.n = m->n; M.bad = m->bad; M.obs_pos = m->obs_pos; M.xw = m->xw; M.normal = m->normal;
  M.max_dist = m->max_dist; M.min_dist = m->min_dist; M.desc = m->desc;
  return M;
}
BirdMapDev bird_dev(const fb_map_points_bird *m) {
  BirdMapDev M;
  M.stride = m->stride; M.n = m->n; M.xw = m->xw; M.desc = m->desc;
  return M;
}
bool map_ok(const fb_frame *f, const fb_map_points *m) {
  return m && m->stride > 0 && m->stride <= f->P.map_cap && m->n && m->bad && m->obs_pos && m->xw && m->normal && m->max_dist && m->min_dist &&
         m->desc && ((uintptr_t)m->desc % 16 == 0);
}
bool bird_ok(const fb_map_points_bird *m) { return m && m->stride > 0 && m->n && m->xw && m->desc && ((uintptr_t)m->desc % 16 == 0); }
dim3 slot_grid(const fb_frame *f) { return dim3((f->cap + TT - 1) / TT, f->B); }

int edges_and_pose(fb_frame *f, const fb_map_points *map, const fb_map_points_bird *mpb, int mode, float wB, float wF, int which,
                   const Commit &C, hipStream_t s) {
  EdgeOut E{f->e_fxw.as<float>(), f->e_fobs.as<float>(), f->e_finf.as<float>(), f->e_fvalid.as<uint8_t>(),
            f->e_bxw.as<float>(), f->e_bxc.as<float>(), f->e_binf.as<float>(), f->e_bvalid.as<uint8_t>()};
  {
    fb::ProfScope prof_(fb::P_GATHER, s);
    k_edges<<<slot_grid(f), TT, 0, s>>>(f->dev(), map_dev(map), bird_dev(mpb), f->sig, E, C);
    FB_HIP(hipGetLastError());
  }
  fb_pose_opt_args A;
  memset(&A, 0, sizeof(A));
  A.batch = f->B; A.mode = mode; A.front_stride = f->cap; A.bird_stride = f->cap;
  A.fx = f->P.K[0]; A.fy = f->P.K[1]; A.cx = f->P.K[2]; A.cy = f->P.K[3];
  A.wF = wF; A.wB = wB;
  A.n_front = f->n.as<int32_t>(); A.front_xw = E.fxw; A.front_obs = E.fobs; A.front_inv_sigma2 = E.finf; A.front_valid = E.fvalid;
  A.n_bird = f->nb.as<int32_t>(); A.bird_xw = E.bxw; A.bird_xc = E.bxc; A.bird_inv_sigma2 = E.binf; A.bird_valid = E.bvalid;
  A.bird_outlier = f->boutlier.as<uint8_t>();
  A.Tcw = f->Tcw.as<float>(); A.front_outlier = f->outlier.as<uint8_t>();
  A.ninliers = f->cnt(which ? FB_CNT_POSE2_INLIERS : FB_CNT_POSE1_INLIERS);
  return fb_pose_opt_batch_dev(&A, s);
}

}  // namespace

extern "C" {

int fb_frame_create(const fb_frame_params *p, fb_frame **out) {
  FB_ARG(p && out && p->batch >= 1 && p->batch <= 65535);
  FB_ARG(p->front_width > 0 && p->front_height > 0 && p->bird_width > 0 && p->bird_height > 0);
  FB_ARG(p->orb.nlevels >= 1 && p->orb.nlevels <= FB_MAX_LEVELS && p->orb.nfeatures > 0 && p->orb.scale_factor > 1.0f);
  FB_ARG(p->map_cap >= 1 && p->local_mp_cap >= 1 && p->local_mpb_cap >= 1);
  FB_ARG(p->bird_nfeatures >= 0 && p->bird_nfeatures <= p->orb.nfeatures);
  FB_TRY(fb::check_device());
  fb_frame *f = new (std::nothrow) fb_frame();
  if (!f) { fb::set_error("fb_frame_create: out of memory"); return FB_ERR_HIP; }
  f->P = *p;
  f->B = p->batch;
  f->cap = fb_orb_capacity(&p->orb);
  {  // the extractor's level tables (mvScaleFactors ... mvInvLevelSigma2, Frame.cc:299-306)
    fb_orb *o = nullptr;
    int rc = fb_orb_create(&p->orb, &o);
    if (rc != FB_OK) { delete f; return rc; }
    fb_orb_get_tables(o, &f->tab);
    fb_orb_destroy(o);
  }
  for (int i = 0; i < FB_MAX_LEVELS; i++) f->sig.inv_sigma2[i] = f->tab.inv_level_sigma2[i];
  f->logScale = (float)log((double)p->orb.scale_factor);  // mfLogScaleFactor = log(mfScaleFactor), Frame.cc:301
  float bounds[4];
  {  // ComputeImageBounds + grid cell sizes (Frame.cc:271-283)
    int rc = fb_image_bounds(p->front_width, p->front_height, p->K, p->D, bounds);
    if (rc != FB_OK) { delete f; return rc; }
  }
  f->gF.min_x = bounds[0]; f->gF.min_y = bounds[2];
  f->gF.inv_w = 64.0f / (bounds[1] - bounds[0]);  // FRAME_GRID_COLS / (mnMaxX - mnMinX), Frame.h:38-39
  f->gF.inv_h = 48.0f / (bounds[3] - bounds[2]);
  f->gF.cols = 64; f->gF.rows = 48;
  f->gB.min_x = 0.f; f->gB.min_y = 0.f;
  f->gB.inv_w = 32.0f / (float)p->bird_width;     // FRAME_GRID_BIRD, Frame.h:40
  f->gB.inv_h = 32.0f / (float)p->bird_height;
  f->gB.cols = 32; f->gB.rows = 32;
  f->cam.fx = p->K[0]; f->cam.fy = p->K[1]; f->cam.cx = p->K[2]; f->cam.cy = p->K[3];
  f->cam.min_x = bounds[0]; f->cam.max_x = bounds[1]; f->cam.min_y = bounds[2]; f->cam.max_y = bounds[3];
  const size_t B = f->B, cap = f->cap, lm = p->local_mp_cap, lb = p->local_mpb_cap;
  const size_t KP = sizeof(fb_keypoint);
  int rc = FB_OK;
#define AL(buf, bytes) if (rc == FB_OK) rc = f->buf.alloc(bytes)
  AL(n, B * 4); AL(kps, B * cap * KP); AL(kps_un, B * cap * KP); AL(desc, B * cap * 32); AL(cs, B * (64 * 48 + 1) * 4); AL(ci, B * cap * 4);
  AL(mp, B * cap * 4); AL(outlier, B * cap);
  AL(nb, B * 4); AL(bkps, B * cap * KP); AL(bdesc, B * cap * 32); AL(bcam, B * cap * 12); AL(bcs, B * (32 * 32 + 1) * 4); AL(bci, B * cap * 4);
  AL(mpb, B * cap * 4); AL(boutlier, B * cap); AL(nb_pre, B * 4); AL(bkps_pre, B * cap * KP); AL(bdesc_pre, B * cap * 32);
  AL(Tcw, B * 48); AL(counts, B * FB_CNT_COUNT * 4);
  AL(m3_valid, B * cap); AL(m3_obs, B * cap); AL(m3_xw, B * cap * 12); AL(m3_desc, B * cap * 32); AL(m3_oct, B * cap * 4); AL(m3_ang, B * cap * 4);
  AL(m_front, B * cap * 4);
  AL(m9_valid, B * lb); AL(m9_xw, B * lb * 12); AL(m9_desc, B * lb * 32); AL(m9_n, B * 4); AL(m_bird, B * cap * 4); AL(ones, B * lb);
  AL(e_fxw, B * cap * 12); AL(e_fobs, B * cap * 8); AL(e_finf, B * cap * 4); AL(e_fvalid, B * cap);
  AL(e_bxw, B * cap * 12); AL(e_bxc, B * cap * 12); AL(e_binf, B * cap * 4); AL(e_bvalid, B * cap);
  AL(m8_m12, B * cap * 4); AL(m8_dist, B * cap * 4); AL(m8_n, B * 4); AL(m8_nd, B * 4);
  AL(l_seen, B * (size_t)p->map_cap); AL(l_blocked, B * cap); AL(l_inview, B * lm); AL(l_obs, B * lm); AL(l_proj, B * lm * 8);
  AL(l_level, B * lm * 4); AL(l_cos, B * lm * 4); AL(l_desc, B * lm * 32); AL(l_n, B * 4); AL(m_local, B * cap * 4);
  AL(m2_ws, fb_match_projection_points_workspace((int)B, (int)lm));
#undef AL
  if (rc != FB_OK) { delete f; return rc; }
  hipError_t e = hipMemset(f->ones.p, 1, B * lb);
  if (e == hipSuccess) e = hipMemset(f->counts.p, 0, B * FB_CNT_COUNT * 4);
  if (e == hipSuccess) e = hipMemset(f->n.p, 0, B * 4);
  if (e == hipSuccess) e = hipMemset(f->nb.p, 0, B * 4);
  if (e == hipSuccess) e = hipMemset(f->Tcw.p, 0, B * 48);
  if (e == hipSuccess) e = hipMemset(f->mp.p, 0xff, B * cap * 4);
  if (e == hipSuccess) e = hipMemset(f->mpb.p, 0xff, B * cap * 4);
  if (e == hipSuccess) e = hipMemset(f->outlier.p, 0, B * cap);
  if (e == hipSuccess) e = hipMemset(f->boutlier.p, 1, B * cap);
  if (e == hipSuccess) e = hipStreamCreateWithFlags(&f->sBird, hipStreamNonBlocking);
  if (e == hipSuccess) e = hipEventCreateWithFlags(&f->evFork, hipEventDisableTiming);
  if (e == hipSuccess) e = hipEventCreateWithFlags(&f->evJoin, hipEventDisableTiming);
  if (e == hipSuccess) e = hipEventCreateWithFlags(&f->evCopy, hipEventDisableTiming);
  if (e == hipSuccess) e = hipHostMalloc(reinterpret_cast<void **>(&f->pinCounts), (B * FB_CNT_COUNT + B * 12) * 4, hipHostMallocDefault);
  if (e != hipSuccess) {
    fb::set_error("fb_frame_create: %s", hipGetErrorString(e));
    (void)hipGetLastError();
    delete f;
    return FB_ERR_HIP;
  }
  *out = f;
  return FB_OK;
}

void fb_frame_destroy(fb_frame *f) { delete f; }

int fb_frame_extract_dev(fb_frame *f, fb_orb *of, fb_orb *ob, const uint8_t *d_front, int front_stride, size_t front_image_stride,
                         const uint8_t *d_bird, int bird_stride, size_t bird_image_stride, const uint8_t *d_contour,
                         const uint8_t *d_mask, void *stream) {
  FB_TRY(fb::check_device());
  FB_ARG(f && of && ob && of != ob && d_front && d_bird);
  FB_ARG(front_stride >= f->P.front_width && bird_stride >= f->P.bird_width);
  FB_ARG(!d_mask || d_contour);  // the detect mask rides in the contour kernel
  if (d_contour) FB_ARG(bird_image_stride == (size_t)bird_stride * f->P.bird_height);  // contour / mask share the bird image's geometry
  // (while every kernel is bracketed for a per-kernel table -- fb_prof_enable without fb_prof_only -- the bird chain stays on
  // the caller's stream, so that the table shows each kernel on its own)
  const bool fork = !(fb::g_prof_on && fb::g_prof_only < 0);
  hipStream_t s = fb::as_stream(stream), sb = fork ? f->sBird : s;
  const int B = f->B, cap = f->cap;
  f->bowDone = false;                         // a new Frame: mBowVec empty
  FB_TRY(fb_orb_set_output_stride(of, cap));  // both extractors write into the frame's arrays (one stride)
  FB_TRY(fb_orb_set_output_stride(ob, cap));
  // bird chain on the handle's stream, beside the front chain
  if (fork) {
    FB_HIP(hipEventRecord(f->evFork, s));
    FB_HIP(hipStreamWaitEvent(sb, f->evFork, 0));
  }
  {
    fb_keypoint *k0 = d_contour ? f->bkps_pre.as<fb_keypoint>() : f->bkps.as<fb_keypoint>();
    uint8_t *d0 = d_contour ? f->bdesc_pre.as<uint8_t>() : f->bdesc.as<uint8_t>();
    int32_t *n0 = d_contour ? f->nb_pre.as<int32_t>() : f->nb.as<int32_t>();
    FB_TRY(fb_orb_extract_batch_dev(ob, d_bird, B, f->P.bird_width, f->P.bird_height, bird_stride, bird_image_stride, k0, d0, n0, sb));
    if (d_contour) {
      fb_bird_guidance_args G;
      memset(&G, 0, sizeof(G));
      G.batch = B; G.kp_stride = cap; G.cols = f->P.bird_width; G.rows = f->P.bird_height; G.pitch = bird_stride;
      G.contour = d_contour; G.mask = d_mask; G.n_in = n0; G.kps_in = k0; G.desc_in = d0;
      G.n_out = f->nb.as<int32_t>(); G.kps_out = f->bkps.as<fb_keypoint>(); G.desc_out = f->bdesc.as<uint8_t>();
      G.keep = f->l_blocked.as<uint8_t>();  // [batch][cap] scratch, free at this point of the frame: the verdicts are computed by many workgroups
      FB_TRY(fb_bird_guidance_dev(&G, sb));
    }
    FB_TRY(fb_bird_keys_to_cam_dev(f->bkps.as<fb_keypoint>(), f->nb.as<int32_t>(), B, cap, f->P.bird_width, f->P.bird_height,
                                   f->P.pixel2meter, f->P.rear_axle_to_center, f->P.Tcb, f->bcam.as<float>(), sb));
    FB_TRY(fb_grid_build_batch_dev(f->bkps.as<fb_keypoint>(), f->nb.as<int32_t>(), B, cap, &f->gB, f->bcs.as<int32_t>(), f->bci.as<int32_t>(), sb));
    if (fork) FB_HIP(hipEventRecord(f->evJoin, sb));
  }
  FB_TRY(fb_orb_extract_batch_dev(of, d_front, B, f->P.front_width, f->P.front_height, front_stride, front_image_stride,
                                  f->kps.as<fb_keypoint>(), f->desc.as<uint8_t>(), f->n.as<int32_t>(), s));
  FB_TRY(fb_undistort_keypoints_dev(f->kps.as<fb_keypoint>(), f->n.as<int32_t>(), B, cap, f->P.K, f->P.D, f->kps_un.as<fb_keypoint>(), s));
  FB_TRY(fb_grid_build_batch_dev(f->kps_un.as<fb_keypoint>(), f->n.as<int32_t>(), B, cap, &f->gF, f->cs.as<int32_t>(), f->ci.as<int32_t>(), s));
  { fb::ProfScope prof_(fb::P_TRACK_GLUE, s);
    k_frame_reset<<<slot_grid(f), TT, 0, s>>>(f->dev(), f->counts.as<int32_t>(), B); }
  FB_HIP(hipGetLastError());
  if (fork) FB_HIP(hipStreamWaitEvent(s, f->evJoin, 0));
  return FB_OK;
}

int fb_frame_extract(fb_frame *f, fb_orb *of, fb_orb *ob, const uint8_t *front, int front_stride, const uint8_t *bird,
                     int bird_stride, const uint8_t *contour, const uint8_t *mask, void *stream) {
  FB_TRY(fb::check_device());
  FB_ARG(f && front && bird && front_stride >= f->P.front_width && bird_stride >= f->P.bird_width);
  const size_t B = f->B, fb_ = (size_t)front_stride * f->P.front_height, bb = (size_t)bird_stride * f->P.bird_height;
  const size_t need = B * (fb_ + 3 * bb);
  hipStream_t s = fb::as_stream(stream);
  if (f->pinBytes < need) {
    if (f->pin) { FB_HIP(hipStreamSynchronize(s)); FB_HIP(hipHostFree(f->pin)); f->pin = nullptr; f->pinBytes = 0; }
    FB_HIP(hipHostMalloc(reinterpret_cast<void **>(&f->pin), need, hipHostMallocDefault));
    f->pinBytes = need;
    FB_TRY(f->img_f.alloc(B * fb_)); FB_TRY(f->img_b.alloc(B * bb)); FB_TRY(f->img_c.alloc(B * bb)); FB_TRY(f->img_m.alloc(B * bb));
  } else {
    // the previous call's copies out of the staging block must be done before it is overwritten
    FB_HIP(hipEventSynchronize(f->evCopy));
  }
  uint8_t *pf = f->pin, *pb = pf + B * fb_, *pc = pb + B * bb, *pm = pc + B * bb;
  memcpy(pf, front, B * fb_);
  memcpy(pb, bird, B * bb);
  if (contour) memcpy(pc, contour, B * bb);
  if (mask) memcpy(pm, mask, B * bb);
  FB_HIP(hipMemcpyAsync(f->img_f.p, pf, B * fb_, hipMemcpyHostToDevice, s));
  FB_HIP(hipMemcpyAsync(f->img_b.p, pb, B * bb, hipMemcpyHostToDevice, s));
  if (contour) FB_HIP(hipMemcpyAsync(f->img_c.p, pc, B * bb, hipMemcpyHostToDevice, s));
  if (mask) FB_HIP(hipMemcpyAsync(f->img_m.p, pm, B * bb, hipMemcpyHostToDevice, s));
  FB_HIP(hipEventRecord(f->evCopy, s));
  return fb_frame_extract_dev(f, of, ob, f->img_f.as<uint8_t>(), front_stride, fb_, f->img_b.as<uint8_t>(), bird_stride, bb,
                              contour ? f->img_c.as<uint8_t>() : nullptr, mask ? f->img_m.as<uint8_t>() : nullptr, stream);
}

int fb_frame_set_pose_dev(fb_frame *f, const float *d_Tcw, void *stream) {
  FB_TRY(fb::check_device());
  FB_ARG(f && d_Tcw);
  FB_HIP(hipMemcpyAsync(f->Tcw.p, d_Tcw, (size_t)f->B * 48, hipMemcpyDeviceToDevice, fb::as_stream(stream)));
  return FB_OK;
}

int fb_frame_predict_pose_dev(fb_frame *cur, const fb_frame *last, const float *d_delta, void *stream) {
  FB_TRY(fb::check_device());
  FB_ARG(cur && last && d_delta && cur != last && cur->B == last->B);
  fb::ProfScope prof_(fb::P_TRACK_GLUE, fb::as_stream(stream));
  k_predict_pose<<<(cur->B + 63) / 64, 64, 0, fb::as_stream(stream)>>>(d_delta, last->Tcw.as<float>(), cur->Tcw.as<float>(), cur->B);
  FB_HIP(hipGetLastError());
  return FB_OK;
}

int fb_frame_clear_map_points_dev(fb_frame *f, void *stream) {
  FB_TRY(fb::check_device());
  FB_ARG(f);
  k_clear_mp<<<slot_grid(f), TT, 0, fb::as_stream(stream)>>>(f->dev());
  FB_HIP(hipGetLastError());
  return FB_OK;
}

int fb_frame_set_map_points_dev(fb_frame *f, const int32_t *d_mp, const int32_t *d_mpb, void *stream) {
  FB_TRY(fb::check_device());
  FB_ARG(f && (d_mp || d_mpb));
  k_set_map_points<<<slot_grid(f), TT, 0, fb::as_stream(stream)>>>(f->dev(), d_mp, d_mpb);
  FB_HIP(hipGetLastError());
  return FB_OK;
}

}  // extern "C"

namespace {

// A matcher leaves its result in the frame's match buffer; `defer` = the pose optimisation that follows folds it into
// mvpMapPoints / mvpMapPointsBird inside its edge kernel (fb_frame_track_dev), otherwise a commit launch does it here.
int launch_commit(fb_frame *f, const Commit &C, hipStream_t s) {
  fb::ProfScope prof_(fb::P_TRACK_GLUE, s);
  k_commit<<<slot_grid(f), TT, 0, s>>>(f->dev(), C);
  FB_HIP(hipGetLastError());
  return FB_OK;
}

Commit commit_m9(const fb_frame *cur, const int32_t *d_local) {
  Commit C;
  memset(&C, 0, sizeof(C));
  C.kind = 1; C.match = nullptr; C.match_bird = cur->m_bird.as<int32_t>(); C.local_mpb = d_local; C.lcap_mpb = cur->P.local_mpb_cap;
  return C;
}

int m9_impl(fb_frame *cur, const fb_map_points_bird *mpb, const int32_t *d_local, const int32_t *d_n_local, int window_size,
            float filter_size, const fb_matcher_params *matcher, hipStream_t s) {
  const int B = cur->B, cap = cur->cap, lcap = cur->P.local_mpb_cap;
  fb_bird_mp_args A;
  memset(&A, 0, sizeof(A));
  { fb::ProfScope prof_(fb::P_TRACK_GLUE, s);
  if (d_local) {
    k_m9_prepare<<<dim3((std::max(lcap, cap) + TT - 1) / TT, B), TT, 0, s>>>(bird_dev(mpb), d_local, d_n_local, lcap, cur->m9_valid.as<uint8_t>(),
                                                                          cur->m9_xw.as<float>(), cur->m9_desc.as<uint8_t>(),
                                                                          cur->m9_n.as<int32_t>(), cur->m_bird.as<int32_t>(), cap);
    A.ref_stride = lcap; A.ref_valid = cur->m9_valid.as<uint8_t>(); A.ref_xw = cur->m9_xw.as<float>(); A.ref_desc = cur->m9_desc.as<uint8_t>();
  } else {
    k_m9_prepare_all<<<slot_grid(cur), TT, 0, s>>>(bird_dev(mpb), cur->m9_n.as<int32_t>(), cur->m_bird.as<int32_t>(), cap);
    A.ref_stride = mpb->stride; A.ref_valid = cur->ones.as<uint8_t>(); A.ref_xw = mpb->xw; A.ref_desc = mpb->desc;
  }
  }
  FB_HIP(hipGetLastError());
  A.batch = B; A.cur_stride = cap;
  A.n_cur = cur->nb.as<int32_t>(); A.cur_kps = cur->bkps.as<fb_keypoint>(); A.cur_desc = cur->bdesc.as<uint8_t>();
  A.cur_cam_xyz = cur->bcam.as<float>(); A.cur_cell_start = cur->bcs.as<int32_t>(); A.cur_cell_items = cur->bci.as<int32_t>();
  A.cur_Tcw = cur->Tcw.as<float>(); A.n_ref = cur->m9_n.as<int32_t>();
  memcpy(A.Tbc, cur->P.Tbc, sizeof(A.Tbc));
  A.bird_cols = cur->P.bird_width; A.bird_rows = cur->P.bird_height;
  A.meter2pixel = cur->P.meter2pixel; A.rear_axle_to_center = cur->P.rear_axle_to_center;
  A.grid = cur->gB; A.window_size = window_size; A.filter_size = filter_size; A.matcher = *matcher;
  A.match_cur_to_ref = cur->m_bird.as<int32_t>(); A.ninliers = cur->cnt(FB_CNT_BIRD_KF_MATCHES);
  return fb_match_bird_mappoints_dev(&A, s);
}

int m3_impl(fb_frame *cur, const fb_frame *last, const fb_map_points *map, float th, const fb_matcher_params *matcher, hipStream_t s,
            int retry_below = 0) {
  M3Scratch S{cur->m3_valid.as<uint8_t>(), cur->m3_obs.as<uint8_t>(), cur->m3_xw.as<float>(), cur->m3_desc.as<uint8_t>(),
              cur->m3_oct.as<int32_t>(), cur->m3_ang.as<float>()};
  { fb::ProfScope prof_(fb::P_TRACK_GLUE, s);
    k_m3_prepare<<<slot_grid(cur), TT, 0, s>>>(cur->dev(), last->dev(), map_dev(map), S); }
  FB_HIP(hipGetLastError());
  fb_proj_frame_args A;
  memset(&A, 0, sizeof(A));
  A.batch = cur->B; A.cur_stride = cur->cap; A.last_stride = cur->cap;
  A.n_cur = cur->n.as<int32_t>(); A.cur_kps = cur->kps_un.as<fb_keypoint>(); A.cur_desc = cur->desc.as<uint8_t>();
  A.cur_cell_start = cur->cs.as<int32_t>(); A.cur_cell_items = cur->ci.as<int32_t>(); A.cur_blocked = nullptr;
  A.cur_Tcw = cur->Tcw.as<float>();
  A.n_last = last->n.as<int32_t>(); A.last_valid = S.valid; A.last_obs_pos = S.obs; A.last_xw = S.xw; A.last_desc = S.desc;
  A.last_octave = S.oct; A.last_angle = S.ang;
  A.cam = cur->cam; A.grid = cur->gF;
  for (int i = 0; i < FB_MAX_LEVELS; i++) A.scale_factors[i] = cur->tab.scale_factor[i];
  A.th = th; A.matcher = *matcher;
  A.match_cur_to_last = cur->m_front.as<int32_t>(); A.nmatches = cur->cnt(FB_CNT_PROJ_MATCHES);
  A.retry_below = retry_below; A.retry_th = 2.0f * th;   // Tracking.cc:1342-1349
  A.retried = retry_below > 0 ? cur->cnt(FB_CNT_PROJ_RETRIED) : nullptr;
  return fb_match_projection_frame_dev(&A, s);
}

int local_impl(fb_frame *f, const fb_map_points *map, const int32_t *d_local, const int32_t *d_n_local, float th,
               const fb_matcher_params *matcher, hipStream_t s, const int32_t *gate_row = nullptr, int gate_min = 0) {
  const int B = f->B, cap = f->cap, lcap = f->P.local_mp_cap;
  LocalScratch S{f->l_seen.as<uint8_t>(), f->l_blocked.as<uint8_t>(), f->l_inview.as<uint8_t>(), f->l_obs.as<uint8_t>(),
                 f->l_proj.as<float>(), f->l_level.as<int32_t>(), f->l_cos.as<float>(), f->l_desc.as<uint8_t>(), f->l_n.as<int32_t>()};
  FrustumK K{f->cam, f->logScale, f->P.orb.nlevels};
  {
    fb::ProfScope prof_(fb::P_FRUSTUM, s);
    k_local_points<<<B, WG, 0, s>>>(f->dev(), map_dev(map), d_local, d_n_local, lcap, K, S, f->counts.as<int32_t>(), B, gate_row, gate_min);
    FB_HIP(hipGetLastError());
  }
  fb_proj_points_args A;
  memset(&A, 0, sizeof(A));
  A.batch = B; A.cur_stride = cap; A.mp_stride = lcap;
  A.n_cur = f->n.as<int32_t>(); A.cur_kps = f->kps_un.as<fb_keypoint>(); A.cur_desc = f->desc.as<uint8_t>();
  A.cur_cell_start = f->cs.as<int32_t>(); A.cur_cell_items = f->ci.as<int32_t>(); A.cur_blocked = S.blocked;
  A.n_mp = S.n_eff; A.mp_track = S.inview; A.mp_obs_pos = S.obs; A.mp_proj = S.proj; A.mp_level = S.level; A.mp_view_cos = S.cosv;
  A.mp_desc = S.desc; A.grid = f->gF;
  for (int i = 0; i < FB_MAX_LEVELS; i++) A.scale_factors[i] = f->tab.scale_factor[i];
  A.th = th; A.matcher = *matcher;
  A.match_cur_to_mp = f->m_local.as<int32_t>(); A.nmatches = f->cnt(FB_CNT_LOCAL_MATCHES);
  A.workspace = f->m2_ws.p; A.workspace_bytes = f->m2_ws.bytes;
  return fb_match_projection_points_dev(&A, s);
}

int bird_points_impl(fb_frame *cur, fb_frame *ref, fb_map_points_bird *mpb, int window_size, float filter_size,
                     const fb_matcher_params *matcher, int only_below, hipStream_t s, const int32_t *gate_row = nullptr, int gate_min = 0) {
  fb_birdview_args A;
  memset(&A, 0, sizeof(A));
  A.batch = cur->B; A.cur_stride = cur->cap; A.ref_stride = cur->cap;
  A.n_cur = cur->nb.as<int32_t>(); A.cur_kps = cur->bkps.as<fb_keypoint>(); A.cur_desc = cur->bdesc.as<uint8_t>();
  A.cur_cell_start = cur->bcs.as<int32_t>(); A.cur_cell_items = cur->bci.as<int32_t>();
  A.n_ref = ref->nb.as<int32_t>(); A.ref_kps = ref->bkps.as<fb_keypoint>(); A.ref_desc = ref->bdesc.as<uint8_t>();
  A.grid = cur->gB; A.window_size = window_size; A.matcher = *matcher;
  A.match_ref_to_cur = cur->m8_m12.as<int32_t>(); A.match_dist = cur->m8_dist.as<int32_t>();
  A.nmatches = cur->m8_n.as<int32_t>(); A.n_dmatches = cur->m8_nd.as<int32_t>();
  FB_TRY(fb_match_birdview_dev(&A, s));
  const size_t lds = (size_t)cur->cap * 4;
  if (lds > 150 * 1024) { fb::set_error("fb_frame_match_bird_points: %d key points per frame beyond the LDS slot table", cur->cap); return FB_ERR_CAPACITY; }
  FB_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(k_bird_commit), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  fb::ProfScope prof_(fb::P_TRACK_GLUE, s);
  k_bird_commit<<<cur->B, WG, lds, s>>>(cur->dev(), ref->dev(), bird_dev(mpb), cur->m8_m12.as<int32_t>(), filter_size,
                                       cur->counts.as<int32_t>(), cur->B, only_below, gate_row, gate_min, cur->m8_n.as<int32_t>());
  FB_HIP(hipGetLastError());
  return FB_OK;
}


int discard_impl(fb_frame *f, const fb_map_points *map, int src_slot, int gate_min, hipStream_t s) {
  fb::ProfScope prof_(fb::P_TRACK_GLUE, s);
  k_discard<<<f->B, WG, 0, s>>>(f->dev(), map_dev(map), f->counts.as<int32_t>(), f->B, src_slot, gate_min);
  FB_HIP(hipGetLastError());
  return FB_OK;
}

fb_feature_vector frame_fv(const fb_frame *f) {
  fb_feature_vector v;
  v.node_stride = f->cap; v.item_stride = f->cap;
  v.n_nodes = f->fv_nn.as<int32_t>(); v.node_ids = f->fv_ids.as<uint32_t>(); v.node_start = f->fv_start.as<int32_t>(); v.items = f->fv_items.as<int32_t>();
  return v;
}

// SearchByBoW(pKF, F, vpMapPointMatches) (ORBmatcher.cc:160-289): result in cur->m_front, count in FB_CNT_BOW_MATCHES
int bow_impl(fb_frame *cur, const fb_frame *kf, const fb_map_points *map, const fb_matcher_params *matcher, hipStream_t s) {
  { fb::ProfScope prof_(fb::P_TRACK_GLUE, s);
    k_bow_prepare<<<slot_grid(cur), TT, 0, s>>>(kf->dev(), map_dev(map), cur->m3_valid.as<uint8_t>()); }
  FB_HIP(hipGetLastError());
  fb_bow_args A;
  memset(&A, 0, sizeof(A));
  A.batch = cur->B; A.kf_stride = cur->cap; A.f_stride = cur->cap;
  A.n_kf = kf->n.as<int32_t>(); A.kf_kps = kf->kps_un.as<fb_keypoint>(); A.kf_desc = kf->desc.as<uint8_t>();
  A.kf_has_mp = cur->m3_valid.as<uint8_t>(); A.kf_fv = frame_fv(kf);
  A.n_f = cur->n.as<int32_t>(); A.f_kps = cur->kps.as<fb_keypoint>(); A.f_desc = cur->desc.as<uint8_t>(); A.f_fv = frame_fv(cur);  // F.mvKeys, :258
  A.matcher = *matcher;
  A.match_f_to_kf = cur->m_front.as<int32_t>(); A.nmatches = cur->cnt(FB_CNT_BOW_MATCHES);
  return fb_match_bow_dev(&A, s);
}

Commit commit_bow(const fb_frame *cur, const fb_frame *kf, int min_matches) {
  Commit C;
  memset(&C, 0, sizeof(C));
  C.kind = 3; C.match = cur->m_front.as<int32_t>(); C.src_mp = kf->mp.as<int32_t>(); C.src_stride = cur->cap;
  C.gate = min_matches > 0 ? cur->cnt(FB_CNT_BOW_MATCHES) : nullptr; C.gate_min = min_matches;
  return C;
}

bool track_args_ok(const fb_frame *cur, const fb_track_args *T) {
  return T && map_ok(cur, &T->map) && bird_ok(&T->mpb) && (!T->d_local_mp || T->d_n_local_mp) && (!T->d_local_mpb || T->d_n_local_mpb) &&
         (T->d_local_mp || T->map.stride <= cur->P.local_mp_cap) && (T->d_local_mpb || T->mpb.stride <= cur->P.local_mpb_cap);
}

const fb_matcher_params M09 = {0.9f, 1};  // ORBmatcher matcher(0.9,true), Tracking.cc:1325,2001,2726
const fb_matcher_params M08 = {0.8f, 1};  // ORBmatcher matcher(0.8), Tracking.cc:1988
const fb_matcher_params M07 = {0.7f, 1};  // ORBmatcher matcher(0.7,true), Tracking.cc:1207

// The bird-side step of a stage runs on the handle's own stream beside the front-side step (they touch disjoint members of
// the frame and both only read its pose): fork after the producer of what both need, join before the edge kernel.  Not while
// every kernel is bracketed for the per-kernel table (the table shows each kernel on its own).
struct SideStream {
  fb_frame *f; hipStream_t s; bool on;
  SideStream(fb_frame *f_, hipStream_t s_) : f(f_), s(s_), on(!(fb::g_prof_on && fb::g_prof_only < 0) && !getenv("FB_TRACK_NO_FORK")) {}
  hipStream_t side() const { return on ? f->sBird : s; }
  int fork() { if (on) { FB_HIP(hipEventRecord(f->evFork, s)); FB_HIP(hipStreamWaitEvent(f->sBird, f->evFork, 0)); } return FB_OK; }
  int join() { if (on) { FB_HIP(hipEventRecord(f->evJoin, f->sBird)); FB_HIP(hipStreamWaitEvent(s, f->evJoin, 0)); } return FB_OK; }
};

// TrackWithMotionModel (Tracking.cc:1312-1385)
int motion_model_impl(fb_frame *cur, fb_frame *last, const fb_track_args *T, hipStream_t s) {
  fb_map_points_bird mpb = T->mpb;
  FB_TRY(fb_frame_predict_pose_dev(cur, last, T->d_delta, s));                                         // :1314-1320
  SideStream side(cur, s);
  FB_TRY(side.fork());
  FB_TRY(m9_impl(cur, &mpb, T->d_local_mpb, T->d_n_local_mpb, 10, 0.05f, &M09, side.side()));          // :1322-1323 -> :1999-2012
  FB_TRY(m3_impl(cur, last, &T->map, 15.0f, &M09, s, 20));                                             // :1339-1349 (incl. the 2 * th retry)
  FB_TRY(side.join());
  Commit C1 = commit_m9(cur, T->d_local_mpb);
  C1.match = cur->m_front.as<int32_t>(); C1.src_mp = last->mp.as<int32_t>(); C1.src_stride = cur->cap;
  // if (nmatches < 20) return false (:1351): the matches are committed, but such a sequence gets no edges (the optimiser leaves
  // its pose and flags alone) and no discard loop
  C1.gate = cur->cnt(FB_CNT_PROJ_MATCHES); C1.gate_min = 20;
  FB_TRY(edges_and_pose(cur, &T->map, &mpb, FB_POSE_FRONT_BIRD, T->wB, T->wF, 0, C1, s));              // :1353
  return discard_impl(cur, &T->map, FB_CNT_PROJ_MATCHES, 20, s);                                       // :1358-1376
}

int finish_impl(fb_frame *f, const fb_map_points *map, const int32_t *gate_row, int gate_min, int keep_outliers, hipStream_t s,
                int min_inliers = 0) {
  f->minInliers = min_inliers > 0 ? min_inliers : 30;
  fb::ProfScope prof_(fb::P_TRACK_GLUE, s);
  k_finish<<<f->B, WG, 0, s>>>(f->dev(), map_dev(map), f->counts.as<int32_t>(), f->B, gate_row, gate_min, f->minInliers, keep_outliers);
  FB_HIP(hipGetLastError());
  return FB_OK;
}

// TrackLocalMap (Tracking.cc:1387-1441) + the end of Track (:1411-1424, 690-701, 721-725).  gated: a sequence whose first
// stage failed (nmatchesMap < 10: TrackWithMotionModel :1384 / TrackReferenceKeyFrame :1243 returned false; the counter is
// still 0 when the stage returned early) is left alone, as if (bOK) of Tracking.cc:642 leaves it
int local_map_impl(fb_frame *cur, fb_frame *ref, const fb_track_args *T, hipStream_t s, bool gated) {
  fb_map_points_bird mpb = T->mpb;
  const int32_t *gr = gated ? cur->cnt(FB_CNT_MATCHES_MAP) : nullptr;
  const int gm = 10;
  SideStream side(cur, s);
  FB_TRY(side.fork());
  FB_TRY(bird_points_impl(cur, ref, &mpb, 10, 0.05f, &M09, 0, side.side(), gr, gm));                   // :1392 -> :2724-2733
  FB_TRY(local_impl(cur, &T->map, T->d_local_mp, T->d_n_local_mp, 1.0f, &M08, s, gr, gm));             // :1396 -> :1947-1997
  FB_TRY(side.join());
  Commit C2;
  memset(&C2, 0, sizeof(C2));
  C2.kind = 2; C2.match = cur->m_local.as<int32_t>(); C2.src_mp = T->d_local_mp; C2.src_stride = cur->P.local_mp_cap;
  C2.gate = gr; C2.gate_min = gm;
  FB_TRY(edges_and_pose(cur, &T->map, &mpb, FB_POSE_FRONT_BIRD, T->wB, T->wF, 1, C2, s));              // :1400
  return finish_impl(cur, &T->map, gr, gm, T->defer_outlier_drop, s, T->min_inliers);
}

}  // namespace

extern "C" {

int fb_frame_bird_mappoint_match_dev(fb_frame *cur, const fb_map_points_bird *mpb, const int32_t *d_local, const int32_t *d_n_local,
                                     int window_size, float filter_size, const fb_matcher_params *matcher, void *stream) {
  FB_TRY(fb::check_device());
  FB_ARG(cur && bird_ok(mpb) && matcher && (!d_local || d_n_local));
  FB_ARG(d_local || mpb->stride <= cur->P.local_mpb_cap);
  hipStream_t s = fb::as_stream(stream);
  FB_TRY(m9_impl(cur, mpb, d_local, d_n_local, window_size, filter_size, matcher, s));
  return launch_commit(cur, commit_m9(cur, d_local), s);
}

int fb_frame_search_by_projection_dev(fb_frame *cur, const fb_frame *last, const fb_map_points *map, float th,
                                      const fb_matcher_params *matcher, void *stream) {
  FB_TRY(fb::check_device());
  FB_ARG(cur && last && cur != last && cur->B == last->B && cur->cap == last->cap && map_ok(cur, map) && matcher);
  hipStream_t s = fb::as_stream(stream);
  FB_TRY(m3_impl(cur, last, map, th, matcher, s));
  Commit C;
  memset(&C, 0, sizeof(C));
  C.kind = 1; C.match = cur->m_front.as<int32_t>(); C.src_mp = last->mp.as<int32_t>(); C.src_stride = cur->cap;
  return launch_commit(cur, C, s);
}

int fb_frame_pose_optimization_dev(fb_frame *f, const fb_map_points *map, const fb_map_points_bird *mpb, int mode, float wB,
                                   float wF, int which, void *stream) {
  FB_TRY(fb::check_device());
  FB_ARG(f && map_ok(f, map) && bird_ok(mpb));
  Commit C;
  memset(&C, 0, sizeof(C));
  return edges_and_pose(f, map, mpb, mode, wB, wF, which, C, fb::as_stream(stream));
}

int fb_frame_discard_outliers_dev(fb_frame *f, const fb_map_points *map, void *stream) {
  FB_TRY(fb::check_device());
  FB_ARG(f && map_ok(f, map));
  return discard_impl(f, map, FB_CNT_PROJ_MATCHES, 0, fb::as_stream(stream));
}

int fb_frame_match_bird_points_dev(fb_frame *cur, fb_frame *ref, fb_map_points_bird *mpb, int window_size, float filter_size,
                                   const fb_matcher_params *matcher, void *stream) {
  FB_TRY(fb::check_device());
  FB_ARG(cur && ref && cur != ref && cur->B == ref->B && cur->cap == ref->cap && bird_ok(mpb) && matcher);
  return bird_points_impl(cur, ref, mpb, window_size, filter_size, matcher, 0, fb::as_stream(stream));
}

int fb_frame_search_local_points_dev(fb_frame *f, const fb_map_points *map, const int32_t *d_local, const int32_t *d_n_local,
                                     float th, const fb_matcher_params *matcher, void *stream) {
  FB_TRY(fb::check_device());
  FB_ARG(f && map_ok(f, map) && matcher && (!d_local || d_n_local));
  FB_ARG(d_local || map->stride <= f->P.local_mp_cap);
  hipStream_t s = fb::as_stream(stream);
  FB_TRY(local_impl(f, map, d_local, d_n_local, th, matcher, s));
  Commit C;
  memset(&C, 0, sizeof(C));
  C.kind = 2; C.match = f->m_local.as<int32_t>(); C.src_mp = d_local; C.src_stride = f->P.local_mp_cap;
  return launch_commit(f, C, s);
}

int fb_frame_finish_dev(fb_frame *f, const fb_map_points *map, void *stream) {
  FB_TRY(fb::check_device());
  FB_ARG(f && map_ok(f, map));
  return finish_impl(f, map, nullptr, 0, 0, fb::as_stream(stream));
}

int fb_frame_drop_outliers_dev(fb_frame *f, void *stream) {
  FB_TRY(fb::check_device());
  FB_ARG(f);
  fb::ProfScope prof_(fb::P_TRACK_GLUE, fb::as_stream(stream));
  k_drop_outliers<<<f->B, WG, 0, fb::as_stream(stream)>>>(f->dev(), f->counts.as<int32_t>(), f->B, f->minInliers);
  FB_HIP(hipGetLastError());
  return FB_OK;
}

int fb_frame_track_motion_model_dev(fb_frame *cur, fb_frame *last, const fb_track_args *T, void *stream) {
  FB_TRY(fb::check_device());
  FB_ARG(cur && last && cur != last && cur->B == last->B && cur->cap == last->cap && track_args_ok(cur, T) && T->d_delta);
  return motion_model_impl(cur, last, T, fb::as_stream(stream));
}

int fb_frame_track_local_map_dev(fb_frame *cur, fb_frame *ref, const fb_track_args *T, void *stream) {
  FB_TRY(fb::check_device());
  FB_ARG(cur && ref && cur != ref && cur->B == ref->B && cur->cap == ref->cap && track_args_ok(cur, T));
  return local_map_impl(cur, ref, T, fb::as_stream(stream), T->gate_local_map != 0);
}

int fb_frame_track_dev(fb_frame *cur, fb_frame *last, const fb_track_args *T, void *stream) {
  FB_TRY(fb::check_device());
  FB_ARG(cur && last && cur != last && cur->B == last->B && cur->cap == last->cap && track_args_ok(cur, T) && T->d_delta);
  FB_TRY(motion_model_impl(cur, last, T, fb::as_stream(stream)));
  return local_map_impl(cur, last, T, fb::as_stream(stream), true);   // if (bOK) bOK = TrackLocalMap(), per sequence
}

int fb_frame_compute_bow_dev(fb_frame *f, const fb_vocabulary *voc, void *stream) {
  FB_TRY(fb::check_device());
  FB_ARG(f && voc);
  if (f->bowDone) return FB_OK;  // if (mBowVec.empty()), Frame.cc:630
  const size_t B = f->B, cap = f->cap;
  if (!f->bow_nw.p) {
    FB_TRY(f->bow_nw.alloc(B * 4)); FB_TRY(f->bow_ids.alloc(B * cap * 4)); FB_TRY(f->bow_vals.alloc(B * cap * 8));
    FB_TRY(f->fv_nn.alloc(B * 4)); FB_TRY(f->fv_ids.alloc(B * cap * 4)); FB_TRY(f->fv_start.alloc(B * (cap + 1) * 4));
    FB_TRY(f->fv_items.alloc(B * cap * 4));
  }
  fb_bow_transform_args A;
  memset(&A, 0, sizeof(A));
  A.batch = f->B; A.f_stride = f->cap; A.n_f = f->n.as<int32_t>(); A.desc = f->desc.as<uint8_t>(); A.levelsup = 4;  // Frame.cc:633
  A.n_words = f->bow_nw.as<int32_t>(); A.bow_ids = f->bow_ids.as<uint32_t>(); A.bow_vals = f->bow_vals.as<double>();
  A.fv_n_nodes = f->fv_nn.as<int32_t>(); A.fv_node_ids = f->fv_ids.as<uint32_t>(); A.fv_node_start = f->fv_start.as<int32_t>();
  A.fv_items = f->fv_items.as<int32_t>();
  FB_TRY(fb_bow_transform_dev(voc, &A, stream));
  f->bowDone = true;
  return FB_OK;
}

int fb_frame_track_using_bird_dev(fb_frame *cur, fb_frame *src, fb_frame *ref, const fb_track_args *T, void *stream) {
  FB_TRY(fb::check_device());
  FB_ARG(cur && src && ref && cur != src && cur != ref && cur->B == src->B && cur->cap == src->cap && cur->B == ref->B && cur->cap == ref->cap);
  FB_ARG(track_args_ok(cur, T) && T->d_delta);
  hipStream_t s = fb::as_stream(stream);
  fb_map_points_bird mpb = T->mpb;
  FB_TRY(fb_frame_predict_pose_dev(cur, src, T->d_delta, s));                                          // :2016-2034
  FB_TRY(m9_impl(cur, &mpb, T->d_local_mpb, T->d_n_local_mpb, 10, 0.05f, &M09, s));                    // :2036
  FB_TRY(launch_commit(cur, commit_m9(cur, T->d_local_mpb), s));
  FB_TRY(bird_points_impl(cur, ref, &mpb, 10, 0.05f, &M09, 11, s));                                    // :2038-2053: only where numPt <= 10
  Commit C;
  memset(&C, 0, sizeof(C));
  C.no_front = 1;
  FB_TRY(edges_and_pose(cur, &T->map, &mpb, FB_POSE_BIRD, 1.0f, 1.0f, 0, C, s));                       // BirdOptimization(&mCurrentFrame, 1.0)
  return bird_points_impl(cur, ref, &mpb, 10, 0.05f, &M09, 0, s);                                      // :2056
}

int fb_frame_copy_dev(fb_frame *dst, const fb_frame *src, void *stream) {
  FB_TRY(fb::check_device());
  FB_ARG(dst && src && dst != src && dst->B == src->B && dst->cap == src->cap);
  hipStream_t s = fb::as_stream(stream);
  const size_t B = src->B, cap = src->cap;
  if (src->bowDone && !dst->bow_nw.p) {
    FB_TRY(dst->bow_nw.alloc(B * 4)); FB_TRY(dst->bow_ids.alloc(B * cap * 4)); FB_TRY(dst->bow_vals.alloc(B * cap * 8));
    FB_TRY(dst->fv_nn.alloc(B * 4)); FB_TRY(dst->fv_ids.alloc(B * cap * 4)); FB_TRY(dst->fv_start.alloc(B * (cap + 1) * 4));
    FB_TRY(dst->fv_items.alloc(B * cap * 4));
  }
#define CP(buf) FB_HIP(hipMemcpyAsync(dst->buf.p, src->buf.p, src->buf.bytes, hipMemcpyDeviceToDevice, s))
  CP(n); CP(kps); CP(kps_un); CP(desc); CP(cs); CP(ci); CP(mp); CP(outlier);
  CP(nb); CP(bkps); CP(bdesc); CP(bcam); CP(bcs); CP(bci); CP(mpb); CP(boutlier);
  CP(Tcw); CP(counts);
  if (src->bowDone) { CP(bow_nw); CP(bow_ids); CP(bow_vals); CP(fv_nn); CP(fv_ids); CP(fv_start); CP(fv_items); }
#undef CP
  dst->bowDone = src->bowDone;
  return FB_OK;
}

int fb_frame_bow_view_dev(fb_frame *f, fb_bow_transform_args *v) {
  FB_ARG(f && v && f->bowDone);
  memset(v, 0, sizeof(*v));
  v->batch = f->B; v->f_stride = f->cap; v->n_f = f->n.as<int32_t>(); v->desc = f->desc.as<uint8_t>(); v->levelsup = 4;
  v->n_words = f->bow_nw.as<int32_t>(); v->bow_ids = f->bow_ids.as<uint32_t>(); v->bow_vals = f->bow_vals.as<double>();
  v->fv_n_nodes = f->fv_nn.as<int32_t>(); v->fv_node_ids = f->fv_ids.as<uint32_t>(); v->fv_node_start = f->fv_start.as<int32_t>();
  v->fv_items = f->fv_items.as<int32_t>();
  return FB_OK;
}

int fb_frame_search_by_bow_dev(fb_frame *cur, const fb_frame *kf, const fb_map_points *map, const fb_matcher_params *matcher,
                               int min_matches, void *stream) {
  FB_TRY(fb::check_device());
  FB_ARG(cur && kf && cur != kf && cur->B == kf->B && cur->cap == kf->cap && map_ok(cur, map) && matcher && min_matches >= 0);
  if (!cur->bowDone || !kf->bowDone) { fb::set_error("fb_frame_search_by_bow: ComputeBoW has not run on both frames"); return FB_ERR_ARG; }
  hipStream_t s = fb::as_stream(stream);
  FB_TRY(bow_impl(cur, kf, map, matcher, s));
  return launch_commit(cur, commit_bow(cur, kf, min_matches), s);
}

int fb_frame_track_reference_dev(fb_frame *cur, fb_frame *kf, fb_frame *ref, const fb_vocabulary *voc, const fb_track_args *T,
                                 void *stream) {
  FB_TRY(fb::check_device());
  FB_ARG(cur && kf && ref && voc && cur != kf && cur != ref && cur->B == kf->B && cur->cap == kf->cap && cur->B == ref->B && cur->cap == ref->cap);
  FB_ARG(track_args_ok(cur, T) && T->d_delta);
  if (!kf->bowDone) { fb::set_error("fb_frame_track_reference: the key frame's BoW is not computed (KeyFrame::ComputeBoW)"); return FB_ERR_ARG; }
  hipStream_t s = fb::as_stream(stream);
  fb_map_points_bird mpb = T->mpb;
  FB_TRY(fb_frame_predict_pose_dev(cur, kf, T->d_delta, s));                                           // :1185-1186
  FB_TRY(m9_impl(cur, &mpb, T->d_local_mpb, T->d_n_local_mpb, 10, 0.05f, &M09, s));                    // :1193-1194
  FB_TRY(launch_commit(cur, commit_m9(cur, T->d_local_mpb), s));
  FB_TRY(bird_points_impl(cur, ref, &mpb, 10, 0.05f, &M09, 10, s));                                    // :1196-1200
  FB_TRY(fb_frame_compute_bow_dev(cur, voc, s));                                                       // :1203
  FB_TRY(bow_impl(cur, kf, &T->map, &M07, s));                                                         // :1207-1210
  FB_TRY(edges_and_pose(cur, &T->map, &mpb, FB_POSE_FRONT_BIRD, T->wB, T->wF, 0, commit_bow(cur, kf, 15), s));  // :1212-1220
  return discard_impl(cur, &T->map, FB_CNT_BOW_MATCHES, 15, s);                                        // :1222-1241
}

int fb_frame_view_dev(fb_frame *f, fb_frame_view *v) {
  FB_ARG(f && v);
  v->batch = f->B; v->kp_stride = f->cap;
  v->n = f->n.as<int32_t>(); v->kps = f->kps.as<fb_keypoint>(); v->kps_un = f->kps_un.as<fb_keypoint>(); v->desc = f->desc.as<uint8_t>();
  v->map_point = f->mp.as<int32_t>(); v->outlier = f->outlier.as<uint8_t>();
  v->n_bird = f->nb.as<int32_t>(); v->kps_bird = f->bkps.as<fb_keypoint>(); v->desc_bird = f->bdesc.as<uint8_t>();
  v->bird_cam_xyz = f->bcam.as<float>(); v->map_point_bird = f->mpb.as<int32_t>(); v->bird_outlier = f->boutlier.as<uint8_t>();
  v->Tcw = f->Tcw.as<float>(); v->counts = f->counts.as<int32_t>();
  return FB_OK;
}

int fb_frame_download(fb_frame *f, const fb_frame_view *h, void *stream) {
  FB_TRY(fb::check_device());
  FB_ARG(f && h);
  hipStream_t s = fb::as_stream(stream);
  const size_t B = f->B, cap = f->cap, KP = sizeof(fb_keypoint);
#define DL(dst, buf, bytes) if (h->dst) FB_HIP(hipMemcpyAsync(h->dst, f->buf.p, bytes, hipMemcpyDeviceToHost, s))
  DL(n, n, B * 4); DL(kps, kps, B * cap * KP); DL(kps_un, kps_un, B * cap * KP); DL(desc, desc, B * cap * 32);
  DL(map_point, mp, B * cap * 4); DL(outlier, outlier, B * cap);
  DL(n_bird, nb, B * 4); DL(kps_bird, bkps, B * cap * KP); DL(desc_bird, bdesc, B * cap * 32); DL(bird_cam_xyz, bcam, B * cap * 12);
  DL(map_point_bird, mpb, B * cap * 4); DL(bird_outlier, boutlier, B * cap);
  DL(Tcw, Tcw, B * 48); DL(counts, counts, B * FB_CNT_COUNT * 4);
#undef DL
  FB_HIP(hipStreamSynchronize(s));
  return FB_OK;
}

int fb_frame_counts(fb_frame *f, int32_t *host_counts, float *host_Tcw, void *stream) {
  FB_TRY(fb::check_device());
  FB_ARG(f && host_counts);
  hipStream_t s = fb::as_stream(stream);
  const size_t B = f->B, nc = B * FB_CNT_COUNT;
  FB_HIP(hipMemcpyAsync(f->pinCounts, f->counts.p, nc * 4, hipMemcpyDeviceToHost, s));
  if (host_Tcw) FB_HIP(hipMemcpyAsync(f->pinCounts + nc, f->Tcw.p, B * 48, hipMemcpyDeviceToHost, s));
  FB_HIP(hipStreamSynchronize(s));
  memcpy(host_counts, f->pinCounts, nc * 4);
  if (host_Tcw) memcpy(host_Tcw, f->pinCounts + nc, B * 48);
  return FB_OK;
}

}  // extern "C"
