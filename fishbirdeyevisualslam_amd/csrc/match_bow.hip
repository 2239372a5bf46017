// match_bow.hip -- the vocabulary-node-gated ORBmatcher entry points on gfx950.
//
// Replaces (reference file:line):
//   ORBmatcher::SearchByBoW(KeyFrame*, Frame&, vector<MapPoint*>&)   src/ORBmatcher.cc:160-289   (M5)
//   ORBmatcher::SearchForTriangulation (+ CheckDistEpipolarLine)      src/ORBmatcher.cc:658-824, 141-158 (M7)
// The DBoW2::FeatureVector of each side arrives as CSR (node ids ascending, items in addFeature order,
// Thirdparty/DBoW2/DBoW2/FeatureVector.cpp:31-45).  One workgroup per keyframe/frame pair; the candidate
// side's descriptor table is staged in LDS; one lane per query feature.  A feature index lives in exactly one
// node, so the merge-join of the reference (lower_bound skips, :181-265) is a binary search per query.
//
// M5 is greedy in keyframe-feature order (a frame feature that already received a MapPoint is skipped,
// :210): resolved with the same fixed-point iteration as M2/M3 (see match.hip), claims here always block.
#include "fb_common.h"

namespace {

constexpr int TH_LOW = 50;        // ORBmatcher.cc:39
constexpr int HISTO_LENGTH = 30;  // :40
constexpr int NONE = 0x7fffffff;
constexpr int BOW_THREADS = 1024;

struct FVd {
  int n;
  const uint32_t *ids;
  const int32_t *start;
  const int32_t *items;
};

__device__ __forceinline__ FVd fv_of(const fb_feature_vector &v, int b) {
  FVd r;
  r.n = v.n_nodes[b];
  r.ids = v.node_ids + (size_t)b * v.node_stride;
  r.start = v.node_start + (size_t)b * (v.node_stride + 1);
  r.items = v.items + (size_t)b * v.item_stride;
  return r;
}

// node that owns flattened item position a: largest k with start[k] <= a
__device__ __forceinline__ int node_of_item(const FVd &v, int a) {
  int lo = 0, hi = v.n;  // start[lo] <= a < start[hi]
  while (hi - lo > 1) {
    const int m = (lo + hi) >> 1;
    if (v.start[m] <= a) lo = m; else hi = m;
  }
  return lo;
}

__device__ __forceinline__ int find_node(const FVd &v, uint32_t id) {  // exact match or -1
  int lo = 0, hi = v.n;
  while (lo < hi) {
    const int m = (lo + hi) >> 1;
    if (v.ids[m] < id) lo = m + 1; else hi = m;
  }
  return (lo < v.n && v.ids[lo] == id) ? lo : -1;
}

__device__ __forceinline__ int rot_bin(float rot) {  // ORBmatcher.cc:237-243
  const float factor = 1.0f / HISTO_LENGTH;
  if (rot < 0.0f) rot += 360.0f;
  int bin = (int)roundf(rot * factor);
  if (bin == HISTO_LENGTH) bin = 0;
  return bin;
}

__device__ void three_maxima(const int *sz, int &ind1, int &ind2, int &ind3) {  // ORBmatcher.cc:1905-1946
  int max1 = 0, max2 = 0, max3 = 0;
  ind1 = ind2 = ind3 = -1;
  for (int i = 0; i < HISTO_LENGTH; i++) {
    const int s = sz[i];
    if (s > max1) { max3 = max2; max2 = max1; max1 = s; ind3 = ind2; ind2 = ind1; ind1 = i; }
    else if (s > max2) { max3 = max2; max2 = s; ind3 = ind2; ind2 = i; }
    else if (s > max3) { max3 = s; ind3 = i; }
  }
  if (max2 < 0.1f * (float)max1) { ind2 = -1; ind3 = -1; }
  else if (max3 < 0.1f * (float)max1) { ind3 = -1; }
}

__device__ __forceinline__ void load_desc(const uint8_t *p, uint32_t d[8]) {
  const uint4 *q = reinterpret_cast<const uint4 *>(p);
  const uint4 a = q[0], b = q[1];
  d[0] = a.x; d[1] = a.y; d[2] = a.z; d[3] = a.w; d[4] = b.x; d[5] = b.y; d[6] = b.z; d[7] = b.w;
}

// ---------------------------------------------------------------------------------------------------------
// KFKF = false: M5 (keyframe -> frame).  KFKF = true: M6, SearchByBoW(pKF1, pKF2) (ORBmatcher.cc:523-656): the
// candidate side must carry a good MapPoint (:575-581), the distance test is strict (:598) and the result is
// indexed by the query side (vpMatches12[idx1], :602).
template <bool KFKF>
__global__ __launch_bounds__(BOW_THREADS) void k_match_bow_t(fb_bow_args A, const uint8_t *f_has_mp, int32_t *match12, int descInLds) {
  extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
  const int b = blockIdx.x, tid = threadIdx.x, nt = blockDim.x;
  const size_t ko = (size_t)b * A.kf_stride, fo = (size_t)b * A.f_stride;
  const int nF = A.n_f[b];
  const FVd K = fv_of(A.kf_fv, b), F = fv_of(A.f_fv, b);
  const int nQ = K.n > 0 ? K.start[K.n] : 0;  // flattened keyframe items = queries in serial order
  // descriptor table of the candidate side: LDS when it fits, else read from HBM/L2 (frames with > ~3400 key points)
  const size_t descBytes = descInLds ? (size_t)A.f_stride * 32 : 0;
  const uint4 *fdesc = descInLds ? reinterpret_cast<const uint4 *>(smem) : reinterpret_cast<const uint4 *>(A.f_desc + fo * 32);
  int *ownerA = reinterpret_cast<int *>(smem + descBytes);  // [f_stride]
  int *ownerB = ownerA + A.f_stride;
  int *assignA = ownerB + A.f_stride;                                // [kf item_stride]
  int *assignB = assignA + A.kf_fv.item_stride;
  int *fitems = assignB + A.kf_fv.item_stride;                       // [f item_stride] F.mFeatVec's items: walked by every query in every round
  __shared__ int s_changed, s_n, s_hist[HISTO_LENGTH], s_ind[3];
  {
    const int nFi = F.n > 0 ? min(F.start[F.n], A.f_fv.item_stride) : 0;
    for (int i = tid; i < nFi; i += nt) fitems[i] = F.items[i];
  }
  if (descInLds) {
    const uint4 *src = reinterpret_cast<const uint4 *>(A.f_desc + fo * 32);
    uint4 *dst = reinterpret_cast<uint4 *>(smem);
    for (int i = tid; i < nF * 2; i += nt) dst[i] = src[i];
  }
  for (int i = tid; i < nF; i += nt) ownerA[i] = NONE;
  for (int q = tid; q < nQ; q += nt) assignA[q] = NONE;
  __syncthreads();
  // What a query needs in every round of the fixed point -- its key-frame feature, whether that carries a MapPoint, the
  // frame-side node (two binary searches over the CSR in HBM) and its descriptor -- does not change between rounds: a lane
  // resolves it once for its (at most QPT) queries and keeps it in registers; a round then only touches LDS.  (Re-deriving it
  // per round was ~15 us of dependent global loads per round, most of the kernel.)
  constexpr int QPT = 4;
  const bool cached = nQ <= QPT * nt;
  int qc0[QPT], qc1[QPT];       // candidate range in F.items, empty = the query cannot match
  uint32_t qd[QPT][8];
#pragma unroll
  for (int s_ = 0; s_ < QPT; s_++) {
    qc0[s_] = 0; qc1[s_] = 0;
#pragma unroll
    for (int w = 0; w < 8; w++) qd[s_][w] = 0;
    const int q = tid + s_ * nt;
    if (cached && q < nQ) {
      const int realIdxKF = K.items[q];
      if (A.kf_has_mp[ko + realIdxKF]) {
        const int fi = find_node(F, K.ids[node_of_item(K, q)]);
        if (fi >= 0) { qc0[s_] = F.start[fi]; qc1[s_] = F.start[fi + 1]; load_desc(A.kf_desc + (ko + realIdxKF) * 32, qd[s_]); }
      }
    }
  }
  auto evaluate = [&](int q, int c0, int c1, const uint32_t *d) {
    int bestDist1 = 256, bestIdxF = -1, bestDist2 = 256;
    for (int c = c0; c < c1; c++) {
      const int realIdxF = fitems[c];
      if (ownerA[realIdxF] < q) continue;  // vpMapPointMatches[realIdxF] already set by an earlier feature
      if (KFKF && !f_has_mp[fo + realIdxF]) continue;
      const int dist = fb::hamming256(d, fdesc + realIdxF * 2);
      if (dist < bestDist1) { bestDist2 = bestDist1; bestDist1 = dist; bestIdxF = realIdxF; }
      else if (dist < bestDist2) bestDist2 = dist;
    }
    const bool low = KFKF ? bestDist1 < TH_LOW : bestDist1 <= TH_LOW;
    return (low && (float)bestDist1 < A.matcher.nnratio * (float)bestDist2) ? bestIdxF : NONE;
  };
  for (int round = 0; round <= nQ + 1; round++) {
    for (int i = tid; i < nF; i += nt) ownerB[i] = NONE;
    if (tid == 0) s_changed = 0;
    __syncthreads();
    if (cached) {
#pragma unroll
      for (int s_ = 0; s_ < QPT; s_++) {
        const int q = tid + s_ * nt;
        if (q >= nQ) continue;
        const int best = qc1[s_] > qc0[s_] ? evaluate(q, qc0[s_], qc1[s_], qd[s_]) : NONE;
        assignB[q] = best;
        if (best != assignA[q]) s_changed = 1;
        if (best != NONE) atomicMin(&ownerB[best], q);
      }
    } else {
      for (int q = tid; q < nQ; q += nt) {
        int best = NONE;
        const int realIdxKF = K.items[q];
        if (A.kf_has_mp[ko + realIdxKF]) {
          const int fi = find_node(F, K.ids[node_of_item(K, q)]);
          if (fi >= 0) {
            uint32_t d[8];
            load_desc(A.kf_desc + (ko + realIdxKF) * 32, d);
            best = evaluate(q, F.start[fi], F.start[fi + 1], d);
          }
        }
        assignB[q] = best;
        if (best != assignA[q]) s_changed = 1;
        if (best != NONE) atomicMin(&ownerB[best], q);
      }
    }
    __syncthreads();
    const int changed = s_changed;
    int *t = ownerA; ownerA = ownerB; ownerB = t;
    t = assignA; assignA = assignB; assignB = t;
    __syncthreads();
    if (!changed) break;
  }
  int *matchL = ownerB;
  for (int i = tid; i < nF; i += nt) matchL[i] = -1;
  if (tid < HISTO_LENGTH) s_hist[tid] = 0;
  if (tid == 0) s_n = 0;
  __syncthreads();
  const bool ori = A.matcher.check_orientation != 0;
  for (int q = tid; q < nQ; q += nt) {
    const int c = assignA[q];
    if (c == NONE) continue;
    const int realIdxKF = K.items[q];
    matchL[c] = realIdxKF;  // unique claimer: a claimed slot blocks every later query
    atomicAdd(&s_n, 1);
    if (ori) {
      const int bin = rot_bin(A.kf_kps[ko + realIdxKF].angle - A.f_kps[fo + c].angle);
      atomicAdd(&s_hist[bin], 1);
      assignB[q] = bin;
    }
  }
  __syncthreads();
  if (ori) {
    if (tid == 0) three_maxima(s_hist, s_ind[0], s_ind[1], s_ind[2]);
    __syncthreads();
    for (int q = tid; q < nQ; q += nt) {
      const int c = assignA[q];
      if (c == NONE) continue;
      const int bin = assignB[q];
      if (bin != s_ind[0] && bin != s_ind[1] && bin != s_ind[2]) { matchL[c] = -1; atomicSub(&s_n, 1); }
    }
    __syncthreads();
  }
  if (KFKF) {
    const int nK = A.n_kf[b];
    for (int i = tid; i < nK; i += nt) match12[ko + i] = -1;
    __syncthreads();
    for (int i = tid; i < nF; i += nt)
      if (matchL[i] >= 0) match12[ko + matchL[i]] = i;
  } else {
    for (int i = tid; i < nF; i += nt) A.match_f_to_kf[fo + i] = matchL[i];
  }
  if (tid == 0) A.nmatches[b] = s_n;
}

// ---------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(BOW_THREADS) void k_match_triangulation(fb_triangulation_args A, int descInLds) {
  extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
  const int b = blockIdx.x, tid = threadIdx.x, nt = blockDim.x;
  const size_t o1 = (size_t)b * A.kf1_stride, o2 = (size_t)b * A.kf2_stride;
  const int n1 = A.n1[b], n2 = A.n2[b];
  const FVd V1 = fv_of(A.fv1, b), V2 = fv_of(A.fv2, b);
  const int nQ = V1.n > 0 ? V1.start[V1.n] : 0;
  const size_t descBytes = descInLds ? (size_t)A.kf2_stride * 32 : 0;
  const uint4 *desc2 = descInLds ? reinterpret_cast<const uint4 *>(smem) : reinterpret_cast<const uint4 *>(A.desc2 + o2 * 32);
  float4 *k2 = reinterpret_cast<float4 *>(smem + descBytes);                         // x, y, angle, octave
  int *m12 = reinterpret_cast<int *>(smem + descBytes + (size_t)A.kf2_stride * 16);  // [kf1_stride]
  int *bins = m12 + A.kf1_stride;                                                    // [kf1_stride]
  __shared__ int s_n, s_hist[HISTO_LENGTH], s_ind[3];
  __shared__ float s_e[2], s_F[9];
  {
    if (descInLds) {
      const uint4 *src = reinterpret_cast<const uint4 *>(A.desc2 + o2 * 32);
      uint4 *dst = reinterpret_cast<uint4 *>(smem);
      for (int i = tid; i < n2 * 2; i += nt) dst[i] = src[i];
    }
    for (int i = tid; i < n2; i += nt) {
      const fb_keypoint k = A.kps2[o2 + i];
      k2[i] = make_float4(k.x, k.y, k.angle, __int_as_float(k.octave));
    }
  }
  for (int i = tid; i < n1; i += nt) { m12[i] = -1; bins[i] = -1; }
  if (tid < HISTO_LENGTH) s_hist[tid] = 0;
  if (tid < 9) s_F[tid] = A.F12[(size_t)b * 9 + tid];
  if (tid == 0) {
    s_n = 0;
    const float *Cw = A.Cw1 + (size_t)b * 3, *R = A.R2w + (size_t)b * 9, *t = A.t2w + (size_t)b * 3;
    float C2[3];
    for (int r = 0; r < 3; r++) C2[r] = ((R[r * 3] * Cw[0] + R[r * 3 + 1] * Cw[1]) + R[r * 3 + 2] * Cw[2]) + t[r];
    const float invz = 1.0f / C2[2];
    s_e[0] = A.fx * C2[0] * invz + A.cx;  // epipole of KF1's centre in KF2, ORBmatcher.cc:665-671
    s_e[1] = A.fy * C2[1] * invz + A.cy;
  }
  __syncthreads();
  const float ex = s_e[0], ey = s_e[1];
  const bool ori = A.matcher.check_orientation != 0;
  for (int q = tid; q < nQ; q += nt) {
    const int idx1 = V1.items[q];
    if (A.has_mp1[o1 + idx1]) continue;
    const int f2 = find_node(V2, V1.ids[node_of_item(V1, q)]);
    if (f2 < 0) continue;
    const fb_keypoint kp1 = A.kps1[o1 + idx1];
    uint32_t d[8];
    load_desc(A.desc1 + (o1 + idx1) * 32, d);
    // epipolar line in image 2: l = x1' F12 (CheckDistEpipolarLine)
    const float la = kp1.x * s_F[0] + kp1.y * s_F[3] + s_F[6];
    const float lb = kp1.x * s_F[1] + kp1.y * s_F[4] + s_F[7];
    const float lc = kp1.x * s_F[2] + kp1.y * s_F[5] + s_F[8];
    const float den = la * la + lb * lb;
    int bestDist = TH_LOW, bestIdx2 = -1;
    for (int c = V2.start[f2]; c < V2.start[f2 + 1]; c++) {
      const int idx2 = V2.items[c];
      if (A.has_mp2[o2 + idx2]) continue;
      const int dist = fb::hamming256(d, desc2 + idx2 * 2);
      if (dist > TH_LOW || dist > bestDist) continue;
      const float4 kp2 = k2[idx2];
      const int oct2 = __float_as_int(kp2.w);
      const float distex = ex - kp2.x, distey = ey - kp2.y;
      if (distex * distex + distey * distey < 100 * A.scale_factors[oct2]) continue;
      const float num = la * kp2.x + lb * kp2.y + lc;
      if (den == 0) continue;
      const float dsqr = num * num / den;
      if (dsqr < 3.84 * A.level_sigma2[oct2]) { bestIdx2 = idx2; bestDist = dist; }
    }
    if (bestIdx2 >= 0) {
      m12[idx1] = bestIdx2;
      atomicAdd(&s_n, 1);
      if (ori) {
        const int bin = rot_bin(kp1.angle - k2[bestIdx2].z);
        bins[idx1] = bin;
        atomicAdd(&s_hist[bin], 1);
      }
    }
  }
  __syncthreads();
  if (ori) {
    if (tid == 0) three_maxima(s_hist, s_ind[0], s_ind[1], s_ind[2]);
    __syncthreads();
    for (int i = tid; i < n1; i += nt) {
      const int bin = bins[i];
      if (bin >= 0 && bin != s_ind[0] && bin != s_ind[1] && bin != s_ind[2]) { m12[i] = -1; atomicSub(&s_n, 1); }
    }
    __syncthreads();
  }
  for (int i = tid; i < n1; i += nt) A.matches12[o1 + i] = m12[i];
  if (tid == 0) A.nmatches[b] = s_n;
}

int lds_ok(size_t bytes, const char *what) {
  if (bytes > 160 * 1024) { fb::set_error("%s: frame too large for the LDS-staged matcher (%zu B)", what, bytes); return FB_ERR_CAPACITY; }
  return FB_OK;
}

// host-pointer upload helper for a feature vector
// a DBoW2::FeatureVector's four arrays join the call's staged upload (fb::Stager)
void stage_fv(fb::Stager &st, const fb_feature_vector &h, fb_feature_vector &d, size_t B) {
  d = h;
  st.in((void **)&d.n_nodes, h.n_nodes, B * 4);
  st.in((void **)&d.node_ids, h.node_ids, B * (size_t)h.node_stride * 4);
  st.in((void **)&d.node_start, h.node_start, B * (size_t)(h.node_stride + 1) * 4);
  st.in((void **)&d.items, h.items, B * (size_t)h.item_stride * 4);
}

}  // namespace

extern "C" {

int fb_match_bow_dev(const fb_bow_args *A, void *stream) {
  FB_TRY(fb::check_device());
  FB_ARG(A && A->batch >= 0 && A->kf_stride > 0 && A->f_stride > 0 && A->kf_fv.item_stride >= 0 && A->f_fv.item_stride >= 0);
  if (A->batch == 0) return FB_OK;
  size_t lds = (size_t)A->f_stride * 32 + (size_t)A->f_stride * 8 + (size_t)A->kf_fv.item_stride * 8 + (size_t)A->f_fv.item_stride * 4 + 16;
  const int descInLds = lds <= 160 * 1024 - 512;
  if (!descInLds) lds -= (size_t)A->f_stride * 32;
  FB_TRY(lds_ok(lds, "fb_match_bow"));
  FB_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(k_match_bow_t<false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  fb::ProfScope prof_(fb::P_BOW, fb::as_stream(stream));
  k_match_bow_t<false><<<A->batch, BOW_THREADS, lds, fb::as_stream(stream)>>>(*A, nullptr, nullptr, descInLds);
  FB_HIP(hipGetLastError());
  return FB_OK;
}

int fb_match_bow_kf_dev(const fb_bow_kf_args *K, void *stream) {
  FB_TRY(fb::check_device());
  FB_ARG(K && K->batch >= 0 && K->kf1_stride > 0 && K->kf2_stride > 0 && K->fv1.item_stride >= 0 && K->fv2.item_stride >= 0);
  FB_ARG(K->has_mp2 && K->matches12);
  if (K->batch == 0) return FB_OK;
  fb_bow_args A{};  // KF1 plays the query ("kf") side, KF2 the candidate ("f") side
  A.batch = K->batch; A.kf_stride = K->kf1_stride; A.f_stride = K->kf2_stride;
  A.n_kf = K->n1; A.kf_kps = K->kps1; A.kf_desc = K->desc1; A.kf_has_mp = K->has_mp1; A.kf_fv = K->fv1;
  A.n_f = K->n2; A.f_kps = K->kps2; A.f_desc = K->desc2; A.f_fv = K->fv2;
  A.matcher = K->matcher; A.match_f_to_kf = nullptr; A.nmatches = K->nmatches;
  size_t lds = (size_t)A.f_stride * 32 + (size_t)A.f_stride * 8 + (size_t)A.kf_fv.item_stride * 8 + (size_t)A.f_fv.item_stride * 4 + 16;
  const int descInLds = lds <= 160 * 1024 - 512;
  if (!descInLds) lds -= (size_t)A.f_stride * 32;
  FB_TRY(lds_ok(lds, "fb_match_bow_kf"));
  FB_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(k_match_bow_t<true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  fb::ProfScope prof_(fb::P_BOW_KF, fb::as_stream(stream));
  k_match_bow_t<true><<<A.batch, BOW_THREADS, lds, fb::as_stream(stream)>>>(A, K->has_mp2, K->matches12, descInLds);
  FB_HIP(hipGetLastError());
  return FB_OK;
}

int fb_match_triangulation_dev(const fb_triangulation_args *A, void *stream) {
  FB_TRY(fb::check_device());
  FB_ARG(A && A->batch >= 0 && A->kf1_stride > 0 && A->kf2_stride > 0);
  if (A->batch == 0) return FB_OK;
  size_t lds = (size_t)A->kf2_stride * 48 + (size_t)A->kf1_stride * 8 + 16;
  const int descInLds = lds <= 160 * 1024 - 512;
  if (!descInLds) lds -= (size_t)A->kf2_stride * 32;
  FB_TRY(lds_ok(lds, "fb_match_triangulation"));
  FB_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(k_match_triangulation), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  fb::ProfScope prof_(fb::P_TRIANG, fb::as_stream(stream));
  k_match_triangulation<<<A->batch, BOW_THREADS, lds, fb::as_stream(stream)>>>(*A, descInLds);
  FB_HIP(hipGetLastError());
  return FB_OK;
}

// host-pointer drop-ins: one staged upload, the same kernels, one staged download (fb::Stager, fb_common.h)
#define UPF(buf, field, bytes) st.in((void **)&D.field, H->field, (bytes));

int fb_match_bow(const fb_bow_args *H) {
  FB_TRY(fb::check_device());
  FB_ARG(H && H->batch >= 0 && H->match_f_to_kf && H->nmatches);
  fb_bow_args D = *H;
  const size_t B = H->batch, ks = H->kf_stride, fs = H->f_stride;
  fb::Stager st;
  UPF(b0, n_kf, B * 4) UPF(b1, kf_kps, B * ks * sizeof(fb_keypoint)) UPF(b2, kf_desc, B * ks * 32) UPF(b3, kf_has_mp, B * ks)
  UPF(b4, n_f, B * 4) UPF(b5, f_kps, B * fs * sizeof(fb_keypoint)) UPF(b6, f_desc, B * fs * 32)
  stage_fv(st, H->kf_fv, D.kf_fv, B);
  stage_fv(st, H->f_fv, D.f_fv, B);
  st.out((void **)&D.match_f_to_kf, H->match_f_to_kf, B * fs * 4, true);  // copy-in: entries past n keep the caller's contents
  st.out((void **)&D.nmatches, H->nmatches, B * 4, false);
  FB_TRY(st.commit(nullptr));
  FB_TRY(fb_match_bow_dev(&D, nullptr));
  return st.fetch(nullptr);
}

int fb_match_bow_kf(const fb_bow_kf_args *H) {
  FB_TRY(fb::check_device());
  FB_ARG(H && H->batch >= 0 && H->matches12 && H->nmatches);
  fb_bow_kf_args D = *H;
  const size_t B = H->batch, s1 = H->kf1_stride, s2 = H->kf2_stride;
  fb::Stager st;
  UPF(b0, n1, B * 4) UPF(b1, kps1, B * s1 * sizeof(fb_keypoint)) UPF(b2, desc1, B * s1 * 32) UPF(b3, has_mp1, B * s1)
  UPF(b4, n2, B * 4) UPF(b5, kps2, B * s2 * sizeof(fb_keypoint)) UPF(b6, desc2, B * s2 * 32) UPF(b7, has_mp2, B * s2)
  stage_fv(st, H->fv1, D.fv1, B);
  stage_fv(st, H->fv2, D.fv2, B);
  st.out((void **)&D.matches12, H->matches12, B * s1 * 4, true);
  st.out((void **)&D.nmatches, H->nmatches, B * 4, false);
  FB_TRY(st.commit(nullptr));
  FB_TRY(fb_match_bow_kf_dev(&D, nullptr));
  return st.fetch(nullptr);
}

int fb_match_triangulation(const fb_triangulation_args *H) {
  FB_TRY(fb::check_device());
  FB_ARG(H && H->batch >= 0 && H->matches12 && H->nmatches);
  fb_triangulation_args D = *H;
  const size_t B = H->batch, s1 = H->kf1_stride, s2 = H->kf2_stride;
  fb::Stager st;
  UPF(b0, n1, B * 4) UPF(b1, kps1, B * s1 * sizeof(fb_keypoint)) UPF(b2, desc1, B * s1 * 32) UPF(b3, has_mp1, B * s1)
  UPF(b4, n2, B * 4) UPF(b5, kps2, B * s2 * sizeof(fb_keypoint)) UPF(b6, desc2, B * s2 * 32) UPF(b7, has_mp2, B * s2)
  UPF(b8, F12, B * 36) UPF(b9, Cw1, B * 12) UPF(b10, R2w, B * 36) UPF(b11, t2w, B * 12)
  stage_fv(st, H->fv1, D.fv1, B);
  stage_fv(st, H->fv2, D.fv2, B);
  st.out((void **)&D.matches12, H->matches12, B * s1 * 4, true);
  st.out((void **)&D.nmatches, H->nmatches, B * 4, false);
  FB_TRY(st.commit(nullptr));
  FB_TRY(fb_match_triangulation_dev(&D, nullptr));
  return st.fetch(nullptr);
}
#undef UPF

}  // extern "C"
