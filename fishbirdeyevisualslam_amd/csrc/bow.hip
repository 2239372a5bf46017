// bow.hip -- DBoW2 vocabulary transform of a frame's descriptors on gfx950 (Frame::ComputeBoW, src/Frame.cc:628-635).
//
// Replaces (reference file:line):
//   TemplatedVocabulary::transform(features, BowVector&, FeatureVector&, levelsup)   Thirdparty/DBoW2/DBoW2/TemplatedVocabulary.h:1126-1194
//   TemplatedVocabulary::transform(feature, word_id, weight, nid, levelsup)          :1217-1259
//   FORB::distance                                                                    Thirdparty/DBoW2/DBoW2/FORB.cpp:85-103
//   BowVector::addWeight / normalize(L1), FeatureVector::addFeature                   BowVector.cpp:30-84, FeatureVector.cpp:31-45
//
// Phase 1 (k_bow_descend, 64 features per workgroup): a lane walks one feature down the tree (per level: Hamming distance to
// the <= k children, first minimum wins).  Phases 2 and 3: one workgroup per image.  Phase 2: the (word, feature) keys are bitonic-sorted in LDS; segment heads sum their
// weights in feature order (the reference's += order), one lane adds the L1 norm in ascending word order (the std::map
// iteration order), everyone divides.  Phase 3: the same sort on (node, feature) keys yields the FeatureVector CSR.
#include "fb_common.h"

namespace {

constexpr int BOW_T = 1024;
constexpr int BOW_MAXF = 4096;
constexpr int BOW_KMAX = 12;  // children per node handled with batched loads (the ORB vocabulary has 10)
constexpr unsigned long long KEY_NONE = ~0ull;

__device__ __forceinline__ void bitonic_sort(unsigned long long *key, int n2, int tid, int nt) {
  for (int k = 2; k <= n2; k <<= 1)
    for (int j = k >> 1; j > 0; j >>= 1) {
      for (int i = tid; i < n2; i += nt) {
        const int ixj = i ^ j;
        if (ixj > i) {
          const unsigned long long a = key[i], b = key[ixj];
          const bool up = (i & k) == 0;
          if ((a > b) == up) { key[i] = b; key[ixj] = a; }
        }
      }
      __syncthreads();
    }
}

// exclusive scan of one int per thread over the BOW_T-thread block (s_wv: [BOW_T / 64]); *total = block sum
__device__ __forceinline__ int bow_excl_scan(int v, int *s_wv, int *total) {
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  int inc = v;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) { const int t = __shfl_up(inc, o, 64); if (lane >= o) inc += t; }
  __syncthreads();
  if (lane == 63) s_wv[wv] = inc;
  __syncthreads();
  int base = 0, tot = 0;
#pragma unroll
  for (int w = 0; w < BOW_T / 64; w++) { const int x = s_wv[w]; if (w < wv) base += x; tot += x; }
  *total = tot;
  return base + inc - v;
}

// Phase 1 on its own grid: 64 features per workgroup.  One workgroup per image walking all of its features kept the whole
// descent -- 2000 x 12 x 6 random 32-byte reads -- on ONE compute unit's L1 (the largest part of the kernel at batch 1).
// Results per feature go through the output arrays, which phase 2 reads before it writes anything:
//   bow_vals[i] = leaf weight, bow_ids[i] = word id, fv_node_ids[i] = node id at nid_level.
__global__ __launch_bounds__(64) void k_bow_descend(fb_vocabulary V, fb_bow_transform_args A) {
  const int b = blockIdx.y, i = blockIdx.x * 64 + threadIdx.x;
  const size_t fo = (size_t)b * A.f_stride;
  if (i >= A.n_f[b] || i >= A.f_stride) return;
  const int nid_level = V.L - A.levelsup;
  unsigned int nid = 0;
  uint32_t d[8];
  const uint4 *dq = reinterpret_cast<const uint4 *>(A.desc + (fo + i) * 32);
  const uint4 d0 = dq[0], d1 = dq[1];
  d[0] = d0.x; d[1] = d0.y; d[2] = d0.z; d[3] = d0.w; d[4] = d1.x; d[5] = d1.y; d[6] = d1.z; d[7] = d1.w;
  int final_id = 0, level = 0;
  int c0 = V.child_start[0], c1 = V.child_start[1];
  do {
    ++level;
    const int cnt = c1 - c0;
    int best, best_d;
    if (cnt <= BOW_KMAX) {
      // the usual k <= 12 children: every child id, then every child descriptor, is requested before the first use (two
      // memory round trips per level instead of two per child); indices past the node's children are clamped and ignored
      int ids[BOW_KMAX], dist[BOW_KMAX];
#pragma unroll
      for (int u = 0; u < BOW_KMAX; u++) ids[u] = V.children[c0 + min(u, cnt - 1)];
#pragma unroll
      for (int u = 0; u < BOW_KMAX; u++) dist[u] = fb::hamming256(d, reinterpret_cast<const uint4 *>(V.descriptors + (size_t)ids[u] * 32));
      best = ids[0]; best_d = dist[0];
#pragma unroll
      for (int u = 1; u < BOW_KMAX; u++)
        if (u < cnt && dist[u] < best_d) { best_d = dist[u]; best = ids[u]; }  // first minimum wins (TemplatedVocabulary.h:1240-1247)
    } else {
      best = V.children[c0];
      best_d = fb::hamming256(d, reinterpret_cast<const uint4 *>(V.descriptors + (size_t)best * 32));
      for (int c = c0 + 1; c < c1; c++) {
        const int id = V.children[c];
        const int dist = fb::hamming256(d, reinterpret_cast<const uint4 *>(V.descriptors + (size_t)id * 32));
        if (dist < best_d) { best_d = dist; best = id; }
      }
    }
    final_id = best;
    if (level == nid_level) nid = (unsigned int)final_id;
    c0 = V.child_start[final_id];
    c1 = V.child_start[final_id + 1];
  } while (c1 > c0 && level < 64);
  A.bow_vals[fo + i] = V.weights[final_id];
  A.bow_ids[fo + i] = (uint32_t)V.word_ids[final_id];
  A.fv_node_ids[fo + i] = nid;
}

__global__ __launch_bounds__(BOW_T) void k_bow_transform(fb_vocabulary V, fb_bow_transform_args A) {
  __shared__ unsigned long long s_key[BOW_MAXF];
  __shared__ double s_w[BOW_MAXF];        // per feature: leaf weight; later per unique word: summed weight
  __shared__ unsigned int s_aux[BOW_MAXF]; // per feature: node id at nid_level; later scan scratch
  __shared__ double s_norm;
  const int b = blockIdx.x, tid = threadIdx.x, nt = blockDim.x;
  const size_t fo = (size_t)b * A.f_stride;
  const int n = min(A.n_f[b], A.f_stride);
  int n2 = 1;
  while (n2 < n) n2 <<= 1;
  if (n2 < 2) n2 = 2;
  // ---- phase 1 happened in k_bow_descend: fetch its per-feature results ---------------------------------------------
  for (int i = tid; i < n2; i += nt) {
    unsigned long long key = KEY_NONE;
    if (i < n) {
      const double w = A.bow_vals[fo + i];
      if (w > 0) key = ((unsigned long long)A.bow_ids[fo + i] << 32) | (unsigned int)i;
      s_w[i] = w;
      s_aux[i] = A.fv_node_ids[fo + i];
    }
    s_key[i] = key;
  }
  if (tid == 0) s_norm = 0.0;
  __syncthreads();
  // ---- phase 2: BowVector ---------------------------------------------------------------------------------------------
  bitonic_sort(s_key, n2, tid, nt);
  // unique words: rank of each segment head = number of heads before it (serial count per lane over a block-strided
  // layout would cost a scan; n <= 4096, so each head counts earlier heads with a block-wide prefix in two steps)
  uint32_t *bow_ids = A.bow_ids + fo;
  double *bow_vals = A.bow_vals + fo;
  // step a: flag heads, per-thread chunk counts
  const int chunk = (n2 + nt - 1) / nt;
  const int i0 = min(tid * chunk, n2), i1 = min(i0 + chunk, n2);
  int heads = 0;
  for (int i = i0; i < i1; i++) {
    const unsigned long long k = s_key[i];
    if (k != KEY_NONE && (i == 0 || (s_key[i - 1] >> 32) != (k >> 32))) heads++;
  }
  __shared__ int s_wv[BOW_T / 64];
  __shared__ double s_sum[BOW_MAXF];   // summed weight per unique word, by rank
  int nw;
  {
    int rank = bow_excl_scan(heads, s_wv, &nw);
    for (int i = i0; i < i1; i++) {
      const unsigned long long k = s_key[i];
      if (k == KEY_NONE) break;
      if (i == 0 || (s_key[i - 1] >> 32) != (k >> 32)) {
        double acc = s_w[(unsigned int)k];  // addWeight: the first occurrence inserts v, later ones += v, in feature order
        for (int j = i + 1; j < n2 && s_key[j] != KEY_NONE && (s_key[j] >> 32) == (k >> 32); j++) acc += s_w[(unsigned int)s_key[j]];
        bow_ids[rank] = (uint32_t)(k >> 32);
        s_sum[rank] = acc;
        rank++;
      }
    }
  }
  __syncthreads();
  if (tid == 0) {  // BowVector::normalize(L1): ascending word order, one lane (the order of the additions is the result), from LDS
    double norm = 0.0;
    int r = 0;
    for (; r + 8 <= nw; r += 8) {
      double v8[8];
#pragma unroll
      for (int u = 0; u < 8; u++) v8[u] = fabs(s_sum[r + u]);
#pragma unroll
      for (int u = 0; u < 8; u++) norm += v8[u];
    }
    for (; r < nw; r++) norm += fabs(s_sum[r]);
    s_norm = norm;
    A.n_words[b] = nw;
  }
  __syncthreads();
  {
    const double norm = s_norm;
    for (int r = tid; r < nw; r += nt) bow_vals[r] = norm > 0.0 ? s_sum[r] / norm : s_sum[r];
  }
  __syncthreads();
  // ---- phase 3: FeatureVector (features with weight > 0 only, :1157-1161) ---------------------------------------------
  for (int i = tid; i < n2; i += nt) {
    unsigned long long key = KEY_NONE;
    if (i < n && s_w[i] > 0) key = ((unsigned long long)s_aux[i] << 32) | (unsigned int)i;
    s_key[i] = key;
  }
  __syncthreads();
  bitonic_sort(s_key, n2, tid, nt);
  heads = 0;
  int valid = 0;
  for (int i = i0; i < i1; i++) {
    const unsigned long long k = s_key[i];
    if (k == KEY_NONE) break;
    valid++;
    if (i == 0 || (s_key[i - 1] >> 32) != (k >> 32)) heads++;
  }
  __shared__ int s_valid;
  if (tid == 0) s_valid = 0;
  __syncthreads();
  if (valid) atomicAdd(&s_valid, valid);
  int nNodes;
  int32_t *st = A.fv_node_start + (size_t)b * (A.f_stride + 1);
  {
    int rank = bow_excl_scan(heads, s_wv, &nNodes);
    for (int i = i0; i < i1; i++) {
      const unsigned long long k = s_key[i];
      if (k == KEY_NONE) break;
      A.fv_items[fo + i] = (int32_t)(unsigned int)k;
      if (i == 0 || (s_key[i - 1] >> 32) != (k >> 32)) {
        A.fv_node_ids[fo + rank] = (uint32_t)(k >> 32);
        st[rank] = i;
        rank++;
      }
    }
  }
  __syncthreads();
  if (tid == 0) { st[nNodes] = s_valid; A.fv_n_nodes[b] = nNodes; }
}

}  // namespace

extern "C" {

int fb_bow_transform_dev(const fb_vocabulary *V, const fb_bow_transform_args *A, void *stream) {
  FB_TRY(fb::check_device());
  FB_ARG(V && A && A->batch >= 0 && A->f_stride > 0 && A->f_stride <= BOW_MAXF && V->n_nodes > 0 && V->L > 0);
  FB_ARG(V->child_start && V->children && V->descriptors && V->weights && V->word_ids);
  FB_ARG(A->n_f && A->desc && A->n_words && A->bow_ids && A->bow_vals && A->fv_n_nodes && A->fv_node_ids && A->fv_node_start && A->fv_items);
  if (A->batch == 0) return FB_OK;
  fb::ProfScope prof_(fb::P_BOWT, fb::as_stream(stream));
  k_bow_descend<<<dim3((A->f_stride + 63) / 64, A->batch), 64, 0, fb::as_stream(stream)>>>(*V, *A);
  k_bow_transform<<<A->batch, BOW_T, 0, fb::as_stream(stream)>>>(*V, *A);
  FB_HIP(hipGetLastError());
  return FB_OK;
}

int fb_bow_transform(const fb_vocabulary *HV, const fb_bow_transform_args *H) {
  FB_TRY(fb::check_device());
  FB_ARG(HV && H && H->batch >= 0 && HV->n_nodes > 0);
  if (H->batch == 0) return FB_OK;
  fb_vocabulary V = *HV;
  fb_bow_transform_args D = *H;
  const size_t B = H->batch, fs = H->f_stride, nn = HV->n_nodes;
  fb::DevBuf v0, v1, v2, v3, v4, b0, b1, o0, o1, o2, o3, o4, o5, o6;
  FB_TRY(v0.upload(HV->child_start, (nn + 1) * 4)); V.child_start = v0.as<int32_t>();
  FB_TRY(v1.upload(HV->children, (size_t)HV->child_start[nn] * 4)); V.children = v1.as<int32_t>();
  FB_TRY(v2.upload(HV->descriptors, nn * 32)); V.descriptors = v2.as<uint8_t>();
  FB_TRY(v3.upload(HV->weights, nn * 8)); V.weights = v3.as<double>();
  FB_TRY(v4.upload(HV->word_ids, nn * 4)); V.word_ids = v4.as<int32_t>();
  FB_TRY(b0.upload(H->n_f, B * 4)); D.n_f = b0.as<int32_t>();
  FB_TRY(b1.upload(H->desc, B * fs * 32)); D.desc = b1.as<uint8_t>();
  FB_TRY(o0.alloc(B * 4)); D.n_words = o0.as<int32_t>();
  FB_TRY(o1.upload(H->bow_ids, B * fs * 4)); D.bow_ids = o1.as<uint32_t>();  // copy-in: entries past n_f keep the caller's contents (those in [count, n_f) are scratch)
  FB_TRY(o2.upload(H->bow_vals, B * fs * 8)); D.bow_vals = o2.as<double>();
  FB_TRY(o3.alloc(B * 4)); D.fv_n_nodes = o3.as<int32_t>();
  FB_TRY(o4.upload(H->fv_node_ids, B * fs * 4)); D.fv_node_ids = o4.as<uint32_t>();
  FB_TRY(o5.upload(H->fv_node_start, B * (fs + 1) * 4)); D.fv_node_start = o5.as<int32_t>();
  FB_TRY(o6.upload(H->fv_items, B * fs * 4)); D.fv_items = o6.as<int32_t>();
  FB_TRY(fb_bow_transform_dev(&V, &D, nullptr));
  FB_HIP(hipDeviceSynchronize());
  FB_TRY(o0.download(H->n_words, B * 4));
  FB_TRY(o1.download(H->bow_ids, B * fs * 4));
  FB_TRY(o2.download(H->bow_vals, B * fs * 8));
  FB_TRY(o3.download(H->fv_n_nodes, B * 4));
  FB_TRY(o4.download(H->fv_node_ids, B * fs * 4));
  FB_TRY(o5.download(H->fv_node_start, B * (fs + 1) * 4));
  return o6.download(H->fv_items, B * fs * 4);
}

}  // extern "C"
