// match.hip -- ORBmatcher hot-path entry points as CDNA4 kernels (gfx950).
//
// Replaces (reference file:line):
//   DescriptorDistance                         src/ORBmatcher.cc:1951-1967
//   Frame::AssignFeaturesToGrid / PosInGrid*   src/Frame.cc:381-411, 548-570
//   Frame::GetFeaturesInArea[Birdview]         src/Frame.cc:493-546, 572-626
//   SearchByProjection(Frame&, const Frame&)   src/ORBmatcher.cc:1329-1471   (M3)
//   SearchByProjection(Frame&, vector<MP*>&)   src/ORBmatcher.cc:46-138      (M2)
//   BirdMapPointMatch                          src/ORBmatcher.cc:1763-1902   (M9)
//   BirdviewMatch (isProject=0)                src/ORBmatcher.cc:1602-1760   (M8)
//   ComputeThreeMaxima                         src/ORBmatcher.cc:1905-1946
//
// Layout: one workgroup per problem (frame pair).  The target frame's descriptor table
// (n x 32 B), keypoint x/y/octave and the grid CSR are staged once into LDS with coalesced
// 16-byte loads; every query lane then walks its grid window out of LDS and scores
// candidates with v_bcnt_u32_b32.  HBM traffic per problem is therefore the algorithmic
// minimum (each input byte read once, each output written once).
//
// Serial semantics: M2/M3 skip a candidate that an EARLIER query already took (with a map
// point that has observations).  That dependency is resolved by a fixed-point iteration:
// every round all queries re-pick their best candidate given the previous round's claims
// ("owner[c] = smallest claiming query index"); query q only honours claims of q' < q.
// By induction on q the iteration converges to exactly the serial result (query 0 is final
// after round 0, query q after at most round q); in practice 2-3 rounds.
#include "fb_common.h"

#include <type_traits>
#include "fb_frame_geom.h"

namespace {

constexpr int TH_HIGH = 100;      // ORBmatcher.cc:38
constexpr int TH_LOW = 50;        // :39
constexpr int HISTO_LENGTH = 30;  // :40
constexpr int NONE = 0x7fffffff;
constexpr int MATCH_THREADS = 1024;

struct TargetLds {  // target frame staged in LDS
  const uint4 *desc;     // [n][2]
  const float2 *xy;      // [n]
  const uint8_t *oct;    // [n] (nullptr when the kernel's callbacks do not ask for it: Carve::withOct)
  const uint16_t *cs;    // [ncell+1]
  const uint32_t *items; // [n] key point index | octave << 16 in cell order: the level test of a walk needs no second load
  uint32_t descLds;      // LDS byte address of the descriptor table, or DESC_NOT_IN_LDS (then `desc` points into HBM / L2)
};
constexpr uint32_t DESC_NOT_IN_LDS = 0xFFFFFFFFu;

// Distance of a query to key point i of the staged frame.  `desc` is a generic pointer (LDS or global, decided at launch), so
// reading through it is a FLAT load; the wave-uniform branch gives each side a typed load (ds_read_b128 /
// global_load_dwordx4).  Worth 2 % of k_proj_frame.  (Probe builds of that kernel's first round: callbacks compiled out 19 k
// of 135 k cycles, constant instead of this distance 131 k, no top-K insertion 21 k -- the insertion block of full_search is
// where the time goes)
__device__ __forceinline__ int target_hamming(const TargetLds &T, const uint32_t a[8], int i) {
  typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
  u32x4 b0, b1;
  if (T.descLds != DESC_NOT_IN_LDS) {
    typedef __attribute__((address_space(3))) const u32x4 lds_u4;
    lds_u4 *p = reinterpret_cast<lds_u4 *>((uintptr_t)(T.descLds + (uint32_t)i * 32u));
    b0 = p[0]; b1 = p[1];
  } else {
    typedef __attribute__((address_space(1))) const u32x4 glb_u4;
    glb_u4 *p = reinterpret_cast<glb_u4 *>((uintptr_t)(T.desc + (size_t)i * 2));
    b0 = p[0]; b1 = p[1];
  }
  return __popc(a[0] ^ b0.x) + __popc(a[1] ^ b0.y) + __popc(a[2] ^ b0.z) + __popc(a[3] ^ b0.w) +
         __popc(a[4] ^ b1.x) + __popc(a[5] ^ b1.y) + __popc(a[6] ^ b1.z) + __popc(a[7] ^ b1.w);
}

// Frame::GetFeaturesInArea (Frame.cc:493-546, inclusive cell loops) and
// Frame::GetFeaturesInAreaBirdview (Frame.cc:572-626, exclusive loops, no min offset).
template <bool BIRD, typename F>
__device__ __forceinline__ void for_area(const fb_grid_geom &g, const TargetLds &T, float x, float y, float r,
                                         int minLevel, int maxLevel, F &&f) {
  int nMinCellX, nMaxCellX, nMinCellY, nMaxCellY;
  if (BIRD) {
    nMinCellX = max(0, (int)floorf((x - r) * g.inv_w));
    if (nMinCellX >= g.cols) return;
    nMaxCellX = min(g.cols - 1, (int)ceilf((x + r) * g.inv_w));
    if (nMaxCellX < 0) return;
    nMinCellY = max(0, (int)floorf((y - r) * g.inv_h));
    if (nMinCellY >= g.rows) return;
    nMaxCellY = min(g.rows - 1, (int)ceilf((y + r) * g.inv_h));
    if (nMaxCellY < 0) return;
    nMaxCellX -= 1;  // ix < nMaxCellX
    nMaxCellY -= 1;
  } else {
    nMinCellX = max(0, (int)floorf((x - g.min_x - r) * g.inv_w));
    if (nMinCellX >= g.cols) return;
    nMaxCellX = min(g.cols - 1, (int)ceilf((x - g.min_x + r) * g.inv_w));
    if (nMaxCellX < 0) return;
    nMinCellY = max(0, (int)floorf((y - g.min_y - r) * g.inv_h));
    if (nMinCellY >= g.rows) return;
    nMaxCellY = min(g.rows - 1, (int)ceilf((y - g.min_y + r) * g.inv_h));
    if (nMaxCellY < 0) return;
  }
  const bool bCheckLevels = (minLevel > 0) || (maxLevel >= 0);
  // SIMT shape: the lanes of a wave walk different windows, so calling f (descriptor distance + bookkeeping) straight from the
  // item loop runs it once per loop position at which ANY lane has a key point, with one or two lanes active.  The key points a
  // lane finds are therefore queued -- eight 16-bit indices in four registers, oldest on top -- and f runs over the queues
  // after the walk: as many passes as the fullest queue of the wave, every lane that still has an entry taking part.  A full
  // queue is emptied on the spot.  The order in which a lane sees its key points is unchanged.  (Measured with the batched
  // loads below: round 0 of k_proj_frame 163 k -> 140 k cycles per frame, profiles/probes/m3_stamps.py.)
  uint32_t q0 = 0, q1 = 0, q2 = 0, q3 = 0;
  int qn = 0;
  auto flush = [&]() {
    for (int t = qn - 1; t >= 0; t--) {  // position t: 0 = newest (low half of q0)
      const uint32_t w = (t >> 1) == 0 ? q0 : ((t >> 1) == 1 ? q1 : ((t >> 1) == 2 ? q2 : q3));
#ifndef FB_WALK_NO_F   // (probe builds: the walk skeleton alone, results wrong)
      f((int)((t & 1) ? (w >> 16) : (w & 0xFFFFu)));
#else
      asm volatile("" :: "v"(w));
#endif
    }
    qn = 0;
  };
  for (int ix = nMinCellX; ix <= nMaxCellX; ix++) {
    const int cbase = ix * g.rows;
    // cells (ix, nMinCellY..nMaxCellY) are contiguous in the CSR (cell id = ix*rows+iy)
    const int j0 = T.cs[cbase + nMinCellY], j1 = T.cs[cbase + nMaxCellY + 1];
    uint32_t itNext = j0 < j1 ? T.items[j0] : 0u;
    for (int j = j0; j < j1; j++) {
      // an item rejected by its level costs no LDS wait (the octave rides in the item word, the next item is in flight): the
      // windows of a dense region scan hundreds of items (up to 290 per walk on the bench's drive, 3.5 of them pass both tests)
      const uint32_t it = itNext;
      itNext = T.items[min(j + 1, j1 - 1)];
      const int idx = (int)(it & 0xFFFFu);
      if (bCheckLevels) {
        const int o = (int)(it >> 16);
        if (o < minLevel) continue;
        if (maxLevel >= 0 && o > maxLevel) continue;
      }
      const float2 p = T.xy[idx];
      const float distx = p.x - x, disty = p.y - y;
      if (fabsf(distx) < r && fabsf(disty) < r) {
        if (qn == 8) flush();
        q3 = (q3 << 16) | (q2 >> 16); q2 = (q2 << 16) | (q1 >> 16); q1 = (q1 << 16) | (q0 >> 16); q0 = (q0 << 16) | (uint32_t)idx;
        qn++;
      }
    }
  }
  flush();
}

__device__ __forceinline__ int rot_bin(float rot) {  // ORBmatcher.cc:1434-1439
  const float factor = 1.0f / HISTO_LENGTH;
  if (rot < 0.0f) rot += 360.0f;
  int bin = (int)roundf(rot * factor);
  if (bin == HISTO_LENGTH) bin = 0;
  return bin;
}

// ComputeThreeMaxima, ORBmatcher.cc:1905-1946
__device__ void three_maxima(const int *sz, int &ind1, int &ind2, int &ind3) {
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

__device__ __forceinline__ void xform(const float *T, const float *X, float *o) {  // rows 0..2 of a 3x4
#pragma unroll
  for (int r = 0; r < 3; r++) o[r] = ((T[r * 4 + 0] * X[0] + T[r * 4 + 1] * X[1]) + T[r * 4 + 2] * X[2]) + T[r * 4 + 3];
}

// LDS carve for a target frame of n keypoints / ncell cells. All offsets 16-B aligned.
struct Carve {
  size_t desc, xy, oct, cs, items, end;
  bool hasOct;
  // withDesc = false: the descriptor table stays in HBM/L2 (frames too large for LDS, e.g. the 2*nFeatures
  // initialisation extractor); only the key point positions, octaves and the grid are staged
  __host__ __device__ Carve(int n, int ncell, bool withDesc = true, bool withOct = true) {
    auto up = [](size_t v) { return (v + 15) & ~(size_t)15; };
    hasOct = withOct;
    desc = 0;
    xy = up(desc + (withDesc ? (size_t)n * 32 : 0));
    oct = up(xy + (size_t)n * 8);
    cs = up(oct + (withOct ? (size_t)n : 0));
    items = up(cs + (size_t)(ncell + 1) * 2);
    end = up(items + (size_t)n * 4);
  }
};

__device__ __forceinline__ TargetLds stage_target(uint8_t *smem, const Carve &cv, int n, int ncell,
                                                  const fb_keypoint *kps, const uint8_t *desc, const int32_t *cs,
                                                  const int32_t *items, bool withDesc = true) {
  uint32_t *ldesc = reinterpret_cast<uint32_t *>(smem + cv.desc);
  float2 *lxy = reinterpret_cast<float2 *>(smem + cv.xy);
  uint8_t *loct = smem + cv.oct;
  uint16_t *lcs = reinterpret_cast<uint16_t *>(smem + cv.cs);
  uint32_t *litems = reinterpret_cast<uint32_t *>(smem + cv.items);
  const int tid = threadIdx.x, nt = blockDim.x;
  // descriptor table: n*32 B as 16-byte vectors (rows are 32-B aligned in the C-ABI arrays)
  const uint4 *src = reinterpret_cast<const uint4 *>(desc);
  uint4 *dst = reinterpret_cast<uint4 *>(ldesc);
  if (withDesc)
    for (int i = tid; i < n * 2; i += nt) dst[i] = src[i];
  for (int i = tid; i < n; i += nt) {
    const fb_keypoint k = kps[i];
    lxy[i] = make_float2(k.x, k.y);
    if (cv.hasOct) loct[i] = (uint8_t)k.octave;
  }
  for (int i = tid; i <= ncell; i += nt) lcs[i] = (uint16_t)cs[i];
  const int nitems = cs[ncell];
  for (int i = tid; i < nitems; i += nt) {
    const int idx = items[i];
    litems[i] = (uint32_t)idx | ((uint32_t)(kps[idx].octave & 0xff) << 16);
  }
  TargetLds T{withDesc ? reinterpret_cast<const uint4 *>(ldesc) : src, lxy, cv.hasOct ? loct : nullptr, lcs, litems,
              withDesc ? (uint32_t)(uintptr_t)ldesc : DESC_NOT_IN_LDS};
  return T;
}

// ---------------------------------------------------------------------------------------
// M3  SearchByProjection(CurrentFrame, LastFrame, th, bMono=true)
// ---------------------------------------------------------------------------------------
// Every round after the first re-decides each query from a per-query cache of its CACHE_K best admissible candidates
// (sorted by distance, ties in grid-walk order = the reference's "first minimum wins"), so the grid walk and the
// Hamming distances are done once; a query whose cached candidates are all taken and whose cache is incomplete
// falls back to the full walk.  cacheK = 0 (LDS too small) keeps the full walk every round.
constexpr int CACHE_K = 4;
// (Measured and dropped, round 3: round 0 -- projection, grid walk and distances of every query -- is 78 % of this kernel for
// one frame (profiles/probes/m3_stamps.py), so it was moved to its own grid of 256-query workgroups with the serial rule
// resolved on candidate lists afterwards, as M2 does.  Same results, no gain: 94 instead of 97 us per frame at batch 1 (a
// query's walk is a chain of dependent loads that is as long on its own workgroup, and the second kernel stages the grid
// again), and 0.226 instead of 0.137 ms per step at batch 256.)
#ifdef FB_MATCH_STAMPS
__device__ unsigned long long g_m3_stamps[8];  // probe build only (block 0): staging | order | round 0 | later rounds | commit | rounds | launches
#define M3_T0() unsigned long long mt_ = 0; if (blockIdx.x == 0 && threadIdx.x == 0) mt_ = __builtin_amdgcn_s_memtime();
#define M3_TICK(slot_) if (blockIdx.x == 0 && threadIdx.x == 0) { const unsigned long long t_ = __builtin_amdgcn_s_memtime(); atomicAdd(&g_m3_stamps[slot_], t_ - mt_); mt_ = t_; }
#define M3_COUNT(slot_) if (blockIdx.x == 0 && threadIdx.x == 0) atomicAdd(&g_m3_stamps[slot_], 1ull);
#else
#define M3_T0()
#define M3_TICK(slot_)
#define M3_COUNT(slot_)
#endif

__global__ __launch_bounds__(MATCH_THREADS) void k_proj_frame(fb_proj_frame_args A, int cacheK, int descInLds) {
  // (its callbacks never ask for a target key point's octave by index: no by-index octave array in LDS, Carve::withOct = false)
  extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
  typedef unsigned short u16;
  constexpr int NONE16 = 0xFFFF;
  const int b = blockIdx.x, tid = threadIdx.x, nt = blockDim.x;
  const int ncell = A.grid.cols * A.grid.rows;
  const size_t co = (size_t)b * A.cur_stride, lo = (size_t)b * A.last_stride;
  const int ncur = A.n_cur[b], nlast = A.n_last[b];
  M3_T0()
  const Carve cv(A.cur_stride, ncell, descInLds != 0, false);
  const TargetLds T = stage_target(smem, cv, ncur, ncell, A.cur_kps + co, A.cur_desc + co * 32,
                                   A.cur_cell_start + (size_t)b * (ncell + 1), A.cur_cell_items + co, descInLds != 0);
  int *ownerA = reinterpret_cast<int *>(smem + cv.end);  // [cur_stride]
  int *ownerB = ownerA + A.cur_stride;                   // [cur_stride]
  uint32_t *cache = reinterpret_cast<uint32_t *>(ownerB + A.cur_stride);  // [last_stride][cacheK]  dist << 16 | idx
  u16 *assignA = reinterpret_cast<u16 *>(cache + (size_t)A.last_stride * cacheK);  // [last_stride]
  u16 *assignB = assignA + A.last_stride;                // [last_stride]
  u16 *perm = assignB + A.last_stride;                   // [last_stride] queries sorted by octave (processing order)
  uint8_t *meta = reinterpret_cast<uint8_t *>(perm + A.last_stride);  // [last_stride] cached count | complete << 7
  __shared__ int s_changed, s_n, s_hist[HISTO_LENGTH], s_ind[3], s_oct[FB_MAX_LEVELS + 1];
  __shared__ float s_T[12];
  if (tid < 12) s_T[tid] = A.cur_Tcw[(size_t)b * 12 + tid];
  if (tid <= FB_MAX_LEVELS) s_oct[tid] = 0;
  const uint8_t *blocked0 = A.cur_blocked ? A.cur_blocked + co : nullptr;
  float thEff = A.th;   // the second attempt (retry_below > 0 and fewer matches than that) searches with retry_th
  __syncthreads();
  M3_TICK(0)
  // The search radius depends only on the octave: hand the lanes of a wave queries of the same octave so that their
  // grid walks have the same length (the query INDEX keeps deciding priorities, only the processing order changes).
  auto oct_bin = [&](int q) -> int {
    const int o = A.last_valid[lo + q] ? A.last_octave[lo + q] : FB_MAX_LEVELS;
    return o < 0 ? 0 : (o > FB_MAX_LEVELS ? FB_MAX_LEVELS : o);
  };
  for (int q = tid; q < nlast; q += nt) atomicAdd(&s_oct[oct_bin(q)], 1);
  __syncthreads();
  if (tid == 0) {
    int run = 0;
    for (int i = 0; i <= FB_MAX_LEVELS; i++) { const int c = s_oct[i]; s_oct[i] = run; run += c; }
  }
  __syncthreads();
  for (int q = tid; q < nlast; q += nt) perm[atomicAdd(&s_oct[oct_bin(q)], 1)] = (u16)q;
  __syncthreads();
  M3_TICK(1)

  // full grid walk of query q against the claims in `owner`; fill = also (re)build the query's cache
  // fill = also (re)build the query's cache.  The callback only APPENDS the eligible candidates (distance
  // <= TH_HIGH, walk order) to eight registers; the stable top-K selection runs after the walk, once, in straight-line code.
  // More than eight eligible candidates: the cache is left empty and incomplete, i.e. later rounds walk again.
  auto full_search = [&](int q, const int *owner, const bool fill) -> int {
    int best = NONE;
    uint32_t e0 = 0xFFFFFFFFu, e1 = e0, e2 = e0, e3 = e0, e4 = e0, e5 = e0, e6 = e0, e7 = e0;  // newest in e0
    int nElig = 0;
    if (A.last_valid[lo + q]) {
      float X[3] = {A.last_xw[(lo + q) * 3], A.last_xw[(lo + q) * 3 + 1], A.last_xw[(lo + q) * 3 + 2]};
      float pc[3];
      xform(s_T, X, pc);
      const float xc = pc[0], yc = pc[1];
      const float invzc = (float)(1.0 / pc[2]);
      if (!(invzc < 0)) {
        const float u = A.cam.fx * xc * invzc + A.cam.cx;
        const float v = A.cam.fy * yc * invzc + A.cam.cy;
        if (!(u < A.cam.min_x || u > A.cam.max_x) && !(v < A.cam.min_y || v > A.cam.max_y)) {
          const int oct = A.last_octave[lo + q];
          const float radius = thEff * A.scale_factors[oct];
          uint32_t d[8];
          const uint4 *dq = reinterpret_cast<const uint4 *>(A.last_desc + (lo + q) * 32);
          const uint4 d0 = dq[0], d1 = dq[1];
          d[0] = d0.x; d[1] = d0.y; d[2] = d0.z; d[3] = d0.w; d[4] = d1.x; d[5] = d1.y; d[6] = d1.z; d[7] = d1.w;
          int bestDist = 256;
          for_area<false>(A.grid, T, u, v, radius, oct - 1, oct + 1, [&](int i2) {
            const int own = owner[i2];
            if (own == -1) return;      // occupied on entry: never a candidate
            const int dist = target_hamming(T, d, i2);
            if (fill && dist <= TH_HIGH) {
              nElig++;
              e7 = e6; e6 = e5; e5 = e4; e4 = e3; e3 = e2; e2 = e1; e1 = e0;
              e0 = ((uint32_t)dist << 16) | (uint32_t)i2;
            }
            if (own < q) return;        // taken by an earlier query
            if (dist < bestDist) { bestDist = dist; best = i2; }
          });
          if (bestDist > TH_HIGH) best = NONE;
        }
      }
    }
    if (fill) {
      // stable top-K by distance over the recorded candidates, oldest first (equal keys keep walk order; once an entry has
      // been displaced everything behind it moves down one place: comparing the displaced entry again would let it jump over
      // an equal-distance neighbour and break the walk order among ties)
      uint32_t top[CACHE_K];
#pragma unroll
      for (int k = 0; k < CACHE_K; k++) top[k] = 0xFFFFFFFFu;
      const bool over = nElig > 8;
      const uint32_t ent[8] = {e7, e6, e5, e4, e3, e2, e1, e0};  // position 7 = newest; the oldest recorded one is at 8 - nElig
#pragma unroll
      for (int i = 0; i < 8; i++) {
        if (i >= 8 - nElig && !over) {
          uint32_t e = ent[i];
          bool shift = false;
#pragma unroll
          for (int k = 0; k < CACHE_K; k++) {
            if (shift || (e >> 16) < (top[k] >> 16)) { const uint32_t t = top[k]; top[k] = e; e = t; shift = true; }
          }
        }
      }
      const int n = over ? 0 : (nElig < CACHE_K ? nElig : CACHE_K);
      uint32_t *cq = cache + (uint32_t)q * CACHE_K;
#pragma unroll
      for (int k = 0; k < CACHE_K; k++) cq[k] = top[k];
      meta[q] = (uint8_t)(n | ((nElig <= CACHE_K) ? 0x80 : 0));
    }
    return best;
  };

  // Tracking.cc:1339-1349 in one launch: nmatches = SearchByProjection(th); if (nmatches < 20) { fill(mvpMapPoints, NULL);
  // nmatches = SearchByProjection(2 * th); } -- the second search starts from scratch on the staged frame
  for (int attempt = 0; attempt < 2; attempt++) {
  for (int i = tid; i < ncur; i += nt) ownerA[i] = (blocked0 && blocked0[i]) ? -1 : NONE;
  for (int q = tid; q < nlast; q += nt) assignA[q] = NONE16;
  __syncthreads();
  for (int round = 0; round <= nlast + 1; round++) {
    for (int i = tid; i < ncur; i += nt) ownerB[i] = (blocked0 && blocked0[i]) ? -1 : NONE;
    if (tid == 0) s_changed = 0;
    __syncthreads();
    for (int pq = tid; pq < nlast; pq += nt) {
      const int q = perm[pq];
      int best = NONE;
      bool walk = round == 0 || cacheK == 0;
      const bool fillNow = walk && cacheK > 0;
      if (!walk) {
        const int m = meta[q], n = m & 0x7f;
        for (int k = 0; k < n; k++) {
          const int i2 = (int)(cache[(size_t)q * cacheK + k] & 0xFFFFu);
          if (!(ownerA[i2] < q)) { best = i2; break; }
        }
        walk = best == NONE && !(m & 0x80);
      }
      if (walk) best = full_search(q, ownerA, fillNow);   // ONE call site: one copy of the walk and its callbacks in the kernel
      const int best16 = best == NONE ? NONE16 : best;
      assignB[q] = (u16)best16;
      if (best16 != assignA[q]) s_changed = 1;
      if (best != NONE && A.last_obs_pos[lo + q]) atomicMin(&ownerB[best], q);
    }
    __syncthreads();
    const int changed = s_changed;
    int *t = ownerA; ownerA = ownerB; ownerB = t;
    u16 *t16 = assignA; assignA = assignB; assignB = t16;
    __syncthreads();
    if (round == 0) { M3_TICK(2) } else { M3_TICK(3) }
    M3_COUNT(5)
    if (!changed) break;
  }

  // commit: last writer wins; rotation histogram culling (ORBmatcher.cc:1446-1468)
  int *matchL = ownerB;  // reuse
  u16 *binQ = assignB;   // reuse: histogram bin of each accepted query
  for (int i = tid; i < ncur; i += nt) matchL[i] = -1;
  if (tid < HISTO_LENGTH) s_hist[tid] = 0;
  if (tid == 0) s_n = 0;
  __syncthreads();
  const bool ori = A.matcher.check_orientation != 0;
  for (int q = tid; q < nlast; q += nt) {
    const int c = assignA[q];
    if (c == NONE16) continue;
    atomicMax(&matchL[c], q);
    atomicAdd(&s_n, 1);
    if (ori) {
      const int bin = rot_bin(A.last_angle[lo + q] - A.cur_kps[co + c].angle);
      atomicAdd(&s_hist[bin], 1);
      binQ[q] = (u16)bin;
    }
  }
  __syncthreads();
  if (ori) {
    if (tid == 0) three_maxima(s_hist, s_ind[0], s_ind[1], s_ind[2]);
    __syncthreads();
    for (int q = tid; q < nlast; q += nt) {
      const int c = assignA[q];
      if (c == NONE16) continue;
      const int bin = binQ[q];
      if (bin != s_ind[0] && bin != s_ind[1] && bin != s_ind[2]) {
        matchL[c] = -1;
        atomicSub(&s_n, 1);
      }
    }
    __syncthreads();
  }
  __syncthreads();
  if (attempt == 0 && A.retry_below > 0 && s_n < A.retry_below) {  // (workgroup-uniform)
    thEff = A.retry_th;
    // matchL / binQ alias ownerB / assignB: which of the two buffers holds what is irrelevant for a search from scratch
    __syncthreads();
    continue;
  }
  for (int i = tid; i < ncur; i += nt) A.match_cur_to_last[co + i] = matchL[i];
  if (tid == 0) { A.nmatches[b] = s_n; if (A.retried) A.retried[b] = attempt; }
  break;
  }
  M3_TICK(4)
  M3_COUNT(6)
}

// ---------------------------------------------------------------------------------------
// M4  SearchByProjection(CurrentFrame, pKF, sAlreadyFound, th, ORBdist)  (ORBmatcher.cc:1473-1600)
// Same greedy structure as M3; the search level comes from MapPoint::PredictScale and every
// accepted slot blocks all later key-frame points (mvpMapPoints[i2] != NULL, :1545).
// ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(MATCH_THREADS) void k_proj_kf(fb_proj_kf_args A, int descInLds) {
  extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
  typedef unsigned short u16;
  constexpr int NONE16 = 0xFFFF;
  const int b = blockIdx.x, tid = threadIdx.x, nt = blockDim.x;
  const int ncell = A.grid.cols * A.grid.rows;
  const size_t co = (size_t)b * A.cur_stride, ko = (size_t)b * A.kf_stride;
  const int ncur = A.n_cur[b], nkf = A.n_kf[b];
  const Carve cv(A.cur_stride, ncell, descInLds != 0);
  const TargetLds T = stage_target(smem, cv, ncur, ncell, A.cur_kps + co, A.cur_desc + co * 32,
                                   A.cur_cell_start + (size_t)b * (ncell + 1), A.cur_cell_items + co, descInLds != 0);
  int *ownerA = reinterpret_cast<int *>(smem + cv.end);  // [cur_stride]
  int *ownerB = ownerA + A.cur_stride;                   // [cur_stride]
  u16 *assignA = reinterpret_cast<u16 *>(ownerB + A.cur_stride);  // [kf_stride] chosen slot or NONE16
  u16 *assignB = assignA + A.kf_stride;                  // [kf_stride]
  __shared__ int s_changed, s_n, s_hist[HISTO_LENGTH], s_ind[3];
  __shared__ float s_T[12], s_Ow[3];
  if (tid < 12) s_T[tid] = A.cur_Tcw[(size_t)b * 12 + tid];
  if (tid == 64) fb::camera_centre(A.cur_Tcw + (size_t)b * 12, s_Ow);
  const uint8_t *blocked0 = A.cur_blocked ? A.cur_blocked + co : nullptr;
  for (int i = tid; i < ncur; i += nt) ownerA[i] = (blocked0 && blocked0[i]) ? -1 : NONE;
  for (int q = tid; q < nkf; q += nt) assignA[q] = NONE16;
  __syncthreads();

  // the geometric gates are cheap and are re-evaluated every round instead of being stored per point
  auto gate = [&](int q, float &u, float &v) -> int {
    if (!A.kf_valid[ko + q]) return -1;
    const float X[3] = {A.kf_xw[(ko + q) * 3], A.kf_xw[(ko + q) * 3 + 1], A.kf_xw[(ko + q) * 3 + 2]};
    float pc[3];
    xform(s_T, X, pc);
    const float invzc = (float)(1.0 / pc[2]);
    u = A.cam.fx * pc[0] * invzc + A.cam.cx;
    v = A.cam.fy * pc[1] * invzc + A.cam.cy;
    if ((u < A.cam.min_x || u > A.cam.max_x) || (v < A.cam.min_y || v > A.cam.max_y)) return -1;
    const float dist3D = fb::norm3(X[0] - s_Ow[0], X[1] - s_Ow[1], X[2] - s_Ow[2]);
    const float maxD = A.kf_max_dist[ko + q];
    if (dist3D < 0.8f * A.kf_min_dist[ko + q] || dist3D > 1.2f * maxD) return -1;
    return fb::predict_scale(maxD, dist3D, A.log_scale_factor, A.n_levels);
  };

  for (int round = 0; round <= nkf + 1; round++) {
    for (int i = tid; i < ncur; i += nt) ownerB[i] = (blocked0 && blocked0[i]) ? -1 : NONE;
    if (tid == 0) s_changed = 0;
    __syncthreads();
    for (int q = tid; q < nkf; q += nt) {
      int best = NONE;
      float u, v;
      const int lvl = gate(q, u, v);
      if (lvl >= 0) {
        const float radius = A.th * A.scale_factors[lvl];
        uint32_t d[8];
        const uint4 *dq = reinterpret_cast<const uint4 *>(A.kf_desc + (ko + q) * 32);
        const uint4 d0 = dq[0], d1 = dq[1];
        d[0] = d0.x; d[1] = d0.y; d[2] = d0.z; d[3] = d0.w; d[4] = d1.x; d[5] = d1.y; d[6] = d1.z; d[7] = d1.w;
        int bestDist = 256;
        for_area<false>(A.grid, T, u, v, radius, lvl - 1, lvl + 1, [&](int i2) {
          if (ownerA[i2] < q) return;
          const int dist = target_hamming(T, d, i2);
          if (dist < bestDist) { bestDist = dist; best = i2; }
        });
        if (bestDist > A.orb_dist) best = NONE;
      }
      const int best16 = best == NONE ? NONE16 : best;
      assignB[q] = (u16)best16;
      if (best16 != assignA[q]) s_changed = 1;
      if (best != NONE) atomicMin(&ownerB[best], q);
    }
    __syncthreads();
    const int changed = s_changed;
    int *t = ownerA; ownerA = ownerB; ownerB = t;
    u16 *t16 = assignA; assignA = assignB; assignB = t16;
    __syncthreads();
    if (!changed) break;
  }

  int *matchL = ownerB;
  u16 *binQ = assignB;
  for (int i = tid; i < ncur; i += nt) matchL[i] = -1;
  if (tid < HISTO_LENGTH) s_hist[tid] = 0;
  if (tid == 0) s_n = 0;
  __syncthreads();
  const bool ori = A.matcher.check_orientation != 0;
  for (int q = tid; q < nkf; q += nt) {
    const int c = assignA[q];
    if (c == NONE16) continue;
    matchL[c] = q;  // a claimed slot blocks every later point: one claimer per slot
    atomicAdd(&s_n, 1);
    if (ori) {
      const int bin = rot_bin(A.kf_angle[ko + q] - A.cur_kps[co + c].angle);
      atomicAdd(&s_hist[bin], 1);
      binQ[q] = (u16)bin;
    }
  }
  __syncthreads();
  if (ori) {
    if (tid == 0) three_maxima(s_hist, s_ind[0], s_ind[1], s_ind[2]);
    __syncthreads();
    for (int q = tid; q < nkf; q += nt) {
      const int c = assignA[q];
      if (c == NONE16) continue;
      const int bin = binQ[q];
      if (bin != s_ind[0] && bin != s_ind[1] && bin != s_ind[2]) {
        matchL[c] = -1;
        atomicSub(&s_n, 1);
      }
    }
    __syncthreads();
  }
  for (int i = tid; i < ncur; i += nt) A.match_cur_to_kf[co + i] = matchL[i];
  if (tid == 0) A.nmatches[b] = s_n;
}

// ---------------------------------------------------------------------------------------
// M2  SearchByProjection(Frame&, const vector<MapPoint*>&, th)
// ---------------------------------------------------------------------------------------
#ifdef FB_MATCH_STAMPS
__device__ int g_m2_rounds[4];  // probe build only: rounds / queries in view / launches of k_proj_points (block 0)
#endif
__global__ __launch_bounds__(MATCH_THREADS) void k_proj_points(fb_proj_points_args A, int descInLds) {
  extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
  const int b = blockIdx.x, tid = threadIdx.x, nt = blockDim.x;
  const int ncell = A.grid.cols * A.grid.rows;
  const size_t co = (size_t)b * A.cur_stride, mo = (size_t)b * A.mp_stride;
  const int ncur = A.n_cur[b], nmp = A.n_mp[b];
  const Carve cv(A.cur_stride, ncell, descInLds != 0);
  const TargetLds T = stage_target(smem, cv, ncur, ncell, A.cur_kps + co, A.cur_desc + co * 32,
                                   A.cur_cell_start + (size_t)b * (ncell + 1), A.cur_cell_items + co, descInLds != 0);
  int *ownerA = reinterpret_cast<int *>(smem + cv.end);
  int *ownerB = ownerA + A.cur_stride;
  int *assignA = ownerB + A.cur_stride;
  int *assignB = assignA + A.mp_stride;
  __shared__ int s_changed, s_n;
  const uint8_t *blocked0 = A.cur_blocked ? A.cur_blocked + co : nullptr;
  for (int i = tid; i < ncur; i += nt) ownerA[i] = (blocked0 && blocked0[i]) ? -1 : NONE;
  for (int q = tid; q < nmp; q += nt) assignA[q] = NONE;
  const bool bFactor = A.th != 1.0f;
  __syncthreads();
  for (int round = 0; round <= nmp + 1; round++) {
    for (int i = tid; i < ncur; i += nt) ownerB[i] = (blocked0 && blocked0[i]) ? -1 : NONE;
    if (tid == 0) s_changed = 0;
    __syncthreads();
    for (int q = tid; q < nmp; q += nt) {
      int best = NONE;
      if (A.mp_track[mo + q]) {
        const int lvl = A.mp_level[mo + q];
        float r = A.mp_view_cos[mo + q] > 0.998 ? 2.5f : 4.0f;  // RadiusByViewingCos
        if (bFactor) r *= A.th;
        uint32_t d[8];
        const uint4 *dq = reinterpret_cast<const uint4 *>(A.mp_desc + (mo + q) * 32);
        const uint4 d0 = dq[0], d1 = dq[1];
        d[0] = d0.x; d[1] = d0.y; d[2] = d0.z; d[3] = d0.w; d[4] = d1.x; d[5] = d1.y; d[6] = d1.z; d[7] = d1.w;
        int bestDist = 256, bestLevel = -1, bestDist2 = 256, bestLevel2 = -1, bestIdx = -1;
        for_area<false>(A.grid, T, A.mp_proj[(mo + q) * 2], A.mp_proj[(mo + q) * 2 + 1], r * A.scale_factors[lvl],
                        lvl - 1, lvl, [&](int idx) {
          if (ownerA[idx] < q) return;
          const int dist = target_hamming(T, d, idx);
          if (dist < bestDist) {
            bestDist2 = bestDist; bestDist = dist; bestLevel2 = bestLevel; bestLevel = T.oct[idx]; bestIdx = idx;
          } else if (dist < bestDist2) {
            bestLevel2 = T.oct[idx]; bestDist2 = dist;
          }
        });
        if (bestDist <= TH_HIGH && !(bestLevel == bestLevel2 && bestDist > A.matcher.nnratio * bestDist2)) best = bestIdx;
      }
      assignB[q] = best;
      if (best != assignA[q]) s_changed = 1;
      if (best != NONE && A.mp_obs_pos[mo + q]) atomicMin(&ownerB[best], q);
    }
    __syncthreads();
    const int changed = s_changed;
    int *t = ownerA; ownerA = ownerB; ownerB = t;
    t = assignA; assignA = assignB; assignB = t;
    __syncthreads();
#ifdef FB_MATCH_STAMPS
    if (b == 0 && tid == 0) atomicAdd(&g_m2_rounds[0], 1);
#endif
    if (!changed) break;
  }
#ifdef FB_MATCH_STAMPS
  if (b == 0 && tid == 0) { atomicAdd(&g_m2_rounds[2], 1); int c = 0; for (int q = 0; q < nmp; q++) c += A.mp_track[mo + q]; atomicAdd(&g_m2_rounds[1], c); }
#endif
  int *matchL = ownerB;
  for (int i = tid; i < ncur; i += nt) matchL[i] = -1;
  if (tid == 0) s_n = 0;
  __syncthreads();
  for (int q = tid; q < nmp; q += nt) {
    const int c = assignA[q];
    if (c == NONE) continue;
    atomicMax(&matchL[c], q);
    atomicAdd(&s_n, 1);
  }
  __syncthreads();
  for (int i = tid; i < ncur; i += nt) A.match_cur_to_mp[co + i] = matchL[i];
  if (tid == 0) A.nmatches[b] = s_n;
}

// ---------------------------------------------------------------------------------------
// M2 in two phases (taken when the caller provides a workspace): the expensive, owner-independent part of a query -- the
// grid walk and the Hamming distances of its candidates -- is done ONCE, by many workgroups per problem; the serial
// "already taken" rule is then resolved by one workgroup per problem on the cached candidate lists.  In the one-kernel
// version every round repeated every walk inside a single workgroup: 0.68 ms per frame for a 4000-point local map at
// batch 1 (rocprof, round 3), most of a tracked frame.
//   candidate record: dist << 20 | octave << 16 | key point index, in grid-walk order (the order decides ties);
//   M2_K records per query + a count; a query with more candidates keeps count = M2_OVER and is walked again each round.
// ---------------------------------------------------------------------------------------
constexpr int M2_K = 8, M2_OVER = 255, M2_CAND_THREADS = 256;

struct GridLds { const float2 *xy; const uint8_t *oct; const uint16_t *cs; const uint16_t *items; };
__device__ __forceinline__ TargetLds stage_grid_only(uint8_t *smem, const Carve &cv, int n, int ncell, const fb_keypoint *kps,
                                                     const uint8_t *desc, const int32_t *cs, const int32_t *items) {
  return stage_target(smem, cv, n, ncell, kps, desc, cs, items, false);
}

__device__ __forceinline__ float m2_radius(const fb_proj_points_args &A, size_t e, bool bFactor, int lvl) {
  float r = A.mp_view_cos[e] > 0.998 ? 2.5f : 4.0f;  // RadiusByViewingCos
  if (bFactor) r *= A.th;
  return r * A.scale_factors[lvl];
}

__global__ __launch_bounds__(M2_CAND_THREADS) void k_m2_candidates(fb_proj_points_args A, uint32_t *cand, uint8_t *ncand) {
  extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
  const int b = blockIdx.y, tid = threadIdx.x;
  const int ncell = A.grid.cols * A.grid.rows;
  const size_t co = (size_t)b * A.cur_stride, mo = (size_t)b * A.mp_stride;
  const int ncur = A.n_cur[b], nmp = A.n_mp[b];
  if ((int)(blockIdx.x * M2_CAND_THREADS) >= nmp) return;
  const Carve cv(A.cur_stride, ncell, false);
  const TargetLds T = stage_grid_only(smem, cv, ncur, ncell, A.cur_kps + co, A.cur_desc + co * 32,
                                      A.cur_cell_start + (size_t)b * (ncell + 1), A.cur_cell_items + co);
  __syncthreads();
  const int q = blockIdx.x * M2_CAND_THREADS + tid;
  if (q >= nmp) return;
  int n = 0;
  if (A.mp_track[mo + q]) {
    const int lvl = A.mp_level[mo + q];
    uint32_t d[8];
    const uint4 *dq = reinterpret_cast<const uint4 *>(A.mp_desc + (mo + q) * 32);
    const uint4 d0 = dq[0], d1 = dq[1];
    d[0] = d0.x; d[1] = d0.y; d[2] = d0.z; d[3] = d0.w; d[4] = d1.x; d[5] = d1.y; d[6] = d1.z; d[7] = d1.w;
    uint32_t *out = cand + (mo + q) * M2_K;
    for_area<false>(A.grid, T, A.mp_proj[(mo + q) * 2], A.mp_proj[(mo + q) * 2 + 1], m2_radius(A, mo + q, A.th != 1.0f, lvl), lvl - 1, lvl,
                    [&](int idx) {
      if (n < M2_K) {
        const int dist = target_hamming(T, d, idx);
        out[n] = ((uint32_t)dist << 20) | ((uint32_t)T.oct[idx] << 16) | (uint32_t)idx;
      }
      n++;
    });
    if (n > M2_K) n = M2_OVER;
  }
  ncand[mo + q] = (uint8_t)n;
}

__global__ __launch_bounds__(MATCH_THREADS) void k_m2_resolve(fb_proj_points_args A, const uint32_t *cand, const uint8_t *ncand) {
  extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
  const int b = blockIdx.x, tid = threadIdx.x, nt = blockDim.x;
  const int ncell = A.grid.cols * A.grid.rows;
  const size_t co = (size_t)b * A.cur_stride, mo = (size_t)b * A.mp_stride;
  const int ncur = A.n_cur[b], nmp = A.n_mp[b];
  const Carve cv(A.cur_stride, ncell, false);
  const TargetLds T = stage_grid_only(smem, cv, ncur, ncell, A.cur_kps + co, A.cur_desc + co * 32,
                                      A.cur_cell_start + (size_t)b * (ncell + 1), A.cur_cell_items + co);
  int *ownerA = reinterpret_cast<int *>(smem + cv.end);
  int *ownerB = ownerA + A.cur_stride;
  int *assignA = ownerB + A.cur_stride;
  int *assignB = assignA + A.mp_stride;
  __shared__ int s_changed, s_n;
  const uint8_t *blocked0 = A.cur_blocked ? A.cur_blocked + co : nullptr;
  for (int i = tid; i < ncur; i += nt) ownerA[i] = (blocked0 && blocked0[i]) ? -1 : NONE;
  for (int q = tid; q < nmp; q += nt) assignA[q] = NONE;
  const bool bFactor = A.th != 1.0f;
  __syncthreads();
  for (int round = 0; round <= nmp + 1; round++) {
    for (int i = tid; i < ncur; i += nt) ownerB[i] = (blocked0 && blocked0[i]) ? -1 : NONE;
    if (tid == 0) s_changed = 0;
    __syncthreads();
    for (int q = tid; q < nmp; q += nt) {
      int best = NONE;
      const int nc = ncand[mo + q];
      if (nc > 0) {
        int bestDist = 256, bestLevel = -1, bestDist2 = 256, bestLevel2 = -1, bestIdx = -1;
        auto take = [&](int idx, int dist, int lev) {
          if (dist < bestDist) { bestDist2 = bestDist; bestDist = dist; bestLevel2 = bestLevel; bestLevel = lev; bestIdx = idx; }
          else if (dist < bestDist2) { bestLevel2 = lev; bestDist2 = dist; }
        };
        if (nc != M2_OVER) {
          const uint4 *cq = reinterpret_cast<const uint4 *>(cand + (mo + q) * M2_K);
          const uint4 c0 = cq[0];
          uint32_t c[M2_K] = {c0.x, c0.y, c0.z, c0.w, 0, 0, 0, 0};
          if (nc > 4) { const uint4 c1 = cq[1]; c[4] = c1.x; c[5] = c1.y; c[6] = c1.z; c[7] = c1.w; }
#pragma unroll
          for (int k = 0; k < M2_K; k++) {
            if (k >= nc) break;
            const int idx = (int)(c[k] & 0xFFFFu);
            if (ownerA[idx] < q) continue;
            take(idx, (int)(c[k] >> 20), (int)((c[k] >> 16) & 0xF));
          }
        } else {  // more candidates than the cache holds: the full walk, descriptors from HBM / L2
          const int lvl = A.mp_level[mo + q];
          uint32_t d[8];
          const uint4 *dq = reinterpret_cast<const uint4 *>(A.mp_desc + (mo + q) * 32);
          const uint4 d0 = dq[0], d1 = dq[1];
          d[0] = d0.x; d[1] = d0.y; d[2] = d0.z; d[3] = d0.w; d[4] = d1.x; d[5] = d1.y; d[6] = d1.z; d[7] = d1.w;
          for_area<false>(A.grid, T, A.mp_proj[(mo + q) * 2], A.mp_proj[(mo + q) * 2 + 1], m2_radius(A, mo + q, bFactor, lvl), lvl - 1, lvl,
                          [&](int idx) {
            if (ownerA[idx] < q) return;
            take(idx, target_hamming(T, d, idx), (int)T.oct[idx]);
          });
        }
        if (bestDist <= TH_HIGH && !(bestLevel == bestLevel2 && bestDist > A.matcher.nnratio * bestDist2)) best = bestIdx;
      }
      assignB[q] = best;
      if (best != assignA[q]) s_changed = 1;
      if (best != NONE && A.mp_obs_pos[mo + q]) atomicMin(&ownerB[best], q);
    }
    __syncthreads();
    const int changed = s_changed;
    int *t = ownerA; ownerA = ownerB; ownerB = t;
    t = assignA; assignA = assignB; assignB = t;
    __syncthreads();
    if (!changed) break;
  }
  int *matchL = ownerB;
  for (int i = tid; i < ncur; i += nt) matchL[i] = -1;
  if (tid == 0) s_n = 0;
  __syncthreads();
  for (int q = tid; q < nmp; q += nt) {
    const int c = assignA[q];
    if (c == NONE) continue;
    atomicMax(&matchL[c], q);
    atomicAdd(&s_n, 1);
  }
  __syncthreads();
  for (int i = tid; i < ncur; i += nt) A.match_cur_to_mp[co + i] = matchL[i];
  if (tid == 0) A.nmatches[b] = s_n;
}

// ---------------------------------------------------------------------------------------
// M9  BirdMapPointMatch
// ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(MATCH_THREADS) void k_bird_mappoints(fb_bird_mp_args A, int descInLds) {
  extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
  const int b = blockIdx.x, tid = threadIdx.x, nt = blockDim.x;
  const int ncell = A.grid.cols * A.grid.rows;
  const size_t co = (size_t)b * A.cur_stride, ro = (size_t)b * A.ref_stride;
  const int ncur = A.n_cur[b], nref = A.n_ref[b];
  const Carve cv(A.cur_stride, ncell, descInLds != 0);
  const TargetLds T = stage_target(smem, cv, ncur, ncell, A.cur_kps + co, A.cur_desc + co * 32,
                                   A.cur_cell_start + (size_t)b * (ncell + 1), A.cur_cell_items + co, descInLds != 0);
  int *writer = reinterpret_cast<int *>(smem + cv.end);  // [cur_stride]
  __shared__ float s_Tcw[12], s_Tbw[12];
  __shared__ int s_n;
  if (tid < 12) s_Tcw[tid] = A.cur_Tcw[(size_t)b * 12 + tid];
  for (int i = tid; i < ncur; i += nt) writer[i] = -1;
  if (tid == 0) s_n = 0;
  __syncthreads();
  if (tid < 12) {  // Tbw = Frame::Tbc * CurF.mTcw, ORBmatcher.cc:1784
    const int r = tid / 4, c = tid % 4;
    float s = (A.Tbc[r * 4 + 0] * s_Tcw[0 * 4 + c] + A.Tbc[r * 4 + 1] * s_Tcw[1 * 4 + c]) + A.Tbc[r * 4 + 2] * s_Tcw[2 * 4 + c];
    if (c == 3) s = s + A.Tbc[r * 4 + 3];
    s_Tbw[tid] = s;
  }
  __syncthreads();
  for (int i1 = tid; i1 < nref; i1 += nt) {
    if (!A.ref_valid[ro + i1]) continue;
    const float X[3] = {A.ref_xw[(ro + i1) * 3], A.ref_xw[(ro + i1) * 3 + 1], A.ref_xw[(ro + i1) * 3 + 2]};
    float lp[3];
    xform(s_Tbw, X, lp);
    if (fabsf(lp[2]) > 0.2) continue;
    // Converter::BaseXY2BirdPixel, Converter.cc:304-310
    const float ptx = (float)(A.bird_cols / 2 - lp[1] * A.meter2pixel);
    const float pty = (float)(A.bird_rows / 2 - (lp[0] - A.rear_axle_to_center) * A.meter2pixel);
    if (ptx < 0 || ptx >= A.bird_cols || pty < 0 || pty >= A.bird_rows) continue;
    uint32_t d[8];
    const uint4 *dq = reinterpret_cast<const uint4 *>(A.ref_desc + (ro + i1) * 32);
    const uint4 d0 = dq[0], d1 = dq[1];
    d[0] = d0.x; d[1] = d0.y; d[2] = d0.z; d[3] = d0.w; d[4] = d1.x; d[5] = d1.y; d[6] = d1.z; d[7] = d1.w;
    int bestDist = INT_MAX, bestDist2 = INT_MAX, bestIdx = -1;
    for_area<true>(A.grid, T, ptx, pty, (float)A.window_size, -1, -1, [&](int i2) {
      if (i2 >= ncur) return;
      const int dist = target_hamming(T, d, i2);
      if (dist < bestDist) { bestDist2 = bestDist; bestDist = dist; bestIdx = i2; }
      else if (dist < bestDist2) bestDist2 = dist;
    });
    if (bestDist > TH_LOW) continue;
    if (!(bestDist < (float)bestDist2 * A.matcher.nnratio)) continue;
    if (!(bestIdx > 0)) continue;  // sic: vnMatches12[i1] > 0, ORBmatcher.cc:1871
    float pc[3];
    xform(s_Tcw, X, pc);
    const float *qv = A.cur_cam_xyz + (co + bestIdx) * 3;
    const float e0 = pc[0] - qv[0], e1 = pc[1] - qv[1], e2 = pc[2] - qv[2];
    const double disC = sqrt((double)e0 * e0 + (double)e1 * e1 + (double)e2 * e2);
    if (disC < A.filter_size) {
      atomicMax(&writer[bestIdx], i1);  // later i1 overwrites earlier
      atomicAdd(&s_n, 1);
    }
  }
  __syncthreads();
  for (int i = tid; i < ncur; i += nt)
    if (writer[i] >= 0) A.match_cur_to_ref[co + i] = writer[i];
  if (tid == 0) A.ninliers[b] = s_n;
}

// ---------------------------------------------------------------------------------------
// M8  BirdviewMatch, isProject = 0
// ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(MATCH_THREADS) void k_birdview(fb_birdview_args A, int descInLds) {
  extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
  const int b = blockIdx.x, tid = threadIdx.x, nt = blockDim.x;
  const int ncell = A.grid.cols * A.grid.rows;
  const size_t co = (size_t)b * A.cur_stride, ro = (size_t)b * A.ref_stride;
  const int ncur = A.n_cur[b], nref = A.n_ref[b];
  const Carve cv(A.cur_stride, ncell, descInLds != 0);
  const TargetLds T = stage_target(smem, cv, ncur, ncell, A.cur_kps + co, A.cur_desc + co * 32,
                                   A.cur_cell_start + (size_t)b * (ncell + 1), A.cur_cell_items + co, descInLds != 0);
  int *m12 = reinterpret_cast<int *>(smem + cv.end);  // [ref_stride]
  int *bins = m12 + A.ref_stride;                      // [ref_stride] histogram bin of i1 or -1
  __shared__ int s_n, s_nd, s_hist[HISTO_LENGTH], s_ind[3];
  if (tid < HISTO_LENGTH) s_hist[tid] = 0;
  if (tid == 0) { s_n = 0; s_nd = 0; }
  __syncthreads();
  const bool ori = A.matcher.check_orientation != 0;
  for (int i1 = tid; i1 < nref; i1 += nt) {
    int m = -1, md = INT_MAX, bin = -1;
    const fb_keypoint kp1 = A.ref_kps[ro + i1];
    if (!(kp1.octave > 0)) {
      uint32_t d[8];
      const uint4 *dq = reinterpret_cast<const uint4 *>(A.ref_desc + (ro + i1) * 32);
      const uint4 d0 = dq[0], d1 = dq[1];
      d[0] = d0.x; d[1] = d0.y; d[2] = d0.z; d[3] = d0.w; d[4] = d1.x; d[5] = d1.y; d[6] = d1.z; d[7] = d1.w;
      int bestDist = INT_MAX, bestDist2 = INT_MAX, bestIdx = -1;
      for_area<true>(A.grid, T, kp1.x, kp1.y, (float)A.window_size, kp1.octave, kp1.octave, [&](int i2) {
        if (i2 >= ncur) return;
        const int dist = target_hamming(T, d, i2);
        if (dist < bestDist) { bestDist2 = bestDist; bestDist = dist; bestIdx = i2; }
        else if (dist < bestDist2) bestDist2 = dist;
      });
      if (bestDist <= TH_LOW) {
        if (bestDist < (float)bestDist2 * A.matcher.nnratio) { m = bestIdx; md = bestDist; atomicAdd(&s_n, 1); }
        if (ori) {  // pushed even when the ratio test failed, ORBmatcher.cc:1712-1722
          bin = rot_bin(kp1.angle - A.cur_kps[co + bestIdx].angle);
          atomicAdd(&s_hist[bin], 1);
        }
      }
    }
    m12[i1] = m;
    bins[i1] = bin;
    A.match_dist[ro + i1] = md;
  }
  __syncthreads();
  if (ori) {
    if (tid == 0) three_maxima(s_hist, s_ind[0], s_ind[1], s_ind[2]);
    __syncthreads();
    for (int i1 = tid; i1 < nref; i1 += nt) {
      const int bin = bins[i1];
      if (bin < 0 || bin == s_ind[0] || bin == s_ind[1] || bin == s_ind[2]) continue;
      if (m12[i1] >= 0) { m12[i1] = -1; atomicSub(&s_n, 1); }
    }
    __syncthreads();
  }
  for (int i1 = tid; i1 < nref; i1 += nt) {
    const int m = m12[i1];
    A.match_ref_to_cur[ro + i1] = m;
    if (m > 0) atomicAdd(&s_nd, 1);  // sic: > 0, ORBmatcher.cc:1755
  }
  __syncthreads();
  if (tid == 0) { A.nmatches[b] = s_n; A.n_dmatches[b] = s_nd; }
}

// ---------------------------------------------------------------------------------------
// DescriptorDistance over n row pairs; one lane per pair, 2 x 16-byte loads per row.
// ---------------------------------------------------------------------------------------
__global__ void k_descriptor_distance(const uint4 *__restrict__ a, const uint4 *__restrict__ b, int n,
                                      int32_t *__restrict__ out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const uint4 a0 = a[2 * i], a1 = a[2 * i + 1], b0 = b[2 * i], b1 = b[2 * i + 1];
  out[i] = __popc(a0.x ^ b0.x) + __popc(a0.y ^ b0.y) + __popc(a0.z ^ b0.z) + __popc(a0.w ^ b0.w) +
           __popc(a1.x ^ b1.x) + __popc(a1.y ^ b1.y) + __popc(a1.z ^ b1.z) + __popc(a1.w ^ b1.w);
}

// ---------------------------------------------------------------------------------------
// AssignFeaturesToGrid: one workgroup per frame. count -> scan -> scatter -> per-cell sort
// (cells hold <1 keypoint on average; the sort restores ascending keypoint index, which is
// the push_back order of Frame.cc:389-396).
// ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_grid_build(const fb_keypoint *__restrict__ kps, const int32_t *__restrict__ n,
                                                    int kp_stride, fb_grid_geom g, int32_t *__restrict__ cell_start,
                                                    int32_t *__restrict__ cell_items, int itemsInLds) {
  extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
  const int b = blockIdx.x, tid = threadIdx.x, nt = blockDim.x;
  const int ncell = g.cols * g.rows;
  int *cnt = reinterpret_cast<int *>(smem);  // [ncell+1]
  int *fillp = cnt + ncell + 1;              // [ncell]
  // the items are scattered, sorted per cell and written out from LDS (itemsInLds): the per-cell insertion sort on the
  // global array was a chain of dependent global accesses per cell, 12 cells per lane -- most of this kernel at batch 1
  int *li = fillp + ncell;                   // [kp_stride] when itemsInLds
  __shared__ int s_part[256];
  const fb_keypoint *k = kps + (size_t)b * kp_stride;
  int32_t *cs = cell_start + (size_t)b * (ncell + 1);
  int32_t *cig = cell_items + (size_t)b * kp_stride;
  int *ci = itemsInLds ? li : cig;
  const int nk = min(max(n[b], 0), kp_stride);
  for (int i = tid; i <= ncell; i += nt) cnt[i] = 0;
  __syncthreads();
  auto cell_of = [&](const fb_keypoint &kp) -> int {
    const int posX = (int)roundf((kp.x - g.min_x) * g.inv_w);  // PosInGrid, Frame.cc:548-558
    const int posY = (int)roundf((kp.y - g.min_y) * g.inv_h);
    if (posX < 0 || posX >= g.cols || posY < 0 || posY >= g.rows) return -1;
    return posX * g.rows + posY;
  };
  for (int i = tid; i < nk; i += nt) {
    const int c = cell_of(k[i]);
    if (c >= 0) atomicAdd(&cnt[c], 1);
  }
  __syncthreads();
  // exclusive scan over ncell counters: per-thread chunk sums, then scan of 256 partials
  const int chunk = (ncell + nt - 1) / nt;
  const int c0 = min(tid * chunk, ncell), c1 = min(c0 + chunk, ncell);
  int s = 0;
  for (int c = c0; c < c1; c++) s += cnt[c];
  // exclusive scan of the 256 chunk sums: shuffles inside a wave, four wave totals through LDS (one lane walking the 256
  // partials in LDS was half of this kernel at batch 1)
  int run;
  {
    const int lane = tid & 63, wv = tid >> 6;
    int inc = s;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { const int t = __shfl_up(inc, o, 64); if (lane >= o) inc += t; }
    if (lane == 63) s_part[wv] = inc;
    __syncthreads();
    int base = 0, tot = 0;
    for (int w = 0; w < (nt >> 6); w++) { const int x = s_part[w]; if (w < wv) base += x; tot += x; }
    run = base + inc - s;
    if (tid == 0) cnt[ncell] = tot;
    __syncthreads();
  }
  for (int c = c0; c < c1; c++) { const int v = cnt[c]; cnt[c] = run; fillp[c] = run; run += v; }
  __syncthreads();
  for (int i = tid; i < nk; i += nt) {
    const int c = cell_of(k[i]);
    if (c >= 0) ci[atomicAdd(&fillp[c], 1)] = i;
  }
  __syncthreads();
  __threadfence_block();
  for (int c = tid; c < ncell; c += nt) {  // insertion sort inside each cell
    const int a0 = cnt[c], a1 = cnt[c + 1];
    for (int i = a0 + 1; i < a1; i++) {
      const int v = ci[i];
      int j = i - 1;
      while (j >= a0 && ci[j] > v) { ci[j + 1] = ci[j]; j--; }
      ci[j + 1] = v;
    }
  }
  for (int i = tid; i <= ncell; i += nt) cs[i] = cnt[i];
  if (itemsInLds) {
    __syncthreads();
    const int nitems = cnt[ncell];
    for (int i = tid; i < nitems; i += nt) cig[i] = li[i];
  }
}

// Frame.cc:365-373: BirdPixel2BaseXY (Converter.cc:284-292) then BaseXY2CamXYZ (:312-318)
struct BirdCamK {
  int cols, rows;
  double pixel2meter, rear;
  float Tcb[12];
};
__global__ void k_bird_keys_to_cam(const fb_keypoint *__restrict__ kps, const int32_t *__restrict__ n, int kp_stride,
                                   BirdCamK K, float *__restrict__ cam) {
  const int b = blockIdx.y;
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n[b]) return;
  const fb_keypoint kp = kps[(size_t)b * kp_stride + i];
  float p[3];
  p[0] = (float)((K.rows / 2 - kp.y) * K.pixel2meter + K.rear);
  p[1] = (float)((K.cols / 2 - kp.x) * K.pixel2meter);
  p[2] = 0.f;
  float o[3];
  xform(K.Tcb, p, o);
  float *dst = cam + ((size_t)b * kp_stride + i) * 3;
  dst[0] = o[0]; dst[1] = o[1]; dst[2] = o[2];
}

size_t match_lds_bytes(int cur_stride, int ncell, int extra_ints, bool withDesc = true, bool withOct = true) {
  return Carve(cur_stride, ncell, withDesc, withOct).end + (size_t)extra_ints * 4;
}

// LDS plan of a matcher: the target frame's descriptor table goes to LDS when everything fits, otherwise it stays in
// HBM/L2 (frames with more key points than ~2900, e.g. nFeatures = 4000) and only positions, octaves and the grid are staged
struct LdsPlan { size_t bytes; int descInLds; };
constexpr size_t LDS_BUDGET = 160 * 1024 - 512;  // leave room for the kernels' static __shared__ variables
int check_lds(size_t bytes, const char *what);
int plan_lds(int cur_stride, int ncell, int extra_ints, const char *what, LdsPlan *p) {
  p->descInLds = 1;
  p->bytes = Carve(cur_stride, ncell, true).end + (size_t)extra_ints * 4;
  if (p->bytes > LDS_BUDGET) {
    p->descInLds = 0;
    p->bytes = Carve(cur_stride, ncell, false).end + (size_t)extra_ints * 4;
  }
  return check_lds(p->bytes, what);
}

int check_lds(size_t bytes, const char *what) {
  if (bytes > 160 * 1024) {
    fb::set_error("%s: frame too large for the LDS-staged matcher (%zu B > 160 KiB)", what, bytes);
    return FB_ERR_CAPACITY;
  }
  return FB_OK;
}

template <typename K>
int set_max_lds(K kernel, size_t bytes) {
  FB_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes));
  return FB_OK;
}

}  // namespace

extern "C" {

int fb_descriptor_distance_dev(const uint8_t *d_a, const uint8_t *d_b, int n, int32_t *d_out, void *stream) {
  FB_TRY(fb::check_device());
  FB_ARG(n >= 0 && (n == 0 || (d_a && d_b && d_out)));
  if (n == 0) return FB_OK;
  fb::ProfScope prof_(fb::P_HAMMING, fb::as_stream(stream));
  k_descriptor_distance<<<(n + 255) / 256, 256, 0, fb::as_stream(stream)>>>(
      reinterpret_cast<const uint4 *>(d_a), reinterpret_cast<const uint4 *>(d_b), n, d_out);
  FB_HIP(hipGetLastError());
  return FB_OK;
}

int fb_descriptor_distance(const uint8_t *a, const uint8_t *b, int n, int32_t *out) {
  FB_TRY(fb::check_device());
  FB_ARG(n >= 0);
  fb::DevBuf da, db, dout;
  FB_TRY(da.upload(a, (size_t)n * 32));
  FB_TRY(db.upload(b, (size_t)n * 32));
  FB_TRY(dout.alloc((size_t)n * 4));
  FB_TRY(fb_descriptor_distance_dev(da.as<uint8_t>(), db.as<uint8_t>(), n, dout.as<int32_t>(), nullptr));
  FB_HIP(hipDeviceSynchronize());
  return dout.download(out, (size_t)n * 4);
}

int fb_grid_build_batch_dev(const fb_keypoint *d_keypoints, const int32_t *d_n, int batch, int kp_stride,
                            const fb_grid_geom *geom, int32_t *d_cell_start, int32_t *d_cell_items, void *stream) {
  FB_TRY(fb::check_device());
  FB_ARG(geom && batch >= 0 && kp_stride > 0 && geom->cols > 0 && geom->rows > 0);
  if (batch == 0) return FB_OK;
  const int ncell = geom->cols * geom->rows;
  size_t lds = (size_t)(2 * ncell + 1) * 4 + (size_t)kp_stride * 4;
  int itemsInLds = 1;
  if (lds > LDS_BUDGET) { itemsInLds = 0; lds = (size_t)(2 * ncell + 1) * 4; }
  FB_TRY(check_lds(lds, "fb_grid_build_batch_dev"));
  FB_TRY(set_max_lds(k_grid_build, lds));
  fb::ProfScope prof_(fb::P_GRID, fb::as_stream(stream));
  k_grid_build<<<batch, 256, lds, fb::as_stream(stream)>>>(d_keypoints, d_n, kp_stride, *geom, d_cell_start, d_cell_items, itemsInLds);
  FB_HIP(hipGetLastError());
  return FB_OK;
}

int fb_bird_keys_to_cam_dev(const fb_keypoint *d_kps, const int32_t *d_n, int batch, int kp_stride, int bird_cols,
                            int bird_rows, double pixel2meter, double rear_axle_to_center, const float *Tcb12,
                            float *d_cam_xyz, void *stream) {
  FB_TRY(fb::check_device());
  FB_ARG(Tcb12 && batch >= 0 && kp_stride > 0);
  if (batch == 0) return FB_OK;
  BirdCamK K;
  K.cols = bird_cols; K.rows = bird_rows; K.pixel2meter = pixel2meter; K.rear = rear_axle_to_center;
  memcpy(K.Tcb, Tcb12, sizeof(K.Tcb));
  dim3 grid((kp_stride + 255) / 256, batch);
  fb::ProfScope prof_(fb::P_BIRDCAM, fb::as_stream(stream));
  k_bird_keys_to_cam<<<grid, 256, 0, fb::as_stream(stream)>>>(d_kps, d_n, kp_stride, K, d_cam_xyz);
  FB_HIP(hipGetLastError());
  return FB_OK;
}

int fb_match_projection_frame_dev(const fb_proj_frame_args *A, void *stream) {
  FB_TRY(fb::check_device());
  FB_ARG(A && A->batch >= 0 && A->cur_stride > 0 && A->last_stride >= 0 && A->cur_stride < 65536);
  if (A->batch == 0) return FB_OK;
  const int ncell = A->grid.cols * A->grid.rows;
  // ints: two owner arrays; per query two u16 assignments + u16 processing order + one meta byte (2 ints) + the cache
  const int base_ints = 2 * A->cur_stride + 2 * A->last_stride + 4;
  // preference: descriptors in LDS + cache, descriptors in LDS, cache only, neither
  static const bool noCache = getenv("FB_M3_NO_CACHE") != nullptr;  // measurements: every round walks the grid again
  int cacheK = noCache ? 0 : CACHE_K, descInLds = 1;
  size_t lds = match_lds_bytes(A->cur_stride, ncell, base_ints + cacheK * A->last_stride, true, false);
  if (lds > LDS_BUDGET) { cacheK = 0; lds = match_lds_bytes(A->cur_stride, ncell, base_ints, true, false); }
  if (lds > LDS_BUDGET) { cacheK = CACHE_K; descInLds = 0; lds = match_lds_bytes(A->cur_stride, ncell, base_ints + cacheK * A->last_stride, false, false); }
  if (lds > LDS_BUDGET) { cacheK = 0; lds = match_lds_bytes(A->cur_stride, ncell, base_ints, false, false); }
  FB_TRY(check_lds(lds, "fb_match_projection_frame"));
  FB_TRY(set_max_lds(k_proj_frame, lds));
  fb::ProfScope prof_(fb::P_PROJ_FRAME, fb::as_stream(stream));
  k_proj_frame<<<A->batch, MATCH_THREADS, lds, fb::as_stream(stream)>>>(*A, cacheK, descInLds);
  FB_HIP(hipGetLastError());
  return FB_OK;
}

int fb_match_projection_keyframe_dev(const fb_proj_kf_args *A, void *stream) {
  FB_TRY(fb::check_device());
  FB_ARG(A && A->batch >= 0 && A->cur_stride > 0 && A->kf_stride >= 0 && A->cur_stride < 65536);
  FB_ARG(A->n_levels > 0 && A->n_levels <= FB_MAX_LEVELS);
  if (A->batch == 0) return FB_OK;
  const int ncell = A->grid.cols * A->grid.rows;
  LdsPlan lp;
  FB_TRY(plan_lds(A->cur_stride, ncell, 2 * A->cur_stride + A->kf_stride + 2, "fb_match_projection_keyframe", &lp));
  const size_t lds = lp.bytes;
  FB_TRY(set_max_lds(k_proj_kf, lds));
  fb::ProfScope prof_(fb::P_PROJ_KF, fb::as_stream(stream));
  k_proj_kf<<<A->batch, MATCH_THREADS, lds, fb::as_stream(stream)>>>(*A, lp.descInLds);
  FB_HIP(hipGetLastError());
  return FB_OK;
}

#ifdef FB_MATCH_STAMPS
int fb_match_debug_m3(unsigned long long *dst8) {  // probe build only
  unsigned long long z[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  FB_HIP(hipMemcpyFromSymbol(dst8, HIP_SYMBOL(g_m3_stamps), sizeof(z)));
  FB_HIP(hipMemcpyToSymbol(HIP_SYMBOL(g_m3_stamps), z, sizeof(z)));
  return FB_OK;
}
int fb_match_debug_m2(int *dst4) {  // probe build only
  int z[4] = {0, 0, 0, 0};
  FB_HIP(hipMemcpyFromSymbol(dst4, HIP_SYMBOL(g_m2_rounds), sizeof(z)));
  FB_HIP(hipMemcpyToSymbol(HIP_SYMBOL(g_m2_rounds), z, sizeof(z)));
  return FB_OK;
}
#endif

int fb_match_projection_points_dev(const fb_proj_points_args *A, void *stream) {
  FB_TRY(fb::check_device());
  FB_ARG(A && A->batch >= 0 && A->cur_stride > 0 && A->mp_stride >= 0 && A->cur_stride < 65536);
  if (A->batch == 0) return FB_OK;
  const int ncell = A->grid.cols * A->grid.rows;
  const size_t wsNeed = fb_match_projection_points_workspace(A->batch, A->mp_stride);
  if (A->workspace && A->workspace_bytes >= wsNeed && A->mp_stride > 0 && A->batch <= 65535 && ((uintptr_t)A->workspace % 16) == 0) {
    // two phases: candidate lists by many workgroups, then the serial rule on the lists (see k_m2_candidates)
    uint32_t *cand = reinterpret_cast<uint32_t *>(A->workspace);
    uint8_t *ncand = reinterpret_cast<uint8_t *>(cand + (size_t)A->batch * A->mp_stride * M2_K);
    const size_t ldsA = Carve(A->cur_stride, ncell, false).end;
    const size_t ldsB = ldsA + ((size_t)2 * A->cur_stride + (size_t)2 * A->mp_stride) * 4;
    if (ldsB <= LDS_BUDGET) {
      FB_TRY(set_max_lds(k_m2_candidates, ldsA));
      FB_TRY(set_max_lds(k_m2_resolve, ldsB));
      fb::ProfScope prof_(fb::P_PROJ_POINTS, fb::as_stream(stream));
      k_m2_candidates<<<dim3((A->mp_stride + M2_CAND_THREADS - 1) / M2_CAND_THREADS, A->batch), M2_CAND_THREADS, ldsA, fb::as_stream(stream)>>>(*A, cand, ncand);
      k_m2_resolve<<<A->batch, MATCH_THREADS, ldsB, fb::as_stream(stream)>>>(*A, cand, ncand);
      FB_HIP(hipGetLastError());
      return FB_OK;
    }
  }
  LdsPlan lp;
  FB_TRY(plan_lds(A->cur_stride, ncell, 2 * A->cur_stride + 2 * A->mp_stride, "fb_match_projection_points", &lp));
  const size_t lds = lp.bytes;
  FB_TRY(set_max_lds(k_proj_points, lds));
  fb::ProfScope prof_(fb::P_PROJ_POINTS, fb::as_stream(stream));
  k_proj_points<<<A->batch, MATCH_THREADS, lds, fb::as_stream(stream)>>>(*A, lp.descInLds);
  FB_HIP(hipGetLastError());
  return FB_OK;
}

size_t fb_match_projection_points_workspace(int batch, int mp_stride) {
  if (batch <= 0 || mp_stride <= 0) return 0;
  return (size_t)batch * mp_stride * (M2_K * 4 + 1) + 16;
}

int fb_match_bird_mappoints_dev(const fb_bird_mp_args *A, void *stream) {
  FB_TRY(fb::check_device());
  FB_ARG(A && A->batch >= 0 && A->cur_stride > 0 && A->ref_stride >= 0 && A->cur_stride < 65536);
  if (A->batch == 0) return FB_OK;
  const int ncell = A->grid.cols * A->grid.rows;
  LdsPlan lp;
  FB_TRY(plan_lds(A->cur_stride, ncell, A->cur_stride, "fb_match_bird_mappoints", &lp));
  const size_t lds = lp.bytes;
  FB_TRY(set_max_lds(k_bird_mappoints, lds));
  fb::ProfScope prof_(fb::P_BIRD_MP, fb::as_stream(stream));
  k_bird_mappoints<<<A->batch, MATCH_THREADS, lds, fb::as_stream(stream)>>>(*A, lp.descInLds);
  FB_HIP(hipGetLastError());
  return FB_OK;
}

int fb_match_birdview_dev(const fb_birdview_args *A, void *stream) {
  FB_TRY(fb::check_device());
  FB_ARG(A && A->batch >= 0 && A->cur_stride > 0 && A->ref_stride >= 0 && A->cur_stride < 65536);
  if (A->batch == 0) return FB_OK;
  const int ncell = A->grid.cols * A->grid.rows;
  LdsPlan lp;
  FB_TRY(plan_lds(A->cur_stride, ncell, 2 * A->ref_stride, "fb_match_birdview", &lp));
  const size_t lds = lp.bytes;
  FB_TRY(set_max_lds(k_birdview, lds));
  fb::ProfScope prof_(fb::P_BIRDVIEW, fb::as_stream(stream));
  k_birdview<<<A->batch, MATCH_THREADS, lds, fb::as_stream(stream)>>>(*A, lp.descInLds);
  FB_HIP(hipGetLastError());
  return FB_OK;
}

// ---- host-pointer drop-ins: ONE staged upload, the same kernels, one staged download (fb::Stager) --------------
#define UP(buf, field, bytes) st.in((void **)&D.field, H->field, (bytes));
#define OUT(field, bytes, copy_in) st.out((void **)&D.field, H->field, (bytes), (copy_in));

int fb_match_projection_frame(const fb_proj_frame_args *H) {
  FB_TRY(fb::check_device());
  FB_ARG(H && H->batch >= 0);
  fb_proj_frame_args D = *H;
  const size_t B = H->batch, cs = H->cur_stride, ls = H->last_stride, ncell = (size_t)H->grid.cols * H->grid.rows;
  fb::Stager st;
  UP(b0, n_cur, B * 4) UP(b1, cur_kps, B * cs * sizeof(fb_keypoint)) UP(b2, cur_desc, B * cs * 32)
  UP(b3, cur_cell_start, B * (ncell + 1) * 4) UP(b4, cur_cell_items, B * cs * 4) UP(b5, cur_blocked, B * cs)
  UP(b6, cur_Tcw, B * 48) UP(b7, n_last, B * 4) UP(b8, last_valid, B * ls) UP(b9, last_obs_pos, B * ls)
  UP(b10, last_xw, B * ls * 12) UP(b11, last_desc, B * ls * 32) UP(b12, last_octave, B * ls * 4)
  UP(b13, last_angle, B * ls * 4)
  OUT(match_cur_to_last, B * cs * 4, true)  // copy-in: entries past n keep the caller's contents
  OUT(nmatches, B * 4, false)
  if (H->retried) { OUT(retried, B * 4, false) } else D.retried = nullptr;
  FB_ARG(H->match_cur_to_last && H->nmatches);
  FB_TRY(st.commit(nullptr));
  FB_TRY(fb_match_projection_frame_dev(&D, nullptr));
  return st.fetch(nullptr);
}

int fb_match_projection_keyframe(const fb_proj_kf_args *H) {
  FB_TRY(fb::check_device());
  FB_ARG(H && H->batch >= 0);
  fb_proj_kf_args D = *H;
  const size_t B = H->batch, cs = H->cur_stride, ks = H->kf_stride, ncell = (size_t)H->grid.cols * H->grid.rows;
  fb::Stager st;
  UP(b0, n_cur, B * 4) UP(b1, cur_kps, B * cs * sizeof(fb_keypoint)) UP(b2, cur_desc, B * cs * 32)
  UP(b3, cur_cell_start, B * (ncell + 1) * 4) UP(b4, cur_cell_items, B * cs * 4) UP(b5, cur_blocked, B * cs)
  UP(b6, cur_Tcw, B * 48) UP(b7, n_kf, B * 4) UP(b8, kf_valid, B * ks) UP(b9, kf_xw, B * ks * 12)
  UP(b10, kf_desc, B * ks * 32) UP(b11, kf_max_dist, B * ks * 4) UP(b12, kf_min_dist, B * ks * 4)
  UP(b13, kf_angle, B * ks * 4)
  OUT(match_cur_to_kf, B * cs * 4, true)
  OUT(nmatches, B * 4, false)
  FB_ARG(H->match_cur_to_kf && H->nmatches);
  FB_TRY(st.commit(nullptr));
  FB_TRY(fb_match_projection_keyframe_dev(&D, nullptr));
  return st.fetch(nullptr);
}

int fb_match_projection_points(const fb_proj_points_args *H) {
  FB_TRY(fb::check_device());
  FB_ARG(H && H->batch >= 0);
  fb_proj_points_args D = *H;
  const size_t B = H->batch, cs = H->cur_stride, ms = H->mp_stride, ncell = (size_t)H->grid.cols * H->grid.rows;
  fb::Stager st;
  UP(b0, n_cur, B * 4) UP(b1, cur_kps, B * cs * sizeof(fb_keypoint)) UP(b2, cur_desc, B * cs * 32)
  UP(b3, cur_cell_start, B * (ncell + 1) * 4) UP(b4, cur_cell_items, B * cs * 4) UP(b5, cur_blocked, B * cs)
  UP(b6, n_mp, B * 4) UP(b7, mp_track, B * ms) UP(b8, mp_obs_pos, B * ms) UP(b9, mp_proj, B * ms * 8)
  UP(b10, mp_level, B * ms * 4) UP(b11, mp_view_cos, B * ms * 4) UP(b12, mp_desc, B * ms * 32)
  OUT(match_cur_to_mp, B * cs * 4, true)
  OUT(nmatches, B * 4, false)
  FB_ARG(H->match_cur_to_mp && H->nmatches);
  fb::DevBuf ws;  // the two-phase matcher (this call is synchronous, so a pooled block is safe as its workspace)
  static const bool onePhase = getenv("FB_M2_ONE_KERNEL") != nullptr;  // measurements / tests of the one-kernel version
  D.workspace = nullptr; D.workspace_bytes = 0;
  if (!onePhase && B * ms > 0) {
    D.workspace_bytes = fb_match_projection_points_workspace((int)B, (int)ms);
    FB_TRY(ws.alloc(D.workspace_bytes));
    D.workspace = ws.p;
  }
  FB_TRY(st.commit(nullptr));
  FB_TRY(fb_match_projection_points_dev(&D, nullptr));
  return st.fetch(nullptr);
}

int fb_match_bird_mappoints(const fb_bird_mp_args *H) {
  FB_TRY(fb::check_device());
  FB_ARG(H && H->batch >= 0);
  fb_bird_mp_args D = *H;
  const size_t B = H->batch, cs = H->cur_stride, rs = H->ref_stride, ncell = (size_t)H->grid.cols * H->grid.rows;
  fb::Stager st;
  UP(b0, n_cur, B * 4) UP(b1, cur_kps, B * cs * sizeof(fb_keypoint)) UP(b2, cur_desc, B * cs * 32)
  UP(b3, cur_cam_xyz, B * cs * 12) UP(b4, cur_cell_start, B * (ncell + 1) * 4) UP(b5, cur_cell_items, B * cs * 4)
  UP(b6, cur_Tcw, B * 48) UP(b7, n_ref, B * 4) UP(b8, ref_valid, B * rs) UP(b9, ref_xw, B * rs * 12)
  UP(b10, ref_desc, B * rs * 32)
  OUT(match_cur_to_ref, B * cs * 4, true)  // in/out
  OUT(ninliers, B * 4, false)
  FB_ARG(H->match_cur_to_ref && H->ninliers);
  FB_TRY(st.commit(nullptr));
  FB_TRY(fb_match_bird_mappoints_dev(&D, nullptr));
  return st.fetch(nullptr);
}

int fb_match_birdview(const fb_birdview_args *H) {
  FB_TRY(fb::check_device());
  FB_ARG(H && H->batch >= 0);
  fb_birdview_args D = *H;
  const size_t B = H->batch, cs = H->cur_stride, rs = H->ref_stride, ncell = (size_t)H->grid.cols * H->grid.rows;
  fb::Stager st;
  UP(b0, n_cur, B * 4) UP(b1, cur_kps, B * cs * sizeof(fb_keypoint)) UP(b2, cur_desc, B * cs * 32)
  UP(b3, cur_cell_start, B * (ncell + 1) * 4) UP(b4, cur_cell_items, B * cs * 4) UP(b5, n_ref, B * 4)
  UP(b6, ref_kps, B * rs * sizeof(fb_keypoint)) UP(b7, ref_desc, B * rs * 32)
  OUT(match_ref_to_cur, B * rs * 4, true)
  OUT(match_dist, B * rs * 4, true)
  OUT(nmatches, B * 4, false)
  OUT(n_dmatches, B * 4, false)
  FB_ARG(H->match_ref_to_cur && H->match_dist && H->nmatches && H->n_dmatches);
  FB_TRY(st.commit(nullptr));
  FB_TRY(fb_match_birdview_dev(&D, nullptr));
  return st.fetch(nullptr);
}
#undef OUT
#undef UP
// (match_kf.inc's wrappers -- loop closing / initialisation, not per-frame -- keep one pooled buffer per argument)
#define UP(buf, field, bytes)                                        \
  fb::DevBuf buf;                                                    \
  if (H->field) { FB_TRY(buf.upload(H->field, (bytes))); D.field = buf.as<std::remove_pointer<decltype(D.field)>::type>(); }

}  // extern "C"

#include "match_kf.inc"
