/*
 * match_oracle.cpp -- CPU restatement of the ORBmatcher entry points on the hot
 * path (TEST INFRASTRUCTURE ONLY; see orb_oracle.cpp header for who may call it).
 *
 * Follows /root/reference/src/ORBmatcher.cc and the Frame grid helpers in
 * /root/reference/src/Frame.cc, serially, in the reference's loop order.
 * Integer Hamming work is exact; the float gating uses the reference's float
 * expressions evaluated left to right without FMA contraction.  cv::Mat products
 * (Rcw*x3Dw+tcw) are restated as ((r0*X + r1*Y) + r2*Z) + t in float: OpenCV's
 * gemm is not vendored -> "parity unpinned" for that rounding.
 */
#include <climits>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <vector>

#include "../include/fishbird.h"

namespace {

const int TH_HIGH = 100;      // ORBmatcher.cc:38
const int TH_LOW = 50;        // :39
const int HISTO_LENGTH = 30;  // :40

// DescriptorDistance, ORBmatcher.cc:1951-1967 (SWAR popcount over 8 x int32)
int descriptor_distance(const uint8_t *a, const uint8_t *b) {
  int dist = 0;
  for (int i = 0; i < 8; i++) {
    uint32_t pa, pb;
    std::memcpy(&pa, a + 4 * i, 4);
    std::memcpy(&pb, b + 4 * i, 4);
    uint32_t v = pa ^ pb;
    v = v - ((v >> 1) & 0x55555555);
    v = (v & 0x33333333) + ((v >> 2) & 0x33333333);
    dist += (((v + (v >> 4)) & 0xF0F0F0F) * 0x1010101) >> 24;
  }
  return dist;
}

// ComputeThreeMaxima, ORBmatcher.cc:1905-1946 (on bin sizes)
void three_maxima(const int *sz, int L, int &ind1, int &ind2, int &ind3) {
  int max1 = 0, max2 = 0, max3 = 0;
  ind1 = ind2 = ind3 = -1;
  for (int i = 0; i < L; i++) {
    const int s = sz[i];
    if (s > max1) { max3 = max2; max2 = max1; max1 = s; ind3 = ind2; ind2 = ind1; ind1 = i; }
    else if (s > max2) { max3 = max2; max2 = s; ind3 = ind2; ind2 = i; }
    else if (s > max3) { max3 = s; ind3 = i; }
  }
  if (max2 < 0.1f * (float)max1) { ind2 = -1; ind3 = -1; }
  else if (max3 < 0.1f * (float)max1) { ind3 = -1; }
}

int rot_bin(float rot) {  // ORBmatcher.cc:1434-1439
  const float factor = 1.0f / HISTO_LENGTH;
  if (rot < 0.0) rot += 360.0f;
  int bin = (int)std::round(rot * factor);
  if (bin == HISTO_LENGTH) bin = 0;
  return bin;
}

struct GridView {
  const fb_grid_geom *g;
  const int32_t *start;
  const int32_t *items;
  const fb_keypoint *kps;
};

// Frame::GetFeaturesInArea, Frame.cc:493-546 (inclusive cell loops)
void features_in_area(const GridView &G, float x, float y, float r, int minLevel, int maxLevel,
                      std::vector<int> &out) {
  out.clear();
  const fb_grid_geom &g = *G.g;
  const int nMinCellX = std::max(0, (int)std::floor((x - g.min_x - r) * g.inv_w));
  if (nMinCellX >= g.cols) return;
  const int nMaxCellX = std::min(g.cols - 1, (int)std::ceil((x - g.min_x + r) * g.inv_w));
  if (nMaxCellX < 0) return;
  const int nMinCellY = std::max(0, (int)std::floor((y - g.min_y - r) * g.inv_h));
  if (nMinCellY >= g.rows) return;
  const int nMaxCellY = std::min(g.rows - 1, (int)std::ceil((y - g.min_y + r) * g.inv_h));
  if (nMaxCellY < 0) return;
  const bool bCheckLevels = (minLevel > 0) || (maxLevel >= 0);
  for (int ix = nMinCellX; ix <= nMaxCellX; ix++)
    for (int iy = nMinCellY; iy <= nMaxCellY; iy++) {
      const int c = ix * g.rows + iy;
      for (int j = G.start[c]; j < G.start[c + 1]; j++) {
        const int idx = G.items[j];
        const fb_keypoint &kp = G.kps[idx];
        if (bCheckLevels) {
          if (kp.octave < minLevel) continue;
          if (maxLevel >= 0 && kp.octave > maxLevel) continue;
        }
        const float distx = kp.x - x, disty = kp.y - y;
        if (std::fabs(distx) < r && std::fabs(disty) < r) out.push_back(idx);
      }
    }
}

// Frame::GetFeaturesInAreaBirdview, Frame.cc:572-626 (EXCLUSIVE cell loops, no min offset)
void features_in_area_bird(const GridView &G, float x, float y, float r, int minLevel, int maxLevel,
                           std::vector<int> &out) {
  out.clear();
  const fb_grid_geom &g = *G.g;
  const int nMinCellX = std::max(0, (int)std::floor((x - r) * g.inv_w));
  if (nMinCellX >= g.cols) return;
  const int nMaxCellX = std::min(g.cols - 1, (int)std::ceil((x + r) * g.inv_w));
  if (nMaxCellX < 0) return;
  const int nMinCellY = std::max(0, (int)std::floor((y - r) * g.inv_h));
  if (nMinCellY >= g.rows) return;
  const int nMaxCellY = std::min(g.rows - 1, (int)std::ceil((y + r) * g.inv_h));
  if (nMaxCellY < 0) return;
  const bool bCheckLevels = (minLevel > 0) || (maxLevel >= 0);
  for (int ix = nMinCellX; ix < nMaxCellX; ix++)
    for (int iy = nMinCellY; iy < nMaxCellY; iy++) {
      const int c = ix * g.rows + iy;
      for (int j = G.start[c]; j < G.start[c + 1]; j++) {
        const int idx = G.items[j];
        const fb_keypoint &kp = G.kps[idx];
        if (bCheckLevels) {
          if (kp.octave < minLevel) continue;
          if (maxLevel >= 0 && kp.octave > maxLevel) continue;
        }
        const float disx = kp.x - x, disy = kp.y - y;
        if (std::fabs(disx) < r && std::fabs(disy) < r) out.push_back(idx);
      }
    }
}

inline void transform(const float *T, const float *X, float *o) {  // rows 0..2 of a 3x4
  for (int r = 0; r < 3; r++) o[r] = ((T[r * 4 + 0] * X[0] + T[r * 4 + 1] * X[1]) + T[r * 4 + 2] * X[2]) + T[r * 4 + 3];
}

}  // namespace

extern "C" {

int orc_descriptor_distance(const uint8_t *a, const uint8_t *b, int n, int32_t *out) {
  for (int i = 0; i < n; i++) out[i] = descriptor_distance(a + (size_t)i * 32, b + (size_t)i * 32);
  return FB_OK;
}

int orc_three_maxima(const int32_t *sizes, int L, int32_t *ind) {
  int a, b, c;
  three_maxima(sizes, L, a, b, c);
  ind[0] = a; ind[1] = b; ind[2] = c;
  return FB_OK;
}

// Frame::AssignFeaturesToGrid + PosInGrid / PosInGridBirdview, Frame.cc:381-411,548-570
int orc_grid_build(const fb_keypoint *kps, const int32_t *n, int batch, int kp_stride, const fb_grid_geom *g,
                   int32_t *cell_start, int32_t *cell_items) {
  const int ncell = g->cols * g->rows;
  for (int b = 0; b < batch; b++) {
    const fb_keypoint *k = kps + (size_t)b * kp_stride;
    int32_t *cs = cell_start + (size_t)b * (ncell + 1);
    int32_t *ci = cell_items + (size_t)b * kp_stride;
    std::vector<std::vector<int>> cells(ncell);
    for (int i = 0; i < n[b]; i++) {
      const int posX = (int)std::round((k[i].x - g->min_x) * g->inv_w);
      const int posY = (int)std::round((k[i].y - g->min_y) * g->inv_h);
      if (posX < 0 || posX >= g->cols || posY < 0 || posY >= g->rows) continue;
      cells[posX * g->rows + posY].push_back(i);
    }
    int off = 0;
    for (int c = 0; c < ncell; c++) {
      cs[c] = off;
      for (int i : cells[c]) ci[off++] = i;
    }
    cs[ncell] = off;
  }
  return FB_OK;
}

// ORBmatcher::SearchByProjection(Frame&, const Frame&, th, bMono=true), ORBmatcher.cc:1329-1471
int orc_match_projection_frame(const fb_proj_frame_args *A) {
  const int ncell = A->grid.cols * A->grid.rows;
  std::vector<int> cand;
  for (int b = 0; b < A->batch; b++) {
    const size_t co = (size_t)b * A->cur_stride, lo = (size_t)b * A->last_stride;
    const int ncur = A->n_cur[b], nlast = A->n_last[b];
    GridView G{&A->grid, A->cur_cell_start + (size_t)b * (ncell + 1), A->cur_cell_items + co, A->cur_kps + co};
    const float *T = A->cur_Tcw + (size_t)b * 12;
    int32_t *match = A->match_cur_to_last + co;
    std::vector<uint8_t> blocked(ncur, 0);
    for (int i = 0; i < ncur; i++) {
      match[i] = -1;
      if (A->cur_blocked) blocked[i] = A->cur_blocked[co + i];
    }
    std::vector<int> rotHist[HISTO_LENGTH];
    int nmatches = 0;
    for (int i = 0; i < nlast; i++) {
      if (!A->last_valid[lo + i]) continue;
      float xc3[3];
      transform(T, A->last_xw + (lo + i) * 3, xc3);
      const float xc = xc3[0], yc = xc3[1];
      const float invzc = (float)(1.0 / xc3[2]);
      if (invzc < 0) continue;
      const float u = A->cam.fx * xc * invzc + A->cam.cx;
      const float v = A->cam.fy * yc * invzc + A->cam.cy;
      if (u < A->cam.min_x || u > A->cam.max_x) continue;
      if (v < A->cam.min_y || v > A->cam.max_y) continue;
      const int nLastOctave = A->last_octave[lo + i];
      const float radius = A->th * A->scale_factors[nLastOctave];
      features_in_area(G, u, v, radius, nLastOctave - 1, nLastOctave + 1, cand);
      if (cand.empty()) continue;
      const uint8_t *dMP = A->last_desc + (lo + i) * 32;
      int bestDist = 256, bestIdx2 = -1;
      for (int i2 : cand) {
        if (blocked[i2]) continue;  // mvpMapPoints[i2] && Observations()>0
        const int dist = descriptor_distance(dMP, A->cur_desc + (co + i2) * 32);
        if (dist < bestDist) { bestDist = dist; bestIdx2 = i2; }
      }
      if (bestDist <= TH_HIGH) {
        match[bestIdx2] = i;
        blocked[bestIdx2] = A->last_obs_pos[lo + i] ? 1 : 0;
        nmatches++;
        if (A->matcher.check_orientation) {
          float rot = A->last_angle[lo + i] - A->cur_kps[co + bestIdx2].angle;
          rotHist[rot_bin(rot)].push_back(bestIdx2);
        }
      }
    }
    if (A->matcher.check_orientation) {
      int sz[HISTO_LENGTH], ind1, ind2, ind3;
      for (int i = 0; i < HISTO_LENGTH; i++) sz[i] = (int)rotHist[i].size();
      three_maxima(sz, HISTO_LENGTH, ind1, ind2, ind3);
      for (int i = 0; i < HISTO_LENGTH; i++)
        if (i != ind1 && i != ind2 && i != ind3)
          for (int idx : rotHist[i]) { match[idx] = -1; nmatches--; }
    }
    A->nmatches[b] = nmatches;
  }
  return FB_OK;
}

// ORBmatcher::BirdMapPointMatch, ORBmatcher.cc:1763-1902
int orc_match_bird_mappoints(const fb_bird_mp_args *A) {
  const int ncell = A->grid.cols * A->grid.rows;
  std::vector<int> cand;
  for (int b = 0; b < A->batch; b++) {
    const size_t co = (size_t)b * A->cur_stride, ro = (size_t)b * A->ref_stride;
    const int ncur = A->n_cur[b], nref = A->n_ref[b];
    GridView G{&A->grid, A->cur_cell_start + (size_t)b * (ncell + 1), A->cur_cell_items + co, A->cur_kps + co};
    const float *Tcw = A->cur_Tcw + (size_t)b * 12;
    // Tbw = Frame::Tbc * CurF.mTcw (:1784), 4x4 float product restated row by row
    float Tbw[12];
    for (int r = 0; r < 3; r++)
      for (int c = 0; c < 4; c++) {
        float s = (A->Tbc[r * 4 + 0] * Tcw[0 * 4 + c] + A->Tbc[r * 4 + 1] * Tcw[1 * 4 + c]) + A->Tbc[r * 4 + 2] * Tcw[2 * 4 + c];
        if (c == 3) s = s + A->Tbc[r * 4 + 3];
        Tbw[r * 4 + c] = s;
      }
    std::vector<int> vnMatches12(nref, -1), vMatchedDistance(nref, INT_MAX);
    int nmatches = 0;
    for (int i1 = 0; i1 < nref; i1++) {
      if (!A->ref_valid[ro + i1]) continue;
      float lp[3];
      transform(Tbw, A->ref_xw + (ro + i1) * 3, lp);
      if (std::fabs(lp[2]) > 0.2) continue;
      // Converter::BaseXY2BirdPixel, Converter.cc:304-310 (double arithmetic, float result)
      const float ptx = (float)(A->bird_cols / 2 - lp[1] * A->meter2pixel);
      const float pty = (float)(A->bird_rows / 2 - (lp[0] - A->rear_axle_to_center) * A->meter2pixel);
      if (ptx < 0 || ptx >= A->bird_cols || pty < 0 || pty >= A->bird_rows) continue;
      features_in_area_bird(G, ptx, pty, (float)A->window_size, -1, -1, cand);
      if (cand.empty()) continue;
      const uint8_t *d1 = A->ref_desc + (ro + i1) * 32;
      int bestDist = INT_MAX, bestDist2 = INT_MAX, bestIdx = -1;
      for (int i2 : cand) {
        if (i2 >= ncur) continue;
        const int dist = descriptor_distance(d1, A->cur_desc + (co + i2) * 32);
        if (dist < bestDist) { bestDist2 = bestDist; bestDist = dist; bestIdx = i2; }
        else if (dist < bestDist2) bestDist2 = dist;
      }
      if (bestDist <= TH_LOW) {
        if (bestDist < (float)bestDist2 * A->matcher.nnratio) {
          vnMatches12[i1] = bestIdx;
          vMatchedDistance[i1] = bestDist;
          nmatches++;
        }
      }
    }
    (void)nmatches;
    int inliers = 0;
    for (int i1 = 0; i1 < nref; i1++) {
      if (vnMatches12[i1] > 0) {  // sic: index 0 is dropped, :1871
        float pc[3];
        transform(Tcw, A->ref_xw + (ro + i1) * 3, pc);
        const float *q = A->cur_cam_xyz + (co + vnMatches12[i1]) * 3;
        const float d0 = pc[0] - q[0], d1 = pc[1] - q[1], d2 = pc[2] - q[2];
        const double disC = std::sqrt((double)d0 * d0 + (double)d1 * d1 + (double)d2 * d2);  // cv::norm L2
        if (disC < A->filter_size) {
          A->match_cur_to_ref[co + vnMatches12[i1]] = i1;
          inliers++;
        }
      }
    }
    A->ninliers[b] = inliers;
  }
  return FB_OK;
}

// ORBmatcher::SearchByProjection(Frame&, const vector<MapPoint*>&, th), ORBmatcher.cc:46-138
int orc_match_projection_points(const fb_proj_points_args *A) {
  const int ncell = A->grid.cols * A->grid.rows;
  std::vector<int> cand;
  const bool bFactor = A->th != 1.0;
  for (int b = 0; b < A->batch; b++) {
    const size_t co = (size_t)b * A->cur_stride, mo = (size_t)b * A->mp_stride;
    const int ncur = A->n_cur[b], nmp = A->n_mp[b];
    GridView G{&A->grid, A->cur_cell_start + (size_t)b * (ncell + 1), A->cur_cell_items + co, A->cur_kps + co};
    int32_t *match = A->match_cur_to_mp + co;
    std::vector<uint8_t> blocked(ncur, 0);
    for (int i = 0; i < ncur; i++) {
      match[i] = -1;
      if (A->cur_blocked) blocked[i] = A->cur_blocked[co + i];
    }
    int nmatches = 0;
    for (int iMP = 0; iMP < nmp; iMP++) {
      if (!A->mp_track[mo + iMP]) continue;
      const int lvl = A->mp_level[mo + iMP];
      float r = A->mp_view_cos[mo + iMP] > 0.998 ? 2.5f : 4.0f;  // RadiusByViewingCos :132-138
      if (bFactor) r *= A->th;
      features_in_area(G, A->mp_proj[(mo + iMP) * 2], A->mp_proj[(mo + iMP) * 2 + 1], r * A->scale_factors[lvl],
                       lvl - 1, lvl, cand);
      if (cand.empty()) continue;
      const uint8_t *dMP = A->mp_desc + (mo + iMP) * 32;
      int bestDist = 256, bestLevel = -1, bestDist2 = 256, bestLevel2 = -1, bestIdx = -1;
      for (int idx : cand) {
        if (blocked[idx]) continue;
        const int dist = descriptor_distance(dMP, A->cur_desc + (co + idx) * 32);
        if (dist < bestDist) {
          bestDist2 = bestDist; bestDist = dist; bestLevel2 = bestLevel;
          bestLevel = A->cur_kps[co + idx].octave; bestIdx = idx;
        } else if (dist < bestDist2) {
          bestLevel2 = A->cur_kps[co + idx].octave; bestDist2 = dist;
        }
      }
      if (bestDist <= TH_HIGH) {
        if (bestLevel == bestLevel2 && bestDist > A->matcher.nnratio * bestDist2) continue;
        match[bestIdx] = iMP;
        blocked[bestIdx] = A->mp_obs_pos[mo + iMP] ? 1 : 0;
        nmatches++;
      }
    }
    A->nmatches[b] = nmatches;
  }
  return FB_OK;
}

// ORBmatcher::BirdviewMatch with isProject = 0, ORBmatcher.cc:1602-1760
int orc_match_birdview(const fb_birdview_args *A) {
  const int ncell = A->grid.cols * A->grid.rows;
  std::vector<int> cand;
  for (int b = 0; b < A->batch; b++) {
    const size_t co = (size_t)b * A->cur_stride, ro = (size_t)b * A->ref_stride;
    const int ncur = A->n_cur[b], nref = A->n_ref[b];
    GridView G{&A->grid, A->cur_cell_start + (size_t)b * (ncell + 1), A->cur_cell_items + co, A->cur_kps + co};
    int32_t *m12 = A->match_ref_to_cur + ro, *md = A->match_dist + ro;
    for (int i = 0; i < nref; i++) { m12[i] = -1; md[i] = INT_MAX; }
    std::vector<int> rotHist[HISTO_LENGTH];
    int nmatches = 0;
    for (int i1 = 0; i1 < nref; i1++) {
      const fb_keypoint &kp1 = A->ref_kps[ro + i1];
      const int level1 = kp1.octave;
      if (level1 > 0) continue;
      features_in_area_bird(G, kp1.x, kp1.y, (float)A->window_size, level1, level1, cand);
      if (cand.empty()) continue;
      const uint8_t *d1 = A->ref_desc + (ro + i1) * 32;
      int bestDist = INT_MAX, bestDist2 = INT_MAX, bestIdx = -1;
      for (int i2 : cand) {
        if (i2 >= ncur) continue;
        const int dist = descriptor_distance(d1, A->cur_desc + (co + i2) * 32);
        if (dist < bestDist) { bestDist2 = bestDist; bestDist = dist; bestIdx = i2; }
        else if (dist < bestDist2) bestDist2 = dist;
      }
      if (bestDist <= TH_LOW) {
        if (bestDist < (float)bestDist2 * A->matcher.nnratio) {
          m12[i1] = bestIdx; md[i1] = bestDist; nmatches++;
        }
        if (A->matcher.check_orientation) {  // pushed even when the ratio test failed, :1712-1722
          float rot = kp1.angle - A->cur_kps[co + bestIdx].angle;
          rotHist[rot_bin(rot)].push_back(i1);
        }
      }
    }
    if (A->matcher.check_orientation) {
      int sz[HISTO_LENGTH], ind1, ind2, ind3;
      for (int i = 0; i < HISTO_LENGTH; i++) sz[i] = (int)rotHist[i].size();
      three_maxima(sz, HISTO_LENGTH, ind1, ind2, ind3);
      for (int i = 0; i < HISTO_LENGTH; i++) {
        if (i == ind1 || i == ind2 || i == ind3) continue;
        for (int idx1 : rotHist[i])
          if (m12[idx1] >= 0) { m12[idx1] = -1; nmatches--; }
      }
    }
    int nd = 0;
    for (int i = 0; i < nref; i++) if (m12[i] > 0) nd++;  // sic: > 0, :1755
    A->nmatches[b] = nmatches;
    A->n_dmatches[b] = nd;
  }
  return FB_OK;
}

}  // extern "C"

// ---- BoW-gated matchers -------------------------------------------------------------------------------------
namespace {
struct FV {  // one problem's DBoW2::FeatureVector view
  int n;
  const uint32_t *ids;
  const int32_t *start;
  const int32_t *items;
};
FV fv_of(const fb_feature_vector &v, int b) {
  return FV{v.n_nodes[b], v.node_ids + (size_t)b * v.node_stride, v.node_start + (size_t)b * (v.node_stride + 1),
            v.items + (size_t)b * v.item_stride};
}
int lower_bound_node(const FV &v, uint32_t id) {  // std::map::lower_bound
  int lo = 0, hi = v.n;
  while (lo < hi) { int m = (lo + hi) / 2; if (v.ids[m] < id) lo = m + 1; else hi = m; }
  return lo;
}
}  // namespace

extern "C" {

// ORBmatcher::SearchByBoW(KeyFrame*, Frame&, vector<MapPoint*>&), ORBmatcher.cc:160-289
int orc_match_bow(const fb_bow_args *A) {
  for (int b = 0; b < A->batch; b++) {
    const size_t ko = (size_t)b * A->kf_stride, fo = (size_t)b * A->f_stride;
    const int nF = A->n_f[b];
    const FV K = fv_of(A->kf_fv, b), F = fv_of(A->f_fv, b);
    int32_t *match = A->match_f_to_kf + fo;
    for (int i = 0; i < nF; i++) match[i] = -1;
    std::vector<int> rotHist[HISTO_LENGTH];
    int nmatches = 0;
    int ki = 0, fi = 0;
    while (ki < K.n && fi < F.n) {
      if (K.ids[ki] == F.ids[fi]) {
        for (int a = K.start[ki]; a < K.start[ki + 1]; a++) {
          const int realIdxKF = K.items[a];
          if (!A->kf_has_mp[ko + realIdxKF]) continue;
          const uint8_t *dKF = A->kf_desc + (ko + realIdxKF) * 32;
          int bestDist1 = 256, bestIdxF = -1, bestDist2 = 256;
          for (int c = F.start[fi]; c < F.start[fi + 1]; c++) {
            const int realIdxF = F.items[c];
            if (match[realIdxF] >= 0) continue;
            const int dist = descriptor_distance(dKF, A->f_desc + (fo + realIdxF) * 32);
            if (dist < bestDist1) { bestDist2 = bestDist1; bestDist1 = dist; bestIdxF = realIdxF; }
            else if (dist < bestDist2) bestDist2 = dist;
          }
          if (bestDist1 <= TH_LOW) {
            if ((float)bestDist1 < A->matcher.nnratio * (float)bestDist2) {
              match[bestIdxF] = realIdxKF;
              if (A->matcher.check_orientation) {
                float rot = A->kf_kps[ko + realIdxKF].angle - A->f_kps[fo + bestIdxF].angle;
                rotHist[rot_bin(rot)].push_back(bestIdxF);
              }
              nmatches++;
            }
          }
        }
        ki++; fi++;
      } else if (K.ids[ki] < F.ids[fi]) ki = lower_bound_node(K, F.ids[fi]);
      else fi = lower_bound_node(F, K.ids[ki]);
    }
    if (A->matcher.check_orientation) {
      int sz[HISTO_LENGTH], ind1, ind2, ind3;
      for (int i = 0; i < HISTO_LENGTH; i++) sz[i] = (int)rotHist[i].size();
      three_maxima(sz, HISTO_LENGTH, ind1, ind2, ind3);
      for (int i = 0; i < HISTO_LENGTH; i++) {
        if (i == ind1 || i == ind2 || i == ind3) continue;
        for (int idx : rotHist[i]) { match[idx] = -1; nmatches--; }
      }
    }
    A->nmatches[b] = nmatches;
  }
  return FB_OK;
}

// ORBmatcher::SearchForTriangulation (bOnlyStereo=false, mono keyframes), ORBmatcher.cc:658-824,
// CheckDistEpipolarLine :141-158
int orc_match_triangulation(const fb_triangulation_args *A) {
  for (int b = 0; b < A->batch; b++) {
    const size_t o1 = (size_t)b * A->kf1_stride, o2 = (size_t)b * A->kf2_stride;
    const int n1 = A->n1[b];
    const FV V1 = fv_of(A->fv1, b), V2 = fv_of(A->fv2, b);
    const float *F12 = A->F12 + (size_t)b * 9, *Cw = A->Cw1 + (size_t)b * 3, *R = A->R2w + (size_t)b * 9, *t = A->t2w + (size_t)b * 3;
    float C2[3];
    for (int r = 0; r < 3; r++) C2[r] = ((R[r * 3] * Cw[0] + R[r * 3 + 1] * Cw[1]) + R[r * 3 + 2] * Cw[2]) + t[r];
    const float invz = 1.0f / C2[2];
    const float ex = A->fx * C2[0] * invz + A->cx, ey = A->fy * C2[1] * invz + A->cy;
    int32_t *m12 = A->matches12 + o1;
    for (int i = 0; i < n1; i++) m12[i] = -1;
    std::vector<int> rotHist[HISTO_LENGTH];
    int nmatches = 0;
    int i1n = 0, i2n = 0;
    while (i1n < V1.n && i2n < V2.n) {
      if (V1.ids[i1n] == V2.ids[i2n]) {
        for (int a = V1.start[i1n]; a < V1.start[i1n + 1]; a++) {
          const int idx1 = V1.items[a];
          if (A->has_mp1[o1 + idx1]) continue;
          const fb_keypoint &kp1 = A->kps1[o1 + idx1];
          const uint8_t *d1 = A->desc1 + (o1 + idx1) * 32;
          int bestDist = TH_LOW, bestIdx2 = -1;
          for (int c = V2.start[i2n]; c < V2.start[i2n + 1]; c++) {
            const int idx2 = V2.items[c];
            if (A->has_mp2[o2 + idx2]) continue;  // vbMatched2 is never set in the reference
            const int dist = descriptor_distance(d1, A->desc2 + (o2 + idx2) * 32);
            if (dist > TH_LOW || dist > bestDist) continue;
            const fb_keypoint &kp2 = A->kps2[o2 + idx2];
            const float distex = ex - kp2.x, distey = ey - kp2.y;
            if (distex * distex + distey * distey < 100 * A->scale_factors[kp2.octave]) continue;
            // CheckDistEpipolarLine
            const float la = kp1.x * F12[0] + kp1.y * F12[3] + F12[6];
            const float lb = kp1.x * F12[1] + kp1.y * F12[4] + F12[7];
            const float lc = kp1.x * F12[2] + kp1.y * F12[5] + F12[8];
            const float num = la * kp2.x + lb * kp2.y + lc;
            const float den = la * la + lb * lb;
            if (den == 0) continue;
            const float dsqr = num * num / den;
            if (dsqr < 3.84 * A->level_sigma2[kp2.octave]) { bestIdx2 = idx2; bestDist = dist; }
          }
          if (bestIdx2 >= 0) {
            m12[idx1] = bestIdx2;
            nmatches++;
            if (A->matcher.check_orientation) {
              float rot = kp1.angle - A->kps2[o2 + bestIdx2].angle;
              rotHist[rot_bin(rot)].push_back(idx1);
            }
          }
        }
        i1n++; i2n++;
      } else if (V1.ids[i1n] < V2.ids[i2n]) i1n = lower_bound_node(V1, V2.ids[i2n]);
      else i2n = lower_bound_node(V2, V1.ids[i1n]);
    }
    if (A->matcher.check_orientation) {
      int sz[HISTO_LENGTH], ind1, ind2, ind3;
      for (int i = 0; i < HISTO_LENGTH; i++) sz[i] = (int)rotHist[i].size();
      three_maxima(sz, HISTO_LENGTH, ind1, ind2, ind3);
      for (int i = 0; i < HISTO_LENGTH; i++) {
        if (i == ind1 || i == ind2 || i == ind3) continue;
        for (int idx : rotHist[i]) { m12[idx] = -1; nmatches--; }
      }
    }
    A->nmatches[b] = nmatches;
  }
  return FB_OK;
}

}  // extern "C"

#include <limits>
#include "match_more_oracle.inc"
#include <algorithm>
#include "match_kf_oracle.inc"
