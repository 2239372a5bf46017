/*
 * bow_oracle.cpp -- CPU restatement of the DBoW2 vocabulary transform (TEST INFRASTRUCTURE ONLY; see orb_oracle.cpp
 * header for who may call it).
 *
 * Follows /root/reference/Thirdparty/DBoW2/DBoW2/TemplatedVocabulary.h:1126-1194 (transform of a feature set, TF_IDF
 * branch), :1217-1259 (descent of one feature), FORB::distance (FORB.cpp:85-103), BowVector::addWeight and
 * BowVector::normalize(L1) (BowVector.cpp:30-84), FeatureVector::addFeature (FeatureVector.cpp:31-45); call site
 * Frame::ComputeBoW (src/Frame.cc:628-635).  DBoW2 IS vendored in the reference, so this restatement is checked against
 * source text, but the reference holds no vocabulary file and no fixture: trees in tests are synthetic.
 */
#include <cmath>
#include <cstdint>
#include <cstring>
#include <map>
#include <vector>

#include "../include/fishbird.h"

namespace {
int forb_distance(const uint8_t *a, const uint8_t *b) {  // FORB.cpp:85-103 (same bit trick as ORBmatcher)
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
}  // namespace

extern "C" int orc_bow_transform(const fb_vocabulary *V, const fb_bow_transform_args *A) {
  for (int b = 0; b < A->batch; b++) {
    const size_t fo = (size_t)b * A->f_stride;
    std::map<uint32_t, double> bow;                      // BowVector
    std::map<uint32_t, std::vector<uint32_t>> fv;        // FeatureVector
    for (int i = 0; i < A->n_f[b]; i++) {
      const uint8_t *feature = A->desc + (fo + i) * 32;
      const int nid_level = V->L - A->levelsup;
      uint32_t nid = 0;                                   // (left unset by the reference when the level is never reached)
      int final_id = 0, current_level = 0;
      do {
        ++current_level;
        const int c0 = V->child_start[final_id], c1 = V->child_start[final_id + 1];
        final_id = V->children[c0];
        double best_d = forb_distance(feature, V->descriptors + (size_t)final_id * 32);
        for (int c = c0 + 1; c < c1; c++) {
          const int id = V->children[c];
          const double d = forb_distance(feature, V->descriptors + (size_t)id * 32);
          if (d < best_d) { best_d = d; final_id = id; }
        }
        if (current_level == nid_level) nid = (uint32_t)final_id;
      } while (V->child_start[final_id + 1] > V->child_start[final_id]);  // !isLeaf()
      const double w = V->weights[final_id];
      if (w > 0) {
        bow[(uint32_t)V->word_ids[final_id]] += w;        // addWeight: insert v or += v (0.0 + v == v exactly)
        fv[nid].push_back((uint32_t)i);
      }
    }
    double norm = 0.0;
    for (auto &kv : bow) norm += std::fabs(kv.second);
    if (norm > 0.0)
      for (auto &kv : bow) kv.second /= norm;
    int k = 0;
    for (auto &kv : bow) { A->bow_ids[fo + k] = kv.first; A->bow_vals[fo + k] = kv.second; k++; }
    A->n_words[b] = k;
    int n = 0, off = 0;
    int32_t *st = A->fv_node_start + (size_t)b * (A->f_stride + 1);
    for (auto &kv : fv) {
      A->fv_node_ids[fo + n] = kv.first;
      st[n] = off;
      for (uint32_t f : kv.second) A->fv_items[fo + off++] = (int32_t)f;
      n++;
    }
    st[n] = off;
    A->fv_n_nodes[b] = n;
  }
  return FB_OK;
}
