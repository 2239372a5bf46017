/*
 * dbow2_shim.cpp -- C entry points over the REFERENCE's own DBoW2::FeatureVector and DBoW2::BowVector, compiled from
 * /root/reference/Thirdparty/DBoW2/DBoW2/{FeatureVector,BowVector}.cpp where they lie (oracle/Makefile, target ref) into
 * oracle/_ref/libref_dbow2.so.  TEST INFRASTRUCTURE ONLY: it pins oracle/bow_oracle.cpp (container order of the
 * FeatureVector, accumulation order and L1 normalisation of the BowVector); the product never loads it.
 * These two files are the only part of the reference that builds here (everything else needs OpenCV / Eigen).
 */
#include <cstdint>

#include "BowVector.h"      // the reference's headers (-I/root/reference/Thirdparty/DBoW2/DBoW2)
#include "FeatureVector.h"

extern "C" {

// FeatureVector::addFeature(node_ids[i], i) for i = 0..n-1 (the loop of TemplatedVocabulary.h:1160-1172), flattened in
// iteration order: out_ids[k], CSR out_start[k..k+1] into out_items.  Returns the number of nodes.
int ref_feature_vector(int n, const uint32_t *node_ids, uint32_t *out_ids, int32_t *out_start, int32_t *out_items) {
  DBoW2::FeatureVector fv;
  for (int i = 0; i < n; i++) fv.addFeature(node_ids[i], (unsigned)i);
  int k = 0, off = 0;
  for (DBoW2::FeatureVector::const_iterator it = fv.begin(); it != fv.end(); ++it, ++k) {
    out_ids[k] = it->first;
    out_start[k] = off;
    for (size_t j = 0; j < it->second.size(); j++) out_items[off++] = (int32_t)it->second[j];
  }
  out_start[k] = off;
  return k;
}

// BowVector::addWeight(word_ids[i], weights[i]) for i = 0..n-1, then normalize(L1) when do_normalize (TemplatedVocabulary.h
// :1150-1158,1188-1192).  Returns the number of words.
int ref_bow_vector(int n, const uint32_t *word_ids, const double *weights, int do_normalize, uint32_t *out_ids, double *out_vals) {
  DBoW2::BowVector v;
  for (int i = 0; i < n; i++) v.addWeight(word_ids[i], weights[i]);
  if (do_normalize) v.normalize(DBoW2::L1);
  int k = 0;
  for (DBoW2::BowVector::const_iterator it = v.begin(); it != v.end(); ++it, ++k) { out_ids[k] = it->first; out_vals[k] = it->second; }
  return k;
}
}
