"""Generates tests/golden/dbow2_ref.npz from the REFERENCE's own DBoW2::FeatureVector / DBoW2::BowVector, compiled from
/root/reference by `make -C oracle ref` (oracle/_ref/libref_dbow2.so).  Run in the build container only (the reference
does not exist on the GPU box); the fixture is data: input id / weight streams and the containers the reference code
produced from them.

    python tests/golden/make_dbow2_ref_golden.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, ".."))
import ref_dbow2  # noqa: E402


def main():
    g = np.random.Generator(np.random.PCG64(8800))
    out = {}
    for c, (n, nnodes, nwords) in enumerate([(2000, 100, 900), (37, 5, 12), (1, 1, 1), (1500, 1000, 100000)]):
        node = g.integers(0, nnodes, n).astype(np.uint32) * 7 + 3
        word = g.integers(0, nwords, n).astype(np.uint32)
        w = g.uniform(0.01, 9.0, n)
        ids, start, items = ref_dbow2.feature_vector(node)
        bid, bval = ref_dbow2.bow_vector(word, w, True)
        bid0, bval0 = ref_dbow2.bow_vector(word, w, False)
        out.update({"c%d_node" % c: node, "c%d_word" % c: word, "c%d_w" % c: w, "c%d_fv_ids" % c: ids, "c%d_fv_start" % c: start,
                    "c%d_fv_items" % c: items, "c%d_bow_ids" % c: bid, "c%d_bow_vals" % c: bval, "c%d_bow_raw" % c: bval0})
    np.savez_compressed(os.path.join(HERE, "dbow2_ref.npz"), **out)
    print("wrote dbow2_ref.npz")


if __name__ == "__main__":
    main()
