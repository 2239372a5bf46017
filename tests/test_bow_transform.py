"""DBoW2 vocabulary transform (Frame::ComputeBoW): oracle known answers on a synthetic tree (the reference ships no
vocabulary file), then HIP parity through the C-ABI -- word ids, node ids and items bit-exact, BowVector values bit-exact
(both sides add the same doubles in the same order)."""
import ctypes as C

import numpy as np
import pytest

import oracle_lib as O
from fishbirdeyevisualslam_amd import cabi, synth
from fishbirdeyevisualslam_amd.cabi import fill


from fishbirdeyevisualslam_amd.bow_problem import make_vocabulary  # noqa: E402  (the tests below and other test modules use it)


def make_args(descs, levelsup=2):
    B = len(descs)
    fs = max(max(len(d) for d in descs), 1)
    keep = dict(n_f=np.array([len(d) for d in descs], np.int32),
                desc=np.stack([np.concatenate([d, np.zeros((fs - len(d), 32), np.uint8)]) for d in descs]))
    out = dict(n_words=np.full(B, -7, np.int32), bow_ids=np.zeros((B, fs), np.uint32), bow_vals=np.zeros((B, fs), np.float64),
               fv_n_nodes=np.full(B, -7, np.int32), fv_node_ids=np.zeros((B, fs), np.uint32),
               fv_node_start=np.zeros((B, fs + 1), np.int32), fv_items=np.full((B, fs), -7, np.int32))
    a = cabi.BowTransformArgs()
    fill(a, batch=B, f_stride=fs, levelsup=levelsup, **keep, **out)
    return a, out, keep


def _descs(seed, sizes, voc_keep, first_leaf):
    g = synth.rng(seed)
    out = []
    for n in sizes:
        leaves = g.integers(first_leaf, len(voc_keep["weights"]), n)
        out.append(synth.flip_bits(g, voc_keep["descriptors"][leaves], p=0.05) if n else np.zeros((0, 32), np.uint8))
    return out


def test_oracle_transform_known_answers():
    v, vk, first_leaf = make_vocabulary(9100, k=10, L=3)
    descs = _descs(9101, [1500, 40], vk, first_leaf)
    a, out, k = make_args(descs, levelsup=2)
    rc = O.lib().orc_bow_transform(C.byref(v), C.byref(a))
    assert rc == 0
    for b, d in enumerate(descs):
        nw, nn = out["n_words"][b], out["fv_n_nodes"][b]
        ids, vals = out["bow_ids"][b, :nw], out["bow_vals"][b, :nw]
        assert (np.diff(ids.astype(np.int64)) > 0).all() and abs(vals.sum() - 1.0) < 1e-12 and (vals > 0).all()
        # brute-force descent in numpy
        words, nids = [], []
        for f in d:
            node = 0
            for level in range(1, 4):
                ch = np.arange(node * 10 + 1, node * 10 + 11)
                dist = np.unpackbits(vk["descriptors"][ch] ^ f, axis=1).sum(1)
                node = int(ch[np.argmin(dist)])          # first minimum
                if level == 1:
                    nid = node                           # nid_level = L - levelsup = 1
            words.append(node - first_leaf)
            nids.append(nid)
        words, nids = np.array(words), np.array(nids)
        kept = vk["weights"][words + first_leaf] > 0
        np.testing.assert_array_equal(ids, np.unique(words[kept]))
        tf = np.array([vk["weights"][w + first_leaf] * (words[kept] == w).sum() for w in ids])
        np.testing.assert_allclose(vals, tf / tf.sum(), rtol=1e-12)
        node_ids = out["fv_node_ids"][b, :nn]
        st = out["fv_node_start"][b, :nn + 1]
        np.testing.assert_array_equal(node_ids, np.unique(nids[kept]))
        for j, nd in enumerate(node_ids):
            np.testing.assert_array_equal(out["fv_items"][b, st[j]:st[j + 1]], np.nonzero(kept & (nids == nd))[0])
        assert st[nn] == kept.sum()


@pytest.mark.gpu
@pytest.mark.parametrize("seed,k,L,levelsup,sizes", [(9100, 10, 3, 2, [2000, 1500, 33]), (9110, 10, 4, 4, [2064, 0, 1]),
                                                      (9120, 4, 5, 2, [4000, 4096, 700])])
def test_gpu_transform_matches_oracle(seed, k, L, levelsup, sizes):
    import fishbirdeyevisualslam_amd as fb
    v, vk, first_leaf = make_vocabulary(seed, k=k, L=L)
    descs = _descs(seed + 1, sizes, vk, first_leaf)
    a, oo, k1 = make_args(descs, levelsup)
    assert O.lib().orc_bow_transform(C.byref(v), C.byref(a)) == 0
    a2, oh, k2 = make_args(descs, levelsup)
    fb.check(fb.lib().fb_bow_transform(C.byref(v), C.byref(a2)), "fb_bow_transform")
    np.testing.assert_array_equal(oh["n_words"], oo["n_words"])
    np.testing.assert_array_equal(oh["fv_n_nodes"], oo["fv_n_nodes"])
    for b in range(len(sizes)):
        nw, nn = oo["n_words"][b], oo["fv_n_nodes"][b]
        np.testing.assert_array_equal(oh["bow_ids"][b, :nw], oo["bow_ids"][b, :nw])
        np.testing.assert_array_equal(oh["bow_vals"][b, :nw], oo["bow_vals"][b, :nw])
        np.testing.assert_array_equal(oh["fv_node_ids"][b, :nn], oo["fv_node_ids"][b, :nn])
        np.testing.assert_array_equal(oh["fv_node_start"][b, :nn + 1], oo["fv_node_start"][b, :nn + 1])
        ni = oo["fv_node_start"][b, nn]
        np.testing.assert_array_equal(oh["fv_items"][b, :ni], oo["fv_items"][b, :ni])
