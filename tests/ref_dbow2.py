"""ctypes loader of oracle/_ref/libref_dbow2.so = the reference's own DBoW2 FeatureVector.cpp / BowVector.cpp compiled
where they lie (oracle/Makefile, target `ref`).  Test infrastructure; absent when neither the prebuilt .so nor
/root/reference is there."""
import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SO = os.path.join(ROOT, "oracle", "_ref", "libref_dbow2.so")
_LIB = None


def available():
    return os.path.exists(SO) or os.path.isdir("/root/reference/Thirdparty/DBoW2/DBoW2")


def lib():
    global _LIB
    if _LIB is None:
        if not os.path.exists(SO):
            subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "ref"])
        _LIB = C.CDLL(SO)
    return _LIB


def feature_vector(node_ids):
    node_ids = np.ascontiguousarray(node_ids, np.uint32)
    n = len(node_ids)
    ids = np.zeros(n + 1, np.uint32)
    start = np.zeros(n + 2, np.int32)
    items = np.zeros(n + 1, np.int32)
    k = lib().ref_feature_vector(n, C.c_void_p(node_ids.ctypes.data), C.c_void_p(ids.ctypes.data), C.c_void_p(start.ctypes.data),
                                 C.c_void_p(items.ctypes.data))
    return ids[:k].copy(), start[:k + 1].copy(), items[:start[k]].copy()


def bow_vector(word_ids, weights, normalize=True):
    word_ids = np.ascontiguousarray(word_ids, np.uint32)
    weights = np.ascontiguousarray(weights, np.float64)
    n = len(word_ids)
    ids = np.zeros(n + 1, np.uint32)
    vals = np.zeros(n + 1, np.float64)
    k = lib().ref_bow_vector(n, C.c_void_p(word_ids.ctypes.data), C.c_void_p(weights.ctypes.data), int(normalize),
                             C.c_void_p(ids.ctypes.data), C.c_void_p(vals.ctypes.data))
    return ids[:k].copy(), vals[:k].copy()
