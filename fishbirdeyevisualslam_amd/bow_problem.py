"""Synthetic problems + C-ABI argument builders for the vocabulary-gated matchers (M5 SearchByBoW,
M7 SearchForTriangulation).  No vocabulary file ships with the reference, so the
DBoW2::FeatureVector of a frame is synthesised deterministically: NodeId = (desc[0] mod 10)*10 + desc[1] mod 10,
feature indices appended in keypoint order exactly like FeatureVector::addFeature."""
import numpy as np

from . import cabi, synth
from .cabi import fill


def feature_vector(desc):
    """-> (node_ids ascending, node_start, items) for one frame."""
    node = (desc[:, 0].astype(np.int64) % 10) * 10 + desc[:, 1].astype(np.int64) % 10
    order = np.argsort(node, kind="stable")
    ids, counts = np.unique(node, return_counts=True)
    start = np.concatenate([[0], np.cumsum(counts)]).astype(np.int32)
    return ids.astype(np.uint32), start, order.astype(np.int32)


def _fv_struct(fvs, node_stride, item_stride):
    B = len(fvs)
    n = np.array([len(f[0]) for f in fvs], np.int32)
    ids = np.zeros((B, node_stride), np.uint32)
    st = np.zeros((B, node_stride + 1), np.int32)
    it = np.zeros((B, item_stride), np.int32)
    for b, (i, s, t) in enumerate(fvs):
        ids[b, : len(i)] = i
        st[b, : len(s)] = s
        st[b, len(s):] = s[-1]
        it[b, : len(t)] = t
    v = cabi.FeatureVector()
    fill(v, node_stride=node_stride, item_stride=item_stride, n_nodes=n, node_ids=ids, node_start=st, items=it)
    return v, (n, ids, st, it)


def make_bow_problem(seed, n_kf=1500, n_f=1500, share_prefix=True):
    """KeyFrame vs Frame: 70% of the frame features are noisy copies of keyframe features.  Copies keep the first
    two descriptor bytes so that they fall into the same synthetic vocabulary node (true matches exist)."""
    g = synth.rng(seed)
    kf = synth.random_keypoints(g, n_kf, 640, 480)
    kd = synth.random_descriptors(g, n_kf)
    src = g.integers(0, n_kf, n_f)
    f = synth.random_keypoints(g, n_f, 640, 480)
    fd = synth.random_descriptors(g, n_f)
    is_copy = g.random(n_f) < 0.7
    cp = synth.flip_bits(g, kd[src[is_copy]], p=0.04)
    if share_prefix:
        cp[:, :2] = kd[src[is_copy], :2]
    fd[is_copy] = cp
    f["angle"][is_copy] = np.mod(kf["angle"][src[is_copy]] - 25.0 + g.normal(0, 4.0, int(is_copy.sum())), 360.0).astype(np.float32)
    return dict(kf_kps=kf, kf_desc=kd, kf_has_mp=(g.random(n_kf) < 0.6).astype(np.uint8), f_kps=f, f_desc=fd)


def bow_args(problems, nnratio=0.7, check_ori=1):
    B = len(problems)
    ks = max(len(p["kf_kps"]) for p in problems)
    fs = max(len(p["f_kps"]) for p in problems)
    stack = lambda key, n, dt, tail=(): np.stack([np.concatenate([p[key], np.zeros((n - len(p[key]),) + tail, dt)]) for p in problems])
    keep = dict(n_kf=np.array([len(p["kf_kps"]) for p in problems], np.int32), kf_kps=stack("kf_kps", ks, cabi.KP_DTYPE),
                kf_desc=stack("kf_desc", ks, np.uint8, (32,)), kf_has_mp=stack("kf_has_mp", ks, np.uint8),
                n_f=np.array([len(p["f_kps"]) for p in problems], np.int32), f_kps=stack("f_kps", fs, cabi.KP_DTYPE),
                f_desc=stack("f_desc", fs, np.uint8, (32,)))
    kfv, k1 = _fv_struct([feature_vector(p["kf_desc"]) for p in problems], 100, ks)
    ffv, k2 = _fv_struct([feature_vector(p["f_desc"]) for p in problems], 100, fs)
    out = dict(match_f_to_kf=np.full((B, fs), -7, np.int32), nmatches=np.full(B, -7, np.int32))
    a = cabi.BowArgs()
    fill(a, batch=B, kf_stride=ks, f_stride=fs, **keep, **out)
    a.kf_fv, a.f_fv = kfv, ffv
    fill(a.matcher, nnratio=nnratio, check_orientation=check_ori)
    return a, out, (keep, k1, k2)


def make_triangulation_problem(seed, n1=1500, n2=1500, w=640, h=480, fx=400.0, fy=400.0):
    """Two keyframes of a static scene: KF2 features are projections of KF1's back-projected features under a
    known relative pose, so that the epipolar constraint holds for true matches (plus outliers)."""
    g = synth.rng(seed)
    cx, cy = w / 2.0, h / 2.0
    K = np.array([[fx, 0, cx], [0, fy, cy], [0, 0, 1.0]])
    T1 = synth.random_pose(g)
    D = np.eye(4)
    D[:3, :3] = synth.so3_exp(g.normal(0, 0.03, 3))
    D[:3, 3] = [0.6, 0.05, 0.1]
    T2 = D @ T1
    k1 = synth.random_keypoints(g, n1, w, h, margin=20)
    d1 = synth.random_descriptors(g, n1)
    z = g.uniform(3.0, 25.0, n1)
    Xc1 = np.stack([(k1["x"] - cx) / fx * z, (k1["y"] - cy) / fy * z, z], 1)
    Xw = (T1[:3, :3].T @ (Xc1 - T1[:3, 3]).T).T
    Xc2 = (T2[:3, :3] @ Xw.T).T + T2[:3, 3]
    u2 = Xc2[:, 0] / Xc2[:, 2] * fx + cx
    v2 = Xc2[:, 1] / Xc2[:, 2] * fy + cy
    src = g.integers(0, n1, n2)
    k2 = synth.random_keypoints(g, n2, w, h, margin=20)
    d2 = synth.random_descriptors(g, n2)
    is_copy = g.random(n2) < 0.7
    k2["x"][is_copy] = (u2[src[is_copy]] + g.normal(0, 0.7, int(is_copy.sum()))).astype(np.float32)
    k2["y"][is_copy] = (v2[src[is_copy]] + g.normal(0, 0.7, int(is_copy.sum()))).astype(np.float32)
    k2["octave"][is_copy] = k1["octave"][src[is_copy]]
    cp = synth.flip_bits(g, d1[src[is_copy]], p=0.04)
    cp[:, :2] = d1[src[is_copy], :2]
    d2[is_copy] = cp
    # F12 = K^-T [t12]x R12 K^-1 with x1' F12 x2 = 0  (LocalMapping::ComputeF12, LocalMapping.cc:560-577)
    R1, t1, R2, t2 = T1[:3, :3], T1[:3, 3], T2[:3, :3], T2[:3, 3]
    R12 = R1 @ R2.T
    t12 = -R1 @ R2.T @ t2 + t1
    tx = np.array([[0, -t12[2], t12[1]], [t12[2], 0, -t12[0]], [-t12[1], t12[0], 0]])
    F12 = np.linalg.inv(K).T @ tx @ R12 @ np.linalg.inv(K)
    Cw1 = -R1.T @ t1
    return dict(kps1=k1, desc1=d1, has_mp1=(g.random(n1) < 0.3).astype(np.uint8), kps2=k2, desc2=d2,
                has_mp2=(g.random(n2) < 0.3).astype(np.uint8), F12=F12.astype(np.float32).reshape(9), Cw1=Cw1.astype(np.float32),
                R2w=R2.astype(np.float32).reshape(9), t2w=t2.astype(np.float32), fx=fx, fy=fy, cx=cx, cy=cy)


def triangulation_args(problems, nnratio=0.6, check_ori=0):
    B = len(problems)
    p0 = problems[0]
    s1 = max(len(p["kps1"]) for p in problems)
    s2 = max(len(p["kps2"]) for p in problems)
    stack = lambda key, n, dt, tail=(): np.stack([np.concatenate([p[key], np.zeros((n - len(p[key]),) + tail, dt)]) for p in problems])
    keep = dict(n1=np.array([len(p["kps1"]) for p in problems], np.int32), kps1=stack("kps1", s1, cabi.KP_DTYPE),
                desc1=stack("desc1", s1, np.uint8, (32,)), has_mp1=stack("has_mp1", s1, np.uint8),
                n2=np.array([len(p["kps2"]) for p in problems], np.int32), kps2=stack("kps2", s2, cabi.KP_DTYPE),
                desc2=stack("desc2", s2, np.uint8, (32,)), has_mp2=stack("has_mp2", s2, np.uint8),
                F12=np.stack([p["F12"] for p in problems]), Cw1=np.stack([p["Cw1"] for p in problems]),
                R2w=np.stack([p["R2w"] for p in problems]), t2w=np.stack([p["t2w"] for p in problems]))
    fv1, k1 = _fv_struct([feature_vector(p["desc1"]) for p in problems], 100, s1)
    fv2, k2 = _fv_struct([feature_vector(p["desc2"]) for p in problems], 100, s2)
    out = dict(matches12=np.full((B, s1), -7, np.int32), nmatches=np.full(B, -7, np.int32))
    a = cabi.TriangulationArgs()
    sf, _, sig2, _ = synth.scale_tables()
    fill(a, batch=B, kf1_stride=s1, kf2_stride=s2, fx=p0["fx"], fy=p0["fy"], cx=p0["cx"], cy=p0["cy"],
         scale_factors=[float(x) for x in sf], level_sigma2=[float(x) for x in sig2], **keep, **out)
    a.fv1, a.fv2 = fv1, fv2
    fill(a.matcher, nnratio=nnratio, check_orientation=check_ori)
    return a, out, (keep, k1, k2)


def make_vocabulary(seed, k=10, L=3, stop_frac=0.05):
    """Complete k-ary tree of depth L in BFS node order (node 0 = root), random node descriptors (every node = its parent
    with a quarter of the bits flipped, so that the descent is meaningful), idf-like leaf weights.  k = 10, L = 6 is the
    shape of the stock ORB vocabulary (1.1 M nodes); built level by level so that it takes seconds, not minutes."""
    g = synth.rng(seed)
    n_nodes = sum(k ** l for l in range(L + 1))
    first_leaf = sum(k ** l for l in range(L))
    child_start = np.minimum(np.arange(n_nodes + 1, dtype=np.int64), first_leaf) * k
    child_start = child_start.astype(np.int32)
    children = np.arange(1, n_nodes, dtype=np.int32)
    desc = np.zeros((n_nodes, 32), np.uint8)
    desc[0] = synth.random_descriptors(g, 1)[0]
    lo = 1
    for l in range(1, L + 1):
        hi = lo + k ** l
        for c0 in range(lo, hi, 65536):   # bounded scratch: 65536 x 256 random numbers at a time
            c1 = min(hi, c0 + 65536)
            parents = (np.arange(c0, c1) - 1) // k
            desc[c0:c1] = synth.flip_bits(g, desc[parents], p=0.25)
        lo = hi
    weights = np.zeros(n_nodes, np.float64)
    weights[first_leaf:] = g.uniform(0.5, 9.0, n_nodes - first_leaf)
    weights[first_leaf:][g.random(n_nodes - first_leaf) < stop_frac] = 0.0   # stopped words
    word_ids = np.full(n_nodes, -1, np.int32)
    word_ids[first_leaf:] = np.arange(n_nodes - first_leaf)
    keep = dict(child_start=child_start, children=children, descriptors=desc, weights=weights, word_ids=word_ids)
    v = cabi.Vocabulary()
    fill(v, n_nodes=n_nodes, L=L, **keep)
    return v, keep, first_leaf
