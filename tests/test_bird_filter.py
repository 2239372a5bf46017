"""Geometric test of Tracking::FilterBirdOutlierInFront (Tracking.cc:1825-1914): oracle invariants on the CPU, HIP parity
on the GPU (flags bit-exact; world points bit-exact where a match is kept)."""
import numpy as np
import pytest

import oracle_lib as O
from fishbirdeyevisualslam_amd import cabi, synth
from fishbirdeyevisualslam_amd.cabi import fill


def make(seed, n1=1000, n2=1000, nm=800, B=3):
    g = synth.rng(seed)
    keep = dict(n_matches=np.full(B, nm, np.int32), query_idx=g.integers(0, n1, (B, max(nm, 1))).astype(np.int32),
                train_idx=g.integers(0, n2, (B, max(nm, 1))).astype(np.int32),
                cam_xyz1=g.uniform(-8, 8, (B, n1, 3)).astype(np.float32), cam_xyz2=np.zeros((B, n2, 3), np.float32),
                Tcw1=np.zeros((B, 12), np.float32), Tcw2=np.zeros((B, 12), np.float32),
                occupied2=(g.random((B, n2)) < 0.15).astype(np.uint8))
    for b in range(B):
        T1 = synth.random_pose(g)
        T2 = synth.perturb_pose(g, T1, rot_sigma=0.01, t_sigma=0.1)
        keep["Tcw1"][b], keep["Tcw2"][b] = synth.to12(T1), synth.to12(T2)
        # frame-2 camera points: the transported frame-1 point of (some) match that trains on them, plus noise
        rel = T2 @ np.linalg.inv(T1)
        keep["cam_xyz2"][b] = g.uniform(-8, 8, (n2, 3))
        for i in range(nm):
            q, t = keep["query_idx"][b, i], keep["train_idx"][b, i]
            p = rel[:3, :3] @ keep["cam_xyz1"][b, q].astype(np.float64) + rel[:3, 3]
            keep["cam_xyz2"][b, t] = p + g.normal(0, 0.03 if g.random() < 0.7 else 0.5, 3)
    out = dict(keep=np.full((B, max(nm, 1)), 9, np.uint8), pt_world=np.full((B, max(nm, 1), 3), -7.0, np.float32))
    a = cabi.BirdFilterArgs()
    fill(a, batch=B, match_stride=max(nm, 1), kp1_stride=n1, kp2_stride=n2, window_size=0.1, **keep, **out)
    return a, out, keep


def test_oracle_filter_invariants():
    a, out, k = make(9000)
    O.call("orc_bird_filter_matches", a)
    for b in range(3):
        kept = out["keep"][b] == 1
        assert 50 < kept.sum() < 800
        t = k["train_idx"][b][kept]
        assert len(np.unique(t)) == kept.sum()                      # one MapPointBird per train slot
        assert (k["occupied2"][b][t] == 0).all()
        T1 = np.vstack([k["Tcw1"][b].reshape(3, 4), [0, 0, 0, 1]]).astype(np.float64)
        T2 = np.vstack([k["Tcw2"][b].reshape(3, 4), [0, 0, 0, 1]]).astype(np.float64)
        pw = (np.linalg.inv(T1) @ np.c_[k["cam_xyz1"][b][k["query_idx"][b][kept]], np.ones(kept.sum())].T).T
        np.testing.assert_allclose(out["pt_world"][b][kept], pw[:, :3], atol=2e-4)
        d = np.linalg.norm((T2 @ pw.T).T[:, :3] - k["cam_xyz2"][b][t], axis=1)
        assert d.max() < 0.1 + 1e-4


@pytest.mark.gpu
@pytest.mark.parametrize("seed,n1,n2,nm", [(9000, 1000, 1000, 800), (9001, 2064, 300, 2000), (9002, 5, 5, 0)])
def test_gpu_filter_matches_oracle(seed, n1, n2, nm):
    import hip_lib as H
    a, oo, k = make(seed, n1, n2, nm)
    O.call("orc_bird_filter_matches", a)
    a2, oh, k2 = make(seed, n1, n2, nm)
    H.call("fb_bird_filter_matches", a2)
    if nm == 0:
        return
    np.testing.assert_array_equal(oh["keep"][:, :nm], oo["keep"][:, :nm])
    kept = oo["keep"][:, :nm] == 1
    np.testing.assert_array_equal(oh["pt_world"][:, :nm][kept], oo["pt_world"][:, :nm][kept])
