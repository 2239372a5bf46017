"""Optimizer::BundleAdjustmentWithOdom (global BA, Optimizer.cc:1786-2135) through fb_global_ba vs the oracle:
one optimize(nIterations), optional Huber kernel, no outlier classification.  Poses / landmarks within 1e-4 relative."""
import ctypes as C

import numpy as np
import pytest

import oracle_lib as O
from fishbirdeyevisualslam_amd import ba_problem, synth

REL_TOL = 1e-4


def _rel(a, b):
    return np.abs(a - b).max() / max(1.0, np.abs(b).max())


def _problem(seed, n_kf, n_mp, n_mpb):
    p = synth.make_ba_problem(seed, n_kf=n_kf, n_fixed=1, n_mp=n_mp, n_mpb=n_mpb)
    p["odom_kf_i"] = p["odom_kf_i"][:0]          # the pose-graph block is commented out in the reference (:2004-2037)
    p["odom_kf_j"] = p["odom_kf_j"][:0]
    p["odom_Tij"] = p["odom_Tij"][:0]
    p["odom_info"] = p["odom_info"][:0]
    return p


def test_oracle_global_ba_reduces_the_error():
    p = _problem(4100, 6, 400, 100)
    a, out, keep = ba_problem.local_ba_args(p, with_odom=1)
    def reproj_rms(kf_T, xw):
        T = np.asarray(kf_T, np.float64).reshape(-1, 3, 4)[p["obs_kf"]]
        X = np.asarray(xw, np.float64).reshape(-1, 3)[p["obs_mp"]]
        pc = np.einsum("nij,nj->ni", T[:, :, :3], X) + T[:, :, 3]
        uv = np.stack([p["fx"] * pc[:, 0] / pc[:, 2] + p["cx"], p["fy"] * pc[:, 1] / pc[:, 2] + p["cy"]], 1)
        r = np.linalg.norm(uv - np.asarray(p["obs_uv"], np.float64).reshape(-1, 2), axis=1)
        return np.sqrt(np.median(r ** 2))                      # median: the problem carries gross outliers
    before = reproj_rms(p["kf_Tcw"], p["mp_xw"])
    rc = O.lib().orc_global_ba(C.byref(a), 10, 1)
    assert rc == 0
    after = reproj_rms(out["kf_Tcw"], out["mp_xw"])
    assert np.isfinite(out["kf_Tcw"]).all() and after < before
    assert (out["obs_outlier"] == keep["obs_outlier"]).all()      # untouched: the global BA classifies nothing


@pytest.mark.gpu
@pytest.mark.parametrize("seed,n_kf,n_mp,n_mpb,its,robust", [(4100, 6, 500, 120, 10, 1), (4101, 12, 3000, 600, 20, 0),
                                                             (4102, 20, 8000, 2000, 10, 1)])
def test_gpu_global_ba_matches_oracle(seed, n_kf, n_mp, n_mpb, its, robust):
    import fishbirdeyevisualslam_amd as fb
    p = _problem(seed, n_kf, n_mp, n_mpb)
    a, oo, k = ba_problem.local_ba_args(p, with_odom=1)
    assert O.lib().orc_global_ba(C.byref(a), its, robust) == 0
    a2, oh, k2 = ba_problem.local_ba_args(p, with_odom=1)
    fb.check(fb.lib().fb_global_ba(C.byref(a2), its, robust), "fb_global_ba")
    assert _rel(oh["kf_Tcw"], oo["kf_Tcw"]) <= REL_TOL
    assert _rel(oh["mp_xw"], oo["mp_xw"]) <= REL_TOL
    assert _rel(oh["mpb_xw"], oo["mpb_xw"]) <= REL_TOL
