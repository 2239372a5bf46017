"""Frames with more key points than the LDS-staged descriptor table holds (~2900): the matchers keep the descriptors in
HBM/L2 and stage only positions, octaves and the grid.  Same bit-exact parity as the regular sizes."""
import numpy as np
import pytest

import hip_lib as H
import oracle_lib as O
from fishbirdeyevisualslam_amd import kf_problems as KP, more_problems as M, problems as P, synth

pytestmark = pytest.mark.gpu


def _both(build, oracle_name, hip_name, outs):
    a, out_o, keep = build()
    O.call(oracle_name, a)
    a2, out_h, keep2 = build()
    H.call(hip_name, a2)
    for k in outs:
        np.testing.assert_array_equal(out_h[k], out_o[k], err_msg=k)
    return out_o


@pytest.mark.parametrize("n", [3200, 5000])
def test_front_matchers_large(n):
    geom = P.grid_geom(synth.front_grid_geom(1280, 720))
    probs = [synth.make_proj_frame_problem(2600 + i, n, n - 300, dup_frac=0.2) for i in range(2)]
    cs, ci = P.build_grid_host([p["cur_kps"] for p in probs], geom, O.grid_build, n)
    out = _both(lambda: P.proj_frame_args(probs, cs, ci), "orc_match_projection_frame", "fb_match_projection_frame",
                ["match_cur_to_last", "nmatches"])
    assert out["nmatches"].min() > 500
    probs = [synth.make_proj_points_problem(2700 + i, n, n + 500) for i in range(2)]
    cs, ci = P.build_grid_host([p["cur_kps"] for p in probs], geom, O.grid_build, n)
    _both(lambda: P.proj_points_args(probs, cs, ci, th=1.0), "orc_match_projection_points", "fb_match_projection_points",
          ["match_cur_to_mp", "nmatches"])
    probs = [M.make_proj_kf_problem(2800 + i, n, n) for i in range(2)]
    cs, ci = P.build_grid_host([p["cur_kps"] for p in probs], geom, O.grid_build, n)
    _both(lambda: M.proj_kf_args(probs, cs, ci), "orc_match_projection_keyframe", "fb_match_projection_keyframe",
          ["match_cur_to_kf", "nmatches"])


def test_bird_matchers_large():
    n = 4000
    geom = P.grid_geom(synth.bird_grid_geom(512, 512))
    probs = [synth.make_bird_mp_problem(2900 + i, n, 3000) for i in range(2)]
    cs, ci = P.build_grid_host([p["cur_kps"] for p in probs], geom, O.grid_build, n)
    _both(lambda: P.bird_mp_args(probs, cs, ci, prefill=-1), "orc_match_bird_mappoints", "fb_match_bird_mappoints",
          ["match_cur_to_ref", "ninliers"])
    probs = [synth.make_birdview_problem(2950 + i, n, 3500) for i in range(2)]
    cs, ci = P.build_grid_host([p["cur_kps"] for p in probs], geom, O.grid_build, n)
    _both(lambda: P.birdview_args(probs, cs, ci), "orc_match_birdview", "fb_match_birdview",
          ["match_ref_to_cur", "match_dist", "nmatches", "n_dmatches"])


def test_keyframe_searches_large():
    n = 4000
    geom = P.grid_geom(synth.front_grid_geom(KP.W, KP.H))
    for sim3, oname, hname in ((False, "orc_fuse_search", "fb_fuse_search"), (True, "orc_fuse_sim3_search", "fb_fuse_sim3_search")):
        probs = [KP.make_kf_points_problem(3000 + i, n, 5000, sim3) for i in range(2)]
        cs, ci = P.build_grid_host([p["kf_kps"] for p in probs], geom, O.grid_build, n)
        _both(lambda: KP.fuse_args(probs, cs, ci), oname, hname, ["best_idx"])
    _both(lambda: KP.proj_sim3_args(probs, cs, ci), "orc_match_projection_sim3", "fb_match_projection_sim3",
          ["match_kf_to_mp", "nmatches"])
    probs = [KP.make_sim3_problem(3100 + i, n, n - 500, 2500) for i in range(2)]
    g1 = P.build_grid_host([p["kps1"] for p in probs], geom, O.grid_build, n)
    g2 = P.build_grid_host([p["kps2"] for p in probs], geom, O.grid_build, n - 500)
    _both(lambda: KP.sim3_args(probs, g1, g2), "orc_match_sim3", "fb_match_sim3", ["matches12", "nfound"])


def test_bow_matchers_large():
    """Key frames made from the 2*nFeatures initialisation frames carry 4000 key points."""
    from fishbirdeyevisualslam_amd import bow_problem as BP
    probs = [BP.make_bow_problem(3200 + i, 4000, 4000) for i in range(2)]
    a, oo, k = BP.bow_args(probs)
    O.call("orc_match_bow", a)
    a2, oh, k2 = BP.bow_args(probs)
    H.call("fb_match_bow", a2)
    np.testing.assert_array_equal(oh["match_f_to_kf"], oo["match_f_to_kf"])
    np.testing.assert_array_equal(oh["nmatches"], oo["nmatches"])
    probs = [M.make_bow_kf_problem(3300 + i, 4000, 3800) for i in range(2)]
    a, oo, k = M.bow_kf_args(probs)
    O.call("orc_match_bow_kf", a)
    a2, oh, k2 = M.bow_kf_args(probs)
    H.call("fb_match_bow_kf", a2)
    np.testing.assert_array_equal(oh["matches12"], oo["matches12"])
    probs = [BP.make_triangulation_problem(3400 + i, 4000, 4000) for i in range(2)]
    a, oo, k = BP.triangulation_args(probs)
    O.call("orc_match_triangulation", a)
    a2, oh, k2 = BP.triangulation_args(probs)
    H.call("fb_match_triangulation", a2)
    np.testing.assert_array_equal(oh["matches12"], oo["matches12"])
    np.testing.assert_array_equal(oh["nmatches"], oo["nmatches"])
