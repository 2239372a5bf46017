#!/usr/bin/env python3
"""Generates the golden fixtures of tests/golden/ with the CPU oracle (oracle/), i.e. by the build's own
restatement: the reference cannot be built or run here (no OpenCV/Eigen) and ships no fixtures of its own.

    python tests/golden/make_golden.py

Inputs are regenerated from seeds by fishbirdeyevisualslam_amd.synth; only small inputs (one 160x120 image)
and the expected OUTPUTS are stored."""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import oracle_lib as O  # noqa: E402
from fishbirdeyevisualslam_amd import ba_problem, cabi, problems as P, synth  # noqa: E402


def golden_cases():
    out = {}
    # 1. ORB extraction of a 160x120 synthetic image, 300 features, 4 levels
    img = synth.synth_image(1234, 160, 120, n_rect=14, n_disc=8)
    params = O.orb_params(nfeatures=300, nlevels=4)
    k, d = O.orb_extract(params, img)
    out["orb"] = dict(image=img, kps=k.view(np.uint8).reshape(len(k), 24), desc=d)
    # 2. matchers on seeded problems
    probs = [synth.make_proj_frame_problem(2000, 400, 400, dup_frac=0.2, obs0_frac=0.2)]
    geom = P.grid_geom(synth.front_grid_geom(1280, 720))
    cs, ci = P.build_grid_host([p["cur_kps"] for p in probs], geom, O.grid_build, 400)
    a, o, keep = P.proj_frame_args(probs, cs, ci)
    O.call("orc_match_projection_frame", a)
    out["m3"] = dict(match=o["match_cur_to_last"], n=o["nmatches"])
    probs = [synth.make_bird_mp_problem(2100, 300, 300)]
    geom = P.grid_geom(synth.bird_grid_geom(512, 512))
    cs, ci = P.build_grid_host([p["cur_kps"] for p in probs], geom, O.grid_build, 300)
    a, o, keep = P.bird_mp_args(probs, cs, ci)
    O.call("orc_match_bird_mappoints", a)
    out["m9"] = dict(match=o["match_cur_to_ref"], n=o["ninliers"])
    probs = [synth.make_proj_points_problem(2200, 400, 600)]
    geom = P.grid_geom(synth.front_grid_geom(1280, 720))
    cs, ci = P.build_grid_host([p["cur_kps"] for p in probs], geom, O.grid_build, 400)
    a, o, keep = P.proj_points_args(probs, cs, ci, th=5.0)
    O.call("orc_match_projection_points", a)
    out["m2"] = dict(match=o["match_cur_to_mp"], n=o["nmatches"])
    probs = [synth.make_birdview_problem(2300, 300, 300)]
    geom = P.grid_geom(synth.bird_grid_geom(512, 512))
    cs, ci = P.build_grid_host([p["cur_kps"] for p in probs], geom, O.grid_build, 300)
    a, o, keep = P.birdview_args(probs, cs, ci)
    O.call("orc_match_birdview", a)
    out["m8"] = dict(match=o["match_ref_to_cur"], dist=o["match_dist"], n=o["nmatches"], nd=o["n_dmatches"])
    # 3. pose optimisation, the three modes
    pp = [synth.make_pose_problem(3000, 300, 120)]
    for mode in (0, 1, 2):
        a, o, keep = P.pose_args(pp, mode=mode)
        O.call("orc_pose_opt", a)
        out["pose%d" % mode] = dict(Tcw=o["Tcw"], front_outlier=o["front_outlier"], bird_outlier=o["bird_outlier"], n=o["ninliers"])
    # 4. local BA (reduced: 6 keyframes x 500 points)
    bp = synth.make_ba_problem(4000, n_kf=6, n_mp=500, n_mpb=120)
    for wo in (1, 0):
        a, o, keep = ba_problem.local_ba_args(bp, with_odom=wo)
        O.call("orc_local_ba", a)
        out["ba%d" % wo] = dict(kf_Tcw=o["kf_Tcw"], mp_xw=o["mp_xw"], mpb_xw=o["mpb_xw"], obs_outlier=o["obs_outlier"],
                                bobs_outlier=o["bobs_outlier"])
    out.update(more_cases(O))
    return out


def more_cases(L, prefix="orc_", call=None, grid_fn=None):
    """Relocalisation / loop-closing matchers and Frame geometry; `L`/`prefix` pick the library (oracle or HIP)."""
    from fishbirdeyevisualslam_amd import more_problems as M
    call = call or O.call
    out = {}
    probs = [M.make_proj_kf_problem(7000, 400, 500)]
    geom = P.grid_geom(synth.front_grid_geom(1280, 720))
    cs, ci = P.build_grid_host([p["cur_kps"] for p in probs], geom, grid_fn or O.grid_build, 400)
    a, o, keep = M.proj_kf_args(probs, cs, ci)
    call(("orc_" if prefix == "orc_" else "fb_") + "match_projection_keyframe", a)
    out["m4"] = dict(match=o["match_cur_to_kf"], n=o["nmatches"])
    a, o, keep = M.bow_kf_args([M.make_bow_kf_problem(7100, 400, 400)])
    call(("orc_" if prefix == "orc_" else "fb_") + "match_bow_kf", a)
    out["m6"] = dict(match=o["matches12"], n=o["nmatches"])
    a, o, keep = M.frustum_args([M.make_frustum_problem(7200, 600)])
    call(("orc_" if prefix == "orc_" else "fb_") + "in_frustum", a)
    v = o["in_view"] == 1
    out["frustum"] = dict(in_view=o["in_view"], proj=o["proj"][v], proj_xr=o["proj_xr"][v], level=o["level"][v],
                          view_cos=o["view_cos"][v])
    kps = synth.random_keypoints(synth.rng(7300), 500, 1280, 720)
    lib = L.lib() if hasattr(L, "lib") else L
    u = M.undistort(lib, prefix, kps)
    out["undistort"] = dict(xy=np.stack([u["x"], u["y"]], 1), bounds=M.image_bounds(lib, prefix, 1280, 720))
    out.update(kf_cases(lib, prefix, call, grid_fn or O.grid_build))
    return out


def kf_cases(lib, prefix, call, grid_fn):
    """Key-frame side matcher entry points + ComputeDistinctiveDescriptors."""
    import ctypes as C
    from fishbirdeyevisualslam_amd import kf_problems as KP
    pre = "orc_" if prefix == "orc_" else "fb_"
    geom = P.grid_geom(synth.front_grid_geom(KP.W, KP.H))
    out = {}
    for sim3, name in ((False, "fuse_search"), (True, "fuse_sim3_search")):
        probs = [KP.make_kf_points_problem(8000, 400, 600, sim3)]
        cs, ci = P.build_grid_host([p["kf_kps"] for p in probs], geom, grid_fn, 400)
        a, o, keep = KP.fuse_args(probs, cs, ci)
        call(pre + name, a)
        out[name] = dict(best_idx=o["best_idx"])
    a, o, keep = KP.proj_sim3_args(probs, cs, ci)
    call(pre + "match_projection_sim3", a)
    out["proj_sim3"] = dict(match=o["match_kf_to_mp"], n=o["nmatches"])
    probs = [KP.make_sim3_problem(8100, 400, 400, 250)]
    g1 = P.build_grid_host([p["kps1"] for p in probs], geom, grid_fn, 400)
    g2 = P.build_grid_host([p["kps2"] for p in probs], geom, grid_fn, 400)
    a, o, keep = KP.sim3_args(probs, g1, g2)
    call(pre + "match_sim3", a)
    out["sim3"] = dict(match=o["matches12"], n=o["nfound"])
    probs = [KP.make_init_problem(8200, 600, 600)]
    cs, ci = P.build_grid_host([p["kps2"] for p in probs], geom, grid_fn, 600)
    a, o, keep = KP.init_args(probs, cs, ci)
    call(pre + "match_initialization", a)
    out["init_match"] = dict(match=o["matches12"], n=o["nmatches"], prev=o["prev_matched"])
    start, desc = KP.make_distinctive_problem(8300, 300, max_obs=12, big=1)
    best = np.zeros(300, np.int32)
    rc = getattr(lib, prefix + "distinctive_descriptors")(C.c_void_p(start.ctypes.data), C.c_void_p(desc.ctypes.data), 300,
                                                         C.c_void_p(best.ctypes.data))
    assert rc == 0
    out["distinctive"] = dict(best=best)
    return out


if __name__ == "__main__":
    for name, d in golden_cases().items():
        np.savez_compressed(os.path.join(HERE, name + ".npz"), **d)
        print(name, {k: v.shape for k, v in d.items()})
