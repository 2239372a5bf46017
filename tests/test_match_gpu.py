"""GPU parity: HIP matchers (through the C-ABI) vs the CPU oracle, bit-exact (integer work)."""
import ctypes as C

import numpy as np
import pytest

import hip_lib as H
import oracle_lib as O
from fishbirdeyevisualslam_amd import cabi, problems as P, synth

pytestmark = pytest.mark.gpu


def _both(build, oracle_name, hip_name, outs):
    a, out_o, keep = build()
    O.call(oracle_name, a)
    a2, out_h, keep2 = build()
    H.call(hip_name, a2)
    for k in outs:
        np.testing.assert_array_equal(out_h[k], out_o[k], err_msg=k)
    return out_o


def test_descriptor_distance_known_answers():
    import fishbirdeyevisualslam_amd as fb
    g = synth.rng(7)
    a = synth.random_descriptors(g, 4096)
    b = synth.random_descriptors(g, 4096)
    b[0] = a[0]                      # distance 0
    a[1] = 0; b[1] = 255             # distance 256
    b[2] = a[2]; b[2, 5] ^= 0x10     # single bit
    out = np.zeros(4096, np.int32)
    fb.check(fb.lib().fb_descriptor_distance(C.c_void_p(a.ctypes.data), C.c_void_p(b.ctypes.data), 4096,
                                             C.c_void_p(out.ctypes.data)), "dd")
    ref = np.unpackbits(a ^ b, axis=1).sum(1)
    np.testing.assert_array_equal(out, ref)
    assert out[0] == 0 and out[1] == 256 and out[2] == 1
    oo = np.zeros(4096, np.int32)
    O.lib().orc_descriptor_distance(C.c_void_p(a.ctypes.data), C.c_void_p(b.ctypes.data), 4096, C.c_void_p(oo.ctypes.data))
    np.testing.assert_array_equal(out, oo)


@pytest.mark.parametrize("n", [0, 1, 17, 2000])
def test_grid_build_matches_oracle(n):
    g = synth.rng(11 + n)
    ks = [synth.random_keypoints(g, n, 1280, 720), synth.random_keypoints(g, max(n - 1, 0), 1280, 720)]
    ks[0]["x"][: n // 50] = -5.0  # out of the grid -> dropped
    geom = P.grid_geom(synth.front_grid_geom(1280, 720))
    stride = max(n, 1)
    cs_o, ci_o = P.build_grid_host(ks, geom, O.grid_build, stride)
    cs_h, ci_h = P.build_grid_host(ks, geom, H.grid_build, stride)
    np.testing.assert_array_equal(cs_h, cs_o)
    for b in range(2):
        np.testing.assert_array_equal(ci_h[b, : cs_o[b, -1]], ci_o[b, : cs_o[b, -1]])


@pytest.mark.parametrize("seed,ncur,nlast", [(2000, 2000, 2000), (2001, 2064, 1500), (2002, 300, 900), (2003, 50, 0)])
def test_search_by_projection_frame(seed, ncur, nlast):
    probs = [synth.make_proj_frame_problem(seed + 10 * i, ncur, nlast, dup_frac=0.3 if i else 0.05) for i in range(3)]
    geom = P.grid_geom(synth.front_grid_geom(1280, 720))
    cs, ci = P.build_grid_host([p["cur_kps"] for p in probs], geom, O.grid_build, ncur)
    for ori in (1, 0):
        out = _both(lambda: P.proj_frame_args(probs, cs, ci, check_ori=ori), "orc_match_projection_frame",
                    "fb_match_projection_frame", ["match_cur_to_last", "nmatches"])
        if nlast >= 900:
            assert out["nmatches"].min() > 50


def test_search_by_projection_frame_contention():
    """Many queries compete for few targets: exercises the serial 'already taken' rule."""
    probs = [synth.make_proj_frame_problem(2500 + i, 120, 2000, dup_frac=0.0, obs0_frac=0.3) for i in range(2)]
    geom = P.grid_geom(synth.front_grid_geom(1280, 720))
    cs, ci = P.build_grid_host([p["cur_kps"] for p in probs], geom, O.grid_build, 120)
    _both(lambda: P.proj_frame_args(probs, cs, ci, th=60.0), "orc_match_projection_frame", "fb_match_projection_frame",
          ["match_cur_to_last", "nmatches"])


@pytest.mark.parametrize("pool,th", [(12, 60.0), (40, 30.0), (3, 90.0)])
def test_search_by_projection_frame_equal_distances(pool, th):
    """Descriptors drawn from a small pool: a query sees many candidates at EXACTLY the same Hamming distance, and the
    first one in grid-walk order must win (ORBmatcher.cc:1411-1421: strict <), also for the candidates a query falls back
    to after its best one was taken by an earlier query.  (Real images produce such ties; random descriptors never do:
    this is the case a drive of the tracking chain found in round 3 -- the cached candidate list had lost the walk
    order among equal distances.)"""
    probs = []
    for i in range(3):
        p = synth.make_proj_frame_problem(2600 + 7 * i + pool, 1500, 1500, dup_frac=0.4)
        g = synth.rng(2700 + i + pool)
        base = synth.random_descriptors(g, pool)
        p["cur_desc"] = base[g.integers(0, pool, len(p["cur_desc"]))].copy()
        q = base[g.integers(0, pool, len(p["last_desc"]))].copy()
        flip = g.random(len(q)) < 0.5
        q[flip, 0] ^= 1                       # half of the queries one bit away: ties at distance 0 and at distance 1
        p["last_desc"] = q
        probs.append(p)
    geom = P.grid_geom(synth.front_grid_geom(1280, 720))
    cs, ci = P.build_grid_host([p["cur_kps"] for p in probs], geom, O.grid_build, 1500)
    for ori in (1, 0):
        out = _both(lambda: P.proj_frame_args(probs, cs, ci, th=th, check_ori=ori), "orc_match_projection_frame",
                    "fb_match_projection_frame", ["match_cur_to_last", "nmatches"])
        assert out["nmatches"].min() > 30


@pytest.mark.parametrize("seed,ncur,nref", [(2100, 1000, 1000), (2101, 2064, 700), (2102, 64, 300)])
def test_bird_mappoint_match(seed, ncur, nref):
    probs = [synth.make_bird_mp_problem(seed + 10 * i, ncur, nref) for i in range(3)]
    geom = P.grid_geom(synth.bird_grid_geom(512, 512))
    cs, ci = P.build_grid_host([p["cur_kps"] for p in probs], geom, O.grid_build, ncur)
    out = _both(lambda: P.bird_mp_args(probs, cs, ci, prefill=-1), "orc_match_bird_mappoints", "fb_match_bird_mappoints",
                ["match_cur_to_ref", "ninliers"])
    if nref >= 700:
        assert out["ninliers"].min() > 20
    # pre-filled entries survive unless overwritten
    _both(lambda: P.bird_mp_args(probs, cs, ci, prefill=12345), "orc_match_bird_mappoints", "fb_match_bird_mappoints",
          ["match_cur_to_ref", "ninliers"])


@pytest.mark.parametrize("seed,ncur,nmp,th", [(2200, 2000, 3000, 1.0), (2201, 2064, 800, 5.0), (2202, 100, 2500, 5.0)])
def test_search_by_projection_points(seed, ncur, nmp, th):
    probs = [synth.make_proj_points_problem(seed + 10 * i, ncur, nmp) for i in range(3)]
    geom = P.grid_geom(synth.front_grid_geom(1280, 720))
    cs, ci = P.build_grid_host([p["cur_kps"] for p in probs], geom, O.grid_build, ncur)
    out = _both(lambda: P.proj_points_args(probs, cs, ci, th=th), "orc_match_projection_points",
                "fb_match_projection_points", ["match_cur_to_mp", "nmatches"])
    assert out["nmatches"].min() > 10


@pytest.mark.parametrize("seed,ncur,nref", [(2300, 1000, 1000), (2301, 2064, 500)])
def test_birdview_match(seed, ncur, nref):
    probs = [synth.make_birdview_problem(seed + 10 * i, ncur, nref) for i in range(3)]
    geom = P.grid_geom(synth.bird_grid_geom(512, 512))
    cs, ci = P.build_grid_host([p["cur_kps"] for p in probs], geom, O.grid_build, ncur)
    for ori in (1, 0):
        out = _both(lambda: P.birdview_args(probs, cs, ci, check_ori=ori), "orc_match_birdview", "fb_match_birdview",
                    ["match_ref_to_cur", "match_dist", "nmatches", "n_dmatches"])
        assert out["nmatches"].min() > 10


def test_one_kernel_version_of_search_local_points():
    """The host-pointer entry point of M2 runs the two-phase matcher (candidate lists by many workgroups, then the serial rule
    on the lists); FB_M2_ONE_KERNEL (read once per process, hence a child process) selects the one-workgroup-per-frame
    kernel a _dev caller without a workspace gets.  Same parity bar."""
    import os, subprocess, sys
    env = dict(os.environ, FB_M2_ONE_KERNEL="1")
    here = os.path.dirname(os.path.abspath(__file__))
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.join(here, "test_match_gpu.py"), "-q", "-x", "-m", "gpu",
                        "-k", "search_by_projection_points", "-p", "no:cacheprovider"],
                       env=env, cwd=os.path.dirname(here), capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert " passed" in r.stdout


def test_search_by_projection_frame_wide_window_retry():
    """retry_below / retry_th: a frame whose search finds fewer matches than retry_below is searched again from scratch with
    retry_th inside the same launch (Tracking.cc:1342-1349); the others keep their first result.  Expected = the oracle called
    twice, per frame."""
    sizes = [(2000, 2000), (400, 300), (2064, 150), (120, 90)]
    probs = [synth.make_proj_frame_problem(2600 + 10 * i, nc, nl, dup_frac=0.1) for i, (nc, nl) in enumerate(sizes)]
    ncur = max(nc for nc, _ in sizes)
    geom = P.grid_geom(synth.front_grid_geom(1280, 720))
    cs, ci = P.build_grid_host([p["cur_kps"] for p in probs], geom, O.grid_build, ncur)
    first, second = {}, {}
    for th, store in ((4.0, first), (15.0, second)):
        a, out, keep = P.proj_frame_args(probs, cs, ci, th=th, cur_stride=ncur)
        O.call("orc_match_projection_frame", a)
        store.update(match=out["match_cur_to_last"].copy(), n=out["nmatches"].copy())
    R = int(np.sort(first["n"])[len(sizes) // 2]) + 1     # half of the frames retry
    assert (first["n"] < R).any() and (first["n"] >= R).any(), first["n"]
    a, out, keep = P.proj_frame_args(probs, cs, ci, th=4.0, cur_stride=ncur)
    retried = np.full(len(sizes), -7, np.int32)
    cabi.fill(a, retry_below=R, retry_th=15.0, retried=retried)
    H.call("fb_match_projection_frame", a)
    for b in range(len(sizes)):
        src = second if first["n"][b] < R else first
        assert retried[b] == (1 if first["n"][b] < R else 0)
        assert out["nmatches"][b] == src["n"][b]
        n = len(probs[b]["cur_kps"])
        np.testing.assert_array_equal(out["match_cur_to_last"][b, :n], src["match"][b, :n])
