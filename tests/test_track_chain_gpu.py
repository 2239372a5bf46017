"""The sequence-faithful tracking chain on the GPU (fb_frame_*, csrc/track.hip) against the oracle chain
(oracle/track_oracle.cpp): consecutive frames of a synthetic drive, frame k's optimised pose, map-point associations,
outlier flags and bird map feeding frame k+1 -- Frame::Frame (Frame.cc:262-379), TrackWithMotionModel + TrackLocalMap
(Tracking.cc:1312-1441) and the end-of-Track clean-up (:690-725), in that order.

Bar (BASELINE.json): key points, descriptors, match indices (mvpMapPoints / mvpMapPointsBird), outlier masks, every
counter and the ids of newly created MapPointBirds bit-exact; poses and the positions of new bird points <= 1e-4 relative.
PARITY UNPINNED: the reference holds no fixture for this path; the oracle is the build's restatement."""
import numpy as np
import pytest

from fishbirdeyevisualslam_amd import cabi, sequence as S, track as T

pytestmark = pytest.mark.gpu
REL_TOL = 1e-4  # BASELINE.json north_star: pose / landmark estimates within 1e-4 relative


def _diff(g, o, k, b, n):
    a, c = g[k][b, :n], o[k][b, :n]
    idx = np.nonzero((a != c) if a.dtype.names is None else np.array([x != y for x, y in zip(a, c)]))[0][:6]
    d = dict(where=idx.tolist(), gpu=[a[i].tolist() for i in idx], oracle=[c[i].tolist() for i in idx],
                gpu_counts=g["counts"][:12, b].tolist(), oracle_counts=o["counts"][:12, b].tolist(),
                gpu_mp=[int(g["map_point"][b, i]) for i in idx] if k == "outlier" else None,
                pose_rel=float(np.abs(g["Tcw"][b] - o["Tcw"][b]).max()))
    print("MISMATCH", k, b, d)
    return d


def _cmp_view(g, o, tag):
    B = g["n"].shape[0]
    assert np.array_equal(g["n"], o["n"]) and np.array_equal(g["n_bird"], o["n_bird"]), (tag, g["n"], o["n"], g["n_bird"], o["n_bird"])
    worst = 0.0
    for b in range(B):
        n, nb = int(o["n"][b]), int(o["n_bird"][b])
        for k in ("kps", "kps_un", "desc", "map_point", "outlier"):
            assert np.array_equal(g[k][b, :n], o[k][b, :n]), (tag, b, k, _diff(g, o, k, b, n))
        for k in ("kps_bird", "desc_bird", "bird_cam_xyz", "map_point_bird", "bird_outlier"):
            assert np.array_equal(g[k][b, :nb], o[k][b, :nb]), (tag, b, k, _diff(g, o, k, b, nb))
        rel = float(np.abs(g["Tcw"][b] - o["Tcw"][b]).max() / max(1.0, np.abs(o["Tcw"][b]).max()))
        worst = max(worst, rel)
        assert rel <= REL_TOL, (tag, b, rel)
    assert np.array_equal(g["counts"][:16], o["counts"][:16]), (tag, g["counts"][:16].T, o["counts"][:16].T)
    return worst


def _lists(M, MB, seed):
    """mvpLocalMapPoints / vlocalMPB as index lists: a shuffled 85 % of the tables."""
    g = np.random.Generator(np.random.PCG64(seed))
    B, mc = M["bad"].shape
    bc = MB["xw"].shape[1]
    lm, nlm, lb, nlb = np.zeros((B, mc), np.int32), np.zeros(B, np.int32), np.zeros((B, bc), np.int32), np.zeros(B, np.int32)
    for b in range(B):
        p = g.permutation(int(M["n"][b]))[: int(0.85 * M["n"][b])]
        lm[b, : len(p)], nlm[b] = p, len(p)
        p = g.permutation(int(MB["n"][b]))[: int(0.85 * MB["n"][b])]
        lb[b, : len(p)], nlb[b] = p, len(p)
    return (lm, nlm), (lb, nlb)


def _run(B, K, front_wh, bird_wh, fx, use_lists, seed, granular=False, contour=True, pipelined=False, bird_nfeatures=0, check_workload=True):
    from oracle import pyoracle as O
    seq = S.Sequence(B, K, seed=seed, front_wh=front_wh, bird_wh=bird_wh, fx=fx, fy=fx, device="cuda:0")
    tc = T.TrackChain(B, front_wh, bird_wh, K=seq.Kc, D=seq.D, use_lists=use_lists, bird_nfeatures=bird_nfeatures)
    oc = O.OracleChain(tc.params, tc.map_cap, tc.bird_cap, use_lists=use_lists)
    import torch
    mask_d = torch.from_numpy(seq.mask).cuda() if contour else None
    mask_h = seq.mask if contour else None
    f, b, c = seq.render(0)
    if not contour:
        c = None
    tc.extract(f, b, c, mask_d)
    h = lambda t: t.cpu().numpy() if t is not None else None
    oc.extract(h(f), h(b), h(c), mask_h)
    v0, o0 = tc.view("cur"), oc.view("cur")
    _cmp_view(v0, o0, "frame 0")
    M, MB, mp0, mpb0, Tcw0 = seq.build_map(v0, tc.tables, map_cap=tc.map_cap, bird_cap=tc.bird_cap)
    lm, lb = _lists(M, MB, seed + 5) if use_lists else (None, None)
    tc.set_map(M, MB, lm, lb)
    oc.set_map(M, MB, lm, lb)
    tc.init_first(mp0, mpb0, Tcw0)
    oc.init_first(mp0, mpb0, Tcw0)
    worst, stats = 0.0, []
    frames = [seq.render(k) for k in range(K)] if pipelined else None
    if pipelined and not contour:   # (the prefetch below takes its images from this list)
        frames = [(f_, b_, None) for f_, b_, c_ in frames]
    if pipelined:
        tc.prefetch(*frames[1], mask_d)
    for k in range(1, K):
        f, b, c = frames[k] if pipelined else seq.render(k)
        if not contour:
            c = None
        d = seq.delta(k)
        if pipelined:  # frame k+1 is constructed on the extraction stream while frame k is tracked
            if k + 1 < K:
                tc.prefetch(*frames[k + 1], mask_d)
            tc.track_prefetched(torch.from_numpy(d).cuda())
            torch.cuda.synchronize()
        else:
            tc.set_delta(d)
            (tc.track_granular if granular else tc.track)(f, b, c, mask_d)
        oc.track(h(f), h(b), h(c), mask_h, d)
        g, o = tc.view(), oc.view()
        worst = max(worst, _cmp_view(g, o, "frame %d" % k))
        # the frame that was `last` during this step received the new MapPointBirds too (Tracking.cc:1899)
        gl, ol = tc.view("prev"), oc.view("cur")
        for bb in range(B):
            nb = int(ol["n_bird"][bb])
            assert np.array_equal(gl["map_point_bird"][bb, :nb], ol["map_point_bird"][bb, :nb]), ("ref frame bird points", k, bb)
        gt, ot = tc.bird_table_host(), oc.bird_table_host()
        assert np.array_equal(gt["n"], ot["n"]), (k, gt["n"], ot["n"])
        for bb in range(B):
            nt = int(ot["n"][bb])
            assert np.array_equal(gt["desc"][bb, :nt], ot["desc"][bb, :nt]), ("bird table descriptors", k, bb)
            scale = max(1.0, float(np.abs(ot["xw"][bb, :nt]).max()))
            assert np.abs(gt["xw"][bb, :nt] - ot["xw"][bb, :nt]).max() / scale <= REL_TOL
        cnt = o["counts"]
        if check_workload:
            # the chain must be doing real work on every frame: it tracks (Tracking.cc:1384: nmatchesMap >= 10; :1438: >= 30 inliers)
            assert (cnt[cabi.FB_CNT["MATCHES_MAP"]] >= 10).all() and (cnt[cabi.FB_CNT["MATCHES_INLIERS"]] >= 30).all(), cnt[:12].T
            for bb in range(B):
                Tt = np.asarray(seq.Tcw_true(k, bb))[:3, :4].reshape(12)
                assert np.abs(o["Tcw"][bb] - Tt).max() < 0.2, ("tracking drifted from the true pose", k, bb)
        stats.append(cnt[:12, 0].tolist())
    tc.close()
    oc.close()
    return worst, stats


def test_chain_six_frames_full_size():
    """BASELINE configs[2] geometry: 1280x720 front + 512x512 bird, fisheye.yaml distortion, contour + mask, 2 sequences."""
    worst, stats = _run(2, 6, (1280, 720), (512, 512), 500.0, use_lists=False, seed=9000)
    # the bench's configuration: a 1000-feature bird extractor writing into the frame's 2064-entry rows
    worst_b, stats_b = _run(2, 4, (1280, 720), (512, 512), 500.0, use_lists=False, seed=9010, bird_nfeatures=1000)
    worst = max(worst, worst_b)
    print("track chain 1280x720+512x512, 5 tracked frames x 2 sequences: worst relative pose difference %.3g; counters of sequence 0 %s" % (worst, stats))


def test_chain_with_local_lists_small():
    """mvpLocalMapPoints / vlocalMPB given as shuffled index lists; 640x480 + 384x384, 3 sequences, 7 tracked frames."""
    worst, stats = _run(3, 8, (640, 480), (384, 384), 250.0, use_lists=True, seed=9100)
    print("track chain with local lists: worst relative pose difference %.3g" % worst)


def test_chain_granular_entry_points():
    """The one-call-per-reference-function entry points (each commit its own launch) give what the fused call gives."""
    worst, stats = _run(2, 5, (640, 480), (384, 384), 250.0, use_lists=True, seed=9200, granular=True)
    worst2, stats2 = _run(2, 5, (640, 480), (384, 384), 250.0, use_lists=True, seed=9200, granular=False)
    assert stats == stats2


def test_chain_pipelined_driver():
    """Frame construction of frame k+1 on a second stream beside the tracking of frame k (three frame handles): same results."""
    worst, stats = _run(2, 7, (640, 480), (384, 384), 250.0, use_lists=False, seed=9600, pipelined=True)
    worst2, stats2 = _run(2, 7, (640, 480), (384, 384), 250.0, use_lists=False, seed=9600, pipelined=False)
    assert stats == stats2


def test_chain_without_contour():
    """No contour / mask image: every bird key point is kept (GuidenceKeyBirdPts skipped)."""
    _run(1, 4, (640, 480), (384, 384), 250.0, use_lists=False, seed=9300, contour=False)


def test_chain_host_images():
    """fb_frame_extract (host images through the handle's pinned staging) == fb_frame_extract_dev."""
    import ctypes as C
    import torch
    seq = S.Sequence(2, 2, seed=9400, front_wh=(640, 480), bird_wh=(384, 384), fx=250.0, fy=250.0, device="cuda:0")
    tc = T.TrackChain(2, (640, 480), (384, 384), K=seq.Kc, D=seq.D)
    f, b, c = seq.render(0)
    mask_d = torch.from_numpy(seq.mask).cuda()
    tc.extract(f, b, c, mask_d)
    v_dev = tc.view("cur")
    tc.k += 1
    fh, bh, ch = (np.ascontiguousarray(t.cpu().numpy()) for t in (f, b, c))
    vp = lambda a: C.c_void_p(a.ctypes.data)
    for _ in range(2):  # twice: the second call reuses the staging block
        rc = tc.L.fb_frame_extract(tc.cur, tc.orb_f, tc.orb_b, vp(fh), 640, vp(bh), 384, vp(ch), vp(seq.mask), tc._stream())
        assert rc == 0, tc.L.fb_last_error()
    v_host = tc.view("cur")
    assert np.array_equal(v_dev["n"], v_host["n"]) and np.array_equal(v_dev["n_bird"], v_host["n_bird"])
    for b in range(2):
        n, nb = int(v_dev["n"][b]), int(v_dev["n_bird"][b])
        for k in ("kps", "kps_un", "desc", "map_point", "outlier"):
            assert np.array_equal(v_dev[k][b, :n], v_host[k][b, :n]), k
        for k in ("kps_bird", "desc_bird", "bird_cam_xyz", "map_point_bird", "bird_outlier"):
            assert np.array_equal(v_dev[k][b, :nb], v_host[k][b, :nb]), k
    tc.close()


def _yaw(delta12, angle):
    """detlaT with a yaw error: E * detlaT, E = rotation about the camera's y axis."""
    c, s_ = np.cos(angle), np.sin(angle)
    E = np.array([[c, 0, s_], [0, 1, 0], [-s_, 0, c]], np.float64)
    M = delta12.reshape(3, 4).astype(np.float64)
    return (E @ M).astype(np.float32).reshape(12)


def _run_modes(B, K, wh, bwh, fx, seed, modes, rekey_at=(), short_list_seq=None, empty_kf_seq=None, use_lists=True, voc_kL=(5, 5), verbose=False,
               few_points=None, yaw_error=None, defer_drop=False, min_inliers=0):
    """A drive where the caller chooses per frame between TrackWithMotionModel, TrackReferenceKeyFrame and the fall-back
    (modes[k] in "motion" / "reference" / "motion+reference"), TrackLocalMap behind each, new key frames after the frames in
    rekey_at.  Returns what the branches saw."""
    import torch
    from oracle import pyoracle as O
    from fishbirdeyevisualslam_amd.bow_problem import make_vocabulary
    seq = S.Sequence(B, K, seed=seed, front_wh=wh, bird_wh=bwh, fx=fx, fy=fx, device="cuda:0")
    tc = T.TrackChain(B, wh, bwh, K=seq.Kc, D=seq.D, use_lists=use_lists)
    oc = O.OracleChain(tc.params, tc.map_cap, tc.bird_cap, use_lists=use_lists)
    mask_d = torch.from_numpy(seq.mask).cuda()
    h = lambda t: t.cpu().numpy()
    f, b, c = seq.render(0)
    tc.extract(f, b, c, mask_d)
    oc.extract(h(f), h(b), h(c), seq.mask)
    v0 = tc.view("cur")
    M, MB, mp0, mpb0, Tcw0 = seq.build_map(v0, tc.tables, map_cap=tc.map_cap, bird_cap=tc.bird_cap)
    lm, lb = _lists(M, MB, seed + 5) if use_lists else (None, None)
    if short_list_seq is not None and use_lists:
        lb[1][short_list_seq] = 8     # vlocalMPB.size() <= 10
    mp0 = mp0.copy()
    if empty_kf_seq is not None:
        mp0[empty_kf_seq, :] = -1     # the first key frame of this sequence has no map points
    for bb_, npts_ in (few_points or {}).items():   # a first frame with only npts_ map points
        keep_ = np.nonzero(mp0[bb_] >= 0)[0][npts_:]
        mp0[bb_, keep_] = -1
    vv, vk, first_leaf = make_vocabulary(seed + 1, k=voc_kL[0], L=voc_kL[1])
    tc.set_vocabulary(vk, voc_kL[1])
    oc.set_vocabulary(vk, voc_kL[1])
    tc.set_map(M, MB, lm, lb)
    oc.set_map(M, MB, lm, lb)
    tc.init_first(mp0, mpb0, Tcw0)
    oc.init_first(mp0, mpb0, Tcw0)
    tc.make_keyframe("last")
    oc.make_keyframe("last")
    if min_inliers:  # TrackLocalMap's success threshold (30; 50 shortly after a relocalisation, Tracking.cc:1435-1438)
        for ch in (tc, oc):
            cabi.fill(ch.targs, min_inliers=min_inliers)
        cabi.fill(tc.targs_kf, min_inliers=min_inliers)
    if defer_drop:   # the host creates its key frame between the clean-up and the outlier drop (Tracking.cc:716-725)
        for ch in (tc, oc):
            cabi.fill(ch.targs, defer_outlier_drop=1)
        cabi.fill(tc.targs_kf, defer_outlier_drop=1)
    kf_at, seen, worst = 0, dict(ref_ok=0, gated=0, bird_branch=0, retried=0, below20=0), 0.0
    for k in range(1, K):
        f, b, c = seq.render(k)
        d, dk = seq.delta(k), seq.delta_between(kf_at, k)
        if yaw_error and k in yaw_error:   # a bad odometry increment for one sequence: the th = 15 window misses, 2 * th finds
            bb_, ang_ = yaw_error[k]
            d = d.copy()
            d[bb_] = _yaw(d[bb_], ang_)
        tc.set_delta(d)
        tc.set_delta_kf(dk)
        tc.track_modes(f, b, c, mask_d, mode=modes[k])
        oc.track_modes(h(f), h(b), h(c), seq.mask, d, dk, mode=modes[k])
        g, o = tc.view(), oc.view()
        worst = max(worst, _cmp_view(g, o, "frame %d (%s)" % (k, modes[k])))
        if defer_drop:
            seen["kept_outliers"] = seen.get("kept_outliers", 0) + int(((o["map_point"] >= 0) & (o["outlier"] != 0)).sum())
            tc.drop_outliers()
            oc.drop_outliers()
            g, o = tc.view(), oc.view()
            _cmp_view(g, o, "frame %d after the outlier drop" % k)
            assert not ((o["map_point"] >= 0) & (o["outlier"] != 0) & (o["counts"][cabi.FB_CNT["MATCHES_INLIERS"]] >= (min_inliers or 30))[:, None]).any(), "outliers left after the drop"
        gt, ot = tc.bird_table_host(), oc.bird_table_host()
        assert np.array_equal(gt["n"], ot["n"])
        for bb in range(B):
            nt = int(ot["n"][bb])
            assert np.array_equal(gt["desc"][bb, :nt], ot["desc"][bb, :nt]), ("bird table descriptors", k, bb)
        cnt = o["counts"]
        if "reference" in modes[k]:
            bow, pts = cnt[cabi.FB_CNT["BOW_MATCHES"]], cnt[cabi.FB_CNT["BIRD_POINTS"]]
            seen["ref_ok"] += int((bow >= 15).sum()); seen["gated"] += int((bow < 15).sum()); seen["bird_branch"] += int((pts < 10).sum())
        if "motion" in modes[k]:
            seen["retried"] += int(cnt[cabi.FB_CNT["PROJ_RETRIED"]].sum()); seen["below20"] += int((cnt[cabi.FB_CNT["PROJ_MATCHES"]] < 20).sum())
        if verbose:
            print("frame", k, modes[k], cnt[:15].T.tolist())
        if k in rekey_at:             # CreateNewKeyFrame from the frame just tracked
            tc.make_keyframe("last")
            oc.make_keyframe("last")
            kf_at = k
            _cmp_view(tc.view("kf"), oc.view("kf"), "key frame")
    tc.close()
    oc.close()
    return seen, worst


def test_chain_reference_keyframe_path():
    """Tracking::TrackReferenceKeyFrame (Tracking.cc:1180-1244) on frame handles -- SetPose from the key frame, GetLocalMapForBird,
    the numPt < 10 branch, ComputeBoW, SearchByBoW(0.7), the < 15 early return, pose optimisation, discard -- alone, and as the
    fall-back after TrackWithMotionModel (Tracking.cc:535-540), with TrackLocalMap behind it and a new key frame mid-drive.
    Sequence 1 has a short vlocalMPB list (<= 10 entries: GetLocalMapForBird matches nothing, so the per-frame bird match
    runs inside TrackReferenceKeyFrame); sequence 2's first key frame holds no map points (SearchByBoW returns 0: the
    early return).  The vocabulary is a synthetic 5-ary tree of depth 5 (the reference ships none)."""
    modes = {1: "reference", 2: "motion", 3: "motion+reference", 4: "reference", 5: "motion", 6: "motion+reference"}
    seen, worst = _run_modes(3, 7, (640, 480), (384, 384), 250.0, 9700, modes, rekey_at=(3,), short_list_seq=1, empty_kf_seq=2, verbose=True)
    # every branch was exercised
    assert seen["ref_ok"] >= 6 and seen["gated"] >= 1 and seen["bird_branch"] >= 1, seen
    print("reference-key-frame path: %s, worst relative pose difference %.3g" % (seen, worst))


def test_chain_track_using_bird():
    """Tracking::TrackUsingBird (Tracking.cc:2014-2061), the frame of a lost tracker: pose from the last frame or from the
    reference key frame, GetLocalMapForBird, the numPt <= 10 branch (sequence 1: a short vlocalMPB list), BirdOptimization
    (bird edges only), GetPerFrameMatchedBirdPoints; then the tracker finds its way back through the reference key frame."""
    modes = {1: "motion", 2: "bird", 3: "bird_kf", 4: "reference", 5: "bird", 6: "motion+reference"}
    seen, worst = _run_modes(2, 7, (640, 480), (384, 384), 250.0, 9900, modes, rekey_at=(1,), short_list_seq=1, verbose=True, defer_drop=True)
    assert seen.get("kept_outliers", 0) > 0, seen   # the deferred drop had something to drop
    print("track-using-bird path: worst relative pose difference %.3g; outliers kept for the key frame: %d" % (worst, seen["kept_outliers"]))


def test_frame_bow_entry_points_against_the_array_api():
    """fb_frame_compute_bow_dev / fb_frame_bow_view_dev / fb_frame_search_by_bow_dev on frame handles give what the array
    entry points (fb_bow_transform, fb_match_bow: oracle-checked in test_bow_transform.py / test_bow_gpu.py) give on the
    frames' downloaded members; a second ComputeBoW on the same frame is a no-op; min_matches gates the commit."""
    import ctypes as C
    import torch
    from fishbirdeyevisualslam_amd.bow_problem import make_vocabulary
    from test_bow_transform import make_args
    B, wh, bwh = 2, (640, 480), (384, 384)
    seq = S.Sequence(B, 3, seed=9800, front_wh=wh, bird_wh=bwh, fx=250.0, fy=250.0, device="cuda:0")
    tc = T.TrackChain(B, wh, bwh, K=seq.Kc, D=seq.D)
    L = tc.L
    mask_d = torch.from_numpy(seq.mask).cuda()
    f, b, c = seq.render(0)
    tc.extract(f, b, c, mask_d)
    v0 = tc.view("cur")
    M, MB, mp0, mpb0, Tcw0 = seq.build_map(v0, tc.tables, map_cap=tc.map_cap, bird_cap=tc.bird_cap)
    vv, vk, first_leaf = make_vocabulary(9801, k=6, L=5)
    tc.set_vocabulary(vk, 5)
    tc.set_map(M, MB)
    tc.init_first(mp0, mpb0, Tcw0)
    tc.make_keyframe("last")
    f, b, c = seq.render(1)
    tc.extract(f, b, c, mask_d)          # the current frame, nothing tracked yet
    s = tc._stream()
    assert L.fb_frame_compute_bow_dev(tc.cur, C.byref(tc.voc), s) == 0
    vcur, vkf = tc.view("cur"), tc.view("kf")
    cap = tc.cap
    hip = C.CDLL("libamdhip64.so")

    def bow_of(frame):
        view = cabi.BowTransformArgs()
        assert L.fb_frame_bow_view_dev(frame, C.byref(view)) == 0, L.fb_last_error()
        torch.cuda.synchronize()
        shapes = dict(n_words=((B,), np.int32), bow_ids=((B, cap), np.uint32), bow_vals=((B, cap), np.float64), fv_n_nodes=((B,), np.int32),
                      fv_node_ids=((B, cap), np.uint32), fv_node_start=((B, cap + 1), np.int32), fv_items=((B, cap), np.int32))
        out = {}
        for k, (shp, dt) in shapes.items():
            out[k] = np.zeros(shp, dt)
            assert hip.hipMemcpy(C.c_void_p(out[k].ctypes.data), C.c_void_p(getattr(view, k)), C.c_size_t(out[k].nbytes), 2) == 0
        return out
    for frame, vw in ((tc.cur, vcur), (tc.kf, vkf)):
        got = bow_of(frame)
        a, ref, keep = make_args([vw["desc"][bb, : vw["n"][bb]] for bb in range(B)], levelsup=4)
        # (make_args pads to the longest frame; the handle's arrays have kp_stride entries per frame)
        fb_lib = L
        assert fb_lib.fb_bow_transform(C.byref(vv), C.byref(a)) == 0, L.fb_last_error()
        for bb in range(B):
            nw, nn = int(ref["n_words"][bb]), int(ref["fv_n_nodes"][bb])
            assert got["n_words"][bb] == nw and got["fv_n_nodes"][bb] == nn
            assert np.array_equal(got["bow_ids"][bb, :nw], ref["bow_ids"][bb, :nw]) and np.array_equal(got["bow_vals"][bb, :nw], ref["bow_vals"][bb, :nw])
            assert np.array_equal(got["fv_node_ids"][bb, :nn], ref["fv_node_ids"][bb, :nn])
            assert np.array_equal(got["fv_node_start"][bb, : nn + 1], ref["fv_node_start"][bb, : nn + 1])
            ni = int(ref["fv_node_start"][bb, nn])
            assert np.array_equal(got["fv_items"][bb, :ni], ref["fv_items"][bb, :ni])
    # a second ComputeBoW is a no-op (if (mBowVec.empty())): poison one value, call again, it stays
    view = cabi.BowTransformArgs()
    assert L.fb_frame_bow_view_dev(tc.cur, C.byref(view)) == 0
    poison = np.array([-123], np.int32)
    assert hip.hipMemcpy(C.c_void_p(view.n_words), C.c_void_p(poison.ctypes.data), C.c_size_t(4), 1) == 0
    assert L.fb_frame_compute_bow_dev(tc.cur, C.byref(tc.voc), s) == 0
    assert bow_of(tc.cur)["n_words"][0] == -123
    tc.extract(f, b, c, mask_d)          # a new Frame: BoW empty again
    view2 = cabi.BowTransformArgs()
    assert L.fb_frame_bow_view_dev(tc.cur, C.byref(view2)) != 0   # no BoW yet
    assert L.fb_frame_search_by_bow_dev(tc.cur, tc.kf, C.byref(tc.targs.map), C.byref(cabi.MatcherParams(0.7, 1)), 15, s) != 0
    assert L.fb_frame_compute_bow_dev(tc.cur, C.byref(tc.voc), s) == 0
    gcur = bow_of(tc.cur)
    gkf = bow_of(tc.kf)
    # ---- SearchByBoW on the handles vs fb_match_bow on the downloaded members
    m07 = cabi.MatcherParams(0.7, 1)
    assert L.fb_frame_search_by_bow_dev(tc.cur, tc.kf, C.byref(tc.targs.map), C.byref(m07), 0, s) == 0, L.fb_last_error()
    after = tc.view("cur")
    bad = M["bad"]
    a = cabi.BowArgs()
    has_mp = ((vkf["map_point"] >= 0) & (np.take_along_axis(bad, np.maximum(vkf["map_point"], 0), 1) == 0)).astype(np.uint8)
    for bb in range(B):
        has_mp[bb, vkf["n"][bb]:] = 0
    keep = dict(n_kf=vkf["n"].copy(), kf_kps=vkf["kps_un"].copy(), kf_desc=vkf["desc"].copy(), kf_has_mp=has_mp,
                n_f=vcur["n"].copy(), f_kps=vcur["kps"].copy(), f_desc=vcur["desc"].copy(),
                match_f_to_kf=np.full((B, cap), -7, np.int32), nmatches=np.zeros(B, np.int32))
    cabi.fill(a, batch=B, kf_stride=cap, f_stride=cap, **keep)
    fvk = {k: np.ascontiguousarray(gkf[k]) for k in ("fv_n_nodes", "fv_node_ids", "fv_node_start", "fv_items")}
    fvc = {k: np.ascontiguousarray(gcur[k]) for k in ("fv_n_nodes", "fv_node_ids", "fv_node_start", "fv_items")}
    cabi.fill(a.kf_fv, node_stride=cap, item_stride=cap, n_nodes=fvk["fv_n_nodes"], node_ids=fvk["fv_node_ids"], node_start=fvk["fv_node_start"], items=fvk["fv_items"])
    cabi.fill(a.f_fv, node_stride=cap, item_stride=cap, n_nodes=fvc["fv_n_nodes"], node_ids=fvc["fv_node_ids"], node_start=fvc["fv_node_start"], items=fvc["fv_items"])
    a.matcher = m07
    assert L.fb_match_bow(C.byref(a)) == 0, L.fb_last_error()
    cnt, _ = tc.counts("cur")
    assert np.array_equal(cnt[cabi.FB_CNT["BOW_MATCHES"]], keep["nmatches"]) and (keep["nmatches"] >= 15).all(), (cnt[cabi.FB_CNT["BOW_MATCHES"]], keep["nmatches"])
    for bb in range(B):
        n = int(vcur["n"][bb])
        m = keep["match_f_to_kf"][bb, :n]
        expect = np.where(m >= 0, vkf["map_point"][bb][np.maximum(m, 0)], -1)
        assert np.array_equal(after["map_point"][bb, :n], expect)
    # ---- min_matches above the count: the sequences keep their mvpMapPoints (here: set to a marker first)
    marker = np.full((B, cap), 5, np.int32)
    md = torch.from_numpy(marker).cuda()
    assert L.fb_frame_set_map_points_dev(tc.cur, C.c_void_p(md.data_ptr()), None, s) == 0
    assert L.fb_frame_search_by_bow_dev(tc.cur, tc.kf, C.byref(tc.targs.map), C.byref(m07), 100000, s) == 0
    kept = tc.view("cur")
    for bb in range(B):
        n = int(vcur["n"][bb])
        assert (kept["map_point"][bb, :n] == 5).all()
    tc.close()


def test_chain_wide_window_retry_and_early_return():
    """Tracking.cc:1342-1352 per sequence inside the fused chain: a sequence whose first SearchByProjection (th = 15) finds fewer
    than 20 matches is searched again with 2 * th in the same launch and goes on when that finds 20 (sequence 0: 80 map points
    and a 3.4-degree yaw error in its first odometry increment); one that stays below 20 "returns false" -- matches committed,
    pose and flags untouched, no discard (sequence 1: 12 map points in its first frame)."""
    modes = {1: "motion", 2: "motion", 3: "motion"}
    seen, worst = _run_modes(2, 4, (640, 480), (384, 384), 250.0, 9950, modes, few_points={0: 80, 1: 12}, yaw_error={1: (0, 0.06)}, verbose=True)
    assert seen["retried"] >= 2 and seen["below20"] >= 1, seen


def test_chain_lost_frames_keep_their_members():
    """A frame whose TrackLocalMap fails (mnMatchesInliers below the threshold, here an impossible one) is a LOST frame: the
    block of Tracking.cc:681-726 does not run, so it keeps its outliers and its points without observations."""
    modes = {1: "motion", 2: "motion", 3: "motion"}
    seen, worst = _run_modes(2, 4, (640, 480), (384, 384), 250.0, 9960, modes, min_inliers=100000, defer_drop=True, verbose=True)
    assert seen.get("kept_outliers", 0) > 0, seen
