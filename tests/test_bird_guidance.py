"""Frame::GuidenceKeyBirdPts / nearEdges / genEdgesPC (src/Frame.cc:671-739) and the bird key -> camera map (Frame.cc:365-373):
known answers of the oracle derived by hand from the reference text (x-as-row quirk, cols/rows clamps, >= 10 rule), then the
HIP kernels against the oracle through the C-ABI, bit for bit."""
import ctypes as C

import numpy as np
import pytest

import oracle_lib as O
from fishbirdeyevisualslam_amd import cabi, synth
from fishbirdeyevisualslam_amd.cabi import fill


def kp_array(xy):
    k = np.zeros(len(xy), cabi.KP_DTYPE)
    k["x"] = [p[0] for p in xy]
    k["y"] = [p[1] for p in xy]
    k["size"] = 31.0
    k["octave"] = np.arange(len(xy)) % 8
    k["response"] = np.arange(len(xy)) + 20
    return k


def make_args(contours, kps_list, descs=None, masks=None, edge_cap=0, stride=None):
    B = len(contours)
    rows, cols = contours[0].shape
    pitch = cols + 5                                     # a pitch that is not the width
    ks = stride or max(max(len(k) for k in kps_list), 1)
    img = np.zeros((B, rows, pitch), np.uint8)
    img[:, :, cols:] = 255                               # padding that must never be read as image
    for b, c in enumerate(contours):
        img[b, :, :cols] = c
    keep = dict(contour=img, n_in=np.array([len(k) for k in kps_list], np.int32),
                kps_in=np.stack([np.concatenate([k, np.zeros(ks - len(k), cabi.KP_DTYPE)]) for k in kps_list]))
    if masks is not None:
        mk = np.zeros((B, rows, pitch), np.uint8)
        for b, m in enumerate(masks):
            mk[b, :, :cols] = m
        keep["mask"] = mk
    if descs is not None:
        keep["desc_in"] = np.stack([np.concatenate([d, np.zeros((ks - len(d), 32), np.uint8)]) for d in descs])
    out = dict(n_out=np.full(B, -7, np.int32), kps_out=np.zeros((B, ks), cabi.KP_DTYPE), keep=np.full((B, ks), 9, np.uint8))
    if descs is not None:
        out["desc_out"] = np.full((B, ks, 32), 0xEE, np.uint8)
    if edge_cap:
        out.update(n_edge_sign=np.full(B, -7, np.int32), n_edge_free=np.full(B, -7, np.int32),
                   edge_sign=np.full((B, edge_cap, 2), -1, np.float32), edge_free=np.full((B, edge_cap, 2), -1, np.float32))
    a = cabi.BirdGuidanceArgs()
    fill(a, batch=B, kp_stride=ks, cols=cols, rows=rows, pitch=pitch, edge_cap=edge_cap, **keep, **out)
    return a, out, keep


def test_oracle_near_edges_known_answers():
    """One marked pixel at (row 100, col 300) of a 512x512 contour image.  nearEdges walks ROWS with the key point's x and
    COLUMNS with its y (Frame.cc:722-729), so the key point that 'sees' that pixel sits at x ~ 100, y ~ 300."""
    c = np.zeros((512, 512), np.uint8)
    c[100, 300] = 10                                     # >= 10 counts (edge), 9 would not
    pts = [(100.0, 300.0),    # centre: kept
           (300.0, 100.0),    # what a non-quirky x=col reading would keep: dropped
           (110.0, 300.0),    # rows trunc(100)..119 include 100: kept
           (110.5, 300.0),    # rows trunc(100.5)=100 ..: kept (truncation, not rounding)
           (111.0, 300.0),    # rows 101..: dropped
           (90.0, 300.0),     # rows 80..99 (row < 100.0 is exclusive): dropped
           (90.5, 300.0),     # rows 80..100 ((float)100 < 100.5): kept
           (100.0, 290.0),    # cols 280..299: dropped
           (100.0, 290.25),   # cols 280..300: kept
           (100.0, 310.0),    # cols 300..: kept
           (100.0, 310.99),   # cols trunc(300.99)=300: kept
           (100.0, 311.0)]    # cols 301..: dropped
    a, out, keep = make_args([c], [kp_array(pts)])
    assert O.lib().orc_bird_guidance(C.byref(a)) == 0
    expect = [1, 0, 1, 1, 0, 0, 1, 0, 1, 1, 1, 0]
    assert out["keep"][0, :len(pts)].tolist() == expect
    assert out["n_out"][0] == sum(expect)
    np.testing.assert_array_equal(out["kps_out"][0, :sum(expect)], keep["kps_in"][0, :len(pts)][np.array(expect, bool)])
    # value 9 is free
    c2 = np.zeros((512, 512), np.uint8)
    c2[100, 300] = 9
    a, out, keep = make_args([c2], [kp_array(pts[:1])])
    assert O.lib().orc_bird_guidance(C.byref(a)) == 0 and out["n_out"][0] == 0
    # clamps: a key point in the corner only looks inside the image (row/col 0, and < cols / < rows)
    c3 = np.zeros((512, 512), np.uint8)
    c3[0, 0] = 200
    c3[511, 511] = 200
    a, out, keep = make_args([c3], [kp_array([(3.0, 4.0), (508.0, 507.0), (20.0, 3.0), (3.0, 20.0)])])
    assert O.lib().orc_bird_guidance(C.byref(a)) == 0
    assert out["keep"][0, :4].tolist() == [1, 1, 0, 0]


def test_oracle_gen_edges_and_mask():
    c = np.zeros((6, 8), np.uint8)
    c[1, 2], c[1, 5], c[3, 0], c[5, 7], c[2, 2] = 10, 150, 149, 255, 9
    m = np.zeros((6, 8), np.uint8)
    m[2, 3] = 1
    pts = [(2.4, 2.6), (2.6, 2.4), (3.4, 1.6)]           # mask pixel (row (int)(y+.5), col (int)(x+.5)): (3,2) (2,3) (2,3)
    a, out, keep = make_args([c], [kp_array(pts)], masks=[m], edge_cap=3)
    assert O.lib().orc_bird_guidance(C.byref(a)) == 0
    assert out["n_edge_sign"][0] == 2 and out["n_edge_free"][0] == 2
    np.testing.assert_array_equal(out["edge_sign"][0, :2], [[2, 1], [0, 3]])     # (x = col, y = row), raster order
    np.testing.assert_array_equal(out["edge_free"][0, :2], [[5, 1], [7, 5]])
    assert out["keep"][0, :3].tolist() == [0, 1, 1]      # every box sees an edge pixel; only the mask decides


def _random_problem(seed, B, rows, cols, n, with_mask, with_desc):
    g = synth.rng(seed)
    contours, kps, descs, masks = [], [], [], []
    for b in range(B):
        c = np.zeros((rows, cols), np.uint8)
        for _ in range(12):                               # a few strokes and blobs of edge / freespace labels
            r0, c0 = g.integers(0, rows), g.integers(0, cols)
            if g.random() < 0.5:
                c[r0, max(0, c0 - 40):c0 + 40] = g.choice([9, 10, 100, 149, 150, 255])
            else:
                c[max(0, r0 - 30):r0 + 30, c0] = g.choice([9, 10, 100, 149, 150, 255])
        c[g.integers(0, rows, 30), g.integers(0, cols, 30)] = g.integers(0, 256, 30)
        nb = n if b % 2 == 0 else max(0, n - 37 * b)
        xy = np.stack([g.uniform(0, cols - 1, nb), g.uniform(0, rows - 1, nb)], 1).astype(np.float32)
        xy[: nb // 4] = np.round(xy[: nb // 4])           # integer positions hit the exclusive bounds
        xy[nb // 4: nb // 2] = np.round(xy[nb // 4: nb // 2] * 2) / 2
        contours.append(c)
        kps.append(kp_array([tuple(p) for p in xy]))
        descs.append(synth.random_descriptors(g, nb))
        masks.append((g.random((rows, cols)) < 0.7).astype(np.uint8) * g.integers(1, 256, (rows, cols)).astype(np.uint8))
    return contours, kps, (descs if with_desc else None), (masks if with_mask else None)


@pytest.mark.gpu
@pytest.mark.parametrize("seed,B,rows,cols,n,with_mask,with_desc,edge_cap", [
    (8900, 3, 512, 512, 2064, False, True, 4096), (8901, 2, 512, 512, 1500, True, True, 0), (8902, 4, 384, 384, 700, True, False, 100),
    (8903, 2, 300, 420, 900, False, True, 50000), (8904, 1, 512, 512, 0, False, True, 16), (8905, 2, 64, 64, 5000, True, True, 8192)])
def test_gpu_bird_guidance_matches_oracle(seed, B, rows, cols, n, with_mask, with_desc, edge_cap):
    import fishbirdeyevisualslam_amd as fb
    contours, kps, descs, masks = _random_problem(seed, B, rows, cols, n, with_mask, with_desc)
    a, oo, k1 = make_args(contours, kps, descs, masks, edge_cap, stride=max(n, 1) + 3)
    assert O.lib().orc_bird_guidance(C.byref(a)) == 0
    a2, oh, k2 = make_args(contours, kps, descs, masks, edge_cap, stride=max(n, 1) + 3)
    fb.check(fb.lib().fb_bird_guidance(C.byref(a2)), "fb_bird_guidance")
    assert 0 < oo["n_out"].sum() < sum(len(k) for k in kps) or n == 0       # the filter is neither empty nor the identity
    for key in oo:
        np.testing.assert_array_equal(oh[key], oo[key], err_msg=key)


@pytest.mark.gpu
def test_gpu_bird_keys_to_cam_matches_oracle():
    """fb_bird_keys_to_cam_dev (Frame.cc:365-373, Converter.cc:284-292,312-318) against the oracle, directly."""
    import torch
    import fishbirdeyevisualslam_amd as fb
    g = synth.rng(8950)
    B, ks = 3, 2100
    n = np.array([2100, 1, 977], np.int32)
    kps = np.zeros((B, ks), cabi.KP_DTYPE)
    kps["x"] = g.uniform(0, 511, (B, ks)).astype(np.float32)
    kps["y"] = g.uniform(0, 511, (B, ks)).astype(np.float32)
    kps["x"][:, :50] = np.round(kps["x"][:, :50])
    Tbc, Tcb = synth.extrinsics()
    T12 = np.ascontiguousarray(Tcb[:3, :4].astype(np.float32))
    ref = np.full((B, ks, 3), -5.0, np.float32)
    O.lib().orc_bird_keys_to_cam.argtypes = None
    rc = O.lib().orc_bird_keys_to_cam(C.c_void_p(kps.ctypes.data), C.c_void_p(n.ctypes.data), B, ks, 512, 512, C.c_double(synth.PIXEL2METER),
                                      C.c_double(synth.REAR_AXLE_TO_CENTER), C.c_void_p(T12.ctypes.data), C.c_void_p(ref.ctypes.data))
    assert rc == 0
    dev = torch.device("cuda:0")
    dk = torch.from_numpy(kps.view(np.uint8).reshape(-1)).to(dev)
    dn = torch.from_numpy(n).to(dev)
    dc = torch.full((B, ks, 3), -5.0, dtype=torch.float32, device=dev)
    fb.check(fb.lib().fb_bird_keys_to_cam_dev(C.c_void_p(dk.data_ptr()), C.c_void_p(dn.data_ptr()), B, ks, 512, 512, C.c_double(synth.PIXEL2METER),
                                              C.c_double(synth.REAR_AXLE_TO_CENTER), C.c_void_p(T12.ctypes.data), C.c_void_p(dc.data_ptr()),
                                              C.c_void_p(torch.cuda.current_stream().cuda_stream)), "fb_bird_keys_to_cam_dev")
    torch.cuda.synchronize()
    got = dc.cpu().numpy()
    assert got.tobytes() == ref.tobytes()
    # hand check of one entry: p = ((rows/2 - y) * p2m + rear, (cols/2 - x) * p2m, 0), then Tcb
    x, y = np.float32(kps["x"][0, 7]), np.float32(kps["y"][0, 7])
    p = np.array([np.float32((np.float32(256 - y)) * synth.PIXEL2METER + synth.REAR_AXLE_TO_CENTER), np.float32(np.float32(256 - x) * synth.PIXEL2METER), 0], np.float32)
    np.testing.assert_allclose(ref[0, 7], T12[:, :3] @ p + T12[:, 3], rtol=1e-5)
