"""Synthetic inputs for the hot path.

All RNG is numpy.random.Generator(PCG64(seed)).  These generators are shared by
tests/ and bench.py so that the CPU oracle and the HIP path see identical bytes.
They use numpy only (no oracle, no HIP).
"""
import math

import numpy as np

from .cabi import KP_DTYPE, FB_MAX_LEVELS

# ---- reference constants -------------------------------------------------------------
ORB_DEFAULT = dict(nfeatures=2000, scale_factor=1.2, nlevels=8, ini_th_fast=15, min_th_fast=5)  # fisheye.yaml:29-41
FRAME_GRID_COLS, FRAME_GRID_ROWS, FRAME_GRID_BIRD = 64, 48, 32  # Frame.h:38-40
PIXEL2METER, METER2PIXEL, REAR_AXLE_TO_CENTER = 0.03984, 25.1, 1.393  # Frame.cc:39-42


def rng(seed):
    return np.random.Generator(np.random.PCG64(seed))


def scale_tables(scale_factor=1.2, nlevels=8, nfeatures=2000):
    """float32 restatement of the ORBextractor tables (ORBextractor.cc:415-446) in numpy.
    Used only to shape synthetic data; the authoritative tables come from fb_orb_get_tables."""
    f = np.float32
    sf = np.ones(nlevels, f)
    for i in range(1, nlevels):
        sf[i] = f(sf[i - 1] * f(scale_factor))
    sig2 = (sf * sf).astype(f)
    return sf, (f(1.0) / sf).astype(f), sig2, (f(1.0) / sig2).astype(f)


def features_per_level(nfeatures=2000, scale_factor=1.2, nlevels=8):
    f = np.float32
    factor = f(1.0) / f(scale_factor)
    nd = f(f(nfeatures) * f(f(1) - factor) / f(f(1) - f(math.pow(float(factor), nlevels))))
    out, s = [], 0
    for _ in range(nlevels - 1):
        v = int(np.rint(nd))
        out.append(v)
        s += v
        nd = f(nd * factor)
    out.append(max(nfeatures - s, 0))
    return out


def synth_image(seed, w, h, n_rect=None, n_disc=None):
    """Mid-grey + rectangles + discs + N(0,3^2) noise."""
    g = rng(seed)
    if n_rect is None:
        n_rect = max(8, int(round(400 * (w * h) / (1280 * 720))))
    if n_disc is None:
        n_disc = max(4, int(round(200 * (w * h) / (1280 * 720))))
    img = np.full((h, w), 128.0, np.float32)
    for _ in range(n_rect):
        sw, sh = g.integers(8, 65), g.integers(8, 65)
        x0, y0 = g.integers(-8, w), g.integers(-8, h)
        val = g.integers(16, 240)
        img[max(y0, 0):max(y0 + sh, 0), max(x0, 0):max(x0 + sw, 0)] = val
    yy, xx = np.mgrid[0:h, 0:w]
    for _ in range(n_disc):
        r = g.integers(3, 13)
        cx, cy = g.integers(0, w), g.integers(0, h)
        val = g.integers(16, 240)
        y0, y1, x0, x1 = max(cy - r, 0), min(cy + r + 1, h), max(cx - r, 0), min(cx + r + 1, w)
        m = (yy[y0:y1, x0:x1] - cy) ** 2 + (xx[y0:y1, x0:x1] - cx) ** 2 <= r * r
        img[y0:y1, x0:x1][m] = val
    img += g.normal(0.0, 3.0, size=img.shape).astype(np.float32)
    return np.clip(np.rint(img), 0, 255).astype(np.uint8)


# ---- extrinsics (Frame::CalculateExtrinsics, Frame.cc:1015-1037) ---------------------
def extrinsics():
    tbc = np.array([3.747, 0.040, 0.736], np.float32)
    qx, qy, qz, qw = 0.631, -0.623, 0.325, -0.330
    n = math.sqrt(qx * qx + qy * qy + qz * qz + qw * qw)
    qx, qy, qz, qw = qx / n, qy / n, qz / n, qw / n
    Rbc = np.array([[1 - 2 * (qy * qy + qz * qz), 2 * (qx * qy - qw * qz), 2 * (qx * qz + qw * qy)],
                    [2 * (qx * qy + qw * qz), 1 - 2 * (qx * qx + qz * qz), 2 * (qy * qz - qw * qx)],
                    [2 * (qx * qz - qw * qy), 2 * (qy * qz + qw * qx), 1 - 2 * (qx * qx + qy * qy)]], np.float32)
    Tbc = np.eye(4, dtype=np.float32)
    Tbc[:3, :3] = Rbc
    Tbc[:3, 3] = tbc
    Tcb = np.eye(4, dtype=np.float32)
    Tcb[:3, :3] = Rbc.T
    Tcb[:3, 3] = -(Rbc.T @ tbc)
    return Tbc, Tcb


def front_grid_geom(w, h):
    """mnMinX=0, mnMaxX=cols (Frame.cc:790-793, zero distortion); Frame.cc:276-277."""
    f = np.float32
    return dict(min_x=0.0, min_y=0.0, inv_w=float(f(FRAME_GRID_COLS) / f(w)), inv_h=float(f(FRAME_GRID_ROWS) / f(h)),
                cols=FRAME_GRID_COLS, rows=FRAME_GRID_ROWS)


def bird_grid_geom(cols, rows):
    f = np.float32
    return dict(min_x=0.0, min_y=0.0, inv_w=float(f(FRAME_GRID_BIRD) / f(cols)), inv_h=float(f(FRAME_GRID_BIRD) / f(rows)),
                cols=FRAME_GRID_BIRD, rows=FRAME_GRID_BIRD)


# ---- random poses ----------------------------------------------------------------------
def rot_xyz(rx, ry, rz):
    cx, sx, cy, sy, cz, sz = math.cos(rx), math.sin(rx), math.cos(ry), math.sin(ry), math.cos(rz), math.sin(rz)
    Rx = np.array([[1, 0, 0], [0, cx, -sx], [0, sx, cx]])
    Ry = np.array([[cy, 0, sy], [0, 1, 0], [-sy, 0, cy]])
    Rz = np.array([[cz, -sz, 0], [sz, cz, 0], [0, 0, 1]])
    return Rz @ Ry @ Rx


def random_pose(g, t_range=5.0):
    """True Tcw: yaw U[0,2pi), small roll/pitch N(0,0.02^2), t U[-5,5]^3."""
    R = rot_xyz(g.normal(0, 0.02), g.uniform(0, 2 * math.pi), g.normal(0, 0.02))
    T = np.eye(4)
    T[:3, :3] = R
    T[:3, 3] = g.uniform(-t_range, t_range, 3)
    return T


def so3_exp(w):
    th = np.linalg.norm(w)
    K = np.array([[0, -w[2], w[1]], [w[2], 0, -w[0]], [-w[1], w[0], 0]])
    if th < 1e-12:
        return np.eye(3) + K
    return np.eye(3) + math.sin(th) / th * K + (1 - math.cos(th)) / th ** 2 * (K @ K)


def perturb_pose(g, T, rot_sigma=0.01, t_sigma=0.05):
    """exp(xi) * T with xi_rot~N(0,rot_sigma^2), xi_t~N(0,t_sigma^2)."""
    D = np.eye(4)
    D[:3, :3] = so3_exp(g.normal(0, rot_sigma, 3))
    D[:3, 3] = g.normal(0, t_sigma, 3)
    return D @ T


def to12(T):
    return np.ascontiguousarray(np.asarray(T)[:3, :4].astype(np.float32).reshape(12))


# ---- descriptors -----------------------------------------------------------------------
def random_descriptors(g, n):
    return g.integers(0, 256, size=(n, 32), dtype=np.uint8)


def flip_bits(g, desc, p=0.08):
    """Each row gets k~Binomial(256,p) flipped bits."""
    n = desc.shape[0]
    bits = np.unpackbits(desc, axis=1)
    flips = g.random((n, 256)) < p
    return np.packbits(bits ^ flips.astype(np.uint8), axis=1)


def random_keypoints(g, n, w, h, nlevels=8, nfeatures=2000, margin=0.0):
    """Keypoints uniform in the image, octave ~ features_per_level, angle U[0,360)."""
    sf, _, _, _ = scale_tables(nlevels=nlevels)
    per = np.array(features_per_level(nfeatures, 1.2, nlevels), np.float64)
    k = np.zeros(n, KP_DTYPE)
    k["x"] = g.uniform(margin, w - margin, n).astype(np.float32)
    k["y"] = g.uniform(margin, h - margin, n).astype(np.float32)
    k["octave"] = g.choice(nlevels, size=n, p=per / per.sum()).astype(np.int32)
    k["angle"] = g.uniform(0, 360, n).astype(np.float32)
    k["size"] = (31 * sf[k["octave"]]).astype(np.int32).astype(np.float32)
    k["response"] = g.integers(5, 120, n).astype(np.float32)
    return k


# ---- M3 problem: SearchByProjection(cur, last) ------------------------------------------
def make_proj_frame_problem(seed, n_cur=2000, n_last=2000, w=1280, h=720, fx=500.0, fy=500.0, dup_frac=0.05,
                            obs0_frac=0.03, blocked_frac=0.02, rot_offset=12.0):
    g = rng(seed)
    cx, cy = w / 2.0, h / 2.0
    cur = random_keypoints(g, n_cur, w, h)
    cur_desc = random_descriptors(g, n_cur)
    Tcw = random_pose(g)
    R, t = Tcw[:3, :3], Tcw[:3, 3]
    # queries: 70% noisy copies of distinct targets, 30% fresh random; some duplicates of a target
    tgt = g.permutation(n_cur)[:n_last] if n_last <= n_cur else g.integers(0, n_cur, n_last)
    ndup = int(dup_frac * n_last)
    if ndup:
        tgt[g.integers(0, n_last, ndup)] = tgt[g.integers(0, n_last, ndup)]
    is_copy = g.random(n_last) < 0.7
    last_desc = random_descriptors(g, n_last)
    last_desc[is_copy] = flip_bits(g, cur_desc[tgt[is_copy]])
    u = cur["x"][tgt] + g.normal(0, 3.0, n_last)
    v = cur["y"][tgt] + g.normal(0, 3.0, n_last)
    z = g.uniform(2.0, 30.0, n_last)
    Xc = np.stack([(u - cx) / fx * z, (v - cy) / fy * z, z], 1)
    Xw = (R.T @ (Xc - t).T).T
    last_oct = np.clip(cur["octave"][tgt] + g.integers(-1, 2, n_last), 0, 7).astype(np.int32)
    last_angle = np.mod(cur["angle"][tgt] + rot_offset + g.normal(0, 4.0, n_last), 360.0).astype(np.float32)
    # a few wildly rotated ones so the histogram culling has something to do
    wild = g.random(n_last) < 0.08
    last_angle[wild] = g.uniform(0, 360, int(wild.sum())).astype(np.float32)
    return dict(
        w=w, h=h, fx=fx, fy=fy, cx=cx, cy=cy, Tcw=to12(Tcw), cur_kps=cur, cur_desc=cur_desc,
        cur_blocked=(g.random(n_cur) < blocked_frac).astype(np.uint8),
        last_valid=(g.random(n_last) < 0.95).astype(np.uint8),
        last_obs_pos=(g.random(n_last) >= obs0_frac).astype(np.uint8),
        last_xw=np.ascontiguousarray(Xw.astype(np.float32)), last_desc=last_desc, last_octave=last_oct,
        last_angle=last_angle)


# ---- M9 problem: BirdMapPointMatch -----------------------------------------------------
def make_bird_mp_problem(seed, n_cur=1000, n_ref=1000, cols=512, rows=512):
    g = rng(seed)
    Tbc, Tcb = extrinsics()
    cur = random_keypoints(g, n_cur, cols, rows)
    cur["octave"] = (g.random(n_cur) < 0.5).astype(np.int32) * g.integers(0, 8, n_cur).astype(np.int32)
    cur_desc = random_descriptors(g, n_cur)
    # cam XYZ of the current bird keypoints (Converter.cc:284-292, 312-318) in float64 then float32
    bx = (rows // 2 - cur["y"].astype(np.float64)) * PIXEL2METER + REAR_AXLE_TO_CENTER
    by = (cols // 2 - cur["x"].astype(np.float64)) * PIXEL2METER
    base = np.stack([bx, by, np.zeros(n_cur)], 1).astype(np.float32)
    cam = (Tcb[:3, :3].astype(np.float64) @ base.T.astype(np.float64)).T + Tcb[:3, 3]
    Tcw = random_pose(g)
    tgt = g.permutation(n_cur)[:n_ref] if n_ref <= n_cur else g.integers(0, n_cur, n_ref)
    is_copy = g.random(n_ref) < 0.7
    ref_desc = random_descriptors(g, n_ref)
    ref_desc[is_copy] = flip_bits(g, cur_desc[tgt[is_copy]], p=0.05)
    # world position of the reference bird map points: the target's camera point (+noise), some off-plane
    pc = cam[tgt] + g.normal(0, 0.01, (n_ref, 3))
    far = g.random(n_ref) < 0.1
    pc[far] += g.uniform(-0.3, 0.3, (int(far.sum()), 3))
    Xw = (Tcw[:3, :3].T @ (pc - Tcw[:3, 3]).T).T
    return dict(cols=cols, rows=rows, Tbc=Tbc, Tcb=Tcb, Tcw=to12(Tcw), cur_kps=cur, cur_desc=cur_desc,
                cur_cam_xyz=np.ascontiguousarray(cam.astype(np.float32)),
                ref_valid=(g.random(n_ref) < 0.95).astype(np.uint8),
                ref_xw=np.ascontiguousarray(Xw.astype(np.float32)), ref_desc=ref_desc)


# ---- pose optimisation problem (seed 3000) ----------------------------------
def make_pose_problem(seed, n_front=2000, n_bird=1000, w=1280, h=720, fx=500.0, fy=500.0, outlier_frac=0.1,
                      bird_half=10.2):
    g = rng(seed)
    cx, cy = w / 2.0, h / 2.0
    _, _, sig2, inv_sig2 = scale_tables()
    per = np.array(features_per_level(), np.float64)
    Tbc, Tcb = extrinsics()
    T = random_pose(g)
    R, t = T[:3, :3], T[:3, 3]
    out = dict(fx=fx, fy=fy, cx=cx, cy=cy, T_true=T)
    # front: points filling the image at depth U[2,30]
    u = g.uniform(0, w, n_front)
    v = g.uniform(0, h, n_front)
    z = g.uniform(2.0, 30.0, n_front)
    Xc = np.stack([(u - cx) / fx * z, (v - cy) / fy * z, z], 1)
    Xw = (R.T @ (Xc - t).T).T
    octv = g.choice(8, size=n_front, p=per / per.sum())
    obs = np.stack([u, v], 1) + g.normal(0, 1.0, (n_front, 2)) * np.sqrt(sig2[octv])[:, None]
    bad = g.random(n_front) < outlier_frac
    obs[bad] += g.uniform(-20, 20, (int(bad.sum()), 2))
    out.update(front_xw=np.ascontiguousarray(Xw.astype(np.float32)), front_obs=np.ascontiguousarray(obs.astype(np.float32)),
               front_inv_sigma2=inv_sig2[octv].astype(np.float32), front_is_outlier=bad)
    # bird: points on the base plane z=0 inside the +-10.2 m square, Xc = Tcb*p_b + noise
    pb = np.stack([g.uniform(-bird_half, bird_half, n_bird) + REAR_AXLE_TO_CENTER, g.uniform(-bird_half, bird_half, n_bird),
                   np.zeros(n_bird)], 1)
    pcam = (Tcb[:3, :3].astype(np.float64) @ pb.T).T + Tcb[:3, 3].astype(np.float64)
    Xwb = (R.T @ (pcam - t).T).T
    xc_meas = pcam + g.normal(0, 0.01, (n_bird, 3))
    badb = g.random(n_bird) < outlier_frac
    xc_meas[badb] += g.uniform(-0.3, 0.3, (int(badb.sum()), 3))
    boct = g.choice(8, size=n_bird, p=per / per.sum())
    out.update(bird_xw=np.ascontiguousarray(Xwb.astype(np.float32)), bird_xc=np.ascontiguousarray(xc_meas.astype(np.float32)),
               bird_inv_sigma2=inv_sig2[boct].astype(np.float32), bird_is_outlier=badb)
    out["Tcw0"] = to12(perturb_pose(g, T))
    return out


# ---- M2 problem: SearchByProjection(F, vpMapPoints, th) ---------------------------------
def make_proj_points_problem(seed, n_cur=2000, n_mp=3000, w=1280, h=720):
    g = rng(seed)
    cur = random_keypoints(g, n_cur, w, h)
    cur_desc = random_descriptors(g, n_cur)
    tgt = g.integers(0, n_cur, n_mp)
    is_copy = g.random(n_mp) < 0.7
    mp_desc = random_descriptors(g, n_mp)
    mp_desc[is_copy] = flip_bits(g, cur_desc[tgt[is_copy]])
    proj = np.stack([cur["x"][tgt] + g.normal(0, 2.0, n_mp), cur["y"][tgt] + g.normal(0, 2.0, n_mp)], 1)
    level = np.clip(cur["octave"][tgt] + g.integers(0, 2, n_mp), 0, 7).astype(np.int32)
    return dict(w=w, h=h, cur_kps=cur, cur_desc=cur_desc,
                cur_blocked=(g.random(n_cur) < 0.05).astype(np.uint8),
                mp_track=(g.random(n_mp) < 0.9).astype(np.uint8),
                mp_obs_pos=(g.random(n_mp) < 0.95).astype(np.uint8),
                mp_proj=np.ascontiguousarray(proj.astype(np.float32)), mp_level=level,
                mp_view_cos=g.uniform(0.99, 1.0, n_mp).astype(np.float32), mp_desc=mp_desc)


# ---- M8 problem: BirdviewMatch(isProject=0) ---------------------------------------------
def make_birdview_problem(seed, n_cur=1000, n_ref=1000, cols=512, rows=512):
    g = rng(seed)
    cur = random_keypoints(g, n_cur, cols, rows)
    cur["octave"] = np.where(g.random(n_cur) < 0.6, 0, g.integers(0, 8, n_cur)).astype(np.int32)
    cur_desc = random_descriptors(g, n_cur)
    tgt = g.integers(0, n_cur, n_ref)
    ref = np.zeros(n_ref, KP_DTYPE)
    ref["x"] = (cur["x"][tgt] + g.normal(0, 2.0, n_ref)).astype(np.float32)
    ref["y"] = (cur["y"][tgt] + g.normal(0, 2.0, n_ref)).astype(np.float32)
    ref["octave"] = np.where(g.random(n_ref) < 0.8, 0, g.integers(1, 8, n_ref)).astype(np.int32)
    ref["angle"] = np.mod(cur["angle"][tgt] + 20.0 + g.normal(0, 5.0, n_ref), 360.0).astype(np.float32)
    wild = g.random(n_ref) < 0.1
    ref["angle"][wild] = g.uniform(0, 360, int(wild.sum())).astype(np.float32)
    is_copy = g.random(n_ref) < 0.7
    ref_desc = random_descriptors(g, n_ref)
    ref_desc[is_copy] = flip_bits(g, cur_desc[tgt[is_copy]], p=0.05)
    return dict(cols=cols, rows=rows, cur_kps=cur, cur_desc=cur_desc, ref_kps=ref, ref_desc=ref_desc)


# ---- local bundle adjustment problem (seed 4000) ------------------------------
def odom_transform(p1, p2, Tbc, Tcb):
    """Frame::GetTransformFromOdometer (Frame.cc:1049-1067): planar odometer poses (x,y,theta) -> float 4x4 Tcb*T12b*Tbc."""
    x1, y1, t1 = p1
    x2, y2, t2 = p2
    t12 = t2 - t1
    x12 = (x2 - x1) * math.cos(t1) + (y2 - y1) * math.sin(t1)
    y12 = (y2 - y1) * math.cos(t1) - (x2 - x1) * math.sin(t1)
    T12b = np.array([[math.cos(t12), -math.sin(t12), 0, x12], [math.sin(t12), math.cos(t12), 0, y12], [0, 0, 1, 0],
                     [0, 0, 0, 1]], np.float32)
    return (Tcb @ T12b @ Tbc).astype(np.float32)


def make_ba_problem(seed=4000, n_kf=20, n_fixed=2, n_mp=8000, n_mpb=2000, w=1280, h=720, fx=500.0, fy=500.0,
                    outlier_frac=0.03, wP=3.0):
    """n_kf keyframes on a planar arc (0.5 m spacing, the n_fixed oldest fixed), n_mp points each observed by
    2..8 keyframes that see it, n_mpb bird points observed by 2..5 consecutive keyframes, odometry chain
    (i,i+1),(i,i+2),(i,i+3) as Optimizer.cc:2419-2495, every estimate perturbed."""
    g = rng(seed)
    cx, cy = w / 2.0, h / 2.0
    _, _, sig2, inv_sig2 = scale_tables()
    per = np.array(features_per_level(), np.float64)
    Tbc, Tcb = extrinsics()
    Tbc64, Tcb64 = Tbc.astype(np.float64), Tcb.astype(np.float64)
    # base (vehicle) poses on an arc of radius 30 m, planar
    odo = []
    for k in range(n_kf):
        s = 0.5 * k
        th = s / 30.0
        odo.append((30.0 * math.sin(th), 30.0 * (1 - math.cos(th)), th))
    Twb = []
    for (x, y, th) in odo:
        T = np.eye(4)
        T[:3, :3] = rot_xyz(0, 0, th)
        T[:3, 3] = [x, y, 0]
        Twb.append(T)
    # camera pose Tcw = Tcb * Tbw
    Tcw_true = [Tcb64 @ np.linalg.inv(T) for T in Twb]
    # map points: sample in the frustum of a random keyframe, observe from 2..8 keyframes that see them
    obs_kf, obs_mp, obs_uv, obs_inf = [], [], [], []
    mp = np.zeros((n_mp, 3))
    j = 0
    while j < n_mp:
        k0 = g.integers(0, n_kf)
        u, v, z = g.uniform(0, w), g.uniform(0, h), g.uniform(4.0, 40.0)
        Xc = np.array([(u - cx) / fx * z, (v - cy) / fy * z, z])
        Xw = np.linalg.inv(Tcw_true[k0])[:3, :3] @ Xc + np.linalg.inv(Tcw_true[k0])[:3, 3]
        vis = []
        for k in range(n_kf):
            pc = Tcw_true[k][:3, :3] @ Xw + Tcw_true[k][:3, 3]
            if pc[2] > 1.0:
                uu, vv = pc[0] / pc[2] * fx + cx, pc[1] / pc[2] * fy + cy
                if 0 <= uu < w and 0 <= vv < h:
                    vis.append((k, uu, vv))
        if len(vis) < 2:
            continue
        nobs = min(len(vis), int(g.integers(2, 9)))
        pick = sorted(g.choice(len(vis), nobs, replace=False))
        mp[j] = Xw
        for i in pick:
            k, uu, vv = vis[i]
            o = int(g.choice(8, p=per / per.sum()))
            n2 = g.normal(0, 1.0, 2) * math.sqrt(sig2[o])
            if g.random() < outlier_frac:
                n2 += g.uniform(-15, 15, 2)
            obs_kf.append(k); obs_mp.append(j); obs_uv.append((uu + n2[0], vv + n2[1])); obs_inf.append(inv_sig2[o])
        j += 1
    # bird points: on the ground plane near the vehicle, observed by 2..5 consecutive keyframes
    bobs_kf, bobs_mpb, bobs_xc, bobs_inf = [], [], [], []
    mpb = np.zeros((n_mpb, 3))
    for j in range(n_mpb):
        k0 = int(g.integers(0, n_kf - 1))
        pb = np.array([g.uniform(-8, 8) + REAR_AXLE_TO_CENTER, g.uniform(-8, 8), 0.0, 1.0])
        Xw = (Twb[k0] @ pb)[:3]
        mpb[j] = Xw
        nobs = int(g.integers(2, 6))
        for k in range(k0, min(k0 + nobs, n_kf)):
            pc = Tcw_true[k][:3, :3] @ Xw + Tcw_true[k][:3, 3]
            n3 = g.normal(0, 0.01, 3)
            if g.random() < outlier_frac:
                n3 += g.uniform(-3, 3, 3)
            bobs_kf.append(k); bobs_mpb.append(j); bobs_xc.append(pc + n3); bobs_inf.append(inv_sig2[int(g.choice(8, p=per / per.sum()))])
    # odometry edges from noisy planar poses
    odo_n = [(x + g.normal(0, 0.01), y + g.normal(0, 0.01), th + g.normal(0, 0.002)) for (x, y, th) in odo]
    oi, oj, oT, oinfo = [], [], [], []
    for i in range(n_kf - 1):
        oi.append(i); oj.append(i + 1); oT.append(odom_transform(odo_n[i], odo_n[i + 1], Tbc, Tcb)); oinfo.append(1e4 * wP)
        if i + 2 < n_kf:
            oi.append(i); oj.append(i + 2); oT.append(odom_transform(odo_n[i], odo_n[i + 2], Tbc, Tcb)); oinfo.append(2e3)
            if i + 3 < n_kf:
                oi.append(i); oj.append(i + 3); oT.append(odom_transform(odo_n[i], odo_n[i + 3], Tbc, Tcb)); oinfo.append(1e3 * wP)
    kf_T = np.stack([to12(T if k < n_fixed else perturb_pose(g, T, 0.005, 0.03)) for k, T in enumerate(Tcw_true)])
    fixed = np.zeros(n_kf, np.uint8)
    fixed[:n_fixed] = 1
    return dict(fx=fx, fy=fy, cx=cx, cy=cy, wP=wP, kf_Tcw=kf_T, kf_fixed=fixed, kf_true=np.stack([to12(T) for T in Tcw_true]),
                mp_xw=(mp + g.normal(0, 0.05, mp.shape)).astype(np.float32), mp_true=mp.astype(np.float32),
                mpb_xw=(mpb + g.normal(0, 0.02, mpb.shape)).astype(np.float32),
                obs_kf=np.array(obs_kf, np.int32), obs_mp=np.array(obs_mp, np.int32),
                obs_uv=np.array(obs_uv, np.float32).reshape(-1, 2), obs_inv_sigma2=np.array(obs_inf, np.float32),
                bobs_kf=np.array(bobs_kf, np.int32), bobs_mpb=np.array(bobs_mpb, np.int32),
                bobs_xc=np.array(bobs_xc, np.float32).reshape(-1, 3), bobs_inv_sigma2=np.array(bobs_inf, np.float32),
                odom_kf_i=np.array(oi, np.int32), odom_kf_j=np.array(oj, np.int32),
                odom_Tij=np.stack([to12(T) for T in oT]) if oT else np.zeros((0, 12), np.float32),
                odom_info=np.array(oinfo, np.float64), odo=np.array(odo_n, np.float64).reshape(-1, 3))
