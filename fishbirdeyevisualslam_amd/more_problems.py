"""Synthetic problems + C-ABI argument builders for the relocalisation / loop-closing matchers (M4
SearchByProjection(Frame, KeyFrame), M6 SearchByBoW(KeyFrame, KeyFrame)) and the Frame geometry entry points
(isInFrustum, UndistortKeyPoints, ComputeImageBounds)."""
import ctypes as C

import numpy as np

from . import bow_problem as BP, cabi, synth
from .cabi import fill

# fisheye intrinsics / distortion of the reference's front camera configuration (SURVEY 8d uses synthetic values of
# the same magnitude; the reference's yaml is data, not code)
FISHEYE_K = np.array([650.0, 648.0, 640.0, 360.0], np.float32)
FISHEYE_D = np.array([-0.035, 0.0021, -0.0049, 0.0011], np.float32)


def _stack(problems, key, n, dt, tail=()):
    return np.stack([np.concatenate([np.asarray(p[key], dt), np.zeros((n - len(p[key]),) + tail, dt)]) for p in problems])


# ---- M4 ----------------------------------------------------------------------------------------------------------
def make_proj_kf_problem(seed, n_cur=2000, n_kf=2000, w=1280, h=720, fx=500.0, fy=500.0):
    """Reuses the M3 generator for the geometry (key-frame points project near noisy copies of current key points)
    and adds the per-point scale-invariance distances so that PredictScale lands around the target's octave."""
    g = synth.rng(seed + 77)
    p = synth.make_proj_frame_problem(seed, n_cur, n_kf, w, h, fx, fy, dup_frac=0.15)
    T = np.asarray(p["Tcw"], np.float64).reshape(3, 4)
    Ow = -T[:, :3].T @ T[:, 3]
    dist = np.linalg.norm(p["last_xw"].astype(np.float64) - Ow, axis=1) if n_kf else np.zeros(0)
    sf = 1.2
    lvl = p["last_octave"].astype(np.float64) + g.uniform(-0.6, 0.6, n_kf)
    max_dist = dist * sf ** lvl                      # PredictScale = ceil(log(max/dist)/log sf)
    min_dist = max_dist / sf ** 7
    # some points outside the invariance region, some degenerate
    far = g.random(n_kf) < 0.05
    max_dist[far] *= 0.3
    out = dict(p)
    out.update(kf_valid=(g.random(n_kf) < 0.9).astype(np.uint8), kf_xw=p["last_xw"], kf_desc=p["last_desc"],
               kf_max_dist=max_dist.astype(np.float32), kf_min_dist=min_dist.astype(np.float32),
               kf_angle=p["last_angle"], cur_blocked=(g.random(n_cur) < 0.1).astype(np.uint8))
    return out


def proj_kf_args(problems, cell_start, cell_items, th=10.0, orb_dist=100, check_ori=1):
    B = len(problems)
    p0 = problems[0]
    cs_ = max(len(p["cur_kps"]) for p in problems)
    ks_ = max(len(p["kf_xw"]) for p in problems)
    keep = dict(
        n_cur=np.array([len(p["cur_kps"]) for p in problems], np.int32),
        cur_kps=_stack(problems, "cur_kps", cs_, cabi.KP_DTYPE),
        cur_desc=_stack(problems, "cur_desc", cs_, np.uint8, (32,)),
        cur_cell_start=np.ascontiguousarray(cell_start), cur_cell_items=np.ascontiguousarray(cell_items),
        cur_blocked=_stack(problems, "cur_blocked", cs_, np.uint8),
        cur_Tcw=np.ascontiguousarray(np.stack([p["Tcw"] for p in problems]), np.float32),
        n_kf=np.array([len(p["kf_xw"]) for p in problems], np.int32),
        kf_valid=_stack(problems, "kf_valid", ks_, np.uint8),
        kf_xw=_stack(problems, "kf_xw", ks_, np.float32, (3,)),
        kf_desc=_stack(problems, "kf_desc", ks_, np.uint8, (32,)),
        kf_max_dist=_stack(problems, "kf_max_dist", ks_, np.float32),
        kf_min_dist=_stack(problems, "kf_min_dist", ks_, np.float32),
        kf_angle=_stack(problems, "kf_angle", ks_, np.float32),
    )
    out = dict(match_cur_to_kf=np.full((B, cs_), -7, np.int32), nmatches=np.full(B, -7, np.int32))
    a = cabi.ProjKfArgs()
    fill(a, batch=B, cur_stride=cs_, kf_stride=ks_, th=th, orb_dist=orb_dist, n_levels=8,
         log_scale_factor=float(np.log(np.float32(1.2))), **keep, **out)
    fill(a.cam, fx=p0["fx"], fy=p0["fy"], cx=p0["cx"], cy=p0["cy"], min_x=0.0, min_y=0.0, max_x=float(p0["w"]),
         max_y=float(p0["h"]))
    fill(a.grid, **synth.front_grid_geom(p0["w"], p0["h"]))
    fill(a, scale_factors=[float(x) for x in synth.scale_tables()[0]])
    fill(a.matcher, nnratio=0.9, check_orientation=check_ori)
    return a, out, keep


# ---- M6 ----------------------------------------------------------------------------------------------------------
def make_bow_kf_problem(seed, n1=1500, n2=1500, share_prefix=True):
    p = BP.make_bow_problem(seed, n1, n2, share_prefix)
    g = synth.rng(seed + 5)
    return dict(kps1=p["kf_kps"], desc1=p["kf_desc"], has_mp1=p["kf_has_mp"], kps2=p["f_kps"], desc2=p["f_desc"],
                has_mp2=(g.random(n2) < 0.7).astype(np.uint8))


def bow_kf_args(problems, nnratio=0.75, check_ori=1):
    B = len(problems)
    s1 = max(len(p["kps1"]) for p in problems)
    s2 = max(len(p["kps2"]) for p in problems)
    keep = dict(n1=np.array([len(p["kps1"]) for p in problems], np.int32), kps1=_stack(problems, "kps1", s1, cabi.KP_DTYPE),
                desc1=_stack(problems, "desc1", s1, np.uint8, (32,)), has_mp1=_stack(problems, "has_mp1", s1, np.uint8),
                n2=np.array([len(p["kps2"]) for p in problems], np.int32), kps2=_stack(problems, "kps2", s2, cabi.KP_DTYPE),
                desc2=_stack(problems, "desc2", s2, np.uint8, (32,)), has_mp2=_stack(problems, "has_mp2", s2, np.uint8))
    fv1, k1 = BP._fv_struct([BP.feature_vector(p["desc1"]) for p in problems], 100, s1)
    fv2, k2 = BP._fv_struct([BP.feature_vector(p["desc2"]) for p in problems], 100, s2)
    out = dict(matches12=np.full((B, s1), -7, np.int32), nmatches=np.full(B, -7, np.int32))
    a = cabi.BowKfArgs()
    fill(a, batch=B, kf1_stride=s1, kf2_stride=s2, **keep, **out)
    a.fv1, a.fv2 = fv1, fv2
    fill(a.matcher, nnratio=nnratio, check_orientation=check_ori)
    return a, out, (keep, k1, k2)


# ---- Frame::isInFrustum --------------------------------------------------------------------------------------------
def make_frustum_problem(seed, n_mp=5000, w=1280, h=720, fx=500.0, fy=500.0):
    g = synth.rng(seed)
    cx, cy = w / 2.0, h / 2.0
    T = synth.random_pose(g)
    R, t = T[:3, :3], T[:3, 3]
    # 75% in front of the camera and inside a slightly enlarged image, the rest anywhere around it
    u = g.uniform(-0.15 * w, 1.15 * w, n_mp)
    v = g.uniform(-0.15 * h, 1.15 * h, n_mp)
    z = g.uniform(1.0, 40.0, n_mp)
    behind = g.random(n_mp) < 0.1
    z[behind] *= -1.0
    Xc = np.stack([(u - cx) / fx * z, (v - cy) / fy * z, z], 1)
    Xw = (R.T @ (Xc - t).T).T
    Ow = -R.T @ t
    PO = Xw - Ow
    dist = np.linalg.norm(PO, axis=1)
    # mean viewing direction: towards the camera, tilted by up to ~75 degrees
    n = PO / dist[:, None] + g.normal(0, 0.6, (n_mp, 3))
    n /= np.linalg.norm(n, axis=1, keepdims=True)
    lvl = g.uniform(-1.0, 8.5, n_mp)
    max_dist = dist * 1.2 ** lvl
    min_dist = max_dist / 1.2 ** 7
    T32 = synth.to12(T)
    Ow32 = np.zeros(3, np.float32)
    for r in range(3):  # mOw = -mRcw.t()*mtcw with double accumulation of the float entries (Frame.cc:432)
        Ow32[r] = np.float32(-sum(float(T32[k * 4 + r]) * float(T32[k * 4 + 3]) for k in range(3)))
    return dict(w=w, h=h, fx=fx, fy=fy, cx=cx, cy=cy, Tcw=T32, Ow=Ow32, mp_valid=(g.random(n_mp) < 0.9).astype(np.uint8),
                mp_xw=np.ascontiguousarray(Xw, np.float32), mp_normal=np.ascontiguousarray(n, np.float32),
                mp_max_dist=max_dist.astype(np.float32), mp_min_dist=min_dist.astype(np.float32))


def frustum_args(problems, cos_limit=0.5, mbf=40.0):
    B = len(problems)
    p0 = problems[0]
    ms = max(len(p["mp_xw"]) for p in problems)
    keep = dict(Tcw=np.ascontiguousarray(np.stack([p["Tcw"] for p in problems]), np.float32),
                Ow=np.ascontiguousarray(np.stack([p["Ow"] for p in problems]), np.float32),
                n_mp=np.array([len(p["mp_xw"]) for p in problems], np.int32),
                mp_valid=_stack(problems, "mp_valid", ms, np.uint8), mp_xw=_stack(problems, "mp_xw", ms, np.float32, (3,)),
                mp_normal=_stack(problems, "mp_normal", ms, np.float32, (3,)),
                mp_max_dist=_stack(problems, "mp_max_dist", ms, np.float32),
                mp_min_dist=_stack(problems, "mp_min_dist", ms, np.float32))
    out = dict(in_view=np.full((B, ms), 9, np.uint8), proj=np.full((B, ms, 2), -7.0, np.float32),
               proj_xr=np.full((B, ms), -7.0, np.float32), level=np.full((B, ms), -7, np.int32),
               view_cos=np.full((B, ms), -7.0, np.float32))
    a = cabi.FrustumArgs()
    fill(a, batch=B, mp_stride=ms, mbf=mbf, viewing_cos_limit=cos_limit, n_levels=8,
         log_scale_factor=float(np.log(np.float32(1.2))), **keep, **out)
    fill(a.cam, fx=p0["fx"], fy=p0["fy"], cx=p0["cx"], cy=p0["cy"], min_x=0.0, min_y=0.0, max_x=float(p0["w"]),
         max_y=float(p0["h"]))
    return a, out, keep


# ---- Frame::UndistortKeyPoints / ComputeImageBounds -------------------------------------------------------------------
def undistort(libobj, prefix, kps, K4=FISHEYE_K, D4=FISHEYE_D):
    """Call <prefix>undistort_keypoints of the HIP library or of the oracle on host arrays."""
    kps = np.ascontiguousarray(kps)
    out = np.zeros_like(kps)
    K4 = np.ascontiguousarray(K4, np.float32)
    D4 = np.ascontiguousarray(D4, np.float32)
    rc = getattr(libobj, prefix + "undistort_keypoints")(C.c_void_p(kps.ctypes.data), C.c_int(len(kps)),
                                                        C.c_void_p(K4.ctypes.data), C.c_void_p(D4.ctypes.data),
                                                        C.c_void_p(out.ctypes.data))
    assert rc == 0, rc
    return out


def image_bounds(libobj, prefix, cols, rows, K4=FISHEYE_K, D4=FISHEYE_D):
    K4 = np.ascontiguousarray(K4, np.float32)
    D4 = np.ascontiguousarray(D4, np.float32)
    out = np.zeros(4, np.float32)
    rc = getattr(libobj, prefix + "image_bounds")(C.c_int(cols), C.c_int(rows), C.c_void_p(K4.ctypes.data),
                                                  C.c_void_p(D4.ctypes.data), C.c_void_p(out.ctypes.data))
    assert rc == 0, rc
    return out
