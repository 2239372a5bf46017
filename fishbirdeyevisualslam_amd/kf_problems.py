"""Synthetic problems + C-ABI argument builders for the key-frame side matcher entry points: Fuse, Fuse-Sim3,
SearchByProjection(KF, Scw), SearchBySim3, SearchForInitialization, and MapPoint::ComputeDistinctiveDescriptors."""
import numpy as np

from . import cabi, synth
from .cabi import fill

W, H, FX, FY = 1280, 720, 500.0, 500.0
SF = 1.2


def _stack(problems, key, n, dt, tail=()):
    return np.stack([np.concatenate([np.asarray(p[key], dt), np.zeros((n - len(p[key]),) + tail, dt)]) for p in problems])


def _points_for(g, T, kps, desc, n_mp, copy_frac=0.75):
    """Map points that project (with pose T) near key points of `kps`, with distances consistent with their octaves."""
    n_kf = len(kps)
    cx, cy = W / 2.0, H / 2.0
    R, t = T[:3, :3], T[:3, 3]
    tgt = g.integers(0, max(n_kf, 1), n_mp)
    is_copy = (g.random(n_mp) < copy_frac) & (n_kf > 0)
    u = np.where(is_copy, kps["x"][tgt] + g.normal(0, 1.2, n_mp), g.uniform(-50, W + 50, n_mp)) if n_kf else g.uniform(0, W, n_mp)
    v = np.where(is_copy, kps["y"][tgt] + g.normal(0, 1.2, n_mp), g.uniform(-50, H + 50, n_mp)) if n_kf else g.uniform(0, H, n_mp)
    z = g.uniform(2.0, 30.0, n_mp)
    z[g.random(n_mp) < 0.05] *= -1.0
    Xc = np.stack([(u - cx) / FX * z, (v - cy) / FY * z, z], 1)
    Xw = (R.T @ (Xc - t).T).T
    Ow = -R.T @ t
    PO = Xw - Ow
    dist = np.linalg.norm(PO, axis=1)
    nrm = PO / dist[:, None] + g.normal(0, 0.45, (n_mp, 3))
    nrm /= np.linalg.norm(nrm, axis=1, keepdims=True)
    oct_t = kps["octave"][tgt].astype(np.float64) if n_kf else np.zeros(n_mp)
    q = oct_t + g.uniform(-0.4, 0.9, n_mp)
    max_dist = dist * SF ** q
    min_dist = max_dist / SF ** 7
    max_dist[g.random(n_mp) < 0.04] *= 0.3
    d = synth.random_descriptors(g, n_mp)
    if n_kf:
        d[is_copy] = synth.flip_bits(g, desc[tgt[is_copy]], p=0.06)
    return dict(mp_valid=(g.random(n_mp) < 0.9).astype(np.uint8), mp_xw=np.ascontiguousarray(Xw, np.float32),
                mp_normal=np.ascontiguousarray(nrm, np.float32), mp_max_dist=max_dist.astype(np.float32),
                mp_min_dist=min_dist.astype(np.float32), mp_desc=d)


def make_kf_points_problem(seed, n_kf=2000, n_mp=3000, sim3=False):
    g = synth.rng(seed)
    kps = synth.random_keypoints(g, n_kf, W, H)
    desc = synth.random_descriptors(g, n_kf)
    T = synth.random_pose(g)
    out = dict(kf_kps=kps, kf_desc=desc, kf_matched=(g.random(n_kf) < 0.1).astype(np.uint8))
    out.update(_points_for(g, T, kps, desc, n_mp))
    T12 = synth.to12(T)
    if sim3:
        s = np.float32(g.uniform(0.8, 1.3))
        out["pose"] = (T12 * s).astype(np.float32)  # rows 0..2 of Scw = s * [Rcw | tcw]
        out["Ow"] = np.zeros(3, np.float32)
    else:
        out["pose"] = T12
        ow = np.zeros(3, np.float32)
        for r in range(3):  # GetCameraCenter(): -Rcw.t()*tcw with double accumulation (KeyFrame.cc SetPose)
            ow[r] = np.float32(-sum(float(T12[k * 4 + r]) * float(T12[k * 4 + 3]) for k in range(3)))
        out["Ow"] = ow
    return out


def kf_target(problems, prefix="kf_", cell_start=None, cell_items=None):
    """-> (cabi.KfTarget, keepalive dict)"""
    ks = max(len(p[prefix + "kps"]) for p in problems)
    ks = max(ks, 1)
    keep = dict(n_kf=np.array([len(p[prefix + "kps"]) for p in problems], np.int32),
                kf_kps=_stack(problems, prefix + "kps", ks, cabi.KP_DTYPE),
                kf_desc=_stack(problems, prefix + "desc", ks, np.uint8, (32,)),
                kf_cell_start=np.ascontiguousarray(cell_start), kf_cell_items=np.ascontiguousarray(cell_items))
    k = cabi.KfTarget()
    fill(k, kf_stride=ks, n_levels=8, log_scale_factor=float(np.log(np.float32(SF))), **keep)
    fill(k.cam, fx=FX, fy=FY, cx=W / 2.0, cy=H / 2.0, min_x=0.0, min_y=0.0, max_x=float(W), max_y=float(H))
    fill(k.grid, **synth.front_grid_geom(W, H))
    sf, _, _, inv_sig2 = synth.scale_tables()
    fill(k, scale_factors=[float(x) for x in sf], inv_level_sigma2=[float(x) for x in inv_sig2])
    return k, keep


def mp_list(problems, suffix=""):
    ms = max(max(len(p["mp_xw" + suffix]) for p in problems), 1)
    keep = dict(n_mp=np.array([len(p["mp_xw" + suffix]) for p in problems], np.int32),
                mp_valid=_stack(problems, "mp_valid" + suffix, ms, np.uint8),
                mp_xw=_stack(problems, "mp_xw" + suffix, ms, np.float32, (3,)),
                mp_normal=_stack(problems, "mp_normal" + suffix, ms, np.float32, (3,)),
                mp_max_dist=_stack(problems, "mp_max_dist" + suffix, ms, np.float32),
                mp_min_dist=_stack(problems, "mp_min_dist" + suffix, ms, np.float32),
                mp_desc=_stack(problems, "mp_desc" + suffix, ms, np.uint8, (32,)))
    m = cabi.MpList()
    fill(m, mp_stride=ms, **keep)
    return m, keep


def fuse_args(problems, cs, ci, th=3.0):
    B = len(problems)
    k, kk = kf_target(problems, "kf_", cs, ci)
    m, mk = mp_list(problems)
    keep = dict(pose=np.ascontiguousarray(np.stack([p["pose"] for p in problems]), np.float32),
                Ow=np.ascontiguousarray(np.stack([p["Ow"] for p in problems]), np.float32))
    out = dict(best_idx=np.full((B, m.mp_stride), -7, np.int32))
    a = cabi.FuseArgs()
    fill(a, batch=B, th=th, **keep, **out)
    a.kf, a.mp = k, m
    return a, out, (kk, mk, keep)


def proj_sim3_args(problems, cs, ci, th=10):
    B = len(problems)
    k, kk = kf_target(problems, "kf_", cs, ci)
    m, mk = mp_list(problems)
    keep = dict(Scw=np.ascontiguousarray(np.stack([p["pose"] for p in problems]), np.float32),
                kf_matched=_stack(problems, "kf_matched", k.kf_stride, np.uint8))
    out = dict(match_kf_to_mp=np.full((B, k.kf_stride), -7, np.int32), nmatches=np.full(B, -7, np.int32))
    a = cabi.ProjSim3Args()
    fill(a, batch=B, th=th, **keep, **out)
    a.kf, a.mp = k, m
    return a, out, (kk, mk, keep)


# ---- SearchBySim3 ---------------------------------------------------------------------------------------------------
def make_sim3_problem(seed, n1=1500, n2=1500, n_shared=900):
    g = synth.rng(seed)
    cx, cy = W / 2.0, H / 2.0
    T2 = synth.random_pose(g)
    T1 = synth.perturb_pose(g, T2, rot_sigma=0.02, t_sigma=0.15)
    n_shared = min(n_shared, n1, n2)
    # shared world points in front of camera 2
    u2 = g.uniform(60, W - 60, n_shared)
    v2 = g.uniform(60, H - 60, n_shared)
    z2 = g.uniform(3.0, 25.0, n_shared)
    Xc2 = np.stack([(u2 - cx) / FX * z2, (v2 - cy) / FY * z2, z2], 1)
    Xw = (T2[:3, :3].T @ (Xc2 - T2[:3, 3]).T).T
    Xc1 = Xw @ T1[:3, :3].T + T1[:3, 3]
    u1 = FX * Xc1[:, 0] / Xc1[:, 2] + cx
    v1 = FY * Xc1[:, 1] / Xc1[:, 2] + cy
    base = synth.random_descriptors(g, n_shared)

    def side(n, u, v, zc):
        kps = synth.random_keypoints(g, n, W, H)
        desc = synth.random_descriptors(g, n)
        slot = g.permutation(n)[:n_shared]
        kps["x"][slot] = (u + g.normal(0, 0.8, n_shared)).astype(np.float32)
        kps["y"][slot] = (v + g.normal(0, 0.8, n_shared)).astype(np.float32)
        desc[slot] = synth.flip_bits(g, base, p=0.04)
        xw = g.uniform(-20, 20, (n, 3))
        xw[slot] = Xw + g.normal(0, 0.01, (n_shared, 3))
        mpd = synth.random_descriptors(g, n)
        mpd[slot] = synth.flip_bits(g, base, p=0.04)
        return kps, desc, slot, xw, mpd

    k1, d1, s1, xw1, md1 = side(n1, u1, v1, Xc1[:, 2])
    k2, d2, s2, xw2, md2 = side(n2, u2, v2, z2)
    # relative similarity: p_c1 = s12 R12 p_c2 + t12 (s12 = 1 up to a small estimation error)
    T12 = T1 @ np.linalg.inv(T2)
    s12 = np.float32(1.0 + g.normal(0, 0.002))
    R12 = T12[:3, :3].astype(np.float32)
    t12 = (T12[:3, 3] + g.normal(0, 0.002, 3)).astype(np.float32)

    def dists(xw, T_other, k_other, slot_self, slot_other, n):
        """invariance distances such that the level predicted in the OTHER camera brackets the other feature's octave"""
        Xc = xw @ T_other[:3, :3].T + T_other[:3, 3]
        dist = np.linalg.norm(Xc, axis=1)
        octv = np.zeros(n)
        octv[slot_self] = k_other["octave"][slot_other]
        q = octv + g.uniform(-0.4, 0.9, n)
        mx = dist * SF ** q
        return mx.astype(np.float32), (mx / SF ** 7).astype(np.float32)

    mx1, mn1 = dists(xw1, T2, k2, s1, s2, n1)
    mx2, mn2 = dists(xw2, T1, k1, s2, s1, n2)
    return dict(kps1=k1, desc1=d1, kps2=k2, desc2=d2,
                mp_valid1=(g.random(n1) < 0.85).astype(np.uint8), mp_xw1=xw1.astype(np.float32),
                mp_normal1=np.zeros((n1, 3), np.float32), mp_max_dist1=mx1, mp_min_dist1=mn1, mp_desc1=md1,
                mp_valid2=(g.random(n2) < 0.85).astype(np.uint8), mp_xw2=xw2.astype(np.float32),
                mp_normal2=np.zeros((n2, 3), np.float32), mp_max_dist2=mx2, mp_min_dist2=mn2, mp_desc2=md2,
                T1w=synth.to12(T1), T2w=synth.to12(T2), s12=s12, R12=R12.reshape(9), t12=t12)


def sim3_args(problems, grid1, grid2, th=7.5):
    B = len(problems)
    k1, kk1 = kf_target([dict(kf_kps=p["kps1"], kf_desc=p["desc1"]) for p in problems], "kf_", *grid1)
    k2, kk2 = kf_target([dict(kf_kps=p["kps2"], kf_desc=p["desc2"]) for p in problems], "kf_", *grid2)
    m1, mk1 = mp_list(problems, "1")
    m2, mk2 = mp_list(problems, "2")
    keep = dict(T1w=np.ascontiguousarray(np.stack([p["T1w"] for p in problems]), np.float32),
                T2w=np.ascontiguousarray(np.stack([p["T2w"] for p in problems]), np.float32),
                s12=np.array([p["s12"] for p in problems], np.float32),
                R12=np.ascontiguousarray(np.stack([p["R12"] for p in problems]), np.float32),
                t12=np.ascontiguousarray(np.stack([p["t12"] for p in problems]), np.float32))
    out = dict(matches12=np.full((B, m1.mp_stride), -7, np.int32), nfound=np.full(B, -7, np.int32))
    a = cabi.Sim3Args()
    fill(a, batch=B, th=th, **keep, **out)
    a.kf1, a.kf2, a.mp1, a.mp2 = k1, k2, m1, m2
    return a, out, (kk1, kk2, mk1, mk2, keep)


# ---- SearchForInitialization -------------------------------------------------------------------------------------------
def make_init_problem(seed, n1=2000, n2=2000):
    g = synth.rng(seed)
    k1 = synth.random_keypoints(g, n1, W, H)
    d1 = synth.random_descriptors(g, n1)
    k2 = synth.random_keypoints(g, n2, W, H)
    d2 = synth.random_descriptors(g, n2)
    m = min(n1, n2)
    src = g.permutation(n1)[:m]
    cp = g.random(m) < 0.7
    dst = np.arange(m)[cp]
    k2["x"][dst] = k1["x"][src[cp]] + 12.0 + g.normal(0, 6.0, int(cp.sum())).astype(np.float32)
    k2["y"][dst] = k1["y"][src[cp]] - 5.0 + g.normal(0, 6.0, int(cp.sum())).astype(np.float32)
    k2["octave"][dst] = k1["octave"][src[cp]]
    k2["angle"][dst] = np.mod(k1["angle"][src[cp]] + 8.0 + g.normal(0, 3.0, int(cp.sum())), 360.0).astype(np.float32)
    d2[dst] = synth.flip_bits(g, d1[src[cp]], p=0.03)
    # near-duplicate F1 features that compete for the same F2 feature (exercises the steal rule, :457-463)
    ndup = n1 // 10
    if ndup and m:
        a = g.integers(0, n1, ndup)
        bsrc = src[cp][g.integers(0, max(int(cp.sum()), 1), ndup)] if cp.any() else a
        for f in ("x", "y", "octave", "angle"):
            k1[f][a] = k1[f][bsrc]
        k1["x"][a] += g.normal(0, 2.0, ndup).astype(np.float32)
        d1[a] = synth.flip_bits(g, d1[bsrc], p=0.02)
    prev = np.stack([k1["x"], k1["y"]], 1).astype(np.float32)  # mvbPrevMatched = F1 key point positions (Tracking.cc:1269-1271)
    return dict(kps1=k1, desc1=d1, kps2=k2, desc2=d2, prev=prev)


def init_args(problems, cs, ci, window=100, nnratio=0.9, check_ori=1):
    B = len(problems)
    s1 = max(max(len(p["kps1"]) for p in problems), 1)
    s2 = max(max(len(p["kps2"]) for p in problems), 1)
    keep = dict(n1=np.array([len(p["kps1"]) for p in problems], np.int32), kps1=_stack(problems, "kps1", s1, cabi.KP_DTYPE),
                desc1=_stack(problems, "desc1", s1, np.uint8, (32,)),
                n2=np.array([len(p["kps2"]) for p in problems], np.int32), kps2=_stack(problems, "kps2", s2, cabi.KP_DTYPE),
                desc2=_stack(problems, "desc2", s2, np.uint8, (32,)),
                f2_cell_start=np.ascontiguousarray(cs), f2_cell_items=np.ascontiguousarray(ci))
    out = dict(prev_matched=_stack(problems, "prev", s1, np.float32, (2,)), matches12=np.full((B, s1), -7, np.int32),
               nmatches=np.full(B, -7, np.int32))
    a = cabi.InitMatchArgs()
    fill(a, batch=B, f1_stride=s1, f2_stride=s2, window_size=window, **keep, **out)
    fill(a.grid, **synth.front_grid_geom(W, H))
    fill(a.matcher, nnratio=nnratio, check_orientation=check_ori)
    return a, out, keep


# ---- ComputeDistinctiveDescriptors ----------------------------------------------------------------------------------------
def make_distinctive_problem(seed, n_mp=3000, max_obs=40, big=3):
    g = synth.rng(seed)
    counts = g.integers(0, max_obs + 1, n_mp)
    counts[g.integers(0, n_mp, big)] = g.integers(100, 300, big)
    start = np.concatenate([[0], np.cumsum(counts)]).astype(np.int32)
    base = synth.random_descriptors(g, n_mp)
    desc = np.zeros((int(start[-1]), 32), np.uint8)
    for p in range(n_mp):
        c = int(counts[p])
        if c:
            desc[start[p]:start[p + 1]] = synth.flip_bits(g, np.repeat(base[p:p + 1], c, 0), p=g.uniform(0.01, 0.2))
    return start, desc
