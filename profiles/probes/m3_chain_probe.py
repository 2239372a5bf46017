"""Probe: the M3 call of the first tracked frame of a synthetic drive through the stand-alone entry points (GPU host-pointer
call vs oracle) -- prints the slots where the two differ and the candidates of the queries involved."""
import ctypes as C
import sys
import numpy as np
import torch
sys.path.insert(0, ".")
from fishbirdeyevisualslam_amd import cabi, sequence as S, track as T, lib, check, problems as P
from oracle import pyoracle as O

seed, B = int(sys.argv[1]) if len(sys.argv) > 1 else 9100, 3
wh, bwh, fx = (640, 480), (384, 384), 250.0
seq = S.Sequence(B, 2, seed=seed, front_wh=wh, bird_wh=bwh, fx=fx, fy=fx, device="cuda:0")
tc = T.TrackChain(B, wh, bwh, K=seq.Kc, D=seq.D)
mask = torch.from_numpy(seq.mask).cuda()
f, b, c = seq.render(0)
tc.extract(f, b, c, mask)
v0 = tc.view("cur")
M, MB, mp0, mpb0, Tcw0 = seq.build_map(v0, tc.tables, map_cap=tc.map_cap, bird_cap=tc.bird_cap)
tc.k += 1
f, b, c = seq.render(1)
tc.extract(f, b, c, mask)
v1 = tc.view("cur")
L = lib()
cap = tc.cap
delta = seq.delta(1)
for bb in range(B):
    n0, n1 = int(v0["n"][bb]), int(v1["n"][bb])
    D, Lt = delta[bb].reshape(3, 4), Tcw0[bb].reshape(3, 4)
    Tc = np.zeros((3, 4), np.float32)
    f32 = np.float32
    for r in range(3):
        for cc in range(4):
            s = f32(f32(f32(D[r, 0] * Lt[0, cc]) + f32(D[r, 1] * Lt[1, cc])) + f32(D[r, 2] * Lt[2, cc]))
            s = f32(s + f32(D[r, 3] * f32(1.0 if cc == 3 else 0.0)))
            Tc[r, cc] = s
    ids = mp0[bb, :n0]
    valid = (ids >= 0).astype(np.uint8)
    sel = np.where(ids >= 0, ids, 0)
    bounds = np.zeros(4, np.float32)
    K4 = np.array(seq.Kc, np.float32); D4 = np.array(seq.D, np.float32)
    O.lib().orc_image_bounds(wh[0], wh[1], C.c_void_p(K4.ctypes.data), C.c_void_p(D4.ctypes.data), C.c_void_p(bounds.ctypes.data))
    geom = cabi.GridGeom(float(bounds[0]), float(bounds[2]), float(f32(64.0) / f32(bounds[1] - bounds[0])), float(f32(48.0) / f32(bounds[3] - bounds[2])), 64, 48)
    res = {}
    for name, fn in (("gpu", L.fb_match_projection_frame), ("orc", O.lib().orc_match_projection_frame)):
        cs = np.zeros((1, 64 * 48 + 1), np.int32); ci = np.zeros((1, cap), np.int32)
        k1 = np.zeros((1, cap), cabi.KP_DTYPE); k1[0] = v1["kps_un"][bb]
        O.grid_build(k1, np.array([n1], np.int32), 1, cap, geom, cs, ci)
        a = cabi.ProjFrameArgs()
        keep = dict(n_cur=np.array([n1], np.int32), cur_kps=k1, cur_desc=np.ascontiguousarray(v1["desc"][bb][None]), cs=cs, ci=ci,
                    Tcw=np.ascontiguousarray(Tc.reshape(1, 12)), n_last=np.array([n0], np.int32),
                    lv=np.zeros((1, cap), np.uint8), lo=np.zeros((1, cap), np.uint8), lx=np.zeros((1, cap, 3), np.float32),
                    ld=np.zeros((1, cap, 32), np.uint8), loct=np.zeros((1, cap), np.int32), lang=np.zeros((1, cap), np.float32),
                    m=np.full((1, cap), -7, np.int32), nm=np.zeros(1, np.int32))
        keep["lv"][0, :n0] = valid
        keep["lo"][0, :n0] = M["obs_pos"][bb][sel] * valid
        keep["lx"][0, :n0] = M["xw"][bb][sel]
        keep["ld"][0, :n0] = M["desc"][bb][sel]
        keep["loct"][0, :n0] = v0["kps"][bb, :n0]["octave"]
        keep["lang"][0, :n0] = v0["kps_un"][bb, :n0]["angle"]
        cabi.fill(a, batch=1, cur_stride=cap, last_stride=cap, n_cur=keep["n_cur"], cur_kps=keep["cur_kps"], cur_desc=keep["cur_desc"],
                  cur_cell_start=cs, cur_cell_items=ci, cur_blocked=None, cur_Tcw=keep["Tcw"], n_last=keep["n_last"], last_valid=keep["lv"],
                  last_obs_pos=keep["lo"], last_xw=keep["lx"], last_desc=keep["ld"], last_octave=keep["loct"], last_angle=keep["lang"], th=15.0,
                  match_cur_to_last=keep["m"], nmatches=keep["nm"], scale_factors=[tc.tables.scale_factor[i] for i in range(16)])
        cabi.fill(a.cam, fx=seq.Kc[0], fy=seq.Kc[1], cx=seq.Kc[2], cy=seq.Kc[3], min_x=float(bounds[0]), min_y=float(bounds[2]), max_x=float(bounds[1]), max_y=float(bounds[3]))
        a.grid = geom
        import os
        cabi.fill(a.matcher, nnratio=0.9, check_orientation=int(os.environ.get('ORI', '1')))
        rc = fn(C.byref(a))
        assert rc == 0, rc
        res[name] = (keep["m"][0, :n1].copy(), int(keep["nm"][0]), keep)
    g, o = res["gpu"], res["orc"]
    diff = np.nonzero(g[0] != o[0])[0]
    print("seq", bb, "n0", n0, "n1", n1, "nmatches gpu/orc", g[1], o[1], "differing slots", diff.tolist(), "gpu", g[0][diff].tolist(), "orc", o[0][diff].tolist())
    keep = g[2]
    for s in diff[:4]:
        for q in {int(g[0][s]), int(o[0][s])} - {-1}:
            d = np.unpackbits(keep["ld"][0, q] ^ v1["desc"][bb, :n1], axis=1).sum(1)
            order = np.argsort(d, kind="stable")[:5]
            X = keep["lx"][0, q].astype(np.float32)
            pc = [f32(f32(f32(f32(Tc[r, 0] * X[0]) + f32(Tc[r, 1] * X[1])) + f32(Tc[r, 2] * X[2])) + Tc[r, 3]) for r in range(3)]
            iz = f32(1.0 / float(pc[2]))
            u = f32(f32(f32(f32(seq.Kc[0]) * pc[0]) * iz) + f32(seq.Kc[2])); vv = f32(f32(f32(f32(seq.Kc[1]) * pc[1]) * iz) + f32(seq.Kc[3]))
            rad = f32(f32(15.0) * f32(tc.tables.scale_factor[int(keep["loct"][0, q])]))
            kk = v1["kps_un"][bb]
            print("      u,v", u, vv, "radius", rad, "slot pos", [(int(t), float(kk[t]["x"]), float(kk[t]["y"]), int(kk[t]["octave"]), int(d[t])) for t in (int(s),)],
                  "angle q", float(keep["lang"][0, q]), "angle slot", float(kk[int(s)]["angle"]))
            print("   slot", int(s), "query", q, "obs", int(keep["lo"][0, q]), "oct", int(keep["loct"][0, q]), "best slots by distance", [(int(i), int(d[i]), int(v1["kps"][bb, i]["octave"])) for i in order])
