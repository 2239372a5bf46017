"""Device-resident per-frame hot path: extract(front) + extract(bird) + grids + front match (M3)
+ bird match (M9) + edge gather + PoseOptimizationWithBird, batched over B independent frame
pairs (one tracked sequence each), every stage through the C-ABI (*_dev entry points).

This is the order Tracking::GrabImageMonocularWithOdom -> TrackWithMotionModel runs them
(Tracking.cc:292-339, 1312-1385).  torch is used only for device memory and the stream.
"""
import ctypes as C

import numpy as np
import torch

from . import cabi, lib, check, synth
from .cabi import fill


def _vp(t):
    return C.c_void_p(t.data_ptr())


class FramePipeline:
    _streams = {}  # device -> (front, bird, pose) streams

    def __init__(self, batch, front_wh=(1280, 720), bird_wh=(512, 512), n_last=2000, n_ref=1000, device="cuda:0",
                 fx=500.0, fy=500.0, orb=None):
        self.L = lib()
        self.B = batch
        self.dev = torch.device(device)
        self.fw, self.fh = front_wh
        self.bw, self.bh = bird_wh
        self.fx, self.fy, self.cx, self.cy = fx, fy, self.fw / 2.0, self.fh / 2.0
        self.params = cabi.OrbParams(**(orb or synth.ORB_DEFAULT))
        self.L.fb_orb_capacity.restype = C.c_int
        self.cap = self.L.fb_orb_capacity(C.byref(self.params))
        self.orb_f, self.orb_b = C.c_void_p(), C.c_void_p()
        check(self.L.fb_orb_create(C.byref(self.params), C.byref(self.orb_f)), "fb_orb_create")
        check(self.L.fb_orb_create(C.byref(self.params), C.byref(self.orb_b)), "fb_orb_create")
        self.tables = cabi.OrbTables()
        check(self.L.fb_orb_get_tables(self.orb_f, C.byref(self.tables)), "fb_orb_get_tables")
        self.Tbc, self.Tcb = synth.extrinsics()
        B, cap, d = batch, self.cap, self.dev
        z = lambda *s, dt=torch.uint8: torch.zeros(*s, dtype=dt, device=d)
        self.f_img, self.b_img = z(B, self.fh, self.fw), z(B, self.bh, self.bw)
        self.f_kps, self.b_kps = z(B, cap * 24), z(B, cap * 24)
        self.f_desc, self.b_desc = z(B, cap, 32), z(B, cap, 32)
        self.f_n, self.b_n = z(B, dt=torch.int32), z(B, dt=torch.int32)
        self.geom_f = fill(cabi.GridGeom(), **synth.front_grid_geom(self.fw, self.fh))
        self.geom_b = fill(cabi.GridGeom(), **synth.bird_grid_geom(self.bw, self.bh))
        self.f_cs = z(B, self.geom_f.cols * self.geom_f.rows + 1, dt=torch.int32)
        self.b_cs = z(B, self.geom_b.cols * self.geom_b.rows + 1, dt=torch.int32)
        self.f_ci, self.b_ci = z(B, cap, dt=torch.int32), z(B, cap, dt=torch.int32)
        self.b_cam = z(B, cap, 3, dt=torch.float32)
        # last frame / reference bird map points (filled by set_map)
        self.nl, self.nr = n_last, n_ref
        self.last = dict(valid=z(B, n_last), obs=z(B, n_last), xw=z(B, n_last, 3, dt=torch.float32), desc=z(B, n_last, 32),
                         octave=z(B, n_last, dt=torch.int32), angle=z(B, n_last, dt=torch.float32), n=z(B, dt=torch.int32))
        self.ref = dict(valid=z(B, n_ref), xw=z(B, n_ref, 3, dt=torch.float32), desc=z(B, n_ref, 32), n=z(B, dt=torch.int32))
        self.Tcw0 = z(B, 12, dt=torch.float32)
        # match buffers
        self.m_front, self.nm_front = z(B, cap, dt=torch.int32), z(B, dt=torch.int32)
        self.m_bird, self.nm_bird = z(B, cap, dt=torch.int32), z(B, dt=torch.int32)
        # pose stage: edge arrays, pose, outlier flags and counts exist TWICE (set k & 1 belongs to step k), so that the edge
        # gather of step k + 1 does not have to wait for the pose kernel of step k to let go of its inputs: consecutive pose
        # kernels run back to back (at B = 1 the wait was 60 us of a 398 us step)
        self._sets = []
        for _ in range(2):
            self._sets.append(dict(
                Tcw=z(B, 12, dt=torch.float32), e_fxw=z(B, cap, 3, dt=torch.float32), e_fobs=z(B, cap, 2, dt=torch.float32),
                e_finf=z(B, cap, dt=torch.float32), e_fvalid=z(B, cap), e_fout=z(B, cap), e_bxw=z(B, cap, 3, dt=torch.float32),
                e_bxc=z(B, cap, 3, dt=torch.float32), e_binf=z(B, cap, dt=torch.float32), e_bvalid=z(B, cap), e_bout=z(B, cap),
                ninl=z(B, dt=torch.int32), e_nf=z(B, dt=torch.int32), e_nb=z(B, dt=torch.int32), evP=torch.cuda.Event(), pending=False))
        self._k = 0  # step counter; results_host() reads the set of the last step
        # three streams: front chain, bird chain, pose optimisation (latency-bound, overlaps the next extraction)
        # (the three streams are shared by every pipeline of a device: the HIP runtime folds streams onto a few hardware queues,
        # and streams that land on one queue serialise -- a second pipeline with streams of its own measured 0.72 instead of
        # 0.47 ms per step at B = 8)
        key = str(d)
        if key not in FramePipeline._streams:
            FramePipeline._streams[key] = tuple(torch.cuda.Stream(device=d) for _ in range(3))
        self.sF, self.sB, self.sP = FramePipeline._streams[key]
        self.evF, self.evB = (torch.cuda.Event() for _ in range(2))
        self._inv_sigma2 = (C.c_float * cabi.FB_MAX_LEVELS)(*self.tables.inv_level_sigma2)
        self._Tcb12 = (C.c_float * 12)(*[float(x) for x in self.Tcb[:3, :4].reshape(12)])
        self._build_args()

    def close(self):
        for h in (self.orb_f, self.orb_b):
            if h:
                self.L.fb_orb_destroy(h)
        self.orb_f = self.orb_b = C.c_void_p()

    # ---- argument structs (pointers are fixed for the lifetime of the pipeline) ----
    def _build_args(self):
        B, cap = self.B, self.cap
        a = cabi.ProjFrameArgs()
        fill(a, batch=B, cur_stride=cap, last_stride=self.nl, n_cur=self.f_n, cur_kps=self.f_kps, cur_desc=self.f_desc,
             cur_cell_start=self.f_cs, cur_cell_items=self.f_ci, cur_blocked=None, cur_Tcw=self.Tcw0,
             n_last=self.last["n"], last_valid=self.last["valid"], last_obs_pos=self.last["obs"], last_xw=self.last["xw"],
             last_desc=self.last["desc"], last_octave=self.last["octave"], last_angle=self.last["angle"], th=15.0,
             match_cur_to_last=self.m_front, nmatches=self.nm_front,
             scale_factors=[self.tables.scale_factor[i] for i in range(cabi.FB_MAX_LEVELS)])
        fill(a.cam, fx=self.fx, fy=self.fy, cx=self.cx, cy=self.cy, min_x=0.0, min_y=0.0, max_x=float(self.fw), max_y=float(self.fh))
        a.grid = self.geom_f
        fill(a.matcher, nnratio=0.9, check_orientation=1)  # ORBmatcher matcher(0.9,true), Tracking.cc:1339
        self.a_m3 = a
        m = cabi.BirdMpArgs()
        fill(m, batch=B, cur_stride=cap, ref_stride=self.nr, n_cur=self.b_n, cur_kps=self.b_kps, cur_desc=self.b_desc,
             cur_cam_xyz=self.b_cam, cur_cell_start=self.b_cs, cur_cell_items=self.b_ci, cur_Tcw=self.Tcw0,
             n_ref=self.ref["n"], ref_valid=self.ref["valid"], ref_xw=self.ref["xw"], ref_desc=self.ref["desc"],
             Tbc=[float(x) for x in self.Tbc[:3, :4].reshape(12)], bird_cols=self.bw, bird_rows=self.bh,
             meter2pixel=synth.METER2PIXEL, rear_axle_to_center=synth.REAR_AXLE_TO_CENTER, window_size=10,
             filter_size=0.05, match_cur_to_ref=self.m_bird, ninliers=self.nm_bird)
        m.grid = self.geom_b
        fill(m.matcher, nnratio=0.9, check_orientation=1)  # Tracking.cc:2008
        self.a_m9 = m
        for S in self._sets:
            p = cabi.PoseOptArgs()
            fill(p, batch=B, mode=cabi.FB_POSE_FRONT_BIRD, front_stride=cap, bird_stride=cap, fx=self.fx, fy=self.fy,
                 cx=self.cx, cy=self.cy, wF=1.0, wB=1.0, n_front=S["e_nf"], front_xw=S["e_fxw"], front_obs=S["e_fobs"],
                 front_inv_sigma2=S["e_finf"], front_valid=S["e_fvalid"], n_bird=S["e_nb"], bird_xw=S["e_bxw"],
                 bird_xc=S["e_bxc"], bird_inv_sigma2=S["e_binf"], bird_valid=S["e_bvalid"], bird_outlier=S["e_bout"],
                 Tcw=S["Tcw"], front_outlier=S["e_fout"], ninliers=S["ninl"])
            S["a_pose"] = p

    # ---- stages ----
    def extract(self, s, which="both"):
        L, B = self.L, self.B
        if which in ("both", "front"):
            check(L.fb_orb_extract_batch_dev(self.orb_f, _vp(self.f_img), B, self.fw, self.fh, self.fw, C.c_size_t(self.fw * self.fh),
                                             _vp(self.f_kps), _vp(self.f_desc), _vp(self.f_n), s), "extract front")
        if which in ("both", "bird"):
            check(L.fb_orb_extract_batch_dev(self.orb_b, _vp(self.b_img), B, self.bw, self.bh, self.bw, C.c_size_t(self.bw * self.bh),
                                             _vp(self.b_kps), _vp(self.b_desc), _vp(self.b_n), s), "extract bird")

    def grids(self, s, which="both"):
        L, B, cap = self.L, self.B, self.cap
        if which in ("both", "front"):
            check(L.fb_grid_build_batch_dev(_vp(self.f_kps), _vp(self.f_n), B, cap, C.byref(self.geom_f), _vp(self.f_cs), _vp(self.f_ci), s), "grid front")
        if which in ("both", "bird"):
            check(L.fb_grid_build_batch_dev(_vp(self.b_kps), _vp(self.b_n), B, cap, C.byref(self.geom_b), _vp(self.b_cs), _vp(self.b_ci), s), "grid bird")
            check(L.fb_bird_keys_to_cam_dev(_vp(self.b_kps), _vp(self.b_n), B, cap, self.bw, self.bh, C.c_double(synth.PIXEL2METER),
                                            C.c_double(synth.REAR_AXLE_TO_CENTER), self._Tcb12, _vp(self.b_cam), s), "bird cam")

    def match_front(self, s):
        check(self.L.fb_match_projection_frame_dev(C.byref(self.a_m3), s), "M3")

    def match_bird(self, s):
        self.m_bird.fill_(-1)  # mvpMapPointsBird starts empty for a new frame
        check(self.L.fb_match_bird_mappoints_dev(C.byref(self.a_m9), s), "M9")

    def gather_front(self, s, S):
        L, B, cap, nl = self.L, self.B, self.cap, self.params.nlevels
        S["e_nf"].copy_(self.f_n)
        S["Tcw"].copy_(self.Tcw0)   # SetPose(prediction), Tracking.cc:1314-1320
        check(L.fb_pose_gather_front_dev(B, cap, self.nl, _vp(self.f_n), _vp(self.f_kps), _vp(self.m_front), _vp(self.last["xw"]),
                                         self._inv_sigma2, nl, _vp(S["e_fxw"]), _vp(S["e_fobs"]), _vp(S["e_finf"]), _vp(S["e_fvalid"]), s), "gather front")

    def gather_bird(self, s, S):
        L, B, cap, nl = self.L, self.B, self.cap, self.params.nlevels
        S["e_nb"].copy_(self.b_n)
        S["e_bout"].fill_(1)        # mvBirdOutlier = vector<bool>(Nbird, true) of a fresh Frame (Frame.cc:356)
        check(L.fb_pose_gather_bird_dev(B, cap, self.nr, _vp(self.b_n), _vp(self.b_kps), _vp(self.b_cam), _vp(self.m_bird), _vp(self.ref["xw"]),
                                        self._inv_sigma2, nl, _vp(S["e_bxw"]), _vp(S["e_bxc"]), _vp(S["e_binf"]), _vp(S["e_bvalid"]), s), "gather bird")

    def pose(self, s, S):
        check(self.L.fb_pose_opt_batch_dev(C.byref(S["a_pose"]), s), "pose opt")

    def step(self):
        """One pass of the hot path over the batch.  The front chain and the bird chain run on their own streams;
        the pose optimisation (one workgroup per frame, latency bound) runs on a third one so that it overlaps the
        next step's extraction.  Callers synchronise with torch.cuda.synchronize() / results_host()."""
        cur = torch.cuda.current_stream(self.dev)
        S = self._sets[self._k & 1]
        self.sF.wait_stream(cur)
        self.sB.wait_stream(cur)
        with torch.cuda.stream(self.sF):
            s = C.c_void_p(self.sF.cuda_stream)
            self.extract(s, "front")
            self.grids(s, "front")
            self.match_front(s)
            if S["pending"]:
                self.sF.wait_event(S["evP"])  # the pose kernel of two steps ago read this set's edge arrays
            self.gather_front(s, S)
            self.evF.record(self.sF)
        with torch.cuda.stream(self.sB):
            s = C.c_void_p(self.sB.cuda_stream)
            self.extract(s, "bird")
            self.grids(s, "bird")
            self.match_bird(s)
            if S["pending"]:
                self.sB.wait_event(S["evP"])
            self.gather_bird(s, S)
            self.evB.record(self.sB)
        with torch.cuda.stream(self.sP):
            self.sP.wait_event(self.evF)
            self.sP.wait_event(self.evB)
            self.pose(C.c_void_p(self.sP.cuda_stream), S)
            S["evP"].record(self.sP)
            S["pending"] = True
        self._last = S
        self._k += 1

    def step_serial(self):
        """The same pass on torch's current stream only (used by tests and for single-stream timing)."""
        cur = torch.cuda.current_stream(self.dev)
        s = C.c_void_p(cur.cuda_stream)
        S = self._sets[self._k & 1]
        if S["pending"]:  # a pose kernel of an overlapped step() may still read this set's edge arrays on the pose stream
            cur.wait_event(S["evP"])
            S["pending"] = False
        self.extract(s)
        self.grids(s)
        self.match_front(s)
        self.match_bird(s)
        self.gather_front(s, S)
        self.gather_bird(s, S)
        self.pose(s, S)
        self._last = S
        self._k += 1

    # ---- synthetic world (untimed setup) ----
    def set_images(self, front, bird):
        self.f_img.copy_(torch.from_numpy(np.ascontiguousarray(front)).to(self.dev))
        self.b_img.copy_(torch.from_numpy(np.ascontiguousarray(bird)).to(self.dev))

    def keypoints_host(self, which="front"):
        kps, desc, n = (self.f_kps, self.f_desc, self.f_n) if which == "front" else (self.b_kps, self.b_desc, self.b_n)
        n = n.cpu().numpy()
        k = kps.cpu().numpy().view(cabi.KP_DTYPE).reshape(self.B, self.cap)
        return [k[b, : n[b]].copy() for b in range(self.B)], [desc[b, : n[b]].cpu().numpy() for b in range(self.B)]

    def build_world(self, seed=5000, rot_sigma=0.002, t_sigma=0.01, outlier_frac=0.1):
        """From one extraction of the current images, synthesise the previous frame's map points and the
        reference bird map points so that matching and pose optimisation have real work."""
        torch.cuda.synchronize()
        s = C.c_void_p(torch.cuda.current_stream(self.dev).cuda_stream)
        self.extract(s)
        self.grids(s)
        torch.cuda.synchronize()
        fk, fd = self.keypoints_host("front")
        bk, bd = self.keypoints_host("bird")
        bcam = self.b_cam.cpu().numpy()
        B, nl, nr = self.B, self.nl, self.nr
        world = []
        for b in range(B):
            g = synth.rng(seed + b)
            T = synth.random_pose(g)
            R, t = T[:3, :3], T[:3, 3]
            k, dsc = fk[b], fd[b]
            m = min(nl, len(k))
            sel = g.permutation(len(k))[:m]
            z = g.uniform(2.0, 30.0, m)
            u = k["x"][sel].astype(np.float64) + g.normal(0, 0.5, m)
            v = k["y"][sel].astype(np.float64) + g.normal(0, 0.5, m)
            bad = g.random(m) < outlier_frac
            u[bad] += g.uniform(-20, 20, int(bad.sum()))
            v[bad] += g.uniform(-20, 20, int(bad.sum()))
            Xc = np.stack([(u - self.cx) / self.fx * z, (v - self.cy) / self.fy * z, z], 1)
            Xw = (R.T @ (Xc - t).T).T
            w = dict(T_true=T, Tcw0=synth.to12(synth.perturb_pose(g, T, rot_sigma, t_sigma)), n_last=m,
                     last_xw=Xw.astype(np.float32), last_desc=synth.flip_bits(g, dsc[sel]),
                     last_octave=k["octave"][sel].astype(np.int32),
                     last_angle=np.mod(k["angle"][sel] + 7.0 + g.normal(0, 3.0, m), 360.0).astype(np.float32))
            kb, db = bk[b], bd[b]
            mb = min(nr, len(kb))
            selb = g.permutation(len(kb))[:mb]
            pc = bcam[b, selb].astype(np.float64) + g.normal(0, 0.005, (mb, 3))
            w.update(n_ref=mb, ref_xw=((R.T @ (pc - t).T).T).astype(np.float32), ref_desc=synth.flip_bits(g, db[selb], p=0.05))
            world.append(w)
        self.set_world(world)
        return world

    def set_world(self, world):
        B = self.B
        up = lambda arr: torch.from_numpy(np.ascontiguousarray(arr)).to(self.dev)

        def stack(key, n, shape, dt):
            out = np.zeros((B, n) + shape, dt)
            for b, w in enumerate(world):
                out[b, : len(w[key])] = w[key]
            return out
        self.last["n"].copy_(up(np.array([w["n_last"] for w in world], np.int32)))
        self.last["xw"].copy_(up(stack("last_xw", self.nl, (3,), np.float32)))
        self.last["desc"].copy_(up(stack("last_desc", self.nl, (32,), np.uint8)))
        self.last["octave"].copy_(up(stack("last_octave", self.nl, (), np.int32)))
        self.last["angle"].copy_(up(stack("last_angle", self.nl, (), np.float32)))
        self.last["valid"].fill_(1)
        self.last["obs"].fill_(1)
        self.ref["n"].copy_(up(np.array([w["n_ref"] for w in world], np.int32)))
        self.ref["xw"].copy_(up(stack("ref_xw", self.nr, (3,), np.float32)))
        self.ref["desc"].copy_(up(stack("ref_desc", self.nr, (32,), np.uint8)))
        self.ref["valid"].fill_(1)
        self.Tcw0.copy_(up(np.stack([w["Tcw0"] for w in world])))
        torch.cuda.synchronize()

    def results_host(self):
        torch.cuda.synchronize()
        S = self._last
        return dict(n_front=self.f_n.cpu().numpy(), n_bird=self.b_n.cpu().numpy(), nm_front=self.nm_front.cpu().numpy(),
                    nm_bird=self.nm_bird.cpu().numpy(), ninliers=S["ninl"].cpu().numpy(), Tcw=S["Tcw"].cpu().numpy(),
                    front_outlier=S["e_fout"].cpu().numpy(), bird_outlier=S["e_bout"].cpu().numpy(),
                    m_front=self.m_front.cpu().numpy(), m_bird=self.m_bird.cpu().numpy())
