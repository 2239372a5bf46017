"""Harness around the device-resident frame chain (fb_frame_*, include/fishbird.h): plays the part of the host's
Tracking state machine for B sequences side by side -- Frame construction, then the OK-state path of Tracking::Track
(TrackWithMotionModel + TrackLocalMap, Tracking.cc:1312-1441) -- with frame k's result feeding frame k+1.
torch only provides device memory and the stream.  Used by tests/ and bench.py.
"""
import ctypes as C

import numpy as np
import torch

from . import cabi, lib, check, synth
from .cabi import fill

VIEW_FIELDS = (("n", np.int32, ()), ("kps", cabi.KP_DTYPE, ("cap",)), ("kps_un", cabi.KP_DTYPE, ("cap",)), ("desc", np.uint8, ("cap", 32)),
               ("map_point", np.int32, ("cap",)), ("outlier", np.uint8, ("cap",)), ("n_bird", np.int32, ()),
               ("kps_bird", cabi.KP_DTYPE, ("cap",)), ("desc_bird", np.uint8, ("cap", 32)), ("bird_cam_xyz", np.float32, ("cap", 3)),
               ("map_point_bird", np.int32, ("cap",)), ("bird_outlier", np.uint8, ("cap",)), ("Tcw", np.float32, (12,)))


def frame_params(batch, front_wh, bird_wh, K, D, map_cap, local_mp_cap, local_mpb_cap, orb=None, bird_nfeatures=0):
    Tbc, Tcb = synth.extrinsics()
    p = cabi.FrameParams()
    fill(p, batch=batch, front_width=front_wh[0], front_height=front_wh[1], bird_width=bird_wh[0], bird_height=bird_wh[1],
         K=[float(x) for x in K], D=[float(x) for x in D], Tbc=[float(x) for x in Tbc[:3, :4].reshape(12)],
         Tcb=[float(x) for x in Tcb[:3, :4].reshape(12)], pixel2meter=synth.PIXEL2METER, meter2pixel=synth.METER2PIXEL,
         rear_axle_to_center=synth.REAR_AXLE_TO_CENTER, map_cap=map_cap, local_mp_cap=local_mp_cap, local_mpb_cap=local_mpb_cap,
         bird_nfeatures=bird_nfeatures)
    p.orb = cabi.OrbParams(**(orb or synth.ORB_DEFAULT))
    return p


def alloc_view(batch, cap):
    """Host buffers for fb_frame_download + the fb_frame_view that points at them."""
    bufs = {}
    for name, dt, shp in VIEW_FIELDS:
        bufs[name] = np.zeros((batch,) + tuple(cap if s == "cap" else s for s in shp), dt)
    bufs["counts"] = np.zeros((cabi.FB_CNT_COUNT, batch), np.int32)
    v = cabi.FrameView()
    fill(v, batch=batch, kp_stride=cap, **bufs)
    return bufs, v


class TrackChain:
    def __init__(self, batch, front_wh=(1280, 720), bird_wh=(512, 512), K=(500.0, 500.0, 640.0, 360.0), D=(0, 0, 0, 0),
                 map_cap=None, bird_cap=None, device="cuda:0", orb=None, use_lists=False, bird_nfeatures=0):
        self.L = lib()
        self.B = batch
        self.dev = torch.device(device)
        self.fw, self.fh = front_wh
        self.bw, self.bh = bird_wh
        orbp = cabi.OrbParams(**(orb or synth.ORB_DEFAULT))
        self.L.fb_orb_capacity.restype = C.c_int
        self.cap = self.L.fb_orb_capacity(C.byref(orbp))
        self.map_cap = map_cap or self.cap
        self.bird_cap = bird_cap or 2 * self.cap
        self.params = frame_params(batch, front_wh, bird_wh, K, D, self.map_cap, self.map_cap, self.bird_cap, orb, bird_nfeatures)
        self.orb_f, self.orb_b = C.c_void_p(), C.c_void_p()
        check(self.L.fb_orb_create(C.byref(orbp), C.byref(self.orb_f)), "fb_orb_create")
        orbb = cabi.OrbParams(**(orb or synth.ORB_DEFAULT))
        if bird_nfeatures:
            orbb.nfeatures = bird_nfeatures
        check(self.L.fb_orb_create(C.byref(orbb), C.byref(self.orb_b)), "fb_orb_create")
        self.tables = cabi.OrbTables()
        check(self.L.fb_orb_get_tables(self.orb_f, C.byref(self.tables)), "fb_orb_get_tables")
        # three handles: frame j lives in frames[j % 3]; with the pipelined driver frame k+1 is being constructed on the
        # extraction stream while frame k is tracked against frame k-1
        self.frames = [C.c_void_p(), C.c_void_p(), C.c_void_p()]
        for f in self.frames:
            check(self.L.fb_frame_create(C.byref(self.params), C.byref(f)), "fb_frame_create")
        self.k = 0          # frames[k % 3] is the current frame, frames[(k - 1) % 3] the last
        self._kext = 0      # next frame index the pipelined driver constructs
        self._pipe = None
        self.use_lists = use_lists
        z = lambda *s, dt=torch.uint8: torch.zeros(*s, dtype=dt, device=self.dev)
        B, mc, bc = batch, self.map_cap, self.bird_cap
        self.map = dict(n=z(B, dt=torch.int32), bad=z(B, mc), obs_pos=z(B, mc), xw=z(B, mc, 3, dt=torch.float32),
                        normal=z(B, mc, 3, dt=torch.float32), max_dist=z(B, mc, dt=torch.float32), min_dist=z(B, mc, dt=torch.float32),
                        desc=z(B, mc, 32))
        self.mpb = dict(n=z(B, dt=torch.int32), xw=z(B, bc, 3, dt=torch.float32), desc=z(B, bc, 32))
        self.local_mp, self.n_local_mp = z(B, mc, dt=torch.int32), z(B, dt=torch.int32)
        self.local_mpb, self.n_local_mpb = z(B, bc, dt=torch.int32), z(B, dt=torch.int32)
        self.delta = z(B, 12, dt=torch.float32)
        self.targs = cabi.TrackArgs()
        fill(self.targs.map, stride=mc, **self.map)
        fill(self.targs.mpb, stride=bc, **self.mpb)
        fill(self.targs, d_delta=self.delta, wB=1.0, wF=1.0, gate_local_map=1)   # if (bOK) bOK = TrackLocalMap(), per sequence
        self._hv = {}

    def close(self):
        for f in self.frames:
            if f:
                self.L.fb_frame_destroy(f)
        for h in (self.orb_f, self.orb_b):
            if h:
                self.L.fb_orb_destroy(h)
        self.frames, self.orb_f, self.orb_b = [], C.c_void_p(), C.c_void_p()

    # ---- map ----
    def set_map(self, M, MB, local_mp=None, local_mpb=None):
        up = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(self.dev)
        for k in self.map:
            self.map[k].copy_(up(M[k]))
        for k in self.mpb:
            self.mpb[k].copy_(up(MB[k]))
        if self.use_lists:
            self.local_mp.copy_(up(local_mp[0])); self.n_local_mp.copy_(up(local_mp[1]))
            self.local_mpb.copy_(up(local_mpb[0])); self.n_local_mpb.copy_(up(local_mpb[1]))
            fill(self.targs, d_local_mp=self.local_mp, d_n_local_mp=self.n_local_mp, d_local_mpb=self.local_mpb,
                 d_n_local_mpb=self.n_local_mpb)
        if hasattr(self, "targs_kf"):
            d = self.targs_kf.d_delta
            C.memmove(C.byref(self.targs_kf), C.byref(self.targs), C.sizeof(self.targs))
            self.targs_kf.d_delta = d
        torch.cuda.synchronize()

    @property
    def cur(self):
        return self.frames[self.k % 3]

    @property
    def last(self):
        return self.frames[(self.k - 1) % 3]

    # ---- pipelined driver: Frame construction of frame k+1 (extraction stream) beside the tracking of frame k ----
    _streams = {}

    def _pipe_init(self):
        if self._pipe is None:
            key = str(self.dev)
            if key not in TrackChain._streams:  # shared per device: HIP folds streams onto few hardware queues
                TrackChain._streams[key] = (torch.cuda.Stream(device=self.dev), torch.cuda.Stream(device=self.dev))
            sE, sT = TrackChain._streams[key]
            self._pipe = dict(sE=sE, sT=sT, evE=[torch.cuda.Event() for _ in range(3)], evT=[torch.cuda.Event() for _ in range(3)],
                              tracked=[False] * 3)
            self._kext = self.k
        return self._pipe

    def prefetch(self, front, bird, contour=None, mask=None):
        """Frame::Frame of the next frame on the extraction stream (the images do not depend on the tracking results)."""
        P = self._pipe_init()
        j = self._kext
        h = j % 3
        sE = P["sE"]
        sE.wait_stream(torch.cuda.current_stream(self.dev))
        if P["tracked"][(j + 1) % 3]:  # handle h was `last` for the frame in handle (j + 1) % 3 == (j - 2) % 3
            sE.wait_event(P["evT"][(j + 1) % 3])
        with torch.cuda.stream(sE):
            self.extract(front, bird, contour, mask, frame=self.frames[h])
            P["evE"][h].record(sE)
        self._kext = j + 1

    def track_prefetched(self, delta_dev=None, sync=False):
        """Track frame k (constructed by prefetch) against frame k-1 on the tracking stream."""
        P = self._pipe_init()
        h = self.k % 3
        sT = P["sT"]
        sT.wait_event(P["evE"][h])
        with torch.cuda.stream(sT):
            if delta_dev is not None:
                self.delta.copy_(delta_dev, non_blocking=True)
            check(self.L.fb_frame_track_dev(self.cur, self.last, C.byref(self.targs), C.c_void_p(sT.cuda_stream)), "fb_frame_track_dev")
            P["evT"][h].record(sT)
            P["tracked"][h] = True
            self.k += 1
            if sync:
                return self.counts()
        return None

    def _stream(self):
        return C.c_void_p(torch.cuda.current_stream(self.dev).cuda_stream)

    # ---- steps ----
    def extract(self, front, bird, contour=None, mask=None, frame=None):
        """Frame::Frame on device images ([B,h,w] u8 tensors)."""
        f = frame or self.cur
        vp = lambda t: C.c_void_p(t.data_ptr()) if t is not None else None
        check(self.L.fb_frame_extract_dev(f, self.orb_f, self.orb_b, vp(front), self.fw, C.c_size_t(self.fw * self.fh), vp(bird), self.bw,
                                          C.c_size_t(self.bw * self.bh), vp(contour), vp(mask), self._stream()), "fb_frame_extract_dev")

    def init_first(self, mp0, mpb0, Tcw0):
        """The current frame enters the chain as an already tracked frame (what initialisation leaves behind)."""
        up = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(self.dev)
        a, b, t = up(mp0), up(mpb0), up(Tcw0)
        check(self.L.fb_frame_set_map_points_dev(self.cur, C.c_void_p(a.data_ptr()), C.c_void_p(b.data_ptr()), self._stream()), "set_map_points")
        check(self.L.fb_frame_set_pose_dev(self.cur, C.c_void_p(t.data_ptr()), self._stream()), "set_pose")
        torch.cuda.synchronize()
        self.k += 1

    def set_delta(self, delta):
        self.delta.copy_(torch.from_numpy(np.ascontiguousarray(delta)).to(self.dev), non_blocking=True)

    def track(self, front, bird, contour=None, mask=None):
        """One tracked frame: Frame construction + fb_frame_track_dev against the previous frame; then it is the last frame."""
        self.extract(front, bird, contour, mask)
        check(self.L.fb_frame_track_dev(self.cur, self.last, C.byref(self.targs), self._stream()), "fb_frame_track_dev")
        self.k += 1

    # ---- reference key frame (Tracking::TrackReferenceKeyFrame, Tracking.cc:1180-1244) ----
    def set_vocabulary(self, voc_arrays, L):
        """Upload a vocabulary (the arrays of cabi.Vocabulary as numpy) once; kept for the chain's lifetime."""
        up = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(self.dev)
        self._voc_keep = {k: up(v) for k, v in voc_arrays.items()}
        self.voc = cabi.Vocabulary()
        fill(self.voc, n_nodes=len(voc_arrays["weights"]), L=L, **self._voc_keep)
        self.kf = C.c_void_p()
        check(self.L.fb_frame_create(C.byref(self.params), C.byref(self.kf)), "fb_frame_create")
        self.frames.append(self.kf)   # (destroyed with the others; the rotation only uses the first three)
        self.delta_kf = torch.zeros(self.B, 12, dtype=torch.float32, device=self.dev)
        self.targs_kf = cabi.TrackArgs.from_buffer_copy(self.targs)
        fill(self.targs_kf, d_delta=self.delta_kf)

    def make_keyframe(self, which="last"):
        """KeyFrame(mCurrentFrame, ...) + ComputeBoW (KeyFrame.cc:32-102) from the frame tracked most recently."""
        check(self.L.fb_frame_copy_dev(self.kf, self._which(which), self._stream()), "fb_frame_copy_dev")
        check(self.L.fb_frame_compute_bow_dev(self.kf, C.byref(self.voc), self._stream()), "fb_frame_compute_bow_dev")

    def set_delta_kf(self, delta):
        self.delta_kf.copy_(torch.from_numpy(np.ascontiguousarray(delta)).to(self.dev), non_blocking=True)

    def track_modes(self, front, bird, contour=None, mask=None, mode="motion"):
        """The host's choice of Tracking.cc:529-540 made by the caller: mode = "motion" (TrackWithMotionModel), "reference"
        (TrackReferenceKeyFrame) or "motion+reference" (the fall-back when the motion model fails), then TrackLocalMap."""
        L, s, cur, last = self.L, self._stream(), self.cur, self.last
        self.extract(front, bird, contour, mask)
        if mode in ("bird", "bird_kf"):   # Tracking::TrackUsingBird (Tracking.cc:2014-2061): no TrackLocalMap behind it (:556)
            src, args = (self.kf, self.targs_kf) if mode == "bird_kf" else (last, self.targs)
            check(L.fb_frame_track_using_bird_dev(cur, src, last, C.byref(args), s), "fb_frame_track_using_bird_dev")
            self.k += 1
            return
        if "motion" in mode:
            check(L.fb_frame_track_motion_model_dev(cur, last, C.byref(self.targs), s), "fb_frame_track_motion_model_dev")
        if "reference" in mode:
            check(L.fb_frame_track_reference_dev(cur, self.kf, last, C.byref(self.voc), C.byref(self.targs_kf), s), "fb_frame_track_reference_dev")
        check(L.fb_frame_track_local_map_dev(cur, last, C.byref(self.targs), s), "fb_frame_track_local_map_dev")
        self.k += 1

    def track_granular(self, front, bird, contour=None, mask=None):
        """The same chain through the one-call-per-reference-function entry points (every commit is its own launch)."""
        L, s, cur, last, T = self.L, self._stream(), self.cur, self.last, self.targs
        self.extract(front, bird, contour, mask)
        m09, m08 = cabi.MatcherParams(0.9, 1), cabi.MatcherParams(0.8, 1)
        vp = lambda x: C.c_void_p(x) if x else None
        lm = (vp(T.d_local_mp), vp(T.d_n_local_mp)) if self.use_lists else (None, None)
        lb = (vp(T.d_local_mpb), vp(T.d_n_local_mpb)) if self.use_lists else (None, None)
        check(L.fb_frame_predict_pose_dev(cur, last, C.c_void_p(self.delta.data_ptr()), s), "predict")
        check(L.fb_frame_bird_mappoint_match_dev(cur, C.byref(T.mpb), lb[0], lb[1], 10, C.c_float(0.05), C.byref(m09), s), "M9")
        check(L.fb_frame_search_by_projection_dev(cur, last, C.byref(T.map), C.c_float(15.0), C.byref(m09), s), "M3")
        # the host's part of Tracking.cc:1342-1352 (wide-window retry, return false below 20 matches) is per sequence; these
        # entry points move all sequences of the handle in lockstep, so a batch that diverges there must be split by the host
        if (self.counts("cur")[0][cabi.FB_CNT["PROJ_MATCHES"]] < 20).any():
            raise NotImplementedError("granular entry points: a sequence has fewer than 20 matches (retry / early return is per sequence)")
        check(L.fb_frame_pose_optimization_dev(cur, C.byref(T.map), C.byref(T.mpb), cabi.FB_POSE_FRONT_BIRD, C.c_float(1.0), C.c_float(1.0), 0, s), "pose 1")
        check(L.fb_frame_discard_outliers_dev(cur, C.byref(T.map), s), "discard")
        if (self.counts("cur")[0][cabi.FB_CNT["MATCHES_MAP"]] < 10).any():
            raise NotImplementedError("granular entry points: TrackWithMotionModel failed for a sequence (the fall-back is per sequence)")
        check(L.fb_frame_match_bird_points_dev(cur, last, C.byref(T.mpb), 10, C.c_float(0.05), C.byref(m09), s), "M8 + filter")
        check(L.fb_frame_search_local_points_dev(cur, C.byref(T.map), lm[0], lm[1], C.c_float(1.0), C.byref(m08), s), "local points")
        check(L.fb_frame_pose_optimization_dev(cur, C.byref(T.map), C.byref(T.mpb), cabi.FB_POSE_FRONT_BIRD, C.c_float(1.0), C.c_float(1.0), 1, s), "pose 2")
        check(L.fb_frame_finish_dev(cur, C.byref(T.map), s), "finish")
        self.k += 1

    def drop_outliers(self, which="last"):
        """Tracking.cc:721-725 after the host's key-frame decision (targs.defer_outlier_drop = 1)."""
        check(self.L.fb_frame_drop_outliers_dev(self._which(which), self._stream()), "fb_frame_drop_outliers_dev")

    # ---- results ----
    def _which(self, which):
        return {"last": self.last, "cur": self.cur, "prev": self.frames[(self.k - 2) % 3], "kf": getattr(self, "kf", None)}[which]

    def view(self, which="last"):
        """last = the frame tracked most recently, prev = the one before it (the reference frame of that step), cur = the next handle."""
        f = self._which(which)
        if "bufs" not in self._hv:
            self._hv["bufs"], self._hv["v"] = alloc_view(self.B, self.cap)
        check(self.L.fb_frame_download(f, C.byref(self._hv["v"]), self._stream()), "fb_frame_download")
        return {k: v.copy() for k, v in self._hv["bufs"].items()}

    def counts(self, which="last"):
        f = self._which(which)
        c = np.zeros((cabi.FB_CNT_COUNT, self.B), np.int32)
        t = np.zeros((self.B, 12), np.float32)
        check(self.L.fb_frame_counts(f, C.c_void_p(c.ctypes.data), C.c_void_p(t.ctypes.data), self._stream()), "fb_frame_counts")
        return c, t

    def bird_table_host(self):
        torch.cuda.synchronize()
        return {k: v.cpu().numpy() for k, v in self.mpb.items()}
