"""Synthetic SEQUENCES for the tracking chain (harness only: tests/ and bench.py).

One static world -- a textured ground plane z = 0 -- seen by both cameras of a vehicle that drives over it:
  * the front fisheye camera (K, D of the settings file, extrinsics Frame::Tbc, Frame.cc:1015-1037: 0.74 m above the
    ground, pitched 35 degrees down) renders the plane through the fisheye model, so that undistorted key points of
    consecutive frames are consistent with ONE rigid motion;
  * the bird's-eye image is the orthographic top view around the vehicle in the reference's pixel <-> base-frame
    convention (Converter::BirdPixel2BaseXY, Converter.cc:284-292), plus a contour image (mBirdviewContourICP) and a
    detect mask with the vehicle body blanked.
Frame k's images therefore really are frame k-1's images after the odometry step, which is what makes the chain
(pose prediction -> projection matching -> pose optimisation -> next frame) meaningful.  Rendering uses torch (float64)
on whatever device the caller passes; everything a parity test compares against the oracle is downloaded from the
rendered bytes, never re-rendered.
"""
import math

import numpy as np
import torch

from . import synth

FISHEYE_D = (-0.0488316, 0.000298406, -0.00591118, 0.00193258)  # Examples/Monocular/fisheye.yaml:13-16 (k1 k2 p1 p2 -> k1..k4)
TEXEL = 0.01          # metres per ground texel
GROUND_X = (-12.0, 44.0)
GROUND_Y = (-18.0, 18.0)


def make_ground(seed):
    """u8 ground texture [ny, nx] (row = y texel, col = x texel) and its contour map (edges of the shapes, dilated)."""
    g = synth.rng(seed)
    nx = int(round((GROUND_X[1] - GROUND_X[0]) / TEXEL))
    ny = int(round((GROUND_Y[1] - GROUND_Y[0]) / TEXEL))
    img = np.full((ny, nx), 128, np.int16)
    n_rect, n_disc = 10000, 3600
    rw, rh = g.integers(8, 90, n_rect), g.integers(8, 90, n_rect)
    rx, ry = g.integers(-8, nx, n_rect), g.integers(-8, ny, n_rect)
    rv = g.integers(16, 240, n_rect)
    for i in range(n_rect):
        img[max(ry[i], 0):max(ry[i] + rh[i], 0), max(rx[i], 0):max(rx[i] + rw[i], 0)] = rv[i]
    dr, dx, dy, dv = g.integers(4, 30, n_disc), g.integers(0, nx, n_disc), g.integers(0, ny, n_disc), g.integers(16, 240, n_disc)
    for i in range(n_disc):
        r = int(dr[i])
        y0, y1, x0, x1 = max(dy[i] - r, 0), min(dy[i] + r + 1, ny), max(dx[i] - r, 0), min(dx[i] + r + 1, nx)
        yy, xx = np.ogrid[y0:y1, x0:x1]
        img[y0:y1, x0:x1][(yy - dy[i]) ** 2 + (xx - dx[i]) ** 2 <= r * r] = dv[i]
    noise = g.integers(-4, 5, size=(ny // 4, nx // 4), dtype=np.int16)  # 4 cm grain
    img += np.repeat(np.repeat(noise, 4, 0), 4, 1)
    tex = np.clip(img, 0, 255).astype(np.uint8)
    # contour map at 4 cm per cell: a cell is an edge cell when the texture has a step inside it or next to it
    q = img[: ny // 4 * 4, : nx // 4 * 4]
    gx = np.zeros(q.shape, bool)
    gy = np.zeros(q.shape, bool)
    gx[:, 1:] = np.abs(q[:, 1:] - q[:, :-1]) > 24
    gy[1:, :] = np.abs(q[1:, :] - q[:-1, :]) > 24
    e = (gx | gy).reshape(ny // 4, 4, nx // 4, 4).any(axis=(1, 3))
    d = e.copy()
    d[1:] |= e[:-1]; d[:-1] |= e[1:]
    e = d.copy()
    e[:, 1:] |= d[:, :-1]; e[:, :-1] |= d[:, 1:]
    contour = np.where(e, 100, 0).astype(np.uint8)
    # some free-space blobs (>= 150), as the reference's contour image carries both labels (Frame.cc:686-715)
    for i in range(0, n_disc, 16):
        r = int(dr[i]) * 3 // 4
        cy_, cx_ = dy[i] // 4, dx[i] // 4
        y0, y1, x0, x1 = max(cy_ - r, 0), min(cy_ + r + 1, ny // 4), max(cx_ - r, 0), min(cx_ + r + 1, nx // 4)
        contour[y0:y1, x0:x1] = np.maximum(contour[y0:y1, x0:x1], 200)
    return tex, contour


def odom_delta(p1, p2, Tbc, Tcb):
    """detlaT = Frame::GetTransformFromOdometer(gt1, gt2).inv() (Tracking.cc:1316, Frame.cc:1049-1067) as float32 3x4.
    The 4x4 float inverse is the host state machine's job (cv::Mat::inv, unpinned); the harness inverts in float64."""
    T12c = synth.odom_transform(p1, p2, Tbc, Tcb).astype(np.float64)
    return np.ascontiguousarray(np.linalg.inv(T12c)[:3, :4].astype(np.float32).reshape(12))


class Sequence:
    """B vehicles on one ground texture.  poses[k][b] = (x, y, theta) of the base frame at frame k."""

    def __init__(self, batch, nframes, seed=9000, front_wh=(1280, 720), bird_wh=(512, 512), fx=500.0, fy=500.0,
                 D=FISHEYE_D, device="cpu", step=0.15, ground=None):
        self.B, self.K = batch, nframes
        self.fw, self.fh = front_wh
        self.bw, self.bh = bird_wh
        self.Kc = (float(fx), float(fy), self.fw / 2.0, self.fh / 2.0)
        self.D = tuple(float(d) for d in D)
        self.dev = torch.device(device)
        self.Tbc, self.Tcb = synth.extrinsics()
        tex, contour = ground if ground is not None else make_ground(seed)
        self.tex = torch.from_numpy(tex).to(self.dev)
        self.contour = torch.from_numpy(contour).to(self.dev)
        self.ny, self.nx = tex.shape
        g = synth.rng(seed + 1)
        self.poses = np.zeros((nframes, batch, 3))
        x0 = g.uniform(0.0, 14.0, batch)
        y0 = g.uniform(-6.0, 6.0, batch)
        th0 = g.uniform(-0.35, 0.35, batch)
        dth = g.uniform(-0.012, 0.012, batch)
        sp = step * g.uniform(0.8, 1.2, batch)
        for b in range(batch):
            x, y, th = x0[b], y0[b], th0[b]
            for k in range(nframes):
                self.poses[k, b] = (x, y, th)
                x, y, th = x + sp[b] * math.cos(th), y + sp[b] * math.sin(th), th + dth[b]
        # odometer readings = true poses + small noise (the prediction is then off by a few mm / mrad)
        self.odom = self.poses + np.stack([g.normal(0, 0.004, (nframes, batch)), g.normal(0, 0.004, (nframes, batch)),
                                           g.normal(0, 0.0008, (nframes, batch))], -1)
        self._rays = self._ray_table()
        m = np.full((self.bh, self.bw), 255, np.uint8)   # detect mask: the vehicle body is blanked
        m[self.bh // 2 - 70:self.bh // 2 + 70, self.bw // 2 - 28:self.bw // 2 + 28] = 0
        self.mask = np.ascontiguousarray(np.broadcast_to(m, (batch, self.bh, self.bw)))

    # ---- geometry -----------------------------------------------------------------------------------------------------
    def Twb(self, k, b):
        x, y, th = self.poses[k, b]
        T = np.eye(4)
        T[:3, :3] = synth.rot_xyz(0, 0, th)
        T[:3, 3] = (x, y, 0.0)
        return T

    def Tcw_true(self, k, b):
        return self.Tcb.astype(np.float64) @ np.linalg.inv(self.Twb(k, b))

    def delta(self, k):
        """[B][12] float32: detlaT between frame k-1 and k from the odometer readings."""
        return self.delta_between(k - 1, k)

    def delta_between(self, k_from, k_to):
        """detlaT that takes frame k_from's pose to frame k_to's (any two frames: the bench drives back and forth)."""
        return np.stack([odom_delta(self.odom[k_from, b], self.odom[k_to, b], self.Tbc, self.Tcb) for b in range(self.B)])

    def _ray_table(self):
        fx, fy, cx, cy = self.Kc
        v, u = torch.meshgrid(torch.arange(self.fh, dtype=torch.float64, device=self.dev),
                              torch.arange(self.fw, dtype=torch.float64, device=self.dev), indexing="ij")
        xd, yd = (u - cx) / fx, (v - cy) / fy
        thd = torch.sqrt(xd * xd + yd * yd)
        k1, k2, k3, k4 = self.D
        th = thd.clone()
        for _ in range(25):  # Newton on theta_d = theta (1 + k1 t^2 + k2 t^4 + k3 t^6 + k4 t^8)
            t2 = th * th
            f = th * (1 + t2 * (k1 + t2 * (k2 + t2 * (k3 + t2 * k4)))) - thd
            df = 1 + t2 * (3 * k1 + t2 * (5 * k2 + t2 * (7 * k3 + t2 * 9 * k4)))
            th = th - f / df
        scale = torch.where(thd > 1e-12, torch.tan(th) / thd.clamp_min(1e-12), torch.ones_like(thd))
        return torch.stack([xd * scale, yd * scale, torch.ones_like(xd)], -1)  # [H, W, 3] camera rays

    def _sample(self, canvas, wx, wy, fill, texel=TEXEL):
        ny, nx = canvas.shape
        ix = torch.floor((wx - GROUND_X[0]) / texel).long()
        iy = torch.floor((wy - GROUND_Y[0]) / texel).long()
        ok = (ix >= 0) & (ix < nx) & (iy >= 0) & (iy < ny)
        val = canvas[iy.clamp(0, ny - 1), ix.clamp(0, nx - 1)]
        return torch.where(ok, val, torch.full_like(val, fill))

    def render(self, k):
        """(front [B,fh,fw], bird [B,bh,bw], contour [B,bh,bw]) u8 tensors on self.dev for frame k."""
        fr, bd, ct = [], [], []
        py, px = torch.meshgrid(torch.arange(self.bh, dtype=torch.float64, device=self.dev),
                                torch.arange(self.bw, dtype=torch.float64, device=self.dev), indexing="ij")
        bx = (self.bh // 2 - py) * synth.PIXEL2METER + synth.REAR_AXLE_TO_CENTER
        by = (self.bw // 2 - px) * synth.PIXEL2METER
        for b in range(self.B):
            Twc = torch.from_numpy(np.linalg.inv(self.Tcw_true(k, b))).to(self.dev)
            d = self._rays @ Twc[:3, :3].T
            O = Twc[:3, 3]
            t = -O[2] / d[..., 2].clamp(max=-1e-9)
            hit = (d[..., 2] < -1e-9) & (t < 80.0)
            img = self._sample(self.tex, O[0] + t * d[..., 0], O[1] + t * d[..., 1], 128)
            fr.append(torch.where(hit, img, torch.full_like(img, 128)))
            x, y, th = self.poses[k, b]
            wx = x + math.cos(th) * bx - math.sin(th) * by
            wy = y + math.sin(th) * bx + math.cos(th) * by
            bd.append(self._sample(self.tex, wx, wy, 128))
            ct.append(self._sample(self.contour, wx, wy, 0, 4 * TEXEL))
        return torch.stack(fr).contiguous(), torch.stack(bd).contiguous(), torch.stack(ct).contiguous()

    # ---- the map the tracker starts from (built from frame 0's extraction) -----------------------------------------------
    def build_map(self, view0, tables, seed=9100, map_cap=None, bird_cap=None, hold_frac=0.7, bird_hold_frac=0.6, extra_views=()):
        """From frame 0's key points (a downloaded frame view: numpy arrays) make, per sequence,
        the MapPoint table (every front key point's ground point), the MapPointBird table (every bird key point) and
        frame 0's initial mvpMapPoints / mvpMapPointsBird (a subset: the rest is for SearchLocalPoints / M9 to find).
        extra_views = [(k, view of frame k), ...]: further "key frames" whose key points are added to the tables
        (the map a LocalMapping thread would have built along the drive)."""
        B = self.B
        cap = view0["kps"].shape[1]
        map_cap = map_cap or cap
        bird_cap = bird_cap or 2 * cap
        sf = np.array(list(tables.scale_factor)[:8], np.float64)
        fx, fy, cx, cy = self.Kc
        M = dict(n=np.zeros(B, np.int32), bad=np.zeros((B, map_cap), np.uint8), obs_pos=np.zeros((B, map_cap), np.uint8),
                 xw=np.zeros((B, map_cap, 3), np.float32), normal=np.zeros((B, map_cap, 3), np.float32),
                 max_dist=np.zeros((B, map_cap), np.float32), min_dist=np.zeros((B, map_cap), np.float32),
                 desc=np.zeros((B, map_cap, 32), np.uint8))
        MB = dict(n=np.zeros(B, np.int32), xw=np.zeros((B, bird_cap, 3), np.float32), desc=np.zeros((B, bird_cap, 32), np.uint8))
        mp0 = np.full((B, cap), -1, np.int32)
        mpb0 = np.full((B, cap), -1, np.int32)
        Tcw0 = np.zeros((B, 12), np.float32)
        views = [(0, view0)] + list(extra_views)
        bird_budget = bird_cap // 2 // len(views)   # the other half of the bird table is for points the chain creates
        for b in range(B):
            g = synth.rng(seed + b)
            Tcw0[b] = synth.to12(self.Tcw_true(0, b))
            for vi, (kf, view) in enumerate(views):
                Twc = np.linalg.inv(self.Tcw_true(kf, b))
                n = int(view["n"][b])
                ku = view["kps_un"][b, :n]
                ray = np.stack([(ku["x"].astype(np.float64) - cx) / fx, (ku["y"].astype(np.float64) - cy) / fy, np.ones(n)], 1)
                d = ray @ Twc[:3, :3].T
                O = Twc[:3, 3]
                good = d[:, 2] < -1e-6
                t = np.where(good, -O[2] / np.where(good, d[:, 2], -1.0), 10.0)
                good &= t * np.linalg.norm(ray, axis=1) < 60.0
                X = O + t[:, None] * d + g.normal(0, 0.001, (n, 3))
                m0 = int(M["n"][b])
                ids = np.nonzero(good)[0][: max(0, map_cap - m0)]
                m = len(ids)
                sl = slice(m0, m0 + m)
                M["n"][b] = m0 + m
                M["xw"][b, sl] = X[ids]
                dist = np.linalg.norm(X[ids] - O, axis=1)
                M["normal"][b, sl] = ((X[ids] - O) / dist[:, None]).astype(np.float32)
                lvl = ku["octave"][ids]
                M["max_dist"][b, sl] = dist * sf[lvl]
                M["min_dist"][b, sl] = dist * sf[lvl] / sf[7]
                M["desc"][b, sl] = synth.flip_bits(g, view["desc"][b, ids], p=0.03)
                M["obs_pos"][b, sl] = (g.random(m) >= 0.03).astype(np.uint8)
                M["bad"][b, sl] = (g.random(m) < 0.01).astype(np.uint8)
                # bird: camera XYZ of the key point -> world
                nb = int(view["n_bird"][b])
                pc = view["bird_cam_xyz"][b, :nb].astype(np.float64)
                Xb = pc @ Twc[:3, :3].T + O + g.normal(0, 0.002, (nb, 3))
                b0 = int(MB["n"][b])
                mb = min(nb, bird_budget)
                MB["n"][b] = b0 + mb
                MB["xw"][b, b0:b0 + mb] = Xb[:mb]
                MB["desc"][b, b0:b0 + mb] = synth.flip_bits(g, view["desc_bird"][b, :mb], p=0.03)
                if vi == 0:
                    hold = g.random(m) < hold_frac
                    mp0[b, ids[hold]] = np.nonzero(hold)[0].astype(np.int32)
                    holdb = g.random(mb) < bird_hold_frac
                    mpb0[b, :mb][holdb] = np.nonzero(holdb)[0].astype(np.int32)
        return M, MB, mp0, mpb0, Tcw0
