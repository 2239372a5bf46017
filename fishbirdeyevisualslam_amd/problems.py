"""Builders that turn the numpy problem dicts of synth.py into the C-ABI argument
structs of include/fishbird.h (host pointers).  Each builder returns (args, outputs, keep)
where `outputs` is a dict of numpy arrays the call fills and `keep` holds references that
must outlive the call.  Used by the Python host wrappers and by the tests (which pass the
very same struct to the HIP library and to the oracle).
"""
import numpy as np

from . import cabi, synth
from .cabi import fill


def _c(a, dtype=None):
    a = np.ascontiguousarray(a if dtype is None else np.asarray(a, dtype))
    return a


def _stack(list_of_arrays, stride, dtype, tail=()):
    """[batch][stride][*tail] zero-padded stack."""
    out = np.zeros((len(list_of_arrays), stride) + tuple(tail), dtype)
    for b, a in enumerate(list_of_arrays):
        out[b, : len(a)] = a
    return out


def grid_geom(d):
    g = cabi.GridGeom()
    return fill(g, **d)


def build_grid_host(kps_list, geom, grid_fn, stride=None):
    """Run a grid builder (oracle or HIP-host wrapper) over a list of keypoint arrays."""
    batch = len(kps_list)
    stride = stride or max(len(k) for k in kps_list)
    kps = _stack(kps_list, stride, cabi.KP_DTYPE)
    n = np.array([len(k) for k in kps_list], np.int32)
    ncell = geom.cols * geom.rows
    cs = np.zeros((batch, ncell + 1), np.int32)
    ci = np.zeros((batch, stride), np.int32)
    grid_fn(kps, n, batch, stride, geom, cs, ci)
    return cs, ci


def proj_frame_args(problems, cell_start, cell_items, th=15.0, nnratio=0.9, check_ori=1, cur_stride=None, scale_factors=None):
    """fb_proj_frame_args from a list of synth.make_proj_frame_problem dicts."""
    B = len(problems)
    p0 = problems[0]
    cs_ = cur_stride or max(len(p["cur_kps"]) for p in problems)
    ls_ = max(len(p["last_xw"]) for p in problems)
    assert cell_items.shape == (B, cs_)
    keep = dict(
        n_cur=np.array([len(p["cur_kps"]) for p in problems], np.int32),
        cur_kps=_stack([p["cur_kps"] for p in problems], cs_, cabi.KP_DTYPE),
        cur_desc=_stack([p["cur_desc"] for p in problems], cs_, np.uint8, (32,)),
        cur_cell_start=_c(cell_start), cur_cell_items=_c(cell_items),
        cur_blocked=_stack([p["cur_blocked"] for p in problems], cs_, np.uint8),
        cur_Tcw=_c(np.stack([p["Tcw"] for p in problems]), np.float32),
        n_last=np.array([len(p["last_xw"]) for p in problems], np.int32),
        last_valid=_stack([p["last_valid"] for p in problems], ls_, np.uint8),
        last_obs_pos=_stack([p["last_obs_pos"] for p in problems], ls_, np.uint8),
        last_xw=_stack([p["last_xw"] for p in problems], ls_, np.float32, (3,)),
        last_desc=_stack([p["last_desc"] for p in problems], ls_, np.uint8, (32,)),
        last_octave=_stack([p["last_octave"] for p in problems], ls_, np.int32),
        last_angle=_stack([p["last_angle"] for p in problems], ls_, np.float32),
    )
    out = dict(match_cur_to_last=np.full((B, cs_), -7, np.int32), nmatches=np.full(B, -7, np.int32))
    a = cabi.ProjFrameArgs()
    fill(a, batch=B, cur_stride=cs_, last_stride=ls_, th=th, **keep, **out)
    fill(a.cam, fx=p0["fx"], fy=p0["fy"], cx=p0["cx"], cy=p0["cy"], min_x=0.0, min_y=0.0, max_x=float(p0["w"]),
         max_y=float(p0["h"]))
    fill(a.grid, **synth.front_grid_geom(p0["w"], p0["h"]))
    sf = synth.scale_tables()[0] if scale_factors is None else scale_factors
    fill(a, scale_factors=[float(x) for x in sf])
    fill(a.matcher, nnratio=nnratio, check_orientation=check_ori)
    return a, out, keep


def bird_mp_args(problems, cell_start, cell_items, window=10, filter_size=0.05, nnratio=0.9, prefill=-1, cur_stride=None):
    B = len(problems)
    p0 = problems[0]
    cs_ = cur_stride or max(len(p["cur_kps"]) for p in problems)
    rs_ = max(len(p["ref_xw"]) for p in problems)
    keep = dict(
        n_cur=np.array([len(p["cur_kps"]) for p in problems], np.int32),
        cur_kps=_stack([p["cur_kps"] for p in problems], cs_, cabi.KP_DTYPE),
        cur_desc=_stack([p["cur_desc"] for p in problems], cs_, np.uint8, (32,)),
        cur_cam_xyz=_stack([p["cur_cam_xyz"] for p in problems], cs_, np.float32, (3,)),
        cur_cell_start=_c(cell_start), cur_cell_items=_c(cell_items),
        cur_Tcw=_c(np.stack([p["Tcw"] for p in problems]), np.float32),
        n_ref=np.array([len(p["ref_xw"]) for p in problems], np.int32),
        ref_valid=_stack([p["ref_valid"] for p in problems], rs_, np.uint8),
        ref_xw=_stack([p["ref_xw"] for p in problems], rs_, np.float32, (3,)),
        ref_desc=_stack([p["ref_desc"] for p in problems], rs_, np.uint8, (32,)),
    )
    out = dict(match_cur_to_ref=np.full((B, cs_), prefill, np.int32), ninliers=np.full(B, -7, np.int32))
    a = cabi.BirdMpArgs()
    fill(a, batch=B, cur_stride=cs_, ref_stride=rs_, bird_cols=p0["cols"], bird_rows=p0["rows"],
         meter2pixel=synth.METER2PIXEL, rear_axle_to_center=synth.REAR_AXLE_TO_CENTER, window_size=window,
         filter_size=filter_size, Tbc=[float(x) for x in p0["Tbc"][:3, :4].reshape(12)], **keep, **out)
    fill(a.grid, **synth.bird_grid_geom(p0["cols"], p0["rows"]))
    fill(a.matcher, nnratio=nnratio, check_orientation=1)
    return a, out, keep


def pose_args(problems, mode=cabi.FB_POSE_FRONT_BIRD, wF=1.0, wB=1.0, front_valid=None, bird_valid=None,
              bird_outlier_in=None):
    B = len(problems)
    p0 = problems[0]
    fs = max(len(p["front_xw"]) for p in problems)
    bs = max(len(p["bird_xw"]) for p in problems)
    keep = dict(
        n_front=np.array([len(p["front_xw"]) for p in problems], np.int32),
        front_xw=_stack([p["front_xw"] for p in problems], fs, np.float32, (3,)),
        front_obs=_stack([p["front_obs"] for p in problems], fs, np.float32, (2,)),
        front_inv_sigma2=_stack([p["front_inv_sigma2"] for p in problems], fs, np.float32),
        n_bird=np.array([len(p["bird_xw"]) for p in problems], np.int32),
        bird_xw=_stack([p["bird_xw"] for p in problems], bs, np.float32, (3,)),
        bird_xc=_stack([p["bird_xc"] for p in problems], bs, np.float32, (3,)),
        bird_inv_sigma2=_stack([p["bird_inv_sigma2"] for p in problems], bs, np.float32),
    )
    if front_valid is not None:
        keep["front_valid"] = _stack(front_valid, fs, np.uint8)
    if bird_valid is not None:
        keep["bird_valid"] = _stack(bird_valid, bs, np.uint8)
    out = dict(
        Tcw=_c(np.stack([p["Tcw0"] for p in problems]), np.float32).copy(),
        front_outlier=np.full((B, fs), 9, np.uint8),
        bird_outlier=(np.zeros((B, bs), np.uint8) if bird_outlier_in is None else _stack(bird_outlier_in, bs, np.uint8)),
        ninliers=np.full(B, -7, np.int32),
    )
    a = cabi.PoseOptArgs()
    fill(a, batch=B, mode=mode, front_stride=fs, bird_stride=bs, fx=p0["fx"], fy=p0["fy"], cx=p0["cx"], cy=p0["cy"],
         wF=wF, wB=wB, **keep, **out)
    return a, out, keep


def proj_points_args(problems, cell_start, cell_items, th=1.0, nnratio=0.8):
    B = len(problems)
    p0 = problems[0]
    cs_ = max(len(p["cur_kps"]) for p in problems)
    ms_ = max(len(p["mp_desc"]) for p in problems)
    keep = dict(
        n_cur=np.array([len(p["cur_kps"]) for p in problems], np.int32),
        cur_kps=_stack([p["cur_kps"] for p in problems], cs_, cabi.KP_DTYPE),
        cur_desc=_stack([p["cur_desc"] for p in problems], cs_, np.uint8, (32,)),
        cur_cell_start=_c(cell_start), cur_cell_items=_c(cell_items),
        cur_blocked=_stack([p["cur_blocked"] for p in problems], cs_, np.uint8),
        n_mp=np.array([len(p["mp_desc"]) for p in problems], np.int32),
        mp_track=_stack([p["mp_track"] for p in problems], ms_, np.uint8),
        mp_obs_pos=_stack([p["mp_obs_pos"] for p in problems], ms_, np.uint8),
        mp_proj=_stack([p["mp_proj"] for p in problems], ms_, np.float32, (2,)),
        mp_level=_stack([p["mp_level"] for p in problems], ms_, np.int32),
        mp_view_cos=_stack([p["mp_view_cos"] for p in problems], ms_, np.float32),
        mp_desc=_stack([p["mp_desc"] for p in problems], ms_, np.uint8, (32,)),
    )
    out = dict(match_cur_to_mp=np.full((B, cs_), -7, np.int32), nmatches=np.full(B, -7, np.int32))
    a = cabi.ProjPointsArgs()
    fill(a, batch=B, cur_stride=cs_, mp_stride=ms_, th=th, **keep, **out)
    fill(a.grid, **synth.front_grid_geom(p0["w"], p0["h"]))
    fill(a, scale_factors=[float(x) for x in synth.scale_tables()[0]])
    fill(a.matcher, nnratio=nnratio, check_orientation=1)
    return a, out, keep


def birdview_args(problems, cell_start, cell_items, window=10, nnratio=0.9, check_ori=1):
    B = len(problems)
    p0 = problems[0]
    cs_ = max(len(p["cur_kps"]) for p in problems)
    rs_ = max(len(p["ref_kps"]) for p in problems)
    keep = dict(
        n_cur=np.array([len(p["cur_kps"]) for p in problems], np.int32),
        cur_kps=_stack([p["cur_kps"] for p in problems], cs_, cabi.KP_DTYPE),
        cur_desc=_stack([p["cur_desc"] for p in problems], cs_, np.uint8, (32,)),
        cur_cell_start=_c(cell_start), cur_cell_items=_c(cell_items),
        n_ref=np.array([len(p["ref_kps"]) for p in problems], np.int32),
        ref_kps=_stack([p["ref_kps"] for p in problems], rs_, cabi.KP_DTYPE),
        ref_desc=_stack([p["ref_desc"] for p in problems], rs_, np.uint8, (32,)),
    )
    out = dict(match_ref_to_cur=np.full((B, rs_), -7, np.int32), match_dist=np.full((B, rs_), -7, np.int32),
               nmatches=np.full(B, -7, np.int32), n_dmatches=np.full(B, -7, np.int32))
    a = cabi.BirdviewArgs()
    fill(a, batch=B, cur_stride=cs_, ref_stride=rs_, window_size=window, **keep, **out)
    fill(a.grid, **synth.bird_grid_geom(p0["cols"], p0["rows"]))
    fill(a.matcher, nnratio=nnratio, check_orientation=check_ori)
    return a, out, keep
