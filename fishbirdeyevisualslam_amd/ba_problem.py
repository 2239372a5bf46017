"""fb_local_ba_args builder from a synth.make_ba_problem dict (host pointers)."""
import numpy as np

from . import cabi
from .cabi import fill


def local_ba_args(p, with_odom=1, wF=1.0, wB=1.0, stop_flag=None):
    keep = dict(
        kf_Tcw=np.ascontiguousarray(p["kf_Tcw"], np.float32).copy(), kf_fixed=np.ascontiguousarray(p["kf_fixed"], np.uint8),
        mp_xw=np.ascontiguousarray(p["mp_xw"], np.float32).copy(), mpb_xw=np.ascontiguousarray(p["mpb_xw"], np.float32).copy(),
        obs_kf=p["obs_kf"], obs_mp=p["obs_mp"], obs_uv=p["obs_uv"], obs_inv_sigma2=p["obs_inv_sigma2"],
        bobs_kf=p["bobs_kf"], bobs_mpb=p["bobs_mpb"], bobs_xc=p["bobs_xc"], bobs_inv_sigma2=p["bobs_inv_sigma2"],
        odom_kf_i=p["odom_kf_i"], odom_kf_j=p["odom_kf_j"], odom_Tij=p["odom_Tij"], odom_info=p["odom_info"],
        obs_outlier=np.full(len(p["obs_kf"]), 9, np.uint8), bobs_outlier=np.full(max(len(p["bobs_kf"]), 1), 9, np.uint8),
    )
    if stop_flag is not None:
        keep["stop_flag"] = stop_flag
    a = cabi.LocalBAArgs()
    fill(a, with_odom=with_odom, fx=p["fx"], fy=p["fy"], cx=p["cx"], cy=p["cy"], wF=wF, wB=wB, wP=p["wP"],
         n_kf=len(p["kf_fixed"]), n_mp=len(p["mp_xw"]), n_mpb=len(p["mpb_xw"]), n_obs=len(p["obs_kf"]),
         n_bobs=len(p["bobs_kf"]), n_odom=len(p["odom_kf_i"]), **keep)
    out = dict(kf_Tcw=keep["kf_Tcw"], mp_xw=keep["mp_xw"], mpb_xw=keep["mpb_xw"], obs_outlier=keep["obs_outlier"],
               bobs_outlier=keep["bobs_outlier"])
    return a, out, keep


DEV_FIELDS = ("kf_Tcw", "mp_xw", "mpb_xw", "obs_kf", "obs_mp", "obs_uv", "obs_inv_sigma2", "bobs_kf", "bobs_mpb", "bobs_xc",
              "bobs_inv_sigma2", "obs_outlier", "bobs_outlier")


def local_ba_args_dev(p, device="cuda:0", **kw):
    """fb_local_ba_args for fb_local_ba_dev: the big arrays as torch tensors on the device, kf_fixed / odometry / stop flag on the host.
    Returns (args, device tensors, host keep)."""
    import torch
    a, out, keep = local_ba_args(p, **kw)
    dev = {}
    for k in DEV_FIELDS:
        arr = np.ascontiguousarray(keep[k])
        dev[k] = torch.from_numpy(arr.copy()).to(device) if arr.size else torch.zeros(1, dtype=torch.from_numpy(arr.reshape(-1)[:0].copy()).dtype, device=device)
    fill(a, **dev)
    return a, dev, keep
