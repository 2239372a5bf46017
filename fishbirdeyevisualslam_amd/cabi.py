"""ctypes mirror of include/fishbird.h (the C-ABI drop-in boundary).

Every Structure here matches a struct in include/fishbird.h field for field.  Pointer
fields are c_void_p so the same struct can carry host pointers (numpy) for the
drop-in entry points or device pointers (torch tensors on the GPU) for the *_dev ones.
"""
import ctypes as C

import numpy as np

FB_MAX_LEVELS = 16
FB_OK, FB_ERR_ARG, FB_ERR_HIP, FB_ERR_CAPACITY, FB_ERR_NODEVICE = 0, -1, -2, -3, -4
FB_POSE_FRONT, FB_POSE_FRONT_BIRD, FB_POSE_BIRD = 0, 1, 2

KP_DTYPE = np.dtype(
    [("x", "<f4"), ("y", "<f4"), ("size", "<f4"), ("angle", "<f4"), ("response", "<f4"), ("octave", "<i4")]
)
assert KP_DTYPE.itemsize == 24

_vp = C.c_void_p
_i32 = C.c_int32
_f32 = C.c_float


def ptr(x):
    """Address of a numpy array / torch tensor / int / None as an integer."""
    if x is None:
        return None
    if isinstance(x, int):
        return x
    if isinstance(x, np.ndarray):
        assert x.flags["C_CONTIGUOUS"], "array must be C-contiguous"
        return x.ctypes.data
    if hasattr(x, "data_ptr"):
        assert x.is_contiguous()
        return x.data_ptr()
    raise TypeError(type(x))


class ProfEntry(C.Structure):
    _fields_ = [("name", C.c_char * 32), ("launches", _i32), ("total_ms", C.c_double)]


class OrbParams(C.Structure):
    _fields_ = [("nfeatures", _i32), ("scale_factor", _f32), ("nlevels", _i32), ("ini_th_fast", _i32),
                ("min_th_fast", _i32)]


class OrbTables(C.Structure):
    _fields_ = [("scale_factor", _f32 * FB_MAX_LEVELS), ("inv_scale_factor", _f32 * FB_MAX_LEVELS),
                ("level_sigma2", _f32 * FB_MAX_LEVELS), ("inv_level_sigma2", _f32 * FB_MAX_LEVELS),
                ("features_per_level", _i32 * FB_MAX_LEVELS), ("umax", _i32 * 16)]


class GridGeom(C.Structure):
    _fields_ = [("min_x", _f32), ("min_y", _f32), ("inv_w", _f32), ("inv_h", _f32), ("cols", _i32), ("rows", _i32)]


class BirdGuidanceArgs(C.Structure):
    _fields_ = [("batch", _i32), ("kp_stride", _i32), ("cols", _i32), ("rows", _i32), ("pitch", _i32),
                ("contour", _vp), ("mask", _vp), ("n_in", _vp), ("kps_in", _vp), ("desc_in", _vp),
                ("n_out", _vp), ("kps_out", _vp), ("desc_out", _vp), ("keep", _vp),
                ("edge_cap", _i32), ("n_edge_sign", _vp), ("n_edge_free", _vp), ("edge_sign", _vp), ("edge_free", _vp)]


class MatcherParams(C.Structure):
    _fields_ = [("nnratio", _f32), ("check_orientation", _i32)]


class Camera(C.Structure):
    _fields_ = [("fx", _f32), ("fy", _f32), ("cx", _f32), ("cy", _f32), ("min_x", _f32), ("min_y", _f32),
                ("max_x", _f32), ("max_y", _f32)]


class ProjFrameArgs(C.Structure):
    _fields_ = [("batch", _i32), ("cur_stride", _i32), ("last_stride", _i32),
                ("n_cur", _vp), ("cur_kps", _vp), ("cur_desc", _vp), ("cur_cell_start", _vp),
                ("cur_cell_items", _vp), ("cur_blocked", _vp), ("cur_Tcw", _vp),
                ("n_last", _vp), ("last_valid", _vp), ("last_obs_pos", _vp), ("last_xw", _vp),
                ("last_desc", _vp), ("last_octave", _vp), ("last_angle", _vp),
                ("cam", Camera), ("grid", GridGeom), ("scale_factors", _f32 * FB_MAX_LEVELS), ("th", _f32),
                ("matcher", MatcherParams), ("match_cur_to_last", _vp), ("nmatches", _vp), ("retry_below", _i32), ("retry_th", _f32), ("retried", _vp)]


class BirdMpArgs(C.Structure):
    _fields_ = [("batch", _i32), ("cur_stride", _i32), ("ref_stride", _i32),
                ("n_cur", _vp), ("cur_kps", _vp), ("cur_desc", _vp), ("cur_cam_xyz", _vp),
                ("cur_cell_start", _vp), ("cur_cell_items", _vp), ("cur_Tcw", _vp),
                ("n_ref", _vp), ("ref_valid", _vp), ("ref_xw", _vp), ("ref_desc", _vp),
                ("Tbc", _f32 * 12), ("bird_cols", _i32), ("bird_rows", _i32), ("meter2pixel", C.c_double),
                ("rear_axle_to_center", C.c_double), ("grid", GridGeom), ("window_size", _i32),
                ("filter_size", _f32), ("matcher", MatcherParams), ("match_cur_to_ref", _vp), ("ninliers", _vp)]


class ProjPointsArgs(C.Structure):
    _fields_ = [("batch", _i32), ("cur_stride", _i32), ("mp_stride", _i32),
                ("n_cur", _vp), ("cur_kps", _vp), ("cur_desc", _vp), ("cur_cell_start", _vp),
                ("cur_cell_items", _vp), ("cur_blocked", _vp),
                ("n_mp", _vp), ("mp_track", _vp), ("mp_obs_pos", _vp), ("mp_proj", _vp), ("mp_level", _vp),
                ("mp_view_cos", _vp), ("mp_desc", _vp),
                ("grid", GridGeom), ("scale_factors", _f32 * FB_MAX_LEVELS), ("th", _f32),
                ("matcher", MatcherParams), ("match_cur_to_mp", _vp), ("nmatches", _vp), ("workspace", _vp), ("workspace_bytes", C.c_size_t)]


class BirdviewArgs(C.Structure):
    _fields_ = [("batch", _i32), ("cur_stride", _i32), ("ref_stride", _i32),
                ("n_cur", _vp), ("cur_kps", _vp), ("cur_desc", _vp), ("cur_cell_start", _vp),
                ("cur_cell_items", _vp), ("n_ref", _vp), ("ref_kps", _vp), ("ref_desc", _vp),
                ("grid", GridGeom), ("window_size", _i32), ("matcher", MatcherParams),
                ("match_ref_to_cur", _vp), ("match_dist", _vp), ("nmatches", _vp), ("n_dmatches", _vp)]


class FeatureVector(C.Structure):
    _fields_ = [("node_stride", _i32), ("item_stride", _i32), ("n_nodes", _vp), ("node_ids", _vp), ("node_start", _vp),
                ("items", _vp)]


class BowArgs(C.Structure):
    _fields_ = [("batch", _i32), ("kf_stride", _i32), ("f_stride", _i32),
                ("n_kf", _vp), ("kf_kps", _vp), ("kf_desc", _vp), ("kf_has_mp", _vp), ("kf_fv", FeatureVector),
                ("n_f", _vp), ("f_kps", _vp), ("f_desc", _vp), ("f_fv", FeatureVector),
                ("matcher", MatcherParams), ("match_f_to_kf", _vp), ("nmatches", _vp)]


class TriangulationArgs(C.Structure):
    _fields_ = [("batch", _i32), ("kf1_stride", _i32), ("kf2_stride", _i32),
                ("n1", _vp), ("kps1", _vp), ("desc1", _vp), ("has_mp1", _vp), ("fv1", FeatureVector),
                ("n2", _vp), ("kps2", _vp), ("desc2", _vp), ("has_mp2", _vp), ("fv2", FeatureVector),
                ("F12", _vp), ("Cw1", _vp), ("R2w", _vp), ("t2w", _vp),
                ("fx", _f32), ("fy", _f32), ("cx", _f32), ("cy", _f32),
                ("scale_factors", _f32 * FB_MAX_LEVELS), ("level_sigma2", _f32 * FB_MAX_LEVELS),
                ("matcher", MatcherParams), ("matches12", _vp), ("nmatches", _vp)]


class ProjKfArgs(C.Structure):
    _fields_ = [("batch", _i32), ("cur_stride", _i32), ("kf_stride", _i32),
                ("n_cur", _vp), ("cur_kps", _vp), ("cur_desc", _vp), ("cur_cell_start", _vp),
                ("cur_cell_items", _vp), ("cur_blocked", _vp), ("cur_Tcw", _vp),
                ("n_kf", _vp), ("kf_valid", _vp), ("kf_xw", _vp), ("kf_desc", _vp),
                ("kf_max_dist", _vp), ("kf_min_dist", _vp), ("kf_angle", _vp),
                ("cam", Camera), ("grid", GridGeom), ("scale_factors", _f32 * FB_MAX_LEVELS),
                ("log_scale_factor", _f32), ("n_levels", _i32), ("th", _f32), ("orb_dist", _i32),
                ("matcher", MatcherParams), ("match_cur_to_kf", _vp), ("nmatches", _vp)]


class BowKfArgs(C.Structure):
    _fields_ = [("batch", _i32), ("kf1_stride", _i32), ("kf2_stride", _i32),
                ("n1", _vp), ("kps1", _vp), ("desc1", _vp), ("has_mp1", _vp), ("fv1", FeatureVector),
                ("n2", _vp), ("kps2", _vp), ("desc2", _vp), ("has_mp2", _vp), ("fv2", FeatureVector),
                ("matcher", MatcherParams), ("matches12", _vp), ("nmatches", _vp)]


class KfTarget(C.Structure):
    _fields_ = [("kf_stride", _i32), ("n_kf", _vp), ("kf_kps", _vp), ("kf_desc", _vp), ("kf_cell_start", _vp),
                ("kf_cell_items", _vp), ("cam", Camera), ("grid", GridGeom),
                ("scale_factors", _f32 * FB_MAX_LEVELS), ("inv_level_sigma2", _f32 * FB_MAX_LEVELS),
                ("log_scale_factor", _f32), ("n_levels", _i32)]


class MpList(C.Structure):
    _fields_ = [("mp_stride", _i32), ("n_mp", _vp), ("mp_valid", _vp), ("mp_xw", _vp), ("mp_normal", _vp),
                ("mp_max_dist", _vp), ("mp_min_dist", _vp), ("mp_desc", _vp)]


class FuseArgs(C.Structure):
    _fields_ = [("batch", _i32), ("kf", KfTarget), ("mp", MpList), ("pose", _vp), ("Ow", _vp), ("th", _f32),
                ("best_idx", _vp)]


class ProjSim3Args(C.Structure):
    _fields_ = [("batch", _i32), ("kf", KfTarget), ("mp", MpList), ("Scw", _vp), ("kf_matched", _vp), ("th", _i32),
                ("match_kf_to_mp", _vp), ("nmatches", _vp)]


class Sim3Args(C.Structure):
    _fields_ = [("batch", _i32), ("kf1", KfTarget), ("kf2", KfTarget), ("mp1", MpList), ("mp2", MpList),
                ("T1w", _vp), ("T2w", _vp), ("s12", _vp), ("R12", _vp), ("t12", _vp), ("th", _f32),
                ("matches12", _vp), ("nfound", _vp)]


class InitMatchArgs(C.Structure):
    _fields_ = [("batch", _i32), ("f1_stride", _i32), ("f2_stride", _i32), ("n1", _vp), ("kps1", _vp), ("desc1", _vp),
                ("n2", _vp), ("kps2", _vp), ("desc2", _vp), ("f2_cell_start", _vp), ("f2_cell_items", _vp),
                ("grid", GridGeom), ("window_size", _i32), ("matcher", MatcherParams),
                ("prev_matched", _vp), ("matches12", _vp), ("nmatches", _vp)]


class Vocabulary(C.Structure):
    _fields_ = [("n_nodes", _i32), ("L", _i32), ("child_start", _vp), ("children", _vp), ("descriptors", _vp),
                ("weights", _vp), ("word_ids", _vp)]


class BowTransformArgs(C.Structure):
    _fields_ = [("batch", _i32), ("f_stride", _i32), ("n_f", _vp), ("desc", _vp), ("levelsup", _i32),
                ("n_words", _vp), ("bow_ids", _vp), ("bow_vals", _vp), ("fv_n_nodes", _vp), ("fv_node_ids", _vp),
                ("fv_node_start", _vp), ("fv_items", _vp)]


class BirdFilterArgs(C.Structure):
    _fields_ = [("batch", _i32), ("match_stride", _i32), ("kp1_stride", _i32), ("kp2_stride", _i32),
                ("n_matches", _vp), ("query_idx", _vp), ("train_idx", _vp), ("cam_xyz1", _vp), ("cam_xyz2", _vp),
                ("Tcw1", _vp), ("Tcw2", _vp), ("occupied2", _vp), ("window_size", _f32), ("keep", _vp),
                ("pt_world", _vp)]


class FrustumArgs(C.Structure):
    _fields_ = [("batch", _i32), ("mp_stride", _i32), ("Tcw", _vp), ("Ow", _vp), ("n_mp", _vp), ("mp_valid", _vp),
                ("mp_xw", _vp), ("mp_normal", _vp), ("mp_max_dist", _vp), ("mp_min_dist", _vp),
                ("cam", Camera), ("mbf", _f32), ("viewing_cos_limit", _f32), ("log_scale_factor", _f32),
                ("n_levels", _i32),
                ("in_view", _vp), ("proj", _vp), ("proj_xr", _vp), ("level", _vp), ("view_cos", _vp)]


class PoseOptArgs(C.Structure):
    _fields_ = [("batch", _i32), ("mode", _i32), ("front_stride", _i32), ("bird_stride", _i32),
                ("fx", _f32), ("fy", _f32), ("cx", _f32), ("cy", _f32), ("wF", _f32), ("wB", _f32),
                ("n_front", _vp), ("front_xw", _vp), ("front_obs", _vp), ("front_inv_sigma2", _vp),
                ("front_valid", _vp),
                ("n_bird", _vp), ("bird_xw", _vp), ("bird_xc", _vp), ("bird_inv_sigma2", _vp),
                ("bird_valid", _vp), ("bird_outlier", _vp),
                ("Tcw", _vp), ("front_outlier", _vp), ("ninliers", _vp)]


class LocalBAArgs(C.Structure):
    _fields_ = [("with_odom", _i32), ("fx", _f32), ("fy", _f32), ("cx", _f32), ("cy", _f32),
                ("wF", _f32), ("wB", _f32), ("wP", _f32),
                ("n_kf", _i32), ("kf_Tcw", _vp), ("kf_fixed", _vp),
                ("n_mp", _i32), ("mp_xw", _vp), ("n_mpb", _i32), ("mpb_xw", _vp),
                ("n_obs", _i32), ("obs_kf", _vp), ("obs_mp", _vp), ("obs_uv", _vp), ("obs_inv_sigma2", _vp),
                ("n_bobs", _i32), ("bobs_kf", _vp), ("bobs_mpb", _vp), ("bobs_xc", _vp), ("bobs_inv_sigma2", _vp),
                ("n_odom", _i32), ("odom_kf_i", _vp), ("odom_kf_j", _vp), ("odom_Tij", _vp), ("odom_info", _vp),
                ("stop_flag", _vp), ("obs_outlier", _vp), ("bobs_outlier", _vp)]


class FrameParams(C.Structure):
    _fields_ = [("batch", _i32), ("front_width", _i32), ("front_height", _i32), ("bird_width", _i32), ("bird_height", _i32),
                ("orb", OrbParams), ("bird_nfeatures", _i32), ("K", _f32 * 4), ("D", _f32 * 4), ("Tbc", _f32 * 12), ("Tcb", _f32 * 12),
                ("pixel2meter", C.c_double), ("meter2pixel", C.c_double), ("rear_axle_to_center", C.c_double),
                ("map_cap", _i32), ("local_mp_cap", _i32), ("local_mpb_cap", _i32)]


class MapPoints(C.Structure):
    _fields_ = [("stride", _i32), ("n", _vp), ("bad", _vp), ("obs_pos", _vp), ("xw", _vp), ("normal", _vp),
                ("max_dist", _vp), ("min_dist", _vp), ("desc", _vp)]


class MapPointsBird(C.Structure):
    _fields_ = [("stride", _i32), ("n", _vp), ("xw", _vp), ("desc", _vp)]


class TrackArgs(C.Structure):
    _fields_ = [("map", MapPoints), ("mpb", MapPointsBird), ("d_delta", _vp), ("d_local_mp", _vp), ("d_n_local_mp", _vp),
                ("d_local_mpb", _vp), ("d_n_local_mpb", _vp), ("wB", _f32), ("wF", _f32), ("gate_local_map", _i32), ("defer_outlier_drop", _i32), ("min_inliers", _i32)]


class FrameView(C.Structure):
    _fields_ = [("batch", _i32), ("kp_stride", _i32), ("n", _vp), ("kps", _vp), ("kps_un", _vp), ("desc", _vp),
                ("map_point", _vp), ("outlier", _vp), ("n_bird", _vp), ("kps_bird", _vp), ("desc_bird", _vp),
                ("bird_cam_xyz", _vp), ("map_point_bird", _vp), ("bird_outlier", _vp), ("Tcw", _vp), ("counts", _vp)]


FB_CNT = dict(BIRD_KF_MATCHES=0, PROJ_MATCHES=1, POSE1_INLIERS=2, MATCHES=3, MATCHES_MAP=4, BIRDVIEW_MATCHES=5, BIRD_INLIERS=6,
              BIRD_NEW=7, TO_MATCH=8, LOCAL_MATCHES=9, POSE2_INLIERS=10, MATCHES_INLIERS=11, BOW_MATCHES=12, BIRD_POINTS=13, PROJ_RETRIED=14, BIRD_POINTS_FINAL=15)
FB_CNT_COUNT = 16

ALLREDUCE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.POINTER(C.c_double), _i32, _i32)


def fill(struct, **kw):
    """Set fields of a Structure; array-likes become pointers, lists fill fixed arrays."""
    for k, v in kw.items():
        ftype = dict(struct._fields_)[k]
        if ftype is _vp:
            setattr(struct, k, ptr(v))
        elif hasattr(ftype, "_length_") and not isinstance(v, ftype):
            arr = getattr(struct, k)
            for i, x in enumerate(v):
                arr[i] = x
        else:
            setattr(struct, k, v)
    return struct


# every symbol include/fishbird.h declares (checked by tests/test_cabi_exports.py)
EXPORTS = [
    "fb_abi_version", "fb_last_error", "fb_device_count", "fb_set_device", "fb_shutdown",
    "fb_prof_enable", "fb_prof_only", "fb_prof_reset", "fb_prof_report",
    "fb_orb_create", "fb_orb_destroy", "fb_orb_get_tables", "fb_orb_capacity", "fb_orb_set_output_stride", "fb_orb_extract", "fb_orb_extract_batch_dev",
    "fb_orb_get_level", "fb_orb_get_blurred_level", "fb_orb_debug_candidates", "fb_orb_debug_timers", "fb_grid_build_batch_dev", "fb_bird_keys_to_cam_dev", "fb_bird_guidance", "fb_bird_guidance_dev",
    "fb_descriptor_distance_dev", "fb_descriptor_distance",
    "fb_match_projection_frame_dev", "fb_match_projection_frame",
    "fb_match_bird_mappoints_dev", "fb_match_bird_mappoints",
    "fb_match_projection_points_dev", "fb_match_projection_points", "fb_match_projection_points_workspace",
    "fb_match_birdview_dev", "fb_match_birdview",
    "fb_match_bow_dev", "fb_match_bow", "fb_match_triangulation_dev", "fb_match_triangulation",
    "fb_match_projection_keyframe_dev", "fb_match_projection_keyframe", "fb_match_bow_kf_dev", "fb_match_bow_kf",
    "fb_fuse_search_dev", "fb_fuse_search", "fb_fuse_sim3_search_dev", "fb_fuse_sim3_search",
    "fb_match_projection_sim3_dev", "fb_match_projection_sim3", "fb_match_sim3_dev", "fb_match_sim3",
    "fb_match_initialization_dev", "fb_match_initialization",
    "fb_distinctive_descriptors_dev", "fb_distinctive_descriptors",
    "fb_bird_filter_matches_dev", "fb_bird_filter_matches", "fb_bow_transform_dev", "fb_bow_transform",
    "fb_in_frustum_dev", "fb_in_frustum", "fb_undistort_keypoints_dev", "fb_undistort_keypoints", "fb_image_bounds",
    "fb_pose_opt_batch_dev", "fb_pose_opt", "fb_pose_gather_front_dev", "fb_pose_gather_bird_dev",
    "fb_frame_create", "fb_frame_destroy", "fb_frame_extract_dev", "fb_frame_extract", "fb_frame_set_pose_dev",
    "fb_frame_predict_pose_dev", "fb_frame_clear_map_points_dev", "fb_frame_set_map_points_dev",
    "fb_frame_bird_mappoint_match_dev", "fb_frame_search_by_projection_dev", "fb_frame_pose_optimization_dev",
    "fb_frame_discard_outliers_dev", "fb_frame_match_bird_points_dev", "fb_frame_search_local_points_dev",
    "fb_frame_finish_dev", "fb_frame_drop_outliers_dev", "fb_frame_track_dev", "fb_frame_track_motion_model_dev", "fb_frame_track_local_map_dev",
    "fb_frame_copy_dev", "fb_frame_compute_bow_dev", "fb_frame_bow_view_dev", "fb_frame_search_by_bow_dev", "fb_frame_track_reference_dev", "fb_frame_track_using_bird_dev", "fb_frame_view_dev", "fb_frame_download", "fb_frame_counts",
    "fb_local_ba", "fb_local_ba_dev", "fb_local_ba_sharded", "fb_local_ba_sharded_rccl", "fb_rccl_get_unique_id", "fb_rccl_comm_init", "fb_rccl_comm_destroy", "fb_rccl_comm_info", "fb_global_ba",
]
