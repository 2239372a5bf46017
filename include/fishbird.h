/*
 * fishbird.h -- C ABI of the MI355X-native front-end + optimiser path.
 *
 * This header is the drop-in boundary: every entry point names the reference
 * interface it replaces (file:line in JingruiYu/FishBirdEyeVisualSLAM).  Plain
 * pointers and sizes only; no C++/torch/OpenCV types.  INTEGRATION.md shows the
 * shim a maintainer adds inside ORBextractor / ORBmatcher / Optimizer.
 *
 * Pointer convention
 *   *_dev entry points take DEVICE pointers (HBM resident) and a hipStream_t
 *   passed as void*; they enqueue work and return without synchronising.
 *   Entry points without the suffix take HOST pointers, copy, run the same
 *   kernels and synchronise before returning (drop-in for the reference call).
 *
 * Threads
 *   Calls that take no handle (matchers, pose optimisation, bundle adjustment)
 *   may run concurrently from several host threads, each on the device that is
 *   current for its thread.  A handle (fb_orb, fb_rccl communicator) carries
 *   device buffers of its own: one call at a time per handle, like the
 *   reference's ORBextractor object.  fb_last_error() is per thread.
 *
 * Status: 0 = ok, <0 = error (fb_last_error() gives the text).
 * There is no CPU fallback: without a HIP device every compute call fails with
 * FB_ERR_NODEVICE.
 */
#ifndef FISHBIRD_H_
#define FISHBIRD_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define FB_ABI_VERSION 1
#define FB_MAX_LEVELS 16
#define FB_DESC_BYTES 32 /* 256-bit rBRIEF, ORBextractor.cc:1068 */

enum {
  FB_OK = 0,
  FB_ERR_ARG = -1,      /* bad argument */
  FB_ERR_HIP = -2,      /* HIP runtime error */
  FB_ERR_CAPACITY = -3, /* an internal/declared capacity was exceeded */
  FB_ERR_NODEVICE = -4  /* no HIP device / kernels not loadable */
};

/* ---- runtime ------------------------------------------------------------ */
int fb_abi_version(void);
const char *fb_last_error(void);
int fb_device_count(void);
int fb_set_device(int device);
/* Releases what the library keeps between calls: the idle blocks of the device scratch pool (all devices) and the calling
 * thread's bundle-adjustment stream / event / pinned control block.  Handles (fb_orb, fb_frame, communicators) stay valid;
 * the next call simply allocates again.  Call it from every thread that ran a bundle adjustment before the thread ends. */
int fb_shutdown(void);

/* ---- per-kernel timing (HIP events on the launch stream) ------------------- */
/* When enabled every kernel launch of this library is bracketed by two hipEvents
 * recorded on the stream it is launched on; fb_prof_report synchronises and
 * returns launches and total milliseconds per kernel since fb_prof_reset.      */
typedef struct fb_prof_entry {
  char name[32];
  int32_t launches;
  double total_ms;
} fb_prof_entry;
int fb_prof_enable(int on);
int fb_prof_only(const char *kernel_name); /* bracket only this kernel (e.g. "k_fast"); NULL or "" = every kernel */
int fb_prof_reset(void);
int fb_prof_report(fb_prof_entry *out, int cap); /* returns the number of entries written */

/* cv::KeyPoint as POD (ORBextractor.cc:837-847: pt, size, angle, response, octave) */
typedef struct fb_keypoint {
  float x, y;
  float size;
  float angle;
  float response;
  int32_t octave;
} fb_keypoint;

/* ======================================================================== */
/* ORBextractor  (include/ORBextractor.h:51-85, src/ORBextractor.cc:410-1132) */
/* ======================================================================== */
typedef struct fb_orb fb_orb;

typedef struct fb_orb_params { /* ctor args, ORBextractor.h:51-52 */
  int32_t nfeatures;
  float scale_factor;
  int32_t nlevels;
  int32_t ini_th_fast;
  int32_t min_th_fast;
} fb_orb_params;

typedef struct fb_orb_tables { /* getters ORBextractor.h:63-83 + mnFeaturesPerLevel, umax */
  float scale_factor[FB_MAX_LEVELS];
  float inv_scale_factor[FB_MAX_LEVELS];
  float level_sigma2[FB_MAX_LEVELS];
  float inv_level_sigma2[FB_MAX_LEVELS];
  int32_t features_per_level[FB_MAX_LEVELS];
  int32_t umax[16];
} fb_orb_tables;

/* replaces ORBextractor::ORBextractor (ORBextractor.cc:410-470) */
int fb_orb_create(const fb_orb_params *params, fb_orb **out);
void fb_orb_destroy(fb_orb *h);
int fb_orb_get_tables(const fb_orb *h, fb_orb_tables *out);
/* Output capacity per image.  DistributeOctTree stops at >= N nodes, so a level can
 * return up to N+2 keypoints (ORBextractor.cc:669,729): capacity = nfeatures + 8*nlevels. */
int fb_orb_capacity(const fb_orb_params *params);
/* Entries per image in the output arrays of fb_orb_extract_batch_dev when the caller's arrays are wider than the
 * capacity (an fb_frame shares one stride between a 2000-feature front and a 1000-feature bird extractor).  0 = capacity. */
int fb_orb_set_output_stride(fb_orb *h, int kp_stride);

/* replaces ORBextractor::operator() (ORBextractor.cc:1043-1105), host buffers.
 * image: u8, row-major, `stride` bytes per row.  keypoints/descriptors must hold
 * cap = fb_orb_capacity() entries (desc: cap*32 bytes).  *n_out <= cap.
 * cap = nfeatures + 8*nlevels covers the quadtree's overshoot of at most 3 per level; a level can also end with up
 * to 4x its number of root nodes (a strip many times wider than high with a tiny feature budget): if that exceeds cap,
 * FB_ERR_CAPACITY is returned.                                                  */
int fb_orb_extract(fb_orb *h, const uint8_t *image, int width, int height, int stride,
                   fb_keypoint *keypoints, uint8_t *descriptors, int32_t *n_out);

/* same, `batch` equally sized images resident in HBM; image b starts at
 * d_images + b*image_stride.  Outputs: d_keypoints[batch][cap],
 * d_descriptors[batch][cap][32], d_n[batch], cap = fb_orb_capacity(); d_n[b] is
 * clamped to cap (asynchronous call: no error channel for the overflow above).
 * Stream semantics: everything is ordered after the work already in `stream` and is complete for work enqueued to
 * `stream` afterwards.  Calls of 64 images or more run one kernel on a stream the handle owns, forked from and joined
 * back into `stream` with events inside the call (legal under stream capture as well).  */
int fb_orb_extract_batch_dev(fb_orb *h, const uint8_t *d_images, int batch, int width, int height,
                             int stride, size_t image_stride, fb_keypoint *d_keypoints,
                             uint8_t *d_descriptors, int32_t *d_n, void *stream);

/* debug/parity access to the pyramid level of image `b` of the last batch
 * (public member mvImagePyramid, ORBextractor.h:85). dst is host, w*h bytes.  */
int fb_orb_get_level(fb_orb *h, int b, int level, uint8_t *dst, int *w, int *hgt);
/* debug/parity: the same level after GaussianBlur(7x7, 2, 2, BORDER_REFLECT_101) (ORBextractor.cc:1080),
 * the image computeDescriptors samples.  dst is host, w*h bytes of fb_orb_get_level's size. */
int fb_orb_get_blurred_level(fb_orb *h, int b, int level, uint8_t *dst);
/* debug/parity: FAST candidates of (image b, level) of the last batch, the input of
 * DistributeOctTree (vToDistributeKeys, ORBextractor.cc:822-824), packed
 * x | y<<12 | response<<24 in level coordinates, unordered.  Returns the count (>=0). */
int fb_orb_debug_candidates(fb_orb *h, int b, int level, uint32_t *dst, int cap);
/* profiling aid: with FB_FAST_DBG=20 one k_fast workgroup in 16 accumulates shader-clock cycles per phase
 * (0 address set-up + load issue, 1 tile wait, 2 sweep, 3/4/5 score/NMS/slot reservation at iniThFAST, 6/7/8 the same
 * at minThFAST, 9 cell decode, 10 whole wave, 11 number of timed waves); reading resets the counters.          */
int fb_orb_debug_timers(fb_orb *h, uint64_t *dst16);

/* ======================================================================== */
/* Frame grid (src/Frame.cc:381-411, 548-570; include/Frame.h:38-40)         */
/* ======================================================================== */
typedef struct fb_grid_geom {
  float min_x, min_y;       /* mnMinX, mnMinY (0 for the bird grid)            */
  float inv_w, inv_h;       /* mfGridElementWidthInv / HeightInv               */
  int32_t cols, rows;       /* 64x48 front, 32x32 bird                         */
} fb_grid_geom;

/* AssignFeaturesToGrid (Frame.cc:381-411): CSR with cell id = ix*rows+iy, items
 * in ascending keypoint index.  d_cell_start[batch][cols*rows+1],
 * d_cell_items[batch][cap].  Keypoints are read at kp_stride entries per image. */
int fb_grid_build_batch_dev(const fb_keypoint *d_keypoints, const int32_t *d_n, int batch,
                            int kp_stride, const fb_grid_geom *geom, int32_t *d_cell_start,
                            int32_t *d_cell_items, void *stream);

/* Bird keypoint -> base XY -> camera XYZ (Frame.cc:365-373, Converter.cc:284-292,312-318):
 * d_cam_xyz[b][i] = Tcb * BirdPixel2BaseXY(kps[i]).  Tcb = rows 0..2 of Frame::Tcb. */
int fb_bird_keys_to_cam_dev(const fb_keypoint *d_kps, const int32_t *d_n, int batch, int kp_stride,
                            int bird_cols, int bird_rows, double pixel2meter,
                            double rear_axle_to_center, const float *Tcb12 /* host */,
                            float *d_cam_xyz, void *stream);

/* Frame::GuidenceKeyBirdPts + nearEdges + genEdgesPC (src/Frame.cc:671-739, called at :342 between the bird
 * detect and compute steps) and the mask of extractorBird->detect(img, kps, mask) (Frame.cc:337-339).
 *
 * A bird key point survives iff
 *   (mask == NULL || mask[(int)(y + .5f)][(int)(x + .5f)] != 0)          -- cv::KeyPointsFilter::runByPixelsMask on the
 *                                                                           level-0 position (the per-level masks of
 *                                                                           cv::ORB are part of the E9 substitution)
 *   && nearEdges(kp): any contour pixel >= 10 in the box  row in [trunc(max(x-10,0)), min(x+10, cols)),
 *                     col in [trunc(max(y-10,0)), min(y+10, rows))  -- sic: the key point's X selects the image ROW and
 *                     is clamped with cols, its Y selects the COLUMN and is clamped with rows (Frame.cc:719-729); loop
 *                     bounds are float compares of a size_t counter, as in the reference.  On a non-square contour image
 *                     the reference reads out of bounds; here pixels outside the image count as 0 (free).
 * Survivors are appended in input order (push_back, Frame.cc:680); descriptors (computed before the filter in this
 * build, after it in the reference) move with their key points.  Outputs must not alias the inputs.
 * genEdgesPC (Frame.cc:686-715; lists nothing in the reference ever reads) is optional: edge_sign = pixels in [10,150),
 * edge_free = pixels >= 150, as (x = col, y = row) float pairs in raster order, at most edge_cap each (counts are
 * the true totals; entries beyond edge_cap are dropped). */
typedef struct fb_bird_guidance_args {
  int32_t batch, kp_stride;
  int32_t cols, rows, pitch;            /* mBirdviewContourICP / mBirdviewMask geometry; images rows*pitch bytes apart */
  const uint8_t *contour;               /* [batch][rows][pitch] u8                                               */
  const uint8_t *mask;                  /* optional, same geometry                                               */
  const int32_t *n_in;                  /* [batch]                                                               */
  const fb_keypoint *kps_in;            /* [batch][kp_stride]  preKeysBird (level-0 pixel coordinates)           */
  const uint8_t *desc_in;               /* optional [batch][kp_stride][32]                                       */
  int32_t *n_out;                       /* [batch]             Nbird                                             */
  fb_keypoint *kps_out;                 /* [batch][kp_stride]  mvKeysBird                                        */
  uint8_t *desc_out;                    /* [batch][kp_stride][32] (required iff desc_in)                         */
  uint8_t *keep;                        /* optional [batch][kp_stride]: 1 = survived                             */
  int32_t edge_cap;                     /* 0 = skip genEdgesPC                                                   */
  int32_t *n_edge_sign, *n_edge_free;   /* [batch]                                                               */
  float *edge_sign, *edge_free;         /* [batch][edge_cap][2]                                                  */
} fb_bird_guidance_args;
int fb_bird_guidance(const fb_bird_guidance_args *a);                    /* host pointers  */
int fb_bird_guidance_dev(const fb_bird_guidance_args *a, void *stream);  /* device pointers */

/* ======================================================================== */
/* ORBmatcher (include/ORBmatcher.h:41-88, src/ORBmatcher.cc)                */
/* ======================================================================== */
typedef struct fb_matcher_params { /* ORBmatcher.h:41 */
  float nnratio;
  int32_t check_orientation;
} fb_matcher_params;

/* DescriptorDistance (ORBmatcher.cc:1951-1967): out[i] = popcount(a[i]^b[i]) over 32 bytes */
int fb_descriptor_distance_dev(const uint8_t *d_a, const uint8_t *d_b, int n, int32_t *d_out,
                               void *stream);
int fb_descriptor_distance(const uint8_t *a, const uint8_t *b, int n, int32_t *out);

/* Camera + frame constants used by the projection matchers (Frame.h statics) */
typedef struct fb_camera {
  float fx, fy, cx, cy;
  float min_x, min_y, max_x, max_y; /* mnMinX.. image bounds, Frame.cc:741-795 */
} fb_camera;

/* --- M3: SearchByProjection(Frame& cur, const Frame& last, th, bMono=true)
 *     (ORBmatcher.cc:1329-1471).  One problem per batch entry.  All arrays are
 *     [batch][stride] with the per-problem counts in n_cur / n_last.            */
typedef struct fb_proj_frame_args {
  int32_t batch;
  int32_t cur_stride;  /* entries per problem in the cur_* arrays            */
  int32_t last_stride; /* entries per problem in the last_* arrays           */
  /* current frame = search targets */
  const int32_t *n_cur;          /* [batch]                                   */
  const fb_keypoint *cur_kps;    /* mvKeysUn: x,y,octave,angle used            */
  const uint8_t *cur_desc;       /* mDescriptors rows                          */
  const int32_t *cur_cell_start; /* [batch][cols*rows+1]                       */
  const int32_t *cur_cell_items; /* [batch][cur_stride]                        */
  const uint8_t *cur_blocked;    /* mvpMapPoints[i] && Observations()>0 on entry; may be NULL */
  const float *cur_Tcw;          /* [batch][12] row-major 3x4 of mTcw          */
  /* last frame = queries */
  const int32_t *n_last;         /* [batch]                                   */
  const uint8_t *last_valid;     /* mvpMapPoints[i] && !mvbOutlier[i]          */
  const uint8_t *last_obs_pos;   /* pMP->Observations()>0                      */
  const float *last_xw;          /* [..][3] pMP->GetWorldPos()                 */
  const uint8_t *last_desc;      /* pMP->GetDescriptor()                       */
  const int32_t *last_octave;    /* LastFrame.mvKeys[i].octave                 */
  const float *last_angle;       /* LastFrame.mvKeysUn[i].angle                */
  /* constants */
  fb_camera cam;
  fb_grid_geom grid;
  float scale_factors[FB_MAX_LEVELS]; /* CurrentFrame.mvScaleFactors           */
  float th;
  fb_matcher_params matcher;
  /* outputs */
  int32_t *match_cur_to_last; /* [batch][cur_stride]: index into last, or -1   */
  int32_t *nmatches;          /* [batch] return value                          */
  /* Tracking.cc:1342-1349 in the same launch: a frame whose search returned fewer than retry_below matches (20 there) is
   * searched again from scratch with retry_th (2 * th there); match_cur_to_last / nmatches then hold the second result.
   * retry_below = 0: one search.                                              */
  int32_t retry_below;
  float retry_th;
  int32_t *retried;           /* [batch] or NULL: 1 = the second search ran     */
} fb_proj_frame_args;
int fb_match_projection_frame_dev(const fb_proj_frame_args *args, void *stream);
int fb_match_projection_frame(const fb_proj_frame_args *args); /* host pointers */

/* --- M9: BirdMapPointMatch(CurF, vRefMapPointsBird, windowSize, filterSize)
 *     (ORBmatcher.cc:1763-1902)                                               */
typedef struct fb_bird_mp_args {
  int32_t batch;
  int32_t cur_stride;
  int32_t ref_stride;
  const int32_t *n_cur;           /* mvKeysBird.size() == mDescriptorsBird.rows */
  const fb_keypoint *cur_kps;     /* mvKeysBird                                */
  const uint8_t *cur_desc;        /* mDescriptorsBird                          */
  const float *cur_cam_xyz;       /* [..][3] mvKeysBirdCamXYZ                  */
  const int32_t *cur_cell_start;  /* bird grid CSR                             */
  const int32_t *cur_cell_items;
  const float *cur_Tcw;           /* [batch][12]                               */
  const int32_t *n_ref;
  const uint8_t *ref_valid;       /* vRefMapPointsBird[i] != NULL              */
  const float *ref_xw;            /* [..][3]                                   */
  const uint8_t *ref_desc;
  float Tbc[12];                  /* Frame::Tbc rows 0..2 (Frame.cc:1018-1030) */
  int32_t bird_cols, bird_rows;   /* Frame::birdviewCols/Rows                  */
  double meter2pixel;             /* Frame.cc:41                               */
  double rear_axle_to_center;     /* Frame.cc:42                               */
  fb_grid_geom grid;              /* 32x32, min=0                              */
  int32_t window_size;
  float filter_size;
  fb_matcher_params matcher;
  int32_t *match_cur_to_ref;      /* [batch][cur_stride]: mvpMapPointsBird as ref index or -1 (in/out: pre-fill) */
  int32_t *ninliers;              /* [batch] return value                      */
} fb_bird_mp_args;
int fb_match_bird_mappoints_dev(const fb_bird_mp_args *args, void *stream);
int fb_match_bird_mappoints(const fb_bird_mp_args *args);

/* --- M2: SearchByProjection(Frame&, vector<MapPoint*>&, th) (ORBmatcher.cc:46-130) */
typedef struct fb_proj_points_args {
  int32_t batch;
  int32_t cur_stride;
  int32_t mp_stride;
  const int32_t *n_cur;
  const fb_keypoint *cur_kps;
  const uint8_t *cur_desc;
  const int32_t *cur_cell_start;
  const int32_t *cur_cell_items;
  const uint8_t *cur_blocked;    /* F.mvpMapPoints[idx] && Observations()>0 on entry */
  const int32_t *n_mp;
  const uint8_t *mp_track;       /* mbTrackInView && !isBad()                  */
  const uint8_t *mp_obs_pos;     /* pMP->Observations()>0                      */
  const float *mp_proj;          /* [..][2] mTrackProjX, mTrackProjY           */
  const int32_t *mp_level;       /* mnTrackScaleLevel                          */
  const float *mp_view_cos;      /* mTrackViewCos                              */
  const uint8_t *mp_desc;
  fb_grid_geom grid;
  float scale_factors[FB_MAX_LEVELS];
  float th;
  fb_matcher_params matcher;
  int32_t *match_cur_to_mp;      /* [batch][cur_stride] index into mp or -1     */
  int32_t *nmatches;
  /* optional device scratch of fb_match_projection_points_workspace(batch, mp_stride) bytes, 16-byte aligned, private to
   * this call until it has run: with it the grid walks and Hamming distances are computed once by many workgroups per
   * problem and only the serial "already taken" rule runs in one workgroup (same results; several times faster for the
   * few-thousand-point local maps of TrackLocalMap, above all at small batch).  NULL = the one-kernel version.        */
  void *workspace;
  size_t workspace_bytes;
} fb_proj_points_args;
size_t fb_match_projection_points_workspace(int batch, int mp_stride);
int fb_match_projection_points_dev(const fb_proj_points_args *args, void *stream);
int fb_match_projection_points(const fb_proj_points_args *args);

/* --- M8: BirdviewMatch(CurF, refKeys, refDesc, refMPBirds, matches, isProject=0, window)
 *     (ORBmatcher.cc:1602-1760), live form isProject=0.
 *     NOT BUILT: the isProject != 0 branch (ORBmatcher.cc:1617-1655: search centre = the reference MapPointBird projected
 *     with Tbc * Tcw, |z| <= 0.2 m gate, no level filter).  No call site of the reference uses it (Tracking.cc:384,407,
 *     450,959,2728 all pass 0); a caller that needs projected bird search has fb_match_bird_mappoints (M9), which is that
 *     projection + window search (BirdMapPointMatch, ORBmatcher.cc:1763-1902).                                        */
typedef struct fb_birdview_args {
  int32_t batch;
  int32_t cur_stride;
  int32_t ref_stride;
  const int32_t *n_cur;
  const fb_keypoint *cur_kps;
  const uint8_t *cur_desc;
  const int32_t *cur_cell_start;
  const int32_t *cur_cell_items;
  const int32_t *n_ref;
  const fb_keypoint *ref_kps;
  const uint8_t *ref_desc;
  fb_grid_geom grid;
  int32_t window_size;
  fb_matcher_params matcher;
  int32_t *match_ref_to_cur; /* [batch][ref_stride] vnMatches12 after culling (-1 = none) */
  int32_t *match_dist;       /* [batch][ref_stride] vMatchedDistance                      */
  int32_t *nmatches;         /* [batch] return value                                      */
  int32_t *n_dmatches;       /* [batch] number of DMatch emitted (vnMatches12[i] > 0)     */
} fb_birdview_args;
int fb_match_birdview_dev(const fb_birdview_args *args, void *stream);
int fb_match_birdview(const fb_birdview_args *args);

/* DBoW2::FeatureVector (Thirdparty/DBoW2/DBoW2/FeatureVector.h:23) = map<NodeId, vector<unsigned>> as CSR:
 * node ids ascending, node n owns items[node_start[n] .. node_start[n+1]) in addFeature order.          */
typedef struct fb_feature_vector {
  int32_t node_stride;        /* entries per problem in node_ids (node_start has node_stride+1)   */
  int32_t item_stride;        /* entries per problem in items                                      */
  const int32_t *n_nodes;     /* [batch]                                                            */
  const uint32_t *node_ids;   /* [batch][node_stride]                                               */
  const int32_t *node_start;  /* [batch][node_stride+1]                                             */
  const int32_t *items;       /* [batch][item_stride] feature indices                               */
} fb_feature_vector;

/* --- M5: SearchByBoW(KeyFrame* pKF, Frame &F, vector<MapPoint*>&) (ORBmatcher.cc:160-289) */
typedef struct fb_bow_args {
  int32_t batch;
  int32_t kf_stride, f_stride;
  const int32_t *n_kf;          /* pKF->N                                                            */
  const fb_keypoint *kf_kps;    /* pKF->mvKeysUn (angle)                                             */
  const uint8_t *kf_desc;
  const uint8_t *kf_has_mp;     /* vpMapPointsKF[i] && !isBad()                                      */
  fb_feature_vector kf_fv;      /* pKF->mFeatVec                                                     */
  const int32_t *n_f;           /* F.N                                                               */
  const fb_keypoint *f_kps;     /* F.mvKeys (angle)                                                  */
  const uint8_t *f_desc;
  fb_feature_vector f_fv;       /* F.mFeatVec                                                        */
  fb_matcher_params matcher;    /* ORBmatcher(0.7,true) at Tracking.cc:1207                          */
  int32_t *match_f_to_kf;       /* [batch][f_stride]: KF feature whose MapPoint lands in slot i, -1  */
  int32_t *nmatches;            /* [batch]                                                           */
} fb_bow_args;
int fb_match_bow_dev(const fb_bow_args *args, void *stream);
int fb_match_bow(const fb_bow_args *args);

/* --- M7: SearchForTriangulation(pKF1, pKF2, F12, pairs, bOnlyStereo=false) (ORBmatcher.cc:658-824) */
typedef struct fb_triangulation_args {
  int32_t batch;
  int32_t kf1_stride, kf2_stride;
  const int32_t *n1;
  const fb_keypoint *kps1;      /* pKF1->mvKeysUn                                                    */
  const uint8_t *desc1;
  const uint8_t *has_mp1;       /* pKF1->GetMapPoint(i) != NULL                                      */
  fb_feature_vector fv1;
  const int32_t *n2;
  const fb_keypoint *kps2;
  const uint8_t *desc2;
  const uint8_t *has_mp2;
  fb_feature_vector fv2;
  const float *F12;             /* [batch][9] row-major                                              */
  const float *Cw1;             /* [batch][3] pKF1->GetCameraCenter()                                */
  const float *R2w;             /* [batch][9] pKF2->GetRotation()                                    */
  const float *t2w;             /* [batch][3]                                                        */
  float fx, fy, cx, cy;         /* pKF2 intrinsics                                                   */
  float scale_factors[FB_MAX_LEVELS]; /* pKF2->mvScaleFactors                                        */
  float level_sigma2[FB_MAX_LEVELS];  /* pKF2->mvLevelSigma2                                         */
  fb_matcher_params matcher;    /* ORBmatcher(0.6,false) at LocalMapping.cc:239                      */
  int32_t *matches12;           /* [batch][kf1_stride] vMatches12                                    */
  int32_t *nmatches;            /* [batch]                                                           */
} fb_triangulation_args;
int fb_match_triangulation_dev(const fb_triangulation_args *args, void *stream);
int fb_match_triangulation(const fb_triangulation_args *args);

/* --- M4: SearchByProjection(Frame &CurrentFrame, KeyFrame *pKF, const set<MapPoint*> &sAlreadyFound,
 *     th, ORBdist) (ORBmatcher.cc:1473-1600; relocalisation, Tracking.cc:1632,1643).              */
typedef struct fb_proj_kf_args {
  int32_t batch;
  int32_t cur_stride;
  int32_t kf_stride;
  const int32_t *n_cur;
  const fb_keypoint *cur_kps;    /* CurrentFrame.mvKeysUn                                            */
  const uint8_t *cur_desc;
  const int32_t *cur_cell_start;
  const int32_t *cur_cell_items;
  const uint8_t *cur_blocked;    /* CurrentFrame.mvpMapPoints[i2] != NULL on entry; may be NULL      */
  const float *cur_Tcw;          /* [batch][12]                                                      */
  const int32_t *n_kf;           /* vpMPs.size()                                                     */
  const uint8_t *kf_valid;       /* vpMPs[i] && !isBad() && !sAlreadyFound.count(vpMPs[i])           */
  const float *kf_xw;            /* [..][3] GetWorldPos()                                            */
  const uint8_t *kf_desc;        /* GetDescriptor()                                                  */
  const float *kf_max_dist;      /* mfMaxDistance (MapPoint.cc:379-383 applies the 1.2f)             */
  const float *kf_min_dist;      /* mfMinDistance (MapPoint.cc:373-377 applies the 0.8f)             */
  const float *kf_angle;         /* pKF->mvKeysUn[i].angle                                           */
  fb_camera cam;
  fb_grid_geom grid;
  float scale_factors[FB_MAX_LEVELS];
  float log_scale_factor;        /* CurrentFrame.mfLogScaleFactor                                    */
  int32_t n_levels;              /* CurrentFrame.mnScaleLevels                                       */
  float th;
  int32_t orb_dist;
  fb_matcher_params matcher;
  int32_t *match_cur_to_kf;      /* [batch][cur_stride]: index into the keyframe's points, or -1     */
  int32_t *nmatches;
} fb_proj_kf_args;
int fb_match_projection_keyframe_dev(const fb_proj_kf_args *args, void *stream);
int fb_match_projection_keyframe(const fb_proj_kf_args *args);

/* --- M6: SearchByBoW(KeyFrame *pKF1, KeyFrame *pKF2, vector<MapPoint*> &vpMatches12)
 *     (ORBmatcher.cc:523-656; loop closing, LoopClosing.cc:263).                                   */
typedef struct fb_bow_kf_args {
  int32_t batch;
  int32_t kf1_stride, kf2_stride;
  const int32_t *n1;            /* vpMapPoints1.size()                                               */
  const fb_keypoint *kps1;      /* pKF1->mvKeysUn (angle)                                            */
  const uint8_t *desc1;
  const uint8_t *has_mp1;       /* vpMapPoints1[i] && !isBad()                                       */
  fb_feature_vector fv1;
  const int32_t *n2;
  const fb_keypoint *kps2;
  const uint8_t *desc2;
  const uint8_t *has_mp2;       /* vpMapPoints2[i] && !isBad()                                       */
  fb_feature_vector fv2;
  fb_matcher_params matcher;    /* ORBmatcher(0.75,true) at LoopClosing.cc:240                       */
  int32_t *matches12;           /* [batch][kf1_stride]: KF2 feature whose MapPoint is vpMatches12[i], -1 */
  int32_t *nmatches;
} fb_bow_kf_args;
int fb_match_bow_kf_dev(const fb_bow_kf_args *args, void *stream);
int fb_match_bow_kf(const fb_bow_kf_args *args);

/* --- M10: the remaining ORBmatcher entry points (same windowed Hamming search) ------------------ */
/* key-frame side of a map-point -> key-frame search (KeyFrame.h members) */
typedef struct fb_kf_target {
  int32_t kf_stride;
  const int32_t *n_kf;            /* pKF->N                                                        */
  const fb_keypoint *kf_kps;      /* pKF->mvKeysUn                                                 */
  const uint8_t *kf_desc;         /* pKF->mDescriptors                                             */
  const int32_t *kf_cell_start;   /* pKF->mGrid as CSR (fb_grid_build_batch_dev)                   */
  const int32_t *kf_cell_items;
  fb_camera cam;                  /* fx, fy, cx, cy, mnMinX/Y, mnMaxX/Y                            */
  fb_grid_geom grid;
  float scale_factors[FB_MAX_LEVELS];    /* pKF->mvScaleFactors                                    */
  float inv_level_sigma2[FB_MAX_LEVELS]; /* pKF->mvInvLevelSigma2                                  */
  float log_scale_factor;         /* pKF->mfLogScaleFactor                                         */
  int32_t n_levels;               /* pKF->mnScaleLevels                                            */
} fb_kf_target;

/* candidate map points (MapPoint getters) */
typedef struct fb_mp_list {
  int32_t mp_stride;
  const int32_t *n_mp;
  const uint8_t *mp_valid;        /* the entry point's own skip rule, see each function             */
  const float *mp_xw;             /* [..][3] GetWorldPos()                                         */
  const float *mp_normal;         /* [..][3] GetNormal()                                           */
  const float *mp_max_dist;       /* mfMaxDistance                                                 */
  const float *mp_min_dist;       /* mfMinDistance                                                 */
  const uint8_t *mp_desc;         /* GetDescriptor()                                               */
} fb_mp_list;

/* Fuse(pKF, vpMapPoints, th) (ORBmatcher.cc:826-976; LocalMapping.cc:513,538) and
 * Fuse(pKF, Scw, vpPoints, th, vpReplacePoint) (:978-1101; LoopClosing.cc:567).
 * The search half of Fuse: best_idx[i] = bestIdx if bestDist <= TH_LOW, else -1.  The map mutation that follows
 * (Replace / AddObservation / AddMapPoint, :950-971 and :1082-1096) stays with the caller, who walks best_idx in
 * order; the search result does not depend on those mutations.
 * mp_valid: Fuse = pMP && !isBad() && !IsInKeyFrame(pKF); Fuse-Sim3 = !isBad() && !spAlreadyFound.count(pMP).   */
typedef struct fb_fuse_args {
  int32_t batch;
  fb_kf_target kf;
  fb_mp_list mp;
  const float *pose;              /* [batch][12]: GetRotation|GetTranslation (Fuse) or rows 0..2 of Scw (Sim3) */
  const float *Ow;                /* [batch][3] GetCameraCenter() (Fuse); unused by the Sim3 variant  */
  float th;
  int32_t *best_idx;              /* [batch][mp_stride]                                             */
} fb_fuse_args;
int fb_fuse_search_dev(const fb_fuse_args *args, void *stream);
int fb_fuse_search(const fb_fuse_args *args);
int fb_fuse_sim3_search_dev(const fb_fuse_args *args, void *stream);
int fb_fuse_sim3_search(const fb_fuse_args *args);

/* SearchByProjection(pKF, Scw, vpPoints, vpMatched, th) (ORBmatcher.cc:291-404; LoopClosing.cc:377).
 * mp_valid = !isBad() && !spAlreadyFound.count(pMP).                                                 */
typedef struct fb_proj_sim3_args {
  int32_t batch;
  fb_kf_target kf;
  fb_mp_list mp;
  const float *Scw;               /* [batch][12]                                                    */
  const uint8_t *kf_matched;      /* vpMatched[idx] != NULL on entry                                */
  int32_t th;
  int32_t *match_kf_to_mp;        /* [batch][kf_stride]: point newly written to vpMatched[idx], or -1 */
  int32_t *nmatches;
} fb_proj_sim3_args;
int fb_match_projection_sim3_dev(const fb_proj_sim3_args *args, void *stream);
int fb_match_projection_sim3(const fb_proj_sim3_args *args);

/* SearchBySim3(pKF1, pKF2, vpMatches12, s12, R12, t12, th) (ORBmatcher.cc:1103-1327; LoopClosing.cc:341).
 * Per key-frame feature i: the MapPoint it holds (mp*.{xw,desc,max,min}[i]), mp*.mp_valid[i] =
 * vpMapPoints[i] && !vbAlreadyMatched[i] && !isBad().  kf1/kf2 strides equal mp1/mp2 strides.        */
typedef struct fb_sim3_args {
  int32_t batch;
  fb_kf_target kf1, kf2;
  fb_mp_list mp1, mp2;
  const float *T1w, *T2w;         /* [batch][12] GetRotation|GetTranslation of pKF1 / pKF2          */
  const float *s12;               /* [batch]                                                        */
  const float *R12;               /* [batch][9]                                                     */
  const float *t12;               /* [batch][3]                                                     */
  float th;
  int32_t *matches12;             /* [batch][kf1 stride]: idx2 with vnMatch1[i1]==idx2 && vnMatch2[idx2]==i1, or -1 */
  int32_t *nfound;
} fb_sim3_args;
int fb_match_sim3_dev(const fb_sim3_args *args, void *stream);
int fb_match_sim3(const fb_sim3_args *args);

/* SearchForInitialization(F1, F2, vbPrevMatched, vnMatches12, windowSize) (ORBmatcher.cc:406-521; Tracking.cc:1293) */
typedef struct fb_init_match_args {
  int32_t batch;
  int32_t f1_stride, f2_stride;
  const int32_t *n1;
  const fb_keypoint *kps1;        /* F1.mvKeysUn                                                    */
  const uint8_t *desc1;
  const int32_t *n2;
  const fb_keypoint *kps2;        /* F2.mvKeysUn                                                    */
  const uint8_t *desc2;
  const int32_t *f2_cell_start;   /* F2.mGrid                                                       */
  const int32_t *f2_cell_items;
  fb_grid_geom grid;
  int32_t window_size;
  fb_matcher_params matcher;      /* ORBmatcher(0.9,true) at Tracking.cc:1292                       */
  float *prev_matched;            /* [batch][f1_stride][2] vbPrevMatched, in/out                    */
  int32_t *matches12;             /* [batch][f1_stride] vnMatches12                                 */
  int32_t *nmatches;
} fb_init_match_args;
int fb_match_initialization_dev(const fb_init_match_args *args, void *stream);
int fb_match_initialization(const fb_init_match_args *args);

/* MapPoint::ComputeDistinctiveDescriptors (MapPoint.cc:242-307): for each map point, the observation descriptor with
 * the least median Hamming distance to the others.  CSR over map points: obs_start[n_mp+1], obs_desc[obs_start[n_mp]][32].
 * best_obs[i] = index (relative to obs_start[i]) of the chosen descriptor, -1 when the point has no observation. */
int fb_distinctive_descriptors_dev(const int32_t *d_obs_start, const uint8_t *d_obs_desc, int n_mp,
                                   int32_t *d_best_obs, void *stream);
int fb_distinctive_descriptors(const int32_t *obs_start, const uint8_t *obs_desc, int n_mp, int32_t *best_obs);

/* ======================================================================== */
/* DBoW2 vocabulary transform (Frame::ComputeBoW, src/Frame.cc:628-635)       */
/* ======================================================================== */
/* DBoW2::TemplatedVocabulary<FORB::TDescriptor, FORB> as flat arrays
 * (Thirdparty/DBoW2/DBoW2/TemplatedVocabulary.h:297-329: Node{weight, children, descriptor, word_id}, m_L).
 * The reference ships no vocabulary file; any tree (k, L) in this form works. */
typedef struct fb_vocabulary {
  int32_t n_nodes;
  int32_t L;                    /* m_L, depth of the leaves                                          */
  const int32_t *child_start;   /* [n_nodes+1] CSR over m_nodes[i].children                           */
  const int32_t *children;      /* NodeIds in the children vector's order                            */
  const uint8_t *descriptors;   /* [n_nodes][32]                                                      */
  const double *weights;        /* [n_nodes] Node::weight (used at the leaves)                        */
  const int32_t *word_ids;      /* [n_nodes] Node::word_id (used at the leaves)                       */
} fb_vocabulary;

/* transform(features, BowVector&, FeatureVector&, levelsup) for TF_IDF weighting + L1 scoring (ORBVocabulary),
 * TemplatedVocabulary.h:1126-1194, per-feature descent :1217-1259, BowVector::addWeight / normalize(L1)
 * (BowVector.cpp:30-84).  Outputs per image: the BowVector as (word id ascending, value) pairs and the FeatureVector
 * in the CSR form the matchers take (fb_feature_vector).  All output arrays have f_stride entries per image
 * (fv_node_start: f_stride + 1); entries at or beyond the returned counts (and below n_f) are scratch of the call. */
typedef struct fb_bow_transform_args {
  int32_t batch;
  int32_t f_stride;
  const int32_t *n_f;           /* features per image (<= 4096)                                       */
  const uint8_t *desc;          /* [batch][f_stride][32] mDescriptors                                 */
  int32_t levelsup;             /* 4 at Frame.cc:633                                                  */
  int32_t *n_words;             /* [batch] mBowVec.size()                                             */
  uint32_t *bow_ids;            /* [batch][f_stride]                                                  */
  double *bow_vals;             /* [batch][f_stride]                                                  */
  int32_t *fv_n_nodes;          /* [batch] mFeatVec.size()                                            */
  uint32_t *fv_node_ids;        /* [batch][f_stride]                                                  */
  int32_t *fv_node_start;       /* [batch][f_stride + 1]                                              */
  int32_t *fv_items;            /* [batch][f_stride]                                                  */
} fb_bow_transform_args;
int fb_bow_transform_dev(const fb_vocabulary *voc /* device arrays */, const fb_bow_transform_args *args, void *stream);
int fb_bow_transform(const fb_vocabulary *voc, const fb_bow_transform_args *args); /* host pointers */

/* ======================================================================== */
/* Frame geometry either side of the matchers (src/Frame.cc)                 */
/* ======================================================================== */
/* --- Frame::isInFrustum(pMP, viewingCosLimit) over a list of map points (Frame.cc:435-491; the loop
 *     around it is Tracking::SearchLocalPoints, Tracking.cc:1071-1091).  Outputs of a point that is
 *     not in view are left untouched, as the reference leaves the MapPoint's track members.        */
typedef struct fb_frustum_args {
  int32_t batch;
  int32_t mp_stride;
  const float *Tcw;             /* [batch][12] mRcw | mtcw                                           */
  const float *Ow;              /* [batch][3] mOw                                                    */
  const int32_t *n_mp;
  const uint8_t *mp_valid;      /* the caller's filter (not already matched, !isBad()); NULL = all   */
  const float *mp_xw;           /* [..][3]                                                           */
  const float *mp_normal;       /* [..][3] GetNormal()                                               */
  const float *mp_max_dist;     /* mfMaxDistance                                                     */
  const float *mp_min_dist;     /* mfMinDistance                                                     */
  fb_camera cam;
  float mbf;
  float viewing_cos_limit;      /* 0.5 at Tracking.cc:1084                                           */
  float log_scale_factor;
  int32_t n_levels;
  uint8_t *in_view;             /* mbTrackInView                                                     */
  float *proj;                  /* [..][2] mTrackProjX, mTrackProjY                                  */
  float *proj_xr;               /* mTrackProjXR; may be NULL                                         */
  int32_t *level;               /* mnTrackScaleLevel                                                 */
  float *view_cos;              /* mTrackViewCos                                                     */
} fb_frustum_args;
int fb_in_frustum_dev(const fb_frustum_args *args, void *stream);
int fb_in_frustum(const fb_frustum_args *args);

/* --- the geometric test of Tracking::FilterBirdOutlierInFront (Tracking.cc:1825-1914) over the bird matches of a
 *     frame pair: match i passes when |Tcw2 * (Twc1 * Xc1[query_i]) - Xc2[train_i]| < windowSize (:1868-1886) and its
 *     train slot holds no MapPointBird yet -- on entry (occupied2) or through an earlier passing match (:1861-1863).
 *     The MapPointBird bookkeeping (:1891-1903) stays with the caller, who walks `keep` in order.                  */
typedef struct fb_bird_filter_args {
  int32_t batch;
  int32_t match_stride, kp1_stride, kp2_stride;
  const int32_t *n_matches;       /* vDMatches12.size()                                              */
  const int32_t *query_idx;       /* [batch][match_stride] DMatch::queryIdx (frame 1)                */
  const int32_t *train_idx;       /* DMatch::trainIdx (frame 2)                                      */
  const float *cam_xyz1;          /* [batch][kp1_stride][3] MatchedFrame1->mvKeysBirdCamXYZ          */
  const float *cam_xyz2;          /* [batch][kp2_stride][3] MatchedFrame2->mvKeysBirdCamXYZ          */
  const float *Tcw1, *Tcw2;       /* [batch][12]                                                     */
  const uint8_t *occupied2;       /* [batch][kp2_stride] MatchedFrame2->mvpMapPointsBird[i] != NULL   */
  float window_size;
  uint8_t *keep;                  /* [batch][match_stride] 1 = pushed to newMatch                     */
  float *pt_world;                /* [batch][match_stride][3] ptwC (position of a new MapPointBird)   */
} fb_bird_filter_args;
int fb_bird_filter_matches_dev(const fb_bird_filter_args *args, void *stream);
int fb_bird_filter_matches(const fb_bird_filter_args *args);

/* --- Frame::UndistortKeyPoints (Frame.cc:636-669): cv::fisheye::undistortPoints(pts, K, D, R=I, P=K) on
 *     every key point; D[0]==0 copies.  K4 = fx,fy,cx,cy, D4 = k1..k4 (host).  d_kps_un may alias d_kps. */
int fb_undistort_keypoints_dev(const fb_keypoint *d_kps, const int32_t *d_n, int batch, int kp_stride,
                               const float *K4, const float *D4, fb_keypoint *d_kps_un, void *stream);
int fb_undistort_keypoints(const fb_keypoint *kps, int n, const float *K4, const float *D4,
                           fb_keypoint *kps_un); /* host pointers */
/* --- Frame::ComputeImageBounds (Frame.cc:741-795): bounds[4] = mnMinX, mnMaxX, mnMinY, mnMaxY.      */
int fb_image_bounds(int cols, int rows, const float *K4, const float *D4, float *bounds);

/* ======================================================================== */
/* Optimizer::PoseOptimization / PoseOptimizationWithBird / BirdOptimization */
/* (include/Optimizer.h:40-68, src/Optimizer.cc:246-835)                     */
/* ======================================================================== */
enum { FB_POSE_FRONT = 0, FB_POSE_FRONT_BIRD = 1, FB_POSE_BIRD = 2 };

typedef struct fb_pose_opt_args {
  int32_t batch;
  int32_t mode;          /* FB_POSE_*                                          */
  int32_t front_stride;  /* entries per problem in front_* arrays              */
  int32_t bird_stride;
  float fx, fy, cx, cy;
  float wF, wB;          /* Optimizer.h:52 defaults 1,1                        */
  /* front edges: one per mvpMapPoints[i] != NULL (Optimizer.cc:525-571)        */
  const int32_t *n_front;        /* [batch]                                    */
  const float *front_xw;         /* [..][3]                                    */
  const float *front_obs;        /* [..][2] kpUn.pt                            */
  const float *front_inv_sigma2; /* mvInvLevelSigma2[kpUn.octave]              */
  const uint8_t *front_valid;    /* slot i carries an edge (mvpMapPoints[i] != NULL); NULL = all */
  /* bird edges: one per mvpMapPointsBird[i] != NULL (Optimizer.cc:575-602)     */
  const int32_t *n_bird;
  const float *bird_xw;          /* [..][3]                                    */
  const float *bird_xc;          /* [..][3] mvKeysBirdCamXYZ                   */
  const float *bird_inv_sigma2;
  const uint8_t *bird_valid;     /* mvpMapPointsBird[i] != NULL; NULL = all       */
  uint8_t *bird_outlier;         /* in/out mvBirdOutlier (not reset by the reference) */
  /* pose in/out: pFrame->mTcw rows 0..2, row-major 3x4 float                   */
  float *Tcw;                    /* [batch][12]                                */
  uint8_t *front_outlier;        /* out mvbOutlier                             */
  int32_t *ninliers;             /* [batch] return value                       */
} fb_pose_opt_args;
int fb_pose_opt_batch_dev(const fb_pose_opt_args *args, void *stream);

/* Device-side form of the edge construction loops of PoseOptimizationWithBird
 * (Optimizer.cc:525-571 front, :575-602 bird): slot i of the frame becomes edge i.
 *   front: valid = match[i] >= 0, Xw = mp_xw[match[i]], obs = kps[i].pt,
 *          inv_sigma2 = inv_level_sigma2[kps[i].octave]
 *   bird : additionally Xc = cam_xyz[i]                                        */
int fb_pose_gather_front_dev(int batch, int kp_stride, int mp_stride, const int32_t *d_n,
                             const fb_keypoint *d_kps, const int32_t *d_match, const float *d_mp_xw,
                             const float *inv_level_sigma2 /* host, nlevels */, int nlevels,
                             float *d_front_xw, float *d_front_obs, float *d_front_inv_sigma2,
                             uint8_t *d_front_valid, void *stream);
int fb_pose_gather_bird_dev(int batch, int kp_stride, int mp_stride, const int32_t *d_n,
                            const fb_keypoint *d_kps, const float *d_cam_xyz, const int32_t *d_match,
                            const float *d_mpb_xw, const float *inv_level_sigma2, int nlevels,
                            float *d_bird_xw, float *d_bird_xc, float *d_bird_inv_sigma2,
                            uint8_t *d_bird_valid, void *stream);
int fb_pose_opt(const fb_pose_opt_args *args); /* host pointers */

/* ======================================================================== */
/* Device-resident Frame and the per-frame tracking chain                    */
/* (include/Frame.h, src/Frame.cc:262-379; src/Tracking.cc:1312-1441,690-746) */
/* ======================================================================== */
/* An fb_frame is the reference's Frame object for `batch` independent sequences, with every per-key-point member living
 * in HBM: mvKeys / mvKeysUn / mDescriptors / mGrid / mvpMapPoints / mvbOutlier, their bird counterparts (mvKeysBird,
 * mDescriptorsBird, mvKeysBirdCamXYZ, mGridBirdview, mvpMapPointsBird, mvBirdOutlier) and mTcw / mOw.  The entry points
 * below take frames where the reference's ORBmatcher / Optimizer methods take Frame& / Frame*, so that consecutive calls
 * (extract -> match -> optimise -> match -> optimise, and frame k -> frame k+1) hand their results over on the device.
 * mvpMapPoints[i] / mvpMapPointsBird[i] are kept as int32 indices into the caller's map tables (-1 = NULL).
 *
 * All *_dev entry points only enqueue on `stream`; the return values of the reference functions (match / inlier counts)
 * land in the frame's counter block, which fb_frame_counts reads back (one synchronisation per frame instead of one per
 * call).  The Tracking state machine stays with the host: it reads the counters and decides on retries (Tracking.cc:1345),
 * LOST handling and key-frame insertion.  One call at a time per frame handle.                                          */
typedef struct fb_frame fb_frame;

typedef struct fb_frame_params {
  int32_t batch;                    /* sequences side by side; 1 = the reference's single Frame                        */
  int32_t front_width, front_height, bird_width, bird_height;
  fb_orb_params orb;                /* ORBextractor.* of the settings file: capacity and level tables (Frame.cc:299-306)  */
  int32_t bird_nfeatures;           /* features of the bird extractor (the reference: cv::ORB::create(2000), Frame.cc:337;
                                       BASELINE configs[2]: 1000 bird edges); 0 = orb.nfeatures.  <= orb.nfeatures        */
  float K[4];                       /* Camera.fx, fy, cx, cy (Tracking.cc:61-75)                                        */
  float D[4];                       /* Camera.k1, k2, p1, p2, fed to the fisheye model as k1..k4 (Frame.cc:657);
                                       D[0] == 0: mvKeysUn = mvKeys (Frame.cc:640-644)                                  */
  float Tbc[12], Tcb[12];           /* Frame::Tbc / Tcb rows 0..2 (CalculateExtrinsics, Frame.cc:1015-1037)             */
  double pixel2meter, meter2pixel, rear_axle_to_center; /* Frame.cc:39-42                                               */
  int32_t map_cap;                  /* largest fb_map_points.stride a call may pass (per-point "seen" flags)            */
  int32_t local_mp_cap;             /* longest mvpLocalMapPoints list a call may pass (>= map stride for NULL lists)    */
  int32_t local_mpb_cap;            /* longest vlocalMPB list (>= bird map stride for NULL lists)                       */
} fb_frame_params;

/* The caller's MapPoint table on the device (MapPoint getters; the Map itself stays with the host, which refreshes the
 * entries LocalMapping changed).  An index into it is what a frame stores in mvpMapPoints.  [batch][stride] arrays.     */
typedef struct fb_map_points {
  int32_t stride;                   /* <= fb_frame_params.map_cap                                                       */
  const int32_t *n;                 /* [batch] entries in use                                                           */
  const uint8_t *bad;               /* isBad()                                                                          */
  const uint8_t *obs_pos;           /* Observations() > 0                                                               */
  const float *xw;                  /* [..][3] GetWorldPos()                                                            */
  const float *normal;              /* [..][3] GetNormal()                                                              */
  const float *max_dist, *min_dist; /* mfMaxDistance, mfMinDistance (the 1.2 / 0.8 factors are applied by the kernels)  */
  const uint8_t *desc;              /* [..][32] GetDescriptor()                                                         */
} fb_map_points;

/* MapPointBird table.  FilterBirdOutlierInFront creates points (Tracking.cc:1896-1901): they are appended at n[b].     */
typedef struct fb_map_points_bird {
  int32_t stride;
  int32_t *n;                       /* [batch] entries in use (in/out)                                                  */
  float *xw;                        /* [..][3] GetWorldPos()                                                            */
  uint8_t *desc;                    /* [..][32] mDescriptor                                                             */
} fb_map_points_bird;

/* slots of a frame's counter block */
enum {
  FB_CNT_BIRD_KF_MATCHES = 0, /* mnBirdKFMatches = BirdMapPointMatch(...)            Tracking.cc:2006                   */
  FB_CNT_PROJ_MATCHES,        /* nmatches = SearchByProjection(cur, last)            Tracking.cc:1339                   */
  FB_CNT_POSE1_INLIERS,       /* PoseOptimizationWithBird                            Tracking.cc:1353                   */
  FB_CNT_MATCHES,             /* nmatches after the outlier discard                  Tracking.cc:1358-1376              */
  FB_CNT_MATCHES_MAP,         /* nmatchesMap                                         Tracking.cc:1373                   */
  FB_CNT_BIRDVIEW_MATCHES,    /* nmatchesBird = BirdviewMatch(...)                   Tracking.cc:2728                   */
  FB_CNT_BIRD_INLIERS,        /* inlier of FilterBirdOutlierInFront                  Tracking.cc:1888                   */
  FB_CNT_BIRD_NEW,            /* mnBirdLastFMatches = buildNew                       Tracking.cc:1913                   */
  FB_CNT_TO_MATCH,            /* nToMatch of SearchLocalPoints                       Tracking.cc:1973-1982              */
  FB_CNT_LOCAL_MATCHES,       /* SearchByProjection(cur, mvpLocalMapPoints, th)      Tracking.cc:1995                   */
  FB_CNT_POSE2_INLIERS,       /* PoseOptimizationWithBird                            Tracking.cc:1400                   */
  FB_CNT_MATCHES_INLIERS,     /* mnMatchesInliers                                    Tracking.cc:1411-1424              */
  FB_CNT_BOW_MATCHES,         /* nmatches = SearchByBoW(mpReferenceKF, cur, ...)     Tracking.cc:1207                   */
  FB_CNT_BIRD_POINTS,         /* numPt = GetBirdMapPointsNum()                       Tracking.cc:1196                   */
  FB_CNT_PROJ_RETRIED,        /* 1 = SearchByProjection ran again with 2 * th        Tracking.cc:1342-1349              */
  FB_CNT_BIRD_POINTS_FINAL,   /* GetBirdMapPointsNum() after the last GetPerFrameMatchedBirdPoints of the frame
                                 (numPt of BirdNeedKF, Tracking.cc:2067)                                                */
  FB_CNT_COUNT = 16
};

int fb_frame_create(const fb_frame_params *params, fb_frame **out);
void fb_frame_destroy(fb_frame *f);

/* Frame::Frame(imGray, BirdGray, ..., birdviewmask, ..., birdviewContourICP, ...) (Frame.cc:262-379) on images in HBM:
 * ExtractORB (front) -> UndistortKeyPoints -> bird extraction (ORBextractor, the E9 substitution) with the detect mask
 * -> GuidenceKeyBirdPts on d_contour -> mvKeysBirdCamXYZ -> mvpMapPoints = NULL, mvbOutlier = false, mvpMapPointsBird =
 * NULL, mvBirdOutlier = TRUE (:355-356) -> AssignFeaturesToGrid.  Image b starts b * image_stride bytes after the base;
 * d_contour / d_mask have the bird image's geometry (rows of bird_stride bytes) and may be NULL (no filter / no mask).
 * The bird chain runs on a stream the handle owns, forked from and joined back into `stream` inside the call.           */
int fb_frame_extract_dev(fb_frame *f, fb_orb *front_extractor, fb_orb *bird_extractor, const uint8_t *d_front,
                         int front_stride, size_t front_image_stride, const uint8_t *d_bird, int bird_stride,
                         size_t bird_image_stride, const uint8_t *d_contour, const uint8_t *d_mask, void *stream);
/* The same from HOST images (the reference's call): the images are copied through page-locked staging buffers the handle
 * owns, asynchronously on `stream`; the call returns when the copies out of the caller's buffers are done, not when the
 * kernels are (fb_frame_counts / fb_frame_download synchronise).  Images are tightly packed batch after batch.           */
int fb_frame_extract(fb_frame *f, fb_orb *front_extractor, fb_orb *bird_extractor, const uint8_t *front, int front_stride,
                     const uint8_t *bird, int bird_stride, const uint8_t *contour, const uint8_t *mask, void *stream);

/* Frame::SetPose(Tcw) + UpdatePoseMatrices (Frame.cc:421-433): d_Tcw [batch][12] device.                                */
int fb_frame_set_pose_dev(fb_frame *f, const float *d_Tcw, void *stream);
/* mCurrentFrame.SetPose(detlaT * mLastFrame.mTcw) (Tracking.cc:1314-1320): d_delta [batch][12] = rows 0..2 of detlaT.   */
int fb_frame_predict_pose_dev(fb_frame *cur, const fb_frame *last, const float *d_delta, void *stream);
/* fill(mvpMapPoints, NULL) (Tracking.cc:1330,1344)                                                                       */
int fb_frame_clear_map_points_dev(fb_frame *f, void *stream);
/* A frame that enters the chain without having been tracked (the first one after initialisation): sets mvpMapPoints /
 * mvpMapPointsBird from host-built index arrays [batch][fb_orb_capacity] (device), mvbOutlier = false.                  */
int fb_frame_set_map_points_dev(fb_frame *f, const int32_t *d_map_point, const int32_t *d_map_point_bird, void *stream);

/* BirdMapPointMatch(CurF, vlocalMPB, windowSize, filterSize) (ORBmatcher.cc:1763-1902, Tracking.cc:1999-2012; M9).
 * d_local [batch][params.local_mpb_cap] lists vlocalMPB as indices into `mpb`, d_n_local [batch] their counts; NULL
 * lists = the whole table in index order.  Tracking.cc:2004: the call is skipped (count 0) unless the list has > 10
 * entries.                                                                                                              */
int fb_frame_bird_mappoint_match_dev(fb_frame *cur, const fb_map_points_bird *mpb, const int32_t *d_local,
                                     const int32_t *d_n_local, int window_size, float filter_size,
                                     const fb_matcher_params *matcher, void *stream);
/* SearchByProjection(CurrentFrame, LastFrame, th, bMono = true) (ORBmatcher.cc:1329-1471; M3)                           */
int fb_frame_search_by_projection_dev(fb_frame *cur, const fb_frame *last, const fb_map_points *map, float th,
                                      const fb_matcher_params *matcher, void *stream);
/* PoseOptimization / PoseOptimizationWithBird / BirdOptimization(pFrame) (Optimizer.cc:246-835), mode FB_POSE_*.
 * which = 0 writes FB_CNT_POSE1_INLIERS, 1 writes FB_CNT_POSE2_INLIERS.                                                 */
int fb_frame_pose_optimization_dev(fb_frame *f, const fb_map_points *map, const fb_map_points_bird *mpb, int mode,
                                   float wB, float wF, int which, void *stream);
/* "Discard outliers" of TrackWithMotionModel (Tracking.cc:1358-1376)                                                     */
int fb_frame_discard_outliers_dev(fb_frame *f, const fb_map_points *map, void *stream);
/* GetPerFrameMatchedBirdPoints (Tracking.cc:2724-2733): BirdviewMatch(cur, ref keys / descriptors, ..., 0, window)
 * (M8) followed by the whole of FilterBirdOutlierInFront(ref, cur, matches, filter_size) (:1825-1914), bookkeeping
 * included: cur.mvBirdOutlier[train] = false, cur.mvpMapPointsBird[train] = ref's point or a NEW MapPointBird (appended
 * to `mpb` with position ptwC and cur's descriptor) that ref.mvpMapPointsBird[query] receives as well.                  */
int fb_frame_match_bird_points_dev(fb_frame *cur, fb_frame *ref, fb_map_points_bird *mpb, int window_size,
                                   float filter_size, const fb_matcher_params *matcher, void *stream);
/* SearchLocalPoints (Tracking.cc:1947-1997): points the frame already holds are marked seen, isInFrustum(pMP, 0.5) on the
 * others of mvpLocalMapPoints (d_local: indices into `map`, [batch][params.local_mp_cap]; NULL = the whole table), then
 * SearchByProjection(cur, mvpLocalMapPoints, th) (ORBmatcher.cc:46-130; M2) if anything is to be matched.               */
int fb_frame_search_local_points_dev(fb_frame *f, const fb_map_points *map, const int32_t *d_local,
                                     const int32_t *d_n_local, float th, const fb_matcher_params *matcher, void *stream);
/* End of Tracking::Track for a tracked frame (Tracking.cc:1411-1424 mnMatchesInliers; :690-701 clean VO matches;
 * :721-725 mvpMapPoints[i] = NULL for outliers) -- after it the frame is what mLastFrame = Frame(mCurrentFrame) copies. */
int fb_frame_finish_dev(fb_frame *f, const fb_map_points *map, void *stream);
/* (both end-of-Track entry points follow `if (bOK)` of Tracking.cc:681 per sequence: the clean-up runs where
 * FB_CNT_MATCHES_INLIERS >= 30, Tracking.cc:1438; a sequence below keeps its members, as a LOST frame does)
 * Tracking.cc:721-725 on its own, after a key-frame decision (fb_track_args.defer_outlier_drop). */
int fb_frame_drop_outliers_dev(fb_frame *f, void *stream);

/* The OK-state path of Tracking::Track in one call: TrackWithMotionModel (Tracking.cc:1312-1385) + TrackLocalMap
 * (:1387-1441) + the end-of-Track clean-up, i.e. predict_pose, M9, M3 (th = 15), PoseOptimizationWithBird, discard,
 * M8 + filter (window 10, 0.05 m), SearchLocalPoints (th = 1, nnratio 0.8), PoseOptimizationWithBird, finish --
 * per sequence as the reference does: the 2 * th retry of :1342-1349 runs inside the matcher launch, and a sequence that
 * still has fewer than 20 matches "returns false" (:1351): its matches stay committed, pose and flags are left alone.   */
typedef struct fb_track_args {
  fb_map_points map;
  fb_map_points_bird mpb;
  const float *d_delta;             /* [batch][12] detlaT                                                               */
  const int32_t *d_local_mp, *d_n_local_mp;   /* mvpLocalMapPoints (NULL = whole table)                                 */
  const int32_t *d_local_mpb, *d_n_local_mpb; /* vlocalMPB (NULL = whole table)                                         */
  float wB, wF;                     /* Optimizer.h:52 defaults 1, 1                                                     */
  /* fb_frame_track_local_map_dev: 1 = per sequence `if (bOK) bOK = TrackLocalMap()` (Tracking.cc:642): a sequence whose first
   * stage returned false (FB_CNT_MATCHES_MAP < 10; the counter is still 0 after an early return) is left alone -- no bird
   * points created, no matching, pose and flags untouched, counters of the stage 0.  0 = every sequence of the handle runs
   * (the host decided for all of them).  fb_frame_track_dev always gates.                                               */
  int32_t gate_local_map;
  /* 1 = leave the final "mvpMapPoints[i] = NULL for outliers" (Tracking.cc:721-725) to fb_frame_drop_outliers_dev: the
   * reference creates its key frame in between (:716-718) and lets the outliers pass to it.                              */
  int32_t defer_outlier_drop;
  /* TrackLocalMap's success threshold on mnMatchesInliers: 0 = 30 (Tracking.cc:1438); a host that relocalised less than
   * mMaxFrames frames ago passes 50 (:1435-1436).                                                                        */
  int32_t min_inliers;
} fb_track_args;
int fb_frame_track_dev(fb_frame *cur, fb_frame *last, const fb_track_args *args, void *stream);
/* The two halves of fb_frame_track_dev on their own, for a host that reads the counters in between (Tracking.cc:529-540:
 * bOK = TrackWithMotionModel(); if (!bOK) bOK = TrackReferenceKeyFrame(); ... if (bOK) bOK = TrackLocalMap()):
 *   fb_frame_track_motion_model_dev  Tracking::TrackWithMotionModel (Tracking.cc:1312-1385); bOK = FB_CNT_MATCHES_MAP >= 10
 *   fb_frame_track_local_map_dev     Tracking::TrackLocalMap (:1387-1441, ref = tmpRefFrame) + the end of Track (:690-725)  */
int fb_frame_track_motion_model_dev(fb_frame *cur, fb_frame *last, const fb_track_args *args, void *stream);
int fb_frame_track_local_map_dev(fb_frame *cur, fb_frame *ref, const fb_track_args *args, void *stream);

/* Tracking::TrackUsingBird (Tracking.cc:2014-2061: the frame of a LOST tracker that could not re-initialise, bHaveBird):
 * SetPose(detlaT * src pose) with src = mpReferenceKF's handle (IsbirdWithRefKF == 1) or tmpRefFrame, GetLocalMapForBird,
 * GetPerFrameMatchedBirdPoints against `ref` first for the sequences with numPt <= 10 (FB_CNT_BIRD_POINTS),
 * Optimizer::BirdOptimization(&frame, 1.0) (bird edges only; inliers -> FB_CNT_POSE1_INLIERS), GetPerFrameMatchedBirdPoints. */
int fb_frame_track_using_bird_dev(fb_frame *cur, fb_frame *src, fb_frame *ref, const fb_track_args *args, void *stream);

/* Frame(const Frame &) (Frame.cc:49-82: tmpRefFrame = new Frame(mCurrentFrame), Tracking.cc:746) and the member copies of
 * KeyFrame::KeyFrame(Frame &F, ...) (KeyFrame.cc:32-91): every member array of src into dst (same parameters), device to
 * device on the stream, including the BoW when src has it.                                                               */
int fb_frame_copy_dev(fb_frame *dst, const fb_frame *src, void *stream);

/* Frame::ComputeBoW (Frame.cc:628-635; KeyFrame::ComputeBoW, KeyFrame.cc:93-102) on the handle: transform(mDescriptors,
 * mBowVec, mFeatVec, 4) into buffers the handle owns; a second call on the same frame is a no-op (if (mBowVec.empty())),
 * fb_frame_extract* empties them again.  voc = device arrays.  fb_frame_bow_view_dev hands out the device pointers
 * (fb_bow_transform_args layout, f_stride = kp_stride).                                                                  */
int fb_frame_compute_bow_dev(fb_frame *f, const fb_vocabulary *voc, void *stream);
int fb_frame_bow_view_dev(fb_frame *f, fb_bow_transform_args *view);
/* ORBmatcher::SearchByBoW(pKF, F, vpMapPointMatches) (ORBmatcher.cc:160-289) with the key frame given as the frame handle
 * it was made from (KeyFrame.cc:32-91 copies mvKeysUn, mDescriptors, mvpMapPoints): count -> FB_CNT_BOW_MATCHES, then
 * mCurrentFrame.mvpMapPoints = vpMapPointMatches (Tracking.cc:1215) for the sequences with count >= min_matches (the
 * others keep their mvpMapPoints, Tracking.cc:1212-1213).  Both frames need their BoW (FB_ERR_ARG otherwise).             */
int fb_frame_search_by_bow_dev(fb_frame *cur, const fb_frame *kf, const fb_map_points *map, const fb_matcher_params *matcher,
                               int min_matches, void *stream);
/* Tracking::TrackReferenceKeyFrame (Tracking.cc:1180-1244, bLooseCouple = true, bHaveBird): SetPose(detlaT * kf pose) with
 * args->d_delta = detlaT of :1185, GetLocalMapForBird, GetPerFrameMatchedBirdPoints against `ref` (tmpRefFrame) for the
 * sequences holding fewer than 10 bird points (count -> FB_CNT_BIRD_POINTS; FB_CNT_BIRDVIEW_MATCHES reads 0 where the
 * step was skipped), ComputeBoW, SearchByBoW(0.7), and for the sequences with >= 15 matches: mvpMapPoints = matches,
 * PoseOptimizationWithBird (FB_CNT_POSE1_INLIERS), the outlier discard (FB_CNT_MATCHES, FB_CNT_MATCHES_MAP).  The
 * reference's return value per sequence = FB_CNT_BOW_MATCHES >= 15 && FB_CNT_MATCHES_MAP >= 10; a sequence below 15 keeps
 * its predicted pose, members and the two discard counters, like the early return.                                       */
int fb_frame_track_reference_dev(fb_frame *cur, fb_frame *kf, fb_frame *ref, const fb_vocabulary *voc, const fb_track_args *args,
                                 void *stream);

/* Pointers to a frame's arrays: device pointers from fb_frame_view_dev (valid for the handle's lifetime, for harnesses
 * that keep working on the device), or host buffers the caller hands to fb_frame_download (NULL members are skipped).
 * kp_stride = fb_orb_capacity(params.orb).                                                                             */
typedef struct fb_frame_view {
  int32_t batch, kp_stride;
  int32_t *n;                       /* [batch] N                                                                        */
  fb_keypoint *kps, *kps_un;        /* mvKeys, mvKeysUn                                                                 */
  uint8_t *desc;                    /* mDescriptors                                                                     */
  int32_t *map_point;               /* mvpMapPoints as table index, -1 = NULL                                           */
  uint8_t *outlier;                 /* mvbOutlier                                                                       */
  int32_t *n_bird;                  /* [batch] Nbird                                                                    */
  fb_keypoint *kps_bird;            /* mvKeysBird                                                                       */
  uint8_t *desc_bird;               /* mDescriptorsBird                                                                 */
  float *bird_cam_xyz;              /* mvKeysBirdCamXYZ                                                                 */
  int32_t *map_point_bird;          /* mvpMapPointsBird                                                                 */
  uint8_t *bird_outlier;            /* mvBirdOutlier                                                                    */
  float *Tcw;                       /* [batch][12]                                                                      */
  int32_t *counts;                  /* [FB_CNT_COUNT][batch] (slot-major: every kernel writes one contiguous row)      */
} fb_frame_view;
int fb_frame_view_dev(fb_frame *f, fb_frame_view *out);
int fb_frame_download(fb_frame *f, const fb_frame_view *host, void *stream); /* synchronises `stream`                  */
int fb_frame_counts(fb_frame *f, int32_t *host_counts /* [FB_CNT_COUNT][batch] */, float *host_Tcw /* [batch][12] or NULL */,
                    void *stream);                                              /* synchronises `stream`                 */

/* ======================================================================== */
/* Optimizer::LocalBundleAdjustment / LocalBundleAdjustmentWithOdom          */
/* (src/Optimizer.cc:838-1165, 2137-2670)                                    */
/* ======================================================================== */
typedef struct fb_local_ba_args {
  int32_t with_odom;   /* 0: LocalBundleAdjustment (SE3Expmap edges), 1: ...WithOdom (Quat edges + bird + odom) */
  float fx, fy, cx, cy;
  float wF, wB, wP;
  int32_t n_kf;               /* local + fixed keyframes, graph insertion order */
  float *kf_Tcw;              /* in/out [n_kf][12]                              */
  const uint8_t *kf_fixed;    /* [n_kf]                                         */
  int32_t n_mp;
  float *mp_xw;               /* in/out [n_mp][3]                               */
  int32_t n_mpb;
  float *mpb_xw;              /* in/out [n_mpb][3]                              */
  /* front observations, grouped by map point in insertion order */
  int32_t n_obs;
  const int32_t *obs_kf;      /* index into kf                                  */
  const int32_t *obs_mp;      /* index into mp                                  */
  const float *obs_uv;        /* [n_obs][2]                                     */
  const float *obs_inv_sigma2;
  /* bird observations */
  int32_t n_bobs;
  const int32_t *bobs_kf;
  const int32_t *bobs_mpb;
  const float *bobs_xc;       /* [n_bobs][3]                                    */
  const float *bobs_inv_sigma2;
  /* odometry edges (Optimizer.cc:2419-2495) */
  int32_t n_odom;
  const int32_t *odom_kf_i;
  const int32_t *odom_kf_j;
  const float *odom_Tij;      /* [n_odom][12]                                   */
  const double *odom_info;    /* information scale per edge (1e4*wP, 2e3, 1e3*wP) */
  const volatile uint8_t *stop_flag; /* host-visible pbStopFlag, may be NULL    */
  uint8_t *obs_outlier;       /* out [n_obs]                                    */
  uint8_t *bobs_outlier;      /* out [n_bobs]                                   */
} fb_local_ba_args;
int fb_local_ba(const fb_local_ba_args *args); /* host pointers */
/* The same with the graph resident in HBM (the LocalMapping-side chain hands its results over on the device: triangulation
 * matches -> new points -> fuse -> local BA, LocalMapping.cc:62-99): kf_Tcw, mp_xw, mpb_xw, every obs_* / bobs_* array and the
 * two outlier arrays are DEVICE pointers; kf_fixed, the odometry edges (odom_*: at most 3 n_kf of them) and stop_flag stay
 * HOST pointers (the host chooses the window, Optimizer.cc:2139-2227, 2419-2495).  The edge records, the CSR by landmark / by
 * key frame and the duplicate check are built by kernels on `stream`, ordered behind the producers of the arrays; results are
 * written back on `stream`.  The call returns when the optimisation has finished (the schedule's length is data dependent:
 * the host watches the device-resident Levenberg-Marquardt loop), like the reference's blocking call.
 * Capacity: <= 23 free key frames, <= 32768 observations per key frame (larger graphs: fb_local_ba).                       */
int fb_local_ba_dev(const fb_local_ba_args *args, void *stream);

/* Optimizer::BundleAdjustmentWithOdom / GlobalBundleAdjustemntWithOdom (Optimizer.cc:1778-2135) on the same graph
 * description: with_odom = 1 (Quat edges), n_odom = 0 (the pose-graph block is commented out in the reference,
 * :2004-2037), kf_fixed[k] = (mnId == 0).  ONE optimize(nIterations) (:2048-2050), Huber delta sqrt(5.99) on every edge
 * iff bRobust (:1836,1891-1895,1980-1984), no chi2 classification: obs_outlier / bobs_outlier are not written.
 * Capacity (local and global BA alike): up to 23 free key frames the Schur-reduced pose system is LDS resident
 * (MFMA Schur kernel + one-workgroup LDL^T); beyond that it lives in HBM (block scatter + blocked multi-kernel LDL^T),
 * up to 682 free key frames.                                                                                        */
int fb_global_ba(const fb_local_ba_args *args, int n_iterations, int robust);

/* One local BA over `world` GPUs (one process per GPU), SURVEY 8(e): rank r owns the landmarks l with
 * l % world == r and all their edges, odometry edges live on rank 0, keyframes are replicated.  Every rank
 * passes the SAME complete problem and receives the complete result.  The library calls `allreduce` (the
 * only callback of this ABI) to combine `n` doubles in host memory in place over the ranks: op 0 = sum,
 * 1 = max.  Two calls per LM trial: the Schur-reduced pose system, then [Hpp, bp, chi2, scale].       */
typedef int (*fb_allreduce_fn)(void *ctx, double *buf, int32_t n, int32_t op);
int fb_local_ba_sharded(const fb_local_ba_args *args, int rank, int world, fb_allreduce_fn allreduce, void *ctx);

/* The same with RCCL inside the library: the two exchanges of an LM trial are ncclAllReduce calls on device buffers,
 * enqueued on the BA's stream between its kernels (no host staging, no read-back: the Levenberg-Marquardt state machine
 * runs on the device and takes identical decisions on every rank from the reduced values; pbStopFlag and a rank's
 * argument errors travel inside the reduced buffers so that every branch is collective).  `comm` is an ncclComm_t over
 * the `world` ranks.  RCCL is resolved at run time (dlopen of librccl.so.1: torch's when the host is Python), so the
 * library has no link-time dependency on it; fb_rccl_* wrap ncclGetUniqueId / ncclCommInitRank / ncclCommDestroy of that
 * same RCCL for hosts that do not own a communicator yet (rank 0 makes the id, the host broadcasts its 128 bytes).  */
typedef struct fb_rccl_unique_id { char internal[128]; } fb_rccl_unique_id;
int fb_rccl_get_unique_id(fb_rccl_unique_id *id);
int fb_rccl_comm_init(const fb_rccl_unique_id *id, int rank, int world, void **comm);
int fb_rccl_comm_destroy(void *comm);
/* ncclCommCount / ncclCommUserRank of a communicator: what RCCL itself says about the ranks it connects (bench.py prints it
 * as local_ba.rccl_ranks_seen, so that a multi-GPU run shows that the collective really spans the ranks). */
int fb_rccl_comm_info(void *comm, int *count, int *rank);
int fb_local_ba_sharded_rccl(const fb_local_ba_args *args, int rank, int world, void *comm);

#ifdef __cplusplus
}
#endif
#endif /* FISHBIRD_H_ */
