// host_abi_bench.cpp -- what ONE frame costs a maintainer who binds the reference to the C-ABI (INTEGRATION.md sections 1-3).
//
// Plain g++ program over include/fishbird.h (no torch, no Python), spawned by bench.py as a child process.  For one
// 1280x720 front + 512x512 bird pair per call it times, with std::chrono::steady_clock as the reference's own
// mono_encoder.cc:160-183 does,
//   (a) the HOST-POINTER entry points the shims of INTEGRATION.md call synchronously, one by one:
//         fb_orb_extract x2 (ORBextractor::operator(), ORBextractor.cc:1043), the host grid build
//         (Frame::AssignFeaturesToGrid, stays C++ in the shim), fb_match_projection_frame (ORBmatcher.cc:1329),
//         fb_match_bird_mappoints (ORBmatcher.cc:1763), fb_pose_opt (Optimizer.cc:478);
//   (b) the same frame through the device-resident Frame handle: fb_frame_extract (host images, pinned staging) +
//       fb_frame_track_dev (the whole TrackWithMotionModel + TrackLocalMap chain) + fb_frame_counts (one sync).
// The workload is the frame matched against itself (last frame = current key points back-projected at 5 m, identity
// motion): every matcher has ~2000 queries with a true match.  Output: one JSON object on stdout.
#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#define __HIP_PLATFORM_AMD__ 1
#include <hip/hip_runtime_api.h>  // the host owns its map tables on the device: plain hipMalloc / hipMemcpy

#include "../../include/fishbird.h"
#include "../../fishbirdeyevisualslam_amd/host/fishbird_host.hpp"  // part (b) goes through the C++ mirror (DeviceFrame)

static double now_ms() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
#define CK(expr) do { int rc_ = (expr); if (rc_ != 0) { std::fprintf(stderr, "%s failed (%d): %s\n", #expr, rc_, fb_last_error()); return 1; } } while (0)

static std::vector<uint8_t> read_raw(const char *path, size_t n) {
  std::vector<uint8_t> v(n);
  FILE *f = std::fopen(path, "rb");
  if (!f || std::fread(v.data(), 1, n, f) != n) { std::fprintf(stderr, "cannot read %s\n", path); std::exit(3); }
  std::fclose(f);
  return v;
}

// Frame::AssignFeaturesToGrid (Frame.cc:381-411) as the shim keeps it: on the host
static void assign_to_grid(const fb_keypoint *k, int n, const fb_grid_geom &g, std::vector<int32_t> &start, std::vector<int32_t> &items, int cap) {
  const int ncell = g.cols * g.rows;
  std::vector<int32_t> cell(n), cnt(ncell + 1, 0);
  for (int i = 0; i < n; i++) {
    const int px = (int)std::round((k[i].x - g.min_x) * g.inv_w), py = (int)std::round((k[i].y - g.min_y) * g.inv_h);
    cell[i] = (px < 0 || px >= g.cols || py < 0 || py >= g.rows) ? -1 : px * g.rows + py;
    if (cell[i] >= 0) cnt[cell[i] + 1]++;
  }
  for (int c = 0; c < ncell; c++) cnt[c + 1] += cnt[c];
  start.assign(cnt.begin(), cnt.end());
  items.assign(cap, 0);
  std::vector<int32_t> fillp(cnt.begin(), cnt.end() - 1);
  for (int i = 0; i < n; i++) if (cell[i] >= 0) items[fillp[cell[i]]++] = i;
}

static double median(std::vector<double> v) { std::sort(v.begin(), v.end()); return v[v.size() / 2]; }

int main(int argc, char **argv) {
  if (argc < 3) { std::fprintf(stderr, "usage: host_abi_bench front.raw bird.raw [iterations]\n"); return 2; }
  const int FW = 1280, FH = 720, BW = 512, BH = 512;
  const int iters = argc > 3 ? std::atoi(argv[3]) : 30, warm = 5;
  std::vector<uint8_t> front = read_raw(argv[1], (size_t)FW * FH), bird = read_raw(argv[2], (size_t)BW * BH);
  fb_orb_params op = {2000, 1.2f, 8, 15, 5};
  const int cap = fb_orb_capacity(&op);
  fb_orb *of = nullptr, *ob = nullptr;
  CK(fb_orb_create(&op, &of));
  CK(fb_orb_create(&op, &ob));
  fb_orb_tables tab;
  CK(fb_orb_get_tables(of, &tab));
  const float fx = 500.f, fy = 500.f, cx = FW / 2.f, cy = FH / 2.f;
  const float Tbc[12] = {0.013296164f, -0.57126045f, 0.82066119f, 3.747f, -0.99991107f, -0.0067514977f, 0.011500627f, 0.04f,
                         -0.0010291613f, -0.82074112f, -0.57129937f, 0.736f};
  float Tcb[12];
  for (int r = 0; r < 3; r++) {
    for (int c = 0; c < 3; c++) Tcb[r * 4 + c] = Tbc[c * 4 + r];
    Tcb[r * 4 + 3] = -(Tbc[0 * 4 + r] * Tbc[3] + Tbc[1 * 4 + r] * Tbc[7] + Tbc[2 * 4 + r] * Tbc[11]);
  }
  const fb_grid_geom gF = {0.f, 0.f, 64.f / FW, 48.f / FH, 64, 48}, gB = {0.f, 0.f, 32.f / BW, 32.f / BH, 32, 32};

  std::vector<fb_keypoint> fk(cap), bk(cap);
  std::vector<uint8_t> fd((size_t)cap * 32), bd((size_t)cap * 32);
  int32_t nf = 0, nb = 0;
  std::vector<int32_t> fcs, fci, bcs, bci;
  // world: every key point of the frame itself, back-projected at 5 m (front) / its camera XYZ (bird), identity pose
  std::vector<float> lxw, lang, bxw, bcam;
  std::vector<uint8_t> lvalid, lobs, bvalid;
  std::vector<int32_t> loct;
  const float Tcw0[12] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0};
  std::vector<int32_t> m3(cap), m9(cap);
  int32_t nm3 = 0, nm9 = 0, ninl = 0;
  std::vector<float> e_fxw((size_t)cap * 3), e_fobs((size_t)cap * 2), e_finf(cap), e_bxw((size_t)cap * 3), e_bxc((size_t)cap * 3), e_binf(cap);
  std::vector<uint8_t> e_fv(cap), e_bv(cap), e_fout(cap), e_bout(cap);

  std::vector<double> t_ef, t_eb, t_grid, t_m3, t_m9, t_gather, t_pose, t_frame;
  for (int it = 0; it < warm + iters; it++) {
    const double a0 = now_ms();
    CK(fb_orb_extract(of, front.data(), FW, FH, FW, fk.data(), fd.data(), &nf));
    const double a1 = now_ms();
    CK(fb_orb_extract(ob, bird.data(), BW, BH, BW, bk.data(), bd.data(), &nb));
    const double a2 = now_ms();
    assign_to_grid(fk.data(), nf, gF, fcs, fci, cap);
    assign_to_grid(bk.data(), nb, gB, bcs, bci, cap);
    bcam.assign((size_t)cap * 3, 0.f);
    for (int i = 0; i < nb; i++) {  // Frame.cc:365-373 (stays host code in the shim)
      const float p0 = (float)((BH / 2 - bk[i].y) * 0.03984 + 1.393), p1 = (float)((BW / 2 - bk[i].x) * 0.03984);
      for (int r = 0; r < 3; r++) bcam[(size_t)i * 3 + r] = (Tcb[r * 4] * p0 + Tcb[r * 4 + 1] * p1) + Tcb[r * 4 + 3];
    }
    const double a3 = now_ms();
    if (it == 0) {
      lxw.assign((size_t)cap * 3, 0.f); lang.assign(cap, 0.f); lvalid.assign(cap, 0); lobs.assign(cap, 1); loct.assign(cap, 0);
      for (int i = 0; i < nf; i++) {
        lvalid[i] = 1; loct[i] = fk[i].octave; lang[i] = fk[i].angle;
        lxw[(size_t)i * 3] = (fk[i].x - cx) / fx * 5.f; lxw[(size_t)i * 3 + 1] = (fk[i].y - cy) / fy * 5.f; lxw[(size_t)i * 3 + 2] = 5.f;
      }
      bxw = bcam; bvalid.assign(cap, 0);
      for (int i = 0; i < nb && i < 1000; i++) bvalid[i] = 1;  // 1000 reference bird points (BASELINE configs[2])
    }
    fb_proj_frame_args A;
    std::memset(&A, 0, sizeof(A));
    A.batch = 1; A.cur_stride = cap; A.last_stride = cap; A.n_cur = &nf; A.cur_kps = fk.data(); A.cur_desc = fd.data();
    A.cur_cell_start = fcs.data(); A.cur_cell_items = fci.data(); A.cur_Tcw = Tcw0; A.n_last = &nf; A.last_valid = lvalid.data();
    A.last_obs_pos = lobs.data(); A.last_xw = lxw.data(); A.last_desc = fd.data(); A.last_octave = loct.data(); A.last_angle = lang.data();
    A.cam = {fx, fy, cx, cy, 0.f, 0.f, (float)FW, (float)FH}; A.grid = gF;
    for (int i = 0; i < FB_MAX_LEVELS; i++) A.scale_factors[i] = tab.scale_factor[i];
    A.th = 15.f; A.matcher = {0.9f, 1}; A.match_cur_to_last = m3.data(); A.nmatches = &nm3;
    CK(fb_match_projection_frame(&A));
    const double a4 = now_ms();
    std::fill(m9.begin(), m9.end(), -1);
    fb_bird_mp_args M;
    std::memset(&M, 0, sizeof(M));
    int32_t nref = std::min<int32_t>(nb, 1000);
    M.batch = 1; M.cur_stride = cap; M.ref_stride = cap; M.n_cur = &nb; M.cur_kps = bk.data(); M.cur_desc = bd.data(); M.cur_cam_xyz = bcam.data();
    M.cur_cell_start = bcs.data(); M.cur_cell_items = bci.data(); M.cur_Tcw = Tcw0; M.n_ref = &nref; M.ref_valid = bvalid.data();
    M.ref_xw = bxw.data(); M.ref_desc = bd.data(); std::memcpy(M.Tbc, Tbc, sizeof(Tbc)); M.bird_cols = BW; M.bird_rows = BH;
    M.meter2pixel = 25.1; M.rear_axle_to_center = 1.393; M.grid = gB; M.window_size = 10; M.filter_size = 0.05f; M.matcher = {0.9f, 1};
    M.match_cur_to_ref = m9.data(); M.ninliers = &nm9;
    CK(fb_match_bird_mappoints(&M));
    const double a5 = now_ms();
    // edge construction of PoseOptimizationWithBird (Optimizer.cc:525-602): host loops in the shim
    for (int i = 0; i < cap; i++) {
      e_fv[i] = i < nf && m3[i] >= 0;
      if (e_fv[i]) {
        std::memcpy(&e_fxw[(size_t)i * 3], &lxw[(size_t)m3[i] * 3], 12);
        e_fobs[(size_t)i * 2] = fk[i].x; e_fobs[(size_t)i * 2 + 1] = fk[i].y; e_finf[i] = tab.inv_level_sigma2[fk[i].octave];
      }
      e_bv[i] = i < nb && m9[i] >= 0;
      if (e_bv[i]) {
        std::memcpy(&e_bxw[(size_t)i * 3], &bxw[(size_t)m9[i] * 3], 12);
        std::memcpy(&e_bxc[(size_t)i * 3], &bcam[(size_t)i * 3], 12); e_binf[i] = tab.inv_level_sigma2[bk[i].octave];
      }
    }
    std::fill(e_bout.begin(), e_bout.end(), 1);
    float Tcw[12];
    std::memcpy(Tcw, Tcw0, sizeof(Tcw));
    Tcw[3] = 0.01f;
    const double a6 = now_ms();
    fb_pose_opt_args P;
    std::memset(&P, 0, sizeof(P));
    P.batch = 1; P.mode = FB_POSE_FRONT_BIRD; P.front_stride = cap; P.bird_stride = cap; P.fx = fx; P.fy = fy; P.cx = cx; P.cy = cy; P.wF = 1.f; P.wB = 1.f;
    P.n_front = &nf; P.front_xw = e_fxw.data(); P.front_obs = e_fobs.data(); P.front_inv_sigma2 = e_finf.data(); P.front_valid = e_fv.data();
    P.n_bird = &nb; P.bird_xw = e_bxw.data(); P.bird_xc = e_bxc.data(); P.bird_inv_sigma2 = e_binf.data(); P.bird_valid = e_bv.data();
    P.bird_outlier = e_bout.data(); P.Tcw = Tcw; P.front_outlier = e_fout.data(); P.ninliers = &ninl;
    CK(fb_pose_opt(&P));
    const double a7 = now_ms();
    if (it >= warm) {
      t_ef.push_back(a1 - a0); t_eb.push_back(a2 - a1); t_grid.push_back(a3 - a2); t_m3.push_back(a4 - a3); t_m9.push_back(a5 - a4);
      t_gather.push_back(a6 - a5); t_pose.push_back(a7 - a6); t_frame.push_back(a7 - a0);
    }
  }

  // ---- (b) the device-resident Frame handle -------------------------------------------------------------------------
  fb_frame_params fp;
  std::memset(&fp, 0, sizeof(fp));
  fp.batch = 1; fp.front_width = FW; fp.front_height = FH; fp.bird_width = BW; fp.bird_height = BH; fp.orb = op;
  fp.K[0] = fx; fp.K[1] = fy; fp.K[2] = cx; fp.K[3] = cy;
  std::memcpy(fp.Tbc, Tbc, sizeof(Tbc)); std::memcpy(fp.Tcb, Tcb, sizeof(Tcb));
  fp.pixel2meter = 0.03984; fp.meter2pixel = 25.1; fp.rear_axle_to_center = 1.393;
  fp.map_cap = cap; fp.local_mp_cap = cap; fp.local_mpb_cap = 2 * cap;
  fishbird::ORBextractor exFront(2000, 1.2f, 8, 15, 5), exBird(2000, 1.2f, 8, 15, 5);
  fishbird::DeviceFrame frameA(fp), frameB(fp);
  fishbird::DeviceFrame *fr[2] = {&frameA, &frameB};
  // map tables on the device: the C-ABI carries no allocator, the host uses its HIP runtime
  struct Dev { void *p = nullptr; };
  auto up = [&](Dev &d, const void *h, size_t n) -> int {
    if (hipMalloc(&d.p, n) != hipSuccess) return 1;
    return hipMemcpy(d.p, h, n, hipMemcpyHostToDevice) != hipSuccess;
  };
  std::vector<uint8_t> zeros((size_t)2 * cap * 32, 0), ones((size_t)2 * cap, 1);
  std::vector<float> nrm((size_t)cap * 3), maxd(cap), mind(cap);
  for (int i = 0; i < nf; i++) {
    const float *X = &lxw[(size_t)i * 3];
    const float d = std::sqrt(X[0] * X[0] + X[1] * X[1] + X[2] * X[2]);
    for (int k = 0; k < 3; k++) nrm[(size_t)i * 3 + k] = X[k] / d;
    maxd[i] = d * tab.scale_factor[fk[i].octave]; mind[i] = maxd[i] / tab.scale_factor[7];
  }
  Dev d_n, d_bad, d_obs, d_xw, d_nrm, d_max, d_min, d_desc, d_bn, d_bxw, d_bdesc, d_delta, d_mp, d_mpb, d_T;
  int32_t nmap = nf, nbird = std::min<int32_t>(nb, 1000);
  std::vector<float> bxw2((size_t)2 * cap * 3, 0.f);
  std::memcpy(bxw2.data(), bxw.data(), (size_t)cap * 12);
  std::vector<uint8_t> bdesc2((size_t)2 * cap * 32, 0);
  std::memcpy(bdesc2.data(), bd.data(), (size_t)cap * 32);
  std::vector<int32_t> mp0(cap, -1), mpb0(cap, -1);
  for (int i = 0; i < nf; i++) mp0[i] = (i % 10) < 7 ? i : -1;          // the frame holds 70 % of its points; the rest is found by SearchLocalPoints
  for (int i = 0; i < nbird; i++) mpb0[i] = (i % 10) < 6 ? i : -1;
  if (up(d_n, &nmap, 4) || up(d_bad, zeros.data(), cap) || up(d_obs, ones.data(), cap) || up(d_xw, lxw.data(), (size_t)cap * 12) ||
      up(d_nrm, nrm.data(), (size_t)cap * 12) || up(d_max, maxd.data(), (size_t)cap * 4) || up(d_min, mind.data(), (size_t)cap * 4) ||
      up(d_desc, fd.data(), (size_t)cap * 32) || up(d_bn, &nbird, 4) || up(d_bxw, bxw2.data(), bxw2.size() * 4) ||
      up(d_bdesc, bdesc2.data(), bdesc2.size()) || up(d_delta, Tcw0, 48) || up(d_mp, mp0.data(), (size_t)cap * 4) ||
      up(d_mpb, mpb0.data(), (size_t)cap * 4) || up(d_T, Tcw0, 48)) { std::fprintf(stderr, "device allocation failed\n"); return 1; }
  fishbird::DeviceMap dmap;
  dmap.points.stride = cap; dmap.points.n = (int32_t *)d_n.p; dmap.points.bad = (uint8_t *)d_bad.p; dmap.points.obs_pos = (uint8_t *)d_obs.p;
  dmap.points.xw = (float *)d_xw.p; dmap.points.normal = (float *)d_nrm.p; dmap.points.max_dist = (float *)d_max.p;
  dmap.points.min_dist = (float *)d_min.p; dmap.points.desc = (uint8_t *)d_desc.p;
  dmap.birdPoints.stride = 2 * cap; dmap.birdPoints.n = (int32_t *)d_bn.p; dmap.birdPoints.xw = (float *)d_bxw.p; dmap.birdPoints.desc = (uint8_t *)d_bdesc.p;
  std::vector<double> t_h_extract, t_h_track, t_h_frame;
  fishbird::DeviceFrame::TrackResult res;
  try {
    fr[0]->Construct(exFront, exBird, front.data(), FW, bird.data(), BW, nullptr, nullptr);
    CK(fb_frame_set_map_points_dev(fr[0]->handle(), (int32_t *)d_mp.p, (int32_t *)d_mpb.p, nullptr));
    fr[0]->SetPose((float *)d_T.p);
    int cur = 1;
    for (int it = 0; it < warm + iters; it++) {
      const double a0 = now_ms();
      fr[cur]->Construct(exFront, exBird, front.data(), FW, bird.data(), BW, nullptr, nullptr);   // Frame::Frame from host images
      const double a1 = now_ms();
      res = fr[cur]->TrackedFrame(*fr[cur ^ 1], dmap, (float *)d_delta.p);                          // the chain + the one synchronisation
      const double a2 = now_ms();
      if (it >= warm) { t_h_extract.push_back(a1 - a0); t_h_track.push_back(a2 - a1); t_h_frame.push_back(a2 - a0); }
      cur ^= 1;
    }
  } catch (const std::exception &e) {
    std::fprintf(stderr, "frame handle path failed: %s\n", e.what());
    return 1;
  }
  std::vector<int32_t> counts(FB_CNT_COUNT);
  for (int i = 0; i < FB_CNT_COUNT; i++) counts[i] = res.count(i);
  std::printf("{\"iterations\": %d, \"warmup\": %d, \"keypoints_front\": %d, \"keypoints_bird\": %d, \"front_matches\": %d, \"bird_matches\": %d, "
              "\"pose_inliers\": %d, \"host_pointer_ms_median\": {\"fb_orb_extract_front\": %.4f, \"fb_orb_extract_bird\": %.4f, "
              "\"host_grids_and_bird_cam\": %.4f, \"fb_match_projection_frame\": %.4f, \"fb_match_bird_mappoints\": %.4f, "
              "\"host_edge_construction\": %.4f, \"fb_pose_opt\": %.4f, \"frame\": %.4f}, "
              "\"frame_handle_ms_median\": {\"fb_frame_extract_enqueue\": %.4f, \"fb_frame_track_dev_plus_counts\": %.4f, \"frame\": %.4f, "
              "\"chain_counters\": {\"proj_matches\": %d, \"pose1_inliers\": %d, \"local_matches\": %d, \"pose2_inliers\": %d, \"bird_kf_matches\": %d}}}\n",
              iters, warm, nf, nb, nm3, nm9, ninl, median(t_ef), median(t_eb), median(t_grid), median(t_m3), median(t_m9), median(t_gather),
              median(t_pose), median(t_frame), median(t_h_extract), median(t_h_track), median(t_h_frame), counts[FB_CNT_PROJ_MATCHES],
              counts[FB_CNT_POSE1_INLIERS], counts[FB_CNT_LOCAL_MATCHES], counts[FB_CNT_POSE2_INLIERS], counts[FB_CNT_BIRD_KF_MATCHES]);
  fb_orb_destroy(of); fb_orb_destroy(ob);
  return 0;
}
