// Drives Optimizer::LocalBundleAdjustment[WithOdom] / GlobalBundleAdjustemntWithOdom of the C++ host mirror with the
// reference's own signatures (KeyFrame*, bool*, Map*) on a map read from a file, and dumps (a) the flattened graph it
// handed to the C-ABI and (b) the map after the write-back, for the Python test to check against the oracle.
#include <cstdio>
#include <cstdlib>
#include <memory>
#include <string>
#include <vector>

#include "../../fishbirdeyevisualslam_amd/host/fishbird_host.hpp"

using namespace fishbird;

template <typename T> static std::vector<T> rd(FILE *f, size_t n) {
  std::vector<T> v(n);
  if (n && std::fread(v.data(), sizeof(T), n, f) != n) { std::fprintf(stderr, "short read\n"); std::exit(3); }
  return v;
}
template <typename T> static void wr(FILE *f, const std::vector<T> &v) {
  const int64_t n = (int64_t)v.size();
  std::fwrite(&n, 8, 1, f);
  if (n) std::fwrite(v.data(), sizeof(T), v.size(), f);
}

int main(int argc, char **argv) {
  if (argc < 4) { std::fprintf(stderr, "usage: map_ba_test map.bin [graph-]local|odom|global out.bin\n"); return 2; }
  std::string mode = argv[2];
  const bool dumpOnly = mode.rfind("graph-", 0) == 0;  // "graph-local" ...: collect and dump, do not optimise (no GPU)
  if (dumpOnly) mode = mode.substr(6);
  FILE *f = std::fopen(argv[1], "rb");
  if (!f) return 3;
  const std::vector<int32_t> hdr = rd<int32_t>(f, 8);  // n_kf n_mp n_mpb n_obs n_bobs cur_kf n_covis global_iterations
  const int nkf = hdr[0], nmp = hdr[1], nmpb = hdr[2], nobs = hdr[3], nbobs = hdr[4], cur = hdr[5], ncov = hdr[6], gits = hdr[7];
  const std::vector<float> K = rd<float>(f, 4);
  const std::vector<float> sf = rd<float>(f, 8), inv2 = rd<float>(f, 8);
  const std::vector<int32_t> isInit = rd<int32_t>(f, nkf);
  const std::vector<float> Tcw = rd<float>(f, (size_t)nkf * 12);
  const std::vector<double> odo = rd<double>(f, (size_t)nkf * 3);
  const std::vector<float> mp = rd<float>(f, (size_t)nmp * 3), mpb = rd<float>(f, (size_t)nmpb * 3);
  const std::vector<int32_t> okf = rd<int32_t>(f, nobs), omp = rd<int32_t>(f, nobs);
  const std::vector<float> ouv = rd<float>(f, (size_t)nobs * 2);
  const std::vector<int32_t> ooct = rd<int32_t>(f, nobs);
  const std::vector<int32_t> bkf = rd<int32_t>(f, nbobs), bmp = rd<int32_t>(f, nbobs);
  const std::vector<float> bxc = rd<float>(f, (size_t)nbobs * 3);
  const std::vector<int32_t> boct = rd<int32_t>(f, nbobs);
  const std::vector<int32_t> covis = rd<int32_t>(f, ncov);
  std::fclose(f);

  Map map;
  std::vector<std::unique_ptr<KeyFrame>> kfs;
  std::vector<std::unique_ptr<MapPoint>> mps;
  std::vector<std::unique_ptr<MapPointBird>> mpbs;
  for (int k = 0; k < nkf; k++) {
    kfs.emplace_back(new KeyFrame());
    KeyFrame &F = *kfs.back();
    F.mnId = k; F.isInit = isInit[k] != 0;
    F.fx = K[0]; F.fy = K[1]; F.cx = K[2]; F.cy = K[3];
    F.SetPose(&Tcw[12 * (size_t)k]);
    for (int i = 0; i < 3; i++) F.mGtPose[i] = odo[3 * (size_t)k + i];
    F.mnScaleLevels = 8; F.mvScaleFactors = sf; F.mvInvLevelSigma2 = inv2;
    map.mspKeyFrames.push_back(&F);
  }
  for (int j = 0; j < nmp; j++) {
    mps.emplace_back(new MapPoint());
    mps.back()->mnId = j; mps.back()->mpMap = &map;
    mps.back()->SetWorldPos(&mp[3 * (size_t)j]);
    map.mspMapPoints.push_back(mps.back().get());
  }
  for (int j = 0; j < nmpb; j++) {
    mpbs.emplace_back(new MapPointBird());
    mpbs.back()->mnId = j;
    mpbs.back()->SetWorldPos(&mpb[3 * (size_t)j]);
    map.mspMapPointsBird.push_back(mpbs.back().get());
  }
  std::vector<int32_t> obsSlot(nobs), bobsSlot(nbobs);
  for (int e = 0; e < nobs; e++) {
    KeyFrame &F = *kfs[okf[e]];
    fb_keypoint kp{};
    kp.x = ouv[2 * (size_t)e]; kp.y = ouv[2 * (size_t)e + 1]; kp.octave = ooct[e];
    obsSlot[e] = (int32_t)F.mvKeysUn.size();
    F.mvKeysUn.push_back(kp);
    F.mvpMapPoints.push_back(mps[omp[e]].get());
    mps[omp[e]]->AddObservation(&F, obsSlot[e]);
    if (!mps[omp[e]]->mpRefKF) mps[omp[e]]->mpRefKF = &F;
  }
  for (int e = 0; e < nbobs; e++) {
    KeyFrame &F = *kfs[bkf[e]];
    fb_keypoint kp{};
    kp.octave = boct[e];
    bobsSlot[e] = (int32_t)F.mvKeysBird.size();
    F.mvKeysBird.push_back(kp);
    F.mvKeysBirdCamXYZ.insert(F.mvKeysBirdCamXYZ.end(), &bxc[3 * (size_t)e], &bxc[3 * (size_t)e] + 3);
    F.mvpMapPointsBird.push_back(mpbs[bmp[e]].get());
    mpbs[bmp[e]]->AddObservation(&F, bobsSlot[e]);
    if (!mpbs[bmp[e]]->mpRefKF) mpbs[bmp[e]]->mpRefKF = &F;
  }
  for (int i = 0; i < ncov; i++) kfs[cur]->mvpOrderedConnectedKeyFrames.push_back(kfs[covis[i]].get());

  FILE *o = std::fopen(argv[3], "wb");
  try {
    {  // the graph the optimiser will see (same calls as the Optimizer methods make), dumped for the oracle
      BAGraph G;
      if (mode == "global") {
        for (auto &k : kfs) G.addKeyFrame(k.get(), k->mnId == 0);
        G.nLocal = G.kfs.size();
        for (auto &p : mps) if (G.addMapPoint(p.get(), nkf - 1) == 0) G.popMapPoint();
        for (auto &p : mpbs) if (G.addMapPointBird(p.get()) == 0) G.popMapPointBird();
      } else {
        collectLocalGraph(kfs[cur].get(), mode == "odom", G);
        if (mode == "odom") G.addOdometryChain(3.f);
        // the collection marks are per current key frame id: clear them so that the real call below collects again
        for (auto &k : kfs) k->mnBALocalForKF = k->mnBAFixedForKF = 0;
        for (auto &p : mps) p->mnBALocalForKF = 0;
        for (auto &p : mpbs) p->mnBALocalForKF = 0;
      }
      std::vector<int32_t> kfIds, mpIds, mpbIds;
      for (KeyFrame *k : G.kfs) kfIds.push_back((int32_t)k->mnId);
      for (MapPoint *p : G.mps) mpIds.push_back((int32_t)p->mnId);
      for (MapPointBird *p : G.mpbs) mpbIds.push_back((int32_t)p->mnId);
      wr(o, kfIds); wr(o, G.kfFixed); wr(o, G.kfTcw); wr(o, mpIds); wr(o, G.mpXw); wr(o, mpbIds); wr(o, G.mpbXw);
      wr(o, G.obsKf); wr(o, G.obsMp); wr(o, G.obsUv); wr(o, G.obsInf);
      wr(o, G.bobsKf); wr(o, G.bobsMpb); wr(o, G.bobsXc); wr(o, G.bobsInf);
      wr(o, G.odomI); wr(o, G.odomJ); wr(o, G.odomT); wr(o, G.odomInfo);
    }
    bool stop = false;
    if (dumpOnly) {}
    else if (mode == "local") Optimizer::LocalBundleAdjustment(kfs[cur].get(), &stop, &map);
    else if (mode == "odom") Optimizer::LocalBundleAdjustmentWithOdom(kfs[cur].get(), &stop, &map);
    else if (mode == "global") Optimizer::GlobalBundleAdjustemntWithOdom(&map, gits, &stop, 0, true);
  } catch (const std::exception &e) {
    std::fprintf(stderr, "error: %s\n", e.what());
    return 1;
  }
  std::vector<float> oT, oMp, oMpb, oNrm, oDist;
  std::vector<uint8_t> bad, erased(nobs), berased(nbobs);
  for (auto &k : kfs) oT.insert(oT.end(), k->Tcw, k->Tcw + 12);
  for (auto &p : mps) {
    oMp.insert(oMp.end(), p->mWorldPos, p->mWorldPos + 3);
    oNrm.insert(oNrm.end(), p->mNormalVector, p->mNormalVector + 3);
    oDist.push_back(p->mfMinDistance); oDist.push_back(p->mfMaxDistance);
    bad.push_back(p->isBad());
  }
  for (auto &p : mpbs) oMpb.insert(oMpb.end(), p->mWorldPos, p->mWorldPos + 3);
  for (int e = 0; e < nobs; e++) erased[e] = kfs[okf[e]]->mvpMapPoints[obsSlot[e]] == nullptr;
  for (int e = 0; e < nbobs; e++) berased[e] = kfs[bkf[e]]->mvpMapPointsBird[bobsSlot[e]] == nullptr;
  wr(o, oT); wr(o, oMp); wr(o, oMpb); wr(o, oNrm); wr(o, oDist); wr(o, bad); wr(o, erased); wr(o, berased);
  std::fclose(o);
  std::printf("ok mode=%s kf=%d mp=%zu\n", mode.c_str(), nkf, map.mspMapPoints.size());
  return 0;
}
