// runtime.hip -- error channel and device selection of the C-ABI (include/fishbird.h).
#include "fb_common.h"

#include <cstring>
#include <map>
#include <mutex>
#include <utility>

namespace fb {

static thread_local char g_err[512] = "";

void set_error(const char *fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

int check_device() {
  int n = 0;
  hipError_t e = hipGetDeviceCount(&n);
  if (e != hipSuccess || n <= 0) {
    set_error("no HIP device available (%s); this library has no CPU fallback",
              e == hipSuccess ? "device count 0" : hipGetErrorString(e));
    return FB_ERR_NODEVICE;
  }
  return FB_OK;
}

// ---- device scratch pool (see DevBuf) ----
namespace {
std::mutex g_pool_mu;
std::map<std::pair<int, size_t>, std::vector<void *>> g_free;  // (device, size class) -> blocks
size_t g_pooled_bytes = 0;
constexpr size_t kPoolCap = (size_t)4 << 30;  // keep at most 4 GiB of idle scratch
}  // namespace

void *pool_take(size_t bytes, size_t *granted) {
  size_t cls = 256;
  while (cls < bytes) cls <<= 1;
  int dev = 0;
  (void)hipGetDevice(&dev);
  {
    std::lock_guard<std::mutex> lk(g_pool_mu);
    auto it = g_free.find({dev, cls});
    if (it != g_free.end() && !it->second.empty()) {
      void *p = it->second.back();
      it->second.pop_back();
      g_pooled_bytes -= cls;
      *granted = cls;
      return p;
    }
  }
  void *p = nullptr;
  hipError_t e = hipMalloc(&p, cls);
  if (e != hipSuccess) {
    set_error("hipMalloc(%zu) -> %s", cls, hipGetErrorString(e));
    return nullptr;
  }
  *granted = cls;
  return p;
}

void pool_give(void *p, size_t granted) {
  int dev = 0;
  (void)hipGetDevice(&dev);
  std::lock_guard<std::mutex> lk(g_pool_mu);
  if (g_pooled_bytes + granted > kPoolCap) { (void)hipFree(p); return; }
  g_free[{dev, granted}].push_back(p);
  g_pooled_bytes += granted;
}

// per-thread page-locked staging block of the host-pointer entry points (Stager)
namespace {
struct PinBlock {  // (no destructor: a thread may end after the HIP runtime has been torn down; fb_shutdown releases it)
  void *p = nullptr;
  size_t bytes = 0;
};
thread_local PinBlock g_pin;
}  // namespace
void *pinned_scratch(size_t bytes) {
  if (g_pin.bytes >= bytes) return g_pin.p;
  if (g_pin.p) { (void)hipHostFree(g_pin.p); g_pin.p = nullptr; g_pin.bytes = 0; }
  size_t cls = 1 << 16;
  while (cls < bytes) cls <<= 1;
  void *p = nullptr;
  const hipError_t e = hipHostMalloc(&p, cls, hipHostMallocDefault);
  if (e != hipSuccess) { set_error("hipHostMalloc(%zu) -> %s", cls, hipGetErrorString(e)); (void)hipGetLastError(); return nullptr; }
  g_pin.p = p; g_pin.bytes = cls;
  return p;
}

void pool_release() {
  if (g_pin.p) { (void)hipHostFree(g_pin.p); g_pin.p = nullptr; g_pin.bytes = 0; }  // the calling thread's staging block
  std::lock_guard<std::mutex> lk(g_pool_mu);
  int cur = 0;
  const bool have = hipGetDevice(&cur) == hipSuccess;
  for (auto &kv : g_free) {
    if (kv.second.empty() || hipSetDevice(kv.first.first) != hipSuccess) continue;
    for (void *p : kv.second) (void)hipFree(p);
    kv.second.clear();
  }
  g_pooled_bytes = 0;
  if (have) (void)hipSetDevice(cur);
}

bool g_prof_on = false;
int g_prof_only = -1;
namespace {
// the names rocprofv3 shows, without the namespace and the argument list (k_fast: one row per tile-pitch instantiation,
// 44 = 1280x720 / 640x480 levels, 56 = 512x512 levels)
const char *kProfNames[P_COUNT] = {"k_resize", "k_fast<44>", "k_octree", "k_describe", "k_grid_build", "k_bird_keys_to_cam",
                                   "k_descriptor_distance", "k_proj_frame", "k_proj_points", "k_bird_mappoints",
                                   "k_birdview", "k_match_bow", "k_match_triangulation", "k_pose_opt", "k_pose_gather", "k_ba_linearize", "k_ba_schur",
                                   "k_ba_solve", "k_ba_update", "k_ba_misc", "k_proj_kf", "k_match_bow_kf", "k_in_frustum",
                                   "k_undistort", "k_blur", "k_kf_search", "k_distinctive", "k_bow_transform", "k_fast<56>", "k_fast<72>", "k_track_glue"};
struct ProfRec { int id; hipEvent_t a, b; };
std::mutex g_prof_mu;
std::vector<ProfRec> g_recs;
std::vector<hipEvent_t> g_pool;
hipEvent_t take_event() {
  if (!g_pool.empty()) { hipEvent_t e = g_pool.back(); g_pool.pop_back(); return e; }
  hipEvent_t e = nullptr;
  (void)hipEventCreate(&e);
  return e;
}
}  // namespace

hipEvent_t prof_begin(int id, hipStream_t s) {
  std::lock_guard<std::mutex> lk(g_prof_mu);
  ProfRec r{id, take_event(), take_event()};
  (void)hipEventRecord(r.a, s);
  g_recs.push_back(r);
  return r.b;
}
void prof_end(hipEvent_t end, hipStream_t s) { (void)hipEventRecord(end, s); }

}  // namespace fb

extern "C" {

int fb_prof_enable(int on) { fb::g_prof_on = on != 0; return FB_OK; }

int fb_prof_only(const char *kernel_name) {
  fb::g_prof_only = -1;
  if (!kernel_name || !*kernel_name) return FB_OK;
  for (int i = 0; i < fb::P_COUNT; i++)
    if (std::strcmp(fb::kProfNames[i], kernel_name) == 0) { fb::g_prof_only = i; return FB_OK; }
  fb::set_error("fb_prof_only: unknown kernel '%s'", kernel_name);
  return FB_ERR_ARG;
}

int fb_prof_reset(void) {
  std::lock_guard<std::mutex> lk(fb::g_prof_mu);
  for (auto &r : fb::g_recs) { fb::g_pool.push_back(r.a); fb::g_pool.push_back(r.b); }
  fb::g_recs.clear();
  return FB_OK;
}

int fb_prof_report(fb_prof_entry *out, int cap) {
  double tot[fb::P_COUNT] = {0};
  int cnt[fb::P_COUNT] = {0};
  std::lock_guard<std::mutex> lk(fb::g_prof_mu);
  for (auto &r : fb::g_recs) {
    if (hipEventSynchronize(r.b) != hipSuccess) continue;
    float ms = 0;
    if (hipEventElapsedTime(&ms, r.a, r.b) != hipSuccess) continue;
    tot[r.id] += ms;
    cnt[r.id]++;
  }
  int n = 0;
  for (int i = 0; i < fb::P_COUNT && n < cap; i++) {
    if (!cnt[i]) continue;
    memset(&out[n], 0, sizeof(out[n]));
    strncpy(out[n].name, fb::kProfNames[i], sizeof(out[n].name) - 1);
    out[n].launches = cnt[i];
    out[n].total_ms = tot[i];
    n++;
  }
  return n;
}

int fb_abi_version(void) { return FB_ABI_VERSION; }

const char *fb_last_error(void) { return fb::g_err; }

int fb_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  return n;
}

int fb_set_device(int device) {
  FB_TRY(fb::check_device());
  FB_HIP(hipSetDevice(device));
  return FB_OK;
}

}  // extern "C"
