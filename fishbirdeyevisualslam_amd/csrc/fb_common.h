/*
 * fb_common.h -- shared host/device plumbing of the HIP library (gfx950 only).
 */
#ifndef FB_COMMON_H_
#define FB_COMMON_H_

#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <climits>
#include <cstring>
#include <type_traits>
#include <vector>

#include "../../include/fishbird.h"

#define FB_HD __host__ __device__
#include "fb_detmath.h"

namespace fb {

void set_error(const char *fmt, ...);
int check_device();  // FB_OK or FB_ERR_NODEVICE (sets error)

#define FB_HIP(expr)                                                                   \
  do {                                                                                 \
    hipError_t e_ = (expr);                                                            \
    if (e_ != hipSuccess) {                                                            \
      fb::set_error("%s:%d: %s -> %s", __FILE__, __LINE__, #expr, hipGetErrorString(e_)); \
      (void)hipGetLastError(); /* the runtime also remembers it: a later hipGetLastError() must not report it again */ \
      return FB_ERR_HIP;                                                               \
    }                                                                                  \
  } while (0)

#define FB_ARG(cond)                                                        \
  do {                                                                      \
    if (!(cond)) {                                                          \
      fb::set_error("%s:%d: bad argument: %s", __FILE__, __LINE__, #cond); \
      return FB_ERR_ARG;                                                    \
    }                                                                       \
  } while (0)

#define FB_TRY(expr)           \
  do {                         \
    int rc_ = (expr);          \
    if (rc_ != FB_OK) return rc_; \
  } while (0)

inline hipStream_t as_stream(void *s) { return reinterpret_cast<hipStream_t>(s); }

// ---- per-kernel timing (fb_prof_* in include/fishbird.h) ----
enum ProfId {
  P_RESIZE, P_FAST, P_OCTREE, P_DESCRIBE, P_GRID, P_BIRDCAM, P_HAMMING, P_PROJ_FRAME, P_PROJ_POINTS, P_BIRD_MP,
  P_BIRDVIEW, P_BOW, P_TRIANG, P_POSE, P_GATHER, P_BA_LINEARIZE, P_BA_SCHUR, P_BA_SOLVE, P_BA_UPDATE, P_BA_MISC,
  P_PROJ_KF, P_BOW_KF, P_FRUSTUM, P_UNDISTORT, P_BLUR, P_FUSE, P_DISTINCT, P_BOWT, P_FAST56, P_FAST72, P_TRACK_GLUE, P_COUNT
};
extern bool g_prof_on;
extern int g_prof_only;  // -1 = every kernel, else only this ProfId is bracketed
hipEvent_t prof_begin(int id, hipStream_t s);  // records the start event, returns the end event of this bracket
void prof_end(hipEvent_t end, hipStream_t s);
struct ProfScope {  // (the record list is mutex-protected and a scope closes its OWN bracket: launches from several host threads)
  hipStream_t s;
  bool on;
  hipEvent_t end = nullptr;
  ProfScope(int id, hipStream_t st) : s(st), on(g_prof_on && (g_prof_only < 0 || g_prof_only == id)) { if (on) end = prof_begin(id, s); }
  ~ProfScope() { if (on) prof_end(end, s); }
};

// Device scratch that lives for one host-pointer call.
// hipMalloc/hipFree cost tens of microseconds each and the host-pointer entry points need dozens of scratch
// buffers per call, so freed blocks are kept in a per-device pool of power-of-two size classes (runtime.hip).
void *pool_take(size_t bytes, size_t *granted);  // nullptr on failure (error set)
void pool_give(void *p, size_t granted);
void pool_release();  // frees every idle block (fb_shutdown)

struct DevBuf {
  void *p = nullptr;
  size_t bytes = 0, granted = 0;
  DevBuf() {}
  DevBuf(const DevBuf &) = delete;
  DevBuf &operator=(const DevBuf &) = delete;
  ~DevBuf() { if (p) pool_give(p, granted); }
  int alloc(size_t n) {
    if (p) { pool_give(p, granted); p = nullptr; }
    bytes = n ? n : 1;
    p = pool_take(bytes, &granted);
    if (!p) return FB_ERR_HIP;
    return FB_OK;
  }
  int upload(const void *src, size_t n) {
    FB_TRY(alloc(n));
    if (n && src) FB_HIP(hipMemcpy(p, src, n, hipMemcpyHostToDevice));
    return FB_OK;
  }
  int download(void *dst, size_t n) const {
    if (n) FB_HIP(hipMemcpy(dst, p, n, hipMemcpyDeviceToHost));
    return FB_OK;
  }
  template <typename T> T *as() const { return reinterpret_cast<T *>(p); }
};

// One staged transfer per host-pointer call.  The drop-in entry points used to upload every argument array with its own
// synchronous hipMemcpy (16 of them for SearchByProjection: ~0.2 ms of a 0.33 ms call, `host_abi` in bench.py): the arrays of a
// call are now packed into one page-locked block (per host thread, grows on demand), travel with ONE copy into one device
// block, and the outputs come back with one copy.  in() = input only; out() = output, optionally uploaded first (in/out
// arrays: entries the kernel does not write keep the caller's contents).
void *pinned_scratch(size_t bytes);  // per-thread page-locked block of at least `bytes` (nullptr on failure, error set)
struct Stager {
  struct Item { void **field; const void *src; void *dst; size_t bytes, off; };
  std::vector<Item> ins, outs;
  DevBuf dev;
  uint8_t *pin = nullptr;
  size_t inBytes = 0, total = 0;
  static size_t up256(size_t v) { return (v + 255) & ~(size_t)255; }
  void in(void **field, const void *src, size_t bytes) { if (src) ins.push_back({field, src, nullptr, bytes, 0}); }
  void out(void **field, void *host, size_t bytes, bool copy_in) { if (host) outs.push_back({field, copy_in ? host : nullptr, host, bytes, 0}); }
  int commit(hipStream_t s) {
    size_t off = 0;
    for (auto &it : ins) { it.off = off; off += up256(it.bytes ? it.bytes : 1); }
    inBytes = off;
    for (auto &it : outs) { it.off = off; off += up256(it.bytes ? it.bytes : 1); }
    total = off ? off : 256;
    pin = static_cast<uint8_t *>(pinned_scratch(total));
    if (!pin) return FB_ERR_HIP;
    FB_TRY(dev.alloc(total));
    bool anyOutIn = false;
    for (auto &it : ins) if (it.bytes) memcpy(pin + it.off, it.src, it.bytes);
    for (auto &it : outs) if (it.src && it.bytes) { memcpy(pin + it.off, it.src, it.bytes); anyOutIn = true; }
    const size_t upBytes = anyOutIn ? total : inBytes;
    if (upBytes) FB_HIP(hipMemcpyAsync(dev.p, pin, upBytes, hipMemcpyHostToDevice, s));
    for (auto &it : ins) *it.field = dev.as<uint8_t>() + it.off;
    for (auto &it : outs) *it.field = dev.as<uint8_t>() + it.off;
    return FB_OK;
  }
  int fetch(hipStream_t s) {  // waits for the stream, then hands the outputs to the caller's arrays
    if (total > inBytes) FB_HIP(hipMemcpyAsync(pin + inBytes, dev.as<uint8_t>() + inBytes, total - inBytes, hipMemcpyDeviceToHost, s));
    FB_HIP(hipStreamSynchronize(s));
    for (auto &it : outs) if (it.bytes) memcpy(it.dst, pin + it.off, it.bytes);
    return FB_OK;
  }
};

// Column sums of up to 32 per-lane values over the 64 lanes as a reduce-scatter butterfly: at the step with mask m a
// lane sends one half of its values to lane^m, keeps the other half and adds what it receives, so the value count halves
// every step (16 + 8 + 4 + 2 + 1 + 1 = 32 fp64 exchanges instead of 6 per value).  On return v[0] of lane L is the sum
// over all lanes of column L >> 1.  (28 accumulators through plain butterflies were most of an LM evaluation: 168
// dependent cross-lane exchanges per wave.)
// c ? a : b on the two halves of a double, in assembly: written as a C++ select between two array elements the compiler
// turns the pair into ONE load with a computed index, which forces the whole array out of registers into scratch memory
__device__ __forceinline__ double select_f64(bool c, double a, double b) {
  const unsigned long long ua = __builtin_bit_cast(unsigned long long, a), ub = __builtin_bit_cast(unsigned long long, b);
  unsigned lo, hi;
  const unsigned long long m = __builtin_amdgcn_ballot_w64(c);
  asm("v_cndmask_b32 %0, %1, %2, %3" : "=v"(lo) : "v"((unsigned)ub), "v"((unsigned)ua), "s"(m));
  asm("v_cndmask_b32 %0, %1, %2, %3" : "=v"(hi) : "v"((unsigned)(ub >> 32)), "v"((unsigned)(ua >> 32)), "s"(m));
  return __builtin_bit_cast(double, ((unsigned long long)hi << 32) | lo);
}

__device__ __forceinline__ double wave_column_sums32(double (&v)[32], int lane) {
#pragma unroll
  for (int half = 16, m = 32; half >= 1; half >>= 1, m >>= 1) {
    const bool up = (lane & m) != 0;
#pragma unroll
    for (int i = 0; i < half; i++) {
      const double send = select_f64(up, v[i], v[i + half]);
      const double keep = select_f64(up, v[i + half], v[i]);
      v[i] = keep + __shfl_xor(send, m, 64);
    }
  }
  return v[0] + __shfl_xor(v[0], 1, 64);
}

// 256-bit Hamming distance of two 32-byte rows given as 8 dwords each
// (DescriptorDistance, ORBmatcher.cc:1951-1967; v_bcnt_u32_b32 instead of the SWAR bithack)
__device__ __forceinline__ int hamming256(const uint32_t a[8], const uint4 *__restrict__ b) {
  const uint4 b0 = b[0], b1 = b[1];
  return __popc(a[0] ^ b0.x) + __popc(a[1] ^ b0.y) + __popc(a[2] ^ b0.z) + __popc(a[3] ^ b0.w) +
         __popc(a[4] ^ b1.x) + __popc(a[5] ^ b1.y) + __popc(a[6] ^ b1.z) + __popc(a[7] ^ b1.w);
}

}  // namespace fb
#endif
