// runtime.hip -- error channel and device selection of the C-ABI (include/fishbird.h).
#include "fb_common.h"

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

}  // namespace fb

extern "C" {

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
