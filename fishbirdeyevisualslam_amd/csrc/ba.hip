// ba.hip -- Optimizer::LocalBundleAdjustment / LocalBundleAdjustmentWithOdom (placeholder until the
// Schur kernels land later in this round; fails loudly, never silently succeeds).
#include "fb_common.h"

extern "C" int fb_local_ba(const fb_local_ba_args *args) {
  (void)args;
  fb::set_error("fb_local_ba: not implemented yet in this build");
  return FB_ERR_ARG;
}
