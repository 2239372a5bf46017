// orb.hip -- ORBextractor as CDNA4 kernels (gfx950), batched over equally sized images.
//
// Replaces (reference file:line):
//   ORBextractor::ORBextractor        src/ORBextractor.cc:410-470   (host tables)
//   ORBextractor::ComputePyramid      :1107-1132   -> k_resize (one launch per level >= 1)
//   ComputeKeyPointsOctTree cell loop :765-832     -> k_fast   (one wave per run of 30-px cells,
//        cv::FAST 9-16 score + 3x3 NMS + ini/min threshold fallback on an LDS tile)
//   DistributeOctTree / DivideNode    :481-763     -> k_octree (one workgroup per image level)
//   GaussianBlur :1086                              -> k_blur   (whole level, separable 7x7, dot4)
//   IC_Angle / computeOrbDescriptor   :77-147      -> k_describe (one wave per two key points: raw and
//        blurred patches staged in LDS -> moments, 256 rBRIEF tests)
//   ORBextractor::operator()          :1043-1105   -> fb_orb_extract*
//
// HBM layout per image: level 0 is the caller's image (never copied); levels 1..n-1 live in
// one pitched buffer (pitch % 64 == 0), the blurred levels in a second one of the same geometry.
// FAST candidates are packed x | y<<12 | score<<24 into per-cell runs of a per-level array sized
// for the NMS worst case, so no kernel can overflow a buffer.
//
// OpenCV semantics (not vendored in the reference -> "parity unpinned") are
// those of oracle/orb_oracle.cpp; this file must agree with it bit for bit.
#include "fb_common.h"

namespace {

constexpr int EDGE_THRESHOLD = 19;  // ORBextractor.cc:74
constexpr int HALF_PATCH = 15;      // :73
constexpr int PATCH_SIZE = 31;      // :72
constexpr int BORDER = EDGE_THRESHOLD - 3;  // minBorderX/Y, :773

// stored as float (the values are small integers): the rotation below works in float, this saves the conversions
__constant__ float c_pattern[1024] = {
#include "orb_pattern.inc"
};

struct LevelInfo {
  int w, h, pitch;
  long long off;  // byte offset of the level inside one image's pyramid buffer (levels >= 1)
  int nCols, nRows, wCell, hCell, cellBase;
  int N;                 // mnFeaturesPerLevel
  long long candBase;    // element offset into one image's candidate array
  int candCap;
  int slotCap;           // candidate slots per FAST cell (k_fast writes cell c's survivors at candBase + c * slotCap)
  int outBase, outCap;   // element offset / capacity in one image's per-level keypoint array
  int nIni;
  float hX;
  float scale;
  int patchSize;         // (int)(PATCH_SIZE*scale), :836
  long long boff;        // byte offset of the level's blurred copy inside one image's blur buffer
};

struct OrbK {
  int nlevels, iniTh, minTh, totalCells, outStride, capOut;
  int kpStride;  // entries per image in the caller's key-point / descriptor arrays (= capOut unless fb_orb_set_output_stride)
  int fastTileBytes, fastMaxOut, fastMaxPix;  // LDS carve of k_fast
  int fastTP;                                 // tile pitch instantiation of k_fast (44 / 56 / 72)
  int dbg;  // FB_FAST_DBG ablation switch (0 = normal)
  unsigned long long *timers;  // FB_FAST_DBG=20: per-phase wave time of k_fast (fb_orb_debug_timers)
  int cellBase[FB_MAX_LEVELS + 1];  // first FAST cell of each level (contiguous copy for one scalar load)
  int grpBase[FB_MAX_LEVELS + 1];   // first k_fast wave of each level (a wave = FAST_CPW consecutive cells)
  int totalGroups;
  long long pyrStride;   // bytes per image of levels >= 1
  long long blurStride;  // bytes per image of the blurred pyramid (all levels)
  int blurStrips[FB_MAX_LEVELS + 1];  // first k_blur strip of each level
  long long candStride;  // candidates per image
  int umax[16];
  LevelInfo L[FB_MAX_LEVELS];
};

struct ResizeTabs {  // per level >= 1, device pointers
  const int *xofs;
  const short *ialpha;
  const int *yofs;
  const short *ibeta;
};

// ------------------------------------------------------------------------------------------
// cv::resize INTER_LINEAR, 8U, 11-bit fixed point (oracle: resize_linear_u8). 4 px per lane.
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ int win_byte(uint32_t w0, uint32_t w1, uint32_t w2, uint32_t w3, int k) {  // byte k of a 16-byte window
  const uint32_t lo = k & 8 ? w2 : w0, hi = k & 8 ? w3 : w1;
  const uint32_t w = k & 4 ? hi : lo;
  return (w >> (8 * (k & 3))) & 0xFF;
}

__global__ __launch_bounds__(256) void k_resize(const uint8_t *__restrict__ src0, long long srcImgStride, int sw,
                                                int sh, int spitch, uint8_t *__restrict__ dst0, long long dstImgStride,
                                                int dw, int dh, int dpitch, ResizeTabs tb) {
  const int b = blockIdx.z;
  const int dy = blockIdx.y * blockDim.y + threadIdx.y;
  const int dx4 = (blockIdx.x * blockDim.x + threadIdx.x) * 4;
  if (dy >= dh || dx4 >= dpitch) return;
  const uint8_t *src = src0 + (long long)b * srcImgStride;
  uint8_t *dst = dst0 + (long long)b * dstImgStride;
  int sy0 = tb.yofs[dy], sy1 = sy0 + 1;
  sy0 = sy0 >= 0 ? (sy0 < sh ? sy0 : sh - 1) : 0;
  sy1 = sy1 >= 0 ? (sy1 < sh ? sy1 : sh - 1) : 0;
  const int b0 = tb.ibeta[dy * 2], b1 = tb.ibeta[dy * 2 + 1];
  const uint8_t *r0p = src + (long long)sy0 * spitch, *r1p = src + (long long)sy1 * spitch;
  uint32_t packed = 0;
  // the 4 destination pixels of this lane read source columns [sxa, sxb+1]; when that span fits a 16-byte
  // window that is 4-byte aligned and inside the row, fetch it with 4 dword loads per row instead of 16 byte loads
  const int dxl = min(dx4 + 3, dw - 1);
  const int sxa = dx4 < dw ? tb.xofs[dx4] : 0, sxb = dx4 < dw ? min(tb.xofs[dxl] + 1, sw - 1) : 0;
  const int a = sxa & ~3;
  const bool wide = ((reinterpret_cast<uintptr_t>(src) | (uintptr_t)spitch) & 3) == 0 && (sxb - a) < 16 && a + 16 <= spitch;
  if (wide) {
    const uint32_t *p0 = reinterpret_cast<const uint32_t *>(r0p + a), *p1 = reinterpret_cast<const uint32_t *>(r1p + a);
    const uint32_t u0 = p0[0], u1 = p0[1], u2 = p0[2], u3 = p0[3], v0 = p1[0], v1 = p1[1], v2 = p1[2], v3 = p1[3];
#pragma unroll
    for (int k = 0; k < 4; k++) {
      const int dx = dx4 + k;
      int v = 0;
      if (dx < dw) {
        const int sx = tb.xofs[dx];
        const int sx1 = min(sx + 1, sw - 1);
        const int a0 = tb.ialpha[dx * 2], a1 = tb.ialpha[dx * 2 + 1];
        const int r0 = win_byte(u0, u1, u2, u3, sx - a) * a0 + win_byte(u0, u1, u2, u3, sx1 - a) * a1;
        const int r1 = win_byte(v0, v1, v2, v3, sx - a) * a0 + win_byte(v0, v1, v2, v3, sx1 - a) * a1;
        v = (((b0 * (r0 >> 4)) >> 16) + ((b1 * (r1 >> 4)) >> 16) + 2) >> 2;
        v = min(max(v, 0), 255);
      }
      packed |= (uint32_t)v << (8 * k);
    }
  } else {
#pragma unroll
    for (int k = 0; k < 4; k++) {
      const int dx = dx4 + k;
      int v = 0;
      if (dx < dw) {
        const int sx = tb.xofs[dx];
        const int sx1 = min(sx + 1, sw - 1);
        const int a0 = tb.ialpha[dx * 2], a1 = tb.ialpha[dx * 2 + 1];
        const int r0 = r0p[sx] * a0 + r0p[sx1] * a1;
        const int r1 = r1p[sx] * a0 + r1p[sx1] * a1;
        v = (((b0 * (r0 >> 4)) >> 16) + ((b1 * (r1 >> 4)) >> 16) + 2) >> 2;
        v = min(max(v, 0), 255);
      }
      packed |= (uint32_t)v << (8 * k);
    }
  }
  *reinterpret_cast<uint32_t *>(dst + (long long)dy * dpitch + dx4) = packed;
}

// Same arithmetic, restructured for throughput (used when the source is dword aligned and the scale is <= 2, i.e.
// always for the ORB pyramid): a lane owns 4 adjacent destination pixels and walks RR destination rows.  The column
// constants (window offset, byte offsets, packed weights) live in registers for the whole walk; per source row the
// lane fetches one 12-byte window (3 dword loads) and does the horizontal pass with v_alignbyte + v_perm + v_dot2;
// a source row shared by two successive destination rows (4 of 5 at scale 1.2) is filtered once.
constexpr int RESIZE_ROWS = 8;

__global__ __launch_bounds__(256) void k_resize_rows(const uint8_t *__restrict__ src0, long long srcImgStride, int sw,
                                                     int sh, int spitch, uint8_t *__restrict__ dst0,
                                                     long long dstImgStride, int dw, int dh, int dpitch, ResizeTabs tb) {
  const int b = blockIdx.z;
  const int dyBase = __builtin_amdgcn_readfirstlane((blockIdx.y * blockDim.y + threadIdx.y) * RESIZE_ROWS);
  const int dx4 = (blockIdx.x * blockDim.x + threadIdx.x) * 4;
  if (dyBase >= dh || dx4 >= dpitch) return;
  const uint8_t *src = src0 + (long long)b * srcImgStride;
  uint8_t *dst = dst0 + (long long)b * dstImgStride;
  // column constants
  const int dxf = min(dx4, dw - 1);
  const int a = tb.xofs[dxf] & ~3;
  const int aw = min(a, spitch - 12);  // keep the 12-byte window inside the row (only the zero-padded tail moves)
  // The taps of the lane's 4 pixels lie within 8 bytes of the first tap (scale <= 2): two v_alignbyte bring those 8
  // bytes to a fixed place, one v_perm per pixel spreads its two taps for v_dot2.
  uint32_t selw[4];                    // v_perm selector: tap0 -> bits 0..7, tap1 -> bits 16..23 of the shifted pair
  uint32_t wgt[4];                     // a0 | a1 << 16
  uint32_t liveMask = 0;               // bytes of the packed result that are destination pixels
  const int o0 = tb.xofs[dxf] - aw;    // byte offset of the first tap inside the 12-byte window, 0..10
  const uint32_t sh0 = (uint32_t)o0 & 3u;
  const int q0 = o0 >> 2;              // 0, except in the clamped window of the row's tail
#pragma unroll
  for (int k = 0; k < 4; k++) {
    const int dx = dx4 + k;
    if (dx < dw) liveMask |= 0xffu << (8 * k);
    const int dxc = min(dx, dw - 1);
    const int sx = tb.xofs[dxc];
    int a0 = tb.ialpha[dxc * 2], a1 = tb.ialpha[dxc * 2 + 1];
    if (sx + 1 > sw - 1) { a0 += a1; a1 = 0; }  // right border: both taps read sx
    const uint32_t ob = (uint32_t)(sx - aw - o0);  // 0..6
    selw[k] = ob | (0x0cu << 8) | ((ob + 1u) << 16) | (0x0cu << 24);
    wgt[k] = (uint32_t)(a0 & 0xffff) | ((uint32_t)a1 << 16);
  }
  const bool anyTail = __ballot(q0 != 0) != 0ull;  // wave-uniform
  const uint32_t mq1 = q0 == 1 ? ~0u : 0u, mq2 = q0 >= 2 ? ~0u : 0u, mq12 = mq1 | mq2;
  auto loadWin = [&](int sy, uint32_t (&wv)[3]) {
    sy = sy >= 0 ? (sy < sh ? sy : sh - 1) : 0;
    const uint32_t *p = reinterpret_cast<const uint32_t *>(src + ((uint32_t)__mul24(sy, spitch) + (uint32_t)aw));
    wv[0] = p[0]; wv[1] = p[1]; wv[2] = p[2];
  };
  auto hpass = [&](const uint32_t (&wv)[3], int h[4]) {
    uint32_t x0 = wv[0], x1 = wv[1], x2 = wv[2];
    if (anyTail) {  // the first tap sits in dword q0 of the window (bit selects: no divergent branches)
      x0 = (wv[0] & ~mq12) | (wv[1] & mq1) | (wv[2] & mq2);
      x1 = (wv[1] & ~mq12) | (wv[2] & mq12);
    }
    const uint32_t e0 = __builtin_amdgcn_alignbyte(x1, x0, sh0), e1 = __builtin_amdgcn_alignbyte(x2, x1, sh0);
#pragma unroll
    for (int k = 0; k < 4; k++) {
      const uint32_t spread = __builtin_amdgcn_perm(e1, e0, selw[k]);  // tap0 | tap1 << 16
      int d;
      asm("v_dot2_i32_i16 %0, %1, %2, 0" : "=v"(d) : "v"(spread), "v"(wgt[k]));
      h[k] = d >> 4;
    }
  };
  int h0[4], h1[4];
  int prevS1 = -0x40000000;
  const int dyEnd = min(dyBase + RESIZE_ROWS, dh);
  // every source window of the RESIZE_ROWS destination rows is requested before the first one is used (a row that two
  // destination rows share is simply requested twice): load -> wait -> use per row serialises the memory latency
  int sy0s[RESIZE_ROWS], b0s[RESIZE_ROWS], b1s[RESIZE_ROWS];
  uint32_t wa[RESIZE_ROWS][3], wb[RESIZE_ROWS][3];
#pragma unroll
  for (int j = 0; j < RESIZE_ROWS; j++) {
    const int dyc = min(dyBase + j, dh - 1);
    sy0s[j] = tb.yofs[dyc];
    b0s[j] = tb.ibeta[dyc * 2];
    b1s[j] = tb.ibeta[dyc * 2 + 1];
    loadWin(sy0s[j], wa[j]);
    loadWin(sy0s[j] + 1, wb[j]);
  }
#pragma unroll
  for (int j = 0; j < RESIZE_ROWS; j++) {
    const int dy = dyBase + j;
    if (dy >= dyEnd) break;
    const int sy0 = sy0s[j];
    const int b0 = b0s[j], b1 = b1s[j];
    if (sy0 == prevS1) {
#pragma unroll
      for (int k = 0; k < 4; k++) h0[k] = h1[k];
    } else {
      hpass(wa[j], h0);
    }
    hpass(wb[j], h1);
    prevS1 = sy0 + 1;
    uint32_t packed = 0;
#pragma unroll
    for (int k = 0; k < 4; k++) {
      // |b| <= 2048, h <= 255 * 2048 >> 4, b0 + b1 = 2048, no negative weight: 0 <= v <= 255 without a clamp
      const int v = ((__mul24(b0, h0[k]) >> 16) + (__mul24(b1, h1[k]) >> 16) + 2) >> 2;
      packed |= (uint32_t)v << (8 * k);
    }
    *reinterpret_cast<uint32_t *>(dst + (long long)dy * dpitch + dx4) = packed & liveMask;
  }
}

// k_resize_rows loads BOTH source rows of every destination row although neighbours share one (at scale 1.2 eight rows ask
// for 16 windows but need 10 or 11 different ones), and the kernel is bound by the bytes it has in flight per wave.  This one
// walks the SOURCE rows of a run of RM_ROWS destination rows once: all RM_SRC windows are requested up front, every source row
// is filtered horizontally once, and a destination row is written when its second source row has been filtered (each source
// row closes at most one destination row because the scale is >= 1).  Same arithmetic as k_resize_rows; used when every run of
// RM_ROWS destination rows spans at most RM_SRC source rows (scale factors up to 1.25), k_resize_rows otherwise.
constexpr int RM_ROWS = 12, RM_SRC = 16;

__global__ __launch_bounds__(256) void k_resize_merge(const uint8_t *__restrict__ src0, long long srcImgStride, int sw,
                                                      int sh, int spitch, uint8_t *__restrict__ dst0,
                                                      long long dstImgStride, int dw, int dh, int dpitch, ResizeTabs tb) {
  const int b = blockIdx.z;
  const int dyBase = __builtin_amdgcn_readfirstlane((blockIdx.y * blockDim.y + threadIdx.y) * RM_ROWS);
  const int dx4 = (blockIdx.x * blockDim.x + threadIdx.x) * 4;
  if (dyBase >= dh || dx4 >= dpitch) return;
  const int lane = threadIdx.x;  // blockDim.x = 64: a wave is one row of the block
  const uint8_t *src = src0 + (long long)b * srcImgStride;
  uint8_t *dst = dst0 + (long long)b * dstImgStride;
  // column constants (as in k_resize_rows)
  const int dxf = min(dx4, dw - 1);
  const int a = tb.xofs[dxf] & ~3;
  const int aw = min(a, spitch - 12);
  uint32_t selw[4], wgt[4], liveMask = 0;
  const int o0 = tb.xofs[dxf] - aw;
  const uint32_t sh0 = (uint32_t)o0 & 3u;
  const int q0 = o0 >> 2;
#pragma unroll
  for (int k = 0; k < 4; k++) {
    const int dx = dx4 + k;
    if (dx < dw) liveMask |= 0xffu << (8 * k);
    const int dxc = min(dx, dw - 1);
    const int sx = tb.xofs[dxc];
    int a0 = tb.ialpha[dxc * 2], a1 = tb.ialpha[dxc * 2 + 1];
    if (sx + 1 > sw - 1) { a0 += a1; a1 = 0; }
    const uint32_t ob = (uint32_t)(sx - aw - o0);
    selw[k] = ob | (0x0cu << 8) | ((ob + 1u) << 16) | (0x0cu << 24);
    wgt[k] = (uint32_t)(a0 & 0xffff) | ((uint32_t)a1 << 16);
  }
  const bool anyTail = __ballot(q0 != 0) != 0ull;
  const uint32_t mq1 = q0 == 1 ? ~0u : 0u, mq2 = q0 >= 2 ? ~0u : 0u, mq12 = mq1 | mq2;
  // row constants: lane j of the wave holds the first source row and the two weights of destination row dyBase + j
  const int nrows = min(RM_ROWS, dh - dyBase);
  const int dyl = min(dyBase + min(lane, RM_ROWS - 1), dh - 1);
  const int syLane = tb.yofs[dyl];
  const int betaLane = (int)((uint32_t)(uint16_t)tb.ibeta[dyl * 2] | ((uint32_t)(uint16_t)tb.ibeta[dyl * 2 + 1] << 16));
  const int base = __builtin_amdgcn_readfirstlane(syLane);  // first source row of the run
  uint32_t w[RM_SRC][3];
#pragma unroll
  for (int r = 0; r < RM_SRC; r++) {
    int sy = base + r;
    sy = sy >= 0 ? (sy < sh ? sy : sh - 1) : 0;
    const uint32_t *p = reinterpret_cast<const uint32_t *>(src + ((uint32_t)__mul24(sy, spitch) + (uint32_t)aw));
    w[r][0] = p[0]; w[r][1] = p[1]; w[r][2] = p[2];
  }
  int hp[4] = {0, 0, 0, 0}, hc[4];
  int j = 0;
#pragma unroll
  for (int r = 0; r < RM_SRC; r++) {
    {  // horizontal pass of source row base + r
      uint32_t x0 = w[r][0], x1 = w[r][1];
      const uint32_t x2 = w[r][2];
      if (anyTail) {
        x0 = (w[r][0] & ~mq12) | (w[r][1] & mq1) | (w[r][2] & mq2);
        x1 = (w[r][1] & ~mq12) | (w[r][2] & mq12);
      }
      const uint32_t e0 = __builtin_amdgcn_alignbyte(x1, x0, sh0), e1 = __builtin_amdgcn_alignbyte(x2, x1, sh0);
#pragma unroll
      for (int k = 0; k < 4; k++) {
        const uint32_t spread = __builtin_amdgcn_perm(e1, e0, selw[k]);
        int d;
        asm("v_dot2_i32_i16 %0, %1, %2, 0" : "=v"(d) : "v"(spread), "v"(wgt[k]));
        hc[k] = d >> 4;
      }
    }
    if (r >= 1 && j < nrows) {
      const int sy0 = __builtin_amdgcn_readlane(syLane, j);
      if (sy0 == base + r - 1) {  // wave-uniform: this source row is the second row of destination row j
        const int bw = __builtin_amdgcn_readlane(betaLane, j);
        const int b0 = (int)(short)(bw & 0xffff), b1 = bw >> 16;
        uint32_t packed = 0;
#pragma unroll
        for (int k = 0; k < 4; k++) {
          const int v = ((__mul24(b0, hp[k]) >> 16) + (__mul24(b1, hc[k]) >> 16) + 2) >> 2;
          packed |= (uint32_t)v << (8 * k);
        }
        *reinterpret_cast<uint32_t *>(dst + (long long)(dyBase + j) * dpitch + dx4) = packed & liveMask;
        j++;
      }
    }
#pragma unroll
    for (int k = 0; k < 4; k++) hp[k] = hc[k];
  }
}

// ------------------------------------------------------------------------------------------
// FAST-9-16 score = max over the 16 arcs of 9 of min(d) (and of min(-d)), minus 1
// (cv::cornerScore<16>); the pixel is a corner at threshold t iff score >= t.
// ------------------------------------------------------------------------------------------
// `corner` points 3 rows above and 3 columns left of the pixel: every tap is corner + a non-negative constant, i.e. one LDS
// read with an immediate offset (taps at negative offsets from the pixel cost a vector add each).
template <int TP>
__device__ __forceinline__ int fast_score16(const uint8_t *__restrict__ corner) {
  constexpr int tp = TP;
  const uint8_t *c = corner + 3 * tp + 3;
  const int v = c[0];
  int d[16];
  d[0] = v - c[3 * tp];      d[1] = v - c[3 * tp + 1];   d[2] = v - c[2 * tp + 2];   d[3] = v - c[tp + 3];
  d[4] = v - c[3];           d[5] = v - c[-tp + 3];      d[6] = v - c[-2 * tp + 2];  d[7] = v - c[-3 * tp + 1];
  d[8] = v - c[-3 * tp];     d[9] = v - c[-3 * tp - 1];  d[10] = v - c[-2 * tp - 2]; d[11] = v - c[-tp - 3];
  d[12] = v - c[-3];         d[13] = v - c[tp - 3];      d[14] = v - c[2 * tp - 2];  d[15] = v - c[3 * tp - 1];
  // sliding min / max over the 16 arcs of 9 in three-operand form (v_min3_i32 / v_max3_i32):
  // arc of 3 -> arc of 9 = three arcs of 3 -> reduction, 40 instructions per polarity
  int mn3[16], mx3[16];
#pragma unroll
  for (int i = 0; i < 16; i++) {
    mn3[i] = min(min(d[i], d[(i + 1) & 15]), d[(i + 2) & 15]);
    mx3[i] = max(max(d[i], d[(i + 1) & 15]), d[(i + 2) & 15]);
  }
  int A = -256, Bm = 256;
#pragma unroll
  for (int i = 0; i < 16; i += 2) {
    const int a0 = min(min(mn3[i], mn3[(i + 3) & 15]), mn3[(i + 6) & 15]);
    const int a1 = min(min(mn3[i + 1], mn3[(i + 4) & 15]), mn3[(i + 7) & 15]);
    const int b0 = max(max(mx3[i], mx3[(i + 3) & 15]), mx3[(i + 6) & 15]);
    const int b1 = max(max(mx3[i + 1], mx3[(i + 4) & 15]), mx3[(i + 7) & 15]);
    A = max(max(A, a0), a1);
    Bm = min(min(Bm, b0), b1);
  }
  return max(A, -Bm) - 1;
}

__device__ __forceinline__ int wave_incl_scan(int v) {  // inclusive scan over the 64 lanes
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const int t = __shfl_up(v, o, 64);
    if ((int)(threadIdx.x & 63) >= o) v += t;
  }
  return v;
}

// exclusive scan of one int per thread over an NT-thread block; *total = block sum
template <int NT>
__device__ __forceinline__ int block_excl_scan(int v, int *s_w /*[NT / 64]*/, int *total) {
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int inc = wave_incl_scan(v);
  __syncthreads();
  if (lane == 63) s_w[wv] = inc;
  __syncthreads();
  int base = 0, tot = 0;
#pragma unroll
  for (int i = 0; i < NT / 64; i++) { if (i < wv) base += s_w[i]; tot += s_w[i]; }
  *total = tot;
  return base + inc - v;
}

// exclusive scan of one int per thread over a 256-thread block; *total = block sum
__device__ __forceinline__ int block_excl_scan256(int v, int *s_w /*[4]*/, int *total) {
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int inc = wave_incl_scan(v);
  __syncthreads();
  if (lane == 63) s_w[wv] = inc;
  __syncthreads();
  int base = 0;
  for (int i = 0; i < wv; i++) base += s_w[i];
  *total = s_w[0] + s_w[1] + s_w[2] + s_w[3];
  return base + inc - v;
}

// inclusive scan over the 64 lanes in the VALU (DPP row shifts, then the row broadcasts of gfx9)
__device__ __forceinline__ int wave_incl_scan_dpp(int v) {
  v += __builtin_amdgcn_update_dpp(0, v, 0x111, 0xf, 0xf, false);  // row_shr:1
  v += __builtin_amdgcn_update_dpp(0, v, 0x112, 0xf, 0xf, false);  // row_shr:2
  v += __builtin_amdgcn_update_dpp(0, v, 0x114, 0xf, 0xf, false);  // row_shr:4
  v += __builtin_amdgcn_update_dpp(0, v, 0x118, 0xf, 0xf, false);  // row_shr:8
  v += __builtin_amdgcn_update_dpp(0, v, 0x142, 0xa, 0xf, false);  // row_bcast:15 -> rows 1 and 3
  v += __builtin_amdgcn_update_dpp(0, v, 0x143, 0xc, 0xf, false);  // row_bcast:31 -> rows 2 and 3
  return v;
}

constexpr int FAST_MAX_TILE = 72;  // cell window <= wCell+6 <= 65 px, +3 alignment slack

// One WAVE per run of FAST_CPW consecutive cells (64-thread workgroups, no cross-wave barriers, up to 31 waves per CU).
// Per cell:
//  1. the cell window is staged in LDS with 4-byte loads, all issued before the first LDS write; the window of the
//     NEXT cell is requested before this one is processed;
//  2. per 128 pixels: a necessary corner test on the 4 compass pixels (any 9-arc holds two consecutive compass
//     points), two pixels per lane in packed 16-bit arithmetic, ballot-compaction of the survivors into an
//     iniThFAST list and a minThFAST list; full score only for listed pixels;
//  3. 3x3 strict NMS over the thresholded score tile, survivors staged in LDS (over the image tile);
//  4. if none survived at iniThFAST the cell is redone at minThFAST (ORBextractor.cc:809-816);
//  5. the survivors go to the cell's own run of output slots + a per-cell count (no atomics; k_octree packs the runs).
//     Candidate order is irrelevant downstream (the quadtree breaks response ties with an order key derived from x,y).
// TP = tile pitch in bytes, a compile-time constant so that every LDS access of the sweep / score / NMS is
// base + immediate offset (44 covers cells up to 35 px wide, i.e. every level of the usual image sizes).
#ifndef FB_FAST_WPE44
#define FB_FAST_WPE44 6  // register budget of the 44-byte instantiation, in waves per SIMD
#endif
constexpr int FAST_CPW = 4;  // consecutive cells per wave; the tile of cell i+1 is in flight while cell i is processed
template <int TP, bool TIMED>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(TP == 44 ? FB_FAST_WPE44 : 4, 8))) void k_fast(OrbK K, const uint8_t *__restrict__ img0, long long imgStride, int pitch0,
                                             const uint8_t *__restrict__ pyr, uint32_t *__restrict__ cand,
                                             int *__restrict__ cellCount) {
  // LDS carve (sized on the host for the largest cell of this image size): tile (later: survivors) | sc | list
  extern __shared__ __attribute__((aligned(16))) uint8_t fsm[];
  uint8_t *tile = fsm;
  uint8_t *sc = fsm + K.fastTileBytes;
  // the NMS survivors overlay the image tile: the first survivor is written only after the last read of the tile
  // (a pass that keeps anything is the final pass; 4 * fastMaxOut <= fastTileBytes)
  uint32_t *s_out = reinterpret_cast<uint32_t *>(fsm);
  unsigned short *s_list = reinterpret_cast<unsigned short *>(fsm + 2 * K.fastTileBytes);  // [fastMaxPix]
  const int b = blockIdx.y, lane = threadIdx.x;
  // FB_FAST_DBG=20: one workgroup in 16 accumulates its phase times in registers and adds them once, at the end
  const bool timed = TIMED && (blockIdx.x & 15) == 0;  // the TIMED instantiation is launched for FB_FAST_DBG=20 only
  uint32_t t_mark = 0, t_acc[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};  // shader clock, low 32 bits
  if (timed) { __builtin_amdgcn_sched_barrier(0); t_mark = (uint32_t)__builtin_amdgcn_s_memtime(); __builtin_amdgcn_sched_barrier(0); }
  const uint32_t t_start = t_mark;
#define FAST_TICK(slot_) if (TIMED && timed) { __builtin_amdgcn_sched_barrier(0); const uint32_t t_ = (uint32_t)__builtin_amdgcn_s_memtime(); __builtin_amdgcn_sched_barrier(0); t_acc[slot_] += t_ - t_mark; t_mark = t_; }
  // XCD-aware mapping: workgroups are dealt round-robin over the 8 XCDs (blocks b and b+8 share one), so give each
  // XCD a contiguous run of cell groups -- neighbouring cells overlap by 6 px and share lines in that XCD's L2
  int grp, l = 0;
  {
    const int nb = gridDim.x, per = (nb + 7) >> 3;
    grp = (int)(blockIdx.x & 7) * per + (int)(blockIdx.x >> 3);
    if (grp >= K.totalGroups) return;  // grid is padded to a multiple of 8
  }
#pragma unroll
  for (int i = 1; i < FB_MAX_LEVELS; i++) l += (i < K.nlevels && grp >= K.grpBase[i]) ? 1 : 0;
  const LevelInfo &Lv = K.L[l];
  const int cFirst = (grp - K.grpBase[l]) * FAST_CPW, cEnd = min(cFirst + FAST_CPW, Lv.nCols * Lv.nRows);
  const int maxBX = Lv.w - BORDER, maxBY = Lv.h - BORDER;
  const uint8_t *img;
  int pitch;
  if (l == 0) { img = img0 + (long long)b * imgStride; pitch = pitch0; }
  else { img = pyr + (long long)b * K.pyrStride + Lv.off; pitch = Lv.pitch; }
  // the window is staged with tile column 0 = image column xa (4-byte aligned when the image allows it)
  const bool aligned = ((reinterpret_cast<uintptr_t>(img) | (uintptr_t)pitch) & 3) == 0;
  constexpr int tp = TP;  // tile pitch
  // fixed lane -> (row within a group, dword) mapping: the global offsets and the LDS indices are constants per lane.
  // ALL loads of a tile are issued together (rows past the window are clamped, their data is dropped): a load -> wait
  // -> write loop serialises one memory latency per RPI rows.
  constexpr int DW = TP / 4, RPI = 64 / DW;  // dwords per tile row, rows per iteration
  constexpr int NIT1 = (40 + RPI - 1) / RPI, NIT2 = (FAST_MAX_TILE + RPI - 1) / RPI - NIT1;  // usual windows: <= 40 rows
  const int r0 = lane / DW, dwc = lane - r0 * DW;
  struct Geo { int x0, y0, x1, y1, cw, ch, xa, ox, wpr; bool ok; };  // wave-uniform
  auto geo = [&](int ci, int cj) {  // cell (row ci, column cj) of the level
    Geo g;
    g.x0 = BORDER + cj * Lv.wCell; g.y0 = BORDER + ci * Lv.hCell;
    g.x1 = min(g.x0 + Lv.wCell + 6, maxBX); g.y1 = min(g.y0 + Lv.hCell + 6, maxBY);
    g.cw = g.x1 - g.x0; g.ch = g.y1 - g.y0;
    g.ok = !(g.y0 >= maxBY - 3 || g.x0 >= maxBX - 6) && g.cw >= 7 && g.ch >= 7;  // ORBextractor.cc:794,802
    g.xa = aligned ? (g.x0 & ~3) : g.x0;
    g.ox = g.x0 - g.xa;
    g.wpr = ((g.x1 - g.xa) + 3) >> 2;  // dwords per window row (<= DW)
    return g;
  };
  uint32_t v[NIT1];
  auto issue = [&](const Geo &g) {  // the first NIT1 * RPI rows of the window -> registers
    if (r0 < RPI && dwc < g.wpr) {
      const uint8_t *src = img + (long long)g.y0 * pitch + g.xa;  // wave-uniform base + 32-bit lane offsets
      const uint32_t lo = (uint32_t)dwc * 4u;
      int r0v = r0;
      asm volatile("" : "+v"(r0v));  // keeps the NIT1 row numbers from being hoisted out of the cell loop into NIT1 registers
#pragma unroll
      for (int it = 0; it < NIT1; it++)
        v[it] = *reinterpret_cast<const uint32_t *>(src + ((uint32_t)__mul24(min(r0v + it * RPI, g.ch - 1), pitch) + lo));
    }
  };
  auto stage = [&](const Geo &g) {  // registers (+ the rows of a tall window) -> LDS tile, score tile cleared
    if (aligned) {
      if (r0 < RPI && dwc < g.wpr) {
        uint32_t *dst = reinterpret_cast<uint32_t *>(&tile[dwc * 4]);
#pragma unroll
        for (int it = 0; it < NIT1; it++) dst[(r0 + it * RPI) * DW] = v[it];  // rows >= ch: copies of the last row, masked later
        // (NIT1 * RPI rows always fit tile + score tile -- the host sizes fastTileBytes for it -- and the score tile is cleared
        // below, after these writes; unconditional stores also let the compiler count the loads it has consumed)
        if (g.ch > NIT1 * RPI) {  // tall cells of small levels
          const uint8_t *src = img + (long long)g.y0 * pitch + g.xa;
          int r0t = r0, dwct = dwc;
          asm volatile("" : "+v"(r0t), "+v"(dwct));  // nothing of this rare path is precomputed (and spilled) outside the cell loop
          const uint32_t lo = (uint32_t)dwct * 4u;
          uint32_t w[NIT2];
#pragma unroll
          for (int it = 0; it < NIT2; it++) w[it] = *reinterpret_cast<const uint32_t *>(src + ((uint32_t)__mul24(min(r0t + (NIT1 + it) * RPI, g.ch - 1), pitch) + lo));
          uint32_t *dstt = reinterpret_cast<uint32_t *>(&tile[dwct * 4]);
#pragma unroll
          for (int it = 0; it < NIT2; it++)
            if (r0t + (NIT1 + it) * RPI < g.ch) dstt[(r0t + (NIT1 + it) * RPI) * DW] = w[it];
        }
      }
    } else {
      for (int i = lane; i < g.cw * g.ch; i += 64) {
        const int yy = i / g.cw, xx = i - yy * g.cw;
        tile[yy * tp + xx] = img[(long long)(g.y0 + yy) * pitch + g.x0 + xx];
      }
    }
    {  // score tile <- 0, 16 bytes per lane and store (the tile sizes are multiples of 16, the carve is 16-byte aligned)
      typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
      const u32x4 z = {0u, 0u, 0u, 0u};
      const int n16 = (tp * g.ch + 15) / 16;
      if (lane < n16) reinterpret_cast<u32x4 *>(sc)[lane] = z;  // the usual 44 x <= 46 bytes: two stores, no loop
      if (lane + 64 < n16) reinterpret_cast<u32x4 *>(sc)[lane + 64] = z;
      if (n16 > 128)
        for (int i = lane + 128; i < n16; i += 64) reinterpret_cast<u32x4 *>(sc)[i] = z;
    }
    __syncthreads();
  };
  uint32_t validKept = 0u;  // the sweep's valid-pixel mask of the last cell size seen (process)
  int validKey = -1;
  auto process = [&](const Geo &g, int cell) {
    const int x0 = g.x0, y0 = g.y0, cw = g.cw, ch = g.ch, ox = g.ox;
    if (K.dbg == 1) { if (tile[lane] == 255 && sc[lane] == 7) cand[0] = 1; return; }
    const int dwid = cw - 6, dhei = ch - 6, npix = dwid * dhei;
    // ---- ONE sweep over the cell: every pixel that can be a corner (necessary compass test) becomes a bit of a
    //      per-lane mask, one mask for iniThFAST (A) and one for minThFAST only (B); the A masks are expanded into a
    //      list of tile offsets (front of s_list), the B masks only if the cell has to be redone at minThFAST (back of
    //      s_list, growing down).  No cross-lane work and no scalar bookkeeping inside the sweep.
    const int cap = K.fastMaxPix;
    // Lanes map to (row-pair, column): 32 columns x 2 row pairs per iteration for the usual <= 32 px wide cells, 64 x 1
    // otherwise, so the LDS offset advances by a constant and no index division is needed.  A lane tests TWO vertically
    // adjacent pixels at once in packed 16-bit arithmetic.  With the compass pixels n0 (below), n4 (right), n8 (above),
    // n12 (left) of centre v, "two ADJACENT compass points brighter than v + T" is "one of {n0, n8} AND one of {n4, n12}"
    // (every such pair is adjacent), i.e. v + T < B with B = min(max(n0, n8), max(n4, n12)); likewise two darker ones are
    // v - T > A with A = max(min(n0, n8), min(n4, n12)).  strength = max(v - A, B - v) > T is the test for ANY threshold:
    // 9 packed operations for two pixels; the sign bits of T - strength shift into the masks (3 operations per threshold).
    (void)npix;
    typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));
    typedef short s16x2 __attribute__((ext_vector_type(2)));
    typedef __attribute__((address_space(3))) unsigned short lds_u16;
    const bool two = dwid <= 32;  // two row pairs per iteration
    const int G = two ? 32 : 64, ppi = two ? 2 : 1;
    const int sxx = lane & (G - 1);
    const bool hiHalf = two && lane >= 32, colok = sxx < dwid;
    const int tp3 = 3 * tp, sstep = __builtin_amdgcn_readfirstlane(2 * ppi * tp);  // scalar: products with it stay off the quarter-rate 32-bit vector multiply
    const int rLane = hiHalf ? 2 : 0;                        // first row of the lane
    const int off = (3 + rLane) * tp + ox + 3 + sxx;         // its upper pixel
    const uint32_t listBase = (uint32_t)(uintptr_t)s_list;  // LDS byte address
    uint32_t baseA = listBase, baseB = listBase + 2u * (uint32_t)(cap - 1);  // next free slot: A grows up, B down
    const u16x2 iniV = {(unsigned short)K.iniTh, (unsigned short)K.iniTh}, minV = {(unsigned short)K.minTh, (unsigned short)K.minTh};
    constexpr int Cc = 3 * TP + 3;  // the centre (upper pixel of the pair) seen from the lane's lowest tap
    // a mask word: bit p of the low half = upper pixel of chunk iteration p - (16 - n), high half = lower pixel
    auto expand = [&](uint32_t W, uint32_t &base, const bool up, int cj) {
      const int cnt = __popc(W);
      const int incl = wave_incl_scan_dpp(cnt);
      const int total = __builtin_amdgcn_readlane(incl, 63);
      if (total == 0) return;  // wave-uniform
      const uint32_t e = 2u * (uint32_t)(incl - cnt);
      uint32_t a = up ? base + e : base - e;
      while (W != 0u) {  // one listed pixel per lane and trip
        const int bpos = __builtin_ctz(W);
        W &= W - 1u;
        // bit p of the low half = iteration p (rows advance by sstep), high half = the lower pixel of the pair (+ tp):
        // cj + (p & 15) * sstep + (p >> 4) * tp as two multiply-adds
        int offp;  // (in assembly: the compiler turns the 24-bit products into quarter-rate 32-bit multiplies here)
        asm("v_mad_u32_u24 %0, %1, %2, %3" : "=v"(offp) : "v"(bpos), "s"(sstep), "v"(cj));
        asm("v_mad_i32_i24 %0, %1, %2, %0" : "+v"(offp) : "v"(bpos >> 4), "s"(__builtin_amdgcn_readfirstlane(tp - 16 * sstep)));
        *reinterpret_cast<lds_u16 *>((uintptr_t)a) = (unsigned short)offp;
        a = up ? a + 2u : a - 2u;
      }
      base = up ? base + 2u * (uint32_t)total : base - 2u * (uint32_t)total;
    };
    // all ten taps at non-negative constant distances from the lane's lowest tap (3 rows up, 3 columns left is the
    // corner of that box), so that every LDS read is base + immediate
    typedef __attribute__((address_space(3))) const uint8_t lds_cu8;
    const uint32_t tileAddr = (uint32_t)(uintptr_t)tile;
    uint32_t lo = tileAddr + (uint32_t)(off - tp3 - 3);  // LDS byte address of the lane's lowest tap
    const int nIt = __builtin_amdgcn_readfirstlane((dhei + 2 * ppi - 1) >> ppi);  // 2 * ppi = 1 << ppi rows per iteration
    uint32_t WB[2] = {0u, 0u};  // the minThFAST-only masks of the (at most two) chunks of 16 iterations
    for (int it0 = 0, ck = 0; it0 < nIt; it0 += 16, ck++) {
      const int n = __builtin_amdgcn_readfirstlane(min(16, nIt - it0));
      uint32_t accA = 0u, accB = 0u;
      const int cj = (int)(lo - tileAddr) + Cc - (16 - n) * sstep;  // tile offset of bit 0
      for (int j = 0; j < n; j++, lo += (uint32_t)sstep) {
        // rows beyond the cell read inside the LDS tile + score tile; their bits are masked below
        asm volatile("" : "+v"(lo));
        lds_cu8 *c = reinterpret_cast<lds_cu8 *>((uintptr_t)lo);
        const uint32_t b0 = c[Cc], b1 = c[Cc + TP], b2 = c[Cc + 3 * TP], b3 = c[Cc + 4 * TP], b4 = c[Cc - 3 * TP], b5 = c[Cc - 2 * TP];
        const uint32_t b6 = c[Cc + 3], b7 = c[Cc + TP + 3], b8 = c[Cc - 3], b9 = c[Cc + TP - 3];
        __builtin_amdgcn_sched_group_barrier(0x100, 10, 0);  // the ten LDS reads back to back, then the arithmetic
        auto pair = [&](uint32_t x, uint32_t yv) { return __builtin_bit_cast(u16x2, x | (yv << 16)); };
        const u16x2 v = pair(b0, b1), n0 = pair(b2, b3), n8 = pair(b4, b5), n4 = pair(b6, b7), n12 = pair(b8, b9);
  #define PMIN(a, b) __builtin_elementwise_min(a, b)
  #define PMAX(a, b) __builtin_elementwise_max(a, b)
        const u16x2 A = PMAX(PMIN(n0, n8), PMIN(n4, n12)), Bv = PMIN(PMAX(n0, n8), PMAX(n4, n12));
        const s16x2 st = PMAX(__builtin_bit_cast(s16x2, (u16x2)(v - A)), __builtin_bit_cast(s16x2, (u16x2)(Bv - v)));
  #undef PMIN
  #undef PMAX
        const u16x2 stu = __builtin_bit_cast(u16x2, st);
        const uint32_t sA = __builtin_bit_cast(uint32_t, (u16x2)(iniV - stu)), sB = __builtin_bit_cast(uint32_t, (u16x2)(minV - stu));
        accA = (accA >> 1) | (sA & 0x80008000u);  // T - strength < 0  <=>  strength > T
        accB = (accB >> 1) | (sB & 0x80008000u);
      }
      // valid bits of this lane: its column inside the cell, its rows inside the cell (kept from the previous cell of the
      // wave when that had the same size, i.e. nearly always)
      uint32_t valid;
      const int key = (dwid << 8) | dhei;
      if (it0 == 0 && key == validKey) {
        valid = validKept;
      } else {
        const int avail = dhei - (it0 * 2 * ppi + rLane);  // rows from the lane's first upper pixel of the chunk to the cell's end
        const int nU = min(max((avail + 2 * ppi - 1) >> ppi, 0), n), nL = min(max((avail + 2 * ppi - 2) >> ppi, 0), n);
        const uint32_t sh = (uint32_t)(16 - n);
        valid = colok ? ((((1u << nU) - 1u) << sh) | (((1u << nL) - 1u) << (sh + 16u))) : 0u;
        if (it0 == 0) { validKept = valid; validKey = key; }
      }
      accA &= valid;
      WB[ck] = accB & valid & ~accA;
      expand(accA, baseA, true, cj);
    }
    const int nlA = (int)((baseA - listBase) >> 1);
    int nlB = 0;
    __syncthreads();
    FAST_TICK(2)  // sweep
      if (K.dbg == 4) { if (nlA + (int)WB[0] == 123456) cand[0] = 1; return; }
    for (int pass = 0; pass < 2; pass++) {
      const int T = pass == 0 ? K.iniTh : K.minTh;
      // ---- scores, stored RAW (the score does not depend on the threshold; clamped at 0): pass 0 scores list A, pass 1
      //      (cells without a corner at iniThFAST) only list B.  The NMS asks "score >= T" of the centre only: a neighbour
      //      below T loses against the centre whether it is stored raw or as the 0 cv::FAST keeps for non-corners, and
      //      a pixel in neither list (score < minThFAST) holds 0
      if (pass == 1) {  // the cell is redone at minThFAST: now its B masks become the B list
        // tile offset of bit 0 of a chunk (as in the sweep): the lane's first tap + the centre - (16 - n) iterations
        const int lo0 = off - tp3 - 3 + Cc;
        expand(WB[0], baseB, false, lo0 - (16 - min(16, nIt)) * sstep);
        if (nIt > 16) expand(WB[1], baseB, false, lo0 + 16 * sstep - (16 - min(16, nIt - 16)) * sstep);
        nlB = (int)((listBase + 2u * (uint32_t)(cap - 1) - baseB) >> 1);
        __syncthreads();
      }
      const int nl = pass == 0 ? nlA : nlA + nlB;  // the NMS of pass 1 visits A then B
      for (int i = (pass == 0 ? 0 : nlA) + lane; i < nl; i += 64) {
        const int o = i < nlA ? s_list[i] : s_list[cap - nlB + (i - nlA)];
        int oc = o - 3 * TP - 3;
        asm volatile("" : "+v"(oc));  // keeps the compiler from re-basing the taps on the pixel
        sc[o] = (uint8_t)max(fast_score16<TP>(&tile[oc]), 0);
      }
      const int Tk = max(T, 1);  // a corner has score >= T and score != 0
      __syncthreads();
      if (pass == 0) { FAST_TICK(3) } else { FAST_TICK(6) }  // score
      if (K.dbg == 2) { if (sc[lane] == 255) cand[0] = 1; return; }
      // ---- 3x3 strict non-max suppression over the candidate list (only listed pixels can hold a score >= T);
      //      the rim holds 0
      int total = 0;
      for (int base = 0; base < nl; base += 64) {
        const int i = base + lane;
        bool keep = false;
        uint32_t rec = 0;
        if (i < nl) {
          const int o = i < nlA ? s_list[i] : s_list[cap - nlB + (i - nlA)];
          int oq = o - TP - 1;  // the upper left neighbour: all nine reads at immediate offsets
          asm volatile("" : "+v"(oq));
          const uint8_t *q = &sc[oq] + TP + 1;
          const int sv = q[0];
          if (sv >= Tk) {
            const int mx = max(max(max(max((int)q[-1], (int)q[1]), (int)q[-tp - 1]), max((int)q[-tp], (int)q[-tp + 1])),
                               max(max((int)q[tp - 1], (int)q[tp]), (int)q[tp + 1]));
            keep = sv > mx;
            const int yy = o / tp, xx = o - yy * tp - ox;
            rec = (uint32_t)(x0 + xx) | ((uint32_t)(y0 + yy) << 12) | ((uint32_t)sv << 24);
          }
        }
        const unsigned long long m = __ballot(keep);
        if (keep) s_out[total + (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u))] = rec;
        total += __popcll(m);
      }
      if (pass == 0) { FAST_TICK(4) } else { FAST_TICK(7) }  // NMS
      if (K.dbg == 3) { if (total == 12345) cand[0] = 1; return; }
      if (total == 0) continue;  // wave-uniform: retry with minThFAST
      __syncthreads();
      // Every cell owns a fixed run of slotCap candidate slots and a count (pre-zeroed by the host); k_octree packs the
      // runs of a level.  (One returning atomic per cell on a per-level counter was a third of a wave's life time: the
      // ~900 cells of a level queue up on one address.)
      uint32_t *out = cand + (long long)b * K.candStride + Lv.candBase + (long long)cell * Lv.slotCap;
#pragma unroll 1
      for (int i = lane; i < total; i += 64) out[i] = s_out[i];
      if (lane == 0) cellCount[(long long)b * K.totalCells + Lv.cellBase + cell] = total;
      if (pass == 0) { FAST_TICK(5) } else { FAST_TICK(8) }  // emission
      break;
    }
  };
  // the one division of the wave (its quotient comes out of the vector ALU: readfirstlane keeps what follows scalar)
  int ci = __builtin_amdgcn_readfirstlane(cFirst / Lv.nCols), cj = cFirst - ci * Lv.nCols;
  Geo g = geo(ci, cj);
  if (timed) { if (g.cw + g.ch + g.x0 + g.y0 == 123456789) cand[0] = 1; }  // forces the kernel-argument loads to have landed
  FAST_TICK(9)  // group decode (scalar loads of the level tables)
  if (g.ok && aligned) issue(g);
  for (int cell = cFirst; cell < cEnd; cell++) {
    const bool haveNext = cell + 1 < cEnd;
    if (haveNext) { cj++; if (cj == Lv.nCols) { cj = 0; ci++; } }
    const Geo gn = geo(ci, cj);
    if (g.ok) {
      stage(g);
      if (timed) { if (tile[lane] == 255 && sc[lane] == 7) cand[0] = 1; }  // forces the wait for the staged tile
      FAST_TICK(0)  // wait for the tile + LDS writes
      // Nothing is outstanding here (the tile has landed), but the compiler cannot see that for the lanes that skipped a
      // row: without this explicit vmcnt(0) it protects their registers with vmcnt(0) waits AFTER the prefetch below,
      // i.e. the wave would sit out the next tile's latency before it starts on this one.
      __builtin_amdgcn_s_waitcnt(0x0F70);
      if (haveNext && gn.ok && aligned) issue(gn);  // in flight while this cell is processed
      FAST_TICK(1)  // prefetch issue
      process(g, cell);
      __syncthreads();  // the next tile overwrites the survivors / lists
    } else if (haveNext && gn.ok && aligned) {
      issue(gn);
    }
    g = gn;
  }
  if (TIMED && timed && lane == 0) {
#pragma unroll
    for (int i = 0; i < 10; i++) atomicAdd(&K.timers[i], (unsigned long long)t_acc[i]);
    atomicAdd(&K.timers[10], (unsigned long long)(t_mark - t_start));
    atomicAdd(&K.timers[11], 1ull);
  }
#undef FAST_TICK
}

// ------------------------------------------------------------------------------------------
// DistributeOctTree.  The std::list of the reference is an ordered array here: every node is
// pushed to the FRONT when created, so list order == descending creation order (initial nodes
// last), which is also the "pointer" order used to break ties in the (size,pointer) sort
// (modelled as creation sequence in the oracle).  Each round: count keypoints per child
// quadrant (LDS atomics), rebuild the list with block scans, remap the keypoints.
// ------------------------------------------------------------------------------------------
struct ONode {
  short x0, y0, x1, y1;
  int cnt;
};

__device__ __forceinline__ int quadrant(const ONode &n, int rx, int ry) {
  const int halfX = (int)ceilf((float)(n.x1 - n.x0) / 2), halfY = (int)ceilf((float)(n.y1 - n.y0) / 2);
  const int mx = n.x0 + halfX, my = n.y0 + halfY;
  return (rx < mx ? 0 : 1) + (ry < my ? 0 : 2);  // n1=0 (UL), n2=1 (UR), n3=2 (BL), n4=3 (BR)
}

__device__ __forceinline__ ONode child_of(const ONode &n, int q, int cnt) {
  const int halfX = (int)ceilf((float)(n.x1 - n.x0) / 2), halfY = (int)ceilf((float)(n.y1 - n.y0) / 2);
  const int mx = n.x0 + halfX, my = n.y0 + halfY;
  ONode c;
  c.x0 = (q & 1) ? mx : n.x0;
  c.x1 = (q & 1) ? n.x1 : mx;
  c.y0 = (q & 2) ? my : n.y0;
  c.y1 = (q & 2) ? n.y1 : my;
  c.cnt = cnt;
  return c;
}

// NT threads: 256 when many (image, level) workgroups share the GPU, 1024 for a small batch, where the workgroup of level 0
// (tens of thousands of candidates, ~10 rounds of three passes over them) is the latency of the whole front chain.
template <int NT>
__global__ __launch_bounds__(NT) void k_octree(OrbK K, const uint32_t *__restrict__ cellCand, const int *__restrict__ cellCount,
                                                uint32_t *__restrict__ cand, int *__restrict__ candCount, uint16_t *__restrict__ nodeOf,
                                                uint32_t *__restrict__ lvlOut, int *__restrict__ lvlCount, int maxNodes) {
  extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
  // level-major grid: the workgroups of level 0, the longest by far, are dispatched first and the short ones fill in behind
  const int l = blockIdx.y, b = blockIdx.x, tid = threadIdx.x;
  const LevelInfo &Lv = K.L[l];
  const int M = maxNodes;
  ONode *listA = reinterpret_cast<ONode *>(smem);        // [M]
  ONode *listB = listA + M;                              // [M]
  int *ccnt = reinterpret_cast<int *>(listB + M);        // [M][4] child counts
  unsigned short *cpos = reinterpret_cast<unsigned short *>(ccnt + 4 * M);  // [M][4] new position of child
  unsigned short *npos = cpos + 4 * M;                   // [M] new position of a surviving node
  unsigned short *order = npos + M;                      // [M] final phase: rank -> position
  unsigned char *split = reinterpret_cast<unsigned char *>(order + M);  // [M] node is split this round
  unsigned long long *best = reinterpret_cast<unsigned long long *>(smem + (((size_t)(split + M - smem)) + 7 & ~(size_t)7));
  __shared__ int s_w[NT / 64], s_S, s_nexp, s_jstar;
  // ---- pack the per-cell candidate runs of k_fast into one dense list (order is irrelevant downstream: the quadtree
  //      breaks response ties with an order key derived from x, y).  offs[] aliases the node lists, not yet in use.
  uint32_t *cdw = cand + (long long)b * K.candStride + Lv.candBase;
  int n;
  {
    int *offs = reinterpret_cast<int *>(smem);  // [ncell + 1]
    const int ncell = Lv.nCols * Lv.nRows;
    const int *cc = cellCount + (long long)b * K.totalCells + Lv.cellBase;
    const uint32_t *sp = cellCand + (long long)b * K.candStride + Lv.candBase;
    int run = 0;
    for (int c0 = 0; c0 < ncell; c0 += NT) {
      const int c = c0 + tid;
      const int v = c < ncell ? cc[c] : 0;
      int tot;
      const int ex = block_excl_scan<NT>(v, s_w, &tot);
      if (c < ncell) offs[c] = run + ex;
      run += tot;
      __syncthreads();  // s_w is reused by the next chunk
    }
    n = run;
    if (tid == 0) { offs[ncell] = n; candCount[b * K.nlevels + l] = n; }
    __syncthreads();
    // dense index -> cell by binary search over the offsets; 4 independent copies per thread and step so that the
    // loads are in flight together
    for (int k0 = tid; k0 < n; k0 += NT * 4) {
      uint32_t val[4];
#pragma unroll
      for (int u = 0; u < 4; u++) {
        const int k = min(k0 + u * NT, n - 1);
        int lo = 0, hi = ncell;  // offs[lo] <= k < offs[hi]
        while (hi - lo > 1) {
          const int mid = (lo + hi) >> 1;
          if (offs[mid] <= k) lo = mid; else hi = mid;
        }
        val[u] = sp[(long long)lo * Lv.slotCap + (k - offs[lo])];
      }
#pragma unroll
      for (int u = 0; u < 4; u++)
        if (k0 + u * NT < n) cdw[k0 + u * NT] = val[u];
    }
    __threadfence_block();
    __syncthreads();  // the lists below overwrite offs[]; the dense list is read back by this block only
  }
  const uint32_t *cd = cdw;
  uint16_t *nof = nodeOf + (long long)b * K.candStride + Lv.candBase;
  int *outCount = lvlCount + b * K.nlevels + l;
  uint32_t *out = lvlOut + (long long)b * K.outStride + Lv.outBase;
  if (n == 0 || Lv.nIni < 1) { if (tid == 0) *outCount = 0; return; }
  const int N = Lv.N;
  const int Hh = Lv.h - 2 * BORDER;
  // ---- initial nodes (ORBextractor.cc:543-593)
  for (int i = tid; i < Lv.nIni; i += NT) {
    ONode nd;
    nd.x0 = (short)(int)(Lv.hX * (float)i);
    nd.x1 = (short)(int)(Lv.hX * (float)(i + 1));
    nd.y0 = 0; nd.y1 = (short)Hh; nd.cnt = 0;
    listB[i] = nd;
  }
  __syncthreads();
  for (int k = tid; k < n; k += NT) {
    const int rx = (int)(cd[k] & 0xFFF) - BORDER;
    int idx = (int)((float)rx / Lv.hX);
    idx = min(idx, Lv.nIni - 1);
    nof[k] = (uint16_t)idx;
    atomicAdd(&listB[idx].cnt, 1);
  }
  __syncthreads();
  if (tid == 0) {  // drop empty initial nodes, keep order
    int S = 0;
    for (int i = 0; i < Lv.nIni; i++) {
      npos[i] = (unsigned short)S;
      if (listB[i].cnt > 0) listA[S++] = listB[i];
    }
    s_S = S;
  }
  __syncthreads();
  for (int k = tid; k < n; k += NT) nof[k] = npos[nof[k]];
  __syncthreads();
  ONode *cur = listA, *nxt = listB;
  int S = s_S;
  bool finalPhase = false;
  for (int guard = 0; guard < 64; guard++) {
    const int prevSize = S;
    // ---- count children of every expandable node
    for (int i = tid; i < 4 * S; i += NT) ccnt[i] = 0;
    __syncthreads();
    // candidate passes: 4 candidates per thread and step, all their global loads requested before the first use
    for (int k0 = tid; k0 < n; k0 += NT * 4) {
      int pk[4];
      uint32_t ck[4];
#pragma unroll
      for (int u = 0; u < 4; u++) { const int kk = min(k0 + u * NT, n - 1); pk[u] = nof[kk]; ck[u] = cd[kk]; }
#pragma unroll
      for (int u = 0; u < 4; u++) {
        if (k0 + u * NT >= n) break;
        const int p = pk[u];
        const ONode nd = cur[p];
        if (nd.cnt > 1) {
          const uint32_t c = ck[u];
          atomicAdd(&ccnt[4 * p + quadrant(nd, (int)(c & 0xFFF) - BORDER, (int)((c >> 12) & 0xFFF) - BORDER)], 1);
        }
      }
    }
    __syncthreads();
    // ---- choose the nodes to split
    if (!finalPhase) {
      for (int p = tid; p < S; p += NT) split[p] = cur[p].cnt > 1;
    } else {
      // sort expandable nodes by (count desc, list position asc) == (size, pointer) sort walked
      // from the back (ORBextractor.cc:684-686); split in that order until the list holds N
      for (int p = tid; p < S; p += NT) {
        split[p] = 0;
        const int c = cur[p].cnt;
        if (c > 1) {
          int rank = 0;
          for (int j = 0; j < S; j++) {
            const int cj = cur[j].cnt;
            if (cj > 1 && (cj > c || (cj == c && j < p))) rank++;
          }
          order[rank] = (unsigned short)p;
        }
      }
      __syncthreads();
      if (tid == 0) {
        int sz = S, j = 0;
        const int C = s_nexp;
        for (; j < C; j++) {
          const int p = order[j];
          const int kk = (ccnt[4 * p] > 0) + (ccnt[4 * p + 1] > 0) + (ccnt[4 * p + 2] > 0) + (ccnt[4 * p + 3] > 0);
          sz += kk - 1;
          split[p] = 1;
          if (sz >= N) { j++; break; }
        }
        s_jstar = j;  // number of nodes split
      }
    }
    __syncthreads();
    // ---- new list: children of split nodes go to the front, most recently created first
    // phase 1 : creation order = list order of parents, q = 0..3  -> position = reverse of it
    // final   : creation order = rank order of parents,  q = 0..3
    const int per = (S + NT - 1) / NT;
    const int a0 = min(tid * per, S), a1 = min(a0 + per, S);
    int myKids = 0, myKeep = 0;
    if (!finalPhase) {
      for (int p = a0; p < a1; p++) {
        if (split[p]) myKids += (ccnt[4 * p] > 0) + (ccnt[4 * p + 1] > 0) + (ccnt[4 * p + 2] > 0) + (ccnt[4 * p + 3] > 0);
        else myKeep++;
      }
    } else {
      const int J = s_jstar;
      const int perj = (J + NT - 1) / NT;
      const int r0 = min(tid * perj, J), r1 = min(r0 + perj, J);
      for (int r = r0; r < r1; r++) {
        const int p = order[r];
        myKids += (ccnt[4 * p] > 0) + (ccnt[4 * p + 1] > 0) + (ccnt[4 * p + 2] > 0) + (ccnt[4 * p + 3] > 0);
      }
      for (int p = a0; p < a1; p++) if (!split[p]) myKeep++;
    }
    int totalKids, totalKeep;
    const int kidsBefore = block_excl_scan<NT>(myKids, s_w, &totalKids);
    const int keepBefore = block_excl_scan<NT>(myKeep, s_w, &totalKeep);
    // children created before mine: kidsBefore -> my first child has creation index kidsBefore,
    // list position = totalKids-1-creationIndex
    {
      int ci = kidsBefore, kp = totalKids + keepBefore;
      if (!finalPhase) {
        for (int p = a0; p < a1; p++) {
          if (split[p]) {
            for (int q = 0; q < 4; q++)
              if (ccnt[4 * p + q] > 0) {
                const int pos = totalKids - 1 - ci++;
                cpos[4 * p + q] = (unsigned short)pos;
                nxt[pos] = child_of(cur[p], q, ccnt[4 * p + q]);
              }
          } else {
            npos[p] = (unsigned short)kp;
            nxt[kp++] = cur[p];
          }
        }
      } else {
        const int J = s_jstar;
        const int perj = (J + NT - 1) / NT;
        const int r0 = min(tid * perj, J), r1 = min(r0 + perj, J);
        for (int r = r0; r < r1; r++) {
          const int p = order[r];
          for (int q = 0; q < 4; q++)
            if (ccnt[4 * p + q] > 0) {
              const int pos = totalKids - 1 - ci++;
              cpos[4 * p + q] = (unsigned short)pos;
              nxt[pos] = child_of(cur[p], q, ccnt[4 * p + q]);
            }
        }
        for (int p = a0; p < a1; p++)
          if (!split[p]) { npos[p] = (unsigned short)kp; nxt[kp++] = cur[p]; }
      }
    }
    const int Snew = totalKids + totalKeep;
    __syncthreads();
    // ---- remap keypoints
    for (int k0 = tid; k0 < n; k0 += NT * 4) {
      int pk[4];
      uint32_t ck[4];
#pragma unroll
      for (int u = 0; u < 4; u++) { const int kk = min(k0 + u * NT, n - 1); pk[u] = nof[kk]; ck[u] = cd[kk]; }
#pragma unroll
      for (int u = 0; u < 4; u++) {
        const int k = k0 + u * NT;
        if (k >= n) break;
        const int p = pk[u];
        if (split[p]) {
          const uint32_t c = ck[u];
          nof[k] = cpos[4 * p + quadrant(cur[p], (int)(c & 0xFFF) - BORDER, (int)((c >> 12) & 0xFFF) - BORDER)];
        } else nof[k] = npos[p];
      }
    }
    // nToExpand of the new list = children with more than one keypoint
    int myExp = 0;
    {
      const int pern = (Snew + NT - 1) / NT;
      const int n0 = min(tid * pern, Snew), n1 = min(n0 + pern, Snew);
      for (int p = n0; p < n1; p++) myExp += (p < totalKids && nxt[p].cnt > 1);
    }
    int nToExpand;
    block_excl_scan<NT>(myExp, s_w, &nToExpand);
    if (tid == 0) s_nexp = nToExpand;
    __syncthreads();
    ONode *t = cur; cur = nxt; nxt = t;
    S = Snew;
    // ---- termination (ORBextractor.cc:669-739)
    if (S >= N || S == prevSize) break;
    if (!finalPhase) {
      if (S + nToExpand * 3 > N) finalPhase = true;
    }
  }
  // ---- best keypoint per node: max response, first in vToDistributeKeys order on ties
  for (int p = tid; p < S; p += NT) best[p] = 0ull;
  __syncthreads();
  for (int k0 = tid; k0 < n; k0 += NT * 4) {
    int pk[4];
    uint32_t ck[4];
#pragma unroll
    for (int u = 0; u < 4; u++) { const int kk = min(k0 + u * NT, n - 1); pk[u] = nof[kk]; ck[u] = cd[kk]; }
#pragma unroll
    for (int u = 0; u < 4; u++) {
    const int k = k0 + u * NT;
    if (k >= n) break;
    const uint32_t c = ck[u];
    const int x = c & 0xFFF, y = (c >> 12) & 0xFFF, r = c >> 24;
    const int cellI = (y - EDGE_THRESHOLD) / Lv.hCell, cellJ = (x - EDGE_THRESHOLD) / Lv.wCell;
    const int dwid = min(Lv.wCell, Lv.w - 2 * EDGE_THRESHOLD - cellJ * Lv.wCell);
    const int inCell = (y - EDGE_THRESHOLD - cellI * Lv.hCell) * dwid + (x - EDGE_THRESHOLD - cellJ * Lv.wCell);
    const unsigned orderKey = (unsigned)(cellI * Lv.nCols + cellJ) * 4096u + (unsigned)inCell;  // < 2^32 (cells < 2^20)
    // response (8 bits) | inverted order key (32 bits) | candidate index (24 bits)
    const unsigned long long key = ((unsigned long long)r << 56) | ((unsigned long long)(0xFFFFFFFFu - orderKey) << 24) | (unsigned)k;
    atomicMax(&best[pk[u]], key);
    }
  }
  __syncthreads();
  for (int p = tid; p < S; p += NT) out[p] = cd[(unsigned)(best[p] & 0xFFFFFFull)];
  if (tid == 0) *outCount = S;
}

// ------------------------------------------------------------------------------------------
// GaussianBlur(level, 7x7, sigma 2, BORDER_REFLECT_101) of every pyramid level (ORBextractor.cc:1080), integer
// kernel {18,34,49,55,49,34,18}/256 per pass, (sum + 2^15) >> 16 (oracle gaussian_blur7).  A wave owns a strip of
// 256 columns x BLUR_ROWS rows: a lane filters 4 adjacent pixels and walks down the rows with the last seven
// horizontal results in registers.  Horizontal pass = 2 x v_dot4_u32_u8 per pixel on a 12-byte window.
// ------------------------------------------------------------------------------------------
constexpr int BLUR_ROWS = 32;

__device__ __forceinline__ int reflect101(int i, int n) {
  if (i < 0) i = -i;
  if (i >= n) i = 2 * n - 2 - i;
  return i;
}

__global__ __launch_bounds__(256) void k_blur(OrbK K, const uint8_t *__restrict__ img0, long long imgStride, int pitch0,
                                              const uint8_t *__restrict__ pyr, uint8_t *__restrict__ blur) {
  const int b = blockIdx.y;
  const int strip = __builtin_amdgcn_readfirstlane(blockIdx.x * 4 + (threadIdx.x >> 6));
  const int lane = threadIdx.x & 63;
  if (strip >= K.blurStrips[K.nlevels]) return;
  int l = 0;
  while (strip >= K.blurStrips[l + 1]) l++;
  const LevelInfo &L = K.L[l];
  const int w = L.w, h = L.h;
  const int sidx = strip - K.blurStrips[l];
  const int nsx = (w + 255) >> 8;
  const int sy = sidx / nsx, sx = sidx - sy * nsx;
  const int x0 = sx * 256 + lane * 4;
  const int y0 = sy * BLUR_ROWS;
  if (x0 >= w) return;
  const uint8_t *img;
  int pitch;
  if (l == 0) { img = img0 + (long long)b * imgStride; pitch = pitch0; }
  else { img = pyr + (long long)b * K.pyrStride + L.off; pitch = L.pitch; }
  uint8_t *out = blur + (long long)b * K.blurStride + L.boff;
  // Column groups whose 12-byte window leaves the row (x0 == 0 and the last one or two groups): the window is loaded
  // from a clamped start instead and its bytes are rearranged with two v_perm per dword so that every out-of-row
  // column holds its BORDER_REFLECT_101 partner (which always lies inside the same 12 bytes).
  const bool edge = x0 < 4 || x0 + 8 > w;
  const int base = edge ? max(min(x0 - 4, w - 12), 0) : x0 - 4;
  uint32_t m1[3] = {0, 0, 0}, m2[3] = {0, 0, 0};
  if (edge) {
#pragma unroll
    for (int i = 0; i < 12; i++) {
      const int c = reflect101(min(x0 - 4 + i, w + 2), w);
      const int sidx8 = min(max(c - base, 0), 11);
      const uint32_t lo = sidx8 < 8 ? (uint32_t)sidx8 : 0x0cu, hi = sidx8 < 8 ? 0x0cu : (uint32_t)(sidx8 - 8);
      m1[i >> 2] |= lo << (8 * (i & 3));
      m2[i >> 2] |= hi << (8 * (i & 3));
    }
  }
  const bool anyEdge = __ballot(edge) != 0ull;
  const uint32_t WA = 18u | (34u << 8) | (49u << 16) | (55u << 24), WB = 49u | (34u << 8) | (18u << 16);
  // vertical pass: the horizontal sums are 16-bit (<= 255 * 256), so two successive rows of a pixel share a register and
  // v_dot2_u32_u16 takes two taps at once: pairs[r % 6][k] = row r-1 | row r << 16 of pixel k
  uint32_t pairs[6][4];
  uint32_t prevh[4] = {0u, 0u, 0u, 0u};
  // The source rows run BLUR_PF rows ahead of the arithmetic: a load -> wait -> use per row would serialise one memory
  // latency per row (22 per wave).  Row indices past the image are reflected, so every prefetch address is valid.
  constexpr int BLUR_PF = 4;
  uint32_t q[BLUR_PF][3];
  auto loadRow = [&](int r, uint32_t (&dst)[3]) {
    const int ys = reflect101(min(y0 + r - 3, h + 2), h);
    const uint8_t *row = img + (uint32_t)__mul24(ys, pitch);  // 32-bit offset: 64-bit multiplies are quarter rate
    const uint32_t *pw = reinterpret_cast<const uint32_t *>(row + base);
    dst[0] = pw[0]; dst[1] = pw[1]; dst[2] = pw[2];
  };
#pragma unroll
  for (int r = 0; r < BLUR_PF; r++) loadRow(r, q[r]);
#pragma unroll
  for (int r = 0; r < BLUR_ROWS + 6; r++) {
    const int yo = y0 + r - 6;  // destination row completed by this source row
    if (r >= 6 && yo >= h) break;
    uint32_t w0 = q[r % BLUR_PF][0], w1 = q[r % BLUR_PF][1], w2 = q[r % BLUR_PF][2];
    if (r + BLUR_PF < BLUR_ROWS + 6) loadRow(r + BLUR_PF, q[r % BLUR_PF]);
    if (anyEdge) {
      const uint32_t e0 = __builtin_amdgcn_perm(w1, w0, m1[0]) | __builtin_amdgcn_perm(w2, w2, m2[0]);
      const uint32_t e1 = __builtin_amdgcn_perm(w1, w0, m1[1]) | __builtin_amdgcn_perm(w2, w2, m2[1]);
      const uint32_t e2 = __builtin_amdgcn_perm(w1, w0, m1[2]) | __builtin_amdgcn_perm(w2, w2, m2[2]);
      if (edge) { w0 = e0; w1 = e1; w2 = e2; }
    }
    uint32_t hr[4];
    // taps of pixel j are window bytes 1+j .. 7+j
    hr[0] = __builtin_amdgcn_udot4(__builtin_amdgcn_alignbyte(w2, w1, 1), WB, __builtin_amdgcn_udot4(__builtin_amdgcn_alignbyte(w1, w0, 1), WA, 0u, false), false);
    hr[1] = __builtin_amdgcn_udot4(__builtin_amdgcn_alignbyte(w2, w1, 2), WB, __builtin_amdgcn_udot4(__builtin_amdgcn_alignbyte(w1, w0, 2), WA, 0u, false), false);
    hr[2] = __builtin_amdgcn_udot4(__builtin_amdgcn_alignbyte(w2, w1, 3), WB, __builtin_amdgcn_udot4(__builtin_amdgcn_alignbyte(w1, w0, 3), WA, 0u, false), false);
    hr[3] = __builtin_amdgcn_udot4(w2, WB, __builtin_amdgcn_udot4(w1, WA, 0u, false), false);
    if (r >= 6) {
      typedef unsigned short u16x2b __attribute__((ext_vector_type(2)));
      const u16x2b W01 = {18, 34}, W23 = {49, 55}, W45 = {49, 34};
      uint32_t packed = 0;
#pragma unroll
      for (int k = 0; k < 4; k++) {
        // rows r-6 .. r with weights 18 34 49 55 49 34 18: three two-tap dots on the row pairs + the new row
        uint32_t sum = (uint32_t)__mul24(18, (int)hr[k]) + (1u << 15);
        sum = __builtin_amdgcn_udot2(__builtin_bit_cast(u16x2b, pairs[(r + 1) % 6][k]), W01, sum, false);  // rows r-6, r-5
        sum = __builtin_amdgcn_udot2(__builtin_bit_cast(u16x2b, pairs[(r + 3) % 6][k]), W23, sum, false);  // rows r-4, r-3
        sum = __builtin_amdgcn_udot2(__builtin_bit_cast(u16x2b, pairs[(r + 5) % 6][k]), W45, sum, false);  // rows r-2, r-1
        packed |= (uint32_t)min((int)(sum >> 16), 255) << (8 * k);
      }
      *reinterpret_cast<uint32_t *>(out + (uint32_t)(__mul24(yo, L.pitch) + x0)) = packed;
    }
#pragma unroll
    for (int k = 0; k < 4; k++) { pairs[r % 6][k] = prevh[k] | (hr[k] << 16); prevh[k] = hr[k]; }
  }
}

// ------------------------------------------------------------------------------------------
// One wave per keypoint: orientation (IC_Angle on the raw level), 256 steered BRIEF tests on the blurred level,
// and the final cv::KeyPoint record.  Key points sit >= 19 px inside the level (EDGE_THRESHOLD), so the radius-15
// orientation patch and the radius-18 test footprint never leave the image.
// ------------------------------------------------------------------------------------------
// One wave per key point.  The radius-15 raw patch (orientation) and the radius-18 blurred patch (BRIEF tests) are
// staged into LDS with coalesced dword loads: the 512 rotated sample positions are then LDS gathers instead of
// scattered byte loads that would each cost a cache-line lookup in the vector L1 (measured: the L1 tag rate, not
// ALU or HBM, bounded the direct-gather version).
constexpr int DP_RAW_DW = 9, DP_RAW_ROWS = 31;    // 31 rows x 36 B  (x-15 .. x+15 after dword alignment)
constexpr int DP_BL_DW = 10, DP_BL_ROWS = 37;     // 37 rows x 40 B  (x-18 .. x+18 after dword alignment)

constexpr int DESC_WPB = 4;  // waves per workgroup (they never talk to each other)
constexpr int DESC_KPW = 2;  // consecutive key points per wave, all their patch loads in flight together

// A wave owns DESC_KPW consecutive key points.  What does not depend on the key point (level counts, sampling
// pattern, orientation tables) is fetched once per wave, the DESC_KPW records with one load, and the patch loads of ALL
// of the wave's key points are requested before the first patch is used.  The kernel is bound by the latency of its
// ~70 scattered cache lines per key point (ablation: 37 % record + table fetch, 53 % patch staging + moments, 10 % the
// 256 tests), not by arithmetic.  Measured at B=256 (per step, both images): one key point per wave 0.97 ms; 8 per wave
// with the NEXT patch prefetched during the tests 1.47 ms (the tests are far too short to cover a patch); 2 per wave
// with both patches requested together 0.93 ms and the best overlap with the other streams; 4 per wave the same.
// More resident waves do not help either (amdgpu_waves_per_eu 6: 76 VGPRs, same time; 7: spills, 1.12 ms).
__global__ __launch_bounds__(64 * DESC_WPB) void k_describe(OrbK K, const uint8_t *__restrict__ img0, long long imgStride,
                                                 int pitch0, const uint8_t *__restrict__ pyr,
                                                 const uint8_t *__restrict__ blur, const uint4 *__restrict__ angTab,
                                                 const uint32_t *__restrict__ lvlOut, const int *__restrict__ lvlCount,
                                                 fb_keypoint *__restrict__ kps, uint8_t *__restrict__ desc,
                                                 int32_t *__restrict__ nOut) {
  __shared__ __attribute__((aligned(16))) uint32_t rawp_all[DESC_WPB][DP_RAW_ROWS * DP_RAW_DW + 4];
  __shared__ __attribute__((aligned(16))) uint32_t blp_all[DESC_WPB][DP_BL_ROWS * DP_BL_DW + 2];
  const int b = blockIdx.y, lane = threadIdx.x & 63;
  const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  uint32_t *rawp = rawp_all[wv], *blp = blp_all[wv];
  // XCD-aware mapping (as in k_fast): consecutive workgroups go to different XCDs; give each XCD a contiguous run of
  // key points (neighbours in the quadtree order overlap in the image) so that their patches share lines in one L2
  const int first = (((int)(blockIdx.x & 7) * ((gridDim.x + 7) >> 3) + (int)(blockIdx.x >> 3)) * DESC_WPB + wv) * DESC_KPW;
  const int *cnts = lvlCount + b * K.nlevels;
  int lstart[FB_MAX_LEVELS + 1];
  lstart[0] = 0;
#pragma unroll
  for (int l = 0; l < FB_MAX_LEVELS; l++) lstart[l + 1] = lstart[l] + (l < K.nlevels ? cnts[l] : 0);
  const int nk = min(lstart[FB_MAX_LEVELS], K.capOut);
  if (blockIdx.x == 0 && threadIdx.x == 0) nOut[b] = nk;
  if (first >= nk) return;
  const int nmine = min(DESC_KPW, nk - first);
  // lanes 0..nmine-1 fetch the records of the wave's key points (level = the run of the per-level counts they fall in)
  int kl = 0;
  uint32_t recv;
  {
    const int ki = first + min(lane, nmine - 1);
    int adj = K.L[0].outBase;
#pragma unroll
    for (int l = 1; l < FB_MAX_LEVELS; l++)
      if (l < K.nlevels && ki >= lstart[l]) { kl = l; adj = K.L[l].outBase - lstart[l]; }
    recv = lvlOut[(long long)b * K.outStride + adj + ki];
  }
  float4 pp[4];
#pragma unroll
  for (int t = 0; t < 4; t++) pp[t] = reinterpret_cast<const float4 *>(c_pattern)[lane * 4 + t];
  const uint4 angW = angTab[(lane < 62 ? lane : 0) * 2], angK = angTab[(lane < 62 ? lane : 0) * 2 + 1];
  if (K.dbg == 11) { if ((int)(pp[0].x + pp[1].y + pp[2].z + pp[3].w) + (int)angW.x + (int)angK.y + (int)recv == 1234567) nOut[0] = 1; return; }
  // fixed lane -> (row within a group, dword) mapping: 6 rows x 10 dwords (7 x 9 for the raw patch) per pass, so the
  // global offsets and the LDS indices are constants per lane (32-bit offsets from wave-uniform bases, 24-bit
  // multiplies: 64/32-bit integer multiplies are quarter rate)
  constexpr int NB_IT = (DP_BL_ROWS + 5) / 6, NR_IT = (DP_RAW_ROWS + 6) / 7;
  const int rb0 = (lane * 205) >> 11, dwb = lane - rb0 * DP_BL_DW;   // lane / 10
  const int rr0 = (lane * 57) >> 9, dwr = lane - rr0 * DP_RAW_DW;    // lane / 9
  struct KP {  // wave-uniform description of one key point
    int cx, cy, resp, myl, pitch, bpitch, bxa, box, rxa, rox, patchSize;
    float scale;
    const uint8_t *img, *bbase;
    bool rawAligned;
  };
  auto decode = [&](int j) {
    KP k;
    const uint32_t rec = (uint32_t)__builtin_amdgcn_readlane((int)recv, j);
    k.myl = __builtin_amdgcn_readlane(kl, j);
    k.cx = rec & 0xFFF; k.cy = (rec >> 12) & 0xFFF; k.resp = rec >> 24;
    const LevelInfo &Lv = K.L[k.myl];
    if (k.myl == 0) { k.img = img0 + (long long)b * imgStride; k.pitch = pitch0; }
    else { k.img = pyr + (long long)b * K.pyrStride + Lv.off; k.pitch = Lv.pitch; }
    k.bbase = blur + (long long)b * K.blurStride + Lv.boff;
    k.bpitch = Lv.pitch; k.scale = Lv.scale; k.patchSize = Lv.patchSize;
    k.bxa = (k.cx - 18) & ~3; k.box = (k.cx - 18) - k.bxa;
    k.rawAligned = ((reinterpret_cast<uintptr_t>(k.img) | (uintptr_t)k.pitch) & 3) == 0;
    k.rxa = (k.cx - 15) & ~3;
    k.rox = k.rawAligned ? (k.cx - 15) - k.rxa : 0;
    return k;
  };
  uint32_t vb[DESC_KPW][NB_IT], vr[DESC_KPW][NR_IT];
  // All global loads of both patches are issued together (rows past a patch are clamped and their data dropped; key
  // points sit >= 19 px inside the level, so neither patch leaves the image)
  auto issue = [&](const KP &k, uint32_t (&vb)[NB_IT], uint32_t (&vr)[NR_IT]) {
    {  // row offsets advance by a constant; only the last group can run past the patch and is clamped
      uint32_t off = (uint32_t)(__mul24(k.cy - 18 + rb0, k.bpitch) + k.bxa + min(dwb, DP_BL_DW - 1) * 4);
      const uint32_t step = 6u * (uint32_t)k.bpitch;
#pragma unroll
      for (int it = 0; it < NB_IT - 1; it++, off += step) vb[it] = *reinterpret_cast<const uint32_t *>(k.bbase + off);
      vb[NB_IT - 1] = *reinterpret_cast<const uint32_t *>(k.bbase + (uint32_t)(__mul24(k.cy - 18 + min((NB_IT - 1) * 6 + rb0, DP_BL_ROWS - 1), k.bpitch) + k.bxa + min(dwb, DP_BL_DW - 1) * 4));
    }
    if (k.rawAligned) {
      uint32_t off = (uint32_t)(__mul24(k.cy - 15 + rr0, k.pitch) + k.rxa + min(dwr, DP_RAW_DW - 1) * 4);
      const uint32_t step = 7u * (uint32_t)k.pitch;
#pragma unroll
      for (int it = 0; it < NR_IT - 1; it++, off += step) vr[it] = *reinterpret_cast<const uint32_t *>(k.img + off);
      vr[NR_IT - 1] = *reinterpret_cast<const uint32_t *>(k.img + (uint32_t)(__mul24(k.cy - 15 + min((NR_IT - 1) * 7 + rr0, DP_RAW_ROWS - 1), k.pitch) + k.rxa + min(dwr, DP_RAW_DW - 1) * 4));
    }
  };
  auto stage = [&](const KP &k, const uint32_t (&vb)[NB_IT], const uint32_t (&vr)[NR_IT]) {
    if (lane < 6 * DP_BL_DW) {
#pragma unroll
      for (int it = 0; it < NB_IT; it++)
        if (it * 6 + rb0 < DP_BL_ROWS) blp[it * 6 * DP_BL_DW + lane] = vb[it];
    }
    if (k.rawAligned) {
      if (lane < 7 * DP_RAW_DW) {
#pragma unroll
        for (int it = 0; it < NR_IT; it++)
          if (it * 7 + rr0 < DP_RAW_ROWS) rawp[it * 7 * DP_RAW_DW + lane] = vr[it];
      }
    } else {  // caller's level-0 image with an odd stride: byte loads
      uint8_t *rb = reinterpret_cast<uint8_t *>(rawp);
      for (int i = lane; i < DP_RAW_ROWS * 31; i += 64) {
        const int yy = i / 31, xx = i - yy * 31;
        rb[yy * (DP_RAW_DW * 4) + xx] = k.img[(long long)(k.cy - 15 + yy) * k.pitch + k.cx - 15 + xx];
      }
    }
    // each wave reads back only what it wrote itself: LDS operations of one wave complete in order, so a wave-level
    // fence (no s_barrier across the workgroup) is all that is needed
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  };
  // the patch loads of ALL of the wave's key points are requested before the first patch is used
  KP kpv[DESC_KPW];
#pragma unroll
  for (int j = 0; j < DESC_KPW; j++) {
    kpv[j] = decode(min(j, nmine - 1));
    if (j < nmine) issue(kpv[j], vb[j], vr[j]);
  }
#pragma unroll
  for (int j = 0; j < DESC_KPW; j++) {
    if (j >= nmine) break;
    const KP &cur = kpv[j];
    stage(cur, vb[j], vr[j]);
    // IC_Angle (ORBextractor.cc:77-104): lanes 2*(v+15) and 2*(v+15)+1 sum the left (u < 0) and right (u >= 0) part
    // of row v of the circular patch.  Each half row is 16 bytes read as dwords; |u| weights and the circular mask
    // (|u| <= umax[|v|]) come from a per-lane table (angTab[lane] = 4 weight dwords + 4 mask dwords) and the sums
    // are v_dot4_u32_u8.
    int m10 = 0, m01 = 0;
    if (lane < 62) {
      const int r = lane >> 1, half = lane & 1;
      const int sb = cur.rox + (half ? 15 : 0);  // left half: u = -15..0 at bytes rox..rox+15, right half: u = 0..15 at rox+15..
      const uint32_t *rowd = rawp + r * DP_RAW_DW + (sb >> 2);
      const uint32_t sh = (uint32_t)(sb & 3);
      const uint32_t d0 = rowd[0], d1 = rowd[1], d2 = rowd[2], d3 = rowd[3], d4 = rowd[4];
      const uint32_t q0 = __builtin_amdgcn_alignbyte(d1, d0, sh), q1 = __builtin_amdgcn_alignbyte(d2, d1, sh);
      const uint32_t q2 = __builtin_amdgcn_alignbyte(d3, d2, sh), q3 = __builtin_amdgcn_alignbyte(d4, d3, sh);
      const uint4 W = angW, Km = angK;
      const uint32_t a10 = __builtin_amdgcn_udot4(q0, W.x, __builtin_amdgcn_udot4(q1, W.y, __builtin_amdgcn_udot4(q2, W.z, __builtin_amdgcn_udot4(q3, W.w, 0u, false), false), false), false);
      const uint32_t rs = __builtin_amdgcn_udot4(q0, Km.x, __builtin_amdgcn_udot4(q1, Km.y, __builtin_amdgcn_udot4(q2, Km.z, __builtin_amdgcn_udot4(q3, Km.w, 0u, false), false), false), false);
      m10 = half ? (int)a10 : -(int)a10;
      m01 = __mul24(r - 15, (int)rs);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { m10 += __shfl_xor(m10, o, 64); m01 += __shfl_xor(m01, o, 64); }
    if (K.dbg == 12) { if (m10 + m01 == 12345678 && (int)(pp[0].x + pp[1].y + pp[2].z + pp[3].w) == 77777) nOut[0] = 1; continue; }
    const float angle = fb_fast_atan2((float)m01, (float)m10);
    // computeOrbDescriptor (ORBextractor.cc:107-147): lane computes tests 4*lane .. 4*lane+3
    const float factorPI = 0x1.1df46ap-6f;
    float sa, ca;
    fb_sincos_f(angle * factorPI, &sa, &ca);
    const float a = ca, bb = sa;
    const uint8_t *centre = reinterpret_cast<const uint8_t *>(blp) + 18 * (DP_BL_DW * 4) + cur.box + 18;
    int nib = 0;
#pragma unroll
    for (int t = 0; t < 4; t++) {
      const float x0 = pp[t].x, y0 = pp[t].y, x1 = pp[t].z, y1 = pp[t].w;
      const int t0 = centre[__mul24(fb_cvround(x0 * bb + y0 * a), DP_BL_DW * 4) + fb_cvround(x0 * a - y0 * bb)];
      const int t1 = centre[__mul24(fb_cvround(x1 * bb + y1 * a), DP_BL_DW * 4) + fb_cvround(x1 * a - y1 * bb)];
      nib |= (t0 < t1) << t;
    }
    // lane 2j holds the low nibble of descriptor byte j, lane 2j+1 the high nibble; lanes 8j assemble dword j
    const int hi = __shfl_down(nib, 1, 64);
    const int byteVal = nib | (hi << 4);
    const int b1 = __shfl_down(byteVal, 2, 64), b2 = __shfl_down(byteVal, 4, 64), b3 = __shfl_down(byteVal, 6, 64);
    const uint32_t dw = (uint32_t)byteVal | ((uint32_t)b1 << 8) | ((uint32_t)b2 << 16) | ((uint32_t)b3 << 24);
    const long long o = (long long)b * K.kpStride + first + j;
    if ((lane & 7) == 0) reinterpret_cast<uint32_t *>(desc + o * 32)[lane >> 3] = dw;
    if (lane == 0) {
      fb_keypoint kp;
      kp.x = (float)cur.cx;
      kp.y = (float)cur.cy;
      if (cur.myl != 0) { kp.x *= cur.scale; kp.y *= cur.scale; }  // ORBextractor.cc:1095-1101
      kp.size = (float)cur.patchSize;
      kp.angle = angle;
      kp.response = (float)cur.resp;
      kp.octave = cur.myl;
      kps[o] = kp;
    }
    // the LDS reads of this key point are ordered before the writes of the next one (same wave, in-order LDS)
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
  }
}

}  // namespace

// ------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------
struct fb_orb {
  fb_orb_params p;
  fb_orb_tables t;
  // k_blur runs on a stream of its own beside k_octree (see fb_orb_extract_batch_dev); created on first use
  hipStream_t sideStream = nullptr;
  hipEvent_t evFork = nullptr, evJoin = nullptr;
  ~fb_orb() {
    if (evFork) (void)hipEventDestroy(evFork);
    if (evJoin) (void)hipEventDestroy(evJoin);
    if (sideStream) (void)hipStreamDestroy(sideStream);
  }
  int kpStride = 0;  // fb_orb_set_output_stride (0 = the capacity)
  // workspace, valid for (w, h, batchCap)
  int w = 0, h = 0, batchCap = 0;
  OrbK K;
  int maxNodes = 0;
  size_t octreeLds = 0;
  fb::DevBuf pyr, blur, cand, cellCand, cellCount, nodeOf, counts, lvlOut, tabs, angTab, timers;
  ResizeTabs rt[FB_MAX_LEVELS];
  bool mergeOK[FB_MAX_LEVELS] = {};  // k_resize_merge: every run of RM_ROWS destination rows spans <= RM_SRC source rows
  bool rowsOK[FB_MAX_LEVELS] = {};  // k_resize_rows' 12-byte window covers every 4-pixel group of the level
  // last call (for fb_orb_get_level)
  const uint8_t *lastImg = nullptr;
  long long lastImgStride = 0;
  int lastPitch0 = 0;
  fb::DevBuf ownedImg;
};

namespace {

void make_tables(const fb_orb_params &p, fb_orb_tables &t) {  // ORBextractor.cc:410-470
  memset(&t, 0, sizeof(t));
  const int nl = p.nlevels;
  t.scale_factor[0] = 1.0f;
  t.level_sigma2[0] = 1.0f;
  for (int i = 1; i < nl; i++) {
    t.scale_factor[i] = t.scale_factor[i - 1] * p.scale_factor;
    t.level_sigma2[i] = t.scale_factor[i] * t.scale_factor[i];
  }
  for (int i = 0; i < nl; i++) {
    t.inv_scale_factor[i] = 1.0f / t.scale_factor[i];
    t.inv_level_sigma2[i] = 1.0f / t.level_sigma2[i];
  }
  float factor = 1.0f / p.scale_factor;
  float nDesired = p.nfeatures * (1 - factor) / (1 - (float)pow((double)factor, (double)nl));
  int sum = 0;
  for (int l = 0; l < nl - 1; l++) {
    t.features_per_level[l] = fb_cvround(nDesired);
    sum += t.features_per_level[l];
    nDesired *= factor;
  }
  t.features_per_level[nl - 1] = std::max(p.nfeatures - sum, 0);
  int umax[HALF_PATCH + 2] = {0};
  int v, v0;
  const int vmax = fb_cvfloor(HALF_PATCH * sqrtf(2.f) / 2 + 1);
  const int vmin = fb_cvceil(HALF_PATCH * sqrtf(2.f) / 2);
  const double hp2 = HALF_PATCH * HALF_PATCH;
  for (v = 0; v <= vmax; ++v) umax[v] = fb_cvround_d(sqrt(hp2 - v * v));
  for (v = HALF_PATCH, v0 = 0; v >= vmin; --v) {
    while (umax[v0] == umax[v0 + 1]) ++v0;
    umax[v] = v0;
    ++v0;
  }
  for (int i = 0; i <= HALF_PATCH; i++) t.umax[i] = umax[i];
}

int capacity_of(const fb_orb_params &p) { return p.nfeatures + 8 * p.nlevels; }

// (re)build the workspace for images of w x h, up to `batch` per call
int prepare(fb_orb *o, int w, int h, int batch) {
  if (o->w == w && o->h == h && batch <= o->batchCap) return FB_OK;
  const fb_orb_params &p = o->p;
  OrbK &K = o->K;
  memset(&K, 0, sizeof(K));
  K.nlevels = p.nlevels;
  K.iniTh = p.ini_th_fast;
  K.minTh = p.min_th_fast;
  K.capOut = capacity_of(p);
  K.kpStride = o->kpStride > 0 ? o->kpStride : K.capOut;
  for (int i = 0; i < 16; i++) K.umax[i] = o->t.umax[i];
  long long pyrOff = 0, candOff = 0, blurOff = 0;
  int cells = 0, outOff = 0, maxNodes = 0, strips = 0;
  std::vector<int> xofs, yofs;
  std::vector<short> ialpha, ibeta;
  std::vector<size_t> tabOff(p.nlevels * 4, 0);
  std::vector<uint8_t> tabBytes;
  auto append = [&](const void *src, size_t n) {
    size_t off = (tabBytes.size() + 15) & ~(size_t)15;
    tabBytes.resize(off + n);
    memcpy(tabBytes.data() + off, src, n);
    return off;
  };
  int pw = w, ph = h;
  for (int l = 0; l < p.nlevels; l++) {
    LevelInfo &L = K.L[l];
    const float scale = o->t.inv_scale_factor[l];
    L.w = fb_cvround((float)w * scale);   // ORBextractor.cc:1112
    L.h = fb_cvround((float)h * scale);
    if (l == 0) { L.w = w; L.h = h; }
    if (L.w < 1 || L.h < 1) {  // cv::resize to an empty Size asserts in the reference as well
      fb::set_error("pyramid level %d of a %dx%d image is empty (scale factor %.3f, %d levels)", l, w, h, (double)p.scale_factor, p.nlevels);
      return FB_ERR_ARG;
    }
    if (L.w >= 4096 || L.h >= 4096) { fb::set_error("image too large (level %d is %dx%d, limit 4095)", l, L.w, L.h); return FB_ERR_ARG; }
    L.pitch = (L.w + 63) & ~63;
    L.off = pyrOff;
    if (l > 0) pyrOff += (long long)L.pitch * L.h;
    L.boff = blurOff;
    blurOff += (long long)L.pitch * L.h;
    K.blurStrips[l] = strips;
    // levels too small to hold a key point (19 px border on every side) are never sampled: no blur strips for them
    if (L.w >= 2 * EDGE_THRESHOLD && L.h >= 2 * EDGE_THRESHOLD) strips += ((L.w + 255) / 256) * ((L.h + BLUR_ROWS - 1) / BLUR_ROWS);
    const int maxBX = L.w - BORDER, maxBY = L.h - BORDER;
    const float width = (float)(maxBX - BORDER), height = (float)(maxBY - BORDER);
    L.nCols = (int)(width / 30.f);
    L.nRows = (int)(height / 30.f);
    if (L.nCols <= 0 || L.nRows <= 0) { L.nCols = L.nRows = 0; L.wCell = L.hCell = 1; }
    else { L.wCell = (int)ceilf(width / L.nCols); L.hCell = (int)ceilf(height / L.nRows); }
    if (L.wCell + 6 > FAST_MAX_TILE - 3 || L.hCell + 6 > FAST_MAX_TILE) { fb::set_error("FAST cell %dx%d exceeds the LDS tile", L.wCell, L.hCell); return FB_ERR_CAPACITY; }
    L.cellBase = cells;
    cells += L.nCols * L.nRows;
    if (L.nCols * L.nRows >= (1 << 20)) { fb::set_error("level %d has more than 2^20 FAST cells", l); return FB_ERR_CAPACITY; }
    L.N = o->t.features_per_level[l];
    // NMS worst case: one survivor per 2x2 block of each cell's detection area
    const int dw = std::max(L.w - 2 * EDGE_THRESHOLD, 0), dh = std::max(L.h - 2 * EDGE_THRESHOLD, 0);
    L.candBase = candOff;
    L.slotCap = ((L.wCell + 1) / 2) * ((L.hCell + 1) / 2);  // strict 3x3 maxima: at most one per 2x2 pixels
    L.candCap = L.nCols * L.nRows * L.slotCap + 16;
    (void)dw; (void)dh;
    if (L.candCap >= (1 << 24)) { fb::set_error("level %d: more than 2^24 FAST candidates possible", l); return FB_ERR_CAPACITY; }
    candOff += (L.candCap + 3) & ~3;
    const int Wr = maxBX - BORDER, Hr = maxBY - BORDER;
    L.nIni = (Hr > 0) ? (int)roundf((float)Wr / Hr) : 0;   // ORBextractor.cc:542
    L.hX = L.nIni > 0 ? (float)Wr / L.nIni : 1.f;
    L.outBase = outOff;
    L.outCap = std::max(L.N + 3, 4 * L.nIni) + 4;
    outOff += L.outCap;
    maxNodes = std::max(maxNodes, L.outCap + 4);
    L.scale = o->t.scale_factor[l];
    L.patchSize = (int)(PATCH_SIZE * o->t.scale_factor[l]);
    if (l > 0) {  // resize tables (oracle resize_linear_u8)
      const int sw = pw, sh = ph, dwd = L.w, dhd = L.h;
      const double inv_scale_x = (double)dwd / sw, inv_scale_y = (double)dhd / sh;
      const double scale_x = 1. / inv_scale_x, scale_y = 1. / inv_scale_y;
      xofs.assign(dwd, 0); ialpha.assign(dwd * 2, 0); yofs.assign(dhd, 0); ibeta.assign(dhd * 2, 0);
      for (int dx = 0; dx < dwd; dx++) {
        float fx = (float)((dx + 0.5) * scale_x - 0.5);
        int sx = fb_cvfloor(fx);
        fx -= sx;
        if (sx < 0) { fx = 0; sx = 0; }
        if (sx >= sw - 1) { fx = 0; sx = sw - 1; }
        xofs[dx] = sx;
        ialpha[dx * 2] = (short)fb_cvround((1.f - fx) * 2048.f);
        ialpha[dx * 2 + 1] = (short)fb_cvround(fx * 2048.f);
      }
      for (int dy = 0; dy < dhd; dy++) {
        float fy = (float)((dy + 0.5) * scale_y - 0.5);
        int sy = fb_cvfloor(fy);
        fy -= sy;
        yofs[dy] = sy;
        ibeta[dy * 2] = (short)fb_cvround((1.f - fy) * 2048.f);
        ibeta[dy * 2 + 1] = (short)fb_cvround(fy * 2048.f);
      }
      bool ok = true;
      // the taps of a 4-pixel group: inside the 12-byte window, and within 8 bytes of the group's first tap (scale <= 2)
      for (int dx4 = 0; dx4 < dwd; dx4 += 4) {
        const int last = xofs[std::min(dx4 + 3, dwd - 1)] + 1;
        ok = ok && (last - (xofs[dx4] & ~3)) <= 11 && (last - xofs[dx4]) <= 7;
      }
      o->rowsOK[l] = ok;
      {
        bool okm = ok;
        for (int y0 = 0; y0 < dhd && okm; y0 += RM_ROWS) {
          const int y1 = std::min(y0 + RM_ROWS, dhd) - 1;
          okm = yofs[y0] >= 0 && yofs[y1] + 1 - yofs[y0] <= RM_SRC - 1;
          for (int y = y0; y < y1 && okm; y++) okm = yofs[y + 1] > yofs[y];  // a source row closes at most one destination row
        }
        o->mergeOK[l] = okm;
      }
      tabOff[l * 4 + 0] = append(xofs.data(), xofs.size() * 4);
      tabOff[l * 4 + 1] = append(ialpha.data(), ialpha.size() * 2);
      tabOff[l * 4 + 2] = append(yofs.data(), yofs.size() * 4);
      tabOff[l * 4 + 3] = append(ibeta.data(), ibeta.size() * 2);
    }
    pw = L.w; ph = L.h;
  }
  K.totalCells = cells;
  {
    int gsum = 0;
    for (int l = 0; l <= FB_MAX_LEVELS; l++) {
      K.grpBase[l] = gsum;
      if (l < p.nlevels) gsum += (K.L[l].nCols * K.L[l].nRows + FAST_CPW - 1) / FAST_CPW;
    }
    K.totalGroups = gsum;
  }
  for (int l = 0; l <= FB_MAX_LEVELS; l++) K.cellBase[l] = l < p.nlevels ? K.L[l].cellBase : cells;
  K.dbg = getenv("FB_FAST_DBG") ? atoi(getenv("FB_FAST_DBG")) : 0;
  {
    int tileB = 0, maxOut = 0, maxPix = 0, tpNeed = 0;
    for (int l = 0; l < p.nlevels; l++) {
      const LevelInfo &L = K.L[l];
      if (L.nCols * L.nRows == 0) continue;
      const int tpMax = (L.wCell + 6 + 3 + 3) & ~3, chMax = L.hCell + 6;
      tpNeed = std::max(tpNeed, tpMax);
      tileB = std::max(tileB, chMax);
      maxOut = std::max(maxOut, ((L.wCell + 1) / 2) * ((L.hCell + 1) / 2));
      maxPix = std::max(maxPix, L.wCell * L.hCell);
    }
    K.fastTP = tpNeed <= 44 ? 44 : tpNeed <= 56 ? 56 : FAST_MAX_TILE;
    K.fastTileBytes = (std::max(K.fastTP * tileB, 4 * ((maxOut + 3) & ~3)) + 15) & ~15;
    K.fastTileBytes = std::max(K.fastTileBytes, (21 * K.fastTP + 15) & ~15);  // k_fast stages 40 / 42 rows unconditionally into tile + score tile
    K.fastMaxOut = (maxOut + 3) & ~3;
    K.fastMaxPix = (maxPix + 7) & ~7;
  }
  K.pyrStride = (pyrOff + 255) & ~255ll;
  K.blurStride = (blurOff + 255) & ~255ll;
  for (int l = p.nlevels; l <= FB_MAX_LEVELS; l++) K.blurStrips[l] = strips;
  K.candStride = candOff;
  K.outStride = outOff;
  o->maxNodes = maxNodes;
  // LDS of k_octree: 2 lists + ccnt + cpos + npos + order + split + best
  const size_t M = maxNodes;
  size_t lds = 2 * M * sizeof(ONode) + 4 * M * 4 + 4 * M * 2 + M * 2 + M * 2 + M;
  lds = ((lds + 7) & ~(size_t)7) + M * 8 + 16;
  {
    size_t maxCells = 0;
    for (int l = 0; l < p.nlevels; l++) maxCells = std::max(maxCells, (size_t)K.L[l].nCols * K.L[l].nRows);
    lds = std::max(lds, (maxCells + 1) * 4 + 16);  // the per-cell offsets of the packing prologue alias the node lists
  }
  // (512 B of the 160 KB are left to the kernel's static LDS: scan scratch and a few scalars; with the dynamic part alone
  // at the limit hipFuncSetAttribute refuses the size, a HIP error instead of this documented refusal)
  if (lds > 160 * 1024 - 512) { fb::set_error("nfeatures too large for the LDS quadtree (%zu B)", lds); return FB_ERR_CAPACITY; }
  o->octreeLds = lds;
  FB_TRY(o->tabs.upload(tabBytes.data(), tabBytes.size()));
  for (int l = 1; l < p.nlevels; l++) {
    const uint8_t *base = o->tabs.as<uint8_t>();
    o->rt[l].xofs = reinterpret_cast<const int *>(base + tabOff[l * 4 + 0]);
    o->rt[l].ialpha = reinterpret_cast<const short *>(base + tabOff[l * 4 + 1]);
    o->rt[l].yofs = reinterpret_cast<const int *>(base + tabOff[l * 4 + 2]);
    o->rt[l].ibeta = reinterpret_cast<const short *>(base + tabOff[l * 4 + 3]);
  }
  const size_t B = batch;
  {  // IC_Angle tables of k_describe: lane 2r+half owns half of patch row v = r-15; per byte j of its 16-byte span the
     // weight |u| and the inclusion mask |u| <= umax[|v|] (left half: u = j-15, u = 0 belongs to the right half)
    std::vector<uint32_t> tab(64 * 8, 0);
    for (int lane = 0; lane < 62; lane++) {
      const int v = (lane >> 1) - 15, half = lane & 1, dmax = K.umax[v < 0 ? -v : v];
      for (int j = 0; j < 16; j++) {
        const int au = half ? j : 15 - j;
        const bool inc = au <= dmax && (half || au != 0);
        if (!inc) continue;
        tab[lane * 8 + (j >> 2)] |= (uint32_t)au << (8 * (j & 3));
        tab[lane * 8 + 4 + (j >> 2)] |= 1u << (8 * (j & 3));
      }
    }
    FB_TRY(o->angTab.upload(tab.data(), tab.size() * 4));
  }
  FB_TRY(o->pyr.alloc(B * K.pyrStride + 256));
  FB_TRY(o->blur.alloc(B * K.blurStride + 256));
  FB_TRY(o->cand.alloc(B * K.candStride * 4 + 16));
  FB_TRY(o->cellCand.alloc(B * K.candStride * 4 + 16));
  FB_TRY(o->cellCount.alloc(B * (size_t)K.totalCells * 4 + 16));
  FB_TRY(o->nodeOf.alloc(B * K.candStride * 2 + 16));
  FB_TRY(o->counts.alloc(B * p.nlevels * 4 * 2));  // candCount | lvlCount
  FB_TRY(o->timers.alloc(16 * 8));
  FB_HIP(hipMemset(o->timers.p, 0, 16 * 8));
  K.timers = o->timers.as<unsigned long long>();
  FB_TRY(o->lvlOut.alloc(B * K.outStride * 4 + 16));
  o->w = w; o->h = h; o->batchCap = batch;
  return FB_OK;
}

}  // namespace

extern "C" {

int fb_orb_capacity(const fb_orb_params *params) {
  if (!params) return FB_ERR_ARG;
  return capacity_of(*params);
}

int fb_orb_set_output_stride(fb_orb *h, int kp_stride) {
  FB_ARG(h && (kp_stride == 0 || kp_stride >= capacity_of(h->p)));
  h->kpStride = kp_stride;
  h->K.kpStride = kp_stride > 0 ? kp_stride : capacity_of(h->p);  // (the workspace itself does not depend on it)
  return FB_OK;
}

int fb_orb_create(const fb_orb_params *params, fb_orb **out) {
  FB_ARG(params && out);
  FB_ARG(params->nlevels >= 1 && params->nlevels <= FB_MAX_LEVELS);
  FB_ARG(params->nfeatures >= 1 && params->nfeatures <= 60000);
  FB_ARG(params->scale_factor > 1.0f);
  FB_ARG(params->min_th_fast >= 1 && params->ini_th_fast >= params->min_th_fast && params->ini_th_fast <= 254);
  fb_orb *o = new fb_orb();
  o->p = *params;
  make_tables(o->p, o->t);
  *out = o;
  return FB_OK;
}

void fb_orb_destroy(fb_orb *h) { delete h; }

int fb_orb_get_tables(const fb_orb *h, fb_orb_tables *out) {
  FB_ARG(h && out);
  *out = h->t;
  return FB_OK;
}

int fb_orb_extract_batch_dev(fb_orb *o, const uint8_t *d_images, int batch, int width, int height, int stride,
                             size_t image_stride, fb_keypoint *d_keypoints, uint8_t *d_descriptors, int32_t *d_n,
                             void *stream) {
  FB_TRY(fb::check_device());
  FB_ARG(o && d_images && d_keypoints && d_descriptors && d_n);
  FB_ARG(batch >= 1 && width >= 2 * EDGE_THRESHOLD + 7 && height >= 2 * EDGE_THRESHOLD + 7 && stride >= width);
  FB_TRY(prepare(o, width, height, batch));
  hipStream_t s = fb::as_stream(stream);
  const OrbK &K = o->K;
  const int nl = K.nlevels;
  int *candCount = o->counts.as<int>();
  int *lvlCount = candCount + (size_t)o->batchCap * nl;
  FB_HIP(hipMemsetAsync(o->cellCount.p, 0, (size_t)batch * K.totalCells * 4, s));
  // pyramid
  for (int l = 1; l < nl; l++) {
    const LevelInfo &D = K.L[l], &S = K.L[l - 1];
    const uint8_t *src = (l == 1) ? d_images : o->pyr.as<uint8_t>() + S.off;
    const long long sstr = (l == 1) ? (long long)image_stride : K.pyrStride;
    const int spitch = (l == 1) ? stride : S.pitch;
    dim3 blk(64, 4), grd((D.pitch / 4 + 63) / 64, (D.h + 3) / 4, batch);
    fb::ProfScope prof_(fb::P_RESIZE, s);
    const bool rows = o->rowsOK[l] && spitch >= 12 && ((reinterpret_cast<uintptr_t>(src) | (uintptr_t)spitch | (uintptr_t)sstr) & 3) == 0;
    if (rows && o->mergeOK[l]) {
      dim3 grdM((D.pitch / 4 + 63) / 64, (D.h + 4 * RM_ROWS - 1) / (4 * RM_ROWS), batch);
      k_resize_merge<<<grdM, blk, 0, s>>>(src, sstr, S.w, S.h, spitch, o->pyr.as<uint8_t>() + D.off, K.pyrStride, D.w, D.h, D.pitch, o->rt[l]);
    } else if (rows) {
      dim3 grdR((D.pitch / 4 + 63) / 64, (D.h + 4 * RESIZE_ROWS - 1) / (4 * RESIZE_ROWS), batch);
      k_resize_rows<<<grdR, blk, 0, s>>>(src, sstr, S.w, S.h, spitch, o->pyr.as<uint8_t>() + D.off, K.pyrStride, D.w, D.h, D.pitch, o->rt[l]);
    } else {
      k_resize<<<grd, blk, 0, s>>>(src, sstr, S.w, S.h, spitch, o->pyr.as<uint8_t>() + D.off, K.pyrStride, D.w, D.h, D.pitch, o->rt[l]);
    }
  }
  if (K.totalCells > 0) {
    fb::ProfScope prof_(K.fastTP == 44 ? fb::P_FAST : K.fastTP == 56 ? fb::P_FAST56 : fb::P_FAST72, s);
    const dim3 grdF((K.totalGroups + 7) / 8 * 8, batch);
    const size_t ldsF = (size_t)2 * K.fastTileBytes + 2 * K.fastMaxPix;
#define FAST_LAUNCH(TP_) { if (K.dbg == 20) k_fast<TP_, true><<<grdF, 64, ldsF, s>>>(K, d_images, (long long)image_stride, stride, o->pyr.as<uint8_t>(), o->cellCand.as<uint32_t>(), o->cellCount.as<int>()); \
                          else k_fast<TP_, false><<<grdF, 64, ldsF, s>>>(K, d_images, (long long)image_stride, stride, o->pyr.as<uint8_t>(), o->cellCand.as<uint32_t>(), o->cellCount.as<int>()); }
    if (K.fastTP == 44)
      FAST_LAUNCH(44)
    else if (K.fastTP == 56)
      FAST_LAUNCH(56)
    else
      FAST_LAUNCH(FAST_MAX_TILE)
#undef FAST_LAUNCH
  }
  // The Gaussian blur (vector-ALU bound) and the quadtree (one workgroup per image level, latency bound: 70 % of its wave
  // cycles wait) are independent and complement each other: the blur goes to a side stream that forks here, after k_fast, and
  // joins before k_describe (calls of FB_ORB_SIDE_MIN = 64 images or more: the fork/join costs ~0.05 ms of latency).
  // (When every kernel is bracketed for the per-kernel table -- fb_prof_enable without fb_prof_only -- everything stays on the
  // caller's stream, so that the table shows each kernel on its own.)
  static const int sideMin = [] {
    const char* e = getenv("FB_ORB_SIDE_MIN");
    return e ? atoi(e) : 64;
  }();
  const bool fork = batch >= sideMin && !(fb::g_prof_on && fb::g_prof_only < 0);
  hipStream_t sBlur = s;
  if (fork) {
    if (!o->sideStream) {
      FB_HIP(hipStreamCreateWithFlags(&o->sideStream, hipStreamNonBlocking));
      FB_HIP(hipEventCreateWithFlags(&o->evFork, hipEventDisableTiming));
      FB_HIP(hipEventCreateWithFlags(&o->evJoin, hipEventDisableTiming));
    }
    sBlur = o->sideStream;
    FB_HIP(hipEventRecord(o->evFork, s));
    FB_HIP(hipStreamWaitEvent(sBlur, o->evFork, 0));
  }
#ifndef FB_ORB_ABLATE_BLUR  // probe build only (profiles/probes/ablate_blur.sh): what the step costs without the blur launch (results are then wrong)
  {
    fb::ProfScope prof_(fb::P_BLUR, sBlur);
    k_blur<<<dim3((K.blurStrips[nl] + 3) / 4, batch), 256, 0, sBlur>>>(K, d_images, (long long)image_stride, stride, o->pyr.as<uint8_t>(),
                                                                        o->blur.as<uint8_t>());
  }
#endif
  if (fork) FB_HIP(hipEventRecord(o->evJoin, sBlur));
  static const int octWideMax = getenv("FB_OCT_WIDE_MAX") ? atoi(getenv("FB_OCT_WIDE_MAX")) : 512;
  const bool octWide = nl * batch <= octWideMax;  // few workgroups: 1024 threads each (two such workgroups fill a CU's wave slots)
  FB_HIP(hipFuncSetAttribute(octWide ? reinterpret_cast<const void *>(k_octree<1024>) : reinterpret_cast<const void *>(k_octree<256>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)o->octreeLds));
  {
  fb::ProfScope prof_(fb::P_OCTREE, s);
  if (octWide)
    k_octree<1024><<<dim3(batch, nl), 1024, o->octreeLds, s>>>(K, o->cellCand.as<uint32_t>(), o->cellCount.as<int>(), o->cand.as<uint32_t>(), candCount, o->nodeOf.as<uint16_t>(),
                                                          o->lvlOut.as<uint32_t>(), lvlCount, o->maxNodes);
  else
    k_octree<256><<<dim3(batch, nl), 256, o->octreeLds, s>>>(K, o->cellCand.as<uint32_t>(), o->cellCount.as<int>(), o->cand.as<uint32_t>(), candCount, o->nodeOf.as<uint16_t>(),
                                                         o->lvlOut.as<uint32_t>(), lvlCount, o->maxNodes);
  }
  if (fork) FB_HIP(hipStreamWaitEvent(s, o->evJoin, 0));  // the blurred levels are complete
  {
  fb::ProfScope prof_(fb::P_DESCRIBE, s);
  k_describe<<<dim3(((K.capOut + DESC_WPB * DESC_KPW - 1) / (DESC_WPB * DESC_KPW) + 7) / 8 * 8, batch), 64 * DESC_WPB, 0, s>>>(K, d_images, (long long)image_stride, stride, o->pyr.as<uint8_t>(),
                                                   o->blur.as<uint8_t>(), o->angTab.as<uint4>(), o->lvlOut.as<uint32_t>(), lvlCount, d_keypoints,
                                                   d_descriptors, d_n);
  }
  FB_HIP(hipGetLastError());
  o->lastImg = d_images;
  o->lastImgStride = (long long)image_stride;
  o->lastPitch0 = stride;
  return FB_OK;
}

int fb_orb_extract(fb_orb *o, const uint8_t *image, int width, int height, int stride, fb_keypoint *keypoints,
                   uint8_t *descriptors, int32_t *n_out) {
  FB_TRY(fb::check_device());
  FB_ARG(o && n_out);
  if (!image || width <= 0 || height <= 0) return FB_OK;  // _image.empty(): silent return, ORBextractor.cc:1046
  FB_ARG(keypoints && descriptors && stride >= width);
  const int cap = capacity_of(o->p);
  // one page-locked block per host thread carries the image in and [n | key points | descriptors | level counts] out: one
  // asynchronous copy each way and ONE synchronisation (five synchronous copies before)
  const size_t imgBytes = (size_t)stride * height, kpBytes = (size_t)cap * sizeof(fb_keypoint), dBytes = (size_t)cap * 32;
  const size_t oN = 0, oK = 256, oD = oK + ((kpBytes + 255) & ~(size_t)255), oC = oD + ((dBytes + 255) & ~(size_t)255), outBytes = oC + 256;
  uint8_t *pin = static_cast<uint8_t *>(fb::pinned_scratch(imgBytes + outBytes));
  if (!pin) return FB_ERR_HIP;
  fb::DevBuf dout;
  FB_TRY(dout.alloc(outBytes));
  if (o->ownedImg.bytes < imgBytes || !o->ownedImg.p) FB_TRY(o->ownedImg.alloc(imgBytes));
  memcpy(pin, image, imgBytes);
  FB_HIP(hipMemcpyAsync(o->ownedImg.p, pin, imgBytes, hipMemcpyHostToDevice, nullptr));
  uint8_t *dbase = dout.as<uint8_t>();
  const int keepStride = o->kpStride;
  if (keepStride) FB_TRY(fb_orb_set_output_stride(o, 0));  // this call's arrays hold exactly `cap` entries
  const int rc = fb_orb_extract_batch_dev(o, o->ownedImg.as<uint8_t>(), 1, width, height, stride, imgBytes, reinterpret_cast<fb_keypoint *>(dbase + oK),
                                          dbase + oD, reinterpret_cast<int32_t *>(dbase + oN), nullptr);
  if (keepStride) (void)fb_orb_set_output_stride(o, keepStride);
  FB_TRY(rc);
  uint8_t *pout = pin + imgBytes;
  FB_HIP(hipMemcpyAsync(pout, dbase, oC, hipMemcpyDeviceToHost, nullptr));
  FB_HIP(hipMemcpyAsync(pout + oC, o->counts.as<int>() + (size_t)o->batchCap * o->p.nlevels, (size_t)o->p.nlevels * 4, hipMemcpyDeviceToHost, nullptr));
  FB_HIP(hipStreamSynchronize(nullptr));
  memcpy(n_out, pout + oN, 4);
  {  // n_out is clamped to the capacity on the device; the per-level counts tell whether anything was cut off (the
     // quadtree can end a level with up to 4 x its number of root nodes: very wide strips with a tiny feature budget)
    const int *lc = reinterpret_cast<const int *>(pout + oC);
    int total = 0;
    for (int l = 0; l < o->p.nlevels; l++) total += lc[l];
    if (total > cap) {
      fb::set_error("keypoint capacity exceeded: the quadtree kept %d key points, fb_orb_capacity() is %d", total, cap);
      return FB_ERR_CAPACITY;
    }
  }
  memcpy(keypoints, pout + oK, (size_t)*n_out * sizeof(fb_keypoint));
  memcpy(descriptors, pout + oD, (size_t)*n_out * 32);
  return FB_OK;
}

int fb_orb_debug_timers(fb_orb *o, uint64_t *dst16) {
  FB_TRY(fb::check_device());
  FB_ARG(o && dst16 && o->timers.p);
  FB_HIP(hipDeviceSynchronize());
  FB_HIP(hipMemcpy(dst16, o->timers.p, 16 * 8, hipMemcpyDeviceToHost));
  FB_HIP(hipMemset(o->timers.p, 0, 16 * 8));
  return FB_OK;
}

int fb_orb_debug_candidates(fb_orb *o, int b, int level, uint32_t *dst, int cap) {
  FB_TRY(fb::check_device());
  FB_ARG(o && o->lastImg && level >= 0 && level < o->p.nlevels && b >= 0 && b < o->batchCap);
  FB_HIP(hipDeviceSynchronize());
  int n = 0;
  FB_HIP(hipMemcpy(&n, o->counts.as<int>() + (size_t)b * o->p.nlevels + level, 4, hipMemcpyDeviceToHost));
  const LevelInfo &L = o->K.L[level];
  const int m = n < cap ? n : cap;
  if (dst && m > 0)
    FB_HIP(hipMemcpy(dst, o->cand.as<uint32_t>() + (size_t)b * o->K.candStride + L.candBase, (size_t)m * 4, hipMemcpyDeviceToHost));
  return n;
}

int fb_orb_get_level(fb_orb *o, int b, int level, uint8_t *dst, int *w, int *hgt) {
  FB_TRY(fb::check_device());
  FB_ARG(o && w && hgt && o->lastImg && level >= 0 && level < o->p.nlevels && b >= 0 && b < o->batchCap);
  const LevelInfo &L = o->K.L[level];
  *w = L.w;
  *hgt = L.h;
  if (!dst) return FB_OK;
  FB_HIP(hipDeviceSynchronize());
  const uint8_t *src = level == 0 ? o->lastImg + (long long)b * o->lastImgStride
                                  : o->pyr.as<uint8_t>() + (long long)b * o->K.pyrStride + L.off;
  const int pitch = level == 0 ? o->lastPitch0 : L.pitch;
  FB_HIP(hipMemcpy2D(dst, L.w, src, pitch, L.w, L.h, hipMemcpyDeviceToHost));
  return FB_OK;
}

int fb_orb_get_blurred_level(fb_orb *o, int b, int level, uint8_t *dst) {
  FB_TRY(fb::check_device());
  FB_ARG(o && dst && o->lastImg && level >= 0 && level < o->p.nlevels && b >= 0 && b < o->batchCap);
  const LevelInfo &L = o->K.L[level];
  FB_HIP(hipDeviceSynchronize());
  const uint8_t *src = o->blur.as<uint8_t>() + (long long)b * o->K.blurStride + L.boff;
  FB_HIP(hipMemcpy2D(dst, L.w, src, L.pitch, L.w, L.h, hipMemcpyDeviceToHost));
  return FB_OK;
}

}  // extern "C"
