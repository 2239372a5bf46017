// pose.hip -- Optimizer::PoseOptimization / PoseOptimizationWithBird / BirdOptimization as
// one CDNA4 kernel (gfx950): one workgroup per frame, whole Levenberg-Marquardt schedule on
// the device.
//
// Replaces (reference file:line):
//   Optimizer::PoseOptimization           src/Optimizer.cc:246-475
//   Optimizer::PoseOptimizationWithBird   src/Optimizer.cc:478-705
//   Optimizer::BirdOptimization           src/Optimizer.cc:708-835
//   EdgeSE3ProjectXYZOnlyPose             Thirdparty/g2o/g2o/types/types_six_dof_expmap.cpp:266-299
//   EdgeSE3ProjectBirdPoint2CamXYZ        src/OdomG2oTypeQuat.cc:61-70, include/OdomG2oTypeQuat.h:89-109
//   OptimizationAlgorithmLevenberg::solve Thirdparty/g2o/g2o/core/optimization_algorithm_levenberg.cpp:61-164
//   BaseUnaryEdge::constructQuadraticForm core/base_unary_edge.hpp:43-72, RobustKernelHuber :78-91
//
// Data layout: the frame's edges (Xw, measurement, information) are staged once into LDS as
// float; every LM evaluation is ONE pass over them that produces the robust chi2, the upper
// triangle of the 6x6 J^T W J and the 6-vector J^T W e (28 doubles), reduced with wavefront
// shuffles and one LDS hop.  The pass at a trial pose doubles as the linearisation of the next
// iteration when the step is accepted, so an LM iteration costs one pass per trial.  The 6x6
// LDL^T, the exponential-map update and the lambda schedule run on lane 0.  Everything is fp64
// like g2o; inputs/outputs are float like the reference's cv::Mat fields.
#include "fb_common.h"
#include "fb_se3.h"

namespace {

constexpr int POSE_THREADS = 256;
constexpr int NW = POSE_THREADS / 64;
constexpr int NACC = 28;  // chi2, 21 x H upper, 6 x b

struct PoseLds {  // fixed-size shared state
  fb::SE3 T, Ttrial, Teval;
  double H[36], b[6], x[6];
  double red[NACC];
  double part[NW][NACC];
  int ok2;
};

__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

struct EdgeView {  // staged (LDS) or global float arrays of one problem
  const float *fxw, *fobs, *finf;
  const float *bxw, *bxc, *binf;
  const uint8_t *flevel, *blevel;  // 0 = active, 1 = outlier level, 2 = no edge
  int nfs, nbs;
  double wf, wb;  // information = (1.0*invSigma2)*w  (Optimizer.cc:303,542,588,756)
};

// accumulate one edge (base_unary_edge.hpp:43-72): b -= rho1 * J^T (info e), H += J^T (rho1 info) J.
// DIM is a compile-time constant and every loop is unrolled so J/err/acc stay in registers.
// ZMASK: bit r*6+i set = J[r][i] is a structural zero; its products are exact zeros and are not formed.
template <int DIM, unsigned ZMASK = 0u>
__device__ __forceinline__ void accumulate_edge(const double (&J)[DIM][6], const double (&err)[DIM], double info,
                                                bool robust, double delta, double (&acc)[NACC]) {
  double chi2 = 0;
#pragma unroll
  for (int r = 0; r < DIM; r++) chi2 += err[r] * (info * err[r]);
  double rho0 = chi2, rho1 = 1.;
  if (robust) fb::huber(chi2, delta, rho0, rho1);
  acc[0] += rho0;
  const double w = rho1 * info;
  double ie[DIM];
#pragma unroll
  for (int r = 0; r < DIM; r++) ie[r] = info * err[r];
  int hidx = 1;
#pragma unroll
  for (int i = 0; i < 6; i++) {
    double s = 0;
#pragma unroll
    for (int r = 0; r < DIM; r++)
      if (!((ZMASK >> (r * 6 + i)) & 1u)) s += J[r][i] * ie[r];
    acc[22 + i] -= rho1 * s;
#pragma unroll
    for (int j = i; j < 6; j++) {
      double h = 0;
#pragma unroll
      for (int r = 0; r < DIM; r++)
        if (!(((ZMASK >> (r * 6 + i)) | (ZMASK >> (r * 6 + j))) & 1u)) h += J[r][i] * w * J[r][j];
      acc[hidx] += h;
      hidx++;
    }
  }
}

// The same for EdgeSE3ProjectBirdPoint2CamXYZ, whose Jacobian -[-skew(p), I] is mostly structural zeros and -1s: only
// the non-zero products are formed, in the row order of the generic loop (a product with a structural 0 adds an exact
// 0, one with -1 is an exact negation), so the sums are the same numbers at a fifth of the arithmetic.
__device__ __forceinline__ void accumulate_bird_edge(const double (&p)[3], const double (&err)[3], double info, bool robust,
                                                     double delta, double (&acc)[NACC]) {
  double chi2 = 0;
#pragma unroll
  for (int r = 0; r < 3; r++) chi2 += err[r] * (info * err[r]);
  double rho0 = chi2, rho1 = 1.;
  if (robust) fb::huber(chi2, delta, rho0, rho1);
  acc[0] += rho0;
  const double w = rho1 * info;
  const double ie0 = info * err[0], ie1 = info * err[1], ie2 = info * err[2];
  const double px = p[0], py = p[1], pz = p[2];
  // b: s_i = sum_r J[r][i] * ie[r]
  acc[22] -= rho1 * (pz * ie1 + (-py) * ie2);
  acc[23] -= rho1 * ((-pz) * ie0 + px * ie2);
  acc[24] -= rho1 * (py * ie0 + (-px) * ie1);
  acc[25] -= rho1 * (-ie0);
  acc[26] -= rho1 * (-ie1);
  acc[27] -= rho1 * (-ie2);
  // H upper triangle, row-major from acc[1]: h_ij = sum_r (J[r][i] * w) * J[r][j]
  const double a = pz * w, b = py * w, c = px * w;
  acc[1] += a * pz + b * py;    // (0,0)
  acc[2] += -(b * px);          // (0,1)
  acc[3] += -(a * px);          // (0,2)
  acc[5] += -a;                 // (0,4)   (0,3) = 0
  acc[6] += b;                  // (0,5)
  acc[7] += a * pz + c * px;    // (1,1)
  acc[8] += -(a * py);          // (1,2)
  acc[9] += a;                  // (1,3)   (1,4) = 0
  acc[11] += -c;                // (1,5)
  acc[12] += b * py + c * px;   // (2,2)
  acc[13] += -b;                // (2,3)
  acc[14] += c;                 // (2,4)   (2,5) = 0
  acc[16] += w;                 // (3,3)   (3,4) = (3,5) = 0
  acc[19] += w;                 // (4,4)   (4,5) = 0
  acc[21] += w;                 // (5,5)
}

// one evaluation: robust chi2 + H + b at pose T over the active edges
__device__ void eval_pass(const EdgeView &E, const fb::SE3 &T, bool robust, double delta, double fx, double fy,
                          double cx, double cy, PoseLds *S) {
  double acc[NACC];
#pragma unroll
  for (int i = 0; i < NACC; i++) acc[i] = 0;
  const int tid = threadIdx.x;
  for (int e = tid; e < E.nfs; e += POSE_THREADS) {  // EdgeSE3ProjectXYZOnlyPose
    if (E.flevel[e] != 0) continue;
    const double Xw[3] = {E.fxw[e * 3], E.fxw[e * 3 + 1], E.fxw[e * 3 + 2]};
    double p[3];
    fb::se3_map(T, Xw, p);
    // ONE fp64 division per edge (it expands to ~35 instructions and the edge pass is what bounds an LM evaluation):
    // x/z and y/z as products with 1/z differ from the quotients by at most one ulp, far inside the 1e-4 bar
    const double X = p[0], Y = p[1], invz = 1.0 / p[2], invz_2 = invz * invz;
    const double err[2] = {(double)E.fobs[e * 2] - ((X * invz) * fx + cx), (double)E.fobs[e * 2 + 1] - ((Y * invz) * fy + cy)};
    const double J[2][6] = {{X * Y * invz_2 * fx, -(1 + (X * X * invz_2)) * fx, Y * invz * fx, -invz * fx, 0, X * invz_2 * fx},
                            {(1 + Y * Y * invz_2) * fy, -X * Y * invz_2 * fy, -X * invz * fy, 0, -invz * fy, Y * invz_2 * fy}};
    accumulate_edge<2, (1u << 4) | (1u << 9)>(J, err, (double)E.finf[e] * E.wf, robust, delta, acc);  // J[0][4] = J[1][3] = 0
  }
  for (int k = tid; k < E.nbs; k += POSE_THREADS) {  // EdgeSE3ProjectBirdPoint2CamXYZ
    if (E.blevel[k] != 0) continue;
    const double Xw[3] = {E.bxw[k * 3], E.bxw[k * 3 + 1], E.bxw[k * 3 + 2]};
    double p[3];
    fb::se3_map(T, Xw, p);
    const double err[3] = {(double)E.bxc[k * 3] - p[0], (double)E.bxc[k * 3 + 1] - p[1], (double)E.bxc[k * 3 + 2] - p[2]};
    accumulate_bird_edge(p, err, (double)E.binf[k] * E.wb, robust, delta, acc);  // J = -[-skew(p), I]
  }
  const int lane = tid & 63, wv = tid >> 6;
  {
    double v[32];
#pragma unroll
    for (int i = 0; i < 32; i++) v[i] = i < NACC ? acc[i] : 0.0;
    const double s = fb::wave_column_sums32(v, lane);
    if ((lane & 1) == 0 && (lane >> 1) < NACC) S->part[wv][lane >> 1] = s;
  }
  __syncthreads();
  if (tid < NACC) {
    double s = 0;
    for (int w2 = 0; w2 < NW; w2++) s += S->part[w2][tid];
    S->red[tid] = s;
  }
  __syncthreads();
}

__device__ __forceinline__ void unpack_system(const double *red, double *H, double *b) {
  int k = 1;
  for (int i = 0; i < 6; i++)
    for (int j = i; j < 6; j++) { H[i * 6 + j] = red[k]; H[j * 6 + i] = red[k]; k++; }
  for (int i = 0; i < 6; i++) b[i] = red[22 + i];
}

// chi2 of one edge at pose T for the inlier / outlier decision (computeError + chi2(), base_edge.h:58-61; compared as
// float with 5.991 / 1.5 at Optimizer.cc:410,645,672).  This file is compiled with fused multiply-adds for the LM
// evaluations, which are held to a tolerance; the DECISION is an integer result, so its arithmetic is written out here
// with contraction off and with g2o's own quotients (project2d: v / v(2)), operation for operation what the oracle
// evaluates: given the same pose the masks are equal by construction, not by luck of the rounding.
__device__ __forceinline__ double edge_chi2(const EdgeView &E, int e, bool bird, const fb::SE3 &T, double fx,
                                            double fy, double cx, double cy) {
#pragma clang fp contract(off)
  const float *X = bird ? E.bxw + e * 3 : E.fxw + e * 3;
  const double v0 = X[0], v1 = X[1], v2 = X[2];
  const double qx = T.r.x, qy = T.r.y, qz = T.r.z, qw = T.r.w;
  // Eigen's quaternion * vector (fb::quat_rotate), then + t (se3quat.h:217-220)
  double u0 = qy * v2 - qz * v1, u1 = qz * v0 - qx * v2, u2 = qx * v1 - qy * v0;
  u0 += u0; u1 += u1; u2 += u2;
  const double p0 = (v0 + qw * u0 + (qy * u2 - qz * u1)) + T.t[0];
  const double p1 = (v1 + qw * u1 + (qz * u0 - qx * u2)) + T.t[1];
  const double p2 = (v2 + qw * u2 + (qx * u1 - qy * u0)) + T.t[2];
  if (!bird) {
    const double e0 = (double)E.fobs[e * 2] - ((p0 / p2) * fx + cx);
    const double e1 = (double)E.fobs[e * 2 + 1] - ((p1 / p2) * fy + cy);
    const double info = (double)E.finf[e] * E.wf;
    double s = 0;
    s += e0 * (info * e0);
    s += e1 * (info * e1);
    return s;
  }
  const double info = (double)E.binf[e] * E.wb;
  const double d0 = (double)E.bxc[e * 3] - p0, d1 = (double)E.bxc[e * 3 + 1] - p1, d2 = (double)E.bxc[e * 3 + 2] - p2;
  double s = 0;
  s += d0 * (info * d0);
  s += d1 * (info * d1);
  s += d2 * (info * d2);
  return s;
}

__global__ __launch_bounds__(POSE_THREADS) void k_pose_opt(fb_pose_opt_args A, int staged) {
  extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
  __shared__ PoseLds S;
  __shared__ int s_cnt[2];
  const int bidx = blockIdx.x, tid = threadIdx.x;
  const int mode = A.mode;
  const size_t fo = (size_t)bidx * A.front_stride, bo = (size_t)bidx * A.bird_stride;
  const int nfs = (mode != FB_POSE_BIRD) ? A.n_front[bidx] : 0;
  const int nbs = (mode != FB_POSE_FRONT) ? A.n_bird[bidx] : 0;
  float *Tcw = A.Tcw + (size_t)bidx * 12;
  // ---- stage the edges
  EdgeView E;
  E.nfs = nfs; E.nbs = nbs;
  uint8_t *flevel, *blevel;
  {
    float *lf = reinterpret_cast<float *>(smem);
    size_t off = 0;
    if (staged) {
      float *fxw = lf; off += (size_t)A.front_stride * 3;
      float *fobs = lf + off; off += (size_t)A.front_stride * 2;
      float *finf = lf + off; off += (size_t)A.front_stride;
      float *bxw = lf + off; off += (size_t)A.bird_stride * 3;
      float *bxc = lf + off; off += (size_t)A.bird_stride * 3;
      float *binf = lf + off; off += (size_t)A.bird_stride;
      for (int i = tid; i < nfs * 3; i += POSE_THREADS) fxw[i] = A.front_xw[fo * 3 + i];
      for (int i = tid; i < nfs * 2; i += POSE_THREADS) fobs[i] = A.front_obs[fo * 2 + i];
      for (int i = tid; i < nbs * 3; i += POSE_THREADS) { bxw[i] = A.bird_xw[bo * 3 + i]; bxc[i] = A.bird_xc[bo * 3 + i]; }
      E.fxw = fxw; E.fobs = fobs; E.finf = finf; E.bxw = bxw; E.bxc = bxc; E.binf = binf;
      for (int i = tid; i < nfs; i += POSE_THREADS) finf[i] = A.front_inv_sigma2[fo + i];
      for (int i = tid; i < nbs; i += POSE_THREADS) binf[i] = A.bird_inv_sigma2[bo + i];
    } else {
      E.fxw = A.front_xw + fo * 3; E.fobs = A.front_obs + fo * 2; E.finf = A.front_inv_sigma2 + fo;
      E.bxw = A.bird_xw + bo * 3; E.bxc = A.bird_xc + bo * 3; E.binf = A.bird_inv_sigma2 + bo;
    }
    flevel = reinterpret_cast<uint8_t *>(lf + off);
    blevel = flevel + ((A.front_stride + 15) & ~15);
    E.flevel = flevel; E.blevel = blevel;
  }
  if (tid < 2) s_cnt[tid] = 0;
  __syncthreads();
  // edge construction (Optimizer.cc:525-602): count edges, clear mvbOutlier of mapped slots
  for (int i = tid; i < nfs; i += POSE_THREADS) {
    const bool v = !A.front_valid || A.front_valid[fo + i];
    flevel[i] = v ? 0 : 2;
    if (v) { A.front_outlier[fo + i] = 0; atomicAdd(&s_cnt[0], 1); }
  }
  for (int i = tid; i < nbs; i += POSE_THREADS) {
    const bool v = !A.bird_valid || A.bird_valid[bo + i];
    blevel[i] = v ? 0 : 2;
    if (v) atomicAdd(&s_cnt[1], 1);
  }
  __syncthreads();
  const int nf = s_cnt[0], nb = s_cnt[1];
  if (mode == FB_POSE_BIRD ? nb < 3 : nf < 3) {  // Optimizer.cc:379,607,776
    if (tid == 0) A.ninliers[bidx] = 0;
    return;
  }
  E.wf = (mode == FB_POSE_FRONT) ? 1.0 : (double)A.wF;
  E.wb = (double)A.wB;
  const double fx = A.fx, fy = A.fy, cx = A.cx, cy = A.cy;
  const double delta = (double)(float)sqrt(5.991);  // const float deltaMono = sqrt(5.991)
  const float chi2Mono = (mode == FB_POSE_FRONT) ? 5.991f : 1.5f;
  const float chi2Bird = 5.991f;
  if (tid == 0) {
    S.T = fb::se3_from_float12(Tcw);
    S.Teval = S.T;
  }
  __syncthreads();
  const fb::SE3 T0 = S.T;

  int nBad = 0, nBadBird = 0;
  for (int it = 0; it < 4; it++) {
    const bool robust = it < 3;  // setRobustKernel(0) after the third round (Optimizer.cc:657,685)
    if (tid == 0) S.T = T0;      // vSE3->setEstimate(toSE3Quat(pFrame->mTcw))
    __syncthreads();
    // active edges = level 0 (initializeOptimization(0))
    int nact = 0;
    for (int i = tid; i < nfs; i += POSE_THREADS) nact += flevel[i] == 0;
    for (int i = tid; i < nbs; i += POSE_THREADS) nact += blevel[i] == 0;
    nact = __syncthreads_count(nact > 0);
    if (nact > 0) {
      // ---- optimize(10): OptimizationAlgorithmLevenberg
      eval_pass(E, S.T, robust, delta, fx, fy, cx, cy, &S);
      if (tid == 0) { unpack_system(S.red, S.H, S.b); S.Teval = S.T; }
      double currentChi = S.red[0];
      __syncthreads();
      double lambda = 0, ni = 2;
      int nBadLM = 0;
      for (int iter = 0; iter < 10; iter++) {
        const double iniChi = currentChi;
        if (iter == 0) {
          double m = 0;
          for (int j = 0; j < 6; j++) m = fmax(fabs(S.H[j * 6 + j]), m);
          lambda = 1e-5 * m;
          ni = 2;
          nBadLM = 0;
        }
        double rho = 0;
        int qmax = 0;
        do {
          if (tid == 0) {
            S.ok2 = fb::ldlt6(S.H, lambda, S.b, S.x) ? 1 : 0;
            S.Ttrial = fb::se3_mul(fb::se3_exp(S.x), S.T);  // oplus
            S.Teval = S.Ttrial;
          }
          __syncthreads();
          eval_pass(E, S.Ttrial, robust, delta, fx, fy, cx, cy, &S);
          double tempChi = S.red[0];
          if (!S.ok2) tempChi = 1.7976931348623157e308;
          rho = currentChi - tempChi;
          double scale = 0;
          for (int j = 0; j < 6; j++) scale += S.x[j] * (lambda * S.x[j] + S.b[j]);
          scale += 1e-3;
          rho /= scale;
          const bool accept = rho > 0 && isfinite(tempChi);
          __syncthreads();  // everyone has read red/x/b
          if (accept) {
            double alpha = 1. - pow((2 * rho - 1), 3);
            alpha = fmin(alpha, 2. / 3.);
            const double scaleFactor = fmax(1. / 3., alpha);
            lambda *= scaleFactor;
            ni = 2;
            currentChi = tempChi;
            if (tid == 0) { S.T = S.Ttrial; unpack_system(S.red, S.H, S.b); }
          } else {
            lambda *= ni;
            ni *= 2;
          }
          __syncthreads();
          qmax++;
        } while (rho < 0 && qmax < 10);
        if (qmax == 10 || rho == 0) break;  // Terminate
        if ((iniChi - currentChi) * 1e3 < iniChi) nBadLM++;
        else nBadLM = 0;
        if (nBadLM >= 3) break;
      }
    }
    // ---- classify (Optimizer.cc:396-431, 627-686, 791-822)
    const fb::SE3 T = S.T, Teval = S.Teval;
    int bad = 0, badb = 0;
    for (int i = tid; i < nfs; i += POSE_THREADS) {
      if (flevel[i] == 2) continue;
      const bool wasOut = A.front_outlier[fo + i] != 0;
      const float chi2 = (float)edge_chi2(E, i, false, wasOut ? T : Teval, fx, fy, cx, cy);
      bool isBad;
      if (mode == FB_POSE_FRONT) isBad = chi2 > chi2Mono;
      else isBad = chi2 > chi2Mono * ((double)A.wF + 1e-9);
      A.front_outlier[fo + i] = isBad ? 1 : 0;
      flevel[i] = isBad ? 1 : 0;
      bad += isBad;
    }
    for (int i = tid; i < nbs; i += POSE_THREADS) {
      if (blevel[i] == 2) continue;
      const bool wasOut = A.bird_outlier[bo + i] != 0;
      const float chi2 = (float)edge_chi2(E, i, true, wasOut ? T : Teval, fx, fy, cx, cy);
      const float chi2Bad = (float)(chi2Bird * ((double)A.wB + 1e-9));
      const bool isBad = chi2 > chi2Bad;
      A.bird_outlier[bo + i] = isBad ? 1 : 0;
      blevel[i] = isBad ? 1 : 0;
      badb += isBad;
    }
    {
      const double sb = wave_sum((double)bad), sbb = wave_sum((double)badb);
      if ((tid & 63) == 0) { S.part[tid >> 6][0] = sb; S.part[tid >> 6][1] = sbb; }
      __syncthreads();
      double tb = 0, tbb = 0;
      for (int w2 = 0; w2 < NW; w2++) { tb += S.part[w2][0]; tbb += S.part[w2][1]; }
      nBad = (int)tb; nBadBird = (int)tbb;
      __syncthreads();
    }
    if (nf + nb < 10) break;  // optimizer.edges().size()<10
  }
  if (tid == 0) {
    fb::se3_to_float12(S.T, Tcw);
    A.ninliers[bidx] = (mode == FB_POSE_BIRD) ? nb - nBadBird : nf - nBad;
  }
}

// --- device-side edge construction (Optimizer.cc:525-602) --------------------------------
struct GatherK { float inv_sigma2[FB_MAX_LEVELS]; int nlevels; };

__global__ void k_gather_front(int kp_stride, int mp_stride, const int32_t *__restrict__ n, const fb_keypoint *__restrict__ kps,
                               const int32_t *__restrict__ match, const float *__restrict__ mp_xw, GatherK G,
                               float *__restrict__ xw, float *__restrict__ obs, float *__restrict__ inf,
                               uint8_t *__restrict__ valid) {
  const int b = blockIdx.y, i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= kp_stride) return;
  const size_t o = (size_t)b * kp_stride + i;
  const int m = (i < n[b]) ? match[o] : -1;
  if (m < 0) { valid[o] = 0; return; }
  const fb_keypoint kp = kps[o];
  const float *X = mp_xw + ((size_t)b * mp_stride + m) * 3;
  xw[o * 3] = X[0]; xw[o * 3 + 1] = X[1]; xw[o * 3 + 2] = X[2];
  obs[o * 2] = kp.x; obs[o * 2 + 1] = kp.y;
  inf[o] = G.inv_sigma2[kp.octave];
  valid[o] = 1;
}

__global__ void k_gather_bird(int kp_stride, int mp_stride, const int32_t *__restrict__ n, const fb_keypoint *__restrict__ kps,
                              const float *__restrict__ cam, const int32_t *__restrict__ match, const float *__restrict__ mpb_xw,
                              GatherK G, float *__restrict__ xw, float *__restrict__ xc, float *__restrict__ inf,
                              uint8_t *__restrict__ valid) {
  const int b = blockIdx.y, i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= kp_stride) return;
  const size_t o = (size_t)b * kp_stride + i;
  const int m = (i < n[b]) ? match[o] : -1;
  if (m < 0) { valid[o] = 0; return; }
  const float *X = mpb_xw + ((size_t)b * mp_stride + m) * 3;
  xw[o * 3] = X[0]; xw[o * 3 + 1] = X[1]; xw[o * 3 + 2] = X[2];
  xc[o * 3] = cam[o * 3]; xc[o * 3 + 1] = cam[o * 3 + 1]; xc[o * 3 + 2] = cam[o * 3 + 2];
  inf[o] = G.inv_sigma2[kps[o].octave];
  valid[o] = 1;
}

}  // namespace

extern "C" {

int fb_pose_opt_batch_dev(const fb_pose_opt_args *A, void *stream) {
  FB_TRY(fb::check_device());
  FB_ARG(A && A->batch >= 0 && A->front_stride >= 0 && A->bird_stride >= 0);
  FB_ARG(A->mode == FB_POSE_FRONT || A->mode == FB_POSE_FRONT_BIRD || A->mode == FB_POSE_BIRD);
  FB_ARG(A->Tcw && A->ninliers);
  // the edge family a mode optimises must be complete (the kernel reads and writes all of it); the other one is ignored
  if (A->mode != FB_POSE_BIRD)
    FB_ARG(A->n_front && A->front_outlier && (A->front_stride == 0 || (A->front_xw && A->front_obs && A->front_inv_sigma2)));
  if (A->mode != FB_POSE_FRONT)
    FB_ARG(A->n_bird && A->bird_outlier && (A->bird_stride == 0 || (A->bird_xw && A->bird_xc && A->bird_inv_sigma2)));
  if (A->batch == 0) return FB_OK;
  const size_t flags = ((size_t)((A->front_stride + 15) & ~15)) + ((A->bird_stride + 15) & ~15);
  const size_t stagedBytes = ((size_t)A->front_stride * 6 + (size_t)A->bird_stride * 7) * 4;
  int staged = 1;
  size_t lds = stagedBytes + flags;
  if (lds > 140 * 1024) { staged = 0; lds = flags; }
  if (lds > 140 * 1024) { fb::set_error("fb_pose_opt: too many edges per frame for LDS flags"); return FB_ERR_CAPACITY; }
  FB_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(k_pose_opt), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  fb::ProfScope prof_(fb::P_POSE, fb::as_stream(stream));
  k_pose_opt<<<A->batch, POSE_THREADS, lds, fb::as_stream(stream)>>>(*A, staged);
  FB_HIP(hipGetLastError());
  return FB_OK;
}

int fb_pose_opt(const fb_pose_opt_args *H) {
  FB_TRY(fb::check_device());
  FB_ARG(H && H->batch >= 0 && H->front_stride >= 0 && H->bird_stride >= 0);
  FB_ARG(H->mode == FB_POSE_FRONT || H->mode == FB_POSE_FRONT_BIRD || H->mode == FB_POSE_BIRD);
  FB_ARG(H->Tcw && H->ninliers);
  if (H->mode != FB_POSE_BIRD) FB_ARG(H->n_front && H->front_outlier);
  if (H->mode != FB_POSE_FRONT) FB_ARG(H->n_bird && H->bird_outlier);
  for (int b = 0; b < H->batch; b++) {  // a count beyond the stride would run into the next frame's edges
    if (H->mode != FB_POSE_BIRD) FB_ARG(H->n_front[b] >= 0 && H->n_front[b] <= H->front_stride);
    if (H->mode != FB_POSE_FRONT) FB_ARG(H->n_bird[b] >= 0 && H->n_bird[b] <= H->bird_stride);
  }
  fb_pose_opt_args D = *H;
  const size_t B = H->batch, fs = H->front_stride, bs = H->bird_stride;
#define UPB(buf, field, bytes)                                                                           \
  fb::DevBuf buf;                                                                                        \
  if (H->field) { FB_TRY(buf.upload(H->field, (bytes))); D.field = buf.as<std::remove_pointer<decltype(D.field)>::type>(); }
  UPB(b0, n_front, B * 4) UPB(b1, front_xw, B * fs * 12) UPB(b2, front_obs, B * fs * 8) UPB(b3, front_inv_sigma2, B * fs * 4)
  UPB(b4, front_valid, B * fs) UPB(b5, n_bird, B * 4) UPB(b6, bird_xw, B * bs * 12) UPB(b7, bird_xc, B * bs * 12)
  UPB(b8, bird_inv_sigma2, B * bs * 4) UPB(b9, bird_valid, B * bs) UPB(b10, bird_outlier, B * bs) UPB(b11, Tcw, B * 48)
  UPB(b12, front_outlier, B * fs)
#undef UPB
  fb::DevBuf o1;
  FB_TRY(o1.alloc(B * 4));
  D.ninliers = o1.as<int32_t>();
  // the family a mode ignores may be absent altogether: the kernel never dereferences it (n = 0 for that family)
  FB_TRY(fb_pose_opt_batch_dev(&D, nullptr));
  FB_HIP(hipDeviceSynchronize());
  FB_TRY(b11.download(H->Tcw, B * 48));
  if (H->front_outlier) FB_TRY(b12.download(H->front_outlier, B * fs));
  if (H->bird_outlier) FB_TRY(b10.download(H->bird_outlier, B * bs));
  return o1.download(H->ninliers, B * 4);
}

int fb_pose_gather_front_dev(int batch, int kp_stride, int mp_stride, const int32_t *d_n, const fb_keypoint *d_kps,
                             const int32_t *d_match, const float *d_mp_xw, const float *inv_level_sigma2, int nlevels,
                             float *d_front_xw, float *d_front_obs, float *d_front_inv_sigma2, uint8_t *d_front_valid,
                             void *stream) {
  FB_TRY(fb::check_device());
  FB_ARG(batch >= 0 && kp_stride > 0 && inv_level_sigma2 && nlevels >= 1 && nlevels <= FB_MAX_LEVELS);
  if (batch == 0) return FB_OK;
  GatherK G;
  memset(&G, 0, sizeof(G));
  G.nlevels = nlevels;
  for (int i = 0; i < nlevels; i++) G.inv_sigma2[i] = inv_level_sigma2[i];
  fb::ProfScope prof_(fb::P_GATHER, fb::as_stream(stream));
  k_gather_front<<<dim3((kp_stride + 255) / 256, batch), 256, 0, fb::as_stream(stream)>>>(
      kp_stride, mp_stride, d_n, d_kps, d_match, d_mp_xw, G, d_front_xw, d_front_obs, d_front_inv_sigma2, d_front_valid);
  FB_HIP(hipGetLastError());
  return FB_OK;
}

int fb_pose_gather_bird_dev(int batch, int kp_stride, int mp_stride, const int32_t *d_n, const fb_keypoint *d_kps,
                            const float *d_cam_xyz, const int32_t *d_match, const float *d_mpb_xw,
                            const float *inv_level_sigma2, int nlevels, float *d_bird_xw, float *d_bird_xc,
                            float *d_bird_inv_sigma2, uint8_t *d_bird_valid, void *stream) {
  FB_TRY(fb::check_device());
  FB_ARG(batch >= 0 && kp_stride > 0 && inv_level_sigma2 && nlevels >= 1 && nlevels <= FB_MAX_LEVELS);
  if (batch == 0) return FB_OK;
  GatherK G;
  memset(&G, 0, sizeof(G));
  G.nlevels = nlevels;
  for (int i = 0; i < nlevels; i++) G.inv_sigma2[i] = inv_level_sigma2[i];
  fb::ProfScope prof_(fb::P_GATHER, fb::as_stream(stream));
  k_gather_bird<<<dim3((kp_stride + 255) / 256, batch), 256, 0, fb::as_stream(stream)>>>(
      kp_stride, mp_stride, d_n, d_kps, d_cam_xyz, d_match, d_mpb_xw, G, d_bird_xw, d_bird_xc, d_bird_inv_sigma2, d_bird_valid);
  FB_HIP(hipGetLastError());
  return FB_OK;
}

}  // extern "C"
