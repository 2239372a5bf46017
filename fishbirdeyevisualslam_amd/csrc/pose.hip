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
#include <type_traits>
#include "fb_common.h"
#include "fb_se3.h"

namespace {

constexpr int POSE_THREADS = 256;  // the LDS-staged kernel
constexpr int NACC = 28;  // chi2, 21 x H upper, 6 x b

// -DFB_POSE_STAMPS (profiles/probes/pose_stamps.py builds that variant on the GPU box): thread 0 of workgroup 0 adds the
// shader-clock time of each phase of an LM evaluation to g_pose_stamps[]; compiled out of the product build.
#ifdef FB_POSE_STAMPS
__device__ unsigned long long g_pose_stamps[16];
#define POSE_T0() unsigned long long pt_ = 0; if (blockIdx.x == 0 && threadIdx.x == 0) { __builtin_amdgcn_sched_barrier(0); pt_ = __builtin_amdgcn_s_memtime(); __builtin_amdgcn_sched_barrier(0); }
#define POSE_TICK(slot_) if (blockIdx.x == 0 && threadIdx.x == 0) { __builtin_amdgcn_sched_barrier(0); const unsigned long long t_ = __builtin_amdgcn_s_memtime(); __builtin_amdgcn_sched_barrier(0); g_pose_stamps[slot_] += t_ - pt_; pt_ = t_; }
#define POSE_COUNT(slot_) if (blockIdx.x == 0 && threadIdx.x == 0) g_pose_stamps[slot_] += 1;
#else
#define POSE_T0()
#define POSE_TICK(slot_)
#define POSE_COUNT(slot_)
#endif

struct PoseLds {  // fixed-size shared state
  fb::SE3 T, Ttrial, Teval, T0;
  double H[36], b[6], x[6];
  double red[NACC];
  double part[8][NACC];  // per-wave partial sums (up to 512 threads)
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
template <int NT>
__device__ void eval_pass(const EdgeView &E, const fb::SE3 &T, bool robust, double delta, double fx, double fy,
                          double cx, double cy, PoseLds *S) {
  constexpr int POSE_THREADS = NT, NW = NT / 64;
  double acc[NACC];
#pragma unroll
  for (int i = 0; i < NACC; i++) acc[i] = 0;
  const int tid = threadIdx.x;
  POSE_T0()
  POSE_COUNT(15)
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
  POSE_TICK(0)
  for (int k = tid; k < E.nbs; k += POSE_THREADS) {  // EdgeSE3ProjectBirdPoint2CamXYZ
    if (E.blevel[k] != 0) continue;
    const double Xw[3] = {E.bxw[k * 3], E.bxw[k * 3 + 1], E.bxw[k * 3 + 2]};
    double p[3];
    fb::se3_map(T, Xw, p);
    const double err[3] = {(double)E.bxc[k * 3] - p[0], (double)E.bxc[k * 3 + 1] - p[1], (double)E.bxc[k * 3 + 2] - p[2]};
    accumulate_bird_edge(p, err, (double)E.binf[k] * E.wb, robust, delta, acc);  // J = -[-skew(p), I]
  }
  POSE_TICK(1)
  const int lane = tid & 63, wv = tid >> 6;
  {
    double v[32];
#pragma unroll
    for (int i = 0; i < 32; i++) v[i] = i < NACC ? acc[i] : 0.0;
    const double s = fb::wave_column_sums32(v, lane);
    if ((lane & 1) == 0 && (lane >> 1) < NACC) S->part[wv][lane >> 1] = s;
  }
  POSE_TICK(2)
  __syncthreads();
  if (tid < NACC) {
    double s = 0;
    for (int w2 = 0; w2 < NW; w2++) s += S->part[w2][tid];
    S->red[tid] = s;
  }
  __syncthreads();
  POSE_TICK(3)
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

// The generic schedule for any number of edges per frame: edges staged in LDS as float (staged = 1) or read from HBM / L2
// in every evaluation (staged = 0), per-edge level bytes in LDS, butterfly sums.  It is the whole kernel for frames beyond
// the register-resident kernel below, and that kernel's in-kernel way out for a frame with more edges than its slots.
template <int NT>
__device__ void pose_generic(const fb_pose_opt_args &A, int staged, uint8_t *smem, PoseLds &S, int *s_cnt) {
  constexpr int POSE_THREADS = NT, NW = NT / 64;
  const int bidx = blockIdx.x, tid = threadIdx.x;
#ifdef FB_POSE_STAMPS
  const unsigned long long pose_t_start = __builtin_amdgcn_s_memtime();
#endif
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
      eval_pass<NT>(E, S.T, robust, delta, fx, fy, cx, cy, &S);
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
          {
            POSE_T0()
            if (tid == 0) {
              S.ok2 = fb::ldlt6(S.H, lambda, S.b, S.x) ? 1 : 0;
              POSE_TICK(4)
              S.Ttrial = fb::se3_mul(fb::se3_exp(S.x), S.T);  // oplus
              S.Teval = S.Ttrial;
              POSE_TICK(5)
            }
            __syncthreads();
            POSE_TICK(6)
          }
          eval_pass<NT>(E, S.Ttrial, robust, delta, fx, fy, cx, cy, &S);
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
    POSE_T0()
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
    POSE_TICK(7)
    if (nf + nb < 10) break;  // optimizer.edges().size()<10
  }
  if (tid == 0) {
    fb::se3_to_float12(S.T, Tcw);
    A.ninliers[bidx] = (mode == FB_POSE_BIRD) ? nb - nBadBird : nf - nBad;
  }
#ifdef FB_POSE_STAMPS
  if (blockIdx.x == 0 && tid == 0) g_pose_stamps[14] += __builtin_amdgcn_s_memtime() - pose_t_start;
#endif
}

__global__ __launch_bounds__(256) void k_pose_opt(fb_pose_opt_args A, int staged) {
  extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
  __shared__ PoseLds S;
  __shared__ int s_cnt[2];
  pose_generic<256>(A, staged, smem, S, s_cnt);
}

// ---------------------------------------------------------------------------------------------------------------------
// k_pose_opt_reg -- the same schedule, built for the latency of ONE frame (single-sequence tracking is bounded by this
// kernel).  Phase stamps of the LDS-staged kernel above (profiles/r02_pose_stamps_before.txt): an LM evaluation took
// ~29 k cycles = 12.6 k edge arithmetic of one wave per SIMD, 7 k for the 28 cross-lane sums (ds_bpermute butterflies
// are latency chains), 4 k for the 6x6 solve + exponential map on one lane.  Here
//  * the edges of a thread live in REGISTERS (edge e -> thread e % NT, slot e / NT, up to 8 front + 4 bird slots), so an
//    evaluation reads nothing but the pose from LDS and NT = 512 threads (two waves per SIMD) cover each other's fp64
//    latencies;
//  * the 28 sums go through an LDS transpose: every thread stores its 28 accumulators (conflict-free rows of NT + CPA
//    doubles), 28 x CPA threads add 32 values each with 4 independent chains, the CPA partial sums of an accumulator
//    sit in adjacent lanes and are combined with DPP row shifts (VALU moves, no LDS crossbar);
//  * the update T <- exp(x) T is formed directly on the quaternion (no rotation matrix, one reciprocal square root per
//    normalisation).
template <int CTRL>
__device__ __forceinline__ double dpp_f64(double v) {  // v of the lane selected by the DPP control, +0.0 where there is none
  int lo = __double2loint(v), hi = __double2hiint(v);
  lo = __builtin_amdgcn_update_dpp(0, lo, CTRL, 0xF, 0xF, true);
  hi = __builtin_amdgcn_update_dpp(0, hi, CTRL, 0xF, 0xF, true);
  return __hiloint2double(hi, lo);
}

// fp64 reciprocal / reciprocal square root from the hardware seed (v_rcp_f64 / v_rsq_f64) and two Newton steps: 6 instead
// of the ~35 dependent instructions of the IEEE division sequence, relative error < 2^-50.  Only for the tolerance-held
// LM evaluations (pose within 1e-4); the inlier / outlier decision keeps true quotients (chi2_*_vals).
__device__ __forceinline__ double rcp_fast(double x) {
  double r = __builtin_amdgcn_rcp(x);
  r = fma(fma(-x, r, 1.0), r, r);
  r = fma(fma(-x, r, 1.0), r, r);
  return r;
}
__device__ __forceinline__ double rsqrt_fast(double x) {
  double r = __builtin_amdgcn_rsq(x);
  r = r * fma(-0.5 * x, r * r, 1.5);
  r = r * fma(-0.5 * x, r * r, 1.5);
  return r;
}
// sin and cos of a small angle (|x| <= 0.5: the half angle of an LM step) by their Taylor series to x^17 / x^18
// (truncation < 1e-19); larger arguments go to the library
__device__ __forceinline__ void sincos_small(double x, double *s, double *c) {
  if (fabs(x) > 0.5) { sincos(x, s, c); return; }
  const double x2 = x * x;
  double ps = -1.0 / 355687428096000.0;
  ps = fma(ps, x2, 1.0 / 1307674368000.0);
  ps = fma(ps, x2, -1.0 / 6227020800.0);
  ps = fma(ps, x2, 1.0 / 39916800.0);
  ps = fma(ps, x2, -1.0 / 362880.0);
  ps = fma(ps, x2, 1.0 / 5040.0);
  ps = fma(ps, x2, -1.0 / 120.0);
  ps = fma(ps, x2, 1.0 / 6.0);
  *s = fma(-(x * x2), ps, x);
  double pc = 1.0 / 6402373705728000.0;
  pc = fma(pc, x2, -1.0 / 20922789888000.0);
  pc = fma(pc, x2, 1.0 / 87178291200.0);
  pc = fma(pc, x2, -1.0 / 479001600.0);
  pc = fma(pc, x2, 1.0 / 3628800.0);
  pc = fma(pc, x2, -1.0 / 40320.0);
  pc = fma(pc, x2, 1.0 / 720.0);
  pc = fma(pc, x2, -1.0 / 24.0);
  pc = fma(pc, x2, 0.5);
  *c = fma(-x2, pc, 1.0);
}

__device__ __forceinline__ void quat_normalize_fast(fb::Quat &q) {  // normalizeRotation (se3quat.h:280-285) with one reciprocal
  if (q.w < 0) { q.x = -q.x; q.y = -q.y; q.z = -q.z; q.w = -q.w; }
  const double inv = rsqrt_fast(q.x * q.x + q.y * q.y + q.z * q.z + q.w * q.w);
  q.x *= inv; q.y *= inv; q.z *= inv; q.w *= inv;
}

// fb::ldlt6 with the pivots' reciprocals from rcp_fast (this routine is one lane's serial work inside every LM trial)
__device__ __forceinline__ bool ldlt6_fast(const double H[36], double lambda, const double b[6], double x[6]) {
  double A[36], d[6], dinv[6], y[6];
#pragma unroll
  for (int i = 0; i < 36; i++) A[i] = H[i];
#pragma unroll
  for (int j = 0; j < 6; j++) A[j * 6 + j] += lambda;
  bool ok = true;
#pragma unroll
  for (int j = 0; j < 6; j++) {
    double dj = A[j * 6 + j];
#pragma unroll
    for (int k = 0; k < j; k++) dj -= A[j * 6 + k] * A[j * 6 + k] * d[k];
    if (dj < 0) ok = false;
    d[j] = dj;
    const double inv = dj != 0 ? rcp_fast(dj) : 0.0;
    dinv[j] = inv;
#pragma unroll
    for (int i = j + 1; i < 6; i++) {
      double s = A[i * 6 + j];
#pragma unroll
      for (int k = 0; k < j; k++) s -= A[i * 6 + k] * A[j * 6 + k] * d[k];
      A[i * 6 + j] = s * inv;
    }
  }
  if (!ok) return false;
#pragma unroll
  for (int i = 0; i < 6; i++) {
    double s = b[i];
#pragma unroll
    for (int k = 0; k < i; k++) s -= A[i * 6 + k] * y[k];
    y[i] = s;
  }
#pragma unroll
  for (int i = 0; i < 6; i++) y[i] = y[i] * dinv[i];
#pragma unroll
  for (int i = 5; i >= 0; i--) {
    double s = y[i];
#pragma unroll
    for (int k = i + 1; k < 6; k++) s -= A[k * 6 + i] * x[k];
    x[i] = s;
  }
  return true;
}

// SE3Quat::exp (se3quat.h:223-257) without the detour over R: q = (sin(t/2) w/t, cos(t/2)), translation V u with
// V = I + b [w]x + d [w]x^2 written as cross products.  The reference's small-angle branch (R = I + [w]x + [w]x^2, sic)
// is kept verbatim through fb::se3_exp.
__device__ __forceinline__ fb::SE3 se3_exp_direct(const double u[6]) {
  const double wx = u[0], wy = u[1], wz = u[2];
  const double theta = sqrt(wx * wx + wy * wy + wz * wz);
  if (theta < 0.00001) return fb::se3_exp(u);
  double sh, ch;
  sincos_small(0.5 * theta, &sh, &ch);
  const double it = rcp_fast(theta), it2 = it * it;
  const double k = sh * it;
  fb::SE3 T;
  T.r.x = wx * k; T.r.y = wy * k; T.r.z = wz * k; T.r.w = ch;
  const double sn = 2.0 * sh * ch;                        // sin(theta)
  const double b = 2.0 * sh * sh * it2;                   // (1 - cos(theta)) / theta^2
  const double d = (theta - sn) * (it2 * it);             // (theta - sin(theta)) / theta^3
  const double c1x = wy * u[5] - wz * u[4], c1y = wz * u[3] - wx * u[5], c1z = wx * u[4] - wy * u[3];
  const double c2x = wy * c1z - wz * c1y, c2y = wz * c1x - wx * c1z, c2z = wx * c1y - wy * c1x;
  T.t[0] = u[3] + b * c1x + d * c2x;
  T.t[1] = u[4] + b * c1y + d * c2y;
  T.t[2] = u[5] + b * c1z + d * c2z;
  quat_normalize_fast(T.r);
  return T;
}

__device__ __forceinline__ fb::SE3 se3_mul_fast(const fb::SE3 &a, const fb::SE3 &b) {  // se3quat.h:104-110
  fb::SE3 r = a;
  double rt[3];
  fb::quat_rotate(a.r, b.t, rt);
  r.t[0] += rt[0]; r.t[1] += rt[1]; r.t[2] += rt[2];
  r.r = fb::quat_mul(a.r, b.r);
  quat_normalize_fast(r.r);
  return r;
}

__device__ __forceinline__ void front_edge_acc(const fb::SE3 &T, float x0, float x1, float x2, float o0, float o1, double info,
                                               bool robust, double delta, double fx, double fy, double cx, double cy,
                                               double (&acc)[NACC]) {
  const double Xw[3] = {x0, x1, x2};
  double p[3];
  fb::se3_map(T, Xw, p);
  const double X = p[0], Y = p[1], invz = rcp_fast(p[2]), invz_2 = invz * invz;
  const double err[2] = {(double)o0 - ((X * invz) * fx + cx), (double)o1 - ((Y * invz) * fy + cy)};
  const double J[2][6] = {{X * Y * invz_2 * fx, -(1 + (X * X * invz_2)) * fx, Y * invz * fx, -invz * fx, 0, X * invz_2 * fx},
                          {(1 + Y * Y * invz_2) * fy, -X * Y * invz_2 * fy, -X * invz * fy, 0, -invz * fy, Y * invz_2 * fy}};
  accumulate_edge<2, (1u << 4) | (1u << 9)>(J, err, info, robust, delta, acc);
}

__device__ __forceinline__ void bird_edge_acc(const fb::SE3 &T, float x0, float x1, float x2, float c0, float c1, float c2, double info,
                                              bool robust, double delta, double (&acc)[NACC]) {
  const double Xw[3] = {x0, x1, x2};
  double p[3];
  fb::se3_map(T, Xw, p);
  const double err[3] = {(double)c0 - p[0], (double)c1 - p[1], (double)c2 - p[2]};
  accumulate_bird_edge(p, err, info, robust, delta, acc);
}

// decision chi2 from register-resident edge data (same unfused arithmetic as edge_chi2 above)
__device__ __forceinline__ double chi2_front_vals(float x0, float x1, float x2, float o0, float o1, double info, const fb::SE3 &T,
                                                  double fx, double fy, double cx, double cy) {
#pragma clang fp contract(off)
  const double v0 = x0, v1 = x1, v2 = x2;
  const double qx = T.r.x, qy = T.r.y, qz = T.r.z, qw = T.r.w;
  double u0 = qy * v2 - qz * v1, u1 = qz * v0 - qx * v2, u2 = qx * v1 - qy * v0;
  u0 += u0; u1 += u1; u2 += u2;
  const double p0 = (v0 + qw * u0 + (qy * u2 - qz * u1)) + T.t[0];
  const double p1 = (v1 + qw * u1 + (qz * u0 - qx * u2)) + T.t[1];
  const double p2 = (v2 + qw * u2 + (qx * u1 - qy * u0)) + T.t[2];
  const double e0 = (double)o0 - ((p0 / p2) * fx + cx);
  const double e1 = (double)o1 - ((p1 / p2) * fy + cy);
  double s = 0;
  s += e0 * (info * e0);
  s += e1 * (info * e1);
  return s;
}
__device__ __forceinline__ double chi2_bird_vals(float x0, float x1, float x2, float c0, float c1, float c2, double info, const fb::SE3 &T) {
#pragma clang fp contract(off)
  const double v0 = x0, v1 = x1, v2 = x2;
  const double qx = T.r.x, qy = T.r.y, qz = T.r.z, qw = T.r.w;
  double u0 = qy * v2 - qz * v1, u1 = qz * v0 - qx * v2, u2 = qx * v1 - qy * v0;
  u0 += u0; u1 += u1; u2 += u2;
  const double p0 = (v0 + qw * u0 + (qy * u2 - qz * u1)) + T.t[0];
  const double p1 = (v1 + qw * u1 + (qz * u0 - qx * u2)) + T.t[1];
  const double p2 = (v2 + qw * u2 + (qx * u1 - qy * u0)) + T.t[2];
  const double d0 = (double)c0 - p0, d1 = (double)c1 - p1, d2 = (double)c2 - p2;
  double s = 0;
  s += d0 * (info * d0);
  s += d1 * (info * d1);
  s += d2 * (info * d2);
  return s;
}

// EF / EB = front / bird edge slots per thread.  <512, 5, 3> (up to 2560 + 1536 edges: the 2000-feature extractor, capacity
// 2064) keeps acc[28] + the edges + the per-edge temporaries inside the 256 registers two waves per SIMD leave a thread;
// <512, 8, 4> (up to 4096 + 2048 edges, the 4000-feature initialisation extractor) spills a little.
template <int NT, int EF, int EB>
__global__ __launch_bounds__(NT) void k_pose_opt_reg(fb_pose_opt_args A) {
  constexpr int CPA = NT / 32;   // threads that share one accumulator in the column sums (each adds 32 values)
  constexpr int RS = NT + CPA;   // row stride in doubles: 2 * RS mod 64 = 2 * CPA, so the CPA-wide windows of the accumulators
                                 // handled by one ds_read_b64 lane group fall into disjoint banks; the row writes are contiguous
  constexpr int NWR = NT / 64;
  extern __shared__ __attribute__((aligned(16))) double s_part[];  // [NACC][RS]
  __shared__ PoseLds S;
  __shared__ int s_cnt[2];
  __shared__ int s_wcnt[2][NWR];
  const int bidx = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
#ifdef FB_POSE_STAMPS
  const unsigned long long pose_t_start = __builtin_amdgcn_s_memtime();
#endif
  const int mode = A.mode;
  const size_t fo = (size_t)bidx * A.front_stride, bo = (size_t)bidx * A.bird_stride;
  const int nfs = (mode != FB_POSE_BIRD) ? A.n_front[bidx] : 0;
  const int nbs = (mode != FB_POSE_FRONT) ? A.n_bird[bidx] : 0;
  float *Tcw = A.Tcw + (size_t)bidx * 12;
  // ---- edge construction (Optimizer.cc:525-602).  The slots with an edge are compacted (rank among the valid slots, in
  //      slot order) so that a thread's registers hold real edges only: wave w scans the contiguous range of slots
  //      [w * per, (w + 1) * per), pass 1 counts, pass 2 writes the slot index of the edge with rank r to s_idx[r].
  unsigned short *s_idxF = reinterpret_cast<unsigned short *>(s_part);   // [EF * NT]
  unsigned short *s_idxB = s_idxF + EF * NT;                             // [EB * NT]
  const int perF = (((nfs + NWR - 1) / NWR) + 63) & ~63, perB = (((nbs + NWR - 1) / NWR) + 63) & ~63;
  {
    int cf = 0, cb = 0;
    for (int i = wv * perF + lane; i < min((wv + 1) * perF, nfs); i += 64) cf += (!A.front_valid || A.front_valid[fo + i]) ? 1 : 0;
    for (int i = wv * perB + lane; i < min((wv + 1) * perB, nbs); i += 64) cb += (!A.bird_valid || A.bird_valid[bo + i]) ? 1 : 0;
    cf = (int)wave_sum((double)cf); cb = (int)wave_sum((double)cb);
    if (lane == 0) { s_wcnt[0][wv] = cf; s_wcnt[1][wv] = cb; }
  }
  __syncthreads();
  int nf = 0, nb = 0, baseF = 0, baseB = 0;
#pragma unroll
  for (int w = 0; w < NWR; w++) {
    if (w < wv) { baseF += s_wcnt[0][w]; baseB += s_wcnt[1][w]; }
    nf += s_wcnt[0][w]; nb += s_wcnt[1][w];
  }
  if (nf > EF * NT || nb > EB * NT || A.front_stride > 65535 || A.bird_stride > 65535) {
    // more edges than register slots: the generic schedule, edges read from HBM / L2 (the level bytes take the LDS)
    __syncthreads();
    pose_generic<NT>(A, 0, reinterpret_cast<uint8_t *>(s_part), S, s_cnt);
    return;
  }
  if (mode == FB_POSE_BIRD ? nb < 3 : nf < 3) {  // Optimizer.cc:379,607,776
    // (the edge slots' mvbOutlier entries are still cleared, as the edge construction loop does before the count is known)
    for (int i = tid; i < nfs; i += NT)
      if (!A.front_valid || A.front_valid[fo + i]) A.front_outlier[fo + i] = 0;
    if (tid == 0) A.ninliers[bidx] = 0;
    return;
  }
  for (int i0 = wv * perF; i0 < min((wv + 1) * perF, nfs); i0 += 64) {
    const int i = i0 + lane;
    const bool v = i < min((wv + 1) * perF, nfs) && (!A.front_valid || A.front_valid[fo + i]);
    const unsigned long long m = __ballot(v);
    if (v) s_idxF[baseF + (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u))] = (unsigned short)i;
    baseF += __popcll(m);
  }
  for (int i0 = wv * perB; i0 < min((wv + 1) * perB, nbs); i0 += 64) {
    const int i = i0 + lane;
    const bool v = i < min((wv + 1) * perB, nbs) && (!A.bird_valid || A.bird_valid[bo + i]);
    const unsigned long long m = __ballot(v);
    if (v) s_idxB[baseB + (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u))] = (unsigned short)i;
    baseB += __popcll(m);
  }
  __syncthreads();
  // this thread's edges -> registers: edge with rank r = s * NT + tid.  level 0 = active, 1 = outlier level, 2 = no edge
  // (two bits per slot); fidx / bidx_ = the frame slot the edge came from (where its outlier flag lives)
  float fx0[EF], fx1[EF], fx2[EF], fo0[EF], fo1[EF], fin[EF];
  float bx0[EB], bx1[EB], bx2[EB], bc0[EB], bc1[EB], bc2[EB], bin[EB];
  int fidx[EF], bidx_[EB];
  unsigned flev = 0, blev = 0, fout = 0, bout = 0;
#pragma unroll
  for (int s = 0; s < EF; s++) {
    const int r = s * NT + tid;
    const bool v = r < nf;
    fx0[s] = fx1[s] = fx2[s] = fo0[s] = fo1[s] = fin[s] = 0.f;
    fidx[s] = 0;
    if (v) {
      const int e = s_idxF[r];
      fidx[s] = e;
      const float *X = A.front_xw + (fo + e) * 3, *O = A.front_obs + (fo + e) * 2;
      fx0[s] = X[0]; fx1[s] = X[1]; fx2[s] = X[2]; fo0[s] = O[0]; fo1[s] = O[1];
      fin[s] = A.front_inv_sigma2[fo + e];
      A.front_outlier[fo + e] = 0;
    }
    flev |= (v ? 0u : 2u) << (2 * s);
  }
#pragma unroll
  for (int s = 0; s < EB; s++) {
    const int r = s * NT + tid;
    const bool v = r < nb;
    bx0[s] = bx1[s] = bx2[s] = bc0[s] = bc1[s] = bc2[s] = bin[s] = 0.f;
    bidx_[s] = 0;
    if (v) {
      const int e = s_idxB[r];
      bidx_[s] = e;
      const float *X = A.bird_xw + (bo + e) * 3, *Cc = A.bird_xc + (bo + e) * 3;
      bx0[s] = X[0]; bx1[s] = X[1]; bx2[s] = X[2]; bc0[s] = Cc[0]; bc1[s] = Cc[1]; bc2[s] = Cc[2];
      bin[s] = A.bird_inv_sigma2[bo + e];
      if (A.bird_outlier[bo + e]) bout |= 1u << s;  // the incoming mvBirdOutlier decides which chi2 the first round recomputes
    }
    blev |= (v ? 0u : 2u) << (2 * s);
  }
  __syncthreads();  // s_idx lives in the buffer the evaluations overwrite
  const double wf = (mode == FB_POSE_FRONT) ? 1.0 : (double)A.wF, wb = (double)A.wB;
  const double fx = A.fx, fy = A.fy, cx = A.cx, cy = A.cy;
  const double delta = (double)(float)sqrt(5.991);
  const float chi2Mono = (mode == FB_POSE_FRONT) ? 5.991f : 1.5f;
  const float chi2Bird = 5.991f;
  if (tid == 0) {
    S.T = fb::se3_from_float12(Tcw);
    S.Teval = S.T;
  }
  __syncthreads();
  const fb::SE3 T0 = S.T;

  // one evaluation at pose T: robust chi2 + H + b over this thread's active edges, then the 28 column sums -> S.red
  auto eval = [&](const fb::SE3 &T, bool robust) {
    double acc[NACC];
#pragma unroll
    for (int i = 0; i < NACC; i++) acc[i] = 0;
    POSE_T0()
    POSE_COUNT(15)
#pragma unroll
    for (int s = 0; s < EF; s++)
      if (((flev >> (2 * s)) & 3u) == 0u) {
        // (opaque copies: otherwise the float -> double conversions and the weight product of every edge are hoisted out of
        // the LM loop and held in registers -- three times the registers of the floats themselves, i.e. spills)
        float a0 = fx0[s], a1 = fx1[s], a2 = fx2[s], a3 = fo0[s], a4 = fo1[s], a5 = fin[s];
        asm volatile("" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5));
        front_edge_acc(T, a0, a1, a2, a3, a4, (double)a5 * wf, robust, delta, fx, fy, cx, cy, acc);
      }
    POSE_TICK(0)
#pragma unroll
    for (int s = 0; s < EB; s++)
      if (((blev >> (2 * s)) & 3u) == 0u) {
        float a0 = bx0[s], a1 = bx1[s], a2 = bx2[s], a3 = bc0[s], a4 = bc1[s], a5 = bc2[s], a6 = bin[s];
        asm volatile("" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6));
        bird_edge_acc(T, a0, a1, a2, a3, a4, a5, (double)a6 * wb, robust, delta, acc);
      }
    POSE_TICK(1)
#pragma unroll
    for (int i = 0; i < NACC; i++) s_part[i * RS + tid] = acc[i];
    __syncthreads();
    POSE_TICK(2)
    if (tid < NACC * CPA) {
      const int a = tid / CPA, c = tid - a * CPA;
      const double *row = s_part + a * RS + c;
      double s0 = 0, s1 = 0, s2 = 0, s3 = 0;
#pragma unroll
      for (int k = 0; k < 32; k += 4) {
        s0 += row[(k + 0) * CPA]; s1 += row[(k + 1) * CPA]; s2 += row[(k + 2) * CPA]; s3 += row[(k + 3) * CPA];
      }
      double v = (s0 + s1) + (s2 + s3);
      // the CPA partial sums of accumulator a sit in CPA adjacent lanes of one 16-lane DPP row: shift-add towards the last lane
      if (CPA == 16) v += dpp_f64<0x118>(v);  // row_shr:8
      v += dpp_f64<0x114>(v);                 // row_shr:4
      v += dpp_f64<0x112>(v);                 // row_shr:2
      v += dpp_f64<0x111>(v);                 // row_shr:1
      if (c == CPA - 1) S.red[a] = v;
    }
    __syncthreads();
    POSE_TICK(3)
  };

  int nBad = 0, nBadBird = 0;
  for (int it = 0; it < 4; it++) {
    const bool robust = it < 3;
    if (tid == 0) S.T = T0;
    bool mine = false;
#pragma unroll
    for (int s = 0; s < EF; s++) mine |= ((flev >> (2 * s)) & 3u) == 0u;
#pragma unroll
    for (int s = 0; s < EB; s++) mine |= ((blev >> (2 * s)) & 3u) == 0u;
    const int nact = __syncthreads_count(mine);  // also publishes S.T
    if (nact > 0) {
      eval(S.T, robust);
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
          {
            POSE_T0()
            if (tid == 0) {
              S.ok2 = ldlt6_fast(S.H, lambda, S.b, S.x) ? 1 : 0;
              POSE_TICK(4)
              S.Ttrial = se3_mul_fast(se3_exp_direct(S.x), S.T);  // oplus
              S.Teval = S.Ttrial;
              POSE_TICK(5)
            }
            __syncthreads();
            POSE_TICK(6)
          }
          eval(S.Ttrial, robust);
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
        if (qmax == 10 || rho == 0) break;
        if ((iniChi - currentChi) * 1e3 < iniChi) nBadLM++;
        else nBadLM = 0;
        if (nBadLM >= 3) break;
      }
    }
    // ---- classify (Optimizer.cc:396-431, 627-686, 791-822)
    POSE_T0()
    const fb::SE3 T = S.T, Teval = S.Teval;
    int bad = 0, badb = 0;
#pragma unroll
    for (int s = 0; s < EF; s++) {
      if (((flev >> (2 * s)) & 3u) == 2u) continue;
      const bool wasOut = (fout >> s) & 1u;
      float a0 = fx0[s], a1 = fx1[s], a2 = fx2[s], a3 = fo0[s], a4 = fo1[s], a5 = fin[s];
      asm volatile("" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5));
      const float chi2 = (float)chi2_front_vals(a0, a1, a2, a3, a4, (double)a5 * wf, wasOut ? T : Teval, fx, fy, cx, cy);
      bool isBad;
      if (mode == FB_POSE_FRONT) isBad = chi2 > chi2Mono;
      else isBad = chi2 > chi2Mono * ((double)A.wF + 1e-9);
      A.front_outlier[fo + fidx[s]] = isBad ? 1 : 0;
      flev = (flev & ~(3u << (2 * s))) | ((isBad ? 1u : 0u) << (2 * s));
      fout = (fout & ~(1u << s)) | ((isBad ? 1u : 0u) << s);
      bad += isBad;
    }
#pragma unroll
    for (int s = 0; s < EB; s++) {
      if (((blev >> (2 * s)) & 3u) == 2u) continue;
      const bool wasOut = (bout >> s) & 1u;
      float a0 = bx0[s], a1 = bx1[s], a2 = bx2[s], a3 = bc0[s], a4 = bc1[s], a5 = bc2[s], a6 = bin[s];
      asm volatile("" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6));
      const float chi2 = (float)chi2_bird_vals(a0, a1, a2, a3, a4, a5, (double)a6 * wb, wasOut ? T : Teval);
      const float chi2Bad = (float)(chi2Bird * ((double)A.wB + 1e-9));
      const bool isBad = chi2 > chi2Bad;
      A.bird_outlier[bo + bidx_[s]] = isBad ? 1 : 0;
      blev = (blev & ~(3u << (2 * s))) | ((isBad ? 1u : 0u) << (2 * s));
      bout = (bout & ~(1u << s)) | ((isBad ? 1u : 0u) << s);
      badb += isBad;
    }
    {
      // per-wave counts through the ballots of each slot would cost 12 ballots; two wave sums + one LDS hop instead
      const double sb = wave_sum((double)bad), sbb = wave_sum((double)badb);
      if ((tid & 63) == 0) { S.part[0][tid >> 6] = sb; S.part[1][tid >> 6] = sbb; }
      __syncthreads();
      double tb = 0, tbb = 0;
      for (int w2 = 0; w2 < NWR; w2++) { tb += S.part[0][w2]; tbb += S.part[1][w2]; }
      nBad = (int)tb; nBadBird = (int)tbb;
      __syncthreads();
    }
    POSE_TICK(7)
    if (nf + nb < 10) break;  // optimizer.edges().size()<10
  }
  if (tid == 0) {
    fb::se3_to_float12(S.T, Tcw);
    A.ninliers[bidx] = (mode == FB_POSE_BIRD) ? nb - nBadBird : nf - nBad;
  }
#ifdef FB_POSE_STAMPS
  if (blockIdx.x == 0 && tid == 0) g_pose_stamps[14] += __builtin_amdgcn_s_memtime() - pose_t_start;
#endif
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

// ------------------------------------------------------------------------------------------
// k_pose_opt_split -- k_pose_opt_reg with the serial part of an LM step on a wave of its own.  In k_pose_opt_reg the 6x6
// solve and the exponential map run on lane 0 of a wave that also carries edges: the registers of the solve come on top of
// the edge registers and ~117 of them spill (704 B of scratch per lane, reloaded in every evaluation).  Here waves
// 0 .. NWE-1 carry the edges and the last wave does nothing but the solve: the two roles are two instantiations of one
// generic lambda, so each is register-allocated on its own (no edge register is live in the solver's code and vice versa)
// and they meet at the same barriers.  Same arithmetic, same order of every sum as k_pose_opt_reg.
// ------------------------------------------------------------------------------------------
template <int NT, int EF, int EB>
__global__ __launch_bounds__(NT) void k_pose_opt_split(fb_pose_opt_args A) {
  constexpr int NE = NT - 64;    // edge threads
  constexpr int NWE = NE / 64, NWR = NT / 64;
  constexpr int CPA = NE / NACC; // threads that share one accumulator in the column sums: NACC * CPA = NE, each adds NE / CPA values
  static_assert(CPA == 16 && NACC * CPA == NE, "the column sums are laid out for 448 edge threads");
  constexpr int RS = NE + CPA;   // row stride in doubles (2 * RS mod 64 = 2 * CPA: the CPA-wide windows fall into disjoint banks)
  extern __shared__ __attribute__((aligned(16))) double s_part[];  // [NACC][RS]
  __shared__ PoseLds S;
  __shared__ int s_cnt[2];
  __shared__ int s_wcnt[2][NWR];
  const int bidx = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int mode = A.mode;
  const size_t fo = (size_t)bidx * A.front_stride, bo = (size_t)bidx * A.bird_stride;
  const int nfs = (mode != FB_POSE_BIRD) ? A.n_front[bidx] : 0;
  const int nbs = (mode != FB_POSE_FRONT) ? A.n_bird[bidx] : 0;
  float *Tcw = A.Tcw + (size_t)bidx * 12;
  // ---- edge construction (Optimizer.cc:525-602): the slots with an edge are compacted in slot order (all NWR waves scan)
  unsigned short *s_idxF = reinterpret_cast<unsigned short *>(s_part);   // [EF * NE]
  unsigned short *s_idxB = s_idxF + EF * NE;                             // [EB * NE]
  const int perF = (((nfs + NWR - 1) / NWR) + 63) & ~63, perB = (((nbs + NWR - 1) / NWR) + 63) & ~63;
  {
    int cf = 0, cb = 0;
    for (int i = wv * perF + lane; i < min((wv + 1) * perF, nfs); i += 64) cf += (!A.front_valid || A.front_valid[fo + i]) ? 1 : 0;
    for (int i = wv * perB + lane; i < min((wv + 1) * perB, nbs); i += 64) cb += (!A.bird_valid || A.bird_valid[bo + i]) ? 1 : 0;
    cf = (int)wave_sum((double)cf); cb = (int)wave_sum((double)cb);
    if (lane == 0) { s_wcnt[0][wv] = cf; s_wcnt[1][wv] = cb; }
  }
  __syncthreads();
  int nf = 0, nb = 0, baseF = 0, baseB = 0;
#pragma unroll
  for (int w = 0; w < NWR; w++) {
    if (w < wv) { baseF += s_wcnt[0][w]; baseB += s_wcnt[1][w]; }
    nf += s_wcnt[0][w]; nb += s_wcnt[1][w];
  }
  if (nf > EF * NE || nb > EB * NE || A.front_stride > 65535 || A.bird_stride > 65535) {
    // more edges than register slots: the generic schedule, edges read from HBM / L2 (the level bytes take the LDS)
    __syncthreads();
    pose_generic<NT>(A, 0, reinterpret_cast<uint8_t *>(s_part), S, s_cnt);
    return;
  }
  if (mode == FB_POSE_BIRD ? nb < 3 : nf < 3) {  // Optimizer.cc:379,607,776
    for (int i = tid; i < nfs; i += NT)
      if (!A.front_valid || A.front_valid[fo + i]) A.front_outlier[fo + i] = 0;
    if (tid == 0) A.ninliers[bidx] = 0;
    return;
  }
  for (int i0 = wv * perF; i0 < min((wv + 1) * perF, nfs); i0 += 64) {
    const int i = i0 + lane;
    const bool v = i < min((wv + 1) * perF, nfs) && (!A.front_valid || A.front_valid[fo + i]);
    const unsigned long long m = __ballot(v);
    if (v) s_idxF[baseF + (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u))] = (unsigned short)i;
    baseF += __popcll(m);
  }
  for (int i0 = wv * perB; i0 < min((wv + 1) * perB, nbs); i0 += 64) {
    const int i = i0 + lane;
    const bool v = i < min((wv + 1) * perB, nbs) && (!A.bird_valid || A.bird_valid[bo + i]);
    const unsigned long long m = __ballot(v);
    if (v) s_idxB[baseB + (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u))] = (unsigned short)i;
    baseB += __popcll(m);
  }
  if (tid == NE) {
    S.T = fb::se3_from_float12(Tcw);
    S.Teval = S.T;
    S.T0 = S.T;
  }
  __syncthreads();

  auto role = [&](auto edgeRole) {
    constexpr bool EDGE = decltype(edgeRole)::value;
    // this thread's edges -> registers (edge waves only): edge with rank r = s * NE + tid.  level 0 = active, 1 = outlier level,
    // 2 = no edge (two bits per slot); fidx / bidx_ = the frame slot the edge came from (where its outlier flag lives)
    float fx0[EF], fx1[EF], fx2[EF], fo0[EF], fo1[EF], fin[EF];
    float bx0[EB], bx1[EB], bx2[EB], bc0[EB], bc1[EB], bc2[EB], bin[EB];
    int fidx[EF], bidx_[EB];
    unsigned flev = 0, blev = 0, fout = 0, bout = 0;
    if constexpr (EDGE) {
#pragma unroll
      for (int s = 0; s < EF; s++) {
        const int r = s * NE + tid;
        const bool v = r < nf;
        fx0[s] = fx1[s] = fx2[s] = fo0[s] = fo1[s] = fin[s] = 0.f;
        fidx[s] = 0;
        if (v) {
          const int e = s_idxF[r];
          fidx[s] = e;
          const float *X = A.front_xw + (fo + e) * 3, *O = A.front_obs + (fo + e) * 2;
          fx0[s] = X[0]; fx1[s] = X[1]; fx2[s] = X[2]; fo0[s] = O[0]; fo1[s] = O[1];
          fin[s] = A.front_inv_sigma2[fo + e];
          A.front_outlier[fo + e] = 0;
        }
        flev |= (v ? 0u : 2u) << (2 * s);
      }
#pragma unroll
      for (int s = 0; s < EB; s++) {
        const int r = s * NE + tid;
        const bool v = r < nb;
        bx0[s] = bx1[s] = bx2[s] = bc0[s] = bc1[s] = bc2[s] = bin[s] = 0.f;
        bidx_[s] = 0;
        if (v) {
          const int e = s_idxB[r];
          bidx_[s] = e;
          const float *X = A.bird_xw + (bo + e) * 3, *Cc = A.bird_xc + (bo + e) * 3;
          bx0[s] = X[0]; bx1[s] = X[1]; bx2[s] = X[2]; bc0[s] = Cc[0]; bc1[s] = Cc[1]; bc2[s] = Cc[2];
          bin[s] = A.bird_inv_sigma2[bo + e];
          if (A.bird_outlier[bo + e]) bout |= 1u << s;  // the incoming mvBirdOutlier decides which chi2 the first round recomputes
        }
        blev |= (v ? 0u : 2u) << (2 * s);
      }
    }
    __syncthreads();  // s_idx lives in the buffer the evaluations overwrite
    const double wf = (mode == FB_POSE_FRONT) ? 1.0 : (double)A.wF, wb = (double)A.wB;
    const double fx = A.fx, fy = A.fy, cx = A.cx, cy = A.cy;
    const double delta = (double)(float)sqrt(5.991);
    const float chi2Mono = (mode == FB_POSE_FRONT) ? 5.991f : 1.5f;
    const float chi2Bird = 5.991f;

    // one evaluation at pose *Tp: robust chi2 + H + b over this thread's active edges, then the 28 column sums -> S.red
    // (the solver wave only keeps the two barriers)
    auto eval = [&](const fb::SE3 *Tp, bool robust) {
      if constexpr (EDGE) {
        const fb::SE3 T = *Tp;
        double acc[NACC];
#pragma unroll
        for (int i = 0; i < NACC; i++) acc[i] = 0;
#pragma unroll
        for (int s = 0; s < EF; s++)
          if (((flev >> (2 * s)) & 3u) == 0u) {
            // (opaque copies: otherwise the float -> double conversions and the weight product of every edge are hoisted out
            // of the LM loop and held in registers -- three times the registers of the floats themselves)
            float a0 = fx0[s], a1 = fx1[s], a2 = fx2[s], a3 = fo0[s], a4 = fo1[s], a5 = fin[s];
            asm volatile("" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5));
            front_edge_acc(T, a0, a1, a2, a3, a4, (double)a5 * wf, robust, delta, fx, fy, cx, cy, acc);
          }
#pragma unroll
        for (int s = 0; s < EB; s++)
          if (((blev >> (2 * s)) & 3u) == 0u) {
            float a0 = bx0[s], a1 = bx1[s], a2 = bx2[s], a3 = bc0[s], a4 = bc1[s], a5 = bc2[s], a6 = bin[s];
            asm volatile("" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6));
            bird_edge_acc(T, a0, a1, a2, a3, a4, a5, (double)a6 * wb, robust, delta, acc);
          }
#pragma unroll
        for (int i = 0; i < NACC; i++) s_part[i * RS + tid] = acc[i];
      }
      __syncthreads();
      if constexpr (EDGE) {
        const int a = tid / CPA, c = tid - a * CPA;  // NACC * CPA = NE: every edge thread sums one window
        const double *row = s_part + a * RS + c;
        double s0 = 0, s1 = 0, s2 = 0, s3 = 0;
#pragma unroll
        for (int k = 0; k < NE / CPA; k += 4) {
          s0 += row[(k + 0) * CPA]; s1 += row[(k + 1) * CPA]; s2 += row[(k + 2) * CPA]; s3 += row[(k + 3) * CPA];
        }
        double v = (s0 + s1) + (s2 + s3);
        v += dpp_f64<0x118>(v);  // row_shr:8: the CPA partial sums of accumulator a sit in the 16 lanes of one DPP row
        v += dpp_f64<0x114>(v);  // row_shr:4
        v += dpp_f64<0x112>(v);  // row_shr:2
        v += dpp_f64<0x111>(v);  // row_shr:1
        if (c == CPA - 1) S.red[a] = v;
      }
      __syncthreads();
    };

    int nBad = 0, nBadBird = 0;
    for (int it = 0; it < 4; it++) {
      const bool robust = it < 3;
      if (!EDGE && tid == NE) S.T = S.T0;
      bool mine = false;
      if constexpr (EDGE) {
#pragma unroll
        for (int s = 0; s < EF; s++) mine |= ((flev >> (2 * s)) & 3u) == 0u;
#pragma unroll
        for (int s = 0; s < EB; s++) mine |= ((blev >> (2 * s)) & 3u) == 0u;
      }
      const int nact = __syncthreads_count(mine);  // also publishes S.T
      if (nact > 0) {
        eval(&S.T, robust);
        if (!EDGE && tid == NE) { unpack_system(S.red, S.H, S.b); S.Teval = S.T; }
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
            if (!EDGE && tid == NE) {
              S.ok2 = ldlt6_fast(S.H, lambda, S.b, S.x) ? 1 : 0;
              S.Ttrial = se3_mul_fast(se3_exp_direct(S.x), S.T);  // oplus
              S.Teval = S.Ttrial;
            }
            __syncthreads();
            eval(&S.Ttrial, robust);
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
              if (!EDGE && tid == NE) { S.T = S.Ttrial; unpack_system(S.red, S.H, S.b); }
            } else {
              lambda *= ni;
              ni *= 2;
            }
            __syncthreads();
            qmax++;
          } while (rho < 0 && qmax < 10);
          if (qmax == 10 || rho == 0) break;
          if ((iniChi - currentChi) * 1e3 < iniChi) nBadLM++;
          else nBadLM = 0;
          if (nBadLM >= 3) break;
        }
      }
      // ---- classify (Optimizer.cc:396-431, 627-686, 791-822)
      int bad = 0, badb = 0;
      if constexpr (EDGE) {
        const fb::SE3 T = S.T, Teval = S.Teval;
#pragma unroll
        for (int s = 0; s < EF; s++) {
          if (((flev >> (2 * s)) & 3u) == 2u) continue;
          const bool wasOut = (fout >> s) & 1u;
          float a0 = fx0[s], a1 = fx1[s], a2 = fx2[s], a3 = fo0[s], a4 = fo1[s], a5 = fin[s];
          asm volatile("" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5));
          const float chi2 = (float)chi2_front_vals(a0, a1, a2, a3, a4, (double)a5 * wf, wasOut ? T : Teval, fx, fy, cx, cy);
          bool isBad;
          if (mode == FB_POSE_FRONT) isBad = chi2 > chi2Mono;
          else isBad = chi2 > chi2Mono * ((double)A.wF + 1e-9);
          A.front_outlier[fo + fidx[s]] = isBad ? 1 : 0;
          flev = (flev & ~(3u << (2 * s))) | ((isBad ? 1u : 0u) << (2 * s));
          fout = (fout & ~(1u << s)) | ((isBad ? 1u : 0u) << s);
          bad += isBad;
        }
#pragma unroll
        for (int s = 0; s < EB; s++) {
          if (((blev >> (2 * s)) & 3u) == 2u) continue;
          const bool wasOut = (bout >> s) & 1u;
          float a0 = bx0[s], a1 = bx1[s], a2 = bx2[s], a3 = bc0[s], a4 = bc1[s], a5 = bc2[s], a6 = bin[s];
          asm volatile("" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6));
          const float chi2 = (float)chi2_bird_vals(a0, a1, a2, a3, a4, a5, (double)a6 * wb, wasOut ? T : Teval);
          const float chi2Bad = (float)(chi2Bird * ((double)A.wB + 1e-9));
          const bool isBad = chi2 > chi2Bad;
          A.bird_outlier[bo + bidx_[s]] = isBad ? 1 : 0;
          blev = (blev & ~(3u << (2 * s))) | ((isBad ? 1u : 0u) << (2 * s));
          bout = (bout & ~(1u << s)) | ((isBad ? 1u : 0u) << s);
          badb += isBad;
        }
        const double sb = wave_sum((double)bad), sbb = wave_sum((double)badb);
        if ((tid & 63) == 0) { S.part[0][tid >> 6] = sb; S.part[1][tid >> 6] = sbb; }
      }
      __syncthreads();
      {
        double tb = 0, tbb = 0;
        for (int w2 = 0; w2 < NWE; w2++) { tb += S.part[0][w2]; tbb += S.part[1][w2]; }
        nBad = (int)tb; nBadBird = (int)tbb;
      }
      __syncthreads();
      if (nf + nb < 10) break;  // optimizer.edges().size()<10
    }
    if (!EDGE && tid == NE) {
      fb::se3_to_float12(S.T, Tcw);
      A.ninliers[bidx] = (mode == FB_POSE_BIRD) ? nb - nBadBird : nf - nBad;
    }
  };
  if (wv < NWE) role(std::true_type{});
  else role(std::false_type{});
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
  // register-resident kernel whenever its in-kernel way out (level bytes of every slot in LDS) fits: by default
  // k_pose_opt_split (448 edge threads + a solver wave; 120 B of scratch per lane); FB_POSE_NT=512 / 256 / 0 selects
  // k_pose_opt_reg with 512 threads / with one wave per SIMD / the LDS-staged kernel (measurements only)
  // Without FB_POSE_NT, batches of 64 frames or more take the 256-thread kernel instead: alone it is 9 % slower
  // (0.38 vs 0.35 ms), but its workgroup leaves 148 registers per SIMD lane to the extractor kernels of the other streams
  // (the split kernel holds a CU's whole register file while it runs): +1.5 % pairs/s in the overlapped step.
  static const int envNT = [] { const char *e = getenv("FB_POSE_NT"); const int v = e ? atoi(e) : -1; return v < 0 || v == 256 || v == 512 || v == 448 ? v : 0; }();
  const int regNT = envNT >= 0 ? envNT : (A->batch >= 64 ? 256 : 448);
  if (regNT) {
    const size_t lds = regNT == 448 ? (size_t)NACC * (448 + 16) * sizeof(double) : (size_t)NACC * (regNT + regNT / 32) * sizeof(double);
    const size_t flagBytes = ((size_t)((A->front_stride + 15) & ~15)) + ((A->bird_stride + 15) & ~15);
    if (flagBytes <= lds && A->front_stride <= 65535 && A->bird_stride <= 65535) {
      fb::ProfScope prof_(fb::P_POSE, fb::as_stream(stream));
      if (regNT == 448) {  // 448 edge threads + the solver wave
        FB_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(k_pose_opt_split<512, 5, 3>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        k_pose_opt_split<512, 5, 3><<<A->batch, 512, lds, fb::as_stream(stream)>>>(*A);
      } else if (regNT == 512) {
        FB_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(k_pose_opt_reg<512, 5, 3>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        k_pose_opt_reg<512, 5, 3><<<A->batch, 512, lds, fb::as_stream(stream)>>>(*A);
      } else {
        FB_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(k_pose_opt_reg<256, 10, 6>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        k_pose_opt_reg<256, 10, 6><<<A->batch, 256, lds, fb::as_stream(stream)>>>(*A);
      }
      FB_HIP(hipGetLastError());
      return FB_OK;
    }
  }
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

#ifdef FB_POSE_STAMPS
// probe build only: copies and clears the 16 phase accumulators
int fb_pose_debug_stamps(uint64_t *dst16) {
  unsigned long long h[16], z[16] = {0};
  FB_HIP(hipMemcpyFromSymbol(h, HIP_SYMBOL(g_pose_stamps), sizeof(h)));
  FB_HIP(hipMemcpyToSymbol(HIP_SYMBOL(g_pose_stamps), z, sizeof(z)));
  for (int i = 0; i < 16; i++) dst16[i] = h[i];
  return FB_OK;
}
#endif

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
  // one staged upload / download (fb::Stager): 13 synchronous copies were a quarter of this call at one frame per call
  fb::Stager st;
#define UPB(field, bytes) st.in((void **)&D.field, H->field, (bytes));
  UPB(n_front, B * 4) UPB(front_xw, B * fs * 12) UPB(front_obs, B * fs * 8) UPB(front_inv_sigma2, B * fs * 4)
  UPB(front_valid, B * fs) UPB(n_bird, B * 4) UPB(bird_xw, B * bs * 12) UPB(bird_xc, B * bs * 12)
  UPB(bird_inv_sigma2, B * bs * 4) UPB(bird_valid, B * bs)
#undef UPB
  st.out((void **)&D.bird_outlier, H->bird_outlier, B * bs, true);   // in/out: mvBirdOutlier
  st.out((void **)&D.Tcw, H->Tcw, B * 48, true);
  st.out((void **)&D.front_outlier, H->front_outlier, B * fs, true);
  st.out((void **)&D.ninliers, H->ninliers, B * 4, false);
  // the family a mode ignores may be absent altogether: the kernel never dereferences it (n = 0 for that family)
  FB_TRY(st.commit(nullptr));
  FB_TRY(fb_pose_opt_batch_dev(&D, nullptr));
  return st.fetch(nullptr);
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
