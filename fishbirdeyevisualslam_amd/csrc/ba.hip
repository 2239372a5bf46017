// ba.hip -- Optimizer::LocalBundleAdjustment / LocalBundleAdjustmentWithOdom on gfx950.
//
// Replaces (reference file:line):
//   Optimizer::LocalBundleAdjustment            src/Optimizer.cc:838-1165
//   Optimizer::LocalBundleAdjustmentWithOdom    src/Optimizer.cc:2137-2670
//   EdgeSE3ProjectXYZ                           Thirdparty/g2o/g2o/types/types_six_dof_expmap.cpp:103-147
//   EdgeSE3ProjectXYZ2UVQuat / 2XYZQuat / EdgeSE3Quat   src/OdomG2oTypeQuat.cc:109-212
//   BaseBinaryEdge::constructQuadraticForm      Thirdparty/g2o/g2o/core/base_binary_edge.hpp:55-120
//   BlockSolver::solve (Schur complement)       core/block_solver.hpp:354-486
//   OptimizationAlgorithmLevenberg::solve       core/optimization_algorithm_levenberg.cpp:61-164
//
// Kernels per LM evaluation (all fp64 like g2o):
//   k_ba_linearize  one lane per landmark: residuals, Huber weights, Hll (3x3), bl, and the 6x3
//                   pose-landmark blocks W_e of its edges; per-block partial chi2
//   k_ba_pose       one workgroup per free keyframe: Hpp diagonal block + bp (wave-shuffle reduce)
//   k_ba_odom       pose-pose EdgeSE3Quat blocks (few edges) added into the dense Hpp
// and per LM trial (given lambda):
//   k_ba_schur      THE MFMA KERNEL: S_part = sum_l (W_l D_l^-1) W_l^T and the reduced rhs, as dense
//                   LDS panels (6n x 3c per chunk of c landmarks) multiplied with
//                   v_mfma_f64_16x16x4_f64; upper-triangular 16x16 tiles only, like g2o
//   k_ba_solve      S = Hpp + lambda I - sum S_part, dense LDL^T in LDS, triangular solves in one wave
//   k_ba_update     landmark back-substitution, exp-map pose update, trial state, scale term
// The accept/reject logic of Levenberg-Marquardt runs on the host between trials (a few scalars
// are read back per trial); the abort flag (pbStopFlag) is polled there, between iterations.
#include "fb_common.h"

#include <thread>
#include "fb_se3.h"

#include <algorithm>
#include <chrono>
#include <cstdlib>

namespace {

using fb::SE3;

constexpr int T_PROJ = 0, T_XYZ = 1;
constexpr int LIN_THREADS = 128;
constexpr int POSE_THREADS = 256;
constexpr int SCHUR_THREADS = 256;
constexpr int SOLVE_THREADS = 256;
constexpr int CHUNK = 16;  // landmarks per MFMA panel (K = 48)
constexpr int KP = 3 * CHUNK;
constexpr int KPAD = KP + 2;  // LDS row stride in doubles (breaks the 2-way bank conflict of stride 48)

struct BADev {  // device pointers + sizes, passed by value
  int n_kf, np, npt, nE, nO, quat;
  double fx, fy, cx, cy, delta;
  const int *poseIdx;          // [n_kf] free index or -1
  const int *e_pt, *e_kf;      // [nE]
  const uint8_t *e_type;       // [nE]
  const int *e_pj;             // [nE] poseIdx[e_kf[e]] (one dependent load less on the landmark-centric paths)
  const float *e_meas;         // [nE][3] (the measurements arrive as floats: kept so, half the staging bytes)
  const double *e_info;        // [nE]
  uint8_t *e_level;            // [nE]
  double *e_chi2;              // [nE] chi2 of the last evaluation that covered the edge
  const int *lm_start, *lm_edges;  // CSR by landmark
  const int *ps_start, *ps_edges;  // CSR by free pose
  const int *o_i, *o_j;        // [nO]
  const SE3 *o_Zinv;           // [nO]
  const double *o_info;        // [nO]
  const int *od_start, *od_edges;  // CSR by free pose: incident odometry edges, ascending edge index
};

struct LinBuf {  // linearisation at one state
  double *Hll;   // [npt][9]
  double *bl;    // [npt][3]
  double *W;     // [nE][18]
  double *Hpp;   // [P6][P6] dense
  double *bp;    // [P6]
  double *chiPart;  // [nLinBlocks + 1] (last = odom chi2)
  double *maxPart;  // [nLinBlocks] max |diag Hll| of the block's landmarks (lambda_0 = 1e-5 max diag; device-resident schedule)
};

struct State {
  SE3 *pose;     // [n_kf]
  double *pt;    // [npt][3]
};

// reciprocal from the hardware seed + two Newton steps (relative error < 2^-50) for the pivots of the reduced pose system:
// the BA is held to 1e-4 on poses and landmarks, and an IEEE division is ~35 dependent instructions on the critical path
// of every block step of k_ba_solve
__device__ __forceinline__ double ba_rcp(double x) {
  double r = __builtin_amdgcn_rcp(x);
  r = fma(fma(-x, r, 1.0), r, r);
  r = fma(fma(-x, r, 1.0), r, r);
  return r;
}

__device__ __forceinline__ double wave_sum_d(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

// residual + Jacobians of one point-pose edge (vertex 0 = point, vertex 1 = pose)
struct EdgeLin {
  double err[3];
  double Ji[3][3];  // d err / d point
  double Jj[3][6];  // d err / d pose
};

__device__ __forceinline__ void edge_residual(const BADev &D, int type, const SE3 &T, const double *X, const double *meas,
                                              double p[3], double err[3]) {
  fb::se3_map(T, X, p);
  if (type == T_PROJ) {
    if (D.quat) {  // OdomG2oTypeQuat.cc:138-144
      err[0] = meas[0] - (D.fx * p[0] / p[2] + D.cx);
      err[1] = meas[1] - (D.fy * p[1] / p[2] + D.cy);
    } else {       // types_six_dof_expmap.cpp:141-147
      err[0] = meas[0] - ((p[0] / p[2]) * D.fx + D.cx);
      err[1] = meas[1] - ((p[1] / p[2]) * D.fy + D.cy);
    }
    err[2] = 0;
  } else {
    err[0] = meas[0] - p[0]; err[1] = meas[1] - p[1]; err[2] = meas[2] - p[2];
  }
}

__device__ __forceinline__ void edge_jacobians(const BADev &D, int type, const SE3 &T, const double p[3], EdgeLin &L) {
  double R[9];
  fb::quat_to_R(T.r, R);
  const double X = p[0], Y = p[1], Z = p[2];
  if (type == T_PROJ) {
    if (D.quat) {  // EdgeSE3ProjectXYZ2UVQuat::linearizeOplus, OdomG2oTypeQuat.cc:109-129
      const double z2 = Z * Z;
      const double jep[2][3] = {{-(D.fx / Z), -0.0, -(-D.fx * X / z2)}, {-0.0, -(D.fy / Z), -(-D.fy * Y / z2)}};
      const double jpk[3][6] = {{-0.0, Z, -Y, 1, 0, 0}, {-Z, -0.0, X, 0, 1, 0}, {Y, -X, -0.0, 0, 0, 1}};  // [-skew(p), I]
#pragma unroll
      for (int r = 0; r < 2; r++) {
#pragma unroll
        for (int c = 0; c < 6; c++) L.Jj[r][c] = jep[r][0] * jpk[0][c] + jep[r][1] * jpk[1][c] + jep[r][2] * jpk[2][c];
#pragma unroll
        for (int c = 0; c < 3; c++) L.Ji[r][c] = jep[r][0] * R[c] + jep[r][1] * R[3 + c] + jep[r][2] * R[6 + c];
      }
    } else {  // EdgeSE3ProjectXYZ::linearizeOplus, types_six_dof_expmap.cpp:103-139
      const double z_2 = Z * Z;
      const double tmp[2][3] = {{D.fx, 0, -X / Z * D.fx}, {0, D.fy, -Y / Z * D.fy}};
      const double s = -1. / Z;
#pragma unroll
      for (int r = 0; r < 2; r++)
#pragma unroll
        for (int c = 0; c < 3; c++) L.Ji[r][c] = (s * tmp[r][0]) * R[c] + (s * tmp[r][1]) * R[3 + c] + (s * tmp[r][2]) * R[6 + c];
      L.Jj[0][0] = X * Y / z_2 * D.fx; L.Jj[0][1] = -(1 + (X * X / z_2)) * D.fx; L.Jj[0][2] = Y / Z * D.fx;
      L.Jj[0][3] = -1. / Z * D.fx;     L.Jj[0][4] = 0;                            L.Jj[0][5] = X / z_2 * D.fx;
      L.Jj[1][0] = (1 + Y * Y / z_2) * D.fy; L.Jj[1][1] = -X * Y / z_2 * D.fy;   L.Jj[1][2] = -X / Z * D.fy;
      L.Jj[1][3] = 0;                  L.Jj[1][4] = -1. / Z * D.fy;               L.Jj[1][5] = Y / z_2 * D.fy;
    }
#pragma unroll
    for (int c = 0; c < 6; c++) L.Jj[2][c] = 0;
#pragma unroll
    for (int c = 0; c < 3; c++) L.Ji[2][c] = 0;
  } else {  // EdgeSE3ProjectXYZ2XYZQuat::linearizeOplus, OdomG2oTypeQuat.cc:157-169
    const double jj[3][6] = {{0, -Z, Y, -1, -0.0, -0.0}, {Z, 0, -X, -0.0, -1, -0.0}, {-Y, X, 0, -0.0, -0.0, -1}};
#pragma unroll
    for (int r = 0; r < 3; r++) {
#pragma unroll
      for (int c = 0; c < 6; c++) L.Jj[r][c] = jj[r][c];
#pragma unroll
      for (int c = 0; c < 3; c++) L.Ji[r][c] = -R[r * 3 + c];
    }
  }
}

// one point-pose edge of the landmark at X: chi2, Huber weight, its terms of Hll / bl, and its 6x3 block W
__device__ __forceinline__ void lin_edge(const BADev &D, const LinBuf &B, const double (&X)[3], int e, int kf, int lvl, int type, double info,
                                         const double (&measv)[3], const SE3 &T, int robust, double (&H)[9], double (&b3)[3], double &chi) {
  double *W = B.W + (size_t)e * 18;
  const int pj = D.poseIdx[kf];
  if (lvl != 0) {
#pragma unroll
    for (int i = 0; i < 18; i++) W[i] = 0;
    return;
  }
  const double meas[3] = {measv[0], measv[1], measv[2]};
  double p[3];
  EdgeLin L;
  edge_residual(D, type, T, X, meas, p, L.err);
  edge_jacobians(D, type, T, p, L);
  double chi2 = 0;
#pragma unroll
  for (int r = 0; r < 3; r++) chi2 += L.err[r] * (info * L.err[r]);
  D.e_chi2[e] = chi2;
  double rho0 = chi2, rho1 = 1.;
  if (robust) fb::huber(chi2, D.delta, rho0, rho1);
  chi += rho0;
  const double w = rho1 * info;
  double orr[3];
#pragma unroll
  for (int r = 0; r < 3; r++) orr[r] = -(info * L.err[r]) * rho1;
#pragma unroll
  for (int i = 0; i < 3; i++) {
    double s = 0;
#pragma unroll
    for (int r = 0; r < 3; r++) s += L.Ji[r][i] * orr[r];
    b3[i] += s;
#pragma unroll
    for (int j = 0; j < 3; j++) {
      double h = 0;
#pragma unroll
      for (int r = 0; r < 3; r++) h += L.Ji[r][i] * w * L.Ji[r][j];
      H[3 * i + j] += h;
    }
  }
  if (pj >= 0) {
#pragma unroll
    for (int i = 0; i < 6; i++)
#pragma unroll
      for (int j = 0; j < 3; j++) {
        double h = 0;
#pragma unroll
        for (int r = 0; r < 3; r++) h += L.Jj[r][i] * w * L.Ji[r][j];
        W[i * 3 + j] = h;
      }
  } else {
#pragma unroll
    for (int i = 0; i < 18; i++) W[i] = 0;
  }
}

template <int CTRL>
__device__ __forceinline__ double ba_dpp_f64(double v) {  // v of the lane selected by the DPP control
  int lo = __double2loint(v), hi = __double2hiint(v);
  lo = __builtin_amdgcn_update_dpp(0, lo, CTRL, 0xF, 0xF, true);
  hi = __builtin_amdgcn_update_dpp(0, hi, CTRL, 0xF, 0xF, true);
  return __hiloint2double(hi, lo);
}

// The same with FOUR lanes per landmark (device-resident schedule): lane `sub` of a quad takes the landmark's edges sub,
// sub + 4, ...; Hll, bl and chi2 are summed over the quad with two DPP quad permutes ((a0 + a1) + (a2 + a3)).  A landmark has
// 2..10 observers, so one lane per landmark walked them serially with three dependent gathers each: the landmark role was
// the long pole of the linearisation launch.
template <int TH>
__device__ __forceinline__ void linearize_quad_body(const BADev &D, const State &S, const LinBuf &B, int robust, double *s_part) {
  const int g = blockIdx.x * TH + threadIdx.x, l = g >> 2, sub = g & 3;
  double chi = 0, hmax = 0;
  double H[9], b3[3] = {0, 0, 0};
#pragma unroll
  for (int i = 0; i < 9; i++) H[i] = 0;
  if (l < D.npt) {
    const double X[3] = {S.pt[3 * l], S.pt[3 * l + 1], S.pt[3 * l + 2]};
    const int cBeg = D.lm_start[l], cEnd = D.lm_start[l + 1];
    for (int c0 = cBeg + sub; c0 < cEnd; c0 += 8) {  // two edges of this lane per trip, their gathers issued together
      int ee[2], kfv[2], lvl[2], typ[2];
      double infov[2], measv[2][3];
      SE3 Tv[2];
#pragma unroll
      for (int u = 0; u < 2; u++) ee[u] = D.lm_edges[min(c0 + 4 * u, cEnd - 1)];
#pragma unroll
      for (int u = 0; u < 2; u++) {
        kfv[u] = D.e_kf[ee[u]]; lvl[u] = D.e_level[ee[u]]; typ[u] = D.e_type[ee[u]]; infov[u] = D.e_info[ee[u]];
        measv[u][0] = D.e_meas[3 * ee[u]]; measv[u][1] = D.e_meas[3 * ee[u] + 1]; measv[u][2] = D.e_meas[3 * ee[u] + 2];
      }
#pragma unroll
      for (int u = 0; u < 2; u++) Tv[u] = S.pose[kfv[u]];
#pragma unroll
      for (int u = 0; u < 2; u++) {
        if (c0 + 4 * u >= cEnd) break;
        lin_edge(D, B, X, ee[u], kfv[u], lvl[u], typ[u], infov[u], measv[u], Tv[u], robust, H, b3, chi);
      }
    }
  }
  // quad sums (every lane of the wave takes part: lanes beyond npt carry zeros)
#pragma unroll
  for (int i = 0; i < 9; i++) { H[i] += ba_dpp_f64<0xB1>(H[i]); H[i] += ba_dpp_f64<0x4E>(H[i]); }
#pragma unroll
  for (int i = 0; i < 3; i++) { b3[i] += ba_dpp_f64<0xB1>(b3[i]); b3[i] += ba_dpp_f64<0x4E>(b3[i]); }
  if (l < D.npt && sub == 0) {
#pragma unroll
    for (int i = 0; i < 9; i++) B.Hll[(size_t)9 * l + i] = H[i];
#pragma unroll
    for (int i = 0; i < 3; i++) B.bl[(size_t)3 * l + i] = b3[i];
    hmax = fmax(fmax(fabs(H[0]), fabs(H[4])), fabs(H[8]));
  }
  const double ws = wave_sum_d(chi);
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) hmax = fmax(hmax, __shfl_xor(hmax, o, 64));
  if ((threadIdx.x & 63) == 0) { s_part[threadIdx.x >> 6] = ws; s_part[TH / 64 + (threadIdx.x >> 6)] = hmax; }
  __syncthreads();
  if (threadIdx.x == 0) {
    double s = 0, m = 0;
    for (int i = 0; i < TH / 64; i++) { s += s_part[i]; m = fmax(m, s_part[TH / 64 + i]); }
    B.chiPart[blockIdx.x] = s;
    if (B.maxPart) B.maxPart[blockIdx.x] = m;
  }
}

// ------------------------------------------------------------------------------------------
// k_ba_linearize: one lane per landmark
// ------------------------------------------------------------------------------------------
template <int TH>
__device__ __forceinline__ void linearize_body(const BADev &D, const State &S, const LinBuf &B, int robust, double *s_part) {
  const int l = blockIdx.x * TH + threadIdx.x;
  double chi = 0, hmax = 0;
  if (l < D.npt) {
    const double X[3] = {S.pt[3 * l], S.pt[3 * l + 1], S.pt[3 * l + 2]};
    double H[9], b3[3] = {0, 0, 0};
#pragma unroll
    for (int i = 0; i < 9; i++) H[i] = 0;
    // The edges of a landmark are walked in chunks of 4 with the three dependent gathers (edge id -> its scalars and key
    // frame -> the key frame's pose) each issued for the whole chunk: three memory latencies per chunk, not per edge.
    const int cBeg = D.lm_start[l], cEnd = D.lm_start[l + 1];
    for (int c0 = cBeg; c0 < cEnd; c0 += 4) {
      int ee[4], kfv[4], lvl[4], typ[4];
      double infov[4], measv[4][3];
      SE3 Tv[4];
#pragma unroll
      for (int u = 0; u < 4; u++) ee[u] = D.lm_edges[min(c0 + u, cEnd - 1)];
#pragma unroll
      for (int u = 0; u < 4; u++) {
        kfv[u] = D.e_kf[ee[u]]; lvl[u] = D.e_level[ee[u]]; typ[u] = D.e_type[ee[u]]; infov[u] = D.e_info[ee[u]];
        measv[u][0] = D.e_meas[3 * ee[u]]; measv[u][1] = D.e_meas[3 * ee[u] + 1]; measv[u][2] = D.e_meas[3 * ee[u] + 2];
      }
#pragma unroll
      for (int u = 0; u < 4; u++) Tv[u] = S.pose[kfv[u]];
#pragma unroll
      for (int u = 0; u < 4; u++) {
      if (c0 + u >= cEnd) break;
      lin_edge(D, B, X, ee[u], kfv[u], lvl[u], typ[u], infov[u], measv[u], Tv[u], robust, H, b3, chi);
      }
    }
#pragma unroll
    for (int i = 0; i < 9; i++) B.Hll[(size_t)9 * l + i] = H[i];
#pragma unroll
    for (int i = 0; i < 3; i++) B.bl[(size_t)3 * l + i] = b3[i];
    hmax = fmax(fmax(fabs(H[0]), fabs(H[4])), fabs(H[8]));
  }
  const double ws = wave_sum_d(chi);
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) hmax = fmax(hmax, __shfl_xor(hmax, o, 64));
  if ((threadIdx.x & 63) == 0) { s_part[threadIdx.x >> 6] = ws; s_part[TH / 64 + (threadIdx.x >> 6)] = hmax; }
  __syncthreads();
  if (threadIdx.x == 0) {
    double s = 0, m = 0;
    for (int i = 0; i < TH / 64; i++) { s += s_part[i]; m = fmax(m, s_part[TH / 64 + i]); }
    B.chiPart[blockIdx.x] = s;
    if (B.maxPart) B.maxPart[blockIdx.x] = m;
  }
}

// ------------------------------------------------------------------------------------------
// k_ba_pose: one workgroup per free keyframe -> Hpp(k,k) and bp(k)
// ------------------------------------------------------------------------------------------
// part / nparts: this workgroup sums every nparts-th edge of key frame k starting at `part`; partOut != nullptr: the 27 sums
// (21 upper Hpp entries, 6 bp) go to partOut instead of Hpp / bp (k_ba_control adds the parts in order)
__device__ __forceinline__ void pose_body(const BADev &D, const State &S, const LinBuf &B, int robust, int P6, double (*s_part)[27], int k,
                                          int part = 0, int nparts = 1, double *partOut = nullptr) {
  const int tid = threadIdx.x;
  double acc[27];
#pragma unroll
  for (int i = 0; i < 27; i++) acc[i] = 0;
  for (int c = D.ps_start[k] + part * POSE_THREADS + tid; c < D.ps_start[k + 1]; c += nparts * POSE_THREADS) {
    const int e = D.ps_edges[c];
    if (D.e_level[e] != 0) continue;
    const int type = D.e_type[e], l = D.e_pt[e];
    const SE3 T = S.pose[D.e_kf[e]];
    const double X[3] = {S.pt[3 * l], S.pt[3 * l + 1], S.pt[3 * l + 2]};
    const double meas[3] = {D.e_meas[3 * e], D.e_meas[3 * e + 1], D.e_meas[3 * e + 2]};
    double p[3];
    EdgeLin L;
    edge_residual(D, type, T, X, meas, p, L.err);
    edge_jacobians(D, type, T, p, L);
    const double info = D.e_info[e];
    double chi2 = 0;
#pragma unroll
    for (int r = 0; r < 3; r++) chi2 += L.err[r] * (info * L.err[r]);
    double rho0 = chi2, rho1 = 1.;
    if (robust) fb::huber(chi2, D.delta, rho0, rho1);
    const double w = rho1 * info;
    int h = 0;
#pragma unroll
    for (int i = 0; i < 6; i++) {
      double s = 0;
#pragma unroll
      for (int r = 0; r < 3; r++) s += L.Jj[r][i] * (-(info * L.err[r]) * rho1);
      acc[21 + i] += s;
#pragma unroll
      for (int j = i; j < 6; j++) {
        double v = 0;
#pragma unroll
        for (int r = 0; r < 3; r++) v += L.Jj[r][i] * w * L.Jj[r][j];
        acc[h] += v;
        h++;
      }
    }
  }
  {
    double v[32];
#pragma unroll
    for (int i = 0; i < 32; i++) v[i] = i < 27 ? acc[i] : 0.0;
    const int lane = tid & 63;
    const double s = fb::wave_column_sums32(v, lane);  // lane L: column L >> 1
    if ((lane & 1) == 0 && (lane >> 1) < 27) s_part[tid >> 6][lane >> 1] = s;
  }
  __syncthreads();
  if (tid < 27) {
    double s = 0;
    for (int w2 = 0; w2 < POSE_THREADS / 64; w2++) s += s_part[w2][tid];
    s_part[0][tid] = s;
  }
  __syncthreads();
  if (partOut) {
    if (tid < 27) partOut[tid] = s_part[0][tid];
    return;
  }
  if (tid < 36) {
    const int i = tid / 6, j = tid % 6;
    const int a = i < j ? i : j, b2 = i < j ? j : i;
    const int idx = a * 6 - a * (a - 1) / 2 + (b2 - a);  // upper-triangular packing
    B.Hpp[(size_t)(6 * k + i) * P6 + 6 * k + j] = s_part[0][idx];
  }
  if (tid < 6) B.bp[6 * k + tid] = s_part[0][21 + tid];
}

// ------------------------------------------------------------------------------------------
// k_ba_odom: EdgeSE3Quat (OdomG2oTypeQuat.cc:180-204). Phase a: one lane per edge computes the
// residual and Jacobians into LDS; phase b: one lane per (free pose, row) adds the blocks into Hpp
// in edge order (deterministic).
// ------------------------------------------------------------------------------------------
struct OdomLin { double A[36], Bm[36], err[6]; };

template <bool GLOBAL>  // GLOBAL: the per-edge linearisations live in HBM scratch (they do not fit LDS)
__device__ __forceinline__ void odom_body(const BADev &D, const State &S, const LinBuf &B, int P6, int chiSlot, OdomLin *ol, double *s_chi,
                                          double *Hout, double *bout, bool zeroFirst, double *tmp = nullptr) {
  const int tid = threadIdx.x;
  double chi = 0;
  // tmp != nullptr (LDS, 108 doubles per edge: adj(T2) | adj(T1^-1) | J adj(T2)): the two 6x6x6 products of an edge are
  // spread over 36 lanes each instead of running serially on the edge's lane (same sums in the same order)
  for (int e = tid; e < D.nO; e += 256) {
    const SE3 T1 = S.pose[D.o_i[e]], T2 = S.pose[D.o_j[e]];
    const SE3 d = fb::se3_mul(fb::se3_mul(D.o_Zinv[e], T1), fb::se3_inverse(T2));
    double er[6];
    fb::se3_log(d, er);
    double J[36], a2[36], a1[36], t1[36];
#pragma unroll
    for (int i = 0; i < 36; i++) J[i] = 0;
    const double w0 = er[0], w1 = er[1], w2 = er[2], u0 = er[3], u1 = er[4], u2 = er[5];
    // JRInv: 0.5*[[skew(w),0],[skew(u),skew(w)]] + I
    J[1] = -w2; J[2] = w1; J[6] = w2; J[8] = -w0; J[12] = -w1; J[13] = w0;
    J[21 + 1] = -w2; J[21 + 2] = w1; J[27] = w2; J[27 + 2] = -w0; J[33] = -w1; J[33 + 1] = w0;
    J[18 + 1] = -u2; J[18 + 2] = u1; J[24] = u2; J[24 + 2] = -u0; J[30] = -u1; J[30 + 1] = u0;
#pragma unroll
    for (int i = 0; i < 36; i++) J[i] = 0.5 * J[i];
#pragma unroll
    for (int i = 0; i < 6; i++) J[i * 6 + i] += 1.0;
    fb::se3_adj(T2, a2);
    fb::se3_adj(fb::se3_inverse(T1), a1);
    if (tmp) {
      double *q = tmp + (size_t)e * 108;
      for (int i = 0; i < 36; i++) { q[i] = a2[i]; q[36 + i] = a1[i]; }
    } else {
      for (int i = 0; i < 6; i++)
        for (int j = 0; j < 6; j++) {
          double s = 0;
          for (int k = 0; k < 6; k++) s += J[i * 6 + k] * a2[k * 6 + j];
          t1[i * 6 + j] = s;
        }
      for (int i = 0; i < 6; i++)
        for (int j = 0; j < 6; j++) {
          double s = 0;
          for (int k = 0; k < 6; k++) s += t1[i * 6 + k] * a1[k * 6 + j];
          ol[e].A[i * 6 + j] = s;
        }
    }
    for (int i = 0; i < 36; i++) ol[e].Bm[i] = -J[i];
    const double info = D.o_info[e];
    const bool act = !(D.poseIdx[D.o_i[e]] < 0 && D.poseIdx[D.o_j[e]] < 0);  // allVerticesFixed edges are not active
    for (int i = 0; i < 6; i++) { ol[e].err[i] = er[i]; if (act) chi += er[i] * (info * er[i]); }
  }
  if (tmp) {
    __syncthreads();
    for (int idx = tid; idx < D.nO * 36; idx += 256) {  // J adj(T2), J = -Bm
      const int e = idx / 36, ij = idx - e * 36, i = ij / 6, j = ij - i * 6;
      const double *q = tmp + (size_t)e * 108;
      double s2 = 0;
      for (int k = 0; k < 6; k++) s2 += (-ol[e].Bm[i * 6 + k]) * q[k * 6 + j];
      tmp[(size_t)e * 108 + 72 + ij] = s2;
    }
    __syncthreads();
    for (int idx = tid; idx < D.nO * 36; idx += 256) {  // (J adj(T2)) adj(T1^-1)
      const int e = idx / 36, ij = idx - e * 36, i = ij / 6, j = ij - i * 6;
      const double *q = tmp + (size_t)e * 108;
      double s2 = 0;
      for (int k = 0; k < 6; k++) s2 += q[72 + i * 6 + k] * q[36 + k * 6 + j];
      ol[e].A[ij] = s2;
    }
  }
  s_chi[tid] = chi;
  if (GLOBAL) __threadfence_block();
  __syncthreads();
  if (tid == 0) {
    double s = 0;
    const int nl = D.nO < 256 ? D.nO : 256;  // lanes at or beyond nO hold 0: the same sum in the same order without them
    for (int i = 0; i < nl; i++) s += s_chi[i];
    B.chiPart[chiSlot] = s;
  }
  // phase b: lane (k, i) = row i of free pose k.  The row's diagonal block and right-hand side entry are accumulated in
  // registers over the incident edges and written once; an off-diagonal block is loaded, updated and stored as a group of six
  // (one memory round trip per block: the 72 dependent read-modify-writes per row this replaces were most of the role's time)
  if (zeroFirst) {  // Hout / bout are this role's own scratch: clear it with coalesced stores
    for (int i = tid; i < P6 * P6; i += 256) Hout[i] = 0;
    __syncthreads();
  }
  for (int row = tid; row < P6; row += 256) {
    const int k = row / 6, i = row % 6;
    double dg[6] = {0, 0, 0, 0, 0, 0}, bacc = 0;
    for (int cc = D.od_start[k]; cc < D.od_start[k + 1]; cc++) {  // incident edges only, in edge order
      const int e = D.od_edges[cc];
      const int pi = D.poseIdx[D.o_i[e]], pj = D.poseIdx[D.o_j[e]];
      const double info = D.o_info[e];
      const OdomLin &o = ol[e];
      // this row belongs to vertex i (Jacobian A) or vertex j (Jacobian Bm)
      const double *Mine = (pi == k) ? o.A : o.Bm;
      double bsum = 0;
      for (int r = 0; r < 6; r++) bsum += Mine[r * 6 + i] * (-(info * o.err[r]));
      bacc += bsum;
      for (int side = 0; side < 2; side++) {
        const int pc = side == 0 ? pi : pj;
        if (pc < 0) continue;
        const double *Oth = side == 0 ? o.A : o.Bm;
        double v[6];
        for (int j = 0; j < 6; j++) {
          double s = 0;
          for (int r = 0; r < 6; r++) s += Mine[r * 6 + i] * info * Oth[r * 6 + j];
          v[j] = s;
        }
        if (pc == k) {
          for (int j = 0; j < 6; j++) dg[j] += v[j];
        } else {
          double *dst = Hout + (size_t)row * P6 + 6 * pc, t6[6];
          for (int j = 0; j < 6; j++) t6[j] = dst[j];
          for (int j = 0; j < 6; j++) dst[j] = t6[j] + v[j];
        }
      }
      if (pi == k && pj == k) {  // degenerate self edge: also the Bm rows
        double b2 = 0;
        for (int r = 0; r < 6; r++) b2 += o.Bm[r * 6 + i] * (-(info * o.err[r]));
        bacc += b2;
      }
    }
    double *dd = Hout + (size_t)row * P6 + 6 * k;
    if (zeroFirst) {
      for (int j = 0; j < 6; j++) dd[j] = dg[j];
      bout[row] = bacc;
    } else {
      for (int j = 0; j < 6; j++) dd[j] += dg[j];
      bout[row] += bacc;
    }
  }
}

// ------------------------------------------------------------------------------------------
// k_ba_schur: MFMA fp64 accumulation of the Schur complement.
//   C[NT*16][NT*16] (upper tiles) += Yp[rows][K] * Wp[rows][K]^T over chunks of CHUNK landmarks,
//   Yp = W_e * Dinv_l scattered at rows 6*pose.., columns 3*li..; Wp likewise with W_e;
//   Wp row P6 carries bl so that column P6 of C is the reduced right-hand side (coefficients).
// v_mfma_f64_16x16x4_f64 operand layout (lane L): a = A[L%16][L/16], b = B[L/16][L%16],
// c[r] = C[(L/16)+4*r][L%16]  (measured on gfx950, scratch probe; not the f32 16x16x4 layout).
// ------------------------------------------------------------------------------------------
typedef double double4_t __attribute__((ext_vector_type(4)));

template <int MAXT>  // accumulator tiles per wave (compile-time so the C tiles stay in registers)
__device__ __forceinline__ void schur_body(const BADev &D, const LinBuf &B, double lambda, double *Dinv, double *Spart,
                                           int P6, int NT, int lmPerWg, uint8_t *smem) {
  const int rows = NT * 16;
  double *Yp = reinterpret_cast<double *>(smem);  // [rows][KPAD]
  double *Wp = Yp + (size_t)rows * KPAD;           // [rows][KPAD]
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int l0 = blockIdx.x * lmPerWg, l1 = min(l0 + lmPerWg, D.npt);
  // upper-triangular tiles, dealt round-robin to the 4 waves
  const int nTiles = NT * (NT + 1) / 2;
  double4_t acc[MAXT];
#pragma unroll
  for (int t = 0; t < MAXT; t++) acc[t] = (double4_t){0, 0, 0, 0};
  // the panels are sparse (a landmark touches only the rows of its observing key frames): zero them once, and after each
  // chunk's MFMA phase clear exactly the entries that chunk wrote instead of re-zeroing 2 x rows x KPAD doubles
  for (int i = tid; i < 2 * rows * KPAD; i += SCHUR_THREADS) Yp[i] = 0;
  __syncthreads();
  for (int c0 = l0; c0 < l1; c0 += CHUNK) {
    // 16 lanes per landmark of the chunk: every lane inverts D (cheap), lane 0 of the group publishes it, the
    // group's lanes scatter the landmark's edge blocks in parallel
    {
      const int li = tid >> 4, sub = tid & 15;
      const int l = c0 + li;
      if (l < l1) {
        double M[9];
#pragma unroll
        for (int i = 0; i < 9; i++) M[i] = B.Hll[(size_t)9 * l + i];
        M[0] += lambda; M[4] += lambda; M[8] += lambda;
        const double c00 = M[4] * M[8] - M[5] * M[7], c01 = M[5] * M[6] - M[3] * M[8], c02 = M[3] * M[7] - M[4] * M[6];
        const double det = M[0] * c00 + M[1] * c01 + M[2] * c02;
        const double id = 1.0 / det;
        double Di[9];
        Di[0] = c00 * id; Di[1] = (M[2] * M[7] - M[1] * M[8]) * id; Di[2] = (M[1] * M[5] - M[2] * M[4]) * id;
        Di[3] = c01 * id; Di[4] = (M[0] * M[8] - M[2] * M[6]) * id; Di[5] = (M[2] * M[3] - M[0] * M[5]) * id;
        Di[6] = c02 * id; Di[7] = (M[1] * M[6] - M[0] * M[7]) * id; Di[8] = (M[0] * M[4] - M[1] * M[3]) * id;
        if (sub == 0) {
#pragma unroll
          for (int i = 0; i < 9; i++) Dinv[(size_t)9 * l + i] = Di[i];
#pragma unroll
          for (int c = 0; c < 3; c++) Wp[(size_t)P6 * KPAD + 3 * li + c] = B.bl[(size_t)3 * l + c];
        }
        for (int cc = D.lm_start[l] + sub; cc < D.lm_start[l + 1]; cc += 16) {
          const int e = D.lm_edges[cc];
          const int pj = D.e_pj[e];
          if (pj < 0 || D.e_level[e] != 0) continue;
          const double *W = B.W + (size_t)e * 18;
#pragma unroll
          for (int i = 0; i < 6; i++) {
            const double w0 = W[i * 3], w1 = W[i * 3 + 1], w2 = W[i * 3 + 2];
            double *yr = Yp + (size_t)(6 * pj + i) * KPAD + 3 * li;
            double *wr = Wp + (size_t)(6 * pj + i) * KPAD + 3 * li;
            wr[0] = w0; wr[1] = w1; wr[2] = w2;
            yr[0] = w0 * Di[0] + w1 * Di[3] + w2 * Di[6];
            yr[1] = w0 * Di[1] + w1 * Di[4] + w2 * Di[7];
            yr[2] = w0 * Di[2] + w1 * Di[5] + w2 * Di[8];
          }
        }
      }
    }
    __syncthreads();
#pragma unroll
    for (int t = 0; t < MAXT; t++) {
      const int tile = wv + 4 * t;
      if (tile < nTiles) {
        // tile -> (ti, tj), ti <= tj
        int ti = 0, rem = tile;
        while (rem >= NT - ti) { rem -= NT - ti; ti++; }
        const int tj = ti + rem;
        const double *ya = Yp + (size_t)(ti * 16 + (lane & 15)) * KPAD + (lane >> 4);
        const double *wb = Wp + (size_t)(tj * 16 + (lane & 15)) * KPAD + (lane >> 4);
        double4_t c = acc[t];
#pragma unroll
        for (int k0 = 0; k0 < KP; k0 += 4) c = __builtin_amdgcn_mfma_f64_16x16x4f64(ya[k0], wb[k0], c, 0, 0, 0);
        acc[t] = c;
      }
    }
    __syncthreads();
    {  // clear what this chunk scattered (same traversal as above)
      const int li = tid >> 4, sub = tid & 15;
      const int l = c0 + li;
      if (l < l1) {
        if (sub == 0) {
#pragma unroll
          for (int c = 0; c < 3; c++) Wp[(size_t)P6 * KPAD + 3 * li + c] = 0;
        }
        for (int cc = D.lm_start[l] + sub; cc < D.lm_start[l + 1]; cc += 16) {
          const int e = D.lm_edges[cc];
          const int pj = D.e_pj[e];
          if (pj < 0 || D.e_level[e] != 0) continue;
#pragma unroll
          for (int i = 0; i < 6; i++) {
            double *yr = Yp + (size_t)(6 * pj + i) * KPAD + 3 * li;
            double *wr = Wp + (size_t)(6 * pj + i) * KPAD + 3 * li;
            yr[0] = 0; yr[1] = 0; yr[2] = 0;
            wr[0] = 0; wr[1] = 0; wr[2] = 0;
          }
        }
      }
    }
    __syncthreads();
  }
  // write this workgroup's partial (upper tiles) : Spart[wg][rows][rows]
  double *out = Spart + (size_t)blockIdx.x * rows * rows;
#pragma unroll
  for (int t = 0; t < MAXT; t++) {
    const int tile = wv + 4 * t;
    if (tile < nTiles) {
      int ti = 0, rem = tile;
      while (rem >= NT - ti) { rem -= NT - ti; ti++; }
      const int tj = ti + rem;
#pragma unroll
      for (int r = 0; r < 4; r++) out[(size_t)(ti * 16 + (lane >> 4) + 4 * r) * rows + tj * 16 + (lane & 15)] = acc[t][r];
    }
  }
}

// ------------------------------------------------------------------------------------------
// k_ba_solve: reduce partials, S = Hpp + lambda I - sum, LDL^T (no pivoting) and solves.
// ------------------------------------------------------------------------------------------
// HppO != nullptr: the dense Hpp is not materialised: B.Hpp holds the key frames' diagonal blocks only and HppO the
// odometry (pose-pose) blocks (device-resident schedule)
__device__ __forceinline__ void solve_body(const LinBuf &B, double lambda, const double *Spart, int nWg, int P6,
                                           int NT, double *xp, double *okFlag, uint8_t *smem, int &s_ok, const double *HppO = nullptr) {
  double *A = reinterpret_cast<double *>(smem);  // [P6][P6+1] lower triangle used
  const int ld = P6 + 1;
  double *rhs = A + (size_t)P6 * ld;             // [P6]
  const int tid = threadIdx.x, rows = NT * 16;
  if (tid == 0) s_ok = 1;
  for (int idx = tid; idx < P6 * P6; idx += SOLVE_THREADS) {
    const int i = idx / P6, j = idx % P6;
    if (j > i) continue;  // lower triangle: A[i][j] = S[j][i] (upper, as g2o's solver reads it)
    double s = 0;
    for (int w = 0; w < nWg; w++) s += Spart[(size_t)w * rows * rows + (size_t)j * rows + i];
    double h;
    if (HppO) h = (i / 6 == j / 6 ? B.Hpp[(size_t)j * P6 + i] : 0.0) + HppO[(size_t)j * P6 + i];
    else h = B.Hpp[(size_t)j * P6 + i];
    if (i == j) h += lambda;
    A[(size_t)i * ld + j] = h - s;
  }
  for (int i = tid; i < P6; i += SOLVE_THREADS) {
    double s = 0;
    for (int w = 0; w < nWg; w++) s += Spart[(size_t)w * rows * rows + (size_t)i * rows + P6];
    rhs[i] = B.bp[i] - s;
  }
  __syncthreads();
  // Blocked right-looking LDL^T with 6x6 blocks (one keyframe per block): per block step every thread factors the
  // diagonal block redundantly in registers (no broadcast barrier), one thread per row solves the panel, and the
  // trailing sub-matrix gets a rank-6 update -- two barriers per keyframe instead of one per column.
  // On exit A holds the unit-lower L below the diagonal and D on it.
  double *Up = rhs + P6;  // [P6][6] panel u = l * d of the current block step
  const int nb6 = P6 / 6;
  for (int J = 0; J < nb6; J++) {
    const int c0 = 6 * J, c1 = c0 + 6;
    double Lb[6][6], d[6], dinv[6];
#pragma unroll
    for (int r = 0; r < 6; r++)
#pragma unroll
      for (int c = 0; c <= r; c++) Lb[r][c] = A[(size_t)(c0 + r) * ld + c0 + c];
    bool neg = false;
#pragma unroll
    for (int c = 0; c < 6; c++) {
      double dc = Lb[c][c];
#pragma unroll
      for (int m = 0; m < c; m++) dc -= Lb[c][m] * Lb[c][m] * d[m];
      if (dc == 0) neg = true;  // SimplicialLDLT: the factorisation fails on a pivot that is exactly 0, negative pivots proceed
      d[c] = dc;
      const double inv = dc != 0 ? ba_rcp(dc) : 0.0;
      dinv[c] = inv;
#pragma unroll
      for (int r = c + 1; r < 6; r++) {
        double v = Lb[r][c];
#pragma unroll
        for (int m = 0; m < c; m++) v -= Lb[r][m] * Lb[c][m] * d[m];
        Lb[r][c] = v * inv;
      }
    }
    if (neg && tid == 0) s_ok = 0;
    __syncthreads();  // everyone has read the diagonal block before it is overwritten
    if (tid == 0) {
#pragma unroll
      for (int r = 0; r < 6; r++) {
        A[(size_t)(c0 + r) * ld + c0 + r] = d[r];
#pragma unroll
        for (int c = 0; c < r; c++) A[(size_t)(c0 + r) * ld + c0 + c] = Lb[r][c];
      }
    }
    for (int i = c1 + tid; i < P6; i += SOLVE_THREADS) {  // panel: row i of L below the block
      double u[6];
#pragma unroll
      for (int c = 0; c < 6; c++) {
        double v = A[(size_t)i * ld + c0 + c];
#pragma unroll
        for (int m = 0; m < c; m++) v -= u[m] * Lb[c][m];
        u[c] = v;
      }
#pragma unroll
      for (int c = 0; c < 6; c++) {
        Up[(size_t)i * 6 + c] = u[c];
        A[(size_t)i * ld + c0 + c] = u[c] * dinv[c];  // dinv = 0 for a zero pivot, as the division rule above
      }
    }
    __syncthreads();
    const int tr = tid >> 4, tc = tid & 15;
    for (int i = c1 + tr; i < P6; i += 16) {
      double u[6];
#pragma unroll
      for (int c = 0; c < 6; c++) u[c] = Up[(size_t)i * 6 + c];
      for (int k = c1 + tc; k <= i; k += 16) {
        double acc2 = 0;
#pragma unroll
        for (int c = 0; c < 6; c++) acc2 += u[c] * A[(size_t)k * ld + c0 + c];
        A[(size_t)i * ld + k] -= acc2;
      }
    }
    __syncthreads();
  }
  // Triangular solves, blocked like the factorisation: per 6-row block every thread solves the 6x6 triangle
  // redundantly from LDS (no broadcast barrier), then the rows outside the block take their rank-6 update in parallel
  // -- one barrier per key frame and direction instead of 2 x P6 dependent shuffle steps in a single wave.
  for (int J = 0; J < nb6; J++) {  // forward: L y = b
    const int c0 = 6 * J, c1 = c0 + 6;
    double yJ[6];
#pragma unroll
    for (int r = 0; r < 6; r++) {
      double v = rhs[c0 + r];
#pragma unroll
      for (int m = 0; m < r; m++) v -= A[(size_t)(c0 + r) * ld + c0 + m] * yJ[m];
      yJ[r] = v;
    }
    __syncthreads();  // everyone has read rhs[c0..c1) before it is overwritten / the rows below are updated
    if (tid < 6) rhs[c0 + tid] = yJ[tid];
    for (int i = c1 + tid; i < P6; i += SOLVE_THREADS) {
      double v = rhs[i];
#pragma unroll
      for (int c = 0; c < 6; c++) v -= A[(size_t)i * ld + c0 + c] * yJ[c];
      rhs[i] = v;
    }
    __syncthreads();
  }
  for (int i = tid; i < P6; i += SOLVE_THREADS) {  // D^-1
    const double dd = A[(size_t)i * ld + i];
    rhs[i] = dd != 0 ? rhs[i] * ba_rcp(dd) : 0.0;
  }
  __syncthreads();
  for (int J = nb6 - 1; J >= 0; J--) {  // backward: L^T x = y
    const int c0 = 6 * J;
    double xJ[6];
#pragma unroll
    for (int r = 5; r >= 0; r--) {
      double v = rhs[c0 + r];
#pragma unroll
      for (int m = 5; m > r; m--) v -= A[(size_t)(c0 + m) * ld + c0 + r] * xJ[m];
      xJ[r] = v;
    }
    __syncthreads();
    if (tid < 6) rhs[c0 + tid] = xJ[tid];
    for (int i = tid; i < c0; i += SOLVE_THREADS) {
      double v = rhs[i];
#pragma unroll
      for (int c = 0; c < 6; c++) v -= A[(size_t)(c0 + c) * ld + i] * xJ[c];
      rhs[i] = v;
    }
    __syncthreads();
  }
  for (int i = tid; i < P6; i += SOLVE_THREADS) xp[i] = rhs[i];
  __syncthreads();
  if (tid == 0) *okFlag = s_ok ? 1.0 : 0.0;
}

// ------------------------------------------------------------------------------------------
// solve_lookahead: the same LDL^T (6x6 blocks, no pivoting, fails on a zero pivot) arranged for latency -- this one
// workgroup sits on the critical path of every LM trial.
//  * the right-hand side rides along as row P6 of the matrix: its panel steps ARE the forward substitution, so after the
//    factorisation row P6 holds z = D^-1 L^-1 b and only the backward substitution remains;
//  * look-ahead: in the trailing update of step J wave 0 updates only the NEXT diagonal block, factors it (a chain of ~100
//    dependent fp64 operations) and publishes L, d, 1/d through LDS, while the other three waves update the rest of the
//    trailing matrix; every thread used to factor the block redundantly BETWEEN the barriers;
//  * two barriers per key frame instead of three.
// HppO as in solve_body.
#ifdef FB_BA_STAMPS
__device__ unsigned long long g_ba_stamps[16];
#define BA_T0() unsigned long long bt_ = 0; if (threadIdx.x == 0) { __builtin_amdgcn_sched_barrier(0); bt_ = __builtin_amdgcn_s_memtime(); __builtin_amdgcn_sched_barrier(0); }
#define BA_TICK(slot_) if (threadIdx.x == 0) { __builtin_amdgcn_sched_barrier(0); const unsigned long long t_ = __builtin_amdgcn_s_memtime(); __builtin_amdgcn_sched_barrier(0); g_ba_stamps[slot_] += t_ - bt_; bt_ = t_; }
#else
#define BA_T0()
#define BA_TICK(slot_)
#endif
template <int TS>  // threads of the workgroup
__device__ __forceinline__ void solve_lookahead(const LinBuf &B, double lambda, const double *Spart, int P6, int NT, double *xp, double *okFlag,
                                                uint8_t *smem, int &s_ok, const double *HppO) {
  const int ld = P6 + 1, rows = NT * 16, tid = threadIdx.x;
  BA_T0()
  double *A = reinterpret_cast<double *>(smem);   // [P6 + 1][ld], lower triangle; row P6 = right-hand side
  double *Up = A + (size_t)(P6 + 1) * ld;         // [P6 + 1][6] panel u = l * d of the current step
  double *fact2 = Up + (size_t)(P6 + 1) * 6;      // two buffers (even / odd block) of 36 L (row-major, lower) | 6 d | 6 1/d
  double *rhs = fact2 + 96;                       // [P6] backward substitution
  if (tid == 0) s_ok = 1;
  {
    // A[i][j] = S[j][i] for i >= j (the upper triangle, as g2o's solver reads it): 16 lanes walk a row j of the upper
    // triangle (contiguous in memory), 16 row groups, four rows in flight per thread.  (The phase stamps had a quarter of the
    // solve in this loop when it divided a flat index and waited for each element's three loads in turn.)
    const int tr = tid >> 4, tc = tid & 15;
    constexpr int MAXM = 9;  // ceil(138 / 16): the LDS-resident system has at most 23 free key frames
    for (int j = tr; j < P6; j += TS / 16) {
      const int jb = j / 6;
      const double *srow = Spart + (size_t)j * rows, *orow = HppO + (size_t)j * P6, *hrow = B.Hpp + (size_t)j * P6;
      // every load of the row segment is issued before the first use: unconditional loads at clamped indices (a predicated
      // load makes the compiler wait for each element in turn), the selection happens on the values
      double o[MAXM], sv[MAXM], hv[MAXM];
#pragma unroll
      for (int m = 0; m < MAXM; m++) o[m] = orow[min(j + tc + 16 * m, P6 - 1)];
#pragma unroll
      for (int m = 0; m < MAXM; m++) sv[m] = srow[min(j + tc + 16 * m, P6 - 1)];
#pragma unroll
      for (int m = 0; m < MAXM; m++) hv[m] = hrow[min(j + tc + 16 * m, P6 - 1)];
#pragma unroll
      for (int m = 0; m < MAXM; m++) {
        const int i = j + tc + 16 * m;
        double h = o[m] - sv[m];
        if (i < 6 * jb + 6) h += hv[m];  // same key frame: the diagonal block
        if (i == j) h += lambda;
        if (i < P6) A[(size_t)i * ld + j] = h;
      }
    }
  }
  for (int i = tid; i < P6; i += TS) A[(size_t)P6 * ld + i] = B.bp[i] - Spart[(size_t)i * rows + P6];
  __syncthreads();
  const int nb6 = P6 / 6;
  // factor the 6x6 diagonal block at c0 (wave-redundant in registers), lane `wlane` 0 publishes it
  auto factor_block = [&](int c0) {
#pragma clang fp contract(fast)
    double *fact = fact2 + 48 * ((c0 / 6) & 1);  // wave 0 publishes block J + 1 while slower waves may still read block J's
    // the dependent chain of this routine is the critical path of every block step: fused multiply-adds and the products
    // U[c][m] = L[c][m] d[m] kept next to L halve its length (the BA is tolerance-held)
    double Lb[6][6], Ub[6][6], d[6], dinv[6];
#pragma unroll
    for (int r = 0; r < 6; r++)
#pragma unroll
      for (int c = 0; c <= r; c++) Lb[r][c] = A[(size_t)(c0 + r) * ld + c0 + c];
    bool neg = false;
#pragma unroll
    for (int c = 0; c < 6; c++) {
      double dc = Lb[c][c];
#pragma unroll
      for (int m = 0; m < c; m++) dc -= Lb[c][m] * Ub[c][m];
      if (dc == 0) neg = true;  // SimplicialLDLT: the factorisation fails on a pivot that is exactly 0, negative pivots proceed
      d[c] = dc;
      const double inv = dc != 0 ? ba_rcp(dc) : 0.0;
      dinv[c] = inv;
#pragma unroll
      for (int r = c + 1; r < 6; r++) {
        double v = Lb[r][c];
#pragma unroll
        for (int m = 0; m < c; m++) v -= Lb[r][m] * Ub[c][m];
        Ub[r][c] = v;          // u = l d
        Lb[r][c] = v * inv;
      }
    }
    if ((tid & 63) == 0) {
      if (neg) s_ok = 0;
#pragma unroll
      for (int r = 0; r < 6; r++) {
        A[(size_t)(c0 + r) * ld + c0 + r] = d[r];
        fact[36 + r] = d[r];
        fact[42 + r] = dinv[r];
#pragma unroll
        for (int c = 0; c < r; c++) { A[(size_t)(c0 + r) * ld + c0 + c] = Lb[r][c]; fact[r * 6 + c] = Lb[r][c]; }
      }
    }
  };
  // (Measured and dropped: wave 0 running ahead -- taking the next block's panel rows itself so that its factor chain
  // overlaps the panel of the others, with a software barrier among the panel waves.  Same time: the chain
  // [factor read -> panel rows -> block update -> 6x6 factor] of the next block is the critical path either way.)
  if (tid < 64 && nb6 > 0) factor_block(0);
  __syncthreads();
  BA_TICK(1)  // first factor
  for (int J = 0; J < nb6; J++) {
    const int c0 = 6 * J, c1 = c0 + 6;
    const double *fact = fact2 + 48 * (J & 1);
    double Lb[6][6], dinv[6];
#pragma unroll
    for (int r = 1; r < 6; r++)
#pragma unroll
      for (int c = 0; c < r; c++) Lb[r][c] = fact[r * 6 + c];
#pragma unroll
    for (int c = 0; c < 6; c++) dinv[c] = fact[42 + c];
    for (int i = c1 + tid; i <= P6; i += TS) {  // panel: row i of L below the block (row P6: forward substitution)
      double u[6];
#pragma unroll
      for (int c = 0; c < 6; c++) {
        double v = A[(size_t)i * ld + c0 + c];
#pragma unroll
        for (int m = 0; m < c; m++) v -= u[m] * Lb[c][m];
        u[c] = v;
      }
#pragma unroll
      for (int c = 0; c < 6; c++) {
        Up[(size_t)i * 6 + c] = u[c];
        A[(size_t)i * ld + c0 + c] = u[c] * dinv[c];  // dinv = 0 for a zero pivot
      }
    }
    BA_TICK(2)  // read fact + panel
    __syncthreads();
    BA_TICK(3)  // barrier after panel
    if (tid < 64) {  // wave 0: the next diagonal block first, then its factorisation
      if (c1 < P6) {
        if (tid < 21) {
          int r = 0, rem = tid;
          while (rem > r) { rem -= r + 1; r++; }  // tid -> (r, cc), cc <= r
          const int i = c1 + r, k = c1 + rem;
          double acc2 = 0;
#pragma unroll
          for (int c = 0; c < 6; c++) acc2 += Up[(size_t)i * 6 + c] * A[(size_t)k * ld + c0 + c];
          A[(size_t)i * ld + k] -= acc2;
        }
        factor_block(c1);
      }
      BA_TICK(4)  // wave 0: next diagonal block update + factorisation
    } else {  // the other waves: the rest of the trailing matrix, rows c1 + 6 .. P6 (row P6: columns < P6)
      const int q = tid - 64, tr = q >> 4, tc = q & 15;
      for (int i = c1 + 6 + tr; i <= P6; i += (TS - 64) / 16) {
        double u[6];
#pragma unroll
        for (int c = 0; c < 6; c++) u[c] = Up[(size_t)i * 6 + c];
        const int kEnd = i < P6 ? i : P6 - 1;
        for (int k = c1 + tc; k <= kEnd; k += 16) {
          double acc2 = 0;
#pragma unroll
          for (int c = 0; c < 6; c++) acc2 += u[c] * A[(size_t)k * ld + c0 + c];
          A[(size_t)i * ld + k] -= acc2;
        }
      }
    }
    __syncthreads();
    BA_TICK(5)  // wait for the trailing update of the other waves
  }
  // row P6 now holds z = D^-1 L^-1 b.  Backward substitution L^T x = z on ONE wave: 18 dependent block steps with two
  // workgroup barriers each cost more in barriers than in arithmetic; inside a wave the LDS operations are ordered, so the
  // steps need no barrier at all.
  if (tid < 64) {
    for (int i = tid; i < P6; i += 64) rhs[i] = A[(size_t)P6 * ld + i];
    for (int J = nb6 - 1; J >= 0; J--) {
      const int c0 = 6 * J;
      double xJ[6];
#pragma unroll
      for (int r = 5; r >= 0; r--) {
        double v = rhs[c0 + r];
#pragma unroll
        for (int m = 5; m > r; m--) v -= A[(size_t)(c0 + m) * ld + c0 + r] * xJ[m];
        xJ[r] = v;
      }
      __builtin_amdgcn_wave_barrier();  // every lane has read rhs[c0..c0+6) before it is overwritten below
      if (tid < 6) rhs[c0 + tid] = xJ[tid];
      for (int i = tid; i < c0; i += 64) {
        double v = rhs[i];
#pragma unroll
        for (int c = 0; c < 6; c++) v -= A[(size_t)(c0 + c) * ld + i] * xJ[c];
        rhs[i] = v;
      }
      __builtin_amdgcn_wave_barrier();
    }
    for (int i = tid; i < P6; i += 64) xp[i] = rhs[i];
  }
  __syncthreads();
  BA_TICK(6)  // backward substitution
#ifdef FB_BA_STAMPS
  if (tid == 0) g_ba_stamps[15] += 1;
#endif
  if (tid == 0) *okFlag = s_ok ? 1.0 : 0.0;
}

// ------------------------------------------------------------------------------------------
// k_ba_update: back-substitution, trial state, scale term sum x (lambda x + b)
// ------------------------------------------------------------------------------------------
// Four lanes per landmark (as in the linearisation): lane q takes the landmark's edges q, q + 4, ..., the partial sums meet
// in a two-step butterfly (the same value on all four lanes), lane q < 3 then finishes coordinate q.  One lane per landmark
// left the back-substitution of 10 k landmarks to 79 two-wave workgroups.  Threads [4 npt, 4 npt + n_kf) = key frames.
__device__ __forceinline__ void update_body(const BADev &D, const LinBuf &B, const State &cur, const State &trial, const double *Dinv,
                                            const double *xp, double lambda, double *scalePart, int countPoses, double *s_part) {
  const int g = blockIdx.x * LIN_THREADS + threadIdx.x;
  double sc = 0;
  if (g < 4 * D.npt) {
    const int l = g >> 2, q = g & 3;
    double cl[3] = {0, 0, 0};
    for (int cc = D.lm_start[l] + q; cc < D.lm_start[l + 1]; cc += 4) {
      const int e = D.lm_edges[cc];
      const int pj = D.e_pj[e];
      if (pj < 0 || D.e_level[e] != 0) continue;
      const double *W = B.W + (size_t)e * 18;
#pragma unroll
      for (int j = 0; j < 3; j++)
#pragma unroll
        for (int i = 0; i < 6; i++) cl[j] -= W[i * 3 + j] * xp[6 * pj + i];
    }
#pragma unroll
    for (int j = 0; j < 3; j++) {
      cl[j] += __shfl_xor(cl[j], 1, 64);
      cl[j] += __shfl_xor(cl[j], 2, 64);
      cl[j] += B.bl[(size_t)3 * l + j];
    }
    if (q < 3) {
      const double *Di = Dinv + (size_t)9 * l;
      const double xl = Di[q * 3] * cl[0] + Di[q * 3 + 1] * cl[1] + Di[q * 3 + 2] * cl[2];
      trial.pt[3 * l + q] = cur.pt[3 * l + q] + xl;
      sc = xl * (lambda * xl + B.bl[(size_t)3 * l + q]);
    }
  } else if (g < 4 * D.npt + D.n_kf) {
    const int k = g - 4 * D.npt;
    const int pi = D.poseIdx[k];
    if (pi < 0) trial.pose[k] = cur.pose[k];
    else {
      double u[6];
#pragma unroll
      for (int i = 0; i < 6; i++) { u[i] = xp[6 * pi + i]; if (countPoses) sc += u[i] * (lambda * u[i] + B.bp[6 * pi + i]); }
      trial.pose[k] = fb::se3_mul(fb::se3_exp(u), cur.pose[k]);
    }
  }
  const double ws = wave_sum_d(sc);
  if ((threadIdx.x & 63) == 0) s_part[threadIdx.x >> 6] = ws;
  __syncthreads();
  if (threadIdx.x == 0) {
    double s = 0;
    for (int i = 0; i < LIN_THREADS / 64; i++) s += s_part[i];
    scalePart[blockIdx.x] = s;
  }
}

// Spart[0] = sum over the workgroup partials (sharded BA: the local sum that goes through the all-reduce)
// rows > 0: the buffers are [rows][rows] matrices of which only the upper 16x16 tiles (and with them column P6, the reduced
// right-hand side) are ever written and read: the lower tiles are skipped
// 256-thread workgroups: 64 consecutive elements x 4 quarters of the partials (a wave = one quarter, so its loads are
// contiguous); the quarters meet in LDS in a fixed order.  One thread per element over all nWg partials kept only 64
// workgroups busy on 19 MB.
constexpr int SUMPARTS_ELEMS = 64;
__device__ __forceinline__ void sumparts_body(double *Spart, int nWg, int n, int rows = 0) {
  __shared__ double s_q[3][SUMPARTS_ELEMS];
  const int lane = threadIdx.x & 63, q = threadIdx.x >> 6;
  const int i = blockIdx.x * SUMPARTS_ELEMS + lane;
  bool live = i < n;
  if (live && rows > 0) {
    const int r = i / rows, c = i - r * rows;
    live = (r >> 4) <= (c >> 4);
  }
  const int per = (nWg + 3) >> 2, w0 = q * per, w1 = min(nWg, w0 + per);
  // 4 independent chains per thread: a single running sum serialises the memory latencies
  double a[4] = {0, 0, 0, 0};
  if (live) {
    int w = w0;
    for (; w + 4 <= w1; w += 4) {
#pragma unroll
      for (int u = 0; u < 4; u++) a[u] += Spart[(size_t)(w + u) * n + i];
    }
    for (; w < w1; w++) a[0] += Spart[(size_t)w * n + i];
  }
  const double s = (a[0] + a[1]) + (a[2] + a[3]);
  if (q > 0) s_q[q - 1][lane] = s;
  __syncthreads();
  if (q == 0 && live) Spart[i] = ((s + s_q[0][lane]) + s_q[1][lane]) + s_q[2][lane];
}
__global__ void k_ba_sumparts(double *Spart, int nWg, int n) { sumparts_body(Spart, nWg, n); }

// sum of partials in index order + max |diag| (computeLambdaInit)
__global__ void k_ba_scalars(const double *part, int n, const double *Hpp, int P6, const double *Hll, int npt, double *out,
                             int wantDiag) {
  __shared__ double s_m[256];
  const int tid = threadIdx.x;
  if (tid == 0) {
    double s = 0;
    for (int i = 0; i < n; i++) s += part[i];
    out[0] = s;
  }
  if (wantDiag) {
    double m = 0;
    for (int i = tid; i < P6; i += 256) m = fmax(m, fabs(Hpp[(size_t)i * P6 + i]));
    for (int i = tid; i < npt * 3; i += 256) m = fmax(m, fabs(Hll[(size_t)9 * (i / 3) + 4 * (i % 3)]));
    s_m[tid] = m;
    __syncthreads();
    if (tid == 0) {
      double mm = 0;
      for (int i = 0; i < 256; i++) mm = fmax(mm, s_m[i]);
      out[1] = mm;
    }
  }
}

// gating between the two optimisation rounds and the final outlier flags
// (Optimizer.cc:1059-1073,1102-1115 / 2534-2565,2573-2607)
__device__ __forceinline__ void gate_edge(const BADev &D, const State &S, int setLevel, uint8_t *outFlag, int e) {
  if (D.e_level[e] == 2) {  // edge owned by another rank of a sharded BA
    if (!setLevel) outFlag[e] = 0;
    return;
  }
  const int type = D.e_type[e], l = D.e_pt[e];
  const SE3 T = S.pose[D.e_kf[e]];
  const double X[3] = {S.pt[3 * l], S.pt[3 * l + 1], S.pt[3 * l + 2]};
  bool bad;
  if (type == T_PROJ) {
    double p[3];
    fb::se3_map(T, X, p);
    bad = D.e_chi2[e] > 5.991 || !(p[2] > 0.0);
  } else {
    if (setLevel) {  // e->computeError() first
      const double meas[3] = {D.e_meas[3 * e], D.e_meas[3 * e + 1], D.e_meas[3 * e + 2]};
      double p[3], err[3];
      edge_residual(D, type, T, X, meas, p, err);
      const double info = D.e_info[e];
      D.e_chi2[e] = err[0] * (info * err[0]) + err[1] * (info * err[1]) + err[2] * (info * err[2]);
    }
    bad = D.e_chi2[e] > 5.991;
  }
  if (setLevel) { if (bad) D.e_level[e] = 1; }
  else outFlag[e] = bad ? 1 : 0;
}

// ---- __global__ entry points of the bodies above: explicit arguments (host-driven Levenberg-Marquardt: sharded BA, the
//      HBM-resident reduced system of ba_big.inc, FB_BA_TRACE) ---------------------------------------------------------
__global__ __launch_bounds__(LIN_THREADS) void k_ba_linearize(BADev D, State S, LinBuf B, int robust) {
  __shared__ double s_part[2 * LIN_THREADS / 64];
  linearize_body<LIN_THREADS>(D, S, B, robust, s_part);
}
__global__ __launch_bounds__(POSE_THREADS) void k_ba_pose(BADev D, State S, LinBuf B, int robust, int P6) {
  __shared__ double s_part[POSE_THREADS / 64][27];
  pose_body(D, S, B, robust, P6, s_part, blockIdx.x);
}
template <bool GLOBAL>
__global__ __launch_bounds__(256) void k_ba_odom(BADev D, State S, LinBuf B, int P6, int chiSlot, OdomLin *olGlobal) {
  extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
  __shared__ double s_chi[256];
  odom_body<GLOBAL>(D, S, B, P6, chiSlot, GLOBAL ? olGlobal : reinterpret_cast<OdomLin *>(smem), s_chi, B.Hpp, B.bp, false);
}
template <int MAXT>
__global__ __launch_bounds__(SCHUR_THREADS) void k_ba_schur(BADev D, LinBuf B, double lambda, double *Dinv, double *Spart, int P6, int NT, int lmPerWg) {
  extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
  schur_body<MAXT>(D, B, lambda, Dinv, Spart, P6, NT, lmPerWg, smem);
}
__global__ __launch_bounds__(SOLVE_THREADS) void k_ba_solve(LinBuf B, double lambda, const double *Spart, int nWg, int P6, int NT, double *xp, double *okFlag) {
  extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
  __shared__ int s_ok;
  solve_body(B, lambda, Spart, nWg, P6, NT, xp, okFlag, smem, s_ok);
}
__global__ __launch_bounds__(LIN_THREADS) void k_ba_update(BADev D, LinBuf B, State cur, State trial, const double *Dinv, const double *xp, double lambda,
                                                           double *scalePart, int countPoses) {
  __shared__ double s_part[LIN_THREADS / 64];
  update_body(D, B, cur, trial, Dinv, xp, lambda, scalePart, countPoses, s_part);
}
__global__ void k_ba_gate(BADev D, State S, int setLevel, uint8_t *outFlag) {
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e < D.nE) gate_edge(D, S, setLevel, outFlag, e);
}

// ---- device-resident Levenberg-Marquardt ------------------------------------------------------------------------------
// The accept / reject logic of OptimizationAlgorithmLevenberg::solve (optimization_algorithm_levenberg.cpp:61-164), the
// iteration loop of SparseOptimizer::optimize and the two-stage schedule of LocalBundleAdjustment[WithOdom]
// (Optimizer.cc:2504-2570: optimize(5) with kernels, chi2 gate, optimize(10) without) live in BACtl in device memory.  The
// host enqueues identical "slots" of kernels without reading anything back; every kernel of a slot looks at BACtl and
// does nothing once the schedule has finished.  A slot is either the linearisation that opens an optimize() (needInit:
// chi2_0, lambda_0 = 1e-5 max diag) or one LM trial: Schur + solve at (cur, lambda) -> trial state -> its linearisation ->
// k_ba_control decides.  pbStopFlag: the host writes `abort` from a side stream while it waits.
struct BACtl {
  double lambda, ni, currentChi, iniChi;
  int phase;     // 0 = first optimize(), 1 = second optimize() (after the chi2 gate), 2 = finished
  int it, qmax, nBad;
  int cur;       // index of the accepted state and of its linearisation
  int needInit;  // this slot linearises the accepted state instead of running a trial
  int needGate;  // this slot starts with the chi2 gate
  int abort;     // terminate(): set by the host when it sees *pbStopFlag
  int trials, slots;
  int badArgs;   // device-side graph builder (fb_local_ba_dev): 1 = index out of range, 2 = duplicate (key frame, point)
};
struct BASched { int its1, robust1, gate, its2; };
struct St2 { State s[2]; };
struct Lb2 { LinBuf b[2]; };
// The per-linearisation "exchange block" (one per buffer t, `stride` doubles apart): everything a rank contributes to the
// pose system of a linearisation, contiguous so that the sharded BA sums it over the ranks with ONE all-reduce:
//   [0, oH) key-frame parts (np x POSE_PARTS x 27) | [oH, oB) HppO (P6 x P6) | [oB, oS) bpO (P6) |
//   [oS, oM) chi2, scale term, abort request, spare | [oM, stride) max |diag Hll| of each rank (own slot, others 0)
struct XBLay {
  double *base;
  int stride, oH, oB, oS, oM;
  __host__ __device__ double *at(int t) const { return base + (size_t)t * stride; }
};


template <int MAXT>
__global__ __launch_bounds__(SCHUR_THREADS) void k_ba_schur_c(BADev D, Lb2 lb, const BACtl *c, double *Dinv, double *Spart, int P6, int NT, int lmPerWg) {
  extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
  if (c->phase == 2 || c->needInit) return;
  schur_body<MAXT>(D, lb.b[c->cur], c->lambda, Dinv, Spart, P6, NT, lmPerWg, smem);
}
__global__ void k_ba_sumparts_c(const BACtl *c, double *Spart, int nWg, int n, int rows) {
  if (c->phase == 2 || c->needInit) return;
  sumparts_body(Spart, nWg, n, rows);
}
constexpr int SOLVE_C_THREADS = 512;  // wave 0 factors the next diagonal block, seven waves share the trailing update
__global__ __launch_bounds__(SOLVE_C_THREADS) void k_ba_solve_c(Lb2 lb, const BACtl *c, const double *Spart, int P6, int NT, double *xp, double *okFlag,
                                                                XBLay xb) {
  extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
  __shared__ int s_ok;
  if (c->phase == 2 || c->needInit) return;
  solve_lookahead<SOLVE_C_THREADS>(lb.b[c->cur], c->lambda, Spart, P6, NT, xp, okFlag, smem, s_ok, xb.at(c->cur) + xb.oH);
}
__global__ __launch_bounds__(LIN_THREADS) void k_ba_update_c(BADev D, Lb2 lb, St2 st, const BACtl *c, const double *Dinv, const double *xp, double *scalePart,
                                                             int countPoses) {
  __shared__ double s_part[LIN_THREADS / 64];
  if (c->phase == 2) return;
  if (c->needInit) {
    if (c->needGate)  // edges beyond chi2 5.991 (or behind the camera) go to level 1 (Optimizer.cc:1059-1073 / 2534-2565)
      for (int e = blockIdx.x * LIN_THREADS + threadIdx.x; e < D.nE; e += gridDim.x * LIN_THREADS) gate_edge(D, st.s[c->cur], 1, nullptr, e);
    return;
  }
  update_body(D, lb.b[c->cur], st.s[c->cur], st.s[1 - c->cur], Dinv, xp, c->lambda, scalePart, countPoses, s_part);
}
constexpr int POSE_PARTS = 4;
// The decision at the end of a slot (one workgroup): assemble the key frames' diagonal blocks and bp from the parts, sum the
// chi2 / scale partials in index order (thread 0, from LDS), maximum of the diagonal (computeLambdaInit,
// optimization_algorithm_levenberg.cpp:166-180), then the LM state machine.
// Sharded BA: the block of this rank is summed over the ranks between this kernel and the control kernel; the chi2 / scale
// partials are therefore added up here (index order), next to the abort request and this rank's max |diag Hll|.
__global__ __launch_bounds__(256) void k_ba_prex(BADev D, Lb2 lb, const BACtl *c, int nLin, int chiSlot, const double *scalePart, int nScale, XBLay xb,
                                                 const int *abortLocal, int rank, int world) {
  __shared__ double s_m[256];
  const int tid = threadIdx.x;
  const int init = c->needInit, t = init ? c->cur : 1 - c->cur;
  double *X = xb.at(t);
  if (c->phase == 2) {  // keep taking part in the exchange with harmless numbers
    if (tid < 4) X[xb.oS + tid] = (tid == 2 && *abortLocal) ? 1.0 : 0.0;
    if (tid < world) X[xb.oM + tid] = 0.0;
    return;
  }
  const LinBuf &B = lb.b[t];
  double m = 0;
  if (init) for (int i = tid; i < nLin; i += 256) m = fmax(m, B.maxPart[i]);
  s_m[tid] = m;
  __syncthreads();
  if (tid != 0) return;
  double chi = 0;
  for (int i = 0; i < nLin; i++) chi += B.chiPart[i];
  chi += B.chiPart[chiSlot];
  double scale = 0;
  if (!init) for (int i = 0; i < nScale; i++) scale += scalePart[i];
  double mm = 0;
  for (int i = 0; i < 256; i++) mm = fmax(mm, s_m[i]);
  X[xb.oS] = chi; X[xb.oS + 1] = scale; X[xb.oS + 2] = *abortLocal ? 1.0 : 0.0; X[xb.oS + 3] = 0.0;
  for (int r = 0; r < world; r++) X[xb.oM + r] = r == rank ? mm : 0.0;
}

template <bool SHARDED>
__device__ __forceinline__ void control_body(const BADev &D, const Lb2 &lb, BACtl *c, const BASched &sc, int nLin, int chiSlot, const double *scalePart, int nScale,
                                             const double *okFlag, int P6, const XBLay &xb, const int *abortLocal, int world, double *s_m /*[256]*/,
                                             double *s_chiP /*[512]*/, double *s_scaleP /*[512]*/) {
  const int tid = threadIdx.x;
  if (c->phase == 2) { if (tid == 0) c->slots++; return; }
  const int init = c->needInit;
  const int t = init ? c->cur : 1 - c->cur;
  const LinBuf &B = lb.b[t];
  const double *XB = xb.at(t);
  const double *HppO = XB + xb.oH, *bpO = XB + xb.oB, *poseP = XB;
  double m = 0;
  for (int idx = tid; idx < D.np * 42; idx += 256) {  // 36 block entries + 6 bp entries per key frame
    const int k = idx / 42, r = idx - k * 42;
    const double *pp = poseP + (size_t)k * POSE_PARTS * 27;
    if (r < 36) {
      const int i = r / 6, j = r % 6;
      const int a = i < j ? i : j, b2 = i < j ? j : i;
      const int u = a * 6 - a * (a - 1) / 2 + (b2 - a);  // upper-triangular packing
      double v = 0;
      for (int q = 0; q < POSE_PARTS; q++) v += pp[q * 27 + u];
      B.Hpp[(size_t)(6 * k + i) * P6 + 6 * k + j] = v;
      if (i == j && init) m = fmax(m, fabs(v + HppO[(size_t)(6 * k + i) * P6 + 6 * k + i]));
    } else {
      const int i = r - 36;
      double v = 0;
      for (int q = 0; q < POSE_PARTS; q++) v += pp[q * 27 + 21 + i];
      B.bp[6 * k + i] = v + bpO[6 * k + i];
    }
  }
  if (!SHARDED) {
    for (int i = tid; i < min(nLin, 512); i += 256) s_chiP[i] = B.chiPart[i];
    if (!init) for (int i = tid; i < min(nScale, 512); i += 256) s_scaleP[i] = scalePart[i];
    if (init) for (int i = tid; i < nLin; i += 256) m = fmax(m, B.maxPart[i]);
  }
  s_m[tid] = m;
  __syncthreads();
  double chi = 0, scale = 0;
  if (!SHARDED && tid < 64) {
    // the partials in a fixed order: lane t adds entries t, t + 64, ..., then the butterfly over the wave (deterministic,
    // and a few hundred dependent additions on one lane were a third of this kernel)
    for (int i = tid; i < nLin; i += 64) chi += i < 512 ? s_chiP[i] : B.chiPart[i];
    if (!init) for (int i = tid; i < nScale; i += 64) scale += i < 512 ? s_scaleP[i] : scalePart[i];
    chi = wave_sum_d(chi);
    scale = wave_sum_d(scale);
  }
  if (tid != 0) return;
  int abortReq;
  if (SHARDED) {  // reduced over the ranks: identical on all of them, and so is every decision below
    chi = XB[xb.oS]; scale = XB[xb.oS + 1];
    abortReq = XB[xb.oS + 2] > 0.0;
  } else {
    chi += B.chiPart[chiSlot];
    abortReq = *abortLocal;
  }
  double maxDiag = 0;
  if (init) {
    for (int i = 0; i < 256; i++) maxDiag = fmax(maxDiag, s_m[i]);
    if (SHARDED) for (int r = 0; r < world; r++) maxDiag = fmax(maxDiag, XB[xb.oM + r]);
  }
  BACtl k = *c;
  k.abort = abortReq;
  const int its = k.phase == 0 ? sc.its1 : sc.its2;
  bool endOpt = false;
  k.slots++;
  k.needGate = 0;
  if (init) {
    k.needInit = 0;
    k.currentChi = chi; k.iniChi = chi;
    k.lambda = 1e-5 * maxDiag; k.ni = 2; k.nBad = 0; k.it = 0; k.qmax = 0;
    if (its <= 0 || k.abort) endOpt = true;
  } else {
    k.trials++;
    double tempChi = chi;
    if (*okFlag == 0.0) tempChi = 1.7976931348623157e308;  // solver failure (levenberg.cpp:126-127)
    double rho = k.currentChi - tempChi;
    rho /= scale + 1e-3;
    if (rho > 0 && isfinite(tempChi)) {
      double alpha = 1. - pow((2 * rho - 1), 3);
      alpha = fmin(alpha, 2. / 3.);
      k.lambda *= fmax(1. / 3., alpha);
      k.ni = 2;
      k.currentChi = tempChi;
      k.cur = 1 - k.cur;  // discardTop: the trial state and its linearisation become current
    } else {
      k.lambda *= k.ni;
      k.ni *= 2;  // pop
    }
    k.qmax++;
    if (!(rho < 0 && k.qmax < 10 && !k.abort)) {  // the iteration is over
      if (k.qmax == 10 || rho == 0) endOpt = true;
      else {
        if ((k.iniChi - k.currentChi) * 1e3 < k.iniChi) k.nBad++;
        else k.nBad = 0;
        if (k.nBad >= 3) endOpt = true;
      }
      if (!endOpt) {
        k.it++;
        if (k.it >= its || k.abort) endOpt = true;
        else { k.iniChi = k.currentChi; k.qmax = 0; }
      }
    }
  }
  if (endOpt) {
    // the chi2 gate runs in the next slot's k_ba_update_c launch (that slot only linearises)
    if (k.phase == 0 && sc.gate && !k.abort && D.nE > 0) { k.phase = 1; k.needInit = 1; k.needGate = 1; }
    else k.phase = 2;
  }
  *c = k;
}

template <bool SHARDED>
__global__ __launch_bounds__(256) void k_ba_control(BADev D, Lb2 lb, BACtl *c, BASched sc, int nLin, int chiSlot, const double *scalePart, int nScale,
                                                    const double *okFlag, int P6, XBLay xb, const int *abortLocal, int world) {
  __shared__ double s_m[256];
  __shared__ double s_chiP[512], s_scaleP[512];
  control_body<SHARDED>(D, lb, c, sc, nLin, chiSlot, scalePart, nScale, okFlag, P6, xb, abortLocal, world, s_m, s_chiP, s_scaleP);
}

// The linearisation of a slot as ONE launch of 256-thread workgroups with three roles (they are independent of each
// other): blocks [0, nLin) = landmarks (Hll, bl, W, chi2); the next POSE_PARTS * np blocks = a quarter of the edges of one
// free key frame each (27 partial sums of its Hpp diagonal block and bp); the last block = the odometry edges (their
// pose-pose blocks HppO, bpO).  The dense Hpp is never materialised: k_ba_control adds the key-frame parts into the diagonal
// blocks and bp, k_ba_solve reads diagonal blocks + HppO.  Scratch per linearisation buffer t: [HppO | bpO], poseP.
// (Measured and dropped: the workgroup that finishes last -- a ticket in BACtl behind a device-scope fence -- taking the slot's
// decision itself instead of a k_ba_control launch.  A device-scope release / acquire on this part writes back and invalidates
// the XCD's L2 (buffer_wbl2 sc1 / buffer_inv sc1), once per workgroup: the linearisation went from 21 to 49 us.)
template <bool GLOBAL>
__global__ __launch_bounds__(256) void k_ba_lin_c(BADev D, St2 st, Lb2 lb, const BACtl *c, BASched sc, int P6, int nLin, int chiSlot, OdomLin *olGlobal,
                                                  XBLay xb) {
  extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
  __shared__ double s_buf[256];
  if (c->phase == 2) return;
  const int t = c->needInit ? c->cur : 1 - c->cur;
  const int robust = c->phase == 0 ? sc.robust1 : 0;
  const int bx = blockIdx.x;
  if (bx < nLin) {
    linearize_quad_body<256>(D, st.s[t], lb.b[t], robust, s_buf);
  } else if (bx < nLin + POSE_PARTS * D.np) {
    const int q = bx - nLin, k = q / POSE_PARTS, part = q - k * POSE_PARTS;
    pose_body(D, st.s[t], lb.b[t], robust, P6, reinterpret_cast<double(*)[27]>(s_buf), k, part, POSE_PARTS, xb.at(t) + (size_t)q * 27);
  } else {
    OdomLin *ol = GLOBAL ? olGlobal : reinterpret_cast<OdomLin *>(smem);
    double *tmp = GLOBAL ? nullptr : reinterpret_cast<double *>(ol + max(D.nO, 1));  // the launch reserves 108 doubles per edge behind the records
    odom_body<GLOBAL>(D, st.s[t], lb.b[t], P6, chiSlot, ol, s_buf, xb.at(t) + xb.oH, xb.at(t) + xb.oB, true, tmp);
  }
}


__global__ void k_ba_gate_final_c(BADev D, St2 st, const BACtl *c, uint8_t *outFlag) {
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e < D.nE) gate_edge(D, st.s[c->cur], 0, outFlag, e);
}

__global__ void k_ba_export(int n_kf, int npt, const SE3 *pose, const double *pt, const uint8_t *fixed, float *kfT, float *ptOut) {
  const int g = blockIdx.x * blockDim.x + threadIdx.x;
  if (g < n_kf) {
    if (!fixed[g]) fb::se3_to_float12(pose[g], kfT + 12 * g);
  } else if (g < n_kf + npt * 3) ptOut[g - n_kf] = (float)pt[g - n_kf];
}

// sharded result: every rank contributes its own landmarks (points as exported floats) and edge flags, the others zeros
__global__ void k_ba_final_pack(int npt, int nE, int rank, int world, const float *ptOut, const uint8_t *flags, double *ex) {
  const int g = blockIdx.x * blockDim.x + threadIdx.x;
  if (g < npt * 3) ex[g] = ((g / 3) % world == rank) ? (double)ptOut[g] : 0.0;
  else if (g < npt * 3 + nE) ex[g] = flags[g - npt * 3];
}
__global__ void k_ba_final_unpack(int npt, int nE, const double *ex, float *ptOut, uint8_t *flags) {
  const int g = blockIdx.x * blockDim.x + threadIdx.x;
  if (g < npt * 3) ptOut[g] = (float)ex[g];
  else if (g < npt * 3 + nE) flags[g - npt * 3] = ex[g] != 0.0;
}

__global__ void k_ba_export_c(int n_kf, int npt, St2 st, const BACtl *c, const uint8_t *fixed, float *kfT, float *ptOut) {
  const SE3 *pose = st.s[c->cur].pose;
  const double *pt = st.s[c->cur].pt;
  const int g = blockIdx.x * blockDim.x + threadIdx.x;
  if (g < n_kf) {
    if (!fixed[g]) fb::se3_to_float12(pose[g], kfT + 12 * g);
  } else if (g < n_kf + npt * 3) ptOut[g - n_kf] = (float)pt[g - n_kf];
}

}  // namespace

#define BA_UP(buf, vec) FB_TRY(buf.upload((vec).data(), (vec).size() * sizeof((vec)[0])))

#include "ba_big.inc"

// ---- exchange transport of the landmark-sharded BA ------------------------------------------------------------------
// RCCL: the all-reduces are enqueued on the BA's stream and work on device buffers (nothing is staged through the host);
// the entry points are resolved at run time from the RCCL the process already has (torch's librccl.so.1 when the host is
// Python, /opt/rocm/lib otherwise), so the library carries no link-time dependency and single-GPU users never load it.
// HOST: the fb_allreduce_fn callback of fb_local_ba_sharded (gloo in the CPU tests; host buffer).
#include <dlfcn.h>
namespace {
struct RcclApi {
  void *lib = nullptr;
  int (*GetUniqueId)(void *) = nullptr;                              // ncclGetUniqueId(ncclUniqueId *)
  int (*CommInitRank)(void **, int, fb_rccl_unique_id, int) = nullptr; // ncclCommInitRank(comm *, nranks, id BY VALUE, rank)
  int (*CommDestroy)(void *) = nullptr;
  int (*AllReduce)(const void *, void *, size_t, int, int, void *, hipStream_t) = nullptr;
  const char *(*GetErrorString)(int) = nullptr;
  int (*CommCount)(void *, int *) = nullptr;
  int (*CommUserRank)(void *, int *) = nullptr;
};
RcclApi *rccl_api() {
  // loaded once (a function-local static is initialised thread-safely)
  static RcclApi api = [] {
    RcclApi a;
    for (const char *name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
      a.lib = dlopen(name, RTLD_NOW | RTLD_LOCAL);
      if (a.lib) break;
    }
    if (a.lib) {
      a.GetUniqueId = reinterpret_cast<decltype(a.GetUniqueId)>(dlsym(a.lib, "ncclGetUniqueId"));
      a.CommInitRank = reinterpret_cast<decltype(a.CommInitRank)>(dlsym(a.lib, "ncclCommInitRank"));
      a.CommDestroy = reinterpret_cast<decltype(a.CommDestroy)>(dlsym(a.lib, "ncclCommDestroy"));
      a.AllReduce = reinterpret_cast<decltype(a.AllReduce)>(dlsym(a.lib, "ncclAllReduce"));
      a.GetErrorString = reinterpret_cast<decltype(a.GetErrorString)>(dlsym(a.lib, "ncclGetErrorString"));
      a.CommCount = reinterpret_cast<decltype(a.CommCount)>(dlsym(a.lib, "ncclCommCount"));
      a.CommUserRank = reinterpret_cast<decltype(a.CommUserRank)>(dlsym(a.lib, "ncclCommUserRank"));
      if (!a.GetUniqueId || !a.CommInitRank || !a.CommDestroy || !a.AllReduce) a.lib = nullptr;
    }
    return a;
  }();
  return api.lib ? &api : nullptr;
}
constexpr int kNcclDouble = 8, kNcclSum = 0;  // ncclFloat64, ncclSum (rccl.h)

struct Xchg {
  int world = 1;
  void *comm = nullptr;            // ncclComm_t, or
  fb_allreduce_fn cb = nullptr;    // host callback
  void *ctx = nullptr;
  bool active() const { return world > 1 || comm != nullptr; }  // a 1-rank communicator still goes through RCCL (tests)
  // in-place sum over the ranks of n doubles in DEVICE memory, ordered behind the work already on stream s
  // (src may differ from dbuf: out-of-place, the source stays as it is)
  int sum_dev(double *dbuf, size_t n, hipStream_t s, std::vector<double> &scratch, const double *src = nullptr) const {
    if (!src) src = dbuf;
    if (!active() || n == 0) return FB_OK;
    if (comm) {
      const int rc = rccl_api()->AllReduce(src, dbuf, n, kNcclDouble, kNcclSum, comm, s);
      if (rc != 0) { fb::set_error("fb_local_ba_sharded: ncclAllReduce failed: %s", rccl_api()->GetErrorString ? rccl_api()->GetErrorString(rc) : "?"); return FB_ERR_HIP; }
      return FB_OK;
    }
    scratch.resize(n);
    FB_HIP(hipStreamSynchronize(s));
    FB_HIP(hipMemcpy(scratch.data(), src, n * 8, hipMemcpyDeviceToHost));
    if (cb(ctx, scratch.data(), (int32_t)n, 0) != 0) { fb::set_error("fb_local_ba_sharded: all-reduce callback failed"); return FB_ERR_ARG; }
    FB_HIP(hipMemcpy(dbuf, scratch.data(), n * 8, hipMemcpyHostToDevice));
    return FB_OK;
  }
  // in-place reduction of n doubles in HOST memory (op 0 = sum, 1 = max); the host-driven schedule uses it
  int reduce_host(double *hbuf, int n, int op) const {
    if (!active() || n <= 0) return FB_OK;
    if (cb) {
      if (cb(ctx, hbuf, n, op) != 0) { fb::set_error("fb_local_ba_sharded: all-reduce callback failed"); return FB_ERR_ARG; }
      return FB_OK;
    }
    fb::DevBuf d;
    FB_TRY(d.upload(hbuf, (size_t)n * 8));
    const int rc = rccl_api()->AllReduce(d.p, d.p, (size_t)n, kNcclDouble, op == 0 ? kNcclSum : 2 /* ncclMax */, comm, nullptr);
    if (rc != 0) { fb::set_error("fb_local_ba_sharded: ncclAllReduce failed (%d)", rc); return FB_ERR_HIP; }
    FB_HIP(hipStreamSynchronize(nullptr));
    return d.download(hbuf, (size_t)n * 8);
  }
};
}  // namespace

#ifdef FB_BA_STAMPS
extern "C" int fb_ba_debug_stamps(uint64_t *dst16) {  // probe build only: copies and clears the phase accumulators of the solve
  unsigned long long h[16], z[16] = {0};
  FB_HIP(hipMemcpyFromSymbol(h, HIP_SYMBOL(g_ba_stamps), sizeof(h)));
  FB_HIP(hipMemcpyToSymbol(HIP_SYMBOL(g_ba_stamps), z, sizeof(z)));
  for (int i = 0; i < 16; i++) dst16[i] = h[i];
  return FB_OK;
}
#endif

extern "C" int fb_rccl_get_unique_id(fb_rccl_unique_id *id) {
  FB_ARG(id);
  RcclApi *r = rccl_api();
  if (!r) { fb::set_error("fb_rccl_get_unique_id: no RCCL in this process (librccl.so.1 not found)"); return FB_ERR_NODEVICE; }
  const int rc = r->GetUniqueId(id);
  if (rc != 0) { fb::set_error("ncclGetUniqueId failed (%d)", rc); return FB_ERR_HIP; }
  return FB_OK;
}
extern "C" int fb_rccl_comm_info(void *comm, int *count, int *rank) {
  FB_ARG(comm && count && rank);
  RcclApi *r = rccl_api();
  if (!r || !r->CommCount || !r->CommUserRank) { fb::set_error("fb_rccl_comm_info: ncclCommCount / ncclCommUserRank not available"); return FB_ERR_NODEVICE; }
  int rc = r->CommCount(comm, count);
  if (rc == 0) rc = r->CommUserRank(comm, rank);
  if (rc != 0) { fb::set_error("fb_rccl_comm_info: RCCL error %d", rc); return FB_ERR_HIP; }
  return FB_OK;
}
extern "C" int fb_rccl_comm_init(const fb_rccl_unique_id *id, int rank, int world, void **comm) {
  FB_TRY(fb::check_device());
  FB_ARG(id && comm && world >= 1 && rank >= 0 && rank < world);
  RcclApi *r = rccl_api();
  if (!r) { fb::set_error("fb_rccl_comm_init: no RCCL in this process (librccl.so.1 not found)"); return FB_ERR_NODEVICE; }
  const int rc = r->CommInitRank(comm, world, *id, rank);
  if (rc != 0) { fb::set_error("ncclCommInitRank failed: %s", r->GetErrorString ? r->GetErrorString(rc) : "?"); return FB_ERR_HIP; }
  return FB_OK;
}
extern "C" int fb_rccl_comm_destroy(void *comm) {
  RcclApi *r = rccl_api();
  if (comm && r) r->CommDestroy(comm);
  return FB_OK;
}

// optimisation schedule: LocalBundleAdjustment[WithOdom] = optimize(5) robust, chi2 gate, optimize(10) plain
// (Optimizer.cc:2504-2560); BundleAdjustmentWithOdom = ONE optimize(nIterations), robust iff bRobust, no gate (:2048-2050)
struct BASchedule {
  int its1, robust1;
  bool gate;
  int its2;
  double delta;  // Huber delta: sqrt(5.991) local (:2290), sqrt(5.99) global (:1836)
};
struct DevIn { hipStream_t stream = nullptr; };  // fb_local_ba_dev: the big arrays of fb_local_ba_args are DEVICE pointers
static int local_ba_impl(const fb_local_ba_args *A, int rank, const Xchg &X, const BASchedule &sc, const DevIn *dv = nullptr);

// ---- device-side graph builder (fb_local_ba_dev): what local_ba_impl's host passes do, as kernels ----------------------
namespace {
struct BuildIn {
  const int32_t *obs_kf, *obs_mp; const float *obs_uv, *obs_inv;
  const int32_t *bobs_kf, *bobs_mpb; const float *bobs_xc, *bobs_inv;
  const float *kf_Tcw, *mp_xw, *mpb_xw;
  int nF, nB, n_kf, n_mp, n_mpb, odom;
  double wF, wB;
};
// per edge: the flat edge record (Optimizer.cc:2346-2367, 2399-2414) + the counts of the two CSR structures
__global__ void k_bld_edges(BuildIn I, const int *poseIdx, int *e_pt, int *e_kf, int *e_pj, uint8_t *e_type, uint8_t *e_level,
                            float *e_meas, double *e_info, int *lm_cnt, int *ps_cnt, int *bad) {
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= I.nF + I.nB) return;
  int pt, kf;
  if (e < I.nF) {
    pt = I.obs_mp[e]; kf = I.obs_kf[e];
    if (!(pt >= 0 && pt < I.n_mp && kf >= 0 && kf < I.n_kf)) { atomicOr(bad, 1); pt = 0; kf = 0; }
    e_type[e] = T_PROJ;
    e_meas[3 * e] = I.obs_uv[2 * e]; e_meas[3 * e + 1] = I.obs_uv[2 * e + 1]; e_meas[3 * e + 2] = 0.0f;
    e_info[e] = I.odom ? (1.0 * (double)I.obs_inv[e]) * I.wF : (double)I.obs_inv[e];
  } else {
    const int i = e - I.nF;
    int pb = I.bobs_mpb[i];
    kf = I.bobs_kf[i];
    if (!(pb >= 0 && pb < I.n_mpb && kf >= 0 && kf < I.n_kf)) { atomicOr(bad, 1); pb = 0; kf = 0; }
    pt = I.n_mp + pb;
    e_type[e] = T_XYZ;
    for (int k = 0; k < 3; k++) e_meas[3 * e + k] = I.bobs_xc[3 * i + k];
    e_info[e] = (1.0 * (double)I.bobs_inv[i]) * I.wB;
  }
  e_pt[e] = pt; e_kf[e] = kf; e_level[e] = 0;
  const int pj = poseIdx[kf];
  e_pj[e] = pj;
  atomicAdd(&lm_cnt[pt + 1], 1);
  if (pj >= 0) atomicAdd(&ps_cnt[pj + 1], 1);
}
__global__ void k_bld_state(BuildIn I, SE3 *poses, double *pts, float *kfT) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < I.n_kf) {
    poses[i] = fb::se3_from_float12(I.kf_Tcw + 12 * i);
    for (int k = 0; k < 12; k++) kfT[12 * i + k] = I.kf_Tcw[12 * i + k];
  }
  if (i < 3 * I.n_mp) pts[i] = I.mp_xw[i];
  if (i < 3 * I.n_mpb) pts[3 * I.n_mp + i] = I.mpb_xw[i];
}
// in-place: cnt[0] = 0, cnt[i + 1] = count of bucket i  ->  exclusive starts; one workgroup, n up to millions
__global__ __launch_bounds__(1024) void k_bld_scan(int *cnt, int n, int *fill) {
  __shared__ int s_w[16], s_run;
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  if (tid == 0) s_run = 0;
  __syncthreads();
  for (int base = 0; base <= n; base += 1024) {
    const int i = base + tid;
    const int v = i <= n ? cnt[i] : 0;
    int inc = v;
    for (int o = 1; o < 64; o <<= 1) { const int t = __shfl_up(inc, o, 64); if (lane >= o) inc += t; }
    if (lane == 63) s_w[wv] = inc;
    __syncthreads();
    int off = s_run;
    for (int w = 0; w < wv; w++) off += s_w[w];
    const int incl = off + inc;
    if (i <= n) { cnt[i] = incl; if (fill && i < n) fill[i] = incl; }  // start of bucket i = inclusive sum up to cnt[i] (cnt[0] = 0)
    __syncthreads();
    if (tid == 1023) s_run = incl;
    __syncthreads();
  }
}
__global__ void k_bld_scatter(int nE, const int *e_pt, const int *e_pj, int *fillL, int *fillP, int *lm_edges, int *ps_edges) {
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= nE) return;
  lm_edges[atomicAdd(&fillL[e_pt[e]], 1)] = e;
  if (e_pj[e] >= 0) ps_edges[atomicAdd(&fillP[e_pj[e]], 1)] = e;
}
// a landmark's edges in ascending edge index (the order the host builder produces: sums must not depend on the atomics'
// order) + a key frame observes a point at most once (map<KeyFrame*, size_t>)
__global__ void k_bld_sort_lm(int npt, const int *lm_start, int *lm_edges, const int *e_kf, int *bad) {
  const int l = blockIdx.x * blockDim.x + threadIdx.x;
  if (l >= npt) return;
  const int a0 = lm_start[l], a1 = lm_start[l + 1];
  for (int i = a0 + 1; i < a1; i++) {
    const int v = lm_edges[i];
    int j = i - 1;
    while (j >= a0 && lm_edges[j] > v) { lm_edges[j + 1] = lm_edges[j]; j--; }
    lm_edges[j + 1] = v;
  }
  for (int i = a0; i < a1; i++)
    for (int j = i + 1; j < a1; j++)
      if (e_kf[lm_edges[i]] == e_kf[lm_edges[j]]) atomicOr(bad, 2);
}
// a key frame's edges in ascending edge index: bitonic sort of its segment in LDS (one workgroup per free key frame)
__global__ __launch_bounds__(1024) void k_bld_sort_ps(const int *ps_start, int *ps_edges, int cap, int *bad) {
  extern __shared__ int s_v[];
  const int k = blockIdx.x, tid = threadIdx.x;
  const int a0 = ps_start[k], n = ps_start[k + 1] - a0;
  int m = 1;
  while (m < n) m <<= 1;
  if (m > cap) { if (tid == 0) atomicOr(bad, 4); return; }  // one key frame with more observations than the LDS sort holds
  for (int i = tid; i < m; i += 1024) s_v[i] = i < n ? ps_edges[a0 + i] : 0x7fffffff;
  __syncthreads();
  for (int kk = 2; kk <= m; kk <<= 1)
    for (int j = kk >> 1; j > 0; j >>= 1) {
      for (int i = tid; i < m; i += 1024) {
        const int p = i ^ j;
        if (p > i) {
          const int a = s_v[i], b = s_v[p];
          const bool up = (i & kk) == 0;
          if ((a > b) == up) { s_v[i] = b; s_v[p] = a; }
        }
      }
      __syncthreads();
    }
  for (int i = tid; i < n; i += 1024) ps_edges[a0 + i] = s_v[i];
}
__global__ void k_bld_check(const int *bad, BACtl *c) {
  if (*bad) { c->badArgs = *bad; c->phase = 2; }
}
}  // namespace
static const BASchedule kLocalSchedule = {5, 1, true, 10, (double)(float)sqrt(5.991)};

extern "C" int fb_local_ba(const fb_local_ba_args *A) { return local_ba_impl(A, 0, Xchg(), kLocalSchedule); }
extern "C" int fb_local_ba_dev(const fb_local_ba_args *A, void *stream) {
  DevIn dv;
  dv.stream = fb::as_stream(stream);
  return local_ba_impl(A, 0, Xchg(), kLocalSchedule, &dv);
}

extern "C" int fb_global_ba(const fb_local_ba_args *A, int n_iterations, int robust) {
  FB_ARG(n_iterations >= 0);
  const BASchedule sc = {n_iterations, robust ? 1 : 0, false, 0, (double)(float)sqrt(5.99)};
  return local_ba_impl(A, 0, Xchg(), sc);
}

extern "C" int fb_local_ba_sharded(const fb_local_ba_args *A, int rank, int world, fb_allreduce_fn allreduce, void *ctx) {
  FB_ARG(world >= 1 && rank >= 0 && rank < world && (world == 1 || allreduce));
  Xchg X;
  X.world = world; X.cb = allreduce; X.ctx = ctx;
  return local_ba_impl(A, rank, X, kLocalSchedule);
}

extern "C" int fb_local_ba_sharded_rccl(const fb_local_ba_args *A, int rank, int world, void *comm) {
  FB_ARG(world >= 1 && rank >= 0 && rank < world && comm);
  if (!rccl_api()) { fb::set_error("fb_local_ba_sharded_rccl: no RCCL in this process"); return FB_ERR_NODEVICE; }
  Xchg X;
  X.world = world; X.comm = comm;
  return local_ba_impl(A, rank, X, kLocalSchedule);
}

// Landmark-partitioned BA: rank r owns the landmarks l with l % world == r and all their edges, the
// odometry edges live on rank 0, the keyframe state is replicated.  Per LM trial two small all-reduces: the
// Schur-reduced system (after k_ba_schur) and [Hpp, bp, chi2, scale] (after the linearisation at the trial state).
// per host thread and device: the side stream for the abort request, the pinned mirror of the control block and the event
// the host waits on (concurrent callers must not share them; a stream belongs to its device).  fb_shutdown releases the
// calling thread's set.
struct PerDev { hipStream_t sAux = nullptr; BACtl *hCtl = nullptr; hipEvent_t evDone = nullptr; };
static thread_local PerDev g_perDev[64];

extern "C" int fb_shutdown(void) {
  int cur = 0;
  const bool have = hipGetDevice(&cur) == hipSuccess;
  for (int d = 0; d < 64; d++) {
    PerDev &pd = g_perDev[d];
    if (!pd.sAux && !pd.hCtl && !pd.evDone) continue;
    if (hipSetDevice(d) != hipSuccess) continue;
    if (pd.evDone) (void)hipEventDestroy(pd.evDone);
    if (pd.hCtl) (void)hipHostFree(pd.hCtl);
    if (pd.sAux) (void)hipStreamDestroy(pd.sAux);
    pd = PerDev();
  }
  if (have) (void)hipSetDevice(cur);
  fb::pool_release();
  return FB_OK;
}

static int local_ba_impl(const fb_local_ba_args *A, int rank, const Xchg &X, const BASchedule &sc, const DevIn *dv) {
  FB_TRY(fb::check_device());
  const int world = X.world;
  const bool timing = getenv("FB_BA_TIMING") != nullptr;  // host-side phase times on stderr (probe)
  const auto tStart = std::chrono::steady_clock::now();
  auto lap = [&](const char *what) {
    if (timing) fprintf(stderr, "[fb_local_ba] %-28s %8.1f us\n", what, std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - tStart).count());
  };
  const bool sharded = X.active();
  const bool devIn = dv != nullptr;   // observations / poses / points / result arrays live in HBM (fb_local_ba_dev)
  hipStream_t s0 = devIn ? dv->stream : nullptr;
  FB_ARG(!(devIn && sharded));
  auto reduce = [&](double *buf, int n, int op) -> int { return sharded ? X.reduce_host(buf, n, op) : FB_OK; };
  // (sharded: a rank must not return alone on ITS argument error -- the others would wait for it in the first exchange;
  // the agreement block below reports it on every rank together)
  if (!sharded) FB_ARG(A && A->n_kf > 0 && A->n_mp >= 0 && A->n_mpb >= 0 && A->n_obs >= 0 && A->kf_Tcw && A->kf_fixed);
  // Optimizer.cc:902-906 / 2498-2500.  Sharded: a rank must not leave on its OWN view of the flag (the others would wait for
  // it in the first exchange): every stop decision below goes through a reduction
  if (!sharded && A->stop_flag && *A->stop_flag) return FB_OK;
  const bool argsOk = A && A->n_kf > 0 && A->n_mp >= 0 && A->n_mpb >= 0 && A->n_obs >= 0 && A->kf_Tcw && A->kf_fixed;
  if (sharded) {
    // Agreement on the arguments before the first exchange: a rank that rejected its input alone would leave the others
    // waiting in a collective.  (The same checks run again below, where they can no longer fail.)
    int bad = argsOk ? 0 : 1;
    if (!bad) {
      const bool od = A->with_odom != 0;
      int free = 0;
      for (int k = 0; k < A->n_kf; k++) free += A->kf_fixed[k] ? 0 : 1;
      if (6 * free > 4096) bad = 1;
      std::vector<long long> pairs;
      pairs.reserve((size_t)A->n_obs + (od ? A->n_bobs : 0));
      for (int i = 0; i < A->n_obs && !bad; i++) {
        if (!(A->obs_mp[i] >= 0 && A->obs_mp[i] < A->n_mp && A->obs_kf[i] >= 0 && A->obs_kf[i] < A->n_kf)) bad = 1;
        else pairs.push_back(((long long)A->obs_mp[i] << 32) | (unsigned)A->obs_kf[i]);
      }
      for (int i = 0; od && i < A->n_bobs && !bad; i++) {
        if (!(A->bobs_mpb[i] >= 0 && A->bobs_mpb[i] < A->n_mpb && A->bobs_kf[i] >= 0 && A->bobs_kf[i] < A->n_kf)) bad = 1;
        else pairs.push_back(((long long)(A->n_mp + A->bobs_mpb[i]) << 32) | (unsigned)A->bobs_kf[i]);
      }
      for (int i = 0; od && rank == 0 && i < A->n_odom && !bad; i++)
        if (!(A->odom_kf_i[i] >= 0 && A->odom_kf_i[i] < A->n_kf && A->odom_kf_j[i] >= 0 && A->odom_kf_j[i] < A->n_kf)) bad = 1;
      if (!bad) {
        std::sort(pairs.begin(), pairs.end());
        if (std::adjacent_find(pairs.begin(), pairs.end()) != pairs.end()) bad = 1;
      }
    }
    double v = bad;
    FB_TRY(X.reduce_host(&v, 1, 1));
    if (v > 0.0) {
      fb::set_error(bad ? "fb_local_ba_sharded: bad arguments on this rank" : "fb_local_ba_sharded: another rank rejected its arguments");
      return FB_ERR_ARG;
    }
  }
  auto stopped = [&]() -> bool {
    double v = (A->stop_flag && *A->stop_flag) ? 1.0 : 0.0;
    if (sharded && X.reduce_host(&v, 1, 1) != FB_OK) return true;
    return v > 0.0;
  };
  const bool odom = A->with_odom != 0;
  const int n_kf = A->n_kf, n_mp = A->n_mp, n_mpb = odom ? A->n_mpb : 0;
  const int npt = A->n_mp + A->n_mpb;  // bird points keep their slots even when unused
  const int nF = A->n_obs, nB = odom ? A->n_bobs : 0, nE = nF + nB, nO = (odom && rank == 0) ? A->n_odom : 0;
  (void)n_mpb;
  // ---- host preprocessing: indices, CSR by landmark / pose.  Everything the kernels read is laid out in ONE host staging
  //      buffer and goes to the device with one copy (two dozen small synchronous copies were a quarter of a millisecond)
  std::vector<int> poseIdx(n_kf, -1);
  int np = 0;
  for (int k = 0; k < n_kf; k++) if (!A->kf_fixed[k]) poseIdx[k] = np++;
  const int P6 = 6 * np;
  if (P6 > 4096) { fb::set_error("fb_local_ba: more than 682 free keyframes"); return FB_ERR_CAPACITY; }
  static thread_local std::vector<uint8_t> stage;
  size_t stageBytes = 0;
  auto reserve = [&](size_t bytes) { const size_t off = stageBytes; stageBytes += (bytes + 255) & ~(size_t)255; return off; };
  const int nE1 = std::max(nE, 1), nO1 = std::max(nO, 1);
  // small, host-built arrays first: the device-input variant uploads only this header
  const size_t o_poseIdx = reserve((size_t)n_kf * 4), o_fixed = reserve(n_kf), o_oi = reserve((size_t)nO1 * 4), o_oj = reserve((size_t)nO1 * 4),
               o_oz = reserve((size_t)nO1 * sizeof(SE3)), o_oinfo = reserve((size_t)nO1 * 8), o_ods = reserve((size_t)(np + 1) * 4),
               o_ode = reserve((size_t)(2 * nO1) * 4), o_ctl = reserve(sizeof(BACtl)), o_abort = reserve(16);
  const size_t headerBytes = stageBytes;   // (the control block and the abort word ride in the header: no separate synchronous copies)
  const size_t o_lms = reserve((size_t)(npt + 1) * 4), o_pss = reserve((size_t)(np + 1) * 4);   // (zeroed together by the device builder)
  const size_t o_ept = reserve((size_t)nE1 * 4), o_ekf = reserve((size_t)nE1 * 4),
               o_epj = reserve((size_t)nE1 * 4), o_etype = reserve(nE1), o_elevel = reserve(nE1), o_emeas = reserve((size_t)nE1 * 12),
               o_einfo = reserve((size_t)nE1 * 8), o_lme = reserve((size_t)nE1 * 4), o_pse = reserve((size_t)nE1 * 4),
               o_poses = reserve((size_t)n_kf * sizeof(SE3)), o_pts = reserve((size_t)std::max(npt, 1) * 24),
               o_kfT = reserve((size_t)n_kf * 48);
  if (stage.size() < stageBytes) stage.resize(stageBytes);
  uint8_t *hs = stage.data();
  int *h_poseIdx = reinterpret_cast<int *>(hs + o_poseIdx), *e_pt = reinterpret_cast<int *>(hs + o_ept), *e_kf = reinterpret_cast<int *>(hs + o_ekf),
      *e_pj = reinterpret_cast<int *>(hs + o_epj), *lm_start = reinterpret_cast<int *>(hs + o_lms), *lm_edges = reinterpret_cast<int *>(hs + o_lme),
      *ps_start = reinterpret_cast<int *>(hs + o_pss), *ps_edges = reinterpret_cast<int *>(hs + o_pse), *o_i = reinterpret_cast<int *>(hs + o_oi),
      *o_j = reinterpret_cast<int *>(hs + o_oj), *od_start = reinterpret_cast<int *>(hs + o_ods), *od_edges = reinterpret_cast<int *>(hs + o_ode);
  uint8_t *e_type = hs + o_etype, *e_level = hs + o_elevel;
  float *e_meas = reinterpret_cast<float *>(hs + o_emeas);
  double *e_info = reinterpret_cast<double *>(hs + o_einfo), *o_info = reinterpret_cast<double *>(hs + o_oinfo);
  SE3 *oZinv = reinterpret_cast<SE3 *>(hs + o_oz), *poses = reinterpret_cast<SE3 *>(hs + o_poses);
  double *pts = reinterpret_cast<double *>(hs + o_pts);
  memcpy(h_poseIdx, poseIdx.data(), (size_t)n_kf * 4);
  memcpy(hs + o_fixed, A->kf_fixed, n_kf);
  {  // initial Levenberg-Marquardt control block + abort word of the device-resident schedule
    BACtl init;
    memset(&init, 0, sizeof(init));
    // (sharded: nE / nO are this rank's view; every rank holds all edges, so `anything` agrees across the ranks)
    const bool anything = nE + (odom ? A->n_odom : 0) > 0 && (np > 0 || npt > 0);
    init.phase = anything ? 0 : 2;
    init.needInit = 1;
    memcpy(hs + o_ctl, &init, sizeof(init));
    const int abort0 = (A->stop_flag && *A->stop_flag) ? 1 : 0;  // sharded: raised before the call on this rank only
    memset(hs + o_abort, 0, 16);
    memcpy(hs + o_abort, &abort0, sizeof(int));
  }
  if (!devIn) {
  memcpy(hs + o_kfT, A->kf_Tcw, (size_t)n_kf * 48);
  // one pass over the observations fills the per-edge arrays and counts the two CSR structures, a second one scatters
  for (int l = 0; l <= npt; l++) lm_start[l] = 0;
  for (int k = 0; k <= np; k++) ps_start[k] = 0;
  const double wFd = (double)A->wF, wBd = (double)A->wB;
  // Observations usually arrive grouped by point (the reference walks its local map points): then the CSR by landmark is
  // the edge order itself and the duplicate check (a key frame sees a point at most once) rides in this pass.
  static thread_local std::vector<int> seen;
  seen.assign(n_kf, -1);
  bool grouped = true, dup = false;
  int prevPt = -1;
  for (int i = 0; i < nF; i++) {
    const int pt = A->obs_mp[i], kf = A->obs_kf[i];
    FB_ARG(pt >= 0 && pt < n_mp && kf >= 0 && kf < n_kf);
    grouped = grouped && pt >= prevPt;
    prevPt = pt;
    dup = dup || seen[kf] == pt;
    seen[kf] = pt;
    e_pt[i] = pt; e_kf[i] = kf; e_type[i] = T_PROJ;
    e_meas[3 * i] = A->obs_uv[2 * i]; e_meas[3 * i + 1] = A->obs_uv[2 * i + 1]; e_meas[3 * i + 2] = 0.0f;
    e_info[i] = odom ? (1.0 * (double)A->obs_inv_sigma2[i]) * wFd : (double)A->obs_inv_sigma2[i];
    e_level[i] = (sharded && pt % world != rank) ? 2 : 0;  // 2 = not this rank's landmark
    const int pj = poseIdx[kf];
    e_pj[i] = pj;
    lm_start[pt + 1]++;
    if (pj >= 0) ps_start[pj + 1]++;
  }
  for (int i = 0; i < nB; i++) {
    FB_ARG(A->bobs_mpb[i] >= 0 && A->bobs_mpb[i] < A->n_mpb && A->bobs_kf[i] >= 0 && A->bobs_kf[i] < n_kf);
    const int e = nF + i, pt = n_mp + A->bobs_mpb[i], kf = A->bobs_kf[i];
    grouped = grouped && pt >= prevPt;
    prevPt = pt;
    dup = dup || seen[kf] == pt;
    seen[kf] = pt;
    e_pt[e] = pt; e_kf[e] = kf; e_type[e] = T_XYZ;
    for (int k = 0; k < 3; k++) e_meas[3 * e + k] = A->bobs_xc[3 * i + k];
    e_info[e] = (1.0 * (double)A->bobs_inv_sigma2[i]) * wBd;
    e_level[e] = (sharded && pt % world != rank) ? 2 : 0;
    const int pj = poseIdx[kf];
    e_pj[e] = pj;
    lm_start[pt + 1]++;
    if (pj >= 0) ps_start[pj + 1]++;
  }
  for (int l = 0; l < npt; l++) lm_start[l + 1] += lm_start[l];
  for (int k = 0; k < np; k++) ps_start[k + 1] += ps_start[k];
  {
    static thread_local std::vector<int> fillL, fillP;
    fillP.assign(ps_start, ps_start + np);
    if (grouped) {
      for (int e = 0; e < nE; e++) {
        lm_edges[e] = e;
        if (e_pj[e] >= 0) ps_edges[fillP[e_pj[e]]++] = e;
      }
    } else {
      fillL.assign(lm_start, lm_start + npt);
      for (int e = 0; e < nE; e++) {
        lm_edges[fillL[e_pt[e]]++] = e;
        if (e_pj[e] >= 0) ps_edges[fillP[e_pj[e]]++] = e;
      }
      // a keyframe observes a point at most once (map<KeyFrame*,size_t>): stamp per key frame = last landmark seen
      dup = false;
      seen.assign(n_kf, -1);
      for (int l = 0; l < npt && !dup; l++)
        for (int c = lm_start[l]; c < lm_start[l + 1]; c++) {
          int &sk = seen[e_kf[lm_edges[c]]];
          dup = dup || sk == l;
          sk = l;
        }
    }
    if (dup) { fb::set_error("fb_local_ba: duplicate (keyframe, point) observation"); return FB_ERR_ARG; }
  }
  for (int k = 0; k < n_kf; k++) poses[k] = fb::se3_from_float12(A->kf_Tcw + 12 * k);
  for (int i = 0; i < 3 * n_mp; i++) pts[i] = A->mp_xw[i];
  for (int i = 0; i < 3 * A->n_mpb; i++) pts[3 * n_mp + i] = A->mpb_xw[i];
  }  // !devIn
  for (int i = 0; i < nO; i++) {
    FB_ARG(A->odom_kf_i[i] >= 0 && A->odom_kf_i[i] < n_kf && A->odom_kf_j[i] >= 0 && A->odom_kf_j[i] < n_kf);
    o_i[i] = A->odom_kf_i[i]; o_j[i] = A->odom_kf_j[i]; o_info[i] = A->odom_info[i];
    oZinv[i] = fb::se3_inverse(fb::se3_from_float12(A->odom_Tij + 12 * i));
  }
  {
    std::vector<std::vector<int>> inc(np);
    for (int e = 0; e < nO; e++) {
      const int pi = poseIdx[o_i[e]], pj = poseIdx[o_j[e]];
      if (pi >= 0) inc[pi].push_back(e);
      if (pj >= 0 && pj != pi) inc[pj].push_back(e);
    }
    od_start[0] = 0;
    int q = 0;
    for (int k = 0; k < np; k++) { for (int e : inc[k]) od_edges[q++] = e; od_start[k + 1] = q; }
  }
  // ---- device buffers: the staged graph (one copy) + one scratch allocation carved below
  const int NT = (P6 + 1 + 15) / 16;
  const int rows = NT * 16;
  const int nLinBlocks = (npt + LIN_THREADS - 1) / LIN_THREADS;
  const int nUpdBlocks = (4 * npt + n_kf + LIN_THREADS - 1) / LIN_THREADS;  // four lanes per landmark + one per key frame
  int nWg = std::min(getenv("FB_BA_NWG") ? atoi(getenv("FB_BA_NWG")) : 256, std::max(1, (npt + CHUNK - 1) / CHUNK));
  const int lmPerWg = ((npt + nWg - 1) / nWg + CHUNK - 1) / CHUNK * CHUNK;
  nWg = std::max(1, (npt + lmPerWg - 1) / std::max(lmPerWg, 1));
  lap("host preprocessing done");
  fb::DevBuf d_stage, d_scratch, d_bld;
  if (!devIn) {
    FB_TRY(d_stage.upload(hs, stageBytes));
  } else {
    // only the header travels; the edge records, both CSR structures and the double-precision state are built by kernels
    // from the caller's device arrays (same contents as the host passes above, incl. the ascending edge order inside a
    // landmark / a key frame that the sums depend on)
    FB_TRY(d_stage.alloc(stageBytes));
    uint8_t *dsb = d_stage.as<uint8_t>();
    FB_HIP(hipMemcpyAsync(dsb, hs, headerBytes, hipMemcpyHostToDevice, s0));
    FB_ARG(A->kf_Tcw && (n_mp == 0 || A->mp_xw) && (A->n_mpb == 0 || A->mpb_xw));
    FB_ARG(nF == 0 || (A->obs_kf && A->obs_mp && A->obs_uv && A->obs_inv_sigma2 && A->obs_outlier));
    FB_ARG(nB == 0 || (A->bobs_kf && A->bobs_mpb && A->bobs_xc && A->bobs_inv_sigma2 && A->bobs_outlier));
    FB_TRY(d_bld.alloc(((size_t)npt + np + 2) * 4 + 16));
    int *fillL = d_bld.as<int>(), *fillP = fillL + npt + 1, *badDev = fillP + np + 1;
    FB_HIP(hipMemsetAsync(dsb + o_lms, 0, (o_ept - o_lms), s0));   // lm_start | ps_start (counts accumulate into them)
    FB_HIP(hipMemsetAsync(d_bld.p, 0, d_bld.bytes, s0));
    BuildIn I;
    I.obs_kf = A->obs_kf; I.obs_mp = A->obs_mp; I.obs_uv = A->obs_uv; I.obs_inv = A->obs_inv_sigma2;
    I.bobs_kf = A->bobs_kf; I.bobs_mpb = A->bobs_mpb; I.bobs_xc = A->bobs_xc; I.bobs_inv = A->bobs_inv_sigma2;
    I.kf_Tcw = A->kf_Tcw; I.mp_xw = A->mp_xw; I.mpb_xw = A->mpb_xw;
    I.nF = nF; I.nB = nB; I.n_kf = n_kf; I.n_mp = n_mp; I.n_mpb = A->n_mpb; I.odom = odom ? 1 : 0; I.wF = (double)A->wF; I.wB = (double)A->wB;
    int *dlms = reinterpret_cast<int *>(dsb + o_lms), *dpss = reinterpret_cast<int *>(dsb + o_pss);
    int *dept = reinterpret_cast<int *>(dsb + o_ept), *dekf = reinterpret_cast<int *>(dsb + o_ekf), *depj = reinterpret_cast<int *>(dsb + o_epj);
    int *dlme = reinterpret_cast<int *>(dsb + o_lme), *dpse = reinterpret_cast<int *>(dsb + o_pse);
    int segCap = 1;
    while (segCap < std::max(nE, 1) && segCap < 32768) segCap <<= 1;  // observations of ONE key frame: at most 32768 (128 KB of LDS)
    { fb::ProfScope pr(fb::P_BA_MISC, s0);
      if (nE > 0) k_bld_edges<<<(nE + 255) / 256, 256, 0, s0>>>(I, reinterpret_cast<const int *>(dsb + o_poseIdx), dept, dekf, depj, dsb + o_etype, dsb + o_elevel,
                                                              reinterpret_cast<float *>(dsb + o_emeas), reinterpret_cast<double *>(dsb + o_einfo), dlms, dpss, badDev);
      k_bld_state<<<(std::max(n_kf, 3 * std::max(n_mp, A->n_mpb)) + 255) / 256, 256, 0, s0>>>(I, reinterpret_cast<SE3 *>(dsb + o_poses), reinterpret_cast<double *>(dsb + o_pts),
                                                                                              reinterpret_cast<float *>(dsb + o_kfT));
      k_bld_scan<<<1, 1024, 0, s0>>>(dlms, npt, fillL);
      k_bld_scan<<<1, 1024, 0, s0>>>(dpss, np, fillP);
      if (nE > 0) {
        k_bld_scatter<<<(nE + 255) / 256, 256, 0, s0>>>(nE, dept, depj, fillL, fillP, dlme, dpse);
        k_bld_sort_lm<<<(npt + 255) / 256, 256, 0, s0>>>(npt, dlms, dlme, dekf, badDev);
        if (np > 0) {
          FB_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(k_bld_sort_ps), hipFuncAttributeMaxDynamicSharedMemorySize, segCap * 4));
          k_bld_sort_ps<<<np, 1024, (size_t)segCap * 4, s0>>>(dpss, dpse, segCap, badDev);
        }
      }
      FB_HIP(hipGetLastError()); }
  }
  lap("graph uploaded");
  uint8_t *ds = d_stage.as<uint8_t>();
  size_t scratchBytes = 0;
  auto carve = [&](size_t bytes) { const size_t off = scratchBytes; scratchBytes += (bytes + 255) & ~(size_t)255; return off; };
  const size_t c_echi2 = carve((size_t)nE1 * 8);
  size_t c_pose1 = carve((size_t)n_kf * sizeof(SE3)), c_pt1 = carve((size_t)std::max(npt, 1) * 24);
  size_t c_Hll[2], c_bl[2], c_W[2], c_Hpp[2], c_bp[2], c_chi[2], c_max[2];
  for (int q = 0; q < 2; q++) {
    c_Hll[q] = carve((size_t)std::max(npt, 1) * 72); c_bl[q] = carve((size_t)std::max(npt, 1) * 24); c_W[q] = carve((size_t)nE1 * 144);
    c_Hpp[q] = carve((size_t)std::max(P6 * P6, 1) * 8); c_bp[q] = carve((size_t)std::max(P6, 1) * 8); c_chi[q] = carve((size_t)(2 * nLinBlocks + 2) * 8); c_max[q] = carve((size_t)(2 * nLinBlocks + 2) * 8);
  }
  FB_TRY(d_scratch.alloc(scratchBytes));
  uint8_t *dc = d_scratch.as<uint8_t>();
  FB_HIP(hipMemsetAsync(dc + c_echi2, 0, (size_t)nE1 * 8, s0));
  // state 0 = the staged copy, state 1 starts as the same poses / points
  FB_HIP(hipMemcpyAsync(dc + c_pose1, ds + o_poses, (size_t)n_kf * sizeof(SE3), hipMemcpyDeviceToDevice, s0));
  FB_HIP(hipMemcpyAsync(dc + c_pt1, ds + o_pts, (size_t)std::max(npt, 1) * 24, hipMemcpyDeviceToDevice, s0));
  const uint8_t *d_fixed = ds + o_fixed;   // kf_fixed
  const uint8_t *d_kfT0 = ds + o_kfT;      // the caller's float poses (fixed key frames are returned untouched)
  BADev D;
  D.n_kf = n_kf; D.np = np; D.npt = npt; D.nE = nE; D.nO = nO; D.quat = odom ? 1 : 0;
  D.fx = A->fx; D.fy = A->fy; D.cx = A->cx; D.cy = A->cy; D.delta = sc.delta;
  D.e_pj = reinterpret_cast<int *>(ds + o_epj);
  D.poseIdx = reinterpret_cast<int *>(ds + o_poseIdx); D.e_pt = reinterpret_cast<int *>(ds + o_ept); D.e_kf = reinterpret_cast<int *>(ds + o_ekf);
  D.e_type = ds + o_etype; D.e_meas = reinterpret_cast<float *>(ds + o_emeas); D.e_info = reinterpret_cast<double *>(ds + o_einfo);
  D.e_level = ds + o_elevel; D.e_chi2 = reinterpret_cast<double *>(dc + c_echi2);
  D.lm_start = reinterpret_cast<int *>(ds + o_lms); D.lm_edges = reinterpret_cast<int *>(ds + o_lme);
  D.ps_start = reinterpret_cast<int *>(ds + o_pss); D.ps_edges = reinterpret_cast<int *>(ds + o_pse);
  D.o_i = reinterpret_cast<int *>(ds + o_oi); D.o_j = reinterpret_cast<int *>(ds + o_oj); D.o_Zinv = reinterpret_cast<SE3 *>(ds + o_oz);
  D.o_info = reinterpret_cast<double *>(ds + o_oinfo);
  D.od_start = reinterpret_cast<int *>(ds + o_ods); D.od_edges = reinterpret_cast<int *>(ds + o_ode);
  State st[2];
  LinBuf lb[2];
  st[0].pose = reinterpret_cast<SE3 *>(ds + o_poses); st[0].pt = reinterpret_cast<double *>(ds + o_pts);
  st[1].pose = reinterpret_cast<SE3 *>(dc + c_pose1); st[1].pt = reinterpret_cast<double *>(dc + c_pt1);
  for (int q = 0; q < 2; q++) {
    lb[q].Hll = reinterpret_cast<double *>(dc + c_Hll[q]); lb[q].bl = reinterpret_cast<double *>(dc + c_bl[q]); lb[q].W = reinterpret_cast<double *>(dc + c_W[q]);
    lb[q].Hpp = reinterpret_cast<double *>(dc + c_Hpp[q]); lb[q].bp = reinterpret_cast<double *>(dc + c_bp[q]); lb[q].chiPart = reinterpret_cast<double *>(dc + c_chi[q]); lb[q].maxPart = reinterpret_cast<double *>(dc + c_max[q]);
  }
  const size_t schurLds = (size_t)2 * rows * KPAD * 8;
  const size_t solveLds = ((size_t)(P6 + 1) * (P6 + 1) + (size_t)(P6 + 1) * 6 + 96 + P6 + 2) * 8;  // the larger of k_ba_solve / solve_lookahead
  // beyond ~23 free key frames the reduced system no longer fits LDS: HBM-resident path of ba_big.inc
  const bool big = schurLds > 160 * 1024 || solveLds > 160 * 1024 || P6 + 1 > 256;
  fb::DevBuf d_Dinv, d_Spart, d_xp, d_ok, d_scale, d_scal, d_bigS, d_bigM, d_bigU, d_bigR;
  FB_TRY(d_Dinv.alloc((size_t)npt * 9 * 8)); FB_TRY(d_Spart.alloc(big ? 8 : (size_t)nWg * rows * rows * 8));
  fb::DevBuf d_blkStart, d_blkPr, d_blkPc, d_entEr, d_entEc;
  BigLists lists{};
  if (big) {
    FB_TRY(d_bigS.alloc((size_t)P6 * P6 * 8 + (size_t)P6 * 8));  // S | r
    FB_TRY(d_bigM.alloc((size_t)P6 * P6 * 8 + (size_t)P6 * 8));  // M | rhs
    FB_TRY(d_bigU.alloc((size_t)P6 * BIG_NB * 8));
    // (row key frame <= column key frame) -> the edge pairs of the landmarks both observe; counting sort by block
    std::vector<int> cnt((size_t)np * np + 1, 0);
    auto for_pairs = [&](auto &&f) {
      for (int l = 0; l < npt; l++)
        for (int a = lm_start[l]; a < lm_start[l + 1]; a++) {
          const int pa = poseIdx[e_kf[lm_edges[a]]];
          if (pa < 0) continue;
          for (int b2 = a; b2 < lm_start[l + 1]; b2++) {
            const int pb = poseIdx[e_kf[lm_edges[b2]]];
            if (pb < 0) continue;
            if (pa <= pb) f(pa, pb, lm_edges[a], lm_edges[b2]);
            else f(pb, pa, lm_edges[b2], lm_edges[a]);
          }
        }
    };
    for_pairs([&](int pr, int pc, int, int) { cnt[(size_t)pr * np + pc + 1]++; });
    std::vector<int> blkStart(1, 0), blkPr, blkPc, slot((size_t)np * np, -1);
    for (size_t key = 0; key < (size_t)np * np; key++)
      if (cnt[key + 1] > 0) {
        slot[key] = (int)blkPr.size();
        blkPr.push_back((int)(key / np));
        blkPc.push_back((int)(key % np));
        blkStart.push_back(blkStart.back() + cnt[key + 1]);
      }
    std::vector<int> fillp(blkStart.begin(), blkStart.end() - 1), entEr(blkStart.back()), entEc(blkStart.back());
    for_pairs([&](int pr, int pc, int er, int ec) {
      const int q = fillp[slot[(size_t)pr * np + pc]]++;
      entEr[q] = er;
      entEc[q] = ec;
    });
    BA_UP(d_blkStart, blkStart); BA_UP(d_blkPr, blkPr); BA_UP(d_blkPc, blkPc); BA_UP(d_entEr, entEr); BA_UP(d_entEc, entEc);
    lists.nBlocks = (int)blkPr.size();
    lists.blk_start = d_blkStart.as<int>(); lists.blk_pr = d_blkPr.as<int>(); lists.blk_pc = d_blkPc.as<int>();
    lists.ent_er = d_entEr.as<int>(); lists.ent_ec = d_entEc.as<int>();
  }
  FB_TRY(d_xp.alloc((size_t)std::max(P6, 1) * 8)); FB_TRY(d_ok.alloc(4)); FB_TRY(d_scale.alloc((size_t)nUpdBlocks * 8));
  FB_TRY(d_scal.alloc(4 * 8));
  FB_HIP(hipMemsetAsync(d_scal.p, 0, 4 * 8, s0));
  // accumulator tiles per wave: NT<=8 -> 9, NT<=12 -> 20, NT<=16 -> 34
  auto schurKernel = NT <= 8 ? k_ba_schur<9> : (NT <= 12 ? k_ba_schur<20> : k_ba_schur<34>);
  if (!big) {
    FB_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(schurKernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)schurLds));
    FB_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(k_ba_solve), hipFuncAttributeMaxDynamicSharedMemorySize, (int)solveLds));
  } else {
    FB_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(k_big_solve), hipFuncAttributeMaxDynamicSharedMemorySize, P6 * 8));
  }
  size_t odomLds = (size_t)std::max(nO, 1) * sizeof(OdomLin);
  fb::DevBuf d_ol;
  OdomLin *olGlobal = nullptr;
  if (odomLds > 150 * 1024) {  // long odometry chains: per-edge linearisations in HBM
    FB_TRY(d_ol.alloc(odomLds));
    olGlobal = d_ol.as<OdomLin>();
    odomLds = 0;
  } else {
    FB_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(k_ba_odom<false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)odomLds));
  }

  // ---- device-resident Levenberg-Marquardt (LDS-resident reduced system): no read-back inside the schedule.  Sharded:
  //      two all-reduces per slot on this stream (the Schur-reduced system; the exchange block of the linearisation), every
  //      rank enqueues the same slots and takes the same decisions from the reduced values.
  if (devIn && big) { fb::set_error("fb_local_ba_dev: more than 23 free key frames (use fb_local_ba)"); return FB_ERR_CAPACITY; }
  if (!big && (devIn || (!getenv("FB_BA_TRACE") && !getenv("FB_BA_HOST_LM")))) {
    // per host thread and device (concurrent callers must not share the pinned mirror; a stream belongs to its device):
    // side stream for the abort request (does not synchronise with the null stream), pinned mirror of the control block,
    // the event the host polls
    int devId = 0;
    FB_HIP(hipGetDevice(&devId));
    if (devId < 0 || devId >= 64) { fb::set_error("fb_local_ba: device id %d", devId); return FB_ERR_NODEVICE; }
    PerDev &pd = g_perDev[devId];
    if (!pd.sAux) FB_HIP(hipStreamCreateWithFlags(&pd.sAux, hipStreamNonBlocking));
    if (!pd.hCtl) FB_HIP(hipHostMalloc(reinterpret_cast<void **>(&pd.hCtl), sizeof(BACtl), hipHostMallocDefault));
    if (!pd.evDone) FB_HIP(hipEventCreateWithFlags(&pd.evDone, hipEventDisableTiming));
    hipStream_t sAux = pd.sAux;
    BACtl *hCtl = pd.hCtl;
    hipEvent_t evDone = pd.evDone;
    fb::DevBuf d_flags, d_kfT, d_ptOut, d_xb, d_ex;
    const bool anything = nE + (odom ? A->n_odom : 0) > 0 && (np > 0 || npt > 0);
    int *const d_abortp = reinterpret_cast<int *>(ds + o_abort);
    BACtl *ctl = reinterpret_cast<BACtl *>(ds + o_ctl);
    if (devIn) {  // a rejected graph (index out of range, duplicate observation) ends the schedule before it starts
      k_bld_check<<<1, 1, 0, s0>>>(d_bld.as<int>() + npt + 1 + np + 1, ctl);
      FB_HIP(hipGetLastError());
    }
    const BASched sched = {sc.its1, sc.robust1, sc.gate ? 1 : 0, sc.its2};
    St2 st2; st2.s[0] = st[0]; st2.s[1] = st[1];
    Lb2 lb2; lb2.b[0] = lb[0]; lb2.b[1] = lb[1];
    auto schurC = NT <= 8 ? k_ba_schur_c<9> : (NT <= 12 ? k_ba_schur_c<20> : k_ba_schur_c<34>);
    FB_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(schurC), hipFuncAttributeMaxDynamicSharedMemorySize, (int)schurLds));
    FB_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(k_ba_solve_c), hipFuncAttributeMaxDynamicSharedMemorySize, (int)solveLds));
    const int nS = rows * rows;
    FB_HIP(hipMemsetAsync(d_Spart.p, 0, (size_t)nS * 8, s0));  // the lower tiles of the summed system are never written: keep them finite
    const int nLin256 = (4 * npt + 255) / 256;  // four lanes per landmark; chi2 slot of the odometry role = nLin256
    XBLay xb;
    xb.oH = np * POSE_PARTS * 27; xb.oB = xb.oH + P6 * P6; xb.oS = xb.oB + P6; xb.oM = xb.oS + 4; xb.stride = xb.oM + world;
    // sharded: the kernels of a linearisation fill the RAW blocks; exchange 2 sums BOTH raw blocks into the REDUCED ones out of
    // place (which of the two the slot wrote is device-side knowledge; the raw block of the accepted linearisation is not
    // touched until it is overwritten, so re-reducing it is idempotent); k_ba_control and k_ba_solve read the reduced blocks
    FB_TRY(d_xb.alloc((size_t)(sharded ? 4 : 2) * xb.stride * 8));
    FB_HIP(hipMemsetAsync(d_xb.p, 0, (size_t)(sharded ? 4 : 2) * xb.stride * 8, s0));
    xb.base = d_xb.as<double>();
    XBLay xr = xb;  // reduced
    if (sharded) xr.base = xb.base + (size_t)2 * xb.stride;
    const size_t linLds = olGlobal ? 0 : odomLds + (size_t)std::max(nO, 1) * 108 * 8;
    bool linGlobal = olGlobal != nullptr;
    fb::DevBuf d_olDev;
    if (!linGlobal && linLds > 150 * 1024) {  // the products' scratch does not fit next to the records: HBM records, serial products
      FB_TRY(d_olDev.alloc((size_t)std::max(nO, 1) * sizeof(OdomLin)));
      olGlobal = d_olDev.as<OdomLin>();
      linGlobal = true;
    }
    if (!linGlobal) FB_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(k_ba_lin_c<false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)linLds));
    const int linGrid = nLin256 + POSE_PARTS * np + 1;
    std::vector<double> hostScratch;
    int rcSlot = FB_OK;
    auto slot = [&]() {
      if (rcSlot != FB_OK) return;
      { fb::ProfScope pr(fb::P_BA_SCHUR, s0);
        schurC<<<nWg, SCHUR_THREADS, schurLds, s0>>>(D, lb2, ctl, d_Dinv.as<double>(), d_Spart.as<double>(), P6, NT, lmPerWg); }
      { fb::ProfScope pr(fb::P_BA_SOLVE, s0);
        k_ba_sumparts_c<<<(nS + SUMPARTS_ELEMS - 1) / SUMPARTS_ELEMS, 256, 0, s0>>>(ctl, d_Spart.as<double>(), nWg, nS, rows);
        if (sharded) rcSlot = X.sum_dev(d_Spart.as<double>(), (size_t)nS, s0, hostScratch);  // exchange 1: the Schur-reduced system
        k_ba_solve_c<<<1, SOLVE_C_THREADS, solveLds, s0>>>(lb2, ctl, d_Spart.as<double>(), P6, NT, d_xp.as<double>(), d_scal.as<double>() + 3, xr); }
      { fb::ProfScope pr(fb::P_BA_UPDATE, s0);
        k_ba_update_c<<<nUpdBlocks, LIN_THREADS, 0, s0>>>(D, lb2, st2, ctl, d_Dinv.as<double>(), d_xp.as<double>(), d_scale.as<double>(), rank == 0 ? 1 : 0); }
      { fb::ProfScope pr(fb::P_BA_LINEARIZE, s0);
        if (linGlobal) k_ba_lin_c<true><<<linGrid, 256, 0, s0>>>(D, st2, lb2, ctl, sched, P6, nLin256, nLin256, olGlobal, xb);
        else k_ba_lin_c<false><<<linGrid, 256, linLds, s0>>>(D, st2, lb2, ctl, sched, P6, nLin256, nLin256, nullptr, xb); }
      { fb::ProfScope pr(fb::P_BA_MISC, s0);
        if (sharded) {
          k_ba_prex<<<1, 256, 0, s0>>>(D, lb2, ctl, nLin256, nLin256, d_scale.as<double>(), nUpdBlocks, xb, d_abortp, rank, world);
          // exchange 2: both raw blocks -> the reduced blocks, one all-reduce
          if (rcSlot == FB_OK) rcSlot = X.sum_dev(xr.at(0), (size_t)2 * xb.stride, s0, hostScratch, xb.at(0));
          k_ba_control<true><<<1, 256, 0, s0>>>(D, lb2, ctl, sched, nLin256, nLin256, d_scale.as<double>(), nUpdBlocks, d_scal.as<double>() + 3, P6, xr, d_abortp, world);
        } else {
          k_ba_control<false><<<1, 256, 0, s0>>>(D, lb2, ctl, sched, nLin256, nLin256, d_scale.as<double>(), nUpdBlocks, d_scal.as<double>() + 3, P6, xb, d_abortp, world);
        } }
    };
    FB_TRY(d_flags.alloc(std::max(nE, 1)));
    FB_TRY(d_kfT.alloc((size_t)n_kf * 48));
    FB_HIP(hipMemcpyAsync(d_kfT.p, d_kfT0, (size_t)n_kf * 48, hipMemcpyDeviceToDevice, s0));
    FB_TRY(d_ptOut.alloc((size_t)std::max(npt, 1) * 12));
    if (sharded) FB_TRY(d_ex.alloc(((size_t)npt * 3 + nE + 1) * 8));
    bool abortSent = false;
    static const int one = 1;
    // a typical schedule takes one trial per iteration: its1 + its2 trials + the two opening linearisations
    int batch = anything ? sc.its1 + (sc.gate ? sc.its2 + 1 : 0) + 1 + 2 : 0;
    int rcLoop = FB_OK;
    lap("buffers ready");
    for (int round = 0; round < 64; round++) {
      for (int i = 0; i < batch; i++) slot();
      if (rcSlot != FB_OK) { rcLoop = rcSlot; break; }
      lap("slots enqueued");
      // the results of the state that is current now (final when the schedule has finished, which is the common case)
      if (nE > 0) k_ba_gate_final_c<<<(nE + 255) / 256, 256, 0, s0>>>(D, st2, ctl, d_flags.as<uint8_t>());
      k_ba_export_c<<<(n_kf + npt * 3 + 255) / 256, 256, 0, s0>>>(n_kf, npt, st2, ctl, d_fixed, d_kfT.as<float>(), d_ptOut.as<float>());
      if (sharded) {  // every rank returns the complete result
        const int nx = npt * 3 + nE;
        k_ba_final_pack<<<(nx + 255) / 256, 256, 0, s0>>>(npt, nE, rank, world, d_ptOut.as<float>(), d_flags.as<uint8_t>(), d_ex.as<double>());
        rcLoop = X.sum_dev(d_ex.as<double>(), (size_t)nx, s0, hostScratch);
        if (rcLoop != FB_OK) break;
        k_ba_final_unpack<<<(nx + 255) / 256, 256, 0, s0>>>(npt, nE, d_ex.as<double>(), d_ptOut.as<float>(), d_flags.as<uint8_t>());
      }
      if (hipGetLastError() != hipSuccess) { fb::set_error("fb_local_ba: kernel launch failed"); rcLoop = FB_ERR_HIP; break; }
      if (hipMemcpyAsync(hCtl, ctl, sizeof(BACtl), hipMemcpyDeviceToHost, s0) != hipSuccess || hipEventRecord(evDone, s0) != hipSuccess) {
        fb::set_error("fb_local_ba: control block read-back failed"); rcLoop = FB_ERR_HIP; break;
      }
      if (!A->stop_flag) {  // nothing to forward: block in the driver instead of holding a host core
        const hipError_t q = hipEventSynchronize(evDone);
        if (q != hipSuccess) { fb::set_error("fb_local_ba: %s", hipGetErrorString(q)); rcLoop = FB_ERR_HIP; }
      } else {
        for (;;) {  // wait; meanwhile forward pbStopFlag (the control kernel sees it at the end of the slot that is running)
          const hipError_t q = hipEventQuery(evDone);
          if (q == hipSuccess) break;
          if (q != hipErrorNotReady) { fb::set_error("fb_local_ba: %s", hipGetErrorString(q)); rcLoop = FB_ERR_HIP; break; }
          if (!abortSent && *A->stop_flag) {
            (void)hipMemcpyAsync(d_abortp, &one, sizeof(int), hipMemcpyHostToDevice, sAux);
            abortSent = true;
          }
          // a BA lasts milliseconds and the flag only has to reach the device before the running slot (~0.1 ms) ends:
          // poll every 20 us instead of spinning on the LocalMapping thread's core
          std::this_thread::sleep_for(std::chrono::microseconds(20));
        }
      }
      if (rcLoop != FB_OK || hCtl->phase == 2) break;
      batch = 6;
    }
    lap("schedule finished");
    if (rcLoop != FB_OK) return rcLoop;
    if (hCtl->phase != 2) { fb::set_error("fb_local_ba: the LM schedule did not finish"); return FB_ERR_HIP; }
    if (devIn) {
      if (hCtl->badArgs & 4) { fb::set_error("fb_local_ba_dev: a key frame with more than 32768 observations (use fb_local_ba)"); return FB_ERR_CAPACITY; }
      if (hCtl->badArgs) {
        fb::set_error(hCtl->badArgs & 1 ? "fb_local_ba_dev: observation index out of range" : "fb_local_ba_dev: duplicate (keyframe, point) observation");
        return FB_ERR_ARG;
      }
      if (sc.gate) {
        if (nF > 0) FB_HIP(hipMemcpyAsync(A->obs_outlier, d_flags.p, (size_t)nF, hipMemcpyDeviceToDevice, s0));
        if (nB > 0) FB_HIP(hipMemcpyAsync(A->bobs_outlier, d_flags.as<uint8_t>() + nF, (size_t)nB, hipMemcpyDeviceToDevice, s0));
      }
      FB_HIP(hipMemcpyAsync(A->kf_Tcw, d_kfT.p, (size_t)n_kf * 48, hipMemcpyDeviceToDevice, s0));
      if (n_mp > 0) FB_HIP(hipMemcpyAsync(A->mp_xw, d_ptOut.p, (size_t)n_mp * 12, hipMemcpyDeviceToDevice, s0));
      if (A->n_mpb > 0) FB_HIP(hipMemcpyAsync(A->mpb_xw, d_ptOut.as<float>() + (size_t)3 * n_mp, (size_t)A->n_mpb * 12, hipMemcpyDeviceToDevice, s0));
      FB_HIP(hipStreamSynchronize(s0));  // the scratch goes back to the pool when this function returns
      lap("results copied (device)");
      return FB_OK;
    }
    std::vector<uint8_t> flags(std::max(nE, 1));
    FB_TRY(d_flags.download(flags.data(), std::max(nE, 1)));
    std::vector<float> po((size_t)std::max(npt, 1) * 3);
    FB_TRY(d_ptOut.download(po.data(), (size_t)npt * 12));
    if (sc.gate) {  // the global BA classifies nothing
      for (int i = 0; i < nF; i++) A->obs_outlier[i] = flags[i];
      for (int i = 0; i < nB; i++) A->bobs_outlier[i] = flags[nF + i];
    }
    FB_TRY(d_kfT.download(A->kf_Tcw, (size_t)n_kf * 48));
    for (int i = 0; i < 3 * n_mp; i++) A->mp_xw[i] = po[i];
    for (int i = 0; i < 3 * A->n_mpb; i++) A->mpb_xw[i] = po[3 * n_mp + i];
    lap("results copied out");
    return FB_OK;
  }

  double lastScale = 0;  // sum x (lambda x + b) of the most recent k_ba_update (all ranks)
  bool lastOk = true;    // LDL^T status of the most recent k_ba_solve
  // one linearisation at state `si` into buffer `bi`; returns chi2 (and max diagonal when wanted)
  auto linearize = [&](int si, int bi, int robust, bool wantDiag, double *chi, double *maxDiag) -> int {
    if (P6 > 0) FB_HIP(hipMemsetAsync(lb[bi].Hpp, 0, (size_t)P6 * P6 * 8, s0));
    { fb::ProfScope pr(fb::P_BA_LINEARIZE, s0);
      if (nLinBlocks > 0) k_ba_linearize<<<nLinBlocks, LIN_THREADS, 0, s0>>>(D, st[si], lb[bi], robust);
      if (np > 0) k_ba_pose<<<np, POSE_THREADS, 0, s0>>>(D, st[si], lb[bi], robust, P6);
      if (olGlobal) k_ba_odom<true><<<1, 256, 0, s0>>>(D, st[si], lb[bi], P6, nLinBlocks, olGlobal);
      else k_ba_odom<false><<<1, 256, odomLds, s0>>>(D, st[si], lb[bi], P6, nLinBlocks, nullptr); }
    { fb::ProfScope pr(fb::P_BA_MISC, s0);
      k_ba_scalars<<<1, 256, 0, s0>>>(lb[bi].chiPart, nLinBlocks + 1, lb[bi].Hpp, P6, lb[bi].Hll, npt, d_scal.as<double>(), wantDiag ? 1 : 0); }
    double h[4];
    FB_HIP(hipMemcpy(h, d_scal.p, 32, hipMemcpyDeviceToHost));  // chi2, max diagonal, scale term + solver status of the last trial
    lastOk = h[3] != 0.0;
    if (sharded) {
      std::vector<double> ex((size_t)P6 * P6 + P6 + 2);
      if (P6 > 0) {
        FB_HIP(hipMemcpy(ex.data(), lb[bi].Hpp, (size_t)P6 * P6 * 8, hipMemcpyDeviceToHost));
        FB_HIP(hipMemcpy(ex.data() + (size_t)P6 * P6, lb[bi].bp, (size_t)P6 * 8, hipMemcpyDeviceToHost));
      }
      ex[(size_t)P6 * P6 + P6] = h[0];
      ex[(size_t)P6 * P6 + P6 + 1] = h[2];
      FB_TRY(reduce(ex.data(), (int)ex.size(), 0));
      if (P6 > 0) {
        FB_HIP(hipMemcpy(lb[bi].Hpp, ex.data(), (size_t)P6 * P6 * 8, hipMemcpyHostToDevice));
        FB_HIP(hipMemcpy(lb[bi].bp, ex.data() + (size_t)P6 * P6, (size_t)P6 * 8, hipMemcpyHostToDevice));
      }
      h[0] = ex[(size_t)P6 * P6 + P6];
      h[2] = ex[(size_t)P6 * P6 + P6 + 1];
      if (wantDiag) {  // the pose diagonals add up over the ranks: take the maximum on the REDUCED Hpp
        double hp = 0;
        for (int i = 0; i < P6; i++) hp = std::max(hp, std::fabs(ex[(size_t)i * P6 + i]));
        k_ba_scalars<<<1, 256, 0, s0>>>(lb[bi].chiPart, 0, nullptr, 0, lb[bi].Hll, npt, d_scal.as<double>(), 1);
        double hl[2];
        FB_HIP(hipMemcpy(hl, d_scal.p, 16, hipMemcpyDeviceToHost));
        h[1] = std::max(hp, hl[1]);
        FB_TRY(reduce(&h[1], 1, 1));
      }
    }
    *chi = h[0];
    if (wantDiag) *maxDiag = h[1];
    lastScale = h[2];
    return FB_OK;
  };
  // SparseOptimizer::optimize + OptimizationAlgorithmLevenberg::solve, host-driven
  int cur = 0;  // index of the accepted state / its linearisation
  const bool trace = getenv("FB_BA_TRACE") != nullptr;
  auto optimize = [&](int iterations, int robust) -> int {
    double currentChi = 0, maxDiag = 0;
    FB_TRY(linearize(cur, cur, robust, true, &currentChi, &maxDiag));
    if (trace) fprintf(stderr, "[hip] optimize(%d) chi0=%.17g maxDiag=%.17g\n", iterations, currentChi, maxDiag);
    double lambda = 0, ni = 2;
    int nBad = 0;
    for (int it = 0; it < iterations; it++) {
      if (stopped()) break;  // terminate()
      const double iniChi = currentChi;
      if (it == 0) { lambda = 1e-5 * maxDiag; ni = 2; nBad = 0; }
      double rho = 0;
      int qmax = 0;
      do {
        const int tr = 1 - cur;
        if (big) {
          double *S = d_bigS.as<double>(), *rS = S + (size_t)P6 * P6, *M = d_bigM.as<double>(), *rhs = M + (size_t)P6 * P6;
          { fb::ProfScope pr(fb::P_BA_SCHUR, s0);
            FB_HIP(hipMemsetAsync(S, 0, (size_t)P6 * P6 * 8 + (size_t)P6 * 8, s0));  // blocks without a shared landmark stay 0
            if (npt > 0) k_ba_dinv<<<(npt + 255) / 256, 256, 0, s0>>>(D, lb[cur], lambda, d_Dinv.as<double>());
            if (lists.nBlocks > 0) k_ba_schur_gather<<<(lists.nBlocks + 3) / 4, 256, 0, s0>>>(D, lb[cur], d_Dinv.as<double>(), lists, S, P6);
            if (np > 0) k_ba_rhs_gather<<<(np + 3) / 4, 256, 0, s0>>>(D, lb[cur], d_Dinv.as<double>(), rS); }
          { fb::ProfScope pr(fb::P_BA_SOLVE, s0);
            if (sharded) {  // exchange step 1: the Schur-reduced system and right-hand side
              std::vector<double> ex((size_t)P6 * P6 + P6);
              FB_HIP(hipMemcpy(ex.data(), S, ex.size() * 8, hipMemcpyDeviceToHost));
              FB_TRY(reduce(ex.data(), (int)ex.size(), 0));
              FB_HIP(hipMemcpy(S, ex.data(), ex.size() * 8, hipMemcpyHostToDevice));
            }
            const long long nel = (long long)P6 * P6;
            k_big_assemble<<<(unsigned)((nel + 255) / 256), 256, 0, s0>>>(lb[cur], lambda, S, rS, M, rhs, P6, d_scal.as<double>() + 3);
            for (int k0 = 0; k0 < P6; k0 += BIG_NB) {
              const int kw = std::min(BIG_NB, P6 - k0);
              k_big_panel<<<1, 256, 0, s0>>>(M, P6, k0, kw, d_bigU.as<double>(), d_scal.as<double>() + 3);
              const int nbt = (P6 - k0 - kw + BIG_NB - 1) / BIG_NB;
              if (nbt > 0) k_big_update<<<nbt * (nbt + 1) / 2, 256, 0, s0>>>(M, P6, k0, kw, d_bigU.as<double>());
            }
            k_big_solve<<<1, 256, (size_t)P6 * 8, s0>>>(M, rhs, P6, d_xp.as<double>()); }
        } else {
        { fb::ProfScope pr(fb::P_BA_SCHUR, s0);
          schurKernel<<<nWg, SCHUR_THREADS, schurLds, s0>>>(D, lb[cur], lambda, d_Dinv.as<double>(), d_Spart.as<double>(), P6, NT, lmPerWg); }
        { fb::ProfScope pr(fb::P_BA_SOLVE, s0);
          // the workgroup partials are summed by a full-width kernel (one workgroup reading nWg x rows^2 doubles is slow)
          const int nS = rows * rows;
          k_ba_sumparts<<<(nS + SUMPARTS_ELEMS - 1) / SUMPARTS_ELEMS, 256, 0, s0>>>(d_Spart.as<double>(), nWg, nS);
          const int nParts = 1;
          if (sharded) {  // exchange step 1: the Schur-reduced system
            std::vector<double> ex(nS);
            FB_HIP(hipMemcpy(ex.data(), d_Spart.p, (size_t)nS * 8, hipMemcpyDeviceToHost));
            FB_TRY(reduce(ex.data(), nS, 0));
            FB_HIP(hipMemcpy(d_Spart.p, ex.data(), (size_t)nS * 8, hipMemcpyHostToDevice));
          }
          k_ba_solve<<<1, SOLVE_THREADS, solveLds, s0>>>(lb[cur], lambda, d_Spart.as<double>(), nParts, P6, NT, d_xp.as<double>(), d_scal.as<double>() + 3); }
        }
        { fb::ProfScope pr(fb::P_BA_UPDATE, s0);
          k_ba_update<<<nUpdBlocks, LIN_THREADS, 0, s0>>>(D, lb[cur], st[cur], st[tr], d_Dinv.as<double>(), d_xp.as<double>(), lambda, d_scale.as<double>(), rank == 0 ? 1 : 0);
          k_ba_scalars<<<1, 256, 0, s0>>>(d_scale.as<double>(), nUpdBlocks, nullptr, 0, nullptr, 0, d_scal.as<double>() + 2, 0); }
        double tempChi = 0, dummy = 0;
        FB_TRY(linearize(tr, tr, robust, false, &tempChi, &dummy));
        double hs[1] = {lastScale};  // exchange step 2 happened inside linearize()
        const int ok2 = lastOk ? 1 : 0;
        if (!ok2) tempChi = 1.7976931348623157e308;
        rho = currentChi - tempChi;
        const double scale = hs[0] + 1e-3;
        rho /= scale;
        if (trace) fprintf(stderr, "[hip]  it=%d q=%d lambda=%.17g tempChi=%.17g scale=%.17g rho=%.17g ok=%d\n", it, qmax, lambda, tempChi, scale, rho, ok2);
        if (rho > 0 && std::isfinite(tempChi)) {
          double alpha = 1. - pow((2 * rho - 1), 3);
          alpha = std::min(alpha, 2. / 3.);
          lambda *= std::max(1. / 3., alpha);
          ni = 2;
          currentChi = tempChi;
          cur = tr;  // discardTop: the trial state and its linearisation become current
        } else {
          lambda *= ni;
          ni *= 2;  // pop: keep `cur`
        }
        qmax++;
      } while (rho < 0 && qmax < 10 && !stopped());
      if (qmax == 10 || rho == 0) break;
      if ((iniChi - currentChi) * 1e3 < iniChi) nBad++;
      else nBad = 0;
      if (nBad >= 3) break;
    }
    return FB_OK;
  };
  if (nE + nO > 0 && (np > 0 || npt > 0)) FB_TRY(optimize(sc.its1, sc.robust1));
  const bool more = sc.gate && !stopped();
  if (more && nE > 0) {
    k_ba_gate<<<(nE + 255) / 256, 256, 0, s0>>>(D, st[cur], 1, nullptr);
    if (nE + nO > 0) FB_TRY(optimize(sc.its2, 0));
  }
  fb::DevBuf d_flags, d_kfT, d_ptOut;
  FB_TRY(d_flags.alloc(std::max(nE, 1)));
  if (nE > 0) k_ba_gate<<<(nE + 255) / 256, 256, 0, s0>>>(D, st[cur], 0, d_flags.as<uint8_t>());
  FB_TRY(d_kfT.upload(A->kf_Tcw, (size_t)n_kf * 48));
  FB_TRY(d_ptOut.alloc((size_t)std::max(npt, 1) * 12));
  k_ba_export<<<(n_kf + npt * 3 + 255) / 256, 256, 0, s0>>>(n_kf, npt, st[cur].pose, st[cur].pt, d_fixed,
                                                            d_kfT.as<float>(), d_ptOut.as<float>());
  FB_HIP(hipGetLastError());
  FB_HIP(hipDeviceSynchronize());
  std::vector<uint8_t> flags(std::max(nE, 1));
  FB_TRY(d_flags.download(flags.data(), std::max(nE, 1)));
  std::vector<float> po((size_t)std::max(npt, 1) * 3);
  FB_TRY(d_ptOut.download(po.data(), (size_t)npt * 12));
  if (sharded) {  // every rank returns the complete result: owned landmarks / edges are summed with zeros
    std::vector<double> ex((size_t)npt * 3 + nE);
    for (int l = 0; l < npt; l++)
      for (int c = 0; c < 3; c++) ex[(size_t)3 * l + c] = (l % world == rank) ? (double)po[(size_t)3 * l + c] : 0.0;
    for (int e = 0; e < nE; e++) ex[(size_t)npt * 3 + e] = flags[e];
    FB_TRY(reduce(ex.data(), (int)ex.size(), 0));
    for (size_t i = 0; i < (size_t)npt * 3; i++) po[i] = (float)ex[i];
    for (int e = 0; e < nE; e++) flags[e] = ex[(size_t)npt * 3 + e] != 0.0;
  }
  if (sc.gate) {  // the global BA classifies nothing
    for (int i = 0; i < nF; i++) A->obs_outlier[i] = flags[i];
    for (int i = 0; i < nB; i++) A->bobs_outlier[i] = flags[nF + i];
  }
  FB_TRY(d_kfT.download(A->kf_Tcw, (size_t)n_kf * 48));
  for (int i = 0; i < 3 * n_mp; i++) A->mp_xw[i] = po[i];
  for (int i = 0; i < 3 * A->n_mpb; i++) A->mpb_xw[i] = po[3 * n_mp + i];
  return FB_OK;
}
