/*
 * fb_se3.h -- device-side double-precision SE3 (unit quaternion + translation), the
 * parameterisation g2o::SE3Quat uses.  Follows Thirdparty/g2o/g2o/types/se3quat.h:41-296
 * and se3_ops.hpp:27-47 of the reference; Eigen's Quaterniond(R) / toRotationMatrix / q*v
 * (not vendored there) are restated from their published algorithms.
 */
#ifndef FB_SE3_H_
#define FB_SE3_H_

#include <hip/hip_runtime.h>
#include <math.h>

namespace fb {

struct Quat { double x, y, z, w; };
struct SE3 { Quat r; double t[3]; };

__host__ __device__ inline void quat_normalize(Quat &q) {
  const double n = sqrt(q.x * q.x + q.y * q.y + q.z * q.z + q.w * q.w);
  q.x /= n; q.y /= n; q.z /= n; q.w /= n;
}

__host__ __device__ inline void se3_normalize_rotation(SE3 &T) {  // se3quat.h:280-285
  if (T.r.w < 0) { T.r.x = -T.r.x; T.r.y = -T.r.y; T.r.z = -T.r.z; T.r.w = -T.r.w; }
  quat_normalize(T.r);
}

__host__ __device__ inline Quat quat_from_R(const double R[9]) {  // Eigen Quaternion(Matrix3)
  Quat q;
  double t = R[0] + R[4] + R[8];
  if (t > 0) {
    t = sqrt(t + 1.0);
    q.w = 0.5 * t;
    t = 0.5 / t;
    q.x = (R[7] - R[5]) * t;
    q.y = (R[2] - R[6]) * t;
    q.z = (R[3] - R[1]) * t;
  } else {
    // i = index of the largest diagonal entry, (i,j,k) cyclic; written out per case so that no
    // array is indexed at run time (keeps everything in registers on the GPU)
    int i = 0;
    if (R[4] > R[0]) i = 1;
    if (R[8] > (i == 1 ? R[4] : R[0])) i = 2;
    if (i == 0) {
      t = sqrt(R[0] - R[4] - R[8] + 1.0);
      q.x = 0.5 * t;
      t = 0.5 / t;
      q.w = (R[7] - R[5]) * t;
      q.y = (R[3] + R[1]) * t;
      q.z = (R[6] + R[2]) * t;
    } else if (i == 1) {
      t = sqrt(R[4] - R[8] - R[0] + 1.0);
      q.y = 0.5 * t;
      t = 0.5 / t;
      q.w = (R[2] - R[6]) * t;
      q.z = (R[7] + R[5]) * t;
      q.x = (R[1] + R[3]) * t;
    } else {
      t = sqrt(R[8] - R[0] - R[4] + 1.0);
      q.z = 0.5 * t;
      t = 0.5 / t;
      q.w = (R[3] - R[1]) * t;
      q.x = (R[2] + R[6]) * t;
      q.y = (R[5] + R[7]) * t;
    }
  }
  return q;
}

__host__ __device__ inline void quat_to_R(const Quat &q, double R[9]) {  // Eigen toRotationMatrix
  const double tx = 2 * q.x, ty = 2 * q.y, tz = 2 * q.z;
  const double twx = tx * q.w, twy = ty * q.w, twz = tz * q.w;
  const double txx = tx * q.x, txy = ty * q.x, txz = tz * q.x;
  const double tyy = ty * q.y, tyz = tz * q.y, tzz = tz * q.z;
  R[0] = 1 - (tyy + tzz); R[1] = txy - twz;       R[2] = txz + twy;
  R[3] = txy + twz;       R[4] = 1 - (txx + tzz); R[5] = tyz - twx;
  R[6] = txz - twy;       R[7] = tyz + twx;       R[8] = 1 - (txx + tyy);
}

__host__ __device__ inline void quat_rotate(const Quat &q, const double v[3], double o[3]) {  // Eigen _transformVector
  double uv[3] = {q.y * v[2] - q.z * v[1], q.z * v[0] - q.x * v[2], q.x * v[1] - q.y * v[0]};
  uv[0] += uv[0]; uv[1] += uv[1]; uv[2] += uv[2];
  o[0] = v[0] + q.w * uv[0] + (q.y * uv[2] - q.z * uv[1]);
  o[1] = v[1] + q.w * uv[1] + (q.z * uv[0] - q.x * uv[2]);
  o[2] = v[2] + q.w * uv[2] + (q.x * uv[1] - q.y * uv[0]);
}

__host__ __device__ inline Quat quat_mul(const Quat &a, const Quat &b) {
  Quat r;
  r.w = a.w * b.w - a.x * b.x - a.y * b.y - a.z * b.z;
  r.x = a.w * b.x + a.x * b.w + a.y * b.z - a.z * b.y;
  r.y = a.w * b.y + a.y * b.w + a.z * b.x - a.x * b.z;
  r.z = a.w * b.z + a.z * b.w + a.x * b.y - a.y * b.x;
  return r;
}

__host__ __device__ inline void se3_map(const SE3 &T, const double x[3], double o[3]) {  // se3quat.h:217-220
  quat_rotate(T.r, x, o);
  o[0] += T.t[0]; o[1] += T.t[1]; o[2] += T.t[2];
}

__host__ __device__ inline SE3 se3_mul(const SE3 &a, const SE3 &b) {  // se3quat.h:104-110
  SE3 r = a;
  double rt[3];
  quat_rotate(a.r, b.t, rt);
  r.t[0] += rt[0]; r.t[1] += rt[1]; r.t[2] += rt[2];
  r.r = quat_mul(a.r, b.r);
  se3_normalize_rotation(r);
  return r;
}

__host__ __device__ inline SE3 se3_inverse(const SE3 &a) {  // se3quat.h:123-128
  SE3 r;
  r.r.x = -a.r.x; r.r.y = -a.r.y; r.r.z = -a.r.z; r.r.w = a.r.w;
  const double nt[3] = {a.t[0] * -1., a.t[1] * -1., a.t[2] * -1.};
  quat_rotate(r.r, nt, r.t);
  return r;
}

__host__ __device__ inline void skew3(const double v[3], double S[9]) {  // se3_ops.hpp:27-35
  S[0] = 0; S[1] = -v[2]; S[2] = v[1];
  S[3] = v[2]; S[4] = 0; S[5] = -v[0];
  S[6] = -v[1]; S[7] = v[0]; S[8] = 0;
}

__host__ __device__ inline void mat3_mul(const double A[9], const double B[9], double C[9]) {
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++) C[i * 3 + j] = A[i * 3] * B[j] + A[i * 3 + 1] * B[3 + j] + A[i * 3 + 2] * B[6 + j];
}

__host__ __device__ inline SE3 se3_exp(const double u[6]) {  // SE3Quat::exp, se3quat.h:223-257
  const double omega[3] = {u[0], u[1], u[2]}, ups[3] = {u[3], u[4], u[5]};
  const double theta = sqrt(omega[0] * omega[0] + omega[1] * omega[1] + omega[2] * omega[2]);
  double Om[9], Om2[9], R[9], V[9];
  skew3(omega, Om);
  mat3_mul(Om, Om, Om2);
  const double I[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
  if (theta < 0.00001) {
    for (int i = 0; i < 9; i++) { R[i] = I[i] + Om[i] + Om2[i]; V[i] = R[i]; }  // sic
  } else {
    double s, c;
    sincos(theta, &s, &c);
    const double it = 1.0 / theta, it2 = it * it;  // one division instead of three
    const double a = s * it, b = (1 - c) * it2, d = (theta - s) * (it2 * it);
    for (int i = 0; i < 9; i++) {
      R[i] = I[i] + a * Om[i] + b * Om2[i];
      V[i] = I[i] + b * Om[i] + d * Om2[i];
    }
  }
  SE3 T;
  T.r = quat_from_R(R);
  for (int i = 0; i < 3; i++) T.t[i] = V[i * 3] * ups[0] + V[i * 3 + 1] * ups[1] + V[i * 3 + 2] * ups[2];
  se3_normalize_rotation(T);
  return T;
}

__host__ __device__ inline void se3_log(const SE3 &T, double res[6]) {  // SE3Quat::log, se3quat.h:178-215
  double R[9];
  quat_to_R(T.r, R);
  const double d = 0.5 * (R[0] + R[4] + R[8] - 1);
  const double dR[3] = {R[7] - R[5], R[2] - R[6], R[3] - R[1]};
  double omega[3], Om[9], Om2[9], Vinv[9];
  const double I[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
  if (d > 0.99999) {
    for (int i = 0; i < 3; i++) omega[i] = 0.5 * dR[i];
    skew3(omega, Om);
    mat3_mul(Om, Om, Om2);
    for (int i = 0; i < 9; i++) Vinv[i] = I[i] - 0.5 * Om[i] + (1. / 12.) * Om2[i];
  } else {
    const double theta = acos(d);
    const double f = theta / (2 * sqrt(1 - d * d));
    for (int i = 0; i < 3; i++) omega[i] = f * dR[i];
    skew3(omega, Om);
    mat3_mul(Om, Om, Om2);
    const double g = (1 - theta / (2 * tan(theta / 2))) / (theta * theta);
    for (int i = 0; i < 9; i++) Vinv[i] = I[i] - 0.5 * Om[i] + g * Om2[i];
  }
  for (int i = 0; i < 3; i++) {
    res[i] = omega[i];
    res[i + 3] = Vinv[i * 3] * T.t[0] + Vinv[i * 3 + 1] * T.t[1] + Vinv[i * 3 + 2] * T.t[2];
  }
}

__host__ __device__ inline void se3_adj(const SE3 &T, double A[36]) {  // SE3Quat::adj, se3quat.h:259-268
  double R[9], S[9], SR[9];
  quat_to_R(T.r, R);
  skew3(T.t, S);
  mat3_mul(S, R, SR);
  for (int i = 0; i < 36; i++) A[i] = 0;
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++) {
      A[i * 6 + j] = R[i * 3 + j];
      A[(i + 3) * 6 + j + 3] = R[i * 3 + j];
      A[(i + 3) * 6 + j] = SR[i * 3 + j];
    }
}

__host__ __device__ inline SE3 se3_from_float12(const float *T) {  // Converter::toSE3Quat, Converter.cc:38-48
  const double R[9] = {T[0], T[1], T[2], T[4], T[5], T[6], T[8], T[9], T[10]};
  SE3 s;
  s.r = quat_from_R(R);
  s.t[0] = T[3]; s.t[1] = T[7]; s.t[2] = T[11];
  se3_normalize_rotation(s);
  return s;
}

__host__ __device__ inline void se3_to_float12(const SE3 &s, float *T) {  // Converter::toCvMat(SE3Quat), Converter.cc:50-72
  double R[9];
  quat_to_R(s.r, R);
  for (int i = 0; i < 3; i++) {
    for (int j = 0; j < 3; j++) T[i * 4 + j] = (float)R[i * 3 + j];
    T[i * 4 + 3] = (float)s.t[i];
  }
}

// RobustKernelHuber::robustify, robust_kernel_impl.cpp:78-91
__host__ __device__ inline void huber(double e, double delta, double &rho0, double &rho1) {
  const double dsqr = delta * delta;
  if (e <= dsqr) { rho0 = e; rho1 = 1.; }
  else {
    const double sqrte = sqrt(e);
    rho0 = 2 * sqrte * delta - dsqr;
    rho1 = delta / sqrte;
  }
}

// Symmetric 6x6 solve, LDL^T without pivoting; false on a negative pivot
// (LinearSolverDense, solvers/linear_solver_dense.h:65-113: Eigen LDLT::isPositive()).
__host__ __device__ inline bool ldlt6(const double H[36], double lambda, const double b[6], double x[6]) {
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
    // one reciprocal per pivot: this routine runs on ONE lane inside every LM trial of k_pose_opt, and an fp64 division is
    // ~35 dependent instructions (21 of them were a sixth of a trial); s * (1/dj) is within one ulp of s / dj
    const double inv = dj != 0 ? 1.0 / dj : 0.0;
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

}  // namespace fb
#endif
