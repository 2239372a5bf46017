/*
 * se3_oracle.h -- plain-double restatement of g2o::SE3Quat and the Levenberg-Marquardt
 * driver (TEST INFRASTRUCTURE ONLY; see orb_oracle.cpp header).
 *
 * Follows /root/reference/Thirdparty/g2o/g2o/types/se3quat.h:41-296, se3_ops.hpp:27-47,
 * core/optimization_algorithm_levenberg.cpp:61-189, core/sparse_optimizer.cpp:354-419,
 * core/robust_kernel_impl.cpp:78-91 and src/Converter.cc:38-72.
 * Eigen (Quaterniond(R), toRotationMatrix, q*v, LDLT) is not vendored in the reference:
 * its published algorithms are restated -> "parity unpinned" at that boundary; any
 * correct variant agrees to ~1e-15, far inside the 1e-4 pose tolerance.
 */
#ifndef SE3_ORACLE_H_
#define SE3_ORACLE_H_

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <limits>
#include <vector>

namespace orc {

struct Quat { double x = 0, y = 0, z = 0, w = 1; };

struct SE3 {
  Quat r;
  double t[3] = {0, 0, 0};
};

inline void quat_normalize(Quat &q) {
  double n = std::sqrt(q.x * q.x + q.y * q.y + q.z * q.z + q.w * q.w);
  q.x /= n; q.y /= n; q.z /= n; q.w /= n;
}

inline void se3_normalize_rotation(SE3 &T) {  // se3quat.h:280-285
  if (T.r.w < 0) { T.r.x = -T.r.x; T.r.y = -T.r.y; T.r.z = -T.r.z; T.r.w = -T.r.w; }
  quat_normalize(T.r);
}

// Eigen::Quaterniond(Matrix3d) (Shepperd-style; Eigen/src/Geometry/Quaternion.h quaternionbase_assign_impl)
inline Quat quat_from_R(const double R[9]) {
  Quat q;
  double t = R[0] + R[4] + R[8];
  if (t > 0) {
    t = std::sqrt(t + 1.0);
    q.w = 0.5 * t;
    t = 0.5 / t;
    q.x = (R[7] - R[5]) * t;
    q.y = (R[2] - R[6]) * t;
    q.z = (R[3] - R[1]) * t;
  } else {
    int i = 0;
    if (R[4] > R[0]) i = 1;
    if (R[8] > R[i * 3 + i]) i = 2;
    int j = (i + 1) % 3, k = (j + 1) % 3;
    t = std::sqrt(R[i * 3 + i] - R[j * 3 + j] - R[k * 3 + k] + 1.0);
    double v[3];
    v[i] = 0.5 * t;
    t = 0.5 / t;
    q.w = (R[k * 3 + j] - R[j * 3 + k]) * t;
    v[j] = (R[j * 3 + i] + R[i * 3 + j]) * t;
    v[k] = (R[k * 3 + i] + R[i * 3 + k]) * t;
    q.x = v[0]; q.y = v[1]; q.z = v[2];
  }
  return q;
}

// Eigen QuaternionBase::toRotationMatrix
inline void quat_to_R(const Quat &q, double R[9]) {
  const double tx = 2 * q.x, ty = 2 * q.y, tz = 2 * q.z;
  const double twx = tx * q.w, twy = ty * q.w, twz = tz * q.w;
  const double txx = tx * q.x, txy = ty * q.x, txz = tz * q.x;
  const double tyy = ty * q.y, tyz = tz * q.y, tzz = tz * q.z;
  R[0] = 1 - (tyy + tzz); R[1] = txy - twz;       R[2] = txz + twy;
  R[3] = txy + twz;       R[4] = 1 - (txx + tzz); R[5] = tyz - twx;
  R[6] = txz - twy;       R[7] = tyz + twx;       R[8] = 1 - (txx + tyy);
}

// Eigen QuaternionBase::_transformVector: v + w*uv + q x uv, uv = 2 q x v
inline void quat_rotate(const Quat &q, const double v[3], double o[3]) {
  double uv[3] = {q.y * v[2] - q.z * v[1], q.z * v[0] - q.x * v[2], q.x * v[1] - q.y * v[0]};
  uv[0] += uv[0]; uv[1] += uv[1]; uv[2] += uv[2];
  o[0] = v[0] + q.w * uv[0] + (q.y * uv[2] - q.z * uv[1]);
  o[1] = v[1] + q.w * uv[1] + (q.z * uv[0] - q.x * uv[2]);
  o[2] = v[2] + q.w * uv[2] + (q.x * uv[1] - q.y * uv[0]);
}

inline Quat quat_mul(const Quat &a, const Quat &b) {
  Quat r;
  r.w = a.w * b.w - a.x * b.x - a.y * b.y - a.z * b.z;
  r.x = a.w * b.x + a.x * b.w + a.y * b.z - a.z * b.y;
  r.y = a.w * b.y + a.y * b.w + a.z * b.x - a.x * b.z;
  r.z = a.w * b.z + a.z * b.w + a.x * b.y - a.y * b.x;
  return r;
}

inline void se3_map(const SE3 &T, const double x[3], double o[3]) {  // se3quat.h:217-220
  quat_rotate(T.r, x, o);
  o[0] += T.t[0]; o[1] += T.t[1]; o[2] += T.t[2];
}

inline SE3 se3_mul(const SE3 &a, const SE3 &b) {  // se3quat.h:104-110
  SE3 r = a;
  double rt[3];
  quat_rotate(a.r, b.t, rt);
  r.t[0] += rt[0]; r.t[1] += rt[1]; r.t[2] += rt[2];
  r.r = quat_mul(a.r, b.r);
  se3_normalize_rotation(r);
  return r;
}

inline SE3 se3_inverse(const SE3 &a) {  // se3quat.h:123-128
  SE3 r;
  r.r.x = -a.r.x; r.r.y = -a.r.y; r.r.z = -a.r.z; r.r.w = a.r.w;
  double nt[3] = {a.t[0] * -1., a.t[1] * -1., a.t[2] * -1.};
  quat_rotate(r.r, nt, r.t);
  return r;
}

inline void skew(const double v[3], double S[9]) {  // se3_ops.hpp:27-35
  S[0] = 0; S[1] = -v[2]; S[2] = v[1];
  S[3] = v[2]; S[4] = 0; S[5] = -v[0];
  S[6] = -v[1]; S[7] = v[0]; S[8] = 0;
}

inline void mat3_mul(const double A[9], const double B[9], double C[9]) {
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++) C[i * 3 + j] = A[i * 3] * B[j] + A[i * 3 + 1] * B[3 + j] + A[i * 3 + 2] * B[6 + j];
}

// SE3Quat::exp, se3quat.h:223-257 (update = [omega, upsilon])
inline SE3 se3_exp(const double u[6]) {
  const double omega[3] = {u[0], u[1], u[2]}, ups[3] = {u[3], u[4], u[5]};
  const double theta = std::sqrt(omega[0] * omega[0] + omega[1] * omega[1] + omega[2] * omega[2]);
  double Om[9], Om2[9], R[9], V[9];
  skew(omega, Om);
  mat3_mul(Om, Om, Om2);
  const double I[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
  if (theta < 0.00001) {
    for (int i = 0; i < 9; i++) { R[i] = I[i] + Om[i] + Om2[i]; V[i] = R[i]; }  // sic, no 1/2
  } else {
    const double s = std::sin(theta), c = std::cos(theta);
    const double a = s / theta, b = (1 - c) / (theta * theta), d = (theta - s) / (theta * theta * theta);
    for (int i = 0; i < 9; i++) {
      R[i] = I[i] + a * Om[i] + b * Om2[i];
      V[i] = I[i] + b * Om[i] + d * Om2[i];
    }
  }
  SE3 T;
  T.r = quat_from_R(R);
  for (int i = 0; i < 3; i++) T.t[i] = V[i * 3] * ups[0] + V[i * 3 + 1] * ups[1] + V[i * 3 + 2] * ups[2];
  se3_normalize_rotation(T);  // SE3Quat(Quaterniond, Vector3d) ctor, se3quat.h:62-64
  return T;
}

// SE3Quat::log, se3quat.h:178-215  -> [omega, upsilon]
inline void se3_log(const SE3 &T, double res[6]) {
  double R[9];
  quat_to_R(T.r, R);
  const double d = 0.5 * (R[0] + R[4] + R[8] - 1);
  const double dR[3] = {R[7] - R[5], R[2] - R[6], R[3] - R[1]};  // deltaR, se3_ops.hpp:37-44
  double omega[3], Om[9], Om2[9], Vinv[9];
  const double I[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
  if (d > 0.99999) {
    for (int i = 0; i < 3; i++) omega[i] = 0.5 * dR[i];
    skew(omega, Om);
    mat3_mul(Om, Om, Om2);
    for (int i = 0; i < 9; i++) Vinv[i] = I[i] - 0.5 * Om[i] + (1. / 12.) * Om2[i];
  } else {
    const double theta = std::acos(d);
    const double f = theta / (2 * std::sqrt(1 - d * d));
    for (int i = 0; i < 3; i++) omega[i] = f * dR[i];
    skew(omega, Om);
    mat3_mul(Om, Om, Om2);
    const double g = (1 - theta / (2 * std::tan(theta / 2))) / (theta * theta);
    for (int i = 0; i < 9; i++) Vinv[i] = I[i] - 0.5 * Om[i] + g * Om2[i];
  }
  for (int i = 0; i < 3; i++) {
    res[i] = omega[i];
    res[i + 3] = Vinv[i * 3] * T.t[0] + Vinv[i * 3 + 1] * T.t[1] + Vinv[i * 3 + 2] * T.t[2];
  }
}

// SE3Quat::adj, se3quat.h:259-268 (6x6 row-major)
inline void se3_adj(const SE3 &T, double A[36]) {
  double R[9], S[9], SR[9];
  quat_to_R(T.r, R);
  skew(T.t, S);
  mat3_mul(S, R, SR);
  for (int i = 0; i < 36; i++) A[i] = 0;
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++) {
      A[i * 6 + j] = R[i * 3 + j];
      A[(i + 3) * 6 + j + 3] = R[i * 3 + j];
      A[(i + 3) * 6 + j] = SR[i * 3 + j];
    }
}

// Converter::toSE3Quat, Converter.cc:38-48 (float 3x4 row-major -> double, normalised quaternion)
inline SE3 se3_from_float12(const float *T) {
  double R[9] = {T[0], T[1], T[2], T[4], T[5], T[6], T[8], T[9], T[10]};
  SE3 s;
  s.r = quat_from_R(R);
  s.t[0] = T[3]; s.t[1] = T[7]; s.t[2] = T[11];
  se3_normalize_rotation(s);
  return s;
}

// Converter::toCvMat(SE3Quat), Converter.cc:50-54,64-72 (double -> float)
inline void se3_to_float12(const SE3 &s, float *T) {
  double R[9];
  quat_to_R(s.r, R);
  for (int i = 0; i < 3; i++) {
    for (int j = 0; j < 3; j++) T[i * 4 + j] = (float)R[i * 3 + j];
    T[i * 4 + 3] = (float)s.t[i];
  }
}

// RobustKernelHuber::robustify, robust_kernel_impl.cpp:78-91 (rho[2] unused, base_edge.h:96-102)
inline void huber(double e, double delta, double rho[2]) {
  const double dsqr = delta * delta;
  if (e <= dsqr) { rho[0] = e; rho[1] = 1.; }
  else {
    const double sqrte = std::sqrt(e);
    rho[0] = 2 * sqrte * delta - dsqr;
    rho[1] = delta / sqrte;
  }
}

// Dense symmetric solve restating LinearSolverDense (solvers/linear_solver_dense.h:65-113):
// LDL^T; returns false on a negative pivot (Eigen LDLT::isPositive() == false).
// simplicial = the failure rule of LinearSolverEigen (solvers/linear_solver_eigen.h:94-124: SimplicialLDLT::info() !=
// Success), which every bundle adjustment of the reference uses (Optimizer.cc:64,895,1799,2233): Eigen's LDL^T
// factorisation stops with NumericalIssue on a pivot that is exactly 0 and carries on through negative ones (Eigen is
// not vendored: restated from general knowledge of SimplicialCholesky_impl.h, PARITY UNPINNED).
inline bool ldlt_solve(std::vector<double> A, int n, const double *b, double *x, bool simplicial = false) {
  std::vector<double> d(n);
  for (int j = 0; j < n; j++) {
    double dj = A[j * n + j];
    for (int k = 0; k < j; k++) dj -= A[j * n + k] * A[j * n + k] * d[k];
    if (simplicial ? dj == 0 : dj < 0) return false;
    d[j] = dj;
    for (int i = j + 1; i < n; i++) {
      double s = A[i * n + j];
      for (int k = 0; k < j; k++) s -= A[i * n + k] * A[j * n + k] * d[k];
      A[i * n + j] = dj != 0 ? s / dj : 0;
    }
  }
  std::vector<double> y(n);
  for (int i = 0; i < n; i++) {
    double s = b[i];
    for (int k = 0; k < i; k++) s -= A[i * n + k] * y[k];
    y[i] = s;
  }
  for (int i = 0; i < n; i++) y[i] = d[i] != 0 ? y[i] / d[i] : 0;
  for (int i = n - 1; i >= 0; i--) {
    double s = y[i];
    for (int k = i + 1; k < n; k++) s -= A[k * n + i] * x[k];
    x[i] = s;
  }
  return true;
}

// ---- Levenberg-Marquardt driver ------------------------------------------------
struct LMProblem {
  virtual ~LMProblem() {}
  virtual void computeActiveErrors() = 0;     // sparse_optimizer.cpp:61-76
  virtual double activeRobustChi2() = 0;      // :100-114
  virtual void buildSystem() = 0;             // block_solver.hpp:502-560
  virtual double maxDiagonal() = 0;           // optimization_algorithm_levenberg.cpp:166-180
  virtual bool solve(double lambda) = 0;      // setLambda + solve + restoreDiagonal -> x
  virtual double scaleTerm(double lambda) = 0;  // :182-189  sum x_j (lambda x_j + b_j)
  virtual void push() = 0;
  virtual void pop() = 0;
  virtual void discardTop() = 0;
  virtual void applyUpdate() = 0;             // oplus(x)
  virtual bool terminate() { return false; }  // forceStopFlag
  virtual bool hasActive() = 0;
};

struct LMState {
  double lambda = -1, ni = 2;
  int nBad = 0;
};

// OptimizationAlgorithmLevenberg::solve, optimization_algorithm_levenberg.cpp:61-164.
// returns 0 = OK, 1 = Terminate
inline int lm_solve(LMProblem &P, LMState &S, int iteration) {
  P.computeActiveErrors();
  double currentChi = P.activeRobustChi2();
  double tempChi = currentChi;
  const double iniChi = currentChi;
  P.buildSystem();
  static const bool trace = std::getenv("FB_BA_TRACE") != nullptr;
  if (iteration == 0) {
    S.lambda = 1e-5 * P.maxDiagonal();
    S.ni = 2;
    S.nBad = 0;
    if (trace) std::fprintf(stderr, "[orc] optimize chi0=%.17g maxDiag=%.17g\n", currentChi, P.maxDiagonal());
  }
  double rho = 0;
  int qmax = 0;
  do {
    P.push();
    const bool ok2 = P.solve(S.lambda);
    P.applyUpdate();
    P.computeActiveErrors();
    tempChi = P.activeRobustChi2();
    if (!ok2) tempChi = std::numeric_limits<double>::max();
    rho = (currentChi - tempChi);
    double scale = P.scaleTerm(S.lambda);
    scale += 1e-3;
    rho /= scale;
    if (trace) std::fprintf(stderr, "[orc]  it=%d q=%d lambda=%.17g tempChi=%.17g scale=%.17g rho=%.17g ok=%d\n", iteration, qmax, S.lambda, tempChi, scale, rho, (int)ok2);
    if (rho > 0 && std::isfinite(tempChi)) {
      double alpha = 1. - std::pow((2 * rho - 1), 3);
      alpha = std::min(alpha, 2. / 3.);
      const double scaleFactor = std::max(1. / 3., alpha);
      S.lambda *= scaleFactor;
      S.ni = 2;
      currentChi = tempChi;
      P.discardTop();
    } else {
      S.lambda *= S.ni;
      S.ni *= 2;
      P.pop();
    }
    qmax++;
  } while (rho < 0 && qmax < 10 && !P.terminate());
  if (qmax == 10 || rho == 0) return 1;
  if ((iniChi - currentChi) * 1e3 < iniChi) S.nBad++;
  else S.nBad = 0;
  if (S.nBad >= 3) return 1;
  return 0;
}

// SparseOptimizer::optimize, sparse_optimizer.cpp:354-419
inline int lm_optimize(LMProblem &P, int iterations) {
  if (!P.hasActive()) return -1;
  LMState S;
  int cj = 0;
  bool ok = true;
  for (int i = 0; i < iterations && !P.terminate() && ok; i++) {
    ok = lm_solve(P, S, i) == 0;
    ++cj;
  }
  return cj;
}

}  // namespace orc
#endif
