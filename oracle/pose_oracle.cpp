/*
 * pose_oracle.cpp -- CPU restatement of Optimizer::PoseOptimization,
 * PoseOptimizationWithBird and BirdOptimization (TEST INFRASTRUCTURE ONLY).
 *
 * Follows /root/reference/src/Optimizer.cc:246-475 (front), :478-705 (front+bird),
 * :708-835 (bird), the edges in Thirdparty/g2o/g2o/types/types_six_dof_expmap.cpp:266-296
 * and src/OdomG2oTypeQuat.cc:61-70, and the g2o machinery restated in se3_oracle.h.
 * Edges are visited in insertion order (front, then bird) as g2o does after
 * sortVectorContainers (sparse_optimizer.cpp:166-190,482-487).
 */
#include <cmath>
#include <cstdint>
#include <cstring>
#include <vector>

#include "../include/fishbird.h"
#include "se3_oracle.h"

namespace {
using namespace orc;

struct Edge {
  int dim;          // 2 = EdgeSE3ProjectXYZOnlyPose, 3 = EdgeSE3ProjectBirdPoint2CamXYZ
  double Xw[3];
  double meas[3];   // obs (u,v) or Xc
  double info;      // information = info * I
  int slot = 0;     // index of the frame slot (keypoint) this edge came from
  int level = 0;
  bool robust = true;
  double err[3] = {0, 0, 0};  // _error, as last computed
};

struct PoseProblem : LMProblem {
  double fx, fy, cx, cy;
  double delta;  // (double)(float)sqrt(5.991)
  std::vector<Edge> edges;
  std::vector<int> active;
  SE3 T;
  std::vector<SE3> stack;
  double H[36], b[6], x[6];

  void computeError(Edge &e) const {
    double p[3];
    se3_map(T, e.Xw, p);
    if (e.dim == 2) {  // types_six_dof_expmap.h:136-140, .cpp:293-299
      const double invz = 1. / p[2];  // project2d: v/v(2) done as division below
      (void)invz;
      e.err[0] = e.meas[0] - ((p[0] / p[2]) * fx + cx);
      e.err[1] = e.meas[1] - ((p[1] / p[2]) * fy + cy);
      e.err[2] = 0;
    } else {  // OdomG2oTypeQuat.h:97-102
      for (int i = 0; i < 3; i++) e.err[i] = e.meas[i] - p[i];
    }
  }
  static double chi2(const Edge &e) {  // base_edge.h:58-61, information = info*I
    double s = 0;
    for (int i = 0; i < e.dim; i++) s += e.err[i] * (e.info * e.err[i]);
    return s;
  }
  void initialize() {  // initializeOptimization(0)
    active.clear();
    for (size_t i = 0; i < edges.size(); i++)
      if (edges[i].level == 0) active.push_back((int)i);
  }
  bool hasActive() override { return !active.empty(); }
  void computeActiveErrors() override {
    for (int i : active) computeError(edges[i]);
  }
  double activeRobustChi2() override {
    double chi = 0;
    for (int i : active) {
      const Edge &e = edges[i];
      if (e.robust) {
        double rho[2];
        huber(chi2(e), delta, rho);
        chi += rho[0];
      } else chi += chi2(e);
    }
    return chi;
  }
  void buildSystem() override {
    for (int i = 0; i < 36; i++) H[i] = 0;
    for (int i = 0; i < 6; i++) b[i] = 0;
    for (int ei : active) {
      const Edge &e = edges[ei];
      double p[3], J[18];
      se3_map(T, e.Xw, p);
      if (e.dim == 2) {  // EdgeSE3ProjectXYZOnlyPose::linearizeOplus
        const double X = p[0], Y = p[1], invz = 1.0 / p[2], invz_2 = invz * invz;
        J[0] = X * Y * invz_2 * fx;
        J[1] = -(1 + (X * X * invz_2)) * fx;
        J[2] = Y * invz * fx;
        J[3] = -invz * fx;
        J[4] = 0;
        J[5] = X * invz_2 * fx;
        J[6] = (1 + Y * Y * invz_2) * fy;
        J[7] = -X * Y * invz_2 * fy;
        J[8] = -X * invz * fy;
        J[9] = 0;
        J[10] = -invz * fy;
        J[11] = Y * invz_2 * fy;
      } else {  // EdgeSE3ProjectBirdPoint2CamXYZ::linearizeOplus: -[-skew(p), I]
        double S[9];
        skew(p, S);
        for (int r = 0; r < 3; r++)
          for (int c = 0; c < 3; c++) {
            J[r * 6 + c] = S[r * 3 + c];
            J[r * 6 + 3 + c] = (r == c) ? -1.0 : -0.0;
          }
      }
      double rho1 = 1.;
      if (e.robust) {
        double rho[2];
        huber(chi2(e), delta, rho);
        rho1 = rho[1];
      }
      // base_unary_edge.hpp:43-72: b -= rho1 * A^T * omega * e ; H += A^T * (rho1*omega) * A
      const double w = rho1 * e.info;
      for (int i = 0; i < 6; i++) {
        double s = 0;
        for (int r = 0; r < e.dim; r++) s += J[r * 6 + i] * (e.info * e.err[r]);
        b[i] -= rho1 * s;
        for (int j = 0; j < 6; j++) {
          double h = 0;
          for (int r = 0; r < e.dim; r++) h += J[r * 6 + i] * w * J[r * 6 + j];
          H[i * 6 + j] += h;
        }
      }
    }
  }
  double maxDiagonal() override {
    double m = 0;
    for (int j = 0; j < 6; j++) m = std::max(std::fabs(H[j * 6 + j]), m);
    return m;
  }
  bool solve(double lambda) override {
    std::vector<double> A(H, H + 36);
    for (int j = 0; j < 6; j++) A[j * 6 + j] += lambda;
    return ldlt_solve(A, 6, b, x);
  }
  double scaleTerm(double lambda) override {
    double s = 0;
    for (int j = 0; j < 6; j++) s += x[j] * (lambda * x[j] + b[j]);
    return s;
  }
  void push() override { stack.push_back(T); }
  void pop() override { T = stack.back(); stack.pop_back(); }
  void discardTop() override { stack.pop_back(); }
  void applyUpdate() override { T = se3_mul(se3_exp(x), T); }  // VertexSE3Expmap::oplusImpl
};

// Test instrumentation: the smallest relative distance |chi2 - threshold| / threshold over every inlier / outlier
// decision taken since the last reset (float values, as compared at Optimizer.cc:410,645,672).  It is the margin a
// differently rounded implementation has before a mask can flip.
double g_min_margin = 1e300;
long g_decisions = 0;
inline void note_margin(float chi2, double thr) {
  const double m = std::fabs((double)chi2 - thr) / thr;
  if (m < g_min_margin) g_min_margin = m;
  g_decisions++;
}

int pose_opt_one(const fb_pose_opt_args *A, int bidx) {
  const int mode = A->mode;
  const size_t fo = (size_t)bidx * A->front_stride, bo = (size_t)bidx * A->bird_stride;
  const int nfs = (mode != FB_POSE_BIRD) ? A->n_front[bidx] : 0;  // slots
  const int nbs = (mode != FB_POSE_FRONT) ? A->n_bird[bidx] : 0;
  int nf = 0, nb = 0;                                             // edges
  float *Tcw = A->Tcw + (size_t)bidx * 12;
  PoseProblem P;
  P.fx = A->fx; P.fy = A->fy; P.cx = A->cx; P.cy = A->cy;
  const float deltaMono = (float)std::sqrt(5.991);  // Optimizer.cc:280,514,733
  P.delta = deltaMono;
  for (int i = 0; i < nfs; i++) {
    if (A->front_valid && !A->front_valid[fo + i]) continue;
    Edge e;
    e.dim = 2;
    e.slot = i;
    nf++;
    for (int k = 0; k < 3; k++) e.Xw[k] = A->front_xw[(fo + i) * 3 + k];
    e.meas[0] = A->front_obs[(fo + i) * 2];
    e.meas[1] = A->front_obs[(fo + i) * 2 + 1];
    e.meas[2] = 0;
    const float invSigma2 = A->front_inv_sigma2[fo + i];
    e.info = (mode == FB_POSE_FRONT) ? (double)invSigma2 : (1.0 * (double)invSigma2) * (double)A->wF;  // :303, :542
    P.edges.push_back(e);
    A->front_outlier[fo + i] = 0;  // :298, :531
  }
  for (int i = 0; i < nbs; i++) {
    if (A->bird_valid && !A->bird_valid[bo + i]) continue;
    Edge e;
    e.dim = 3;
    e.slot = i;
    nb++;
    for (int k = 0; k < 3; k++) { e.Xw[k] = A->bird_xw[(bo + i) * 3 + k]; e.meas[k] = A->bird_xc[(bo + i) * 3 + k]; }
    e.info = (1.0 * (double)A->bird_inv_sigma2[bo + i]) * (double)A->wB;  // :588, :756
    P.edges.push_back(e);
  }
  A->ninliers[bidx] = 0;
  if (mode == FB_POSE_BIRD) { if (nb < 3) return FB_OK; }  // :776
  else if (nf < 3) return FB_OK;                           // :379, :607
  const float chi2Mono = (mode == FB_POSE_FRONT) ? 5.991f : 1.5f;  // :384, :611
  const float chi2Bird = 5.991f;                                   // :612, :781
  int nBad = 0, nBadBird = 0;
  const SE3 T0 = se3_from_float12(Tcw);
  for (int it = 0; it < 4; it++) {
    P.T = T0;           // vSE3->setEstimate(toSE3Quat(pFrame->mTcw))
    P.initialize();     // initializeOptimization(0)
    lm_optimize(P, 10);
    nBad = 0;
    for (int k = 0; k < nf; k++) {
      Edge &e = P.edges[k];
      const int i = e.slot;
      if (A->front_outlier[fo + i]) P.computeError(e);
      const float chi2 = (float)PoseProblem::chi2(e);
      bool bad;
      if (mode == FB_POSE_FRONT) bad = chi2 > chi2Mono;                         // :410-412
      else bad = chi2 > chi2Mono * ((double)A->wF + 1e-9);                       // :645
      note_margin(chi2, mode == FB_POSE_FRONT ? (double)chi2Mono : chi2Mono * ((double)A->wF + 1e-9));
      if (bad) { A->front_outlier[fo + i] = 1; e.level = 1; nBad++; }
      else { A->front_outlier[fo + i] = 0; e.level = 0; }
      if (it == 2) e.robust = false;
    }
    nBadBird = 0;
    for (int k = 0; k < nb; k++) {
      Edge &e = P.edges[nf + k];
      const int i = e.slot;
      if (A->bird_outlier[bo + i]) P.computeError(e);
      const float chi2 = (float)PoseProblem::chi2(e);
      const float chi2Bad = (float)(chi2Bird * ((double)A->wB + 1e-9));          // :672, :806
      note_margin(chi2, (double)chi2Bad);
      if (chi2 > chi2Bad) { A->bird_outlier[bo + i] = 1; e.level = 1; nBadBird++; }
      else { A->bird_outlier[bo + i] = 0; e.level = 0; }
      if (it == 2) e.robust = false;
    }
    if (P.edges.size() < 10) break;  // :462, :688, :824
  }
  se3_to_float12(P.T, Tcw);
  A->ninliers[bidx] = (mode == FB_POSE_BIRD) ? nb - nBadBird : nf - nBad;
  return FB_OK;
}

}  // namespace

extern "C" {

int orc_pose_opt(const fb_pose_opt_args *A) {
  for (int b = 0; b < A->batch; b++) {
    int rc = pose_opt_one(A, b);
    if (rc != FB_OK) return rc;
  }
  return FB_OK;
}

void orc_pose_margin_reset() { g_min_margin = 1e300; g_decisions = 0; }
int orc_pose_margin_get(double *min_margin, long *decisions) { *min_margin = g_min_margin; *decisions = g_decisions; return FB_OK; }

// known-answer hooks -------------------------------------------------------------
int orc_se3_exp(const double *u6, double *q4t3) {  // -> qx,qy,qz,qw,tx,ty,tz
  orc::SE3 T = orc::se3_exp(u6);
  q4t3[0] = T.r.x; q4t3[1] = T.r.y; q4t3[2] = T.r.z; q4t3[3] = T.r.w;
  for (int i = 0; i < 3; i++) q4t3[4 + i] = T.t[i];
  return FB_OK;
}
int orc_se3_log(const double *q4t3, double *u6) {
  orc::SE3 T;
  T.r.x = q4t3[0]; T.r.y = q4t3[1]; T.r.z = q4t3[2]; T.r.w = q4t3[3];
  for (int i = 0; i < 3; i++) T.t[i] = q4t3[4 + i];
  orc::se3_log(T, u6);
  return FB_OK;
}
int orc_huber(double e, double delta, double *rho2) {
  orc::huber(e, delta, rho2);
  return FB_OK;
}
// residual + analytic Jacobian of one pose-only edge at pose Tcw (float 3x4); dim 2 or 3
int orc_pose_edge(int dim, const float *Tcw, const double *Xw, const double *meas, const double *K4, double *err,
                  double *J) {
  PoseProblem P;
  P.fx = K4[0]; P.fy = K4[1]; P.cx = K4[2]; P.cy = K4[3];
  P.delta = 1e30;
  Edge e;
  e.dim = dim;
  for (int k = 0; k < 3; k++) { e.Xw[k] = Xw[k]; e.meas[k] = meas[k]; }
  e.info = 1;
  e.robust = false;
  P.edges.push_back(e);
  P.T = orc::se3_from_float12(Tcw);
  P.initialize();
  P.computeActiveErrors();
  P.buildSystem();
  for (int i = 0; i < dim; i++) err[i] = P.edges[0].err[i];
  // with info=1, no kernel: b = -J^T e, H = J^T J; return b and H for the caller to check
  for (int i = 0; i < 6; i++) J[i] = P.b[i];
  for (int i = 0; i < 36; i++) J[6 + i] = P.H[i];
  return FB_OK;
}

}  // extern "C"
