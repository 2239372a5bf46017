/*
 * ba_oracle.cpp -- CPU restatement of Optimizer::LocalBundleAdjustment and
 * Optimizer::LocalBundleAdjustmentWithOdom (TEST INFRASTRUCTURE ONLY).
 *
 * Follows /root/reference/src/Optimizer.cc:838-1165 (O4) and :2137-2670 (O5), the edges in
 * src/OdomG2oTypeQuat.cc:109-212 / include/OdomG2oTypeQuat.h:133-192 and
 * Thirdparty/g2o/g2o/types/types_six_dof_expmap.{h:78-108,cpp:103-147}, the quadratic forms of
 * core/base_binary_edge.hpp:55-120, the Schur complement of core/block_solver.hpp:354-486 and the
 * LM driver restated in se3_oracle.h.  The sparse SimplicialLDLT of linear_solver_eigen.h:94-124
 * (Eigen, not vendored) is restated as a dense LDL^T of the reduced pose system.
 *
 * Graph conventions of the C-ABI (fb_local_ba_args): keyframes, map points, bird map points and
 * the three edge lists are given in graph insertion order; the pose blocks of the reduced system
 * are ordered as the free keyframes appear in kf[].
 */
#include <cmath>
#include <cstdint>
#include <cstring>
#include <vector>

#include "../include/fishbird.h"
#include "se3_oracle.h"

namespace {
using namespace orc;

enum { E_PROJ = 0, E_XYZ = 1, E_ODOM = 2 };

struct BEdge {
  int type;
  int a, b;          // PROJ/XYZ: a = point vertex, b = keyframe; ODOM: a = kf i, b = kf j
  double meas[3];
  SE3 Zinv;          // ODOM: inverse measurement
  double info;
  int level = 0;
  bool robust = false;
  double err[6] = {0, 0, 0, 0, 0, 0};
  int dim() const { return type == E_PROJ ? 2 : (type == E_XYZ ? 3 : 6); }
};

struct PLBlock { int pose; double W[18]; };  // Hpl block (6x3) of one (pose, landmark) pair

struct BAProblem : LMProblem {
  bool quat_edges;  // O5 (EdgeSE3ProjectXYZ2UVQuat) vs O4 (EdgeSE3ProjectXYZ)
  double fx, fy, cx, cy, delta;
  std::vector<SE3> pose;
  std::vector<uint8_t> fixed;
  std::vector<double> pt;  // 3 per point vertex (map points, then bird map points)
  std::vector<BEdge> edges;
  const volatile uint8_t *stop = nullptr;
  // active structure
  std::vector<int> active, poseIdx, ptIdx;
  int np = 0, nl = 0;
  std::vector<double> Hpp, Hll, b, x;
  std::vector<std::vector<PLBlock>> Hpl;  // per active landmark, sorted by pose index
  std::vector<std::vector<SE3>> poseStack;
  std::vector<std::vector<double>> ptStack;

  bool terminate() override { return stop && *stop; }
  bool hasActive() override { return !active.empty() && (np + nl) > 0; }

  static double chi2(const BEdge &e) {
    double s = 0;
    for (int i = 0; i < e.dim(); i++) s += e.err[i] * (e.info * e.err[i]);
    return s;
  }
  void computeError(BEdge &e) const {
    if (e.type == E_ODOM) {  // EdgeSE3Quat::computeError, OdomG2oTypeQuat.h:182-188
      SE3 d = se3_mul(se3_mul(e.Zinv, pose[e.a]), se3_inverse(pose[e.b]));
      se3_log(d, e.err);
      return;
    }
    double p[3];
    se3_map(pose[e.b], &pt[3 * e.a], p);
    if (e.type == E_PROJ) {
      if (quat_edges) {  // OdomG2oTypeQuat.cc:138-144: fx * x / z + cx
        e.err[0] = e.meas[0] - (fx * p[0] / p[2] + cx);
        e.err[1] = e.meas[1] - (fy * p[1] / p[2] + cy);
      } else {           // types_six_dof_expmap.cpp:141-147: (x/z)*fx + cx
        e.err[0] = e.meas[0] - ((p[0] / p[2]) * fx + cx);
        e.err[1] = e.meas[1] - ((p[1] / p[2]) * fy + cy);
      }
    } else {
      for (int i = 0; i < 3; i++) e.err[i] = e.meas[i] - p[i];
    }
  }
  bool depthPositive(const BEdge &e) const {
    double p[3];
    se3_map(pose[e.b], &pt[3 * e.a], p);
    return p[2] > 0.0;
  }
  // SparseOptimizer::initializeOptimization(level), sparse_optimizer.cpp:199-267
  void initialize(int level) {
    active.clear();
    const int nk = (int)pose.size(), npt = (int)pt.size() / 3;
    std::vector<int> useP(nk, 0), useL(npt, 0);
    for (size_t i = 0; i < edges.size(); i++) {
      const BEdge &e = edges[i];
      if (e.level != level) continue;
      if (e.type == E_ODOM) {
        if (fixed[e.a] && fixed[e.b]) continue;  // allVerticesFixed
        useP[e.a] = useP[e.b] = 1;
      } else {
        useL[e.a] = 1;
        useP[e.b] = 1;
      }
      active.push_back((int)i);
    }
    poseIdx.assign(nk, -1);
    ptIdx.assign(npt, -1);
    np = nl = 0;
    for (int k = 0; k < nk; k++) if (useP[k] && !fixed[k]) poseIdx[k] = np++;
    for (int j = 0; j < npt; j++) if (useL[j]) ptIdx[j] = nl++;
  }
  void computeActiveErrors() override {
    for (int i : active) computeError(edges[i]);
  }
  double activeRobustChi2() override {
    double chi = 0;
    for (int i : active) {
      const BEdge &e = edges[i];
      if (e.robust) { double rho[2]; huber(chi2(e), delta, rho); chi += rho[0]; }
      else chi += chi2(e);
    }
    return chi;
  }
  void addPL(int l, int pi, const double *W) {
    auto &v = Hpl[l];
    size_t k = 0;
    while (k < v.size() && v[k].pose < pi) k++;
    if (k < v.size() && v[k].pose == pi) { for (int i = 0; i < 18; i++) v[k].W[i] += W[i]; return; }
    PLBlock blk;
    blk.pose = pi;
    std::memcpy(blk.W, W, sizeof(blk.W));
    v.insert(v.begin() + k, blk);
  }
  void buildSystem() override {
    const int P6 = 6 * np;
    Hpp.assign((size_t)P6 * P6, 0.0);
    Hll.assign((size_t)9 * nl, 0.0);
    b.assign((size_t)P6 + 3 * nl, 0.0);
    Hpl.assign(nl, {});
    for (int ei : active) {
      const BEdge &e = edges[ei];
      if (e.type == E_ODOM) {  // EdgeSE3Quat::linearizeOplus, OdomG2oTypeQuat.cc:190-204
        double J[36], A[36], Bm[36], t1[36], a2[36], a1[36];
        {
          const double *er = e.err;
          double S1[9], S2[9];
          skew(er, S1);      // omega
          skew(er + 3, S2);  // upsilon
          for (int i = 0; i < 36; i++) J[i] = 0;
          for (int r = 0; r < 3; r++)
            for (int c = 0; c < 3; c++) {
              J[r * 6 + c] = S1[r * 3 + c];
              J[(r + 3) * 6 + c + 3] = S1[r * 3 + c];
              J[(r + 3) * 6 + c] = S2[r * 3 + c];
            }
          for (int i = 0; i < 36; i++) J[i] = 0.5 * J[i];
          for (int i = 0; i < 6; i++) J[i * 6 + i] += 1.0;
        }
        se3_adj(pose[e.b], a2);
        se3_adj(se3_inverse(pose[e.a]), a1);
        auto mul6 = [](const double *X, const double *Y, double *Z) {
          for (int i = 0; i < 6; i++)
            for (int j = 0; j < 6; j++) {
              double s = 0;
              for (int k = 0; k < 6; k++) s += X[i * 6 + k] * Y[k * 6 + j];
              Z[i * 6 + j] = s;
            }
        };
        mul6(J, a2, t1);
        mul6(t1, a1, A);                       // _jacobianOplusXi
        for (int i = 0; i < 36; i++) Bm[i] = -J[i];  // _jacobianOplusXj
        const int pi = poseIdx[e.a], pj = poseIdx[e.b];
        // no robust kernel: omega_r = -omega*e; b += A^T omega_r; H += A^T omega A (base_binary_edge.hpp:72-87)
        double orr[6];
        for (int r = 0; r < 6; r++) orr[r] = -(e.info * e.err[r]);
        auto acc = [&](const double *X, int px, const double *Y, int py) {
          for (int i = 0; i < 6; i++)
            for (int j = 0; j < 6; j++) {
              double s = 0;
              for (int r = 0; r < 6; r++) s += X[r * 6 + i] * e.info * Y[r * 6 + j];
              Hpp[(size_t)(6 * px + i) * P6 + 6 * py + j] += s;
            }
        };
        if (pi >= 0) {
          for (int i = 0; i < 6; i++) { double s = 0; for (int r = 0; r < 6; r++) s += A[r * 6 + i] * orr[r]; b[6 * pi + i] += s; }
          acc(A, pi, A, pi);
          if (pj >= 0) {  // off-diagonal block, both triangles kept in the dense matrix
            acc(A, pi, Bm, pj);
            for (int i = 0; i < 6; i++)
              for (int j = 0; j < 6; j++) {
                double s = 0;
                for (int r = 0; r < 6; r++) s += Bm[r * 6 + i] * e.info * A[r * 6 + j];
                Hpp[(size_t)(6 * pj + i) * P6 + 6 * pi + j] += s;
              }
          }
        }
        if (pj >= 0) {
          for (int i = 0; i < 6; i++) { double s = 0; for (int r = 0; r < 6; r++) s += Bm[r * 6 + i] * orr[r]; b[6 * pj + i] += s; }
          acc(Bm, pj, Bm, pj);
        }
        continue;
      }
      // point-pose edges: vertex 0 = point (Xi), vertex 1 = pose (Xj)
      const int D = e.dim();
      double p[3], R[9], Ji[9], Jj[18];  // Ji: D x 3, Jj: D x 6
      se3_map(pose[e.b], &pt[3 * e.a], p);
      quat_to_R(pose[e.b].r, R);
      if (e.type == E_PROJ) {
        const double X = p[0], Y = p[1], Z = p[2];
        if (quat_edges) {  // OdomG2oTypeQuat.cc:109-129
          const double z2 = Z * Z;
          const double jep[6] = {-(fx / Z), -0.0, -(-fx * X / z2), -0.0, -(fy / Z), -(-fy * Y / z2)};
          double S[9];
          skew(p, S);
          double jpk[18];  // [-skew(p), I]
          for (int r = 0; r < 3; r++)
            for (int c = 0; c < 3; c++) { jpk[r * 6 + c] = -S[r * 3 + c]; jpk[r * 6 + 3 + c] = (r == c) ? 1.0 : 0.0; }
          for (int r = 0; r < 2; r++) {
            for (int c = 0; c < 6; c++) Jj[r * 6 + c] = jep[r * 3] * jpk[c] + jep[r * 3 + 1] * jpk[6 + c] + jep[r * 3 + 2] * jpk[12 + c];
            for (int c = 0; c < 3; c++) Ji[r * 3 + c] = jep[r * 3] * R[c] + jep[r * 3 + 1] * R[3 + c] + jep[r * 3 + 2] * R[6 + c];
          }
        } else {  // types_six_dof_expmap.cpp:103-139
          const double z_2 = Z * Z;
          const double tmp[6] = {fx, 0, -X / Z * fx, 0, fy, -Y / Z * fy};
          const double s = -1. / Z;
          for (int r = 0; r < 2; r++)
            for (int c = 0; c < 3; c++)
              Ji[r * 3 + c] = (s * tmp[r * 3]) * R[c] + (s * tmp[r * 3 + 1]) * R[3 + c] + (s * tmp[r * 3 + 2]) * R[6 + c];
          Jj[0] = X * Y / z_2 * fx; Jj[1] = -(1 + (X * X / z_2)) * fx; Jj[2] = Y / Z * fx;
          Jj[3] = -1. / Z * fx;     Jj[4] = 0;                          Jj[5] = X / z_2 * fx;
          Jj[6] = (1 + Y * Y / z_2) * fy; Jj[7] = -X * Y / z_2 * fy;    Jj[8] = -X / Z * fy;
          Jj[9] = 0;                Jj[10] = -1. / Z * fy;              Jj[11] = Y / z_2 * fy;
        }
      } else {  // EdgeSE3ProjectXYZ2XYZQuat, OdomG2oTypeQuat.cc:157-169
        double S[9];
        skew(p, S);
        for (int r = 0; r < 3; r++)
          for (int c = 0; c < 3; c++) {
            Jj[r * 6 + c] = S[r * 3 + c];                 // -(-skew(p))
            Jj[r * 6 + 3 + c] = (r == c) ? -1.0 : -0.0;   // -I
            Ji[r * 3 + c] = -R[r * 3 + c];
          }
      }
      double rho1 = 1.;
      if (e.robust) { double rho[2]; huber(chi2(e), delta, rho); rho1 = rho[1]; }
      const double w = rho1 * e.info;
      const int l = ptIdx[e.a], pj = poseIdx[e.b];
      // omega_r = -omega*e (*rho1); point: b += Ji^T omega_r, Hll += Ji^T w Ji
      double orr[3];
      for (int r = 0; r < D; r++) orr[r] = -(e.info * e.err[r]) * rho1;
      for (int i = 0; i < 3; i++) {
        double s = 0;
        for (int r = 0; r < D; r++) s += Ji[r * 3 + i] * orr[r];
        b[(size_t)P6 + 3 * l + i] += s;
        for (int j = 0; j < 3; j++) {
          double h = 0;
          for (int r = 0; r < D; r++) h += Ji[r * 3 + i] * w * Ji[r * 3 + j];
          Hll[9 * l + 3 * i + j] += h;
        }
      }
      if (pj >= 0) {
        double W[18];  // Hpl block: Jj^T w Ji  (6 x 3)
        for (int i = 0; i < 6; i++) {
          double s = 0;
          for (int r = 0; r < D; r++) s += Jj[r * 6 + i] * orr[r];
          b[6 * pj + i] += s;
          for (int j = 0; j < 6; j++) {
            double h = 0;
            for (int r = 0; r < D; r++) h += Jj[r * 6 + i] * w * Jj[r * 6 + j];
            Hpp[(size_t)(6 * pj + i) * P6 + 6 * pj + j] += h;
          }
          for (int j = 0; j < 3; j++) {
            double h = 0;
            for (int r = 0; r < D; r++) h += Jj[r * 6 + i] * w * Ji[r * 3 + j];
            W[i * 3 + j] = h;
          }
        }
        addPL(l, pj, W);
      }
    }
  }
  double maxDiagonal() override {
    double m = 0;
    const int P6 = 6 * np;
    for (int j = 0; j < P6; j++) m = std::max(std::fabs(Hpp[(size_t)j * P6 + j]), m);
    for (int l = 0; l < nl; l++)
      for (int j = 0; j < 3; j++) m = std::max(std::fabs(Hll[9 * l + 4 * j]), m);
    return m;
  }
  static void inv3(const double *M, double *I) {  // Eigen Matrix3d::inverse (cofactors / determinant)
    const double c00 = M[4] * M[8] - M[5] * M[7], c01 = M[5] * M[6] - M[3] * M[8], c02 = M[3] * M[7] - M[4] * M[6];
    const double det = M[0] * c00 + M[1] * c01 + M[2] * c02;
    const double id = 1.0 / det;
    I[0] = c00 * id; I[1] = (M[2] * M[7] - M[1] * M[8]) * id; I[2] = (M[1] * M[5] - M[2] * M[4]) * id;
    I[3] = c01 * id; I[4] = (M[0] * M[8] - M[2] * M[6]) * id; I[5] = (M[2] * M[3] - M[0] * M[5]) * id;
    I[6] = c02 * id; I[7] = (M[1] * M[6] - M[0] * M[7]) * id; I[8] = (M[0] * M[4] - M[1] * M[3]) * id;
  }
  // BlockSolver::solve with setLambda/restoreDiagonal, block_solver.hpp:354-486,564-604
  bool solve(double lambda) override {
    const int P6 = 6 * np;
    x.assign((size_t)P6 + 3 * nl, 0.0);
    std::vector<double> S(Hpp);
    for (int j = 0; j < P6; j++) S[(size_t)j * P6 + j] += lambda;
    std::vector<double> coeff(P6, 0.0), Dinv((size_t)9 * nl);
    for (int l = 0; l < nl; l++) {
      double D[9];
      for (int i = 0; i < 9; i++) D[i] = Hll[9 * l + i];
      D[0] += lambda; D[4] += lambda; D[8] += lambda;
      double *Di = &Dinv[9 * l];
      inv3(D, Di);
      const double *bl = &b[(size_t)P6 + 3 * l];
      double db[3];
      for (int i = 0; i < 3; i++) db[i] = Di[i * 3] * bl[0] + Di[i * 3 + 1] * bl[1] + Di[i * 3 + 2] * bl[2];
      const auto &col = Hpl[l];
      for (size_t a = 0; a < col.size(); a++) {
        const int i1 = col[a].pose;
        double BD[18];
        for (int i = 0; i < 6; i++)
          for (int j = 0; j < 3; j++)
            BD[i * 3 + j] = col[a].W[i * 3] * Di[j] + col[a].W[i * 3 + 1] * Di[3 + j] + col[a].W[i * 3 + 2] * Di[6 + j];
        for (int i = 0; i < 6; i++) coeff[6 * i1 + i] += col[a].W[i * 3] * db[0] + col[a].W[i * 3 + 1] * db[1] + col[a].W[i * 3 + 2] * db[2];
        for (size_t c = a; c < col.size(); c++) {  // upper blocks i2 >= i1, mirrored
          const int i2 = col[c].pose;
          for (int i = 0; i < 6; i++)
            for (int j = 0; j < 6; j++) {
              const double v = BD[i * 3] * col[c].W[j * 3] + BD[i * 3 + 1] * col[c].W[j * 3 + 1] + BD[i * 3 + 2] * col[c].W[j * 3 + 2];
              S[(size_t)(6 * i1 + i) * P6 + 6 * i2 + j] -= v;
              if (i2 != i1) S[(size_t)(6 * i2 + j) * P6 + 6 * i1 + i] -= v;
            }
        }
      }
    }
    if (P6 > 0) {
      std::vector<double> bs(P6);
      for (int i = 0; i < P6; i++) bs[i] = b[i] - coeff[i];
      // the solver sees the upper triangle only: symmetrise from it
      for (int i = 0; i < P6; i++)
        for (int j = 0; j < i; j++) S[(size_t)i * P6 + j] = S[(size_t)j * P6 + i];
      if (!ldlt_solve(S, P6, bs.data(), x.data(), true)) return false;
    }
    for (int l = 0; l < nl; l++) {  // xl = Dinv (bl - B^T xp)
      double cl[3] = {b[(size_t)P6 + 3 * l], b[(size_t)P6 + 3 * l + 1], b[(size_t)P6 + 3 * l + 2]};
      for (const auto &blk : Hpl[l])
        for (int j = 0; j < 3; j++)
          for (int i = 0; i < 6; i++) cl[j] -= blk.W[i * 3 + j] * x[6 * blk.pose + i];
      const double *Di = &Dinv[9 * l];
      for (int i = 0; i < 3; i++) x[(size_t)P6 + 3 * l + i] = Di[i * 3] * cl[0] + Di[i * 3 + 1] * cl[1] + Di[i * 3 + 2] * cl[2];
    }
    return true;
  }
  double scaleTerm(double lambda) override {
    double s = 0;
    for (size_t j = 0; j < x.size(); j++) s += x[j] * (lambda * x[j] + b[j]);
    return s;
  }
  void push() override { poseStack.push_back(pose); ptStack.push_back(pt); }
  void pop() override { pose = poseStack.back(); pt = ptStack.back(); poseStack.pop_back(); ptStack.pop_back(); }
  void discardTop() override { poseStack.pop_back(); ptStack.pop_back(); }
  void applyUpdate() override {
    const int P6 = 6 * np;
    for (size_t k = 0; k < pose.size(); k++)
      if (poseIdx[k] >= 0) pose[k] = se3_mul(se3_exp(&x[6 * poseIdx[k]]), pose[k]);  // VertexSE3Quat::oplusImpl
    for (size_t j = 0; j < pt.size() / 3; j++)
      if (ptIdx[j] >= 0) for (int i = 0; i < 3; i++) pt[3 * j + i] += x[(size_t)P6 + 3 * ptIdx[j] + i];  // VertexSBAPointXYZ
  }
};

}  // namespace

// schedule: its1 < 0 selects LocalBundleAdjustment[WithOdom] (optimize(5) robust, gate, optimize(10)); otherwise
// BundleAdjustmentWithOdom (Optimizer.cc:1786-2135): ONE optimize(its1), robust iff `robust`, delta sqrt(5.99), no gate
static int ba_common(const fb_local_ba_args *A, int its1, int robust);
static double g_ba_min_margin = 1e300;
static long g_ba_decisions = 0;
// margin behind "identical outlier flags" for an implementation that rounds differently (tests/test_parity_sweeps_gpu.py)
extern "C" void orc_ba_margin_reset() { g_ba_min_margin = 1e300; g_ba_decisions = 0; }
extern "C" int orc_ba_margin_get(double *min_margin, long *decisions) { *min_margin = g_ba_min_margin; *decisions = g_ba_decisions; return FB_OK; }
extern "C" int orc_local_ba(const fb_local_ba_args *A) { return ba_common(A, -1, 1); }
extern "C" int orc_global_ba(const fb_local_ba_args *A, int n_iterations, int robust) { return ba_common(A, n_iterations, robust); }

static int ba_common(const fb_local_ba_args *A, int its1, int robust1) {
  const bool global = its1 >= 0;
  BAProblem P;
  P.quat_edges = A->with_odom != 0;
  P.fx = A->fx; P.fy = A->fy; P.cx = A->cx; P.cy = A->cy;
  P.delta = global ? (float)std::sqrt(5.99) : (float)std::sqrt(5.991);  // thHuber2D (:1836) / thHuberMono
  P.stop = A->stop_flag;
  P.pose.resize(A->n_kf);
  P.fixed.assign(A->kf_fixed, A->kf_fixed + A->n_kf);
  for (int k = 0; k < A->n_kf; k++) P.pose[k] = se3_from_float12(A->kf_Tcw + 12 * k);
  P.pt.resize((size_t)3 * (A->n_mp + A->n_mpb));
  for (int i = 0; i < 3 * A->n_mp; i++) P.pt[i] = A->mp_xw[i];
  for (int i = 0; i < 3 * A->n_mpb; i++) P.pt[3 * A->n_mp + i] = A->mpb_xw[i];
  const double wF = A->with_odom ? (double)A->wF : 1.0;
  for (int i = 0; i < A->n_obs; i++) {
    BEdge e;
    e.type = E_PROJ; e.a = A->obs_mp[i]; e.b = A->obs_kf[i];
    e.meas[0] = A->obs_uv[2 * i]; e.meas[1] = A->obs_uv[2 * i + 1]; e.meas[2] = 0;
    e.info = A->with_odom ? (1.0 * (double)A->obs_inv_sigma2[i]) * wF : (double)A->obs_inv_sigma2[i];
    e.robust = true;
    P.edges.push_back(e);
  }
  const int nb = A->with_odom ? A->n_bobs : 0;
  for (int i = 0; i < nb; i++) {
    BEdge e;
    e.type = E_XYZ; e.a = A->n_mp + A->bobs_mpb[i]; e.b = A->bobs_kf[i];
    for (int k = 0; k < 3; k++) e.meas[k] = A->bobs_xc[3 * i + k];
    e.info = (1.0 * (double)A->bobs_inv_sigma2[i]) * (double)A->wB;
    e.robust = true;
    P.edges.push_back(e);
  }
  const int no = A->with_odom ? A->n_odom : 0;
  for (int i = 0; i < no; i++) {
    BEdge e;
    e.type = E_ODOM; e.a = A->odom_kf_i[i]; e.b = A->odom_kf_j[i];
    e.Zinv = se3_inverse(se3_from_float12(A->odom_Tij + 12 * i));  // Converter::toMatrix4d(float) -> SE3Quat(R,t)
    e.info = A->odom_info[i];
    e.robust = false;
    P.edges.push_back(e);
  }
  auto gate = [](double chi2) {  // every chi2 > 5.991 decision (Optimizer.cc:2534-2565 gate, :2579-2610 collection) notes its margin
    const double m = std::fabs(chi2 - 5.991) / 5.991;
    if (m < g_ba_min_margin) g_ba_min_margin = m;
    g_ba_decisions++;
    return chi2 > 5.991;
  };
  if (A->stop_flag && *A->stop_flag) return FB_OK;  // Optimizer.cc:2498-2500
  if (global && !robust1)
    for (auto &e : P.edges) e.robust = false;
  P.initialize(0);
  lm_optimize(P, global ? its1 : 5);
  const bool more = !global && !(A->stop_flag && *A->stop_flag);
  if (more) {
    for (int i = 0; i < A->n_obs; i++) {
      BEdge &e = P.edges[i];
      if (gate(BAProblem::chi2(e)) || !P.depthPositive(e)) e.level = 1;
      e.robust = false;
    }
    for (int i = 0; i < nb; i++) {
      BEdge &e = P.edges[A->n_obs + i];
      P.computeError(e);
      if (gate(BAProblem::chi2(e))) e.level = 1;
      e.robust = false;
    }
    P.initialize(0);
    lm_optimize(P, 10);
  }
  if (!global) {
    for (int i = 0; i < A->n_obs; i++) {
      const BEdge &e = P.edges[i];
      A->obs_outlier[i] = (gate(BAProblem::chi2(e)) || !P.depthPositive(e)) ? 1 : 0;
    }
    for (int i = 0; i < nb; i++) A->bobs_outlier[i] = gate(BAProblem::chi2(P.edges[A->n_obs + i])) ? 1 : 0;
  }
  for (int k = 0; k < A->n_kf; k++)
    if (!A->kf_fixed[k]) se3_to_float12(P.pose[k], A->kf_Tcw + 12 * k);  // local keyframes only; fixed ones unchanged
  for (int i = 0; i < 3 * A->n_mp; i++) A->mp_xw[i] = (float)P.pt[i];
  for (int i = 0; i < 3 * A->n_mpb; i++) A->mpb_xw[i] = (float)P.pt[3 * A->n_mp + i];
  return FB_OK;
}
