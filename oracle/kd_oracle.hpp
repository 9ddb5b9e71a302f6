// TEST INFRASTRUCTURE — CPU restatement of the kd-tree odometry used while the system initialises
// (voxelslam.cpp:1102-1252, VOXEL_SLAM::lio_state_estimation_kdtree).  Only tests/, __graft_entry__.smoke() and bench.py's
// cpu_baseline leg may use anything under oracle/.  Parity unpinned against the reference binary (PCL/FLANN/Eigen absent):
// the exact 5-nearest-neighbour search of pcl::KdTreeFLANN (L2 on float x,y,z) is restated as a brute-force scan, and
// Eigen's colPivHouseholderQr().solve() as Householder QR with column pivoting (least squares).
#pragma once
#include "map_oracle.hpp"
#include "scan_oracle.hpp"

namespace vso {

// min ||A x - b||, A 5x3: Householder QR with column pivoting (Eigen::ColPivHouseholderQR semantics for a full-rank A)
inline V3 lstsq_colpiv_5x3(double A[5][3], double b[5]) {
  int perm[3] = {0, 1, 2};
  double cn[3];
  for (int c = 0; c < 3; c++) { cn[c] = 0; for (int r = 0; r < 5; r++) cn[c] += A[r][c] * A[r][c]; }
  for (int k = 0; k < 3; k++) {
    int piv = k;
    for (int c = k + 1; c < 3; c++) if (cn[c] > cn[piv]) piv = c;
    if (piv != k) {
      for (int r = 0; r < 5; r++) std::swap(A[r][k], A[r][piv]);
      std::swap(perm[k], perm[piv]); std::swap(cn[k], cn[piv]);
    }
    double nrm = 0;
    for (int r = k; r < 5; r++) nrm += A[r][k] * A[r][k];
    nrm = std::sqrt(nrm);
    if (nrm == 0) continue;
    const double alpha = (A[k][k] > 0) ? -nrm : nrm;
    double v[5] = {0, 0, 0, 0, 0};
    for (int r = k; r < 5; r++) v[r] = A[r][k];
    v[k] -= alpha;
    double vv = 0;
    for (int r = k; r < 5; r++) vv += v[r] * v[r];
    if (vv > 0) {
      for (int c = k; c < 3; c++) {
        double s = 0;
        for (int r = k; r < 5; r++) s += v[r] * A[r][c];
        s = 2 * s / vv;
        for (int r = k; r < 5; r++) A[r][c] -= s * v[r];
      }
      double s = 0;
      for (int r = k; r < 5; r++) s += v[r] * b[r];
      s = 2 * s / vv;
      for (int r = k; r < 5; r++) b[r] -= s * v[r];
    }
    for (int c = k + 1; c < 3; c++) { cn[c] = 0; for (int r = k + 1; r < 5; r++) cn[c] += A[r][c] * A[r][c]; }
  }
  double y[3];
  for (int k = 2; k >= 0; k--) {
    double s = b[k];
    for (int c = k + 1; c < 3; c++) s -= A[k][c] * y[c];
    y[k] = s / A[k][k];
  }
  V3 x;
  for (int k = 0; k < 3; k++) x[perm[k]] = y[k];
  return x;
}

struct KdOdomOracle {
  std::vector<float> tree;   // pl_tree: x, y, z per point

  void append_world(const std::vector<V3> &pnt, const IMUST &x) {
    for (const V3 &p : pnt) { V3 w = x.R * p + x.p; tree.push_back((float)w[0]); tree.push_back((float)w[1]); tree.push_back((float)w[2]); }
  }

  // voxelslam.cpp:1102-1252.  Returns the number of EKF iterations run (0 while the tree is being seeded).
  int lio_state_estimation_kdtree(const std::vector<V3> &pnt, IMUST &x_curr) {
    const int NMATCH = 5;
    if (tree.size() / 3 < 100) { append_world(pnt, x_curr); return 0; }          // VS:1105-1118
    const int num_max_iter = 4;
    IMUST x_prop = x_curr;
    const int psize = (int)pnt.size(), m = (int)(tree.size() / 3);
    bool EKF_stop_flg = false, flg_EKF_converged = false;
    Mat<15, 15> G, H_T_H, I_STATE = Mat<15, 15>::Identity();
    G.setZero(); H_T_H.setZero();
    int rematch_num = 0, iters = 0;
    Mat<15, 15> cov_inv = inverse_lu<15>(x_curr.cov);
    std::vector<double> ds(psize, -1);
    std::vector<V3> directs(psize);
    bool refind = true;
    for (int iterCount = 0; iterCount < num_max_iter; iterCount++) {
      iters++;
      M6 HTH; V6 HTz;
      for (int i = 0; i < psize; i++) {
        M3 phat = hat(pnt[i]);
        V3 wld = x_curr.R * pnt[i] + x_curr.p;
        if (refind) {
          const float qx = (float)wld[0], qy = (float)wld[1], qz = (float)wld[2];
          float bd[NMATCH]; int bi[NMATCH];
          for (int k = 0; k < NMATCH; k++) { bd[k] = 3.4e38f; bi[k] = -1; }
          for (int j = 0; j < m; j++) {
            const float dx = qx - tree[3 * j], dy = qy - tree[3 * j + 1], dz = qz - tree[3 * j + 2];
            float d = 0; d += dx * dx; d += dy * dy; d += dz * dz;
            if (d < bd[NMATCH - 1]) {
              int k = NMATCH - 1;
              while (k > 0 && bd[k - 1] > d) { bd[k] = bd[k - 1]; bi[k] = bi[k - 1]; k--; }
              bd[k] = d; bi[k] = j;
            }
          }
          double A[5][3], Aw[5][3], b[5];
          for (int k = 0; k < NMATCH; k++) { for (int c = 0; c < 3; c++) { A[k][c] = tree[3 * bi[k] + c]; Aw[k][c] = A[k][c]; } b[k] = -1.0; }
          V3 direct = lstsq_colpiv_5x3(Aw, b);
          bool check_flag = false;
          for (int k = 0; k < NMATCH; k++)
            if (std::fabs(direct[0] * A[k][0] + direct[1] * A[k][1] + direct[2] * A[k][2] + 1.0) > 0.1) check_flag = true;
          if (check_flag) { ds[i] = -1; continue; }
          const double d = 1.0 / direct.norm();
          ds[i] = d;
          directs[i] = direct * d;
        }
        if (ds[i] >= 0) {
          const double pd2 = dot(directs[i], wld) + ds[i];
          V6 jac;
          V3 jr3 = phat * x_curr.R.transpose() * directs[i];
          for (int k = 0; k < 3; k++) { jac[k] = jr3[k]; jac[3 + k] = directs[i][k]; }
          HTH += jac * jac.transpose();
          HTz += jac * (-pd2);
        }
      }
      H_T_H.setBlock<6, 6>(0, 0, HTH);
      Mat<15, 15> K_1 = inverse_lu<15>(H_T_H + cov_inv * (1.0 / 1000));
      Mat<15, 6> K6 = K_1.block<15, 6>(0, 0);
      G.setBlock<15, 6>(0, 0, K6 * HTH);
      Mat<15, 1> vec;
      vec.setBlock<3, 1>(0, 0, Log(x_curr.R.transpose() * x_prop.R));
      vec.setBlock<3, 1>(3, 0, x_prop.p - x_curr.p);
      vec.setBlock<3, 1>(6, 0, x_prop.v - x_curr.v);
      vec.setBlock<3, 1>(9, 0, x_prop.bg - x_curr.bg);
      vec.setBlock<3, 1>(12, 0, x_prop.ba - x_curr.ba);
      Mat<15, 1> solution = K6 * HTz + vec - G.block<15, 6>(0, 0) * vec.block<6, 1>(0, 0);
      x_curr.R = x_curr.R * Exp(solution.block<3, 1>(0, 0));
      x_curr.p += solution.block<3, 1>(3, 0);
      x_curr.v += solution.block<3, 1>(6, 0);
      x_curr.bg += solution.block<3, 1>(9, 0);
      x_curr.ba += solution.block<3, 1>(12, 0);
      V3 rot_add = solution.block<3, 1>(0, 0), tra_add = solution.block<3, 1>(3, 0);
      refind = false;
      if ((rot_add.norm() * 57.3 < 0.01) && (tra_add.norm() * 100 < 0.015)) { refind = true; flg_EKF_converged = true; rematch_num++; }
      if (iterCount == num_max_iter - 2 && !flg_EKF_converged) refind = true;
      if (rematch_num >= 2 || (iterCount == num_max_iter - 1)) { x_curr.cov = (I_STATE - G) * x_curr.cov; EKF_stop_flg = true; }
      if (EKF_stop_flg) break;
    }
    // map update VS:1238-1250
    append_world(pnt, x_curr);
    std::vector<V3> pl(tree.size() / 3);
    for (size_t i = 0; i < pl.size(); i++) pl[i] = v3(tree[3 * i], tree[3 * i + 1], tree[3 * i + 2]);
    std::vector<DsPoint> out;
    down_sampling_voxel(pl, 0.5, out);
    tree.clear();
    for (const DsPoint &p : out) { tree.push_back(p.x); tree.push_back(p.y); tree.push_back(p.z); }
    return iters;
  }
};

}  // namespace vso
