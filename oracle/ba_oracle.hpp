// ORACLE — TEST INFRASTRUCTURE ONLY.  Not part of the shipped product path.
// Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may use
// anything under oracle/.
//
// CPU restatement (C++17, no Eigen) of the Voxel-SLAM local-BA factor and the
// three Levenberg-Marquardt optimizers, in the reference's order of operations.
// Reference files (read-only, /root/reference/VoxelSLAM/src):
//   TL = tools.hpp, VM = voxel_map.hpp, PI = preintegration.hpp
// Parity status: "parity unpinned" against the reference binary (unbuildable here:
// TL:4 needs Eigen, VM:11 needs ROS).  Pinned by KATs in tests/: finite-difference
// gradient/Hessian of lambda_min, numpy eigh / solve / inv fixtures (tests/golden/).
// Third-party arithmetic restated from its published algorithm: Eigen 3.3.7
// (README.md:26) SelfAdjointEigenSolver -> cyclic Jacobi (same result up to
// rounding and eigenvector sign); LDLT -> Eigen's diagonal-pivoted unblocked LDLT;
// Matrix<15,15>::inverse() -> partial-pivot LU.
#pragma once
#include "smallmat.hpp"
#include <cfloat>
#include <deque>
#include <thread>
#include <functional>
#include <cstdio>

namespace vso {

static const int DIM = 15;   // TL:16
static const int DVEL = 6;   // VM:502

// ---------------------------------------------------------------- SO(3) helpers
inline M3 hat(const V3 &v) {  // TL:93-100
  M3 o;
  o(0, 1) = -v[2]; o(0, 2) = v[1];
  o(1, 0) = v[2];  o(1, 2) = -v[0];
  o(2, 0) = -v[1]; o(2, 1) = v[0];
  return o;
}

inline M3 Exp(const V3 &ang) {  // TL:51-66
  double n = ang.norm();
  if (n >= 1e-11) {
    V3 ax = ang / n;
    M3 K = hat(ax);
    return M3::Identity() + K * std::sin(n) + (K * K) * (1.0 - std::cos(n));
  }
  return M3::Identity();
}

inline M3 Exp(const V3 &ang_vel, double dt) {  // TL:68-84
  double n = ang_vel.norm();
  if (n > 1e-7) {
    V3 ax = ang_vel / n;
    M3 K = hat(ax);
    double r = n * dt;
    return M3::Identity() + K * std::sin(r) + (K * K) * (1.0 - std::cos(r));
  }
  return M3::Identity();
}

inline V3 Log(const M3 &R) {  // TL:86-91
  double tr = R.trace();
  double theta = (tr > 3.0 - 1e-6) ? 0.0 : std::acos(0.5 * (tr - 1));
  V3 K = v3(R(2, 1) - R(1, 2), R(0, 2) - R(2, 0), R(1, 0) - R(0, 1));
  return (std::fabs(theta) < 0.001) ? (K * 0.5) : (K * (0.5 * theta / std::sin(theta)));
}

inline M3 jr(V3 vec) {  // TL:102-116
  double ang = vec.norm();
  if (ang < 1e-9) return M3::Identity();
  vec /= ang;
  double ra = std::sin(ang) / ang;
  return M3::Identity() * ra + (vec * vec.transpose()) * (1 - ra) - hat(vec) * ((1 - std::cos(ang)) / ang);
}

// Eigen::AngleAxisd(Matrix3d): rotation matrix -> quaternion -> angle/axis
// (Eigen/src/Geometry/Quaternion.h quaternionbase_assign_impl<Matrix,3,3>,
//  AngleAxis.h operator=(QuaternionBase)); used by jr_inv (TL:120-122).
inline void angle_axis_from_R(const M3 &m, double &angle, V3 &axis) {
  double w, x, y, z;
  double t = m.trace();
  if (t > 0) {
    t = std::sqrt(t + 1.0);
    w = 0.5 * t;
    t = 0.5 / t;
    x = (m(2, 1) - m(1, 2)) * t;
    y = (m(0, 2) - m(2, 0)) * t;
    z = (m(1, 0) - m(0, 1)) * t;
  } else {
    int i = 0;
    if (m(1, 1) > m(0, 0)) i = 1;
    if (m(2, 2) > m(i, i)) i = 2;
    int j = (i + 1) % 3, k = (j + 1) % 3;
    t = std::sqrt(m(i, i) - m(j, j) - m(k, k) + 1.0);
    double q[3];
    q[i] = 0.5 * t;
    t = 0.5 / t;
    w = (m(k, j) - m(j, k)) * t;
    q[j] = (m(j, i) + m(i, j)) * t;
    q[k] = (m(k, i) + m(i, k)) * t;
    x = q[0]; y = q[1]; z = q[2];
  }
  double n = std::sqrt(x * x + y * y + z * z);
  if (n != 0.0) {
    angle = 2.0 * std::atan2(n, std::fabs(w));
    if (w < 0) n = -n;
    axis = v3(x / n, y / n, z / n);
  } else {
    angle = 0;
    axis = v3(1, 0, 0);
  }
}

inline M3 jr_inv(const M3 &rotR) {  // TL:118-133
  double ang; V3 axi;
  angle_axis_from_R(rotR, ang, axi);
  if (ang < 1e-9) return M3::Identity();
  double ctt = ang / 2 / std::tan(ang / 2);
  return M3::Identity() * ctt + (axi * axi.transpose()) * (1 - ctt) + hat(axi) * (ang / 2);
}

// ---------------------------------------------------------------- state (TL:135-199)
struct IMUST {
  double t = 0;
  M3 R = M3::Identity();
  V3 p, v, bg, ba, g;
  Mat<15, 15> cov;
  IMUST() { setZero(); }
  void setZero() {  // TL:188-197 (g is left untouched there; default-zero here)
    t = 0; R.setIdentity(); p.setZero(); v.setZero(); bg.setZero(); ba.setZero();
    cov.setIdentity(); cov *= 0.0001;
    for (int i = 9; i < 15; i++) for (int j = 9; j < 15; j++) cov(i, j) = (i == j) ? 0.00001 : 0.0;
  }
};

// ---------------------------------------------------------------- PointCluster (TL:304-365)
struct PointCluster {
  M3 P; V3 v; int N = 0;
  void clear() { P.setZero(); v.setZero(); N = 0; }
  void push(const V3 &vec) { N++; P += vec * vec.transpose(); v += vec; }  // TL:326-331
  M3 cov() const { V3 c = v / (double)N; return P / (double)N - c * c.transpose(); }  // TL:333-337
  PointCluster &operator+=(const PointCluster &s) { P += s.P; v += s.v; N += s.N; return *this; }
  PointCluster &operator-=(const PointCluster &s) { P -= s.P; v -= s.v; N -= s.N; return *this; }
  void transform(const PointCluster &s, const IMUST &st) {  // TL:357-363
    N = s.N;
    v = st.R * s.v + st.p * (double)N;
    M3 rp = st.R * s.v * st.p.transpose();
    P = st.R * s.P * st.R.transpose() + rp + rp.transpose() + (st.p * st.p.transpose()) * (double)N;
  }
};

// ---------------------------------------------------------------- symmetric 3x3 eigen
// Stands in for Eigen::SelfAdjointEigenSolver<Matrix3d> (VM:312, VM:1416, VM:1525):
// eigenvalues ascending, eigenvectors orthonormal in columns, sign arbitrary; only the
// lower triangle of the input is read.  Cyclic Jacobi (Rutishauser's stable rotation).
inline void eig3_sym(const M3 &Ain, V3 &w, M3 &V) {
  double a[3][3];
  for (int i = 0; i < 3; i++) for (int j = 0; j <= i; j++) a[i][j] = a[j][i] = Ain(i, j);
  double v[3][3] = {{1, 0, 0}, {0, 1, 0}, {0, 0, 1}};
  for (int sweep = 0; sweep < 60; sweep++) {
    double off = std::fabs(a[0][1]) + std::fabs(a[0][2]) + std::fabs(a[1][2]);
    if (off == 0.0) break;
    for (int p = 0; p < 2; p++)
      for (int q = p + 1; q < 3; q++) {
        double apq = a[p][q];
        if (apq == 0.0) continue;
        double g = 100.0 * std::fabs(apq);
        if (sweep > 3 && std::fabs(a[p][p]) + g == std::fabs(a[p][p]) && std::fabs(a[q][q]) + g == std::fabs(a[q][q])) {
          a[p][q] = a[q][p] = 0.0;
          continue;
        }
        double h = a[q][q] - a[p][p], t;
        if (std::fabs(h) + g == std::fabs(h)) {
          t = apq / h;
        } else {
          double theta = 0.5 * h / apq;
          t = 1.0 / (std::fabs(theta) + std::sqrt(1.0 + theta * theta));
          if (theta < 0) t = -t;
        }
        double c = 1.0 / std::sqrt(1 + t * t), s = t * c;
        // A <- J^T A J with J the rotation in plane (p,q)
        int r = 3 - p - q;
        double arp = a[r][p], arq = a[r][q];
        a[p][p] -= t * apq;
        a[q][q] += t * apq;
        a[p][q] = a[q][p] = 0.0;
        a[r][p] = a[p][r] = c * arp - s * arq;
        a[r][q] = a[q][r] = s * arp + c * arq;
        for (int k = 0; k < 3; k++) {
          double vkp = v[k][p], vkq = v[k][q];
          v[k][p] = c * vkp - s * vkq;
          v[k][q] = s * vkp + c * vkq;
        }
      }
  }
  int idx[3] = {0, 1, 2};
  double d[3] = {a[0][0], a[1][1], a[2][2]};
  for (int i = 0; i < 2; i++)
    for (int j = i + 1; j < 3; j++)
      if (d[idx[j]] < d[idx[i]]) std::swap(idx[i], idx[j]);
  for (int c = 0; c < 3; c++) {
    w[c] = d[idx[c]];
    for (int r = 0; r < 3; r++) V(r, c) = v[r][idx[c]];
  }
}

// ---------------------------------------------------------------- dense solvers
// Eigen 3.3.7 LDLT (Cholesky/LDLT.h ldlt_inplace<Lower>::unblocked + _solve_impl):
// diagonal pivoting on max |d_ii|, unit-lower L, solve via P, L, D (zero where |d|<=tol), L^T, P^T.
struct LDLT {
  int n = 0;
  MatX m;
  std::vector<int> tr;
  void compute(const MatX &A) {
    n = A.rows;
    m = A;
    tr.assign(n, 0);
    std::vector<double> temp(n);
    for (int k = 0; k < n; k++) {
      int piv = k;
      double big = std::fabs(m(k, k));
      for (int i = k + 1; i < n; i++)
        if (std::fabs(m(i, i)) > big) { big = std::fabs(m(i, i)); piv = i; }
      tr[k] = piv;
      if (piv != k) {
        for (int j = 0; j < k; j++) std::swap(m(k, j), m(piv, j));
        for (int i = piv + 1; i < n; i++) std::swap(m(i, k), m(i, piv));
        std::swap(m(k, k), m(piv, piv));
        for (int i = k + 1; i < piv; i++) std::swap(m(i, k), m(piv, i));
      }
      int rs = n - k - 1;
      if (k > 0) {
        for (int j = 0; j < k; j++) temp[j] = m(j, j) * m(k, j);
        double s = 0;
        for (int j = 0; j < k; j++) s += m(k, j) * temp[j];
        m(k, k) -= s;
        for (int i = k + 1; i < n; i++) {
          double t = 0;
          for (int j = 0; j < k; j++) t += m(i, j) * temp[j];
          m(i, k) -= t;
        }
      }
      double akk = m(k, k);
      bool valid = std::fabs(akk) > 0.0;
      if (k == 0 && !valid) {
        for (int j = 0; j < n; j++) tr[j] = j;
        break;
      }
      if (rs > 0 && valid)
        for (int i = k + 1; i < n; i++) m(i, k) /= akk;
    }
  }
  VecX solve(const VecX &b) const {
    VecX x = b;
    for (int k = 0; k < n; k++) if (tr[k] != k) std::swap(x[k], x[tr[k]]);
    for (int i = 0; i < n; i++) {
      double s = x[i];
      for (int j = 0; j < i; j++) s -= m(i, j) * x[j];
      x[i] = s;
    }
    const double tol = DBL_MIN;
    for (int i = 0; i < n; i++) {
      double d = m(i, i);
      if (std::fabs(d) > tol) x[i] /= d; else x[i] = 0.0;
    }
    for (int i = n - 1; i >= 0; i--) {
      double s = x[i];
      for (int j = i + 1; j < n; j++) s -= m(j, i) * x[j];
      x[i] = s;
    }
    for (int k = n - 1; k >= 0; k--) if (tr[k] != k) std::swap(x[k], x[tr[k]]);
    return x;
  }
};

// Matrix<double,15,15>::inverse() (PI:166, PI:244): Eigen uses PartialPivLU for N>4.
template <int N>
inline Mat<N, N> inverse_lu(const Mat<N, N> &A) {
  double lu[N][N], inv[N][N];
  int perm[N];
  for (int i = 0; i < N; i++) { perm[i] = i; for (int j = 0; j < N; j++) lu[i][j] = A(i, j); }
  for (int k = 0; k < N; k++) {
    int piv = k; double big = std::fabs(lu[k][k]);
    for (int i = k + 1; i < N; i++) if (std::fabs(lu[i][k]) > big) { big = std::fabs(lu[i][k]); piv = i; }
    if (piv != k) { for (int j = 0; j < N; j++) std::swap(lu[k][j], lu[piv][j]); std::swap(perm[k], perm[piv]); }
    for (int i = k + 1; i < N; i++) {
      lu[i][k] /= lu[k][k];
      for (int j = k + 1; j < N; j++) lu[i][j] -= lu[i][k] * lu[k][j];
    }
  }
  for (int c = 0; c < N; c++) {
    double y[N];
    for (int i = 0; i < N; i++) {
      double s = (perm[i] == c) ? 1.0 : 0.0;
      for (int j = 0; j < i; j++) s -= lu[i][j] * y[j];
      y[i] = s;
    }
    for (int i = N - 1; i >= 0; i--) {
      double s = y[i];
      for (int j = i + 1; j < N; j++) s -= lu[i][j] * inv[j][c];
      inv[i][c] = s / lu[i][i];
    }
  }
  Mat<N, N> R;
  for (int i = 0; i < N; i++) for (int j = 0; j < N; j++) R(i, j) = inv[i][j];
  return R;
}

// ---------------------------------------------------------------- LidarFactor (VM:124-339)
struct LidarFactor {
  std::vector<PointCluster> sig_vecs;                     // VM:128
  std::vector<std::vector<PointCluster>> plvec_voxels;    // VM:129
  std::vector<double> coeffs;                             // VM:130
  std::vector<V3> eig_values;                             // VM:131
  std::vector<M3> eig_vectors;                            // VM:132
  std::vector<PointCluster> pcr_adds;                     // VM:133
  int win_size;
  explicit LidarFactor(int w) : win_size(w) {}

  void push_voxel(const std::vector<PointCluster> &vec_orig, const PointCluster &fix, double coe,
                  const V3 &eig_value, const M3 &eig_vector, const PointCluster &pcr_add) {  // VM:139-147
    plvec_voxels.push_back(vec_orig);
    sig_vecs.push_back(fix);
    coeffs.push_back(coe);
    eig_values.push_back(eig_value);
    eig_vectors.push_back(eig_vector);
    pcr_adds.push_back(pcr_add);
  }

  // VM:150-282.  Hess is (6W x 6W), JacT is 6W.
  void acc_evaluate2(const std::vector<IMUST> &xs, int head, int end, MatX &Hess, VecX &JacT, double &residual) {
    Hess.setZero(); JacT.setZero(); residual = 0;
    const int kk = 0;
    std::vector<V3> viRiTuk(win_size);
    std::vector<M3> viRiTukukT(win_size);
    std::vector<Mat<3, 6>> Auk(win_size);
    M3 umumT;
    for (int a = head; a < end; a++) {
      std::vector<PointCluster> &sig_orig = plvec_voxels[a];
      double coe = coeffs[a];
      V3 lmbd = eig_values[a];
      M3 U = eig_vectors[a];
      int NN = pcr_adds[a].N;
      V3 vBar = pcr_adds[a].v / (double)NN;
      V3 u[3] = {U.col(0), U.col(1), U.col(2)};
      V3 &uk = u[kk];
      M3 ukukT = uk * uk.transpose();
      umumT.setZero();
      for (int i = 0; i < 3; i++)
        if (i != kk) umumT += (u[i] * u[i].transpose()) * (2.0 / (lmbd[kk] - lmbd[i]));

      for (int i = 0; i < win_size; i++)
        if (sig_orig[i].N != 0) {
          M3 Pi = sig_orig[i].P;
          V3 vi = sig_orig[i].v;
          M3 Ri = xs[i].R;
          double ni = sig_orig[i].N;
          M3 vihat = hat(vi);
          V3 RiTuk = Ri.transpose() * uk;
          M3 RiTukhat = hat(RiTuk);
          V3 PiRiTuk = Pi * RiTuk;
          viRiTuk[i] = vihat * RiTuk;
          viRiTukukT[i] = viRiTuk[i] * uk.transpose();
          V3 ti_v = xs[i].p - vBar;
          double ukTti_v = dot(uk, ti_v);
          M3 combo1 = hat(PiRiTuk) + vihat * ukTti_v;
          V3 combo2 = Ri * vi + ti_v * ni;
          M3 Arot = (Ri * Pi + ti_v * vi.transpose()) * RiTukhat - Ri * combo1;
          M3 Atr = combo2 * uk.transpose() + M3::Identity() * dot(combo2, uk);
          Auk[i].template setBlock<3, 3>(0, 0, Arot);
          Auk[i].template setBlock<3, 3>(0, 3, Atr);
          Auk[i] /= (double)NN;

          V6 jjt = Auk[i].transpose() * uk;
          JacT.addSeg<6>(6 * i, jjt * coe);

          M3 HRt = viRiTukukT[i] * (2.0 / NN * (1.0 - ni / NN));
          M6 Hb = Auk[i].transpose() * umumT * Auk[i];
          M3 jr3 = hat(jjt.block<3, 1>(0, 0));
          M3 rr = ((combo1 - RiTukhat * Pi) * RiTukhat) * (2.0 / NN) -
                  (viRiTuk[i] * viRiTuk[i].transpose()) * (2.0 / NN / NN) - jr3 * 0.5;
          Hb.addBlock<3, 3>(0, 0, rr);
          Hb.addBlock<3, 3>(0, 3, HRt);
          Hb.addBlock<3, 3>(3, 0, HRt.transpose());
          Hb.addBlock<3, 3>(3, 3, ukukT * (2.0 / NN * (ni - ni * ni / NN)));
          Hess.addBlock<6, 6>(6 * i, 6 * i, Hb * coe);
        }

      for (int i = 0; i < win_size - 1; i++)
        if (sig_orig[i].N != 0) {
          double ni = sig_orig[i].N;
          for (int j = i + 1; j < win_size; j++)
            if (sig_orig[j].N != 0) {
              double nj = sig_orig[j].N;
              M6 Hb = Auk[i].transpose() * umumT * Auk[j];
              Hb.addBlock<3, 3>(0, 0, (viRiTuk[i] * viRiTuk[j].transpose()) * (-2.0 / NN / NN));
              Hb.addBlock<3, 3>(0, 3, viRiTukukT[i] * (-2.0 * nj / NN / NN));
              Hb.addBlock<3, 3>(3, 0, viRiTukukT[j].transpose() * (-2.0 * ni / NN / NN));
              Hb.addBlock<3, 3>(3, 3, ukukT * (-2.0 * ni * nj / NN / NN));
              Hess.addBlock<6, 6>(6 * i, 6 * j, Hb * coe);
            }
        }
      residual += coe * lmbd[kk];
    }
    for (int i = 1; i < win_size; i++)
      for (int j = 0; j < i; j++)
        for (int r = 0; r < 6; r++)
          for (int c = 0; c < 6; c++) Hess(6 * i + r, 6 * j + c) = Hess(6 * j + c, 6 * i + r);
  }

  // VM:285-325
  void evaluate_only_residual(const std::vector<IMUST> &xs, int head, int end, double &residual) {
    residual = 0;
    int kk = 0;
    PointCluster pcr;
    for (int a = head; a < end; a++) {
      const std::vector<PointCluster> &sig_orig = plvec_voxels[a];
      PointCluster sig = sig_vecs[a];
      for (int i = 0; i < win_size; i++)
        if (sig_orig[i].N != 0) {
          pcr.transform(sig_orig[i], xs[i]);
          sig += pcr;
        }
      V3 vBar = sig.v / (double)sig.N;
      M3 cov = sig.P / (double)sig.N - vBar * vBar.transpose();
      V3 lmbd; M3 U;
      eig3_sym(cov, lmbd, U);
      eig_values[a] = lmbd;
      eig_vectors[a] = U;
      pcr_adds[a] = sig;
      residual += coeffs[a] * lmbd[kk];
    }
  }

  void clear() {  // VM:328-336
    sig_vecs.clear(); plvec_voxels.clear(); eig_values.clear(); eig_vectors.clear(); pcr_adds.clear(); coeffs.clear();
  }
};

// ---------------------------------------------------------------- IMU pre-integration (PI:11-331)
struct ImuSample { double t; V3 gyr, acc; };  // sensor_msgs::Imu fields used at PI:59-66

struct ImuNoise {  // PI:8-9 globals
  double scale_gravity = 1.0;
  M6 noiseMeas, noiseWalk;
};

struct IMU_PRE {
  M3 R_delta; V3 p_delta, v_delta; V3 bg, ba;
  M3 R_bg, p_bg, p_ba, v_bg, v_ba;
  double dtime = 0;
  V3 dbg, dba, dbg_buf, dba_buf;
  Mat<15, 15> cov;
  IMU_PRE(const V3 &bg1 = V3(), const V3 &ba1 = V3()) {  // PI:32-48
    bg = bg1; ba = ba1; R_delta.setIdentity();
  }

  void push_imu(const std::deque<ImuSample> &imus, const ImuNoise &nz) {  // PI:50-73
    for (size_t k = 1; k < imus.size(); k++) {
      const ImuSample &i1 = imus[k - 1], &i2 = imus[k];
      double dt = i2.t - i1.t;
      V3 cur_gyr = (i1.gyr + i2.gyr) * 0.5;
      V3 cur_acc = (i1.acc + i2.acc) * 0.5;
      cur_gyr = cur_gyr - bg;
      cur_acc = cur_acc * nz.scale_gravity - ba;
      add_imu(cur_gyr, cur_acc, dt, nz);
    }
  }

  void add_imu(const V3 &cur_gyr, const V3 &cur_acc, double dt, const ImuNoise &nz) {  // PI:75-135
    dtime += dt;
    M3 R_inc = Exp(cur_gyr, dt);
    M3 R_jr = jr(cur_gyr * dt);
    M3 R_dt = R_delta * dt;
    M3 R_dt2_2 = R_delta * (0.5 * dt * dt);
    M3 acc_skew = hat(cur_acc);
    p_ba = p_ba + v_ba * dt - R_dt2_2;
    p_bg = p_bg + v_bg * dt - R_dt2_2 * acc_skew * R_bg;
    v_ba = v_ba - R_dt;
    v_bg = v_bg - R_dt * acc_skew * R_bg;
    R_bg = R_inc.transpose() * R_bg - R_jr * dt;

    Mat<9, 9> A = Mat<9, 9>::Identity();
    Mat<9, 6> B;
    A.setBlock<3, 3>(0, 0, R_inc.transpose());
    A.setBlock<3, 3>(3, 0, -(R_dt2_2 * acc_skew));
    A.setBlock<3, 3>(3, 6, M3::Identity() * dt);
    A.setBlock<3, 3>(6, 0, -(R_dt * acc_skew));
    B.setBlock<3, 3>(0, 0, R_jr * dt);
    B.setBlock<3, 3>(3, 3, R_dt2_2);
    B.setBlock<3, 3>(6, 3, R_dt);
    Mat<9, 9> c9 = cov.block<9, 9>(0, 0);
    c9 = A * c9 * A.transpose() + B * nz.noiseMeas * B.transpose();
    cov.setBlock<9, 9>(0, 0, c9);
    cov.addBlock<6, 6>(9, 9, nz.noiseWalk * dt);

    p_delta += v_delta * dt + R_dt2_2 * cur_acc;
    v_delta += R_dt * cur_acc;
    R_delta = R_delta * R_inc;
  }

  // PI:137-212 (with_g=false) and PI:214-294 (with_g=true).  jtj is (30|33)^2 row-major, gg (30|33).
  double give_evaluate_impl(const IMUST &st1, const IMUST &st2, MatX &jtj, VecX &gg, bool jac_enable, bool with_g) {
    Mat<15, 15> joca, jocb;
    Mat<15, 1> rr;
    Mat<15, 3> jocg;
    M3 R_correct = R_delta * Exp(R_bg * dbg);
    V3 t_correct = p_delta + p_bg * dbg + p_ba * dba;
    V3 v_correct = v_delta + v_bg * dbg + v_ba * dba;
    M3 res_r = R_correct.transpose() * st1.R.transpose() * st2.R;
    V3 exp_v = st1.R.transpose() * (st2.v - st1.v - st1.g * dtime);
    V3 res_v = exp_v - v_correct;
    V3 exp_t = st1.R.transpose() * (st2.p - st1.p - st1.v * dtime - st1.g * (0.5 * dtime * dtime));
    V3 res_t = exp_t - t_correct;
    V3 res_bg = st2.bg - st1.bg;
    V3 res_ba = st2.ba - st1.ba;
    double b_wei = 1;
    rr.setBlock<3, 1>(0, 0, Log(res_r));
    rr.setBlock<3, 1>(3, 0, res_t);
    rr.setBlock<3, 1>(6, 0, res_v);
    rr.setBlock<3, 1>(9, 0, res_bg * b_wei);
    rr.setBlock<3, 1>(12, 0, res_ba * b_wei);
    Mat<15, 15> cov_inv = inverse_lu<15>(cov);
    if (jac_enable) {
      M3 I = M3::Identity();
      M3 JR_inv = jr_inv(res_r);
      joca.setBlock<3, 3>(0, 0, -(JR_inv * st2.R.transpose() * st1.R));
      jocb.setBlock<3, 3>(0, 0, JR_inv);
      joca.setBlock<3, 3>(0, 9, -(JR_inv * res_r.transpose() * jr(R_bg * dbg) * R_bg));
      joca.setBlock<3, 3>(3, 0, hat(exp_t));
      joca.setBlock<3, 3>(3, 3, -st1.R.transpose());
      joca.setBlock<3, 3>(3, 6, -(st1.R.transpose() * dtime));
      joca.setBlock<3, 3>(3, 9, -p_bg);
      joca.setBlock<3, 3>(3, 12, -p_ba);
      jocb.setBlock<3, 3>(3, 3, st1.R.transpose());
      joca.setBlock<3, 3>(6, 0, hat(exp_v));
      joca.setBlock<3, 3>(6, 6, -st1.R.transpose());
      joca.setBlock<3, 3>(6, 9, -v_bg);
      joca.setBlock<3, 3>(6, 12, -v_ba);
      jocb.setBlock<3, 3>(6, 6, st1.R.transpose());
      joca.setBlock<3, 3>(9, 9, -(I * b_wei));
      joca.setBlock<3, 3>(12, 12, -(I * b_wei));
      jocb.setBlock<3, 3>(9, 9, I * b_wei);
      jocb.setBlock<3, 3>(12, 12, I * b_wei);
      int nc = with_g ? 33 : 30;
      if (with_g) {
        jocg.setBlock<3, 3>(3, 0, st1.R.transpose() * (-0.5 * dtime * dtime));
        jocg.setBlock<3, 3>(6, 0, st1.R.transpose() * (-dtime));
      }
      // joc = [joca | jocb | jocg] (15 x nc); jtj = joc^T cov_inv joc ; gg = joc^T cov_inv rr
      std::vector<double> joc(15 * nc, 0.0);
      for (int r = 0; r < 15; r++) {
        for (int c = 0; c < 15; c++) { joc[r * nc + c] = joca(r, c); joc[r * nc + 15 + c] = jocb(r, c); }
        if (with_g) for (int c = 0; c < 3; c++) joc[r * nc + 30 + c] = jocg(r, c);
      }
      std::vector<double> cj(15 * nc, 0.0);  // cov_inv * joc
      for (int r = 0; r < 15; r++)
        for (int c = 0; c < nc; c++) {
          double s = 0;
          for (int k = 0; k < 15; k++) s += cov_inv(r, k) * joc[k * nc + c];
          cj[r * nc + c] = s;
        }
      Mat<15, 1> cr = cov_inv * rr;
      for (int r = 0; r < nc; r++) {
        for (int c = 0; c < nc; c++) {
          double s = 0;
          for (int k = 0; k < 15; k++) s += joc[k * nc + r] * cj[k * nc + c];
          jtj(r, c) = s;
        }
        double s = 0;
        for (int k = 0; k < 15; k++) s += joc[k * nc + r] * cr[k];
        gg[r] = s;
      }
    }
    return dot(rr, Mat<15, 1>(cov_inv * rr));
  }
  double give_evaluate(const IMUST &a, const IMUST &b, MatX &jtj, VecX &gg, bool jac) { return give_evaluate_impl(a, b, jtj, gg, jac, false); }
  double give_evaluate_g(const IMUST &a, const IMUST &b, MatX &jtj, VecX &gg, bool jac) { return give_evaluate_impl(a, b, jtj, gg, jac, true); }

  void update_state(const Mat<15, 1> &dxi) {  // PI:296-303
    dbg_buf = dbg; dba_buf = dba;
    dbg += dxi.block<3, 1>(9, 0);
    dba += dxi.block<3, 1>(12, 0);
  }
};

// ---------------------------------------------------------------- worker partition
// The reference spawns std::thread workers over contiguous voxel ranges
// [part*i, part*(i+1)) with double->int truncation (VM:371-374, VM:541-546).
// run_parallel=false evaluates the same ranges sequentially (deterministic, for tests).
struct ThreadCfg { bool run_parallel = true; };

// ---------------------------------------------------------------- Lidar_BA_Optimizer (VM:342-498)
struct Lidar_BA_Optimizer {
  int win_size = 0, jac_leng = 0, thd_num = 2;
  ThreadCfg tc;

  double divide_thread(std::vector<IMUST> &x_stats, LidarFactor &voxhess, MatX &Hess, VecX &JacT) {  // VM:347-389
    double residual = 0;
    Hess.setZero(); JacT.setZero();
    std::vector<MatX> hessians(thd_num, MatX(jac_leng, jac_leng));
    std::vector<VecX> jacobins(thd_num, VecX(jac_leng));
    int tthd_num = thd_num;
    std::vector<double> resis(tthd_num, 0);
    int g_size = (int)voxhess.plvec_voxels.size();
    if (g_size < tthd_num) tthd_num = 1;
    std::vector<std::thread> th;
    double part = 1.0 * g_size / tthd_num;
    for (int i = 1; i < tthd_num; i++) {
      auto fn = [&, i]() { voxhess.acc_evaluate2(x_stats, (int)(part * i), (int)(part * (i + 1)), hessians[i], jacobins[i], resis[i]); };
      if (tc.run_parallel) th.emplace_back(fn); else fn();
    }
    for (int i = 0; i < tthd_num; i++) {
      if (i != 0) { if (tc.run_parallel) th[i - 1].join(); }
      else voxhess.acc_evaluate2(x_stats, 0, (int)part, hessians[0], jacobins[0], resis[0]);
      Hess += hessians[i]; JacT += jacobins[i]; residual += resis[i];
    }
    return residual;
  }

  double only_residual(std::vector<IMUST> &x_stats, LidarFactor &voxhess, bool *too_few = nullptr) {  // VM:391-420
    double residual1 = 0;
    std::vector<double> residuals(thd_num, 0);
    int g_size = (int)voxhess.plvec_voxels.size();
    if (g_size < thd_num) {  // VM:399-403: printf("Too Less Voxel"); exit(0);
      if (too_few) *too_few = true;
      return 0;
    }
    std::vector<std::thread> th;
    double part = 1.0 * g_size / thd_num;
    for (int i = 1; i < thd_num; i++) {
      auto fn = [&, i]() { voxhess.evaluate_only_residual(x_stats, (int)(part * i), (int)(part * (i + 1)), residuals[i]); };
      if (tc.run_parallel) th.emplace_back(fn); else fn();
    }
    for (int i = 0; i < thd_num; i++) {
      if (i != 0) { if (tc.run_parallel) th[i - 1].join(); }
      else voxhess.evaluate_only_residual(x_stats, (int)(part * i), (int)(part * (i + 1)), residuals[i]);
      residual1 += residuals[i];
    }
    return residual1;
  }

  // VM:422-497.  Returns is_converge; *status = -1 when the reference would exit(0).
  bool damping_iter(std::vector<IMUST> &x_stats, LidarFactor &voxhess, MatX *hess, std::vector<double> &resis,
                    int max_iter = 3, int *status = nullptr, std::vector<double> *trace = nullptr) {
    win_size = voxhess.win_size;
    jac_leng = win_size * 6;
    double u = 0.01, v = 2;
    MatX D(jac_leng, jac_leng), Hess(jac_leng, jac_leng);
    VecX JacT(jac_leng), dxi(jac_leng);
    hess->resize(jac_leng, jac_leng);
    double residual1 = 0, residual2 = 0, q;
    bool is_calc_hess = true;
    std::vector<IMUST> x_stats_temp = x_stats;
    bool is_converge = true;
    if (status) *status = 0;
    for (int i = 0; i < max_iter; i++) {
      if (is_calc_hess) {
        residual1 = divide_thread(x_stats, voxhess, Hess, JacT);
        *hess = Hess;
      }
      if (i == 0) resis.push_back(residual1);
      for (int r = 0; r < 6; r++) for (int c = 0; c < jac_leng; c++) { Hess(r, c) = 0; Hess(c, r) = 0; }
      for (int r = 0; r < 6; r++) { Hess(r, r) = 1; JacT[r] = 0; }
      MatX A = Hess;
      for (int r = 0; r < jac_leng; r++) A(r, r) += u * Hess(r, r);
      VecX rhs(jac_leng);
      for (int r = 0; r < jac_leng; r++) rhs[r] = -JacT[r];
      LDLT ld; ld.compute(A);
      dxi = ld.solve(rhs);
      for (int j = 0; j < win_size; j++) {
        x_stats_temp[j].R = x_stats[j].R * Exp(dxi.seg<3>(6 * j));
        x_stats_temp[j].p = x_stats[j].p + dxi.seg<3>(6 * j + 3);
      }
      double q1 = 0;
      for (int r = 0; r < jac_leng; r++) q1 += dxi[r] * (u * Hess(r, r) * dxi[r] - JacT[r]);
      q1 *= 0.5;
      bool too_few = false;
      residual2 = only_residual(x_stats_temp, voxhess, &too_few);
      if (too_few) { if (status) *status = -1; return false; }
      q = (residual1 - residual2);
      if (trace) { trace->push_back(residual1); trace->push_back(residual2); trace->push_back(u); trace->push_back(v); trace->push_back(q1); }
      if (q > 0) {
        x_stats = x_stats_temp;
        double one_three = 1.0 / 3;
        q = q / q1;
        v = 2;
        q = 1 - std::pow(2 * q - 1, 3);
        u *= (q < one_three ? one_three : q);
        is_calc_hess = true;
      } else {
        u = u * v;
        v = 2 * v;
        is_calc_hess = false;
        is_converge = false;
      }
      if (std::fabs((residual1 - residual2) / residual1) < 1e-6) break;
    }
    resis.push_back(residual2);
    return is_converge;
  }
};

// ---------------------------------------------------------------- LI_BA_Optimizer (VM:504-714) and
// LI_BA_OptimizerGravity (VM:717-976).  One class, gravity=false/true selects the variant.
struct LI_BA_Optimizer {
  int win_size = 0, jac_leng = 0, imu_leng = 0;
  bool gravity = false;
  double imu_coef = 1e-4;  // VM:500
  ThreadCfg tc;

  void hess_plus(MatX &Hess, VecX &JacT, MatX &hs, VecX &js) {  // VM:509-517
    for (int i = 0; i < win_size; i++) {
      for (int r = 0; r < DVEL; r++) JacT[i * DIM + r] += js[i * DVEL + r];
      for (int j = 0; j < win_size; j++)
        for (int r = 0; r < DVEL; r++)
          for (int c = 0; c < DVEL; c++) Hess(i * DIM + r, j * DIM + c) += hs(i * DVEL + r, j * DVEL + c);
    }
  }

  double divide_thread(std::vector<IMUST> &x_stats, LidarFactor &voxhess, std::deque<IMU_PRE *> &imus_factor,
                       MatX &Hess, VecX &JacT) {  // VM:519-584, VM:746-825
    int thd_num = 5;
    double residual = 0;
    Hess.setZero(); JacT.setZero();
    std::vector<MatX> hessians(thd_num, MatX(jac_leng, jac_leng));
    std::vector<VecX> jacobins(thd_num, VecX(jac_leng));
    std::vector<double> resis(thd_num, 0);
    int tthd_num = thd_num;
    int g_size = (int)voxhess.plvec_voxels.size();
    if (g_size < tthd_num) tthd_num = 1;
    double part = 1.0 * g_size / tthd_num;
    std::vector<std::thread> th;
    for (int i = 1; i < tthd_num; i++) {
      auto fn = [&, i]() { voxhess.acc_evaluate2(x_stats, (int)(part * i), (int)(part * (i + 1)), hessians[i], jacobins[i], resis[i]); };
      if (tc.run_parallel) th.emplace_back(fn); else fn();
    }
    int nb = gravity ? 2 * DIM + 3 : 2 * DIM;
    MatX jtj(nb, nb);
    VecX gg(nb);
    for (int i = 0; i < win_size - 1; i++) {
      jtj.setZero(); gg.setZero();
      if (!gravity) {
        residual += imus_factor[i]->give_evaluate(x_stats[i], x_stats[i + 1], jtj, gg, true);
        for (int r = 0; r < 2 * DIM; r++) {
          for (int c = 0; c < 2 * DIM; c++) Hess(i * DIM + r, i * DIM + c) += jtj(r, c);
          JacT[i * DIM + r] += gg[r];
        }
      } else {
        residual += imus_factor[i]->give_evaluate_g(x_stats[i], x_stats[i + 1], jtj, gg, true);
        for (int r = 0; r < 2 * DIM; r++) {
          for (int c = 0; c < 2 * DIM; c++) Hess(i * DIM + r, i * DIM + c) += jtj(r, c);
          for (int c = 0; c < 3; c++) {
            Hess(i * DIM + r, imu_leng - 3 + c) += jtj(r, 2 * DIM + c);
            Hess(imu_leng - 3 + c, i * DIM + r) += jtj(2 * DIM + c, r);
          }
          JacT[i * DIM + r] += gg[r];
        }
        for (int r = 0; r < 3; r++) {
          for (int c = 0; c < 3; c++) Hess(imu_leng - 3 + r, imu_leng - 3 + c) += jtj(2 * DIM + r, 2 * DIM + c);
          JacT[imu_leng - 3 + r] += gg[2 * DIM + r];
        }
      }
    }
    Hess *= imu_coef;
    JacT *= imu_coef;
    residual *= (imu_coef * 0.5);
    for (int i = 0; i < tthd_num; i++) {
      if (i != 0) { if (tc.run_parallel) th[i - 1].join(); }
      else voxhess.acc_evaluate2(x_stats, 0, (int)part, hessians[0], jacobins[0], resis[0]);
      hess_plus(Hess, JacT, hessians[i], jacobins[i]);
      residual += resis[i];
    }
    return residual;
  }

  double only_residual(std::vector<IMUST> &x_stats, LidarFactor &voxhess, std::deque<IMU_PRE *> &imus_factor) {  // VM:586-622, VM:831-870
    double residual1 = 0, residual2 = 0;
    MatX jtj(2 * DIM + 3, 2 * DIM + 3);
    VecX gg(2 * DIM + 3);
    int thd_num = 5;
    std::vector<double> residuals(thd_num, 0);
    int g_size = (int)voxhess.plvec_voxels.size();
    if (g_size < thd_num) thd_num = 1;
    std::vector<std::thread> th;
    double part = 1.0 * g_size / thd_num;
    for (int i = 1; i < thd_num; i++) {
      auto fn = [&, i]() { voxhess.evaluate_only_residual(x_stats, (int)(part * i), (int)(part * (i + 1)), residuals[i]); };
      if (tc.run_parallel) th.emplace_back(fn); else fn();
    }
    for (int i = 0; i < win_size - 1; i++)
      residual1 += gravity ? imus_factor[i]->give_evaluate_g(x_stats[i], x_stats[i + 1], jtj, gg, false)
                           : imus_factor[i]->give_evaluate(x_stats[i], x_stats[i + 1], jtj, gg, false);
    residual1 *= (imu_coef * 0.5);
    for (int i = 0; i < thd_num; i++) {
      if (i != 0) { if (tc.run_parallel) th[i - 1].join(); }
      else voxhess.evaluate_only_residual(x_stats, (int)(part * i), (int)(part * (i + 1)), residuals[i]);
      residual2 += residuals[i];
    }
    return (residual1 + residual2);
  }

  // VM:624-713 (gravity=false: max_iter fixed 3, gauge fixes DIM rows, no resis)
  // VM:878-975 (gravity=true: max_iter param, gauge fixes 6 rows, resis gets [first,last]).
  void damping_iter(std::vector<IMUST> &x_stats, LidarFactor &voxhess, std::deque<IMU_PRE *> &imus_factor,
                    std::vector<double> *resis, MatX *hess, int max_iter, std::vector<double> *trace = nullptr) {
    win_size = voxhess.win_size;
    jac_leng = win_size * 6;
    imu_leng = win_size * DIM + (gravity ? 3 : 0);
    double u = 0.01, v = 2;
    MatX Hess(imu_leng, imu_leng);
    VecX JacT(imu_leng), dxi(imu_leng);
    hess->resize(imu_leng, imu_leng);
    double residual1 = 0, residual2 = 0, q;
    bool is_calc_hess = true;
    std::vector<IMUST> x_stats_temp = x_stats;
    int gauge = gravity ? 6 : DIM;
    if (!gravity) max_iter = 3;
    for (int i = 0; i < max_iter; i++) {
      if (is_calc_hess) {
        residual1 = divide_thread(x_stats, voxhess, imus_factor, Hess, JacT);
        *hess = Hess;
      }
      if (gravity && i == 0 && resis) resis->push_back(residual1);
      for (int r = 0; r < gauge; r++) for (int c = 0; c < imu_leng; c++) { Hess(r, c) = 0; Hess(c, r) = 0; }
      for (int r = 0; r < gauge; r++) { Hess(r, r) = 1; JacT[r] = 0; }
      MatX A = Hess;
      for (int r = 0; r < imu_leng; r++) A(r, r) += u * Hess(r, r);
      VecX rhs(imu_leng);
      for (int r = 0; r < imu_leng; r++) rhs[r] = -JacT[r];
      LDLT ld; ld.compute(A);
      dxi = ld.solve(rhs);
      if (gravity) x_stats_temp[0].g += dxi.seg<3>(imu_leng - 3);  // VM:921
      for (int j = 0; j < win_size; j++) {
        x_stats_temp[j].R = x_stats[j].R * Exp(dxi.seg<3>(DIM * j));
        x_stats_temp[j].p = x_stats[j].p + dxi.seg<3>(DIM * j + 3);
        x_stats_temp[j].v = x_stats[j].v + dxi.seg<3>(DIM * j + 6);
        x_stats_temp[j].bg = x_stats[j].bg + dxi.seg<3>(DIM * j + 9);
        x_stats_temp[j].ba = x_stats[j].ba + dxi.seg<3>(DIM * j + 12);
        if (gravity) x_stats_temp[j].g = x_stats_temp[0].g;  // VM:930
      }
      for (int j = 0; j < win_size - 1; j++) imus_factor[j]->update_state(dxi.seg<15>(DIM * j));
      double q1 = 0;
      for (int r = 0; r < imu_leng; r++) q1 += dxi[r] * (u * Hess(r, r) * dxi[r] - JacT[r]);
      q1 *= 0.5;
      residual2 = only_residual(x_stats_temp, voxhess, imus_factor);
      q = (residual1 - residual2);
      if (trace) { trace->push_back(residual1); trace->push_back(residual2); trace->push_back(u); trace->push_back(v); trace->push_back(q1); }
      if (q > 0) {
        x_stats = x_stats_temp;
        double one_three = 1.0 / 3;
        q = q / q1;
        v = 2;
        q = 1 - std::pow(2 * q - 1, 3);
        u *= (q < one_three ? one_three : q);
        is_calc_hess = true;
      } else {
        u = u * v;
        v = 2 * v;
        is_calc_hess = false;
        for (int j = 0; j < win_size - 1; j++) {
          imus_factor[j]->dbg = imus_factor[j]->dbg_buf;
          imus_factor[j]->dba = imus_factor[j]->dba_buf;
        }
      }
      if (std::fabs((residual1 - residual2) / residual1) < 1e-6) break;
    }
    if (gravity && resis) resis->push_back(residual2);
  }
};

}  // namespace vso
