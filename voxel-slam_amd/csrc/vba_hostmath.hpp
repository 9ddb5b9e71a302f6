// Host-side small dense math of the product (plain C arrays, row-major), used by the LM drivers and the
// IMU pre-integration factor.  References: tools.hpp (TL) Exp/Log/hat/jr/jr_inv TL:51-133, Eigen 3.3.7 LDLT
// (voxel_map.hpp:458/659/918 call sites) and PartialPivLU inverse (preintegration.hpp:166/244).
#pragma once
#include <cmath>
#include <cfloat>
#include <cstring>
#include <vector>
#include <algorithm>

#if defined(__HIPCC__)
#define VBH_HD __host__ __device__
#else
#define VBH_HD
#endif

namespace vbh {

// ---- 3-vectors / 3x3 (row-major)
VBH_HD inline void m3_identity(double *M) { for (int i = 0; i < 9; i++) M[i] = 0.0; M[0] = M[4] = M[8] = 1.0; }
VBH_HD inline void m3_mul(const double *A, const double *B, double *C) {  // C = A B (C may not alias)
  for (int r = 0; r < 3; r++)
    for (int c = 0; c < 3; c++) C[3 * r + c] = A[3 * r] * B[c] + A[3 * r + 1] * B[3 + c] + A[3 * r + 2] * B[6 + c];
}
VBH_HD inline void m3_mulT(const double *A, const double *B, double *C) {  // C = A B^T
  for (int r = 0; r < 3; r++)
    for (int c = 0; c < 3; c++) C[3 * r + c] = A[3 * r] * B[3 * c] + A[3 * r + 1] * B[3 * c + 1] + A[3 * r + 2] * B[3 * c + 2];
}
VBH_HD inline void m3_Tmul(const double *A, const double *B, double *C) {  // C = A^T B
  for (int r = 0; r < 3; r++)
    for (int c = 0; c < 3; c++) C[3 * r + c] = A[r] * B[c] + A[3 + r] * B[3 + c] + A[6 + r] * B[6 + c];
}
VBH_HD inline void m3_vec(const double *A, const double *x, double *y) {
  for (int r = 0; r < 3; r++) y[r] = A[3 * r] * x[0] + A[3 * r + 1] * x[1] + A[3 * r + 2] * x[2];
}
VBH_HD inline void m3_Tvec(const double *A, const double *x, double *y) {
  for (int r = 0; r < 3; r++) y[r] = A[r] * x[0] + A[3 + r] * x[1] + A[6 + r] * x[2];
}
VBH_HD inline void m3_transpose(const double *A, double *T) {
  for (int r = 0; r < 3; r++) for (int c = 0; c < 3; c++) T[3 * c + r] = A[3 * r + c];
}
VBH_HD inline void hat(const double *v, double *M) {  // TL:93-100
  M[0] = 0; M[1] = -v[2]; M[2] = v[1];
  M[3] = v[2]; M[4] = 0; M[5] = -v[0];
  M[6] = -v[1]; M[7] = v[0]; M[8] = 0;
}
VBH_HD inline double norm3(const double *v) { return std::sqrt(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]); }

// Rodrigues about unit axis ax with angle th: I + sin K + (1-cos) K^2
VBH_HD inline void rodrigues(const double *ax, double th, double *R) {
  double K[9], K2[9];
  hat(ax, K);
  m3_mul(K, K, K2);
  const double s = std::sin(th), c1 = 1.0 - std::cos(th);
  for (int i = 0; i < 9; i++) R[i] = s * K[i] + c1 * K2[i];
  R[0] += 1.0; R[4] += 1.0; R[8] += 1.0;
}
VBH_HD inline void so3_exp(const double *w, double *R) {  // TL:51-66 (threshold 1e-11)
  const double n = norm3(w);
  if (n >= 1e-11) { const double ax[3] = {w[0] / n, w[1] / n, w[2] / n}; rodrigues(ax, n, R); }
  else m3_identity(R);
}
VBH_HD inline void so3_exp_dt(const double *w, double dt, double *R) {  // TL:68-84 (threshold 1e-7)
  const double n = norm3(w);
  if (n > 1e-7) { const double ax[3] = {w[0] / n, w[1] / n, w[2] / n}; rodrigues(ax, n * dt, R); }
  else m3_identity(R);
}
VBH_HD inline void so3_log(const double *R, double *w) {  // TL:86-91
  const double tr = R[0] + R[4] + R[8];
  const double theta = (tr > 3.0 - 1e-6) ? 0.0 : std::acos(0.5 * (tr - 1));
  const double K[3] = {R[7] - R[5], R[2] - R[6], R[3] - R[1]};
  const double f = (std::fabs(theta) < 0.001) ? 0.5 : (0.5 * theta / std::sin(theta));
  for (int i = 0; i < 3; i++) w[i] = f * K[i];
}
VBH_HD inline void so3_jr(const double *vec, double *J) {  // TL:102-116
  const double ang = norm3(vec);
  if (ang < 1e-9) { m3_identity(J); return; }
  const double a[3] = {vec[0] / ang, vec[1] / ang, vec[2] / ang};
  const double ra = std::sin(ang) / ang, k = (1 - std::cos(ang)) / ang;
  double H[9];
  hat(a, H);
  for (int r = 0; r < 3; r++)
    for (int c = 0; c < 3; c++) J[3 * r + c] = (r == c ? ra : 0.0) + (1 - ra) * a[r] * a[c] - k * H[3 * r + c];
}
// Eigen::AngleAxisd(Matrix3d) = matrix -> quaternion -> angle/axis (used at TL:120-122)
VBH_HD inline void angle_axis(const double *m, double &angle, double *axis) {
  double q[4];  // w x y z
  double t = m[0] + m[4] + m[8];
  if (t > 0) {
    t = std::sqrt(t + 1.0);
    q[0] = 0.5 * t;
    t = 0.5 / t;
    q[1] = (m[7] - m[5]) * t; q[2] = (m[2] - m[6]) * t; q[3] = (m[3] - m[1]) * t;
  } else {
    int i = 0;
    if (m[4] > m[0]) i = 1;
    if (m[8] > m[4 * i]) i = 2;
    const int j = (i + 1) % 3, k = (j + 1) % 3;
    t = std::sqrt(m[4 * i] - m[4 * j] - m[4 * k] + 1.0);
    q[1 + i] = 0.5 * t;
    t = 0.5 / t;
    q[0] = (m[3 * k + j] - m[3 * j + k]) * t;
    q[1 + j] = (m[3 * j + i] + m[3 * i + j]) * t;
    q[1 + k] = (m[3 * k + i] + m[3 * i + k]) * t;
  }
  double n = std::sqrt(q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
  if (n != 0.0) {
    angle = 2.0 * std::atan2(n, std::fabs(q[0]));
    if (q[0] < 0) n = -n;
    axis[0] = q[1] / n; axis[1] = q[2] / n; axis[2] = q[3] / n;
  } else { angle = 0; axis[0] = 1; axis[1] = 0; axis[2] = 0; }
}
VBH_HD inline void so3_jr_inv(const double *R, double *J) {  // TL:118-133
  double ang, a[3];
  angle_axis(R, ang, a);
  if (ang < 1e-9) { m3_identity(J); return; }
  const double ctt = ang / 2 / std::tan(ang / 2);
  double H[9];
  hat(a, H);
  for (int r = 0; r < 3; r++)
    for (int c = 0; c < 3; c++) J[3 * r + c] = (r == c ? ctt : 0.0) + (1 - ctt) * a[r] * a[c] + (ang / 2) * H[3 * r + c];
}

// ---- general small dense (row-major) helpers
inline void mat_mul(const double *A, const double *B, double *C, int n, int k, int m) {  // C(n x m) = A(n x k) B(k x m)
  for (int r = 0; r < n; r++)
    for (int c = 0; c < m; c++) {
      double s = 0;
      for (int j = 0; j < k; j++) s += A[r * k + j] * B[j * m + c];
      C[r * m + c] = s;
    }
}
inline void mat_mul_ABt(const double *A, const double *B, double *C, int n, int k, int m) {  // C = A(n x k) B(m x k)^T
  for (int r = 0; r < n; r++)
    for (int c = 0; c < m; c++) {
      double s = 0;
      for (int j = 0; j < k; j++) s += A[r * k + j] * B[c * k + j];
      C[r * m + c] = s;
    }
}
VBH_HD inline void set_block3(double *M, int ld, int r0, int c0, const double *B, double scale = 1.0) {
  for (int r = 0; r < 3; r++) for (int c = 0; c < 3; c++) M[(r0 + r) * ld + c0 + c] = scale * B[3 * r + c];
}

// inverse via LU with partial pivoting (Eigen PartialPivLU path for fixed sizes > 4); n <= 16 works on the stack
inline void inverse_pplu(const double *A, double *Ainv, int n) {
  double lu_s[256], y_s[16];
  int perm_s[16];
  std::vector<double> lu_v, y_v;
  std::vector<int> perm_v;
  double *lu = lu_s, *y = y_s;
  int *perm = perm_s;
  if (n > 16) { lu_v.resize((size_t)n * n); y_v.resize(n); perm_v.resize(n); lu = lu_v.data(); y = y_v.data(); perm = perm_v.data(); }
  for (int i = 0; i < n * n; i++) lu[i] = A[i];
  for (int i = 0; i < n; i++) perm[i] = i;
  for (int k = 0; k < n; k++) {
    int piv = k;
    double big = std::fabs(lu[k * n + k]);
    for (int i = k + 1; i < n; i++)
      if (std::fabs(lu[i * n + k]) > big) { big = std::fabs(lu[i * n + k]); piv = i; }
    if (piv != k) {
      for (int j = 0; j < n; j++) std::swap(lu[k * n + j], lu[piv * n + j]);
      std::swap(perm[k], perm[piv]);
    }
    for (int i = k + 1; i < n; i++) {
      const double l = (lu[i * n + k] /= lu[k * n + k]);
      for (int j = k + 1; j < n; j++) lu[i * n + j] -= l * lu[k * n + j];
    }
  }
  for (int c = 0; c < n; c++) {
    for (int i = 0; i < n; i++) {
      double s = (perm[i] == c) ? 1.0 : 0.0;
      for (int j = 0; j < i; j++) s -= lu[i * n + j] * y[j];
      y[i] = s;
    }
    for (int i = n - 1; i >= 0; i--) {
      double s = y[i];
      for (int j = i + 1; j < n; j++) s -= lu[i * n + j] * Ainv[j * n + c];
      Ainv[i * n + c] = s / lu[i * n + i];
    }
  }
}

// Symmetric solve A x = b in the manner of Eigen::LDLT (diagonal pivoting on max |d_ii|, lower storage,
// zero solution component where |d| <= DBL_MIN).  A (n x n, row-major) is overwritten.
inline void ldlt_solve_inplace(double *A, const double *b, double *x, int n) {
  std::vector<int> tr(n);
  std::vector<double> tmp(n);
#define AT(r, c) A[(size_t)(r) * n + (c)]
  for (int k = 0; k < n; k++) {
    int piv = k;
    double big = std::fabs(AT(k, k));
    for (int i = k + 1; i < n; i++)
      if (std::fabs(AT(i, i)) > big) { big = std::fabs(AT(i, i)); piv = i; }
    tr[k] = piv;
    if (piv != k) {
      for (int j = 0; j < k; j++) std::swap(AT(k, j), AT(piv, j));
      for (int i = piv + 1; i < n; i++) std::swap(AT(i, k), AT(i, piv));
      std::swap(AT(k, k), AT(piv, piv));
      for (int i = k + 1; i < piv; i++) std::swap(AT(i, k), AT(piv, i));
    }
    if (k > 0) {
      for (int j = 0; j < k; j++) tmp[j] = AT(j, j) * AT(k, j);
      double s = 0;
      for (int j = 0; j < k; j++) s += AT(k, j) * tmp[j];
      AT(k, k) -= s;
      for (int i = k + 1; i < n; i++) {
        double t = 0;
        for (int j = 0; j < k; j++) t += AT(i, j) * tmp[j];
        AT(i, k) -= t;
      }
    }
    const double akk = AT(k, k);
    const bool valid = std::fabs(akk) > 0.0;
    if (k == 0 && !valid) { for (int j = 0; j < n; j++) tr[j] = j; break; }
    if (valid) for (int i = k + 1; i < n; i++) AT(i, k) /= akk;
  }
  for (int i = 0; i < n; i++) x[i] = b[i];
  for (int k = 0; k < n; k++) if (tr[k] != k) std::swap(x[k], x[tr[k]]);
  for (int i = 0; i < n; i++) { double s = x[i]; for (int j = 0; j < i; j++) s -= AT(i, j) * x[j]; x[i] = s; }
  for (int i = 0; i < n; i++) { const double d = AT(i, i); x[i] = (std::fabs(d) > DBL_MIN) ? x[i] / d : 0.0; }
  for (int i = n - 1; i >= 0; i--) { double s = x[i]; for (int j = i + 1; j < n; j++) s -= AT(j, i) * x[j]; x[i] = s; }
  for (int k = n - 1; k >= 0; k--) if (tr[k] != k) std::swap(x[k], x[tr[k]]);
#undef AT
}

// ---- IMU pre-integration factor on the flat imu_pre[304] layout of voxelba.h
struct ImuPre {
  double R_delta[9], p_delta[3], v_delta[3], bg[3], ba[3], R_bg[9], p_bg[9], p_ba[9], v_bg[9], v_ba[9];
  double dtime, dbg[3], dba[3], dbg_buf[3], dba_buf[3], cov[225];
};
static_assert(sizeof(ImuPre) == 304 * sizeof(double), "imu_pre flat layout");

struct State {  // state[25]
  double t, R[9], p[3], v[3], bg[3], ba[3], g[3];
};
static_assert(sizeof(State) == 25 * sizeof(double), "state flat layout");

inline void imu_init(ImuPre &m, const double *bg, const double *ba) {  // PI:32-48
  std::memset(&m, 0, sizeof(m));
  m3_identity(m.R_delta);
  for (int i = 0; i < 3; i++) { m.bg[i] = bg[i]; m.ba[i] = ba[i]; }
}

// PI:75-135
inline void imu_add(ImuPre &m, const double *gyr, const double *acc, double dt, const double *nm6, const double *nw6) {
  m.dtime += dt;
  double R_inc[9], R_jr[9], gd[3] = {gyr[0] * dt, gyr[1] * dt, gyr[2] * dt};
  so3_exp_dt(gyr, dt, R_inc);
  so3_jr(gd, R_jr);
  double R_dt[9], R_dt2[9], skew[9];
  for (int i = 0; i < 9; i++) { R_dt[i] = dt * m.R_delta[i]; R_dt2[i] = 0.5 * dt * dt * m.R_delta[i]; }
  hat(acc, skew);
  double T1[9], T2[9];
  // p_ba = p_ba + v_ba*dt - R_dt2_2 ;  p_bg = p_bg + v_bg*dt - R_dt2_2*acc_skew*R_bg
  m3_mul(R_dt2, skew, T1); m3_mul(T1, m.R_bg, T2);
  for (int i = 0; i < 9; i++) { m.p_ba[i] = m.p_ba[i] + m.v_ba[i] * dt - R_dt2[i]; m.p_bg[i] = m.p_bg[i] + m.v_bg[i] * dt - T2[i]; }
  // v_ba = v_ba - R_dt ; v_bg = v_bg - R_dt*acc_skew*R_bg
  m3_mul(R_dt, skew, T1); m3_mul(T1, m.R_bg, T2);
  for (int i = 0; i < 9; i++) { m.v_ba[i] -= R_dt[i]; m.v_bg[i] -= T2[i]; }
  // R_bg = R_inc^T R_bg - R_jr*dt
  m3_Tmul(R_inc, m.R_bg, T1);
  for (int i = 0; i < 9; i++) m.R_bg[i] = T1[i] - R_jr[i] * dt;

  double A[81], B[54];
  std::memset(A, 0, sizeof(A)); std::memset(B, 0, sizeof(B));
  for (int i = 0; i < 9; i++) A[i * 9 + i] = 1.0;
  double RiT[9], N1[9], N2[9];
  m3_transpose(R_inc, RiT);
  m3_mul(R_dt2, skew, N1); m3_mul(R_dt, skew, N2);
  set_block3(A, 9, 0, 0, RiT);
  set_block3(A, 9, 3, 0, N1, -1.0);
  for (int i = 0; i < 3; i++) A[(3 + i) * 9 + 6 + i] = dt;
  set_block3(A, 9, 6, 0, N2, -1.0);
  set_block3(B, 6, 0, 0, R_jr, dt);
  set_block3(B, 6, 3, 3, R_dt2);
  set_block3(B, 6, 6, 3, R_dt);
  double c9[81], AC[81], ACAt[81], Bn[54], BnBt[81];
  for (int r = 0; r < 9; r++) for (int c = 0; c < 9; c++) c9[r * 9 + c] = m.cov[r * 15 + c];
  mat_mul(A, c9, AC, 9, 9, 9);
  mat_mul_ABt(AC, A, ACAt, 9, 9, 9);
  for (int r = 0; r < 9; r++) for (int c = 0; c < 6; c++) Bn[r * 6 + c] = B[r * 6 + c] * nm6[c];
  mat_mul_ABt(Bn, B, BnBt, 9, 6, 9);
  for (int r = 0; r < 9; r++) for (int c = 0; c < 9; c++) m.cov[r * 15 + c] = ACAt[r * 9 + c] + BnBt[r * 9 + c];
  for (int i = 0; i < 6; i++) m.cov[(9 + i) * 15 + 9 + i] += nw6[i] * dt;

  double ra[3], rb[3];
  m3_vec(R_dt2, acc, ra); m3_vec(R_dt, acc, rb);
  for (int i = 0; i < 3; i++) { m.p_delta[i] += m.v_delta[i] * dt + ra[i]; }
  for (int i = 0; i < 3; i++) m.v_delta[i] += rb[i];
  m3_mul(m.R_delta, R_inc, T1);
  std::memcpy(m.R_delta, T1, sizeof(T1));
}

// PI:50-73
inline void imu_push(ImuPre &m, int n, const double *t, const double *gyr, const double *acc, const double *nm6, const double *nw6,
                     double scale_gravity) {
  for (int k = 1; k < n; k++) {
    const double dt = t[k] - t[k - 1];
    double g[3], a[3];
    for (int i = 0; i < 3; i++) {
      g[i] = 0.5 * (gyr[3 * (k - 1) + i] + gyr[3 * k + i]) - m.bg[i];
      a[i] = 0.5 * (acc[3 * (k - 1) + i] + acc[3 * k + i]) * scale_gravity - m.ba[i];
    }
    imu_add(m, g, a, dt, nm6, nw6);
  }
}

// PI:137-212 / PI:214-294, the part that does not involve cov: residual rr[15] and (joc != nullptr) the 15 x nb Jacobian
// [joca | jocb | jocg] (row-major, ld = nb, must come in zeroed).  Host and device.
VBH_HD inline void imu_residual_jacobian(const ImuPre &m, const State &s1, const State &s2, bool with_g, double *rr, double *joc, int nb) {
  double rb[3], Eb[9], Rc[9];
  m3_vec(m.R_bg, m.dbg, rb);
  so3_exp(rb, Eb);
  m3_mul(m.R_delta, Eb, Rc);
  double tc[3], vc[3], a3[3], b3[3];
  m3_vec(m.p_bg, m.dbg, a3); m3_vec(m.p_ba, m.dba, b3);
  for (int i = 0; i < 3; i++) tc[i] = m.p_delta[i] + a3[i] + b3[i];
  m3_vec(m.v_bg, m.dbg, a3); m3_vec(m.v_ba, m.dba, b3);
  for (int i = 0; i < 3; i++) vc[i] = m.v_delta[i] + a3[i] + b3[i];
  double T[9], res_r[9];
  m3_transpose(Rc, T);  // R_correct^T
  double R1tR2[9];
  m3_Tmul(s1.R, s2.R, R1tR2);
  m3_mul(T, R1tR2, res_r);
  const double dt = m.dtime;
  double dv[3], dp[3], exp_v[3], exp_t[3];
  for (int i = 0; i < 3; i++) {
    dv[i] = s2.v[i] - s1.v[i] - dt * s1.g[i];
    dp[i] = s2.p[i] - s1.p[i] - s1.v[i] * dt - 0.5 * dt * dt * s1.g[i];
  }
  m3_Tvec(s1.R, dv, exp_v);
  m3_Tvec(s1.R, dp, exp_t);
  double lr[3];
  so3_log(res_r, lr);
  for (int i = 0; i < 3; i++) {
    rr[i] = lr[i];
    rr[3 + i] = exp_t[i] - tc[i];
    rr[6 + i] = exp_v[i] - vc[i];
    rr[9 + i] = s2.bg[i] - s1.bg[i];
    rr[12 + i] = s2.ba[i] - s1.ba[i];
  }
  if (!joc) return;
  double JRi[9], R2tR1[9], M1[9], M2[9], M3[9], R1t[9], H[9], I3[9];
  m3_identity(I3);
  so3_jr_inv(res_r, JRi);
  m3_Tmul(s2.R, s1.R, R2tR1);
  m3_mul(JRi, R2tR1, M1);
  set_block3(joc, nb, 0, 0, M1, -1.0);                 // joca(0,0) = -JR_inv * R2^T * R1
  set_block3(joc, nb, 0, 15, JRi, 1.0);                // jocb(0,0) = JR_inv
  double jrb[9], resT[9];
  so3_jr(rb, jrb);
  m3_transpose(res_r, resT);
  m3_mul(JRi, resT, M1); m3_mul(M1, jrb, M2); m3_mul(M2, m.R_bg, M3);
  set_block3(joc, nb, 0, 9, M3, -1.0);                 // joca(0,9)
  m3_transpose(s1.R, R1t);
  hat(exp_t, H); set_block3(joc, nb, 3, 0, H, 1.0);
  set_block3(joc, nb, 3, 3, R1t, -1.0);
  set_block3(joc, nb, 3, 6, R1t, -dt);
  set_block3(joc, nb, 3, 9, m.p_bg, -1.0);
  set_block3(joc, nb, 3, 12, m.p_ba, -1.0);
  set_block3(joc, nb, 3, 15 + 3, R1t, 1.0);
  hat(exp_v, H); set_block3(joc, nb, 6, 0, H, 1.0);
  set_block3(joc, nb, 6, 6, R1t, -1.0);
  set_block3(joc, nb, 6, 9, m.v_bg, -1.0);
  set_block3(joc, nb, 6, 12, m.v_ba, -1.0);
  set_block3(joc, nb, 6, 15 + 6, R1t, 1.0);
  set_block3(joc, nb, 9, 9, I3, -1.0); set_block3(joc, nb, 12, 12, I3, -1.0);
  set_block3(joc, nb, 9, 15 + 9, I3, 1.0); set_block3(joc, nb, 12, 15 + 12, I3, 1.0);
  if (with_g) { set_block3(joc, nb, 3, 30, R1t, -0.5 * dt * dt); set_block3(joc, nb, 6, 30, R1t, -dt); }
}

// PI:137-212 / PI:214-294.  jtj (nb x nb), gg (nb), nb = 30 (+3 with gravity).  Returns r^T cov^-1 r.
// cinv_in: cov^-1 when the caller already holds it (cov does not change inside damping_iter), else nullptr.
inline double imu_evaluate(const ImuPre &m, const State &s1, const State &s2, bool with_g, bool jac, double *jtj, double *gg,
                           const double *cinv_in = nullptr) {
  const int nb = with_g ? 33 : 30;
  double rr[15];
  std::vector<double> joc;
  if (jac) joc.assign((size_t)15 * nb, 0.0);
  imu_residual_jacobian(m, s1, s2, with_g, rr, jac ? joc.data() : nullptr, nb);
  double cinv_loc[225];
  const double *cinv = cinv_in;
  if (!cinv) { inverse_pplu(m.cov, cinv_loc, 15); cinv = cinv_loc; }
  double cr[15];
  for (int r = 0; r < 15; r++) { double s = 0; for (int k = 0; k < 15; k++) s += cinv[r * 15 + k] * rr[k]; cr[r] = s; }
  if (jac) {
    std::vector<double> cj((size_t)15 * nb);
    mat_mul(cinv, joc.data(), cj.data(), 15, 15, nb);
    for (int r = 0; r < nb; r++) {
      for (int c = 0; c < nb; c++) {
        double s = 0;
        for (int k = 0; k < 15; k++) s += joc[k * nb + r] * cj[k * nb + c];
        jtj[r * nb + c] = s;
      }
      double s = 0;
      for (int k = 0; k < 15; k++) s += joc[k * nb + r] * cr[k];
      gg[r] = s;
    }
  }
  double q = 0;
  for (int k = 0; k < 15; k++) q += rr[k] * cr[k];
  return q;
}

}  // namespace vbh
