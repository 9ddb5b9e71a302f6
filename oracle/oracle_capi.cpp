// ORACLE — TEST INFRASTRUCTURE ONLY (see ba_oracle.hpp header).  extern "C" surface for
// ctypes so tests/ and bench.py's cpu_baseline leg can drive the CPU restatement.
//
// Flat formats (all double, row-major):
//   cluster  : [Pxx,Pxy,Pxz,Pyy,Pyz,Pzz, vx,vy,vz, N]                        (10)
//   pose     : [R(9) row-major, p(3)]                                        (12)
//   state    : [t, R(9), p(3), v(3), bg(3), ba(3), g(3)]                     (25)   (IMUST TL:135-145 minus cov)
//   imu_pre  : [R_delta(9) p_delta(3) v_delta(3) bg(3) ba(3) R_bg(9) p_bg(9) p_ba(9) v_bg(9) v_ba(9)
//               dtime(1) dbg(3) dba(3) dbg_buf(3) dba_buf(3) cov(225)]       (304)  (IMU_PRE PI:15-28)
#include "ba_oracle.hpp"
#include "map_oracle.hpp"
#include "scan_oracle.hpp"
#include "gba_oracle.hpp"
#include "kd_oracle.hpp"
#include <chrono>

using namespace vso;

namespace {
inline PointCluster unpack_cluster(const double *c) {
  PointCluster pc;
  pc.P(0, 0) = c[0]; pc.P(0, 1) = pc.P(1, 0) = c[1]; pc.P(0, 2) = pc.P(2, 0) = c[2];
  pc.P(1, 1) = c[3]; pc.P(1, 2) = pc.P(2, 1) = c[4]; pc.P(2, 2) = c[5];
  pc.v[0] = c[6]; pc.v[1] = c[7]; pc.v[2] = c[8];
  pc.N = (int)c[9];
  return pc;
}
inline void pack_cluster(const PointCluster &pc, double *c) {
  // lower triangle is what the reference's eigen-solver reads (VM:312)
  c[0] = pc.P(0, 0); c[1] = pc.P(1, 0); c[2] = pc.P(2, 0); c[3] = pc.P(1, 1); c[4] = pc.P(2, 1); c[5] = pc.P(2, 2);
  c[6] = pc.v[0]; c[7] = pc.v[1]; c[8] = pc.v[2]; c[9] = (double)pc.N;
}
inline M3 m3_from(const double *p) { M3 m; for (int i = 0; i < 9; i++) m[i] = p[i]; return m; }
inline V3 v3_from(const double *p) { return v3(p[0], p[1], p[2]); }
inline void m3_to(const M3 &m, double *p) { for (int i = 0; i < 9; i++) p[i] = m[i]; }
inline void v3_to(const V3 &v, double *p) { for (int i = 0; i < 3; i++) p[i] = v[i]; }

inline std::vector<IMUST> poses_to_states(const double *poses, int W) {
  std::vector<IMUST> xs(W);
  for (int i = 0; i < W; i++) { xs[i].R = m3_from(poses + 12 * i); xs[i].p = v3_from(poses + 12 * i + 9); }
  return xs;
}
inline void states_to_poses(const std::vector<IMUST> &xs, double *poses) {
  for (size_t i = 0; i < xs.size(); i++) { m3_to(xs[i].R, poses + 12 * i); v3_to(xs[i].p, poses + 12 * i + 9); }
}
inline IMUST state_from(const double *s) {
  IMUST x; x.t = s[0]; x.R = m3_from(s + 1); x.p = v3_from(s + 10); x.v = v3_from(s + 13);
  x.bg = v3_from(s + 16); x.ba = v3_from(s + 19); x.g = v3_from(s + 22);
  return x;
}
inline void state_to(const IMUST &x, double *s) {
  s[0] = x.t; m3_to(x.R, s + 1); v3_to(x.p, s + 10); v3_to(x.v, s + 13); v3_to(x.bg, s + 16); v3_to(x.ba, s + 19); v3_to(x.g, s + 22);
}
inline void imu_from(const double *f, IMU_PRE &m) {
  m.R_delta = m3_from(f); m.p_delta = v3_from(f + 9); m.v_delta = v3_from(f + 12); m.bg = v3_from(f + 15); m.ba = v3_from(f + 18);
  m.R_bg = m3_from(f + 21); m.p_bg = m3_from(f + 30); m.p_ba = m3_from(f + 39); m.v_bg = m3_from(f + 48); m.v_ba = m3_from(f + 57);
  m.dtime = f[66]; m.dbg = v3_from(f + 67); m.dba = v3_from(f + 70); m.dbg_buf = v3_from(f + 73); m.dba_buf = v3_from(f + 76);
  for (int i = 0; i < 225; i++) m.cov[i] = f[79 + i];
}
inline void imu_to(const IMU_PRE &m, double *f) {
  m3_to(m.R_delta, f); v3_to(m.p_delta, f + 9); v3_to(m.v_delta, f + 12); v3_to(m.bg, f + 15); v3_to(m.ba, f + 18);
  m3_to(m.R_bg, f + 21); m3_to(m.p_bg, f + 30); m3_to(m.p_ba, f + 39); m3_to(m.v_bg, f + 48); m3_to(m.v_ba, f + 57);
  f[66] = m.dtime; v3_to(m.dbg, f + 67); v3_to(m.dba, f + 70); v3_to(m.dbg_buf, f + 73); v3_to(m.dba_buf, f + 76);
  for (int i = 0; i < 225; i++) f[79 + i] = m.cov[i];
}
}  // namespace

extern "C" {

// ---- small math (KAT-3, KAT-4)
void vso_eig3(const double *A, double *w, double *V) {
  V3 ww; M3 VV; eig3_sym(m3_from(A), ww, VV); v3_to(ww, w); m3_to(VV, V);
}
void vso_exp(const double *ang, double *R) { m3_to(Exp(v3_from(ang)), R); }
void vso_log(const double *R, double *ang) { v3_to(Log(m3_from(R)), ang); }
void vso_jr(const double *ang, double *J) { m3_to(jr(v3_from(ang)), J); }
void vso_jr_inv(const double *R, double *J) { m3_to(jr_inv(m3_from(R)), J); }
void vso_cluster_from_points(const double *pts, int n, double *cl) {
  PointCluster pc; for (int i = 0; i < n; i++) pc.push(v3_from(pts + 3 * i)); pack_cluster(pc, cl);
}
void vso_cluster_transform(const double *cl, const double *pose, double *out) {
  IMUST st; st.R = m3_from(pose); st.p = v3_from(pose + 9);
  PointCluster r; r.transform(unpack_cluster(cl), st); pack_cluster(r, out);
}
void vso_cluster_cov(const double *cl, double *cov9) { m3_to(unpack_cluster(cl).cov(), cov9); }
// A (n x n, symmetric, row-major), b -> x via the restated Eigen LDLT
void vso_ldlt_solve(const double *A, const double *b, int n, double *x) {
  MatX M(n, n); VecX r(n);
  for (int i = 0; i < n * n; i++) M.a[i] = A[i];
  for (int i = 0; i < n; i++) r[i] = b[i];
  LDLT ld; ld.compute(M); VecX s = ld.solve(r);
  for (int i = 0; i < n; i++) x[i] = s[i];
}
void vso_inverse15(const double *A, double *Ainv) {
  Mat<15, 15> M; for (int i = 0; i < 225; i++) M[i] = A[i];
  Mat<15, 15> R = inverse_lu<15>(M); for (int i = 0; i < 225; i++) Ainv[i] = R[i];
}

// ---- LidarFactor handle
void *vso_factor_create(int win_size) { return new LidarFactor(win_size); }
void vso_factor_destroy(void *h) { delete (LidarFactor *)h; }
void vso_factor_clear(void *h) { ((LidarFactor *)h)->clear(); }
int vso_factor_size(void *h) { return (int)((LidarFactor *)h)->plvec_voxels.size(); }
// clusters [n][W][10], fix [n][10], coe[n], eig_val[n][3], eig_vec[n][9] (row-major, columns = eigenvectors), pcr_add[n][10]
void vso_factor_push(void *h, int n, const double *clusters, const double *fix, const double *coe,
                     const double *eig_val, const double *eig_vec, const double *pcr_add) {
  LidarFactor *f = (LidarFactor *)h;
  int W = f->win_size;
  std::vector<PointCluster> pcs(W);
  for (int a = 0; a < n; a++) {
    for (int i = 0; i < W; i++) pcs[i] = unpack_cluster(clusters + ((size_t)a * W + i) * 10);
    f->push_voxel(pcs, unpack_cluster(fix + (size_t)a * 10), coe[a], v3_from(eig_val + 3 * a), m3_from(eig_vec + 9 * a),
                  unpack_cluster(pcr_add + (size_t)a * 10));
  }
}
void vso_factor_acc_evaluate2(void *h, const double *poses, int head, int end, double *Hess, double *JacT, double *residual) {
  LidarFactor *f = (LidarFactor *)h;
  int W = f->win_size, n = 6 * W;
  std::vector<IMUST> xs = poses_to_states(poses, W);
  MatX H(n, n); VecX g(n);
  f->acc_evaluate2(xs, head, end, H, g, *residual);
  for (int i = 0; i < n * n; i++) Hess[i] = H.a[i];
  for (int i = 0; i < n; i++) JacT[i] = g[i];
}
void vso_factor_evaluate_only_residual(void *h, const double *poses, int head, int end, double *residual) {
  LidarFactor *f = (LidarFactor *)h;
  std::vector<IMUST> xs = poses_to_states(poses, f->win_size);
  f->evaluate_only_residual(xs, head, end, *residual);
}
void vso_factor_read_back(void *h, double *eig_val, double *eig_vec, double *pcr_add) {
  LidarFactor *f = (LidarFactor *)h;
  for (size_t a = 0; a < f->plvec_voxels.size(); a++) {
    v3_to(f->eig_values[a], eig_val + 3 * a);
    m3_to(f->eig_vectors[a], eig_vec + 9 * a);
    pack_cluster(f->pcr_adds[a], pcr_add + 10 * a);
  }
}
void vso_factor_read_inputs(void *h, double *clusters, double *fix, double *coe) {
  LidarFactor *f = (LidarFactor *)h;
  int W = f->win_size;
  for (size_t a = 0; a < f->plvec_voxels.size(); a++) {
    for (int i = 0; i < W; i++) pack_cluster(f->plvec_voxels[a][i], clusters + (a * W + i) * 10);
    pack_cluster(f->sig_vecs[a], fix + a * 10);
    coe[a] = f->coeffs[a];
  }
}

// ---- optimizers.  trace (optional, may be null): per iteration [r1, r2, u, v, q1]; *n_trace = entries written.
int vso_lidar_ba_damping_iter(void *h, double *poses, double *hess, double *resis2, int max_iter, int thd_num, int parallel,
                              int *status, double *trace, int *n_trace) {
  LidarFactor *f = (LidarFactor *)h;
  int W = f->win_size, n = 6 * W;
  std::vector<IMUST> xs = poses_to_states(poses, W);
  Lidar_BA_Optimizer opt; opt.thd_num = thd_num; opt.tc.run_parallel = parallel != 0;
  MatX H; std::vector<double> resis, tr;
  bool conv = opt.damping_iter(xs, *f, &H, resis, max_iter, status, &tr);
  states_to_poses(xs, poses);
  if (hess) for (int i = 0; i < n * n; i++) hess[i] = H.a[i];
  if (resis2) { resis2[0] = resis.size() > 0 ? resis[0] : 0; resis2[1] = resis.size() > 1 ? resis[1] : 0; }
  if (trace && n_trace) { for (size_t i = 0; i < tr.size(); i++) trace[i] = tr[i]; *n_trace = (int)tr.size(); }
  return conv ? 1 : 0;
}

// states [W][25] in/out, imus [W-1][304] in/out, hess out ((15W+3g)^2), resis2 out (gravity only)
void vso_li_ba_damping_iter(void *h, double *states, double *imus, int gravity, double imu_coef, int max_iter, int parallel,
                            double *hess, double *resis2, double *trace, int *n_trace) {
  LidarFactor *f = (LidarFactor *)h;
  int W = f->win_size;
  std::vector<IMUST> xs(W);
  for (int i = 0; i < W; i++) xs[i] = state_from(states + 25 * i);
  std::vector<IMU_PRE> store(W - 1);
  std::deque<IMU_PRE *> fac;
  for (int i = 0; i < W - 1; i++) { imu_from(imus + 304 * i, store[i]); fac.push_back(&store[i]); }
  LI_BA_Optimizer opt; opt.gravity = gravity != 0; opt.imu_coef = imu_coef; opt.tc.run_parallel = parallel != 0;
  MatX H; std::vector<double> resis, tr;
  opt.damping_iter(xs, *f, fac, &resis, &H, max_iter, &tr);
  for (int i = 0; i < W; i++) state_to(xs[i], states + 25 * i);
  for (int i = 0; i < W - 1; i++) imu_to(store[i], imus + 304 * i);
  int n = 15 * W + (gravity ? 3 : 0);
  if (hess) for (int i = 0; i < n * n; i++) hess[i] = H.a[i];
  if (resis2 && resis.size() >= 2) { resis2[0] = resis[0]; resis2[1] = resis[1]; }
  if (trace && n_trace) { for (size_t i = 0; i < tr.size(); i++) trace[i] = tr[i]; *n_trace = (int)tr.size(); }
}

// ---- IMU
// samples: t[n], gyr[n][3], acc[n][3]; noise diagonals (PI:9, VS:934-939); out imu_pre[304]
void vso_imu_preintegrate(int n, const double *t, const double *gyr, const double *acc, const double *bg, const double *ba,
                          const double *noise_meas_diag6, const double *noise_walk_diag6, double scale_gravity, double *out) {
  ImuNoise nz; nz.scale_gravity = scale_gravity;
  for (int i = 0; i < 6; i++) { nz.noiseMeas(i, i) = noise_meas_diag6[i]; nz.noiseWalk(i, i) = noise_walk_diag6[i]; }
  std::deque<ImuSample> imus;
  for (int i = 0; i < n; i++) imus.push_back({t[i], v3_from(gyr + 3 * i), v3_from(acc + 3 * i)});
  IMU_PRE pre(v3_from(bg), v3_from(ba));
  pre.push_imu(imus, nz);
  imu_to(pre, out);
}
// jtj ((30|33)^2), gg (30|33); returns r^T cov^-1 r
double vso_imu_give_evaluate(const double *imu, const double *st1, const double *st2, int with_g, int jac_enable, double *jtj, double *gg) {
  IMU_PRE m; imu_from(imu, m);
  int nb = with_g ? 33 : 30;
  MatX J(nb, nb); VecX g(nb);
  IMUST a = state_from(st1), b = state_from(st2);
  double r = m.give_evaluate_impl(a, b, J, g, jac_enable != 0, with_g != 0);
  if (jac_enable) { for (int i = 0; i < nb * nb; i++) jtj[i] = J.a[i]; for (int i = 0; i < nb; i++) gg[i] = g[i]; }
  return r;
}

// ---- voxel map (map_oracle.hpp)
void *vso_map_create(int win_size, double voxel_size, int max_layer, double min_eigen_value, const double *plane_thre4,
                     const double *min_point4, int max_points, int thread_num) {
  MapConfig c; c.win_size = win_size; c.voxel_size = voxel_size; c.max_layer = max_layer; c.min_eigen_value = min_eigen_value;
  for (int i = 0; i < 4; i++) { c.plane_eigen_value_thre[i] = plane_thre4[i]; c.min_point[i] = min_point4[i]; }
  c.max_points = max_points; c.thread_num = thread_num;
  return new VoxelMapOracle(c);
}
void vso_map_destroy(void *m) { delete (VoxelMapOracle *)m; }
void vso_map_key(double voxel_size, const double *pw, long long *key3) {
  VOXEL_LOC k = voxel_key(v3_from(pw), voxel_size); key3[0] = k.x; key3[1] = k.y; key3[2] = k.z;
}
// pnt_body [n][3], var [n][9] (may be null -> zero), pose(12) used for pw = R p + t; multi: cut_voxel_multi semantics
void vso_map_cut_voxel(void *m, int win_count, int n, const double *pnt_body, const double *var, const double *pose, int multi) {
  VoxelMapOracle *vm = (VoxelMapOracle *)m;
  PVec pv(n); std::vector<V3> pw(n);
  M3 R = m3_from(pose); V3 t = v3_from(pose + 9);
  for (int i = 0; i < n; i++) {
    pv[i].pnt = v3_from(pnt_body + 3 * i);
    if (var) pv[i].var = m3_from(var + 9 * i);
    pw[i] = R * pv[i].pnt + t;
  }
  if (multi) vm->cut_voxel_multi(pv, win_count, pw); else vm->cut_voxel(pv, win_count, pw);
}
void vso_map_cut_voxel_fix(void *m, int n, const double *pnt_world, double jour) {
  VoxelMapOracle *vm = (VoxelMapOracle *)m;
  PVec pv(n);
  for (int i = 0; i < n; i++) pv[i].pnt = v3_from(pnt_world + 3 * i);
  vm->cut_voxel_fix(pv, jour);
}
// recut + tras_opt over the slide map into factor handle fh (multi_recut VS:1682 / VS:699-703)
void vso_map_recut(void *m, int win_count, const double *poses, void *fh, int multi) {
  VoxelMapOracle *vm = (VoxelMapOracle *)m;
  LidarFactor *f = (LidarFactor *)fh;
  std::vector<IMUST> xs = poses_to_states(poses, win_count);
  f->clear(); f->win_size = vm->cfg.win_size;
  vm->recut_all(win_count, xs, *f, multi != 0);
}
void vso_map_margi(void *m, int win_count, const double *poses, void *fh, double jour) {
  VoxelMapOracle *vm = (VoxelMapOracle *)m;
  std::vector<IMUST> xs = poses_to_states(poses, win_count);
  vm->multi_margi(win_count, xs, *(LidarFactor *)fh, jour);
}
void vso_map_prune(void *m, double jour, int dist) { ((VoxelMapOracle *)m)->prune(jour, dist); }
void vso_map_slide(void *m, int mgsize) { ((VoxelMapOracle *)m)->slide(mgsize); }
int vso_map_num_roots(void *m) { return (int)((VoxelMapOracle *)m)->surf_map.size(); }
int vso_map_num_slide_roots(void *m) { return (int)((VoxelMapOracle *)m)->surf_map_slide.size(); }
// leaf dump: for every leaf node in surf_map: [kx,ky,kz, layer, path(octant code), N_add, N_fix, is_plane, isexist, opt_state,
//   eig_value(3), eig_vector(9), pcr_add(10), center(3), normal(3), radius] = 10 + 3 + 9 + 10 + 7 = 39 doubles
int vso_map_dump_leaves(void *m, double *out, int max_leaves) { return ((VoxelMapOracle *)m)->dump_leaves(out, max_leaves); }
int vso_map_dump_plane_var(void *m, double *out, int max_leaves) { return ((VoxelMapOracle *)m)->dump_plane_var(out, max_leaves); }
int vso_map_dump_cov_add(void *m, double *out, int max_leaves) { return ((VoxelMapOracle *)m)->dump_cov_add(out, max_leaves); }

// lio_state_estimation (voxelslam.cpp:962-1098): state[25] + cov[225] in/out; pnt_body [n][3], var_body [n][9];
// trace (optional) receives rows [match_num, |rot_add|, |tra_add|]; returns the bool result.
int vso_map_lio_state_estimation(void *m, int n, const double *pnt_body, const double *var_body, double *state, double *cov,
                                 double *trace, int *n_trace) {
  VoxelMapOracle *vm = (VoxelMapOracle *)m;
  PVec pv(n);
  for (int i = 0; i < n; i++) { pv[i].pnt = v3_from(pnt_body + 3 * i); pv[i].var = m3_from(var_body + 9 * i); }
  IMUST x = state_from(state);
  for (int i = 0; i < 225; i++) x.cov[i] = cov[i];
  std::vector<double> tr;
  bool ok = vm->lio_state_estimation(pv, x, &tr);
  state_to(x, state);
  for (int i = 0; i < 225; i++) cov[i] = x.cov[i];
  if (trace && n_trace) { for (size_t i = 0; i < tr.size(); i++) trace[i] = tr[i]; *n_trace = (int)tr.size(); }
  return ok ? 1 : 0;
}

// var_init (voxelslam.hpp:210-234): pnt [n][3] in/out, var [n][9] out
void vso_var_init(int n, double *pnt, const double *ext_pose, double dept_err, double beam_err, double *var) {
  PVec pv(n);
  for (int i = 0; i < n; i++) pv[i].pnt = v3_from(pnt + 3 * i);
  var_init(m3_from(ext_pose), v3_from(ext_pose + 9), pv, dept_err, beam_err);
  for (int i = 0; i < n; i++) { v3_to(pv[i].pnt, pnt + 3 * i); m3_to(pv[i].var, var + 9 * i); }
}
// pvec_update (voxelslam.hpp:242-265): var [n][9] in/out (body -> world), pwld [n][3] out
void vso_pvec_update(int n, const double *pnt, double *var, const double *state, const double *cov, double *pwld) {
  PVec pv(n);
  for (int i = 0; i < n; i++) { pv[i].pnt = v3_from(pnt + 3 * i); pv[i].var = m3_from(var + 9 * i); }
  IMUST x = state_from(state);
  for (int i = 0; i < 225; i++) x.cov[i] = cov[i];
  std::vector<V3> pw;
  pvec_update(pv, x, pw);
  for (int i = 0; i < n; i++) { m3_to(pv[i].var, var + 9 * i); v3_to(pw[i], pwld + 3 * i); }
}

// down_sampling_voxel (tools.hpp:201-238): pnt [n][3] -> out [<=n][3] (float values), count, first index; returns n_out
int vso_down_sampling_voxel(int n, const double *pnt, double voxel_size, double *out, int *count, int *first) {
  std::vector<V3> in(n);
  for (int i = 0; i < n; i++) in[i] = v3_from(pnt + 3 * i);
  std::vector<DsPoint> o;
  down_sampling_voxel(in, voxel_size, o);
  for (size_t i = 0; i < o.size(); i++) {
    out[3 * i] = o[i].x; out[3 * i + 1] = o[i].y; out[3 * i + 2] = o[i].z;
    count[i] = (int)o[i].curvature; first[i] = o[i].first;
  }
  return (int)o.size();
}
int vso_down_sampling_pvec(int n, const double *pnt, const double *var, double voxel_size, double *out, double *vdiag, int *count) {
  std::vector<V3> p(n); std::vector<M3> v(n);
  for (int i = 0; i < n; i++) { p[i] = v3_from(pnt + 3 * i); v[i] = m3_from(var + 9 * i); }
  std::vector<DsPvec> o;
  down_sampling_pvec(p, v, voxel_size, o);
  for (size_t i = 0; i < o.size(); i++) {
    out[3 * i] = o[i].x; out[3 * i + 1] = o[i].y; out[3 * i + 2] = o[i].z;
    vdiag[3 * i] = o[i].nx; vdiag[3 * i + 1] = o[i].ny; vdiag[3 * i + 2] = o[i].nz; count[i] = o[i].count;
  }
  return (int)o.size();
}
int vso_down_sampling_close(int n, const double *pnt, double voxel_size, int *keep) {
  std::vector<V3> p(n);
  for (int i = 0; i < n; i++) p[i] = v3_from(pnt + 3 * i);
  std::vector<int> k;
  down_sampling_close(p, voxel_size, k);
  for (size_t i = 0; i < k.size(); i++) keep[i] = k[i];
  return (int)k.size();
}
// motion_blur point loop (ekf_imu.hpp:137-163): pnt [n][3] in/out (float values), imu_poses [m][22] = t R p v angvel acc
void vso_undistort(int n, double *pnt, const double *curv, int m, const double *imu_poses, const double *end_pose, const double *ext_pose) {
  std::vector<float> pts(3 * (size_t)n), cv(n);
  for (int i = 0; i < 3 * n; i++) pts[i] = (float)pnt[i];
  for (int i = 0; i < n; i++) cv[i] = (float)curv[i];
  std::vector<ImuPose> ip(m);
  for (int j = 0; j < m; j++) {
    const double *q = imu_poses + 22 * j;
    ip[j].t = q[0]; ip[j].R = m3_from(q + 1); ip[j].p = v3_from(q + 10); ip[j].v = v3_from(q + 13); ip[j].angvel = v3_from(q + 16); ip[j].acc = v3_from(q + 19);
  }
  undistort(pts, cv, ip, m3_from(end_pose), v3_from(end_pose + 9), m3_from(ext_pose), v3_from(ext_pose + 9));
  for (int i = 0; i < 3 * n; i++) pnt[i] = pts[i];
}

// ---- hierarchical global BA (gba_oracle.hpp).  cfg13 = gba_voxel_size, gba_min_eigen_value, gba_eigen_value_array[4],
// voxel_size, min_eigen_value, plane_eigen_value_thre[4], max_layer
static GbaCfg gba_cfg_from(const double *c) {
  GbaCfg g;
  g.gba_voxel_size = c[0]; g.gba_min_eigen_value = c[1];
  for (int k = 0; k < 4; k++) { g.gba_eigen_value_array[k] = c[2 + k]; g.plane_eigen_value_thre[k] = c[8 + k]; }
  g.voxel_size = c[6]; g.min_eigen_value = c[7]; g.max_layer = (int)c[12];
  return g;
}
static std::vector<std::vector<V3>> clouds_from(int wdsize, const int *offsets, const double *pnt) {
  std::vector<std::vector<V3>> cl(wdsize);
  for (int i = 0; i < wdsize; i++)
    for (int k = offsets[i]; k < offsets[i + 1]; k++) cl[i].push_back(v3_from(pnt + 3 * (size_t)k));
  return cl;
}
void vso_gba_build(void *fh, int wdsize, const int *offsets, const double *pnt, const double *poses, const double *cfg13) {
  GbaMap map; map.cfg = gba_cfg_from(cfg13);
  std::vector<IMUST> xs = poses_to_states(poses, wdsize);
  auto cl = clouds_from(wdsize, offsets, pnt);
  for (int i = 0; i < wdsize; i++) map.cut_voxel(xs[i], cl[i], i, wdsize);
  LidarFactor *f = (LidarFactor *)fh;
  f->clear();
  map.multi_recut(*f);
}
// edges_out rows: i, j, rot[9], tra[3], v6[6] (20 doubles)
int vso_hba_add_edge(int wdsize, const int *offsets, const double *pnt, double *poses, const double *cfg13, int max_iter, int thread_num,
                     double *edges_out, int *n_edges, double *cloud_out, int *cloud_cnt, int *n_cloud, double *resis_log, int *n_log) {
  std::vector<IMUST> xs = poses_to_states(poses, wdsize);
  auto cl = clouds_from(wdsize, offsets, pnt);
  std::vector<GbaEdge> edges;
  std::vector<DsPoint> cloud;
  std::vector<double> rl;
  int st = hba_add_edge(xs, cl, gba_cfg_from(cfg13), max_iter, thread_num, edges, cloud_out ? &cloud : nullptr, &rl);
  if (st) return st;
  states_to_poses(xs, poses);
  for (size_t e = 0; e < edges.size(); e++) {
    double *o = edges_out + 20 * e;
    o[0] = edges[e].i; o[1] = edges[e].j; m3_to(edges[e].rot, o + 2); v3_to(edges[e].tra, o + 11);
    for (int k = 0; k < 6; k++) o[14 + k] = edges[e].v6[k];
  }
  *n_edges = (int)edges.size();
  if (cloud_out) {
    for (size_t i = 0; i < cloud.size(); i++) { cloud_out[3 * i] = cloud[i].x; cloud_out[3 * i + 1] = cloud[i].y; cloud_out[3 * i + 2] = cloud[i].z; cloud_cnt[i] = (int)cloud[i].curvature; }
    *n_cloud = (int)cloud.size();
  }
  if (resis_log) { for (size_t i = 0; i < rl.size(); i++) resis_log[i] = rl[i]; *n_log = (int)rl.size(); }
  return 0;
}

// ---- kd-tree odometry of the initialisation phase (kd_oracle.hpp)
void *vso_kd_create() { return new KdOdomOracle(); }
void vso_kd_destroy(void *h) { delete (KdOdomOracle *)h; }
int vso_kd_tree_size(void *h) { return (int)(((KdOdomOracle *)h)->tree.size() / 3); }
void vso_kd_tree_points(void *h, double *out) { auto &t = ((KdOdomOracle *)h)->tree; for (size_t i = 0; i < t.size(); i++) out[i] = t[i]; }
int vso_kd_lio_state_estimation(void *h, int n, const double *pnt_body, double *state, double *cov) {
  std::vector<V3> p(n);
  for (int i = 0; i < n; i++) p[i] = v3_from(pnt_body + 3 * i);
  IMUST x = state_from(state);
  for (int i = 0; i < 225; i++) x.cov[i] = cov[i];
  const int it = ((KdOdomOracle *)h)->lio_state_estimation_kdtree(p, x);
  state_to(x, state);
  for (int i = 0; i < 225; i++) cov[i] = x.cov[i];
  return it;
}

double vso_now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

}  // extern "C"
