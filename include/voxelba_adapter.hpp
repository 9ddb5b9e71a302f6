// voxelba_adapter.hpp — header-only C++ adapter that gives libvoxelba.so (include/voxelba.h) the class/method
// surface of the reference so that it drops in behind the ROS odometry node:
//
//   reference (VoxelSLAM/src/voxel_map.hpp)                         adapter
//   -------------------------------------------------------------   -----------------------------------------
//   class LidarFactor            VM:124-339                         vba::LidarFactor
//   class Lidar_BA_Optimizer     VM:342-498  (damping_iter :422)    vba::Lidar_BA_Optimizer
//   class LI_BA_Optimizer        VM:504-714  (damping_iter :624)    vba::LI_BA_Optimizer
//   class LI_BA_OptimizerGravity VM:717-976  (damping_iter :878)    vba::LI_BA_OptimizerGravity
//   cut_voxel / cut_voxel_multi / cut_voxel(fix)  VM:1896/1964/2108 vba::VoxelMap::cut_voxel[_multi|_fix]
//   multi_recut / multi_margi    VS:1682 / VS:1590                  vba::VoxelMap::multi_recut / multi_margi
//
// The reference's types are Eigen-based (tools.hpp:4).  This header compiles without Eigen (plain-array structs that
// mirror PointCluster / IMUST / IMU_PRE field for field); when <Eigen/Core> is available the Eigen-typed overloads
// below are enabled so call sites such as voxelslam.cpp:1969-1970 compile unchanged after `using namespace vba;`.
#pragma once
#include "voxelba.h"
#include <cstring>
#include <deque>
#include <stdexcept>
#include <string>
#include <vector>

#if defined(__has_include)
#if __has_include(<Eigen/Core>)
#include <Eigen/Core>
#define VBA_ADAPTER_HAVE_EIGEN 1
#endif
#endif

namespace vba {

struct PointCluster {  // tools.hpp:304-365 (P symmetric 3x3 row-major, v, N)
  double P[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
  double v[3] = {0, 0, 0};
  int N = 0;
  void pack(double *c) const { c[0] = P[0]; c[1] = P[3]; c[2] = P[6]; c[3] = P[4]; c[4] = P[7]; c[5] = P[8]; c[6] = v[0]; c[7] = v[1]; c[8] = v[2]; c[9] = N; }
  void unpack(const double *c) {
    P[0] = c[0]; P[1] = P[3] = c[1]; P[2] = P[6] = c[2]; P[4] = c[3]; P[5] = P[7] = c[4]; P[8] = c[5];
    v[0] = c[6]; v[1] = c[7]; v[2] = c[8]; N = (int)c[9];
  }
};

struct IMUST {  // tools.hpp:135-199: exactly the `state` layout of voxelba.h followed by cov
  double t = 0, R[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1}, p[3] = {0, 0, 0}, v[3] = {0, 0, 0}, bg[3] = {0, 0, 0}, ba[3] = {0, 0, 0}, g[3] = {0, 0, 0};
  double cov[225] = {0};
};

struct IMU_PRE {  // preintegration.hpp:11-331: exactly the `imu_pre` layout of voxelba.h
  double f[VBA_IMU_PRE_LEN];
  double bg0[3] = {0, 0, 0}, ba0[3] = {0, 0, 0};
  IMU_PRE(const double *bg1 = nullptr, const double *ba1 = nullptr) {   // IMU_PRE(bg, ba) PI:32-48
    std::memset(f, 0, sizeof(f));
    f[0] = f[4] = f[8] = 1.0;
    if (bg1) { std::memcpy(f + 15, bg1, 24); std::memcpy(bg0, bg1, 24); }
    if (ba1) { std::memcpy(f + 18, ba1, 24); std::memcpy(ba0, ba1, 24); }
  }
  // push_imu(deque<sensor_msgs::Imu::Ptr>&) PI:50-73 with the samples as arrays t[n], gyr[n][3], acc[n][3] and the noise globals
  // noiseMeas / noiseWalk (PI:8) as diagonals
  void push_imu(int n, const double *t, const double *gyr, const double *acc, const double *noise_meas6, const double *noise_walk6,
                double scale_gravity = 1.0) {
    const int st = vba_imu_preintegrate(n, t, gyr, acc, bg0, ba0, noise_meas6, noise_walk6, scale_gravity, f);
    if (st != VBA_OK) throw std::runtime_error(std::string("libvoxelba: push_imu: ") + vba_status_string(st));
  }
};

inline void check(vba_ctx *c, int st) {
  if (st != VBA_OK) throw std::runtime_error(std::string("libvoxelba: ") + vba_status_string(st) + " | " + (c ? vba_last_error(c) : ""));
}

// One context = one HIP stream + HBM factor store + device voxel map.  The reference's globals (VM:98-104, VM:500) are
// the fields of vba_options.
class Context {
 public:
  explicit Context(const vba_options &o) { check(nullptr, vba_create(&o, &c_)); }
  ~Context() { vba_destroy(c_); }
  Context(const Context &) = delete;
  Context &operator=(const Context &) = delete;
  vba_ctx *get() const { return c_; }
 private:
  vba_ctx *c_ = nullptr;
};

// class LidarFactor (VM:124-339).  The voxel data lives in HBM; eig_values / eig_vectors / pcr_adds are fetched on demand.
class LidarFactor {
 public:
  int win_size;
  LidarFactor(Context &ctx, int w) : win_size(w), c_(ctx.get()) {}
  void push_voxel(const std::vector<PointCluster> &vec_orig, const PointCluster &fix, double coe, const double *eig_value3,
                  const double *eig_vector9, const PointCluster &pcr_add) {   // VM:139-147
    std::vector<double> cl((size_t)win_size * 10);
    for (int i = 0; i < win_size; i++) vec_orig[i].pack(&cl[(size_t)i * 10]);
    double fx[10], pa[10];
    fix.pack(fx); pcr_add.pack(pa);
    check(c_, vba_factor_push_voxels(c_, 1, cl.data(), fx, &coe, eig_value3, eig_vector9, pa));
  }
  size_t size() const { return (size_t)vba_factor_size(c_); }   // plvec_voxels.size()
  void acc_evaluate2(const std::vector<IMUST> &xs, int head, int end, double *Hess, double *JacT, double &residual) {   // VM:150-282
    std::vector<double> poses = poses_of(xs);
    check(c_, vba_factor_acc_evaluate2(c_, poses.data(), head, end, Hess, JacT, &residual));
  }
  void evaluate_only_residual(const std::vector<IMUST> &xs, int head, int end, double &residual) {   // VM:285-325
    std::vector<double> poses = poses_of(xs);
    check(c_, vba_factor_evaluate_only_residual(c_, poses.data(), head, end, &residual));
  }
  void read_back(std::vector<double> &eig_values, std::vector<double> &eig_vectors, std::vector<PointCluster> &pcr_adds) {
    const size_t n = size();
    eig_values.resize(n * 3); eig_vectors.resize(n * 9);
    std::vector<double> pa(n * 10);
    check(c_, vba_factor_read_back(c_, eig_values.data(), eig_vectors.data(), pa.data()));
    pcr_adds.resize(n);
    for (size_t a = 0; a < n; a++) pcr_adds[a].unpack(&pa[a * 10]);
  }
  void clear() { check(c_, vba_factor_clear(c_)); }   // VM:328-336
  vba_ctx *ctx() const { return c_; }
  static std::vector<double> poses_of(const std::vector<IMUST> &xs) {
    std::vector<double> p(xs.size() * 12);
    for (size_t i = 0; i < xs.size(); i++) { std::memcpy(&p[i * 12], xs[i].R, 72); std::memcpy(&p[i * 12 + 9], xs[i].p, 24); }
    return p;
  }
 private:
  vba_ctx *c_;
};

// class Lidar_BA_Optimizer (VM:342-498)
class Lidar_BA_Optimizer {
 public:
  int win_size = 0, jac_leng = 0, thd_num = 2;
  // bool damping_iter(vector<IMUST>&, LidarFactor&, MatrixXd *hess, vector<double> &resis, int max_iter = 3, bool is_display = false)  VM:422
  bool damping_iter(std::vector<IMUST> &x_stats, LidarFactor &voxhess, std::vector<double> *hess, std::vector<double> &resis, int max_iter = 3,
                    bool /*is_display*/ = false) {
    win_size = voxhess.win_size; jac_leng = 6 * win_size;
    std::vector<double> poses = LidarFactor::poses_of(x_stats);
    if (hess) hess->assign((size_t)jac_leng * jac_leng, 0.0);
    double r2[2] = {0, 0};
    int conv = 0;
    check(voxhess.ctx(), vba_lidar_ba_damping_iter(voxhess.ctx(), poses.data(), hess ? hess->data() : nullptr, r2, max_iter, thd_num, &conv));
    for (size_t i = 0; i < x_stats.size(); i++) { std::memcpy(x_stats[i].R, &poses[i * 12], 72); std::memcpy(x_stats[i].p, &poses[i * 12 + 9], 24); }
    resis.push_back(r2[0]); resis.push_back(r2[1]);
    return conv != 0;
  }
};

namespace detail {
inline void li_call(std::vector<IMUST> &x_stats, LidarFactor &voxhess, std::deque<IMU_PRE *> &imus_factor, std::vector<double> *resis,
                    std::vector<double> *hess, int gravity, int max_iter) {
  const int W = voxhess.win_size, n = 15 * W + (gravity ? 3 : 0);
  std::vector<double> st((size_t)W * 25), im((size_t)(W - 1) * VBA_IMU_PRE_LEN);
  for (int i = 0; i < W; i++) std::memcpy(&st[(size_t)i * 25], &x_stats[i].t, 25 * sizeof(double));
  for (int i = 0; i < W - 1; i++) std::memcpy(&im[(size_t)i * VBA_IMU_PRE_LEN], imus_factor[i]->f, sizeof(imus_factor[i]->f));
  if (hess) hess->assign((size_t)n * n, 0.0);
  double r2[2] = {0, 0};
  check(voxhess.ctx(), vba_li_ba_damping_iter(voxhess.ctx(), st.data(), im.data(), gravity, max_iter, hess ? hess->data() : nullptr, r2));
  for (int i = 0; i < W; i++) std::memcpy(&x_stats[i].t, &st[(size_t)i * 25], 25 * sizeof(double));
  for (int i = 0; i < W - 1; i++) std::memcpy(imus_factor[i]->f, &im[(size_t)i * VBA_IMU_PRE_LEN], sizeof(imus_factor[i]->f));
  if (gravity && resis) { resis->push_back(r2[0]); resis->push_back(r2[1]); }
}
}  // namespace detail

// class LI_BA_Optimizer (VM:504-714): void damping_iter(x_stats, voxhess, imus_factor, MatrixXd *hess)  VM:624
class LI_BA_Optimizer {
 public:
  void damping_iter(std::vector<IMUST> &x_stats, LidarFactor &voxhess, std::deque<IMU_PRE *> &imus_factor, std::vector<double> *hess) {
    detail::li_call(x_stats, voxhess, imus_factor, nullptr, hess, 0, 3);
  }
};
// class LI_BA_OptimizerGravity (VM:717-976): void damping_iter(x_stats, voxhess, imus_factor, resis, hess, max_iter = 2)  VM:878
class LI_BA_OptimizerGravity {
 public:
  void damping_iter(std::vector<IMUST> &x_stats, LidarFactor &voxhess, std::deque<IMU_PRE *> &imus_factor, std::vector<double> &resis,
                    std::vector<double> *hess, int max_iter = 2) {
    detail::li_call(x_stats, voxhess, imus_factor, &resis, hess, 1, max_iter);
  }
};

// Voxel map: unordered_map<VOXEL_LOC, OctoTree*> surf_map / surf_map_slide + the free functions that operate on them.
struct pointVar { double pnt[3]; double var[9]; };   // VM:18-34
typedef std::vector<pointVar> PVec;

class VoxelMap {
 public:
  explicit VoxelMap(Context &ctx) : c_(ctx.get()) {}
  // cut_voxel(feat_map, pvec, win_count, feat_tem_map, wdsize, pwld, sws) VM:1896 — pwld = R p + t is recomputed on the device from `x`
  void cut_voxel(const PVec &pvec, int win_count, const IMUST &x, bool with_var = true) { insert(pvec, win_count, x, with_var, 0); }
  // cut_voxel_multi(...) VM:1964
  void cut_voxel_multi(const PVec &pvec, int win_count, const IMUST &x, bool with_var = true) { insert(pvec, win_count, x, with_var, 1); }
  // pvec_update(pptr, x_curr, pwld) VH:242-265 + cut_voxel_multi VM:1964 fused on the device: pvec holds BODY-frame points and
  // covariances (as var_init leaves them), x.cov the state covariance whose rotation / translation blocks inflate them
  void pvec_update_cut_voxel_multi(const PVec &pvec, int win_count, const IMUST &x) {
    const size_t n = pvec.size();
    std::vector<double> p(n * 3), v(n * 9);
    for (size_t i = 0; i < n; i++) { std::memcpy(&p[i * 3], pvec[i].pnt, 24); std::memcpy(&v[i * 9], pvec[i].var, 72); }
    double pose[12];
    std::memcpy(pose, x.R, 72); std::memcpy(pose + 9, x.p, 24);
    check(c_, vba_map_pvec_update_cut_voxel(c_, win_count, (int)n, p.data(), v.data(), pose, x.cov, 1));
  }
  // cut_voxel(feat_map, PVec&, wdsize, jour) VM:2108
  void cut_voxel_fix(const PVec &pvec, double jour) {
    std::vector<double> p(pvec.size() * 3);
    for (size_t i = 0; i < pvec.size(); i++) std::memcpy(&p[i * 3], pvec[i].pnt, 24);
    check(c_, vba_map_cut_voxel_fix(c_, (int)pvec.size(), p.data(), jour));
  }
  // multi_recut(feat_map, win_count, xs, voxopt, sws) VS:1682 (fills the context's LidarFactor store)
  void multi_recut(int win_count, const std::vector<IMUST> &xs, bool multi = true) {
    std::vector<double> poses = LidarFactor::poses_of(xs);
    check(c_, vba_map_recut(c_, win_count, poses.data(), multi ? 1 : 0));
  }
  // multi_margi(feat_map, jour, win_count, xs, voxopt, sw) VS:1590
  void multi_margi(double jour, int win_count, const std::vector<IMUST> &xs) {
    std::vector<double> poses = LidarFactor::poses_of(xs);
    check(c_, vba_map_margi(c_, win_count, poses.data(), jour));
  }
  void slide(int mgsize) { check(c_, vba_map_slide(c_, mgsize)); }   // VS:2014-2019
  void reset() { check(c_, vba_map_reset(c_)); }
  size_t size() const { return (size_t)vba_map_num_roots(c_); }
  size_t slide_size() const { return (size_t)vba_map_num_slide_roots(c_); }
 private:
  void insert(const PVec &pvec, int win_count, const IMUST &x, bool with_var, int multi) {
    const size_t n = pvec.size();
    std::vector<double> p(n * 3), v(with_var ? n * 9 : 0);
    for (size_t i = 0; i < n; i++) { std::memcpy(&p[i * 3], pvec[i].pnt, 24); if (with_var) std::memcpy(&v[i * 9], pvec[i].var, 72); }
    double pose[12];
    std::memcpy(pose, x.R, 72); std::memcpy(pose + 9, x.p, 24);
    check(c_, vba_map_cut_voxel(c_, win_count, (int)n, p.data(), with_var ? v.data() : nullptr, pose, multi));
  }
  vba_ctx *c_;
};

// ---- scan pre-processing and hierarchical global BA (free functions of the reference: tools.hpp / voxelslam.cpp)
struct XYZ { float x, y, z; };   // the coordinates of a pcl::PointXYZINormal

// down_sampling_voxel(pl_feat, voxel_size) TL:201-238: pl is replaced by the centroids; counts[i] = the `curvature` field
// after the call, first[i] = index (in the input) of the point whose other fields the reference keeps.
inline void down_sampling_voxel(Context &ctx, std::vector<XYZ> &pl, double voxel_size, std::vector<int> *counts = nullptr,
                                std::vector<int> *first = nullptr) {
  const int n = (int)pl.size();
  std::vector<double> in((size_t)n * 3), out((size_t)n * 3);
  std::vector<int> cnt(n), fst(n);
  for (int i = 0; i < n; i++) { in[3 * i] = pl[i].x; in[3 * i + 1] = pl[i].y; in[3 * i + 2] = pl[i].z; }
  int m = 0;
  check(ctx.get(), vba_scan_down_sampling_voxel(ctx.get(), n, in.data(), voxel_size, out.data(), cnt.data(), fst.data(), &m));
  pl.resize(m);
  for (int i = 0; i < m; i++) { pl[i].x = (float)out[3 * i]; pl[i].y = (float)out[3 * i + 1]; pl[i].z = (float)out[3 * i + 2]; }
  cnt.resize(m); fst.resize(m);
  if (counts) *counts = cnt;
  if (first) *first = fst;
}

struct GbaEdge { int i, j; double rot[9], tra[3], v6[6]; };   // the arguments of PGO_Edges::push, LR:247

// The optimisation part of VOXEL_SLAM::HBA_add_edge (VS:2858-2951) for one connected keyframe set: xs in/out,
// clouds[i] = keyframe i's points in its own frame.  submap (optional) receives the cloud of VS:2954-2989.
inline std::vector<GbaEdge> HBA_add_edge(Context &ctx, std::vector<IMUST> &xs, const std::vector<std::vector<XYZ>> &clouds, double gba_voxel_size,
                                         double gba_min_eigen_value, const std::vector<double> &gba_eigen_value_array, int max_iter, int thread_num,
                                         std::vector<XYZ> *submap = nullptr) {
  const int W = (int)xs.size();
  std::vector<int> off(W + 1, 0);
  for (int i = 0; i < W; i++) off[i + 1] = off[i] + (int)clouds[i].size();
  std::vector<double> pts((size_t)off[W] * 3), poses = LidarFactor::poses_of(xs), edges((size_t)(W * (W - 1) / 2 + 1) * 20);
  for (int i = 0; i < W; i++)
    for (size_t k = 0; k < clouds[i].size(); k++) {
      double *q = &pts[((size_t)off[i] + k) * 3];
      q[0] = clouds[i][k].x; q[1] = clouds[i][k].y; q[2] = clouds[i][k].z;
    }
  double eig[4] = {0, 0, 0, 0};
  for (size_t k = 0; k < 4 && k < gba_eigen_value_array.size(); k++) eig[k] = gba_eigen_value_array[k];
  std::vector<double> cloud(submap ? pts.size() : 0);
  std::vector<int> ccnt(submap ? (size_t)off[W] : 0);
  int ne = 0, nc = 0;
  check(ctx.get(), vba_hba_add_edge(ctx.get(), W, off.data(), pts.data(), poses.data(), gba_voxel_size, gba_min_eigen_value, eig, max_iter, thread_num,
                                    edges.data(), &ne, submap ? cloud.data() : nullptr, submap ? ccnt.data() : nullptr, submap ? &nc : nullptr, nullptr, nullptr));
  for (int i = 0; i < W; i++) { std::memcpy(xs[i].R, &poses[12 * i], 72); std::memcpy(xs[i].p, &poses[12 * i + 9], 24); }
  std::vector<GbaEdge> out(ne);
  for (int e = 0; e < ne; e++) {
    const double *q = &edges[(size_t)e * 20];
    out[e].i = (int)q[0]; out[e].j = (int)q[1];
    std::memcpy(out[e].rot, q + 2, 72); std::memcpy(out[e].tra, q + 11, 24); std::memcpy(out[e].v6, q + 14, 48);
  }
  if (submap) { submap->resize(nc); for (int k = 0; k < nc; k++) (*submap)[k] = XYZ{(float)cloud[3 * k], (float)cloud[3 * k + 1], (float)cloud[3 * k + 2]}; }
  return out;
}

// VOXEL_SLAM::lio_state_estimation_kdtree(pptr) VS:1102-1252: x_curr (state + cov) in/out, the point-cloud map (pl_tree) lives in
// the context.  Returns the number of EKF iterations (0 while the map is only being seeded).
inline int lio_state_estimation_kdtree(Context &ctx, const std::vector<pointVar> &pvec, IMUST &x_curr) {
  const int n = (int)pvec.size();
  std::vector<double> p((size_t)n * 3);
  for (int i = 0; i < n; i++) { p[3 * i] = (float)pvec[i].pnt[0]; p[3 * i + 1] = (float)pvec[i].pnt[1]; p[3 * i + 2] = (float)pvec[i].pnt[2]; }   // PointType is float (VS:1143-1146)
  int iters = 0;
  check(ctx.get(), vba_odom_lio_state_estimation_kdtree(ctx.get(), n, p.data(), &x_curr.t, x_curr.cov, &iters));
  return iters;
}

// FileReaderWriter::save_pcd / save_pose (VS:166-204), pcl::io::loadPCDFile (VS:340), read_lidarstate (VH:268-307)
inline void save_pcd(const std::vector<pointVar> &pvec, int count, const std::string &savename) {
  std::vector<double> p(pvec.size() * 3);
  for (size_t i = 0; i < pvec.size(); i++) std::memcpy(&p[3 * i], pvec[i].pnt, 24);
  const std::string path = savename + "/" + std::to_string(count) + ".pcd";
  if (vba_io_save_pcd(path.c_str(), (int)pvec.size(), p.data())) throw std::runtime_error("save_pcd: " + path);
}
inline std::vector<XYZ> load_pcd(const std::string &path) {
  int n = 0;
  int st = vba_io_load_pcd(path.c_str(), 0, nullptr, nullptr, &n);
  if (st != VBA_OK && st != VBA_ERR_CAPACITY) throw std::runtime_error("load_pcd: " + path);
  std::vector<double> p((size_t)(n > 0 ? n : 1) * 3);
  if (vba_io_load_pcd(path.c_str(), n, p.data(), nullptr, &n)) throw std::runtime_error("load_pcd: " + path);
  std::vector<XYZ> out(n);
  for (int i = 0; i < n; i++) out[i] = XYZ{(float)p[3 * i], (float)p[3 * i + 1], (float)p[3 * i + 2]};
  return out;
}
struct ScanPoseRec { IMUST x; double v6[6]; };   // ScanPose (LR:17-27) without the point pointer
inline void save_pose(const std::vector<ScanPoseRec> &bbuf, const std::string &path) {
  std::vector<double> st(bbuf.size() * 25), v6(bbuf.size() * 6);
  for (size_t i = 0; i < bbuf.size(); i++) { std::memcpy(&st[25 * i], &bbuf[i].x.t, 200); std::memcpy(&v6[6 * i], bbuf[i].v6, 48); }
  if (vba_io_save_pose(path.c_str(), (int)bbuf.size(), st.data(), v6.data())) throw std::runtime_error("save_pose: " + path);
}
inline std::vector<ScanPoseRec> read_lidarstate(const std::string &filename) {
  int n = 0;
  int st = vba_io_read_lidarstate(filename.c_str(), 0, nullptr, nullptr, &n);
  if (st != VBA_OK && st != VBA_ERR_CAPACITY) throw std::runtime_error("read_lidarstate: " + filename);   // the reference exits (VH:271-275)
  std::vector<double> s((size_t)(n > 0 ? n : 1) * 25), v6((size_t)(n > 0 ? n : 1) * 6);
  if (vba_io_read_lidarstate(filename.c_str(), n, s.data(), v6.data(), &n)) throw std::runtime_error("read_lidarstate: " + filename);
  std::vector<ScanPoseRec> out(n);
  for (int i = 0; i < n; i++) {
    std::memcpy(&out[i].x.t, &s[25 * (size_t)i], 200); std::memcpy(out[i].v6, &v6[6 * (size_t)i], 48);
    for (int k = 0; k < 225; k++) out[i].x.cov[k] = 0.0;
    for (int k = 0; k < 15; k++) out[i].x.cov[16 * k] = k < 9 ? 1e-4 : 1e-5;       // IMUST::setZero (TL:188-197)
  }
  return out;
}

#ifdef VBA_ADAPTER_HAVE_EIGEN
// Eigen-typed conveniences so reference call sites keep their argument types (Eigen is column-major: converted here).
inline void to_rowmajor3(const Eigen::Matrix3d &M, double *r) { for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) r[3 * i + j] = M(i, j); }
inline Eigen::MatrixXd to_eigen(const std::vector<double> &h, int n) {
  Eigen::MatrixXd M(n, n);
  for (int i = 0; i < n; i++) for (int j = 0; j < n; j++) M(i, j) = h[(size_t)i * n + j];
  return M;
}
#endif

}  // namespace vba
