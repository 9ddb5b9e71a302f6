// TEST INFRASTRUCTURE — CPU restatement of the hierarchical global BA building blocks.
// Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may use anything under oracle/.
// Parity unpinned against the reference binary (PCL/ROS/Eigen/GTSAM absent here).
//   OctreeGBA                 loop_refine.hpp:273-481   (cut_voxel 439-479, recut 354-400, subdivide 320-352)
//   OctreeGBA_multi_recut     loop_refine.hpp:483-537   (thread split only changes the order voxels are appended)
//   HBA_add_edge              voxelslam.cpp:2822-3015   (octree rebuild + Lidar_BA_Optimizer loop, edges from the Hessian)
#pragma once
#include "ba_oracle.hpp"
#include "map_oracle.hpp"
#include "scan_oracle.hpp"
#include <map>
#include <memory>

namespace vso {

struct GbaCfg {
  double gba_voxel_size = 1.0, gba_min_eigen_value = 0.01;
  double gba_eigen_value_array[4] = {0.25, 0.25, 0.25, 0.25};   // already inverted (VS:3022-3024)
  double voxel_size = 1.0, min_eigen_value = 0.0025;
  double plane_eigen_value_thre[4] = {0.25, 0.25, 0.25, 0.25};  // already inverted (VS:930-931)
  int max_layer = 2;
};

struct OctreeGBA {
  std::vector<std::vector<V3>> locals, worlds;   // LR:276
  PointCluster pcr_add;
  int layer, octo_state = 0, wdsize;
  std::unique_ptr<OctreeGBA> leaves[8];
  double voxel_center[3];
  float quater_length;
  bool is_plane = false;
  const GbaCfg *cfg;

  OctreeGBA(int l, int w, const GbaCfg *c) : locals(w), worlds(w), layer(l), wdsize(w), cfg(c) {}

  bool plane_judge(const V3 &ev) const {  // LR:310-314
    return ev[0] < cfg->gba_min_eigen_value && (ev[0] / ev[2]) < cfg->gba_eigen_value_array[layer];
  }
  void push(int ord, const V3 &local, const V3 &world) {  // LR:316-321
    locals[ord].push_back(local); worlds[ord].push_back(world); pcr_add.push(world);
  }
  void subdivide() {  // LR:323-356
    for (int i = 0; i < wdsize; i++)
      for (size_t j = 0; j < locals[i].size(); j++) {
        const V3 &pw = worlds[i][j];
        int xyz[3] = {0, 0, 0};
        for (int k = 0; k < 3; k++) if (pw[k] > voxel_center[k]) xyz[k] = 1;
        const int leafnum = 4 * xyz[0] + 2 * xyz[1] + xyz[2];
        if (!leaves[leafnum]) {
          leaves[leafnum].reset(new OctreeGBA(layer + 1, wdsize, cfg));
          for (int k = 0; k < 3; k++) leaves[leafnum]->voxel_center[k] = voxel_center[k] + (2 * xyz[k] - 1) * quater_length;
          leaves[leafnum]->quater_length = quater_length / 2;
        }
        leaves[leafnum]->push(i, locals[i][j], pw);
      }
  }
  void recut(LidarFactor &vox_opt) {  // LR:358-404
    if (pcr_add.N <= 10) return;
    V3 eig_value; M3 eig_vector;
    eig3_sym(pcr_add.cov(), eig_value, eig_vector);
    is_plane = plane_judge(eig_value);
    if (is_plane) {
      if (pcr_add.N < 10) return;
      int exi = 0;
      for (int i = 0; i < wdsize; i++) if (!locals[i].empty()) exi++;
      if (exi <= 1) return;
      if (eig_value[0] / eig_value[1] > 0.12) return;
      std::vector<PointCluster> pcrs(wdsize);
      for (int i = 0; i < wdsize; i++) { pcrs[i].clear(); for (const V3 &v : locals[i]) pcrs[i].push(v); }
      PointCluster pcr_fix; pcr_fix.clear();
      vox_opt.push_voxel(pcrs, pcr_fix, 1.0, eig_value, eig_vector, pcr_add);
      return;
    } else if (layer >= cfg->max_layer) {
      return;
    } else {
      subdivide();
      octo_state = 1;
    }
    for (int i = 0; i < 8; i++) if (leaves[i]) leaves[i]->recut(vox_opt);
  }
};

struct GbaMap {
  GbaCfg cfg;
  // insertion-ordered (the reference's unordered_map order is unspecified; only the order of the voxel list depends on it)
  std::unordered_map<VOXEL_LOC, int, VoxelLocHash> index;
  std::vector<std::unique_ptr<OctreeGBA>> roots;

  void cut_voxel(const IMUST &xc, const std::vector<V3> &pl, int win_count, int wdsize) {  // LR:439-479
    for (const V3 &local : pl) {
      V3 world = xc.R * local + xc.p;
      float loc[3];
      for (int j = 0; j < 3; j++) {
        loc[j] = world[j] / cfg.gba_voxel_size;
        if (loc[j] < 0) loc[j] -= 1;
      }
      VOXEL_LOC position(loc[0], loc[1], loc[2]);
      auto it = index.find(position);
      if (it != index.end()) {
        roots[it->second]->push(win_count, local, world);
      } else {
        std::unique_ptr<OctreeGBA> ot(new OctreeGBA(0, wdsize, &cfg));
        ot->push(win_count, local, world);
        ot->voxel_center[0] = (0.5 + position.x) * cfg.gba_voxel_size;
        ot->voxel_center[1] = (0.5 + position.y) * cfg.gba_voxel_size;
        ot->voxel_center[2] = (0.5 + position.z) * cfg.gba_voxel_size;
        ot->quater_length = cfg.gba_voxel_size / 4.0;
        index[position] = (int)roots.size();
        roots.push_back(std::move(ot));
      }
    }
  }
  void multi_recut(LidarFactor &voxhess) {  // LR:483-537
    for (auto &r : roots) r->recut(voxhess);
    roots.clear(); index.clear();
  }
};

struct GbaEdge { int i, j; M3 rot; V3 tra; double v6[6]; };

// voxelslam.cpp:2822-3015 on one connected set of keyframes (the map filter VS:2830-2856 is caller bookkeeping).
// clouds[i]: keyframe i's points in its own frame (PCL float values).  Returns -1 where the reference exits
// ("Too Less Voxel").  cloud_out (when non-null) = the submap cloud of VS:2957-2989 in first-occurrence order.
inline int hba_add_edge(std::vector<IMUST> &xs, const std::vector<std::vector<V3>> &clouds, const GbaCfg &cfg0, int max_iter, int thread_num,
                        std::vector<GbaEdge> &edges, std::vector<DsPoint> *cloud_out, std::vector<double> *resis_log = nullptr) {
  const int wdsize = (int)xs.size();
  GbaCfg cfg = cfg0;
  MatX hess(6 * wdsize, 6 * wdsize);
  const int up = 4;
  int converge_flag = 0;
  double converge_thre = 0.05;
  for (int iterCnt = 0; iterCnt < max_iter; iterCnt++) {
    if (converge_flag == 1 || iterCnt == max_iter - 1) {  // VS:2871-2881
      cfg.gba_voxel_size = cfg0.voxel_size;
      for (int k = 0; k < 4; k++) cfg.gba_eigen_value_array[k] = cfg0.plane_eigen_value_thre[k];
      cfg.gba_min_eigen_value = cfg0.min_eigen_value;
    }
    GbaMap map; map.cfg = cfg;
    for (int i = 0; i < wdsize; i++) map.cut_voxel(xs[i], clouds[i], i, wdsize);
    LidarFactor voxhess(wdsize);
    map.multi_recut(voxhess);
    Lidar_BA_Optimizer opt_lsv;
    opt_lsv.thd_num = thread_num;
    std::vector<double> resis;
    int status = 0;
    bool is_converge = opt_lsv.damping_iter(xs, voxhess, &hess, resis, up, &status);
    if (status) return -1;
    if (resis_log) { resis_log->push_back(resis[0]); resis_log->push_back(resis[1]); }
    if ((std::fabs(resis[0] - resis[1]) / resis[0] < converge_thre && is_converge) || (iterCnt == max_iter - 2 && converge_flag == 0)) {
      converge_thre = 0.01;
      if (converge_flag == 0) converge_flag = 1;
      else if (converge_flag == 1) break;
    }
  }
  edges.clear();
  for (int i = 0; i < wdsize - 1; i++)
    for (int j = i + 1; j < wdsize; j++) {  // VS:2926-2951
      bool isAdd = true;
      GbaEdge e; e.i = i; e.j = j;
      for (int k = 0; k < 6; k++) {
        const double hc = std::fabs(hess(6 * i + k, 6 * j + k));
        if (hc < 1e-6) { isAdd = false; break; }
        e.v6[k] = 1.0 / hc;
      }
      if (isAdd) {
        e.tra = xs[i].R.transpose() * (xs[j].p - xs[i].p);
        e.rot = xs[i].R.transpose() * xs[j].R;
        edges.push_back(e);
      }
    }
  if (cloud_out) {  // VS:2954-2989
    std::vector<V3> pl;
    const IMUST &xc = xs[0];
    for (int i = 0; i < wdsize; i++) {
      V3 dp = xc.R.transpose() * (xs[i].p - xc.p);
      M3 dR = xc.R.transpose() * xs[i].R;
      for (const V3 &ap : clouds[i]) {
        V3 q = dR * ap + dp;
        pl.push_back(v3((double)(float)q[0], (double)(float)q[1], (double)(float)q[2]));
      }
    }
    down_sampling_voxel(pl, cfg0.voxel_size / 8, *cloud_out);
  }
  return 0;
}

}  // namespace vso
