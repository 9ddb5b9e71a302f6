// TEST INFRASTRUCTURE — CPU restatement of the scan pre-processing that feeds the hot path.
// Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may use anything under oracle/.
// Parity unpinned against the reference binary (it cannot be built here: PCL/ROS/Eigen absent); pinned by the
// properties in tests/test_oracle_kat.py (centroid identity, zero-motion identity).
//   down_sampling_voxel   tools.hpp:201-238   (PCL PointXYZINormal: float x,y,z, float curvature = running count)
//   down_sampling_pvec    voxel_map.hpp:39-83    (pointVar: double point, 3x3 covariance; running means in double)
//   down_sampling_close   tools.hpp:240-298
//   motion_blur, point loop   ekf_imu.hpp:137-163 (points sorted by curvature, voxelslam.hpp:92-95)
#pragma once
#include "map_oracle.hpp"
#include <unordered_map>

namespace vso {

struct DsPoint { float x, y, z, curvature; int first; };

// tools.hpp:201-238.  Output order = first occurrence (the reference's order is std::unordered_map iteration order, which
// is unspecified; callers never depend on it).
inline void down_sampling_voxel(const std::vector<V3> &in, double voxel_size, std::vector<DsPoint> &out) {
  out.clear();
  if (voxel_size < 0.001) {                      // TL:203: the cloud is left untouched
    for (size_t i = 0; i < in.size(); i++) out.push_back({(float)in[i][0], (float)in[i][1], (float)in[i][2], 0.f, (int)i});
    return;
  }
  std::unordered_map<VOXEL_LOC, int, VoxelLocHash> feat_map;
  float loc_xyz[3];
  for (size_t i = 0; i < in.size(); i++) {
    const float pc[3] = {(float)in[i][0], (float)in[i][1], (float)in[i][2]};
    for (int j = 0; j < 3; j++) {
      loc_xyz[j] = pc[j] / voxel_size;
      if (loc_xyz[j] < 0) loc_xyz[j] -= 1.0;
    }
    VOXEL_LOC position((int64_t)loc_xyz[0], (int64_t)loc_xyz[1], (int64_t)loc_xyz[2]);
    auto it = feat_map.find(position);
    if (it == feat_map.end()) {
      feat_map[position] = (int)out.size();
      out.push_back({pc[0], pc[1], pc[2], 1.f, (int)i});
    } else {
      DsPoint &pp = out[it->second];
      pp.x = (pp.x * pp.curvature + pc[0]) / (pp.curvature + 1);
      pp.y = (pp.y * pp.curvature + pc[1]) / (pp.curvature + 1);
      pp.z = (pp.z * pp.curvature + pc[2]) / (pp.curvature + 1);
      pp.curvature += 1;
    }
  }
}

// voxel_map.hpp:39-83.  Output (first-occurrence order): PCL point x,y,z and normal_x/y/z = diagonal of the mean covariance.
struct DsPvec { float x, y, z, nx, ny, nz; int count; };
inline void down_sampling_pvec(const std::vector<V3> &pnt, const std::vector<M3> &var, double voxel_size, std::vector<DsPvec> &out) {
  struct Acc { V3 p; M3 v; int n; };
  std::unordered_map<VOXEL_LOC, int, VoxelLocHash> feat_map;
  std::vector<Acc> acc;
  float loc_xyz[3];
  for (size_t i = 0; i < pnt.size(); i++) {
    for (int j = 0; j < 3; j++) {
      loc_xyz[j] = pnt[i][j] / voxel_size;
      if (loc_xyz[j] < 0) loc_xyz[j] -= 1.0;
    }
    VOXEL_LOC position((int64_t)loc_xyz[0], (int64_t)loc_xyz[1], (int64_t)loc_xyz[2]);
    auto it = feat_map.find(position);
    if (it == feat_map.end()) { feat_map[position] = (int)acc.size(); acc.push_back({pnt[i], var[i], 1}); }
    else {
      Acc &pp = acc[it->second];
      pp.p = (pp.p * (double)pp.n + pnt[i]) / (double)(pp.n + 1);
      pp.v = (pp.v * (double)pp.n + var[i]) / (double)(pp.n + 1);
      pp.n += 1;
    }
  }
  out.clear();
  for (const Acc &a : acc) out.push_back({(float)a.p[0], (float)a.p[1], (float)a.p[2], (float)a.v(0, 0), (float)a.v(1, 1), (float)a.v(2, 2), a.n});
}

// tools.hpp:240-298: index of the kept point per voxel, first-occurrence order of the voxels
inline void down_sampling_close(const std::vector<V3> &in, double voxel_size, std::vector<int> &keep) {
  keep.clear();
  if (voxel_size < 0.001) { for (size_t i = 0; i < in.size(); i++) keep.push_back((int)i); return; }
  std::unordered_map<VOXEL_LOC, int, VoxelLocHash> feat_map;
  std::vector<std::vector<int>> groups;
  float loc_xyz[3];
  for (size_t i = 0; i < in.size(); i++) {
    const float pc[3] = {(float)in[i][0], (float)in[i][1], (float)in[i][2]};
    for (int j = 0; j < 3; j++) {
      loc_xyz[j] = pc[j] / voxel_size;
      if (loc_xyz[j] < 0) loc_xyz[j] -= 1.0;
    }
    VOXEL_LOC position((int64_t)loc_xyz[0], (int64_t)loc_xyz[1], (int64_t)loc_xyz[2]);
    auto it = feat_map.find(position);
    if (it == feat_map.end()) { feat_map[position] = (int)groups.size(); groups.push_back({(int)i}); }
    else groups[it->second].push_back((int)i);
  }
  for (const auto &g : groups) {
    float bx = (float)in[g[0]][0], by = (float)in[g[0]][1], bz = (float)in[g[0]][2];
    const int plsize = (int)g.size();
    for (int i = 1; i < plsize; i++) { bx += (float)in[g[i]][0]; by += (float)in[g[i]][1]; bz += (float)in[g[i]][2]; }
    bx /= plsize; by /= plsize; bz /= plsize;
    double ndis = 100;
    int mnum = 0;
    for (int i = 0; i < plsize; i++) {
      const double xx = bx - (float)in[g[i]][0], yy = by - (float)in[g[i]][1], zz = bz - (float)in[g[i]][2];
      const double dis = xx * xx + yy * yy + zz * zz;
      if (dis < ndis) { mnum = i; ndis = dis; }
    }
    keep.push_back(g[mnum]);
  }
}

struct ImuPose { double t; M3 R; V3 p, v, angvel, acc; };   // IMUST(offt, R_imu, pos_imu, vel_imu, angvel_avr, acc_imu)  EK:87

// ekf_imu.hpp:137-163.  pts: float xyz (in/out), curv: per-point time offset (ascending).
inline void undistort(std::vector<float> &pts, const std::vector<float> &curv, const std::vector<ImuPose> &imu_poses, const M3 &R_end,
                      const V3 &p_end, const M3 &Lid_rot_to_IMU, const V3 &Lid_offset_to_IMU) {
  const int n = (int)curv.size();
  if (n == 0) return;
  int it_pcl = n - 1;
  for (int i = (int)imu_poses.size() - 1; i >= 0; i--) {
    const ImuPose &head = imu_poses[i];
    for (; curv[it_pcl] > head.t; it_pcl--) {
      const double dt = curv[it_pcl] - head.t;
      M3 R_i = head.R * Exp(head.angvel, dt);
      V3 T_ei = head.p + head.v * dt + head.acc * (0.5 * dt * dt) - p_end;
      V3 P_i = v3(pts[3 * it_pcl], pts[3 * it_pcl + 1], pts[3 * it_pcl + 2]);
      V3 P_c = Lid_rot_to_IMU.transpose() * (R_end.transpose() * (R_i * (Lid_rot_to_IMU * P_i + Lid_offset_to_IMU) + T_ei) - Lid_offset_to_IMU);
      pts[3 * it_pcl] = (float)P_c[0]; pts[3 * it_pcl + 1] = (float)P_c[1]; pts[3 * it_pcl + 2] = (float)P_c[2];
      if (it_pcl == 0) break;
    }
  }
}

}  // namespace vso
