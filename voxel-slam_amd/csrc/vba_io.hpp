// Session-store formats either side of the path — host code, no device work:
//   <session>/<i>.pcd          one scan, binary PCD of pcl::PointXYZI written by FileReaderWriter::save_pcd (voxelslam.cpp:166-179),
//                              read back by pcl::io::loadPCDFile in previous_map_read (voxelslam.cpp:337-340);
//   <session>/alidarState.txt  one line per scan, FileReaderWriter::save_pose (voxelslam.cpp:181-204) / read_lidarstate
//                              (voxelslam.hpp:268-307).
// The PCD layout is PCL's published one (pcl/io/pcd_io.h, format v0.7): an ASCII header, then for DATA binary the fields of
// every point packed in FIELDS order.  PCL is not in this image: the layout is restated from the format description, not
// checked against PCL itself ("parity unpinned").
#pragma once
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <sstream>
#include <string>
#include <vector>

namespace vba_io {

// Eigen::Quaterniond(Matrix3d): the branch on the trace, then the largest diagonal element (row-major R)
inline void quat_from_rot(const double *R, double q[4] /* x y z w */) {
  double t = R[0] + R[4] + R[8];
  if (t > 0) {
    t = std::sqrt(t + 1.0);
    q[3] = 0.5 * t;
    t = 0.5 / t;
    q[0] = (R[7] - R[5]) * t; q[1] = (R[2] - R[6]) * t; q[2] = (R[3] - R[1]) * t;
  } else {
    int i = 0;
    if (R[4] > R[0]) i = 1;
    if (R[8] > R[i * 3 + i]) i = 2;
    const int j = (i + 1) % 3, k = (j + 1) % 3;
    t = std::sqrt(R[i * 3 + i] - R[j * 3 + j] - R[k * 3 + k] + 1.0);
    q[i] = 0.5 * t;
    t = 0.5 / t;
    q[3] = (R[k * 3 + j] - R[j * 3 + k]) * t;
    q[j] = (R[j * 3 + i] + R[i * 3 + j]) * t;
    q[k] = (R[k * 3 + i] + R[i * 3 + k]) * t;
  }
}
// Quaterniond::matrix(): no normalisation of the stored coefficients
inline void rot_from_quat(const double q[4], double *R) {
  const double x = q[0], y = q[1], z = q[2], w = q[3];
  const double tx = 2 * x, ty = 2 * y, tz = 2 * z, twx = tx * w, twy = ty * w, twz = tz * w, txx = tx * x, txy = ty * x, txz = tz * x, tyy = ty * y,
               tyz = tz * y, tzz = tz * z;
  R[0] = 1 - (tyy + tzz); R[1] = txy - twz; R[2] = txz + twy;
  R[3] = txy + twz; R[4] = 1 - (txx + tzz); R[5] = tyz - twx;
  R[6] = txz - twy; R[7] = tyz + twx; R[8] = 1 - (txx + tyy);
}

struct PcdHeader {
  std::vector<std::string> fields, type;
  std::vector<int> size, count;
  long points = -1, width = -1, height = 1;
  std::string data;
};

}  // namespace vba_io
