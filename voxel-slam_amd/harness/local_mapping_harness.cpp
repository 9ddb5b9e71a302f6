// C++ harness that drives libvoxelba.so through include/voxelba_adapter.hpp in the call order of the reference's
// thd_odometry_localmapping (voxelslam.cpp:1899-1927 and 1951-2043): the class names and call sites below are the
// reference's (LidarFactor voxhess; cut_voxel_multi; multi_recut; LI_BA_Optimizer::damping_iter; multi_margi; the mp[]
// rotation; the x_buf / pvec_buf / imu_pre_buf slide) with the ROS / PCL / Eigen containers replaced by the adapter's
// plain-array ones.  Input and output are flat files of doubles written / read by tests/test_gpu_harness.py, which runs
// the same sequence on the CPU oracle and compares.
//
//   input : [magic 20241004, win_size, n_scans, mode (0 lidar-only | 1 LI | 2 LI + gravity on the first full window),
//            voxel_size, max_layer, max_points, min_eigen_value, plane_thre[4], min_point[4], imu_coef, thread_num]
//           per scan : [n, x_curr (25 state doubles), x_curr.cov (225), n_imu] points[n][3] var_body[n][9]
//                      imu samples of the interval BEFORE this scan: t[n_imu] gyr[n_imu][3] acc[n_imu][3]
//           noise_meas[6] noise_walk[6]
//   output: per optimised window [scan index, W x 25 states, v6[6]] ... then [-1, n_leaves] leaf dump [n][39] plane_var dump [n][86]
#include "../../include/voxelba_adapter.hpp"
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <deque>
#include <memory>
#include <vector>

using namespace vba;

static std::vector<double> read_all(const char *path) {
  FILE *f = std::fopen(path, "rb");
  if (!f) { std::fprintf(stderr, "harness: cannot open %s\n", path); std::exit(2); }
  std::fseek(f, 0, SEEK_END);
  const long bytes = std::ftell(f);
  std::fseek(f, 0, SEEK_SET);
  std::vector<double> v((size_t)bytes / 8);
  if (std::fread(v.data(), 8, v.size(), f) != v.size()) { std::fprintf(stderr, "harness: short read\n"); std::exit(2); }
  std::fclose(f);
  return v;
}

int main(int argc, char **argv) {
  if (argc < 3) { std::fprintf(stderr, "usage: %s <input.bin> <output.bin>\n", argv[0]); return 2; }
  const std::vector<double> in = read_all(argv[1]);
  size_t q = 0;
  auto next = [&]() { return in.at(q++); };
  if (next() != 20241004.0) { std::fprintf(stderr, "harness: bad magic\n"); return 2; }
  const int win_size = (int)next(), n_scans = (int)next(), mode = (int)next();
  vba_options opt;
  vba_default_options(&opt);
  opt.win_size = win_size;
  opt.voxel_size = next(); opt.max_layer = (int)next(); opt.max_points = (int)next(); opt.min_eigen_value = next();
  for (int i = 0; i < 4; i++) opt.plane_eigen_value_thre[i] = next();
  for (int i = 0; i < 4; i++) opt.min_point[i] = next();
  opt.imu_coef = next(); opt.thread_num = (int)next();

  std::vector<double> out;
  try {
    Context ctx(opt);
    VoxelMap surf_map(ctx);                      // surf_map + surf_map_slide
    LidarFactor voxhess(ctx, win_size);          // VS:1758
    std::vector<IMUST> x_buf;                    // VS:1760
    std::vector<std::shared_ptr<PVec>> pvec_buf;
    std::deque<IMU_PRE *> imu_pre_buf;
    std::vector<double> hess;                    // Eigen::MatrixXd hess (VS:1762)
    int win_count = 0, win_base = 0, g_update = (mode == 2) ? 2 : 0;
    const int mgsize = 1, DIM = VBA_DIM;
    double jour = 0;
    // the noise globals of preintegration.hpp:8-9 travel at the end of the file
    const double *noise = &in[in.size() - 12];

    for (int k = 0; k < n_scans; k++) {
      const int n = (int)next();
      IMUST x_curr;
      std::memcpy(&x_curr.t, &in.at(q), 25 * sizeof(double)); q += 25;
      std::memcpy(x_curr.cov, &in.at(q), 225 * sizeof(double)); q += 225;
      const int n_imu = (int)next();
      std::shared_ptr<PVec> pptr(new PVec((size_t)n));
      for (int i = 0; i < n; i++) { std::memcpy((*pptr)[i].pnt, &in.at(q), 24); q += 3; }
      for (int i = 0; i < n; i++) { std::memcpy((*pptr)[i].var, &in.at(q), 72); q += 9; }
      const double *imu_t = &in[q]; q += (size_t)n_imu;
      const double *imu_g = &in[q]; q += (size_t)n_imu * 3;
      const double *imu_a = &in[q]; q += (size_t)n_imu * 3;

      // VS:1905-1913
      win_count++;
      x_buf.push_back(x_curr);
      pvec_buf.push_back(pptr);
      if (win_count > 1) {
        imu_pre_buf.push_back(new IMU_PRE(x_buf[win_count - 2].bg, x_buf[win_count - 2].ba));
        imu_pre_buf[win_count - 2]->push_imu(n_imu, imu_t, imu_g, imu_a, noise, noise + 6);
      }
      // VS:1918-1926: pvec_update (VS:1901) rides in the insert, as the device path fuses the two
      voxhess.clear();
      surf_map.pvec_update_cut_voxel_multi(*pvec_buf[win_count - 1], win_count - 1, x_curr);
      surf_map.multi_recut(win_count, x_buf);

      if (win_count >= win_size) {                               // VS:1951
        if (mode == 0) {                                         // lidar-only windows (what HBA_add_edge runs, VS:2895-2899)
          Lidar_BA_Optimizer opt_lsv;
          std::vector<double> resis;
          opt_lsv.damping_iter(x_buf, voxhess, &hess, resis, 3);
        } else if (g_update == 2) {                              // VS:1955-1964
          LI_BA_OptimizerGravity opt_lsv;
          std::vector<double> resis;
          opt_lsv.damping_iter(x_buf, voxhess, imu_pre_buf, resis, &hess, 5);
          g_update = 0;
        } else {                                                 // VS:1967-1970
          LI_BA_Optimizer opt_lsv;
          opt_lsv.damping_iter(x_buf, voxhess, imu_pre_buf, &hess);
        }
        // VS:1973-1977: v6 = 1 / |diag(hess.block<6,6>(0, DIM))|
        const int nh = (mode == 0) ? 6 * win_size : DIM * win_size + ((int)hess.size() == (DIM * win_size + 3) * (DIM * win_size + 3) ? 3 : 0);
        const int col0 = (mode == 0) ? 6 : DIM;
        double v6[6];
        for (int i = 0; i < 6; i++) v6[i] = 1.0 / std::fabs(hess[(size_t)i * nh + col0 + i]);
        out.push_back((double)k);
        for (int i = 0; i < win_size; i++) out.insert(out.end(), &x_buf[i].t, &x_buf[i].t + 25);
        out.insert(out.end(), v6, v6 + 6);

        surf_map.multi_margi(jour, win_count, x_buf);            // VS:1991
        jour += 0.1;
        surf_map.slide(mgsize);                                  // mp[] rotation VS:2014-2019
        for (int i = mgsize; i < win_count; i++) {               // VS:2022-2028
          x_buf[i - mgsize] = x_buf[i];
          std::swap(pvec_buf[i - mgsize], pvec_buf[i]);
        }
        for (int i = win_count - mgsize; i < win_count; i++) {   // VS:2031-2038
          x_buf.pop_back();
          pvec_buf.pop_back();
          delete imu_pre_buf.front();
          imu_pre_buf.pop_front();
        }
        win_base += mgsize;
        win_count -= mgsize;
      }
    }
    while (!imu_pre_buf.empty()) { delete imu_pre_buf.front(); imu_pre_buf.pop_front(); }
    // final map: leaves + plane covariances
    const int nl = vba_map_dump_leaves(ctx.get(), nullptr, 0);
    out.push_back(-1.0);
    out.push_back((double)nl);
    std::vector<double> leaves((size_t)(nl > 0 ? nl : 1) * 39), pv((size_t)(nl > 0 ? nl : 1) * 86);
    if (nl > 0) { vba_map_dump_leaves(ctx.get(), leaves.data(), nl); vba_map_dump_plane_var(ctx.get(), pv.data(), nl); }
    out.insert(out.end(), leaves.begin(), leaves.begin() + (size_t)nl * 39);
    out.insert(out.end(), pv.begin(), pv.begin() + (size_t)nl * 86);
  } catch (const std::exception &e) {
    std::fprintf(stderr, "harness: %s\n", e.what());
    return 1;
  }
  FILE *f = std::fopen(argv[2], "wb");
  if (!f || std::fwrite(out.data(), 8, out.size(), f) != out.size()) { std::fprintf(stderr, "harness: cannot write %s\n", argv[2]); return 2; }
  std::fclose(f);
  return 0;
}
