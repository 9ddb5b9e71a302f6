// Odometry of the initialisation phase: VOXEL_SLAM::lio_state_estimation_kdtree (voxelslam.cpp:1102-1252).  The reference
// matches every scan point against a point-cloud map through pcl::KdTreeFLANN::nearestKSearch (5 exact nearest neighbours,
// squared L2 on float x, y, z), fits a plane to the five by least squares and runs the same iterated EKF as the voxel-map
// odometry.  The map here holds at most a few 10^4 points (it is re-sampled on a 0.5 m grid after every scan), so the
// exact 5-NN is a tiled brute-force scan: one thread per (scan point, map slice), the slice streamed through LDS 256 points at
// a time, the five best kept sorted in registers, the slices merged by a second kernel — no tree, no traversal divergence.
#pragma once
#include <hip/hip_runtime.h>

namespace vba {

struct KdPose { double R[9], t[3]; };

// least squares min ||A x - b|| for A 5x3 by Householder QR with column pivoting (what Eigen's colPivHouseholderQr().solve
// computes for a full-rank A, VS:1172)
__device__ __forceinline__ void kd_lstsq_5x3(double A[5][3], double b[5], double x[3]) {
  int perm[3] = {0, 1, 2};
  double cn[3];
#pragma unroll
  for (int c = 0; c < 3; c++) { cn[c] = 0; for (int r = 0; r < 5; r++) cn[c] += A[r][c] * A[r][c]; }
#pragma unroll
  for (int k = 0; k < 3; k++) {
    int piv = k;
    for (int c = k + 1; c < 3; c++) if (cn[c] > cn[piv]) piv = c;
    if (piv != k) {
      for (int r = 0; r < 5; r++) { const double t = A[r][k]; A[r][k] = A[r][piv]; A[r][piv] = t; }
      { const int t = perm[k]; perm[k] = perm[piv]; perm[piv] = t; }
      { const double t = cn[k]; cn[k] = cn[piv]; cn[piv] = t; }
    }
    double nrm = 0;
    for (int r = k; r < 5; r++) nrm += A[r][k] * A[r][k];
    nrm = sqrt(nrm);
    if (nrm != 0) {
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
    }
    for (int c = k + 1; c < 3; c++) { cn[c] = 0; for (int r = k + 1; r < 5; r++) cn[c] += A[r][c] * A[r][c]; }
  }
  double y[3];
  for (int k = 2; k >= 0; k--) {
    double s = b[k];
    for (int c = k + 1; c < 3; c++) s -= A[k][c] * y[c];
    y[k] = s / A[k][k];
  }
  for (int k = 0; k < 3; k++) x[perm[k]] = y[k];
}

// refind pass (VS:1156-1196), stage 1: the five nearest map points of every scan point WITHIN ONE SLICE of the map
// (gridDim.y slices: one thread per scan point alone leaves three quarters of the chip idle at 20k-point scans).  A candidate
// is the 64-bit word (float distance bits << 32 | index): positive floats order like their bit patterns, so comparing the
// words orders by distance and, at equal distance, by index — the order a sequential scan with strict `<` produces.
__global__ __launch_bounds__(256) void k_kd_match(int n, const double *__restrict__ pts, KdPose X, int m, const double *__restrict__ tree,
                                                  unsigned long long *__restrict__ cand /*[slices][n][5]*/) {
  __shared__ float tx[256], ty[256], tz[256];
  const int i = blockIdx.x * 256 + threadIdx.x;
  float qx = 0, qy = 0, qz = 0;
  if (i < n) {
    const double x = pts[3 * (size_t)i], y = pts[3 * (size_t)i + 1], z = pts[3 * (size_t)i + 2];
    qx = (float)(X.R[0] * x + X.R[1] * y + X.R[2] * z + X.t[0]);
    qy = (float)(X.R[3] * x + X.R[4] * y + X.R[5] * z + X.t[1]);
    qz = (float)(X.R[6] * x + X.R[7] * y + X.R[8] * z + X.t[2]);
  }
  float bd0 = 3.4e38f, bd1 = 3.4e38f, bd2 = 3.4e38f, bd3 = 3.4e38f, bd4 = 3.4e38f;
  int bi0 = -1, bi1 = -1, bi2 = -1, bi3 = -1, bi4 = -1;
  const int ntile = (m + 255) / 256, per = (ntile + (int)gridDim.y - 1) / (int)gridDim.y;
  const int lo = (int)blockIdx.y * per * 256, hi = (lo + per * 256 < m) ? lo + per * 256 : m;
  for (int base = lo; base < hi; base += 256) {
    const int j = base + threadIdx.x;
    __syncthreads();
    if (j < hi) { tx[threadIdx.x] = (float)tree[3 * (size_t)j]; ty[threadIdx.x] = (float)tree[3 * (size_t)j + 1]; tz[threadIdx.x] = (float)tree[3 * (size_t)j + 2]; }
    __syncthreads();
    const int cnt = (hi - base < 256) ? hi - base : 256;
    for (int k = 0; k < cnt; k++) {
      const float dx = qx - tx[k], dy = qy - ty[k], dz = qz - tz[k];
      float d = dx * dx; d += dy * dy; d += dz * dz;       // FLANN L2_Simple accumulates in float, x then y then z
      if (d < bd4) {                                       // insertion into the sorted five (strict <: the earlier index wins a tie)
        const int id = base + k;
        if (d < bd3) { bd4 = bd3; bi4 = bi3;
          if (d < bd2) { bd3 = bd2; bi3 = bi2;
            if (d < bd1) { bd2 = bd1; bi2 = bi1;
              if (d < bd0) { bd1 = bd0; bi1 = bi0; bd0 = d; bi0 = id; } else { bd1 = d; bi1 = id; }
            } else { bd2 = d; bi2 = id; }
          } else { bd3 = d; bi3 = id; }
        } else { bd4 = d; bi4 = id; }
      }
    }
  }
  if (i >= n) return;
  unsigned long long *o = cand + ((size_t)blockIdx.y * n + i) * 5;
  o[0] = ((unsigned long long)__float_as_uint(bd0) << 32) | (unsigned int)bi0;
  o[1] = ((unsigned long long)__float_as_uint(bd1) << 32) | (unsigned int)bi1;
  o[2] = ((unsigned long long)__float_as_uint(bd2) << 32) | (unsigned int)bi2;
  o[3] = ((unsigned long long)__float_as_uint(bd3) << 32) | (unsigned int)bi3;
  o[4] = ((unsigned long long)__float_as_uint(bd4) << 32) | (unsigned int)bi4;
}

// stage 2: merge the slices' candidates, fit the plane (VS:1166-1190): (unit normal, distance) per scan point, distance < 0 = rejected
__global__ __launch_bounds__(256) void k_kd_fit(int n, int slices, const unsigned long long *__restrict__ cand, const double *__restrict__ tree,
                                                double *__restrict__ planes /*[n][4]*/) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  unsigned long long b0 = ~0ull, b1 = ~0ull, b2 = ~0ull, b3 = ~0ull, b4 = ~0ull;
  for (int s = 0; s < slices; s++) {
    const unsigned long long *c = cand + ((size_t)s * n + i) * 5;
#pragma unroll
    for (int k = 0; k < 5; k++) {
      const unsigned long long key = c[k];
      if ((unsigned int)key == 0xFFFFFFFFu) continue;       // unfilled slot
      if (key < b4) {
        if (key < b3) { b4 = b3;
          if (key < b2) { b3 = b2;
            if (key < b1) { b2 = b1;
              if (key < b0) { b1 = b0; b0 = key; } else b1 = key;
            } else b2 = key;
          } else b3 = key;
        } else b4 = key;
      }
    }
  }
  const unsigned long long bk[5] = {b0, b1, b2, b3, b4};
  double A[5][3], Aw[5][3], b[5];
  for (int k = 0; k < 5; k++) {
    const size_t id = (bk[k] == ~0ull) ? 0 : (size_t)(unsigned int)bk[k];
    for (int c = 0; c < 3; c++) { A[k][c] = (double)(float)tree[3 * id + c]; Aw[k][c] = A[k][c]; }
    b[k] = -1.0;
  }
  double dir[3];
  kd_lstsq_5x3(Aw, b, dir);
  bool bad = b4 == ~0ull;
  for (int k = 0; k < 5; k++) if (fabs(dir[0] * A[k][0] + dir[1] * A[k][1] + dir[2] * A[k][2] + 1.0) > 0.1) bad = true;     // VS:1174-1185
  double *o = planes + 4 * (size_t)i;
  if (bad || !(dir[0] == dir[0])) { o[0] = o[1] = o[2] = 0.0; o[3] = -1.0; return; }
  const double d = 1.0 / sqrt(dir[0] * dir[0] + dir[1] * dir[1] + dir[2] * dir[2]);
  o[0] = dir[0] * d; o[1] = dir[1] * d; o[2] = dir[2] * d; o[3] = d;
}

// HTH (21 unique), HTz (6), valid count: per-block partials [nb][28]  (VS:1198-1209)
__global__ __launch_bounds__(256) void k_kd_accum(int n, const double *__restrict__ pts, KdPose X, const double *__restrict__ planes, double *__restrict__ part) {
  __shared__ double red[4][28];
  const int i = blockIdx.x * 256 + threadIdx.x;
  double s[28];
#pragma unroll
  for (int k = 0; k < 28; k++) s[k] = 0.0;
  if (i < n) {
    const double *pl = planes + 4 * (size_t)i;
    if (pl[3] >= 0) {
      const double x = pts[3 * (size_t)i], y = pts[3 * (size_t)i + 1], z = pts[3 * (size_t)i + 2];
      const double wx = X.R[0] * x + X.R[1] * y + X.R[2] * z + X.t[0], wy = X.R[3] * x + X.R[4] * y + X.R[5] * z + X.t[1], wz = X.R[6] * x + X.R[7] * y + X.R[8] * z + X.t[2];
      const double pd2 = pl[0] * wx + pl[1] * wy + pl[2] * wz + pl[3];
      // jac = [hat(p) R^T n ; n]
      const double rx = X.R[0] * pl[0] + X.R[3] * pl[1] + X.R[6] * pl[2], ry = X.R[1] * pl[0] + X.R[4] * pl[1] + X.R[7] * pl[2], rz = X.R[2] * pl[0] + X.R[5] * pl[1] + X.R[8] * pl[2];
      const double j[6] = {y * rz - z * ry, z * rx - x * rz, x * ry - y * rx, pl[0], pl[1], pl[2]};
      int idx = 0;
#pragma unroll
      for (int r = 0; r < 6; r++)
#pragma unroll
        for (int c = r; c < 6; c++) s[idx++] = j[r] * j[c];
#pragma unroll
      for (int r = 0; r < 6; r++) s[21 + r] = -pd2 * j[r];
      s[27] = 1.0;
    }
  }
#pragma unroll
  for (int k = 0; k < 28; k++) s[k] = wave_sum(s[k]);
  if ((threadIdx.x & 63) == 0)
#pragma unroll
    for (int k = 0; k < 28; k++) red[threadIdx.x >> 6][k] = s[k];
  __syncthreads();
  if (threadIdx.x < 28) part[(size_t)blockIdx.x * 28 + threadIdx.x] = (red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x]);
}

// world points of the scan appended to the map, as PCL floats (VS:1107-1114, 1238-1246)
__global__ void k_kd_append(int n, const double *__restrict__ pts, KdPose X, double *__restrict__ tree_out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const double x = pts[3 * (size_t)i], y = pts[3 * (size_t)i + 1], z = pts[3 * (size_t)i + 2];
  tree_out[3 * (size_t)i] = (double)(float)(X.R[0] * x + X.R[1] * y + X.R[2] * z + X.t[0]);
  tree_out[3 * (size_t)i + 1] = (double)(float)(X.R[3] * x + X.R[4] * y + X.R[5] * z + X.t[1]);
  tree_out[3 * (size_t)i + 2] = (double)(float)(X.R[6] * x + X.R[7] * y + X.R[8] * z + X.t[2]);
}

}  // namespace vba
