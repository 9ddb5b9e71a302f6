// Scan pre-processing that feeds K1, one thread per point:
//   down_sampling_voxel   tools.hpp:201-238    voxel-grid centroid filter (hash grid + f64 sums; the reference's running mean
//                                              in float is order dependent, the centroid is the same up to float rounding)
//   undistortion          ekf_imu.hpp:137-163  per-point motion compensation against the IMU-propagated poses
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace vba {

struct DsSlot {
  unsigned long long key;      // packed voxel index, DS_EMPTY when free
  double sx, sy, sz;
  double vx, vy, vz;           // down_sampling_pvec: sums of the covariance diagonals
  unsigned long long mind;     // down_sampling_close: smallest squared distance to the centroid (bits of a non-negative double)
  int cnt, first, best, pad;
};
static constexpr unsigned long long DS_EMPTY = ~0ull;

// TL:210-217 on PCL float coordinates; dbl != 0: the pointVar form of VM:45-51 (double coordinates)
__device__ __forceinline__ long long ds_axis_key(double v, double voxel_size, int dbl = 0) {
  float loc = (float)((dbl ? v : (double)(float)v) / voxel_size);
  if (loc < 0) loc = (float)((double)loc - 1.0);
  return (long long)loc;
}
__device__ __forceinline__ unsigned long long ds_pack(long long x, long long y, long long z) {
  return ((unsigned long long)(x & 0x1FFFFF) << 42) | ((unsigned long long)(y & 0x1FFFFF) << 21) | (unsigned long long)(z & 0x1FFFFF);
}
__device__ __forceinline__ unsigned int ds_hash(unsigned long long k) {
  k ^= k >> 33; k *= 0xff51afd7ed558ccdull; k ^= k >> 33; k *= 0xc4ceb9fe1a85ec53ull; k ^= k >> 33;
  return (unsigned int)k;
}

__global__ void k_ds_clear(DsSlot *__restrict__ tab, int cap) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= cap) return;
  DsSlot s; s.key = DS_EMPTY; s.sx = s.sy = s.sz = 0.0; s.vx = s.vy = s.vz = 0.0; s.mind = ~0ull; s.cnt = 0; s.first = 0x7fffffff; s.best = 0x7fffffff; s.pad = 0;
  tab[i] = s;
}

// point -> slot (open addressing, linear probing), sums, count, first point of the voxel
// var != nullptr: pointVar input (down_sampling_pvec, VM:39-83): double coordinates, the covariance diagonal is averaged too
__global__ void k_ds_insert(int n, const double *__restrict__ pnt, const double *__restrict__ var, double voxel_size, DsSlot *__restrict__ tab, int cap_mask,
                            int *__restrict__ slot_of) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int dbl = var != nullptr;
  const double x = pnt[3 * (size_t)i], y = pnt[3 * (size_t)i + 1], z = pnt[3 * (size_t)i + 2];
  const unsigned long long key = ds_pack(ds_axis_key(x, voxel_size, dbl), ds_axis_key(y, voxel_size, dbl), ds_axis_key(z, voxel_size, dbl));
  unsigned int h = ds_hash(key) & cap_mask;
  for (int probe = 0; probe <= cap_mask; probe++) {          // the table has >= 2n slots: always terminates
    const unsigned long long old = atomicCAS(&tab[h].key, DS_EMPTY, key);
    if (old == DS_EMPTY || old == key) break;
    h = (h + 1) & cap_mask;
  }
  atomicAdd(&tab[h].sx, dbl ? x : (double)(float)x);
  atomicAdd(&tab[h].sy, dbl ? y : (double)(float)y);
  atomicAdd(&tab[h].sz, dbl ? z : (double)(float)z);
  if (dbl) { atomicAdd(&tab[h].vx, var[9 * (size_t)i]); atomicAdd(&tab[h].vy, var[9 * (size_t)i + 4]); atomicAdd(&tab[h].vz, var[9 * (size_t)i + 8]); }
  atomicAdd(&tab[h].cnt, 1);
  atomicMin(&tab[h].first, i);
  slot_of[i] = h;
}

// block-level count of "first point of its voxel" flags; blk[b] = count of block b
__global__ __launch_bounds__(256) void k_ds_count(int n, const DsSlot *__restrict__ tab, const int *__restrict__ slot_of, int *__restrict__ blk) {
  __shared__ int wsum[4];
  const int i = blockIdx.x * 256 + threadIdx.x;
  const int f = (i < n && tab[slot_of[i]].first == i) ? 1 : 0;
  const unsigned long long m = __ballot(f);
  if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6] = __popcll(m);
  __syncthreads();
  if (threadIdx.x == 0) blk[blockIdx.x] = wsum[0] + wsum[1] + wsum[2] + wsum[3];
}

// exclusive scan of the block counts by one workgroup (nb <= a few thousand); total -> *n_out
__global__ __launch_bounds__(256) void k_ds_scan(int nb, int *__restrict__ blk, int *__restrict__ n_out) {
  __shared__ int part[256];
  const int t = threadIdx.x, per = (nb + 255) / 256;
  int s = 0;
  for (int k = 0; k < per; k++) { const int b = t * per + k; if (b < nb) s += blk[b]; }
  part[t] = s;
  __syncthreads();
  if (t == 0) { int acc = 0; for (int k = 0; k < 256; k++) { const int v = part[k]; part[k] = acc; acc += v; } *n_out = acc; }
  __syncthreads();
  int acc = part[t];
  for (int k = 0; k < per; k++) { const int b = t * per + k; if (b < nb) { const int v = blk[b]; blk[b] = acc; acc += v; } }
}

// stable compaction in first-occurrence order; centroid rounded to float like the PCL cloud it replaces
// mode 0: centroid (down_sampling_voxel); 1: centroid + mean covariance diagonal in vout (down_sampling_pvec);
// 2: first[] receives the index of the point closest to the centroid (down_sampling_close), out = that point's voxel centroid
__global__ __launch_bounds__(256) void k_ds_emit(int n, const DsSlot *__restrict__ tab, const int *__restrict__ slot_of, const int *__restrict__ blk,
                                                 double *__restrict__ out, int *__restrict__ count, int *__restrict__ first, double *__restrict__ vout, int mode) {
  __shared__ int wsum[4];
  const int i = blockIdx.x * 256 + threadIdx.x, lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  DsSlot s;
  int f = 0;
  if (i < n) { s = tab[slot_of[i]]; f = (s.first == i) ? 1 : 0; }
  const unsigned long long m = __ballot(f);
  if (lane == 0) wsum[w] = __popcll(m);
  __syncthreads();
  if (!f) return;
  int pos = blk[blockIdx.x] + __popcll(m & ((1ull << lane) - 1ull));
  for (int k = 0; k < w; k++) pos += wsum[k];
  const double inv = 1.0 / (double)s.cnt;
  out[3 * (size_t)pos] = (double)(float)(s.sx * inv);
  out[3 * (size_t)pos + 1] = (double)(float)(s.sy * inv);
  out[3 * (size_t)pos + 2] = (double)(float)(s.sz * inv);
  count[pos] = s.cnt;
  first[pos] = (mode == 2) ? ((s.best != 0x7fffffff) ? s.best : i) : i;
  if (mode == 1) {
    vout[3 * (size_t)pos] = (double)(float)(s.vx * inv); vout[3 * (size_t)pos + 1] = (double)(float)(s.vy * inv); vout[3 * (size_t)pos + 2] = (double)(float)(s.vz * inv);
  }
}

// down_sampling_close (tools.hpp:240-298): per voxel the point with the smallest squared distance to the (float) centroid,
// first such point in input order, only distances < 100 compete (ndis = 100, mnum = 0 at TL:281-282).
__global__ void k_ds_close_min(int n, const double *__restrict__ pnt, DsSlot *__restrict__ tab, const int *__restrict__ slot_of, double *__restrict__ dist) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  DsSlot *s = tab + slot_of[i];
  const double inv = 1.0 / (double)s->cnt;
  const float cx = (float)(s->sx * inv), cy = (float)(s->sy * inv), cz = (float)(s->sz * inv);
  const double xx = (double)(cx - (float)pnt[3 * (size_t)i]), yy = (double)(cy - (float)pnt[3 * (size_t)i + 1]), zz = (double)(cz - (float)pnt[3 * (size_t)i + 2]);
  const double d = xx * xx + yy * yy + zz * zz;
  dist[i] = d;
  if (d < 100.0) atomicMin(&s->mind, (unsigned long long)__double_as_longlong(d));
}
__global__ void k_ds_close_arg(int n, DsSlot *__restrict__ tab, const int *__restrict__ slot_of, const double *__restrict__ dist) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  DsSlot *s = tab + slot_of[i];
  if ((unsigned long long)__double_as_longlong(dist[i]) == s->mind) atomicMin(&s->best, i);
}

// ---------------------------------------------------------------- undistortion
// prm: [m] imu poses x 22 doubles (t, R[9], p[3], v[3], angvel[3], acc[3]), then end pose R[9] p[3], then extrinsic R[9] t[3].
__device__ __forceinline__ void undist_one(const double *__restrict__ q, const double *__restrict__ Re, const double *__restrict__ pe,
                                           const double *__restrict__ Rx, const double *__restrict__ tx, double curv, double &x, double &y, double &z) {
  const double dt = curv - q[0];
  const double *R = q + 1, *p = q + 10, *v = q + 13, *w = q + 16, *a = q + 19;
  // Exp(angvel, dt)  tools.hpp:68-84 (threshold 1e-7)
  double E[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
  const double nrm = sqrt(w[0] * w[0] + w[1] * w[1] + w[2] * w[2]);
  if (nrm > 1e-7) {
    const double a0 = w[0] / nrm, a1 = w[1] / nrm, a2 = w[2] / nrm, th = nrm * dt;
    const double s = sin(th), c1 = 1.0 - cos(th);
    E[0] = 1.0 + c1 * (a0 * a0 - 1.0); E[1] = -s * a2 + c1 * a0 * a1;     E[2] = s * a1 + c1 * a0 * a2;
    E[3] = s * a2 + c1 * a1 * a0;      E[4] = 1.0 + c1 * (a1 * a1 - 1.0); E[5] = -s * a0 + c1 * a1 * a2;
    E[6] = -s * a1 + c1 * a2 * a0;     E[7] = s * a0 + c1 * a2 * a1;      E[8] = 1.0 + c1 * (a2 * a2 - 1.0);
  }
  double Ri[9];
#pragma unroll
  for (int r = 0; r < 3; r++)
#pragma unroll
    for (int c = 0; c < 3; c++) Ri[3 * r + c] = R[3 * r] * E[c] + R[3 * r + 1] * E[3 + c] + R[3 * r + 2] * E[6 + c];
  double T[3], b[3], u[3], g[3];
#pragma unroll
  for (int k = 0; k < 3; k++) T[k] = p[k] + v[k] * dt + a[k] * (0.5 * dt * dt) - pe[k];
#pragma unroll
  for (int k = 0; k < 3; k++) b[k] = Rx[3 * k] * x + Rx[3 * k + 1] * y + Rx[3 * k + 2] * z + tx[k];           // Lid_rot_to_IMU * P_i + offset
#pragma unroll
  for (int k = 0; k < 3; k++) u[k] = Ri[3 * k] * b[0] + Ri[3 * k + 1] * b[1] + Ri[3 * k + 2] * b[2] + T[k];   // R_i * (...) + T_ei
#pragma unroll
  for (int k = 0; k < 3; k++) g[k] = Re[k] * u[0] + Re[3 + k] * u[1] + Re[6 + k] * u[2] - tx[k];               // xc.R^T * (...) - offset
  x = (double)(float)(Rx[0] * g[0] + Rx[3] * g[1] + Rx[6] * g[2]);                                            // Lid_rot_to_IMU^T * (...)
  y = (double)(float)(Rx[1] * g[0] + Rx[4] * g[1] + Rx[7] * g[2]);
  z = (double)(float)(Rx[2] * g[0] + Rx[5] * g[1] + Rx[8] * g[2]);
}

// Points are time-sorted (voxelslam.hpp:92-95), so the backwards walk of EK:138-163 assigns point i to the last pose with
// t < curvature_i; points at or before the first pose stay untouched.  The very first point is compensated once per
// remaining pose (the `break` at EK:161 leaves the iterator on it while the outer loop continues) — kept.
__global__ void k_undistort(int n, double *__restrict__ pnt, const double *__restrict__ curv, int m, const double *__restrict__ prm) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const double *Re = prm + 22 * (size_t)m, *pe = Re + 9, *Rx = pe + 3, *tx = Rx + 9;
  const double cv = (double)(float)curv[i];
  int lo = 0, hi = m;                       // first pose index with t >= cv
  while (lo < hi) { const int mid = (lo + hi) >> 1; if (prm[22 * (size_t)mid] < cv) lo = mid + 1; else hi = mid; }
  int j = lo - 1;
  if (j < 0) return;
  double x = (double)(float)pnt[3 * (size_t)i], y = (double)(float)pnt[3 * (size_t)i + 1], z = (double)(float)pnt[3 * (size_t)i + 2];
  const int jend = (i == 0) ? 0 : j;
  for (; j >= jend; j--) undist_one(prm + 22 * (size_t)j, Re, pe, Rx, tx, cv, x, y, z);
  pnt[3 * (size_t)i] = x; pnt[3 * (size_t)i + 1] = y; pnt[3 * (size_t)i + 2] = z;
}

}  // namespace vba
