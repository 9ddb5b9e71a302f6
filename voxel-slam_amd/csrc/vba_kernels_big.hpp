// Arbitrary window size (W > 16, or any W different from the context's templated window): the top layer of the
// hierarchical global BA optimises ALL submaps at once (voxelslam.cpp:3103-3113: HBA_add_edge over gba_submaps), so the
// (voxel, frame) occupancy is sparse and the dense per-voxel [W][10] record of the local-BA path does not apply.
//   * octree build: as vba_kernels_gba.hpp, but the per-keyframe body clusters live in a hash table keyed (node, frame);
//   * factor store: CSR over voxels, one entry per occupied (voxel, frame);
//   * residual pass: one thread per voxel over its entries (evaluate_only_residual, VM:285-325);
//   * Hessian pass (acc_evaluate2, VM:167-281) in the rank-3 form of k_hessian2: k_big_slot computes every entry's three
//     6-vectors (g1, g2, h) and adds its diagonal-block remainder E and its gradient, k_big_pairs adds
//     c1 g1_i g1_j^T + c2 g2_i g2_j^T + c3 h_i h_j^T for every pair of entries of a voxel — f64 atomics into the dense
//     (6W)^2 matrix (no tiling / MFMA SYRK yet: this path is about capability, the reference runs it rarely);
//   * the dense (6W)^2 LDL^T runs in HBM (k_bigl_*: panel kernel + MFMA trailing update per 8 columns); the LM bookkeeping,
//     the O(n^2) back substitution and the edge extraction stay on the host.
#pragma once
#include <hip/hip_runtime.h>
#include <string>
#include <vector>

namespace vba {

struct BigView {
  int W, V, E, capV, capE;
  int *vptr;        // [V + 1]
  int *efr, *evox;  // [E] frame / voxel of an entry
  double *ecl;      // [10][capE] body clusters
  double *gv;       // [18][capE] g1, g2, h of an entry (Hessian pass scratch)
  double *eval, *evec, *pcr;   // [3][capV], [9][capV], [10][capV]
  double *poses;    // [W][12]
  double *H, *g, *r;   // dense (6W)^2, 6W, 1
  int *eidx;           // [V][W] entry of (voxel, frame) or -1 (k_big_syrk operand staging)
  double *es;          // [27][capE] diagonal-block remainder E (21 upper) + gradient (6) of an entry, summed per frame by k_big_diag
};

__device__ __forceinline__ void big_cluster_tf(const double *c, const double *R, double *o) { cluster_transform_dev(c, R, o); }

// evaluate_only_residual (VM:285-325) on the sparse store; residual accumulated with one atomic per workgroup
__global__ __launch_bounds__(256) void k_big_residual(BigView b) {
  __shared__ double part[4];
  const int v = blockIdx.x * 256 + threadIdx.x;
  double r = 0.0;
  if (v < b.V) {
    const size_t ce = (size_t)b.capE, cv = (size_t)b.capV;
    double s[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    for (int e = b.vptr[v]; e < b.vptr[v + 1]; e++) {
      double c[10], t[10];
      for (int k = 0; k < 10; k++) c[k] = b.ecl[(size_t)k * ce + e];
      big_cluster_tf(c, b.poses + 12 * b.efr[e], t);
      for (int k = 0; k < 10; k++) s[k] += t[k];
    }
    const double N = s[9], b0 = s[6] / N, b1 = s[7] / N, b2 = s[8] / N;
    double w0, w1, w2, U[9];
    eig3_sym_dev(s[0] / N - b0 * b0, s[1] / N - b1 * b0, s[2] / N - b2 * b0, s[3] / N - b1 * b1, s[4] / N - b2 * b1, s[5] / N - b2 * b2, w0, w1, w2, U);
    b.eval[v] = w0; b.eval[cv + v] = w1; b.eval[2 * cv + v] = w2;
    for (int k = 0; k < 9; k++) b.evec[(size_t)k * cv + v] = U[k];
    for (int k = 0; k < 10; k++) b.pcr[(size_t)k * cv + v] = s[k];
    r = w0;                                   // coeffs = 1 (LR:380)
  }
  r = wave_sum(r);
  if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = r;
  __syncthreads();
  if (threadIdx.x == 0) unsafeAtomicAdd(b.r, part[0] + part[1] + part[2] + part[3]);
}

// Hessian pass, part 1: one thread per entry.  Same per-slot algebra as phase A of k_hessian2 (vba_kernels_factor.hpp).
__global__ void k_big_slot(BigView b) {
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= b.E) return;
  const size_t ce = (size_t)b.capE, cv = (size_t)b.capV;
  const int v = b.evox[e], fi = b.efr[e], n6 = 6 * b.W;
  const double l0 = b.eval[v], l1 = b.eval[cv + v], l2 = b.eval[2 * cv + v];
  const double NN = b.pcr[9 * cv + v];
  if (e == b.vptr[v]) unsafeAtomicAdd(b.r, l0);                               // VM:275
  const double n = b.ecl[9 * ce + e];
  const double pxx = b.ecl[e], pxy = b.ecl[ce + e], pxz = b.ecl[2 * ce + e], pyy = b.ecl[3 * ce + e], pyz = b.ecl[4 * ce + e], pzz = b.ecl[5 * ce + e];
  const double vx = b.ecl[6 * ce + e], vy = b.ecl[7 * ce + e], vz = b.ecl[8 * ce + e];
  const double u00 = b.evec[v], u01 = b.evec[cv + v], u02 = b.evec[2 * cv + v], u10 = b.evec[3 * cv + v], u11 = b.evec[4 * cv + v], u12 = b.evec[5 * cv + v],
               u20 = b.evec[6 * cv + v], u21 = b.evec[7 * cv + v], u22 = b.evec[8 * cv + v];
  const double inn = 1.0 / NN;
  const double bx = b.pcr[6 * cv + v] * inn, by = b.pcr[7 * cv + v] * inn, bz = b.pcr[8 * cv + v] * inn;          // vBar (VM:190)
  const double *R = b.poses + 12 * fi;
  const double k0 = u00, k1 = u10, k2 = u20;
  const double a00 = R[0] * k0 + R[3] * k1 + R[6] * k2, a01 = R[1] * k0 + R[4] * k1 + R[7] * k2, a02 = R[2] * k0 + R[5] * k1 + R[8] * k2;
  const double a10 = R[0] * u01 + R[3] * u11 + R[6] * u21, a11 = R[1] * u01 + R[4] * u11 + R[7] * u21, a12 = R[2] * u01 + R[5] * u11 + R[8] * u21;
  const double a20 = R[0] * u02 + R[3] * u12 + R[6] * u22, a21 = R[1] * u02 + R[4] * u12 + R[7] * u22, a22 = R[2] * u02 + R[5] * u12 + R[8] * u22;
  const double tx = R[9] - bx, ty = R[10] - by, tz = R[11] - bz;                                                   // VM:224
  const double s0 = k0 * tx + k1 * ty + k2 * tz, s1 = u01 * tx + u11 * ty + u21 * tz, s2 = u02 * tx + u12 * ty + u22 * tz;
  const double pa00 = pxx * a00 + pxy * a01 + pxz * a02, pa01 = pxy * a00 + pyy * a01 + pyz * a02, pa02 = pxz * a00 + pyz * a01 + pzz * a02;
  const double pa10 = pxx * a10 + pxy * a11 + pxz * a12, pa11 = pxy * a10 + pyy * a11 + pyz * a12, pa12 = pxz * a10 + pyz * a11 + pzz * a12;
  const double pa20 = pxx * a20 + pxy * a21 + pxz * a22, pa21 = pxy * a20 + pyy * a21 + pyz * a22, pa22 = pxz * a20 + pyz * a21 + pzz * a22;
  const double wx = pa00 + s0 * vx, wy = pa01 + s0 * vy, wz = pa02 + s0 * vz;                                      // combo1 = hat(w) VM:228
  const double c2x = R[0] * vx + R[1] * vy + R[2] * vz + n * tx;                                                   // combo2 VM:229
  const double c2y = R[3] * vx + R[4] * vy + R[5] * vz + n * ty;
  const double c2z = R[6] * vx + R[7] * vy + R[8] * vz + n * tz;
  const double qx = vy * a02 - vz * a01, qy = vz * a00 - vx * a02, qz = vx * a01 - vy * a00;                      // viRiTuk VM:221
  const double d0 = c2x * k0 + c2y * k1 + c2z * k2;
  const double d1 = c2x * u01 + c2y * u11 + c2z * u21;
  const double d2 = c2x * u02 + c2y * u12 + c2z * u22;
  double gj[6];
  gj[0] = 2.0 * (wy * a02 - wz * a01) * inn; gj[1] = 2.0 * (wz * a00 - wx * a02) * inn; gj[2] = 2.0 * (wx * a01 - wy * a00) * inn;
  gj[3] = 2.0 * d0 * k0 * inn; gj[4] = 2.0 * d0 * k1 * inn; gj[5] = 2.0 * d0 * k2 * inn;                          // VM:235-236
  for (int k = 0; k < 6; k++) b.es[(size_t)(21 + k) * ce + e] = gj[k];
  double g1[6], g2[6], hh[6];
  {
    const double bx1 = pa10 + s1 * vx, by1 = pa11 + s1 * vy, bz1 = pa12 + s1 * vz;
    g1[0] = (-(a01 * bz1 - a02 * by1) + (wy * a12 - wz * a11)) * inn;
    g1[1] = (-(a02 * bx1 - a00 * bz1) + (wz * a10 - wx * a12)) * inn;
    g1[2] = (-(a00 * by1 - a01 * bx1) + (wx * a11 - wy * a10)) * inn;
    g1[3] = (k0 * d1 + d0 * u01) * inn; g1[4] = (k1 * d1 + d0 * u11) * inn; g1[5] = (k2 * d1 + d0 * u21) * inn;
    const double bx2 = pa20 + s2 * vx, by2 = pa21 + s2 * vy, bz2 = pa22 + s2 * vz;
    g2[0] = (-(a01 * bz2 - a02 * by2) + (wy * a22 - wz * a21)) * inn;
    g2[1] = (-(a02 * bx2 - a00 * bz2) + (wz * a20 - wx * a22)) * inn;
    g2[2] = (-(a00 * by2 - a01 * bx2) + (wx * a21 - wy * a20)) * inn;
    g2[3] = (k0 * d2 + d0 * u02) * inn; g2[4] = (k1 * d2 + d0 * u12) * inn; g2[5] = (k2 * d2 + d0 * u22) * inn;
  }
  hh[0] = qx; hh[1] = qy; hh[2] = qz; hh[3] = n * k0; hh[4] = n * k1; hh[5] = n * k2;
  for (int d = 0; d < 6; d++) { b.gv[(size_t)d * ce + e] = g1[d]; b.gv[(size_t)(6 + d) * ce + e] = g2[d]; b.gv[(size_t)(12 + d) * ce + e] = hh[d]; }
  // diagonal-block remainder E (VM:242-248; only the symmetric part of the rot-rot block survives)
  const double wa = wx * a00 + wy * a01 + wz * a02;
  const double t00 = a01 * pxz - a02 * pxy, t10 = a02 * pxx - a00 * pxz, t20 = a00 * pxy - a01 * pxx;
  const double t01 = a01 * pyz - a02 * pyy, t11 = a02 * pxy - a00 * pyz, t21 = a00 * pyy - a01 * pxy;
  const double t02 = a01 * pzz - a02 * pyz, t12 = a02 * pxz - a00 * pzz, t22 = a00 * pyz - a01 * pxz;
  const double S00 = t01 * a02 - t02 * a01, S01 = -t00 * a02 + t02 * a00, S02 = t00 * a01 - t01 * a00;
  const double S11 = -t10 * a02 + t12 * a00, S12 = t10 * a01 - t11 * a00;
  const double S22 = t20 * a01 - t21 * a00;
  const double e2 = 2.0 * inn;
  double Eb[6][6];
  Eb[0][0] = e2 * (a00 * wx - wa - S00); Eb[0][1] = e2 * (0.5 * (a00 * wy + wx * a01) - S01); Eb[0][2] = e2 * (0.5 * (a00 * wz + wx * a02) - S02);
  Eb[1][1] = e2 * (a01 * wy - wa - S11); Eb[1][2] = e2 * (0.5 * (a01 * wz + wy * a02) - S12); Eb[2][2] = e2 * (a02 * wz - wa - S22);
  Eb[1][0] = Eb[0][1]; Eb[2][0] = Eb[0][2]; Eb[2][1] = Eb[1][2];
  const double qq[3] = {qx, qy, qz}, kk[3] = {k0, k1, k2};
  for (int a = 0; a < 3; a++)
    for (int c = 0; c < 3; c++) { Eb[a][3 + c] = e2 * qq[a] * kk[c]; Eb[3 + c][a] = Eb[a][3 + c]; Eb[3 + a][3 + c] = e2 * n * kk[a] * kk[c]; }
  int idx = 0;
  for (int a = 0; a < 6; a++)
    for (int c = a; c < 6; c++) b.es[(size_t)(idx++) * ce + e] = Eb[a][c];
}
// One workgroup per frame sums the E blocks and gradients of the frame's entries (fixed order, no atomics) and adds them to
// the diagonal block of H / to g.  (Per-entry atomics put E / W entries on each of the 42 W addresses: 0.26 ms at W = 60.)
__global__ __launch_bounds__(256) void k_big_diag(BigView b) {
  __shared__ double red[4][27];
  const int f = blockIdx.x, tid = threadIdx.x, n6 = 6 * b.W;
  const size_t ce = (size_t)b.capE;
  double acc[27];
#pragma unroll
  for (int k = 0; k < 27; k++) acc[k] = 0.0;
  for (int v = tid; v < b.V; v += 256) {
    const int e = b.eidx[(size_t)v * b.W + f];
    if (e < 0) continue;
#pragma unroll
    for (int k = 0; k < 27; k++) acc[k] += b.es[(size_t)k * ce + e];
  }
#pragma unroll
  for (int k = 0; k < 27; k++) acc[k] = wave_sum(acc[k]);
  if ((tid & 63) == 0)
#pragma unroll
    for (int k = 0; k < 27; k++) red[tid >> 6][k] = acc[k];
  __syncthreads();
  if (tid < 27) {
    const double val = (red[0][tid] + red[1][tid]) + (red[2][tid] + red[3][tid]);
    if (tid >= 21) { b.g[6 * f + tid - 21] += val; return; }
    int a = 0, rem = tid;
    while (rem >= 6 - a) { rem -= 6 - a; a++; }
    const int c = a + rem;
    b.H[(size_t)(6 * f + a) * n6 + 6 * f + c] += val;
    if (a != c) b.H[(size_t)(6 * f + c) * n6 + 6 * f + a] += val;
  }
}

// Hessian pass, part 2: H += sum_v G_v^T C_v G_v as a blocked SYRK.  (An earlier version gave every entry a thread that
// added its 6x6 products against the voxel's other entries with f64 atomics: ~1e8 atomics per pass at W = 60, 1.4 ms — L2
// atomic throughput, not address contention: 32 private copies of H did not change it.)
// Workgroup = one pair (bi <= bj) of 8-frame tiles (48 x 48 outputs, a 3 x 3 patch per thread, accumulators in registers) x
// one slice of the voxels.  Per chunk of 16 voxels the 48 k-rows (voxel, m = g1 | g2 | h) x 48 columns of both sides are
// staged in LDS through the (voxel, frame) -> entry map; the i side carries the weights c_m.  A chunk with no entry in one
// of the two tiles is skipped (sparse windows).  The slice results are added to H with one atomic per output (and its
// mirror when bi != bj).
constexpr int BIG_TF = 8, BIG_TC = 6 * BIG_TF, BIG_VC = 16, BIG_KR = 3 * BIG_VC;

__global__ void k_big_eidx(BigView b) {
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e < b.E) b.eidx[(size_t)b.evox[e] * b.W + b.efr[e]] = e;
}

__global__ __launch_bounds__(256) void k_big_syrk(BigView b, int nt, int nslice) {
  __shared__ double A[BIG_KR][BIG_TC], B[BIG_KR][BIG_TC];
  // block pair from the linear index (row-major upper triangle)
  int bi = 0, rem = blockIdx.x;
  while (rem >= nt - bi) { rem -= nt - bi; bi++; }
  const int bj = bi + rem;
  const int t = threadIdx.x, pr = t >> 4, pc = t & 15;
  const size_t ce = (size_t)b.capE, cv = (size_t)b.capV;
  const int nchunk = (b.V + BIG_VC - 1) / BIG_VC;
  const int per = (nchunk + nslice - 1) / nslice, c0 = blockIdx.y * per, c1 = (c0 + per < nchunk) ? c0 + per : nchunk;
  double acc[3][3] = {{0, 0, 0}, {0, 0, 0}, {0, 0, 0}};
  // fill role of this thread: side (0 = i side, scaled), voxel of the chunk, frame of the tile
  const int side = t >> 7, vl = (t & 127) >> 3, fl = t & 7;
  const int f = (side ? bj : bi) * BIG_TF + fl;
  for (int ch = c0; ch < c1; ch++) {
    const int v = ch * BIG_VC + vl;
    int e = -1;
    if (v < b.V && f < b.W) e = b.eidx[(size_t)v * b.W + f];
    const int any_i = __syncthreads_or(side == 0 && e >= 0), any_j = __syncthreads_or(side == 1 && e >= 0);
    if (!any_i || !any_j) continue;                                   // uniform: the barriers above are reached by every thread
    double w[3] = {1.0, 1.0, 1.0};
    if (e >= 0 && side == 0) {
      const double l0 = b.eval[v], l1 = b.eval[cv + v], l2 = b.eval[2 * cv + v], NN = b.pcr[9 * cv + v];
      w[0] = 2.0 / (l0 - l1); w[1] = 2.0 / (l0 - l2); w[2] = -2.0 / NN / NN;                                      // VM:201, 264-268
    }
    double (*S)[BIG_TC] = side ? B : A;
#pragma unroll
    for (int m = 0; m < 3; m++)
#pragma unroll
      for (int d = 0; d < 6; d++) S[vl * 3 + m][fl * 6 + d] = e >= 0 ? w[m] * b.gv[(size_t)(6 * m + d) * ce + e] : 0.0;
    __syncthreads();
#pragma unroll 4
    for (int k = 0; k < BIG_KR; k++) {
      const double a0 = A[k][3 * pr], a1 = A[k][3 * pr + 1], a2 = A[k][3 * pr + 2];
      const double b0 = B[k][3 * pc], b1 = B[k][3 * pc + 1], b2 = B[k][3 * pc + 2];
      acc[0][0] += a0 * b0; acc[0][1] += a0 * b1; acc[0][2] += a0 * b2;
      acc[1][0] += a1 * b0; acc[1][1] += a1 * b1; acc[1][2] += a1 * b2;
      acc[2][0] += a2 * b0; acc[2][1] += a2 * b1; acc[2][2] += a2 * b2;
    }
    // (the next chunk's barriers order these reads against its LDS writes)
  }
  const int n6 = 6 * b.W;
#pragma unroll
  for (int i = 0; i < 3; i++)
#pragma unroll
    for (int j = 0; j < 3; j++) {
      const int r = bi * BIG_TC + 3 * pr + i, c = bj * BIG_TC + 3 * pc + j;
      if (r >= n6 || c >= n6 || acc[i][j] == 0.0) continue;
      unsafeAtomicAdd(b.H + (size_t)r * n6 + c, acc[i][j]);
      if (bi != bj) unsafeAtomicAdd(b.H + (size_t)c * n6 + r, acc[i][j]);
    }
}

// ---------------------------------------------------------------- octree build with (node, frame) hashed body clusters
struct GbaBigView {
  // roots
  unsigned long long *hkeys; int *hvals; unsigned int hmask;
  // nodes
  int cap, W, npts;
  double *nadd, *ncenter, *neval, *nevec;
  float *nql;
  int *nchild, *nfac, *nexi;
  signed char *nlayer;
  // (node, frame) entries
  unsigned long long *ekeys; unsigned int emask; double *ecl;   // [10][emask + 1]
  // points
  double *pw; const double *pl; int *pframe, *pnode;
  int *perm;                   // points ordered by root voxel (the accumulation pass walks them in this order: see k_gbab_accum)
  unsigned int *skey; int *sval;   // sort input: root id (all ones = no root) / point index
  int *cnt; double *poses; int *offsets;
};

__device__ __forceinline__ int big_frame_of(const int *offsets, int W, int p) {   // largest f with offsets[f] <= p
  int lo = 0, hi = W;
  while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (offsets[mid] <= p) lo = mid; else hi = mid; }
  return lo;
}

__global__ void k_gbab_keys(GbaBigView g, GbaParams P) {
  const int p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= g.npts) return;
  const int f = big_frame_of(g.offsets, g.W, p);
  const double *R = g.poses + 12 * (size_t)f;
  const double x = g.pl[3 * (size_t)p], y = g.pl[3 * (size_t)p + 1], z = g.pl[3 * (size_t)p + 2];
  const double wx = (R[0] * x + R[1] * y + R[2] * z) + R[9], wy = (R[3] * x + R[4] * y + R[5] * z) + R[10], wz = (R[6] * x + R[7] * y + R[8] * z) + R[11];
  const size_t n = (size_t)g.npts;
  g.pw[p] = wx; g.pw[n + p] = wy; g.pw[2 * n + p] = wz;
  g.pframe[p] = f;
  const long long kx = key_axis(wx, P.voxel_size), ky = key_axis(wy, P.voxel_size), kz = key_axis(wz, P.voxel_size);
  if (kx < -KEY_OFF || kx >= KEY_OFF || ky < -KEY_OFF || ky >= KEY_OFF || kz < -KEY_OFF || kz >= KEY_OFF) { g.pnode[p] = -1; atomicExch(&g.cnt[GCNT_OVERFLOW], 2); return; }
  const unsigned long long key = pack_key(kx, ky, kz);
  unsigned int h = (unsigned int)((key * 0x9E3779B97F4A7C15ull) >> 32) & g.hmask;
  for (unsigned int probe = 0; probe <= g.hmask; probe++) {
    unsigned long long old = g.hkeys[h];                               // almost every point finds its root present: read before the CAS
    if (old == key) break;
    if (old == KEY_EMPTY) {
      old = atomicCAS(&g.hkeys[h], KEY_EMPTY, key);
      if (old == KEY_EMPTY || old == key) break;
    }
    h = (h + 1) & g.hmask;
  }
  g.pnode[p] = (int)h;
}
__global__ __launch_bounds__(256) void k_gbab_roots(GbaBigView g, GbaParams P) { gba_roots_body(g, P, false); }
__global__ void k_gbab_rootid(GbaBigView g) {
  const int p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= g.npts) return;
  const int s = g.pnode[p];
  const int id = s >= 0 ? g.hvals[s] : -1;
  if (s >= 0) g.pnode[p] = id;
  g.skey[p] = id >= 0 ? (unsigned int)id : (unsigned int)g.cap;      // "no root" sorts behind every id
  g.sval[p] = p;
}
// (LDS pre-aggregation per (node, frame) as k_gba_accum; the global (node, frame) table is probed once per occupied LDS entry).
// The points are walked in ROOT-VOXEL order (perm, one radix sort per build): the clouds of a top-level window arrive submap by
// submap in the down-sampler's hash order, so 256 consecutive points touched ~150 different (node, frame) pairs and the pass was
// bound by L2 atomic throughput (8 M points: 6-15 ms per level, 92 of the 187 ms of a 199-submap window); in root order a
// workgroup sees a handful of pairs.
__global__ __launch_bounds__(256) void k_gbab_accum(GbaBigView g) {
  __shared__ unsigned long long tkey[256];   // 256 points per workgroup -> at most 256 keys; 42 KB keeps three workgroups per CU
  __shared__ unsigned int tslot[256];
  __shared__ double tacc[20][256];
  const int tid = threadIdx.x;
  tkey[tid] = ~0ull;
  for (int t = tid; t < 20 * 256; t += 256) (&tacc[0][0])[t] = 0.0;
  __syncthreads();
  const int q = blockIdx.x * blockDim.x + tid;
  const size_t n = (size_t)g.npts, cp = (size_t)g.cap, ct = (size_t)g.emask + 1;
  if (q < g.npts) {
    const int p = g.perm[q];
    const int id = g.pnode[p];
    if (id >= 0) {
      const unsigned int e = gba_lds_claim(tkey, ((unsigned long long)(unsigned int)id << 20) | (unsigned int)g.pframe[p]);
      gba_lds_add(tacc, e, g.pl[3 * (size_t)p], g.pl[3 * (size_t)p + 1], g.pl[3 * (size_t)p + 2], g.pw[p], g.pw[n + p], g.pw[2 * n + p]);
    }
  }
  __syncthreads();
  for (int e = tid; e < 256; e += 256) {
    const unsigned long long key = tkey[e];
    if (key == ~0ull) continue;
    unsigned int h = (unsigned int)((key * 0x9E3779B97F4A7C15ull) >> 32) & g.emask;
    for (unsigned int probe = 0; probe <= g.emask; probe++) {
      const unsigned long long old = atomicCAS(&g.ekeys[h], KEY_EMPTY, key);
      if (old == KEY_EMPTY) { atomicAdd(&g.nexi[(int)(key >> 20)], 1); break; }   // a new (node, frame) pair: one more keyframe sees the node
      if (old == key) break;
      h = (h + 1) & g.emask;
    }
    tslot[e] = h;
  }
  __syncthreads();
  for (int t = tid; t < 20 * 256; t += 256) {
    const int k = t >> 8, e = t & 255;
    const unsigned long long key = tkey[e];
    if (key == ~0ull) continue;
    const double v = tacc[k][e];
    if (v == 0.0) continue;
    if (k < 10) unsafeAtomicAdd(g.nadd + (size_t)k * cp + (size_t)(key >> 20), v);
    else unsafeAtomicAdd(g.ecl + (size_t)(k - 10) * ct + tslot[e], v);
  }
}
__global__ void k_gbab_decide(GbaBigView g, GbaParams P, int layer) {
  const int id = blockIdx.x * blockDim.x + threadIdx.x;
  const int nn = g.cnt[GCNT_NODES] < g.cap ? g.cnt[GCNT_NODES] : g.cap;
  if (id >= nn || g.nlayer[id] != layer) return;
  const size_t cp = (size_t)g.cap;
  const double N = g.nadd[9 * cp + id];
  if (N <= 10.0) return;                                                               // LR:360
  const double cx = g.nadd[6 * cp + id] / N, cy = g.nadd[7 * cp + id] / N, cz = g.nadd[8 * cp + id] / N;
  double w0, w1, w2, V[9];
  eig3_sym_dev(g.nadd[id] / N - cx * cx, g.nadd[cp + id] / N - cx * cy, g.nadd[2 * cp + id] / N - cx * cz, g.nadd[3 * cp + id] / N - cy * cy,
               g.nadd[4 * cp + id] / N - cy * cz, g.nadd[5 * cp + id] / N - cz * cz, w0, w1, w2, V);
  if ((w0 < P.min_eigen_value) && ((w0 / w2) < P.eig_array[layer])) {                  // LR:310-314
    if (g.nexi[id] <= 1) return;                                                        // LR:371-375
    if (w0 / w1 > 0.12) return;                                                         // LR:377
    g.nfac[id] = atomicAdd(&g.cnt[GCNT_FACTORS], 1);
    g.neval[id] = w0; g.neval[cp + id] = w1; g.neval[2 * cp + id] = w2;
    for (int k = 0; k < 9; k++) g.nevec[(size_t)k * cp + id] = V[k];
    return;
  }
  if (layer >= P.max_layer) return;
  const int base = atomicAdd(&g.cnt[GCNT_NODES], 8);
  if (base + 8 > g.cap) { atomicExch(&g.cnt[GCNT_OVERFLOW], 1); return; }
  const float ql = g.nql[id];
  const double c0 = g.ncenter[id], c1 = g.ncenter[cp + id], c2 = g.ncenter[2 * cp + id];
  for (int o = 0; o < 8; o++) {
    const int ch = base + o;
    g.ncenter[ch] = c0 + (double)((float)(2 * ((o >> 2) & 1) - 1) * ql);
    g.ncenter[cp + ch] = c1 + (double)((float)(2 * ((o >> 1) & 1) - 1) * ql);
    g.ncenter[2 * cp + ch] = c2 + (double)((float)(2 * (o & 1) - 1) * ql);
    g.nql[ch] = ql / 2;
    g.nlayer[ch] = (signed char)(layer + 1); g.nchild[ch] = -1; g.nfac[ch] = -1;
  }
  g.nchild[id] = base;
}
__global__ void k_gbab_descend(GbaBigView g) {
  const int p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= g.npts) return;
  const int id = g.pnode[p];
  if (id < 0) return;
  const int base = g.nchild[id];
  if (base < 0) { g.pnode[p] = -1; return; }
  const size_t n = (size_t)g.npts, cp = (size_t)g.cap;
  g.pnode[p] = base + 4 * (g.pw[p] > g.ncenter[id] ? 1 : 0) + 2 * (g.pw[n + p] > g.ncenter[cp + id] ? 1 : 0) + (g.pw[2 * n + p] > g.ncenter[2 * cp + id] ? 1 : 0);
}
// planar voxels -> CSR.  vcnt[a] = entries of factor a; after the scan, the table slots are scattered to their rows.
__global__ void k_gbab_vcount(GbaBigView g, int *vcnt) {
  const int id = blockIdx.x * blockDim.x + threadIdx.x;
  const int nn = g.cnt[GCNT_NODES] < g.cap ? g.cnt[GCNT_NODES] : g.cap;
  if (id >= nn) return;
  const int a = g.nfac[id];
  if (a >= 0) vcnt[a] = g.nexi[id];
}
__global__ __launch_bounds__(256) void k_big_scan(int n, const int *__restrict__ in, int *__restrict__ out /*[n+1]*/) {   // one workgroup, exclusive
  __shared__ int part[256];
  const int t = threadIdx.x, per = (n + 255) / 256;
  int s = 0;
  for (int k = 0; k < per; k++) { const int i = t * per + k; if (i < n) s += in[i]; }
  part[t] = s;
  __syncthreads();
  if (t == 0) { int acc = 0; for (int k = 0; k < 256; k++) { const int v = part[k]; part[k] = acc; acc += v; } out[n] = acc; }
  __syncthreads();
  int acc = part[t];
  for (int k = 0; k < per; k++) { const int i = t * per + k; if (i < n) { out[i] = acc; acc += in[i]; } }
}
__global__ void k_gbab_fill(GbaBigView g, BigView b, int *fill) {
  const unsigned int s = blockIdx.x * blockDim.x + threadIdx.x;
  if (s > g.emask) return;
  const unsigned long long key = g.ekeys[s];
  if (key == KEY_EMPTY) return;
  const int id = (int)(key >> 20), fr = (int)(key & 0xFFFFF);
  const int a = g.nfac[id];
  if (a < 0) return;
  const int pos = b.vptr[a] + atomicAdd(&fill[a], 1);
  const size_t ct = (size_t)g.emask + 1, ce = (size_t)b.capE;
  b.efr[pos] = fr; b.evox[pos] = a;
  for (int k = 0; k < 10; k++) b.ecl[(size_t)k * ce + pos] = g.ecl[(size_t)k * ct + s];
}
__global__ void k_gbab_voxels(GbaBigView g, BigView b) {
  const int id = blockIdx.x * blockDim.x + threadIdx.x;
  const int nn = g.cnt[GCNT_NODES] < g.cap ? g.cnt[GCNT_NODES] : g.cap;
  if (id >= nn) return;
  const int a = g.nfac[id];
  if (a < 0) return;
  const size_t cp = (size_t)g.cap, cv = (size_t)b.capV;
  for (int k = 0; k < 3; k++) b.eval[(size_t)k * cv + a] = g.neval[(size_t)k * cp + id];
  for (int k = 0; k < 9; k++) b.evec[(size_t)k * cv + a] = g.nevec[(size_t)k * cp + id];
  for (int k = 0; k < 10; k++) b.pcr[(size_t)k * cv + a] = g.nadd[(size_t)k * cp + id];
}

// ---------------------------------------------------------------- dense LDL^T of the (6W)^2 system in HBM
// Right-looking, panels of 8 columns, static pivot order (Eigen's "largest |diagonal| first" as a permutation computed by
// the caller).  Ab = (NP + 1) x ld row-major, lower triangle of P (H + uD) P^T, identity on the padding n..NP-1, row NP =
// the right-hand side (so D^-1 L^-1 b falls out as that row of L).  Two kernels per panel:
//   k_bigl_panel   one workgroup: every thread factorises the 8x8 diagonal block in registers (as vba_ldlt.hpp) and
//                  substitutes the rows it owns; L overwrites the panel in place, -T = -L D goes to Tb[row][8];
//   k_bigl_update  one workgroup per 64x64 tile of the trailing lower triangle: C -= L (T)^T on the matrix cores
//                  (v_mfma_f64_16x16x4_f64, two per 16x16 sub-tile, operands staged in LDS).
__global__ void k_bigl_setup(const double *__restrict__ H, const double *__restrict__ g, const int *__restrict__ ord, int n, int NP, int ld, double u,
                             double *__restrict__ Ab) {
  const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  const long long tot = (long long)(NP + 1) * NP;
  if (t >= tot) return;
  const int i = (int)(t / NP), j = (int)(t - (long long)i * NP);
  double a = 0.0;
  if (j >= n) a = (i == j) ? 1.0 : 0.0;                      // padding columns
  else if (i == NP) { const int pj = ord[j]; a = (pj < 6) ? 0.0 : -g[pj]; }      // rhs row: -JacT, gauge rows zeroed (VM:455)
  else if (i >= n || i < j) a = 0.0;
  else {
    const int pi = ord[i], pj = ord[j];
    const int rr = pi > pj ? pi : pj, cc = pi > pj ? pj : pi;
    a = (rr < 6 || cc < 6) ? ((rr == cc) ? 1.0 : 0.0) : H[(size_t)rr * n + cc];
    if (i == j) a += u * a;
  }
  Ab[(size_t)i * ld + j] = a;
}

__global__ __launch_bounds__(256) void k_bigl_panel(double *__restrict__ Ab, double *__restrict__ Tb, int NP, int ld, int k0) {
  double D[8][8];
#pragma unroll
  for (int r = 0; r < 8; r++)
#pragma unroll
    for (int c = 0; c <= r; c++) D[r][c] = Ab[(size_t)(k0 + r) * ld + k0 + c];
  double dv[8], di[8];
  bool okv[8];
#pragma unroll
  for (int c = 0; c < 8; c++) {
    const double d = D[c][c];
    const bool ok = fabs(d) > 0.0;
    const double y0 = __builtin_amdgcn_rcp(d);
    const double e = fma(-d, y0, 1.0);
    const double y1 = fma(e, y0, y0), e2 = e * e;
    const double inv = fma(e2, y1, y1);
    const double dinv = ok ? inv : 0.0;
    dv[c] = d; di[c] = dinv; okv[c] = ok;
#pragma unroll
    for (int r = c + 1; r < 8; r++)
#pragma unroll
      for (int c2 = c + 1; c2 <= r; c2++) D[r][c2] = fma(-(D[r][c] * D[c2][c]), dinv, D[r][c2]);
  }
  // D[r][c] (r > c) now holds the UNSCALED column values (T of the block); L = T / d
  __syncthreads();                                            // every thread has read the diagonal block before rows are rewritten
  for (int i = k0 + threadIdx.x; i <= NP; i += 256) {
    const int ib = i - k0;
    double x[8];
#pragma unroll
    for (int c = 0; c < 8; c++) x[c] = Ab[(size_t)i * ld + k0 + c];
#pragma unroll
    for (int c = 0; c < 8; c++) {
      double lv = okv[c] ? x[c] * di[c] : x[c];
      if (i == NP) lv = (fabs(dv[c]) > 2.2250738585072014e-308) ? x[c] * di[c] : 0.0;
      lv = (ib > c) ? lv : 0.0;
      const double tmc = (okv[c] && ib > c) ? x[c] : 0.0;
      if (ib > c) Ab[(size_t)i * ld + k0 + c] = lv;            // L in place (the diagonal keeps d)
      Tb[(size_t)i * 8 + c] = -tmc;
#pragma unroll
      for (int c2 = c + 1; c2 < 8; c2++) x[c2] = fma(-(tmc * D[c2][c]), di[c], x[c2]);
    }
    if (ib < 8) Ab[(size_t)i * ld + k0 + ib] = dv[ib];        // (unchanged value, written for clarity of the layout)
  }
}

__global__ __launch_bounds__(256) void k_bigl_update(double *__restrict__ Ab, const double *__restrict__ Tb, int NP, int ld, int k0) {
  __shared__ double Ls[64][9], Ts[64][9];
  // tile (ti, tj), ti >= tj, of the trailing block that starts at row / column kn = k0 + 8
  const int kn = k0 + 8;
  int tj = 0, rem = blockIdx.x;
  const int nt = (NP + 1 - kn + 63) / 64;
  while (rem >= nt - tj) { rem -= nt - tj; tj++; }
  const int ti = tj + rem;
  const int r0 = kn + 64 * ti, c0 = kn + 64 * tj;
  const int tid = threadIdx.x;
  for (int t = tid; t < 64 * 8; t += 256) {
    const int r = t >> 3, c = t & 7;
    const int gi = r0 + r, gj = c0 + r;
    Ls[r][c] = (gi <= NP) ? ((gi - k0 > c) ? Ab[(size_t)gi * ld + k0 + c] : 0.0) : 0.0;
    Ts[r][c] = (gj < NP) ? Tb[(size_t)gj * 8 + c] : 0.0;
  }
  __syncthreads();
  const int l = tid & 63, w = tid >> 6, lr = l >> 4, lc = l & 15;
#pragma unroll
  for (int q = 0; q < 4; q++) {
    const int st = w * 4 + q, si = st >> 2, sj = st & 3;      // 16 sub-tiles of 16 x 16, four per wave
    if (ti == tj && sj > si) continue;                        // strictly upper sub-tiles of a diagonal tile
    v4f64_l acc;
#pragma unroll
    for (int r = 0; r < 4; r++) {
      const int gi = r0 + 16 * si + lr + 4 * r, gj = c0 + 16 * sj + lc;
      acc[r] = (gi <= NP && gj < NP) ? Ab[(size_t)gi * ld + gj] : 0.0;
    }
    acc = __builtin_amdgcn_mfma_f64_16x16x4f64(Ls[16 * si + lc][lr], Ts[16 * sj + lc][lr], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f64_16x16x4f64(Ls[16 * si + lc][4 + lr], Ts[16 * sj + lc][4 + lr], acc, 0, 0, 0);
#pragma unroll
    for (int r = 0; r < 4; r++) {
      const int gi = r0 + 16 * si + lr + 4 * r, gj = c0 + 16 * sj + lc;
      if (gi <= NP && gj < NP && gj <= gi) Ab[(size_t)gi * ld + gj] = acc[r];
    }
  }
}

// ---------------------------------------------------------------- host side
// diag(H) (the host-side LM bookkeeping needs only it and g: q1 of VM:465, Eigen's pivot order) and, for the edges of
// HBA_add_edge (VS:2926-2951), the six diagonal entries of every 6x6 cross block: out[(i W + j) 6 + k] = H(6 i + k, 6 j + k).
__global__ void k_big_getdiag(const double *__restrict__ H, int n, double *__restrict__ out) {
  const int r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r < n) out[r] = H[(size_t)r * n + r];
}
__global__ void k_big_blockdiag(const double *__restrict__ H, int W, double *__restrict__ out) {
  const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= (long long)W * W * 6) return;
  const int k = (int)(t % 6), j = (int)((t / 6) % W), i = (int)(t / (6LL * W));
  out[t] = H[(size_t)(6 * i + k) * (6 * (size_t)W) + 6 * j + k];
}
// Back substitution x = L^-T z on the device, 64 unknowns per step from the bottom (Ab: rows of L, row NP = z, row stride ld):
//   k_bigl_bs_tri : one wave solves the 64x64 triangular diagonal block in place (lane j owns x[lo + j], v_readlane broadcasts);
//   k_bigl_bs_gemv: every unknown above the block subtracts the block's contribution, z[j] -= sum_i L[i][j] x[i].
// (The host version fetched the whole factor, 8 (NP+1) ld bytes — 47 MB at 400 submaps — for an O(n^2) loop.)
__global__ __launch_bounds__(64) void k_bigl_bs_tri(double *__restrict__ Ab, int NP, int ld, int n, int lo) {
  const int j = threadIdx.x, hi = (lo + 63 < n - 1) ? lo + 63 : n - 1;
  double *z = Ab + (size_t)NP * ld;
  double x = (lo + j <= hi) ? z[lo + j] : 0.0;
  for (int i = hi; i > lo; i--) {
    const double xi = readlane_f64(x, i - lo);
    const double l = (lo + j < i) ? Ab[(size_t)i * ld + lo + j] : 0.0;
    x -= l * xi;
  }
  if (lo + j <= hi) z[lo + j] = x;
}
__global__ __launch_bounds__(256) void k_bigl_bs_gemv(double *__restrict__ Ab, int NP, int ld, int n, int lo) {
  __shared__ double xs[64];
  const int hi = (lo + 63 < n - 1) ? lo + 63 : n - 1, cnt = hi - lo + 1;
  double *z = Ab + (size_t)NP * ld;
  if (threadIdx.x < 64) xs[threadIdx.x] = (int)threadIdx.x < cnt ? z[lo + threadIdx.x] : 0.0;
  __syncthreads();
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= lo) return;
  double acc = z[j];
  const double *col = Ab + (size_t)lo * ld + j;
#pragma unroll 8
  for (int i = 0; i < cnt; i++) acc -= col[(size_t)i * ld] * xs[i];
  z[j] = acc;
}
__global__ void k_bigl_bs_out(const double *__restrict__ Ab, int NP, int ld, int n, const int *__restrict__ ord, double *__restrict__ dxi) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) dxi[ord[i]] = Ab[(size_t)NP * ld + i];
}

struct BigStore {
  int last_cap = 1 << 17;
  BigView b{};
  GbaBigView g{};
  // Device memory comes from an arena of large chunks that survives across builds (reset() rewinds it): hipMalloc / hipFree
  // of ~40 buffers per build, some of them 10^8 bytes, cost more than the kernels of a top-level window.
  struct Chunk { char *base; size_t size, used; };
  std::vector<Chunk> chunks;
  hipError_t arena(void **p, size_t bytes) {
    bytes = (bytes ? bytes : 8) + 255 & ~(size_t)255;
    for (Chunk &ck : chunks)
      if (ck.size - ck.used >= bytes) { *p = ck.base + ck.used; ck.used += bytes; return hipSuccess; }
    Chunk ck{nullptr, bytes > ((size_t)256 << 20) ? bytes : ((size_t)256 << 20), 0};
    hipError_t e = hipMalloc((void **)&ck.base, ck.size);
    if (e != hipSuccess) return e;
    ck.used = bytes; *p = ck.base;
    chunks.push_back(ck);
    return hipSuccess;
  }
  void reset() { for (Chunk &ck : chunks) ck.used = 0; b = BigView(); g = GbaBigView(); }
  int *h_cnt = nullptr;
  int *d_vcnt = nullptr, *d_fill = nullptr;
  double *d_Ab = nullptr, *d_Tb = nullptr; int *d_ord = nullptr;   // dense solver (allocated by big_build)
  double *d_vec = nullptr;                                          // [3 n + 6 W W]: diag(H) | g copy | dxi | cross-block diagonals
  int NP = 0, ld = 0;
  void release() { for (Chunk &ck : chunks) hipFree(ck.base); chunks.clear(); if (h_cnt) hipHostFree(h_cnt); h_cnt = nullptr; b = BigView(); g = GbaBigView(); }
};

#define BIGCHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { err = std::string(#x) + ": " + hipGetErrorString(e_); return VBA_ERR_HIP; } } while (0)

// Builds the octree of `W` keyframes and the sparse factor store (everything is re-allocated per call: the top-level BA
// runs once per loop closure).  pl: device pointer to the local points [n][3].
inline int big_build(BigStore &s, hipStream_t st, int W, const int *offsets, const double *d_pl, const double *poses, const GbaParams &P, std::string &err) {
  s.reset();
  const int n = offsets[W];
  auto al = [&](void **p, size_t bytes) { return s.arena(p, bytes); };
  if (!s.h_cnt) BIGCHK(hipHostMalloc((void **)&s.h_cnt, GCNT_N * sizeof(int), hipHostMallocDefault));
  GbaBigView &g = s.g;
  g.W = W; g.npts = n; g.pl = d_pl;
  unsigned int hcap = 1u << 16; while (hcap < 2u * (unsigned)n && hcap < (1u << 28)) hcap <<= 1;
  unsigned int ecap = 1u << 16;                         // (node, frame) pairs of ALL levels share the table: <= points per level
  while ((unsigned long long)ecap < 2ull * (unsigned long long)n * (unsigned)(P.max_layer + 1) && ecap < (1u << 30)) ecap <<= 1;
  g.hmask = hcap - 1; g.emask = ecap - 1;
  BIGCHK(al((void **)&g.hkeys, (size_t)hcap * 8)); BIGCHK(al((void **)&g.hvals, (size_t)hcap * 4));
  BIGCHK(al((void **)&g.ekeys, (size_t)ecap * 8)); BIGCHK(al((void **)&g.ecl, (size_t)ecap * 10 * 8));
  BIGCHK(al((void **)&g.pw, (size_t)n * 3 * 8)); BIGCHK(al((void **)&g.pframe, (size_t)n * 4)); BIGCHK(al((void **)&g.pnode, (size_t)n * 4));
  unsigned int *skey_b = nullptr; void *sort_tmp = nullptr; size_t sort_bytes = 0;
  BIGCHK(al((void **)&g.skey, (size_t)n * 4)); BIGCHK(al((void **)&skey_b, (size_t)n * 4)); BIGCHK(al((void **)&g.sval, (size_t)n * 4)); BIGCHK(al((void **)&g.perm, (size_t)n * 4));
  if (n > 0) {
    BIGCHK(sort_pairs_u32(nullptr, sort_bytes, g.skey, skey_b, g.sval, g.perm, (size_t)n, 32u, st));
    BIGCHK(al(&sort_tmp, sort_bytes + 256));
  }
  BIGCHK(al((void **)&g.cnt, GCNT_N * sizeof(int))); BIGCHK(al((void **)&g.poses, (size_t)W * 12 * 8)); BIGCHK(al((void **)&g.offsets, (size_t)(W + 1) * 4));
  BIGCHK(hipMemcpyAsync(g.poses, poses, (size_t)W * 12 * 8, hipMemcpyHostToDevice, st));
  BIGCHK(hipMemcpyAsync(g.offsets, offsets, (size_t)(W + 1) * 4, hipMemcpyHostToDevice, st));
  int cap = s.last_cap;                          // (the node capacity the previous build ended with: no doubling attempts, each of which re-clears the tables)
  const dim3 bk(256), gp((n + 255) / 256);
  for (int attempt = 0; attempt < 10; attempt++) {
    const size_t cp = (size_t)cap;
    g.cap = cap;
    void *tmp[9];
    size_t sz[9] = {10 * cp * 8, 3 * cp * 8, 3 * cp * 8, 9 * cp * 8, cp * 4, cp * 4, cp * 4, cp * 4, cp};
    for (int k = 0; k < 9; k++) { if (s.arena(&tmp[k], sz[k]) != hipSuccess) { err = "octree node storage"; return VBA_ERR_HIP; } }
    g.nadd = (double *)tmp[0]; g.ncenter = (double *)tmp[1]; g.neval = (double *)tmp[2]; g.nevec = (double *)tmp[3]; g.nql = (float *)tmp[4];
    g.nchild = (int *)tmp[5]; g.nfac = (int *)tmp[6]; g.nexi = (int *)tmp[7]; g.nlayer = (signed char *)tmp[8];
    BIGCHK(hipMemsetAsync(g.cnt, 0, GCNT_N * sizeof(int), st));
    BIGCHK(hipMemsetAsync(g.hkeys, 0xFF, (size_t)hcap * 8, st));
    BIGCHK(hipMemsetAsync(g.nadd, 0, 10 * cp * 8, st));
    BIGCHK(hipMemsetAsync(g.nexi, 0, cp * 4, st));
    if (n > 0) {
      hipLaunchKernelGGL(k_gbab_keys, gp, bk, 0, st, g, P);
      hipLaunchKernelGGL(k_gbab_roots, dim3((hcap + 4095) / 4096), bk, 0, st, g, P);
      hipLaunchKernelGGL(k_gbab_rootid, gp, bk, 0, st, g);
      {
        unsigned int bits = 1; while (bits < 32 && (1ull << bits) <= (unsigned long long)cap) bits++;     // keys are <= cap
        size_t tb = sort_bytes + 256;
        BIGCHK(sort_pairs_u32(sort_tmp, tb, g.skey, skey_b, g.sval, g.perm, (size_t)n, bits, st));
      }
      for (int L = 0; L <= P.max_layer; L++) {
        // entries of the previous level are dead: a planar node keeps its own entries (it stopped descending), so the
        // table is only cleared of nothing here — finished nodes never receive points again and their keys stay valid
        if (L == 0) {   // (a fill kernel: the runtime's memset moved the 5.4 GB of an 8 M-point window at 750 GB/s, 7.2 ms a time)
          hipLaunchKernelGGL(k_fill_u64, dim3(4096), dim3(256), 0, st, g.ekeys, ~0ull, (size_t)ecap);
          hipLaunchKernelGGL(k_fill_u64, dim3(4096), dim3(256), 0, st, (unsigned long long *)g.ecl, 0ull, (size_t)ecap * 10);
        }
        hipLaunchKernelGGL(k_gbab_accum, gp, bk, 0, st, g);
        hipLaunchKernelGGL(k_gbab_decide, dim3((cap + 255) / 256), bk, 0, st, g, P, L);
        if (L < P.max_layer) hipLaunchKernelGGL(k_gbab_descend, gp, bk, 0, st, g);
      }
    }
    BIGCHK(hipGetLastError());
    BIGCHK(hipStreamSynchronize(st));
    BIGCHK(hipMemcpyAsync(s.h_cnt, g.cnt, GCNT_N * sizeof(int), hipMemcpyDeviceToHost, st));
    BIGCHK(hipStreamSynchronize(st));
    if (s.h_cnt[GCNT_OVERFLOW] == 2) { err = "keyframe point outside the 21-bit voxel index range"; return VBA_ERR_CAPACITY; }
    if (!s.h_cnt[GCNT_OVERFLOW]) { s.last_cap = cap; break; }
    // (the undersized node arrays stay in the arena until the next build rewinds it)
    cap *= 2;
    if (attempt == 9) { err = "octree node capacity"; return VBA_ERR_CAPACITY; }
  }
  // sparse factor store
  BigView &b = s.b;
  const int V = s.h_cnt[GCNT_FACTORS];
  b.W = W; b.V = V; b.capV = V > 0 ? V : 1;
  BIGCHK(al((void **)&b.vptr, (size_t)(V + 1) * 4)); BIGCHK(al((void **)&s.d_vcnt, (size_t)b.capV * 4)); BIGCHK(al((void **)&s.d_fill, (size_t)b.capV * 4));
  BIGCHK(al((void **)&b.eval, (size_t)b.capV * 3 * 8)); BIGCHK(al((void **)&b.evec, (size_t)b.capV * 9 * 8)); BIGCHK(al((void **)&b.pcr, (size_t)b.capV * 10 * 8));
  BIGCHK(al((void **)&b.poses, (size_t)W * 12 * 8));
  const size_t n6 = (size_t)6 * W;
  BIGCHK(al((void **)&b.H, n6 * n6 * 8)); BIGCHK(al((void **)&b.g, n6 * 8)); BIGCHK(al((void **)&b.r, 8));

  s.NP = (int)((n6 + 7) / 8 * 8); s.ld = (int)((s.NP + 63) / 64 * 64);
  BIGCHK(al((void **)&s.d_Ab, (size_t)(s.NP + 1) * s.ld * 8)); BIGCHK(al((void **)&s.d_Tb, (size_t)(s.NP + 1) * 8 * 8)); BIGCHK(al((void **)&s.d_ord, n6 * 4)); BIGCHK(al((void **)&s.d_vec, ((size_t)3 * n6 + (size_t)6 * W * W) * 8));
  BIGCHK(hipMemsetAsync(s.d_fill, 0, (size_t)b.capV * 4, st));
  BIGCHK(hipMemsetAsync(b.vptr, 0, (size_t)(V + 1) * 4, st));
  int E = 0;
  if (V > 0) {
    const int nn = s.h_cnt[GCNT_NODES] < g.cap ? s.h_cnt[GCNT_NODES] : g.cap;
    hipLaunchKernelGGL(k_gbab_vcount, dim3((nn + 255) / 256), bk, 0, st, g, s.d_vcnt);
    hipLaunchKernelGGL(k_big_scan, dim3(1), bk, 0, st, V, s.d_vcnt, b.vptr);
    BIGCHK(hipStreamSynchronize(st));
    BIGCHK(hipMemcpyAsync(&E, b.vptr + V, 4, hipMemcpyDeviceToHost, st));
    BIGCHK(hipStreamSynchronize(st));
  }
  b.E = E; b.capE = E > 0 ? E : 1;
  BIGCHK(al((void **)&b.efr, (size_t)b.capE * 4)); BIGCHK(al((void **)&b.evox, (size_t)b.capE * 4));
  BIGCHK(al((void **)&b.ecl, (size_t)b.capE * 10 * 8)); BIGCHK(al((void **)&b.gv, (size_t)b.capE * 18 * 8)); BIGCHK(al((void **)&b.es, (size_t)b.capE * 27 * 8));
  if (V > 0) {
    const int nn = s.h_cnt[GCNT_NODES] < g.cap ? s.h_cnt[GCNT_NODES] : g.cap;
    hipLaunchKernelGGL(k_gbab_fill, dim3((ecap + 255) / 256), bk, 0, st, g, b, s.d_fill);
    hipLaunchKernelGGL(k_gbab_voxels, dim3((nn + 255) / 256), bk, 0, st, g, b);
    BIGCHK(hipGetLastError());
  }
  BIGCHK(al((void **)&b.eidx, (size_t)b.capV * W * 4));
  BIGCHK(hipMemsetAsync(b.eidx, 0xFF, (size_t)b.capV * W * 4, st));
  if (V > 0 && E > 0) {
    hipLaunchKernelGGL(k_big_eidx, dim3((E + 255) / 256), bk, 0, st, b);
    BIGCHK(hipGetLastError());
  }
  return VBA_OK;
}

// divide_thread (VM:347-389): H, g, r at `poses` on the host side buffers (full layout)
// Hessian pass on the sparse store: H (n x n) and g stay in HBM (the solver reads them there); the host gets diag(H), g and r.
inline int big_hessian(BigStore &s, hipStream_t st, const double *poses, double *hdiag, double *gvec, double *r, std::string &err) {
  BigView &b = s.b;
  const size_t n6 = (size_t)6 * b.W;
  BIGCHK(hipMemcpyAsync(b.poses, poses, (size_t)b.W * 12 * 8, hipMemcpyHostToDevice, st));
  BIGCHK(hipMemsetAsync(b.H, 0, n6 * n6 * 8, st)); BIGCHK(hipMemsetAsync(b.g, 0, n6 * 8, st)); BIGCHK(hipMemsetAsync(b.r, 0, 8, st));
  if (b.E > 0) {
    hipLaunchKernelGGL(k_big_slot, dim3((b.E + 127) / 128), dim3(128), 0, st, b);
    const int nt = (b.W + BIG_TF - 1) / BIG_TF, npair = nt * (nt + 1) / 2, nchunk = (b.V + BIG_VC - 1) / BIG_VC;
    int nslice = (2048 + npair - 1) / npair;
    if (nslice > nchunk) nslice = nchunk;
    if (nslice < 1) nslice = 1;
    hipLaunchKernelGGL(k_big_syrk, dim3(npair, nslice), dim3(256), 0, st, b, nt, nslice);
    hipLaunchKernelGGL(k_big_diag, dim3(b.W), dim3(256), 0, st, b);   // after the SYRK atomics on H (stream order)
  }
  hipLaunchKernelGGL(k_big_getdiag, dim3((unsigned)((n6 + 255) / 256)), dim3(256), 0, st, b.H, (int)n6, s.d_vec);
  BIGCHK(hipGetLastError());
  BIGCHK(hipStreamSynchronize(st));
  BIGCHK(hipMemcpyAsync(hdiag, s.d_vec, n6 * 8, hipMemcpyDeviceToHost, st));
  BIGCHK(hipMemcpyAsync(gvec, b.g, n6 * 8, hipMemcpyDeviceToHost, st));
  BIGCHK(hipMemcpyAsync(r, b.r, 8, hipMemcpyDeviceToHost, st));
  BIGCHK(hipStreamSynchronize(st));
  return VBA_OK;
}
// the six diagonal entries of every 6x6 cross block of the Hessian of the last big_hessian (before the gauge): [W][W][6]
inline int big_block_diagonals(BigStore &s, hipStream_t st, double *out, std::string &err) {
  const int W = s.b.W;
  const size_t n6 = (size_t)6 * W, cnt = (size_t)6 * W * W;
  hipLaunchKernelGGL(k_big_blockdiag, dim3((unsigned)((cnt + 255) / 256)), dim3(256), 0, st, s.b.H, W, s.d_vec + 3 * n6);
  BIGCHK(hipGetLastError());
  BIGCHK(hipStreamSynchronize(st));
  BIGCHK(hipMemcpyAsync(out, s.d_vec + 3 * n6, cnt * 8, hipMemcpyDeviceToHost, st));
  BIGCHK(hipStreamSynchronize(st));
  return VBA_OK;
}
// only_residual (VM:391-420): also refreshes the per-voxel eigen state
inline int big_residual(BigStore &s, hipStream_t st, const double *poses, double *r, std::string &err) {
  BigView &b = s.b;
  BIGCHK(hipMemcpyAsync(b.poses, poses, (size_t)b.W * 12 * 8, hipMemcpyHostToDevice, st));
  BIGCHK(hipMemsetAsync(b.r, 0, 8, st));
  if (b.V > 0) hipLaunchKernelGGL(k_big_residual, dim3((b.V + 255) / 256), dim3(256), 0, st, b);
  BIGCHK(hipGetLastError());
  BIGCHK(hipStreamSynchronize(st));
  BIGCHK(hipMemcpyAsync(r, b.r, 8, hipMemcpyDeviceToHost, st));
  BIGCHK(hipStreamSynchronize(st));
  return VBA_OK;
}

// (H + u D) dxi = -g with the gauge of VM:452-455, H / g = the device buffers of the last big_hessian (before the gauge).
// ord = Eigen's pivot order (host).  Factorisation and back substitution run on the device; the host gets dxi (n doubles).
inline int big_solve(BigStore &s, hipStream_t st, const int *ord, double u, double *dxi, std::string &err) {
  const int n = 6 * s.b.W, NP = s.NP, ld = s.ld;
  BIGCHK(hipMemcpyAsync(s.d_ord, ord, (size_t)n * 4, hipMemcpyHostToDevice, st));
  const long long tot = (long long)(NP + 1) * NP;
  hipLaunchKernelGGL(k_bigl_setup, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, st, s.b.H, s.b.g, s.d_ord, n, NP, ld, u, s.d_Ab);
  for (int k0 = 0; k0 < NP; k0 += 8) {
    hipLaunchKernelGGL(k_bigl_panel, dim3(1), dim3(256), 0, st, s.d_Ab, s.d_Tb, NP, ld, k0);
    const int kn = k0 + 8;
    if (kn <= NP) {
      const int nt = (NP + 1 - kn + 63) / 64;
      if (nt > 0) hipLaunchKernelGGL(k_bigl_update, dim3(nt * (nt + 1) / 2), dim3(256), 0, st, s.d_Ab, s.d_Tb, NP, ld, k0);
    }
  }
  // back substitution on the device, 64 unknowns per step from the bottom
  for (int lo = ((n - 1) / 64) * 64; lo >= 0; lo -= 64) {
    hipLaunchKernelGGL(k_bigl_bs_tri, dim3(1), dim3(64), 0, st, s.d_Ab, NP, ld, n, lo);
    if (lo > 0) hipLaunchKernelGGL(k_bigl_bs_gemv, dim3((lo + 255) / 256), dim3(256), 0, st, s.d_Ab, NP, ld, n, lo);
  }
  double *d_dxi = s.d_vec + 2 * (size_t)n;
  hipLaunchKernelGGL(k_bigl_bs_out, dim3((n + 255) / 256), dim3(256), 0, st, s.d_Ab, NP, ld, n, s.d_ord, d_dxi);
  BIGCHK(hipGetLastError());
  BIGCHK(hipStreamSynchronize(st));
  BIGCHK(hipMemcpyAsync(dxi, d_dxi, (size_t)n * 8, hipMemcpyDeviceToHost, st));
  BIGCHK(hipStreamSynchronize(st));
  return VBA_OK;
}

}  // namespace vba
