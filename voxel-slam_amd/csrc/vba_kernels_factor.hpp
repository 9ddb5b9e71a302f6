// HIP kernels (gfx950 / CDNA4, wave64) for the factor level of the Voxel-SLAM local BA:
//   k_residual   (K4)  <->  LidarFactor::evaluate_only_residual   voxel_map.hpp:285-325 + tools.hpp:357-363
//   k_hessian<W> (K3)  <->  LidarFactor::acc_evaluate2            voxel_map.hpp:150-282
//   k_reduce_partials  <->  the thread-sum after join()           voxel_map.hpp:376-386, 571-581
// Data layout in HBM: SoA [field][frame][voxel] (voxel fastest) so that one wave reads 64 consecutive
// doubles (512 B) per field — see DESIGN.md §3.
#pragma once
#include <hip/hip_runtime.h>

namespace vba {

struct FactorView {
  double *cl;      // [10][W][vs]   body-frame clusters per (frame, voxel): Pxx,Pxy,Pxz,Pyy,Pyz,Pzz,vx,vy,vz,N
  double *fix;     // [10][vs]      sig_vecs (fixed world cluster)
  double *coe;     // [vs]
  double *eigval;  // [3][vs]
  double *eigvec;  // [9][vs]       row-major r*3+c, column c = eigenvector c
  double *pcr;     // [10][vs]      pcr_adds
  int vs;          // voxel stride (capacity)
  int W;
};

// ------------------------------------------------------------------------------------------------
// Symmetric 3x3 eigen-decomposition, ascending eigenvalues, orthonormal eigenvectors in columns.
// Cyclic Jacobi in registers (no indexed arrays -> no scratch).  Replaces Eigen::SelfAdjointEigenSolver
// at voxel_map.hpp:312 / :1416 / :1525 (result equal up to rounding and eigenvector sign).
__device__ __forceinline__ void jacobi_rot(double &app, double &aqq, double &apq, double &arp, double &arq,
                                           double &v0p, double &v0q, double &v1p, double &v1q, double &v2p, double &v2q,
                                           int sweep) {
  if (apq == 0.0) return;
  const double g = 100.0 * fabs(apq);
  if (sweep > 3 && fabs(app) + g == fabs(app) && fabs(aqq) + g == fabs(aqq)) { apq = 0.0; return; }
  const double h = aqq - app;
  double t;
  if (fabs(h) + g == fabs(h)) {
    t = apq / h;
  } else {
    const double theta = 0.5 * h / apq;
    t = 1.0 / (fabs(theta) + sqrt(1.0 + theta * theta));
    if (theta < 0.0) t = -t;
  }
  const double c = 1.0 / sqrt(1.0 + t * t), s = t * c;
  app -= t * apq;
  aqq += t * apq;
  apq = 0.0;
  double x = arp, y = arq;
  arp = c * x - s * y; arq = s * x + c * y;
  x = v0p; y = v0q; v0p = c * x - s * y; v0q = s * x + c * y;
  x = v1p; y = v1q; v1p = c * x - s * y; v1q = s * x + c * y;
  x = v2p; y = v2q; v2p = c * x - s * y; v2q = s * x + c * y;
}

#define VBA_SWAP(a, b) { double _t = a; a = b; b = _t; }

// in: lower triangle a00,a10,a20,a11,a21,a22.  out: w0<=w1<=w2, V (row-major, columns = eigenvectors)
__device__ __forceinline__ void eig3_sym_dev(double a00, double a01, double a02, double a11, double a12, double a22,
                                             double &w0, double &w1, double &w2, double *V) {
  double v00 = 1, v01 = 0, v02 = 0, v10 = 0, v11 = 1, v12 = 0, v20 = 0, v21 = 0, v22 = 1;
  for (int sweep = 0; sweep < 30; sweep++) {
    if (fabs(a01) + fabs(a02) + fabs(a12) == 0.0) break;
    jacobi_rot(a00, a11, a01, a02, a12, v00, v01, v10, v11, v20, v21, sweep);  // (p,q)=(0,1), r=2
    jacobi_rot(a00, a22, a02, a01, a12, v00, v02, v10, v12, v20, v22, sweep);  // (0,2), r=1
    jacobi_rot(a11, a22, a12, a01, a02, v01, v02, v11, v12, v21, v22, sweep);  // (1,2), r=0
  }
  if (a11 < a00) { VBA_SWAP(a00, a11); VBA_SWAP(v00, v01); VBA_SWAP(v10, v11); VBA_SWAP(v20, v21); }
  if (a22 < a00) { VBA_SWAP(a00, a22); VBA_SWAP(v00, v02); VBA_SWAP(v10, v12); VBA_SWAP(v20, v22); }
  if (a22 < a11) { VBA_SWAP(a11, a22); VBA_SWAP(v01, v02); VBA_SWAP(v11, v12); VBA_SWAP(v21, v22); }
  w0 = a00; w1 = a11; w2 = a22;
  V[0] = v00; V[1] = v01; V[2] = v02; V[3] = v10; V[4] = v11; V[5] = v12; V[6] = v20; V[7] = v21; V[8] = v22;
}

// ------------------------------------------------------------------------------------------------
// AoS (reference push_voxel order) -> SoA store.  One thread per (voxel, scalar).
__global__ void k_aos_to_soa(FactorView f, int base, int n, const double *__restrict__ clusters, const double *__restrict__ fix,
                             const double *__restrict__ coe, const double *__restrict__ eig_val, const double *__restrict__ eig_vec,
                             const double *__restrict__ pcr_add) {
  const int W = f.W;
  const int per = 10 * W + 10 + 1 + 3 + 9 + 10;
  const long long tot = (long long)n * per;
  for (long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x; t < tot; t += (long long)gridDim.x * blockDim.x) {
    const int a = (int)(t / per);
    int j = (int)(t % per);
    const int v = base + a;
    if (j < 10 * W) {
      const int i = j / 10, fld = j % 10;
      f.cl[((size_t)fld * W + i) * f.vs + v] = clusters[((size_t)a * W + i) * 10 + fld];
      continue;
    }
    j -= 10 * W;
    if (j < 10) { f.fix[(size_t)j * f.vs + v] = fix[(size_t)a * 10 + j]; continue; }
    j -= 10;
    if (j < 1) { f.coe[v] = coe[a]; continue; }
    j -= 1;
    if (j < 3) { f.eigval[(size_t)j * f.vs + v] = eig_val[(size_t)a * 3 + j]; continue; }
    j -= 3;
    if (j < 9) { f.eigvec[(size_t)j * f.vs + v] = eig_vec[(size_t)a * 9 + j]; continue; }
    j -= 9;
    f.pcr[(size_t)j * f.vs + v] = pcr_add[(size_t)a * 10 + j];
  }
}

__global__ void k_soa_to_aos_out(FactorView f, int n, double *__restrict__ eig_val, double *__restrict__ eig_vec, double *__restrict__ pcr_add) {
  const long long tot = (long long)n * 22;
  for (long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x; t < tot; t += (long long)gridDim.x * blockDim.x) {
    const int v = (int)(t / 22);
    int j = (int)(t % 22);
    if (j < 3) { eig_val[(size_t)v * 3 + j] = f.eigval[(size_t)j * f.vs + v]; continue; }
    j -= 3;
    if (j < 9) { eig_vec[(size_t)v * 9 + j] = f.eigvec[(size_t)j * f.vs + v]; continue; }
    j -= 9;
    pcr_add[(size_t)v * 10 + j] = f.pcr[(size_t)j * f.vs + v];
  }
}

__global__ void k_count_slots(FactorView f, int n, unsigned long long *out) {
  const long long tot = (long long)n * f.W;
  unsigned long long c = 0;
  for (long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x; t < tot; t += (long long)gridDim.x * blockDim.x) {
    const int i = (int)(t / n), v = (int)(t % n);
    if (f.cl[((size_t)9 * f.W + i) * f.vs + v] != 0.0) c++;
  }
  if (c) atomicAdd(out, c);
}

// ------------------------------------------------------------------------------------------------
// K4: residual pass.  One thread per voxel; 64-thread workgroups so that V ~ 3e4 voxels still spread
// over all 256 CUs.  Algorithmic traffic per voxel: read (W_occ+1)*80 + W*8 + 8 B, write 176 B.
__device__ __forceinline__ double wave_sum(double x) {
  for (int m = 32; m >= 1; m >>= 1) x += __shfl_xor(x, m, 64);
  return x;
}

__global__ __launch_bounds__(64) void k_residual(FactorView f, const double *__restrict__ poses, int head, int end,
                                                 double *__restrict__ partial, const int *__restrict__ gate) {
  __shared__ double sp[VBA_MAX_WIN_DEV * 12];
  if (gate && *gate == 0) return;   // device-side LM: the loop has stopped
  const int W = f.W;
  for (int t = threadIdx.x; t < W * 12; t += 64) sp[t] = poses[t];
  __syncthreads();
  const int v = head + blockIdx.x * 64 + threadIdx.x;
  const size_t vs = (size_t)f.vs;
  double r = 0.0;
  if (v < end) {
    double P00 = f.fix[0 * vs + v], P01 = f.fix[1 * vs + v], P02 = f.fix[2 * vs + v];
    double P11 = f.fix[3 * vs + v], P12 = f.fix[4 * vs + v], P22 = f.fix[5 * vs + v];
    double s0 = f.fix[6 * vs + v], s1 = f.fix[7 * vs + v], s2 = f.fix[8 * vs + v];
    double N = f.fix[9 * vs + v];
    for (int i = 0; i < W; i++) {
      const double *c = f.cl + (size_t)i * vs + v;
      const size_t fs = (size_t)W * vs;  // field stride
      const double n = c[9 * fs];
      if (n != 0.0) {
        const double pxx = c[0], pxy = c[fs], pxz = c[2 * fs], pyy = c[3 * fs], pyz = c[4 * fs], pzz = c[5 * fs];
        const double vx = c[6 * fs], vy = c[7 * fs], vz = c[8 * fs];
        const double *R = sp + 12 * i;
        const double tx = R[9], ty = R[10], tz = R[11];
        // Rv = R v ; v' = Rv + n t                                  (tools.hpp:360)
        const double rv0 = R[0] * vx + R[1] * vy + R[2] * vz;
        const double rv1 = R[3] * vx + R[4] * vy + R[5] * vz;
        const double rv2 = R[6] * vx + R[7] * vy + R[8] * vz;
        // M = R P (3x3), then (R P R^T) lower triangle               (tools.hpp:362)
        const double m00 = R[0] * pxx + R[1] * pxy + R[2] * pxz, m01 = R[0] * pxy + R[1] * pyy + R[2] * pyz, m02 = R[0] * pxz + R[1] * pyz + R[2] * pzz;
        const double m10 = R[3] * pxx + R[4] * pxy + R[5] * pxz, m11 = R[3] * pxy + R[4] * pyy + R[5] * pyz, m12 = R[3] * pxz + R[4] * pyz + R[5] * pzz;
        const double m20 = R[6] * pxx + R[7] * pxy + R[8] * pxz, m21 = R[6] * pxy + R[7] * pyy + R[8] * pyz, m22 = R[6] * pxz + R[7] * pyz + R[8] * pzz;
        P00 += (m00 * R[0] + m01 * R[1] + m02 * R[2]) + 2.0 * rv0 * tx + n * tx * tx;
        P01 += (m10 * R[0] + m11 * R[1] + m12 * R[2]) + (rv1 * tx + rv0 * ty) + n * ty * tx;
        P02 += (m20 * R[0] + m21 * R[1] + m22 * R[2]) + (rv2 * tx + rv0 * tz) + n * tz * tx;
        P11 += (m10 * R[3] + m11 * R[4] + m12 * R[5]) + 2.0 * rv1 * ty + n * ty * ty;
        P12 += (m20 * R[3] + m21 * R[4] + m22 * R[5]) + (rv2 * ty + rv1 * tz) + n * tz * ty;
        P22 += (m20 * R[6] + m21 * R[7] + m22 * R[8]) + 2.0 * rv2 * tz + n * tz * tz;
        s0 += rv0 + n * tx; s1 += rv1 + n * ty; s2 += rv2 + n * tz;
        N += n;
      }
    }
    // cov = P/N - vBar vBar^T ; eigen                                     (voxel_map.hpp:308-313)
    const double b0 = s0 / N, b1 = s1 / N, b2 = s2 / N;
    double w0, w1, w2, V[9];
    eig3_sym_dev(P00 / N - b0 * b0, P01 / N - b1 * b0, P02 / N - b2 * b0, P11 / N - b1 * b1, P12 / N - b2 * b1, P22 / N - b2 * b2,
                 w0, w1, w2, V);
    // write back eig_values / eig_vectors / pcr_adds                       (voxel_map.hpp:317-319)
    f.eigval[0 * vs + v] = w0; f.eigval[1 * vs + v] = w1; f.eigval[2 * vs + v] = w2;
#pragma unroll
    for (int k = 0; k < 9; k++) f.eigvec[(size_t)k * vs + v] = V[k];
    f.pcr[0 * vs + v] = P00; f.pcr[1 * vs + v] = P01; f.pcr[2 * vs + v] = P02; f.pcr[3 * vs + v] = P11; f.pcr[4 * vs + v] = P12;
    f.pcr[5 * vs + v] = P22; f.pcr[6 * vs + v] = s0; f.pcr[7 * vs + v] = s1; f.pcr[8 * vs + v] = s2; f.pcr[9 * vs + v] = N;
    r = f.coe[v] * w0;                                                    // voxel_map.hpp:323
  }
  r = wave_sum(r);
  if (threadIdx.x == 0) partial[blockIdx.x] = r;
}

// out[j] = sum_b partial[b*nout + j]   (deterministic, fixed order).  256 threads = 16 outputs x 16 partial groups,
// so ~nout/16 workgroups keep every CU busy on the (nb x nout) partial slab.
__global__ __launch_bounds__(256) void k_reduce_partials(const double *__restrict__ partial, int nb, int nout, double *__restrict__ out,
                                                        const int *__restrict__ gate) {
  __shared__ double s[256];
  if (gate && *gate == 0) return;
  const int j = threadIdx.x & 15, q = threadIdx.x >> 4;
  const int o = blockIdx.x * 16 + j;
  double acc = 0.0;
  if (o < nout)
    for (int b = q; b < nb; b += 16) acc += partial[(size_t)b * nout + o];
  s[threadIdx.x] = acc;
  __syncthreads();
  if (q == 0 && o < nout) {
    double t = 0.0;
#pragma unroll
    for (int k = 0; k < 16; k++) t += s[16 * k + j];
    out[o] = t;
  }
}

// scalar version: out[0] = sum partial[0..nb)  (one workgroup, fixed tree)
__global__ __launch_bounds__(256) void k_sum_scalar(const double *__restrict__ partial, int nb, double *__restrict__ out,
                                                   const int *__restrict__ gate) {
  __shared__ double s[4];
  if (gate && *gate == 0) return;
  double acc = 0.0;
  for (int b = threadIdx.x; b < nb; b += 256) acc += partial[b];
  acc = wave_sum(acc);
  if ((threadIdx.x & 63) == 0) s[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) out[0] = (s[0] + s[1]) + (s[2] + s[3]);
}

// ------------------------------------------------------------------------------------------------
// K3: Hessian / gradient pass.
// Per voxel  H_v = coe * ( G_v^T C_v G_v + blockdiag_i(E_{v,i}) )   with G_v (3 x 6W) rows
//   g_{.,1}, g_{.,2}  (g_{i,m} = Auk_i^T u_m, VM:228-232,240)  and  h_. = [vi x Ri^T uk ; n_i uk],
//   C_v = diag(2/(l0-l1), 2/(l0-l2), -2/NN^2)                                  (VM:198-201, 264-268)
// and E_{v,i} the diagonal-block terms not covered by the rank-3 form (VM:242-248).  This identity is checked
// against the oracle's literal per-pair accumulation in tests/ (and SURVEY.md §3.4).
// Workgroup = 32 voxels x W frames (one thread per (voxel, frame) slot) per tile:
//   phase A: slot threads build their 3 rows x 6 columns of G in LDS, accumulate E / gradient privately;
//   phase B: the workgroup contracts the 96-row tile G^T C G into register patches (4x4, upper triangle).
// Output: one partial [ (6W)^2 | 6W | 1 ] per workgroup, reduced by k_reduce_partials.
template <int W>
struct HessCfg {
  static constexpr int TV = 32;                            // voxels per tile
  static constexpr int NT = ((TV * W + 63) / 64) * 64;     // threads
  static constexpr int NC = 6 * W;                         // columns
  static constexpr int NP = (NC + 3) / 4;                  // 4-wide patches per dimension
  static constexpr int NCP = NP * 4;                       // padded columns
  static constexpr int NPATCH = NP * (NP + 1) / 2;         // upper-triangle patches
  static constexpr int KSPLIT = (NT / NPATCH) < 1 ? 1 : ((NT / NPATCH) > 4 ? 4 : (NT / NPATCH));
  static constexpr int NK = 3 * TV;                        // rows per tile (96)
  static constexpr int NOUT = NC * NC + NC + 1;
  static constexpr size_t LDS_DOUBLES = (size_t)NK * NCP + NK + W * 12 + 8;
  static constexpr size_t LDS_EPI = (size_t)NC * NC + NC + 8;
  static constexpr size_t LDS_BYTES = (LDS_DOUBLES > LDS_EPI ? LDS_DOUBLES : LDS_EPI) * sizeof(double);
};

template <int W>
__global__ __launch_bounds__(HessCfg<W>::NT) void k_hessian(FactorView f, const double *__restrict__ poses, int head, int end,
                                                            int ntiles, double *__restrict__ partial, const int *__restrict__ gate) {
  using C = HessCfg<W>;
  extern __shared__ __attribute__((aligned(16))) double lds[];
  if (gate && *gate == 0) return;   // device-side LM: rejected step -> the Hessian is not recomputed (VM:443)
  double *G = lds;                      // [NK][NCP]
  double *cK = G + (size_t)C::NK * C::NCP;  // [NK]
  double *sp = cK + C::NK;              // [W][12]
  const int tid = threadIdx.x;
  const size_t vs = (size_t)f.vs;
  const size_t fs = (size_t)W * vs;

  for (int t = tid; t < W * 12; t += C::NT) sp[t] = poses[t];
  for (int t = tid; t < C::NK * C::NCP; t += C::NT) G[t] = 0.0;   // padded columns stay zero for the whole kernel
  __syncthreads();

  const int vl = tid & 31, fi = tid >> 5;      // slot = (voxel-in-tile, frame)
  const bool slot_thread = fi < W;
  // phase-B role
  const int patch = tid % C::NPATCH, ks = tid / C::NPATCH;
  const bool syrk_thread = ks < C::KSPLIT;
  int pa = 0, pb = 0;
  {
    int p = patch, row = 0;
    while (p >= C::NP - row) { p -= C::NP - row; row++; }
    pa = row; pb = row + p;
  }
  double acc[4][4];
#pragma unroll
  for (int e = 0; e < 4; e++)
#pragma unroll
    for (int g = 0; g < 4; g++) acc[e][g] = 0.0;
  double Err[6] = {0, 0, 0, 0, 0, 0}, Ert[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0}, Ett[6] = {0, 0, 0, 0, 0, 0};
  double gj[6] = {0, 0, 0, 0, 0, 0};
  double rres = 0.0;

  for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    // ---------------- phase A
    if (slot_thread) {
      const int v = head + tile * C::TV + vl;
      double g1[6] = {0, 0, 0, 0, 0, 0}, g2[6] = {0, 0, 0, 0, 0, 0}, hh[6] = {0, 0, 0, 0, 0, 0};
      double ck1 = 0.0, ck2 = 0.0, ck3 = 0.0;
      if (v < end) {
        const double coe = f.coe[v];
        const double l0 = f.eigval[v], l1 = f.eigval[vs + v], l2 = f.eigval[2 * vs + v];
        const double NN = f.pcr[9 * vs + v];
        const double c1 = 2.0 / (l0 - l1), c2 = 2.0 / (l0 - l2);      // VM:201
        ck1 = coe * c1; ck2 = coe * c2; ck3 = coe * (-2.0 / NN / NN);
        if (fi == 0) rres += coe * l0;                                // VM:275
        const double *cp = f.cl + (size_t)fi * vs + v;
        const double n = cp[9 * fs];
        if (n != 0.0) {
          const double pxx = cp[0], pxy = cp[fs], pxz = cp[2 * fs], pyy = cp[3 * fs], pyz = cp[4 * fs], pzz = cp[5 * fs];
          const double vx = cp[6 * fs], vy = cp[7 * fs], vz = cp[8 * fs];
          const double u00 = f.eigvec[0 * vs + v], u01 = f.eigvec[1 * vs + v], u02 = f.eigvec[2 * vs + v];
          const double u10 = f.eigvec[3 * vs + v], u11 = f.eigvec[4 * vs + v], u12 = f.eigvec[5 * vs + v];
          const double u20 = f.eigvec[6 * vs + v], u21 = f.eigvec[7 * vs + v], u22 = f.eigvec[8 * vs + v];
          const double inn = 1.0 / NN;
          const double bx = f.pcr[6 * vs + v] * inn, by = f.pcr[7 * vs + v] * inn, bz = f.pcr[8 * vs + v] * inn;  // vBar (VM:190)
          const double *R = sp + 12 * fi;
          // eigenvectors uk=u0 (col 0), u1, u2
          const double k0 = u00, k1 = u10, k2 = u20;
          // a_m = R^T u_m
          const double a00 = R[0] * k0 + R[3] * k1 + R[6] * k2, a01 = R[1] * k0 + R[4] * k1 + R[7] * k2, a02 = R[2] * k0 + R[5] * k1 + R[8] * k2;
          const double a10 = R[0] * u01 + R[3] * u11 + R[6] * u21, a11 = R[1] * u01 + R[4] * u11 + R[7] * u21, a12 = R[2] * u01 + R[5] * u11 + R[8] * u21;
          const double a20 = R[0] * u02 + R[3] * u12 + R[6] * u22, a21 = R[1] * u02 + R[4] * u12 + R[7] * u22, a22 = R[2] * u02 + R[5] * u12 + R[8] * u22;
          // ti_v = p - vBar ; s_m = u_m . ti_v                                       (VM:224-225)
          const double tx = R[9] - bx, ty = R[10] - by, tz = R[11] - bz;
          const double s0 = k0 * tx + k1 * ty + k2 * tz, s1 = u01 * tx + u11 * ty + u21 * tz, s2 = u02 * tx + u12 * ty + u22 * tz;
          // P a_m
          const double pa00 = pxx * a00 + pxy * a01 + pxz * a02, pa01 = pxy * a00 + pyy * a01 + pyz * a02, pa02 = pxz * a00 + pyz * a01 + pzz * a02;
          const double pa10 = pxx * a10 + pxy * a11 + pxz * a12, pa11 = pxy * a10 + pyy * a11 + pyz * a12, pa12 = pxz * a10 + pyz * a11 + pzz * a12;
          const double pa20 = pxx * a20 + pxy * a21 + pxz * a22, pa21 = pxy * a20 + pyy * a21 + pyz * a22, pa22 = pxz * a20 + pyz * a21 + pzz * a22;
          // w = P a0 + s0 v   (combo1 = hat(w), VM:228) ; combo2 = R v + n ti_v (VM:229)
          const double wx = pa00 + s0 * vx, wy = pa01 + s0 * vy, wz = pa02 + s0 * vz;
          const double c2x = R[0] * vx + R[1] * vy + R[2] * vz + n * tx;
          const double c2y = R[3] * vx + R[4] * vy + R[5] * vz + n * ty;
          const double c2z = R[6] * vx + R[7] * vy + R[8] * vz + n * tz;
          // q = v x a0  (viRiTuk, VM:221)
          const double qx = vy * a02 - vz * a01, qy = vz * a00 - vx * a02, qz = vx * a01 - vy * a00;
          // g_rot,m = ( -a0 x (P a_m + s_m v) + w x a_m ) / NN ; g_tr,m = ( uk (c2.u_m) + (c2.uk) u_m ) / NN
          const double d0 = c2x * k0 + c2y * k1 + c2z * k2;
          const double d1 = c2x * u01 + c2y * u11 + c2z * u21;
          const double d2 = c2x * u02 + c2y * u12 + c2z * u22;
          // m = 0 (gradient, VM:235): g_rot,0 = 2 (w x a0)/NN, g_tr,0 = 2 d0 uk / NN
          const double j0 = 2.0 * (wy * a02 - wz * a01) * inn, j1 = 2.0 * (wz * a00 - wx * a02) * inn, j2 = 2.0 * (wx * a01 - wy * a00) * inn;
          const double j3 = 2.0 * d0 * k0 * inn, j4 = 2.0 * d0 * k1 * inn, j5 = 2.0 * d0 * k2 * inn;
          gj[0] += coe * j0; gj[1] += coe * j1; gj[2] += coe * j2; gj[3] += coe * j3; gj[4] += coe * j4; gj[5] += coe * j5;
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
          // E_rr = (2/NN) [ sym(a0 w^T) - (w.a0) I - hat(a0) P hat(a0) ]   (symmetric part of VM:242; the -0.5 hat(jjt)
          //        term cancels the antisymmetric part exactly)
          const double wa = wx * a00 + wy * a01 + wz * a02;
          // T = hat(a0) P  (rows: a0 x P[:,c] columnwise -> T[r][c] = (a0 x Pc)_r with Pc = column c of P)
          const double t00 = a01 * pxz - a02 * pxy, t10 = a02 * pxx - a00 * pxz, t20 = a00 * pxy - a01 * pxx;
          const double t01 = a01 * pyz - a02 * pyy, t11 = a02 * pxy - a00 * pyz, t21 = a00 * pyy - a01 * pxy;
          const double t02 = a01 * pzz - a02 * pyz, t12 = a02 * pxz - a00 * pzz, t22 = a00 * pyz - a01 * pxz;
          // S = T hat(a0):  S[r][c] = sum_k T[r][k] hat(a0)[k][c];  hat(a0) = [0 -a02 a01; a02 0 -a00; -a01 a00 0]
          const double S00 = t01 * a02 - t02 * a01, S01 = -t00 * a02 + t02 * a00, S02 = t00 * a01 - t01 * a00;
          const double S11 = -t10 * a02 + t12 * a00, S12 = t10 * a01 - t11 * a00;
          const double S22 = t20 * a01 - t21 * a00;
          const double e2 = 2.0 * inn * coe;
          Err[0] += e2 * (a00 * wx - wa - S00);
          Err[1] += e2 * (0.5 * (a00 * wy + wx * a01) - S01);
          Err[2] += e2 * (0.5 * (a00 * wz + wx * a02) - S02);
          Err[3] += e2 * (a01 * wy - wa - S11);
          Err[4] += e2 * (0.5 * (a01 * wz + wy * a02) - S12);
          Err[5] += e2 * (a02 * wz - wa - S22);
          // E_rt = (2/NN) q uk^T (VM:239,244 without the -ni/NN part that lives in the rank-3 form) ; E_tt = (2 n/NN) uk uk^T
          Ert[0] += e2 * qx * k0; Ert[1] += e2 * qx * k1; Ert[2] += e2 * qx * k2;
          Ert[3] += e2 * qy * k0; Ert[4] += e2 * qy * k1; Ert[5] += e2 * qy * k2;
          Ert[6] += e2 * qz * k0; Ert[7] += e2 * qz * k1; Ert[8] += e2 * qz * k2;
          const double e3 = e2 * n;
          Ett[0] += e3 * k0 * k0; Ett[1] += e3 * k0 * k1; Ett[2] += e3 * k0 * k2; Ett[3] += e3 * k1 * k1; Ett[4] += e3 * k1 * k2; Ett[5] += e3 * k2 * k2;
        }
      }
      double *g = G + (size_t)(3 * vl) * C::NCP + 6 * fi;
#pragma unroll
      for (int d = 0; d < 6; d++) { g[d] = g1[d]; g[C::NCP + d] = g2[d]; g[2 * C::NCP + d] = hh[d]; }
      if (fi == 0) { cK[3 * vl] = ck1; cK[3 * vl + 1] = ck2; cK[3 * vl + 2] = ck3; }
    }
    __syncthreads();
    // ---------------- phase B: acc += sum_k cK[k] G[k][4pa+e] G[k][4pb+g]
    if (syrk_thread) {
      const int k0 = ks * (C::NK / C::KSPLIT), k1 = (ks == C::KSPLIT - 1) ? C::NK : k0 + C::NK / C::KSPLIT;
      const double *ga = G + 4 * pa, *gb = G + 4 * pb;
      for (int k = k0; k < k1; k++) {
        const double ck = cK[k];
        const double2 x0 = *reinterpret_cast<const double2 *>(ga + (size_t)k * C::NCP);
        const double2 x1 = *reinterpret_cast<const double2 *>(ga + (size_t)k * C::NCP + 2);
        const double2 y0 = *reinterpret_cast<const double2 *>(gb + (size_t)k * C::NCP);
        const double2 y1 = *reinterpret_cast<const double2 *>(gb + (size_t)k * C::NCP + 2);
        const double a[4] = {x0.x * ck, x0.y * ck, x1.x * ck, x1.y * ck};
        const double b[4] = {y0.x, y0.y, y1.x, y1.y};
#pragma unroll
        for (int e = 0; e < 4; e++)
#pragma unroll
          for (int g = 0; g < 4; g++) acc[e][g] += a[e] * b[g];
      }
    }
    __syncthreads();
  }

  // ---------------- epilogue: assemble the workgroup's partial in LDS, then store
  double *Hs = lds;                          // [NC][NC]
  double *gs = Hs + (size_t)C::NC * C::NC;   // [NC]
  double *rs = gs + C::NC;                   // [1]
  for (int t = tid; t < C::NC * C::NC + C::NC + 1; t += C::NT) Hs[t] = 0.0;
  __syncthreads();
  for (int s = 0; s < C::KSPLIT; s++) {
    if (syrk_thread && ks == s) {
#pragma unroll
      for (int e = 0; e < 4; e++)
#pragma unroll
        for (int g = 0; g < 4; g++) {
          const int r = 4 * pa + e, c = 4 * pb + g;
          if (r < C::NC && c < C::NC && r <= c) Hs[r * C::NC + c] += acc[e][g];
        }
    }
    __syncthreads();
  }
  // wave-level reduction of the private E / gradient / residual over the 32 voxel lanes of each frame
#pragma unroll
  for (int m = 16; m >= 1; m >>= 1) {
#pragma unroll
    for (int k = 0; k < 6; k++) { Err[k] += __shfl_xor(Err[k], m, 64); Ett[k] += __shfl_xor(Ett[k], m, 64); gj[k] += __shfl_xor(gj[k], m, 64); }
#pragma unroll
    for (int k = 0; k < 9; k++) Ert[k] += __shfl_xor(Ert[k], m, 64);
    rres += __shfl_xor(rres, m, 64);
  }
  if (slot_thread && vl == 0) {
    const int o = 6 * fi;
    // rot-rot (upper incl. diagonal)
    Hs[(o + 0) * C::NC + o + 0] += Err[0]; Hs[(o + 0) * C::NC + o + 1] += Err[1]; Hs[(o + 0) * C::NC + o + 2] += Err[2];
    Hs[(o + 1) * C::NC + o + 1] += Err[3]; Hs[(o + 1) * C::NC + o + 2] += Err[4]; Hs[(o + 2) * C::NC + o + 2] += Err[5];
    // rot-trans
#pragma unroll
    for (int r = 0; r < 3; r++)
#pragma unroll
      for (int c = 0; c < 3; c++) Hs[(o + r) * C::NC + o + 3 + c] += Ert[3 * r + c];
    // trans-trans
    Hs[(o + 3) * C::NC + o + 3] += Ett[0]; Hs[(o + 3) * C::NC + o + 4] += Ett[1]; Hs[(o + 3) * C::NC + o + 5] += Ett[2];
    Hs[(o + 4) * C::NC + o + 4] += Ett[3]; Hs[(o + 4) * C::NC + o + 5] += Ett[4]; Hs[(o + 5) * C::NC + o + 5] += Ett[5];
#pragma unroll
    for (int k = 0; k < 6; k++) gs[o + k] = gj[k];
    if (fi == 0) rs[0] = rres;
  }
  __syncthreads();
  // mirror upper -> lower (VM:279-281) and store
  double *out = partial + (size_t)blockIdx.x * C::NOUT;
  for (int t = tid; t < C::NC * C::NC; t += C::NT) {
    const int r = t / C::NC, c = t % C::NC;
    out[t] = (r <= c) ? Hs[t] : Hs[c * C::NC + r];
  }
  for (int t = tid; t < C::NC + 1; t += C::NT) out[C::NC * C::NC + t] = gs[t];
}

}  // namespace vba
