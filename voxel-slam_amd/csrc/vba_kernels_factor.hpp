// HIP kernels (gfx950 / CDNA4, wave64) for the factor level of the Voxel-SLAM local BA:
//   k_residual   (K4)  <->  LidarFactor::evaluate_only_residual   voxel_map.hpp:285-325 + tools.hpp:357-363
//   k_hessian<W> (K3)  <->  LidarFactor::acc_evaluate2            voxel_map.hpp:150-282
//   k_reduce_partials  <->  the thread-sum after join()           voxel_map.hpp:376-386, 571-581
// Data layout in HBM: SoA [field][frame][voxel] (voxel fastest) so that one wave reads 64 consecutive
// doubles (512 B) per field — see DESIGN.md §3.
#pragma once
#include <hip/hip_runtime.h>
#include "vba_eig3.hpp"

namespace vba {

struct FactorView {
  double *cl;      // [10][W][vs]   body-frame clusters per (frame, voxel): Pxx,Pxy,Pxz,Pyy,Pyz,Pzz,vx,vy,vz,N
  double *fix;     // [10][vs]      sig_vecs (fixed world cluster)
  double *coe;     // [vs]
  double *eigval;  // [3][vs]
  double *eigvec;  // [9][vs]       row-major r*3+c, column c = eigenvector c
  double *pcr;     // [10][vs]      pcr_adds
  unsigned int *occ;   // [vs]      bit i set <=> slot (voxel, frame i) holds points (cl N != 0): what the residual pass tests instead of
                       //           reading the N of all W slots (4 B per voxel instead of 8 W); kept current by k_factor_mask
  int *tiles;      // Hessian-pass tile table (k_factor_tiles): [0] = number of tiles, then (first voxel, voxels, union mask, 0) from [4]
  int vs;          // voxel stride (capacity)
  int W;
};

// Rank of an occupancy mask of `nb` frames in the store order: popcount DESCENDING, masks of one popcount in ascending numeric order
// (colexicographic rank).  Equal masks share a bucket, so the counting sort of the extraction keeps them adjacent.
__host__ __device__ inline int mask_bucket(unsigned int m, int nb) {
  constexpr int C[11][11] = {{1, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, {1, 1, 0, 0, 0, 0, 0, 0, 0, 0, 0}, {1, 2, 1, 0, 0, 0, 0, 0, 0, 0, 0}, {1, 3, 3, 1, 0, 0, 0, 0, 0, 0, 0},
                             {1, 4, 6, 4, 1, 0, 0, 0, 0, 0, 0}, {1, 5, 10, 10, 5, 1, 0, 0, 0, 0, 0}, {1, 6, 15, 20, 15, 6, 1, 0, 0, 0, 0},
                             {1, 7, 21, 35, 35, 21, 7, 1, 0, 0, 0}, {1, 8, 28, 56, 70, 56, 28, 8, 1, 0, 0}, {1, 9, 36, 84, 126, 126, 84, 36, 9, 1, 0},
                             {1, 10, 45, 120, 210, 252, 210, 120, 45, 10, 1}};
  int p = 0;
  for (int b = 0; b < nb; b++) p += (m >> b) & 1u;
  int off = 0;
  for (int q = nb; q > p; q--) off += C[nb][q];
  int r = 0, k = 0;
  for (int b = 0; b < nb; b++)
    if ((m >> b) & 1u) { k++; r += C[b][k]; }
  return off + r;
}

// ------------------------------------------------------------------------------------------------
// Symmetric 3x3 eigen-decomposition, ascending eigenvalues, orthonormal eigenvectors in columns.
// Cyclic Jacobi in registers (no indexed arrays -> no scratch).  Replaces Eigen::SelfAdjointEigenSolver
// at voxel_map.hpp:312 / :1416 / :1525 (result equal up to rounding and eigenvector sign).
// One Jacobi rotation in the (p,q) plane.  The rotation only has to be ORTHOGONAL to full precision, not optimal: the
// tangent t is computed in f32 (v_rcp_f32 / v_sqrt_f32, ~1e-7 relative), c = rsqrt(1 + t^2) in f64 (v_rsq_f64 + two
// Newton steps), s = t c, so c^2 + s^2 = 1 to rounding while the annihilated element is left at ~1e-7 |a_pq| and dies
// in the next sweep.  Measured on MI355X (K4, one wave per SIMD): the textbook form (f64 div, sqrt, div, sqrt, div per
// rotation) cost 10.3k cycles per eigen-solve, 45 % of the residual pass.
__device__ __forceinline__ void jacobi_rot(double &app, double &aqq, double &apq, double &arp, double &arq,
                                           double &v0p, double &v0q, double &v1p, double &v1q, double &v2p, double &v2q,
                                           int sweep) {
  if (apq == 0.0) return;
  const double g = 100.0 * fabs(apq);
  // an off-diagonal below ulp/200 of both diagonals cannot change them any more: drop it (at any sweep)
  if (fabs(app) + g == fabs(app) && fabs(aqq) + g == fabs(aqq)) { apq = 0.0; return; }
  // t = sgn(a) b / (|a| + sqrt(a^2 + b^2)),  a = (aqq - app) / 2, b = apq   (the smaller root of t^2 + 2 theta t - 1 = 0);
  // operands are scaled by the larger magnitude first so that the f32 range cannot over/underflow
  const double a = 0.5 * (aqq - app);
  const double inv_scale = __builtin_amdgcn_rcp(fmax(fabs(a), fabs(apq)));   // raw v_rcp_f64: only the ratio a : b matters
  const float af = (float)(a * inv_scale), bf = (float)(apq * inv_scale);
  const float tf = bf * __builtin_amdgcn_rcpf(fabsf(af) + __builtin_amdgcn_sqrtf(af * af + bf * bf));   // raw v_sqrt_f32 / v_rcp_f32
  const double t = (af < 0.0f) ? -(double)tf : (double)tf;
  const double x = 1.0 + t * t;
  double c = __builtin_amdgcn_rsq(x);            // ~26 good bits
  c = c * (1.5 - 0.5 * x * c * c);
  c = c * (1.5 - 0.5 * x * c * c);
  const double s = t * c;
  // A <- J^T A J:  a_pp' = c^2 a_pp - 2 c s a_pq + s^2 a_qq, a_qq' likewise, a_pq' = c s (a_pp - a_qq) + (c^2 - s^2) a_pq
  const double cc = c * c, ss = s * s, cs = c * s;
  const double npp = cc * app - 2.0 * cs * apq + ss * aqq;
  const double nqq = ss * app + 2.0 * cs * apq + cc * aqq;
  const double npq = cs * (app - aqq) + (cc - ss) * apq;
  app = npp; aqq = nqq; apq = npq;
  double x1 = arp, y1 = arq;
  arp = c * x1 - s * y1; arq = s * x1 + c * y1;
  x1 = v0p; y1 = v0q; v0p = c * x1 - s * y1; v0q = s * x1 + c * y1;
  x1 = v1p; y1 = v1q; v1p = c * x1 - s * y1; v1q = s * x1 + c * y1;
  x1 = v2p; y1 = v2q; v2p = c * x1 - s * y1; v2q = s * x1 + c * y1;
}

#define VBA_SWAP(a, b) { double _t = a; a = b; b = _t; }

// in: lower triangle a00,a10,a20,a11,a21,a22.  out: w0<=w1<=w2, V (row-major, columns = eigenvectors).  Plain cyclic sweeps in f64:
// this is the rare fallback of eig3_sym_dev below (near-double eigenvalue pairs, multiples of the identity, degenerate input),
// so it is written for few registers, not for speed (the sweep loop is not unrolled).
__device__ __forceinline__ Eig3 eig3_jacobi_dev(double a00, double a01, double a02, double a11, double a12, double a22) {
  double v00 = 1, v01 = 0, v02 = 0, v10 = 0, v11 = 1, v12 = 0, v20 = 0, v21 = 0, v22 = 1;
#pragma unroll 1
  for (int sweep = 0; sweep < 30; sweep++) {
    if (fabs(a01) + fabs(a02) + fabs(a12) == 0.0) break;
    jacobi_rot(a00, a11, a01, a02, a12, v00, v01, v10, v11, v20, v21, sweep);  // (p,q)=(0,1), r=2
    jacobi_rot(a00, a22, a02, a01, a12, v00, v02, v10, v12, v20, v22, sweep);  // (0,2), r=1
    jacobi_rot(a11, a22, a12, a01, a02, v01, v02, v11, v12, v21, v22, sweep);  // (1,2), r=0
  }
  if (a11 < a00) { VBA_SWAP(a00, a11); VBA_SWAP(v00, v01); VBA_SWAP(v10, v11); VBA_SWAP(v20, v21); }
  if (a22 < a00) { VBA_SWAP(a00, a22); VBA_SWAP(v00, v02); VBA_SWAP(v10, v12); VBA_SWAP(v20, v22); }
  if (a22 < a11) { VBA_SWAP(a11, a22); VBA_SWAP(v01, v02); VBA_SWAP(v11, v12); VBA_SWAP(v21, v22); }
  Eig3 o;
  o.w0 = a00; o.w1 = a11; o.w2 = a22;
  o.v00 = v00; o.v01 = v01; o.v02 = v02; o.v10 = v10; o.v11 = v11; o.v12 = v12; o.v20 = v20; o.v21 = v21; o.v22 = v22;
  return o;
}

// The plane fit's eigen-solver: direct (vba_eig3.hpp: trigonometric seed + Newton for the isolated root, deflation, eigenvectors
// from cross products and a 2x2 complement problem); matrices with a near-double eigenvalue pair, multiples of the identity and
// non-finite input take the Jacobi sweeps above.
__device__ __forceinline__ void eig3_sym_dev(double a00, double a01, double a02, double a11, double a12, double a22,
                                             double &w0, double &w1, double &w2, double *V) {
  Eig3 o;
  if (!eig3_direct(a00, a01, a02, a11, a12, a22, o)) o = eig3_jacobi_dev(a00, a01, a02, a11, a12, a22);
  w0 = o.w0; w1 = o.w1; w2 = o.w2;
  V[0] = o.v00; V[1] = o.v01; V[2] = o.v02; V[3] = o.v10; V[4] = o.v11; V[5] = o.v12; V[6] = o.v20; V[7] = o.v21; V[8] = o.v22;
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

// occupancy masks of voxels [base, base + n): run after every write of the cluster rows (push_voxels, the map's / the GBA octree's extraction)
__global__ void k_factor_mask(FactorView f, int base, int n) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= n) return;
  const int v = base + t;
  const size_t vs = (size_t)f.vs;
  unsigned int m = 0;
  for (int i = 0; i < f.W; i++) m |= (f.cl[((size_t)9 * f.W + i) * vs + v] != 0.0) ? (1u << i) : 0u;
  f.occ[v] = m;
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

// Calibration of rocprofv3's FETCH_SIZE for this library's access shape (MI355X_MICROARCH.md, "HBM": the counter is exact x1/2 for
// 16-byte-per-lane streams and uncalibrated for others): every thread reads 16 doubles, 8 bytes per lane, 512 contiguous bytes
// per wave instruction — what the SoA factor store is read with — so the launch reads exactly gridDim * blockDim * 128 bytes.
__global__ __launch_bounds__(256) void k_calib_read8(const double *__restrict__ buf, double *__restrict__ sink) {
  const size_t nthr = (size_t)gridDim.x * blockDim.x, t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  double acc = 0.0;
#pragma unroll
  for (int k = 0; k < 16; k++) acc += buf[t + (size_t)k * nthr];
  if (acc == 123.456) sink[0] = acc;     // (never true for the zero-filled buffer: keeps the loads alive)
}

// ------------------------------------------------------------------------------------------------
// K4: residual pass  <->  LidarFactor::evaluate_only_residual voxel_map.hpp:285-325 (+ PointCluster::transform tools.hpp:357-363).
// Algorithmic traffic per voxel: read (W_occ + 1) * 80 + 8 B, write 176 B.
//
// Workgroup = TV voxels x W frames, one thread per (voxel, frame) SLOT; wave 0's first TV lanes double as the voxel threads.
//   trip 1: every slot thread reads its voxel's occupancy mask (4 B, shared by the W slot threads of the voxel), thread (voxel, k)
//           scalar k of the fixed cluster, row 0 coe, W*12 threads the poses;
//   trip 2: occupied slots read their 10 scalars (empty slots cost nothing);
//   slot threads transform their cluster (~95 f64 operations) and park the 10 world-frame scalars in LDS;
//   thread (voxel, k) adds scalar k of the W frames to the fixed cluster IN FRAME ORDER (the reference's order, VM:297-305: the
//   sum does not depend on the launch) and writes pcr_adds (VM:319); the voxel threads solve the 3x3 eigen-problem
//   (vba_eig3.hpp), write eig_values / eig_vectors back (VM:317-318) and reduce coe * lambda_0.
// Why this shape (MI355X, bench window V = 18 391, S/V = 4.3): the first version gave a lane a whole voxel — the wave then issues the
// transform of all W frames for every lane (exec-masked lanes still cost issue cycles): 950 f64 instructions x 4 cycles = 3.8k of
// the pass's 22k cycles, and the Jacobi eigen-solve another 8.5k (in-kernel stamps, profiles/r01_k3_stamps.txt); one lane per slot
// issues the transform once per wave, and ~3 waves per SIMD (instead of 0.3) overlap each other's memory trips.
// PointCluster::transform (tools.hpp:357-363) in the reference's operation order, every operation rounded separately (the
// reference targets baseline x86-64: no FMA contraction):  v' = R v + p N ;  rp = (R v) p^T ;  P' = ((R P R^T + rp) + rp^T) + (p p^T) N,
// matrix products as left-to-right dot products.  The six P scalars of a cluster are its LOWER triangle (what the eigen-solver of the
// reference reads, and what the pushes make symmetric anyway).  With the frames added in frame order (VM:297-305) pcr_adds — which
// margi copies into the map (VM:1498-1500) — comes out bit-identical to the CPU restatement's, so the map's sums stay exact over a session.
struct Cl10 { double p00, p10, p20, p11, p21, p22, v0, v1, v2, n; };
__device__ __forceinline__ Cl10 cluster_transform_exact(double c0, double c1, double c2, double c3, double c4, double c5, double v0, double v1, double v2, double n,
                                                        const double *R) {
#pragma clang fp contract(off)
  const double R0 = R[0], R1 = R[1], R2 = R[2], R3 = R[3], R4 = R[4], R5 = R[5], R6 = R[6], R7 = R[7], R8 = R[8];
  const double tx = R[9], ty = R[10], tz = R[11];
  const double rv0 = (R0 * v0 + R1 * v1) + R2 * v2, rv1 = (R3 * v0 + R4 * v1) + R5 * v2, rv2 = (R6 * v0 + R7 * v1) + R8 * v2;
  // M = R P (P symmetric: P01 = c1, P02 = c2, P12 = c4)
  const double m00 = (R0 * c0 + R1 * c1) + R2 * c2, m01 = (R0 * c1 + R1 * c3) + R2 * c4, m02 = (R0 * c2 + R1 * c4) + R2 * c5;
  const double m10 = (R3 * c0 + R4 * c1) + R5 * c2, m11 = (R3 * c1 + R4 * c3) + R5 * c4, m12 = (R3 * c2 + R4 * c4) + R5 * c5;
  const double m20 = (R6 * c0 + R7 * c1) + R8 * c2, m21 = (R6 * c1 + R7 * c3) + R8 * c4, m22 = (R6 * c2 + R7 * c4) + R8 * c5;
  Cl10 o;
  o.p00 = ((((m00 * R0 + m01 * R1) + m02 * R2) + rv0 * tx) + rv0 * tx) + (tx * tx) * n;
  o.p10 = ((((m10 * R0 + m11 * R1) + m12 * R2) + rv1 * tx) + rv0 * ty) + (ty * tx) * n;
  o.p20 = ((((m20 * R0 + m21 * R1) + m22 * R2) + rv2 * tx) + rv0 * tz) + (tz * tx) * n;
  o.p11 = ((((m10 * R3 + m11 * R4) + m12 * R5) + rv1 * ty) + rv1 * ty) + (ty * ty) * n;
  o.p21 = ((((m20 * R3 + m21 * R4) + m22 * R5) + rv2 * ty) + rv1 * tz) + (tz * ty) * n;
  o.p22 = ((((m20 * R6 + m21 * R7) + m22 * R8) + rv2 * tz) + rv2 * tz) + (tz * tz) * n;
  o.v0 = rv0 + tx * n; o.v1 = rv1 + ty * n; o.v2 = rv2 + tz * n;
  o.n = n;
  return o;
}

__device__ __forceinline__ double wave_sum(double x) {
  for (int m = 32; m >= 1; m >>= 1) x += __shfl_xor(x, m, 64);
  return x;
}

// Sum over the 64 lanes of a wave, result valid in LANE 63 only.  Data-parallel-primitive moves instead of ds_bpermute
// (__shfl_xor goes through the LDS crossbar: ~100+ cycles per step, six dependent steps): quad swaps, row mirrors, then the
// two row broadcasts of gfx9.  Fixed summation tree, so the result does not depend on the launch.
__device__ __forceinline__ double dpp_mov_f64(double x, double old, const int ctrl, const int row_mask) {
  const long long xi = __double_as_longlong(x), oi = __double_as_longlong(old);
  int lo = (int)xi, hi = (int)(xi >> 32);
  const int olo = (int)oi, ohi = (int)(oi >> 32);
  switch (ctrl) {   // (the control word is an immediate operand)
    case 0xB1: lo = __builtin_amdgcn_update_dpp(olo, lo, 0xB1, 0xF, 0xF, false); hi = __builtin_amdgcn_update_dpp(ohi, hi, 0xB1, 0xF, 0xF, false); break;
    case 0x4E: lo = __builtin_amdgcn_update_dpp(olo, lo, 0x4E, 0xF, 0xF, false); hi = __builtin_amdgcn_update_dpp(ohi, hi, 0x4E, 0xF, 0xF, false); break;
    case 0x141: lo = __builtin_amdgcn_update_dpp(olo, lo, 0x141, 0xF, 0xF, false); hi = __builtin_amdgcn_update_dpp(ohi, hi, 0x141, 0xF, 0xF, false); break;
    case 0x140: lo = __builtin_amdgcn_update_dpp(olo, lo, 0x140, 0xF, 0xF, false); hi = __builtin_amdgcn_update_dpp(ohi, hi, 0x140, 0xF, 0xF, false); break;
    case 0x142: lo = __builtin_amdgcn_update_dpp(olo, lo, 0x142, 0xA, 0xF, false); hi = __builtin_amdgcn_update_dpp(ohi, hi, 0x142, 0xA, 0xF, false); break;
    default: lo = __builtin_amdgcn_update_dpp(olo, lo, 0x143, 0xC, 0xF, false); hi = __builtin_amdgcn_update_dpp(ohi, hi, 0x143, 0xC, 0xF, false); break;
  }
  (void)row_mask;
  return __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
}
// sum over each aligned group of 8 lanes, result in all 8 lanes (fixed tree)
__device__ __forceinline__ double group8_sum(double x) {
  x += dpp_mov_f64(x, x, 0xB1, 0xF);      // quad_perm [1,0,3,2]
  x += dpp_mov_f64(x, x, 0x4E, 0xF);      // quad_perm [2,3,0,1]
  x += dpp_mov_f64(x, x, 0x141, 0xF);     // row_half_mirror
  return x;
}
__device__ __forceinline__ double wave_sum_to_lane63(double x) {
  x += dpp_mov_f64(x, x, 0xB1, 0xF);      // quad_perm [1,0,3,2]
  x += dpp_mov_f64(x, x, 0x4E, 0xF);      // quad_perm [2,3,0,1]
  x += dpp_mov_f64(x, x, 0x141, 0xF);     // row_half_mirror
  x += dpp_mov_f64(x, x, 0x140, 0xF);     // row_mirror: every lane of a 16-lane row holds the row's sum
  x += dpp_mov_f64(x, 0.0, 0x142, 0xA);   // row_bcast15 into rows 1 and 3 (the others add 0)
  x += dpp_mov_f64(x, 0.0, 0x143, 0xC);   // row_bcast31 into rows 2 and 3
  return x;
}

template <int W, int TV>
struct ResCfg {
  static constexpr int NT = ((TV * W + 63) / 64) * 64;      // threads per workgroup
  static constexpr int NF = NT / TV;                        // thread rows (>= W): row fi owns frame fi and the scalars k = fi, fi + NF, ... < 10
  static constexpr int KPT = (10 + NF - 1) / NF;            // scalars per thread in the frame sum
  static_assert(TV <= 64 && (TV & (TV - 1)) == 0 && NT % TV == 0, "the voxel threads are the first TV lanes of wave 0");
  static_assert(W * 12 <= NT, "one thread per pose scalar");
};

template <int W, int TV, bool STAMPS>
__global__ __launch_bounds__((ResCfg<W, TV>::NT)) void k_residual_s(FactorView f, const double *__restrict__ poses, int head, int end,
                                                                  double *__restrict__ partial, const int *__restrict__ gate,
                                                                  long long *__restrict__ stamps) {
  using C = ResCfg<W, TV>;
  __shared__ double T[10][W][TV];                               // world-frame cluster of every slot (zeros for empty slots)
  __shared__ double S[10][TV];                                  // pcr_add of every voxel
  __shared__ double sp[W * 12];
  // The gate (a flag the previous kernel wrote, ~2 us away on another XCD) and the first round of loads are requested
  // TOGETHER; the early exit is taken only after both are back, so a live pass pays one memory trip here, not two.
  const int gate_v = gate ? *gate : 1;
  const int tid = threadIdx.x, vl = tid % TV, fi = tid / TV;
  const int v = head + blockIdx.x * TV + vl;
  const size_t vs = (size_t)f.vs, fs = (size_t)W * vs;
  const int vc = v < end ? v : end - 1;                         // clamped: trip 1 is issued unconditionally (end > head)
  const int fic = fi < W ? fi : W - 1;
  long long st0 = 0, st1 = 0, st2 = 0, st3 = 0;
  if (STAMPS) st0 = clock64();
  // ---- trip 1: the voxel's occupancy mask, this thread's scalars of the fixed cluster, coe (row 0), the poses
  const unsigned int occm = f.occ[vc];
  const double pose_s = poses[tid < W * 12 ? tid : 0];
  double acc[C::KPT];
#pragma unroll
  for (int j = 0; j < C::KPT; j++) { const int k = fi + j * C::NF; acc[j] = f.fix[(size_t)(k < 10 ? k : 9) * vs + vc]; }
  double coe = f.coe[vc];
  unsigned int occv = occm;
  asm volatile("" : "+v"(occv), "+v"(coe), "+v"(acc[0]));       // keep the loads above the exit (they would be sunk below it)
  if (gate_v == 0) return;
  if (tid < W * 12) sp[tid] = pose_s;
  // ---- trip 2: the 10 scalars of an occupied slot (an empty slot costs nothing beyond its mask bit)
  const bool occ = fi < W && v < end && ((occv >> fic) & 1u);
  double c[9], n = 0.0;
#pragma unroll
  for (int k = 0; k < 9; k++) c[k] = occ ? f.cl[(size_t)k * fs + (size_t)fi * vs + v] : 0.0;
  if (occ) n = f.cl[9 * fs + (size_t)fi * vs + v];
  __syncthreads();
  if (occ) {
    const Cl10 w = cluster_transform_exact(c[0], c[1], c[2], c[3], c[4], c[5], c[6], c[7], c[8], n, sp + 12 * fi);   // tools.hpp:357-363
    T[0][fi][vl] = w.p00; T[1][fi][vl] = w.p10; T[2][fi][vl] = w.p20; T[3][fi][vl] = w.p11; T[4][fi][vl] = w.p21; T[5][fi][vl] = w.p22;
    T[6][fi][vl] = w.v0; T[7][fi][vl] = w.v1; T[8][fi][vl] = w.v2;
    T[9][fi][vl] = n;
  } else if (fi < W) {
#pragma unroll
    for (int k = 0; k < 10; k++) T[k][fi][vl] = 0.0;            // x + 0.0 = x: the frame sum below needs no occupancy test
  }
  __syncthreads();
  // ---- frame sum: thread (vl, k) adds scalar k of the W frames to the fixed cluster IN FRAME ORDER (sig = sig_vecs[a]; sig += ...
  //      VM:297-305), keeps it for the voxel thread and writes pcr_adds[a] (VM:319), 10 coalesced stores per workgroup row
#pragma unroll
  for (int j = 0; j < C::KPT; j++) {
    const int k = fi + j * C::NF;
    if (k < 10) {
      double a = acc[j];
#pragma unroll
      for (int i = 0; i < W; i++) a += T[k][i][vl];
      S[k][vl] = a;
      if (v < end) f.pcr[(size_t)k * vs + v] = a;
    }
  }
  __syncthreads();
  if (STAMPS) st1 = clock64();
  double r = 0.0;
  if (fi == 0 && v < end) {
    // cov = P/N - vBar vBar^T ; eigen                                     (voxel_map.hpp:308-313)
    const double Nv = S[9][vl];
    double iN = __builtin_amdgcn_rcp(Nv);                              // 1 / N to full precision: v_rcp_f64 + two Newton steps
    iN = iN * (2.0 - Nv * iN);
    iN = iN * (2.0 - Nv * iN);
    const double b0 = S[6][vl] * iN, b1 = S[7][vl] * iN, b2 = S[8][vl] * iN;
    double w0, w1, w2, V[9];
    eig3_sym_dev(S[0][vl] * iN - b0 * b0, S[1][vl] * iN - b1 * b0, S[2][vl] * iN - b2 * b0, S[3][vl] * iN - b1 * b1, S[4][vl] * iN - b2 * b1,
                 S[5][vl] * iN - b2 * b2, w0, w1, w2, V);
    if (STAMPS) st2 = clock64();
    // write back eig_values / eig_vectors                                  (voxel_map.hpp:317-318)
    f.eigval[0 * vs + v] = w0; f.eigval[1 * vs + v] = w1; f.eigval[2 * vs + v] = w2;
#pragma unroll
    for (int k = 0; k < 9; k++) f.eigvec[(size_t)k * vs + v] = V[k];
    r = coe * w0;                                                         // voxel_map.hpp:323
  }
  if (tid < 64) {
    r = wave_sum_to_lane63(r);
    if (tid == 63) partial[blockIdx.x] = r;
    if (STAMPS && tid == 0 && blockIdx.x < 2048) {
      st3 = clock64();
      long long *o = stamps + (size_t)blockIdx.x * 4;
      o[0] = st0; o[1] = st1; o[2] = st2; o[3] = st3;
    }
  }
}

// Voxel-per-lane form of the pass — used for LARGE stores (throughput-bound: V > kResidualVoxelPerLane in voxelba.hip) — for stores in OCCUPANCY-MASK ORDER (what the map's extraction produces, vba_kernels_map.hpp
// k_extract_key): the 64 voxels of a wave then see (nearly) the same frames, so a lane can own a whole voxel without the wave issuing
// the transform of frames its lanes do not have — the reason the slot-parallel form above exists.  No LDS, no barriers, no second
// kernel: trip 1 = mask, fixed cluster, coe; trip 2 = every scalar of every frame the WAVE sees, all in flight at once (wave-uniform
// skip of the others); the frames are added in frame order (VM:297-305, the same sums as the slot-parallel form: bit-identical);
// eigen-solve and stores with all 64 lanes active.  One wave per workgroup; ~43 512-byte loads in flight per wave.
// Measured on MI355X: V = 9.9e5 (570 MB algorithmic, beyond the Infinity Cache) 104 us = 0.69 of the 8 TB/s peak against 157 us
// for the slot-parallel form (and 194-208 us before the store was ordered); V = 2.8e5: 28.5 us = 0.72 against 42.7 us; at the bench
// window (V = 1.8e4: 288 waves, one per CU, the frames of a voxel serial in its lane) 7.8 us against 5.5 us — stores below
// ~45k voxels keep the slot-parallel form.
template <int W, bool STAMPS>
__global__ __launch_bounds__(64) void k_residual_v(FactorView f, const double *__restrict__ poses, int head, int end,
                                                   double *__restrict__ partial, const int *__restrict__ gate, long long *__restrict__ stamps) {
  const int gate_v = gate ? *gate : 1;
  const int lane = threadIdx.x;
  const int v = head + blockIdx.x * 64 + lane;
  const size_t vs = (size_t)f.vs, fs = (size_t)W * vs;
  const int vc = v < end ? v : end - 1;
  long long st0 = 0, st1 = 0, st2 = 0, st3 = 0;
  if (STAMPS) st0 = clock64();
  unsigned int occm = f.occ[vc];
  double acc[10];
#pragma unroll
  for (int k = 0; k < 10; k++) acc[k] = f.fix[(size_t)k * vs + vc];
  double coe = f.coe[vc];
  asm volatile("" : "+v"(occm), "+v"(coe), "+v"(acc[0]));       // keep the loads above the exit
  if (gate_v == 0) return;
  if (v >= end) occm = 0;
  // trip 2: all frames the wave sees
  double c[W][10];
#pragma unroll
  for (int i = 0; i < W; i++) {
    const bool on = (occm >> i) & 1u;
    if (__ballot(on) != 0ull) {                                   // wave-uniform
#pragma unroll
      for (int k = 0; k < 10; k++) c[i][k] = on ? f.cl[(size_t)k * fs + (size_t)i * vs + v] : 0.0;
    } else {
#pragma unroll
      for (int k = 0; k < 10; k++) c[i][k] = 0.0;
    }
  }
#pragma unroll
  for (int i = 0; i < W; i++) {
    const bool on = (occm >> i) & 1u;
    if (__ballot(on) != 0ull) {
      if (on) {
        const Cl10 w = cluster_transform_exact(c[i][0], c[i][1], c[i][2], c[i][3], c[i][4], c[i][5], c[i][6], c[i][7], c[i][8], c[i][9], poses + 12 * i);
        acc[0] += w.p00; acc[1] += w.p10; acc[2] += w.p20; acc[3] += w.p11; acc[4] += w.p21; acc[5] += w.p22;
        acc[6] += w.v0; acc[7] += w.v1; acc[8] += w.v2; acc[9] += w.n;
      }
    }
  }
  if (STAMPS) st1 = clock64();
  double r = 0.0;
  if (v < end) {
#pragma unroll
    for (int k = 0; k < 10; k++) f.pcr[(size_t)k * vs + v] = acc[k];      // pcr_adds[a]  VM:319
    const double Nv = acc[9];
    double iN = __builtin_amdgcn_rcp(Nv);
    iN = iN * (2.0 - Nv * iN);
    iN = iN * (2.0 - Nv * iN);
    const double b0 = acc[6] * iN, b1 = acc[7] * iN, b2 = acc[8] * iN;
    double w0, w1, w2, V[9];
    eig3_sym_dev(acc[0] * iN - b0 * b0, acc[1] * iN - b1 * b0, acc[2] * iN - b2 * b0, acc[3] * iN - b1 * b1, acc[4] * iN - b2 * b1,
                 acc[5] * iN - b2 * b2, w0, w1, w2, V);
    if (STAMPS) st2 = clock64();
    f.eigval[0 * vs + v] = w0; f.eigval[1 * vs + v] = w1; f.eigval[2 * vs + v] = w2;
#pragma unroll
    for (int k = 0; k < 9; k++) f.eigvec[(size_t)k * vs + v] = V[k];
    r = coe * w0;                                                         // voxel_map.hpp:323
  }
  r = wave_sum_to_lane63(r);
  if (lane == 63) partial[blockIdx.x] = r;
  if (STAMPS && lane == 0 && blockIdx.x < 2048) {
    st3 = clock64();
    long long *o = stamps + (size_t)blockIdx.x * 4;
    o[0] = st0; o[1] = st1; o[2] = st2; o[3] = st3;
  }
}

// out[j] = sum_b partial[b*nout + j]   (deterministic, fixed order).  256 threads = 16 outputs x 16 partial groups,
// so ~nout/16 workgroups keep every CU busy on the (nb x nout) partial slab.
__global__ __launch_bounds__(256) void k_reduce_partials(const double *__restrict__ partial, int nb, int nout, double *__restrict__ out,
                                                        const int *__restrict__ gate) {
  __shared__ double s[256];
  const int gate_v = gate ? *gate : 1;            // requested together with the first 16 partials (one memory trip, see k_residual_s)
  const int j = threadIdx.x & 15, q = threadIdx.x >> 4;
  const int o = blockIdx.x * 16 + j, oc = o < nout ? o : nout - 1;
  double pv[16];
#pragma unroll
  for (int k = 0; k < 16; k++) { const int b = q + 16 * k; pv[k] = partial[(size_t)(b < nb ? b : 0) * nout + oc]; }
#pragma unroll
  for (int k = 0; k < 16; k++) asm volatile("" : "+v"(pv[k]));
  if (gate_v == 0) return;
  double acc = 0.0;
  if (o < nout) {
#pragma unroll
    for (int k = 0; k < 16; k++) acc += (q + 16 * k < nb) ? pv[k] : 0.0;
    for (int b = q + 256; b < nb; b += 16) acc += partial[(size_t)b * nout + o];
  }
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
// and E_{v,i} the diagonal-block terms not covered by the rank-3 form (VM:242-248; its -0.5 hat(jjt) term cancels the
// antisymmetric part of the rot-rot block exactly, so only the symmetric part is formed).  The identity is checked against
// the oracle's literal per-pair accumulation in tests/ (and SURVEY.md 3.4).
// Workgroup = TV voxels x W frames (one thread per (voxel, frame) slot) per tile:
//   phase A: slot threads build their 3 rows x 6 columns of G in LDS, accumulate E / gradient / residual privately;
//   phase B: the workgroup contracts the tile G^T C G on the matrix cores.
// Output per workgroup ("tile layout", NOUT2 doubles): [NU upper-triangle 16x16 accumulator tiles in MFMA register order
// | E 21 per frame | g 6 per frame | r]; k_reduce_partials sums the workgroups, tl_fetch() maps it back to H(row, col).
// Structure chosen from MI355X measurements of the first version (VALU contraction, 5-wave tiles, full 60x60 LDS image in
// the epilogue: 34.5 us per pass at V = 1.8e4, in-kernel stamps in profiles/r01_k3_stamps.txt):
//   * phase A issues ALL of a slot's loads in one batch (no dependent second round trip on N != 0) and the loads of the
//     NEXT tile are issued before the current tile's contraction, so they fly under phase B;
//   * phase B is the dense contraction H += G^T C G on the matrix cores: v_mfma_f64_16x16x4_f64, each wave owns
//     ceil(NU / waves) of the NU upper-triangle 16x16 tiles of the (6W)^2 output (W = 10: 10 tiles over 5 waves);
//     LDS image G[k][GS] with GS = 16 (mod 32) doubles, so the 4 rows x 16 columns a wave reads per operand are
//     bank-conflict free (rows k, k+1 land in opposite halves of the 64 banks);
//   * the epilogue writes the accumulator tiles straight into the LDS H image (no zero fill).
// f64 MFMA fragment layout (cdna_hip_programming.md, "f64 MFMA does NOT use these maps"):
//   A[i = l & 15][k = l >> 4], B[k = l >> 4][j = l & 15], D[row = (l >> 4) + 4 r][col = l & 15], r = 0..3.
typedef double v4f64 __attribute__((ext_vector_type(4)));

constexpr int hess2_tv(int W) {   // voxels per tile: ~256 / W slot threads = exactly 4 waves, multiple of 8, K-split divisible
  return W == 2 ? 128 : W == 3 ? 80 : W == 4 ? 64 : W == 5 ? 48 : W == 6 ? 40 : (W == 7 || W == 8) ? 32 : (W == 9 || W == 10) ? 24 : 16;
}
constexpr int gcd_i(int a, int b) { return b == 0 ? a : gcd_i(b, a % b); }

__device__ __forceinline__ double rcp_f64(double x) {
  double y = __builtin_amdgcn_rcp(x);
  y = y * (2.0 - x * y);
  y = y * (2.0 - x * y);
  return y;
}

template <int W>
struct HessCfg2 {
  static constexpr int TV = hess2_tv(W);
  static constexpr int NT = 256;                         // 4 waves: one per SIMD (5 waves measured 2x slower per tile)
  static constexpr int NC = 6 * W;
  static constexpr int NT16 = (NC + 15) / 16;
  static constexpr int NCP = 16 * NT16;
  // row stride of the LDS images in doubles.  ODD: the slot threads of one frame write rows 3 vl, 3 vl + 1, 3 vl + 2 — with the earlier
  // stride = 16 (mod 32) their stores all fell on two bank groups (12-way conflicts); an odd stride spreads them, and the MFMA operand
  // reads (4 rows x 16 consecutive doubles per wave) stay at 2-3 passes, far from binding
  static constexpr int GS = ((NCP % 32 == 16) ? NCP : NCP + 16) + 1;
  static constexpr int NK = 3 * TV;
  static constexpr int NWV = NT / 64;
  static constexpr int NU = NT16 * (NT16 + 1) / 2;       // upper-triangle 16x16 tiles
  static constexpr int KS = NWV / gcd_i(NU, NWV);        // K-splits so that NU*KS units divide evenly over the waves
  static constexpr int UPW = NU * KS / NWV;              // (tile, k-split) units per wave
  static constexpr int KSTEPS = NK / 4 / KS;             // MFMAs per unit
  static constexpr int NOUT = NC * NC + NC + 1;           // full layout [H | g | r]
  static constexpr int EB = NU * 256;                     // tile layout: offset of the per-frame E blocks
  static constexpr int GB = EB + 21 * W;                  //              offset of the gradient
  static constexpr int RB = GB + 6 * W;                   //              offset of the residual
  static constexpr int NOUT2 = RB + 1;
  static constexpr int NG8 = NT / 8;
  static_assert(TV * W <= NT && TV % 8 == 0 && (NK / 4) % KS == 0 && (NU * KS) % NWV == 0, "tile shape");
  // PRESCALE: a second LDS image holds c_k * G, so the MFMA loop carries no f64 VALU multiply (f64 VALU and f64 MFMA share the issue on
  // gfx950: one multiply per MFMA cost 21 of 95 cycles, tools/micro/mfma_lds.hip); every window except W = 3 has the LDS for it
  static constexpr bool PRESCALE = ((size_t)2 * NK * GS + NK + W * 12 + 8) * sizeof(double) <= (size_t)150 * 1024;
  static constexpr size_t LDS_MAIN = (size_t)NK * GS * (PRESCALE ? 2 : 1) + NK + W * 12 + 8;
  static constexpr size_t LDS_EPI = (size_t)(KS > 1 ? NU * 256 : 0) + (size_t)28 * NG8 + 8;
  static constexpr size_t LDS_BYTES = (LDS_MAIN > LDS_EPI ? LDS_MAIN : LDS_EPI) * sizeof(double);
};

struct SlotLoad {   // everything one (voxel, frame) slot thread reads from HBM
  double n, c[9], coe, l0, l1, l2, NN, vs0, vs1, vs2, U[9];
  bool valid;
};

// `occm` = the voxel's occupancy mask (f.occ[v], requested one step earlier by the caller): the 10 cluster scalars of an EMPTY slot are
// not read (VM:203-205 `if (sig_orig[i].N != 0)`; 57 % of the slots of the bench window are empty: 8.3 MB of the 23 MB the pass
// fetched in round 2), the slot then counts as n = 0.
template <int W>
__device__ __forceinline__ void slot_load(const FactorView &f, int v, int fi, int end, unsigned int occm, SlotLoad &q) {
  const size_t vs = (size_t)f.vs, fs = (size_t)W * vs;
  q.valid = v < end;
  if (q.valid) {
    const double *cp = f.cl + (size_t)fi * vs + v;
    const bool on = (occm >> fi) & 1u;
    q.n = on ? cp[9 * fs] : 0.0;
#pragma unroll
    for (int k = 0; k < 9; k++) q.c[k] = on ? cp[(size_t)k * fs] : 0.0;
    q.coe = f.coe[v];
    q.l0 = f.eigval[v]; q.l1 = f.eigval[vs + v]; q.l2 = f.eigval[2 * vs + v];
    q.NN = f.pcr[9 * vs + v]; q.vs0 = f.pcr[6 * vs + v]; q.vs1 = f.pcr[7 * vs + v]; q.vs2 = f.pcr[8 * vs + v];
#pragma unroll
    for (int k = 0; k < 9; k++) q.U[k] = f.eigvec[(size_t)k * vs + v];
  }
}

// Phase A of the Hessian pass for one (voxel, frame) slot: the slot's three rows g1, g2, h of G (6 columns each), the scales c_k,
// and its contributions to the diagonal-block remainder E (Err 6 | Ert 9 | Ett 6), the gradient and the residual (VM:167-275).
// R = the frame's pose [R | t]; `first` = this thread adds the voxel's coe * lambda_0 (one thread per voxel does).
__device__ __forceinline__ void slot_terms(const SlotLoad &q, const double *R, bool first, double *g1, double *g2, double *hh, double &ck1, double &ck2, double &ck3,
                                           double *Err, double *Ert, double *Ett, double *gj, double &rres) {
if (q.valid) {
  const double coe = q.coe, l0 = q.l0, NN = q.NN;
  // reciprocals by v_rcp_f64 + two Newton steps (full f64 precision to an ulp) instead of five IEEE divisions per slot
  const double inn = rcp_f64(NN);
  const double c1 = 2.0 * rcp_f64(l0 - q.l1), c2 = 2.0 * rcp_f64(l0 - q.l2);      // VM:201
  ck1 = coe * c1; ck2 = coe * c2; ck3 = coe * (-2.0 * inn * inn);
  if (first) rres += coe * l0;                                    // VM:275
  const double n = q.n;
  if (n != 0.0) {
    const double pxx = q.c[0], pxy = q.c[1], pxz = q.c[2], pyy = q.c[3], pyz = q.c[4], pzz = q.c[5];
    const double vx = q.c[6], vy = q.c[7], vz = q.c[8];
    const double u00 = q.U[0], u01 = q.U[1], u02 = q.U[2], u10 = q.U[3], u11 = q.U[4], u12 = q.U[5], u20 = q.U[6], u21 = q.U[7], u22 = q.U[8];
    const double bx = q.vs0 * inn, by = q.vs1 * inn, bz = q.vs2 * inn;                                      // vBar (VM:190)
    
    const double k0 = u00, k1 = u10, k2 = u20;
    const double a00 = R[0] * k0 + R[3] * k1 + R[6] * k2, a01 = R[1] * k0 + R[4] * k1 + R[7] * k2, a02 = R[2] * k0 + R[5] * k1 + R[8] * k2;
    const double a10 = R[0] * u01 + R[3] * u11 + R[6] * u21, a11 = R[1] * u01 + R[4] * u11 + R[7] * u21, a12 = R[2] * u01 + R[5] * u11 + R[8] * u21;
    const double a20 = R[0] * u02 + R[3] * u12 + R[6] * u22, a21 = R[1] * u02 + R[4] * u12 + R[7] * u22, a22 = R[2] * u02 + R[5] * u12 + R[8] * u22;
    const double tx = R[9] - bx, ty = R[10] - by, tz = R[11] - bz;                                           // VM:224
    const double s0 = k0 * tx + k1 * ty + k2 * tz, s1 = u01 * tx + u11 * ty + u21 * tz, s2 = u02 * tx + u12 * ty + u22 * tz;
    const double pa00 = pxx * a00 + pxy * a01 + pxz * a02, pa01 = pxy * a00 + pyy * a01 + pyz * a02, pa02 = pxz * a00 + pyz * a01 + pzz * a02;
    const double pa10 = pxx * a10 + pxy * a11 + pxz * a12, pa11 = pxy * a10 + pyy * a11 + pyz * a12, pa12 = pxz * a10 + pyz * a11 + pzz * a12;
    const double pa20 = pxx * a20 + pxy * a21 + pxz * a22, pa21 = pxy * a20 + pyy * a21 + pyz * a22, pa22 = pxz * a20 + pyz * a21 + pzz * a22;
    const double wx = pa00 + s0 * vx, wy = pa01 + s0 * vy, wz = pa02 + s0 * vz;                              // combo1 = hat(w) VM:228
    const double c2x = R[0] * vx + R[1] * vy + R[2] * vz + n * tx;                                           // combo2 VM:229
    const double c2y = R[3] * vx + R[4] * vy + R[5] * vz + n * ty;
    const double c2z = R[6] * vx + R[7] * vy + R[8] * vz + n * tz;
    const double qx = vy * a02 - vz * a01, qy = vz * a00 - vx * a02, qz = vx * a01 - vy * a00;              // viRiTuk VM:221
    const double d0 = c2x * k0 + c2y * k1 + c2z * k2;
    const double d1 = c2x * u01 + c2y * u11 + c2z * u21;
    const double d2 = c2x * u02 + c2y * u12 + c2z * u22;
    const double j0 = 2.0 * (wy * a02 - wz * a01) * inn, j1 = 2.0 * (wz * a00 - wx * a02) * inn, j2 = 2.0 * (wx * a01 - wy * a00) * inn;
    const double j3 = 2.0 * d0 * k0 * inn, j4 = 2.0 * d0 * k1 * inn, j5 = 2.0 * d0 * k2 * inn;
    gj[0] += coe * j0; gj[1] += coe * j1; gj[2] += coe * j2; gj[3] += coe * j3; gj[4] += coe * j4; gj[5] += coe * j5;   // VM:235-236
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
    const double wa = wx * a00 + wy * a01 + wz * a02;
    const double t00 = a01 * pxz - a02 * pxy, t10 = a02 * pxx - a00 * pxz, t20 = a00 * pxy - a01 * pxx;
    const double t01 = a01 * pyz - a02 * pyy, t11 = a02 * pxy - a00 * pyz, t21 = a00 * pyy - a01 * pxy;
    const double t02 = a01 * pzz - a02 * pyz, t12 = a02 * pxz - a00 * pzz, t22 = a00 * pyz - a01 * pxz;
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
    Ert[0] += e2 * qx * k0; Ert[1] += e2 * qx * k1; Ert[2] += e2 * qx * k2;
    Ert[3] += e2 * qy * k0; Ert[4] += e2 * qy * k1; Ert[5] += e2 * qy * k2;
    Ert[6] += e2 * qz * k0; Ert[7] += e2 * qz * k1; Ert[8] += e2 * qz * k2;
    const double e3 = e2 * n;
    Ett[0] += e3 * k0 * k0; Ett[1] += e3 * k0 * k1; Ett[2] += e3 * k0 * k2; Ett[3] += e3 * k1 * k1; Ett[4] += e3 * k1 * k2; Ett[5] += e3 * k2 * k2;
  }
}
}

// Device-resident LM loop, single rank: the accept / reject bookkeeping of the PREVIOUS iteration (VM:467-494) rides in the
// prologue of the Hessian pass instead of being a kernel of its own.  Every workgroup re-derives the decision from r1 and the
// residual pass' partials (same summation order everywhere, so the decision is bit-identical), runs only after an accepted
// step (then on the trial poses xt, which the bookkeeping copies to x), and workgroup 0 writes the new LM state — nothing a
// late workgroup still reads depends on what workgroup 0 changes.  (vba_kernels_lm.hpp defines the pieces.)
struct LmDev;
__device__ int lm_prev_stop(const LmDev *s);
__device__ double lm_r1(const LmDev *s);
__device__ const double *lm_xt(const LmDev *s);
__device__ void lm_update_apply(LmDev *s, double r2, int W);
__device__ __forceinline__ double lm_sum_partials(const double *__restrict__ r2_dev, int nb, int lane) {   // one wave; every lane gets the sum
  double pv[8];
#pragma unroll
  for (int k = 0; k < 8; k++) { const int b = lane + 64 * k; pv[k] = r2_dev[(nb > 0 && b < nb) ? b : 0]; }
  double acc = 0.0;
#pragma unroll
  for (int k = 0; k < 8; k++) acc += (lane + 64 * k < nb) ? pv[k] : 0.0;
  for (int b = lane + 512; b < nb; b += 64) acc += r2_dev[b];
  for (int m = 32; m >= 1; m >>= 1) acc += __shfl_xor(acc, m, 64);
  return acc;
}

// LI-BA rides along: when `li.dev` is set the launch has ONE workgroup more than `nwg` (block 0), and that workgroup computes the IMU factors'
// Hessian / gradient (li_imu_body, vba_kernels_li.hpp: VM:551-567) while the others do the lidar pass — the two are independent, the
// IMU part is a single workgroup of ~28 us, and as a kernel of its own it sat in front of this pass on the stream.
struct LiDev;
struct LiJob { const LmDev *lm; LiDev *dev; const double *imu; double *himu, *gimu; };
__device__ void li_imu_body(const LmDev *s, LiDev *li, const double *imu, double *himu, double *gimu, double *lds);

template <int W>
__global__ __launch_bounds__(HessCfg2<W>::NT) void k_hessian2(FactorView f, const double *__restrict__ poses, int head, int end,
                                                             int ntiles, double *__restrict__ partial, const int *__restrict__ gate,
                                                             long long *__restrict__ stamps, LmDev *lm, const double *__restrict__ k4_partial, int k4_nb,
                                                             int nwg, LiJob li) {
  using C = HessCfg2<W>;
  extern __shared__ __attribute__((aligned(16))) double lds[];
  // (every wave of this kernel holds ~300 registers, so a CU takes ONE workgroup: the IMU workgroup is block 0 — dispatched first —
  //  and the host launches one lidar workgroup fewer than there are CUs, or it would run after them instead of beside them)
  if (li.dev && blockIdx.x == 0) {
    li_imu_body(li.lm, li.dev, li.imu, li.himu, li.gimu, lds);
    return;
  }
  const int bid = (int)blockIdx.x - (li.dev ? 1 : 0);
  // A workgroup takes a CONTIGUOUS, balanced range of tiles (round 2 dealt them with stride nwg): its rows of the SoA store are then
  // 3 TV consecutive voxels per scalar — 576 B at W = 10 — instead of three 192-byte pieces that each straddle a 128-byte line and
  // share it with a workgroup on another XCD (which fetched it again).
  const int tile0 = (int)(((long long)bid * ntiles) / nwg), tile1 = (int)(((long long)(bid + 1) * ntiles) / nwg);
  const int my_vl = threadIdx.x % HessCfg2<W>::TV, my_fi = threadIdx.x / HessCfg2<W>::TV;
  // the occupancy mask of the first tile's voxel: requested before anything else, it comes back with the prologue's other reads
  unsigned int occ_nx = 0;
  if (my_fi < W && tile0 < tile1) { const int v0 = head + tile0 * HessCfg2<W>::TV + my_vl; occ_nx = f.occ[v0 < end ? v0 : end - 1]; }
  __shared__ int lm_dec[2];
  int gate_v = (gate && !lm) ? *gate : 1;     // consumed after the first tile's loads have been requested (one trip, not two)
  double lm_r2 = 0.0;
  if (lm) {
    if (threadIdx.x < 64) {
      lm_r2 = lm_sum_partials(k4_partial, k4_nb, threadIdx.x);
      if (threadIdx.x == 0) {
        const double r1 = lm_r1(lm);
        const int stop = lm_prev_stop(lm);
        const bool accept = (r1 - lm_r2) > 0, nstop = fabs((r1 - lm_r2) / r1) < 1e-6;
        lm_dec[0] = (!stop && accept && !nstop) ? 1 : 0;
        lm_dec[1] = stop;
      }
    }
    poses = lm_xt(lm);
  }
  // diagnostic stamps (stamps == nullptr in production): per workgroup [start, prologue, A0, B0, A1, B1, ..., reduce, end]
#define VBA_STAMP(i) do { if (stamps && threadIdx.x == 0) stamps[(size_t)bid * 16 + (i)] = wall_clock64(); } while (0)
  VBA_STAMP(0);
  double *G = lds;                            // [NK][GS]
  double *GA = C::PRESCALE ? G + (size_t)C::NK * C::GS : G;   // [NK][GS] rows scaled by c_k (the A operand)
  double *cK = GA + (size_t)C::NK * C::GS;    // [NK]
  double *sp = cK + C::NK;                    // [W][12]
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;

  for (int t = tid; t < W * 12; t += C::NT) sp[t] = poses[t];
  constexpr int PADC = C::GS - C::NC;                               // padded columns (none at W = 8) stay zero for the whole kernel
  if constexpr (PADC > 0)
    for (int t = tid; t < C::NK * PADC; t += C::NT) { G[(size_t)(t / PADC) * C::GS + C::NC + t % PADC] = 0.0; if (C::PRESCALE) GA[(size_t)(t / PADC) * C::GS + C::NC + t % PADC] = 0.0; }

  const int vl = tid % C::TV, fi = tid / C::TV;
  const bool slot_thread = fi < W;
  v4f64 acc[C::UPW];
#pragma unroll
  for (int t = 0; t < C::UPW; t++) acc[t] = (v4f64){0.0, 0.0, 0.0, 0.0};
  double Err[6] = {0, 0, 0, 0, 0, 0}, Ert[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0}, Ett[6] = {0, 0, 0, 0, 0, 0};
  double gj[6] = {0, 0, 0, 0, 0, 0};
  double rres = 0.0;

  SlotLoad nx;
  nx.valid = false;
  if (slot_thread && tile0 < tile1) slot_load<W>(f, head + tile0 * C::TV + vl, fi, end, occ_nx, nx);
  if (slot_thread && tile0 + 1 < tile1) { const int v1 = head + (tile0 + 1) * C::TV + vl; occ_nx = f.occ[v1 < end ? v1 : end - 1]; }   // one tile ahead
  asm volatile("" : "+v"(nx.n));
  if (gate_v == 0) return;                    // uniform
  __syncthreads();
  if (lm) {
    const int run = lm_dec[0], was_stopped = lm_dec[1];
    if (bid == 0 && threadIdx.x < 64 && !was_stopped) lm_update_apply(lm, lm_r2, W);
    if (!run) return;                         // uniform: rejected step or converged -> no Hessian pass (VM:443)
  }
  VBA_STAMP(1);
  int stamp_i = 2;

  for (int tile = tile0; tile < tile1; tile++) {
    // ---------------- phase A (registers of this tile were loaded one iteration ago)
    if (slot_thread) {
      const SlotLoad q = nx;
      double g1[6] = {0, 0, 0, 0, 0, 0}, g2[6] = {0, 0, 0, 0, 0, 0}, hh[6] = {0, 0, 0, 0, 0, 0};
      double ck1 = 0.0, ck2 = 0.0, ck3 = 0.0;
      slot_terms(q, sp + 12 * fi, fi == 0, g1, g2, hh, ck1, ck2, ck3, Err, Ert, Ett, gj, rres);
      double *g = G + (size_t)(3 * vl) * C::GS + 6 * fi;
#pragma unroll
      for (int d = 0; d < 6; d++) { g[d] = g1[d]; g[C::GS + d] = g2[d]; g[2 * C::GS + d] = hh[d]; }
      if (C::PRESCALE) {
        double *ga = GA + (size_t)(3 * vl) * C::GS + 6 * fi;
#pragma unroll
        for (int d = 0; d < 6; d++) { ga[d] = g1[d] * ck1; ga[C::GS + d] = g2[d] * ck2; ga[2 * C::GS + d] = hh[d] * ck3; }
      }
      if (fi == 0) { cK[3 * vl] = ck1; cK[3 * vl + 1] = ck2; cK[3 * vl + 2] = ck3; }
      // the next tile's loads fly under this tile's contraction
      const int nt = tile + 1;
      nx.valid = false;
      if (nt < tile1) {
        slot_load<W>(f, head + nt * C::TV + vl, fi, end, occ_nx, nx);
        if (nt + 1 < tile1) { const int v2 = head + (nt + 1) * C::TV + vl; occ_nx = f.occ[v2 < end ? v2 : end - 1]; }
      }
    }
    __syncthreads();
    if (stamp_i < 12) { VBA_STAMP(stamp_i); stamp_i++; }
    // ---------------- phase B: MFMA contraction of the tile's NK rows; unit = (16x16 tile, k-split)
    // The wave's UPW units are advanced ROUND-ROBIN, one k-step each per round: consecutive MFMAs then write different
    // accumulators (a chain on one accumulator ran at ~115 cycles per v_mfma_f64_16x16x4_f64 — its dependent latency — against
    // 64 cycles of issue), and the operands of round r + 1 are requested from LDS before the MFMAs of round r are issued (left
    // to itself the compiler waits for every operand pair right before its MFMA).
    {
      const int kr = lane >> 4, cl = lane & 15;
      const double *gap[C::UPW], *gbp[C::UPW], *ckp[C::UPW];
#pragma unroll
      for (int t = 0; t < C::UPW; t++) {
        const int unit = wv * C::UPW + t;
        const int u = unit % C::NU, ksp = unit / C::NU;
        int p = u, ta = 0;
        while (p >= C::NT16 - ta) { p -= C::NT16 - ta; ta++; }
        const int tb = ta + p;
        const int kb = ksp * C::KSTEPS * 4;
        gap[t] = GA + (size_t)(kb + kr) * C::GS + 16 * ta + cl;
        gbp[t] = G + (size_t)(kb + kr) * C::GS + 16 * tb + cl;
        ckp[t] = cK + kb + kr;
      }
      double ar[C::UPW], cr[C::UPW], br[C::UPW];
#pragma unroll
      for (int t = 0; t < C::UPW; t++) { ar[t] = gap[t][0]; cr[t] = C::PRESCALE ? 1.0 : ckp[t][0]; br[t] = gbp[t][0]; }
#pragma unroll
      for (int ks = 0; ks < C::KSTEPS; ks++) {
        double av[C::UPW], bv[C::UPW];
#pragma unroll
        for (int t = 0; t < C::UPW; t++) { av[t] = C::PRESCALE ? ar[t] : ar[t] * cr[t]; bv[t] = br[t]; }
        if (ks + 1 < C::KSTEPS) {
          const int k = 4 * (ks + 1);
#pragma unroll
          for (int t = 0; t < C::UPW; t++) { ar[t] = gap[t][(size_t)k * C::GS]; if (!C::PRESCALE) cr[t] = ckp[t][k]; br[t] = gbp[t][(size_t)k * C::GS]; }
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int t = 0; t < C::UPW; t++) acc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(av[t], bv[t], acc[t], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    __syncthreads();
    if (stamp_i < 12) { VBA_STAMP(stamp_i); stamp_i++; }
  }
  VBA_STAMP(12);

  // ---------------- epilogue: accumulator tiles go to HBM in register order (coalesced 512 B per wave store)
  double *out = partial + (size_t)bid * C::NOUT2;
  double *Tb = lds;                                   // [NU][256] staging to combine k-splits
  double *Eb = Tb + (C::KS > 1 ? C::NU * 256 : 0);    // [28][NG8]
  if (C::KS > 1) {
#pragma unroll
    for (int t = 0; t < C::UPW; t++) {
      const int unit = wv * C::UPW + t;
      if (unit / C::NU != 0) {
        const int u = unit % C::NU;
#pragma unroll
        for (int r = 0; r < 4; r++) {
          if (unit / C::NU == 1) Tb[u * 256 + r * 64 + lane] = acc[t][r];
        }
      }
    }
    for (int ksp = 2; ksp < C::KS; ksp++) {
      __syncthreads();
#pragma unroll
      for (int t = 0; t < C::UPW; t++) {
        const int unit = wv * C::UPW + t;
        if (unit / C::NU == ksp) {
          const int u = unit % C::NU;
#pragma unroll
          for (int r = 0; r < 4; r++) Tb[u * 256 + r * 64 + lane] += acc[t][r];
        }
      }
    }
  }
  // E / gradient / residual: 8-lane groups never straddle a frame (TV % 8 == 0).  Data-parallel-primitive adds: the same 28 x 3
  // steps through __shfl_xor (ds_bpermute, ~100 cycles of LDS round trip per dependent step) took most of the 7.7 us epilogue.
#pragma unroll
  for (int k = 0; k < 6; k++) { Err[k] = group8_sum(Err[k]); Ett[k] = group8_sum(Ett[k]); gj[k] = group8_sum(gj[k]); }
#pragma unroll
  for (int k = 0; k < 9; k++) Ert[k] = group8_sum(Ert[k]);
  rres = group8_sum(rres);
  if ((tid & 7) == 0) {
    const int g8 = tid >> 3;
#pragma unroll
    for (int k = 0; k < 6; k++) { Eb[(size_t)k * C::NG8 + g8] = Err[k]; Eb[(size_t)(15 + k) * C::NG8 + g8] = Ett[k]; Eb[(size_t)(21 + k) * C::NG8 + g8] = gj[k]; }
#pragma unroll
    for (int k = 0; k < 9; k++) Eb[(size_t)(6 + k) * C::NG8 + g8] = Ert[k];
    Eb[(size_t)27 * C::NG8 + g8] = rres;
  }
  __syncthreads();
  VBA_STAMP(13);
#pragma unroll
  for (int t = 0; t < C::UPW; t++) {
    const int unit = wv * C::UPW + t;
    if (unit / C::NU == 0) {
      const int u = unit % C::NU;
#pragma unroll
      for (int r = 0; r < 4; r++) {
        double val = acc[t][r];
        if (C::KS > 1) val += Tb[u * 256 + r * 64 + lane];
        out[u * 256 + r * 64 + lane] = val;
      }
    }
  }
  for (int t = tid; t < 28 * W; t += C::NT) {
    const int k = t % 28, fr = t / 28;
    double sum = 0.0;
#pragma unroll
    for (int g = 0; g < C::TV / 8; g++) sum += Eb[(size_t)k * C::NG8 + fr * (C::TV / 8) + g];
    if (k < 21) out[C::EB + 21 * fr + k] = sum;
    else if (k < 27) out[C::GB + 6 * fr + (k - 21)] = sum;
    else if (fr == 0) out[C::RB] = sum;
  }
  VBA_STAMP(14);
#undef VBA_STAMP
}

// Tile layout -> H(row, col) of the (6W x 6W) Hessian (symmetric: the upper triangle is stored, VM:279-281 mirrors it).
template <int W>
__device__ __forceinline__ double tl_fetch(const double *__restrict__ red, int row, int col) {
  using C = HessCfg2<W>;
  if (row > col) { const int t = row; row = col; col = t; }
  const int ta = row >> 4, tb = col >> 4;
  const int u = ta * C::NT16 - ta * (ta - 1) / 2 + (tb - ta);
  const int rt = row & 15, ct = col & 15;
  double val = red[u * 256 + (rt >> 2) * 64 + (rt & 3) * 16 + ct];
  const int fr = row / 6;
  if (col / 6 == fr) {
    const int a = row - 6 * fr, b = col - 6 * fr;
    int idx;
    if (b < 3) idx = a * 3 - a * (a - 1) / 2 + (b - a);
    else if (a < 3) idx = 6 + 3 * a + (b - 3);
    else { const int a2 = a - 3, b2 = b - 3; idx = 15 + a2 * 3 - a2 * (a2 - 1) / 2 + (b2 - a2); }
    val += red[C::EB + 21 * fr + idx];
  }
  return val;
}

// Tile layout -> full layout [H | g | r] for host consumers (acc_evaluate2, *hess of the optimizers).
template <int W>
__global__ void k_tiles_to_full(const double *__restrict__ red, double *__restrict__ full) {
  using C = HessCfg2<W>;
  for (int t = blockIdx.x * blockDim.x + threadIdx.x; t < C::NOUT; t += gridDim.x * blockDim.x) {
    if (t < C::NC * C::NC) full[t] = tl_fetch<W>(red, t / C::NC, t % C::NC);
    else if (t < C::NC * C::NC + C::NC) full[t] = red[C::GB + (t - C::NC * C::NC)];
    else full[t] = red[C::RB];
  }
}

}  // namespace vba
