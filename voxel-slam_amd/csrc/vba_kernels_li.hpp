// Device-resident LM loop of LI_BA_Optimizer / LI_BA_OptimizerGravity::damping_iter (voxel_map.hpp:624-713, 872-975):
//   k_hessian -> k_reduce -> [all-reduce] -> k_li_imu (IMU factors: joc^T cov^-1 joc, VM:551-567) -> k_li_solve
//   -> k_residual -> k_li_update (IMU residual at the trial states + accept/reject, VM:675-706)
// The lidar part reuses the pose-only passes unchanged (they read the R,p view kept in LmDev); everything the IMU adds
// lives in LiDev and three flat device arrays.  Up to W = 10 (n = 15 W + 3 <= 153) the packed factor of the (n+1)-row
// augmented system fits the 160 KB LDS; for W = 11..16 (n <= 243) the same kernel keeps the staged matrix and L in a
// device scratch buffer (GL = true) and only the panel buffers in LDS.
#pragma once
#include <hip/hip_runtime.h>
#include "vba_hostmath.hpp"

namespace vba {

constexpr int LI_MAX_W = 16;
constexpr int LI_MAX_N = 15 * LI_MAX_W + 3;

struct LiDev {
  int W, n, nb, gravity, gauge, F;
  double imu_coef;
  double ex[LI_MAX_W * 12];     // accepted v, bg, ba, g per frame
  double ext[LI_MAX_W * 12];    // trial
  double tstamp[LI_MAX_W];
  double rimu[2];               // sum_f r^T cov^-1 r at the accepted states [0] / at the trial states [1]
  // speculative damping candidates 1..LM_SPEC-1 of k_li_solve (see k_lm_solve_m): what a consumed candidate adds to the trial state
  double ext_spec[LM_SPEC][LI_MAX_W * 12];   // trial v, bg, ba (the g slots are unused: gravity accumulates on the trial state, VM:921)
  double ginc_spec[LM_SPEC][4];              // dxi.tail(3)
  double binc_spec[LM_SPEC][LI_MAX_W * 6];   // bias increments of IMU_PRE::update_state (PI:296-303) per factor
};

__device__ __forceinline__ void li_state(const double *pose12, const double *ex12, double t, vbh::State &s) {
  s.t = t;
#pragma unroll
  for (int k = 0; k < 9; k++) s.R[k] = pose12[k];
#pragma unroll
  for (int k = 0; k < 3; k++) { s.p[k] = pose12[9 + k]; s.v[k] = ex12[k]; s.bg[k] = ex12[3 + k]; s.ba[k] = ex12[6 + k]; s.g[k] = ex12[9 + k]; }
}

// compact IMU Hessian layout: [pair (a, b), |a - b| <= 1][15][15] | (R, g col k) [15W][3] | (g row k, C) [3][15W] | corner [3][3]
__host__ __device__ inline int li_hb_pair(int a, int b) { return (3 * a + (b - a)) * 225; }
__host__ __device__ inline int li_hb_ne1(int W) { return (3 * W - 2) * 225; }
__host__ __device__ inline int li_hb_size(int W, int grav) { return li_hb_ne1(W) + (grav ? 90 * W + 9 : 0); }
__host__ __device__ inline double li_hb_get(const double *hb, int W, int n, int R, int C) {   // dense (R, C) from the compact image
  const int nw = 15 * W;
  if (R < nw && C < nw) { const int a = R / 15, b = C / 15; if (a - b > 1 || b - a > 1) return 0.0; return hb[li_hb_pair(a, b) + (R - 15 * a) * 15 + (C - 15 * b)]; }
  const int ne1 = li_hb_ne1(W);
  if (R < nw) return hb[ne1 + R * 3 + (C - nw)];
  if (C < nw) return hb[ne1 + 45 * W + (R - nw) * nw + C];
  return hb[ne1 + 90 * W + (R - nw) * 3 + (C - nw)];
}

// IMU part of divide_thread (VM:551-567 / 783-801).  imu[f] = the ImuPre image of factor f with `cov` REPLACED by cov^-1
// (cov is constant inside damping_iter; the host inverts it once per call, preintegration.hpp:166 does it per evaluation).
// Output-centric: every entry of the block-tridiagonal (+ gravity border) Hessian sums its <= 2 (corner: F) factors in
// ascending factor order, so the result does not depend on scheduling.  himu is COMPACT (li_hb_* below): the 3W-2 state
// blocks, then the gravity border and corner — 7.2k doubles at W = 10 instead of the 23k of the dense matrix, because
// the solve kernel pays ~2 us per batch of loads for data another kernel wrote.
// (one workgroup of blockDim.x threads, a multiple of 64; `lds` = the launch's dynamic LDS.  Runs as the extra workgroup of the lidar
//  Hessian launch k_hessian2 (256 threads) or as the kernel k_li_imu below)
constexpr int LI_IMU_NT = 256;
__device__ void li_imu_body(const LmDev *s, LiDev *li, const double *imu, double *himu, double *gimu, double *lds) {
  const int NTH = blockDim.x;
  if (s->stop || !s->is_calc_hess) return;
  const int W = li->W, F = li->F, nb = li->nb, n = li->n, grav = li->gravity, tid = threadIdx.x;
  long long *stp = ((s->pad & 64) && tid == 0) ? const_cast<long long *>(s->stamps) + 40 : nullptr;   // diagnostic
  if (stp) stp[0] = clock64();
  double *joc = lds, *cj = joc + (size_t)F * 15 * nb, *rr = cj + (size_t)F * 15 * nb, *cr = rr + F * 15, *qf = cr + F * 15;
  // (cov^-1 is read from memory where it is needed: keeping a copy in LDS put the footprint over 80 KB at W = 10, and two workgroups
  //  per CU — this one beside a workgroup of the lidar pass — need it under that)
  for (int t = tid; t < F * 15 * nb; t += NTH) joc[t] = 0.0;
  __syncthreads();
  if (tid < F) {
    vbh::State s1, s2;
    li_state(s->x + 12 * tid, li->ex + 12 * tid, li->tstamp[tid], s1);
    li_state(s->x + 12 * (tid + 1), li->ex + 12 * (tid + 1), li->tstamp[tid + 1], s2);
    vbh::imu_residual_jacobian(*reinterpret_cast<const vbh::ImuPre *>(imu + 304 * (size_t)tid), s1, s2, grav != 0, rr + 15 * tid, joc + (size_t)tid * 15 * nb, nb);
  }
  __syncthreads();
  if (stp) stp[1] = clock64();
  {   // cj_f = cov^-1 joc_f in 3 x 3 register tiles (6 LDS loads per 9 FMAs; one output per thread was LDS-bandwidth bound)
    const int ncb = nb / 3, ntask = F * 5 * ncb;
    for (int t = tid; t < ntask; t += NTH) {
      const int f = t / (5 * ncb), rem = t - f * 5 * ncb, kb = rem / ncb, cb = rem - kb * ncb;
      const double *jf = joc + (size_t)f * 15 * nb + 3 * cb, *cv = imu + 304 * (size_t)f + 79 + 45 * kb;
      double a[3][3] = {{0, 0, 0}, {0, 0, 0}, {0, 0, 0}};
#pragma unroll 5
      for (int k2 = 0; k2 < 15; k2++) {
        const double j0 = jf[k2 * nb], j1 = jf[k2 * nb + 1], j2 = jf[k2 * nb + 2];
        const double c0 = cv[k2], c1 = cv[15 + k2], c2 = cv[30 + k2];
        a[0][0] += c0 * j0; a[0][1] += c0 * j1; a[0][2] += c0 * j2;
        a[1][0] += c1 * j0; a[1][1] += c1 * j1; a[1][2] += c1 * j2;
        a[2][0] += c2 * j0; a[2][1] += c2 * j1; a[2][2] += c2 * j2;
      }
      double *o = cj + (size_t)f * 15 * nb + (size_t)(3 * kb) * nb + 3 * cb;
#pragma unroll
      for (int r = 0; r < 3; r++)
#pragma unroll
        for (int c = 0; c < 3; c++) o[r * nb + c] = a[r][c];
    }
  }
  if (tid < F * 15) {
    const int f = tid / 15, k = tid % 15;
    const double *ci = imu + 304 * (size_t)f + 79 + 15 * k;
    double a = 0;
    for (int k2 = 0; k2 < 15; k2++) a += ci[k2] * rr[15 * f + k2];
    cr[tid] = a;
  }
  __syncthreads();
  if (stp) stp[2] = clock64();
  if (tid < F) { double q = 0; for (int k = 0; k < 15; k++) q += rr[15 * tid + k] * cr[15 * tid + k]; qf[tid] = q; }
  // jtj_f(lr, lc) = sum_k joc_f[k][lr] cj_f[k][lc]
  auto jtj = [&](int f, int lr, int lc) -> double {
    const double *jf = joc + (size_t)f * 15 * nb + lr, *cf = cj + (size_t)f * 15 * nb + lc;
    double a = 0;
    for (int k = 0; k < 15; k++) a += jf[k * nb] * cf[k * nb];
    return a;
  };
  // state blocks on the matrix cores: a wave owns a 15 x 15 block pair (a, b); J_f^T (cov^-1 J_f) restricted to the block's rows and
  // columns is one 16 x 16 tile with K = 15 (four v_mfma_f64_16x16x4_f64, the 16th k padded with zeros); the <= 2 factors of a
  // diagonal block accumulate in ascending order in the same tile.  (As 3 x 3 register tiles on the VALU this was 19.6 of the
  // workgroup's 32 us: 540 LDS loads per thread.)
  {
    typedef double li_v4f64 __attribute__((ext_vector_type(4)));
    const int npair = 3 * W - 2, lane = tid & 63, wv = tid >> 6, nwv = NTH >> 6, m = lane & 15, kq = lane >> 4;
    for (int pr = wv; pr < npair; pr += nwv) {
      const int a = (pr + 1) / 3, b = a + ((pr + 1) % 3) - 1;
      li_v4f64 acc = {0.0, 0.0, 0.0, 0.0};
      for (int pass = 0; pass < 2; pass++) {
        int f, ro, co;
        if (b == a) { if (pass == 0) { f = a - 1; ro = 15; co = 15; } else { f = a; ro = 0; co = 0; } }
        else if (b == a + 1) { if (pass) continue; f = a; ro = 0; co = 15; }
        else { if (pass) continue; f = b; ro = 15; co = 0; }
        if (f < 0 || f >= F) continue;
        const double *jf = joc + (size_t)f * 15 * nb + ro + m, *cf = cj + (size_t)f * 15 * nb + co + m;   // (lane 15 of a row reads a neighbour: its products only reach the discarded row / column 15)
#pragma unroll
        for (int k0 = 0; k0 < 16; k0 += 4) {
          const int k = k0 + kq;
          const double av = k < 15 ? jf[k * nb] : 0.0, bv = k < 15 ? cf[k * nb] : 0.0;
          acc = __builtin_amdgcn_mfma_f64_16x16x4f64(av, bv, acc, 0, 0, 0);
        }
      }
      double *o = himu + li_hb_pair(a, b);
#pragma unroll
      for (int r = 0; r < 4; r++) { const int row = kq + 4 * r; if (row < 15 && m < 15) o[15 * row + m] = acc[r]; }
    }
  }
  if (grav) {                                                // gravity border VM:788-795 and the 3 x 3 corner
    for (int e = tid; e < 15 * W * 3; e += NTH) {
      const int R = e / 3, k = e - 3 * R, a = R / 15, r = R - 15 * a;
      double u1 = 0, u2 = 0;
      if (a >= 1) { u1 += jtj(a - 1, 15 + r, 30 + k); u2 += jtj(a - 1, 30 + k, 15 + r); }
      if (a <= W - 2) { u1 += jtj(a, r, 30 + k); u2 += jtj(a, 30 + k, r); }
      himu[li_hb_ne1(W) + R * 3 + k] = u1;
      himu[li_hb_ne1(W) + 45 * W + k * 15 * W + R] = u2;
    }
    if (tid < 9) {
      const int r = tid / 3, k = tid - 3 * r;
      double acc = 0;
      for (int f = 0; f < F; f++) acc += jtj(f, 30 + r, 30 + k);
      himu[li_hb_ne1(W) + 90 * W + 3 * r + k] = acc;
    }
  }
  if (tid < n) {                                             // gradient: gg_f(lr) = sum_k joc_f[k][lr] cr_f[k]
    auto gg = [&](int f, int lr) -> double {
      const double *jf = joc + (size_t)f * 15 * nb + lr;
      double a = 0;
      for (int k = 0; k < 15; k++) a += jf[k * nb] * cr[15 * f + k];
      return a;
    };
    double acc = 0;
    if (tid < 15 * W) {
      const int a = tid / 15, r = tid - 15 * a;
      if (a >= 1) acc += gg(a - 1, 15 + r);
      if (a <= W - 2) acc += gg(a, r);
    } else {
      for (int f = 0; f < F; f++) acc += gg(f, 30 + (tid - 15 * W));
    }
    gimu[tid] = acc;
  }
  __syncthreads();
  if (stp) stp[3] = clock64();
  if (tid == 0) { double q = 0; for (int f = 0; f < F; f++) q += qf[f]; li->rimu[0] = q; }
}
__global__ __launch_bounds__(LI_IMU_NT) void k_li_imu(const LmDev *s, LiDev *li, const double *imu, double *himu, double *gimu) {
  extern __shared__ __attribute__((aligned(16))) double lds_imu[];
  li_imu_body(s, li, imu, himu, gimu, lds_imu);
}

// lower triangle of the assembled Hessian (VM:565-578 / 803-814), before the gauge: imu_coef * IMU part + lidar 6-blocks.
// Branch-free (every load is issued unconditionally on a clamped index and selected afterwards) so that the 28 gathers a
// lane performs while the system is dealt to the accumulator tiles overlap instead of serialising behind branches.
template <int W>
__device__ __forceinline__ double li_hfull(const double *__restrict__ himu, const double *__restrict__ src, double coef, int n, int r, int c) {
  using C = HessCfg2<W>;
  const double vi = himu[(size_t)r * n + c];
  const int a = r / 15, lr = r - 15 * a, b = c / 15, lc = c - 15 * b;
  const bool lid = (r < 15 * W) && (c < 15 * W) && (lr < 6) && (lc < 6);
  int row = lid ? 6 * a + lr : 0, col = lid ? 6 * b + lc : 0;
  if (row > col) { const int t = row; row = col; col = t; }
  const int ta = row >> 4, tb = col >> 4;
  const int ut = ta * C::NT16 - ta * (ta - 1) / 2 + (tb - ta);
  const int rt = row & 15, ct = col & 15;
  const double v1 = src[ut * 256 + (rt >> 2) * 64 + (rt & 3) * 16 + ct];
  const int fr = row / 6;
  const bool dg = (col / 6 == fr);
  const int aa = row - 6 * fr, bb = dg ? col - 6 * fr : aa;
  int idx;
  if (bb < 3) idx = aa * 3 - aa * (aa - 1) / 2 + (bb - aa);
  else if (aa < 3) idx = 6 + 3 * aa + (bb - 3);
  else { const int a2 = aa - 3, b2 = bb - 3; idx = 15 + a2 * 3 - a2 * (a2 - 1) / 2 + (b2 - a2); }
  const double v2 = src[C::EB + 21 * fr + idx];
  return coef * vi + (lid ? v1 + (dg ? v2 : 0.0) : 0.0);
}

// (H + u D) dxi = -g for the 15W(+3) system in Eigen-LDLT pivot order, then the retraction of VM:661-671 / 921-934.
template <int W, int NT, bool GL>
__global__ __launch_bounds__(NT) void k_li_solve(LmDev *s, LiDev *li, const double *__restrict__ red, double *__restrict__ raw, int copy_raw,
                                                 const double *__restrict__ himu, const double *__restrict__ gimu, double *__restrict__ imu,
                                                 int n, int gauge, int grav, double coef, double *__restrict__ lscratch) {
  using C2 = HessCfg2<W>;
  constexpr int NMAX = 15 * W + 3, NP = ((NMAX + 1 + 15) / 16) * 16;
  using LC = LdltCfg<NP>;
  extern __shared__ __attribute__((aligned(16))) double lds[];
  const int sb = blockIdx.x;                      // damping candidate of this workgroup (speculative damping, see k_lm_solve_m)
  // GL: L (and the staged matrix before it) in device memory — written and read by this one workgroup only, ordered by its barriers
  constexpr size_t LSCR = (size_t)LC::LTOT > (size_t)NMAX * (NMAX + 1) / 2 ? (size_t)LC::LTOT : (size_t)NMAX * (NMAX + 1) / 2;
  double *Lst = GL ? lscratch + (size_t)sb * LSCR : lds, *Tp = GL ? lds : Lst + LC::LTOT, *P = Tp + NP * LC::LS;
  double *hd = P + NP * 8, *gs = hd + NMAX, *dsh = gs + NMAX, *xs = dsh + NMAX, *dxs = xs + NP, *red8 = dxs + NMAX;
  int *ord = (int *)(red8 + 32);
  const int tid = threadIdx.x;
  // Everything this kernel reads was written by OTHER kernels, in general on another XCD: each dependent trip to memory
  // costs ~2 us (measured 5k cycles).  So the sizes are kernel arguments and every load the prologue needs — LM flags, the
  // IMU Hessian, the lidar tiles — is issued up front, before the first branch, and consumed afterwards.
  const long long t_begin = clock64();
  // lower part of the compact IMU Hessian: blocks (a, a) and (a, a - 1), then the gravity rows and corner
  constexpr int NLB = (2 * W - 1) * 225, NLOW = NLB + 45 * W + 9, QH = (NLOW + NT - 1) / NT, NL = 36 * W * W, QL = (NL + NT - 1) / NT;
  const int nlow = NLB + (grav ? 45 * W + 9 : 0);
  double hv[QH], lv[QL];
#pragma unroll
  for (int q = 0; q < QH; q++) {
    const int e = tid + NT * q, ec = e < nlow ? e : 0;
    int idx;
    if (ec < NLB) { const int blk = ec / 225, a = (blk + 1) >> 1, b = a - (blk & 1); idx = li_hb_pair(a, b) + (ec - 225 * blk); }
    else idx = li_hb_ne1(W) + 45 * W + (ec - NLB);                  // (g row k, C) rows then the corner, contiguous
    hv[q] = himu[idx];
  }
  if (!copy_raw) {
#pragma unroll
    for (int q = 0; q < QL; q++) {
      const int e = tid + NT * q, ec = e < NL ? e : NL - 1;
      const int row = ec / (6 * W), col = ec - row * (6 * W);
      lv[q] = tl_fetch<W>(red, row, col);
    }
  }
  double gimu_v = 0.0, glid_v = 0.0;
  if (tid < n) {
    gimu_v = gimu[tid];
    if (!copy_raw && tid < 15 * W) { const int a = tid / 15, lr = tid - 15 * a; glid_v = (lr < 6) ? red[C2::GB + 6 * a + lr] : 0.0; }
  }
  const int stop = s->stop, calc = s->is_calc_hess, iter0 = s->iter, dbg = sb == 0 ? s->pad : 0, use_spec = s->use_spec;
  const double u0 = s->u, v0 = s->v, rimu0 = li->rimu[0], rlid0 = copy_raw ? 0.0 : red[C2::RB];
  if (stop || use_spec) return;
  double u = u0;
  { double vb = v0; for (int k = 0; k < sb; k++) { u = u * vb; vb = 2 * vb; } }                         // VM:696-697, sb times
  if ((dbg & 16) && tid == 0) s->stamps[0] = t_begin;
  const double *__restrict__ src = (copy_raw && !calc) ? raw : red;
  if (copy_raw) {                                  // multi-rank: the valid copy depends on is_calc_hess, so these loads come second
    if (calc && sb == 0)
      for (int t = tid; t < C2::NOUT2; t += NT) raw[t] = src[t];
#pragma unroll
    for (int q = 0; q < QL; q++) {
      const int e = tid + NT * q, ec = e < NL ? e : NL - 1;
      const int row = ec / (6 * W), col = ec - row * (6 * W);
      lv[q] = tl_fetch<W>(src, row, col);
    }
    if (tid < 15 * W) { const int a = tid / 15, lr = tid - 15 * a; glid_v = (lr < 6) ? src[C2::GB + 6 * a + lr] : 0.0; }
  }
  if (tid == 0 && calc && sb == 0) {
    const double rl = copy_raw ? src[C2::RB] : rlid0, r = coef * 0.5 * rimu0 + rl;
    s->r1 = r; if (iter0 == 0) s->resis_first = r;
    if ((dbg & 32) && iter0 < 3) { s->stamps[58 + 2 * iter0] = __double_as_longlong(rimu0); s->stamps[59 + 2 * iter0] = __double_as_longlong(rl); }
  }
  // the assembled lower triangle (VM:565-578) is staged in LDS (the region of L is free until the first panel); the
  // permuted gather into the accumulator tiles then never leaves the CU
  double *stage = Lst;
  for (int e = tid; e < n * (n + 1) / 2; e += NT) stage[e] = 0.0;
  __syncthreads();
#pragma unroll
  for (int q = 0; q < QH; q++) {
    const int e = tid + NT * q;
    if (e < nlow) {
      int R, Cc;
      if (e < NLB) { const int blk = e / 225, rc = e - 225 * blk, a = (blk + 1) >> 1, b = a - (blk & 1); R = 15 * a + rc / 15; Cc = 15 * b + rc % 15; }
      else if (e < NLB + 45 * W) { const int q2 = e - NLB, k = q2 / (15 * W); R = 15 * W + k; Cc = q2 - k * 15 * W; }
      else { const int q2 = e - NLB - 45 * W; R = 15 * W + q2 / 3; Cc = 15 * W + q2 % 3; }
      if (Cc <= R) stage[R * (R + 1) / 2 + Cc] = coef * hv[q];
    }
  }
  __syncthreads();
  if ((dbg & 16) && tid == 0) s->stamps[50] = clock64();
#pragma unroll
  for (int q = 0; q < QL; q++) {                    // lidar 6-blocks (hess_plus VM:509-517)
    const int e = tid + NT * q;
    const int row = e / (6 * W), col = e - row * (6 * W);
    if (e < NL && col <= row) {
      const int R = 15 * (row / 6) + row % 6, Cc = 15 * (col / 6) + col % 6;
      stage[R * (R + 1) / 2 + Cc] += lv[q];
    }
  }
  __syncthreads();
  if ((dbg & 16) && tid == 0) s->stamps[51] = clock64();
  if (tid < n) {
    double h = 1.0, g = 0.0;
    if (tid >= gauge) {
      h = stage[tid * (tid + 1) / 2 + tid];
      g = coef * gimu_v + glid_v;
    }
    hd[tid] = h; gs[tid] = g; dsh[tid] = fabs(h + u * h);
  }
  __syncthreads();
  if ((dbg & 16) && tid == 0) s->stamps[52] = clock64();
  // Elimination order = the STRUCTURE of the system ("eliminate the 9W velocity / bias block first"): the v, bg, ba
  // of frame 0, 1, ... W-1 (block-tridiagonal through the IMU factors, VM:551-567), then the 6W pose scalars (dense through the
  // lidar part), then gravity.  Column c of frame i's block then reaches only the rest of its block, frame i+1's block, the poses
  // of frames 0..i+1 (i-1, i, i+1 directly, the older ones as fill of the chain), gravity and the right-hand side: the trailing
  // update of most tiles is skipped (li_live_rows).  The reference's Eigen LDLT pivots by the largest remaining diagonal (VM:659);
  // the system is symmetric positive definite (damped), for which every order is backward stable — the results move in the last
  // digits, inside the bars of the LI parity tests (same bars as before).
  if (tid < n) {
    int o = tid;
    if (tid < 9 * W) { const int i = tid / 9; o = 15 * i + 6 + (tid - 9 * i); }
    else if (tid < 15 * W) { const int q = tid - 9 * W, i = q / 6; o = 15 * i + (q - 6 * i); }
    ord[tid] = o;
  }
  __syncthreads();
  // element (i, j) of the padded system: P (H + u D) P^T lower triangle, row n = -g, identity on the padding
  auto elem = [&](int i, int j) -> double {
    const int jc = j < n ? j : n - 1, ic = i < n ? i : n - 1;
    const int pj = ord[jc], pi = ord[ic];
    const int rr = pi > pj ? pi : pj, cc = pi > pj ? pj : pi;
    double a = stage[rr * (rr + 1) / 2 + cc];
    a = (rr < gauge || cc < gauge) ? ((rr == cc) ? 1.0 : 0.0) : a;
    a = (i == j) ? a + u * a : a;
    a = (i == n) ? -gs[pj] : a;
    a = (i > n || i < j) ? 0.0 : a;
    return (j >= n) ? ((i == j) ? 1.0 : 0.0) : a;
  };
  long long *stamps = ((dbg & 16) != 0) ? s->stamps : nullptr;
  if (stamps && tid == 0) stamps[1] = clock64();
  auto live = [&](int kb) -> unsigned {               // 16-row blocks that can hold a non-zero of L in the columns [8 kb, 8 kb + 8)
    const int c0 = 8 * kb, c1 = c0 + 7;
    auto span = [](int a, int b) -> unsigned { return (a < b) ? ((2u << ((b - 1) >> 4)) - 1u) & ~((1u << (a >> 4)) - 1u) : 0u; };   // rows [a, b)
    if (c0 >= 9 * W) return span(c0, NP);
    const int ihi = (c1 < 9 * W ? c1 : 9 * W - 1) / 9;                                   // last frame the panel touches
    unsigned m = span(c0, (9 * (ihi + 2) < 9 * W) ? 9 * (ihi + 2) : 9 * W);            // its own blocks and the next frame's
    m |= span(9 * W, (6 * (ihi + 2) < 6 * W) ? 9 * W + 6 * (ihi + 2) : 15 * W);        // poses of frames 0 .. ihi + 1
    m |= span(15 * W, n + 1);                                                            // gravity, right-hand side row
    if (c1 >= 9 * W) m |= span(9 * W, NP);                                               // a panel that straddles into the pose block
    return m;
  };
  ldlt_mfma<NP, NT>(Lst, Tp, P, n, elem, stamps, live);
  if (stamps && tid == 0) stamps[3] = clock64();
  if (tid < n) xs[tid] = Lst[LC::lat(n, tid)];                                  // z = D^-1 L^-1 P (-g)
  __syncthreads();
  const double x = ldlt_backsub<NP>(Lst, xs, n);
  if (stamps && tid == 0) stamps[4] = clock64();
  if (tid < n) dxs[ord[tid]] = x;
  __syncthreads();
  // retraction VM:661-671 / 921-934 and the bias increments of IMU_PRE::update_state (PI:296-303)
  // (candidates sb > 0 park what they would add in the *_spec buffers; k_li_update applies a candidate when it is consumed)
  double *gl = red8 + 16;
  if (grav && tid < 3) {
    if (sb == 0) { const double gn = li->ext[9 + tid] + dxs[n - 3 + tid]; gl[tid] = gn; }   // accumulates on x_stats_temp[0].g (VM:921)
    else li->ginc_spec[sb][tid] = dxs[n - 3 + tid];
  }
  __syncthreads();
  if (tid < W) {
    double E[9];
    so3_exp_dev(dxs + 15 * tid, E);
    const double *R = s->x + 12 * tid;
    double *Rt = (sb == 0 ? s->xt : s->xt_spec[sb]) + 12 * tid;
    double *et = (sb == 0 ? li->ext : li->ext_spec[sb]) + 12 * tid;
    for (int r = 0; r < 3; r++)
      for (int c = 0; c < 3; c++) Rt[3 * r + c] = R[3 * r] * E[c] + R[3 * r + 1] * E[3 + c] + R[3 * r + 2] * E[6 + c];
    for (int k = 0; k < 3; k++) Rt[9 + k] = R[9 + k] + dxs[15 * tid + 3 + k];
    for (int k = 0; k < 9; k++) et[k] = li->ex[12 * tid + k] + dxs[15 * tid + 6 + k];
    if (grav && sb == 0) for (int k = 0; k < 3; k++) et[9 + k] = gl[k];
  }
  if (tid >= 64 && tid < 64 + li->F) {
    const int f = tid - 64;
    if (sb == 0) {
      double *m = imu + 304 * (size_t)f;               // dtime 66, dbg 67, dba 70, dbg_buf 73, dba_buf 76
      for (int k = 0; k < 3; k++) {
        m[73 + k] = m[67 + k]; m[76 + k] = m[70 + k];
        m[67 + k] += dxs[15 * f + 9 + k]; m[70 + k] += dxs[15 * f + 12 + k];
      }
    } else {
      for (int k = 0; k < 6; k++) li->binc_spec[sb][6 * f + k] = dxs[15 * f + 9 + k];
    }
  }
  double q = tid < n ? dxs[tid] * (u * hd[tid] * dxs[tid] - gs[tid]) : 0.0;            // VM:673
  for (int m = 32; m >= 1; m >>= 1) q += __shfl_xor(q, m, 64);
  if ((tid & 63) == 0) red8[tid >> 6] = q;
  __syncthreads();
  if (tid == 0) {
    double t = 0;
    for (int w2 = 0; w2 < NT / 64; w2++) t += red8[w2];
    s->q1_spec[sb] = 0.5 * t;
    if (sb == 0) { s->q1 = 0.5 * t; s->spec_n = gridDim.x; s->spec_i = 0; }
  }
  if (stamps && tid == 0) stamps[5] = clock64();
}

// only_residual's IMU part at the trial states (VM:605-607 / 851-854) + the accept / reject bookkeeping of VM:675-706.
// One workgroup of 256 threads: lane f of wave 0 evaluates factor f's residual (a serial chain), thread (f, r) then forms row r of
// cov^-1 r_f (the 225 loads per factor that one lane used to issue alone), lane f closes r^T (cov^-1 r) in the reference's order;
// the bookkeeping stays on wave 0.  Everything the bookkeeping reads is requested before the residual chain starts.
constexpr int LI_UPD_NT = 256;
__global__ __launch_bounds__(LI_UPD_NT) void k_li_update(LmDev *s, LiDev *li, double *__restrict__ imu, const double *__restrict__ r2_dev, int nb) {
  __shared__ double rr_s[LI_MAX_W][15], a_s[LI_MAX_W][15];
  if (s->stop) return;
  const int tid = threadIdx.x, lane = tid & 63, W = li->W, F = li->F;
  // (hoisted: one memory trip together with the states below instead of one more after the residual)
  const double r1 = s->r1, q1 = s->q1, u0 = s->u, v0 = s->v;
  const int ntr = s->n_trace, mtr = s->max_trace, it = s->iter, spec_n = s->spec_n, spec_nx = s->spec_i + 1;
  double r2acc = 0.0;
  if (tid < 64) {
    if (nb > 0) for (int b = lane; b < nb; b += 64) r2acc += r2_dev[b];
    else if (lane == 0) r2acc = *r2_dev;
  }
  if (tid < F) {
    vbh::State s1, s2;
    li_state(s->xt + 12 * tid, li->ext + 12 * tid, li->tstamp[tid], s1);
    li_state(s->xt + 12 * (tid + 1), li->ext + 12 * (tid + 1), li->tstamp[tid + 1], s2);
    double rr[15];
    vbh::imu_residual_jacobian(*reinterpret_cast<const vbh::ImuPre *>(imu + 304 * (size_t)tid), s1, s2, li->gravity != 0, rr, nullptr, li->nb);
    for (int k = 0; k < 15; k++) rr_s[tid][k] = rr[k];
  }
  __syncthreads();
  if (tid < 15 * F) {
    const int f = tid / 15, r = tid - 15 * f;
    const double *m = imu + 304 * (size_t)f + 79 + 15 * r;
    double a = 0;
    for (int k = 0; k < 15; k++) a += m[k] * rr_s[f][k];
    a_s[f][r] = a;
  }
  __syncthreads();
  if (tid >= 64) return;
  double qi = 0.0;
  if (lane < F) for (int r = 0; r < 15; r++) qi += rr_s[lane][r] * a_s[lane][r];
  // (sum over factors in ascending order, like the reference's loop)
  double rimu = 0.0;
  for (int f = 0; f < F; f++) rimu += readlane_f64(qi, f);
  double r2;
  if (nb > 0) {
    double acc = r2acc;
    for (int m = 32; m >= 1; m >>= 1) acc += __shfl_xor(acc, m, 64);
    r2 = acc;
  } else {
    r2 = readlane_f64(r2acc, 0);
  }
  r2 += rimu * (li->imu_coef * 0.5);
  const bool have_spec = spec_nx < spec_n;        // a solved candidate for the damping a rejection leads to (k_li_solve)
  const int sc = have_spec ? spec_nx : 0;
  double q = r1 - r2, u = u0, v = v0;
  const bool accept = q > 0;
  if (accept) {
    for (int t = lane; t < 12 * W; t += 64) { s->x[t] = s->xt[t]; li->ex[t] = li->ext[t]; }
    const double one_three = 1.0 / 3;
    q = q / q1;
    v = 2;
    const double t = 2 * q - 1;
    q = 1 - t * t * t;
    u *= (q < one_three ? one_three : q);
  } else {
    u = u * v;
    v = 2 * v;
    if (lane < F) { double *m = imu + 304 * (size_t)lane; for (int k = 0; k < 3; k++) { m[67 + k] = m[73 + k]; m[70 + k] = m[76 + k]; } }   // VM:701-705
    if (have_spec) {
      // what the next launch of k_li_solve would do with this (u, H, x): trial poses and v/bg/ba from the accepted state, gravity on
      // top of the (rejected) trial value, bias increments on top of the restored ones (buffers = restored values)
      double gn[3] = {0, 0, 0};
      if (li->gravity) for (int k = 0; k < 3; k++) gn[k] = li->ext[9 + k] + li->ginc_spec[sc][k];
      for (int t = lane; t < 12 * W; t += 64) {
        s->xt[t] = s->xt_spec[sc][t];
        const int k = t % 12;
        if (k < 9) li->ext[t] = li->ext_spec[sc][t];
        else if (li->gravity) li->ext[t] = gn[k - 9];
      }
      if (lane < F) {
        double *m = imu + 304 * (size_t)lane;
        for (int k = 0; k < 6; k++) m[67 + k] = m[73 + k] + li->binc_spec[sc][6 * lane + k];
      }
    }
  }
  if (lane != 0) return;
  if (!accept && have_spec) { s->q1 = s->q1_spec[sc]; s->spec_i = sc; s->use_spec = 1; }
  else { s->use_spec = 0; s->spec_n = 0; }
  li->rimu[1] = rimu;
  s->r2 = r2;
  if (ntr < mtr) {
    double *t = s->trace + 5 * ntr;
    t[0] = r1; t[1] = r2; t[2] = u0; t[3] = v0; t[4] = q1;
    s->n_trace = ntr + 1;
  }
  s->u = u; s->v = v;
  s->is_calc_hess = accept ? 1 : 0;
  s->last_accepted = accept ? 1 : 0;
  if (!accept) s->all_accepted = 0;
  s->iter = it + 1;
  const int nstop = (fabs((r1 - r2) / r1) < 1e-6) ? 1 : 0;
  s->stop = nstop;
  s->run_res = nstop ? 0 : 1;
  s->run_hess = (accept && !nstop) ? 1 : 0;
}

// gathers up to 8 device arrays into one block (one D2H copy instead of one per array)
struct PackSegs { int n; const double *src[8]; size_t off[8], len[8]; };
__global__ void k_pack_segments(PackSegs p, double *__restrict__ dst) {
  const int s = blockIdx.y;
  if (s >= p.n) return;
  const double *src = p.src[s];
  double *d = dst + p.off[s];
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < p.len[s]; i += (size_t)gridDim.x * blockDim.x) d[i] = src[i];
}

}  // namespace vba
