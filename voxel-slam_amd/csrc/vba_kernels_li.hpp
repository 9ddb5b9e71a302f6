// Device-resident LM loop of LI_BA_Optimizer / LI_BA_OptimizerGravity::damping_iter (voxel_map.hpp:624-713, 872-975):
//   k_hessian -> k_reduce -> [all-reduce] -> k_li_imu (IMU factors: joc^T cov^-1 joc, VM:551-567) -> k_li_solve
//   -> k_residual -> k_li_update (IMU residual at the trial states + accept/reject, VM:675-706)
// The lidar part reuses the pose-only passes unchanged (they read the R,p view kept in LmDev); everything the IMU adds
// lives in LiDev and three flat device arrays.  Supported on the device for W <= 10 (n = 15 W + 3 <= 153: the packed
// factor of the (n+1)-row augmented system fits the 160 KB LDS); larger windows use the host solve in voxelba.hip.
#pragma once
#include <hip/hip_runtime.h>
#include "vba_hostmath.hpp"

namespace vba {

constexpr int LI_MAX_W = 10;
constexpr int LI_MAX_N = 15 * LI_MAX_W + 3;

struct LiDev {
  int W, n, nb, gravity, gauge, F;
  double imu_coef;
  double ex[LI_MAX_W * 12];     // accepted v, bg, ba, g per frame
  double ext[LI_MAX_W * 12];    // trial
  double tstamp[LI_MAX_W];
  double rimu[2];               // sum_f r^T cov^-1 r at the accepted states [0] / at the trial states [1]
};

__device__ __forceinline__ void li_state(const double *pose12, const double *ex12, double t, vbh::State &s) {
  s.t = t;
#pragma unroll
  for (int k = 0; k < 9; k++) s.R[k] = pose12[k];
#pragma unroll
  for (int k = 0; k < 3; k++) { s.p[k] = pose12[9 + k]; s.v[k] = ex12[k]; s.bg[k] = ex12[3 + k]; s.ba[k] = ex12[6 + k]; s.g[k] = ex12[9 + k]; }
}

// IMU part of divide_thread (VM:551-567 / 783-801).  imu[f] = the ImuPre image of factor f with `cov` REPLACED by cov^-1
// (cov is constant inside damping_iter; the host inverts it once per call, preintegration.hpp:166 does it per evaluation).
// Output-centric: every entry of the block-tridiagonal (+ gravity border) Hessian sums its <= 2 (corner: F) factors in
// ascending factor order, so the result does not depend on scheduling.  himu is dense n x n (entries off the band stay 0).
__global__ __launch_bounds__(256) void k_li_imu(const LmDev *__restrict__ s, LiDev *__restrict__ li, const double *__restrict__ imu, double *__restrict__ himu,
                                                double *__restrict__ gimu) {
  extern __shared__ __attribute__((aligned(16))) double lds[];
  if (s->stop || !s->is_calc_hess) return;
  const int W = li->W, F = li->F, nb = li->nb, n = li->n, grav = li->gravity, tid = threadIdx.x;
  double *joc = lds, *cj = joc + (size_t)F * 15 * nb, *rr = cj + (size_t)F * 15 * nb, *cr = rr + F * 15, *qf = cr + F * 15;
  for (int t = tid; t < F * 15 * nb; t += 256) joc[t] = 0.0;
  __syncthreads();
  if (tid < F) {
    vbh::State s1, s2;
    li_state(s->x + 12 * tid, li->ex + 12 * tid, li->tstamp[tid], s1);
    li_state(s->x + 12 * (tid + 1), li->ex + 12 * (tid + 1), li->tstamp[tid + 1], s2);
    vbh::imu_residual_jacobian(*reinterpret_cast<const vbh::ImuPre *>(imu + 304 * (size_t)tid), s1, s2, grav != 0, rr + 15 * tid, joc + (size_t)tid * 15 * nb, nb);
  }
  __syncthreads();
  for (int f = 0; f < F; f++) {                              // cj_f = cov^-1 joc_f
    const double *jf = joc + (size_t)f * 15 * nb, *cv = imu + 304 * (size_t)f + 79;
    for (int t = tid; t < 15 * nb; t += 256) {
      const int k = t / nb, c = t - k * nb;
      double a = 0;
      for (int k2 = 0; k2 < 15; k2++) a += cv[15 * k + k2] * jf[k2 * nb + c];
      cj[(size_t)f * 15 * nb + t] = a;
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
  if (tid < F) { double q = 0; for (int k = 0; k < 15; k++) q += rr[15 * tid + k] * cr[15 * tid + k]; qf[tid] = q; }
  // jtj_f(lr, lc) = sum_k joc_f[k][lr] cj_f[k][lc]
  auto jtj = [&](int f, int lr, int lc) -> double {
    const double *jf = joc + (size_t)f * 15 * nb + lr, *cf = cj + (size_t)f * 15 * nb + lc;
    double a = 0;
    for (int k = 0; k < 15; k++) a += jf[k * nb] * cf[k * nb];
    return a;
  };
  // state blocks: the (a, b) pair is uniform per iteration, every thread keeps one (r, c) of the 15 x 15 block
  if (tid < 225) {
    const int r = tid / 15, c = tid - 15 * r;
    for (int a = 0; a < W; a++)
      for (int b = (a > 0 ? a - 1 : 0); b <= (a + 1 < W ? a + 1 : W - 1); b++) {
        double acc = 0;
        if (b == a) { if (a >= 1) acc += jtj(a - 1, 15 + r, 15 + c); if (a <= W - 2) acc += jtj(a, r, c); }
        else if (b == a + 1) acc = jtj(a, r, 15 + c);
        else acc = jtj(b, 15 + r, c);
        himu[(size_t)(15 * a + r) * n + 15 * b + c] = acc;
      }
  }
  if (grav) {                                                // gravity border VM:788-795 and the 3 x 3 corner
    for (int e = tid; e < 15 * W * 3; e += 256) {
      const int R = e / 3, k = e - 3 * R, a = R / 15, r = R - 15 * a;
      double u1 = 0, u2 = 0;
      if (a >= 1) { u1 += jtj(a - 1, 15 + r, 30 + k); u2 += jtj(a - 1, 30 + k, 15 + r); }
      if (a <= W - 2) { u1 += jtj(a, r, 30 + k); u2 += jtj(a, 30 + k, r); }
      himu[(size_t)R * n + n - 3 + k] = u1;
      himu[(size_t)(n - 3 + k) * n + R] = u2;
    }
    if (tid < 9) {
      const int r = tid / 3, k = tid - 3 * r;
      double acc = 0;
      for (int f = 0; f < F; f++) acc += jtj(f, 30 + r, 30 + k);
      himu[(size_t)(n - 3 + r) * n + n - 3 + k] = acc;
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
  if (tid == 0) { double q = 0; for (int f = 0; f < F; f++) q += qf[f]; li->rimu[0] = q; }
}

// lower triangle of the assembled Hessian (VM:565-578 / 803-814), before the gauge: imu_coef * IMU part + lidar 6-blocks
template <int W>
__device__ __forceinline__ double li_hfull(const double *__restrict__ himu, const double *__restrict__ src, double coef, int n, int r, int c) {
  double v = coef * himu[(size_t)r * n + c];
  if (r < 15 * W && c < 15 * W) { const int a = r / 15, lr = r - 15 * a, b = c / 15, lc = c - 15 * b; if (lr < 6 && lc < 6) v += tl_fetch<W>(src, 6 * a + lr, 6 * b + lc); }
  return v;
}
// ------------------------------------------------------------------------------------------------
// Blocked LDL^T of a symmetric NP x NP system (NP a multiple of 16) by ONE workgroup, trailing matrix resident in MFMA
// accumulators.  Pivoting is static (the caller passes the matrix already permuted), which is what Eigen's LDLT amounts
// to once the order "largest |stored diagonal| first" has been fixed.
//   * the lower triangle lives as 16x16 f64 accumulator tiles (v_mfma_f64_16x16x4_f64 layout: lane l holds rows
//     (l >> 4) + 4 r, column l & 15), tiles dealt round-robin to the waves, longest-living tile columns first;
//   * per panel of 8 columns: the owners publish the panel P[row][8] to LDS; every row-lane factorises the 8x8 diagonal
//     block REDUNDANTLY in registers (no communication on the pivot chain) and forward-substitutes its own row, writing
//     L (kept for the back substitution, column blocks of 8, row stride 9 doubles: conflict-free operand reads) and
//     -T = -L D; two MFMAs per live tile apply the rank-8 update C += L (-T)^T;
//   * two barriers per panel instead of one or two per column.
// A right-hand side carried as an extra ROW (rhs_row) leaves D^-1 L^-1 b in that row of L.
// Zero pivots follow Eigen (ldlt_inplace: a column with |d| == 0 is left unscaled; solve(): |d| <= DBL_MIN gives 0).
template <int NP>
struct LdltCfg {
  static constexpr int NTL = NP / 16, NTILES = NTL * (NTL + 1) / 2, NBLK = NP / 8, LS = 9;
  static constexpr int LTOT = LS * 4 * NBLK * (NBLK + 1);                       // sum over panels of (NP - 8 kb) rows x LS
  __host__ __device__ static constexpr int lst_off(int kb) { return LS * 8 * (kb * NBLK - kb * (kb - 1) / 2); }
  static constexpr int DOUBLES = LTOT + NP * LS + NP * 8;                       // Lst | Tp | P
  __device__ static __forceinline__ int lat(int j, int i) { return lst_off(i >> 3) + (j - (i & ~7)) * LS + (i & 7); }   // L[j][i], j > i
};

template <int NP, int NT, typename F>
__device__ __forceinline__ void ldlt_mfma(double *__restrict__ Lst, double *__restrict__ Tp, double *__restrict__ P, int rhs_row, F elem) {
  using C = LdltCfg<NP>;
  constexpr int NW = NT / 64, TPW = (C::NTILES + NW - 1) / NW, LS = C::LS;
  const int tid = threadIdx.x, l = tid & 63, w = tid >> 6, lr = l >> 4, lc = l & 15;
  // tile t (ordered by tile column descending, then tile row) -> (ti, tj); wave w owns t = w, w + NW, ...
  int tti[TPW], ttj[TPW];
  v4f64 acc[TPW];
#pragma unroll
  for (int u = 0; u < TPW; u++) {
    const int t = w + NW * u;
    int tj = C::NTL - 1, rem = t;
    while (tj > 0 && rem >= C::NTL - tj) { rem -= C::NTL - tj; tj--; }          // column tj has NTL - tj tiles
    tti[u] = tj + rem; ttj[u] = tj;
    if (t < C::NTILES) {
#pragma unroll
      for (int r = 0; r < 4; r++) acc[u][r] = elem(16 * tti[u] + lr + 4 * r, 16 * ttj[u] + lc);
      if (ttj[u] == 0 && lc < 8)
#pragma unroll
        for (int r = 0; r < 4; r++) P[(16 * tti[u] + lr + 4 * r) * 8 + lc] = acc[u][r];
    }
  }
  for (int kb = 0; kb < C::NBLK; kb++) {
    const int k0 = 8 * kb;
    double *Lk = Lst + C::lst_off(kb);
    __syncthreads();
    if (tid < NP && tid >= k0) {
      const int i = tid, ib = i - k0;
      double D[8][8], dd[8], dinv[8], x[8], tm[8];
      bool ok[8];
#pragma unroll
      for (int r = 0; r < 8; r++)
#pragma unroll
        for (int c = 0; c <= r; c++) D[r][c] = P[(k0 + r) * 8 + c];
      {
        const double2 x0 = *reinterpret_cast<const double2 *>(P + i * 8), x1 = *reinterpret_cast<const double2 *>(P + i * 8 + 2),
                      x2 = *reinterpret_cast<const double2 *>(P + i * 8 + 4), x3 = *reinterpret_cast<const double2 *>(P + i * 8 + 6);
        x[0] = x0.x; x[1] = x0.y; x[2] = x1.x; x[3] = x1.y; x[4] = x2.x; x[5] = x2.y; x[6] = x3.x; x[7] = x3.y;
      }
#pragma unroll
      for (int c = 0; c < 8; c++) {            // right-looking elimination of the diagonal block, identical in every lane
        const double d = D[c][c];
        ok[c] = fabs(d) > 0.0;
        double inv = __builtin_amdgcn_rcp(d);
        inv = fma(fma(-d, inv, 1.0), inv, inv);
        inv = fma(fma(-d, inv, 1.0), inv, inv);
        dd[c] = d; dinv[c] = ok[c] ? inv : 0.0;
#pragma unroll
        for (int r = c + 1; r < 8; r++) D[r][c] = ok[c] ? D[r][c] * inv : D[r][c];      // L[r][c]
#pragma unroll
        for (int r = c + 1; r < 8; r++) {
          const double t = D[r][c] * d;                                                  // T[r][c]
#pragma unroll
          for (int c2 = c + 1; c2 <= r; c2++) D[r][c2] -= t * D[c2][c];
        }
      }
      const bool is_rhs = (i == rhs_row);
#pragma unroll
      for (int c = 0; c < 8; c++) {            // this lane's row against the block: l_c = (x_c - sum_m t_m Ld[c][m]) / d_c
        double sacc = x[c];
#pragma unroll
        for (int m = 0; m < c; m++) sacc -= tm[m] * D[c][m];
        double lv = ok[c] ? sacc * dinv[c] : sacc;
        if (is_rhs) lv = (fabs(dd[c]) > 2.2250738585072014e-308) ? sacc * dinv[c] : 0.0;
        lv = (ib > c) ? lv : 0.0;             // rows of the diagonal block: strictly lower part only
        tm[c] = lv * dd[c];
        Lk[ib * LS + c] = lv;
        Tp[i * LS + c] = -tm[c];
      }
    }
    __syncthreads();
    const int kn = k0 + 8;
    if (kn >= NP) break;
    const int tjn = kn >> 4, cb0 = kn & 15;
#pragma unroll
    for (int u = 0; u < TPW; u++) {
      const int t = w + NW * u;
      if (t < C::NTILES && ttj[u] >= tjn) {    // wave-uniform
        const double *la = Lk + (16 * tti[u] + lc - k0) * LS + lr, *tb = Tp + (16 * ttj[u] + lc) * LS + lr;
        acc[u] = __builtin_amdgcn_mfma_f64_16x16x4f64(la[0], tb[0], acc[u], 0, 0, 0);
        acc[u] = __builtin_amdgcn_mfma_f64_16x16x4f64(la[4], tb[4], acc[u], 0, 0, 0);
        if (ttj[u] == tjn && lc >= cb0 && lc < cb0 + 8)
#pragma unroll
          for (int r = 0; r < 4; r++) P[(16 * tti[u] + lr + 4 * r) * 8 + (lc - cb0)] = acc[u][r];
      }
    }
  }
  __syncthreads();
}

// x = L^-T z for the first n rows (z in xs on entry, x in xs on exit), 64 rows per wave, blocks from the bottom.
template <int NP>
__device__ __forceinline__ double ldlt_backsub(const double *__restrict__ Lst, double *__restrict__ xs, int n) {
  using C = LdltCfg<NP>;
  const int tid = threadIdx.x, wv = tid >> 6;
  const int nblk = (n + 63) >> 6;
  double x = (tid < n) ? xs[tid] : 0.0;
  for (int b = nblk - 1; b >= 0; b--) {
    const int hiR = (64 * b + 63 < n - 1) ? 64 * b + 63 : n - 1;
    if (wv == b) {
      for (int j = hiR; j > 64 * b; j--) {
        const double xj = readlane_f64(x, j - 64 * b);
        if (tid < j && tid < n) x -= Lst[C::lat(j, tid)] * xj;
      }
      if (tid < n) xs[tid] = x;
    }
    __syncthreads();
    if (wv < b && tid < n)
      for (int j = 64 * b; j <= hiR; j++) x -= Lst[C::lat(j, tid)] * xs[j];
  }
  return x;
}

// (H + u D) dxi = -g for the 15W(+3) system in Eigen-LDLT pivot order, then the retraction of VM:661-671 / 921-934.
template <int W, int NT>
__global__ __launch_bounds__(NT) void k_li_solve(LmDev *s, LiDev *li, const double *__restrict__ red, double *__restrict__ raw, int copy_raw,
                                                 const double *__restrict__ himu, const double *__restrict__ gimu, double *__restrict__ imu) {
  using C2 = HessCfg2<W>;
  constexpr int NMAX = 15 * W + 3, NP = ((NMAX + 1 + 15) / 16) * 16;
  using LC = LdltCfg<NP>;
  extern __shared__ __attribute__((aligned(16))) double lds[];
  double *Lst = lds, *Tp = Lst + LC::LTOT, *P = Tp + NP * LC::LS;
  double *hd = P + NP * 8, *gs = hd + NMAX, *dsh = gs + NMAX, *xs = dsh + NMAX, *dxs = xs + NP, *red8 = dxs + NMAX;
  int *ord = (int *)(red8 + 32);
  const int tid = threadIdx.x;
  if (s->stop) return;
  const int n = li->n, gauge = li->gauge, calc = s->is_calc_hess, iter0 = s->iter, grav = li->gravity;
  const double u = s->u, coef = li->imu_coef;
  const double *__restrict__ src = (copy_raw && !calc) ? raw : red;
  if (copy_raw && calc)
    for (int t = tid; t < C2::NOUT2; t += NT) raw[t] = src[t];
  if (tid == 0 && calc) { const double r = coef * 0.5 * li->rimu[0] + src[C2::RB]; s->r1 = r; if (iter0 == 0) s->resis_first = r; }
  if (tid < n) {
    double h = 1.0, g = 0.0;
    if (tid >= gauge) {
      h = li_hfull<W>(himu, src, coef, n, tid, tid);
      g = coef * gimu[tid];
      if (tid < 15 * W) { const int a = tid / 15, lr = tid - 15 * a; if (lr < 6) g += src[C2::GB + 6 * a + lr]; }
    }
    hd[tid] = h; gs[tid] = g; dsh[tid] = fabs(h + u * h);
  }
  __syncthreads();
  if (tid < n) {
    const double me = dsh[tid];
    int rank = 0;
    for (int j = 0; j < n; j++) { const double o = dsh[j]; rank += (o > me || (o == me && j < tid)) ? 1 : 0; }
    ord[rank] = tid;
  }
  __syncthreads();
  // element (i, j) of the padded system: P (H + u D) P^T lower triangle, row n = -g, identity on the padding
  auto elem = [&](int i, int j) -> double {
    if (j >= n) return (i == j) ? 1.0 : 0.0;
    if (i > n || i < j) return 0.0;
    const int pj = ord[j];
    if (i == n) return -gs[pj];
    const int pi = ord[i];
    const int rr = pi > pj ? pi : pj, cc = pi > pj ? pj : pi;
    double a = (rr < gauge || cc < gauge) ? ((rr == cc) ? 1.0 : 0.0) : li_hfull<W>(himu, src, coef, n, rr, cc);
    if (i == j) a += u * a;
    return a;
  };
  ldlt_mfma<NP, NT>(Lst, Tp, P, n, elem);
  if (tid < n) xs[tid] = Lst[LC::lat(n, tid)];                                  // z = D^-1 L^-1 P (-g)
  __syncthreads();
  const double x = ldlt_backsub<NP>(Lst, xs, n);
  if (tid < n) dxs[ord[tid]] = x;
  __syncthreads();
  // retraction VM:661-671 / 921-934 and the bias increments of IMU_PRE::update_state (PI:296-303)
  double *gl = red8 + 16;
  if (grav && tid < 3) { const double gn = li->ext[9 + tid] + dxs[n - 3 + tid]; gl[tid] = gn; }   // accumulates on x_stats_temp[0].g (VM:921)
  __syncthreads();
  if (tid < W) {
    double E[9];
    so3_exp_dev(dxs + 15 * tid, E);
    const double *R = s->x + 12 * tid;
    double *Rt = s->xt + 12 * tid;
    for (int r = 0; r < 3; r++)
      for (int c = 0; c < 3; c++) Rt[3 * r + c] = R[3 * r] * E[c] + R[3 * r + 1] * E[3 + c] + R[3 * r + 2] * E[6 + c];
    for (int k = 0; k < 3; k++) Rt[9 + k] = R[9 + k] + dxs[15 * tid + 3 + k];
    for (int k = 0; k < 9; k++) li->ext[12 * tid + k] = li->ex[12 * tid + k] + dxs[15 * tid + 6 + k];
    if (grav) for (int k = 0; k < 3; k++) li->ext[12 * tid + 9 + k] = gl[k];
  }
  if (tid >= 64 && tid < 64 + li->F) {
    const int f = tid - 64;
    double *m = imu + 304 * (size_t)f;               // dtime 66, dbg 67, dba 70, dbg_buf 73, dba_buf 76
    for (int k = 0; k < 3; k++) {
      m[73 + k] = m[67 + k]; m[76 + k] = m[70 + k];
      m[67 + k] += dxs[15 * f + 9 + k]; m[70 + k] += dxs[15 * f + 12 + k];
    }
  }
  double q = tid < n ? dxs[tid] * (u * hd[tid] * dxs[tid] - gs[tid]) : 0.0;            // VM:673
  for (int m = 32; m >= 1; m >>= 1) q += __shfl_xor(q, m, 64);
  if ((tid & 63) == 0) red8[tid >> 6] = q;
  __syncthreads();
  if (tid == 0) { double t = 0; for (int w2 = 0; w2 < NT / 64; w2++) t += red8[w2]; s->q1 = 0.5 * t; }
}

// only_residual's IMU part at the trial states (VM:605-607 / 851-854) + the accept / reject bookkeeping of VM:675-706.
__global__ __launch_bounds__(64) void k_li_update(LmDev *s, LiDev *li, double *__restrict__ imu, const double *__restrict__ r2_dev, int nb) {
  if (s->stop) return;
  const int lane = threadIdx.x, W = li->W, F = li->F;
  double qi = 0.0;
  if (lane < F) {
    vbh::State s1, s2;
    li_state(s->xt + 12 * lane, li->ext + 12 * lane, li->tstamp[lane], s1);
    li_state(s->xt + 12 * (lane + 1), li->ext + 12 * (lane + 1), li->tstamp[lane + 1], s2);
    double rr[15];
    const double *m = imu + 304 * (size_t)lane;
    vbh::imu_residual_jacobian(*reinterpret_cast<const vbh::ImuPre *>(m), s1, s2, li->gravity != 0, rr, nullptr, li->nb);
    for (int r = 0; r < 15; r++) { double a = 0; for (int k = 0; k < 15; k++) a += m[79 + 15 * r + k] * rr[k]; qi += rr[r] * a; }
  }
  // (sum over factors in ascending order, like the reference's loop)
  double rimu = 0.0;
  for (int f = 0; f < F; f++) rimu += readlane_f64(qi, f);
  double r2;
  if (nb > 0) {
    double acc = 0.0;
    for (int b = lane; b < nb; b += 64) acc += r2_dev[b];
    for (int m = 32; m >= 1; m >>= 1) acc += __shfl_xor(acc, m, 64);
    r2 = acc;
  } else {
    r2 = *r2_dev;
  }
  r2 += rimu * (li->imu_coef * 0.5);
  const double r1 = s->r1, q1 = s->q1, u0 = s->u, v0 = s->v;
  const int ntr = s->n_trace, mtr = s->max_trace, it = s->iter;
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
  }
  if (lane != 0) return;
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

}  // namespace vba
