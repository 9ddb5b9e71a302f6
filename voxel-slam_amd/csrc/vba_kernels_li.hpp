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
  for (int t = tid; t < F * 15 * nb; t += 256) {            // cj_f = cov^-1 joc_f
    const int f = t / (15 * nb), k = (t / nb) % 15, c = t % nb;
    const double *ci = imu + 304 * (size_t)f + 79 + 15 * k, *jf = joc + (size_t)f * 15 * nb;
    double a = 0;
    for (int k2 = 0; k2 < 15; k2++) a += ci[k2] * jf[k2 * nb + c];
    cj[t] = a;
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
  // entries: [0, NE1) state blocks (a, b = a-1..a+1), then gravity border, then g
  const int npair = 3 * W - 2, NE1 = npair * 225, NEG = grav ? (2 * 15 * W * 3 + 9) : 0, NG = n;
  for (int e = tid; e < NE1 + NEG + NG; e += 256) {
    int R, C, isg = 0;
    if (e < NE1) {
      const int pr = e / 225, r = (e % 225) / 15, c = e % 15;
      // pair index -> (a, b): a = (pr + 1) / 3, b = a + ((pr + 1) % 3) - 1
      const int a = (pr + 1) / 3, b = a + ((pr + 1) % 3) - 1;
      R = 15 * a + r; C = 15 * b + c;
    } else if (e < NE1 + NEG) {
      const int q = e - NE1;
      if (q < 15 * W * 3) { R = q / 3; C = n - 3 + q % 3; }
      else if (q < 2 * 15 * W * 3) { const int q2 = q - 15 * W * 3; R = n - 3 + q2 % 3; C = q2 / 3; }
      else { const int q2 = q - 2 * 15 * W * 3; R = n - 3 + q2 / 3; C = n - 3 + q2 % 3; }
    } else { R = e - NE1 - NEG; C = 0; isg = 1; }
    // factors touching row R / column C and the local indices inside their 30(+3) window
    const int aR = R < 15 * W ? R / 15 : -1, aC = C < 15 * W ? C / 15 : -1;
    double acc = 0;
    for (int f = 0; f < F; f++) {
      int lr, lc;
      if (aR >= 0) { if (aR == f) lr = R - 15 * f; else if (aR == f + 1) lr = 15 + R - 15 * (f + 1); else continue; }
      else lr = 30 + (R - (n - 3));
      const double *jf = joc + (size_t)f * 15 * nb;
      if (isg) {
        double a = 0;
        for (int k = 0; k < 15; k++) a += jf[k * nb + lr] * cr[15 * f + k];
        acc += a;
        continue;
      }
      if (aC >= 0) { if (aC == f) lc = C - 15 * f; else if (aC == f + 1) lc = 15 + C - 15 * (f + 1); else continue; }
      else lc = 30 + (C - (n - 3));
      const double *cf = cj + (size_t)f * 15 * nb;
      double a = 0;
      for (int k = 0; k < 15; k++) a += jf[k * nb + lr] * cf[k * nb + lc];
      acc += a;
    }
    if (isg) gimu[R] = acc; else himu[(size_t)R * n + C] = acc;
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
// element e of the packed, permuted, gauged and damped system (row n = right-hand side)
template <int W>
__device__ __forceinline__ double li_elem_init(int e, int n, int gauge, double u, double coef, const double *__restrict__ himu,
                                                        const double *__restrict__ src, const double *gs, const int *ord, const int *off, int *ij) {
  int lo = 0, hi = n;                          // largest j with off[j] <= e
  while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (off[mid] <= e) lo = mid; else hi = mid; }
  const int j = lo, i = j + (e - off[j]);
  *ij = (i << 16) | j;
  const int pj = ord[j];
  if (i == n) return -gs[pj];
  const int pi = ord[i];
  const int rr = pi > pj ? pi : pj, cc = pi > pj ? pj : pi;
  double a = (rr < gauge || cc < gauge) ? ((rr == cc) ? 1.0 : 0.0) : li_hfull<W>(himu, src, coef, n, rr, cc);
  if (i == j) a += u * a;
  return a;
}

// The elements a thread owns, as a compile-time recursion: every member is its own scalar (an indexed register array
// here turned into one 32-register tuple that was copied and spilled on every branch: 1600 spill instructions).
// Elements are dealt in packed order, so a thread's list is sorted by column: those with j > k form a suffix.
template <int Q, int NT>
struct LiElems {
  double v; int i, j;
  LiElems<Q - 1, NT> next;
  __device__ __forceinline__ void init(int e, int M, int n, const int *off, const double *Lm) {
    i = n; j = n; v = 0.0;                      // unused slot: never published, its dummy update reads cb[n]
    if (e < M) {
      int lo = 0, hi = n;                       // largest column with off[col] <= e
      while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (off[mid] <= e) lo = mid; else hi = mid; }
      j = lo; i = lo + (e - off[lo]);
      v = Lm[e];
    }
    next.init(e + NT, M, n, off, Lm);
  }
  __device__ __forceinline__ void publish(int k, double *cb) const { if (j == k) cb[i] = v; next.publish(k, cb); }
  // returns false once an element with j < k has been met (everything before it is finished too)
  __device__ __forceinline__ bool update(int k, int n, int e, const double *cb, double inv, bool ok, double invz, double *Lm) {
    if (!next.update(k, n, e + NT, cb, inv, ok, invz, Lm)) return false;
    if (j < k) return false;
    if (j == k) {
      if (i != k) {
        const double l = (i == n) ? v * invz : (ok ? v * inv : v);
        v = l; Lm[e] = l;
      }
    } else {
      v -= cb[i] * (cb[j] * inv);
    }
    return true;
  }
  __device__ __forceinline__ void emit_rhs(int n, double *xs) const { if (i == n && j < n) xs[j] = v; next.emit_rhs(n, xs); }
};
template <int NT>
struct LiElems<0, NT> {
  __device__ __forceinline__ void init(int, int, int, const int *, const double *) {}
  __device__ __forceinline__ void publish(int, double *) const {}
  __device__ __forceinline__ bool update(int, int, int, const double *, double, bool, double, double *) { return true; }
  __device__ __forceinline__ void emit_rhs(int, double *) const {}
};

// (H + u D) dxi = -g for the 15W(+3) system, Eigen-LDLT pivot order (largest |stored diagonal| first), then the
// retraction of VM:661-671 / 921-934.  One workgroup of NT threads; the lower triangle of the permuted matrix plus the
// right-hand side as an extra ROW (so that D^-1 L^-1 (-g) falls out of the factorisation) is dealt cyclically to the
// threads and lives in REGISTERS; per column k the owners publish column k, one barrier, every thread applies the rank-1
// update to the elements it owns.  L is kept packed in LDS for the back substitution.
template <int W, int NT>
__global__ __launch_bounds__(NT) void k_li_solve(LmDev *s, LiDev *li, const double *__restrict__ red, double *__restrict__ raw, int copy_raw,
                                                 const double *__restrict__ himu, const double *__restrict__ gimu, double *__restrict__ imu) {
  using C2 = HessCfg2<W>;
  constexpr int NMAX = 15 * W + 3;
  constexpr int MMAX = NMAX * (NMAX + 3) / 2;
  constexpr int Q = (MMAX + NT - 1) / NT;
  extern __shared__ __attribute__((aligned(16))) double lds[];
  double *Lm = lds;                              // [MMAX] packed columns: column j holds rows j..n (row n = rhs)
  double *colbuf = Lm + MMAX;                    // [2][NMAX + 1]
  double *hd = colbuf + 2 * (NMAX + 1), *gs = hd + NMAX, *dsh = gs + NMAX, *xs = dsh + NMAX, *dxs = xs + NMAX + 1;
  int *ord = (int *)(dxs + NMAX), *off = ord + NMAX;      // off[j] = first packed index of column j, off[n] = M
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
  if (tid <= n) off[tid] = tid * (n + 1) - tid * (tid - 1) / 2;
  __syncthreads();
  if (tid < n) {
    const double me = dsh[tid];
    int rank = 0;
    for (int j = 0; j < n; j++) { const double o = dsh[j]; rank += (o > me || (o == me && j < tid)) ? 1 : 0; }
    ord[rank] = tid;
  }
  __syncthreads();
  const int M = off[n];
  // stage the system through LDS with a ROLLED loop (the gather from the tile layout is register hungry), then deal the
  // elements to registers with a light unrolled loop; Lm[e] is overwritten by the final L value of the same element later
  for (int e = tid; e < M; e += NT) { int ij; Lm[e] = li_elem_init<W>(e, n, gauge, u, coef, himu, src, gs, ord, off, &ij); }
  __syncthreads();
  LiElems<Q, NT> el;
  el.init(tid, M, n, off, Lm);
  for (int k = 0; k < n; k++) {
    double *cb = colbuf + (k & 1) * (NMAX + 1);
    el.publish(k, cb);
    __syncthreads();
    const double dk = cb[k];
    const bool ok = fabs(dk) > 0.0;
    double inv = __builtin_amdgcn_rcp(dk);
    inv = fma(fma(-dk, inv, 1.0), inv, inv);
    inv = fma(fma(-dk, inv, 1.0), inv, inv);
    if (!ok) inv = 0.0;
    const double invz = (fabs(dk) > 2.2250738585072014e-308) ? inv : 0.0;      // D^-1 y with Eigen's tolerance (rhs row)
    el.update(k, n, tid, cb, inv, ok, invz, Lm);
  }
  el.emit_rhs(n, xs);                                                          // z = D^-1 L^-1 P (-g)
  __syncthreads();
  // x = L^-T z, 64 rows per wave, blocks from the bottom
  const int nblk = (n + 63) >> 6, wv = tid >> 6, lane = tid & 63;
  double x = (tid < n) ? xs[tid] : 0.0;
  for (int b = nblk - 1; b >= 0; b--) {
    if (wv == b) {
      const int hiR = (64 * b + 63 < n - 1) ? 64 * b + 63 : n - 1;
      for (int j = hiR; j > 64 * b; j--) {
        const double xj = readlane_f64(x, j - 64 * b);
        if (tid < j && tid < n) x -= Lm[off[tid] + (j - tid)] * xj;
      }
      if (tid < n) xs[tid] = x;
    }
    __syncthreads();
    if (wv < b && tid < n) {
      const int hiR = (64 * b + 63 < n - 1) ? 64 * b + 63 : n - 1;
      const double *col = Lm + off[tid] - tid;
      for (int j = 64 * b; j <= hiR; j++) x -= col[j] * xs[j];
    }
  }
  if (tid < n) dxs[ord[tid]] = x;
  __syncthreads();
  // retraction VM:661-671 / 921-934 and the bias increments of IMU_PRE::update_state (PI:296-303)
  double *gl = colbuf + 32;                                          // (colbuf is free again)
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
  if (lane == 0) colbuf[wv] = q;
  __syncthreads();
  if (tid == 0) { double t = 0; for (int w2 = 0; w2 < NT / 64; w2++) t += colbuf[w2]; s->q1 = 0.5 * t; }
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
