// Device-resident Levenberg-Marquardt loop of Lidar_BA_Optimizer::damping_iter (voxel_map.hpp:422-497): the state
// (poses, u, v, flags, H, g, trace) lives in HBM, so one LM iteration is a fixed chain of launches
//   k_hessian -> k_reduce_partials -> [all-reduce] -> k_lm_solve -> k_residual -> k_sum_scalar -> [all-reduce] -> k_lm_update
// with no host round trip; data-dependent control (is_calc_hess, the 1e-6 stop test) is carried by device flags that the
// kernels test on entry.
#pragma once
#include <hip/hip_runtime.h>
#include "vba_ldlt.hpp"

namespace vba {

constexpr int LM_SPEC = 4;           // damping candidates solved per launch of the solve kernel (one workgroup each)

struct LmDev {
  double x[VBA_MAX_WIN_DEV * 12];    // x_stats       (accepted poses)
  double xt[VBA_MAX_WIN_DEV * 12];   // x_stats_temp  (trial poses)
  double u, v, r1, r2, q1, resis_first;
  int is_calc_hess, stop, iter, n_trace, all_accepted, last_accepted, max_trace, run_hess, run_res, pad;
  double trace[5 * 64];              // rows [r1, r2, u, v, q1]
  long long stamps[64];              // diagnostic (VBA_DEBUG_SOLVE bit 16)
  // speculative damping (see k_lm_solve_m): trial poses / q1 for the damping values the next LM_SPEC - 1 consecutive rejections would use
  int spec_n, spec_i, use_spec, pad2;
  double q1_spec[LM_SPEC];
  double xt_spec[LM_SPEC][VBA_MAX_WIN_DEV * 12];
};

__device__ __forceinline__ void so3_exp_dev(const double *w, double *R) {   // tools.hpp:51-66
  const double n = sqrt(w[0] * w[0] + w[1] * w[1] + w[2] * w[2]);
  if (n >= 1e-11) {
    const double a0 = w[0] / n, a1 = w[1] / n, a2 = w[2] / n;
    const double s = sin(n), c1 = 1.0 - cos(n);
    // K = hat(a), K^2 = a a^T - I
    R[0] = 1.0 + c1 * (a0 * a0 - 1.0); R[1] = -s * a2 + c1 * a0 * a1;    R[2] = s * a1 + c1 * a0 * a2;
    R[3] = s * a2 + c1 * a1 * a0;      R[4] = 1.0 + c1 * (a1 * a1 - 1.0); R[5] = -s * a0 + c1 * a1 * a2;
    R[6] = -s * a1 + c1 * a2 * a0;     R[7] = s * a0 + c1 * a2 * a1;     R[8] = 1.0 + c1 * (a2 * a2 - 1.0);
  } else {
    R[0] = 1; R[1] = 0; R[2] = 0; R[3] = 0; R[4] = 1; R[5] = 0; R[6] = 0; R[7] = 0; R[8] = 1;
  }
}


// `red` = the reduced [H | g | r] of the last Hessian pass.  It is never modified here: the gauge (first 6 rows/cols ->
// identity, JacT.head(6) = 0, VM:452-455) and the damping u*diag are applied while the system is loaded, so a rejected
// step re-reads the same H with a new u (VM:443) and `red[0..n^2)` doubles as *hess (VM:446).  With more than one rank
// the exchange step all-reduces `red` in place on every iteration, so the valid copy is kept in `raw` (copy_raw != 0).
// Elimination order = Eigen's LDLT pivoting: largest |diagonal| of the *stored* matrix first (ldlt_inplace::unblocked),
// realised as a rank computation (first index wins ties).
// (History, measured on MI355X at n = 60: unblocked LDS rows 58 us -> one wave blocked by pose 30.8 us -> this kernel 24 us,
//  profiles/r01_solve_ablation.txt; the two earlier kernels are no longer part of the library.)
//
// Blocked variant (vba_ldlt.hpp) for every supported window (n = 6W <= 96): trailing matrix in MFMA accumulators, panels
// of 8 columns, two barriers per panel.  Everything the kernel reads was written by other kernels (in general on another
// XCD, ~2 us per dependent trip), so the whole reduced system is requested up front and staged in LDS before the first branch.
//
// Speculative damping.  The solve is one workgroup on a 256-CU chip and the longest kernel of an iteration, and a REJECTED step
// repeats it on the same H and x with a damping that is known in advance (u <- u v, v <- 2 v, VM:485-486).  So the launch has
// LM_SPEC workgroups: workgroup b solves with the damping b consecutive rejections from now would use (the same f64 products the
// update would form, so every candidate is bit-identical to the sequential solve) and parks its trial poses and q1 in
// xt_spec[b] / q1_spec[b].  When the update rejects a step and a candidate is left it copies that candidate into xt / q1 and
// sets use_spec; the next launch of this kernel then returns at once.  An accepted step discards the candidates.  No field a
// workgroup reads on entry is written inside this kernel, so the workgroups need no ordering among themselves.
template <int W>
__global__ __launch_bounds__(256) void k_lm_solve_m(LmDev *s, const double *__restrict__ red, double *__restrict__ raw, int copy_raw) {
  const int sb = blockIdx.x;
  using C2 = HessCfg2<W>;
  constexpr int n = 6 * W, NT = 256, NP = ((n + 1 + 15) / 16) * 16;
  using LC = LdltCfg<NP>;
  constexpr int STG = n * (n + 1) / 2;
  __shared__ __attribute__((aligned(16))) double lds[LC::DOUBLES > STG ? LC::DOUBLES : STG];
  __shared__ double hd[n], gs[n], dsh[n], xs[NP], dxs[n], red8[8];
  __shared__ int ord[n];
  double *Lst = lds, *Tp = Lst + LC::LTOT, *P = Tp + NP * LC::LS, *stage = lds;
  const int tid = threadIdx.x;
  constexpr int NL = n * n, QL = (NL + NT - 1) / NT;
  double lv[QL];
  if (!copy_raw) {
#pragma unroll
    for (int q = 0; q < QL; q++) {
      const int e = tid + NT * q, ec = e < NL ? e : NL - 1;
      const int row = ec / n, col = ec - row * n;
      lv[q] = tl_fetch<W>(red, row, col);
    }
  }
  double g_v = (!copy_raw && tid < n) ? red[C2::GB + tid] : 0.0;
  double xr[12];
  if (tid < W)
#pragma unroll
    for (int k = 0; k < 12; k++) xr[k] = s->x[12 * tid + k];
  const long long t_begin = clock64();
  const int stop = s->stop, calc = s->is_calc_hess, iter0 = s->iter, dbg = s->pad, use_spec = s->use_spec;
  const double u0 = s->u, v0 = s->v, r_v = copy_raw ? 0.0 : red[C2::RB];
  if (stop || use_spec) return;
  double u = u0;
  { double vb = v0; for (int k = 0; k < sb; k++) { u = u * vb; vb = 2 * vb; } }                         // VM:485-486, sb times
  long long *stamps = ((dbg & 16) && tid == 0 && sb == 0) ? s->stamps : nullptr;
  if (stamps) stamps[0] = t_begin;
  const double *__restrict__ src = (copy_raw && !calc) ? raw : red;
  if (copy_raw) {
    if (calc && sb == 0)
      for (int t = tid; t < C2::NOUT2; t += NT) raw[t] = src[t];
#pragma unroll
    for (int q = 0; q < QL; q++) {
      const int e = tid + NT * q, ec = e < NL ? e : NL - 1;
      const int row = ec / n, col = ec - row * n;
      lv[q] = tl_fetch<W>(src, row, col);
    }
    if (tid < n) g_v = src[C2::GB + tid];
  }
  if (tid == 0 && calc && sb == 0) { const double r = copy_raw ? src[C2::RB] : r_v; s->r1 = r; if (iter0 == 0) s->resis_first = r; }   // VM:445, 449-450
#pragma unroll
  for (int q = 0; q < QL; q++) {
    const int e = tid + NT * q;
    const int row = e / n, col = e - row * n;
    if (e < NL && col <= row) stage[row * (row + 1) / 2 + col] = lv[q];
  }
  __syncthreads();
  if (tid < n) {
    const double h = tid < 6 ? 1.0 : stage[tid * (tid + 1) / 2 + tid];                                  // gauge VM:452-455
    hd[tid] = h; gs[tid] = tid < 6 ? 0.0 : g_v;
    dsh[tid] = fabs(h + u * h);
    ord[tid] = 0;
  }
  __syncthreads();
  {
    constexpr int parts = NT / n;
    const int i = tid % n, part = tid / n;
    if (part < parts) {
      const double me = dsh[i];
      constexpr int seg = (n + parts - 1) / parts;
      const int j0 = part * seg, j1 = (j0 + seg < n) ? j0 + seg : n;
      int cnt = 0;
#pragma unroll 8
      for (int j = j0; j < j1; j++) { const double o = dsh[j]; cnt += (o > me || (o == me && j < i)) ? 1 : 0; }
      atomicAdd(&ord[i], cnt);
    }
  }
  __syncthreads();
  const int my_rank = (tid < n) ? ord[tid] : 0;
  __syncthreads();
  if (tid < n) ord[my_rank] = tid;
  __syncthreads();
  auto elem = [&](int i, int j) -> double {        // P (Hess + u D) P^T lower triangle, row n = -g, identity on the padding
    const int jc = j < n ? j : n - 1, ic = i < n ? i : n - 1;
    const int pj = ord[jc], pi = ord[ic];
    const int rr = pi > pj ? pi : pj, cc = pi > pj ? pj : pi;
    double a = stage[rr * (rr + 1) / 2 + cc];
    a = (rr < 6 || cc < 6) ? ((rr == cc) ? 1.0 : 0.0) : a;
    a = (i == j) ? a + u * a : a;
    a = (i == n) ? -gs[pj] : a;
    a = (i > n || i < j) ? 0.0 : a;
    return (j >= n) ? ((i == j) ? 1.0 : 0.0) : a;
  };
  if (stamps) stamps[1] = clock64();
  ldlt_mfma<NP, NT>(Lst, Tp, P, n, elem, ((dbg & 16) && sb == 0) ? s->stamps : nullptr);
  if (stamps) stamps[3] = clock64();
  if (tid < n) xs[tid] = Lst[LC::lat(n, tid)];
  __syncthreads();
  const double x = ldlt_backsub<NP>(Lst, xs, n);
  if (stamps) stamps[4] = clock64();
  if (tid < n) dxs[ord[tid]] = x;
  __syncthreads();
  if (tid < W) {                                                                                        // VM:460-464
    double E[9];
    so3_exp_dev(dxs + 6 * tid, E);
    const double *R = xr;
    double *Rt = (sb == 0 ? s->xt : s->xt_spec[sb]) + 12 * tid;
#pragma unroll
    for (int r = 0; r < 3; r++)
#pragma unroll
      for (int c = 0; c < 3; c++) Rt[3 * r + c] = R[3 * r] * E[c] + R[3 * r + 1] * E[3 + c] + R[3 * r + 2] * E[6 + c];
#pragma unroll
    for (int k = 0; k < 3; k++) Rt[9 + k] = R[9 + k] + dxs[6 * tid + 3 + k];
  }
  double q = tid < n ? dxs[tid] * (u * hd[tid] * dxs[tid] - gs[tid]) : 0.0;                              // VM:465
  q = wave_sum_to_lane63(q);                                  // DPP adds (vba_kernels_factor.hpp), no LDS round trips
  if ((tid & 63) == 63) red8[tid >> 6] = q;
  __syncthreads();
  if (tid == 0) {
    const double q1 = 0.5 * (red8[0] + red8[1] + red8[2] + red8[3]);
    s->q1_spec[sb] = q1;
    if (sb == 0) { s->q1 = q1; s->spec_n = gridDim.x; s->spec_i = 0; }
  }
  if (stamps) stamps[5] = clock64();
}

// (Round 3 tried two single-wave forms of this solve for n <= 64 — lane i owns row i, no barriers, no tiles — and measured both slower on
//  MI355X at n = 60 than the blocked kernel's 24 us: rows in registers with the 1770 updates fully unrolled 31 us (a dependent f64
//  operation costs ~50 cycles with one wave per SIMD, so the 60 pivot steps are a ~350-cycle chain each: readlane, reciprocal + two
//  Newton steps, scale, first update), rows in LDS with rolled loops 62 us (every read-modify-write of a row element waits for its
//  own LDS round trip).  The blocked kernel's panel of 8 columns amortises that chain over 8 pivots.)

// Accept / reject bookkeeping of VM:467-494 (one thread).  r2_dev = the reduced residual of the trial poses.
// nb > 0: r2 is first summed here from the residual pass' nb workgroup partials (single rank: saves one launch);
// nb == 0: r2_dev already holds the (all-reduced) scalar.  One wave; every lane takes the (uniform) decision so that the
// pose copy x <- x_temp is a parallel copy and no dependent chain of single-lane global accesses remains.
__device__ int lm_prev_stop(const LmDev *s) { return s->stop; }
__device__ double lm_r1(const LmDev *s) { return s->r1; }
__device__ const double *lm_xt(const LmDev *s) { return s->xt; }

// One wave (all 64 lanes call it with the same r2).  Reads of the state come first, in one batch.
__device__ void lm_update_apply(LmDev *s, double r2, int W) {
  const int ntr = s->n_trace, mtr = s->max_trace, it = s->iter, spec_n = s->spec_n, spec_nx = s->spec_i + 1;
  const double r1 = s->r1, q1 = s->q1, u0 = s->u, v0 = s->v;
  const int lane = threadIdx.x & 63;
  const bool have_spec = spec_nx < spec_n;                  // a candidate for the damping a rejection leads to (k_lm_solve_m)
  const double *xs_ = s->xt_spec[have_spec ? spec_nx : 0];
  double sp0 = 0.0, sp1 = 0.0, sp2 = 0.0;
  if (have_spec) {
    if (lane < 12 * W) sp0 = xs_[lane];
    if (lane + 64 < 12 * W) sp1 = xs_[lane + 64];
    if (lane + 128 < 12 * W) sp2 = xs_[lane + 128];
  }
  const double q1_nx = s->q1_spec[have_spec ? spec_nx : 0];
  double xt0 = 0.0, xt1 = 0.0, xt2 = 0.0;                 // up to 192 pose scalars = 3 per lane
  if (lane < 12 * W) xt0 = s->xt[lane];
  if (lane + 64 < 12 * W) xt1 = s->xt[lane + 64];
  if (lane + 128 < 12 * W) xt2 = s->xt[lane + 128];
  double q = r1 - r2, u = u0, v = v0;
  const bool accept = q > 0;
  if (accept) {                                             // VM:473-483
    if (lane < 12 * W) s->x[lane] = xt0;
    if (lane + 64 < 12 * W) s->x[lane + 64] = xt1;
    if (lane + 128 < 12 * W) s->x[lane + 128] = xt2;
    const double one_three = 1.0 / 3;
    q = q / q1;
    v = 2;
    const double t = 2 * q - 1;
    q = 1 - t * t * t;                                      // pow(2q-1, 3)
    u *= (q < one_three ? one_three : q);
  } else {                                                  // VM:484-490
    u = u * v;
    v = 2 * v;
    if (have_spec) {                                        // the solve for this (u, H, x) has already been done
      if (lane < 12 * W) s->xt[lane] = sp0;
      if (lane + 64 < 12 * W) s->xt[lane + 64] = sp1;
      if (lane + 128 < 12 * W) s->xt[lane + 128] = sp2;
    }
  }
  if (lane != 0) return;
  if (!accept && have_spec) { s->q1 = q1_nx; s->spec_i = spec_nx; s->use_spec = 1; }
  else { s->use_spec = 0; s->spec_n = 0; }
  s->r2 = r2;
  if (ntr < mtr) {
    double *t = s->trace + 5 * ntr;
    t[0] = r1; t[1] = r2; t[2] = u0; t[3] = v0; t[4] = q1;
    s->n_trace = ntr + 1;
  }
  s->u = u; s->v = v;
  s->is_calc_hess = accept ? 1 : 0;
  s->last_accepted = accept ? 1 : 0;
  if (!accept) s->all_accepted = 0;                         // is_converge = false   VM:489
  s->iter = it + 1;
  const int nstop = (fabs((r1 - r2) / r1) < 1e-6) ? 1 : 0;  // VM:492-493
  s->stop = nstop;
  s->run_res = nstop ? 0 : 1;
  s->run_hess = (accept && !nstop) ? 1 : 0;
}

// Stand-alone form (multi-rank flow, the last iteration of a call, callers that want the flags after every iteration).
__global__ __launch_bounds__(64) void k_lm_update(LmDev *s, const double *__restrict__ r2_dev, int nb, int W) {
  const int stop = s->stop;
  const double r2 = (nb > 0) ? lm_sum_partials(r2_dev, nb, threadIdx.x) : r2_dev[0];
  if (stop) return;
  lm_update_apply(s, r2, W);
}

}  // namespace vba
