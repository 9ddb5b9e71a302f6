// Blocked LDL^T + back substitution for the small dense systems of the LM solvers (6W for Lidar_BA_Optimizer, 15W+3 for
// LI_BA_Optimizer): one workgroup, trailing matrix in MFMA accumulators.  Shared by vba_kernels_lm.hpp / vba_kernels_li.hpp.
#pragma once
#include <hip/hip_runtime.h>

namespace vba {

typedef double v4f64_l __attribute__((ext_vector_type(4)));

__device__ __forceinline__ double readlane_f64(double v, int lane) {
  int lo = __double2loint(v), hi = __double2hiint(v);
  lo = __builtin_amdgcn_readlane(lo, lane);
  hi = __builtin_amdgcn_readlane(hi, lane);
  return __hiloint2double(hi, lo);
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

// `live(kb)` = bit b set when rows [16 b, 16 b + 16) can hold a non-zero of L in the columns of panel kb (a SUPERSET of the
// structure is fine): the rank-8 update of a tile whose row block or column block is dead for the panel is skipped.  A dense
// system passes LdltDense.
struct LdltDense { __device__ __forceinline__ unsigned operator()(int) const { return ~0u; } };
template <int NP, int NT, typename F, typename LV = LdltDense>
__device__ __forceinline__ void ldlt_mfma(double *__restrict__ Lst, double *__restrict__ Tp, double *__restrict__ P, int rhs_row, F elem, long long *stamps = nullptr, LV live = LV()) {
  using C = LdltCfg<NP>;
  constexpr int NW = NT / 64, TPW = (C::NTILES + NW - 1) / NW, LS = C::LS;
  const int tid = threadIdx.x, l = tid & 63, w = tid >> 6, lr = l >> 4, lc = l & 15;
  // tile t (ordered by tile column descending, then tile row) -> (ti, tj); wave w owns t = w, w + NW, ...
  int tti[TPW], ttj[TPW];
  v4f64_l acc[TPW];
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
  if (stamps && tid == 0) stamps[2] = clock64();
  for (int kb = 0; kb < C::NBLK; kb++) {
    const int k0 = 8 * kb;
    double *Lk = Lst + C::lst_off(kb);
    __syncthreads();
    if (stamps && tid == 0 && kb < 20) stamps[8 + 2 * kb] = clock64();
    if ((tid | 63) >= k0 && (tid & ~63) < NP) {               // wave-uniform: some lane of this wave owns a row of the trailing matrix
      // The 8 x 8 diagonal block is factorised ONCE PER WAVE, spread over its lanes: lane r < 8 keeps row r of the block, a pivot and
      // the scaled pivot column reach the other lanes by v_readlane (no LDS, no barrier).  Every row lane used to factorise the
      // block redundantly in its own registers (~430 f64 instructions per panel, the constant 2.6-2.9k cycles of DESIGN.md 5/9);
      // the arithmetic per element is unchanged, so the factor is bit-identical.
      const bool rowlane = tid < NP && tid >= k0;
      const int i = rowlane ? tid : k0, ib = i - k0;
      double Dr[8], x[8];
      {
        const double *dr = P + (k0 + (l & 7)) * 8;
        const double2 d0 = *reinterpret_cast<const double2 *>(dr), d1 = *reinterpret_cast<const double2 *>(dr + 2),
                      d2 = *reinterpret_cast<const double2 *>(dr + 4), d3 = *reinterpret_cast<const double2 *>(dr + 6);
        Dr[0] = d0.x; Dr[1] = d0.y; Dr[2] = d1.x; Dr[3] = d1.y; Dr[4] = d2.x; Dr[5] = d2.y; Dr[6] = d3.x; Dr[7] = d3.y;
        const double2 x0 = *reinterpret_cast<const double2 *>(P + i * 8), x1 = *reinterpret_cast<const double2 *>(P + i * 8 + 2),
                      x2 = *reinterpret_cast<const double2 *>(P + i * 8 + 4), x3 = *reinterpret_cast<const double2 *>(P + i * 8 + 6);
        x[0] = x0.x; x[1] = x0.y; x[2] = x1.x; x[3] = x1.y; x[4] = x2.x; x[5] = x2.y; x[6] = x3.x; x[7] = x3.y;
      }
      const bool is_rhs = rowlane && (i == rhs_row);
#pragma unroll
      for (int c = 0; c < 8; c++) {
        // pivot c of the diagonal block.  f64 dependent-issue latency is ~40 cycles here, so the recurrence is arranged to keep
        // the chain pivot -> next pivot short: 1/d by v_rcp_f64 and two Newton steps in three levels (e^2 is formed beside y1)
        const double d = readlane_f64(Dr[c], c);
        const bool ok = fabs(d) > 0.0;
        const double y0 = __builtin_amdgcn_rcp(d);
        const double e = fma(-d, y0, 1.0);
        const double y1 = fma(e, y0, y0), e2 = e * e;
        const double inv = fma(e2, y1, y1);
        const double dinv = ok ? inv : 0.0;
        // this lane's row, right-looking: x[c] is final once columns < c have been applied
        double lv = ok ? x[c] * dinv : x[c];
        if (is_rhs) lv = (fabs(d) > 2.2250738585072014e-308) ? x[c] * dinv : 0.0;
        lv = (ib > c) ? lv : 0.0;               // rows of the diagonal block: strictly lower part only
        const double tmc = (ok && ib > c) ? x[c] : 0.0;   // T = L d = the unscaled value (0 for a zero pivot)
        if (rowlane) { Lk[ib * LS + c] = lv; Tp[i * LS + c] = -tmc; }
        const double lrc = Dr[c] * dinv;        // lane r: L[r][c] of the block (0 for a zero pivot: no update)
#pragma unroll
        for (int c2 = c + 1; c2 < 8; c2++) {
          const double Lc2 = readlane_f64(lrc, c2);
          x[c2] = fma(-tmc, Lc2, x[c2]);
          Dr[c2] = fma(-Dr[c], Lc2, Dr[c2]);
        }
      }
    }
    __syncthreads();
    if (stamps && tid == 0 && kb < 20) stamps[9 + 2 * kb] = clock64();
    const int kn = k0 + 8;
    if (kn >= NP) break;
    const int tjn = kn >> 4, cb0 = kn & 15;
    const unsigned lm = (unsigned)__builtin_amdgcn_readfirstlane((int)live(kb));
    // operands of every live tile first (unconditional loads on clamped rows: they overlap), then the MFMAs
    double la0[TPW], la1[TPW], tb0[TPW], tb1[TPW];
    bool actv[TPW];
#pragma unroll
    for (int u = 0; u < TPW; u++) {
      const bool act = (w + NW * u < C::NTILES) && (ttj[u] >= tjn) && ((lm >> tti[u]) & 1u) && ((lm >> ttj[u]) & 1u);    // wave-uniform
      actv[u] = act;
      const double *la = Lk + (act ? (16 * tti[u] + lc - k0) : 0) * LS + lr, *tb = Tp + (act ? (16 * ttj[u] + lc) : k0) * LS + lr;
      la0[u] = la[0]; la1[u] = la[4]; tb0[u] = tb[0]; tb1[u] = tb[4];
    }
    // (first the first half of the rank-8 update on every live tile, then the second: two MFMAs in a row on ONE accumulator run at
    //  the instruction's dependent latency, ~115 cycles, instead of its 64-cycle issue rate)
#pragma unroll
    for (int u = 0; u < TPW; u++)
      if (actv[u]) acc[u] = __builtin_amdgcn_mfma_f64_16x16x4f64(la0[u], tb0[u], acc[u], 0, 0, 0);
#pragma unroll
    for (int u = 0; u < TPW; u++) {
      if (actv[u]) acc[u] = __builtin_amdgcn_mfma_f64_16x16x4f64(la1[u], tb1[u], acc[u], 0, 0, 0);
      // the next panel's columns are published by their tile column whether or not this panel touched the tile
      if ((w + NW * u < C::NTILES) && ttj[u] == tjn && lc >= cb0 && lc < cb0 + 8)
#pragma unroll
        for (int r = 0; r < 4; r++) P[(16 * tti[u] + lr + 4 * r) * 8 + (lc - cb0)] = acc[u][r];
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
    // L[j][tid] for fixed tid is linear in j (row stride LS inside one column block): step a pointer, load 8 ahead
    const double *col = Lst + C::lst_off((tid < n ? tid : 0) >> 3) + (0 - ((tid < n ? tid : 0) & ~7)) * C::LS + ((tid < n ? tid : 0) & 7);   // &L[0][tid]
    if (wv == b) {
      int j = hiR;
      for (; j - 7 > 64 * b; j -= 8) {
        double lv[8];
#pragma unroll
        for (int q = 0; q < 8; q++) lv[q] = (tid < j - q) ? col[(j - q) * C::LS] : 0.0;
#pragma unroll
        for (int q = 0; q < 8; q++) { const double xj = readlane_f64(x, j - q - 64 * b); x -= lv[q] * xj; }
      }
      for (; j > 64 * b; j--) {
        const double xj = readlane_f64(x, j - 64 * b);
        if (tid < j && tid < n) x -= col[j * C::LS] * xj;
      }
      if (tid < n) xs[tid] = x;
    }
    __syncthreads();
    if (wv < b && tid < n) {
      int j = 64 * b;
      for (; j + 7 <= hiR; j += 8) {
        double lv[8];
#pragma unroll
        for (int q = 0; q < 8; q++) lv[q] = col[(j + q) * C::LS];
#pragma unroll
        for (int q = 0; q < 8; q++) x -= lv[q] * xs[j + q];
      }
      for (; j <= hiR; j++) x -= col[j * C::LS] * xs[j];
    }
  }
  return x;
}


}  // namespace vba
