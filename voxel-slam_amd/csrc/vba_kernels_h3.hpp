// K3, third form: the Hessian pass on OCCUPANCY-COMPACT tiles  <->  LidarFactor::acc_evaluate2  voxel_map.hpp:150-282
//
// The reference skips the empty frames of a voxel (VM:203-205, 255-272: `if (sig_orig[i].N != 0)`); on the bench window 57 % of the
// (voxel, frame) slots are empty and only 18 % of the 6x6 blocks of a voxel's G^T C G are non-zero.  k_hessian2 (vba_kernels_factor.hpp)
// gave every slot a thread and contracted the dense 3 TV x 6 W image of every tile: 10 of 10 upper-triangle MFMA units whatever the
// tile saw.  Here the store is in (popcount descending, mask) order and a TILE is a run of voxels whose masks' UNION has p frames with
//   voxels x p <= 256 slot threads,   rows x row stride within the LDS image,   voxels <= 96        (k_factor_tiles, greedy)
// so that  * phase A runs one thread per slot of the union (70 % of them occupied instead of 43 %, and 1.7 tiles per workgroup
//            instead of 3),
//          * phase B contracts the COMPACT 3 nv x 6 p image: T = ceil(6 p / 16) column tiles, T (T + 1) / 2 units instead of 10
//            (2.4x fewer MFMAs on the bench window), the units and K-chunks dealt to the four waves per tile,
//          * the accumulators are added into a dense (6 W)^2 image in LDS after every tile (the compact layout changes from tile to
//            tile); the epilogue writes that image in the tile layout k_reduce_partials / tl_fetch already read.
// Per-tile data are requested one tile ahead (slot scalars) and two tiles ahead (masks), as k_hessian2 did for its fixed tiles.
#pragma once
#include "vba_kernels_factor.hpp"

namespace vba {

constexpr int H3_IMG = 4896;          // doubles per LDS image (rows x stride): 288 x 17, 144 x 33, 96 x 49, 72 x 65 all fit
constexpr int H3_MAXNV = 96;
__host__ __device__ constexpr int h3_T(int p) { return (6 * p + 15) / 16; }
__host__ __device__ constexpr bool h3_fits(int nv, int p) {
  return nv * p <= 256 && nv <= H3_MAXNV && (4 * ((3 * nv + 3) / 4)) * (16 * h3_T(p) + 1) <= H3_IMG;
}

// Tile table of the store [0, n): classes = runs of equal popcount (the extraction's order; a store pushed by the host in any order is
// one class); inside a class one wave walks the masks 64 at a time with a running union and cuts a tile where the budget breaks
// (the budget is monotone in both the union and the count, so the first violating lane is the cut).  tiles[0] = count; entry t =
// (first voxel, voxels, union of the masks, 0) at tiles[4 + 4 t].
__global__ __launch_bounds__(1024) void k_factor_tiles(FactorView f, int n) {
  __shared__ int bnd[18];
  __shared__ int nbnd;
  __shared__ int cnt[16], base[17];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  if (tid == 0) nbnd = 0;
  __syncthreads();
  for (int v = tid; v < n; v += 1024) {
    const bool b = v == 0 || __popc(f.occ[v]) != __popc(f.occ[v - 1]);
    if (b) { const int i = atomicAdd(&nbnd, 1); if (i < 17) bnd[i] = v; }
  }
  __syncthreads();
  if (tid == 0) {
    int nb = nbnd;
    if (nb > 16 || n == 0) { nb = n > 0 ? 1 : 0; bnd[0] = 0; }         // not in popcount order: one class
    else for (int i = 1; i < nb; i++) { const int x = bnd[i]; int j = i - 1; while (j >= 0 && bnd[j] > x) { bnd[j + 1] = bnd[j]; j--; } bnd[j + 1] = x; }
    bnd[nb] = n;
    nbnd = nb;
  }
  __syncthreads();
  const int ncls = nbnd;
  // pass 1 counts the tiles of every class, pass 2 writes them behind the classes before it
  for (int pass = 0; pass < 2; pass++) {
    if (wave < ncls) {
      const int v0 = bnd[wave], v1 = bnd[wave + 1];
      int *out = f.tiles + 4 + 4 * (pass ? base[wave] : 0);
      int ntile = 0, start = v0, nv = 0;
      unsigned int U = 0;
      int i = v0;
      while (i < v1) {
        const unsigned int m = (i + lane < v1) ? f.occ[i + lane] : 0u;
        unsigned int pre = m;                                            // inclusive prefix OR over the lanes
        for (int off = 1; off < 64; off <<= 1) { const unsigned int o = __shfl_up(pre, off, 64); if (lane >= off) pre |= o; }
        const unsigned int Ul = U | pre;
        const bool ok = (i + lane < v1) && h3_fits(nv + lane + 1, __popc(Ul));
        const unsigned long long bad = ~__ballot(ok);
        const int take = bad ? __ffsll((long long)bad) - 1 : 64;         // lanes [0, take) join the current tile
        if (take == 64) { U = __shfl(Ul, 63, 64); nv += 64; i += 64; continue; }
        const unsigned int Ut = take > 0 ? __shfl(Ul, take - 1, 64) : U;   // union of the tile that ends here
        if (i + take >= v1) { nv += take; U = Ut; i = v1; break; }         // the class ends inside this step
        // cut: the tile is [start, i + take)
        nv += take;
        if (pass && lane == 0) { out[4 * ntile] = start; out[4 * ntile + 1] = nv; out[4 * ntile + 2] = (int)Ut; out[4 * ntile + 3] = 0; }
        ntile++;
        start = i + take; i = start; nv = 0; U = 0;
      }
      if (nv > 0) { if (pass && lane == 0) { out[4 * ntile] = start; out[4 * ntile + 1] = nv; out[4 * ntile + 2] = (int)U; out[4 * ntile + 3] = 0; } ntile++; }
      if (!pass && lane == 0) cnt[wave] = ntile;
    }
    __syncthreads();
    if (!pass && tid == 0) { int o = 0; for (int c = 0; c < ncls; c++) { base[c] = o; o += cnt[c]; } base[ncls] = o; f.tiles[0] = o; f.tiles[1] = ncls; }
    __syncthreads();
  }
}

template <int W>
struct HessCfg3 {
  static constexpr int NT = 256;
  static constexpr int NC = 6 * W;
  static constexpr int HS = 64;                                      // row stride of the dense H image
  // dynamic LDS (doubles): G | GA | Himg | staging [4 waves][5 units][256] | Eacc [27][W] | sp [W][12]   then ints: masks [2][96] | cmap [64]
  static constexpr size_t LDS_DOUBLES = 2 * (size_t)H3_IMG + 64 * 64 + 4 * 5 * 256 + 27 * W + 12 * W + 8;
  static constexpr size_t LDS_BYTES = LDS_DOUBLES * 8 + (2 * H3_MAXNV + 64 + 16) * 4;
  using C2 = HessCfg2<W>;                                            // the output (tile) layout is k_hessian2's
};

// everything a slot thread keeps about the tile it holds loads for
struct H3Slot { int vl, r, fi; bool on; };

__device__ __forceinline__ int h3_select_bit(unsigned int U, int r) {      // position of the r-th set bit
  for (int k = 0; k < r; k++) U &= U - 1;
  return __ffs((int)U) - 1;
}

// unit u of a T-column-tile image = the u-th pair (ta <= tb) in row-major order of the upper triangle
__device__ __forceinline__ void h3_unit(int T, int u, int &ta, int &tb) {
  int a = 0;
  while (u >= T - a) { u -= T - a; a++; }
  ta = a; tb = a + u;
}

// MFMA phase of one wave: UPW units starting at unit u0, K-steps [k0, k1) of the compact image (operands of step k + 1 are requested
// before the MFMAs of step k are issued); the accumulators go to the wave's staging area P[t][r][lane] (the K-chunks of a unit belong to
// different waves: they are combined after the barrier, in chunk order)
// (the operand pointers are LDS-typed: as generic pointers — a loop-carried array of them — the compiler emitted FLAT loads, whose
//  s_waitcnt vmcnt(0) also waits for the next tile's global loads in flight: the prefetch was dead and phase B took 3.5 us more)
typedef __attribute__((address_space(3))) double lds_f64;
template <int UPW>
__device__ __forceinline__ void h3_phase_b(const double *G_, const double *GA_, int GS, int T, int u0, int k0, int k1, int lane, double *P_) {
  const lds_f64 *G = (const lds_f64 *)G_, *GA = (const lds_f64 *)GA_;
  lds_f64 *P = (lds_f64 *)P_;
  const int kr = lane >> 4, cl = lane & 15;
  int oa[UPW], ob[UPW];
  v4f64 acc[UPW];
#pragma unroll
  for (int t = 0; t < UPW; t++) {
    int ta, tb;
    h3_unit(T, u0 + t, ta, tb);
    oa[t] = (4 * k0 + kr) * GS + 16 * ta + cl; ob[t] = (4 * k0 + kr) * GS + 16 * tb + cl;
    acc[t] = (v4f64){0.0, 0.0, 0.0, 0.0};
  }
  const int step = 4 * GS;
  double ar[UPW], br[UPW];
  if (k0 < k1) {
#pragma unroll
    for (int t = 0; t < UPW; t++) { ar[t] = GA[oa[t]]; br[t] = G[ob[t]]; }
  }
  int off = 0;
  for (int ks = k0; ks < k1; ks++) {
    double av[UPW], bv[UPW];
#pragma unroll
    for (int t = 0; t < UPW; t++) { av[t] = ar[t]; bv[t] = br[t]; }
    off += step;
    if (ks + 1 < k1) {
#pragma unroll
      for (int t = 0; t < UPW; t++) { ar[t] = GA[oa[t] + off]; br[t] = G[ob[t] + off]; }
    }
#pragma unroll
    for (int t = 0; t < UPW; t++) acc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(av[t], bv[t], acc[t], 0, 0, 0);
  }
#pragma unroll
  for (int t = 0; t < UPW; t++)
#pragma unroll
    for (int r = 0; r < 4; r++) P[(t * 4 + r) * 64 + lane] = acc[t][r];
}

template <int W>
__global__ __launch_bounds__(256, 1) void k_hessian3(FactorView f, const double *__restrict__ poses, int nvox, double *__restrict__ partial, const int *__restrict__ gate,
                                                  LmDev *lm, const double *__restrict__ k4_partial, int k4_nb, int nwg, LiJob li, long long *__restrict__ stamps) {
  using C = HessCfg3<W>;
  using C2 = HessCfg2<W>;
  extern __shared__ __attribute__((aligned(16))) double lds[];
  if (li.dev && blockIdx.x == 0) {
    li_imu_body(li.lm, li.dev, li.imu, li.himu, li.gimu, lds);
    return;
  }
  const int bid = (int)blockIdx.x - (li.dev ? 1 : 0);
  // (values that are the same for a whole wave are made so for the compiler too — v_readfirstlane — or it treats the tile loop, the
  //  K loop of the MFMA phase and their bounds as divergent: the first build moved all 40 accumulator registers between the vector and
  //  the accumulator file around EVERY MFMA and ran the pass in 34 us)
  const int tid = threadIdx.x, lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  // diagnostic stamps (stamps == nullptr in production): per workgroup [start, prologue, then per tile A, next-tile requests, MFMA loop (wave 0), barrier, combine, E ..., end at 15]
  int stamp_i = 0;
#define H3_STAMP() do { if (stamps && tid == 0 && stamp_i < 15) stamps[(size_t)bid * 16 + stamp_i] = wall_clock64(); stamp_i++; } while (0)
#define H3_STAMP_END() do { if (stamps && tid == 0) stamps[(size_t)bid * 16 + 15] = wall_clock64(); } while (0)
  H3_STAMP();
  double *G = lds, *GA = G + H3_IMG, *Himg = GA + H3_IMG, *Pst = Himg + 64 * 64, *Eacc = Pst + 4 * 5 * 256, *sp = Eacc + 27 * W;
  int *msk = (int *)(sp + 12 * W + 8), *cmap = msk + 2 * H3_MAXNV;
  __shared__ int lm_dec[2];
  __shared__ double rsum[4];
  // ---- requests that depend on nothing: the table entries (first voxel, voxels, union) of this workgroup's first tiles
  const int ntiles = __builtin_amdgcn_readfirstlane(f.tiles[0]);
  const int4 *tab = reinterpret_cast<const int4 *>(f.tiles + 4);
  int t_cur = bid, t_nx = bid + nwg, t_nx2 = bid + 2 * nwg;
  int4 e_cur = make_int4(0, 0, 0, 0), e_nx = e_cur, e_nx2 = e_cur;
  if (t_cur < ntiles) e_cur = tab[t_cur];
  if (t_nx < ntiles) e_nx = tab[t_nx];
  if (t_nx2 < ntiles) e_nx2 = tab[t_nx2];
  int gate_v = (gate && !lm) ? *gate : 1;
  double lm_r2 = 0.0;
  if (lm) {
    if (tid < 64) {
      lm_r2 = lm_sum_partials(k4_partial, k4_nb, tid);
      if (tid == 0) {
        const double r1 = lm_r1(lm);
        const int stop = lm_prev_stop(lm);
        const bool accept = (r1 - lm_r2) > 0, nstop = fabs((r1 - lm_r2) / r1) < 1e-6;
        lm_dec[0] = (!stop && accept && !nstop) ? 1 : 0;
        lm_dec[1] = stop;
      }
    }
    poses = lm_xt(lm);
  }
  for (int t = tid; t < W * 12; t += C::NT) sp[t] = poses[t];
  for (int t = tid; t < 64 * 64; t += C::NT) Himg[t] = 0.0;
  for (int t = tid; t < 27 * W; t += C::NT) Eacc[t] = 0.0;
  // ---- slot of this thread in a tile with nv voxels and union U: thread (vl, r) <-> frame = the r-th set bit of U
  auto slot_of = [&](int nv, unsigned int U, H3Slot &s) {
    const int p = __popc(U);
    int r = 0, x = tid;
    while (x >= nv && r < p) { x -= nv; r++; }                          // (tid / nv without a division: r <= 10)
    s.vl = x; s.r = r; s.fi = 0; s.on = false;
    if (r < p) s.fi = h3_select_bit(U, r); else s.vl = -1;              // no slot for this thread in this tile
  };
  // the first tile: its union comes with the table entry, so the slot scalars are requested UNCONDITIONALLY together with the masks
  // (one memory trip; an empty slot's scalars are zeros in the store) — later tiles know their masks a tile ahead and skip empty slots
  SlotLoad nx; nx.valid = false;
  H3Slot sl_cur; sl_cur.vl = -1; sl_cur.r = 0; sl_cur.fi = 0; sl_cur.on = false;
  unsigned int mk_nx = 0;
  if (t_cur < ntiles) {
    slot_of(e_cur.y, (unsigned int)e_cur.z, sl_cur);
    if (sl_cur.vl >= 0) slot_load<W>(f, e_cur.x + sl_cur.vl, sl_cur.fi, nvox, ~0u, nx);
  }
  if (tid < H3_MAXNV && t_nx < ntiles && tid < e_nx.y) mk_nx = f.occ[e_nx.x + tid];
  asm volatile("" : "+v"(nx.n));
  if (gate_v == 0) return;                    // uniform
  __syncthreads();
  if (lm) {
    const int run = lm_dec[0], was_stopped = lm_dec[1];
    if (bid == 0 && tid < 64 && !was_stopped) lm_update_apply(lm, lm_r2, W);
    if (!run) return;                         // uniform: rejected step or converged -> no Hessian pass (VM:443)
  }
  double rres = 0.0;
  H3_STAMP();
  int buf = 0;
  while (__builtin_amdgcn_readfirstlane(t_cur) < ntiles) {
    const int v0 = __builtin_amdgcn_readfirstlane(e_cur.x), nv = __builtin_amdgcn_readfirstlane(e_cur.y);
    const unsigned int U = (unsigned int)__builtin_amdgcn_readfirstlane(e_cur.z);
    const int p = __popc(U), T = h3_T(p), GS = 16 * T + 1;
    const int ksteps = (3 * nv + 3) / 4, NK = 4 * ksteps;
    const H3Slot sl = sl_cur;
    // ---------------- phase A
    double Err[6] = {0, 0, 0, 0, 0, 0}, Ert[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0}, Ett[6] = {0, 0, 0, 0, 0, 0}, gj[6] = {0, 0, 0, 0, 0, 0};
    // zero what the slot threads do not write: the pad columns [6 p, 16 T] and the pad rows [3 nv, NK)
    for (int row = tid; row < NK; row += C::NT)
      for (int col = 6 * p; col < GS; col++) { G[(size_t)row * GS + col] = 0.0; GA[(size_t)row * GS + col] = 0.0; }
    for (int row = 3 * nv; row < NK; row++)
      for (int col = tid; col < 6 * p; col += C::NT) { G[(size_t)row * GS + col] = 0.0; GA[(size_t)row * GS + col] = 0.0; }
    if (sl.vl >= 0) {
      const SlotLoad q = nx;
      double g1[6] = {0, 0, 0, 0, 0, 0}, g2[6] = {0, 0, 0, 0, 0, 0}, hh[6] = {0, 0, 0, 0, 0, 0};
      double ck1 = 0.0, ck2 = 0.0, ck3 = 0.0;
      slot_terms(q, sp + 12 * sl.fi, sl.r == 0, g1, g2, hh, ck1, ck2, ck3, Err, Ert, Ett, gj, rres);
      double *g = G + (size_t)(3 * sl.vl) * GS + 6 * sl.r, *ga = GA + (size_t)(3 * sl.vl) * GS + 6 * sl.r;
#pragma unroll
      for (int d = 0; d < 6; d++) {
        g[d] = g1[d]; g[GS + d] = g2[d]; g[2 * GS + d] = hh[d];
        ga[d] = g1[d] * ck1; ga[GS + d] = g2[d] * ck2; ga[2 * GS + d] = hh[d] * ck3;
      }
    }
    if (tid < 64) { const int cc = tid; cmap[tid] = cc < 6 * p ? 6 * h3_select_bit(U, cc / 6) + cc % 6 : -1; }   // compact -> dense column
    // the next tile: its masks go to LDS, its slots are derived after the barrier, then its loads fly under phase B
    if (tid < H3_MAXNV) msk[buf * H3_MAXNV + tid] = (int)mk_nx;
    __syncthreads();
    H3_STAMP();
    H3Slot sl_nx; sl_nx.vl = -1; sl_nx.r = 0; sl_nx.fi = 0; sl_nx.on = false;
    nx.valid = false;
    if (t_nx < ntiles) {
      slot_of(e_nx.y, (unsigned int)e_nx.z, sl_nx);
      if (sl_nx.vl >= 0) slot_load<W>(f, e_nx.x + sl_nx.vl, sl_nx.fi, nvox, (unsigned int)msk[buf * H3_MAXNV + sl_nx.vl], nx);
    }
    // two tiles ahead: table entry and masks
    const int t_nx3 = t_nx2 + nwg;
    int4 e_nx3 = make_int4(0, 0, 0, 0);
    if (t_nx3 < ntiles) e_nx3 = tab[t_nx3];
    unsigned int mk_nx2 = 0;
    if (tid < H3_MAXNV && t_nx2 < ntiles && tid < e_nx2.y) mk_nx2 = f.occ[e_nx2.x + tid];
    H3_STAMP();
    // ---------------- phase B: units (ta <= tb < T) x K-chunks dealt to the waves
    // T <= 2: 1 or 3 units, K in quarters (every wave takes all units of its quarter); T = 3, 4: 6 or 10 units, K in halves
    // (waves 0, 1 take the first half of the units, waves 2, 3 the second)
    const int NU = T * (T + 1) / 2;
    const int KS = T <= 2 ? 4 : 2, kc = T <= 2 ? wv : (wv & 1);
    const int per = T <= 2 ? NU : NU / 2, u0 = T <= 2 ? 0 : (wv >> 1) * per;
    {
      const int k0 = (ksteps * kc) / KS, k1 = (ksteps * (kc + 1)) / KS;
      double *P = Pst + wv * 5 * 256;
      if (per == 1) h3_phase_b<1>(G, GA, GS, T, u0, k0, k1, lane, P);
      else if (per == 3) h3_phase_b<3>(G, GA, GS, T, u0, k0, k1, lane, P);
      else h3_phase_b<5>(G, GA, GS, T, u0, k0, k1, lane, P);
    }
    H3_STAMP();
    __syncthreads();
    H3_STAMP();
    // ---- the tile's accumulators join the dense image: thread e owns element e of every unit; the K-chunks of a unit are summed in
    //      chunk order, the sum goes into the image with an LDS add that nobody waits for (one thread per element: no two adds meet)
    {
      const int l = tid & 63, r = tid >> 6, ru = (l >> 4) + 4 * r, cu = l & 15;
      int mapR[4], mapC[4];
#pragma unroll
      for (int a = 0; a < 4; a++) { mapR[a] = a < T ? cmap[16 * a + ru] : -1; mapC[a] = a < T ? cmap[16 * a + cu] : -1; }
      int u = 0;
#pragma unroll
      for (int ta = 0; ta < 4; ta++)
#pragma unroll
        for (int tb = ta; tb < 4; tb++) {
          if (ta < T && tb < T) {                                  // uniform
            const int h = T <= 2 ? 0 : (u >= per ? 1 : 0), ul = T <= 2 ? u : u - h * per;
            const double *src = Pst + ((T <= 2 ? 0 : 2 * h) * 5 + ul) * 256 + tid;
            double s = src[0] + src[5 * 256];
            if (T <= 2) s = (s + src[10 * 256]) + src[15 * 256];
            const int R = mapR[ta], Cc = mapC[tb];
            if ((ta < tb || ru <= cu) && R >= 0 && Cc >= 0) unsafeAtomicAdd(&Himg[R * C::HS + Cc], s);
            u++;
          }
        }
    }
    H3_STAMP();
    // ---- E / gradient of the tile: slot sums per (term, frame), through the (now free) image area
    {
      double *S = G;                                  // [256 slots][27 terms]: slot-major, 27 is odd (term-major put the 27 readers of a frame on one bank)
      if (sl.vl >= 0) {
        double *s_ = S + tid * 27;
#pragma unroll
        for (int k = 0; k < 6; k++) { s_[k] = Err[k]; s_[15 + k] = Ett[k]; s_[21 + k] = gj[k]; }
#pragma unroll
        for (int k = 0; k < 9; k++) s_[6 + k] = Ert[k];
      }
      __syncthreads();
      for (int t = tid; t < 27 * p; t += C::NT) {
        const int k = t % 27, r = t / 27;
        const double *sp_ = S + (size_t)(r * nv) * 27 + k;
        double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
        int vl = 0;
        for (; vl + 4 <= nv; vl += 4) { s0 += sp_[vl * 27]; s1 += sp_[(vl + 1) * 27]; s2 += sp_[(vl + 2) * 27]; s3 += sp_[(vl + 3) * 27]; }
        for (; vl < nv; vl++) s0 += sp_[vl * 27];
        Eacc[k * W + h3_select_bit(U, r)] += (s0 + s1) + (s2 + s3);
      }
      __syncthreads();
    }
    H3_STAMP();
    // ---- advance
    buf ^= 1;
    t_cur = t_nx; t_nx = t_nx2; t_nx2 = t_nx3;
    e_cur = e_nx; e_nx = e_nx2; e_nx2 = e_nx3;
    mk_nx = mk_nx2;
    sl_cur = sl_nx;
  }
  // ---------------- epilogue: the dense image in the tile layout of k_hessian2 (k_reduce_partials / tl_fetch), E blocks, gradient, residual
  double *out = partial + (size_t)bid * C2::NOUT2;
  {
    const int l = tid & 63, r = tid >> 6, ru = (l >> 4) + 4 * r, cu = l & 15;
    int u = 0;
#pragma unroll
    for (int ta = 0; ta < C2::NT16; ta++)
#pragma unroll
      for (int tb = ta; tb < C2::NT16; tb++) {
        const int row = 16 * ta + ru, col = 16 * tb + cu;
        out[u * 256 + tid] = (row <= col && col < C::NC) ? Himg[row * C::HS + col] : 0.0;
        u++;
      }
  }
  for (int t = tid; t < 27 * W; t += C::NT) {
    const int k = t / W, fr = t % W;
    if (k < 21) out[C2::EB + 21 * fr + k] = Eacc[t];
    else out[C2::GB + 6 * fr + (k - 21)] = Eacc[t];
  }
  rres = wave_sum(rres);
  if (lane == 0) rsum[wv] = rres;
  __syncthreads();
  if (tid == 0) out[C2::RB] = (rsum[0] + rsum[1]) + (rsum[2] + rsum[3]);
  H3_STAMP_END();
#undef H3_STAMP
#undef H3_STAMP_END
}

}  // namespace vba
