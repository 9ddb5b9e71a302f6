// Device-resident voxel hash map + adaptive octree of Voxel-SLAM's local mapping (gfx950 / wave64):
//   K1 insert       <-> cut_voxel / cut_voxel_multi VM:1896-2096, OctoTree::allocate/push VM:1204/1105, Bf_var VM:106,
//                       cut_voxel(fix) VM:2108-2152, allocate_fix / push_fix_novar VM:1239/1168
//   K2 recut        <-> OctoTree::recut VM:1396-1456 (plane_judge 1185, fix_divide 1270, subdivide 1307),
//                       multi_recut VS:1682-1737, tras_opt VM:1605-1638 (writes the SoA factor store directly)
//   K5 marginalise  <-> OctoTree::margi VM:1465-1598, plane_update VM:1344-1388, multi_margi VS:1590-1679
// Design: explicit per-node state in SoA arrays, roots in an open-addressing hash table keyed by the
// packed voxel key, children allocated as blocks of 8 when a leaf is subdivided, raw window points kept in a ring of
// W scan slots with a per-point leaf assignment (the "points of leaf X, frame i" lists of the reference), fixed points
// in an append-only pool with a per-point owner.  The tree is processed level-synchronously (<= max_layer+1 passes).
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstring>
#include <string>
#include <vector>
#include <functional>
#include "../../include/voxelba.h"
#include "vba_kernels_factor.hpp"

namespace vba {

// vba_sort.hip (rocPRIM): stable radix sort of (key, value) pairs on key bits [0, end_bit); tmp == nullptr queries tmp_bytes
hipError_t sort_pairs_u32(void *tmp, size_t &tmp_bytes, const unsigned int *keys_in, unsigned int *keys_out, const int *vals_in, int *vals_out,
                          size_t n, unsigned int end_bit, hipStream_t stream);

// 16-bit bucket of a root voxel key; ranks own contiguous bucket ranges
__host__ __device__ inline uint64_t shard_bucket(int64_t kx, int64_t ky, int64_t kz) {
  uint64_t h = (uint64_t)kx * 0x9E3779B97F4A7C15ull;
  h ^= (uint64_t)ky * 0xC2B2AE3D27D4EB4Full + (h << 6) + (h >> 2);
  h ^= (uint64_t)kz * 0x165667B19E3779F9ull + (h << 6) + (h >> 2);
  h ^= h >> 29; h *= 0xBF58476D1CE4E5B9ull; h ^= h >> 32;
  return h & 0xFFFFull;
}

static constexpr unsigned long long KEY_EMPTY = ~0ull;
static constexpr unsigned long long KEY_TOMB = ~0ull - 1;   // erased root (map pruning): probes walk past it
static constexpr int KEY_BITS = 21, KEY_OFF = 1 << 20;

__host__ __device__ inline unsigned long long pack_key(long long kx, long long ky, long long kz) {
  return ((unsigned long long)(kx + KEY_OFF) << 42) | ((unsigned long long)(ky + KEY_OFF) << 21) | (unsigned long long)(kz + KEY_OFF);
}
__host__ __device__ inline void unpack_key(unsigned long long k, long long &kx, long long &ky, long long &kz) {
  kx = (long long)((k >> 42) & 0x1FFFFF) - KEY_OFF; ky = (long long)((k >> 21) & 0x1FFFFF) - KEY_OFF; kz = (long long)(k & 0x1FFFFF) - KEY_OFF;
}
// The reference's key quirk VM:1907-1918: float narrowing, -1 if negative, truncation toward zero.
__host__ __device__ inline long long key_axis(double pw, double voxel_size) {
  float loc = (float)(pw / voxel_size);
  if (loc < 0) loc -= 1.0f;
  return (long long)loc;
}

// CNT_USED = hash slots that are not EMPTY (live roots + tombstones); CNT_FREE_ROOTS / CNT_FREE_BLOCKS = depth of the two
// free-node stacks that map pruning fills and node creation drains
enum { CNT_NODES = 0, CNT_FIX, CNT_SLIDE, CNT_OVERFLOW, CNT_TOUCH, CNT_ROOTS, CNT_FACTORS, CNT_NEWSLOTS, CNT_LEAVES, CNT_BADKEY, CNT_SNAP,
       CNT_USED, CNT_FREE_ROOTS, CNT_FREE_BLOCKS,
       CNT_FBLK, CNT_TAKE,         // blocks of the fixed-point pool / leaves whose oldest frame moves to the pool in this margi
       CNT_WLB, CNT_CURSOR,        // leaves of the work list with more than 64 points of the scan (their own kernel) / segment allocator of the scan
       CNT_WL, CNT_SPLIT,          // ordered accumulation: leaf segments of the scan being inserted / leaves split by the current recut level
       CNT_DBG0, CNT_DBG1, CNT_DBG2, CNT_DBG3, // diagnostics build only (-DVBA_DIAG)
       CNT_SLIDE_G, CNT_TOUCH_G,   // the two counts the reference's 'fewer voxels than threads' quirks test, summed over the ranks when the map is sharded
       CNT_N };

struct MapParams {
  int W, max_layer, max_points, thread_num;
  double voxel_size, min_eigen_value;
  double plane_thre[4], min_point[4];
  int mp[VBA_MAX_WIN];
  int rank, n_ranks;
};

struct MapView {
  // hash table of roots
  unsigned long long *hkeys; int *hvals; unsigned int hmask;
  // nodes
  int cap;
  unsigned long long *nkey; int *nroot; int *nparent; int *nchild; int *npath; int *nopt; int *nflist /* factor index -> leaf (tras_opt order) */; int *nfl2 /* the same before the occupancy sort */; unsigned int *nfkey; int *fhist /* [EXTRACT_NB_MAX] */; int *nlast; int *nstamp; int *nsplit; int *ntake; int *nclear; int *ndead;
  int *nfree_root, *nfree_blk;   // stacks of recycled node ids: single root nodes / bases of 8-node child blocks (map_prune)
  int *nseg_a, *nseg_b;          // [W][cap]: the points a scan slot gave to a leaf AT INSERTION = perm[slot][nseg_a .. nseg_b) (scan order)
  int *ncnt;                     // [cap] points of the scan being inserted per leaf, then the scatter cursor; zero between inserts
  int *nsl;                      // [cap] leaves split by the current recut level (margi: leaves whose oldest frame joins the fixed points)
  int *nfb_head, *nfb_tail;      // [cap] a leaf's fixed points (point_fix) arrive in BLOCKS of consecutive pool entries; the blocks are chained in arrival order
  signed char *nlayer; signed char *nstate;
  unsigned char *f_exist, *f_sw, *f_plane, *f_touched; int *f_slide;
  float *nql; double *ncenter; double *njour;
  double *nadd, *nfix, *ncov, *neval, *nevec, *nplane, *nlc;
  // scan ring
  int max_pts;
  double *px;   // [W][max_pts][3]  (AoS: the per-leaf kernels gather whole points by index)
  double *pvar; // [W][max_pts][9]
  int *pnode;   // [W][max_pts]
  int *phash;   // [max_pts] temp
  int *newslots;  // [max_pts] temp
  int *perm;    // [W][max_pts] point indices of a slot grouped by insertion leaf, scan order inside a group
  // the slot's points IN THAT ORDER (what sw->points[mord] of the leaves hold in the reference): the passes that walk a leaf's points again
  // (subdivide, the move of the oldest frame to point_fix) stream them instead of chasing perm -> point
  double *sx;   // [3][W][max_pts]
  double *svar; // [9][W][max_pts]
  int *pleaf;   // [W][max_pts] the leaf that holds the point NOW (-1: none / released)
  unsigned int *skey_a, *skey_b;   // [max_pts] sort keys (leaf id) in / out
  int *sval_a;  // [max_pts] sort values in (the point index)
  int *wl;      // [max_pts] leaves that received points of the scan being inserted
  int *wlb;     // [max_pts] those with more than 64 points
  int4 *wl4;    // [max_pts] work list entries (leaf, segment start, points, -) for the per-leaf kernels: one 16-byte load
  // fixed-point pool
  int cap_fix;
  double *fx;   // [cap_fix][3]  (AoS, as the window points: a leaf's block is one contiguous run)
  double *fvar; // [cap_fix][9]
  int *fnode;
  int *fb_base, *fb_len, *fb_next;   // [cap_fix] block table of the pool (a block has >= 1 point)
  int *sval_b;  // [max_pts] sort values out where the destination is not a slot's perm (fixed-point insertion)
  int *cnt;     // counters [CNT_N]
  double *poses;  // [W][12]
};

// ------------------------------------------------------------------------------------------------ helpers
// ------------------------------------------------------------------------------------------------ order-preserving accumulation
// The reference pushes a leaf's points one by one in SCAN ORDER (cut_voxel VM:1899-1948 / cut_voxel_multi VM:2061-2095 ->
// allocate -> push VM:1134-1140): pcrs_local, pcr_add and cov_add are chains  ((s + t_1) + t_2) + ...  of separately rounded
// IEEE operations (the reference targets baseline x86-64: no FMA contraction).  Floating-point addition is not associative, so
// the only way to reproduce those sums bit for bit is to run the same chains: the scan's points are grouped by leaf with a STABLE
// sort (scan order survives inside a group), one wave owns a leaf, lane j prepares the terms of the group's j-th point, lane k
// then adds term k of the points IN ORDER to scalar k of the leaf.  No f64 atomics: the sums do not depend on the run either.
// Every expression below is written in the reference's operation order with contraction off.
__device__ __forceinline__ void world_point(const double *R, double bx, double by, double bz, double &x, double &y, double &z) {
#pragma clang fp contract(off)
  x = ((R[0] * bx + R[1] * by) + R[2] * bz) + R[9];      // s = 0; s += R(r,k) p(k); ... + t(r)
  y = ((R[3] * bx + R[4] * by) + R[5] * bz) + R[10];
  z = ((R[6] * bx + R[7] * by) + R[8] * bz) + R[11];
}

// terms of one point: [0..8] body cluster (P upper triangle, v), [9..17] world cluster, [18..62] Bf_var upper triangle (VM:106-121);
// the two point counts are integers and are added outside the chains
template <bool HAS_VAR>
struct OrdCfg { static constexpr int NT = HAS_VAR ? 63 : 18; static constexpr int TS = HAS_VAR ? 63 : 19; };   // TS odd: conflict-free rows

template <bool HAS_VAR>
__device__ __forceinline__ void ord_terms(double *t, double bx, double by, double bz, double x, double y, double z, const double *var) {
#pragma clang fp contract(off)
  t[0] = bx * bx; t[1] = bx * by; t[2] = bx * bz; t[3] = by * by; t[4] = by * bz; t[5] = bz * bz; t[6] = bx; t[7] = by; t[8] = bz;
  t[9] = x * x; t[10] = x * y; t[11] = x * z; t[12] = y * y; t[13] = y * z; t[14] = z * z; t[15] = x; t[16] = y; t[17] = z;
  if (HAS_VAR) {
    const double Bi[6][3] = {{2 * x, 0, 0}, {y, x, 0}, {z, 0, x}, {0, 2 * y, 0}, {0, z, y}, {0, 0, 2 * z}};
    double Bu[6][3];
#pragma unroll
    for (int r = 0; r < 6; r++)
#pragma unroll
      for (int c = 0; c < 3; c++) Bu[r][c] = (Bi[r][0] * var[0 * 3 + c] + Bi[r][1] * var[1 * 3 + c]) + Bi[r][2] * var[2 * 3 + c];
    int idx = 18;
#pragma unroll
    for (int r = 0; r < 9; r++)
#pragma unroll
      for (int c = r; c < 9; c++) {
        double val;
        if (r < 6 && c < 6) val = (Bu[r][0] * Bi[c][0] + Bu[r][1] * Bi[c][1]) + Bu[r][2] * Bi[c][2];
        else if (r < 6) val = Bu[r][c - 6];
        else val = var[(r - 6) * 3 + (c - 6)];
        t[idx++] = val;
      }
  }
}
// The sums of a node are stored NODE-MAJOR (nlc[id][slot][10], nadd / nfix[id][10], ncov[id][45]): the accumulation kernels own a
// leaf per wave and touch all its scalars (63 consecutive-ish doubles instead of 63 cache lines), the per-node kernels read whole
// records anyway.
__device__ __forceinline__ double &nlc_at(const MapView &m, size_t W, int k, int slot, size_t id) { return m.nlc[(id * W + slot) * 10 + k]; }
__device__ __forceinline__ double &nadd_at(const MapView &m, int k, size_t id) { return m.nadd[id * 10 + k]; }
__device__ __forceinline__ double &nfix_at(const MapView &m, int k, size_t id) { return m.nfix[id * 10 + k]; }
__device__ __forceinline__ double &ncov_at(const MapView &m, int k, size_t id) { return m.ncov[id * 45 + k]; }
__device__ __forceinline__ double &neval_at(const MapView &m, int k, size_t id) { return m.neval[id * 3 + k]; }   // eigen-pairs of the node's plane fit: node-major too
__device__ __forceinline__ double &nevec_at(const MapView &m, int k, size_t id) { return m.nevec[id * 9 + k]; }   // (the extraction reads them per factor)
// where scalar k of a leaf lives: [0..8] nlc (slot cluster), [9..17] nadd (pcr_add), [18..62] ncov (cov_add)
__device__ __forceinline__ double *ord_target(const MapView &m, int W, int slot, int k, int leaf) {
  return k < 9 ? &nlc_at(m, (size_t)W, k, slot, leaf) : k < 18 ? &nadd_at(m, k - 9, leaf) : &ncov_at(m, k - 18, leaf);
}

__device__ __forceinline__ int octant_of(const MapView &m, int node, double x, double y, double z) {
  const size_t cp = (size_t)m.cap;
  const int ox = x > m.ncenter[node] ? 1 : 0, oy = y > m.ncenter[cp + node] ? 1 : 0, oz = z > m.ncenter[2 * cp + node] ? 1 : 0;
  return 4 * ox + 2 * oy + oz;   // VM:1214-1219
}

__device__ __forceinline__ void init_node(const MapView &m, int W, int id, unsigned long long key, int root, int parent, int layer, int path,
                                          double cx, double cy, double cz, float ql) {
  const size_t cp = (size_t)m.cap;
  // (nseg_a / nseg_b of a fresh id are zero: node storage is zero-filled at allocation and k_prune_zero clears what it recycles)
  m.nfb_head[id] = -1; m.nfb_tail[id] = -1;
  m.nkey[id] = key; m.nroot[id] = root; m.nparent[id] = parent; m.nchild[id] = -1; m.npath[id] = path; m.nopt[id] = -1; m.nlast[id] = 0;
  m.nstamp[id] = 0; m.nsplit[id] = 0; m.ntake[id] = 0; m.nclear[id] = 0; m.ndead[id] = 0;
  m.nlayer[id] = (signed char)layer; m.nstate[id] = 0;
  m.f_exist[id] = 0; m.f_sw[id] = 0; m.f_plane[id] = 0; m.f_touched[id] = 0; m.f_slide[id] = 0;
  m.nql[id] = ql; m.ncenter[id] = cx; m.ncenter[cp + id] = cy; m.ncenter[2 * cp + id] = cz; m.njour[id] = 0.0;
}

// Node storage is recycled through two stacks that map pruning fills (k_prune_*): single nodes (roots) and 8-node child blocks.
// Only pops run concurrently (node creation), pushes only inside the prune kernels, so a counter decrement is a safe pop; a pop
// that finds the stack empty gives its decrement back and takes fresh storage from the end of the node arrays.
__device__ __forceinline__ int alloc_nodes(const MapView &m, int which_cnt, const int *stack, int count) {
  if (m.cnt[which_cnt] > 0) {
    const int k = atomicSub(&m.cnt[which_cnt], 1) - 1;
    if (k >= 0) return stack[k];
    atomicAdd(&m.cnt[which_cnt], 1);
  }
  return atomicAdd(&m.cnt[CNT_NODES], count);
}

// The reference tests "#voxels < thread_num" on the whole map (VM:2044, VS:1616, VS:1693).  A sharded rank holds only its bucket
// range, so those tests read the counts summed over the ranks (refreshed by a tiny all-reduce before the kernels that test them).
__device__ __forceinline__ int slide_count(const MapView &m, const MapParams &P) { return P.n_ranks > 1 ? m.cnt[CNT_SLIDE_G] : m.cnt[CNT_SLIDE]; }
__device__ __forceinline__ int touch_count(const MapView &m, const MapParams &P) { return P.n_ranks > 1 ? m.cnt[CNT_TOUCH_G] : m.cnt[CNT_TOUCH]; }
__global__ void k_cnt_to_f64(const int *cnt, int which, double *out) { out[0] = (double)cnt[which]; }
__global__ void k_f64_to_cnt(const double *in, int *cnt, int which) { cnt[which] = (int)(in[0] + 0.5); }

// ------------------------------------------------------------------------------------------------ K1: insert
// Phase 1: world transform, key, find-or-claim the hash slot of the root voxel.
__global__ __launch_bounds__(256) void k_ins_keys(MapView m, MapParams P, int slot, int n, int world_given, int stamp) {
  const int p = blockIdx.x * blockDim.x + threadIdx.x;
  if (!world_given) {   // the slot's previous occupant is gone: its insertion segments are cleared here (k_ins_scan writes the new ones)
    int *sa = m.nseg_a + (size_t)slot * m.cap, *sb = m.nseg_b + (size_t)slot * m.cap;
    const int nn = m.cnt[CNT_NODES] < m.cap ? m.cnt[CNT_NODES] : m.cap;   // (ids that were never handed out still hold the zeros of their allocation)
    for (int i = p; i < nn; i += gridDim.x * blockDim.x) { sa[i] = 0; sb[i] = 0; }
    int *pl = m.pleaf + (size_t)slot * m.max_pts;
    for (int i = p; i < m.max_pts; i += gridDim.x * blockDim.x) pl[i] = -1;
  }
  const size_t mpz = (size_t)m.max_pts, W = (size_t)P.W;
  int hslot = -1, root = -1, claimed = -1;                   // claimed: the hash slot this thread took for a new root (bit 30: a reused tombstone)
  if (p < n) {
    double x, y, z;
    if (world_given) {  // fixed points arrive in world coordinates, staged in the pool tail (see map_cut_voxel_fix)
      { const double *fp = m.fx + (size_t)(slot + p) * 3; x = fp[0]; y = fp[1]; z = fp[2]; }
    } else {
      const double *pp = m.px + ((size_t)slot * mpz + p) * 3;
      const double bx = pp[0], by = pp[1], bz = pp[2];
      world_point(m.poses, bx, by, bz, x, y, z);   // the scan pose is staged at poses[0..12)
    }
    const long long kx = key_axis(x, P.voxel_size), ky = key_axis(y, P.voxel_size), kz = key_axis(z, P.voxel_size);
    bool ok = true;
    if (kx < -KEY_OFF || kx >= KEY_OFF || ky < -KEY_OFF || ky >= KEY_OFF || kz < -KEY_OFF || kz >= KEY_OFF) { atomicAdd(&m.cnt[CNT_BADKEY], 1); ok = false; }
    if (ok && P.n_ranks > 1 && (int)((shard_bucket(kx, ky, kz) * (uint64_t)P.n_ranks) >> 16) != P.rank) ok = false;  // not this rank's bucket range
    if (ok) {
      const unsigned long long key = pack_key(kx, ky, kz);
      const unsigned int h0 = (unsigned int)((key * 0x9E3779B97F4A7C15ull) >> 32) & m.hmask;
      // Find the key, else claim a slot for it: the FIRST tombstone of the probe sequence (slots of pruned roots are reused, so
      // the table cannot silt up with them over a long session) or, when the sequence holds none, its first EMPTY slot.
      // A claim lost to another key restarts the scan; the scan is bounded by the table size, the restarts by a small count.
      unsigned int h = h0;
      bool found = false;
      for (int attempt = 0; attempt < 64 && !found; attempt++) {
        h = h0;
        int tomb = -1;
        bool restart = false;
        for (unsigned int probe = 0; probe <= m.hmask; probe++) {
          const unsigned long long cur = m.hkeys[h];
          if (cur == key) { found = true; break; }
          if (cur == KEY_TOMB && tomb < 0) tomb = (int)h;
          if (cur == KEY_EMPTY) {
            if (tomb >= 0) {
              const unsigned long long prev = atomicCAS(&m.hkeys[tomb], KEY_TOMB, key);
              if (prev == KEY_TOMB) { claimed = tomb | (int)0x40000000; h = (unsigned int)tomb; found = true; break; }
              if (prev == key) { h = (unsigned int)tomb; found = true; break; }
              restart = true; break;                        // another key took the tombstone: scan again
            }
            const unsigned long long prev = atomicCAS(&m.hkeys[h], KEY_EMPTY, key);
            if (prev == KEY_EMPTY) { claimed = (int)h; found = true; break; }
            if (prev == key) { found = true; break; }
            // another key took this slot: it is an ordinary occupied slot now, keep probing
          }
          h = (h + 1) & m.hmask;
        }
        if (!found && !restart) break;                      // a full sweep without the key, a tombstone or an EMPTY slot
      }
      if (!found) atomicMax(&m.cnt[CNT_OVERFLOW], 4);                   // table full: never fall through to an unrelated slot
      else {
        hslot = (int)h;
        // hvals is only written by the next kernel, so a valid id here means "this root existed before the call"
        if (!world_given) root = m.hvals[h];
      }
    }
    m.phash[p] = hslot;
  }
  {  // the new slots join the list of k_ins_newroots: one returning atomic per wave (thousands of them on one address serialise in L2)
    const unsigned long long mc = __ballot(claimed >= 0);
    if (mc) {
      const int ln = threadIdx.x & 63, lead = __ffsll((long long)mc) - 1;
      int b0 = 0;
      if (ln == lead) b0 = atomicAdd(&m.cnt[CNT_NEWSLOTS], __popcll(mc));
      b0 = __shfl(b0, lead, 64);
      if (claimed >= 0) m.newslots[b0 + __popcll(mc & ((1ull << ln) - 1ull))] = claimed;
    }
  }
  // Roots that existed before this call are marked here (isexist, sliding-map membership, per-scan touch count:
  // VM:1997-2001); roots created by this call are marked by k_ins_newroots.  Consecutive points of a scan mostly share
  // their root: only the first lane of each run issues the (contended) atomics.
  // The two counters are bumped once per workgroup: thousands of single increments on one address serialise in L2.
  __shared__ int won[2][4];
  const int prev_root = __shfl_up(root, 1, 64);
  bool w_slide = false, w_touch = false;
  if (root >= 0 && ((threadIdx.x & 63) == 0 || prev_root != root)) {
    m.f_exist[root] = 1;
    w_slide = m.f_slide[root] == 0 && atomicExch(&m.f_slide[root], 1) == 0;
    w_touch = m.nstamp[root] != stamp && atomicExch(&m.nstamp[root], stamp) != stamp;
  }
  const unsigned long long ms = __ballot(w_slide), mt = __ballot(w_touch);
  if ((threadIdx.x & 63) == 0) { won[0][threadIdx.x >> 6] = __popcll(ms); won[1][threadIdx.x >> 6] = __popcll(mt); }
  __syncthreads();
  if (threadIdx.x == 0) {
    const int ns = won[0][0] + won[0][1] + won[0][2] + won[0][3], nt = won[1][0] + won[1][1] + won[1][2] + won[1][3];
    if (ns) atomicAdd(&m.cnt[CNT_SLIDE], ns);
    if (nt) atomicAdd(&m.cnt[CNT_TOUCH], nt);
  }
}

// Phase 2: one thread per newly claimed hash slot creates the root node (VM:1935-1946).
__global__ void k_ins_newroots(MapView m, MapParams P, int is_fix, double jour, int stamp) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= m.cnt[CNT_NEWSLOTS]) return;
  const int hs = m.newslots[i];
  const int h = hs & 0x3FFFFFFF;
  if (!(hs & 0x40000000)) atomicAdd(&m.cnt[CNT_USED], 1);     // a formerly EMPTY slot (a reused tombstone was counted when first claimed)
  const int id = alloc_nodes(m, CNT_FREE_ROOTS, m.nfree_root, 1);
  if (id >= m.cap) { atomicMax(&m.cnt[CNT_OVERFLOW], 1); return; }
  long long kx, ky, kz;
  const unsigned long long key = m.hkeys[h];
  unpack_key(key, kx, ky, kz);
  init_node(m, P.W, id, key, id, -1, 0, 0, (0.5 + kx) * P.voxel_size, (0.5 + ky) * P.voxel_size, (0.5 + kz) * P.voxel_size, (float)(P.voxel_size / 4.0));
  if (is_fix) m.njour[id] = jour;   // VM:2147
  else {                            // VM:2016-2017: a root created by a window scan enters the sliding map
    m.f_exist[id] = 1; m.f_slide[id] = 1; m.nstamp[id] = stamp;
    atomicAdd(&m.cnt[CNT_SLIDE], 1); atomicAdd(&m.cnt[CNT_TOUCH], 1);
  }
  m.hvals[h] = id;
  atomicAdd(&m.cnt[CNT_ROOTS], 1);
}

// Phase 3: descend to the leaf (OctoTree::allocate VM:1204-1237) and count the leaf's points.  Grouping by leaf needs no global
// sort: count -> segments of the touched leaves (in any order) -> scatter (arrival order inside a segment) -> the wave that owns
// the leaf puts its segment into scan order.
__global__ __launch_bounds__(256) void k_ins_leaf(MapView m, MapParams P, int slot, int n, int multi) {
  const int p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= n) return;
  const size_t mpz = (size_t)m.max_pts;
  const bool dropped = multi && touch_count(m, P) < P.thread_num;   // VM:2044-2045: the scan is dropped
  const int h = m.phash[p];
  int node = (!dropped && h >= 0) ? m.hvals[h] : -1;
  if (node >= 0) {
    const double *pp = m.px + ((size_t)slot * mpz + p) * 3;
    const double bx = pp[0], by = pp[1], bz = pp[2];
    double x, y, z;
    world_point(m.poses, bx, by, bz, x, y, z);
    while (m.nstate[node] == 1) node = m.nchild[node] + octant_of(m, node, x, y, z);
    const int arrival = atomicAdd(&m.ncnt[node], 1);      // the point's place in its leaf's segment (arrival order)
    if (arrival == 0) { m.f_sw[node] = 1; m.f_exist[node] = 1; m.f_touched[node] = 1; }
    m.phash[p] = arrival;
  }
  m.pnode[(size_t)slot * mpz + p] = node;
}

// Phase 4: one thread per NODE; every node that counted points gets its segment [nseg_a, nseg_b) of perm[slot] and an entry of the
// work list.  The ORDER of the segments carries no meaning, so no global scan is needed: a workgroup scans its 256 counts and
// reserves their sum (and its run of list entries) with one atomic each.  The list comes out in runs of ascending node index:
// neighbouring waves of the accumulation kernel then work on neighbouring leaves, whose sums share the cache lines of the
// component-major leaf arrays.  ncnt returns to zero; leaves with more than 64 points also go onto the list of the
// workgroup-per-leaf kernel.
__global__ __launch_bounds__(256) void k_ins_scan(MapView m, int slot) {
  __shared__ int wsum[4], wnum[4];
  __shared__ int sbase, lbase;
  const int NN = m.cnt[CNT_NODES];
  if ((int)(blockIdx.x * blockDim.x) >= NN) return;          // (the grid covers an upper bound of the node count)
  const int leaf = blockIdx.x * blockDim.x + threadIdx.x;
  const size_t cp = (size_t)m.cap;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int c = leaf < NN ? m.ncnt[leaf] : 0;
  int incl = c;
  for (int off = 1; off < 64; off <<= 1) { const int v = __shfl_up(incl, off, 64); if (lane >= off) incl += v; }
  const unsigned long long mask = __ballot(c > 0);
  if (lane == 63) wsum[wave] = incl;
  if (lane == 0) wnum[wave] = __popcll(mask);
  __syncthreads();
  if (threadIdx.x == 0) {
    const int tot = wsum[0] + wsum[1] + wsum[2] + wsum[3], num = wnum[0] + wnum[1] + wnum[2] + wnum[3];
    sbase = num ? atomicAdd(&m.cnt[CNT_CURSOR], tot) : 0;
    lbase = num ? atomicAdd(&m.cnt[CNT_WL], num) : 0;
  }
  __syncthreads();
  if (c > 0) {
    int wb = sbase, lb = lbase;
    for (int w = 0; w < wave; w++) { wb += wsum[w]; lb += wnum[w]; }
    const int start = wb + incl - c, i = lb + __popcll(mask & ((1ull << lane) - 1ull));
    m.nseg_a[(size_t)slot * cp + leaf] = start;
    m.nseg_b[(size_t)slot * cp + leaf] = start + c;
    m.ncnt[leaf] = 0;
    m.wl4[i] = make_int4(leaf, start, c, 0);
    if (c > 64) m.wlb[atomicAdd(&m.cnt[CNT_WLB], 1)] = i;   // (few)
  }
}

// Phase 5: every point takes its place in its leaf's segment (the arrival rank k_ins_leaf drew; the owner of the leaf orders the segment)
__global__ __launch_bounds__(256) void k_ins_scatter(MapView m, int slot, int n) {
  const int p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= n) return;
  const size_t mpz = (size_t)m.max_pts;
  const int node = m.pnode[(size_t)slot * mpz + p];
  if (node >= 0) m.perm[(size_t)slot * mpz + m.nseg_a[(size_t)slot * m.cap + node] + m.phash[p]] = p;
}

// one chunk of <= 64 points of a leaf in scan order: lane j prepares the terms of point j, lane k adds term k of the points in order.
// The terms are staged in two rounds of <= 32 (the LDS image of all 63 would hold a CU to 4 waves).
template <bool HAS_VAR>
__device__ __forceinline__ void ord_chunk(const MapView &m, const MapParams &P, int slot, int p, int pos, int leaf, int cnt, int lane, double *T, double &acc) {
  using C = OrdCfg<HAS_VAR>;
  constexpr int NR = HAS_VAR ? 2 : 1, RT = HAS_VAR ? 32 : 18, TS = RT | 1;
  const size_t mpz = (size_t)m.max_pts, W = (size_t)P.W;
  double t[C::NT];
  if (lane < cnt) {
    const double *pp = m.px + ((size_t)slot * mpz + p) * 3;
      const double bx = pp[0], by = pp[1], bz = pp[2];
    double var[9];
    if (HAS_VAR) {
#pragma unroll
      for (int k = 0; k < 9; k++) var[k] = m.pvar[((size_t)slot * mpz + p) * 9 + k];
    }
    double x, y, z;
    world_point(m.poses, bx, by, bz, x, y, z);
    ord_terms<HAS_VAR>(t, bx, by, bz, x, y, z, var);
    // the point joins the leaf's ordered storage
    { double *sp3 = m.sx + ((size_t)slot * mpz + pos) * 3; sp3[0] = bx; sp3[1] = by; sp3[2] = bz; }
    if (HAS_VAR) {
#pragma unroll
      for (int k = 0; k < 9; k++) m.svar[((size_t)slot * mpz + pos) * 9 + k] = var[k];
    }
    m.pleaf[(size_t)slot * mpz + pos] = leaf;
  }
#pragma unroll
  for (int r = 0; r < NR; r++) {
    if (lane < cnt) {
#pragma unroll
      for (int k = 0; k < RT; k++) if (r * RT + k < C::NT) T[lane * TS + k] = t[r * RT + k];
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup"); __builtin_amdgcn_wave_barrier();
    if (lane >= r * RT && lane < (r + 1) * RT && lane < C::NT)
      for (int j = 0; j < cnt; j++) acc += T[j * TS + (lane - r * RT)];
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup"); __builtin_amdgcn_wave_barrier();
  }
}

// Phase 6: one wave per leaf of the work list with <= 64 points — push VM:1129-1140 for the leaf's points in scan order.  The
// segment arrives in arrival order; a point's place in scan order is the number of smaller indices in the segment (64 lane reads).
// (Packing several small leaves into one wave was measured: no gain at any pack size — the kernel is bound by its scattered
// 8-byte accesses, not by per-wave latency.)
template <bool HAS_VAR>
__global__ __launch_bounds__(64) void k_ins_accum_ord(MapView m, MapParams P, int slot) {
  using C = OrdCfg<HAS_VAR>;
  constexpr int TS = (HAS_VAR ? 32 : 18) | 1;
  __shared__ double T[64 * TS];
  __shared__ int sp[64];
  const int lane = threadIdx.x;
  const int nseg = m.cnt[CNT_WL];
  const size_t W = (size_t)P.W;
  int *perm = m.perm + (size_t)slot * m.max_pts;
  // Workgroups go round-robin over the 8 XCDs (one L2 each): XCD x takes the x-th eighth of the list, so that the waves in flight on
  // one L2 work on neighbouring list entries (neighbouring leaves).  gridDim.x is a multiple of 8.
  const int xcd = blockIdx.x & 7, per = (nseg + 7) >> 3, s_end = (xcd + 1) * per < nseg ? (xcd + 1) * per : nseg;
  for (int s = xcd * per + (blockIdx.x >> 3); s < s_end; s += gridDim.x >> 3) {
    const int4 e = m.wl4[s];
    const int leaf = e.x, start = e.y, cnt = e.z;
    if (cnt > 64) continue;                              // k_ins_accum_big
    double *tgt = ord_target(m, P.W, slot, lane < C::NT ? lane : 0, leaf);
    double acc = lane < C::NT ? *tgt : 0.0;
    const int pa = lane < cnt ? perm[start + lane] : 0x7FFFFFFF;
    int rank = 0;
    for (int i = 0; i < cnt; i++) rank += (__shfl(pa, i, 64) < pa) ? 1 : 0;
    if (lane < cnt) { sp[rank] = pa; }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup"); __builtin_amdgcn_wave_barrier();
    const int p = lane < cnt ? sp[lane] : 0;
    if (lane < cnt) perm[start + lane] = p;              // the segment in scan order (recut / margi read it again)
    ord_chunk<HAS_VAR>(m, P, slot, p, start + lane, leaf, cnt, lane, T, acc);
    if (lane < C::NT) *tgt = acc;
    if (lane == 63) {                                    // N of both clusters: integers, exact in f64
      nlc_at(m, W, 9, slot, leaf) += (double)cnt;
      nadd_at(m, 9, leaf) += (double)cnt;
    }
  }
}

// Leaves with more than 64 points of the scan: one workgroup per leaf.  Scan order from a BITMAP of the scan's point indices in LDS
// (set the bits of the segment's points, count the bits below each set bit): O(n / 64 + points) whatever the segment size; scans of
// more than `win` points are covered window by window.
template <bool HAS_VAR>
__global__ __launch_bounds__(256) void k_ins_accum_big(MapView m, MapParams P, int slot, int n, int win) {
  using C = OrdCfg<HAS_VAR>;
  constexpr int TS = (HAS_VAR ? 32 : 18) | 1;
  extern __shared__ __attribute__((aligned(16))) unsigned long long bm[];     // [win / 64] bitmap, then [win / 64] int prefix
  const int nw = win / 64;
  int *pre = (int *)(bm + nw);
  double *T = (double *)bm;                                 // the four term images of the chain phase take over the bitmap's space
  __shared__ int wsum[4];
  __shared__ int sbase;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int nbig = m.cnt[CNT_WLB];
  const size_t mpz = (size_t)m.max_pts, W = (size_t)P.W, cp = (size_t)m.cap;
  int *perm = m.perm + (size_t)slot * mpz;
  for (int s = blockIdx.x; s < nbig; s += gridDim.x) {
    const int4 e = m.wl4[m.wlb[s]];
    const int leaf = e.x, start = e.y, cnt = e.z;
    // ---- order the segment: its points' indices are distinct integers in [0, n)
    int written = 0;
    for (int w0 = 0; w0 < n; w0 += win) {
      for (int i = tid; i < nw; i += 256) bm[i] = 0ull;
      __syncthreads();
      for (int i = tid; i < cnt; i += 256) {
        const int p = perm[start + i] - w0;
        if (p >= 0 && p < win) atomicOr(&bm[p >> 6], 1ull << (p & 63));
      }
      __syncthreads();
      // exclusive prefix of the word popcounts
      int loc = 0;
      const int per = (nw + 255) / 256;
      for (int k = 0; k < per; k++) { const int i = tid * per + k; if (i < nw) loc += __popcll(bm[i]); }
      int incl = loc;
      for (int off = 1; off < 64; off <<= 1) { const int v = __shfl_up(incl, off, 64); if (lane >= off) incl += v; }
      if (lane == 63) wsum[wave] = incl;
      __syncthreads();
      int wb = 0;
      for (int w = 0; w < wave; w++) wb += wsum[w];
      int run = wb + incl - loc;
      for (int k = 0; k < per; k++) { const int i = tid * per + k; if (i < nw) { pre[i] = run; run += __popcll(bm[i]); } }
      if (tid == 255) sbase = run;
      __syncthreads();
      // (the ordered indices go to sval_b first: perm still holds unread arrival-order entries of later windows)
      for (int i = tid; i < nw; i += 256) {
        unsigned long long b = bm[i];
        int o = pre[i];
        while (b) { const int j = __ffsll((long long)b) - 1; b &= b - 1; m.sval_b[start + written + o++] = w0 + i * 64 + j; }
      }
      __syncthreads();
      written += sbase;
      __syncthreads();
    }
    for (int i = tid; i < cnt; i += 256) perm[start + i] = m.sval_b[start + i];
    __syncthreads();
    // ---- the chains: the four waves prepare the terms of four chunks of 64 points side by side (the memory trips of a chunk are
    //      the long part), wave 0 then runs the chains through the four images in order.  The images reuse the bitmap's LDS.
    {
      constexpr int NR = HAS_VAR ? 2 : 1, RT = HAS_VAR ? 32 : 18;
      double *tgt = ord_target(m, P.W, slot, lane < C::NT ? lane : 0, leaf);
      double acc = (wave == 0 && lane < C::NT) ? *tgt : 0.0;
      double *Tw = T + (size_t)wave * 64 * TS;
      for (int c0 = 0; c0 < cnt; c0 += 256) {
        const int cb = c0 + wave * 64;
        const int cc = cnt - cb < 0 ? 0 : cnt - cb < 64 ? cnt - cb : 64;
        double t[C::NT];
        if (lane < cc) {
          const int pos = start + cb + lane, p = perm[pos];
          const double *pp = m.px + ((size_t)slot * mpz + p) * 3;
          const double bx = pp[0], by = pp[1], bz = pp[2];
          double var[9];
          if (HAS_VAR) {
#pragma unroll
            for (int k = 0; k < 9; k++) var[k] = m.pvar[((size_t)slot * mpz + p) * 9 + k];
          }
          double x, y, z;
          world_point(m.poses, bx, by, bz, x, y, z);
          ord_terms<HAS_VAR>(t, bx, by, bz, x, y, z, var);
          { double *sp3 = m.sx + ((size_t)slot * mpz + pos) * 3; sp3[0] = bx; sp3[1] = by; sp3[2] = bz; }
          if (HAS_VAR) {
#pragma unroll
            for (int k = 0; k < 9; k++) m.svar[((size_t)slot * mpz + pos) * 9 + k] = var[k];
          }
          m.pleaf[(size_t)slot * mpz + pos] = leaf;
        }
#pragma unroll
        for (int r = 0; r < NR; r++) {
          if (lane < cc) {
#pragma unroll
            for (int k = 0; k < RT; k++) if (r * RT + k < C::NT) Tw[lane * TS + k] = t[r * RT + k];
          }
          __syncthreads();
          if (wave == 0 && lane >= r * RT && lane < (r + 1) * RT && lane < C::NT) {
            const int rem = cnt - c0 < 256 ? cnt - c0 : 256;                      // points of the four images, in order
            for (int j = 0; j < rem; j++) acc += T[j * TS + (lane - r * RT)];
          }
          __syncthreads();
        }
      }
      if (wave == 0 && lane < C::NT) *tgt = acc;
      if (tid == 63) {
        nlc_at(m, W, 9, slot, leaf) += (double)cnt;
        nadd_at(m, 9, leaf) += (double)cnt;
      }
    }
    __syncthreads();
  }
}

// Fixed points (VM:2108-2152): new root -> push_fix_novar on the root; else allocate_fix (descend while layer < max_layer).
// Same order-preserving scheme as the window scans: leaf per point, stable sort by leaf, one wave per leaf.  The points of a leaf
// become ONE BLOCK of consecutive pool entries (in call order) that joins the leaf's chain: point_fix of the reference.
__global__ void k_fix_leaf(MapView m, MapParams P, int base, int n) {
  const int p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= n) return;
  const size_t cf = (size_t)m.cap_fix;
  const int q = base + p;
  const int h = m.phash[p];
  int node = h >= 0 ? m.hvals[h] : -1;
  if (node >= 0) {
    const double x = m.fx[(size_t)q * 3], y = m.fx[(size_t)q * 3 + 1], z = m.fx[(size_t)q * 3 + 2];
    while (m.nstate[node] == 1) node = m.nchild[node] + octant_of(m, node, x, y, z);
  }
  m.skey_a[p] = node >= 0 ? (unsigned int)node : 0xFFFFFFFFu;
  m.sval_a[p] = p;
}
__global__ __launch_bounds__(256) void k_fix_heads(MapView m, int n) {       // work list = start positions of the leaf groups
  __shared__ int wbase[4];
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  bool head = false;
  if (i < n) {
    const unsigned int key = m.skey_b[i];
    head = key != 0xFFFFFFFFu && ((i == 0) || m.skey_b[i - 1] != key);
  }
  const unsigned long long mask = __ballot(head);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (lane == 0) wbase[wave] = __popcll(mask);
  __syncthreads();
  if (threadIdx.x == 0) {
    int tot = 0;
    for (int w = 0; w < 4; w++) { const int c = wbase[w]; wbase[w] = tot; tot += c; }
    const int b0 = tot ? atomicAdd(&m.cnt[CNT_WL], tot) : 0;
    for (int w = 0; w < 4; w++) wbase[w] += b0;
  }
  __syncthreads();
  if (head) m.wl[wbase[wave] + __popcll(mask & ((1ull << lane) - 1ull))] = i;
}
// link block [qb, qb + len) to the end of leaf's chain (one lane; a leaf is served by one wave per call)
__device__ __forceinline__ void fix_chain_append(const MapView &m, int leaf, int qb, int len) {
  const int blk = atomicAdd(&m.cnt[CNT_FBLK], 1);
  if (blk >= m.cap_fix) { atomicMax(&m.cnt[CNT_OVERFLOW], 3); return; }
  m.fb_base[blk] = qb; m.fb_len[blk] = len; m.fb_next[blk] = -1;
  const int tail = m.nfb_tail[leaf];
  if (tail < 0) m.nfb_head[leaf] = blk; else m.fb_next[tail] = blk;
  m.nfb_tail[leaf] = blk;
}
// one wave per leaf group: push_fix_novar VM:1168-1178 for the group's points in call order; pts = the caller's [n][3] array
__global__ __launch_bounds__(64) void k_fix_accum_ord(MapView m, MapParams P, int base, int n, const double *__restrict__ pts) {
  __shared__ double T[64 * 9];
  const int lane = threadIdx.x;
  const int nseg = m.cnt[CNT_WL];
  const size_t cp = (size_t)m.cap, cf = (size_t)m.cap_fix;
  for (int s = blockIdx.x; s < nseg; s += gridDim.x) {
    const int start = m.wl[s];
    const int leaf = (int)m.skey_b[start];
    int end = start;                               // the group ends where the key changes
    while (true) {
      const int i = end + lane;
      const bool same = i < n && (int)m.skey_b[i] == leaf;
      const unsigned long long mk = __ballot(same);
      if (mk == ~0ull) { end += 64; continue; }
      end += __ffsll((long long)~mk) - 1;
      break;
    }
    const bool store = m.nlayer[leaf] < P.max_layer;     // point_fix.push_back only below max_layer  VM:1171-1172
    double *tgt = lane < 9 ? &nfix_at(m, lane, leaf) : &nadd_at(m, lane < 18 ? lane - 9 : 0, leaf);
    double acc = lane < 18 ? *tgt : 0.0;
    for (int c0 = start; c0 < end; c0 += 64) {
      const int i = c0 + lane;
      if (i < end) {
        const int p = m.sval_b[i];
        const double x = pts[(size_t)p * 3], y = pts[(size_t)p * 3 + 1], z = pts[(size_t)p * 3 + 2];
        const int q = base + i;                      // pool entries in GROUP order: the group is one block
        m.fx[(size_t)q * 3] = x; m.fx[(size_t)q * 3 + 1] = y; m.fx[(size_t)q * 3 + 2] = z;
        m.fnode[q] = store ? leaf : -1;
        double *t = T + lane * 9;
        t[0] = x * x; t[1] = x * y; t[2] = x * z; t[3] = y * y; t[4] = y * z; t[5] = z * z; t[6] = x; t[7] = y; t[8] = z;
      }
      __syncthreads();
      const int cnt = end - c0 < 64 ? end - c0 : 64;
      if (lane < 18) {
        const int k = lane < 9 ? lane : lane - 9;    // pcr_fix.push(pnt) and pcr_add.push(pnt): the same terms, two chains
        for (int j = 0; j < cnt; j++) acc += T[j * 9 + k];
      }
      __syncthreads();
    }
    if (lane < 18) *tgt = acc;
    if (lane == 63) {
      const double dn = (double)(end - start);
      nfix_at(m, 9, leaf) += dn; nadd_at(m, 9, leaf) += dn;
      m.f_touched[leaf] = 1;
      if (store) fix_chain_append(m, leaf, base + start, end - start);
    }
  }
}

// ------------------------------------------------------------------------------------------------ K2: recut
__device__ __forceinline__ bool in_scope(const MapView &m, const MapParams &P, int node, int multi) {
  if (m.nlayer[node] < 0) return false;   // pruned
  if (!multi) return true;
  return m.f_slide[m.nroot[node]] != 0;
}

// One thread per node of layer L: leaf logic of OctoTree::recut VM:1399-1450.
// The grid covers the node CAPACITY; the live range is the node count snapshotted on the device before the launch
// (CNT_SNAP) — nodes created by this launch's own splits must not be visited by it, and reading the count back to size
// the grid cost one ~18 us host round trip per level.
__global__ __launch_bounds__(256) void k_recut_leaf(MapView m, MapParams P, int L, int multi, int epoch) {
  __shared__ int wbase[4];
  const int id = blockIdx.x * blockDim.x + threadIdx.x;
  const int nn = m.cnt[CNT_SNAP] < m.cap ? m.cnt[CNT_SNAP] : m.cap;
  if ((int)(blockIdx.x * blockDim.x) >= nn) return;         // (the grid covers the capacity, not the node count)
  const size_t cp = (size_t)m.cap;
  bool split = false;
  do {
    if (id >= nn) break;
    if (m.nlayer[id] != L || m.nstate[id] != 0) break;
    if (multi && slide_count(m, P) < P.thread_num) break;   // VS:1693-1694
    if (!in_scope(m, P, id, multi)) break;
    m.nopt[id] = -1;
    const double N = nadd_at(m, 9, id);
    if (N <= P.min_point[L]) { m.f_plane[id] = 0; break; }   // VM:1406-1410
    if (!m.f_exist[id] || !m.f_sw[id]) break;                // VM:1412-1413
    const double b0 = nadd_at(m, 6, id) / N, b1 = nadd_at(m, 7, id) / N, b2 = nadd_at(m, 8, id) / N;
    double w0, w1, w2, V[9];
    eig3_sym_dev(nadd_at(m, 0, id) / N - b0 * b0, nadd_at(m, 1, id) / N - b1 * b0, nadd_at(m, 2, id) / N - b2 * b0,
                 nadd_at(m, 3, id) / N - b1 * b1, nadd_at(m, 4, id) / N - b2 * b1, nadd_at(m, 5, id) / N - b2 * b2, w0, w1, w2, V);
    neval_at(m, 0, id) = w0; neval_at(m, 1, id) = w1; neval_at(m, 2, id) = w2;
#pragma unroll
    for (int k = 0; k < 9; k++) nevec_at(m, k, id) = V[k];
    const bool plane = (w0 < P.min_eigen_value) && ((w0 / w2) < P.plane_thre[L]);   // plane_judge VM:1194
    m.f_plane[id] = plane ? 1 : 0;
    if (plane || L >= P.max_layer) break;
    // subdivide: children are created as a block of 8 (untouched octants stay empty leaves, which every traversal skips)
    const int base = alloc_nodes(m, CNT_FREE_BLOCKS, m.nfree_blk, 8);
    if (base + 8 > m.cap) { atomicMax(&m.cnt[CNT_OVERFLOW], 1); break; }
    const double cx = m.ncenter[id], cy = m.ncenter[cp + id], cz = m.ncenter[2 * cp + id];
    const float ql = m.nql[id];
    for (int o = 0; o < 8; o++) {
      const int ox = o >> 2, oy = (o >> 1) & 1, oz = o & 1;
      // VM:1227-1231: double + int * float
      init_node(m, P.W, base + o, m.nkey[id], m.nroot[id], id, L + 1, (L == 0 ? o : m.npath[id] * 8 + o), cx + (2 * ox - 1) * ql, cy + (2 * oy - 1) * ql,
                cz + (2 * oz - 1) * ql, ql / 2);
    }
    m.nchild[id] = base;
    m.nsplit[id] = epoch;    // the point kernels of this pass move this leaf's points to the children
    m.nstate[id] = 1;        // VM:1449
    m.f_sw[id] = 0;          // sw->clear(); sws.push_back(sw); sw = nullptr  VM:1445-1447
    for (int k = 0; k < 10 * P.W; k++) m.nlc[(size_t)id * 10 * P.W + k] = 0.0;
    split = true;
  } while (false);
  // the leaves this level split form the work list of k_recut_push (one returning atomic per workgroup; the order carries no meaning)
  const unsigned long long mask = __ballot(split);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (lane == 0) wbase[wave] = __popcll(mask);
  __syncthreads();
  if (threadIdx.x == 0) {
    int tot = 0;
    for (int w = 0; w < 4; w++) { const int c = wbase[w]; wbase[w] = tot; tot += c; }
    const int base = tot ? atomicAdd(&m.cnt[CNT_SPLIT], tot) : 0;
    for (int w = 0; w < 4; w++) wbase[w] += base;
  }
  __syncthreads();
  if (split) m.nsl[wbase[wave] + __popcll(mask & ((1ull << lane) - 1ull))] = id;
}

// Window points of split leaves -> children, keyed with the CURRENT poses (subdivide VM:1307-1338 + push VM:1105-1143), in the
// reference's order: frames 0 .. win_count-1, inside a frame the leaf's points in scan order.  One WORKGROUP of four waves per split
// leaf X (a recut level splits a few hundred to a few thousand leaves, some with thousands of points: the level lasts as long as
// its largest leaf, so the leaf itself is spread out).  X's points of a slot are the entries with pleaf == X inside the segment the
// slot gave to X — or to the ancestor of X that was the leaf when the scan was inserted — in scan order.  Per group of 256
// candidates thread j prepares point j's terms (the memory trips and the arithmetic: the long part); wave w then serves children
// 2w and 2w+1: lane k adds term k of the child's points, in order, to the child's scalar k (kept in LDS between groups).  The terms
// are staged in rounds of 21 (LDS).  The children were created empty by this level's k_recut_leaf, so every chain starts from zero
// like a new OctoTree.
template <bool HAS_VAR>
__global__ __launch_bounds__(256) void k_recut_push(MapView m, MapParams P, int win_count, int child_layer) {
  using C = OrdCfg<HAS_VAR>;
  constexpr int NR = HAS_VAR ? 3 : 1, RT = HAS_VAR ? 21 : 18, TS = RT | 1, GC = 256;   // TS odd: conflict-free rows
  __shared__ double T[GC * TS];
  __shared__ double A[8 * 64];          // child accumulators: [0..8] body cluster of the current frame, [9..17] pcr_add, [18..62] cov_add
  __shared__ unsigned long long cm[8][4];   // per child: which candidates of each of the four 64-point images are its points
  __shared__ int fj[GC];
  constexpr int FBL = 128;
  __shared__ int fbq[FBL], fbo[FBL + 1], fbn[2];   // blocks of X's fixed points (pool start, offset in the common index space), their number, 'more to come'
  __shared__ int nw[8], nb[8], nf[8], curf[8];   // per child: window points so far / points of the frame being added / fixed points / that frame
  __shared__ int fs[VBA_MAX_WIN], flen[VBA_MAX_WIN], foff[VBA_MAX_WIN + 1];   // per frame: start and length of the candidate segment; offsets in the common index space
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int nsplit = m.cnt[CNT_SPLIT];
  const size_t mpz = (size_t)m.max_pts, W = (size_t)P.W, cp = (size_t)m.cap, cf = (size_t)m.cap_fix;
  // one round of the chains over the group's images: T holds terms [r * RT, (r + 1) * RT) of the 256 candidates, cm their children
  auto chains = [&](int r, bool framed, int base) {
#pragma unroll 1
    for (int cc = 0; cc < 2; cc++) {
      const int c = 2 * wave + cc;
      const unsigned long long k0 = cm[c][0], k1 = cm[c][1], k2 = cm[c][2], k3 = cm[c][3];
      if ((k0 | k1 | k2 | k3) == 0ull) continue;
      if (lane >= 9 && lane >= r * RT && lane < (r + 1) * RT && lane < C::NT) {   // pcr_add / cov_add: one chain through all frames
        double acc = A[c * 64 + lane];
#pragma unroll 1
        for (int i = 0; i < 4; i++) {
          unsigned long long mk = i == 0 ? k0 : i == 1 ? k1 : i == 2 ? k2 : k3;
          while (mk) { const int j = __ffsll((long long)mk) - 1; mk &= mk - 1; acc += T[(i * 64 + j) * TS + lane - r * RT]; }
        }
        A[c * 64 + lane] = acc;
      }
      if (r == 0 && !framed && lane == 63) nf[c] += __popcll(k0) + __popcll(k1) + __popcll(k2) + __popcll(k3);
      if (r == 0 && framed && (lane < 9 || lane == 63)) {       // pcrs_local: a chain per frame; lane 63 keeps the counts
        double acc = A[c * 64 + (lane < 9 ? lane : 0)];
        int cfr = curf[c], cnt = nb[c];
#pragma unroll 1
        for (int i = 0; i < 4; i++) {
          unsigned long long mk = i == 0 ? k0 : i == 1 ? k1 : i == 2 ? k2 : k3;
          while (mk) {
            const int j = __ffsll((long long)mk) - 1; mk &= mk - 1;
            const int fjj = fj[i * 64 + j];
            if (fjj != cfr) {
              if (cfr >= 0 && cnt > 0) {
                if (lane < 9) nlc_at(m, W, lane, P.mp[cfr], base + c) = acc;
                else nlc_at(m, W, 9, P.mp[cfr], base + c) = (double)cnt;
              }
              acc = 0.0; cnt = 0; cfr = fjj;
            }
            if (lane < 9) acc += T[(i * 64 + j) * TS + lane];
            cnt++;
          }
        }
        if (lane < 9) A[c * 64 + lane] = acc;
        else { curf[c] = cfr; nb[c] = cnt; nw[c] += __popcll(k0) + __popcll(k1) + __popcll(k2) + __popcll(k3); }
      }
    }
  };
  // the group's terms go through T round by round; `child` < 0 = not a point of X
  auto rounds = [&](const double *t, int child, bool framed, int base) {
    unsigned long long mine = 0ull;
#pragma unroll
    for (int c = 0; c < 8; c++) { const unsigned long long b = __ballot(child == c); if (lane == c) mine = b; }
    if (lane < 8) cm[lane][wave] = mine;
#pragma unroll
    for (int r = 0; r < NR; r++) {
      if (child >= 0) {
#pragma unroll
        for (int k = 0; k < RT; k++) if (r * RT + k < C::NT) T[tid * TS + k] = t[r * RT + k];
      }
      __syncthreads();
      chains(r, framed, base);
      __syncthreads();
    }
  };
  for (int s = blockIdx.x; s < nsplit; s += gridDim.x) {
    const int X = m.nsl[s];
    const int base = m.nchild[X];
    const double ctr[3] = {m.ncenter[X], m.ncenter[cp + X], m.ncenter[2 * cp + X]};     // (requested with the leaf's other fields)
    for (int i = tid; i < 8 * 64; i += GC) A[i] = 0.0;
    if (tid < 8) { nw[tid] = 0; nf[tid] = 0; }
    __syncthreads();
    // ---- fix_divide (VM:1270-1299) + push_fix (VM:1149-1162): X's fixed points in point_fix order = the chains of X's ancestors
    //      (blocks that arrived while the ancestor was the leaf), oldest ancestor first, then X's own chain; entries with fnode == X
    if (nfix_at(m, 9, X) != 0.0) {               // VM:1433: if (pcr_fix.N != 0)
      // thread 0 walks the chains (one dependent load per block) and lists the blocks; their entries then form ONE index space, so
      // that a leaf with many short blocks pays the memory trips and the chain rounds once per 256 entries, not once per block
      int path[8], np = 0, pi = 0, blk = -1;
      if (tid == 0) {
        for (int a = X; a >= 0 && np < 8; a = m.nparent[a]) path[np++] = a;
        pi = np - 1; blk = m.nfb_head[path[pi]];
      }
      while (true) {
        if (tid == 0) {
          int nbk = 0, o = 0;
          while (nbk < FBL && pi >= 0) {
            if (blk < 0) { pi--; if (pi >= 0) blk = m.nfb_head[path[pi]]; continue; }
            fbq[nbk] = m.fb_base[blk]; fbo[nbk] = o; o += m.fb_len[blk]; nbk++;
            blk = m.fb_next[blk];
          }
          fbo[nbk] = o; fbn[0] = nbk; fbn[1] = pi >= 0 ? 1 : 0;
        }
        __syncthreads();
        const int nbk = fbn[0], ftotal = fbo[nbk], more = fbn[1];
#ifdef VBA_DIAG
        if (tid == 0) { atomicAdd(&m.cnt[CNT_DBG2], nbk); atomicAdd(&m.cnt[CNT_DBG3], ftotal); }
#endif
        for (int c0 = 0; c0 < ftotal; c0 += GC) {
          const int tpos = c0 + tid;
          int child = -1;
          double t[C::NT];
          if (tpos < ftotal) {
            int lo = 0, hi = nbk - 1;                   // the block of entry tpos: last b with fbo[b] <= tpos
            while (lo < hi) { const int mid = (lo + hi + 1) >> 1; if (fbo[mid] <= tpos) lo = mid; else hi = mid - 1; }
            const int q = fbq[lo] + (tpos - fbo[lo]);
            if (m.fnode[q] == X) {
              const double x = m.fx[(size_t)q * 3], y = m.fx[(size_t)q * 3 + 1], z = m.fx[(size_t)q * 3 + 2];
              double var[9];
              if (HAS_VAR) {
#pragma unroll
                for (int k = 0; k < 9; k++) var[k] = m.fvar[(size_t)q * 9 + k];
              }
              child = 4 * (x > ctr[0] ? 1 : 0) + 2 * (y > ctr[1] ? 1 : 0) + (z > ctr[2] ? 1 : 0);
              ord_terms<HAS_VAR>(t, 0.0, 0.0, 0.0, x, y, z, var);     // pcr_fix.push(pnt); pcr_add.push(pnt); cov_add += Bf_var(pv, pnt)
              m.fnode[q] = (child_layer < P.max_layer) ? base + child : -1;          // VM:1152-1153
            }
          }
          rounds(t, child, false, base);
        }
        __syncthreads();                                  // (the block list is rewritten by the next batch)
        if (!more) break;
      }
      // pcr_fix of the children = the state of the pcr_add chains after the fixed points
      for (int c = wave; c < 8; c += 4) {
        if (nf[c] == 0) continue;
        if (lane >= 9 && lane < 18) nfix_at(m, lane - 9, base + c) = A[c * 64 + lane];
        if (lane == 63) nfix_at(m, 9, base + c) = (double)nf[c];
      }
    }
    // ---- subdivide (VM:1307-1338): the frames' candidate entries form ONE index space [0, total) in frame order, so short segments
    //      share a group and the memory trips of the frames overlap instead of following each other
    if (tid < win_count) {
      const int slot = P.mp[tid];
      int anc = X;                       // the node that was the leaf when this slot's scan was inserted
      while (anc >= 0 && m.nseg_b[(size_t)slot * cp + anc] == m.nseg_a[(size_t)slot * cp + anc]) anc = m.nparent[anc];
      fs[tid] = anc >= 0 ? m.nseg_a[(size_t)slot * cp + anc] : 0;
      flen[tid] = anc >= 0 ? m.nseg_b[(size_t)slot * cp + anc] - fs[tid] : 0;
    }
    if (tid < 8) { curf[tid] = -1; nb[tid] = 0; }
    __syncthreads();
    if (tid == 0) { int o = 0; for (int k = 0; k < win_count; k++) { foff[k] = o; o += flen[k]; } foff[win_count] = o; }
    __syncthreads();
    const int total = foff[win_count];
#ifdef VBA_DIAG
    if (tid == 0) atomicAdd(&m.cnt[CNT_DBG0], total);
#endif
    for (int c0 = 0; c0 < total; c0 += GC) {
      const int tpos = c0 + tid;
      int child = -1, f = 0;
      double t[C::NT];
      if (tpos < total) {
        while (tpos >= foff[f + 1]) f++;
        const int slot = P.mp[f];
        const size_t p = (size_t)(fs[f] + (tpos - foff[f]));      // position in the slot's ordered storage: no indirection
        // the point is requested together with its owner tag (one memory trip instead of two: this pass is a chain of trips per leaf,
        // the bytes of the candidates that turn out to be a sibling's do not matter)
        const int owner = m.pleaf[(size_t)slot * mpz + p];
        const double *sp3 = m.sx + ((size_t)slot * mpz + p) * 3;
        const double bx = sp3[0], by = sp3[1], bz = sp3[2];
        double var[9];
        if (HAS_VAR) {
#pragma unroll
          for (int k = 0; k < 9; k++) var[k] = m.svar[((size_t)slot * mpz + p) * 9 + k];
        }
        if (owner == X) {
          double x, y, z;
          world_point(m.poses + 12 * f, bx, by, bz, x, y, z);
          child = 4 * (x > ctr[0] ? 1 : 0) + 2 * (y > ctr[1] ? 1 : 0) + (z > ctr[2] ? 1 : 0);   // octant_of(X): VM:1214-1219
          ord_terms<HAS_VAR>(t, bx, by, bz, x, y, z, var);
          m.pleaf[(size_t)slot * mpz + p] = base + child;
#ifdef VBA_DIAG
          atomicAdd(&m.cnt[CNT_DBG1], 1);
#endif
        }
      }
      fj[tid] = f;
      rounds(t, child, true, base);
    }
    // the last frame of every child, then the children's world sums
    for (int c = wave; c < 8; c += 4) {
      const int cfr = curf[c];
      if (cfr >= 0 && nb[c] > 0) {
        if (lane < 9) nlc_at(m, W, lane, P.mp[cfr], base + c) = A[c * 64 + lane];
        if (lane == 9) nlc_at(m, W, 9, P.mp[cfr], base + c) = (double)nb[c];
      }
      if (nw[c] + nf[c] == 0) continue;
      const int ch = base + c;
      if (lane >= 9 && lane < C::NT) *ord_target(m, P.W, 0, lane, ch) = A[c * 64 + lane];
      if (lane == 63) {
        nadd_at(m, 9, ch) = (double)(nw[c] + nf[c]);
        m.f_touched[ch] = 1;
        if (nw[c] > 0) { m.f_sw[ch] = 1; m.f_exist[ch] = 1; }     // push attaches a SlideWindow and sets isexist (VM:1110-1125); push_fix does neither
      }
    }
    __syncthreads();
  }
}

// tras_opt VM:1605-1638, pass 1: assign factor indices.
// (one returning atomic per workgroup on the factor counter, as k_margi_points)
__global__ __launch_bounds__(256) void k_extract_count(MapView m, MapParams P, int multi) {
  __shared__ int wbase[4];
  const int id = blockIdx.x * blockDim.x + threadIdx.x;
  const int nn = m.cnt[CNT_NODES] < m.cap ? m.cnt[CNT_NODES] : m.cap;
  if ((int)(blockIdx.x * blockDim.x) >= nn) return;
  bool take = id < nn && m.nstate[id] == 0;
  if (take && multi && slide_count(m, P) < P.thread_num) take = false;
  if (take && !in_scope(m, P, id, multi)) take = false;
  if (take && !(m.f_exist[id] && m.f_plane[id] && m.f_sw[id])) take = false;
  const size_t cp = (size_t)m.cap;
  if (take && neval_at(m, 0, id) / neval_at(m, 1, id) > 0.12) take = false;   // VM:1615
  const unsigned long long mask = __ballot(take);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (lane == 0) wbase[wave] = __popcll(mask);
  __syncthreads();
  if (threadIdx.x == 0) {
    int tot = 0;
    for (int w = 0; w < 4; w++) { const int c = wbase[w]; wbase[w] = tot; tot += c; }
    const int base = tot ? atomicAdd(&m.cnt[CNT_FACTORS], tot) : 0;
    for (int w = 0; w < 4; w++) wbase[w] += base;
  }
  __syncthreads();
  if (take) {
    const int a = wbase[wave] + __popcll(mask & ((1ull << lane) - 1ull));
    m.nopt[id] = a;                                          // opt_state  VM:1626
    m.nflist[a] = id;                                        // (a < number of nodes <= cap)
  }
}
// pass 1b: order the factors by OCCUPANCY MASK (which frames of the window see the voxel).  The factor store is SoA with the voxel
// index fastest and a scalar of an empty (voxel, frame) slot is never read — but memory moves whole lines: in tras_opt order (43 %
// of the slots occupied, scattered) the passes fetched 2.06x the occupied slots' bytes (64-byte lines; PMC: 17.0 MB against 11.2 MB
// algorithmic per residual pass); with equal masks adjacent a frame's occupied slots are runs and the factor is 1.04x.  The order of
// the factors carries no meaning (push_voxel order of a recursive traversal in the reference; opt_state follows it, VM:1626).
// Counting sort over the masks of the first min(W, 10) frames (<= 1024 buckets): key + histogram, exclusive scan (one workgroup),
// scatter.  Both atomic stages aggregate in LDS first: thousands of factors share the popular masks and returning atomics on one
// address serialise in L2 (~10 ns each) — the direct form took 24 + 20 us for 18k factors.
constexpr int EXTRACT_NB_MAX = 1024;
__global__ __launch_bounds__(256) void k_extract_key(MapView m, MapParams P, int nfac, int nb) {
  __shared__ int lh[EXTRACT_NB_MAX];
  for (int t = threadIdx.x; t < nb; t += 256) lh[t] = 0;
  __syncthreads();
  const int a = blockIdx.x * blockDim.x + threadIdx.x;
  if (a < nfac) {
    const int id = m.nflist[a];
    const size_t cp = (size_t)m.cap, W = (size_t)P.W;
    unsigned int key = 0;
    for (int i = 0; i < P.W && i < 10; i++) key |= (nlc_at(m, W, 9, P.mp[i], id) != 0.0) ? (1u << i) : 0u;
    key = (unsigned int)mask_bucket(key, P.W < 10 ? P.W : 10);   // order of the store: popcount DESCENDING, then the mask (heavy tiles of the Hessian pass first)
    m.nfkey[a] = key;
    m.nfl2[a] = id;
    atomicAdd(&lh[key], 1);
  }
  __syncthreads();
  for (int t = threadIdx.x; t < nb; t += 256) { const int c = lh[t]; if (c) atomicAdd(&m.fhist[t], c); }
}
__global__ __launch_bounds__(1024) void k_extract_scan(MapView m, int nbuckets) {   // fhist[k] <- number of factors with a smaller mask
  __shared__ int part[1024];
  const int tid = threadIdx.x;
  const int c = tid < nbuckets ? m.fhist[tid] : 0;
  part[tid] = c;
  __syncthreads();
  for (int off = 1; off < 1024; off <<= 1) {
    const int v = (tid >= off) ? part[tid - off] : 0;
    __syncthreads();
    part[tid] += v;
    __syncthreads();
  }
  if (tid < nbuckets) m.fhist[tid] = part[tid] - c;
}
__global__ __launch_bounds__(256) void k_extract_scatter(MapView m, int nfac, int nb) {
  __shared__ int lh[EXTRACT_NB_MAX];
  for (int t = threadIdx.x; t < nb; t += 256) lh[t] = 0;
  __syncthreads();
  const int a = blockIdx.x * blockDim.x + threadIdx.x;
  unsigned int key = 0;
  int local = 0;
  if (a < nfac) { key = m.nfkey[a]; local = atomicAdd(&lh[key], 1); }     // rank among this workgroup's factors with the same mask
  __syncthreads();
  for (int t = threadIdx.x; t < nb; t += 256) { const int c = lh[t]; if (c) lh[t] = atomicAdd(&m.fhist[t], c); }   // the workgroup's range
  __syncthreads();
  if (a < nfac) {
    const int id = m.nfl2[a];
    const int pos = lh[key] + local;
    m.nflist[pos] = id;
    m.nopt[id] = pos;                                        // opt_state  VM:1626
  }
}
// pass 2: write the SoA factor store (push_voxel VM:139-147), frames in ring order pcrs[i] = pcrs_local[mp[i]] VM:1623-1624.
// A node's record (body clusters of all slots, fix, pcr) is contiguous, the store is component-major: a workgroup moves 32 factors
// through an LDS tile — record-order reads (consecutive lanes, consecutive scalars of one node), factor-order writes.  Rows of a
// factor: 0..10W-1 = the body clusters [k][i], then fix (10), pcr (10), coe, eigval (3), eigvec (9).
constexpr int XW_F = 32;
__global__ __launch_bounds__(256) void k_extract_write(MapView m, MapParams P, FactorView f, int nfac) {
  extern __shared__ double xw_buf[];                          // [NR][XW_F + 1]
  __shared__ int ids[XW_F];
  __shared__ int inv[VBA_MAX_WIN];
  const size_t cp = (size_t)m.cap, vs = (size_t)f.vs;
  const int W = P.W, ncl = 10 * W, NR = ncl + 33;
  const int a0 = blockIdx.x * XW_F;
  int nf = nfac < (int)f.vs ? nfac : (int)f.vs;
  nf = nf - a0 < XW_F ? nf - a0 : XW_F;
  if (nf <= 0) return;
  if (threadIdx.x < nf) ids[threadIdx.x] = m.nflist[a0 + threadIdx.x];
  if (threadIdx.x >= 64 && threadIdx.x < 64 + W) inv[P.mp[threadIdx.x - 64]] = threadIdx.x - 64;     // slot -> place in the ring
  __syncthreads();
  for (int e = threadIdx.x; e < nf * NR; e += 256) {
    const int j = e / NR, q = e - j * NR;
    const size_t id = (size_t)ids[j];
    double v; int row = q;
    if (q < ncl) { const int s = q / 10, k = q - 10 * s; v = m.nlc[id * ncl + q]; row = k * W + inv[s]; }
    else {
      const int r = q - ncl;
      if (r < 10) v = nfix_at(m, r, id);
      else if (r < 20) v = nadd_at(m, r - 10, id);
      else if (r == 20) v = 1.0;                              // coe  VM:1619
      else if (r < 24) v = neval_at(m, r - 21, id);
      else v = nevec_at(m, r - 24, id);
    }
    xw_buf[row * (XW_F + 1) + j] = v;
  }
  __syncthreads();
  for (int e = threadIdx.x; e < NR * XW_F; e += 256) {
    const int row = e / XW_F, j = e - row * XW_F;
    if (j >= nf) continue;
    const double v = xw_buf[row * (XW_F + 1) + j];
    const size_t a = (size_t)(a0 + j);
    if (row < ncl) f.cl[(size_t)row * vs + a] = v;
    else {
      const int r = row - ncl;
      if (r < 10) f.fix[(size_t)r * vs + a] = v;
      else if (r < 20) f.pcr[(size_t)(r - 10) * vs + a] = v;
      else if (r == 20) f.coe[a] = v;
      else if (r < 24) f.eigval[(size_t)(r - 21) * vs + a] = v;
      else f.eigvec[(size_t)(r - 24) * vs + a] = v;
    }
  }
}

// ------------------------------------------------------------------------------------------------ K5: marginalise
__device__ __forceinline__ void cluster_transform_dev(const double *c /*10*/, const double *R /*12*/, double *o /*10*/) {
  const Cl10 w = cluster_transform_exact(c[0], c[1], c[2], c[3], c[4], c[5], c[6], c[7], c[8], c[9], R);   // the reference's operation order
  o[0] = w.p00; o[1] = w.p10; o[2] = w.p20; o[3] = w.p11; o[4] = w.p21; o[5] = w.p22; o[6] = w.v0; o[7] = w.v1; o[8] = w.v2; o[9] = w.n;
}

// plane_update VM:1344-1388.  nplane layout: center(3) normal(3) radius(1) plane_var(36 row-major)
// Everything is unrolled onto registers (the caller is launched with 256-thread bounds): the first version indexed its local
// arrays at run time and re-read cov_add from memory inside the triple loop — 564 B of scratch per lane, 243 loads.
__device__ __forceinline__ void plane_update_dev(const MapView &m, int id, const double *add /*10*/, const double *ev /*3*/, const double *U /*9*/) {
  const size_t cp = (size_t)m.cap;
  double cv[45];                                           // cov_add, upper triangle of the symmetric 9x9
#pragma unroll
  for (int k = 0; k < 45; k++) cv[k] = ncov_at(m, k, id);
  const double nv = 1.0 / add[9];
  const double c[3] = {add[6] * nv, add[7] * nv, add[8] * nv};
  const double u[3][3] = {{U[0], U[3], U[6]}, {U[1], U[4], U[7]}, {U[2], U[5], U[8]}};   // u[k] = column k
  double uc[3][9];
#pragma unroll
  for (int r = 0; r < 3; r++)
#pragma unroll
    for (int k = 0; k < 9; k++) uc[r][k] = 0.0;
#pragma unroll
  for (int k = 1; k < 3; k++) {
    // ukl = u[k] u[0]^T ; fkl(0..5) from its symmetric part, fkl(6..8) = -(u[k].c u[0] + u[0].c u[k])
    double fkl[9];
    const double a[3] = {u[k][0], u[k][1], u[k][2]}, b[3] = {u[0][0], u[0][1], u[0][2]};
    fkl[0] = a[0] * b[0]; fkl[1] = a[1] * b[0] + a[0] * b[1]; fkl[2] = a[2] * b[0] + a[0] * b[2];
    fkl[3] = a[1] * b[1]; fkl[4] = a[1] * b[2] + a[2] * b[1]; fkl[5] = a[2] * b[2];
    const double ac = a[0] * c[0] + a[1] * c[1] + a[2] * c[2], bc = b[0] * c[0] + b[1] * c[1] + b[2] * c[2];
#pragma unroll
    for (int j = 0; j < 3; j++) fkl[6 + j] = -(ac * b[j] + bc * a[j]);
    const double s = nv / (ev[0] - ev[k]);
#pragma unroll
    for (int r = 0; r < 3; r++)
#pragma unroll
      for (int j = 0; j < 9; j++) uc[r][j] += s * a[r] * fkl[j];
  }
  // Jc = u_c * cov_add (cov_add symmetric, upper triangle stored)
  double Jc[3][9];
#pragma unroll
  for (int r = 0; r < 3; r++)
#pragma unroll
    for (int j = 0; j < 9; j++) {
      double s = 0;
#pragma unroll
      for (int k = 0; k < 9; k++) {
        const int rr = k < j ? k : j, cc = k < j ? j : k;
        s += uc[r][k] * cv[rr * 9 - rr * (rr - 1) / 2 + (cc - rr)];
      }
      Jc[r][j] = s;
    }
  double *pl = m.nplane;
#pragma unroll
  for (int j = 0; j < 3; j++) { pl[(size_t)j * cp + id] = c[j]; pl[(size_t)(3 + j) * cp + id] = u[0][j]; }
  pl[(size_t)6 * cp + id] = (double)(float)ev[2];   // float radius (VM:89, VM:1387)
  double pv[36];
#pragma unroll
  for (int r = 0; r < 3; r++)
#pragma unroll
    for (int cidx = 0; cidx < 3; cidx++) {
      double s = 0;
#pragma unroll
      for (int k = 0; k < 9; k++) s += Jc[r][k] * uc[cidx][k];
      pv[r * 6 + cidx] = s;                                // Jc * u_c^T
      const double jn = nv * Jc[r][6 + cidx];
      pv[r * 6 + 3 + cidx] = jn; pv[(3 + cidx) * 6 + r] = jn;   // Jc_N and its transpose
      const int rr = 6 + (r < cidx ? r : cidx), cc = 6 + (r < cidx ? cidx : r);
      pv[(3 + r) * 6 + 3 + cidx] = nv * nv * cv[rr * 9 - rr * (rr - 1) / 2 + (cc - rr)];
    }
#pragma unroll
  for (int k = 0; k < 36; k++) pl[(size_t)(7 + k) * cp + id] = pv[k];
}

// One thread per leaf: OctoTree::margi leaf branch VM:1468-1584 with mgsize = 1.
__device__ __forceinline__ int margi_leaf_body(const MapView &m, const MapParams &P, const FactorView &f, int nfac, int win_count, int epoch, int id, const int *smp);
__global__ __launch_bounds__(256) void k_margi_leaf(MapView m, MapParams P, FactorView f, int nfac, int win_count, int epoch) {
  __shared__ int wtake[4], wcnt[4], bases[3];
  __shared__ int smp[VBA_MAX_WIN];                       // the ring map: indexed per lane below (a kernel argument cannot be)
  if (threadIdx.x < VBA_MAX_WIN) smp[threadIdx.x] = P.mp[threadIdx.x];
  __syncthreads();
  const int id = blockIdx.x * blockDim.x + threadIdx.x;
  const int nn = m.cnt[CNT_NODES] < m.cap ? m.cnt[CNT_NODES] : m.cap;
  // > 0: the leaf's oldest frame (that many points) joins point_fix.  The pool range, the block id and the place on the work list
  // of k_margi_take are reserved here with ONE returning atomic each per workgroup (per leaf they serialised: ~10 ns each in L2)
  const int count = id < nn ? margi_leaf_body(m, P, f, nfac, win_count, epoch, id, smp) : 0;
  const bool take = count > 0;
  const unsigned long long mask = __ballot(take);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  int incl = count;
  for (int off = 1; off < 64; off <<= 1) { const int v = __shfl_up(incl, off, 64); if (lane >= off) incl += v; }
  if (lane == 63) wcnt[wave] = incl;
  if (lane == 0) wtake[wave] = __popcll(mask);
  __syncthreads();
  if (threadIdx.x == 0) {
    int nt = 0, nc = 0;
    for (int w = 0; w < 4; w++) { const int a = wtake[w], b = wcnt[w]; wtake[w] = nt; wcnt[w] = nc; nt += a; nc += b; }
    bases[0] = nt ? atomicAdd(&m.cnt[CNT_TAKE], nt) : 0;
    bases[1] = nt ? atomicAdd(&m.cnt[CNT_FBLK], nt) : 0;
    bases[2] = nc ? atomicAdd(&m.cnt[CNT_FIX], nc) : 0;
  }
  __syncthreads();
  if (take) {
    const int r = wtake[wave] + __popcll(mask & ((1ull << lane) - 1ull));
    const int qb = bases[2] + wcnt[wave] + incl - count;
    const int blk = bases[1] + r;
    m.nsl[bases[0] + r] = id; m.wl[bases[0] + r] = qb; m.wlb[bases[0] + r] = count;
    if (qb + count > m.cap_fix || blk >= m.cap_fix) { atomicMax(&m.cnt[CNT_OVERFLOW], 3); m.wlb[bases[0] + r] = 0; }
    else {                                                 // link the block to the leaf's chain (this thread owns the leaf)
      m.fb_base[blk] = qb; m.fb_len[blk] = count; m.fb_next[blk] = -1;
      const int tail = m.nfb_tail[id];
      if (tail < 0) m.nfb_head[id] = blk; else m.fb_next[tail] = blk;
      m.nfb_tail[id] = blk;
    }
  }
}
__device__ __forceinline__ int margi_leaf_body(const MapView &m, const MapParams &P, const FactorView &f, int nfac, int win_count, int epoch, int id, const int *smp) {
  int take = 0;
  if (m.nstate[id] != 0 || m.nlayer[id] < 0) return 0;        // internal node, or freed storage
  if (slide_count(m, P) < P.thread_num) return 0;              // VS:1616-1617
  if (m.f_slide[m.nroot[id]] == 0) return 0;
  if (!m.f_exist[id] || !m.f_sw[id]) return 0;                // VM:1471-1472
  const size_t cp = (size_t)m.cap, W = (size_t)P.W, vs = (size_t)f.vs;
  double add[10], fix[10], pw0[10], ev[3], U[9], lc[10];
  for (int k = 0; k < 10; k++) fix[k] = nfix_at(m, k, id);
  for (int k = 0; k < 3; k++) ev[k] = neval_at(m, k, id);
  for (int k = 0; k < 9; k++) U[k] = nevec_at(m, k, id);
  for (int k = 0; k < 10; k++) pw0[k] = 0.0;
  const int opt = m.nopt[id];
  if (opt >= nfac) { atomicMax(&m.cnt[CNT_OVERFLOW], 2); return 0; }     // VM:1488-1492 "Error: opt_state"
  const int slot0 = P.mp[0];
  if (opt >= 0) {                                           // VM:1495-1509
    for (int k = 0; k < 10; k++) add[k] = f.pcr[(size_t)k * vs + opt];
    for (int k = 0; k < 3; k++) ev[k] = f.eigval[(size_t)k * vs + opt];
    for (int k = 0; k < 9; k++) U[k] = f.eigvec[(size_t)k * vs + opt];
    m.nopt[id] = -1;
    for (int k = 0; k < 10; k++) lc[k] = nlc_at(m, W, k, slot0, id);
    if (lc[9] != 0.0) cluster_transform_dev(lc, m.poses, pw0);
  } else {                                                  // VM:1510-1529
    for (int k = 0; k < 10; k++) add[k] = fix[k];
    // which frames hold points: the N column of every slot in ONE memory trip (unrolled loads), then a rolled loop over the occupied
    // frames only (unrolled over 16 frames the exact-order transform was 2400 f64 operations of straight-line code per thread)
    unsigned int occm = 0;
    {
      double nn[VBA_MAX_WIN_DEV];
#pragma unroll
      for (int i = 0; i < VBA_MAX_WIN_DEV; i++) nn[i] = (i < win_count) ? nlc_at(m, W, 9, P.mp[i < win_count ? i : 0], id) : 0.0;
#pragma unroll
      for (int i = 0; i < VBA_MAX_WIN_DEV; i++) occm |= (nn[i] != 0.0) ? (1u << i) : 0u;
    }
#pragma unroll 1
    while (occm) {
      const int i = __ffs((int)occm) - 1;
      occm &= occm - 1;
      const int slot = smp[i];
      for (int k = 0; k < 10; k++) lc[k] = nlc_at(m, W, k, slot, id);
      double t[10];
      cluster_transform_dev(lc, m.poses + 12 * i, t);
      for (int k = 0; k < 10; k++) add[k] += t[k];
      if (i == 0) for (int k = 0; k < 10; k++) pw0[k] = t[k];
    }
    if (m.f_plane[id]) {
      const double N = add[9], b0 = add[6] / N, b1 = add[7] / N, b2 = add[8] / N;
      eig3_sym_dev(add[0] / N - b0 * b0, add[1] / N - b1 * b0, add[2] / N - b2 * b0, add[3] / N - b1 * b1, add[4] / N - b2 * b1, add[5] / N - b2 * b2,
                   ev[0], ev[1], ev[2], U);
    }
  }
  for (int k = 0; k < 3; k++) neval_at(m, k, id) = ev[k];
  for (int k = 0; k < 9; k++) nevec_at(m, k, id) = U[k];
  if (fix[9] < P.max_points && m.f_plane[id]) {             // VM:1532-1538
    const int last = m.nlast[id];
    if ((int)add[9] - last >= 5 || last <= 10) { plane_update_dev(m, id, add, ev, U); m.nlast[id] = (int)add[9]; }
  }
  if (fix[9] < P.max_points) {                              // VM:1541-1555
    if (pw0[9] != 0.0) {
      for (int k = 0; k < 10; k++) fix[k] += pw0[k];
      if (m.nlayer[id] < P.max_layer) { m.ntake[id] = epoch; take = (int)pw0[9]; }  // its frame-0 points (pcrs_local[mp[0]].N of them) move to the fixed pool (k_margi_take)
    }
  } else {                                                  // VM:1556-1566
    if (pw0[9] != 0.0) for (int k = 0; k < 10; k++) add[k] -= pw0[k];
    m.nclear[id] = epoch;                                   // PVec().swap(point_fix)
    m.nfb_head[id] = -1; m.nfb_tail[id] = -1;               // (entries it owns inside older blocks are released by k_margi_fixclear)
  }
  for (int k = 0; k < 10; k++) { nadd_at(m, k, id) = add[k]; nfix_at(m, k, id) = fix[k]; }
  for (int k = 0; k < 10; k++) nlc_at(m, W, k, slot0, id) = 0.0;   // VM:1569-1574
  m.f_exist[id] = (fix[9] >= add[9]) ? 0 : 1;               // VM:1577-1580
  return take;
}

// Frame-0 points of leaves that still collect fixed points -> pool, in world coordinates (VM:1549-1553), in the leaf's scan order:
// one wave per such leaf X.  X's frame-0 points are the entries with pnode == X of the segment slot mp[0] gave to X (or to the ancestor
// that was the leaf when that scan was inserted).  They become ONE block of consecutive pool entries appended to X's chain, so
// point_fix of the reference — older blocks first, scan order inside a block — can be replayed in order by the next fix_divide.
// A leaf's frame-0 segment holds a handful of points: a GROUP OF 16 LANES per leaf, four leaves per wave (the pass is a chain of
// dependent memory trips per leaf: what counts is how many leaves are in flight).
__global__ __launch_bounds__(64) void k_margi_take(MapView m, MapParams P, int has_var) {
  const int lane = threadIdx.x, gl = lane & 15, gsh = lane & 48;                 // lane in the group, first lane of the group
  const int ntake = m.cnt[CNT_TAKE];
  const size_t mpz = (size_t)m.max_pts, W = (size_t)P.W, cp = (size_t)m.cap, cf = (size_t)m.cap_fix;
  const int slot = P.mp[0];
  const int *pleaf = m.pleaf + (size_t)slot * mpz;
  for (int s = blockIdx.x * 4 + (lane >> 4); s < ntake; s += gridDim.x * 4) {
    const int X = m.nsl[s], qb = m.wl[s], count = m.wlb[s];      // pool range and chain block were reserved by k_margi_leaf
    if (count == 0) continue;
    int anc = X;
    while (anc >= 0 && m.nseg_b[(size_t)slot * cp + anc] == m.nseg_a[(size_t)slot * cp + anc]) anc = m.nparent[anc];
    if (anc < 0) continue;
    const int start = m.nseg_a[(size_t)slot * cp + anc], end = m.nseg_b[(size_t)slot * cp + anc];
    int off = 0;
    for (int c0 = start; c0 < end; c0 += 16) {
      const int i = c0 + gl;
      const size_t p = (size_t)(i < end ? i : start);          // position in the slot's ordered storage
      const bool mine = i < end && pleaf[p] == X;
      const unsigned int mk = (unsigned int)((__ballot(mine) >> gsh) & 0xFFFFull);   // the group's 16 bits (its lanes run in step)
      if (mine) {
        const int o = off + __popc(mk & ((1u << gl) - 1u));
        if (o < count) {
          const int q = qb + o;
          const double *sp3 = m.sx + ((size_t)slot * mpz + p) * 3;
          const double bx = sp3[0], by = sp3[1], bz = sp3[2];
          double wx, wy, wz;
          world_point(m.poses, bx, by, bz, wx, wy, wz);                    // pv.pnt = x_buf[0].R * pv.pnt + x_buf[0].p  VM:1551
          m.fx[(size_t)q * 3] = wx; m.fx[(size_t)q * 3 + 1] = wy; m.fx[(size_t)q * 3 + 2] = wz;
          for (int k = 0; k < 9; k++) m.fvar[(size_t)q * 9 + k] = has_var ? m.svar[((size_t)slot * mpz + p) * 9 + k] : 0.0;
          m.fnode[q] = X;
        }
      }
      off += __popc(mk);
    }
    // (a leaf at layer < max_layer keeps every raw point, so the segment holds exactly pcrs_local[mp[0]].N entries of X;
    //  should it hold fewer, the rest of the reserved range stays unowned)
    for (int o = off + gl; o < count; o += 16) m.fnode[qb + o] = -1;
  }
}
// slot mp[0] is emptied (VM:1569-1574: points[mp[0]].clear())
__global__ __launch_bounds__(1024) void k_margi_points(MapView m, MapParams P) {
  const int p = blockIdx.x * 1024 + threadIdx.x;
  if (p < m.max_pts) m.pleaf[(size_t)P.mp[0] * m.max_pts + p] = -1;
}
__global__ void k_margi_fixclear(MapView m, int epoch) {
  const int q = blockIdx.x * blockDim.x + threadIdx.x;
  const int nf = m.cnt[CNT_FIX] < m.cap_fix ? m.cnt[CNT_FIX] : m.cap_fix;
  if (q >= nf) return;
  const int node = m.fnode[q];
  if (node >= 0 && m.nclear[node] == epoch) m.fnode[q] = -1;
}
// Internal nodes, bottom-up: isexist = OR(children)  VM:1585-1597
__global__ void k_margi_up(MapView m, MapParams P, int L) {
  const int id = blockIdx.x * blockDim.x + threadIdx.x;
  const int nn = m.cnt[CNT_NODES] < m.cap ? m.cnt[CNT_NODES] : m.cap;
  if (id >= nn) return;
  if (m.nlayer[id] != L || m.nstate[id] != 1) return;
  if (slide_count(m, P) < P.thread_num || m.f_slide[m.nroot[id]] == 0) return;
  unsigned char e = 0;
  const int base = m.nchild[id];
  for (int o = 0; o < 8; o++) e |= m.f_exist[base + o];
  m.f_exist[id] = e;
}
// Roots: jour stamp (VS:1628); roots with !isexist leave the sliding map (VS:1665-1674)
__global__ void k_margi_roots(MapView m, MapParams P, double jour, int epoch, int n_slide_before) {
  const int id = blockIdx.x * blockDim.x + threadIdx.x;
  const int nn = m.cnt[CNT_NODES] < m.cap ? m.cnt[CNT_NODES] : m.cap;
  if (id >= nn) return;
  if (m.nlayer[id] != 0 || m.f_slide[id] == 0) return;
  if (n_slide_before < P.thread_num) return;
  m.njour[id] = jour;
  if (!m.f_exist[id]) { m.f_slide[id] = 0; atomicSub(&m.cnt[CNT_SLIDE], 1); m.ndead[id] = epoch; }
}
// clear_slwd (VM:1856-1880) over the subtrees of the removed roots
__global__ void k_margi_clear_nodes(MapView m, MapParams P, int epoch) {
  const int id = blockIdx.x * blockDim.x + threadIdx.x;
  const int nn = m.cnt[CNT_NODES] < m.cap ? m.cnt[CNT_NODES] : m.cap;
  if (id >= nn) return;
  if (m.nlayer[id] < 0 || m.ndead[m.nroot[id]] != epoch || !m.f_sw[id]) return;
  const size_t cp = (size_t)m.cap;
  for (int k = 0; k < 10 * P.W; k++) m.nlc[(size_t)id * 10 * P.W + k] = 0.0;
  m.f_sw[id] = 0;
}
__global__ void k_margi_clear_points(MapView m, MapParams P, int epoch) {
  const int p = blockIdx.x * blockDim.x + threadIdx.x;
  const int slot = blockIdx.y;
  if (p >= m.max_pts) return;
  int *pn = m.pleaf + (size_t)slot * m.max_pts + p;
  const int node = *pn;
  if (node >= 0 && m.ndead[m.nroot[node]] == epoch) *pn = -1;
}

// ------------------------------------------------------------------------------------------------ pruning
// "release the features not used for a long time" (voxelslam.cpp:1800-1823): roots with int(jour - root.jour) >= dist
// leave surf_map together with their subtrees.  The hash slot becomes a tombstone; node storage is not recycled (a
// later compaction pass can do that), the subtree is only made unreachable and its fixed points are dropped.
__global__ void k_prune_roots(MapView m, double jour, int dist, int epoch) {
  const unsigned int h = blockIdx.x * blockDim.x + threadIdx.x;
  if (h > m.hmask) return;
  const unsigned long long key = m.hkeys[h];
  if (key == KEY_EMPTY || key == KEY_TOMB) return;
  const int root = m.hvals[h];
  if (root < 0) return;
  const int dis = (int)(jour - m.njour[root]);
  if (dis < dist) return;
  m.hkeys[h] = KEY_TOMB; m.hvals[h] = -1;
  m.ndead[root] = epoch;
  atomicSub(&m.cnt[CNT_ROOTS], 1);
  if (m.f_slide[root]) { m.f_slide[root] = 0; atomicSub(&m.cnt[CNT_SLIDE], 1); }
}
// Freed nodes go onto the free stacks: an internal node hands back its child block, a root itself.  (Every node of a dead
// subtree is visited: children are freed by their parent, so a child never pushes itself.)
__global__ void k_prune_nodes(MapView m, int epoch) {
  const int id = blockIdx.x * blockDim.x + threadIdx.x;
  const int nn = m.cnt[CNT_NODES] < m.cap ? m.cnt[CNT_NODES] : m.cap;
  if (id >= nn) return;
  if (m.nlayer[id] < 0 || m.ndead[m.nroot[id]] != epoch) return;
  if (m.nstate[id] == 1) { const int k = atomicAdd(&m.cnt[CNT_FREE_BLOCKS], 1); m.nfree_blk[k] = m.nchild[id]; }
  if (m.nroot[id] == id) { const int k = atomicAdd(&m.cnt[CNT_FREE_ROOTS], 1); m.nfree_root[k] = id; }
  else m.nlayer[id] = -1;                       // children first become unreachable ...
}
__global__ void k_prune_finish(MapView m, int epoch) {
  const int id = blockIdx.x * blockDim.x + threadIdx.x;
  const int nn = m.cnt[CNT_NODES] < m.cap ? m.cnt[CNT_NODES] : m.cap;
  if (id >= nn) return;
  if (m.nroot[id] == id && m.ndead[id] == epoch && m.nlayer[id] >= 0) m.nlayer[id] = -1;   // ... then the roots themselves
}
// The accumulators of a freed node are zeroed so that its next owner starts like fresh storage (which is zero-filled at
// allocation): one thread per (node, row) of [pcr_add 10 | pcr_fix 10 | cov_add 45 | eig 12 | plane 43 | local clusters 10 W].
__global__ void k_prune_zero(MapView m, int W, int epoch) {
  const int id = blockIdx.x * blockDim.x + threadIdx.x;
  const int nn = m.cnt[CNT_NODES] < m.cap ? m.cnt[CNT_NODES] : m.cap;
  if (id >= nn) return;
  if (m.nlayer[id] >= 0) return;
  if (m.ndead[m.nroot[id]] != epoch) return;    // freed by an earlier prune (zeroed then)
  const size_t cp = (size_t)m.cap;
  int r = blockIdx.y;
  if (r == 0) {   // a stale flag must not make a later pass take the freed node for a live leaf (its nroot may be re-owned by then)
    m.f_exist[id] = 0; m.f_sw[id] = 0; m.f_plane[id] = 0; m.f_touched[id] = 0; m.nstate[id] = 0; m.nopt[id] = -1; m.nchild[id] = -1;
  }
  if (r < 10) { nadd_at(m, r, id) = 0.0; return; }
  r -= 10;
  if (r < 10) { nfix_at(m, r, id) = 0.0; return; }
  r -= 10;
  if (r < 45) { ncov_at(m, r, id) = 0.0; return; }
  r -= 45;
  if (r < 3) { neval_at(m, r, id) = 0.0; return; }
  r -= 3;
  if (r < 9) { nevec_at(m, r, id) = 0.0; return; }
  r -= 9;
  if (r < 43) { m.nplane[(size_t)r * cp + id] = 0.0; return; }
  r -= 43;
  if (r < 10 * W) { m.nlc[(size_t)id * 10 * W + r] = 0.0; if (r < W) { m.nseg_a[(size_t)r * cp + id] = 0; m.nseg_b[(size_t)r * cp + id] = 0; } }
}
__global__ void k_prune_fix(MapView m) {
  const int q = blockIdx.x * blockDim.x + threadIdx.x;
  const int nf = m.cnt[CNT_FIX] < m.cap_fix ? m.cnt[CNT_FIX] : m.cap_fix;
  if (q >= nf) return;
  const int node = m.fnode[q];
  if (node >= 0 && m.nlayer[node] < 0) m.fnode[q] = -1;
}

// ------------------------------------------------------------------------------------------------ odometry scan-to-map
// One EKF iteration's point loop of VOXEL_SLAM::lio_state_estimation (voxelslam.cpp:1004-1052) with match()
// (voxel_map.hpp:2167-2205) and OctoTree::match (voxel_map.hpp:1649-1721): per point world covariance, root lookup
// (read-only probe), octant descent, the two 3-sigma gates, then the weighted normal equations
//   HTH (6x6 sym, 21) | HTz (6) | nnt (3x3 sym, 6) | match_num   = 34 sums per workgroup.
// The reference's per-point cache octos[i] (voxelslam.cpp:1020) only short-cuts the lookup; the full lookup is done here.
struct OdomState { double R[9], t[3], rot_var[9], tsl_var[9]; };

__global__ __launch_bounds__(256) void k_odom_match(MapView m, MapParams P, OdomState X, int n, const double *__restrict__ pts,
                                                    const double *__restrict__ var, double *__restrict__ partial) {
  __shared__ double red[4][34];
  const int p = blockIdx.x * blockDim.x + threadIdx.x;
  const size_t cp = (size_t)m.cap;
  double acc[34];
#pragma unroll
  for (int k = 0; k < 34; k++) acc[k] = 0.0;
  if (p < n) {
    const double bx = pts[3 * (size_t)p], by = pts[3 * (size_t)p + 1], bz = pts[3 * (size_t)p + 2];
    const double *R = X.R;
    const double wx = R[0] * bx + R[1] * by + R[2] * bz + X.t[0], wy = R[3] * bx + R[4] * by + R[5] * bz + X.t[1], wz = R[6] * bx + R[7] * by + R[8] * bz + X.t[2];
    const long long kx = key_axis(wx, P.voxel_size), ky = key_axis(wy, P.voxel_size), kz = key_axis(wz, P.voxel_size);
    int node = -1;
    if (!(kx < -KEY_OFF || kx >= KEY_OFF || ky < -KEY_OFF || ky >= KEY_OFF || kz < -KEY_OFF || kz >= KEY_OFF)) {
      const unsigned long long key = pack_key(kx, ky, kz);
      unsigned int h = (unsigned int)((key * 0x9E3779B97F4A7C15ull) >> 32) & m.hmask;
      for (unsigned int probe = 0; probe <= m.hmask; probe++) {
        const unsigned long long cur = m.hkeys[h];
        if (cur == key) { node = m.hvals[h]; break; }
        if (cur == KEY_EMPTY) break;
        h = (h + 1) & m.hmask;
      }
    }
    if (node >= 0) {
      while (m.nstate[node] == 1) node = m.nchild[node] + octant_of(m, node, wx, wy, wz);
      if (m.f_plane[node]) {
        const double *pl = m.nplane;
        const double cx = pl[0 * cp + node], cy = pl[1 * cp + node], cz = pl[2 * cp + node];
        const double nx = pl[3 * cp + node], ny = pl[4 * cp + node], nz = pl[5 * cp + node];
        const float radius = (float)pl[6 * cp + node];
        const double dx = wx - cx, dy = wy - cy, dz = wz - cz;
        const double resi = nx * dx + ny * dy + nz * dz;
        const float dis_to_plane = (float)fabs(resi);                                  // VM:1657
        const float dis_to_center = (float)(dx * dx + dy * dy + dz * dz);              // VM:1659
        const float range_dis = dis_to_center - dis_to_plane * dis_to_plane;           // VM:1661
        if (range_dis <= 9.0f * radius) {                                              // VM:1664
          // var_world = R var R^T + phat rot_var phat^T + tsl_var                        voxelslam.cpp:1009
          double v[9], RV[9], vw[9];
#pragma unroll
          for (int k = 0; k < 9; k++) v[k] = var[9 * (size_t)p + k];
#pragma unroll
          for (int r = 0; r < 3; r++)
#pragma unroll
            for (int c = 0; c < 3; c++) RV[3 * r + c] = (R[3 * r] * v[c] + R[3 * r + 1] * v[3 + c]) + R[3 * r + 2] * v[6 + c];
          const double ph[9] = {0, -bz, by, bz, 0, -bx, -by, bx, 0};
          double PR[9];
#pragma unroll
          for (int r = 0; r < 3; r++)
#pragma unroll
            for (int c = 0; c < 3; c++) PR[3 * r + c] = ph[3 * r] * X.rot_var[c] + ph[3 * r + 1] * X.rot_var[3 + c] + ph[3 * r + 2] * X.rot_var[6 + c];
#pragma unroll
          for (int r = 0; r < 3; r++)
#pragma unroll
            for (int c = 0; c < 3; c++)
              vw[3 * r + c] = (RV[3 * r] * R[3 * c] + RV[3 * r + 1] * R[3 * c + 1] + RV[3 * r + 2] * R[3 * c + 2]) +
                              (PR[3 * r] * ph[3 * c] + PR[3 * r + 1] * ph[3 * c + 1] + PR[3 * r + 2] * ph[3 * c + 2]) + X.tsl_var[3 * r + c];
          // sigma_l = J plane_var J^T + n^T var_world n,  J = [wld - center, -normal]      VM:1667-1672
          const double J[6] = {dx, dy, dz, -nx, -ny, -nz};
          double sigma_l = 0.0;
#pragma unroll
          for (int r = 0; r < 6; r++) {
            double sr = 0.0;
#pragma unroll
            for (int c = 0; c < 6; c++) sr += pl[(size_t)(7 + 6 * r + c) * cp + node] * J[c];
            sigma_l += J[r] * sr;
          }
          const double nvec[3] = {nx, ny, nz};
#pragma unroll
          for (int r = 0; r < 3; r++) sigma_l += nvec[r] * (vw[3 * r] * nx + vw[3 * r + 1] * ny + vw[3 * r + 2] * nz);
          if ((double)dis_to_plane < 3.0 * sqrt(sigma_l)) {                             // VM:1675
            const double Rinv = 1.0 / (0.0005 + sigma_l);                               // voxelslam.cpp:1033
            // jac = [phat R^T n ; n]                                                     voxelslam.cpp:1039-1040
            const double a0 = R[0] * nx + R[3] * ny + R[6] * nz, a1 = R[1] * nx + R[4] * ny + R[7] * nz, a2 = R[2] * nx + R[5] * ny + R[8] * nz;
            const double jac[6] = {by * a2 - bz * a1, bz * a0 - bx * a2, bx * a1 - by * a0, nx, ny, nz};
            int idx = 0;
#pragma unroll
            for (int r = 0; r < 6; r++)
#pragma unroll
              for (int c = r; c < 6; c++) acc[idx++] = Rinv * jac[r] * jac[c];           // HTH upper triangle
#pragma unroll
            for (int r = 0; r < 6; r++) acc[21 + r] = -Rinv * jac[r] * resi;             // HTz
            acc[27] = nx * nx; acc[28] = nx * ny; acc[29] = nx * nz; acc[30] = ny * ny; acc[31] = ny * nz; acc[32] = nz * nz;   // nnt
            acc[33] = 1.0;                                                               // match_num
          }
        }
      }
    }
  }
#pragma unroll
  for (int k = 0; k < 34; k++) {
    double x = acc[k];
    for (int s = 32; s >= 1; s >>= 1) x += __shfl_xor(x, s, 64);
    acc[k] = x;
  }
  if ((threadIdx.x & 63) == 0)
#pragma unroll
    for (int k = 0; k < 34; k++) red[threadIdx.x >> 6][k] = acc[k];
  __syncthreads();
  if (threadIdx.x < 34) partial[(size_t)blockIdx.x * 34 + threadIdx.x] = (red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x]);
}

// ------------------------------------------------------------------------------------------------ dumps
__global__ void k_dump_leaves(MapView m, double *out, int max_leaves) {
  const int id = blockIdx.x * blockDim.x + threadIdx.x;
  const int nn = m.cnt[CNT_NODES] < m.cap ? m.cnt[CNT_NODES] : m.cap;
  if (id >= nn) return;
  if (m.nstate[id] != 0 || m.nlayer[id] < 0) return;  // internal node, or pruned
  if (m.nlayer[id] > 0 && !m.f_touched[id]) return;   // never-touched octants do not exist in the reference tree
  const int i = atomicAdd(&m.cnt[CNT_LEAVES], 1);
  if (i >= max_leaves) return;
  const size_t cp = (size_t)m.cap;
  double *o = out + (size_t)i * 39;
  long long kx, ky, kz;
  unpack_key(m.nkey[id], kx, ky, kz);
  o[0] = (double)kx; o[1] = (double)ky; o[2] = (double)kz; o[3] = m.nlayer[id]; o[4] = m.npath[id];
  o[5] = nadd_at(m, 9, id); o[6] = nfix_at(m, 9, id); o[7] = m.f_plane[id]; o[8] = m.f_exist[id]; o[9] = m.nopt[id];
  for (int k = 0; k < 3; k++) o[10 + k] = neval_at(m, k, id);
  for (int k = 0; k < 9; k++) o[13 + k] = nevec_at(m, k, id);
  for (int k = 0; k < 10; k++) o[22 + k] = nadd_at(m, k, id);
  for (int k = 0; k < 7; k++) o[32 + k] = m.nplane[(size_t)k * cp + id];
}

// plane.plane_var (6x6, VM:1356-1383) and cov_add (9x9 symmetric, upper triangle; VM:106-121, 1138-1140) of every leaf, keyed like
// k_dump_leaves: [kx,ky,kz, layer, path, plane_var(36 row-major), cov_add upper triangle (45)] = 86 doubles.
__global__ void k_dump_plane_var(MapView m, double *out, int max_leaves) {
  const int id = blockIdx.x * blockDim.x + threadIdx.x;
  const int nn = m.cnt[CNT_NODES] < m.cap ? m.cnt[CNT_NODES] : m.cap;
  if (id >= nn) return;
  if (m.nstate[id] != 0 || m.nlayer[id] < 0) return;
  if (m.nlayer[id] > 0 && !m.f_touched[id]) return;
  const int i = atomicAdd(&m.cnt[CNT_LEAVES], 1);
  if (i >= max_leaves) return;
  const size_t cp = (size_t)m.cap;
  double *o = out + (size_t)i * 86;
  long long kx, ky, kz;
  unpack_key(m.nkey[id], kx, ky, kz);
  o[0] = (double)kx; o[1] = (double)ky; o[2] = (double)kz; o[3] = m.nlayer[id]; o[4] = m.npath[id];
  for (int k = 0; k < 36; k++) o[5 + k] = m.nplane[(size_t)(7 + k) * cp + id];
  for (int k = 0; k < 45; k++) o[41 + k] = ncov_at(m, k, id);
}

__global__ void k_fill_u64(unsigned long long *p, unsigned long long v, size_t n) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = v;
}
__global__ void k_rehash(const unsigned long long *okeys, const int *ovals, unsigned int omask, unsigned long long *nkeys, int *nvals, unsigned int nmask) {
  const unsigned int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i > omask) return;
  const unsigned long long key = okeys[i];
  if (key == KEY_EMPTY || key == KEY_TOMB) return;
  unsigned int h = (unsigned int)((key * 0x9E3779B97F4A7C15ull) >> 32) & nmask;
  while (true) {
    if (atomicCAS(&nkeys[h], KEY_EMPTY, key) == KEY_EMPTY) { nvals[h] = ovals[i]; return; }
    h = (h + 1) & nmask;
  }
}
// var_init (voxelslam.hpp:210-234) = calcBodyVar (voxelslam.hpp:180-200) + extrinsic: one thread per point.
// DEG2RAD is PCL's macro ((x) * 0.017453293), see oracle/map_oracle.hpp.
__global__ void k_var_init(int n, const double *__restrict__ pin, double *__restrict__ pout, double *__restrict__ var, const double *__restrict__ ext,
                           float range_inc, float degree_inc) {
  const int p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= n) return;
  double x = pin[3 * (size_t)p], y = pin[3 * (size_t)p + 1], z = pin[3 * (size_t)p + 2];
  if (z == 0) z = 0.0001;
  const float range = (float)sqrt(x * x + y * y + z * z);
  const float range_var = range_inc * range_inc;
  const double sn = sin((degree_inc) * 0.017453293), dv = sn * sn;
  const double nrm = sqrt(x * x + y * y + z * z);
  const double d0 = x / nrm, d1 = y / nrm, d2 = z / nrm;
  double b1x = 1, b1y = 1, b1z = -(d0 + d1) / d2;
  const double n1 = sqrt(b1x * b1x + b1y * b1y + b1z * b1z);
  b1x /= n1; b1y /= n1; b1z /= n1;
  double b2x = b1y * d2 - b1z * d1, b2y = b1z * d0 - b1x * d2, b2z = b1x * d1 - b1y * d0;   // b1 x direction
  const double n2 = sqrt(b2x * b2x + b2y * b2y + b2z * b2z);
  b2x /= n2; b2y /= n2; b2z /= n2;
  // A = range * hat(direction) * [b1 b2]
  const double r = (double)range;
  const double a1x = r * (d1 * b1z - d2 * b1y), a1y = r * (d2 * b1x - d0 * b1z), a1z = r * (d0 * b1y - d1 * b1x);
  const double a2x = r * (d1 * b2z - d2 * b2y), a2y = r * (d2 * b2x - d0 * b2z), a2z = r * (d0 * b2y - d1 * b2x);
  const double rv = (double)range_var;
  const double d[3] = {d0, d1, d2}, a1[3] = {a1x, a1y, a1z}, a2[3] = {a2x, a2y, a2z};
  double vb[9];
#pragma unroll
  for (int i = 0; i < 3; i++)
#pragma unroll
    for (int j = 0; j < 3; j++) vb[3 * i + j] = d[i] * rv * d[j] + (a1[i] * dv * a1[j] + a2[i] * dv * a2[j]);
  // extrinsic: pnt = R p + t ; var = R var R^T
  const double *R = ext;
  pout[3 * (size_t)p] = R[0] * x + R[1] * y + R[2] * z + R[9];
  pout[3 * (size_t)p + 1] = R[3] * x + R[4] * y + R[5] * z + R[10];
  pout[3 * (size_t)p + 2] = R[6] * x + R[7] * y + R[8] * z + R[11];
  double RV[9];
#pragma unroll
  for (int i = 0; i < 3; i++)
#pragma unroll
    for (int j = 0; j < 3; j++) RV[3 * i + j] = R[3 * i] * vb[j] + R[3 * i + 1] * vb[3 + j] + R[3 * i + 2] * vb[6 + j];
#pragma unroll
  for (int i = 0; i < 3; i++)
#pragma unroll
    for (int j = 0; j < 3; j++) var[9 * (size_t)p + 3 * i + j] = RV[3 * i] * R[3 * j] + RV[3 * i + 1] * R[3 * j + 1] + RV[3 * i + 2] * R[3 * j + 2];
}

// pvec_update (voxelslam.hpp:242-265) fused into the staging of a scan: var_world = R var R^T + phat rot_var phat^T + tsl_var
// (pw = R p + t is recomputed by the insert kernels).  cov6 = [rot_var(9) | tsl_var(9)].
__global__ __launch_bounds__(256) void k_scan_to_soa_pvec_update(MapView m, int W, int slot, int n, const double *pts, const double *var, const double *pose, const double *cov6) {
#pragma clang fp contract(off)      // the reference's operation order, separately rounded (cov_add sums these values in order: see ord_terms)
  // A thread owns a point, but 72-byte records read with one 8-byte load per lane touch 36 lines per instruction: the workgroup moves
  // its 256 records through LDS with consecutive lanes on consecutive doubles (rows of 9 / 3 doubles: conflict-free for the owner).
  __shared__ double sv[256 * 9], sp[256 * 3];
  const int tid = threadIdx.x, base = blockIdx.x * 256, p = base + tid;
  if (p == 0) { m.cnt[CNT_NEWSLOTS] = 0; m.cnt[CNT_TOUCH] = 0; m.cnt[CNT_WL] = 0; m.cnt[CNT_WLB] = 0; m.cnt[CNT_CURSOR] = 0; }   // the insert's counters (no kernels of their own)
  const int cnt = n - base < 256 ? n - base : 256;
  for (int i = tid; i < cnt * 9; i += 256) sv[i] = var[(size_t)base * 9 + i];
  for (int i = tid; i < cnt * 3; i += 256) sp[i] = pts[(size_t)base * 3 + i];
  __syncthreads();
  const size_t mpz = (size_t)m.max_pts;
  if (tid < cnt) {
    const double bx = sp[tid * 3], by = sp[tid * 3 + 1], bz = sp[tid * 3 + 2];
    const double *R = pose;
    double v[9], RV[9], PR[9];
#pragma unroll
    for (int k = 0; k < 9; k++) v[k] = sv[tid * 9 + k];
#pragma unroll
    for (int r = 0; r < 3; r++)
#pragma unroll
      for (int c = 0; c < 3; c++) RV[3 * r + c] = (R[3 * r] * v[c] + R[3 * r + 1] * v[3 + c]) + R[3 * r + 2] * v[6 + c];
    const double ph[9] = {0, -bz, by, bz, 0, -bx, -by, bx, 0};
#pragma unroll
    for (int r = 0; r < 3; r++)
#pragma unroll
      for (int c = 0; c < 3; c++) PR[3 * r + c] = (ph[3 * r] * cov6[c] + ph[3 * r + 1] * cov6[3 + c]) + ph[3 * r + 2] * cov6[6 + c];
#pragma unroll
    for (int r = 0; r < 3; r++)
#pragma unroll
      for (int c = 0; c < 3; c++)
        sv[tid * 9 + 3 * r + c] = (((RV[3 * r] * R[3 * c] + RV[3 * r + 1] * R[3 * c + 1]) + RV[3 * r + 2] * R[3 * c + 2]) +
                                   ((PR[3 * r] * ph[3 * c] + PR[3 * r + 1] * ph[3 * c + 1]) + PR[3 * r + 2] * ph[3 * c + 2])) + cov6[9 + 3 * r + c];
  }
  __syncthreads();
  double *ov = m.pvar + ((size_t)slot * mpz + base) * 9, *op = m.px + ((size_t)slot * mpz + base) * 3;
  for (int i = tid; i < cnt * 9; i += 256) ov[i] = sv[i];
  for (int i = tid; i < cnt * 3; i += 256) op[i] = sp[i];
}

// host layout [n][3] / [n][9] -> the scan slot's staging arrays (same AoS layout)
__global__ void k_scan_to_soa(MapView m, int W, int slot, int n, const double *pts, const double *var) {
  const int p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p == 0) { m.cnt[CNT_NEWSLOTS] = 0; m.cnt[CNT_TOUCH] = 0; m.cnt[CNT_WL] = 0; m.cnt[CNT_WLB] = 0; m.cnt[CNT_CURSOR] = 0; }   // the insert's counters (no kernels of their own)
  if (p >= n) return;
  const size_t mpz = (size_t)m.max_pts;
  for (int k = 0; k < 3; k++) m.px[((size_t)slot * mpz + p) * 3 + k] = pts[(size_t)p * 3 + k];
  if (var) for (int k = 0; k < 9; k++) m.pvar[((size_t)slot * mpz + p) * 9 + k] = var[(size_t)p * 9 + k];
}
__global__ void k_fix_to_soa(MapView m, int base, int n, const double *pts) {
  const int p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= n) return;
  const size_t cf = (size_t)m.cap_fix;
  for (int k = 0; k < 3; k++) m.fx[(size_t)(base + p) * 3 + k] = pts[(size_t)p * 3 + k];
  for (int k = 0; k < 9; k++) m.fvar[(size_t)(base + p) * 9 + k] = 0.0;   // push_fix_novar: no covariance
  m.fnode[base + p] = -1;
}

// ================================================================================================ host side
__global__ void k_set_counter(int *cnt, int which, int val) { cnt[which] = val; }
__global__ void k_set_counter2(int *cnt, int a, int va, int b, int vb) { cnt[a] = va; cnt[b] = vb; }
// per recut level: node snapshot, empty split list; the first level clears the overflow flag (a full hash table reported by an
// insertion, code 4, stays up for the read-back at the end of the pass), the last one the factor counter of k_extract_count
__global__ void k_recut_prep(int *cnt, int first, int last) {
  cnt[CNT_SNAP] = cnt[CNT_NODES]; cnt[CNT_SPLIT] = 0;
  if (first && cnt[CNT_OVERFLOW] != 4) cnt[CNT_OVERFLOW] = 0;
  if (last) cnt[CNT_FACTORS] = 0;
}
__global__ void k_copy_counter(int *cnt, int from, int to) { cnt[to] = cnt[from]; }
struct DevArr {  // a [rows][cap] device array that can grow its cap keeping [rows][used]
  void **slot; size_t elem, rows;
};

struct MapStore {
  vba_options opt;
  int rank = 0, n_ranks = 1;
  MapView v{};
  bool allocated = false;
  bool have_var = false;
  int mp[VBA_MAX_WIN];
  int npts[VBA_MAX_WIN];
  int epoch = 1, stamp = 1;
  unsigned int hcap = 0;
  int *h_cnt = nullptr;    // pinned
  // Inserts are enqueued without reading the counters back: the host keeps pessimistic upper bounds (every point may
  // create a root) and re-reads the true counters only when a bound would exceed a capacity.
  long long ub_nodes = 0, ub_roots = 0, ub_used = 0;   // ub_used: hash slots that are not EMPTY (live roots + tombstones)
  bool cnt_stale = false;
  double *h_pose_ring = nullptr; hipEvent_t pose_ev[8] = {nullptr}; int pose_next = 0;
  void *d_stage = nullptr; size_t stage_bytes = 0;
  void *d_sort_tmp = nullptr; size_t sort_tmp_bytes = 0; int sort_tmp_for = 0;   // rocPRIM scratch, sized for max_pts pairs
  // sharded map: SUM all-reduce of n doubles in HBM over the ranks, stream-ordered (set by the context); d_gc = its 2-double scratch
  std::function<int(double *, size_t)> allreduce;
  double *d_gc = nullptr;
};

inline void map_init(MapStore &s, const vba_options &o) {
  s.opt = o;
  for (int i = 0; i < VBA_MAX_WIN; i++) { s.mp[i] = i; s.npts[i] = 0; }   // VS:3158-3160
}

#define MAPCHK(expr)                                                                 \
  do {                                                                               \
    hipError_t _e = (expr);                                                          \
    if (_e != hipSuccess) { err = std::string(#expr) + ": " + hipGetErrorString(_e); return VBA_ERR_HIP; } \
  } while (0)

inline MapParams map_params(const MapStore &s) {
  MapParams P;
  P.W = s.opt.win_size; P.max_layer = s.opt.max_layer; P.max_points = s.opt.max_points; P.thread_num = s.opt.thread_num;
  P.voxel_size = s.opt.voxel_size; P.min_eigen_value = s.opt.min_eigen_value;
  for (int i = 0; i < 4; i++) { P.plane_thre[i] = s.opt.plane_eigen_value_thre[i]; P.min_point[i] = s.opt.min_point[i]; }
  for (int i = 0; i < VBA_MAX_WIN; i++) P.mp[i] = s.mp[i];
  P.rank = s.rank; P.n_ranks = s.n_ranks;
  return P;
}

inline std::vector<DevArr> node_arrays(MapView &v, int W) {
  return {
      {(void **)&v.nkey, 8, 1}, {(void **)&v.nroot, 4, 1}, {(void **)&v.nparent, 4, 1}, {(void **)&v.nchild, 4, 1}, {(void **)&v.npath, 4, 1},
      {(void **)&v.nopt, 4, 1}, {(void **)&v.nflist, 4, 1}, {(void **)&v.nfl2, 4, 1}, {(void **)&v.nfkey, 4, 1}, {(void **)&v.nlast, 4, 1}, {(void **)&v.nstamp, 4, 1}, {(void **)&v.nsplit, 4, 1}, {(void **)&v.ntake, 4, 1},
      {(void **)&v.nclear, 4, 1}, {(void **)&v.ndead, 4, 1}, {(void **)&v.nfree_root, 4, 1}, {(void **)&v.nfree_blk, 4, 1},
      {(void **)&v.nseg_a, 4, (size_t)W}, {(void **)&v.nseg_b, 4, (size_t)W}, {(void **)&v.nsl, 4, 1}, {(void **)&v.ncnt, 4, 1}, {(void **)&v.nfb_head, 4, 1}, {(void **)&v.nfb_tail, 4, 1}, {(void **)&v.nlayer, 1, 1}, {(void **)&v.nstate, 1, 1}, {(void **)&v.f_exist, 1, 1},
      {(void **)&v.f_sw, 1, 1}, {(void **)&v.f_plane, 1, 1}, {(void **)&v.f_touched, 1, 1}, {(void **)&v.f_slide, 4, 1}, {(void **)&v.nql, 4, 1},
      {(void **)&v.ncenter, 8, 3}, {(void **)&v.njour, 8, 1}, {(void **)&v.nadd, 80, 1}, {(void **)&v.nfix, 80, 1}, {(void **)&v.ncov, 360, 1},
      {(void **)&v.neval, 24, 1}, {(void **)&v.nevec, 72, 1}, {(void **)&v.nplane, 8, 43}, {(void **)&v.nlc, (size_t)80 * W, 1},
  };
}
inline std::vector<DevArr> scan_arrays(MapView &v, int W) {
  return {{(void **)&v.px, 24, (size_t)W}, {(void **)&v.pvar, 72, (size_t)W}, {(void **)&v.pnode, 4, (size_t)W}, {(void **)&v.phash, 4, 1}, {(void **)&v.newslots, 4, 1},
          {(void **)&v.perm, 4, (size_t)W}, {(void **)&v.sx, 24, (size_t)W}, {(void **)&v.svar, 72, (size_t)W}, {(void **)&v.pleaf, 4, (size_t)W}, {(void **)&v.skey_a, 4, 1}, {(void **)&v.skey_b, 4, 1}, {(void **)&v.sval_a, 4, 1}, {(void **)&v.sval_b, 4, 1}, {(void **)&v.wl, 4, 1}, {(void **)&v.wlb, 4, 1}, {(void **)&v.wl4, 16, 1}};
}
inline std::vector<DevArr> fix_arrays(MapView &v) {
  return {{(void **)&v.fx, 24, 1}, {(void **)&v.fvar, 72, 1}, {(void **)&v.fnode, 4, 1}, {(void **)&v.fb_base, 4, 1}, {(void **)&v.fb_len, 4, 1}, {(void **)&v.fb_next, 4, 1}};
}

// grow a family of [rows][cap] arrays from oldcap to newcap, keeping the first `used` columns; new space zero-filled
inline int grow_arrays(std::vector<DevArr> arrs, size_t oldcap, size_t newcap, size_t used, hipStream_t st, std::string &err) {
  for (auto &a : arrs) {
    void *np = nullptr;
    MAPCHK(hipMalloc(&np, a.elem * a.rows * newcap));
    MAPCHK(hipMemsetAsync(np, 0, a.elem * a.rows * newcap, st));
    if (*a.slot && used > 0)
      MAPCHK(hipMemcpy2DAsync(np, newcap * a.elem, *a.slot, oldcap * a.elem, used * a.elem, a.rows, hipMemcpyDeviceToDevice, st));
    MAPCHK(hipStreamSynchronize(st));
    if (*a.slot) hipFree(*a.slot);
    *a.slot = np;
  }
  return VBA_OK;
}

// device -> pinned host memory by a kernel: a D2H copy queued behind in-flight kernels completes much later (measured), and draining
// the stream first costs a second host round trip; a 64-thread kernel that stores through the host mapping needs one.
__global__ void k_words_to_host(const int *__restrict__ src, int *__restrict__ dst, int n) {
  for (int i = threadIdx.x + blockIdx.x * blockDim.x; i < n; i += blockDim.x * gridDim.x) dst[i] = src[i];
}
inline int map_read_counters(MapStore &s, hipStream_t st, std::string &err) {
  hipLaunchKernelGGL(k_words_to_host, dim3(1), dim3(64), 0, st, s.v.cnt, s.h_cnt, (int)CNT_N);
  MAPCHK(hipGetLastError());
  MAPCHK(hipStreamSynchronize(st));
  s.ub_nodes = s.h_cnt[CNT_NODES]; s.ub_roots = s.h_cnt[CNT_ROOTS]; s.ub_used = s.h_cnt[CNT_USED]; s.cnt_stale = false;
  return VBA_OK;
}

inline int map_hash_alloc(MapStore &s, unsigned int cap, hipStream_t st, std::string &err) {
  unsigned long long *nk = nullptr; int *nv = nullptr;
  MAPCHK(hipMalloc((void **)&nk, (size_t)cap * 8));
  MAPCHK(hipMalloc((void **)&nv, (size_t)cap * 4));
  hipLaunchKernelGGL(k_fill_u64, dim3(1024), dim3(256), 0, st, nk, KEY_EMPTY, (size_t)cap);
  MAPCHK(hipMemsetAsync(nv, 0xFF, (size_t)cap * 4, st));
  if (s.v.hkeys) {
    hipLaunchKernelGGL(k_rehash, dim3((s.hcap + 255) / 256), dim3(256), 0, st, s.v.hkeys, s.v.hvals, s.v.hmask, nk, nv, cap - 1);
    hipLaunchKernelGGL(k_copy_counter, dim3(1), dim3(1), 0, st, s.v.cnt, (int)CNT_ROOTS, (int)CNT_USED);   // the tombstones are gone
    MAPCHK(hipStreamSynchronize(st));
    hipFree(s.v.hkeys); hipFree(s.v.hvals);
    s.ub_used = s.ub_roots;
  }
  s.v.hkeys = nk; s.v.hvals = nv; s.v.hmask = cap - 1; s.hcap = cap;
  return VBA_OK;
}

inline int map_base(MapStore &s, hipStream_t st, std::string &err) {
  if (s.allocated) return VBA_OK;
  MAPCHK(hipMalloc((void **)&s.v.cnt, CNT_N * sizeof(int)));
  MAPCHK(hipMalloc((void **)&s.v.fhist, (size_t)EXTRACT_NB_MAX * sizeof(int)));
  MAPCHK(hipMemsetAsync(s.v.cnt, 0, CNT_N * sizeof(int), st));
  MAPCHK(hipMalloc((void **)&s.v.poses, VBA_MAX_WIN * 12 * sizeof(double)));
  MAPCHK(hipHostMalloc((void **)&s.h_cnt, CNT_N * sizeof(int) + 64, hipHostMallocDefault));
  std::memset(s.h_cnt, 0, CNT_N * sizeof(int) + 64);
  // initial root table: 2^20 slots, or (with the max_points_per_scan capacity hint) the power of two above 4x the hint
  unsigned int hc = 1u << 20;
  if (s.opt.max_points_per_scan) { hc = 1u << 10; while ((size_t)hc < 4 * s.opt.max_points_per_scan && hc < (1u << 30)) hc *= 2; }
  int st2 = map_hash_alloc(s, hc, st, err);
  if (st2) return st2;
  s.allocated = true;
  return VBA_OK;
}

inline int map_ensure(MapStore &s, hipStream_t st, size_t need_nodes, size_t need_pts, size_t need_fix, std::string &err) {
  const int W = s.opt.win_size;
  int rb = map_base(s, st, err);
  if (rb) return rb;
  // capacity hints of vba_options: taken at the first allocation of each array family
  if (s.v.cap == 0 && need_nodes > 0 && s.opt.max_map_nodes > need_nodes) need_nodes = s.opt.max_map_nodes;
  if (s.v.max_pts == 0 && need_pts > 0 && s.opt.max_points_per_scan > need_pts) need_pts = s.opt.max_points_per_scan;
  if (s.v.cap_fix == 0 && need_fix > 0 && s.opt.max_fix_points > need_fix) need_fix = s.opt.max_fix_points;
  if (need_nodes > (size_t)s.v.cap) {
    size_t nc = s.v.cap ? (size_t)s.v.cap : (size_t)1 << 18;
    while (nc < need_nodes) nc *= 2;
    int r = grow_arrays(node_arrays(s.v, W), (size_t)s.v.cap, nc, (size_t)s.v.cap, st, err);
    if (r) return r;
    s.v.cap = (int)nc;
  }
  if (need_pts > (size_t)s.v.max_pts) {
    size_t nc = s.v.max_pts ? (size_t)s.v.max_pts : (size_t)1 << 16;
    while (nc < need_pts) nc *= 2;
    int r = grow_arrays(scan_arrays(s.v, W), (size_t)s.v.max_pts, nc, (size_t)s.v.max_pts, st, err);
    if (r) return r;
    if (s.v.max_pts == 0) { MAPCHK(hipMemsetAsync(s.v.pnode, 0xFF, (size_t)W * nc * 4, st)); MAPCHK(hipMemsetAsync(s.v.pleaf, 0xFF, (size_t)W * nc * 4, st)); }
    else {  // new tail of every slot must read "no node"
      for (int sl = 0; sl < W; sl++) {
        MAPCHK(hipMemsetAsync(s.v.pnode + (size_t)sl * nc + s.v.max_pts, 0xFF, (nc - s.v.max_pts) * 4, st));
        MAPCHK(hipMemsetAsync(s.v.pleaf + (size_t)sl * nc + s.v.max_pts, 0xFF, (nc - s.v.max_pts) * 4, st));
      }
    }
    s.v.max_pts = (int)nc;
  }
  if (need_fix > (size_t)s.v.cap_fix) {
    size_t nc = s.v.cap_fix ? (size_t)s.v.cap_fix : (size_t)1 << 20;
    while (nc < need_fix) nc *= 2;
    int r = grow_arrays(fix_arrays(s.v), (size_t)s.v.cap_fix, nc, (size_t)s.v.cap_fix, st, err);
    if (r) return r;
    s.v.cap_fix = (int)nc;
  }
  // keep the hash table under ~50 % load, counting the tombstones of pruned roots (insertion reuses them, lookups walk past
  // them): when the live roots alone would fit, the table is re-hashed at its current size, which drops the tombstones
  if (2 * ((size_t)s.ub_used + need_pts) > (size_t)s.hcap) {
    if (s.cnt_stale) { int r = map_read_counters(s, st, err); if (r) return r; }
    if (2 * ((size_t)s.ub_used + need_pts) > (size_t)s.hcap) {
      unsigned int nc = s.hcap;
      while ((size_t)nc < 2 * ((size_t)s.ub_roots + need_pts) && nc < (1u << 30)) nc *= 2;
      int r = map_hash_alloc(s, nc, st, err);
      if (r) return r;
    }
  }
  return VBA_OK;
}

inline void map_free(MapStore &s) {
  if (!s.allocated) return;
  const int W = s.opt.win_size;
  for (auto &a : node_arrays(s.v, W)) if (*a.slot) hipFree(*a.slot);
  for (auto &a : scan_arrays(s.v, W)) if (*a.slot) hipFree(*a.slot);
  for (auto &a : fix_arrays(s.v)) if (*a.slot) hipFree(*a.slot);
  if (s.v.hkeys) hipFree(s.v.hkeys);
  if (s.v.hvals) hipFree(s.v.hvals);
  if (s.v.cnt) hipFree(s.v.cnt);
  if (s.v.fhist) hipFree(s.v.fhist);
  if (s.v.poses) hipFree(s.v.poses);
  if (s.h_cnt) hipHostFree(s.h_cnt);
  if (s.h_pose_ring) hipHostFree(s.h_pose_ring);
  for (int i = 0; i < 8; i++) if (s.pose_ev[i]) hipEventDestroy(s.pose_ev[i]);
  if (s.d_stage) hipFree(s.d_stage);
  if (s.d_sort_tmp) { hipFree(s.d_sort_tmp); s.d_sort_tmp = nullptr; s.sort_tmp_bytes = 0; s.sort_tmp_for = 0; }
  if (s.d_gc) { hipFree(s.d_gc); s.d_gc = nullptr; }
  s.v = MapView{};
  s.allocated = false;
}

inline bool is_device_ptr(const void *p) {
  hipPointerAttribute_t a;
  if (hipPointerGetAttributes(&a, p) != hipSuccess) { (void)hipGetLastError(); return false; }
  return a.type == hipMemoryTypeDevice;
}
inline int map_stage(MapStore &s, size_t bytes, std::string &err) {
  if (bytes <= s.stage_bytes) return VBA_OK;
  if (s.d_stage) hipFree(s.d_stage);
  s.d_stage = nullptr; s.stage_bytes = 0;
  MAPCHK(hipMalloc(&s.d_stage, bytes));
  s.stage_bytes = bytes;
  return VBA_OK;
}
inline int map_set_counter(MapStore &s, hipStream_t st, int which, int val, std::string &err) {   // stream-ordered, no host sync
  hipLaunchKernelGGL(k_set_counter, dim3(1), dim3(1), 0, st, s.v.cnt, which, val);
  MAPCHK(hipGetLastError());
  return VBA_OK;
}

// cnt[to] = sum over the ranks of cnt[from]  (no-op for an unsharded map)
inline int map_global_count(MapStore &s, hipStream_t st, int from, int to, std::string &err) {
  if (s.n_ranks <= 1 || !s.allreduce) return VBA_OK;
  if (!s.d_gc) MAPCHK(hipMalloc((void **)&s.d_gc, 2 * sizeof(double)));
  hipLaunchKernelGGL(k_cnt_to_f64, dim3(1), dim3(1), 0, st, s.v.cnt, from, s.d_gc);
  if (s.allreduce(s.d_gc, 1)) { err = "collective failed while summing a map counter over the ranks"; return VBA_ERR_HIP; }
  hipLaunchKernelGGL(k_f64_to_cnt, dim3(1), dim3(1), 0, st, s.d_gc, s.v.cnt, to);
  MAPCHK(hipGetLastError());
  return VBA_OK;
}

// rocPRIM scratch for sorting up to max_pts (leaf, point) pairs
inline int map_sort_reserve(MapStore &s, hipStream_t st, std::string &err) {
  if (s.sort_tmp_for >= s.v.max_pts) return VBA_OK;
  size_t need = 0;
  MAPCHK(sort_pairs_u32(nullptr, need, s.v.skey_a, s.v.skey_b, s.v.sval_a, s.v.sval_b, (size_t)s.v.max_pts, 32u, st));
  if (need > s.sort_tmp_bytes) {
    MAPCHK(hipStreamSynchronize(st));
    if (s.d_sort_tmp) hipFree(s.d_sort_tmp);
    s.d_sort_tmp = nullptr; s.sort_tmp_bytes = 0;
    MAPCHK(hipMalloc(&s.d_sort_tmp, need + 256));
    s.sort_tmp_bytes = need + 256;
  }
  s.sort_tmp_for = s.v.max_pts;
  return VBA_OK;
}
// sort key = node id < cap; "no leaf" = all ones, which must sort behind every id
inline unsigned int map_key_bits(const MapStore &s) {
  unsigned int bits = 1;
  while (bits < 32 && (1ull << bits) <= (unsigned long long)s.v.cap) bits++;
  return bits;
}

// cut_voxel / cut_voxel_multi for one scan
inline int map_cut_voxel(MapStore &s, hipStream_t st, int win_count, int n, const double *pnt_body, const double *var, const double *pose,
                         bool multi, std::string &err, const double *cov6 = nullptr) {
  const int W = s.opt.win_size;
  if (win_count < 0 || win_count >= W || n < 0 || !pose || (n > 0 && !pnt_body)) return VBA_ERR_BAD_ARG;
  int r = map_base(s, st, err);
  if (r) return r;
  if (s.cnt_stale && (s.ub_nodes + n + 64 > (long long)s.v.cap || 2 * (s.ub_used + n) > (long long)s.hcap)) {
    r = map_read_counters(s, st, err);      // bounds too pessimistic for the current capacity: fetch the true counts
    if (r) return r;
  }
  r = map_ensure(s, st, (size_t)s.ub_nodes + (size_t)n + 64, (size_t)n, (size_t)1, err);
  if (r) return r;
  const int slot = s.mp[win_count];
  s.npts[slot] = n;
  if (n == 0) return VBA_OK;
  // stage the points into the slot's SoA arrays
  const double *d_pts = pnt_body, *d_var = var;
  if (!is_device_ptr(pnt_body)) {
    const size_t bytes = (size_t)n * 3 * 8 + (var ? (size_t)n * 9 * 8 : 0);
    r = map_stage(s, bytes, err);
    if (r) return r;
    MAPCHK(hipMemcpyAsync(s.d_stage, pnt_body, (size_t)n * 3 * 8, hipMemcpyHostToDevice, st));
    d_pts = (const double *)s.d_stage;
    if (var) {
      MAPCHK(hipMemcpyAsync((char *)s.d_stage + (size_t)n * 3 * 8, var, (size_t)n * 9 * 8, hipMemcpyHostToDevice, st));
      d_var = (const double *)((char *)s.d_stage + (size_t)n * 3 * 8);
    }
  }
  if (var) s.have_var = true;
  {  // pose upload through a pinned ring: no implicit synchronisation of a pageable copy
    if (!s.h_pose_ring) MAPCHK(hipHostMalloc((void **)&s.h_pose_ring, 8 * 40 * sizeof(double), hipHostMallocDefault));
    const int k = s.pose_next; s.pose_next = (k + 1) & 7;
    if (!s.pose_ev[k]) MAPCHK(hipEventCreateWithFlags(&s.pose_ev[k], hipEventDisableTiming));
    else MAPCHK(hipEventSynchronize(s.pose_ev[k]));
    // entry = the device image poses[0 .. 34): pose (12) | 4 unused | rot_var, tsl_var (18) — one copy command
    std::memcpy(s.h_pose_ring + 40 * k, pose, 12 * sizeof(double));
    if (cov6) std::memcpy(s.h_pose_ring + 40 * k + 16, cov6, 18 * sizeof(double));
    MAPCHK(hipMemcpyAsync(s.v.poses, s.h_pose_ring + 40 * k, (cov6 ? 34 : 12) * sizeof(double), hipMemcpyHostToDevice, st));
    MAPCHK(hipEventRecord(s.pose_ev[k], st));
  }
  const MapParams P = map_params(s);
  const int nb = (n + 255) / 256;
  if (cov6 && var) hipLaunchKernelGGL(k_scan_to_soa_pvec_update, dim3(nb), dim3(256), 0, st, s.v, W, slot, n, d_pts, d_var, s.v.poses, s.v.poses + 16);
  else hipLaunchKernelGGL(k_scan_to_soa, dim3(nb), dim3(256), 0, st, s.v, W, slot, n, d_pts, d_var);
  s.stamp++;
  hipLaunchKernelGGL(k_ins_keys, dim3(nb), dim3(256), 0, st, s.v, P, slot, n, 0, s.stamp);
  hipLaunchKernelGGL(k_ins_newroots, dim3(nb), dim3(256), 0, st, s.v, P, 0, 0.0, s.stamp);
  if (multi) { r = map_global_count(s, st, CNT_TOUCH, CNT_TOUCH_G, err); if (r) return r; }   // VM:2044 tests the whole scan's voxel count
  // order-preserving accumulation: leaf of every point + per-leaf counts -> segments (scan over the touched leaves) -> scatter ->
  // one wave (workgroup for big leaves) per leaf puts its segment into scan order and adds in that order
  hipLaunchKernelGGL(k_ins_leaf, dim3(nb), dim3(256), 0, st, s.v, P, slot, n, multi ? 1 : 0);
  {
    long long ubn = (long long)s.ub_nodes + n;               // the insert creates at most one node (a root) per point
    if (ubn > s.v.cap) ubn = s.v.cap;
    hipLaunchKernelGGL(k_ins_scan, dim3((unsigned)((ubn + 255) / 256)), dim3(256), 0, st, s.v, slot);
  }
  hipLaunchKernelGGL(k_ins_scatter, dim3(nb), dim3(256), 0, st, s.v, slot, n);
  {
    const int nwg = n < 8192 ? ((n + 7) & ~7) : 8192;   // grid-stride over the work list (its length stays on the device); a multiple of 8
    int win = 128; while (win < n && win < (1 << 19)) win *= 2;         // bitmap window of the big-leaf kernel (<= 96 KB of LDS)
    size_t lds_big = (size_t)(win / 64) * 12 + 16;
    if (lds_big < (size_t)4 * 64 * 33 * 8) lds_big = (size_t)4 * 64 * 33 * 8;             // bitmap + prefix, then the four term images in the same space
    static bool attr_set[64] = {false};
    int dev = 0; hipGetDevice(&dev);
    if (!attr_set[dev & 63]) {
      hipFuncSetAttribute((const void *)k_ins_accum_big<true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 64);
      hipFuncSetAttribute((const void *)k_ins_accum_big<false>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 64);
      attr_set[dev & 63] = true;
    }
    if (var) {
      hipLaunchKernelGGL((k_ins_accum_ord<true>), dim3(nwg), dim3(64), 0, st, s.v, P, slot);
      hipLaunchKernelGGL((k_ins_accum_big<true>), dim3(256), dim3(256), lds_big, st, s.v, P, slot, n, win);
    } else {
      hipLaunchKernelGGL((k_ins_accum_ord<false>), dim3(nwg), dim3(64), 0, st, s.v, P, slot);
      hipLaunchKernelGGL((k_ins_accum_big<false>), dim3(256), dim3(256), lds_big, st, s.v, P, slot, n, win);
    }
  }
  MAPCHK(hipGetLastError());
  // no read-back: capacity was reserved for the worst case (n new roots), so this call cannot overflow
  s.ub_nodes += n; s.ub_roots += n; s.ub_used += n; s.cnt_stale = true;
  if (!is_device_ptr(pnt_body)) MAPCHK(hipStreamSynchronize(st));   // the caller's host point buffers may go away (pose and covariance went through the pinned ring)
  return VBA_OK;
}

inline int map_cut_voxel_fix(MapStore &s, hipStream_t st, int n, const double *pnt_world, double jour, std::string &err) {
  if (n < 0 || (n > 0 && !pnt_world)) return VBA_ERR_BAD_ARG;
  if (n == 0) return VBA_OK;
  int r = map_base(s, st, err);
  if (r) return r;
  if (s.cnt_stale) { r = map_read_counters(s, st, err); if (r) return r; }
  r = map_ensure(s, st, (size_t)s.h_cnt[CNT_NODES] + (size_t)n + 64, (size_t)n, (size_t)s.h_cnt[CNT_FIX] + (size_t)n, err);
  if (r) return r;
  const double *d_pts = pnt_world;
  if (!is_device_ptr(pnt_world)) {
    r = map_stage(s, (size_t)n * 3 * 8, err);
    if (r) return r;
    MAPCHK(hipMemcpyAsync(s.d_stage, pnt_world, (size_t)n * 3 * 8, hipMemcpyHostToDevice, st));
    d_pts = (const double *)s.d_stage;
  }
  const MapParams P = map_params(s);
  const int base = s.h_cnt[CNT_FIX];
  const int nb = (n + 255) / 256;
  hipLaunchKernelGGL(k_fix_to_soa, dim3(nb), dim3(256), 0, st, s.v, base, n, d_pts);
  r = map_set_counter(s, st, CNT_NEWSLOTS, 0, err); if (r) return r;
  r = map_set_counter(s, st, CNT_FIX, base + n, err); if (r) return r;
  hipLaunchKernelGGL(k_ins_keys, dim3(nb), dim3(256), 0, st, s.v, P, base, n, 1, 0);
  hipLaunchKernelGGL(k_ins_newroots, dim3(nb), dim3(256), 0, st, s.v, P, 1, jour, 0);
  r = map_sort_reserve(s, st, err); if (r) return r;
  r = map_set_counter(s, st, CNT_WL, 0, err); if (r) return r;
  hipLaunchKernelGGL(k_fix_leaf, dim3(nb), dim3(256), 0, st, s.v, P, base, n);
  {
    size_t tb = s.sort_tmp_bytes;
    MAPCHK(sort_pairs_u32(s.d_sort_tmp, tb, s.v.skey_a, s.v.skey_b, s.v.sval_a, s.v.sval_b, (size_t)n, map_key_bits(s), st));
  }
  hipLaunchKernelGGL(k_fix_heads, dim3(nb), dim3(256), 0, st, s.v, n);
  hipLaunchKernelGGL(k_fix_accum_ord, dim3(n < 4096 ? n : 4096), dim3(64), 0, st, s.v, P, base, n, d_pts);
  MAPCHK(hipGetLastError());
  r = map_read_counters(s, st, err);
  if (r) return r;
  if (s.h_cnt[CNT_OVERFLOW]) { err = "voxel map capacity exceeded during fixed-point insert"; return VBA_ERR_CAPACITY; }
  return VBA_OK;
}

// recut over the scope + factor index assignment; *n_factors = number of planar leaves selected by tras_opt
inline int map_recut(MapStore &s, hipStream_t st, int win_count, const double *poses, bool multi, std::string &err, int *n_factors) {
  const int W = s.opt.win_size;
  *n_factors = 0;
  if (win_count < 0 || win_count > W || !poses) return VBA_ERR_BAD_ARG;
  if (!s.allocated) return VBA_OK;
  if (multi) { int r0 = map_global_count(s, st, CNT_SLIDE, CNT_SLIDE_G, err); if (r0) return r0; }   // VS:1693 tests surf_map_slide.size() of the whole map
  for (int attempt = 0; attempt < 8; attempt++) {
    // The first attempt works from the host's upper bound of the node count (exact at the last read-back + the points inserted
    // since): no read-back, hence no drain of the stream, before the pass.  The pass ends with the one read-back that serves the
    // overflow check, the factor count and the next call's bounds.
    int r = VBA_OK;
    if (attempt > 0) { r = map_read_counters(s, st, err); if (r) return r; }
    const size_t nodes_ub = attempt ? (size_t)s.h_cnt[CNT_NODES] : (size_t)s.ub_nodes;
    // room for every current leaf to split once per level (checked again through the overflow flag)
    r = map_ensure(s, st, nodes_ub + 8 * (size_t)(attempt ? s.h_cnt[CNT_NODES] : 65536), 0, 0, err);
    if (r) return r;
    MAPCHK(hipMemcpyAsync(s.v.poses, poses, (size_t)(win_count > 0 ? win_count : 1) * 12 * sizeof(double), hipMemcpyHostToDevice, st));
    const MapParams P = map_params(s);
    int max_n = 0;
    for (int i = 0; i < win_count; i++) if (s.npts[s.mp[i]] > max_n) max_n = s.npts[s.mp[i]];
    const int grid_nodes = (s.v.cap + 255) / 256;
    if (nodes_ub > 0) {
      for (int L = 0; L <= s.opt.max_layer; L++) {
        s.epoch++;
        hipLaunchKernelGGL(k_recut_prep, dim3(1), dim3(1), 0, st, s.v.cnt, L == 0 ? 1 : 0, L == s.opt.max_layer ? 1 : 0);
        hipLaunchKernelGGL(k_recut_leaf, dim3(grid_nodes), dim3(256), 0, st, s.v, P, L, multi ? 1 : 0, s.epoch);
        if (L < s.opt.max_layer) {
          if (s.have_var) hipLaunchKernelGGL((k_recut_push<true>), dim3(4096), dim3(256), 0, st, s.v, P, win_count, L + 1);
          else hipLaunchKernelGGL((k_recut_push<false>), dim3(4096), dim3(256), 0, st, s.v, P, win_count, L + 1);
        }
#ifdef VBA_DIAG
        if (getenv("VBA_RECUT_STATS")) {
          int h[CNT_N];
          hipStreamSynchronize(st);
          hipMemcpy(h, s.v.cnt, sizeof(h), hipMemcpyDeviceToHost);
          fprintf(stderr, "[recut] level %d: nodes %d, split leaves %d, candidates scanned %d, matched %d, fix blocks walked %d (%d entries)\n", L, h[CNT_NODES], h[CNT_SPLIT], h[CNT_DBG0], h[CNT_DBG1], h[CNT_DBG2], h[CNT_DBG3]);
          const int z[4] = {0, 0, 0, 0};
          hipMemcpy(s.v.cnt + CNT_DBG0, z, sizeof(z), hipMemcpyHostToDevice);
        }
#endif
      }
      // tras_opt pass 1 rides in the same submission: one counter read-back serves the overflow check and the factor count
      hipLaunchKernelGGL(k_extract_count, dim3(grid_nodes), dim3(256), 0, st, s.v, P, multi ? 1 : 0);
    }
    MAPCHK(hipGetLastError());
    r = map_read_counters(s, st, err);
    if (r) return r;
    if (s.h_cnt[CNT_OVERFLOW] == 4) { err = "root hash table full during scan insertion"; return VBA_ERR_CAPACITY; }
    if (!s.h_cnt[CNT_OVERFLOW]) break;
    // a leaf could not be split for lack of node space: clamp the counter, grow and run the pass again (idempotent)
    if (s.h_cnt[CNT_NODES] > s.v.cap) { r = map_set_counter(s, st, CNT_NODES, s.v.cap, err); if (r) return r; }
    if (attempt == 7) { err = "voxel map node capacity exceeded during recut"; return VBA_ERR_CAPACITY; }
  }
  if (s.h_cnt[CNT_NODES] == 0) s.h_cnt[CNT_FACTORS] = 0;
  *n_factors = s.h_cnt[CNT_FACTORS];
  s.h_cnt[CNT_N] = multi ? 1 : 0;   // remembered for map_extract_factors
  return VBA_OK;
}

inline int map_extract_factors(MapStore &s, hipStream_t st, FactorView f, std::string &err, int *n_factors) {
  *n_factors = 0;
  if (!s.allocated) return VBA_OK;
  const MapParams P = map_params(s);
  const int nfac = s.h_cnt[CNT_NODES] > 0 ? s.h_cnt[CNT_FACTORS] : 0;
  if (nfac > 1) {
    const int nbuckets = 1 << (s.opt.win_size < 10 ? s.opt.win_size : 10);
    MAPCHK(hipMemsetAsync(s.v.fhist, 0, (size_t)nbuckets * sizeof(int), st));
    hipLaunchKernelGGL(k_extract_key, dim3((nfac + 255) / 256), dim3(256), 0, st, s.v, P, nfac, nbuckets);
    hipLaunchKernelGGL(k_extract_scan, dim3(1), dim3(1024), 0, st, s.v, nbuckets);
    hipLaunchKernelGGL(k_extract_scatter, dim3((nfac + 255) / 256), dim3(256), 0, st, s.v, nfac, nbuckets);
  }
  if (nfac > 0) hipLaunchKernelGGL(k_extract_write, dim3((nfac + XW_F - 1) / XW_F), dim3(256), (size_t)(10 * s.opt.win_size + 33) * (XW_F + 1) * 8, st, s.v, P, f, nfac);
  MAPCHK(hipGetLastError());
  *n_factors = s.h_cnt[CNT_FACTORS];
  return VBA_OK;
}

inline int map_margi(MapStore &s, hipStream_t st, int win_count, const double *poses, double jour, FactorView f, int nfac, std::string &err) {
  const int W = s.opt.win_size;
  if (win_count < 1 || win_count > W || !poses) return VBA_ERR_BAD_ARG;
  if (!s.allocated) return VBA_OK;
  int r = VBA_OK;
  r = map_global_count(s, st, CNT_SLIDE, CNT_SLIDE_G, err); if (r) return r;   // VS:1616 tests the whole sliding map (every rank enters this collective)
  if (s.cnt_stale || s.n_ranks > 1) { r = map_read_counters(s, st, err); if (r) return r; }    // (the recut before the optimisation left them current)
  const int slot0 = s.mp[0];
  r = map_ensure(s, st, 0, 0, (size_t)s.h_cnt[CNT_FIX] + (size_t)s.npts[slot0] + 1, err);
  if (r) return r;
  hipLaunchKernelGGL(k_set_counter2, dim3(1), dim3(1), 0, st, s.v.cnt, (int)CNT_OVERFLOW, 0, (int)CNT_TAKE, 0);
  MAPCHK(hipMemcpyAsync(s.v.poses, poses, (size_t)win_count * 12 * sizeof(double), hipMemcpyHostToDevice, st));
  const MapParams P = map_params(s);
  const int nn = s.h_cnt[CNT_NODES] < s.v.cap ? s.h_cnt[CNT_NODES] : s.v.cap;
  const int n_slide_before = s.n_ranks > 1 ? s.h_cnt[CNT_SLIDE_G] : s.h_cnt[CNT_SLIDE];
  if (nn == 0) return VBA_OK;
  s.epoch++;
  const dim3 gn((nn + 255) / 256), b(256);
  hipLaunchKernelGGL(k_margi_leaf, gn, b, 0, st, s.v, P, f, nfac, win_count, s.epoch);
  if (n_slide_before >= s.opt.thread_num) {
    if (s.npts[slot0] > 0) {
      hipLaunchKernelGGL(k_margi_take, dim3(4096), dim3(64), 0, st, s.v, P, s.have_var ? 1 : 0);
      hipLaunchKernelGGL(k_margi_points, dim3((s.v.max_pts + 1023) / 1024), dim3(1024), 0, st, s.v, P);
    }
    if (s.h_cnt[CNT_FIX] > 0) hipLaunchKernelGGL(k_margi_fixclear, dim3((s.h_cnt[CNT_FIX] + 255) / 256), b, 0, st, s.v, s.epoch);
    for (int L = s.opt.max_layer - 1; L >= 0; L--) hipLaunchKernelGGL(k_margi_up, gn, b, 0, st, s.v, P, L);
    hipLaunchKernelGGL(k_margi_roots, gn, b, 0, st, s.v, P, jour, s.epoch, n_slide_before);
    hipLaunchKernelGGL(k_margi_clear_nodes, gn, b, 0, st, s.v, P, s.epoch);
    hipLaunchKernelGGL(k_margi_clear_points, dim3((s.v.max_pts + 255) / 256, W), b, 0, st, s.v, P, s.epoch);
    s.npts[slot0] = 0;
  }
  MAPCHK(hipGetLastError());
  r = map_read_counters(s, st, err);
  if (r) return r;
  if (s.h_cnt[CNT_OVERFLOW] == 2) { err = "Error: opt_state out of range"; return VBA_ERR_OPT_STATE; }
  if (s.h_cnt[CNT_OVERFLOW]) { err = "fixed-point pool capacity exceeded"; return VBA_ERR_CAPACITY; }
  return VBA_OK;
}

inline int map_slide(MapStore &s, int mgsize) {   // VS:2014-2019
  const int W = s.opt.win_size;
  if (mgsize < 0 || mgsize > W) return VBA_ERR_BAD_ARG;
  for (int i = 0; i < W; i++) { s.mp[i] += mgsize; if (s.mp[i] >= W) s.mp[i] -= W; }
  return VBA_OK;
}

inline int map_reset(MapStore &s, hipStream_t st, std::string &err) {
  if (!s.allocated) return VBA_OK;
  const int W = s.opt.win_size;
  hipStreamSynchronize(st);
  for (auto &a : node_arrays(s.v, W)) MAPCHK(hipMemsetAsync(*a.slot, 0, a.elem * a.rows * (size_t)s.v.cap, st));
  if (s.v.pnode) MAPCHK(hipMemsetAsync(s.v.pnode, 0xFF, (size_t)W * s.v.max_pts * 4, st));
  if (s.v.pleaf) MAPCHK(hipMemsetAsync(s.v.pleaf, 0xFF, (size_t)W * s.v.max_pts * 4, st));
  if (s.v.fnode) MAPCHK(hipMemsetAsync(s.v.fnode, 0xFF, (size_t)s.v.cap_fix * 4, st));
  hipLaunchKernelGGL(k_fill_u64, dim3(1024), dim3(256), 0, st, s.v.hkeys, KEY_EMPTY, (size_t)s.hcap);
  MAPCHK(hipMemsetAsync(s.v.hvals, 0xFF, (size_t)s.hcap * 4, st));
  MAPCHK(hipMemsetAsync(s.v.cnt, 0, CNT_N * sizeof(int), st));
  MAPCHK(hipStreamSynchronize(st));
  std::memset(s.h_cnt, 0, CNT_N * sizeof(int));
  for (int i = 0; i < VBA_MAX_WIN; i++) { s.mp[i] = i; s.npts[i] = 0; }
  s.have_var = false; s.ub_nodes = 0; s.ub_roots = 0; s.ub_used = 0; s.cnt_stale = false;
  return VBA_OK;
}

inline int map_num_roots(MapStore &s, hipStream_t st, bool slide) {
  if (!s.allocated) return 0;
  std::string err;
  if (map_read_counters(s, st, err)) return -1;
  return slide ? s.h_cnt[CNT_SLIDE] : s.h_cnt[CNT_ROOTS];
}

// storage statistics: [node high-water mark, free root nodes, free child blocks, hash capacity, hash slots in use (roots +
// tombstones), roots, sliding-map roots, fixed points]
inline int map_stats(MapStore &s, hipStream_t st, long long *out8, std::string &err) {
  for (int k = 0; k < 8; k++) out8[k] = 0;
  if (!s.allocated) return VBA_OK;
  int r = map_read_counters(s, st, err);
  if (r) return r;
  out8[0] = s.h_cnt[CNT_NODES]; out8[1] = s.h_cnt[CNT_FREE_ROOTS]; out8[2] = s.h_cnt[CNT_FREE_BLOCKS]; out8[3] = s.hcap;
  out8[4] = s.h_cnt[CNT_USED]; out8[5] = s.h_cnt[CNT_ROOTS]; out8[6] = s.h_cnt[CNT_SLIDE]; out8[7] = s.h_cnt[CNT_FIX];
  return VBA_OK;
}

inline int map_dump_leaves(MapStore &s, hipStream_t st, double *out, int max_leaves, std::string &err) {
  if (!s.allocated) return 0;
  if (map_read_counters(s, st, err)) return -1;
  const int nn = s.h_cnt[CNT_NODES] < s.v.cap ? s.h_cnt[CNT_NODES] : s.v.cap;
  if (nn == 0) return 0;
  const int cap_out = out ? max_leaves : 0;
  double *d_out = nullptr;
  if (cap_out > 0 && hipMalloc((void **)&d_out, (size_t)cap_out * 39 * 8) != hipSuccess) return -1;
  if (map_set_counter(s, st, CNT_LEAVES, 0, err)) return -1;
  hipLaunchKernelGGL(k_dump_leaves, dim3((nn + 255) / 256), dim3(256), 0, st, s.v, d_out, cap_out);
  if (map_read_counters(s, st, err)) return -1;
  const int n = s.h_cnt[CNT_LEAVES];
  if (cap_out > 0) {
    hipMemcpy(out, d_out, (size_t)(n < cap_out ? n : cap_out) * 39 * 8, hipMemcpyDeviceToHost);
    hipFree(d_out);
  }
  return n;
}

inline int map_dump_plane_var(MapStore &s, hipStream_t st, double *out, int max_leaves, std::string &err) {
  if (!s.allocated) return 0;
  if (map_read_counters(s, st, err)) return -1;
  const int nn = s.h_cnt[CNT_NODES] < s.v.cap ? s.h_cnt[CNT_NODES] : s.v.cap;
  if (nn == 0) return 0;
  const int cap_out = out ? max_leaves : 0;
  double *d_out = nullptr;
  if (cap_out > 0 && hipMalloc((void **)&d_out, (size_t)cap_out * 86 * 8) != hipSuccess) return -1;
  if (map_set_counter(s, st, CNT_LEAVES, 0, err)) return -1;
  hipLaunchKernelGGL(k_dump_plane_var, dim3((nn + 255) / 256), dim3(256), 0, st, s.v, d_out, cap_out);
  if (map_read_counters(s, st, err)) return -1;
  const int n = s.h_cnt[CNT_LEAVES];
  if (cap_out > 0) {
    hipMemcpy(out, d_out, (size_t)(n < cap_out ? n : cap_out) * 86 * 8, hipMemcpyDeviceToHost);
    hipFree(d_out);
  }
  return n;
}

inline int map_prune(MapStore &s, hipStream_t st, double jour, int dist, std::string &err) {
  if (!s.allocated) return VBA_OK;
  int r = map_read_counters(s, st, err);
  if (r) return r;
  const int nn = s.h_cnt[CNT_NODES] < s.v.cap ? s.h_cnt[CNT_NODES] : s.v.cap;
  if (nn == 0) return VBA_OK;
  s.epoch++;
  hipLaunchKernelGGL(k_prune_roots, dim3((s.hcap + 255) / 256), dim3(256), 0, st, s.v, jour, dist, s.epoch);
  hipLaunchKernelGGL(k_prune_nodes, dim3((nn + 255) / 256), dim3(256), 0, st, s.v, s.epoch);
  hipLaunchKernelGGL(k_prune_finish, dim3((nn + 255) / 256), dim3(256), 0, st, s.v, s.epoch);
  hipLaunchKernelGGL(k_prune_zero, dim3((nn + 255) / 256, 130 + 10 * s.opt.win_size), dim3(256), 0, st, s.v, s.opt.win_size, s.epoch);
  if (s.h_cnt[CNT_FIX] > 0) hipLaunchKernelGGL(k_prune_fix, dim3((s.h_cnt[CNT_FIX] + 255) / 256), dim3(256), 0, st, s.v);
  MAPCHK(hipGetLastError());
  return map_read_counters(s, st, err);
}

// One scan-to-map accumulation: out34 (host) = [HTH upper (21) | HTz (6) | nnt upper (6) | match_num]
inline int map_odom_accumulate(MapStore &s, hipStream_t st, const OdomState &X, int n, const double *d_pts, const double *d_var,
                               double *d_partial, double *d_out34, double *out34, std::string &err) {
  if (!s.allocated) { for (int k = 0; k < 34; k++) out34[k] = 0.0; return VBA_OK; }
  const MapParams P = map_params(s);
  const int nb = (n + 255) / 256;
  hipLaunchKernelGGL(k_odom_match, dim3(nb), dim3(256), 0, st, s.v, P, X, n, d_pts, d_var, d_partial);
  hipLaunchKernelGGL(k_reduce_partials, dim3(3), dim3(256), 0, st, d_partial, nb, 34, d_out34, (const int *)nullptr);
  MAPCHK(hipGetLastError());
  MAPCHK(hipStreamSynchronize(st));
  MAPCHK(hipMemcpyAsync(out34, d_out34, 34 * sizeof(double), hipMemcpyDeviceToHost, st));
  MAPCHK(hipStreamSynchronize(st));
  return VBA_OK;
}

}  // namespace vba
