// Keyframe-window octree of the hierarchical global BA: OctreeGBA::cut_voxel over all keyframes of a
// window (loop_refine.hpp:439-479) + OctreeGBA_multi_recut / recut / subdivide (LR:320-404, 483-537), rebuilt from the
// current poses at every outer iteration of HBA_add_edge (voxelslam.cpp:2884-2891).  Unlike the local map nothing is
// incremental, so the build is level-synchronous over ALL points:
//   k_gba_keys   world point + root key (float quirk of LR:446-451) + hash insert
//   k_gba_roots  hash slots -> node ids, root geometry (centre, quater_length as float, LR:470-474)
//   per layer:   k_gba_accum (world cluster + per-keyframe body clusters, f64 atomics)
//                k_gba_decide (N <= 10 / plane_judge / exi / 0.12 ratio gates LR:358-383, or 8 children)
//                k_gba_descend (octant of LR:327-331)
//   k_gba_extract  planar voxels -> the SoA factor store (push_voxel, LR:383)
// Data-dependent sizes (roots, nodes, factors) live in device counters; the host reads them once per build.
#pragma once
#include <hip/hip_runtime.h>
#include <string>
#include <vector>

namespace vba {

enum { GCNT_NODES = 0, GCNT_FACTORS, GCNT_OVERFLOW, GCNT_ROOTS, GCNT_N };

struct GbaView {
  unsigned long long *hkeys; int *hvals; unsigned int hmask;
  int cap, W, npts;
  double *nadd;     // [10][cap]     world cluster (pcr_add)
  double *nlc;      // [10][W][cap]  body clusters per keyframe
  double *ncenter;  // [3][cap]
  float *nql;       // [cap]
  int *nchild, *nfac;
  signed char *nlayer;
  double *neval, *nevec;   // [3][cap], [9][cap]
  double *pw;       // [3][npts] world points
  const double *pl; // [npts][3] local points (caller's layout)
  int *pframe, *pnode;
  int *cnt;
  double *poses;    // [W][12]
  int *offsets;     // [W+1]
};

struct GbaParams { double voxel_size, min_eigen_value, eig_array[4]; int max_layer; };

__global__ void k_gba_keys(GbaView g, GbaParams P) {
  const int p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= g.npts) return;
  int f = 0;
  while (f + 1 < g.W && p >= g.offsets[f + 1]) f++;
  const double *R = g.poses + 12 * f;
  const double x = g.pl[3 * (size_t)p], y = g.pl[3 * (size_t)p + 1], z = g.pl[3 * (size_t)p + 2];
  const double wx = (R[0] * x + R[1] * y + R[2] * z) + R[9], wy = (R[3] * x + R[4] * y + R[5] * z) + R[10], wz = (R[6] * x + R[7] * y + R[8] * z) + R[11];
  const size_t n = (size_t)g.npts;
  g.pw[p] = wx; g.pw[n + p] = wy; g.pw[2 * n + p] = wz;
  g.pframe[p] = f;
  const long long kx = key_axis(wx, P.voxel_size), ky = key_axis(wy, P.voxel_size), kz = key_axis(wz, P.voxel_size);
  if (kx < -KEY_OFF || kx >= KEY_OFF || ky < -KEY_OFF || ky >= KEY_OFF || kz < -KEY_OFF || kz >= KEY_OFF) { g.pnode[p] = -1; atomicExch(&g.cnt[GCNT_OVERFLOW], 2); return; }
  const unsigned long long key = pack_key(kx, ky, kz);
  unsigned long long hsh = key * 0x9E3779B97F4A7C15ull;
  unsigned int h = (unsigned int)(hsh >> 32) & g.hmask;
  for (unsigned int probe = 0; probe <= g.hmask; probe++) {
    unsigned long long old = g.hkeys[h];                               // almost every point finds its root present: read before the CAS
    if (old == key) break;
    if (old == KEY_EMPTY) {
      old = atomicCAS(&g.hkeys[h], KEY_EMPTY, key);
      if (old == KEY_EMPTY || old == key) break;
    }
    h = (h + 1) & g.hmask;
  }
  g.pnode[p] = (int)h;     // slot for now; k_gba_rootid turns it into the node id
}

// One root node per occupied hash slot.  A workgroup scans 4096 slots (16 per thread) and claims its node ids with ONE
// returning atomic: per-slot claims serialise on the counter (~10 ns per wave-level operation, one per root: 48 us for a
// few thousand roots in a 10^6-slot table).
template <class View>
__device__ __forceinline__ void gba_roots_body(View &g, const GbaParams &P, bool count_roots) {
  __shared__ int wbase[4];
  const unsigned int s0 = blockIdx.x * 4096u + threadIdx.x;
  unsigned long long mk[16];
  int wave_tot = 0;
#pragma unroll
  for (int j = 0; j < 16; j++) {
    const unsigned int s = s0 + 256u * j;
    const bool occ = s <= g.hmask && g.hkeys[s] != KEY_EMPTY;
    mk[j] = __ballot(occ);
    wave_tot += __popcll(mk[j]);
  }
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (lane == 0) wbase[wave] = wave_tot;
  __syncthreads();
  if (threadIdx.x == 0) {
    int tot = 0;
    for (int w = 0; w < 4; w++) { const int c = wbase[w]; wbase[w] = tot; tot += c; }
    const int base = tot ? atomicAdd(&g.cnt[GCNT_NODES], tot) : 0;
    if (tot && count_roots) atomicAdd(&g.cnt[GCNT_ROOTS], tot);
    for (int w = 0; w < 4; w++) wbase[w] += base;
  }
  __syncthreads();
  int running = wbase[wave];
  const size_t cp = (size_t)g.cap;
#pragma unroll
  for (int j = 0; j < 16; j++) {
    const unsigned int s = s0 + 256u * j;
    if ((mk[j] >> lane) & 1ull) {
      const int id = running + __popcll(mk[j] & ((1ull << lane) - 1ull));
      if (id >= g.cap) { atomicExch(&g.cnt[GCNT_OVERFLOW], 1); g.hvals[s] = -1; }
      else {
        g.hvals[s] = id;
        long long kx, ky, kz;
        unpack_key(g.hkeys[s], kx, ky, kz);
        g.ncenter[id] = (0.5 + (double)kx) * P.voxel_size;
        g.ncenter[cp + id] = (0.5 + (double)ky) * P.voxel_size;
        g.ncenter[2 * cp + id] = (0.5 + (double)kz) * P.voxel_size;
        g.nql[id] = (float)(P.voxel_size / 4.0);
        g.nlayer[id] = 0; g.nchild[id] = -1; g.nfac[id] = -1;
      }
    }
    running += __popcll(mk[j]);
  }
}
__global__ __launch_bounds__(256) void k_gba_roots(GbaView g, GbaParams P) { gba_roots_body(g, P, true); }

__global__ void k_gba_rootid(GbaView g) {
  const int p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= g.npts) return;
  const int s = g.pnode[p];
  if (s >= 0) g.pnode[p] = g.hvals[s];
}

// 256 consecutive points of a keyframe fall into a handful of nodes: the 20 cluster scalars are first summed per
// (node, frame) in an LDS hash table of the workgroup (as k_ins_accum, vba_kernels_map.hpp) and only its occupied entries
// go to HBM as f64 atomics — at layer 0 every point of a window lands in ~10^2 root voxels and per-point global atomics
// serialise on them.
__device__ __forceinline__ unsigned int gba_lds_claim(unsigned long long *tkey, unsigned long long key) {
  unsigned int e = (unsigned int)((key * 0x9E3779B97F4A7C15ull) >> 56);
  while (true) {
    const unsigned long long old = atomicCAS(&tkey[e], ~0ull, key);
    if (old == ~0ull || old == key) break;
    e = (e + 1) & 255;
  }
  return e;
}
__device__ __forceinline__ void gba_lds_add(double (*tacc)[256], unsigned int e, double x, double y, double z, double wx, double wy, double wz) {
  unsafeAtomicAdd(&tacc[0][e], wx * wx); unsafeAtomicAdd(&tacc[1][e], wx * wy); unsafeAtomicAdd(&tacc[2][e], wx * wz);
  unsafeAtomicAdd(&tacc[3][e], wy * wy); unsafeAtomicAdd(&tacc[4][e], wy * wz); unsafeAtomicAdd(&tacc[5][e], wz * wz);
  unsafeAtomicAdd(&tacc[6][e], wx); unsafeAtomicAdd(&tacc[7][e], wy); unsafeAtomicAdd(&tacc[8][e], wz); unsafeAtomicAdd(&tacc[9][e], 1.0);
  unsafeAtomicAdd(&tacc[10][e], x * x); unsafeAtomicAdd(&tacc[11][e], x * y); unsafeAtomicAdd(&tacc[12][e], x * z);
  unsafeAtomicAdd(&tacc[13][e], y * y); unsafeAtomicAdd(&tacc[14][e], y * z); unsafeAtomicAdd(&tacc[15][e], z * z);
  unsafeAtomicAdd(&tacc[16][e], x); unsafeAtomicAdd(&tacc[17][e], y); unsafeAtomicAdd(&tacc[18][e], z); unsafeAtomicAdd(&tacc[19][e], 1.0);
}

__global__ __launch_bounds__(256) void k_gba_accum(GbaView g) {
  __shared__ unsigned long long tkey[256];   // 256 points per workgroup -> at most 256 keys; 42 KB keeps three workgroups per CU
  __shared__ double tacc[20][256];
  const int tid = threadIdx.x;
  tkey[tid] = ~0ull;
  for (int t = tid; t < 20 * 256; t += 256) (&tacc[0][0])[t] = 0.0;
  __syncthreads();
  const int p = blockIdx.x * blockDim.x + tid;
  const size_t n = (size_t)g.npts, cp = (size_t)g.cap;
  if (p < g.npts) {
    const int id = g.pnode[p];
    if (id >= 0) {
      const unsigned int e = gba_lds_claim(tkey, ((unsigned long long)(unsigned int)id << 20) | (unsigned int)g.pframe[p]);
      gba_lds_add(tacc, e, g.pl[3 * (size_t)p], g.pl[3 * (size_t)p + 1], g.pl[3 * (size_t)p + 2], g.pw[p], g.pw[n + p], g.pw[2 * n + p]);
    }
  }
  __syncthreads();
  for (int t = tid; t < 20 * 256; t += 256) {
    const int k = t >> 8, e = t & 255;
    const unsigned long long key = tkey[e];
    if (key == ~0ull) continue;
    const double v = tacc[k][e];
    if (v == 0.0) continue;
    const size_t id = (size_t)(key >> 20), f = (size_t)(key & 0xFFFFFull);
    if (k < 10) unsafeAtomicAdd(g.nadd + (size_t)k * cp + id, v);
    else unsafeAtomicAdd(g.nlc + ((size_t)(k - 10) * g.W + f) * cp + id, v);
  }
}

__global__ void k_gba_decide(GbaView g, GbaParams P, int layer) {
  const int id = blockIdx.x * blockDim.x + threadIdx.x;
  const int nn = g.cnt[GCNT_NODES] < g.cap ? g.cnt[GCNT_NODES] : g.cap;
  if (id >= nn || g.nlayer[id] != layer) return;
  const size_t cp = (size_t)g.cap;
  const double N = g.nadd[9 * cp + id];
  if (N <= 10.0) return;                                                               // LR:360
  const double inv = 1.0 / N;   // (N is an integer count: v / N and P / N below follow PointCluster::cov, TL:333-337)
  const double cx = g.nadd[6 * cp + id] / N, cy = g.nadd[7 * cp + id] / N, cz = g.nadd[8 * cp + id] / N;
  const double a00 = g.nadd[id] / N - cx * cx, a01 = g.nadd[cp + id] / N - cx * cy, a02 = g.nadd[2 * cp + id] / N - cx * cz;
  const double a11 = g.nadd[3 * cp + id] / N - cy * cy, a12 = g.nadd[4 * cp + id] / N - cy * cz, a22 = g.nadd[5 * cp + id] / N - cz * cz;
  (void)inv;
  double w0, w1, w2, V[9];
  eig3_sym_dev(a00, a01, a02, a11, a12, a22, w0, w1, w2, V);
  const bool is_plane = (w0 < P.min_eigen_value) && ((w0 / w2) < P.eig_array[layer]);     // LR:310-314
  if (is_plane) {
    int exi = 0;
    for (int i = 0; i < g.W; i++) exi += (g.nlc[((size_t)9 * g.W + i) * cp + id] != 0.0) ? 1 : 0;
    if (exi <= 1) return;                                                               // LR:371-375
    if (w0 / w1 > 0.12) return;                                                         // LR:377
    const int a = atomicAdd(&g.cnt[GCNT_FACTORS], 1);
    g.nfac[id] = a;
    g.neval[id] = w0; g.neval[cp + id] = w1; g.neval[2 * cp + id] = w2;
    for (int k = 0; k < 9; k++) g.nevec[(size_t)k * cp + id] = V[k];
    return;
  }
  if (layer >= P.max_layer) return;                                                     // LR:388
  const int base = atomicAdd(&g.cnt[GCNT_NODES], 8);                                    // subdivide LR:320-356
  if (base + 8 > g.cap) { atomicExch(&g.cnt[GCNT_OVERFLOW], 1); return; }
  const float ql = g.nql[id];
  const double c0 = g.ncenter[id], c1 = g.ncenter[cp + id], c2 = g.ncenter[2 * cp + id];
  for (int o = 0; o < 8; o++) {
    const int ch = base + o;
    const int bx = (o >> 2) & 1, by = (o >> 1) & 1, bz = o & 1;
    g.ncenter[ch] = c0 + (double)((float)(2 * bx - 1) * ql);
    g.ncenter[cp + ch] = c1 + (double)((float)(2 * by - 1) * ql);
    g.ncenter[2 * cp + ch] = c2 + (double)((float)(2 * bz - 1) * ql);
    g.nql[ch] = ql / 2;
    g.nlayer[ch] = (signed char)(layer + 1); g.nchild[ch] = -1; g.nfac[ch] = -1;
  }
  g.nchild[id] = base;
}

__global__ void k_gba_descend(GbaView g) {
  const int p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= g.npts) return;
  const int id = g.pnode[p];
  if (id < 0) return;
  const int base = g.nchild[id];
  if (base < 0) { g.pnode[p] = -1; return; }
  const size_t n = (size_t)g.npts, cp = (size_t)g.cap;
  const int bx = g.pw[p] > g.ncenter[id] ? 1 : 0, by = g.pw[n + p] > g.ncenter[cp + id] ? 1 : 0, bz = g.pw[2 * n + p] > g.ncenter[2 * cp + id] ? 1 : 0;
  g.pnode[p] = base + 4 * bx + 2 * by + bz;
}

// One thread per (node, row of the factor record) — rows 0..10W-1 the body clusters, then fix (10, zero), pcr (10), coe,
// eigval (3), eigvec (9); one thread per node copying all 10W+33 scalars in turn was latency-bound (32 us per window).
__global__ void k_gba_extract(GbaView g, FactorView f) {
  const int id = blockIdx.x * blockDim.x + threadIdx.x;
  const int nn = g.cnt[GCNT_NODES] < g.cap ? g.cnt[GCNT_NODES] : g.cap;
  if (id >= nn) return;
  const int a = g.nfac[id];
  if (a < 0 || a >= f.vs) return;
  const size_t cp = (size_t)g.cap, vs = (size_t)f.vs;
  const int row = blockIdx.y, ncl = 10 * g.W;
  if (row < ncl) f.cl[(size_t)row * vs + a] = g.nlc[(size_t)row * cp + id];          // both are [k][W][.] with the same (k, i) order
  else {
    const int r = row - ncl;
    if (r < 10) f.fix[(size_t)r * vs + a] = 0.0;
    else if (r < 20) f.pcr[(size_t)(r - 10) * vs + a] = g.nadd[(size_t)(r - 10) * cp + id];
    else if (r == 20) f.coe[a] = 1.0;
    else if (r < 24) f.eigval[(size_t)(r - 21) * vs + a] = g.neval[(size_t)(r - 21) * cp + id];
    else f.eigvec[(size_t)(r - 24) * vs + a] = g.nevec[(size_t)(r - 24) * cp + id];
  }
}

// submap cloud VS:2957-2975: every keyframe's points in the frame of keyframe 0, stored as PCL floats
__global__ void k_gba_to_ref(int n, int W, const int *__restrict__ offsets, const double *__restrict__ pl, const double *__restrict__ rel /*[W][12]*/,
                             double *__restrict__ out) {
  const int p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= n) return;
  int f = 0;
  while (f + 1 < W && p >= offsets[f + 1]) f++;
  const double *R = rel + 12 * f;
  const double x = pl[3 * (size_t)p], y = pl[3 * (size_t)p + 1], z = pl[3 * (size_t)p + 2];
  out[3 * (size_t)p] = (double)(float)((R[0] * x + R[1] * y + R[2] * z) + R[9]);
  out[3 * (size_t)p + 1] = (double)(float)((R[3] * x + R[4] * y + R[5] * z) + R[10]);
  out[3 * (size_t)p + 2] = (double)(float)((R[6] * x + R[7] * y + R[8] * z) + R[11]);
}

// ---------------------------------------------------------------- host side
struct GbaStore {
  GbaView v{};
  int cap_pts = 0, cap_hash = 0;
  double *d_pl = nullptr;      // device copy of the local points [n][3]
  int *h_cnt = nullptr;        // pinned
  std::vector<void *> node_bufs;

  void free_nodes() { for (void *p : node_bufs) hipFree(p); node_bufs.clear(); v.cap = 0; }
  void free_all() {
    free_nodes();
    hipFree(v.hkeys); hipFree(v.hvals); hipFree(v.pw); hipFree(v.pframe); hipFree(v.pnode); hipFree(d_pl); hipFree(v.cnt); hipFree(v.poses); hipFree(v.offsets);
    if (h_cnt) hipHostFree(h_cnt);
    *this = GbaStore();
  }
};

#define GBACHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { err = std::string(#x) + ": " + hipGetErrorString(e_); return VBA_ERR_HIP; } } while (0)

inline int gba_alloc_nodes(GbaStore &s, int cap, int W, std::string &err) {
  s.free_nodes();
  const size_t cp = (size_t)cap;
  auto al = [&](void **p, size_t bytes) { hipError_t e = hipMalloc(p, bytes); if (e == hipSuccess) s.node_bufs.push_back(*p); return e; };
  GBACHK(al((void **)&s.v.nadd, 10 * cp * 8)); GBACHK(al((void **)&s.v.nlc, 10 * cp * W * 8)); GBACHK(al((void **)&s.v.ncenter, 3 * cp * 8));
  GBACHK(al((void **)&s.v.nql, cp * 4)); GBACHK(al((void **)&s.v.nchild, cp * 4)); GBACHK(al((void **)&s.v.nfac, cp * 4)); GBACHK(al((void **)&s.v.nlayer, cp));
  GBACHK(al((void **)&s.v.neval, 3 * cp * 8)); GBACHK(al((void **)&s.v.nevec, 9 * cp * 8));
  s.v.cap = cap; s.v.W = W;
  return VBA_OK;
}

// Builds the octree of one keyframe window and leaves the planar voxels in device node storage; *n_factors = their count.
// pl may be a host or device pointer ([n][3] local points, keyframe i = rows offsets[i]..offsets[i+1]).
inline int gba_build(GbaStore &s, hipStream_t st, int W, const int *offsets, const double *pl, const double *poses, const GbaParams &P, int *n_factors,
                     std::string &err) {
  const int n = offsets[W];
  *n_factors = 0;
  if (!s.h_cnt) {
    GBACHK(hipHostMalloc((void **)&s.h_cnt, GCNT_N * sizeof(int), hipHostMallocDefault));
    GBACHK(hipMalloc((void **)&s.v.cnt, GCNT_N * sizeof(int)));
    GBACHK(hipMalloc((void **)&s.v.poses, VBA_MAX_WIN * 12 * sizeof(double)));
    GBACHK(hipMalloc((void **)&s.v.offsets, (VBA_MAX_WIN + 1) * sizeof(int)));
  }
  if (n > s.cap_pts) {
    hipFree(s.v.pw); hipFree(s.v.pframe); hipFree(s.v.pnode); hipFree(s.d_pl);
    const size_t c = (size_t)n + n / 4 + 1024;
    GBACHK(hipMalloc((void **)&s.v.pw, 3 * c * 8)); GBACHK(hipMalloc((void **)&s.v.pframe, c * 4)); GBACHK(hipMalloc((void **)&s.v.pnode, c * 4));
    GBACHK(hipMalloc((void **)&s.d_pl, 3 * c * 8));
    s.cap_pts = (int)c;
  }
  int hcap = 1 << 16;
  while (hcap < 2 * n && hcap < (1 << 28)) hcap <<= 1;
  if (hcap > s.cap_hash) {
    hipFree(s.v.hkeys); hipFree(s.v.hvals);
    GBACHK(hipMalloc((void **)&s.v.hkeys, (size_t)hcap * 8)); GBACHK(hipMalloc((void **)&s.v.hvals, (size_t)hcap * 4));
    s.cap_hash = hcap;
  }
  s.v.hmask = (unsigned int)(s.cap_hash - 1);
  s.v.npts = n;
  GBACHK(hipMemcpyAsync(s.d_pl, pl, (size_t)n * 3 * 8, hipMemcpyDefault, st));
  s.v.pl = s.d_pl;
  GBACHK(hipMemcpyAsync(s.v.poses, poses, (size_t)W * 12 * 8, hipMemcpyHostToDevice, st));
  GBACHK(hipMemcpyAsync(s.v.offsets, offsets, (size_t)(W + 1) * 4, hipMemcpyHostToDevice, st));
  if (s.v.cap == 0 || s.v.W != W) { int r = gba_alloc_nodes(s, 1 << 17, W, err); if (r) return r; }
  const dim3 b(256), gp((n + 255) / 256);
  for (int attempt = 0; attempt < 8; attempt++) {
    const size_t cp = (size_t)s.v.cap;
    GBACHK(hipMemsetAsync(s.v.cnt, 0, GCNT_N * sizeof(int), st));
    GBACHK(hipMemsetAsync(s.v.hkeys, 0xFF, (size_t)s.cap_hash * 8, st));
    GBACHK(hipMemsetAsync(s.v.nadd, 0, 10 * cp * 8, st));
    GBACHK(hipMemsetAsync(s.v.nlc, 0, 10 * cp * W * 8, st));
    if (n > 0) {
      hipLaunchKernelGGL(k_gba_keys, gp, b, 0, st, s.v, P);
      hipLaunchKernelGGL(k_gba_roots, dim3((s.cap_hash + 4095) / 4096), b, 0, st, s.v, P);
      hipLaunchKernelGGL(k_gba_rootid, gp, b, 0, st, s.v);
      for (int L = 0; L <= P.max_layer; L++) {
        hipLaunchKernelGGL(k_gba_accum, gp, b, 0, st, s.v);
        hipLaunchKernelGGL(k_gba_decide, dim3((s.v.cap + 255) / 256), b, 0, st, s.v, P, L);
        if (L < P.max_layer) hipLaunchKernelGGL(k_gba_descend, gp, b, 0, st, s.v);
      }
    }
    GBACHK(hipGetLastError());
    GBACHK(hipStreamSynchronize(st));   // drain first (see map_read_counters)
    GBACHK(hipMemcpyAsync(s.h_cnt, s.v.cnt, GCNT_N * sizeof(int), hipMemcpyDeviceToHost, st));
    GBACHK(hipStreamSynchronize(st));
    if (s.h_cnt[GCNT_OVERFLOW] == 2) { err = "keyframe point outside the 21-bit voxel index range"; return VBA_ERR_CAPACITY; }
    if (!s.h_cnt[GCNT_OVERFLOW]) { *n_factors = s.h_cnt[GCNT_FACTORS]; return VBA_OK; }
    int want = s.v.cap * 2;
    while (want < s.h_cnt[GCNT_NODES] + 64) want *= 2;
    int r = gba_alloc_nodes(s, want, W, err);
    if (r) return r;
  }
  err = "octree node capacity";
  return VBA_ERR_CAPACITY;
}

}  // namespace vba
