// libvoxelba.so — implementation of include/voxelba.h for MI355X (gfx950).
// Host side: context / HBM store management, the three LM drivers (voxel_map.hpp:342-976) and the IMU factor;
// device side: the kernels in vba_kernels_factor.hpp / vba_kernels_map.hpp.  No CPU compute fallback exists.
#include "../../include/voxelba.h"

#define VBA_MAX_WIN_DEV VBA_MAX_WIN
#include "vba_kernels_factor.hpp"
#include "vba_kernels_h3.hpp"
#include "vba_kernels_map.hpp"
#include "vba_kernels_lm.hpp"
#include "vba_kernels_li.hpp"
#include "vba_kernels_scan.hpp"
#include "vba_kernels_gba.hpp"
#include "vba_kernels_big.hpp"
#include "vba_kernels_kd.hpp"
#include "vba_io.hpp"
#include <cstddef>
#include "vba_hostmath.hpp"

#include <hip/hip_runtime.h>
#include <rccl/rccl.h>     // TYPES only: the entry points are resolved at run time (rccl_api below), the library does not link librccl
#include <dlfcn.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <atomic>
#include <string>
#include <thread>
#include <vector>
#include <array>
#include <map>
#include <chrono>
#include <deque>

using namespace vba;

#define HIPCHK(ctx, expr)                                                                        \
  do {                                                                                           \
    hipError_t _e = (expr);                                                                      \
    if (_e != hipSuccess) {                                                                      \
      (ctx)->set_error(std::string(#expr) + ": " + hipGetErrorString(_e));                       \
      return VBA_ERR_HIP;                                                                        \
    }                                                                                            \
  } while (0)

namespace {
struct TimedSpan { hipEvent_t a, b; };

// RCCL entry points, resolved on first use: the copy the process has ALREADY loaded wins (a host program that imported torch carries
// torch/lib/librccl.so; binding to a second build would split the communicator state), then the system librccl.so.1.  A process
// that never asks for the in-library exchange step (vba_rccl_init / vba_set_rccl_comm / vba_rccl_get_unique_id) needs no RCCL at all.
struct RcclApi {
  ncclResult_t (*GetUniqueId)(ncclUniqueId *) = nullptr;
  ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*AllReduce)(const void *, void *, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*AllGather)(const void *, void *, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
  const char *(*GetErrorString)(ncclResult_t) = nullptr;
  bool ok = false;
  std::string why;
};
const RcclApi &rccl_api() {
  static const RcclApi api = [] {
    RcclApi a;
    void *h = nullptr;
    if (dlsym(RTLD_DEFAULT, "ncclAllReduce")) h = RTLD_DEFAULT;            // already in the process (e.g. torch's copy)
    if (!h) h = dlopen("librccl.so.1", RTLD_NOW | RTLD_LOCAL);
    if (!h) h = dlopen("librccl.so", RTLD_NOW | RTLD_LOCAL);
    if (!h) h = dlopen("/opt/rocm/lib/librccl.so.1", RTLD_NOW | RTLD_LOCAL);
    if (!h) { const char *e = dlerror(); a.why = std::string("librccl.so.1 not found: ") + (e ? e : "?"); return a; }
    bool all = true;
    auto get = [&](const char *name) { void *s = dlsym(h, name); if (!s) { all = false; a.why = std::string("RCCL lacks ") + name; } return s; };
    a.GetUniqueId = (decltype(a.GetUniqueId))get("ncclGetUniqueId");
    a.CommInitRank = (decltype(a.CommInitRank))get("ncclCommInitRank");
    a.CommDestroy = (decltype(a.CommDestroy))get("ncclCommDestroy");
    a.AllReduce = (decltype(a.AllReduce))get("ncclAllReduce");
    a.AllGather = (decltype(a.AllGather))get("ncclAllGather");
    a.GetErrorString = (decltype(a.GetErrorString))get("ncclGetErrorString");
    a.ok = all;
    return a;
  }();
  return api;
}

// Diagnostic switches (in-kernel stamps, host-side phase timers, ablation forms) exist only in a -DVBA_DIAG build (make diag ->
// libvoxelba_diag.so, tools/README.md); the shipped library reads no environment variable.
inline const char *diag_env(const char *name) {
#ifdef VBA_DIAG
  return std::getenv(name);
#else
  (void)name;
  return nullptr;
#endif
}
}

struct vba_ctx {
  vba_options opt;
  int device = 0;
  hipStream_t stream = nullptr;
  bool own_stream = false;
  std::string err;

  // factor store (HBM, SoA)
  FactorView fv{};
  int nvox = 0;   // voxels stored
  int nvox_global = 0;   // the same summed over the ranks (set by vba_lm_begin when the factor store is sharded)
  int cap = 0;    // capacity = stride
  double *d_poses = nullptr;     // [W][12]
  double *d_partial = nullptr;   // workgroup partials
  size_t partial_doubles = 0;
  double *d_out = nullptr;       // reduced Hessian pass, tile layout (vba_kernels_factor.hpp)
  double *d_full = nullptr;      // the same in full layout [H | g | r] for host consumers
  double *d_scal = nullptr;      // reduced residual scalar
  double *h_pin = nullptr;       // pinned host staging
  size_t pin_doubles = 0;
  void *d_stage = nullptr;       // AoS upload staging
  size_t stage_bytes = 0;

  // multi-GPU
  vba_allreduce_fn allreduce = nullptr;
  void *allreduce_user = nullptr;
  int rank = 0, n_ranks = 1;
  bool force_collective = false;  // vba_options::force_collective (rehearsal: run the exchange step with one rank)
  int max_blocks_hess = 256;      // vba_options::hessian_workgroups
  int residual_vpl_from = 45000;  // vba_options::residual_vpl_from
  bool use_h3 = false;            // vba_options::hessian_compact_tiles != 0
  ncclComm_t comm = nullptr;      // RCCL communicator: the exchange step is issued by the library on the context's stream
  bool own_comm = false;
  bool collective_off = false;    // replica phases (bottom-layer HBA windows) run their LM loops without the exchange step
  bool collective() const { return !collective_off && (allreduce || comm) && (n_ranks > 1 || force_collective); }

  // timing
  bool timing = false;
  std::string timing_only;        // when non-empty only this kernel family is bracketed by events
  int timing_every = 1; unsigned timing_ctr = 0;   // bracket every n-th launch of the selected family
  int lm_spec = LM_SPEC;          // damping candidates per solve launch
  std::vector<std::array<double, 450>> covinv_cache; size_t covinv_next = 0;   // li_ba_device: (cov, cov^-1) of recently seen IMU factors
  std::map<std::string, std::vector<TimedSpan>> spans;

  // device-resident LM state (lm_begin / lm_iterate / lm_end)
  LmDev *d_lm = nullptr;
  LmDev *h_lm = nullptr;          // pinned mirror (download side)
  static const int kLmRing = 8;
  LmDev *h_lm_up[kLmRing] = {nullptr};   // pinned upload ring: lm_begin never has to drain the stream
  hipEvent_t lm_up_ev[kLmRing] = {nullptr};
  int lm_up_next = 0;
  double *d_raw = nullptr;        // last valid all-reduced [H|g|r] (multi-rank only; single rank reads d_out in place)
  struct { bool active = false; int thd_num = 2; bool have_hess = false; bool pending_update = false; int k4_nb = 0; } lm;   // pending_update: the accept/reject step of the last iteration rides in the next Hessian pass
  int k4part_cap = 0;
  double *d_k4part = nullptr;     // residual-pass partials of the LM loop (the Hessian pass reuses d_partial while they are still read)   // have_hess: [H|g|r] of the next solve is already reduced (multi-rank)
  std::vector<double> trace;

  // device-resident LI-BA (vba_kernels_li.hpp)
  LiDev *d_li = nullptr;
  double *d_imu = nullptr, *d_himu = nullptr, *d_gimu = nullptr;

  MapStore map;
  GbaStore gba;
  BigStore big;                   // arbitrary-window path (top-level global BA)
  double *d_kdtree[2] = {nullptr, nullptr};   // pl_tree of the initialisation odometry (float-valued xyz), ping-pong for the re-sampling
  size_t kd_cap = 0; int kd_n = 0, kd_cur = 0;
  double *d_refpts = nullptr;     // submap cloud staging (HBA_add_edge)
  size_t refpts_doubles = 0;
  double *d_lipack = nullptr; size_t lipack_doubles = 0;       // li_ba_device: results gathered for one D2H copy
  double *d_liscr = nullptr; size_t liscr_doubles = 0;         // k_li_solve at W > 10: staged matrix / L outside the LDS
  double *d_hba_all = nullptr; size_t hba_all_doubles = 0;   // vba_hba_global: keyframe clouds + submap clouds, kept across calls
  std::vector<vba_ctx *> hba_workers;                         // vba_hba_global: extra contexts (own stream, own octree) that optimise bottom-layer windows side by side

  void set_error(const std::string &s) { err = s; }
};

namespace {

// damping candidates per solve launch (vba_kernels_lm.hpp, "Speculative damping"); 1 = the plain sequential solve.  Read when a
// context is created (vba_options::lm_spec; tests compare the two forms bit for bit).
static const int kMaxDevices = 64;         // per-device "kernel attribute set" flags
static const int kMaxBlocksHess = 256;     // upper bound of vba_options::hessian_workgroups (sizes the partial slab): one workgroup per CU

int nout_of(int W) { return 36 * W * W + 6 * W + 1; }   // full layout [H | g | r]
int nout_tl(int W) {                                      // tile layout produced by k_hessian2 (HessCfg2<W>::NOUT2)
  const int nt16 = (6 * W + 15) / 16;
  return nt16 * (nt16 + 1) / 2 * 256 + 27 * W + 1;
}

static inline bool span_on(vba_ctx *c, const char *name) { return c->timing && (c->timing_only.empty() || c->timing_only == name); }
void span_begin(vba_ctx *c, const char *name, TimedSpan &s) {
  s.a = s.b = nullptr;
  if (!span_on(c, name)) return;
  if (c->timing_every > 1 && (c->timing_ctr++ % c->timing_every) != 0) return;   // sampled bracketing (vba_timing_sample_every)
  hipEventCreate(&s.a); hipEventCreate(&s.b);
  hipEventRecord(s.a, c->stream);
}
void span_end(vba_ctx *c, const char *name, TimedSpan &s) {
  if (!s.a) return;
  hipEventRecord(s.b, c->stream);
  c->spans[name].push_back(s);
}

int ensure_pin(vba_ctx *c, size_t n) {
  if (n <= c->pin_doubles) return VBA_OK;
  if (c->h_pin) hipHostFree(c->h_pin);
  c->h_pin = nullptr; c->pin_doubles = 0;
  HIPCHK(c, hipHostMalloc((void **)&c->h_pin, n * sizeof(double), hipHostMallocDefault));
  c->pin_doubles = n;
  return VBA_OK;
}
int ensure_stage(vba_ctx *c, size_t bytes) {
  if (bytes <= c->stage_bytes) return VBA_OK;
  if (c->d_stage) hipFree(c->d_stage);
  c->d_stage = nullptr; c->stage_bytes = 0;
  HIPCHK(c, hipMalloc(&c->d_stage, bytes));
  c->stage_bytes = bytes;
  return VBA_OK;
}

// (re)allocate the SoA factor store with stride newcap, preserving the first nvox voxels
int factor_reserve(vba_ctx *c, int need) {
  if (need <= c->cap) return VBA_OK;
  int newcap = c->cap ? c->cap : 4096;
  while (newcap < need) newcap *= 2;
  // the SoA rows are `stride` doubles apart and every pass streams ~100 of them at the same offset: a power-of-two stride would
  // put all those streams on the same HBM channels / cache sets, so the stride is skewed by an odd number of 512-byte blocks
  newcap = (newcap + 63) / 64 * 64 + 64 * 33;
  const int W = c->opt.win_size;
  FactorView n = c->fv;
  n.vs = newcap; n.W = W;
  const size_t rows[6] = {(size_t)10 * W, 10, 1, 3, 9, 10};
  double **np[6] = {&n.cl, &n.fix, &n.coe, &n.eigval, &n.eigvec, &n.pcr};
  double *op[6] = {c->fv.cl, c->fv.fix, c->fv.coe, c->fv.eigval, c->fv.eigvec, c->fv.pcr};
  for (int k = 0; k < 6; k++) {
    HIPCHK(c, hipMalloc((void **)np[k], rows[k] * newcap * sizeof(double)));
    HIPCHK(c, hipMemsetAsync(*np[k], 0, rows[k] * newcap * sizeof(double), c->stream));
    if (c->nvox > 0 && op[k])
      HIPCHK(c, hipMemcpy2DAsync(*np[k], (size_t)newcap * sizeof(double), op[k], (size_t)c->cap * sizeof(double),
                                 (size_t)c->nvox * sizeof(double), rows[k], hipMemcpyDeviceToDevice, c->stream));
  }
  unsigned int *oocc = c->fv.occ;
  int *otiles = c->fv.tiles;
  HIPCHK(c, hipMalloc((void **)&n.tiles, ((size_t)4 * newcap + 16) * sizeof(int)));
  HIPCHK(c, hipMemsetAsync(n.tiles, 0, ((size_t)4 * newcap + 16) * sizeof(int), c->stream));
  HIPCHK(c, hipMalloc((void **)&n.occ, (size_t)newcap * sizeof(unsigned int)));
  HIPCHK(c, hipMemsetAsync(n.occ, 0, (size_t)newcap * sizeof(unsigned int), c->stream));
  if (c->nvox > 0 && oocc) HIPCHK(c, hipMemcpyAsync(n.occ, oocc, (size_t)c->nvox * sizeof(unsigned int), hipMemcpyDeviceToDevice, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  for (int k = 0; k < 6; k++) if (op[k]) hipFree(op[k]);
  if (oocc) hipFree(oocc);
  if (otiles) hipFree(otiles);
  c->fv = n;
  if (c->nvox > 0 && c->use_h3) hipLaunchKernelGGL(k_factor_tiles, dim3(1), dim3(1024), 0, c->stream, c->fv, c->nvox);   // (the table lives in the new allocation)
  c->cap = newcap;
  return VBA_OK;
}

// the occupancy masks of voxels [base, base + n) follow every write of the cluster rows
void factor_update_mask(vba_ctx *c, int base, int n) {
  if (n > 0) hipLaunchKernelGGL(k_factor_mask, dim3((n + 255) / 256), dim3(256), 0, c->stream, c->fv, base, n);
  // the Hessian pass' tile table of the whole store [0, base + n) (vba_kernels_h3.hpp)
  if (base + n > 0 && c->use_h3) hipLaunchKernelGGL(k_factor_tiles, dim3(1), dim3(1024), 0, c->stream, c->fv, base + n);
}

int upload_poses(vba_ctx *c, const double *poses) {
  const int W = c->opt.win_size;
  int st = ensure_pin(c, 65536);
  if (st) return st;
  std::memcpy(c->h_pin, poses, (size_t)W * 12 * sizeof(double));
  HIPCHK(c, hipMemcpyAsync(c->d_poses, c->h_pin, (size_t)W * 12 * sizeof(double), hipMemcpyHostToDevice, c->stream));
  return VBA_OK;
}

template <int W>
int launch_hessian2_t(vba_ctx *c, const double *poses_dev, const int *gate, int head, int end, int *nblocks_out, LmDev *lm, const double *k4p, int k4nb,
                      const LiJob &li, size_t li_lds) {
  using C = HessCfg2<W>;
  const int ntiles = (end - head + C::TV - 1) / C::TV;
  const int maxb = li.dev ? c->max_blocks_hess - 1 : c->max_blocks_hess;      // (the IMU workgroup of LI-BA takes a CU of its own)
  int nb = ntiles < maxb ? ntiles : maxb;
  if (nb < 1) nb = 1;
  static bool attr_set[kMaxDevices] = {false};      // the attribute is per DEVICE (a process may hold contexts on several)
  if (!attr_set[c->device % kMaxDevices]) {
    // (the IMU workgroup of LI-BA needs up to 150 KB at W = 16; a lidar-only launch asks for C::LDS_BYTES)
    const size_t li_max = 150 * 1024;
    hipFuncSetAttribute((const void *)k_hessian2<W>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(C::LDS_BYTES > li_max ? C::LDS_BYTES : li_max));
    attr_set[c->device % kMaxDevices] = true;
  }
  long long *stamps = nullptr;
  static const bool want_stamps = diag_env("VBA_K3_STAMPS") != nullptr;     // -DVBA_DIAG builds only
  if (want_stamps) {
    static long long *d_st = nullptr;
    if (!d_st) hipMalloc((void **)&d_st, (size_t)kMaxBlocksHess * 16 * 8);
    hipMemsetAsync(d_st, 0, (size_t)kMaxBlocksHess * 16 * 8, c->stream);
    stamps = d_st;
  }
  const size_t lds = (li.dev && li_lds > C::LDS_BYTES) ? li_lds : C::LDS_BYTES;
  hipLaunchKernelGGL(k_hessian2<W>, dim3(nb + (li.dev ? 1 : 0)), dim3(C::NT), lds, c->stream, c->fv, poses_dev, head, end, ntiles, c->d_partial, gate, stamps, lm, k4p, k4nb,
                     nb, li);
  if (want_stamps) {
    std::vector<long long> h((size_t)nb * 16);
    hipStreamSynchronize(c->stream);
    hipMemcpy(h.data(), stamps, h.size() * 8, hipMemcpyDeviceToHost);
    long long t0 = h[0];
    for (int b = 0; b < nb; b++) if (h[(size_t)b * 16] && h[(size_t)b * 16] < t0) t0 = h[(size_t)b * 16];
    for (int b : {0, 1, nb / 2, nb - 1}) {
      fprintf(stderr, "[k3 stamps] wg %d:", b);
      for (int i = 0; i < 15; i++) fprintf(stderr, " %lld", h[(size_t)b * 16 + i] ? h[(size_t)b * 16 + i] - t0 : -1);
      fprintf(stderr, "\n");
    }
    long long tmax = 0;
    for (int b = 0; b < nb; b++) if (h[(size_t)b * 16 + 14] - t0 > tmax) tmax = h[(size_t)b * 16 + 14] - t0;
    fprintf(stderr, "[k3 stamps] last workgroup ends at %lld ticks (100 MHz wall clock: 1 tick = 10 ns)\n", tmax);
  }
  *nblocks_out = nb;
  return VBA_OK;
}

template <int W>
int launch_hessian3_t(vba_ctx *c, const double *poses_dev, const int *gate, int *nblocks_out, LmDev *lm, const double *k4p, int k4nb, const LiJob &li, size_t li_lds) {
  using C = HessCfg3<W>;
  const int nb = li.dev ? c->max_blocks_hess - 1 : c->max_blocks_hess;      // (the IMU workgroup of LI-BA takes a CU of its own)
  static bool attr_set[kMaxDevices] = {false};
  if (!attr_set[c->device % kMaxDevices]) {
    const size_t li_max = 150 * 1024;
    hipFuncSetAttribute((const void *)k_hessian3<W>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(C::LDS_BYTES > li_max ? C::LDS_BYTES : li_max));
    attr_set[c->device % kMaxDevices] = true;
  }
  const size_t lds = (li.dev && li_lds > C::LDS_BYTES) ? li_lds : C::LDS_BYTES;
  long long *stamps = nullptr;
  static const bool want_stamps = diag_env("VBA_K3_STAMPS") != nullptr;     // -DVBA_DIAG builds only
  if (want_stamps) {
    static long long *d_st = nullptr;
    if (!d_st) hipMalloc((void **)&d_st, (size_t)kMaxBlocksHess * 16 * 8);
    hipMemsetAsync(d_st, 0, (size_t)kMaxBlocksHess * 16 * 8, c->stream);
    stamps = d_st;
  }
  hipLaunchKernelGGL(k_hessian3<W>, dim3(nb + (li.dev ? 1 : 0)), dim3(C::NT), lds, c->stream, c->fv, poses_dev, c->nvox, c->d_partial, gate, lm, k4p, k4nb, nb, li, stamps);
  if (want_stamps) {
    std::vector<long long> h((size_t)nb * 16);
    hipStreamSynchronize(c->stream);
    hipMemcpy(h.data(), stamps, h.size() * 8, hipMemcpyDeviceToHost);
    long long t0 = h[0];
    for (int b = 0; b < nb; b++) if (h[(size_t)b * 16] && h[(size_t)b * 16] < t0) t0 = h[(size_t)b * 16];
    for (int b : {0, 1, nb / 2, nb - 1}) {
      fprintf(stderr, "[k3 stamps] wg %d:", b);
      for (int i = 0; i < 16; i++) fprintf(stderr, " %lld", h[(size_t)b * 16 + i] ? h[(size_t)b * 16 + i] - t0 : -1);
      fprintf(stderr, "\n");
    }
    long long tmax = 0;
    for (int b = 0; b < nb; b++) if (h[(size_t)b * 16 + 15] - t0 > tmax) tmax = h[(size_t)b * 16 + 15] - t0;
    fprintf(stderr, "[k3 stamps] last workgroup ends at %lld ticks (100 MHz wall clock: 1 tick = 10 ns); per workgroup: start, prologue, then per tile [A, B, combine, E]\n", tmax);
  }
  *nblocks_out = nb;
  return VBA_OK;
}

int launch_hessian(vba_ctx *c, const double *pd, const int *gate, int head, int end, int *nb, LmDev *lm = nullptr, const double *k4p = nullptr, int k4nb = 0,
                   const LiJob &li = LiJob{}, size_t li_lds = 0) {
  // whole store, W <= 10: the occupancy-compact pass (vba_kernels_h3.hpp); sub-ranges and wider windows: the dense-tile pass
  if (head == 0 && end == c->nvox && c->opt.win_size <= 10 && c->use_h3) {
    switch (c->opt.win_size) {
#define VBA_H3_CASE(WW) case WW: return launch_hessian3_t<WW>(c, pd, gate, nb, lm, k4p, k4nb, li, li_lds);
      VBA_H3_CASE(2) VBA_H3_CASE(3) VBA_H3_CASE(4) VBA_H3_CASE(5) VBA_H3_CASE(6) VBA_H3_CASE(7) VBA_H3_CASE(8) VBA_H3_CASE(9) VBA_H3_CASE(10)
#undef VBA_H3_CASE
    }
  }
  switch (c->opt.win_size) {
#define VBA_H_CASE(WW) case WW: return launch_hessian2_t<WW>(c, pd, gate, head, end, nb, lm, k4p, k4nb, li, li_lds);
    VBA_H_CASE(2) VBA_H_CASE(3) VBA_H_CASE(4) VBA_H_CASE(5) VBA_H_CASE(6) VBA_H_CASE(7) VBA_H_CASE(8) VBA_H_CASE(9) VBA_H_CASE(10)
    VBA_H_CASE(11) VBA_H_CASE(12) VBA_H_CASE(13) VBA_H_CASE(14) VBA_H_CASE(15) VBA_H_CASE(16)
#undef VBA_H_CASE
    default: return VBA_ERR_UNSUPPORTED_WINDOW;
  }
}

#ifndef VBA_K4_TV
#define VBA_K4_TV 32        // voxels per workgroup of the residual pass (tools/ builds 16 / 64 for comparison)
#endif
// number of workgroups (= residual partials) of the residual pass over n voxels: small stores (latency-bound) take the
// slot-parallel kernel k_residual_s, large ones (throughput-bound) the voxel-per-lane kernel k_residual_v (vba_kernels_factor.hpp)
// (measured crossover on MI355X, hesai200k_w10 scene tiled: 36.8k voxels 6.4 vs 7.1 us, 55.1k voxels 8.7 vs 7.5 us)
// (the crossover is vba_options::residual_vpl_from, default 45000)
inline bool residual_vpl(const vba_ctx *c, int n) { return n > c->residual_vpl_from; }
inline int residual_nb(const vba_ctx *c, int n) { return residual_vpl(c, n) ? (n + 63) / 64 : (n + VBA_K4_TV - 1) / VBA_K4_TV; }

// diagnostic (VBA_K4_STAMPS=1): in-kernel clock stamps of a separate STAMPS instantiation; the production kernel holds none
void k4_stamps_dump(vba_ctx *c, int nb, long long *d_st) {
  const int n = nb < 2048 ? nb : 2048;
  std::vector<long long> h((size_t)n * 4);
  hipStreamSynchronize(c->stream);
  hipMemcpy(h.data(), d_st, h.size() * 8, hipMemcpyDeviceToHost);
  long long t0 = h[0];
  for (int b = 0; b < n; b++) if (h[b * 4] && h[b * 4] < t0) t0 = h[b * 4];
  double a = 0, e = 0, w = 0, last = 0;
  for (int b = 0; b < n; b++) { a += h[b * 4 + 1] - h[b * 4]; e += h[b * 4 + 2] - h[b * 4 + 1]; w += h[b * 4 + 3] - h[b * 4 + 2]; if (h[b * 4 + 3] - t0 > last) last = h[b * 4 + 3] - t0; }
  fprintf(stderr, "[k4 stamps] %d workgroups: loads+transforms %.0f, frame sum + eigen %.0f, stores+reduce %.0f cycles (mean per workgroup); last one ends at %.0f cycles\n", n, a / n, e / n, w / n, last);
}

// residual pass over voxels [head, end) (end > head); partials (one per workgroup) go to dst or d_partial; returns their number
int launch_residual(vba_ctx *c, const double *pd, const int *gate, int head, int end, double *dst = nullptr) {
  double *part = dst ? dst : c->d_partial;
  const int nb = residual_nb(c, end - head);
  static const bool want_stamps = diag_env("VBA_K4_STAMPS") != nullptr;
  static long long *d_st = nullptr;
  if (want_stamps) {
    if (!d_st) hipMalloc((void **)&d_st, 2048 * 4 * 8);
    hipMemsetAsync(d_st, 0, 2048 * 4 * 8, c->stream);
  }
  if (residual_vpl(c, end - head)) {
#define VBA_RESV_CASE(WW) case WW: \
    if (want_stamps) hipLaunchKernelGGL((k_residual_v<WW, true>), dim3(nb), dim3(64), 0, c->stream, c->fv, pd, head, end, part, gate, d_st); \
    else hipLaunchKernelGGL((k_residual_v<WW, false>), dim3(nb), dim3(64), 0, c->stream, c->fv, pd, head, end, part, gate, (long long *)nullptr); break;
    switch (c->opt.win_size) {
      VBA_RESV_CASE(2) VBA_RESV_CASE(3) VBA_RESV_CASE(4) VBA_RESV_CASE(5) VBA_RESV_CASE(6) VBA_RESV_CASE(7) VBA_RESV_CASE(8) VBA_RESV_CASE(9) VBA_RESV_CASE(10)
      VBA_RESV_CASE(11) VBA_RESV_CASE(12) VBA_RESV_CASE(13) VBA_RESV_CASE(14) VBA_RESV_CASE(15) VBA_RESV_CASE(16)
    }
#undef VBA_RESV_CASE
  } else {
#define VBA_RES_CASE(WW) case WW: { using RC = ResCfg<WW, VBA_K4_TV>; \
    if (want_stamps) hipLaunchKernelGGL((k_residual_s<WW, VBA_K4_TV, true>), dim3(nb), dim3(RC::NT), 0, c->stream, c->fv, pd, head, end, part, gate, d_st); \
    else hipLaunchKernelGGL((k_residual_s<WW, VBA_K4_TV, false>), dim3(nb), dim3(RC::NT), 0, c->stream, c->fv, pd, head, end, part, gate, (long long *)nullptr); break; }
    switch (c->opt.win_size) {
      VBA_RES_CASE(2) VBA_RES_CASE(3) VBA_RES_CASE(4) VBA_RES_CASE(5) VBA_RES_CASE(6) VBA_RES_CASE(7) VBA_RES_CASE(8) VBA_RES_CASE(9) VBA_RES_CASE(10)
      VBA_RES_CASE(11) VBA_RES_CASE(12) VBA_RES_CASE(13) VBA_RES_CASE(14) VBA_RES_CASE(15) VBA_RES_CASE(16)
    }
#undef VBA_RES_CASE
  }
  if (want_stamps) k4_stamps_dump(c, nb, d_st);
  return nb;
}

// SUM all-reduce of n doubles in HBM across the ranks, ordered on the context's stream: RCCL when the context holds a
// communicator (vba_rccl_init / vba_set_rccl_comm), else the host program's hook (gloo rehearsals on the CPU side of tests).
int ctx_allreduce(vba_ctx *c, double *buf, size_t n) {
  if (c->comm) {
    const RcclApi &R = rccl_api();
    const ncclResult_t r = R.AllReduce(buf, buf, n, ncclDouble, ncclSum, c->comm, c->stream);
    if (r != ncclSuccess) { c->set_error(std::string("ncclAllReduce: ") + R.GetErrorString(r)); return VBA_ERR_HIP; }
    return VBA_OK;
  }
  if (!c->allreduce) { c->set_error("no collective configured"); return VBA_ERR_BAD_ARG; }
  if (c->allreduce(c->allreduce_user, buf, n, c->stream)) { c->set_error("allreduce hook failed"); return VBA_ERR_HIP; }
  return VBA_OK;
}
// All-gather in place: buf holds n_ranks chunks of `chunk` doubles, rank r has filled chunk r.  RCCL moves every chunk once;
// the hook (SUM only) emulates it by zeroing the foreign chunks first.
int ctx_allgather(vba_ctx *c, double *buf, size_t chunk) {
  if (chunk == 0) return VBA_OK;
  if (c->comm) {
    const RcclApi &R = rccl_api();
    const ncclResult_t r = R.AllGather(buf + (size_t)c->rank * chunk, buf, chunk, ncclDouble, c->comm, c->stream);
    if (r != ncclSuccess) { c->set_error(std::string("ncclAllGather: ") + R.GetErrorString(r)); return VBA_ERR_HIP; }
    return VBA_OK;
  }
  for (int r = 0; r < c->n_ranks; r++)
    if (r != c->rank) HIPCHK(c, hipMemsetAsync(buf + (size_t)r * chunk, 0, chunk * sizeof(double), c->stream));
  return ctx_allreduce(c, buf, chunk * (size_t)c->n_ranks);
}

// device passes on device-resident poses (gate == nullptr: unconditional)
int hessian_pass(vba_ctx *c, const double *poses_dev, const int *gate, int head, int end, LmDev *lm = nullptr, const double *k4p = nullptr, int k4nb = 0,
                 const LiJob &li = LiJob{}, size_t li_lds = 0) {
  const int W = c->opt.win_size, nout = nout_tl(W);
  if (end <= head) {
    HIPCHK(c, hipMemsetAsync(c->d_out, 0, (size_t)nout * sizeof(double), c->stream));
  } else {
    int nb = 0;
    TimedSpan s1{}, s2{};
    span_begin(c, "hessian", s1);
    int st = launch_hessian(c, poses_dev, gate, head, end, &nb, lm, k4p, k4nb, li, li_lds);
    if (st) return st;
    span_end(c, "hessian", s1);
    span_begin(c, "reduce", s2);
    hipLaunchKernelGGL(k_reduce_partials, dim3((nout + 15) / 16), dim3(256), 0, c->stream, c->d_partial, nb, nout, c->d_out, gate);
    span_end(c, "reduce", s2);
    HIPCHK(c, hipGetLastError());
  }
  if (c->collective()) {
    int rc = ctx_allreduce(c, c->d_out, (size_t)nout);
    if (rc) return rc;
  }
  return VBA_OK;
}

int residual_pass(vba_ctx *c, const double *poses_dev, const int *gate, int head, int end, double *d_scalar_out) {
  if (end <= head) {
    HIPCHK(c, hipMemsetAsync(d_scalar_out, 0, sizeof(double), c->stream));
  } else {
    const int nb = residual_nb(c, end - head);
    if ((size_t)nb > c->partial_doubles) { c->set_error("partial buffer too small"); return VBA_ERR_CAPACITY; }
    TimedSpan s1{}, s2{};
    span_begin(c, "residual", s1);
    launch_residual(c, poses_dev, gate, head, end);
    span_end(c, "residual", s1);
    span_begin(c, "reduce", s2);
    hipLaunchKernelGGL(k_sum_scalar, dim3(1), dim3(256), 0, c->stream, c->d_partial, nb, d_scalar_out, gate);
    span_end(c, "reduce", s2);
    HIPCHK(c, hipGetLastError());
  }
  if (c->collective()) {
    int rc = ctx_allreduce(c, d_scalar_out, 1);
    if (rc) return rc;
  }
  return VBA_OK;
}

// tile layout -> full layout [H | g | r] in d_full (host consumers only; the device LM reads the tile layout directly)
int tiles_to_full(vba_ctx *c, const double *src) {
#define VBA_TF_CASE(WW) case WW: hipLaunchKernelGGL(k_tiles_to_full<WW>, dim3(16), dim3(256), 0, c->stream, src, c->d_full); break;
  switch (c->opt.win_size) {
    VBA_TF_CASE(2) VBA_TF_CASE(3) VBA_TF_CASE(4) VBA_TF_CASE(5) VBA_TF_CASE(6) VBA_TF_CASE(7) VBA_TF_CASE(8) VBA_TF_CASE(9) VBA_TF_CASE(10)
    VBA_TF_CASE(11) VBA_TF_CASE(12) VBA_TF_CASE(13) VBA_TF_CASE(14) VBA_TF_CASE(15) VBA_TF_CASE(16)
    default: return VBA_ERR_UNSUPPORTED_WINDOW;
  }
#undef VBA_TF_CASE
  HIPCHK(c, hipGetLastError());
  return VBA_OK;
}

// device: d_full = [H | g | r] over voxels [head,end) for host poses (+ all-reduce across ranks when configured)
int eval_hessian_dev(vba_ctx *c, const double *poses, int head, int end) {
  int st = upload_poses(c, poses);
  if (st) return st;
  st = hessian_pass(c, c->d_poses, nullptr, head, end);
  if (st) return st;
  return tiles_to_full(c, c->d_out);
}

int eval_residual_dev(vba_ctx *c, const double *poses, int head, int end, double *d_scalar_out) {
  int st = upload_poses(c, poses);
  if (st) return st;
  return residual_pass(c, c->d_poses, nullptr, head, end, d_scalar_out);
}

int ensure_partial(vba_ctx *c, size_t doubles) {
  if (doubles <= c->partial_doubles) return VBA_OK;
  if (c->d_partial) hipFree(c->d_partial);
  c->d_partial = nullptr; c->partial_doubles = 0;
  HIPCHK(c, hipMalloc((void **)&c->d_partial, doubles * sizeof(double)));
  c->partial_doubles = doubles;
  return VBA_OK;
}

// host copies of the reduced device results
int fetch(vba_ctx *c, const double *d_src, size_t n, double *dst) {
  int st = ensure_pin(c, n + 65536);
  if (st) return st;
  double *stage = c->h_pin + 32768;  // poses live in the first part
  HIPCHK(c, hipStreamSynchronize(c->stream));   // drain first: a D2H copy queued behind in-flight kernels completes much later (measured)
  HIPCHK(c, hipMemcpyAsync(stage, d_src, n * sizeof(double), hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  std::memcpy(dst, stage, n * sizeof(double));
  return VBA_OK;
}

// states <-> poses
void states_to_poses(const double *states, int W, double *poses) {
  for (int i = 0; i < W; i++) { std::memcpy(poses + 12 * i, states + 25 * i + 1, 9 * sizeof(double)); std::memcpy(poses + 12 * i + 9, states + 25 * i + 10, 3 * sizeof(double)); }
}

}  // namespace

extern "C" {

void vba_default_options(vba_options *o) {
  std::memset(o, 0, sizeof(*o));
  o->win_size = 10; o->voxel_size = 1.0; o->max_layer = 2; o->max_points = 100; o->min_eigen_value = 0.0025;
  for (int i = 0; i < 4; i++) { o->plane_eigen_value_thre[i] = 0.25; o->min_point[i] = 5; }
  o->imu_coef = 1e-4; o->thread_num = 5; o->device = -1; o->stream = nullptr;
}

const char *vba_status_string(int s) {
  switch (s) {
    case VBA_OK: return "ok";
    case VBA_ERR_NO_DEVICE: return "no HIP device (libvoxelba has no CPU path)";
    case VBA_ERR_BAD_ARG: return "bad argument";
    case VBA_ERR_UNSUPPORTED_WINDOW: return "unsupported window size";
    case VBA_ERR_TOO_FEW_VOXELS: return "too few voxels (reference: 'Too Less Voxel' exit)";
    case VBA_ERR_OPT_STATE: return "opt_state out of range (reference: exit)";
    case VBA_ERR_HIP: return "HIP runtime error";
    case VBA_ERR_CAPACITY: return "capacity exceeded";
    case VBA_ERR_IO: return "file missing or malformed";
    case VBA_ERR_UNSUPPORTED: return "not available in this process (RCCL could not be resolved)";
    default: return "unknown";
  }
}

int vba_create(const vba_options *opt, vba_ctx **out) {
  if (!opt || !out) return VBA_ERR_BAD_ARG;
  *out = nullptr;
  if (opt->win_size < 2 || opt->win_size > VBA_MAX_WIN) return VBA_ERR_UNSUPPORTED_WINDOW;
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return VBA_ERR_NO_DEVICE;
  vba_ctx *c = new vba_ctx();
  c->opt = *opt;
  if (opt->device >= 0) {
    if (hipSetDevice(opt->device) != hipSuccess) { delete c; return VBA_ERR_NO_DEVICE; }
    c->device = opt->device;
  } else {
    hipGetDevice(&c->device);
  }
  c->lm_spec = opt->lm_spec > 0 ? std::min(opt->lm_spec, (int)LM_SPEC) : (int)LM_SPEC;
  c->force_collective = opt->force_collective != 0;
  c->max_blocks_hess = opt->hessian_workgroups > 0 ? std::min(opt->hessian_workgroups, kMaxBlocksHess) : kMaxBlocksHess;
  if (c->max_blocks_hess < 2) c->max_blocks_hess = 2;      // (LI-BA gives one workgroup's CU to the IMU factors)
  c->residual_vpl_from = opt->residual_vpl_from > 0 ? opt->residual_vpl_from : 45000;
  c->use_h3 = opt->hessian_compact_tiles != 0;
  if (opt->stream) { c->stream = (hipStream_t)opt->stream; c->own_stream = false; }
  else {
    if (hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess) { delete c; return VBA_ERR_HIP; }
    c->own_stream = true;
  }
  const int W = opt->win_size, nout = nout_of(W);
  c->fv.W = W;
  if (hipMalloc((void **)&c->d_poses, (size_t)VBA_MAX_WIN * 12 * sizeof(double)) != hipSuccess ||
      hipMalloc((void **)&c->d_out, ((size_t)nout_tl(W) + 64) * sizeof(double)) != hipSuccess ||
      hipMalloc((void **)&c->d_full, ((size_t)nout + 64) * sizeof(double)) != hipSuccess ||
      hipMalloc((void **)&c->d_scal, 64 * sizeof(double)) != hipSuccess) { vba_destroy(c); return VBA_ERR_HIP; }
  if (ensure_partial(c, (size_t)kMaxBlocksHess * (nout_tl(W) > nout ? nout_tl(W) : nout)) != VBA_OK || ensure_pin(c, 65536 + (size_t)nout + 1024) != VBA_OK) { vba_destroy(c); return VBA_ERR_HIP; }
  if (hipMalloc((void **)&c->d_lm, sizeof(LmDev)) != hipSuccess || hipMalloc((void **)&c->d_raw, ((size_t)nout_tl(W) + 64) * 8) != hipSuccess ||
      hipHostMalloc((void **)&c->h_lm, sizeof(LmDev), hipHostMallocDefault) != hipSuccess) { vba_destroy(c); return VBA_ERR_HIP; }
  if (opt->max_voxels && factor_reserve(c, (int)opt->max_voxels) != VBA_OK) { vba_destroy(c); return VBA_ERR_HIP; }
  map_init(c->map, c->opt);
  *out = c;
  return VBA_OK;
}

void vba_destroy(vba_ctx *c) {
  if (!c) return;
  hipSetDevice(c->device);
  if (c->stream) hipStreamSynchronize(c->stream);
  if (c->comm && c->own_comm) rccl_api().CommDestroy(c->comm);      // (a communicator exists only if the API resolved)
  map_free(c->map);
  c->gba.free_all();
  c->big.release();
  for (vba_ctx *w : c->hba_workers) vba_destroy(w);
  c->hba_workers.clear();
  if (c->d_hba_all) hipFree(c->d_hba_all);
  if (c->d_lipack) hipFree(c->d_lipack);
  if (c->d_liscr) hipFree(c->d_liscr);
  for (int i = 0; i < 2; i++) if (c->d_kdtree[i]) hipFree(c->d_kdtree[i]);
  if (c->d_refpts) hipFree(c->d_refpts);
  if (c->d_li) hipFree(c->d_li);
  if (c->d_k4part) hipFree(c->d_k4part);
  if (c->d_imu) hipFree(c->d_imu);
  if (c->d_himu) hipFree(c->d_himu);
  if (c->d_gimu) hipFree(c->d_gimu);
  double *p[] = {c->fv.cl, c->fv.fix, c->fv.coe, c->fv.eigval, c->fv.eigvec, c->fv.pcr, c->d_poses, c->d_partial, c->d_out, c->d_full, c->d_scal};
  for (double *q : p) if (q) hipFree(q);
  if (c->fv.occ) hipFree(c->fv.occ);
  if (c->fv.tiles) hipFree(c->fv.tiles);
  if (c->d_stage) hipFree(c->d_stage);
  if (c->h_pin) hipHostFree(c->h_pin);
  if (c->d_lm) hipFree(c->d_lm);
  if (c->d_raw) hipFree(c->d_raw);
  if (c->h_lm) hipHostFree(c->h_lm);
  for (int i = 0; i < vba_ctx::kLmRing; i++) { if (c->h_lm_up[i]) hipHostFree(c->h_lm_up[i]); if (c->lm_up_ev[i]) hipEventDestroy(c->lm_up_ev[i]); }
  for (auto &kv : c->spans) for (auto &s : kv.second) { hipEventDestroy(s.a); hipEventDestroy(s.b); }
  if (c->own_stream && c->stream) hipStreamDestroy(c->stream);
  delete c;
}

const char *vba_last_error(vba_ctx *c) { return c ? c->err.c_str() : ""; }
int vba_synchronize(vba_ctx *c) { HIPCHK(c, hipStreamSynchronize(c->stream)); return VBA_OK; }

// ---------------------------------------------------------------- factor level
int vba_factor_clear(vba_ctx *c) { c->nvox = 0; return VBA_OK; }
int vba_factor_size(vba_ctx *c) { return c->nvox; }

int vba_factor_push_voxels(vba_ctx *c, int n, const double *clusters, const double *fix, const double *coe, const double *eig_val,
                           const double *eig_vec, const double *pcr_add) {
  if (n < 0) return VBA_ERR_BAD_ARG;
  if (n == 0) return VBA_OK;
  const int W = c->opt.win_size;
  int st = factor_reserve(c, c->nvox + n);
  if (st) return st;
  const size_t per = (size_t)10 * W + 33;
  st = ensure_stage(c, per * n * sizeof(double));
  if (st) return st;
  double *d = (double *)c->d_stage;
  double *d_cl = d, *d_fix = d_cl + (size_t)n * W * 10, *d_coe = d_fix + (size_t)n * 10, *d_ev = d_coe + n, *d_evec = d_ev + (size_t)n * 3,
         *d_pcr = d_evec + (size_t)n * 9;
  HIPCHK(c, hipMemcpyAsync(d_cl, clusters, (size_t)n * W * 10 * sizeof(double), hipMemcpyHostToDevice, c->stream));
  HIPCHK(c, hipMemcpyAsync(d_fix, fix, (size_t)n * 10 * sizeof(double), hipMemcpyHostToDevice, c->stream));
  HIPCHK(c, hipMemcpyAsync(d_coe, coe, (size_t)n * sizeof(double), hipMemcpyHostToDevice, c->stream));
  HIPCHK(c, hipMemcpyAsync(d_ev, eig_val, (size_t)n * 3 * sizeof(double), hipMemcpyHostToDevice, c->stream));
  HIPCHK(c, hipMemcpyAsync(d_evec, eig_vec, (size_t)n * 9 * sizeof(double), hipMemcpyHostToDevice, c->stream));
  HIPCHK(c, hipMemcpyAsync(d_pcr, pcr_add, (size_t)n * 10 * sizeof(double), hipMemcpyHostToDevice, c->stream));
  const long long tot = (long long)n * per;
  int nb = (int)((tot + 255) / 256);
  if (nb > 4096) nb = 4096;
  hipLaunchKernelGGL(k_aos_to_soa, dim3(nb), dim3(256), 0, c->stream, c->fv, c->nvox, n, d_cl, d_fix, d_coe, d_ev, d_evec, d_pcr);
  factor_update_mask(c, c->nvox, n);
  HIPCHK(c, hipGetLastError());
  HIPCHK(c, hipStreamSynchronize(c->stream));  // caller's arrays may go away
  c->nvox += n;
  return VBA_OK;
}

int vba_factor_acc_evaluate2(vba_ctx *c, const double *poses, int head, int end, double *Hess, double *JacT, double *residual) {
  if (head < 0 || end > c->nvox || head > end) return VBA_ERR_BAD_ARG;
  const int W = c->opt.win_size, n = 6 * W, nout = nout_of(W);
  int st = eval_hessian_dev(c, poses, head, end);
  if (st) return st;
  std::vector<double> buf(nout);
  st = fetch(c, c->d_full, nout, buf.data());
  if (st) return st;
  if (Hess) std::memcpy(Hess, buf.data(), (size_t)n * n * sizeof(double));
  if (JacT) std::memcpy(JacT, buf.data() + (size_t)n * n, (size_t)n * sizeof(double));
  if (residual) *residual = buf[(size_t)n * n + n];
  return VBA_OK;
}

int vba_factor_evaluate_only_residual(vba_ctx *c, const double *poses, int head, int end, double *residual) {
  if (head < 0 || end > c->nvox || head > end) return VBA_ERR_BAD_ARG;
  double *d_r = c->d_scal;
  int st = eval_residual_dev(c, poses, head, end, d_r);
  if (st) return st;
  double r = 0;
  st = fetch(c, d_r, 1, &r);
  if (st) return st;
  if (residual) *residual = r;
  return VBA_OK;
}

int vba_factor_read_back(vba_ctx *c, double *eig_val, double *eig_vec, double *pcr_add) {
  const int n = c->nvox;
  if (n == 0) return VBA_OK;
  int st = ensure_stage(c, (size_t)n * 22 * sizeof(double));
  if (st) return st;
  double *d = (double *)c->d_stage;
  double *d_ev = d, *d_evec = d + (size_t)n * 3, *d_pcr = d_evec + (size_t)n * 9;
  int nb = (int)(((long long)n * 22 + 255) / 256);
  if (nb > 4096) nb = 4096;
  hipLaunchKernelGGL(k_soa_to_aos_out, dim3(nb), dim3(256), 0, c->stream, c->fv, n, d_ev, d_evec, d_pcr);
  HIPCHK(c, hipGetLastError());
  if (eig_val) HIPCHK(c, hipMemcpyAsync(eig_val, d_ev, (size_t)n * 3 * sizeof(double), hipMemcpyDeviceToHost, c->stream));
  if (eig_vec) HIPCHK(c, hipMemcpyAsync(eig_vec, d_evec, (size_t)n * 9 * sizeof(double), hipMemcpyDeviceToHost, c->stream));
  if (pcr_add) HIPCHK(c, hipMemcpyAsync(pcr_add, d_pcr, (size_t)n * 10 * sizeof(double), hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  return VBA_OK;
}

int vba_factor_occupancy_masks(vba_ctx *c, unsigned int *masks) {
  if (!masks && c->nvox > 0) return VBA_ERR_BAD_ARG;
  if (c->nvox == 0) return VBA_OK;
  HIPCHK(c, hipMemcpyAsync(masks, c->fv.occ, (size_t)c->nvox * sizeof(unsigned int), hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  return VBA_OK;
}
int vba_factor_occupied_slots(vba_ctx *c, long long *slots) {
  if (!slots) return VBA_ERR_BAD_ARG;
  *slots = 0;
  if (c->nvox == 0) return VBA_OK;
  int st = ensure_stage(c, 64);
  if (st) return st;
  HIPCHK(c, hipMemsetAsync(c->d_stage, 0, 8, c->stream));
  hipLaunchKernelGGL(k_count_slots, dim3(512), dim3(256), 0, c->stream, c->fv, c->nvox, (unsigned long long *)c->d_stage);
  unsigned long long h = 0;
  HIPCHK(c, hipMemcpyAsync(&h, c->d_stage, 8, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  *slots = (long long)h;
  return VBA_OK;
}

// ---------------------------------------------------------------- Lidar_BA_Optimizer (VM:342-498), device-resident loop
int vba_lm_begin(vba_ctx *c, const double *poses, int thd_num) {
  const int W = c->opt.win_size;
  const int slot = c->lm_up_next;
  c->lm_up_next = (slot + 1) % vba_ctx::kLmRing;
  if (!c->h_lm_up[slot]) {
    HIPCHK(c, hipHostMalloc((void **)&c->h_lm_up[slot], sizeof(LmDev), hipHostMallocDefault));
    HIPCHK(c, hipEventCreateWithFlags(&c->lm_up_ev[slot], hipEventDisableTiming));
  } else {
    HIPCHK(c, hipEventSynchronize(c->lm_up_ev[slot]));      // the copy that last used this slot has long completed
  }
  LmDev *h = c->h_lm_up[slot];
  std::memset(h, 0, sizeof(LmDev));
  std::memcpy(h->x, poses, (size_t)W * 12 * sizeof(double));
  std::memcpy(h->xt, poses, (size_t)W * 12 * sizeof(double));   // vector<IMUST> x_stats_temp = x_stats  VM:435
  h->u = 0.01; h->v = 2;                                        // VM:427
  h->is_calc_hess = 1; h->stop = 0; h->iter = 0; h->n_trace = 0; h->all_accepted = 1; h->last_accepted = 0; h->max_trace = 64;
  h->run_hess = 1; h->run_res = 1;
  { const char *e = diag_env("VBA_DEBUG_SOLVE"); h->pad = e ? atoi(e) : 0; }   // ablation / stamp mask of -DVBA_DIAG builds (0 otherwise)
  HIPCHK(c, hipMemcpyAsync(c->d_lm, h, sizeof(LmDev), hipMemcpyHostToDevice, c->stream));
  HIPCHK(c, hipEventRecord(c->lm_up_ev[slot], c->stream));
  c->lm.active = true; c->lm.thd_num = thd_num; c->lm.have_hess = false; c->lm.pending_update = false;
  c->trace.clear();
  // "Too Less Voxel" (VM:399-403) is a statement about the whole window: a sharded rank decides it from the voxel count summed over
  // the ranks, so every rank takes the same branch and none is left waiting in the next collective
  c->nvox_global = c->nvox;
  if (c->collective()) {
    int st = ensure_pin(c, 65536);
    if (st) return st;
    c->h_pin[60000] = (double)c->nvox;
    HIPCHK(c, hipMemcpyAsync(c->d_scal + 8, c->h_pin + 60000, sizeof(double), hipMemcpyHostToDevice, c->stream));
    st = ctx_allreduce(c, c->d_scal + 8, 1);
    if (st) return st;
    HIPCHK(c, hipStreamSynchronize(c->stream));
    HIPCHK(c, hipMemcpyAsync(c->h_pin + 60000, c->d_scal + 8, sizeof(double), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    c->nvox_global = (int)(c->h_pin[60000] + 0.5);
  }
  return VBA_OK;
}

// Re-creates the per-voxel eigen state (eig_values / eig_vectors / pcr_adds) at the poses loaded by vba_lm_begin: one
// residual pass on the device, nothing is fetched.  In the reference this state comes from recut/tras_opt right before
// damping_iter (VM:1628); a caller that restarts the optimiser on an unchanged factor store uses this instead.
int vba_lm_refresh_eigen(vba_ctx *c) {
  if (!c->lm.active) return VBA_ERR_BAD_ARG;
  const double *x_dev = reinterpret_cast<const double *>(reinterpret_cast<char *>(c->d_lm) + offsetof(LmDev, x));
  if (c->nvox <= 0) return VBA_OK;
  // only the pass' side effect is wanted (eig_values / eig_vectors / pcr_adds at the begin poses): its partials are not summed
  if ((size_t)residual_nb(c, c->nvox) > c->partial_doubles) { c->set_error("partial buffer too small"); return VBA_ERR_CAPACITY; }
  TimedSpan s1{};
  span_begin(c, "residual", s1);
  launch_residual(c, x_dev, nullptr, 0, c->nvox);
  span_end(c, "residual", s1);
  HIPCHK(c, hipGetLastError());
  return VBA_OK;
}

int vba_timing_launch_hessian(vba_ctx *c) {
  if (!c->lm.active || c->nvox <= 0) return VBA_ERR_BAD_ARG;
  const double *x_dev = reinterpret_cast<const double *>(reinterpret_cast<char *>(c->d_lm) + offsetof(LmDev, x));
  int nb = 0;
  const int st = launch_hessian(c, x_dev, nullptr, 0, c->nvox, &nb);
  if (st) return st;
  HIPCHK(c, hipGetLastError());
  return VBA_OK;
}

// One trip through the loop body VM:441-494, enqueued without host synchronisation unless the caller asks for the flags.
int vba_lm_iterate(vba_ctx *c, int *accepted, int *stop) {
  if (!c->lm.active) return VBA_ERR_BAD_ARG;
  const int W = c->opt.win_size, nout = nout_of(W), V = c->nvox;
  if (c->nvox_global < c->lm.thd_num) return VBA_ERR_TOO_FEW_VOXELS;   // VM:399-403 (and g_size checks of VM:367); the same on every rank
  char *base = reinterpret_cast<char *>(c->d_lm);
  const double *x_dev = reinterpret_cast<const double *>(base + offsetof(LmDev, x));
  const double *xt_dev = reinterpret_cast<const double *>(base + offsetof(LmDev, xt));
  const int *run_hess = reinterpret_cast<const int *>(base + offsetof(LmDev, run_hess));
  const int *run_res = reinterpret_cast<const int *>(base + offsetof(LmDev, run_res));
  // Multi-rank: ONE collective per iteration.  After the residual pass at the trial poses the Hessian pass is run there
  // too (speculating that the step is accepted) and [H | g | r] is all-reduced once: its r (the sum of the eigenvalues the
  // residual pass just stored) is the trial residual the accept test needs, and on acceptance H is already the next
  // iteration's Hessian; on a reject the solve keeps using its saved copy (`raw`), exactly as VM:443 skips divide_thread.
  const int copy_raw = c->collective() ? 1 : 0;
  int st = VBA_OK;
  static const bool no_fuse = diag_env("VBA_NO_FUSED_UPDATE") != nullptr;   // diagnostic: accept/reject always as its own kernel
  if (!(copy_raw && c->lm.have_hess)) {
    if (c->lm.pending_update) {               // the previous iteration's accept/reject rides in this pass (runs on xt after an accepted step)
      st = hessian_pass(c, x_dev, run_hess, 0, V, c->d_lm, c->d_k4part, c->lm.k4_nb);
      c->lm.pending_update = false;
    } else {
      st = hessian_pass(c, x_dev, run_hess, 0, V);   // divide_thread  VM:445 (skipped on device after a reject)
    }
  }
  if (st) return st;
  TimedSpan sp{};
  span_begin(c, "solve", sp);
  switch (W) {
#define VBA_SM_CASE(WW) case WW: hipLaunchKernelGGL(k_lm_solve_m<WW>, dim3(c->lm_spec), dim3(256), 0, c->stream, c->d_lm, c->d_out, c->d_raw, copy_raw); break;
    VBA_SM_CASE(2) VBA_SM_CASE(3) VBA_SM_CASE(4) VBA_SM_CASE(5) VBA_SM_CASE(6) VBA_SM_CASE(7) VBA_SM_CASE(8) VBA_SM_CASE(9) VBA_SM_CASE(10)
    VBA_SM_CASE(11) VBA_SM_CASE(12) VBA_SM_CASE(13) VBA_SM_CASE(14) VBA_SM_CASE(15) VBA_SM_CASE(16)
#undef VBA_SM_CASE
    default: return VBA_ERR_UNSUPPORTED_WINDOW;
  }
  span_end(c, "solve", sp);
  double *d_r = c->d_scal;
  if (copy_raw) {
    if (V > 0) launch_residual(c, xt_dev, run_res, 0, V);   // only_residual VM:467: refreshes the eigen state at the trial poses
    st = hessian_pass(c, xt_dev, run_res, 0, V);                      // speculative divide_thread there + the one all-reduce
    if (st) return st;
    c->lm.have_hess = true;
    hipLaunchKernelGGL(k_lm_update, dim3(1), dim3(64), 0, c->stream, c->d_lm, c->d_out + (nout_tl(W) - 1), 0, W);
  } else {
    const int nb = residual_nb(c, V);
    if (nb > c->k4part_cap) {
      if (c->d_k4part) { HIPCHK(c, hipStreamSynchronize(c->stream)); hipFree(c->d_k4part); c->d_k4part = nullptr; }
      const int cap = nb > 65536 ? 2 * nb : 65536;
      HIPCHK(c, hipMalloc((void **)&c->d_k4part, (size_t)cap * sizeof(double)));
      c->k4part_cap = cap;
    }
    const bool fuse = !no_fuse && !(accepted || stop);
    TimedSpan s1{};
    span_begin(c, "residual", s1);
    launch_residual(c, xt_dev, run_res, 0, V, c->d_k4part);
    span_end(c, "residual", s1);
    if (fuse) { c->lm.pending_update = true; c->lm.k4_nb = nb; }
    else hipLaunchKernelGGL(k_lm_update, dim3(1), dim3(64), 0, c->stream, c->d_lm, c->d_k4part, nb, W);   // sums the partials itself
  }
  HIPCHK(c, hipGetLastError());
  if (accepted || stop) {
    HIPCHK(c, hipMemcpyAsync(c->h_lm, c->d_lm, sizeof(LmDev), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    if (accepted) *accepted = c->h_lm->last_accepted;
    if (stop) *stop = c->h_lm->stop;
  }
  return VBA_OK;
}

int vba_lm_end(vba_ctx *c, double *poses, double *hess, double *resis2) {
  if (!c->lm.active) return VBA_ERR_BAD_ARG;
  const int W = c->opt.win_size, n = 6 * W;
  if (c->lm.pending_update) {                 // the last iteration's accept/reject has no Hessian pass to ride in
    hipLaunchKernelGGL(k_lm_update, dim3(1), dim3(64), 0, c->stream, c->d_lm, c->d_k4part, c->lm.k4_nb, W);
    c->lm.pending_update = false;
  }
  if (!poses && !hess && !resis2) { c->lm.active = false; return VBA_OK; }    // nothing requested: no synchronisation
  // One host round trip: the LM state (and *hess) reach the pinned mirrors through kernels that store via the host mapping — a D2H
  // copy queued behind in-flight kernels completes much later (see li_ba_device), and draining the stream first is a second trip.
  hipLaunchKernelGGL(k_words_to_host, dim3(4), dim3(256), 0, c->stream, (const int *)c->d_lm, (int *)c->h_lm, (int)(sizeof(LmDev) / 4));
  if (hess) {
    int st = ensure_pin(c, 65536 + (size_t)n * n + 1024);
    if (st) return st;
    const double *src = c->collective() ? c->d_raw : c->d_out;    // *hess = Hess before gauge fixing (VM:446)
    st = tiles_to_full(c, src);
    if (st) return st;
    hipLaunchKernelGGL(k_words_to_host, dim3(16), dim3(256), 0, c->stream, (const int *)c->d_full, (int *)(c->h_pin + 32768), n * n * 2);
  }
  HIPCHK(c, hipGetLastError());
  HIPCHK(c, hipStreamSynchronize(c->stream));
  const LmDev *h = c->h_lm;
  if (poses) std::memcpy(poses, h->x, (size_t)W * 12 * sizeof(double));
  if (hess) std::memcpy(hess, c->h_pin + 32768, (size_t)n * n * sizeof(double));
  if (resis2) { resis2[0] = h->resis_first; resis2[1] = h->r2; }
  c->trace.assign(h->trace, h->trace + 5 * h->n_trace);
  if (h->pad & 16) {
    fprintf(stderr, "[k_lm_solve_m cycles] prologue %lld | tile load %lld | factorisation %lld | backsub %lld | epilogue %lld | panels:", h->stamps[1] - h->stamps[0],
            h->stamps[2] - h->stamps[1], h->stamps[3] - h->stamps[2], h->stamps[4] - h->stamps[3], h->stamps[5] - h->stamps[4]);
    for (int kb = 0; kb < 8; kb++) fprintf(stderr, " %lld+%lld", h->stamps[9 + 2 * kb] - h->stamps[8 + 2 * kb], kb < 7 ? h->stamps[10 + 2 * kb] - h->stamps[9 + 2 * kb] : 0LL);
    fprintf(stderr, "\n");
  }
  c->lm.active = false;
  return VBA_OK;
}

int vba_lidar_ba_damping_iter(vba_ctx *c, double *poses, double *hess, double *resis2, int max_iter, int thd_num, int *is_converge) {
  int st = vba_lm_begin(c, poses, thd_num);
  if (st) return st;
  for (int i = 0; i < max_iter; i++) {           // the 1e-6 break (VM:492) is a device flag: later launches return at once
    st = vba_lm_iterate(c, nullptr, nullptr);
    if (st) { c->lm.active = false; return st; }
  }
  st = vba_lm_end(c, poses, hess, resis2);
  if (is_converge) *is_converge = c->h_lm->all_accepted;
  return st;
}

int vba_last_lm_trace(vba_ctx *c, double *rows, int max_rows) {
  int n = (int)(c->trace.size() / 5);
  if (n > max_rows) n = max_rows;
  if (rows) std::memcpy(rows, c->trace.data(), (size_t)n * 5 * sizeof(double));
  return n;
}

// ---------------------------------------------------------------- LI_BA_Optimizer / LI_BA_OptimizerGravity on the device
extern "C++" {
template <int W, int NT = (W > 10 ? 1024 : 512)>
static int launch_li_solve(vba_ctx *c, int copy_raw, int n, int gauge, int grav) {
  constexpr int NMAX = 15 * W + 3, NP = ((NMAX + 1 + 15) / 16) * 16;
  constexpr bool GL = W > 10;                    // L of the 15 W + 3 system exceeds the LDS: it lives in c->d_liscr
  constexpr size_t l_doubles = (size_t)LdltCfg<NP>::LTOT > (size_t)NMAX * (NMAX + 1) / 2 ? (size_t)LdltCfg<NP>::LTOT : (size_t)NMAX * (NMAX + 1) / 2;
  constexpr size_t lds = ((GL ? (size_t)LdltCfg<NP>::DOUBLES - LdltCfg<NP>::LTOT : (size_t)LdltCfg<NP>::DOUBLES) + 4 * NMAX + NP + 32) * 8 + (size_t)NMAX * 4 + 64;
  static_assert(GL || l_doubles == (size_t)LdltCfg<NP>::LTOT, "the staged triangle must fit the region of L");
  static bool attr_set[kMaxDevices] = {false};
  if (!attr_set[c->device % kMaxDevices]) { hipFuncSetAttribute((const void *)k_li_solve<W, NT, GL>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); attr_set[c->device % kMaxDevices] = true; }
  if (GL && c->liscr_doubles < l_doubles * LM_SPEC) {           // one region per damping candidate; W is fixed per context, so this runs once
    HIPCHK(c, hipStreamSynchronize(c->stream));
    if (c->d_liscr) hipFree(c->d_liscr);
    c->d_liscr = nullptr; c->liscr_doubles = 0;
    HIPCHK(c, hipMalloc((void **)&c->d_liscr, l_doubles * LM_SPEC * sizeof(double)));
    c->liscr_doubles = l_doubles * LM_SPEC;
  }
  hipLaunchKernelGGL((k_li_solve<W, NT, GL>), dim3(c->lm_spec), dim3(NT), lds, c->stream, c->d_lm, c->d_li, c->d_out, c->d_raw, copy_raw, c->d_himu, c->d_gimu, c->d_imu, n, gauge, grav,
                     c->opt.imu_coef, c->d_liscr);
  return VBA_OK;
}
}  // extern "C++"

static bool li_device_supported(int W) { return W >= 2 && W <= LI_MAX_W; }

static int li_ba_device(vba_ctx *c, double *states, double *imus, int gravity, int max_iter, double *hess, double *resis2) {
  static const bool want_times = diag_env("VBA_LI_TIMES") != nullptr;   // diagnostic: host-side phases of one call
  const auto t_0 = std::chrono::steady_clock::now();
  auto since = [&](std::chrono::steady_clock::time_point a) { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - a).count(); };
  const int W = c->opt.win_size, V = c->nvox, DIM = VBA_DIM, F = W - 1;
  const int n = W * DIM + (gravity ? 3 : 0), nb = gravity ? 33 : 30, n6 = 6 * W;
  if (!gravity) max_iter = 3;                                         // VM:643
  if (!c->d_li) {
    HIPCHK(c, hipMalloc((void **)&c->d_li, sizeof(LiDev)));
    HIPCHK(c, hipMalloc((void **)&c->d_imu, (size_t)LI_MAX_W * 304 * sizeof(double)));
    HIPCHK(c, hipMalloc((void **)&c->d_himu, (size_t)LI_MAX_N * LI_MAX_N * sizeof(double)));
    HIPCHK(c, hipMalloc((void **)&c->d_gimu, (size_t)LI_MAX_N * sizeof(double)));
  }
  // upload: LM state (poses view), the IMU extras, the factors with cov^-1 in place of cov
  std::vector<double> poses((size_t)W * 12);
  states_to_poses(states, W, poses.data());
  int st = vba_lm_begin(c, poses.data(), 0);
  if (st) return st;
  LiDev h{};
  h.W = W; h.n = n; h.nb = nb; h.gravity = gravity ? 1 : 0; h.gauge = gravity ? 6 : DIM; h.F = F; h.imu_coef = c->opt.imu_coef;   // VM:653-656 / 906-909
  for (int i = 0; i < W; i++) {
    const double *sx = states + 25 * i;
    h.tstamp[i] = sx[0];
    for (int k = 0; k < 12; k++) h.ex[12 * i + k] = h.ext[12 * i + k] = sx[13 + k];
  }
  std::vector<double> fimg((size_t)F * 304);
  std::memcpy(fimg.data(), imus, fimg.size() * sizeof(double));
  // cov^-1 (PI:166 / 244) is a property of the factor, and a sliding window hands the same factors in again scan after scan: a small
  // content-addressed cache (exact comparison of the 225 doubles) saves the host inversions (~4 us each)
  for (int f = 0; f < F; f++) {
    const double *cov = imus + 304 * (size_t)f + 79;
    double *dst = fimg.data() + 304 * (size_t)f + 79;
    bool hit = false;
    for (auto &e : c->covinv_cache)
      if (std::memcmp(e.data(), cov, 225 * sizeof(double)) == 0) { std::memcpy(dst, e.data() + 225, 225 * sizeof(double)); hit = true; break; }
    if (!hit) {
      vbh::inverse_pplu(cov, dst, 15);
      const bool grow = c->covinv_cache.size() < 32;
      if (grow) c->covinv_cache.emplace_back();
      auto &e = grow ? c->covinv_cache.back() : c->covinv_cache[c->covinv_next++ % 32];   // (round-robin replacement once full)
      std::memcpy(e.data(), cov, 225 * sizeof(double)); std::memcpy(e.data() + 225, dst, 225 * sizeof(double));
    }
  }
  {
    // (small pageable uploads are staged synchronously by the runtime, so the stack / vector sources may die after this)
    HIPCHK(c, hipMemcpyAsync(c->d_li, &h, sizeof(LiDev), hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipMemcpyAsync(c->d_imu, fimg.data(), fimg.size() * sizeof(double), hipMemcpyHostToDevice, c->stream));
  }
  HIPCHK(c, hipMemsetAsync(c->d_himu, 0, (size_t)li_hb_size(W, 1) * sizeof(double), c->stream));
  HIPCHK(c, hipMemsetAsync(c->d_gimu, 0, (size_t)n * sizeof(double), c->stream));
  char *base = reinterpret_cast<char *>(c->d_lm);
  const double *x_dev = reinterpret_cast<const double *>(base + offsetof(LmDev, x));
  const double *xt_dev = reinterpret_cast<const double *>(base + offsetof(LmDev, xt));
  const int *run_hess = reinterpret_cast<const int *>(base + offsetof(LmDev, run_hess));
  const int *run_res = reinterpret_cast<const int *>(base + offsetof(LmDev, run_res));
  const int copy_raw = c->collective() ? 1 : 0;
  const size_t lds_imu = ((size_t)2 * F * 15 * nb + 2 * F * 15 + F + 16) * sizeof(double);
  {
    static bool attr_set[kMaxDevices] = {false};      // W = 10 with gravity: 88 KB
    constexpr int FM = LI_MAX_W - 1;
    if (!attr_set[c->device % kMaxDevices]) { hipFuncSetAttribute((const void *)k_li_imu, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(((size_t)2 * FM * 15 * 33 + 2 * FM * 15 + FM + 16) * sizeof(double))); attr_set[c->device % kMaxDevices] = true; }
  }
  const double t_up = since(t_0);
  for (int it = 0; it < max_iter; it++) {
    // The IMU factors' workgroup rides in the lidar Hessian launch as block 0 on a CU of its own (k_hessian2).  As a kernel of its own
    // in front of the lidar pass it cost its 28 us + a kernel boundary; on a side stream (fork / join events around it) the two
    // cross-stream dependencies cost more than they hid (209 vs 196 us per iteration).
    const bool lidar_now = !(copy_raw && c->lm.have_hess) && V > 0 && lds_imu <= 150 * 1024;
    if (lidar_now) {          // the IMU workgroup rides in the lidar Hessian launch
      const LiJob job{c->d_lm, c->d_li, c->d_imu, c->d_himu, c->d_gimu};
      st = hessian_pass(c, x_dev, run_hess, 0, V, nullptr, nullptr, 0, job, lds_imu);   // lidar part of divide_thread (+ all-reduce)
    } else {
      hipLaunchKernelGGL(k_li_imu, dim3(1), dim3(LI_IMU_NT), lds_imu, c->stream, c->d_lm, c->d_li, c->d_imu, c->d_himu, c->d_gimu);
      if (!(copy_raw && c->lm.have_hess)) st = hessian_pass(c, x_dev, run_hess, 0, V);
    }
    if (st) { c->lm.active = false; return st; }
    TimedSpan s1{};
    span_begin(c, "solve", s1);
    switch (W) {
      case 2: st = launch_li_solve<2>(c, copy_raw, n, h.gauge, h.gravity); break;
      case 3: st = launch_li_solve<3>(c, copy_raw, n, h.gauge, h.gravity); break;
      case 4: st = launch_li_solve<4>(c, copy_raw, n, h.gauge, h.gravity); break;
      case 5: st = launch_li_solve<5>(c, copy_raw, n, h.gauge, h.gravity); break;
      case 6: st = launch_li_solve<6>(c, copy_raw, n, h.gauge, h.gravity); break;
      case 7: st = launch_li_solve<7>(c, copy_raw, n, h.gauge, h.gravity); break;
      case 9: st = launch_li_solve<9>(c, copy_raw, n, h.gauge, h.gravity); break;
      case 8: st = launch_li_solve<8>(c, copy_raw, n, h.gauge, h.gravity); break;
      case 10: {
        static const int nt = diag_env("VBA_LI_NT") ? atoi(diag_env("VBA_LI_NT")) : 512;      // -DVBA_DIAG builds only
        if (nt == 256) st = launch_li_solve<10, 256>(c, copy_raw, n, h.gauge, h.gravity); else if (nt == 1024) st = launch_li_solve<10, 1024>(c, copy_raw, n, h.gauge, h.gravity); else st = launch_li_solve<10, 512>(c, copy_raw, n, h.gauge, h.gravity);
        break;
      }
      case 11: st = launch_li_solve<11>(c, copy_raw, n, h.gauge, h.gravity); break;
      case 12: st = launch_li_solve<12>(c, copy_raw, n, h.gauge, h.gravity); break;
      case 13: st = launch_li_solve<13>(c, copy_raw, n, h.gauge, h.gravity); break;
      case 14: st = launch_li_solve<14>(c, copy_raw, n, h.gauge, h.gravity); break;
      case 15: st = launch_li_solve<15>(c, copy_raw, n, h.gauge, h.gravity); break;
      case 16: st = launch_li_solve<16>(c, copy_raw, n, h.gauge, h.gravity); break;
      default: c->lm.active = false; return VBA_ERR_UNSUPPORTED_WINDOW;
    }
    span_end(c, "solve", s1);
    if (st) { c->lm.active = false; return st; }                     // (scratch allocation of the W > 10 solve failed: nothing was launched)
    if (copy_raw) {                                                   // one collective per iteration (see vba_lm_iterate)
      if (V > 0) launch_residual(c, xt_dev, run_res, 0, V);
      st = hessian_pass(c, xt_dev, run_res, 0, V);
      if (st) { c->lm.active = false; return st; }
      c->lm.have_hess = true;
      hipLaunchKernelGGL(k_li_update, dim3(1), dim3(LI_UPD_NT), 0, c->stream, c->d_lm, c->d_li, c->d_imu, c->d_out + (nout_tl(W) - 1), 0);
    } else if (V == 0) {
      st = residual_pass(c, xt_dev, run_res, 0, V, c->d_scal);
      if (st) { c->lm.active = false; return st; }
      hipLaunchKernelGGL(k_li_update, dim3(1), dim3(LI_UPD_NT), 0, c->stream, c->d_lm, c->d_li, c->d_imu, c->d_scal, 0);
    } else {
      const int nbk = residual_nb(c, V);
      TimedSpan s2{};
      span_begin(c, "residual", s2);
      launch_residual(c, xt_dev, run_res, 0, V);
      span_end(c, "residual", s2);
      hipLaunchKernelGGL(k_li_update, dim3(1), dim3(LI_UPD_NT), 0, c->stream, c->d_lm, c->d_li, c->d_imu, c->d_partial, nbk);
    }
    HIPCHK(c, hipGetLastError());
  }
  const double t_enq = since(t_0);
  // (no drain before the download: with everything packed into ONE copy, queueing it behind the kernels is as fast as draining
  //  first — 497 vs 503 us per call; with five separate copies draining first had been 2-3x faster)
  const double t_gpu = since(t_0);
  // download: accepted state, the factors' bias increments, trace, and (on request) *hess = Hess before gauge fixing —
  // everything lands in ONE pinned block (pageable destinations make every copy a blocking staged transfer)
  // — gathered on the device into one block first: five separate D2H copies cost ~20 us each (110 us per call, measured)
  static_assert(sizeof(LiDev) % 8 == 0 && sizeof(LmDev) % 8 == 0, "packed as doubles");
  const size_t o_li = 0, o_img = o_li + sizeof(LiDev) / 8, o_hb = o_img + fimg.size(), o_lid = o_hb + (size_t)li_hb_size(W, 1),
               o_lm = o_lid + (size_t)n6 * n6, o_end = o_lm + sizeof(LmDev) / 8;
  st = ensure_pin(c, o_end + 64);
  if (st) return st;
  if (o_end + 64 > c->lipack_doubles) {
    if (c->d_lipack) hipFree(c->d_lipack);
    c->d_lipack = nullptr; c->lipack_doubles = 0;
    HIPCHK(c, hipMalloc((void **)&c->d_lipack, (o_end + 64) * sizeof(double)));
    c->lipack_doubles = o_end + 64;
  }
  PackSegs segs{};
  segs.n = 3;
  segs.src[0] = (const double *)c->d_li; segs.off[0] = o_li; segs.len[0] = sizeof(LiDev) / 8;
  segs.src[1] = c->d_imu; segs.off[1] = o_img; segs.len[1] = fimg.size();
  segs.src[2] = (const double *)c->d_lm; segs.off[2] = o_lm; segs.len[2] = sizeof(LmDev) / 8;
  if (hess) {
    st = tiles_to_full(c, copy_raw ? c->d_raw : c->d_out);
    if (st) return st;
    segs.n = 5;
    segs.src[3] = c->d_full; segs.off[3] = o_lid; segs.len[3] = (size_t)n6 * n6;
    segs.src[4] = c->d_himu; segs.off[4] = o_hb; segs.len[4] = (size_t)li_hb_size(W, gravity);
  }
  // (gathered on the device, then ONE copy: letting the gather kernel store the ~120 KB straight through the host mapping was
  //  measured slower — 187 against 174 us per iteration; for the few KB of the counters and of the lidar LM state it is faster)
  hipLaunchKernelGGL(k_pack_segments, dim3(64, segs.n), dim3(256), 0, c->stream, segs, c->d_lipack);
  HIPCHK(c, hipGetLastError());
  HIPCHK(c, hipMemcpyAsync(c->h_pin, c->d_lipack, o_end * sizeof(double), hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  std::memcpy(c->h_lm, c->h_pin + o_lm, sizeof(LmDev));
  std::memcpy(&h, c->h_pin + o_li, sizeof(LiDev));
  std::memcpy(fimg.data(), c->h_pin + o_img, fimg.size() * sizeof(double));
  const double *himu_h = c->h_pin + o_hb;
  c->lm.active = false;
  const LmDev *hl = c->h_lm;
  for (int i = 0; i < W; i++) {
    double *sx = states + 25 * i;
    for (int k = 0; k < 12; k++) sx[1 + k] = hl->x[12 * i + k];
    for (int k = 0; k < 12; k++) sx[13 + k] = h.ex[12 * i + k];
  }
  // (a rejected LAST step has installed the next damping candidate, which nothing evaluated: the caller sees what the sequential loop
  //  leaves behind, the restored increments of VM:701-705 = the buffers)
  if (hl->use_spec)
    for (int f = 0; f < F; f++) std::memcpy(fimg.data() + 304 * (size_t)f + 67, fimg.data() + 304 * (size_t)f + 73, 6 * sizeof(double));
  for (int f = 0; f < F; f++) std::memcpy(imus + 304 * (size_t)f + 67, fimg.data() + 304 * (size_t)f + 67, 12 * sizeof(double));   // dbg, dba, dbg_buf, dba_buf
  if (hess) {
    const double *lid = c->h_pin + o_lid;
    for (int r = 0; r < n; r++)
      for (int k = 0; k < n; k++) hess[(size_t)r * n + k] = c->opt.imu_coef * li_hb_get(himu_h, W, n, r, k);                     // VM:565
    for (int i = 0; i < W; i++)
      for (int j = 0; j < W; j++)
        for (int r = 0; r < 6; r++)
          for (int k = 0; k < 6; k++) hess[(size_t)(i * DIM + r) * n + j * DIM + k] += lid[(size_t)(i * 6 + r) * n6 + j * 6 + k];   // hess_plus VM:509-517
  }
  if (gravity && resis2) { resis2[0] = hl->resis_first; resis2[1] = hl->r2; }
  c->trace.assign(hl->trace, hl->trace + 5 * hl->n_trace);
  if (want_times && (hl->pad & 16)) {
    fprintf(stderr, "[k_li_solve prologue] stage imu %lld | stage lidar %lld | diag+g %lld | rank %lld\n", hl->stamps[50] - hl->stamps[0], hl->stamps[51] - hl->stamps[50], hl->stamps[52] - hl->stamps[51], hl->stamps[1] - hl->stamps[52]);
    fprintf(stderr, "[k_li_solve cycles] prologue %lld | tile load %lld | factorisation %lld | backsub %lld | epilogue %lld | panels:", hl->stamps[1] - hl->stamps[0],
            hl->stamps[2] - hl->stamps[1], hl->stamps[3] - hl->stamps[2], hl->stamps[4] - hl->stamps[3], hl->stamps[5] - hl->stamps[4]);
    for (int kb = 0; kb < 20; kb++) fprintf(stderr, " %lld+%lld", hl->stamps[9 + 2 * kb] - hl->stamps[8 + 2 * kb], kb < 19 ? hl->stamps[10 + 2 * kb] - hl->stamps[9 + 2 * kb] : 0LL);
    fprintf(stderr, "\n");
  }
  if (want_times && (hl->pad & 64)) fprintf(stderr, "[k_li_imu cycles] factor algebra (one lane per factor) %lld | cov^-1 joc %lld | contractions %lld\n", hl->stamps[41] - hl->stamps[40], hl->stamps[42] - hl->stamps[41], hl->stamps[43] - hl->stamps[42]);
  if (want_times && (hl->pad & 32)) { double v[6]; std::memcpy(v, &hl->stamps[58], sizeof(v)); fprintf(stderr, "[li r1 parts] rank %d: rimu %.10g lidar %.10g | %.10g %.10g | %.10g %.10g\n", c->rank, v[0], v[1], v[2], v[3], v[4], v[5]); }
  if (want_times) fprintf(stderr, "[li_ba_device] upload %.1f us | enqueue %.1f | gpu drained at %.1f | total %.1f\n", t_up, t_enq - t_up, t_gpu, since(t_0));
  return VBA_OK;
}

// ---------------------------------------------------------------- LI_BA_Optimizer / LI_BA_OptimizerGravity (VM:504-976)
int vba_li_ba_damping_iter(vba_ctx *c, double *states, double *imus, int gravity, int max_iter, double *hess, double *resis2) {
  // the whole optimiser runs on the device (k_li_imu / k_li_solve / k_li_update) for every window the context accepts (2..16)
  if (!li_device_supported(c->opt.win_size)) return VBA_ERR_UNSUPPORTED_WINDOW;
  return li_ba_device(c, states, imus, gravity, max_iter, hess, resis2);
}

// ---------------------------------------------------------------- initialisation odometry on a point-cloud map (vba_kernels_kd.hpp)
static int kd_reserve(vba_ctx *c, size_t pts) {
  if (pts <= c->kd_cap) return VBA_OK;
  size_t cap = c->kd_cap ? c->kd_cap : 65536;
  while (cap < pts) cap *= 2;
  for (int i = 0; i < 2; i++) {
    double *nw = nullptr;
    HIPCHK(c, hipMalloc((void **)&nw, cap * 3 * sizeof(double)));
    if (c->d_kdtree[i]) {
      if (i == c->kd_cur && c->kd_n > 0) HIPCHK(c, hipMemcpyAsync(nw, c->d_kdtree[i], (size_t)c->kd_n * 3 * sizeof(double), hipMemcpyDeviceToDevice, c->stream));
      HIPCHK(c, hipStreamSynchronize(c->stream));
      hipFree(c->d_kdtree[i]);
    }
    c->d_kdtree[i] = nw;
  }
  c->kd_cap = cap;
  return VBA_OK;
}
int vba_odom_kdtree_reset(vba_ctx *c) { c->kd_n = 0; return VBA_OK; }
int vba_odom_kdtree_size(vba_ctx *c) { return c->kd_n; }
int vba_odom_kdtree_points(vba_ctx *c, double *out) {
  if (!out && c->kd_n > 0) return VBA_ERR_BAD_ARG;
  if (c->kd_n == 0) return VBA_OK;
  HIPCHK(c, hipStreamSynchronize(c->stream));
  HIPCHK(c, hipMemcpyAsync(out, c->d_kdtree[c->kd_cur], (size_t)c->kd_n * 3 * sizeof(double), hipMemcpyDefault, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  return VBA_OK;
}

int vba_odom_lio_state_estimation_kdtree(vba_ctx *c, int n, const double *pnt_body, double *state, double *cov, int *iterations) {
  if (n < 0 || (n > 0 && !pnt_body) || !state || !cov) return VBA_ERR_BAD_ARG;
  if (iterations) *iterations = 0;
  const int DIM = VBA_DIM, nb = (n + 255) / 256;
  const int kd_slices = nb >= 512 ? 1 : (nb >= 128 ? 4 : 8);               // enough workgroups to cover the chip
  int st = ensure_stage(c, ((size_t)n * 7 + (size_t)nb * 28 + (size_t)kd_slices * n * 5 + 64) * sizeof(double));
  if (st) return st;
  double *d_pts = (double *)c->d_stage, *d_pl = d_pts + (size_t)n * 3, *d_part = d_pl + (size_t)n * 4;
  unsigned long long *d_cand = (unsigned long long *)(d_part + (size_t)nb * 28);
  if (n > 0) HIPCHK(c, hipMemcpyAsync(d_pts, pnt_body, (size_t)n * 3 * sizeof(double), hipMemcpyDefault, c->stream));
  vbh::State x_curr, x_prop;
  std::memcpy(&x_curr, state, sizeof(x_curr));
  auto pose_of = [](const vbh::State &x) { KdPose X; std::memcpy(X.R, x.R, sizeof(X.R)); std::memcpy(X.t, x.p, sizeof(X.t)); return X; };
  st = kd_reserve(c, (size_t)c->kd_n + (size_t)n + 16);
  if (st) return st;
  if (c->kd_n < 100) {                                                       // VS:1105-1118: the map is only seeded
    if (n > 0) hipLaunchKernelGGL(k_kd_append, dim3(nb), dim3(256), 0, c->stream, n, d_pts, pose_of(x_curr), c->d_kdtree[c->kd_cur] + (size_t)c->kd_n * 3);
    HIPCHK(c, hipGetLastError());
    HIPCHK(c, hipStreamSynchronize(c->stream));
    c->kd_n += n;
    return VBA_OK;
  }
  x_prop = x_curr;
  std::vector<double> P(cov, cov + 225), cov_inv(225), part((size_t)nb * 28);
  vbh::inverse_pplu(P.data(), cov_inv.data(), DIM);                          // VS:1134
  const int num_max_iter = 4;
  int rematch_num = 0, iters = 0;
  bool refind = true, converged_once = false;
  double G[225];
  std::memset(G, 0, sizeof(G));
  for (int iter = 0; iter < num_max_iter; iter++) {
    iters++;
    const KdPose X = pose_of(x_curr);
    double s28[28];
    std::memset(s28, 0, sizeof(s28));
    if (n > 0) {
      if (refind) {
        hipLaunchKernelGGL(k_kd_match, dim3(nb, kd_slices), dim3(256), 0, c->stream, n, d_pts, X, c->kd_n, c->d_kdtree[c->kd_cur], d_cand);
        hipLaunchKernelGGL(k_kd_fit, dim3(nb), dim3(256), 0, c->stream, n, kd_slices, d_cand, c->d_kdtree[c->kd_cur], d_pl);
      }
      hipLaunchKernelGGL(k_kd_accum, dim3(nb), dim3(256), 0, c->stream, n, d_pts, X, d_pl, d_part);
      HIPCHK(c, hipGetLastError());
      HIPCHK(c, hipStreamSynchronize(c->stream));
      HIPCHK(c, hipMemcpyAsync(part.data(), d_part, part.size() * sizeof(double), hipMemcpyDeviceToHost, c->stream));
      HIPCHK(c, hipStreamSynchronize(c->stream));
      for (int b = 0; b < nb; b++) for (int k = 0; k < 28; k++) s28[k] += part[(size_t)b * 28 + k];
    }
    double HTH[36], HTz[6];
    { int idx = 0; for (int r = 0; r < 6; r++) for (int k = r; k < 6; k++) { HTH[r * 6 + k] = s28[idx]; HTH[k * 6 + r] = s28[idx]; idx++; } }
    for (int r = 0; r < 6; r++) HTz[r] = s28[21 + r];
    // K_1 = (H_T_H + cov_inv / 1000)^-1 ; G(:,0:6) = K_1(:,0:6) HTH ; solution = K_1(:,0:6) HTz + vec - G(:,0:6) vec(0:6)   VS:1213-1217
    std::vector<double> A(225), K1(225);
    for (int k = 0; k < 225; k++) A[k] = cov_inv[k] / 1000;
    for (int r = 0; r < 6; r++) for (int k = 0; k < 6; k++) A[r * DIM + k] += HTH[r * 6 + k];
    vbh::inverse_pplu(A.data(), K1.data(), DIM);
    for (int r = 0; r < DIM; r++)
      for (int k = 0; k < 6; k++) { double sacc = 0; for (int j = 0; j < 6; j++) sacc += K1[r * DIM + j] * HTH[j * 6 + k]; G[r * DIM + k] = sacc; }
    double vec[15], RtR[9], lg[3];
    vbh::m3_Tmul(x_curr.R, x_prop.R, RtR);
    vbh::so3_log(RtR, lg);
    for (int k = 0; k < 3; k++) { vec[k] = lg[k]; vec[3 + k] = x_prop.p[k] - x_curr.p[k]; vec[6 + k] = x_prop.v[k] - x_curr.v[k]; vec[9 + k] = x_prop.bg[k] - x_curr.bg[k]; vec[12 + k] = x_prop.ba[k] - x_curr.ba[k]; }
    double sol[15];
    for (int r = 0; r < DIM; r++) {
      double a = 0, b = 0;
      for (int j = 0; j < 6; j++) { a += K1[r * DIM + j] * HTz[j]; b += G[r * DIM + j] * vec[j]; }
      sol[r] = a + vec[r] - b;
    }
    double E[9], Rn[9];
    vbh::so3_exp(sol, E);
    vbh::m3_mul(x_curr.R, E, Rn);
    std::memcpy(x_curr.R, Rn, sizeof(Rn));
    for (int k = 0; k < 3; k++) { x_curr.p[k] += sol[3 + k]; x_curr.v[k] += sol[6 + k]; x_curr.bg[k] += sol[9 + k]; x_curr.ba[k] += sol[12 + k]; }
    const double rot_add = vbh::norm3(sol), tra_add = vbh::norm3(sol + 3);
    refind = false;                                                          // VS:1223-1234
    if ((rot_add * 57.3 < 0.01) && (tra_add * 100 < 0.015)) { refind = true; converged_once = true; rematch_num++; }
    if (iter == num_max_iter - 2 && !converged_once) refind = true;
    if (rematch_num >= 2 || (iter == num_max_iter - 1)) {
      std::vector<double> IG(225), Pn(225);
      for (int r = 0; r < DIM; r++) for (int k = 0; k < DIM; k++) IG[r * DIM + k] = (r == k ? 1.0 : 0.0) - G[r * DIM + k];
      vbh::mat_mul(IG.data(), P.data(), Pn.data(), DIM, DIM, DIM);
      P = Pn;
      break;
    }
  }
  // map update VS:1238-1250: append the scan in the refined pose, re-sample on a 0.5 m grid
  if (n > 0) hipLaunchKernelGGL(k_kd_append, dim3(nb), dim3(256), 0, c->stream, n, d_pts, pose_of(x_curr), c->d_kdtree[c->kd_cur] + (size_t)c->kd_n * 3);
  HIPCHK(c, hipGetLastError());
  HIPCHK(c, hipStreamSynchronize(c->stream));
  const int tot = c->kd_n + n;
  {
    std::vector<int> cnt(tot), first(tot);
    int m = 0;
    st = vba_scan_down_sampling_voxel(c, tot, c->d_kdtree[c->kd_cur], 0.5, c->d_kdtree[c->kd_cur ^ 1], cnt.data(), first.data(), &m);
    if (st) return st;
    c->kd_cur ^= 1; c->kd_n = m;
  }
  std::memcpy(state, &x_curr, sizeof(x_curr));
  std::memcpy(cov, P.data(), 225 * sizeof(double));
  if (iterations) *iterations = iters;
  return VBA_OK;
}

// ---------------------------------------------------------------- hierarchical global BA (vba_kernels_gba.hpp)
static GbaParams gba_params(vba_ctx *c, double voxel_size, double min_eig, const double *eig_array) {
  GbaParams P;
  P.voxel_size = voxel_size; P.min_eigen_value = min_eig; P.max_layer = c->opt.max_layer;
  for (int k = 0; k < 4; k++) P.eig_array[k] = eig_array[k];
  return P;
}
static int gba_build_into_store(vba_ctx *c, int wdsize, const int *offsets, const double *pl, const double *poses, const GbaParams &P) {
  int nf = 0;
  TimedSpan sp{};
  span_begin(c, "gba_build", sp);
  int st = gba_build(c->gba, c->stream, wdsize, offsets, pl, poses, P, &nf, c->err);
  if (st) return st;
  c->nvox = 0;
  st = factor_reserve(c, nf > 0 ? nf : 1);
  if (st) return st;
  if (nf > 0) {
    const int nn = c->gba.h_cnt[GCNT_NODES] < c->gba.v.cap ? c->gba.h_cnt[GCNT_NODES] : c->gba.v.cap;
    hipLaunchKernelGGL(k_gba_extract, dim3((nn + 255) / 256, 10 * wdsize + 33), dim3(256), 0, c->stream, c->gba.v, c->fv);
    factor_update_mask(c, 0, nf);
    HIPCHK(c, hipGetLastError());
  }
  span_end(c, "gba_build", sp);
  c->nvox = nf;
  return VBA_OK;
}
static int gba_check(vba_ctx *c, int wdsize, const int *offsets, const double *pl, const double *poses) {
  if (wdsize != c->opt.win_size) return VBA_ERR_UNSUPPORTED_WINDOW;
  if (!offsets || !poses || offsets[0] != 0) return VBA_ERR_BAD_ARG;
  for (int i = 0; i < wdsize; i++) if (offsets[i + 1] < offsets[i]) return VBA_ERR_BAD_ARG;
  if (offsets[wdsize] > 0 && !pl) return VBA_ERR_BAD_ARG;
  return VBA_OK;
}
int vba_gba_build(vba_ctx *c, int wdsize, const int *offsets, const double *pnt_local, const double *poses, double gba_voxel_size,
                  double gba_min_eigen_value, const double *gba_eigen_value_array) {
  int st = gba_check(c, wdsize, offsets, pnt_local, poses);
  if (st) return st;
  if (!gba_eigen_value_array) return VBA_ERR_BAD_ARG;
  return gba_build_into_store(c, wdsize, offsets, pnt_local, poses, gba_params(c, gba_voxel_size, gba_min_eigen_value, gba_eigen_value_array));
}

// Lidar_BA_Optimizer::damping_iter (VM:422-497) for an arbitrary window: device Hessian / residual passes on the sparse
// store, gauge + (H + uD) LDL^T + retraction on the host.
// hdiag6_out: [W][W][6] = the six diagonal entries of every 6x6 block of *hess (all that HBA_add_edge reads of it, VS:2926-2951);
// the n x n Hessian itself stays in HBM.
static int big_damping_iter(vba_ctx *c, int W, double *poses, std::vector<double> &hdiag6_out, double *resis2, int max_iter, int thd_num, int *is_converge) {
  BigStore &S = c->big;
  const int n = 6 * W;
  if (S.b.V < thd_num) return VBA_ERR_TOO_FEW_VOXELS;                 // VM:399-403
  std::vector<double> x(poses, poses + (size_t)W * 12), xt(x), hd(n), JacT(n), dxi(n);
  double u = 0.01, v = 2, residual1 = 0, residual2 = 0;
  bool is_calc_hess = true, conv = true;
  c->trace.clear();
  for (int it = 0; it < max_iter; it++) {
    if (is_calc_hess) {
      int st = big_hessian(S, c->stream, x.data(), hd.data(), JacT.data(), &residual1, c->err);   // *hess = Hess (VM:446) stays on the device
      if (st) return st;
      for (int r = 0; r < 6; r++) { hd[r] = 1.0; JacT[r] = 0.0; }     // gauge VM:452-455 (k_bigl_setup applies it to the matrix)
    }
    if (it == 0) resis2[0] = residual1;
    {
      // pivot order of Eigen's LDLT (largest |stored diagonal| first, first index wins ties), then the device factorisation
      std::vector<int> ord(n);
      for (int r = 0; r < n; r++) ord[r] = r;
      std::vector<double> dabs(n);
      for (int r = 0; r < n; r++) dabs[r] = std::fabs(hd[r] + u * hd[r]);
      std::stable_sort(ord.begin(), ord.end(), [&](int a, int b) { return dabs[a] > dabs[b]; });
      int st2 = big_solve(S, c->stream, ord.data(), u, dxi.data(), c->err);
      if (st2) return st2;
    }
    for (int j = 0; j < W; j++) {
      double E[9];
      vbh::so3_exp(&dxi[6 * j], E);
      vbh::m3_mul(&x[12 * j], E, &xt[12 * j]);
      for (int k = 0; k < 3; k++) xt[12 * j + 9 + k] = x[12 * j + 9 + k] + dxi[6 * j + 3 + k];
    }
    double q1 = 0;
    for (int r = 0; r < n; r++) q1 += dxi[r] * (u * hd[r] * dxi[r] - JacT[r]);
    q1 *= 0.5;
    int st = big_residual(S, c->stream, xt.data(), &residual2, c->err);
    if (st) return st;
    double q = residual1 - residual2;
    const double tr[5] = {residual1, residual2, u, v, q1};
    c->trace.insert(c->trace.end(), tr, tr + 5);
    if (q > 0) {
      x = xt;
      q = q / q1;
      v = 2;
      q = 1 - std::pow(2 * q - 1, 3);
      u *= (q < 1.0 / 3 ? 1.0 / 3 : q);
      is_calc_hess = true;
    } else {
      u = u * v; v = 2 * v;
      is_calc_hess = false; conv = false;
    }
    if (std::fabs((residual1 - residual2) / residual1) < 1e-6) break;
  }
  resis2[1] = residual2;
  std::memcpy(poses, x.data(), x.size() * sizeof(double));
  if (is_converge) *is_converge = conv ? 1 : 0;
  hdiag6_out.resize((size_t)6 * W * W);
  return big_block_diagonals(S, c->stream, hdiag6_out.data(), c->err);   // b.H still holds the last evaluated Hessian (a rejected step does not recompute it)
}

int vba_hba_add_edge(vba_ctx *c, int wdsize, const int *offsets, const double *pnt_local, double *poses, double gba_voxel_size,
                     double gba_min_eigen_value, const double *gba_eigen_value_array, int max_iter, int thread_num, double *edges_out, int *n_edges,
                     double *cloud_out, int *cloud_count, int *n_cloud, double *resis_log, int *n_log) {
  const bool big = (wdsize != c->opt.win_size);      // any other window size (the top-level BA over all submaps): sparse path
  if (big && wdsize < 2) return VBA_ERR_BAD_ARG;
  int st = VBA_OK;
  if (!big) st = gba_check(c, wdsize, offsets, pnt_local, poses);
  else {
    if (!offsets || !poses || offsets[0] != 0) return VBA_ERR_BAD_ARG;
    for (int i = 0; i < wdsize; i++) if (offsets[i + 1] < offsets[i]) return VBA_ERR_BAD_ARG;
    if (offsets[wdsize] > 0 && !pnt_local) return VBA_ERR_BAD_ARG;
  }
  if (st) return st;
  if (!gba_eigen_value_array || !edges_out || !n_edges || (cloud_out && (!cloud_count || !n_cloud))) return VBA_ERR_BAD_ARG;
  const int W = wdsize, n6 = 6 * W, n = offsets[W];
  *n_edges = 0;
  if (n_log) *n_log = 0;
  static const bool want_times = diag_env("VBA_HBA_TIMES") != nullptr;      // diagnostic: wall-clock split of the call on stderr
  double t_ph[5] = {0, 0, 0, 0, 0};
  auto now = [] { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
  double t_mark = want_times ? now() : 0.0;
  auto lap = [&](int k) { if (want_times) { hipStreamSynchronize(c->stream); const double t = now(); t_ph[k] += t - t_mark; t_mark = t; } };
  // the keyframe clouds stay in HBM for the whole call (every outer iteration re-cuts them with the current poses)
  if ((size_t)n * 3 > c->refpts_doubles) {
    if (c->d_refpts) hipFree(c->d_refpts);
    c->refpts_doubles = (size_t)n * 3 + 3072;
    HIPCHK(c, hipMalloc((void **)&c->d_refpts, 2 * c->refpts_doubles * sizeof(double)));
  }
  double *d_pl = c->d_refpts, *d_ref = c->d_refpts + c->refpts_doubles;
  if (n > 0) HIPCHK(c, hipMemcpyAsync(d_pl, pnt_local, (size_t)n * 3 * sizeof(double), hipMemcpyDefault, c->stream));
  GbaParams P = gba_params(c, gba_voxel_size, gba_min_eigen_value, gba_eigen_value_array);
  std::vector<double> hess(big ? 0 : (size_t)n6 * n6, 0.0), hd6;      // the any-window path keeps *hess in HBM and returns its block diagonals
  lap(0);
  const int up = 4;                                                       // VS:2866
  int converge_flag = 0;
  double converge_thre = 0.05;
  for (int iterCnt = 0; iterCnt < max_iter; iterCnt++) {
    if (converge_flag == 1 || iterCnt == max_iter - 1)                    // VS:2871-2881: last pass with the local-map parameters
      P = gba_params(c, c->opt.voxel_size, c->opt.min_eigen_value, c->opt.plane_eigen_value_thre);
    double resis[2] = {0, 0};
    int is_converge = 0;
    if (!big) {
      st = gba_build_into_store(c, W, offsets, d_pl, poses, P);
      if (st) return st;
      lap(1);
      st = vba_lidar_ba_damping_iter(c, poses, hess.data(), resis, up, thread_num, &is_converge);
    } else {
      st = big_build(c->big, c->stream, W, offsets, d_pl, poses, P, c->err);
      if (st) return st;
      lap(1);
      st = big_damping_iter(c, W, poses, hd6, resis, up, thread_num, &is_converge);
    }
    if (st) return st;
    lap(2);
    if (resis_log && n_log) { resis_log[2 * *n_log] = resis[0]; resis_log[2 * *n_log + 1] = resis[1]; (*n_log)++; }
    if ((std::fabs(resis[0] - resis[1]) / resis[0] < converge_thre && is_converge) || (iterCnt == max_iter - 2 && converge_flag == 0)) {
      converge_thre = 0.01;                                               // VS:2903-2915
      if (converge_flag == 0) converge_flag = 1;
      else if (converge_flag == 1) break;
    }
  }
  int ne = 0;
  for (int i = 0; i < W - 1; i++)
    for (int j = i + 1; j < W; j++) {                                     // VS:2926-2951
      bool isAdd = true;
      double v6[6];
      for (int k = 0; k < 6; k++) {
        const double hc = std::fabs(big ? hd6[((size_t)i * W + j) * 6 + k] : hess[(size_t)(6 * i + k) * n6 + 6 * j + k]);
        if (hc < 1e-6) { isAdd = false; break; }
        v6[k] = 1.0 / hc;
      }
      if (!isAdd) continue;
      double *o = edges_out + 20 * (size_t)ne++;
      const double *Ri = poses + 12 * i, *Rj = poses + 12 * j;
      o[0] = i; o[1] = j;
      vbh::m3_Tmul(Ri, Rj, o + 2);
      const double d[3] = {Rj[9] - Ri[9], Rj[10] - Ri[10], Rj[11] - Ri[11]};
      vbh::m3_Tvec(Ri, d, o + 11);
      for (int k = 0; k < 6; k++) o[14 + k] = v6[k];
    }
  *n_edges = ne;
  lap(3);
  if (cloud_out) {                                                        // VS:2954-2989
    *n_cloud = 0;
    if (n > 0) {
      std::vector<double> rel((size_t)W * 12);
      for (int i = 0; i < W; i++) {
        const double *R0 = poses, *Ri = poses + 12 * i;
        vbh::m3_Tmul(R0, Ri, rel.data() + 12 * i);
        const double d[3] = {Ri[9] - R0[9], Ri[10] - R0[10], Ri[11] - R0[11]};
        vbh::m3_Tvec(R0, d, rel.data() + 12 * i + 9);
      }
      double *d_rel = big ? c->big.g.poses : c->gba.v.poses;
      const int *d_off = big ? c->big.g.offsets : c->gba.v.offsets;
      HIPCHK(c, hipMemcpyAsync(d_rel, rel.data(), rel.size() * sizeof(double), hipMemcpyHostToDevice, c->stream));
      hipLaunchKernelGGL(k_gba_to_ref, dim3((n + 255) / 256), dim3(256), 0, c->stream, n, W, d_off, d_pl, d_rel, d_ref);
      HIPCHK(c, hipGetLastError());
      HIPCHK(c, hipStreamSynchronize(c->stream));      // rel is a host temporary
      std::vector<int> first(n);
      st = vba_scan_down_sampling_voxel(c, n, d_ref, c->opt.voxel_size / 8, cloud_out, cloud_count, first.data(), n_cloud);
      if (st) return st;
    }
  }
  lap(4);
  if (want_times)
    std::fprintf(stderr, "[hba_add_edge W=%d n=%d] upload %.0f  build %.0f  LM %.0f  edges %.0f  cloud %.0f us\n", W, n, t_ph[0], t_ph[1], t_ph[2], t_ph[3], t_ph[4]);
  return VBA_OK;
}

// thd_globalmapping (VS:3018-3141), the optimisation work of the hierarchical global BA over one map:
//   bottom layer  windows of `wdsize` keyframes, stride `mgsize` (VS:3033-3034, 3064-3066, 3136-3137): HBA_add_edge(xs = x0 of the
//                 window, max_iter 1, thread_num 2) -> edges1 + one submap (pose x0 of the window's first keyframe, cloud
//                 = the window's down-sampled points in that frame, VS:3084-3089);
//   top layer     HBA_add_edge over all submaps with their CURRENT poses (VS:3096-3110): edges2.
// Edge rows carry GLOBAL keyframe indices.  (Queue handling, map switching and the GTSAM pose graph stay with the caller.)
int vba_hba_global(vba_ctx *c, int n_kf, const int *offsets, const double *pnt_local, const double *poses_x0, const double *poses_now,
                   double gba_voxel_size, double gba_min_eigen_value, const double *gba_eigen_value_array, int total_max_iter, int wdsize, int mgsize,
                   double *edges1_out, int cap1, int *n_edges1, double *edges2_out, int cap2, int *n_edges2) {
  if (n_kf < 0 || wdsize < 2 || mgsize < 1 || !offsets || !poses_x0 || !poses_now || !gba_eigen_value_array || !n_edges1 || !n_edges2 ||
      (offsets[n_kf] > 0 && !pnt_local))
    return VBA_ERR_BAD_ARG;
  *n_edges1 = 0; *n_edges2 = 0;
  std::vector<int> sub_first, sub_n;                // global id of every submap's first keyframe, points of its cloud
  std::vector<double> edges((size_t)(wdsize * (wdsize - 1) / 2 + 1) * 20);
  // the keyframe clouds go to HBM once (the stride-5 windows overlap: every keyframe is used twice) and the submap clouds
  // never leave it: every window's down-sampled cloud is written behind the previous one and the top-level BA reads them there
  const size_t n_all = (size_t)offsets[n_kf];
  size_t n_sub_cap = 0, n_win_max = 0;
  for (int start = 0; start + wdsize <= n_kf; start += mgsize) {
    const size_t nw = (size_t)(offsets[start + wdsize] - offsets[start]);
    n_sub_cap += nw; if (nw > n_win_max) n_win_max = nw;
  }
  // More than one rank ("windows are independent problems => replicas across GPUs for the bottom layer"): window
  // wi is optimised by rank wi % n_ranks with the exchange step switched off; every rank packs its windows' clouds and its
  // [points, edges, status | edge rows] records into ITS chunk of two buffers, and one ALL-GATHER of each hands every rank all of
  // them (a rank receives each foreign byte once).  A window that fails on one rank travels as its status word: every rank
  // enters both collectives and all of them return the same error afterwards — no rank is left waiting in a collective.
  // The top-level window then runs replicated (identical inputs on every rank).
  const bool replicas = c->collective() && c->n_ranks > 1;
  int n_win = 0;
  for (int start = 0; start + wdsize <= n_kf; start += mgsize) n_win++;
  // ONE rank: the windows are independent problems too, and one window is a chain of small kernels and host round trips that leaves
  // most of the chip idle — KL worker contexts (own stream, own octree and LM state; host threads drive them) optimise windows
  // side by side, with the bookkeeping of the replicas: worker t takes windows t, t + KL, ... and writes their clouds into its chunk.
  const int kl_opt = c->opt.hba_workers > 0 ? (c->opt.hba_workers < 8 ? c->opt.hba_workers : 8) : 4;
  const int KL = (!replicas && n_win >= 2 * kl_opt) ? kl_opt : 1;
  const bool local_rep = KL > 1, chunked = replicas || local_rep;
  const int NR = replicas ? c->n_ranks : KL;
  const size_t meta_per = 3 + (size_t)(wdsize * (wdsize - 1) / 2) * 20;
  const size_t win_per_rank = chunked ? (size_t)(n_win + NR - 1) / NR : 0, meta_chunk = meta_per * win_per_rank;
  std::vector<size_t> rank_cap(NR, 0), win_roff(n_win > 0 ? n_win : 1, 0);      // points capacity per rank chunk, window offset inside it
  if (chunked) {
    int w = 0;
    for (int start = 0; start + wdsize <= n_kf; start += mgsize, w++) {
      win_roff[w] = rank_cap[w % NR];
      rank_cap[w % NR] += (size_t)(offsets[start + wdsize] - offsets[start]);
    }
  }
  size_t chunk_pts = 0;
  for (int r = 0; r < NR; r++) if (rank_cap[r] > chunk_pts) chunk_pts = rank_cap[r];
  if (!chunked) chunk_pts = 0;
  const size_t need = (n_all + n_sub_cap + (size_t)NR * chunk_pts) * 3 + (size_t)NR * meta_chunk + 64;
  if (need > c->hba_all_doubles) {
    if (c->d_hba_all) hipFree(c->d_hba_all);
    c->d_hba_all = nullptr; c->hba_all_doubles = 0;
    HIPCHK(c, hipMalloc((void **)&c->d_hba_all, need * sizeof(double)));
    c->hba_all_doubles = need;
  }
  double *d_all = c->d_hba_all, *d_sub = c->d_hba_all + n_all * 3;
  if (n_all > 0 && !local_rep) HIPCHK(c, hipMemcpyAsync(d_all, pnt_local, n_all * 3 * sizeof(double), hipMemcpyDefault, c->stream));   // (the worker path uploads in chunks, under the first windows)
  std::vector<int> ccnt(n_win_max > 0 ? n_win_max : 1);
  size_t sub_off = 0;
  double *d_rep = d_sub + n_sub_cap * 3, *d_meta = d_rep + (size_t)NR * chunk_pts * 3;      // replica mode only
  std::vector<double> meta((size_t)NR * meta_chunk, 0.0);
  struct Restore { vba_ctx *c; bool was; ~Restore() { c->collective_off = was; } } restore{c, c->collective_off};
  if (replicas) c->collective_off = true;                                       // the windows' own LM loops must not enter a collective
  int wi = -1;
  static const bool want_times = diag_env("VBA_HBA_TIMES") != nullptr;
  auto now = [] { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
  const double t_g0 = want_times ? (hipStreamSynchronize(c->stream), now()) : 0.0;
  double t_g1 = 0;
  if (local_rep) {
    while ((int)c->hba_workers.size() < KL - 1) {
      vba_options o = c->opt; o.stream = nullptr; o.device = c->device;
      vba_ctx *w = nullptr;
      const int stc = vba_create(&o, &w);
      if (stc) { c->set_error("vba_hba_global: could not create a worker context"); return stc; }
      c->hba_workers.push_back(w);
    }
    HIPCHK(c, hipStreamSynchronize(c->stream));
    for (int w = 0; w < n_win; w++) meta[(size_t)(w % KL) * meta_chunk + meta_per * (size_t)(w / KL) + 2] = -1.0;   // "not run"
    std::vector<std::string> werr(KL);
    // the keyframe clouds travel to HBM in chunks on a stream of their own while the first windows are already being optimised
    // (2.4 GB at full length: as long as the windows themselves); a window starts when its keyframes have arrived
    std::atomic<int> kf_ready{0}, give_up{0};
    auto work = [&](int tw) {
      vba_ctx *cx = tw == 0 ? c : c->hba_workers[tw - 1];
      hipSetDevice(c->device);
      std::vector<double> ed(edges.size());
      std::vector<int> cc(ccnt.size()), off(wdsize + 1);
      for (int w = tw; w < n_win; w += KL) {
        const int start = w * mgsize;
        while (kf_ready.load(std::memory_order_acquire) < start + wdsize && !give_up.load()) std::this_thread::sleep_for(std::chrono::microseconds(20));
        if (give_up.load()) return;
        for (int i = 0; i <= wdsize; i++) off[i] = offsets[start + i] - offsets[start];
        std::vector<double> xs(poses_x0 + (size_t)start * 12, poses_x0 + (size_t)(start + wdsize) * 12);
        int ne = 0, nc = 0;
        double *mrec = &meta[(size_t)tw * meta_chunk + meta_per * (size_t)(w / KL)];
        const int st = vba_hba_add_edge(cx, wdsize, off.data(), d_all + (size_t)offsets[start] * 3, xs.data(), gba_voxel_size, gba_min_eigen_value,
                                        gba_eigen_value_array, 1, 2, ed.data(), &ne, d_rep + ((size_t)tw * chunk_pts + win_roff[w]) * 3,
                                        cc.data(), &nc, nullptr, nullptr);
        mrec[2] = st;
        if (st != VBA_OK) { werr[tw] = cx->err; return; }
        mrec[0] = nc; mrec[1] = ne;
        std::memcpy(mrec + 3, ed.data(), (size_t)ne * 20 * sizeof(double));
      }
    };
    int up_status = VBA_OK;
    {
      std::vector<std::thread> th;
      for (int tw = 0; tw < KL; tw++) th.emplace_back(work, tw);
      // (one uploader: three threads staging chunks in turn moved the pageable copy no faster — 4-5 GB/s either way; at full
      //  length the call is bound by this copy once the windows overlap it)
      hipStream_t up = nullptr;
      if (hipStreamCreateWithFlags(&up, hipStreamNonBlocking) != hipSuccess) { up_status = VBA_ERR_HIP; give_up.store(1); }
      const int CH = 16;                                                           // keyframes per chunk
      for (int k0 = 0; k0 < n_kf && up_status == VBA_OK; k0 += CH) {
        const int k1 = k0 + CH < n_kf ? k0 + CH : n_kf;
        const size_t o0 = (size_t)offsets[k0] * 3, nb = (size_t)(offsets[k1] - offsets[k0]) * 3 * sizeof(double);
        if (nb > 0 && (hipMemcpyAsync(d_all + o0, pnt_local + o0, nb, hipMemcpyDefault, up) != hipSuccess || hipStreamSynchronize(up) != hipSuccess)) {
          up_status = VBA_ERR_HIP; give_up.store(1); break;
        }
        kf_ready.store(k1, std::memory_order_release);
      }
      if (up) hipStreamDestroy(up);
      for (auto &x : th) x.join();
    }
    hipSetDevice(c->device);
    if (up_status != VBA_OK) { c->set_error("vba_hba_global: uploading the keyframe clouds failed"); return up_status; }
    for (int w = 0; w < n_win; w++) {                                            // the first failing window in window order decides
      const int stw = (int)meta[(size_t)(w % KL) * meta_chunk + meta_per * (size_t)(w / KL) + 2];
      if (stw > 0) { if (!werr[w % KL].empty()) c->set_error(werr[w % KL]); return stw; }
    }
    for (int w = 0; w < n_win; w++) {
      const double *mrec = &meta[(size_t)(w % KL) * meta_chunk + meta_per * (size_t)(w / KL)];
      if ((int)mrec[2] != VBA_OK) { c->set_error("vba_hba_global: a bottom-layer window was not run"); return VBA_ERR_HIP; }
      const int nc = (int)mrec[0], ne = (int)mrec[1], start = w * mgsize;
      for (int e = 0; e < ne; e++) {
        if (*n_edges1 >= cap1) return VBA_ERR_CAPACITY;
        double *o = edges1_out + (size_t)(*n_edges1) * 20;
        std::memcpy(o, mrec + 3 + (size_t)e * 20, 20 * sizeof(double));
        o[0] += start; o[1] += start;
        (*n_edges1)++;
      }
      if (nc > 0) HIPCHK(c, hipMemcpyAsync(d_sub + sub_off * 3, d_rep + ((size_t)(w % KL) * chunk_pts + win_roff[w]) * 3, (size_t)nc * 3 * sizeof(double),
                                           hipMemcpyDeviceToDevice, c->stream));
      sub_first.push_back(start);
      sub_n.push_back(nc);
      sub_off += (size_t)nc;
    }
  }
  for (int start = 0; !local_rep && start + wdsize <= n_kf; start += mgsize) {
    std::vector<int> off(wdsize + 1);
    for (int i = 0; i <= wdsize; i++) off[i] = offsets[start + i] - offsets[start];
    std::vector<double> xs(poses_x0 + (size_t)start * 12, poses_x0 + (size_t)(start + wdsize) * 12);
    int ne = 0, nc = 0;
    wi++;
    if (replicas) {
      sub_first.push_back(start);
      if (wi % NR != c->rank) continue;
      double *mrec = &meta[(size_t)c->rank * meta_chunk + meta_per * (size_t)(wi / NR)];
      const int st = vba_hba_add_edge(c, wdsize, off.data(), d_all + (size_t)offsets[start] * 3, xs.data(), gba_voxel_size, gba_min_eigen_value,
                                      gba_eigen_value_array, 1, 2, edges.data(), &ne, d_rep + ((size_t)c->rank * chunk_pts + win_roff[wi]) * 3,
                                      ccnt.data(), &nc, nullptr, nullptr);
      mrec[2] = st;                            // travels with the gather: every rank learns it
      if (st == VBA_OK) {
        mrec[0] = nc; mrec[1] = ne;
        std::memcpy(mrec + 3, edges.data(), (size_t)ne * 20 * sizeof(double));
      }
      continue;
    }
    int st = vba_hba_add_edge(c, wdsize, off.data(), d_all + (size_t)offsets[start] * 3, xs.data(), gba_voxel_size, gba_min_eigen_value,
                              gba_eigen_value_array, 1, 2, edges.data(), &ne, d_sub + sub_off * 3, ccnt.data(), &nc, nullptr, nullptr);
    if (st) return st;
    for (int e = 0; e < ne; e++) {
      if (*n_edges1 >= cap1) return VBA_ERR_CAPACITY;
      double *o = edges1_out + (size_t)(*n_edges1) * 20;
      std::memcpy(o, &edges[(size_t)e * 20], 20 * sizeof(double));
      o[0] += start; o[1] += start;
      (*n_edges1)++;
    }
    sub_first.push_back(start);
    sub_n.push_back(nc);
    sub_off += (size_t)nc;
  }
  if (replicas) {
    c->collective_off = restore.was;
    if (meta_chunk > 0)
      HIPCHK(c, hipMemcpyAsync(d_meta + (size_t)c->rank * meta_chunk, meta.data() + (size_t)c->rank * meta_chunk, meta_chunk * sizeof(double),
                               hipMemcpyHostToDevice, c->stream));
    int rc = ctx_allgather(c, d_rep, chunk_pts * 3);
    if (rc) return rc;
    rc = ctx_allgather(c, d_meta, meta_chunk);
    if (rc) return rc;
    HIPCHK(c, hipStreamSynchronize(c->stream));
    HIPCHK(c, hipMemcpyAsync(meta.data(), d_meta, meta.size() * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    int worst = VBA_OK;
    for (int w = 0; w < n_win; w++) {
      const int stw = (int)meta[(size_t)(w % NR) * meta_chunk + meta_per * (size_t)(w / NR) + 2];
      if (stw != VBA_OK && worst == VBA_OK) worst = stw;
    }
    if (worst != VBA_OK) { c->set_error("a bottom-layer window failed on one of the ranks"); return worst; }   // the same on every rank
    for (int w = 0; w < n_win; w++) {
      const double *mrec = &meta[(size_t)(w % NR) * meta_chunk + meta_per * (size_t)(w / NR)];
      const int nc = (int)mrec[0], ne = (int)mrec[1], start = sub_first[w];
      for (int e = 0; e < ne; e++) {
        if (*n_edges1 >= cap1) return VBA_ERR_CAPACITY;
        double *o = edges1_out + (size_t)(*n_edges1) * 20;
        std::memcpy(o, mrec + 3 + (size_t)e * 20, 20 * sizeof(double));
        o[0] += start; o[1] += start;
        (*n_edges1)++;
      }
      if (nc > 0) HIPCHK(c, hipMemcpyAsync(d_sub + sub_off * 3, d_rep + ((size_t)(w % NR) * chunk_pts + win_roff[w]) * 3, (size_t)nc * 3 * sizeof(double),
                                           hipMemcpyDeviceToDevice, c->stream));
      sub_n.push_back(nc);
      sub_off += (size_t)nc;
    }
  }
  const int ns = (int)sub_first.size();
  if (want_times) t_g1 = now();
  struct Report { bool on; double t0, *t1; decltype(now) *clk; ~Report() { if (on) std::fprintf(stderr, "[hba_global] windows %.0f us, top %.0f us (after the upload)\n", *t1 - t0, (*clk)() - *t1); } } report{want_times, t_g0, &t_g1, &now};
  if (ns >= 2) {
    std::vector<int> off(ns + 1, 0);
    for (int i = 0; i < ns; i++) off[i + 1] = off[i] + sub_n[i];
    std::vector<double> xs((size_t)ns * 12), e2((size_t)(ns * (ns - 1) / 2 + 1) * 20);
    for (int i = 0; i < ns; i++) std::memcpy(&xs[(size_t)i * 12], poses_now + (size_t)sub_first[i] * 12, 12 * sizeof(double));
    int ne = 0;
    int st = vba_hba_add_edge(c, ns, off.data(), d_sub, xs.data(), gba_voxel_size, gba_min_eigen_value, gba_eigen_value_array, total_max_iter, 5,
                              e2.data(), &ne, nullptr, nullptr, nullptr, nullptr, nullptr);
    if (st) return st;
    for (int e = 0; e < ne; e++) {
      if (*n_edges2 >= cap2) return VBA_ERR_CAPACITY;
      double *o = edges2_out + (size_t)(*n_edges2) * 20;
      std::memcpy(o, &e2[(size_t)e * 20], 20 * sizeof(double));
      o[0] = sub_first[(int)e2[(size_t)e * 20]]; o[1] = sub_first[(int)e2[(size_t)e * 20 + 1]];
      (*n_edges2)++;
    }
  }
  return VBA_OK;
}

// ---------------------------------------------------------------- IMU factor (host)
int vba_imu_preintegrate(int n, const double *t, const double *gyr, const double *acc, const double *bg, const double *ba,
                         const double *nm6, const double *nw6, double scale_gravity, double *out) {
  if (n < 1 || !t || !gyr || !acc || !out) return VBA_ERR_BAD_ARG;
  vbh::ImuPre m;
  vbh::imu_init(m, bg, ba);
  vbh::imu_push(m, n, t, gyr, acc, nm6, nw6, scale_gravity);
  std::memcpy(out, &m, sizeof(m));
  return VBA_OK;
}
int vba_imu_give_evaluate(const double *imu_pre, const double *s1, const double *s2, int with_gravity, int jac_enable, double *jtj,
                          double *gg, double *resid) {
  if (!imu_pre || !s1 || !s2) return VBA_ERR_BAD_ARG;
  const double r = vbh::imu_evaluate(*reinterpret_cast<const vbh::ImuPre *>(imu_pre), *reinterpret_cast<const vbh::State *>(s1),
                                     *reinterpret_cast<const vbh::State *>(s2), with_gravity != 0, jac_enable != 0, jtj, gg);
  if (resid) *resid = r;
  return VBA_OK;
}

// ---------------------------------------------------------------- multi-GPU plumbing
int vba_set_allreduce(vba_ctx *c, vba_allreduce_fn fn, void *user) { c->allreduce = fn; c->allreduce_user = user; return VBA_OK; }

// RCCL inside the library (north_star: "RCCL all-reduce of the (6W)x(6W) Hessian over xGMI"): the communicator lives in the context
// and ncclAllReduce(ncclDouble, ncclSum) is issued on the context's stream, no host code in the LM loop.
int vba_rccl_get_unique_id(void *out128) {
  if (!out128) return VBA_ERR_BAD_ARG;
  static_assert(sizeof(ncclUniqueId) == 128, "ncclUniqueId is 128 bytes");
  const RcclApi &R = rccl_api();
  if (!R.ok) return VBA_ERR_UNSUPPORTED;
  ncclUniqueId id;
  if (R.GetUniqueId(&id) != ncclSuccess) return VBA_ERR_HIP;
  std::memcpy(out128, &id, sizeof(id));
  return VBA_OK;
}
int vba_rccl_init(vba_ctx *c, const void *unique_id128, int rank, int n_ranks) {
  if (!c || !unique_id128 || n_ranks < 1 || rank < 0 || rank >= n_ranks) return VBA_ERR_BAD_ARG;
  const RcclApi &R = rccl_api();
  if (!R.ok) { c->set_error(R.why); return VBA_ERR_UNSUPPORTED; }
  if (c->comm && c->own_comm) { R.CommDestroy(c->comm); c->comm = nullptr; }
  ncclUniqueId id;
  std::memcpy(&id, unique_id128, sizeof(id));
  HIPCHK(c, hipSetDevice(c->device));
  const ncclResult_t r = R.CommInitRank(&c->comm, n_ranks, id, rank);
  if (r != ncclSuccess) { c->comm = nullptr; c->set_error(std::string("ncclCommInitRank: ") + R.GetErrorString(r)); return VBA_ERR_HIP; }
  c->own_comm = true;
  return vba_set_shard(c, rank, n_ranks);
}
int vba_set_rccl_comm(vba_ctx *c, void *nccl_comm) {
  if (!c) return VBA_ERR_BAD_ARG;
  const RcclApi &R = rccl_api();
  if (!R.ok) { c->set_error(R.why); return VBA_ERR_UNSUPPORTED; }
  if (c->comm && c->own_comm) R.CommDestroy(c->comm);
  c->comm = (ncclComm_t)nccl_comm; c->own_comm = false;
  return VBA_OK;
}
int vba_shard_owner(int64_t kx, int64_t ky, int64_t kz, int n_ranks) {
  if (n_ranks <= 1) return 0;
  return (int)(vba::shard_bucket(kx, ky, kz) * (uint64_t)n_ranks >> 16);   // contiguous bucket ranges per rank
}
int vba_set_shard(vba_ctx *c, int rank, int n_ranks) {
  if (n_ranks < 1 || rank < 0 || rank >= n_ranks) return VBA_ERR_BAD_ARG;
  c->rank = rank; c->n_ranks = n_ranks;
  c->map.rank = rank; c->map.n_ranks = n_ranks;
  c->map.allreduce = [c](double *buf, size_t n) { return (c->allreduce || c->comm) ? ctx_allreduce(c, buf, n) : (int)VBA_ERR_BAD_ARG; };
  return VBA_OK;
}

// ---------------------------------------------------------------- timing
// Measurement aid: one launch that reads exactly n_bytes (rounded down to a multiple of 32 KiB) from a zero-filled scratch buffer
// in the access shape of the factor store (k_calib_read8).  Under `rocprofv3 --pmc FETCH_SIZE` its counter value calibrates the
// read-side correction factor tools/prof_summary.py applies to the residual pass.
int vba_timing_calibration_read(vba_ctx *c, size_t n_bytes) {
  const size_t nblk = n_bytes / (256 * 128);
  if (nblk == 0 || nblk > 0x7fffffffu) return VBA_ERR_BAD_ARG;
  double *buf = nullptr;
  HIPCHK(c, hipMalloc((void **)&buf, nblk * 256 * 128 + 64));
  HIPCHK(c, hipMemsetAsync(buf, 0, nblk * 256 * 128 + 64, c->stream));
  hipLaunchKernelGGL(k_calib_read8, dim3((unsigned)nblk), dim3(256), 0, c->stream, buf, buf + nblk * 256 * 16);
  HIPCHK(c, hipGetLastError());
  HIPCHK(c, hipStreamSynchronize(c->stream));
  hipFree(buf);
  return VBA_OK;
}
int vba_timing_enable(vba_ctx *c, int on) { c->timing = on != 0; return VBA_OK; }
int vba_timing_null_span(vba_ctx *c) {   // an event pair around nothing: the bracketing overhead itself (recorded as "null")
  TimedSpan s{};
  const bool was = c->timing;
  const std::string only = c->timing_only;
  c->timing = true; c->timing_only.clear();
  span_begin(c, "null", s);
  span_end(c, "null", s);
  c->timing = was; c->timing_only = only;
  return VBA_OK;
}
int vba_timing_select(vba_ctx *c, const char *name) { c->timing_only = name ? name : ""; return VBA_OK; }
int vba_timing_sample_every(vba_ctx *c, int n) { c->timing_every = n > 1 ? n : 1; c->timing_ctr = 0; return VBA_OK; }
int vba_timing_reset(vba_ctx *c) {
  hipStreamSynchronize(c->stream);
  for (auto &kv : c->spans) for (auto &s : kv.second) { hipEventDestroy(s.a); hipEventDestroy(s.b); }
  c->spans.clear();
  return VBA_OK;
}
int vba_timing_get(vba_ctx *c, const char *name, double *total_us, int *count) {
  hipStreamSynchronize(c->stream);
  double tot = 0; int n = 0;
  auto it = c->spans.find(name);
  if (it != c->spans.end())
    for (auto &s : it->second) { float ms = 0; if (hipEventElapsedTime(&ms, s.a, s.b) == hipSuccess) { tot += (double)ms * 1000.0; n++; } }
  if (total_us) *total_us = tot;
  if (count) *count = n;
  return VBA_OK;
}

// ---------------------------------------------------------------- map level (vba_kernels_map.hpp)
int vba_map_cut_voxel(vba_ctx *c, int win_count, int n, const double *pnt_body, const double *var, const double *pose, int multi) {
  TimedSpan s{};
  span_begin(c, "insert", s);
  const int st = map_cut_voxel(c->map, c->stream, win_count, n, pnt_body, var, pose, multi != 0, c->err);
  span_end(c, "insert", s);
  return st;
}
int vba_map_pvec_update_cut_voxel(vba_ctx *c, int win_count, int n, const double *pnt_body, const double *var_body, const double *pose,
                                  const double *cov, int multi) {
  if (!cov || !var_body) return VBA_ERR_BAD_ARG;
  double cov6[18];
  for (int r = 0; r < 3; r++) for (int k = 0; k < 3; k++) { cov6[3 * r + k] = cov[r * VBA_DIM + k]; cov6[9 + 3 * r + k] = cov[(3 + r) * VBA_DIM + 3 + k]; }
  TimedSpan s{};
  span_begin(c, "insert", s);
  const int st = map_cut_voxel(c->map, c->stream, win_count, n, pnt_body, var_body, pose, multi != 0, c->err, cov6);
  span_end(c, "insert", s);
  return st;
}
int vba_scan_var_init(vba_ctx *c, int n, const double *pnt_in, const double *ext_pose, double dept_err, double beam_err, double *pnt_out,
                      double *var_out) {
  if (n < 0 || (n > 0 && (!pnt_in || !pnt_out || !var_out)) || !ext_pose) return VBA_ERR_BAD_ARG;
  if (n == 0) return VBA_OK;
  int st = ensure_stage(c, ((size_t)n * 15 + 16) * sizeof(double));
  if (st) return st;
  double *d_in = (double *)c->d_stage, *d_out = d_in + (size_t)n * 3, *d_var = d_out + (size_t)n * 3, *d_ext = d_var + (size_t)n * 9;
  HIPCHK(c, hipMemcpyAsync(d_in, pnt_in, (size_t)n * 3 * sizeof(double), hipMemcpyDefault, c->stream));
  HIPCHK(c, hipMemcpyAsync(d_ext, ext_pose, 12 * sizeof(double), hipMemcpyHostToDevice, c->stream));
  hipLaunchKernelGGL(k_var_init, dim3((n + 255) / 256), dim3(256), 0, c->stream, n, d_in, d_out, d_var, d_ext, (float)dept_err, (float)beam_err);
  HIPCHK(c, hipGetLastError());
  HIPCHK(c, hipMemcpyAsync(pnt_out, d_out, (size_t)n * 3 * sizeof(double), hipMemcpyDefault, c->stream));
  HIPCHK(c, hipMemcpyAsync(var_out, d_var, (size_t)n * 9 * sizeof(double), hipMemcpyDefault, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  return VBA_OK;
}
// mode 0 down_sampling_voxel, 1 down_sampling_pvec (var in, vout out), 2 down_sampling_close (first_out = chosen indices)
static int ds_common(vba_ctx *c, int mode, int n, const double *pnt, const double *var, double voxel_size, double *pnt_out, double *vout, int *count_out,
                     int *first_out, int *n_out) {
  if (n < 0 || !n_out || (n > 0 && (!pnt || !first_out)) || (mode != 2 && n > 0 && (!pnt_out || !count_out)) || (mode == 1 && n > 0 && (!var || !vout)))
    return VBA_ERR_BAD_ARG;
  *n_out = 0;
  if (n == 0) return VBA_OK;
  if (voxel_size < 0.001 && mode != 1) {                                // TL:203 / TL:242: the cloud is left untouched
    if (pnt_out) HIPCHK(c, hipMemcpyAsync(pnt_out, pnt, (size_t)n * 3 * sizeof(double), hipMemcpyDefault, c->stream));
    std::vector<int> z(n, 0), id(n);
    for (int i = 0; i < n; i++) id[i] = i;
    if (count_out) HIPCHK(c, hipMemcpyAsync(count_out, z.data(), (size_t)n * sizeof(int), hipMemcpyDefault, c->stream));
    HIPCHK(c, hipMemcpyAsync(first_out, id.data(), (size_t)n * sizeof(int), hipMemcpyDefault, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    *n_out = n;
    return VBA_OK;
  }
  int cap = 1024;
  while (cap < 2 * n) cap <<= 1;
  const int nb = (n + 255) / 256;
  const size_t b_tab = (size_t)cap * sizeof(DsSlot), b_pnt = (size_t)n * 3 * sizeof(double), b_i = (((size_t)n * sizeof(int)) + 15) & ~(size_t)15,
               b_blk = (((size_t)nb + 1) * sizeof(int) + 15) & ~(size_t)15, b_var = mode == 1 ? (size_t)n * 9 * sizeof(double) : 0,
               b_dist = mode == 2 ? (size_t)n * sizeof(double) : 0;
  int st = ensure_stage(c, b_tab + 3 * b_pnt + b_var + b_dist + 3 * b_i + b_blk + 64);
  if (st) return st;
  char *base = (char *)c->d_stage;
  DsSlot *tab = (DsSlot *)base;
  double *d_in = (double *)(base + b_tab), *d_out = (double *)(base + b_tab + b_pnt), *d_vout = (double *)(base + b_tab + 2 * b_pnt),
         *d_var = (double *)(base + b_tab + 3 * b_pnt), *d_dist = (double *)(base + b_tab + 3 * b_pnt + b_var);
  int *d_slot = (int *)(base + b_tab + 3 * b_pnt + b_var + b_dist), *d_cnt = (int *)((char *)d_slot + b_i), *d_first = (int *)((char *)d_cnt + b_i),
      *d_blk = (int *)((char *)d_first + b_i), *d_n = d_blk + nb;
  HIPCHK(c, hipMemcpyAsync(d_in, pnt, b_pnt, hipMemcpyDefault, c->stream));
  if (mode == 1) HIPCHK(c, hipMemcpyAsync(d_var, var, b_var, hipMemcpyDefault, c->stream));
  TimedSpan sp{};
  span_begin(c, "downsample", sp);
  hipLaunchKernelGGL(k_ds_clear, dim3((cap + 255) / 256), dim3(256), 0, c->stream, tab, cap);
  hipLaunchKernelGGL(k_ds_insert, dim3(nb), dim3(256), 0, c->stream, n, d_in, mode == 1 ? d_var : nullptr, voxel_size, tab, cap - 1, d_slot);
  if (mode == 2) {
    hipLaunchKernelGGL(k_ds_close_min, dim3(nb), dim3(256), 0, c->stream, n, d_in, tab, d_slot, d_dist);
    hipLaunchKernelGGL(k_ds_close_arg, dim3(nb), dim3(256), 0, c->stream, n, tab, d_slot, d_dist);
  }
  hipLaunchKernelGGL(k_ds_count, dim3(nb), dim3(256), 0, c->stream, n, tab, d_slot, d_blk);
  hipLaunchKernelGGL(k_ds_scan, dim3(1), dim3(256), 0, c->stream, nb, d_blk, d_n);
  hipLaunchKernelGGL(k_ds_emit, dim3(nb), dim3(256), 0, c->stream, n, tab, d_slot, d_blk, d_out, d_cnt, d_first, d_vout, mode);
  span_end(c, "downsample", sp);
  HIPCHK(c, hipGetLastError());
  int m = 0;
  HIPCHK(c, hipStreamSynchronize(c->stream));
  HIPCHK(c, hipMemcpyAsync(&m, d_n, sizeof(int), hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  if (m > 0) {
    if (pnt_out) HIPCHK(c, hipMemcpyAsync(pnt_out, d_out, (size_t)m * 3 * sizeof(double), hipMemcpyDefault, c->stream));
    if (count_out) HIPCHK(c, hipMemcpyAsync(count_out, d_cnt, (size_t)m * sizeof(int), hipMemcpyDefault, c->stream));
    HIPCHK(c, hipMemcpyAsync(first_out, d_first, (size_t)m * sizeof(int), hipMemcpyDefault, c->stream));
    if (mode == 1) HIPCHK(c, hipMemcpyAsync(vout, d_vout, (size_t)m * 3 * sizeof(double), hipMemcpyDefault, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
  }
  *n_out = m;
  return VBA_OK;
}
int vba_scan_down_sampling_voxel(vba_ctx *c, int n, const double *pnt, double voxel_size, double *pnt_out, int *count_out, int *first_out,
                                 int *n_out) {
  return ds_common(c, 0, n, pnt, nullptr, voxel_size, pnt_out, nullptr, count_out, first_out, n_out);
}
int vba_scan_down_sampling_pvec(vba_ctx *c, int n, const double *pnt, const double *var, double voxel_size, double *pnt_out, double *vardiag_out,
                                int *count_out, int *n_out) {
  std::vector<int> first(n > 0 ? n : 1);
  return ds_common(c, 1, n, pnt, var, voxel_size, pnt_out, vardiag_out, count_out, first.data(), n_out);
}
int vba_scan_down_sampling_close(vba_ctx *c, int n, const double *pnt, double voxel_size, int *index_out, int *n_out) {
  return ds_common(c, 2, n, pnt, nullptr, voxel_size, nullptr, nullptr, nullptr, index_out, n_out);
}
int vba_scan_undistort(vba_ctx *c, int n, double *pnt, const double *curv, int m, const double *imu_poses, const double *end_pose,
                       const double *ext_pose) {
  if (n < 0 || m < 0 || (n > 0 && (!pnt || !curv)) || (m > 0 && !imu_poses) || !end_pose || !ext_pose) return VBA_ERR_BAD_ARG;
  if (n == 0 || m == 0) return VBA_OK;
  const size_t nprm = (size_t)22 * m + 24;
  int st = ensure_stage(c, ((size_t)n * 4 + nprm) * sizeof(double));
  if (st) return st;
  double *d_p = (double *)c->d_stage, *d_c = d_p + (size_t)n * 3, *d_prm = d_c + n;
  std::vector<double> prm(nprm);
  std::memcpy(prm.data(), imu_poses, (size_t)22 * m * sizeof(double));
  std::memcpy(prm.data() + (size_t)22 * m, end_pose, 12 * sizeof(double));
  std::memcpy(prm.data() + (size_t)22 * m + 12, ext_pose, 12 * sizeof(double));
  HIPCHK(c, hipMemcpyAsync(d_p, pnt, (size_t)n * 3 * sizeof(double), hipMemcpyDefault, c->stream));
  HIPCHK(c, hipMemcpyAsync(d_c, curv, (size_t)n * sizeof(double), hipMemcpyDefault, c->stream));
  HIPCHK(c, hipMemcpyAsync(d_prm, prm.data(), nprm * sizeof(double), hipMemcpyHostToDevice, c->stream));
  TimedSpan sp{};
  span_begin(c, "undistort", sp);
  hipLaunchKernelGGL(k_undistort, dim3((n + 255) / 256), dim3(256), 0, c->stream, n, d_p, d_c, m, d_prm);
  span_end(c, "undistort", sp);
  HIPCHK(c, hipGetLastError());
  HIPCHK(c, hipMemcpyAsync(pnt, d_p, (size_t)n * 3 * sizeof(double), hipMemcpyDefault, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));      // prm is a host temporary
  return VBA_OK;
}
int vba_map_cut_voxel_fix(vba_ctx *c, int n, const double *pnt_world, double jour) {
  return map_cut_voxel_fix(c->map, c->stream, n, pnt_world, jour, c->err);
}
int vba_map_recut(vba_ctx *c, int win_count, const double *poses, int multi) {
  int nf = 0;
  TimedSpan sp{};
  span_begin(c, "recut", sp);
  int st = map_recut(c->map, c->stream, win_count, poses, multi != 0, c->err, &nf);
  if (st) return st;
  // tras_opt: the map writes the planar leaves straight into the SoA factor store (no host round trip)
  c->nvox = 0;
  st = factor_reserve(c, nf > 0 ? nf : 1);
  if (st) return st;
  st = map_extract_factors(c->map, c->stream, c->fv, c->err, &nf);
  factor_update_mask(c, 0, nf);
  span_end(c, "recut", sp);
  if (st) return st;
  c->nvox = nf;
  return VBA_OK;
}
int vba_map_margi(vba_ctx *c, int win_count, const double *poses, double jour) {
  TimedSpan s{};
  span_begin(c, "margi", s);
  const int st = map_margi(c->map, c->stream, win_count, poses, jour, c->fv, c->nvox, c->err);
  span_end(c, "margi", s);
  return st;
}
int vba_map_slide(vba_ctx *c, int mgsize) { return map_slide(c->map, mgsize); }
int vba_map_prune(vba_ctx *c, double jour, int dist) { return map_prune(c->map, c->stream, jour, dist, c->err); }
int vba_map_reset(vba_ctx *c) { return map_reset(c->map, c->stream, c->err); }
int vba_map_num_roots(vba_ctx *c) { return map_num_roots(c->map, c->stream, false); }
int vba_map_num_slide_roots(vba_ctx *c) { return map_num_roots(c->map, c->stream, true); }
int vba_map_stats(vba_ctx *c, long long *out8) { return out8 ? map_stats(c->map, c->stream, out8, c->err) : VBA_ERR_BAD_ARG; }
// ---------------------------------------------------------------- odometry scan-to-map (VS:962-1098)
int vba_odom_lio_state_estimation(vba_ctx *c, int n, const double *pnt_body, const double *var_body, double *state, double *cov, int *ok) {
  if (n < 0 || (n > 0 && (!pnt_body || !var_body)) || !state || !cov) return VBA_ERR_BAD_ARG;
  const int DIM = VBA_DIM;
  // stage the scan once: [pts n*3 | var n*9 | partial nb*34 | out 34]
  const int nb = (n + 255) / 256;
  const size_t bytes = ((size_t)n * 12 + (size_t)nb * 34 + 64) * sizeof(double);
  int st = ensure_stage(c, bytes);
  if (st) return st;
  double *d_pts = (double *)c->d_stage, *d_var = d_pts + (size_t)n * 3, *d_part = d_var + (size_t)n * 9, *d_o34 = d_part + (size_t)nb * 34;
  if (n > 0) {
    HIPCHK(c, hipMemcpyAsync(d_pts, pnt_body, (size_t)n * 3 * sizeof(double), hipMemcpyDefault, c->stream));
    HIPCHK(c, hipMemcpyAsync(d_var, var_body, (size_t)n * 9 * sizeof(double), hipMemcpyDefault, c->stream));
  }
  vbh::State x_curr, x_prop;
  std::memcpy(&x_curr, state, sizeof(x_curr));
  x_prop = x_curr;                                                         // VS:965
  std::vector<double> P(cov, cov + 225), cov_inv(225);
  vbh::inverse_pplu(P.data(), cov_inv.data(), DIM);                        // VS:987
  const int num_max_iter = 4;
  int rematch_num = 0;
  double nnt[9] = {0};
  double G[225];
  for (int iter = 0; iter < num_max_iter; iter++) {
    OdomState X;
    std::memcpy(X.R, x_curr.R, sizeof(X.R)); std::memcpy(X.t, x_curr.p, sizeof(X.t));
    for (int r = 0; r < 3; r++) for (int k = 0; k < 3; k++) { X.rot_var[3 * r + k] = P[r * DIM + k]; X.tsl_var[3 * r + k] = P[(3 + r) * DIM + 3 + k]; }   // VS:1000-1001
    double s34[34];
    if (n > 0) { st = map_odom_accumulate(c->map, c->stream, X, n, d_pts, d_var, d_part, d_o34, s34, c->err); if (st) return st; }
    else std::memset(s34, 0, sizeof(s34));
    double HTH[36], HTz[6];
    { int idx = 0; for (int r = 0; r < 6; r++) for (int k = r; k < 6; k++) { HTH[r * 6 + k] = s34[idx]; HTH[k * 6 + r] = s34[idx]; idx++; } }
    for (int r = 0; r < 6; r++) HTz[r] = s34[21 + r];
    nnt[0] = s34[27]; nnt[1] = nnt[3] = s34[28]; nnt[2] = nnt[6] = s34[29]; nnt[4] = s34[30]; nnt[5] = nnt[7] = s34[31]; nnt[8] = s34[32];
    // K_1 = (H_T_H + cov_inv)^-1 ; G(:,0:6) = K_1(:,0:6) HTH ; solution = K_1(:,0:6) HTz + vec - G(:,0:6) vec(0:6)      VS:1056-1060
    std::vector<double> A(cov_inv), K1(225);
    for (int r = 0; r < 6; r++) for (int k = 0; k < 6; k++) A[r * DIM + k] += HTH[r * 6 + k];
    vbh::inverse_pplu(A.data(), K1.data(), DIM);
    std::memset(G, 0, sizeof(G));
    for (int r = 0; r < DIM; r++)
      for (int k = 0; k < 6; k++) { double sacc = 0; for (int j = 0; j < 6; j++) sacc += K1[r * DIM + j] * HTH[j * 6 + k]; G[r * DIM + k] = sacc; }
    double vec[15], RtR[9], lg[3];
    vbh::m3_Tmul(x_curr.R, x_prop.R, RtR);                                 // x_prop - x_curr: Log(b.R^T this.R)  TL:164-173
    vbh::so3_log(RtR, lg);
    for (int k = 0; k < 3; k++) { vec[k] = lg[k]; vec[3 + k] = x_prop.p[k] - x_curr.p[k]; vec[6 + k] = x_prop.v[k] - x_curr.v[k]; vec[9 + k] = x_prop.bg[k] - x_curr.bg[k]; vec[12 + k] = x_prop.ba[k] - x_curr.ba[k]; }
    double sol[15];
    for (int r = 0; r < DIM; r++) {
      double a = 0, b = 0;
      for (int j = 0; j < 6; j++) { a += K1[r * DIM + j] * HTz[j]; b += G[r * DIM + j] * vec[j]; }
      sol[r] = a + vec[r] - b;
    }
    double E[9], Rn[9];                                                    // x_curr += solution  TL:154-162
    vbh::so3_exp(sol, E);
    vbh::m3_mul(x_curr.R, E, Rn);
    std::memcpy(x_curr.R, Rn, sizeof(Rn));
    for (int k = 0; k < 3; k++) { x_curr.p[k] += sol[3 + k]; x_curr.v[k] += sol[6 + k]; x_curr.bg[k] += sol[9 + k]; x_curr.ba[k] += sol[12 + k]; }
    const double rot_add = vbh::norm3(sol), tra_add = vbh::norm3(sol + 3);
    const bool converged = (rot_add * 57.3 < 0.01) && (tra_add * 100 < 0.015);     // VS:1072
    if (converged || ((rematch_num == 0) && (iter == num_max_iter - 2))) rematch_num++;   // VS:1076-1079
    if (rematch_num >= 2 || (iter == num_max_iter - 1)) {                  // x_curr.cov = (I - G) cov   VS:1082-1086
      std::vector<double> IG(225), Pn(225);
      for (int r = 0; r < DIM; r++) for (int k = 0; k < DIM; k++) IG[r * DIM + k] = (r == k ? 1.0 : 0.0) - G[r * DIM + k];
      vbh::mat_mul(IG.data(), P.data(), Pn.data(), DIM, DIM, DIM);
      P = Pn;
      break;
    }
  }
  std::memcpy(state, &x_curr, sizeof(x_curr));
  std::memcpy(cov, P.data(), 225 * sizeof(double));
  if (ok) {
    // SelfAdjointEigenSolver(nnt).eigenvalues()[0] < 14 -> false  (VS:1090-1097); closed form is not needed here: Jacobi on host
    double a[3][3] = {{nnt[0], nnt[1], nnt[2]}, {nnt[3], nnt[4], nnt[5]}, {nnt[6], nnt[7], nnt[8]}};
    for (int sweep = 0; sweep < 60; sweep++) {
      const double off = std::fabs(a[0][1]) + std::fabs(a[0][2]) + std::fabs(a[1][2]);
      if (off == 0.0) break;
      for (int p = 0; p < 2; p++)
        for (int q = p + 1; q < 3; q++) {
          if (a[p][q] == 0.0) continue;
          const double theta = 0.5 * (a[q][q] - a[p][p]) / a[p][q];
          double t = 1.0 / (std::fabs(theta) + std::sqrt(1.0 + theta * theta));
          if (theta < 0) t = -t;
          const double cth = 1.0 / std::sqrt(1 + t * t), sth = t * cth, apq = a[p][q];
          const int r = 3 - p - q;
          const double arp = a[r][p], arq = a[r][q];
          a[p][p] -= t * apq; a[q][q] += t * apq; a[p][q] = a[q][p] = 0.0;
          a[r][p] = a[p][r] = cth * arp - sth * arq; a[r][q] = a[q][r] = sth * arp + cth * arq;
          if (std::fabs(a[r][p]) < 1e-300) a[r][p] = a[p][r] = 0.0;
          if (std::fabs(a[r][q]) < 1e-300) a[r][q] = a[q][r] = 0.0;
        }
      if (off < 1e-14 * (std::fabs(a[0][0]) + std::fabs(a[1][1]) + std::fabs(a[2][2]))) break;
    }
    const double emin = std::min(a[0][0], std::min(a[1][1], a[2][2]));
    *ok = (emin < 14) ? 0 : 1;
  }
  return VBA_OK;
}

int vba_map_dump_leaves(vba_ctx *c, double *out, int max_leaves) { return map_dump_leaves(c->map, c->stream, out, max_leaves, c->err); }
int vba_map_dump_plane_var(vba_ctx *c, double *out, int max_leaves) { return map_dump_plane_var(c->map, c->stream, out, max_leaves, c->err); }

// ---------------------------------------------------------------- session-store formats (vba_io.hpp), host only
int vba_io_save_pcd(const char *path, int n, const double *xyz) {
  if (!path || n < 0 || (n > 0 && !xyz)) return VBA_ERR_BAD_ARG;
  FILE *f = std::fopen(path, "wb");
  if (!f) return VBA_ERR_IO;
  std::fprintf(f, "# .PCD v0.7 - Point Cloud Data file format\nVERSION 0.7\nFIELDS x y z intensity\nSIZE 4 4 4 4\nTYPE F F F F\nCOUNT 1 1 1 1\n"
                  "WIDTH %d\nHEIGHT 1\nVIEWPOINT 0 0 0 1 0 0 0\nPOINTS %d\nDATA binary\n", n, n);
  std::vector<float> rec((size_t)n * 4);
  for (int i = 0; i < n; i++) {                                // save_pcd sets x, y, z only: intensity keeps PointXYZI's default 0 (VS:170-176)
    rec[4 * (size_t)i] = (float)xyz[3 * (size_t)i]; rec[4 * (size_t)i + 1] = (float)xyz[3 * (size_t)i + 1];
    rec[4 * (size_t)i + 2] = (float)xyz[3 * (size_t)i + 2]; rec[4 * (size_t)i + 3] = 0.f;
  }
  const size_t w = n > 0 ? std::fwrite(rec.data(), 16, (size_t)n, f) : 0;
  const int bad = std::fclose(f);
  return (w == (size_t)n && !bad) ? VBA_OK : VBA_ERR_IO;
}

int vba_io_load_pcd(const char *path, int cap, double *xyz, double *intensity, int *n_out) {
  if (!path || !n_out || cap < 0 || (cap > 0 && !xyz)) return VBA_ERR_BAD_ARG;
  *n_out = 0;
  FILE *f = std::fopen(path, "rb");
  if (!f) return VBA_ERR_IO;
  struct Close { FILE *f; ~Close() { std::fclose(f); } } closer{f};
  vba_io::PcdHeader h;
  char line[1024];
  while (h.data.empty()) {
    if (!std::fgets(line, sizeof(line), f)) return VBA_ERR_IO;
    std::istringstream ss(line);
    std::string key, tok;
    if (!(ss >> key) || key[0] == '#') continue;
    if (key == "FIELDS" || key == "COLUMNS") while (ss >> tok) h.fields.push_back(tok);
    else if (key == "SIZE") while (ss >> tok) h.size.push_back(std::atoi(tok.c_str()));
    else if (key == "TYPE") while (ss >> tok) h.type.push_back(tok);
    else if (key == "COUNT") while (ss >> tok) h.count.push_back(std::atoi(tok.c_str()));
    else if (key == "WIDTH") ss >> h.width;
    else if (key == "HEIGHT") ss >> h.height;
    else if (key == "POINTS") ss >> h.points;
    else if (key == "DATA") ss >> h.data;
  }
  const size_t nf = h.fields.size();
  if (nf == 0 || h.size.size() != nf || h.type.size() != nf) return VBA_ERR_IO;
  if (h.count.empty()) h.count.assign(nf, 1);
  if (h.count.size() != nf) return VBA_ERR_IO;
  if (h.points < 0) h.points = h.width * h.height;
  if (h.points < 0) return VBA_ERR_IO;
  int fx = -1, fy = -1, fz = -1, fi = -1;
  std::vector<size_t> off(nf);
  size_t stride = 0;
  for (size_t k = 0; k < nf; k++) {
    off[k] = stride; stride += (size_t)h.size[k] * (size_t)h.count[k];
    if (h.fields[k] == "x") fx = (int)k; else if (h.fields[k] == "y") fy = (int)k; else if (h.fields[k] == "z") fz = (int)k;
    else if (h.fields[k] == "intensity") fi = (int)k;
  }
  if (fx < 0 || fy < 0 || fz < 0) return VBA_ERR_IO;
  *n_out = (int)h.points;
  if (h.points > cap) return VBA_ERR_CAPACITY;                 // *n_out tells the caller what to allocate
  auto scalar = [&](const unsigned char *p, size_t k) -> double {
    const char t = h.type[k][0]; const int sz = h.size[k];
    if (t == 'F') { if (sz == 4) { float v; std::memcpy(&v, p, 4); return v; } if (sz == 8) { double v; std::memcpy(&v, p, 8); return v; } }
    if (t == 'U') { uint64_t v = 0; std::memcpy(&v, p, (size_t)sz); return (double)v; }              // little endian
    if (t == 'I') { int64_t v = 0; std::memcpy(&v, p, (size_t)sz); const int sh = 64 - 8 * sz; return (double)((v << sh) >> sh); }
    return 0.0;
  };
  if (h.data == "binary") {
    std::vector<unsigned char> buf((size_t)h.points * stride);
    if (h.points > 0 && std::fread(buf.data(), stride, (size_t)h.points, f) != (size_t)h.points) return VBA_ERR_IO;
    for (long i = 0; i < h.points; i++) {
      const unsigned char *r = buf.data() + (size_t)i * stride;
      xyz[3 * i] = scalar(r + off[fx], fx); xyz[3 * i + 1] = scalar(r + off[fy], fy); xyz[3 * i + 2] = scalar(r + off[fz], fz);
      if (intensity) intensity[i] = fi >= 0 ? scalar(r + off[fi], fi) : 0.0;
    }
  } else if (h.data == "ascii") {
    for (long i = 0; i < h.points; i++) {
      if (!std::fgets(line, sizeof(line), f)) return VBA_ERR_IO;
      std::istringstream ss(line);
      if (intensity) intensity[i] = 0.0;
      for (size_t k = 0; k < nf; k++)
        for (int cidx = 0; cidx < h.count[k]; cidx++) {
          double v;
          if (!(ss >> v)) return VBA_ERR_IO;
          if (cidx) continue;
          if ((int)k == fx) xyz[3 * i] = v; else if ((int)k == fy) xyz[3 * i + 1] = v; else if ((int)k == fz) xyz[3 * i + 2] = v;
          else if ((int)k == fi && intensity) intensity[i] = v;
        }
    }
  } else {
    return VBA_ERR_IO;                                          // binary_compressed: never written by the reference (VS:178)
  }
  return VBA_OK;
}

int vba_io_save_pose(const char *path, int n, const double *states, const double *v6) {
  if (!path || n < 0 || (n > 0 && (!states || !v6))) return VBA_ERR_BAD_ARG;
  if (n < 100) return VBA_OK;                                   // VS:183-184: short sessions are not saved
  FILE *f = std::fopen(path, "w");
  if (!f) return VBA_ERR_IO;
  for (int i = 0; i < n; i++) {
    vbh::State x;
    std::memcpy(&x, states + (size_t)i * 25, sizeof(x));
    double q[4];
    vba_io::quat_from_rot(x.R, q);
    std::fprintf(f, "%.6f ", x.t);                              // fixed, precision 6; then precision 7 for the rest (VS:192-193)
    std::fprintf(f, "%.7f %.7f %.7f ", x.p[0], x.p[1], x.p[2]);
    std::fprintf(f, "%.7f %.7f %.7f %.7f", q[0], q[1], q[2], q[3]);
    const double *grp[4] = {x.v, x.bg, x.ba, x.g};
    for (int g = 0; g < 4; g++) std::fprintf(f, " %.7f %.7f %.7f", grp[g][0], grp[g][1], grp[g][2]);
    for (int j = 0; j < 6; j++) std::fprintf(f, " %.7f", v6[(size_t)i * 6 + j]);
    std::fprintf(f, "\n");
  }
  return std::fclose(f) ? VBA_ERR_IO : VBA_OK;
}

int vba_io_read_lidarstate(const char *path, int cap, double *states, double *v6, int *n_out) {
  if (!path || !n_out || cap < 0 || (cap > 0 && !states)) return VBA_ERR_BAD_ARG;
  *n_out = 0;
  FILE *f = std::fopen(path, "r");
  if (!f) return VBA_ERR_IO;                                    // the reference prints "not found" and exits (VH:271-275)
  struct Close { FILE *f; ~Close() { std::fclose(f); } } closer{f};
  std::vector<char> line(1 << 16);
  int n = 0;
  while (std::fgets(line.data(), (int)line.size(), f)) {
    std::vector<double> nums;
    char *p = line.data();
    for (;;) {
      char *e = nullptr;
      const double v = std::strtod(p, &e);
      if (e == p) break;
      nums.push_back(v); p = e;
    }
    if (nums.size() < 8) { if (nums.empty()) continue; return VBA_ERR_IO; }
    if (n < cap) {
      vbh::State x;
      std::memset(&x, 0, sizeof(x));
      x.g[2] = -9.8;                                            // lines without g: the reference leaves IMUST::g unset (TL:188-197); gravity here
      x.t = nums[0];
      for (int k = 0; k < 3; k++) x.p[k] = nums[1 + k];
      const double q[4] = {nums[4], nums[5], nums[6], nums[7]};
      vba_io::rot_from_quat(q, x.R);
      if (nums.size() >= 20)
        for (int k = 0; k < 3; k++) { x.v[k] = nums[8 + k]; x.bg[k] = nums[11 + k]; x.ba[k] = nums[14 + k]; x.g[k] = nums[17 + k]; }
      std::memcpy(states + (size_t)n * 25, &x, sizeof(x));
      if (v6) for (int k = 0; k < 6; k++) v6[(size_t)n * 6 + k] = nums.size() >= 26 ? nums[20 + k] : 0.0;
    }
    n++;
  }
  *n_out = n;
  return n > cap ? VBA_ERR_CAPACITY : VBA_OK;
}

}  // extern "C"
