// Microbenchmark: the MFMA loop of K3's phase B (5 units x 9 k-steps per wave, operands from LDS) in isolation, with variants,
// to find what keeps it at ~2x the 64-cycle issue rate of v_mfma_f64_16x16x4_f64.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double v4 __attribute__((ext_vector_type(4)));
constexpr int GS = 80, NK = 72, UPW = 5, KSTEPS = 9;
template <int MODE>
__global__ __launch_bounds__(256) void k(int iters, double *out, long long *cyc) {
  __shared__ double G[NK * GS + NK];
  double *cK = G + NK * GS;
  for (int t = threadIdx.x; t < NK * GS + NK; t += 256) G[t] = 1e-3 * (t % 97);
  __syncthreads();
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, kr = lane >> 4, cl = lane & 15;
  v4 acc[UPW];
  for (int t = 0; t < UPW; t++) acc[t] = (v4){0, 0, 0, 0};
  const double *gap[UPW], *gbp[UPW], *ckp[UPW];
  for (int t = 0; t < UPW; t++) {
    const int unit = wv * UPW + t, u = unit % 10, ksp = unit / 10;
    int p = u, ta = 0;
    while (p >= 4 - ta) { p -= 4 - ta; ta++; }
    const int tb = ta + p, kb = ksp * KSTEPS * 4;
    gap[t] = G + (kb + kr) * GS + 16 * ta + cl;
    gbp[t] = G + (kb + kr) * GS + 16 * tb + cl;
    ckp[t] = cK + kb + kr;
  }
  const long long t0 = clock64();
  for (int it = 0; it < iters; it++) {
    if (MODE == 0) {            // as in K3: round-robin, prefetch one round ahead
      double ar[UPW], cr[UPW], br[UPW];
#pragma unroll
      for (int t = 0; t < UPW; t++) { ar[t] = gap[t][0]; cr[t] = ckp[t][0]; br[t] = gbp[t][0]; }
#pragma unroll
      for (int ks = 0; ks < KSTEPS; ks++) {
        double av[UPW], bv[UPW];
#pragma unroll
        for (int t = 0; t < UPW; t++) { av[t] = ar[t] * cr[t]; bv[t] = br[t]; }
        if (ks + 1 < KSTEPS) {
          const int kk = 4 * (ks + 1);
#pragma unroll
          for (int t = 0; t < UPW; t++) { ar[t] = gap[t][kk * GS]; cr[t] = ckp[t][kk]; br[t] = gbp[t][kk * GS]; }
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int t = 0; t < UPW; t++) acc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(av[t], bv[t], acc[t], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
      }
    } else if (MODE == 1) {     // operands loaded once (no LDS traffic in the loop), same MFMA sequence
      double ar[UPW], br[UPW];
#pragma unroll
      for (int t = 0; t < UPW; t++) { ar[t] = gap[t][0]; br[t] = gbp[t][0]; }
#pragma unroll
      for (int ks = 0; ks < KSTEPS; ks++) {
#pragma unroll
        for (int t = 0; t < UPW; t++) acc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(ar[t], br[t], acc[t], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
      }
    } else if (MODE == 2) {     // LDS loads but no multiply (A already scaled)
      double ar[UPW], br[UPW];
#pragma unroll
      for (int t = 0; t < UPW; t++) { ar[t] = gap[t][0]; br[t] = gbp[t][0]; }
#pragma unroll
      for (int ks = 0; ks < KSTEPS; ks++) {
        double av[UPW], bv[UPW];
#pragma unroll
        for (int t = 0; t < UPW; t++) { av[t] = ar[t]; bv[t] = br[t]; }
        if (ks + 1 < KSTEPS) {
          const int kk = 4 * (ks + 1);
#pragma unroll
          for (int t = 0; t < UPW; t++) { ar[t] = gap[t][kk * GS]; br[t] = gbp[t][kk * GS]; }
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int t = 0; t < UPW; t++) acc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(av[t], bv[t], acc[t], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
      }
    } else if (MODE == 3) {     // MODE 0 in VGPR accumulators? (same code; compiled with accumulators forced through "v" constraints)
      double ar[UPW], cr[UPW], br[UPW];
#pragma unroll
      for (int t = 0; t < UPW; t++) { ar[t] = gap[t][0]; cr[t] = ckp[t][0]; br[t] = gbp[t][0]; }
#pragma unroll
      for (int ks = 0; ks < KSTEPS; ks++) {
        double av[UPW], bv[UPW];
#pragma unroll
        for (int t = 0; t < UPW; t++) { av[t] = ar[t] * cr[t]; bv[t] = br[t]; }
        if (ks + 1 < KSTEPS) {
          const int kk = 4 * (ks + 1);
#pragma unroll
          for (int t = 0; t < UPW; t++) { ar[t] = gap[t][kk * GS]; cr[t] = ckp[t][kk]; br[t] = gbp[t][kk * GS]; }
        }
#pragma unroll
        for (int t = 0; t < UPW; t++) { asm volatile("" : "+v"(acc[t])); acc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(av[t], bv[t], acc[t], 0, 0, 0); asm volatile("" : "+v"(acc[t])); }
      }
    }
  }
  const long long t1 = clock64();
  double r = 0;
  for (int t = 0; t < UPW; t++) r += acc[t][0] + acc[t][1] + acc[t][2] + acc[t][3];
  out[blockIdx.x * 256 + threadIdx.x] = r;
  if (lane == 0) cyc[blockIdx.x * 4 + wv] = t1 - t0;
}
template <int MODE>
void run(const char *name, int nb, double *out, long long *cyc) {
  const int iters = 200;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(k<MODE>, dim3(nb), dim3(256), 0, 0, iters, out, cyc);
  hipDeviceSynchronize();
  hipEventRecord(e0, 0);
  hipLaunchKernelGGL(k<MODE>, dim3(nb), dim3(256), 0, 0, iters, out, cyc);
  hipEventRecord(e1, 0);
  hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  long long h[4]; hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
  printf("grid %3d  %-44s %8.1f us | cycles per MFMA: %.1f %.1f %.1f %.1f | %.1f ns per MFMA\n", nb, name, ms * 1e3, h[0] / (45.0 * iters), h[1] / (45.0 * iters),
         h[2] / (45.0 * iters), h[3] / (45.0 * iters), ms * 1e6 / (45.0 * iters));
}
int main() {
  double *out; long long *cyc;
  hipMalloc(&out, 256 * 256 * 8); hipMalloc(&cyc, 256 * 4 * 8);
  for (int nb : {1, 256}) {
    run<0>("as K3 (LDS a, b, ck; multiply; prefetch)", nb, out, cyc);
    run<1>("operands in registers, no LDS in the loop", nb, out, cyc);
    run<2>("LDS a, b only (no ck, no multiply)", nb, out, cyc);
    run<3>("as K3, accumulators pinned to VGPRs", nb, out, cyc);
  }
  return 0;
}
