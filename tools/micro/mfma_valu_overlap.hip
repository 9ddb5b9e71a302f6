// Microbenchmark: do a f64-MFMA wave and a f64-VALU wave on the same SIMD run concurrently on gfx950?
// (decides whether K3's slot preparation (VALU) can hide behind its MFMA phase with specialised waves)
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double v4 __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(512) void k(int mode, int iters, double *out, long long *cyc) {
  const int w = threadIdx.x >> 6;
  const bool do_mfma = (mode == 0) || ((mode == 2 || mode == 3) && w < 4);
  const bool do_valu = (mode == 1) || ((mode == 2 || mode == 4) && w >= 4);
  const long long t0 = clock64();
  double r = 0;
  if (do_mfma) {
    v4 a0 = {0, 0, 0, 0}, a1 = a0, a2 = a0, a3 = a0;
    double x = threadIdx.x * 1e-3, y = 1.0 + threadIdx.x * 1e-6;
    for (int i = 0; i < iters; i++) {
      a0 = __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, a0, 0, 0, 0);
      a1 = __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, a1, 0, 0, 0);
      a2 = __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, a2, 0, 0, 0);
      a3 = __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, a3, 0, 0, 0);
    }
    r = a0[0] + a1[1] + a2[2] + a3[3];
  } else if (do_valu) {
    double c0 = threadIdx.x, c1 = c0 + 1, c2 = c0 + 2, c3 = c0 + 3, c4 = c0 + 4, c5 = c0 + 5, c6 = c0 + 6, c7 = c0 + 7;
    const double m = 1.0000001, b = 1e-9;
    for (int i = 0; i < iters; i++) {
#pragma unroll
      for (int q = 0; q < 2; q++) {
        c0 = fma(c0, m, b); c1 = fma(c1, m, b); c2 = fma(c2, m, b); c3 = fma(c3, m, b);
        c4 = fma(c4, m, b); c5 = fma(c5, m, b); c6 = fma(c6, m, b); c7 = fma(c7, m, b);
      }
    }
    r = c0 + c1 + c2 + c3 + c4 + c5 + c6 + c7;
  }
  const long long t1 = clock64();
  out[blockIdx.x * 512 + threadIdx.x] = r;
  if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 8 + w] = t1 - t0;
}
int main() {
  double *out; long long *cyc;
  const int NB = 256;
  hipMalloc(&out, NB * 512 * 8); hipMalloc(&cyc, NB * 8 * 8);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int iters = 2000;
  const char *names[] = {"all 8 waves MFMA (2/SIMD)", "all 8 waves VALU f64 (2/SIMD)", "4 MFMA + 4 VALU waves (1+1/SIMD)", "4 MFMA waves only (1/SIMD)", "4 VALU waves only (1/SIMD)"};
  for (int nb : {1, NB})
    for (int mode = 0; mode < 5; mode++) {
      hipLaunchKernelGGL(k, dim3(nb), dim3(512), 0, 0, mode, iters, out, cyc);
      hipDeviceSynchronize();
      hipEventRecord(e0, 0);
      hipLaunchKernelGGL(k, dim3(nb), dim3(512), 0, 0, mode, iters, out, cyc);
      hipEventRecord(e1, 0);
      hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      long long h[8]; hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
      // per-wave work: MFMA wave = 4*iters MFMAs; VALU wave = 16*iters FMAs
      printf("grid %3d  %-36s %8.1f us | wave0 %lld ticks, wave4 %lld ticks | per MFMA %.1f ns, per f64 FMA %.2f ns\n", nb, names[mode], ms * 1e3, h[0], h[4],
             ms * 1e6 / (4.0 * iters), ms * 1e6 / (16.0 * iters));
    }
  return 0;
}
