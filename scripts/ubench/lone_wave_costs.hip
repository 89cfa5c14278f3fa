// micro-benchmark (r02): what a wavefront ALONE on its SIMD pays for the patterns of the solver's equality rounds, in shader cycles
// (s_memtime): dependent fp64 fma, fma -> DPP hand-over -> fma, fma -> v_cndmask -> fma, LDS store -> load of the same word,
// and the same LDS round trip with independent fp64 work in its shadow.
#include <hip/hip_runtime.h>
#include <cstdio>
__device__ __forceinline__ double dppx(double x) {
  int lo = __double2loint(x), hi = __double2hiint(x);
  lo = __builtin_amdgcn_update_dpp(0, lo, 0xB1, 0xF, 0xF, true);
  hi = __builtin_amdgcn_update_dpp(0, hi, 0xB1, 0xF, 0xF, true);
  return __hiloint2double(hi, lo);
}
#define T0 long long t0 = __builtin_readcyclecounter()
#define T1(slot, per) do { long long t1 = __builtin_readcyclecounter(); if (threadIdx.x == 0 && blockIdx.x == 0) out[64 + slot] = (double)(t1 - t0) / (per); } while (0)
__global__ void k_dep(double* out, int n, double a, double b) {
  double x = out[threadIdx.x];
  T0;
  for (int i = 0; i < n; i++) {
#pragma unroll
    for (int k = 0; k < 16; k++) x = fma(x, a, b);
  }
  T1(0, 16.0 * n);
  out[threadIdx.x] = x;
}
__global__ void k_dpp(double* out, int n, double a, double b) {  // fma -> dpp -> fma -> dpp ...
  double x = out[threadIdx.x];
  T0;
  for (int i = 0; i < n; i++) {
#pragma unroll
    for (int k = 0; k < 8; k++) x = fma(dppx(x), a, b);
  }
  T1(1, 8.0 * n);
  out[threadIdx.x] = x;
}
__global__ void k_sel(double* out, int n, double a, double b) {  // fma -> cndmask -> fma
  double x = out[threadIdx.x], y = x + 1;
  const bool s = threadIdx.x & 1;
  T0;
  for (int i = 0; i < n; i++) {
#pragma unroll
    for (int k = 0; k < 8; k++) { double t = fma(x, a, b); x = s ? t : y; }
  }
  T1(2, 8.0 * n);
  out[threadIdx.x] = x;
}
__global__ void k_lds(double* out, int n, double a, double b) {  // store -> load of the same word -> fma
  __shared__ double L[64 * 4];
  double x = out[threadIdx.x];
  volatile double* p = L + threadIdx.x;
  T0;
  for (int i = 0; i < n; i++) {
#pragma unroll
    for (int k = 0; k < 8; k++) { *p = x; x = fma(*p, a, b); }
  }
  T1(3, 8.0 * n);
  out[threadIdx.x] = x;
}
__global__ void k_lds_shadow(double* out, int n, double a, double b) {  // the same with 8 independent fmas between store and use
  __shared__ double L[64 * 4];
  double x = out[threadIdx.x], y0 = x + 1, y1 = x + 2, y2 = x + 3, y3 = x + 4;
  volatile double* p = L + threadIdx.x;
  T0;
  for (int i = 0; i < n; i++) {
#pragma unroll
    for (int k = 0; k < 8; k++) {
      *p = x;
      double v = *p;
      y0 = fma(y0, a, b); y1 = fma(y1, a, b); y2 = fma(y2, a, b); y3 = fma(y3, a, b);
      y0 = fma(y0, a, b); y1 = fma(y1, a, b); y2 = fma(y2, a, b); y3 = fma(y3, a, b);
      x = fma(v, a, b);
    }
  }
  T1(4, 8.0 * n);
  out[threadIdx.x] = x + y0 + y1 + y2 + y3;
}
__global__ void k_indep(double* out, int n, double a, double b) {  // 8 independent fma streams: issue cadence
  double x0 = out[threadIdx.x], x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3, x4 = x0 + 4, x5 = x0 + 5, x6 = x0 + 6, x7 = x0 + 7;
  T0;
  for (int i = 0; i < n; i++) {
#pragma unroll
    for (int k = 0; k < 2; k++) { x0 = fma(x0, a, b); x1 = fma(x1, a, b); x2 = fma(x2, a, b); x3 = fma(x3, a, b); x4 = fma(x4, a, b); x5 = fma(x5, a, b); x6 = fma(x6, a, b); x7 = fma(x7, a, b); }
  }
  T1(5, 16.0 * n);
  out[threadIdx.x] = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7;
}
__global__ void k_int(double* out, int n) {  // dependent 32-bit integer adds
  int x = (int)out[threadIdx.x];
  T0;
  for (int i = 0; i < n; i++) {
#pragma unroll
    for (int k = 0; k < 16; k++) x = x * 3 + k;
  }
  T1(6, 16.0 * n);
  out[threadIdx.x] = x;
}
int main() {
  double* d; hipMalloc(&d, 2048); hipMemset(d, 0, 2048);
  for (int rep = 0; rep < 2; rep++) {
    hipLaunchKernelGGL(k_dep, dim3(1), dim3(64), 0, 0, d, 2048, 0.999, 0.001);
    hipLaunchKernelGGL(k_dpp, dim3(1), dim3(64), 0, 0, d, 2048, 0.999, 0.001);
    hipLaunchKernelGGL(k_sel, dim3(1), dim3(64), 0, 0, d, 2048, 0.999, 0.001);
    hipLaunchKernelGGL(k_lds, dim3(1), dim3(64), 0, 0, d, 2048, 0.999, 0.001);
    hipLaunchKernelGGL(k_lds_shadow, dim3(1), dim3(64), 0, 0, d, 2048, 0.999, 0.001);
    hipLaunchKernelGGL(k_indep, dim3(1), dim3(64), 0, 0, d, 2048, 0.999, 0.001);
    hipLaunchKernelGGL(k_int, dim3(1), dim3(64), 0, 0, d, 2048);
    hipDeviceSynchronize();
  }
  double h[96]; hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
  printf("s_memtime ticks per step, one wavefront alone: dep fma %.1f | fma+dpp hand-over %.1f | fma+cndmask %.1f | LDS store->load->fma %.1f | same with 8 indep fma in the shadow %.1f | indep fma (issue) %.1f | dep int mad %.1f\n",
         h[64], h[65], h[66], h[67], h[68], h[69], h[70]);
  return 0;
}
