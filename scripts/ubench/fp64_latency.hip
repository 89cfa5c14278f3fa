// micro-benchmark: dependent vs independent v_fma_f64 / rcp chains, one wave per SIMD (MI355X)
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void dep_chain(double* out, int n, double a, double b) {
  double x = out[threadIdx.x];
  long long t0 = clock64();
  for (int i = 0; i < n; i++) {
#pragma unroll
    for (int k = 0; k < 16; k++) x = fma(x, a, b);
  }
  long long t1 = clock64();
  out[threadIdx.x] = x;
  if (threadIdx.x == 0 && blockIdx.x == 0) out[64] = (double)(t1 - t0) / (16.0 * n);
}
__global__ void indep_chain(double* out, int n, double a, double b) {
  double x0 = out[threadIdx.x], x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3;
  long long t0 = clock64();
  for (int i = 0; i < n; i++) {
#pragma unroll
    for (int k = 0; k < 4; k++) { x0 = fma(x0, a, b); x1 = fma(x1, a, b); x2 = fma(x2, a, b); x3 = fma(x3, a, b); }
  }
  long long t1 = clock64();
  out[threadIdx.x] = x0 + x1 + x2 + x3;
  if (threadIdx.x == 0 && blockIdx.x == 0) out[65] = (double)(t1 - t0) / (16.0 * n);
}
__global__ void rcp_chain(double* out, int n) {
  double x = out[threadIdx.x] + 1.5;
  long long t0 = clock64();
  for (int i = 0; i < n; i++) {
#pragma unroll
    for (int k = 0; k < 8; k++) x = __builtin_amdgcn_rcp(x) + 1.0;
  }
  long long t1 = clock64();
  out[threadIdx.x] = x;
  if (threadIdx.x == 0 && blockIdx.x == 0) out[66] = (double)(t1 - t0) / (8.0 * n);
}
__global__ void dep_chain_f32(float* out, int n, float a, float b) {
  float x = out[threadIdx.x];
  long long t0 = clock64();
  for (int i = 0; i < n; i++) {
#pragma unroll
    for (int k = 0; k < 16; k++) x = fmaf(x, a, b);
  }
  long long t1 = clock64();
  out[threadIdx.x] = x;
  if (threadIdx.x == 0 && blockIdx.x == 0) out[67] = (float)((double)(t1 - t0) / (16.0 * n));
}
int main() {
  double* d; hipMalloc(&d, 1024); hipMemset(d, 0, 1024);
  float* f; hipMalloc(&f, 1024); hipMemset(f, 0, 1024);
  for (int rep = 0; rep < 2; rep++) {
    hipLaunchKernelGGL(dep_chain, dim3(1), dim3(64), 0, 0, d, 4096, 0.999, 0.001);
    hipLaunchKernelGGL(indep_chain, dim3(1), dim3(64), 0, 0, d, 4096, 0.999, 0.001);
    hipLaunchKernelGGL(rcp_chain, dim3(1), dim3(64), 0, 0, d, 4096);
    hipLaunchKernelGGL(dep_chain_f32, dim3(1), dim3(64), 0, 0, f, 4096, 0.999f, 0.001f);
    hipDeviceSynchronize();
  }
  double h[72]; hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
  float hf[72]; hipMemcpy(hf, f, sizeof hf, hipMemcpyDeviceToHost);
  printf("clock64 ticks per op (s_memtime ticks = shader cycles? see guide): dep fma_f64 %.2f  indep(4) fma_f64 %.2f  dep rcp_f64+add %.2f  dep fma_f32 %.2f\n", h[64], h[65], h[66], hf[67]);
  return 0;
}
