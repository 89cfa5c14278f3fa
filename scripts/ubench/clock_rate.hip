// micro-benchmark: what clock64() (s_memtime) counts, and the real rate of a lone wavefront's dependent / independent fp64 FMAs
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void dep(double* out, int n, double a, double b) {
  double x = out[threadIdx.x];
  long long t0 = clock64(), w0 = wall_clock64();
  for (int i = 0; i < n; i++) {
#pragma unroll
    for (int k = 0; k < 16; k++) x = fma(x, a, b);
  }
  long long t1 = clock64(), w1 = wall_clock64();
  out[threadIdx.x] = x;
  if (threadIdx.x == 0) { out[64 + 2 * blockIdx.x] = (double)(t1 - t0); out[65 + 2 * blockIdx.x] = (double)(w1 - w0); }
}
__global__ void indep(double* out, int n, double a, double b) {
  double x0 = out[threadIdx.x], x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3;
  long long t0 = clock64(), w0 = wall_clock64();
  for (int i = 0; i < n; i++) {
#pragma unroll
    for (int k = 0; k < 4; k++) { x0 = fma(x0, a, b); x1 = fma(x1, a, b); x2 = fma(x2, a, b); x3 = fma(x3, a, b); }
  }
  long long t1 = clock64(), w1 = wall_clock64();
  out[threadIdx.x] = x0 + x1 + x2 + x3;
  if (threadIdx.x == 0) { out[64 + 2 * blockIdx.x] = (double)(t1 - t0); out[65 + 2 * blockIdx.x] = (double)(w1 - w0); }
}
int main() {
  double* d; hipMalloc(&d, 65536); hipMemset(d, 0, 65536);
  double h[4096];
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  int wall_khz = 0; hipDeviceGetAttribute(&wall_khz, hipDeviceAttributeWallClockRate, 0);
  int clk_khz = 0; hipDeviceGetAttribute(&clk_khz, hipDeviceAttributeClockRate, 0);
  printf("wall clock rate %d kHz, device clock rate attribute %d kHz\n", wall_khz, clk_khz);
  const int n = 200000;
  for (int blocks : {1, 512, 1024, 4096}) {
    for (int kind = 0; kind < 2; kind++) {
      for (int rep = 0; rep < 2; rep++) {
        hipEventRecord(e0);
        if (kind == 0) dep<<<blocks, 64>>>(d, n, 0.999, 1e-3); else indep<<<blocks, 64>>>(d, n, 0.999, 1e-3);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        hipMemcpy(h, d, 1024, hipMemcpyDeviceToHost);
        if (rep == 1)
          printf("%s blocks %4d: %.3f ms, %.2f ns per FMA, clock64 %.2f ticks/FMA (%.0f MHz), wall_clock64 %.0f ticks\n", kind ? "indep" : "dep  ", blocks, ms,
                 ms * 1e6 / (16.0 * n), h[64] / (16.0 * n), h[64] / (ms * 1e3), h[65]);
      }
    }
  }
  return 0;
}
