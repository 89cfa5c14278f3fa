// micro-benchmark: v_mfma_f64_4x4x4_4b_f64 as a cross-lane sum / broadcast inside 16-lane rows, against the DPP quad sum.
// Prints the register layout (which lanes of A/B feed which lanes of D) and the issue/latency cost in a dependent chain.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

__device__ __forceinline__ double mfma444(double a, double b, double c) { return __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c, 0, 0, 0); }

template <int CTRL>
__device__ __forceinline__ double dpp(double x) {
  int lo = __double2loint(x), hi = __double2hiint(x);
  lo = __builtin_amdgcn_update_dpp(0, lo, CTRL, 0xF, 0xF, true);
  hi = __builtin_amdgcn_update_dpp(0, hi, CTRL, 0xF, 0xF, true);
  return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double qsum(double x) { x += dpp<0xB1>(x); x += dpp<0x4E>(x); return x; }

// layout probes: out[0][l] = D when B = lane id, A = 1; out[1][l] = D when A = lane id, B = 1; out[2]: A = (lane/4 == 2), B = lane id
__global__ void layout(double* out) {
  const int l = threadIdx.x;
  out[l] = mfma444(1.0, (double)l, 0.0);
  out[64 + l] = mfma444((double)l, 1.0, 0.0);
  out[128 + l] = mfma444((l >> 2 & 3) == 2 ? 1.0 : 0.0, (double)l, 0.0);
  out[192 + l] = mfma444((l & 3) == 2 ? 1.0 : 0.0, (double)l, 0.0);
}

__global__ void chain_mfma(double* out, int n) {  // dependent: x -> sum -> fma -> sum ...
  double x = out[threadIdx.x];
  long long t0 = clock64();
  for (int i = 0; i < n; i++) {
#pragma unroll
    for (int k = 0; k < 8; k++) { x = mfma444(1.0, x, 0.0); x = fma(x, 0.25, 1e-3); }
  }
  long long t1 = clock64();
  out[threadIdx.x] = x;
  if (threadIdx.x == 0 && blockIdx.x == 0) out[256] = (double)(t1 - t0) / (8.0 * n);
}
__global__ void chain_dpp(double* out, int n) {
  double x = out[threadIdx.x];
  long long t0 = clock64();
  for (int i = 0; i < n; i++) {
#pragma unroll
    for (int k = 0; k < 8; k++) { x = qsum(x); x = fma(x, 0.25, 1e-3); }
  }
  long long t1 = clock64();
  out[threadIdx.x] = x;
  if (threadIdx.x == 0 && blockIdx.x == 0) out[257] = (double)(t1 - t0) / (8.0 * n);
}
__global__ void five_mfma(double* out, int n) {  // five independent sums per step (the J' df group), then a dependent use
  double a = out[threadIdx.x], b = a + 1, c = a + 2, d = a + 3, e = a + 4;
  long long t0 = clock64();
  for (int i = 0; i < n; i++) {
#pragma unroll
    for (int k = 0; k < 4; k++) {
      double s0 = mfma444(1.0, a, 0.0), s1 = mfma444(1.0, b, 0.0), s2 = mfma444(1.0, c, 0.0), s3 = mfma444(1.0, d, 0.0), s4 = mfma444(1.0, e, 0.0);
      a = fma(s0, 0.2, 1e-3); b = fma(s1, 0.2, 1e-3); c = fma(s2, 0.2, 1e-3); d = fma(s3, 0.2, 1e-3); e = fma(s4, 0.2, 1e-3);
    }
  }
  long long t1 = clock64();
  out[threadIdx.x] = a + b + c + d + e;
  if (threadIdx.x == 0 && blockIdx.x == 0) out[258] = (double)(t1 - t0) / (4.0 * n);
}
__global__ void five_dpp(double* out, int n) {
  double a = out[threadIdx.x], b = a + 1, c = a + 2, d = a + 3, e = a + 4;
  long long t0 = clock64();
  for (int i = 0; i < n; i++) {
#pragma unroll
    for (int k = 0; k < 4; k++) {
      double s0 = qsum(a), s1 = qsum(b), s2 = qsum(c), s3 = qsum(d), s4 = qsum(e);
      a = fma(s0, 0.2, 1e-3); b = fma(s1, 0.2, 1e-3); c = fma(s2, 0.2, 1e-3); d = fma(s3, 0.2, 1e-3); e = fma(s4, 0.2, 1e-3);
    }
  }
  long long t1 = clock64();
  out[threadIdx.x] = a + b + c + d + e;
  if (threadIdx.x == 0 && blockIdx.x == 0) out[259] = (double)(t1 - t0) / (4.0 * n);
}

int main() {
  double* d;
  hipMalloc(&d, 4096); hipMemset(d, 0, 4096);
  layout<<<1, 64>>>(d);
  std::vector<double> h(512);
  hipMemcpy(h.data(), d, 4096, hipMemcpyDeviceToHost);
  const char* names[4] = {"D for A=1, B=lane", "D for A=lane, B=1", "D for A=(lane/4==2), B=lane", "D for A=(lane%4==2), B=lane"};
  for (int p = 0; p < 4; p++) {
    printf("%s\n", names[p]);
    for (int l = 0; l < 32; l++) printf("%s%5.0f", l % 16 == 0 ? "  " : "", h[64 * p + l]), (l % 16 == 15 ? printf("\n") : 0);
  }
  hipMemset(d, 0, 4096);
  chain_mfma<<<1, 64>>>(d, 1000); chain_dpp<<<1, 64>>>(d, 1000); five_mfma<<<1, 64>>>(d, 1000); five_dpp<<<1, 64>>>(d, 1000);
  hipMemcpy(h.data(), d, 4096, hipMemcpyDeviceToHost);
  printf("dependent (sum, fma) pair: mfma %.1f ticks, dpp quad sum %.1f ticks\n", h[256], h[257]);
  printf("five independent sums + five fma: mfma %.1f ticks, dpp %.1f ticks\n", h[258], h[259]);
  return 0;
}
