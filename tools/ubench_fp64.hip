// Micro-benchmark (dev tool): issue cost of fp64 VALU ops for ONE wave per SIMD on gfx950.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/ubench tools/ubench_fp64.hip && /tmp/ubench
#include <hip/hip_runtime.h>
#include <cstdio>
template <int CHAINS>
__global__ void k_fma(double* out, long long* cyc, int iters, double a, double b) {
  double x[CHAINS];
  for (int c = 0; c < CHAINS; ++c) x[c] = threadIdx.x * 1e-3 + c;
  long long t0 = clock64();
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int r = 0; r < 8; ++r)
#pragma unroll
      for (int c = 0; c < CHAINS; ++c) x[c] = __builtin_fma(x[c], a, b);
  }
  long long t1 = clock64();
  double s = 0;
  for (int c = 0; c < CHAINS; ++c) s += x[c];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}
template <int CHAINS>
__global__ void k_fma32(float* out, long long* cyc, int iters, float a, float b) {
  float x[CHAINS];
  for (int c = 0; c < CHAINS; ++c) x[c] = threadIdx.x * 1e-3f + c;
  long long t0 = clock64();
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int r = 0; r < 8; ++r)
#pragma unroll
      for (int c = 0; c < CHAINS; ++c) x[c] = __builtin_fmaf(x[c], a, b);
  }
  long long t1 = clock64();
  float s = 0;
  for (int c = 0; c < CHAINS; ++c) s += x[c];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}
// same chain, but only the first ACTIVE lanes of the wave execute it (does the SIMD skip idle 16-lane quarters?)
template <int CHAINS, int ACTIVE>
__global__ void k_fma_part(double* out, long long* cyc, int iters, double a, double b) {
  double x[CHAINS];
  for (int c = 0; c < CHAINS; ++c) x[c] = threadIdx.x * 1e-3 + c;
  long long t0 = clock64();
  if (threadIdx.x < ACTIVE) {
    for (int i = 0; i < iters; ++i) {
#pragma unroll
      for (int r = 0; r < 8; ++r)
#pragma unroll
        for (int c = 0; c < CHAINS; ++c) x[c] = __builtin_fma(x[c], a, b);
    }
  }
  long long t1 = clock64();
  double s = 0;
  for (int c = 0; c < CHAINS; ++c) s += x[c];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}
template <class K, class T>
void run(const char* name, K kern, int chains, int threads, T a, T b) {
  T* out; long long* cyc; long long h;
  hipMalloc(&out, 1 << 20); hipMalloc(&cyc, 8);
  const int iters = 2000;
  kern<<<1, threads>>>(out, cyc, iters, a, b);
  kern<<<1, threads>>>(out, cyc, iters, a, b);
  hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
  printf("%-28s chains=%d threads=%4d : %.2f cycles per wave-instruction\n", name, chains, threads, double(h) / (double(iters) * 8 * chains));
  hipFree(out); hipFree(cyc);
}
int main() {
  run("v_fma_f64 16 of 64 lanes", k_fma_part<1, 16>, 1, 64, 1.0000001, 1e-9);
  run("v_fma_f64 16 of 64 lanes", k_fma_part<4, 16>, 4, 64, 1.0000001, 1e-9);
  run("v_fma_f64 32 of 64 lanes", k_fma_part<4, 32>, 4, 64, 1.0000001, 1e-9);
  run("v_fma_f64 16 lanes x 4 waves", k_fma_part<4, 16>, 4, 256, 1.0000001, 1e-9);
  for (int threads : {64, 256, 512, 1024}) {
    run("v_fma_f64", k_fma<1>, 1, threads, 1.0000001, 1e-9);
    run("v_fma_f64", k_fma<2>, 2, threads, 1.0000001, 1e-9);
    run("v_fma_f64", k_fma<4>, 4, threads, 1.0000001, 1e-9);
    run("v_fma_f64", k_fma<8>, 8, threads, 1.0000001, 1e-9);
    run("v_fma_f32", k_fma32<1>, 1, threads, 1.0000001f, 1e-9f);
    run("v_fma_f32", k_fma32<4>, 4, threads, 1.0000001f, 1e-9f);
  }
  return 0;
}
