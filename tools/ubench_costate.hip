// Micro-benchmark (dev tool): the stage loop of the costate sweep as wave 0 of the chunk-parallel form runs it — per
// stage 3 x ds_read_b128 + ds_read_b64 (operands, software-pipelined two stages ahead), ~12 dependent-ish fp64 FMAs,
// ds_write_b64 — with 16 or 64 active lanes, alone or next to three waves running the transfer-matrix loop
// (2 x ds_read_b128, 9 FMAs, ds_write_b64 per stage).
//   hipcc --offload-arch=gfx950 -O3 -o _diag/ubench_costate tools/ubench_costate.hip && ./_diag/ubench_costate
#include <hip/hip_runtime.h>
#include <cstdio>

struct Pair { double a, b; };

template <bool HOM>
__device__ __forceinline__ void run(const double* R, double* out, int lane_addr, int stages, double dtau, double* sink) {
  double l0 = 1, l1 = 2, l2 = 3, l3 = 4;
  const Pair* q = reinterpret_cast<const Pair*>(R) + lane_addr;
  double* o = out + lane_addr;
  constexpr int STEP = 48;  // pairs per stage
  Pair A0 = q[0], A1 = q[16], A2 = HOM ? A0 : q[32];
  double ao = HOM ? 0.0 : o[0];
  Pair B0 = q[STEP], B1 = q[STEP + 16], B2 = HOM ? B0 : q[STEP + 32];
  double bo = HOM ? 0.0 : o[3];
  for (int s = 0; s < stages; ++s) {
    q += STEP, o += 3;
    Pair C0 = q[STEP], C1 = q[STEP + 16], C2 = HOM ? C0 : q[STEP + 32];
    double co = HOM ? 0.0 : o[3];
    const double dF = l2 * 0.5 + A1.b * l3;
    const double n0 = HOM ? __builtin_fma(A0.a, l3, l0) : (l0 + A2.a) + A0.a * l3;
    const double n1 = HOM ? __builtin_fma(A0.b, l3, l1) : (l1 + A2.b) + A0.b * l3;
    const double n2 = __builtin_fma(A1.a, l3, __builtin_fma(dtau, l0, 0.99 * l2));
    const double n3 = __builtin_fma(dtau, l1, 0.98 * l3);
    l0 = n0, l1 = n1, l2 = n2, l3 = n3;
    o[-3] = HOM ? dF : ao + dF * 0.25;
    A0 = B0, A1 = B1, A2 = B2, ao = bo;
    B0 = C0, B1 = C1, B2 = C2, bo = co;
  }
  *sink = l0 + l1 + l2 + l3;
}

__global__ __launch_bounds__(256) void k(long long* cyc, double* sink, int mode, int lanes, int stages, double dtau) {
  extern __shared__ __align__(16) unsigned char smem[];
  double* R = reinterpret_cast<double*>(smem);           // 64 stages x 96 doubles
  double* out = R + 64 * 96;                             // rows
  for (int i = threadIdx.x; i < 64 * 96 + 4096; i += 256) R[i] = 1e-3 * (i & 127);
  __syncthreads();
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  long long t0 = 0, t1 = 0;
  if (wave == 0) {
    if (lane < lanes) {
      t0 = __builtin_readcyclecounter();
      for (int rep = 0; rep < 20; ++rep) run<false>(R, out, (lane & 15) + (lane >> 4) * 12 * 48, stages, dtau, sink + threadIdx.x);
      t1 = __builtin_readcyclecounter();
    }
  } else if (mode == 1) {
    for (int rep = 0; rep < 20; ++rep) run<true>(R, out + 1024 + wave * 64 * 16, (lane & 15), stages, dtau, sink + threadIdx.x);
  }
  __syncthreads();
  if (threadIdx.x == 0) cyc[0] = t1 - t0;
}

int main() {
  long long* cyc;
  double* sink;
  hipMalloc(&cyc, 64);
  hipMalloc(&sink, 4096);
  hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024);
  for (int mode = 0; mode < 2; ++mode)
    for (int lanes = 16; lanes <= 64; lanes += 48) {
      const int stages = 12;
      k<<<1, 256, 96 * 1024>>>(cyc, sink, mode, lanes, stages, 1e-3);
      k<<<1, 256, 96 * 1024>>>(cyc, sink, mode, lanes, stages, 1e-3);
      long long h;
      hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
      printf("wave 0 with %2d active lanes, other waves %s: %.1f cycles per stage\n", lanes, mode ? "run the transfer-matrix loop" : "idle",
             double(h) / (20.0 * stages));
    }
  return 0;
}
