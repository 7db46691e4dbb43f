// Micro-benchmark (dev tool): LDS pipe occupancy of reads/writes on gfx950 as a function of the number of waves issuing
// them and of the address pattern (the numbers behind WgCtx::sweep_costate_par's lane layout).
//   hipcc --offload-arch=gfx950 -O3 -o _diag/ubench_lds tools/ubench_lds.hip && ./_diag/ubench_lds
// One workgroup of 256 threads; `nw` waves run the loop, the others wait at the barrier.  Patterns:
//   0: 64 lanes, distinct 16-byte words (lane*16)                 1: lanes 0..15 only (exec mask), distinct
//   2: 64 lanes, 16 distinct words each read by 4 lanes (broadcast; (lane&15)*16)
//   3: 64 lanes, 4 groups of 16 at a stride of 9216 bytes (same banks across groups)
//   4: as 3 with the groups skewed by 64 bytes each (group g at +g*9216 + g*64)
#include <hip/hip_runtime.h>
#include <cstdio>
#define REPT(n, body) ".rept " #n "\n" body "\n.endr\n"

template <int OP>
__global__ __launch_bounds__(256) void k(long long* cyc, float* sink, int nw, int pat, int iters) {
  extern __shared__ unsigned char smem[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int i = threadIdx.x; i < 16384; i += 256) reinterpret_cast<float*>(smem)[i] = i;
  __syncthreads();
  int addr = wave * 1024;
  if (pat == 0) addr += lane * 16;
  if (pat == 1) addr += lane * 16;
  if (pat == 2) addr += (lane & 15) * 16;
  if (pat == 3) addr += (lane & 15) * 16 + (lane >> 4) * 9216;
  if (pat == 4) addr += (lane & 15) * 16 + (lane >> 4) * (9216 + 64);
  typedef float v4 __attribute__((ext_vector_type(4)));
  typedef float v2 __attribute__((ext_vector_type(2)));
  v4 a = {0, 0, 0, 0}, b = a, c = a, d = a;
  v2 e = {1, 2}, f = e, g = e, h = e;
  long long t0 = 0, t1 = 0;
  if (wave < nw && (pat != 1 || lane < 16)) {
    t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
      if (OP == 0)
        asm volatile(REPT(8, "ds_read_b128 %0, %4\n ds_read_b128 %1, %4 offset:256\n ds_read_b128 %2, %4 offset:512\n ds_read_b128 %3, %4 offset:768\n")
                     "s_waitcnt lgkmcnt(0)" : "=v"(a), "=v"(b), "=v"(c), "=v"(d) : "v"(addr) : "memory");
      if (OP == 1)
        asm volatile(REPT(32, "ds_write_b64 %1, %0\n") "s_waitcnt lgkmcnt(0)" : : "v"(e), "v"(addr) : "memory");
      if (OP == 2)
        asm volatile(REPT(8, "ds_read_b64 %0, %4\n ds_read_b64 %1, %4 offset:256\n ds_read_b64 %2, %4 offset:512\n ds_read_b64 %3, %4 offset:768\n")
                     "s_waitcnt lgkmcnt(0)" : "=v"(e), "=v"(f), "=v"(g), "=v"(h) : "v"(addr) : "memory");
      if (OP == 3)
        asm volatile(REPT(32, "ds_write_b128 %1, %0\n") "s_waitcnt lgkmcnt(0)" : : "v"(a), "v"(addr) : "memory");
    }
    t1 = __builtin_readcyclecounter();
  }
  __syncthreads();
  if (lane == 0) cyc[wave] = t1 - t0;
  sink[threadIdx.x] = a.x + b.y + c.z + d.w + e.x + f.y + g.x + h.y;
}

int main() {
  long long* cyc;
  float* sink;
  hipMalloc(&cyc, 64);
  hipMalloc(&sink, 4096);
  const char* ops[] = {"ds_read_b128", "ds_write_b64", "ds_read_b64", "ds_write_b128"};
  auto run = [&](auto kern, int op) {
    hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
    for (int pat = 0; pat < 5; ++pat)
      for (int nw = 1; nw <= 4; nw += 3) {
        const int iters = 200;
        kern<<<1, 256, 65536>>>(cyc, sink, nw, pat, iters);
        kern<<<1, 256, 65536>>>(cyc, sink, nw, pat, iters);
        long long h[4];
        hipMemcpy(h, cyc, 32, hipMemcpyDeviceToHost);
        printf("%-14s pattern %d  %d wave(s): %.1f cycles per instruction per wave  (%.1f per instruction on the CU)\n", ops[op], pat, nw,
               double(h[0]) / (iters * 32), double(h[0]) / (iters * 32) / nw);
      }
  };
  run(k<0>, 0);
  run(k<1>, 1);
  run(k<2>, 2);
  run(k<3>, 3);
  return 0;
}
