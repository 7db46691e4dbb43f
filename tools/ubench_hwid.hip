// Micro-benchmark (dev tool): where do the waves of co-resident workgroups land, and does it matter which wave of a
// workgroup runs a serial (single-wave, issue-bound) phase?
//   hipcc --offload-arch=gfx950 -O3 -o _diag/ubench_hwid tools/ubench_hwid.hip && ./_diag/ubench_hwid [lds_bytes] [wgs]
// Each workgroup = 256 threads (4 waves) + `lds_bytes` of dynamic LDS.  One wave per workgroup (the "sweep" wave)
// executes a long dependent FMA chain, the other three wait at the barrier — the shape of the tick kernel's horizon
// sweeps.  Modes:
//   0: sweep wave = wave 0 everywhere
//   1: sweep wave = the wave that sits on SIMD (2 * slot), slot = this workgroup's ordinal on its CU (claimed in a
//      per-CU bit mask in global memory, released at exit)
// Prints the wave -> SIMD placement, how many workgroups shared a CU, and the kernel time per mode.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <map>
#include <vector>

struct Rec {
  unsigned hw_id, xcc, slot, sweep_wave;
  long long t0, t1;
};

__device__ __forceinline__ unsigned hw_id() {
  unsigned v;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(v));
  return v;
}
__device__ __forceinline__ unsigned xcc_id() {
  unsigned v;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v));
  return v & 0xf;
}

template <class T>
__global__ __launch_bounds__(256) void k(Rec* rec, unsigned* cu_mask, T* sink, int mode, int iters) {
  extern __shared__ unsigned char smem[];
  __shared__ int s_slot, s_simd[4];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const unsigned hw = hw_id(), xcc = xcc_id();
  const unsigned cu_key = (xcc << 8) | ((hw >> 8) & 0xff);  // xcc | se,sh,cu
  if (lane == 0) s_simd[wave] = (hw >> 4) & 3;
  if (threadIdx.x == 0) {
    int slot = 0;
    if (mode == 1) {
      unsigned old = atomicOr(&cu_mask[cu_key], 1u);
      if (old & 1u) {
        old = atomicOr(&cu_mask[cu_key], 2u);
        slot = (old & 2u) ? 2 : 1;  // 2 = no free slot (third workgroup on the CU)
      }
    }
    s_slot = slot;
  }
  reinterpret_cast<volatile unsigned*>(smem)[threadIdx.x] = hw;  // touch the dynamic LDS
  __syncthreads();
  int sweep = 0;
  if (mode == 1) {
    const int want = (s_slot & 1) * 2;
    for (int w = 0; w < 4; ++w)
      if (s_simd[w] == want) sweep = w;
  }
  const long long t0 = (long long)__builtin_amdgcn_s_memrealtime();
  T a = T(threadIdx.x) * T(1e-3) + T(1.0);
  if (wave == sweep) {
    const T b = T(0.999999), c = T(1e-7);
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int j = 0; j < 64; ++j) a = a * b + c;
    }
  }
  __syncthreads();
  const long long t1 = (long long)__builtin_amdgcn_s_memrealtime();
  if (lane == 0) {
    Rec r{hw, xcc, (unsigned)s_slot, (unsigned)sweep, t0, t1};
    rec[blockIdx.x * 4 + wave] = r;
  }
  if (wave == sweep) sink[blockIdx.x * 64 + lane] = a;
  if (threadIdx.x == 0 && mode == 1 && s_slot < 2) atomicAnd(&cu_mask[cu_key], ~(1u << s_slot));
}

template <class T>
void run(const char* name, int lds, int wgs, int iters) {
  Rec* rec;
  unsigned* mask;
  T* sink;
  hipMalloc(&rec, sizeof(Rec) * wgs * 4);
  hipMalloc(&mask, 4 * 4096);
  hipMalloc(&sink, sizeof(T) * wgs * 64);
  hipFuncSetAttribute(reinterpret_cast<const void*>(k<T>), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
  hipEvent_t e0, e1;
  hipEventCreate(&e0), hipEventCreate(&e1);
  for (int mode = 0; mode < 2; ++mode) {
    float best = 1e30f;
    for (int rep = 0; rep < 4; ++rep) {
      hipMemset(mask, 0, 4 * 4096);
      hipEventRecord(e0);
      k<T><<<wgs, 256, lds>>>(rec, mask, sink, mode, iters);
      hipEventRecord(e1);
      hipEventSynchronize(e1);
      float ms;
      hipEventElapsedTime(&ms, e0, e1);
      if (rep && ms < best) best = ms;
    }
    std::vector<Rec> h(wgs * 4);
    hipMemcpy(h.data(), rec, sizeof(Rec) * wgs * 4, hipMemcpyDeviceToHost);
    std::map<unsigned, std::vector<int>> per_cu;  // cu key -> workgroups
    int simd_hist[4][4] = {}, same = 0, pairs = 0, slot_hist[3] = {};
    for (int g = 0; g < wgs; ++g) {
      per_cu[(h[g * 4].xcc << 8) | ((h[g * 4].hw_id >> 8) & 0xff)].push_back(g);
      for (int w = 0; w < 4; ++w) simd_hist[w][(h[g * 4 + w].hw_id >> 4) & 3]++;
      slot_hist[h[g * 4].slot]++;
    }
    // co-resident pairs (time overlap) and whether their sweep waves share a SIMD
    int max_share = 0;
    for (auto& kv : per_cu) {
      auto& v = kv.second;
      for (size_t i = 0; i < v.size(); ++i)
        for (size_t j = i + 1; j < v.size(); ++j) {
          const Rec &a = h[v[i] * 4], &b = h[v[j] * 4];
          if (a.t0 < b.t1 && b.t0 < a.t1) {
            ++pairs;
            const int sa = (h[v[i] * 4 + a.sweep_wave].hw_id >> 4) & 3, sb = (h[v[j] * 4 + b.sweep_wave].hw_id >> 4) & 3;
            same += sa == sb;
          }
        }
      if ((int)v.size() > max_share) max_share = (int)v.size();
    }
    printf("%s lds=%d wgs=%d mode=%d: %.3f ms | CUs used %zu, max WGs per CU %d, overlapping pairs %d (sweep waves on the same SIMD: %d) | slots %d/%d/%d\n",
           name, lds, wgs, mode, best, per_cu.size(), max_share, pairs, same, slot_hist[0], slot_hist[1], slot_hist[2]);
    if (mode == 0) {
      printf("   wave -> SIMD histogram:");
      for (int w = 0; w < 4; ++w) printf("  w%d:[%d %d %d %d]", w, simd_hist[w][0], simd_hist[w][1], simd_hist[w][2], simd_hist[w][3]);
      printf("\n");
    }
  }
  hipFree(rec), hipFree(mask), hipFree(sink);
}

int main(int argc, char** argv) {
  const int lds = argc > 1 ? atoi(argv[1]) : 79 * 1024;
  const int wgs = argc > 2 ? atoi(argv[2]) : 512;
  const int iters = 2000;
  run<float>("f32", lds, wgs, iters);
  run<double>("f64", lds, wgs, iters);
  run<float>("f32 (one WG per CU)", 150 * 1024, wgs / 2, iters);
  run<float>("f32 (two rounds)", 150 * 1024, wgs, iters);
  return 0;
}
