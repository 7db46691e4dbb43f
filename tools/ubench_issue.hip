// Micro-benchmark (dev tool): per-instruction issue cost seen by ONE wave on gfx950, measured with s_memtime
// around loops of .rept-expanded inline assembly (64 instructions per taken branch, so the branch is amortised;
// the branch itself is measured separately).
//   hipcc --offload-arch=gfx950 -O3 -o _diag/ubench_issue tools/ubench_issue.hip && ./_diag/ubench_issue
#include <hip/hip_runtime.h>
#include <cstdio>

#define REPT(n, body) ".rept " #n "\n" body "\n.endr\n"

template <int TEST>
__global__ void k(double* out, long long* cyc, int iters, double b, double c) {
  double a = threadIdx.x * 1e-3 + 1.0, d = threadIdx.x * 2e-3 + 1.5, e = a + 3, f = d + 4;
  int i0 = threadIdx.x, i1 = threadIdx.x * 3;
  typedef int v4i __attribute__((ext_vector_type(4)));
  v4i i128 = {0, 0, 0, 0};
  __shared__ double lds[2048];
  lds[threadIdx.x] = a;
  __syncthreads();
  double* gp = out + 4096 + (threadIdx.x >> 2) * 2;
  int addr = (threadIdx.x >> 2) * 8, addr2 = threadIdx.x * 8, addr3 = threadIdx.x * 16;
  long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
    if (TEST == 0) asm volatile(REPT(64, "v_fma_f64 %0, %0, %1, %2") : "+v"(a) : "v"(b), "v"(c));
    if (TEST == 1)
      asm volatile(REPT(32, "v_fma_f64 %0, %0, %2, %3\n v_fma_f64 %1, %1, %2, %3") : "+v"(a), "+v"(d) : "v"(b), "v"(c));
    if (TEST == 2)
      asm volatile(REPT(16, "v_fma_f64 %0, %0, %4, %5\n v_fma_f64 %1, %1, %4, %5\n v_fma_f64 %2, %2, %4, %5\n v_fma_f64 %3, %3, %4, %5")
                   : "+v"(a), "+v"(d), "+v"(e), "+v"(f) : "v"(b), "v"(c));
    if (TEST == 3) asm volatile(REPT(64, "v_mul_f64 %0, %0, %1") : "+v"(a) : "v"(b));
    if (TEST == 4) asm volatile(REPT(64, "v_add_f64 %0, %0, %1") : "+v"(a) : "v"(c));
    if (TEST == 5) asm volatile(REPT(64, "v_rndne_f64 %0, %0") : "+v"(a));
    if (TEST == 6) asm volatile(REPT(32, "v_cvt_i32_f64 %1, %0\n v_cvt_f64_i32 %0, %1") : "+v"(a), "+v"(i0));
    if (TEST == 7) asm volatile(REPT(64, "v_cndmask_b32 %0, %0, %1, vcc") : "+v"(i0) : "v"(i1) : "vcc");
    if (TEST == 8)
      asm volatile(REPT(64, "v_mov_b32_dpp %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n s_nop 1") : "+v"(i0));
    if (TEST == 9)
      asm volatile(REPT(32, "v_mov_b32_dpp %0, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n"
                            "v_mov_b32_dpp %1, %0 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf")
                   : "+v"(i0), "+v"(i1));  // independent-ish pairs: no nop needed between different registers? (hazard: see note)
    if (TEST == 10) asm volatile(REPT(64, "v_add_u32 %0, %0, %1") : "+v"(i0) : "v"(i1));
    if (TEST == 11) asm volatile(REPT(64, "v_bitop3_b32 %0, %0, %1, %1 bitop3:0x6c") : "+v"(i0) : "v"(i1));
    if (TEST == 12) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(a) : "v"(b), "v"(c));  // 1 instr per taken branch
    if (TEST == 13) asm volatile(REPT(64, "ds_read_b64 %0, %1\n s_waitcnt lgkmcnt(0)") : "+v"(a) : "v"(addr) : "memory");
    if (TEST == 14) asm volatile(REPT(64, "ds_write_b64 %1, %0") : : "v"(a), "v"(addr) : "memory");
    if (TEST == 15) asm volatile(REPT(64, "ds_write2_b64 %1, %0, %0 offset1:16") : : "v"(a), "v"(addr) : "memory");
    if (TEST == 16) asm volatile(REPT(64, "v_fma_f64 %0, %0, %1, %2") : "+v"(a) : "s"(b), "v"(c));
    if (TEST == 17) asm volatile(REPT(32, "v_fma_f64 %0, %0, %2, %3\n v_cndmask_b32 %1, %1, %1, vcc") : "+v"(a), "+v"(i0) : "v"(b), "v"(c) : "vcc");
    if (TEST == 18) asm volatile(REPT(64, "v_cmp_nlt_f64 vcc, |%0|, %1") : : "v"(a), "v"(b) : "vcc");
    if (TEST == 19) asm volatile(REPT(64, "v_fmac_f64 %0, %1, %2") : "+v"(a) : "v"(b), "v"(c));
    if (TEST == 20) asm volatile(REPT(32, "v_fma_f64 %0, %0, %1, %2\n s_nop 0") : "+v"(a) : "v"(b), "v"(c));
    if (TEST == 21) asm volatile(REPT(64, "v_mov_b32 %0, %1") : "+v"(i0) : "v"(i1));
    if (TEST == 22) asm volatile(REPT(64, "v_mov_b64 %0, %1") : "+v"(a) : "v"(d));
    if (TEST == 23) asm volatile(REPT(64, "v_accvgpr_read_b32 %0, a0") : "+v"(i0));
    if (TEST == 24) asm volatile(REPT(64, "s_add_u32 s20, s20, 1") : : : "s20", "scc");
    if (TEST == 25) asm volatile(REPT(64, "v_readlane_b32 s20, %0, 3") : : "v"(i0) : "s20");
    // LDS stores with one active lane per quad (exec = 0x1111...) / 16 contiguous active lanes
    if (TEST == 26)
      asm volatile("s_mov_b64 s[20:21], exec\n s_mov_b32 s22, 0x11111111\n s_mov_b32 s23, 0x11111111\n s_mov_b64 exec, s[22:23]\n"
                   REPT(64, "ds_write_b64 %1, %0") "s_mov_b64 exec, s[20:21]" : : "v"(a), "v"(addr) : "memory", "s20", "s21", "s22", "s23");
    if (TEST == 27)
      asm volatile("s_mov_b64 s[20:21], exec\n s_mov_b32 s22, 0x11111111\n s_mov_b32 s23, 0x11111111\n s_mov_b64 exec, s[22:23]\n"
                   REPT(64, "ds_write2_b64 %1, %0, %0 offset1:16") "s_mov_b64 exec, s[20:21]" : : "v"(a), "v"(addr) : "memory", "s20", "s21", "s22", "s23");
    if (TEST == 28)
      asm volatile("s_mov_b64 s[20:21], exec\n s_mov_b64 exec, 0xffff\n"
                   REPT(64, "ds_write2_b64 %1, %0, %0 offset1:16") "s_mov_b64 exec, s[20:21]" : : "v"(a), "v"(addr) : "memory", "s20", "s21");
    if (TEST == 31) asm volatile(REPT(64, "ds_write_b64 %1, %0") : : "v"(a), "v"(addr2) : "memory");
    if (TEST == 32) asm volatile(REPT(64, "ds_write_b32 %1, %0") : : "v"(i0), "v"(addr2) : "memory");
    if (TEST == 33) asm volatile(REPT(64, "ds_write_b128 %1, %0") : : "v"(i128), "v"(addr3) : "memory");
    if (TEST == 34) asm volatile(REPT(64, "ds_write2_b64 %1, %0, %0 offset1:16") : : "v"(a), "v"(addr2) : "memory");
    if (TEST == 35) asm volatile(REPT(64, "ds_read_b128 %0, %1") : "=v"(i128) : "v"(addr3) : "memory");
    if (TEST == 36) asm volatile(REPT(32, "ds_write_b64 %1, %0\n v_fma_f64 %2, %2, %3, %4\n v_fma_f64 %2, %2, %3, %4\n v_fma_f64 %2, %2, %3, %4") : : "v"(a), "v"(addr2), "v"(d), "v"(b), "v"(c) : "memory");
    if (TEST == 37) asm volatile(REPT(64, "global_store_dwordx2 %1, %0, off") : : "v"(a), "v"(gp) : "memory");
    if (TEST == 38) asm volatile(REPT(64, "global_store_dwordx4 %1, %0, off") : : "v"(i128), "v"(gp) : "memory");
    if (TEST == 39) asm volatile(REPT(32, "global_store_dwordx2 %1, %0, off\n v_fma_f64 %2, %2, %3, %4\n v_fma_f64 %2, %2, %3, %4\n v_fma_f64 %2, %2, %3, %4") : : "v"(a), "v"(gp), "v"(d), "v"(b), "v"(c) : "memory");
    if (TEST == 40) asm volatile(REPT(64, "global_load_dwordx2 %0, %1, off") : "=v"(a) : "v"(gp) : "memory");
    // the shape of one state-sweep stage: ~52 VALU + the stage-table stores (write2 + 2 writes), clustered or spread
    if (TEST == 41)
      asm volatile("ds_write2_b64 %1, %0, %0 offset1:16\n ds_write_b64 %5, %0\n ds_write_b64 %5, %0 offset:2048\n"
                   REPT(52, "v_fma_f64 %2, %2, %3, %4") : : "v"(a), "v"(addr), "v"(d), "v"(b), "v"(c), "v"(addr2) : "memory");
    if (TEST == 42)
      asm volatile("ds_write2_b64 %1, %0, %0 offset1:16\n" REPT(17, "v_fma_f64 %2, %2, %3, %4") "ds_write_b64 %5, %0\n"
                   REPT(17, "v_fma_f64 %2, %2, %3, %4") "ds_write_b64 %5, %0 offset:2048\n" REPT(18, "v_fma_f64 %2, %2, %3, %4")
                   : : "v"(a), "v"(addr), "v"(d), "v"(b), "v"(c), "v"(addr2) : "memory");
    if (TEST == 43) asm volatile(REPT(55, "v_fma_f64 %0, %0, %1, %2") : "+v"(a) : "v"(b), "v"(c));
    if (TEST == 29) asm volatile(REPT(64, "ds_read2_b64 %0, %1 offset1:16") : "=v"(i128) : "v"(addr) : "memory");
    if (TEST == 30) asm volatile(REPT(64, "ds_read_b64 %0, %1") : "=v"(a) : "v"(addr) : "memory");
  }
  long long t1 = __builtin_readcyclecounter();
  if (TEST == 14 || TEST == 15) __syncthreads();
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  out[blockIdx.x * blockDim.x + threadIdx.x] = i128[0] + a + d + e + f + i0 + i1 + lds[(threadIdx.x * 7) & 1023];
  if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}

template <int TEST>
void run(const char* name, int per_iter) {
  double* out;
  long long *cyc, h;
  (void)hipMalloc(&out, 1 << 20);
  (void)hipMalloc(&cyc, 8);
  const int iters = 2000;
  k<TEST><<<1, 64>>>(out, cyc, iters, 1.0000001, 1e-9);
  k<TEST><<<1, 64>>>(out, cyc, iters, 1.0000001, 1e-9);
  (void)hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
  printf("%-58s %7.2f cycles/instr  (%d instr per taken branch)\n", name, double(h) / (double(iters) * per_iter), per_iter);
  fflush(stdout);
  (void)hipFree(out);
  (void)hipFree(cyc);
}

int main() {
  run<0>("v_fma_f64 dependent chain", 64);
  run<1>("v_fma_f64 2 chains", 64);
  run<2>("v_fma_f64 4 chains", 64);
  run<16>("v_fma_f64 dependent, one SGPR source", 64);
  run<19>("v_fmac_f64 dependent", 64);
  run<20>("v_fma_f64 dependent + s_nop 0 (per pair)", 64);
  run<3>("v_mul_f64 dependent", 64);
  run<4>("v_add_f64 dependent", 64);
  run<5>("v_rndne_f64 dependent", 64);
  run<6>("v_cvt_i32_f64 / v_cvt_f64_i32 dependent", 64);
  run<7>("v_cndmask_b32 dependent", 64);
  run<17>("v_fma_f64 + v_cndmask_b32 alternating", 64);
  run<18>("v_cmp_nlt_f64 -> vcc", 64);
  run<8>("v_mov_b32_dpp quad_perm dependent + s_nop 1", 64);
  run<9>("v_mov_b32_dpp ping-pong (no nop)", 64);
  run<10>("v_add_u32 dependent", 64);
  run<11>("v_bitop3_b32 dependent", 64);
  run<21>("v_mov_b32", 64);
  run<22>("v_mov_b64", 64);
  run<23>("v_accvgpr_read_b32", 64);
  run<24>("s_add_u32 dependent", 64);
  run<25>("v_readlane_b32", 64);
  run<12>("loop of 1 v_fma_f64 (taken branch cost)", 1);
  run<13>("ds_read_b64 + wait (LDS latency)", 64);
  run<14>("ds_write_b64 64 lanes, 4 lanes/address", 64);
  run<15>("ds_write2_b64 64 lanes, 4 lanes/address", 64);
  run<26>("ds_write_b64 one lane per quad (16 active)", 64);
  run<27>("ds_write2_b64 one lane per quad (16 active)", 64);
  run<28>("ds_write2_b64 lanes 0-15 active", 64);
  run<31>("ds_write_b64 lane-contiguous", 64);
  run<32>("ds_write_b32 lane-contiguous", 64);
  run<33>("ds_write_b128 lane-contiguous", 64);
  run<34>("ds_write2_b64 lane-contiguous", 64);
  run<35>("ds_read_b128 lane-contiguous (no wait)", 64);
  run<36>("ds_write_b64 + 3 v_fma_f64 (per 4 instr)", 128);
  run<37>("global_store_dwordx2, 4 lanes/address", 64);
  run<38>("global_store_dwordx4, 4 lanes/address", 64);
  run<39>("global_store_dwordx2 + 3 v_fma_f64 (per 4 instr)", 128);
  run<40>("global_load_dwordx2 back to back (L1/L2 hit)", 64);
  run<41>("stage shape: 3 LDS stores clustered + 52 v_fma_f64", 55);
  run<42>("stage shape: 3 LDS stores spread   + 52 v_fma_f64", 55);
  run<43>("stage shape: 55 v_fma_f64 only", 55);
  run<29>("ds_read2_b64 back to back (no wait)", 64);
  run<30>("ds_read_b64 back to back (no wait)", 64);
  return 0;
}
