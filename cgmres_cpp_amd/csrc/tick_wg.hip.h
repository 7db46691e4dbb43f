// Variants 2 and 3 ("wg", "wg-lean"): one workgroup owns IPW controller instances for a whole control tick.
//
// Mapping (gfx950, wave64):
//   * block = IPW*16 threads; thread (inst = tid/16, r = tid%16).  The 16 lanes of one DPP row own the
//     length-L vectors of one instance, lane r holding elements e = r + 16*m (m < MAXM) in registers.
//     Dot products / norms are lane partial sums + a 4-step DPP butterfly inside the row
//     (quad_perm, quad_perm, row_half_mirror, row_mirror): every lane of the row ends with the
//     bit-identical sum, so row-uniform branches (early exit, breakdown) never diverge inside a row.
//   * a horizon sweep (F_func, cgmres.hpp:113-162) has three phases:
//       1. state sweep  — serial in the stage index, on wave 0: four lanes (one DPP quad) per instance for the
//                         pendulum (PendulumDev::quad_stage), one lane per instance otherwise; x(s) and the
//                         reusable trig values go to the LDS stage table R;
//       2. coefficients — everything of the backward stage that does not involve the costate (dHdx and dHdu
//                         are affine in it, models.hip.h) for ALL (stage, instance) pairs at once; inside the
//                         Arnoldi loop waves 1..3 do this chunk by chunk BEHIND the state sweep (f_eval);
//       3. costate sweep — serial again, but only the short affine recurrence is left (pendulum: 12 fp64 ops
//                         per stage instead of 46), fused with the costate part of dH/du.
//     U, F(U,x+hf,t+h), the work vector, the stage table and ptau all sit in LDS (odd row pitches: conflict-free
//     for the sweep lanes); there is no HBM access inside the stage loops.  (Long vectors: WgParams::fh_hbm.)
//   * the Krylov basis V (IPW x (k_max+1) x L) does not fit in 160 KB of LDS at IPW = 16, it lives in
//     HBM/L2 as zero-padded, pair-interleaved rows of 16*MAXM scalars (16-byte accesses, no guards) and is
//     streamed once per Gram-Schmidt step through a register ring — the traffic SURVEY.md §8(d) prices.
//   * the serial phases run on ONE wave, which issues one instruction of any kind per ~4.4 cycles
//     (tools/ubench_issue.hip): the stage loops are written for instruction count.
//   * instances never exchange data: the only synchronisation is the workgroup barrier between phases.
//   * variant 3 ("wg-lean", template parameter LEAN): the same kernel on half the LDS — U in the row lanes' registers (or
//     re-read from its HBM row), F(U,x+hf,t+h) and ptau fetched from HBM/L2 a group of items ahead — so that two
//     workgroups share a CU and their serial phases run side by side on different SIMDs (WgLds).
//   * inside a chunk of stages the pendulum sweep advances its sin/cos values by rotation through the exact angle
//     increment instead of re-evaluating them (PendulumDev::quad_stage_rot), with per-chunk fallbacks; in fixed-k mode
//     (tol = 0) the Hessenberg column of an iteration is processed by an idle wave during the next sweep (gmres()).
//   * template parameter NWT = 1 (pendulum fp64, full plan; the library's choice for the headline batch): inside the
//     Arnoldi loop the three phases above give way to ROW-PARALLEL sweeps — every row of 16 lanes runs its instance's
//     state recurrence as Newton's method on the whole trajectory (four stages per lane, in-row DPP scans) and the
//     costate recurrence as three scans down the row, all in registers, with no workgroup barrier in the loop
//     (row_newton_sweep / row_costate); the serial state sweep remains in the preamble only.
//     NWT = 2 (a state equation affine in x: the semi-active damper): the same with plain scans — no Newton, no serial
//     sweep and no stage table anywhere in the tick (row_affine_sweep).
// Statement order inside each instance follows cgmres.hpp:78-175 / gmres.hpp:28-112; what differs from the
// reference is the association order of sums (16 partial sums + butterfly; affine regrouping of the costate step).
#pragma once
#include <hip/hip_runtime.h>

#include <cfloat>
#include <type_traits>

#include "wave_scan.hip.h"
#include "dpp.hip.h"
#include "models.hip.h"
#include "tick_lane.hip.h"  // sqrt_t / abs_t / FOut

namespace cgm {

constexpr int CGM_FUSE_MAX = CGMRES_HIP_TICKS_PER_LAUNCH;
static_assert(CGM_FUSE_MAX == 10, "WgParams::dtau_tab");

enum WgMode { WG_TICK = 0, WG_HOOK_F = 1, WG_HOOK_PREPARE = 2, WG_HOOK_AX = 3, WG_HOOK_GMRES = 4 };

template <class T>
struct WgParams {
  int B, dv, kmax, L, Lp, Lg, Lv, Pp, Hp, fh_hbm, lds_bytes;  // fh_hbm: F(U,x+hf,t+h) is kept in HBM only (P.Fh), see WgLds
  int wave_dbg;   // wave mapping (tick_wave.hip.h): 1 = Newton with fresh sin/cos, 2 = serial state sweep in every mat-vec
  int cs_chunks;  // chunks of the two-pass costate sweep (WgCtx::sweep_costate_2pass): 3 or 4, what the LDS budget allows
  // row-parallel Newton kernel (WgCtx::NWT = 1): byte offsets, from the start of the workgroup's LDS, of the arrays that hold
  // the tick's base trajectory during the Arnoldi loop (ctx_wg.hip.h places them: in the stage table and in the scratch of
  // the chunk-parallel costate sweep, both idle then, and behind everything else where those do not suffice)
  int base_off[8];
   // Lp/Pp/Hp: odd LDS row pitches (Hp: the COMPACT Hessenberg, column k = rows 0..k at offset k(k+1)/2; h(k+1,k) of the
   // column in progress lives in WgLds::hsub);
   // Lg: global row pitch (multiple of 16); Lv = 16*MAXM: pitch of the Krylov rows (pads kept zero, no guards)
  T h, dt, tol, inv_h, one_m_zh, dtau_h, dtau_0;
  // closed loop on the device: up to CGM_FUSE_MAX consecutive ticks per launch, the controller state (U in LDS, dUdt
  // in registers, x in LDS) carried from tick to tick without going through HBM; dtau_tab[2*k], [2*k+1] = the two
  // horizon steps of tick k (cgmres.hpp:88,91), computed by the host exactly as for single launches
  int n_ticks;
  T dtau_tab[2 * 10];
  // time-varying reference (cgmres_hip_closed_loop_device_ptau): ptau of tick k of this launch =
  // ptau_seq[k*pseq_tick + b*pseq_inst + ...]; null = the handle's ptau for every tick
  const T* ptau_seq;
  size_t pseq_tick;
  int pseq_inst;
  // instance-major HBM state
  T *U, *dUdt, *Fh, *V, *xdxh, *ptau;  // [B][Lg], [B][Lg], [B][Lg], [B][kmax+1][Lv], [B][NX], [B][NP*(dv+1)]
  T* kry;                              // [B][KS]: H (k1*k1 col-major) | rho (k1) | g (3*kmax)
  T* scr;                              // [workgroups][2][(dv+TAB_PAD)*NSTG*IPW]: parked stage tables of the preamble sweeps
  T* pT;                               // lean plan: [workgroups][(dv+1)*NP][IPW] parameter horizon, transposed
  T* park;                             // [workgroups*IPW][Lv]: the solution vector during the Arnoldi loop (MAXM > 10 only)
  int *n_ax, *reason;
  // Placement: slot q = workgroup*IPW + row of the launch holds instance perm[q] (null = identity).  Instances never
  // exchange data; the closed loop of an oversubscribed batch bins instances by the Arnoldi count of their last tick so
  // that the rows of a workgroup leave the loop together (gmres.hpp:93-95 makes the count data-dependent;
  // util_kernels.hip.h: bin_by_count_kernel, a stable sort: the same counts give the same placement).  Where an
  // instance sits changes its bits in ONE case: the pendulum's quad sweep picks the form of its trig update (rotation /
  // fresh evaluation, sweep_state) per WAVE, i.e. for the 16 instances of a workgroup together, and the two forms round
  // differently (~1e-16) — in fast motion an instance's rounding depends on its workgroup mates.  Everything else of a
  // tick is independent of the placement bit for bit.
  const int* perm;
  const T* x_in;  // [B][NX]
  T* u_out;       // [B][NU]
  T* x_next;      // [B][NX] or null
  // hooks
  int mode;
  const T *hook_in0, *hook_in1;  // instance-major [B][L]
  T* hook_out;
  T hook_dtau;
};

#ifdef CGM_DEBUG_LDS
// Diagnostic build only: set by the costate sweep when one of its LDS addresses falls outside the workgroup's allocation
// (0x10000 | what << 12 | thread); read back through cgmres_hip_plugin_debug_lds_oob (user_model.hip.h)
__device__ int g_cgm_lds_oob;
#endif
// Diagnostic build only: accumulate shader-clock deltas per phase (thread 0 of block 0).  Compiles to nothing
// in the product build.
#ifdef CGM_STAMPS
// Diagnostic build only (tools/phase_stamps.py): per-phase shader-clock totals of block 0 in a device-global buffer
// (kept out of WgParams on purpose: any use of the kernel-argument struct through a pointer made hipcc spill the whole
// struct to scratch and distorted the very thing being measured).
__device__ long long g_cgm_stamps[64];
// The per-phase totals are accumulated in LDS (static, outside the dynamic carve-up) and added to the global buffer
// once at the end of the kernel: a global atomic per stamp queues behind the basis-row loads in flight and made the
// stamps after them look ~1.5 k cycles long.
__device__ __forceinline__ long long* cgm_stamp_lds() {
  __shared__ long long acc[66];  // [0..31] cycles, [32..63] visits, [64] last stamp
  return acc;
}
__device__ __forceinline__ void cgm_stamp(int id) {
  if (threadIdx.x == 0 && blockIdx.x == 0) {
    long long* acc = cgm_stamp_lds();
    const long long now = (long long)__builtin_amdgcn_s_memtime();
    if (id >= 0) {
      acc[id] += now - acc[64];
      acc[32 + id] += 1;
    } else {
      for (int i = 0; i < 64; ++i) acc[i] = 0;
      g_cgm_stamps[30] = (long long)__builtin_amdgcn_s_memrealtime();
      g_cgm_stamps[31] = now;
    }
    acc[64] = now;
  }
}
__device__ __forceinline__ void cgm_stamp_flush() {
  if (threadIdx.x == 0 && blockIdx.x == 0) {
    long long* acc = cgm_stamp_lds();
    for (int i = 0; i < 28; ++i) g_cgm_stamps[i] += acc[i], g_cgm_stamps[32 + i] += acc[32 + i];
    g_cgm_stamps[28] += (long long)__builtin_amdgcn_s_memrealtime() - g_cgm_stamps[30];
    g_cgm_stamps[29] += (long long)__builtin_amdgcn_s_memtime() - g_cgm_stamps[31];
  }
}
#define CGM_STAMP(ctx, id) cgm_stamp(id)
#else
#define CGM_STAMP(ctx, id) ((void)0)
#endif

// ---- LDS carve-up -------------------------------------------------------------------------------
// Three plans (WgParams::fh_hbm selects between the first two at run time, LEAN is a kernel template parameter):
//   full    U, F(U,x+hf,t+h) and the work vector W as rows, the stage table, ptau, the small Krylov arrays
//   fh_hbm  the row array of F(U,x+hf,t+h) is left out — the coefficient phase, its only reader after the preamble,
//           takes it from HBM (P.Fh).  Used when that is what lets 16 instances fit (MSD at N = 50: L = 300).
//   lean    only W, the stage table and the small arrays: U lives in the row lanes' REGISTERS (and is published into W
//           for the two unperturbed sweeps of the preamble), F(U,x+hf,t+h) and ptau are read from HBM/L2 by the
//           coefficient phase.  Half the footprint: TWO workgroups share a CU (<= 79.5 KB each), and the hardware
//           places their sweep waves on different SIMDs (tools/ubench_hwid.hip) — the serial phases of 32 instances
//           per CU overlap.  Chosen when a batch needs more workgroups than the GPU has CUs, or when controllers of
//           two models share the GPU (multiple_controller).
enum WgPlan { PLAN_FULL = 0, PLAN_FH_HBM = 1, PLAN_LEAN = 2 };
template <class M, class T, int IPW, int TABX = 0>  // TABX: spare scalars behind every stage of the stage table (WgCtx::NWT)
struct WgLds {
  // stage table: NSTG values per (stage, instance), layout [stage][slot][IPW], dv + TAB_PAD stages
  //   after phase 1: slots 0..NSLOT-1 = x(s), trig(s) (model's slot map);  after phase 2: slots 0..NBW-1 = costate coefficients
  static constexpr int NSTG = M::NSLOT > M::NBW ? M::NSLOT : M::NBW;
  static constexpr int TAB_PAD = M::TAB_PAD;
  T *U, *Fh, *W, *R, *p, *H, *rho, *g, *xs, *xh, *xT, *u0;  // xT: terminal states of the state sweeps in flight
  T* hsub;  // h(k+1,k) of the column in progress, per instance (the compact Hessenberg keeps rows 0..k of column k only)
  T* scan;  // scratch of the chunk-parallel costate sweep (WgCtx::sweep_costate_par), full plans only
  unsigned char* lds0;  // start of the workgroup's LDS (WgParams::base_off)
  int *flag, *reason, *nax, *ksolve;
  int* binst;  // global instance of every row of this workgroup (WgParams::perm applied)
  static __host__ __device__ size_t tab_count(int dv) {
    if constexpr (TABX != 0) return size_t(dv + TAB_PAD) * (NSTG * IPW + TABX);
    return size_t(dv + TAB_PAD) * NSTG * IPW;
  }
  static __host__ __device__ size_t count_T(int dv, int kmax, int Lp, int Pp, int Hp, int plan = PLAN_FULL) {
    const int k1 = kmax + 1;
    const int rows = plan == PLAN_FULL ? 3 : (plan == PLAN_FH_HBM ? 2 : 1);
    return size_t(rows) * IPW * Lp + tab_count(dv) + (plan == PLAN_LEAN ? 0 : size_t(IPW) * Pp) + size_t(IPW) * Hp +
           size_t(IPW) * k1 + size_t(IPW) * 3 * kmax + size_t(IPW) + size_t(plan == PLAN_LEAN ? 4 : 5) * M::NX * IPW +
           (plan == PLAN_LEAN ? size_t(M::NU) * IPW : 0);
  }
  // chunk-parallel costate sweep (WgCtx::sweep_costate_par): dF of the homogeneous lanes for the 3*(dv/4) stages of
  // chunks 1..3, then four boundary records per instance (k = 0..3): [NX*NX transfer matrix, row-major][NX vector]
  // [pad], read with 16-byte loads — SCAN_REC doubles apart so that the 16 instances of a read fall into distinct banks
  // (22 scalars = 44 banks for NX = 4: i*44 mod 64 are 16 distinct multiples of 4)
  static constexpr int SCAN_REC = M::NX * M::NX + M::NX + 2;
  static __host__ __device__ size_t scan_count(int dv) {
    return size_t(3 * (dv >> 2)) * M::NUL * M::NX * IPW + size_t(4) * IPW * SCAN_REC;
  }
  // two-pass costate sweep (WgCtx::sweep_costate_2pass), `chunks` of them: the end value of chunk 0 per instance, then one
  // boundary record per instance for each of the chunks 1 .. chunks-2
  static __host__ __device__ size_t scan2_count(int chunks) {
    return size_t(IPW) * M::NX + size_t(chunks - 2) * IPW * SCAN_REC;
  }
  // scan_T: scalars of costate-sweep scratch behind the small arrays (scan_count(dv), scan2_count(chunks) or 0)
  static __host__ __device__ size_t bytes(int dv, int kmax, int Lp, int Pp, int Hp, int plan = PLAN_FULL, size_t scan_T = 0) {
    return count_T(dv, kmax, Lp, Pp, Hp, plan) * sizeof(T) + 5 * IPW * sizeof(int) + 16 + (scan_T ? scan_T * sizeof(T) + 16 : 0);
  }
  __device__ __forceinline__ WgLds(unsigned char* base, const WgParams<T>& P, int plan) {
    T* q = reinterpret_cast<T*>(base);
    lds0 = base;
    const int k1 = P.kmax + 1;
    const bool lean = plan == PLAN_LEAN;
    // The costate sweep's look-ahead reads up to three stages BELOW the start of its operand arrays (values never used):
    // below the stage table that is a row array, below the first row of `out` it is whatever precedes the row arrays.
    // In the full plans U / Fh precede W; in the lean plan W would be the first array of the allocation, so the small
    // per-instance state arrays are placed in front of it (an access below the allocation is an aperture violation).
    if (lean) {
      xs = q, q += M::NX * IPW;
      xh = q, q += M::NX * IPW;
      xT = q, q += 2 * M::NX * IPW;  // lean: two concurrent preamble sweeps
      u0 = q, q += M::NU * IPW;      // lean: the control of the current tick for the plant step
    }
    U = q, q += lean ? 0 : IPW * P.Lp;                 // lean: U aliases W (never dereferenced as U)
    Fh = q, q += plan == PLAN_FULL ? IPW * P.Lp : 0;  // otherwise Fh aliases W (the preamble moves the result to HBM)
    W = q, q += IPW * P.Lp;
    R = q, q += tab_count(P.dv);
    p = q, q += lean ? 0 : IPW * P.Pp;
    H = q, q += IPW * P.Hp;
    rho = q, q += IPW * k1;
    g = q, q += IPW * 3 * P.kmax;
    hsub = q, q += IPW;
    if (!lean) {
      xs = q, q += M::NX * IPW;
      xh = q, q += M::NX * IPW;
      xT = q, q += 3 * M::NX * IPW;  // full plans: three concurrent preamble sweeps
      u0 = q;
    }
    int* z = reinterpret_cast<int*>(q);
    flag = z, reason = z + IPW, nax = z + 2 * IPW, ksolve = z + 3 * IPW, binst = z + 4 * IPW;
    // 16-byte aligned (records are read in pairs) — as an OFFSET from the base: a pointer-integer-pointer round trip
    // loses the LDS address space and every access through `scan` becomes a flat load behind s_waitcnt vmcnt(0)
    const unsigned scan_off = unsigned(reinterpret_cast<unsigned char*>(z + 5 * IPW) - base);
    scan = reinterpret_cast<T*>(base + ((scan_off + 15u) & ~15u));
  }
};

// Per-thread view of one workgroup's job.
// PAR: form of the costate sweep — 0 serial (one lane per instance walks all stages), 1 chunk-parallel with per-stage
// scratch in LDS (sweep_costate_par: full plans with short vectors), 2 chunk-parallel in two passes, boundary records only
// (sweep_costate_2pass: every plan).
// NWT = 1: the state sweeps of the Arnoldi loop run as Newton iterations on the whole trajectory, row-parallel on all
// four waves (row_newton_sweep) instead of the serial quad sweep on wave 0; the stage table then has NWT_TABX spare
// scalars per stage so that the 16 lanes of a row, which own four consecutive stages each, store into distinct banks.
constexpr int NWT_TABX = 1;
template <class M, class T, int IPW, int MAXM, bool LEAN = false, int PAR = 0, int NWT = 0>
struct WgCtx {
  using Lds = WgLds<M, T, IPW, NWT ? NWT_TABX : 0>;
  static constexpr int NSTG = Lds::NSTG;
  // offset of (stage s, slot) in a stage table
  static __device__ __forceinline__ int tab_off(int s, int slot = 0) {
    if constexpr (NWT != 0) return s * TAB_STEP + slot * IPW;
    return (s * NSTG + slot) * IPW;
  }
  static constexpr int TAB_STEP = NSTG * IPW + (NWT ? NWT_TABX : 0);  // scalars from one stage to the next
  const WgParams<T>& P;
  Lds S;
  // lean plan: this row's U in the row layout for the whole launch — in registers, unless a row is 160 bytes per lane or
  // more (fp64 with MAXM = 20): those kernels are register-starved at 256 registers per wave and re-read their U row
  // from HBM/L2 where they need it (once per Arnoldi iteration, ~1/30 of the iteration's traffic)
  static constexpr bool U_IN_REGS = LEAN && sizeof(T) * MAXM < 160;
  static constexpr bool VK_IN_REGS = !LEAN || U_IN_REGS;  // v_k kept in registers between its creation and the next MGS
  T ureg[U_IN_REGS ? MAXM : 1];
  T* pTw;                   // lean plan: this workgroup's transposed parameter horizon [(stage*NP + j)*IPW + i]
  int tid, inst, r, b;  // b = global instance of this thread's row
  bool valid;           // row has a real instance
  bool sweep_lane;      // this thread runs the serial sweeps (for instance `tid`)
  T dtau_h, dtau_0;     // horizon steps of the current tick: F(., t+h) and F(., t)
  int bi;               // global instance of the sweep lane
  // S.flag of the instances this thread works on in the sweeps of the current Arnoldi iteration — instance tid mod IPW
  // (items of the coefficient and costate phases, the lane-per-instance state sweep) and instance (tid mod 64) / 4 (quad
  // state sweep) — read once per iteration with the loop-top flag reads (gmres()) instead of one LDS round trip at the
  // start of every phase; only meaningful where `only_active` is passed (the Arnoldi loop)
  int act_i = 1, act_q = 1;
  int rot_level = 0, rot_hold = 0;  // quad sweep: form of the trig update the sweeps of this wave currently take (sweep_state)
  typename M::template MathFor<LEAN> mc;  // per-thread math context (pinned sin/cos constants); A/B: keeping it live for the whole
                        // kernel is 13 us/tick FASTER than re-creating it inside every sweep
  __device__ __forceinline__ WgCtx(const WgParams<T>& P_, unsigned char* smem)
      : P(P_), S(smem, P_, LEAN ? PLAN_LEAN : (MAXM > 10 && P_.fh_hbm ? PLAN_FH_HBM : PLAN_FULL)), tid(threadIdx.x),
        inst(threadIdx.x >> 4), r(threadIdx.x & 15) {
    pTw = LEAN ? P.pT + size_t(blockIdx.x) * (P.dv + 1) * (M::NP > 0 ? M::NP : 1) * IPW : nullptr;
    const int q = blockIdx.x * IPW + inst, qs = blockIdx.x * IPW + tid;  // slots of this row / of the sweep lane
    valid = q < P.B;
    sweep_lane = tid < IPW && qs < P.B;
    b = (valid && P.perm) ? P.perm[q] : q;
    bi = (sweep_lane && P.perm) ? P.perm[qs] : qs;
    dtau_h = P.dtau_h, dtau_0 = P.dtau_0;
    mc.init();
    CGM_STAMP(*this, -1);
  }
  __device__ __forceinline__ int elem(int m) const { return r + 16 * m; }

  // rows: HBM [B][Lg] -> LDS row / registers
  __device__ __forceinline__ void load_row_to_lds(T* lds, const T* g) const {
    if (!valid) return;
    const T* src = g + size_t(b) * P.Lg;
    T* row = lds + inst * P.Lp;
    each_elem([&](int m, auto full) {
      if (decltype(full)::value || elem(m) < P.L) row[elem(m)] = src[elem(m)];
    });
  }
  __device__ __forceinline__ void load_row_to_reg(T* reg, const T* g, size_t pitch) const {
    const T* src = g + size_t(b) * pitch;
    each_elem([&](int m, auto full) {
      reg[m] = (valid && (decltype(full)::value || elem(m) < P.L)) ? src[elem(m)] : T(0);
    });
  }
  // (Elements r + 16 m with m < L / 16 exist in every lane.  A per-lane guard on every element is an exec-mask sequence
  // of ~8 instructions each — the masks live in spilled SGPRs — and lds_to_reg / publish_direction run in every Arnoldi
  // iteration on every wave.  When all but the last two elements are full — L > 16 (MAXM - 2): what a kernel of this
  // MAXM is normally chosen for — those are moved without a guard, in their own arm of a SCALAR branch (guards inside
  // one loop, `m < L / 16 || e < L`, the compiler folds back into ten per-lane masks).)
  static constexpr int MFAST = MAXM - 2;
  __device__ __forceinline__ bool mostly_full() const { return (P.L >> 4) >= MFAST; }
  template <class F>
  __device__ __forceinline__ void each_elem(F&& f) const {  // f(m, full): full = element m exists in every lane
    if (mostly_full()) {
#pragma unroll
      for (int m = 0; m < MFAST; ++m) f(m, std::true_type{});
#pragma unroll
      for (int m = MFAST; m < MAXM; ++m) f(m, std::false_type{});
    } else {
#pragma unroll
      for (int m = 0; m < MAXM; ++m) f(m, std::false_type{});
    }
  }
  __device__ __forceinline__ void lds_to_reg(T* reg, const T* lds) const {
    const T* row = lds + inst * P.Lp;
    each_elem([&](int m, auto full) { reg[m] = (decltype(full)::value || elem(m) < P.L) ? row[elem(m)] : T(0); });
  }
  __device__ __forceinline__ void reg_to_lds(T* lds, const T* reg) const {
    T* row = lds + inst * P.Lp;
    each_elem([&](int m, auto full) {
      if (decltype(full)::value || elem(m) < P.L) row[elem(m)] = reg[m];
    });
  }
  // The direction d of a matrix-vector product is published to the sweeps as the perturbed control U + h*d
  // (cgmres.hpp:166-168 forms it in every stage); the sweep phases then read one array instead of two.
  __device__ __forceinline__ void publish_direction(const T* reg) const {
    T uu[MAXM];  // all reads first (pad lanes read in-bounds words of the next row, never stored)
    if constexpr (LEAN) {
      get_urow(uu);
    } else {
#pragma unroll
      for (int m = 0; m < MAXM; ++m) uu[m] = S.U[inst * P.Lp + elem(m)];
    }
    T* row = S.W + inst * P.Lp;
    if constexpr (NWT != 0) {
      // also: did the direction change any control of this row?  One that the rounding of U + h*d absorbs completely
      // leaves F unchanged bit for bit in the serial sweeps — A*d = 0 exactly, the reference's breakdown case
      // (gmres.hpp:63-65) — while Newton's fixed point agrees with the serial trajectory up to rounding only: gmres()
      // answers that case from this flag.
      bool moved = false;
      each_elem([&](int m, auto full) {
        if (decltype(full)::value || elem(m) < P.L) {
          const T wv = reg[m] * P.h + uu[m];
          moved = moved || wv != uu[m];
          row[elem(m)] = wv;
        }
      });
      row_moved = ((__ballot(moved) >> (16 * ((tid >> 4) & 3))) & 0xffffull) != 0;
      return;
    }
    each_elem([&](int m, auto full) {  // (see lds_to_reg)
      if (decltype(full)::value || elem(m) < P.L) row[elem(m)] = reg[m] * P.h + uu[m];
    });
  }
  // lean plan: the unperturbed sweeps read U through W as well
  __device__ __forceinline__ void publish_U() const {
    if constexpr (LEAN) {
      T uu[MAXM];
      get_urow(uu);
      reg_to_lds(S.W, uu);
    }
  }
  // lean plan: this row's U (registers, or its HBM row — written by this same thread, so program order suffices)
  __device__ __forceinline__ void get_urow(T* dst) const {
    if constexpr (U_IN_REGS) {
#pragma unroll
      for (int m = 0; m < MAXM; ++m) dst[m] = ureg[m];
    } else {
      load_row_to_reg(dst, P.U, P.Lg);
    }
  }
  __device__ __forceinline__ void reg_to_row(T* g, size_t pitch, const T* reg) const {
    if (!valid) return;
    T* dst = g + size_t(b) * pitch;
    each_elem([&](int m, auto full) {
      if (decltype(full)::value || elem(m) < P.L) dst[elem(m)] = reg[m];
    });
  }
  __device__ __forceinline__ T* vrow(int j) const {  // Krylov vector j of this row's instance
    if constexpr (NWT != 0) {
      // From an opaque copy of the instance index at every use: the row addresses are invariant over the whole launch,
      // and hoisted to the kernel entry they are a dozen 64-bit values in scratch — each reload in the Gram-Schmidt rounds
      // sits behind s_waitcnt vmcnt(0), i.e. waits for every basis row in flight.
      int bo = b;
      asm volatile("" : "+v"(bo));
      return P.V + (size_t(bo) * (P.kmax + 1) + j) * P.Lv;
    }
    return P.V + (size_t(b) * (P.kmax + 1) + j) * P.Lv;
  }
  // Krylov rows have pitch 16*MAXM and zero pads (every vector written here has zero pads: lds_to_reg clears
  // them and all later operations are linear), so neither direction needs a guard.  That matters for more than the
  // compare: a guarded load is an exec-masked branch, and the compiler cannot count outstanding loads across
  // branches — every wait in the Gram-Schmidt rounds became vmcnt(0) and serialised the register ring.
  // In HBM a Krylov row is stored pair-interleaved — element (r + 16 m) at (m/2)*32 + 2r + (m&1) — so a lane moves
  // two of its elements per 16-byte access: half the vector-memory instructions (the CU's address unit takes ~16
  // cycles per 64-lane access and four waves issue their rows at once).  ctx_wg undoes the permutation on export.
  struct alignas(2 * sizeof(T)) Pair {
    T a, b;
  };
  static_assert(MAXM % 2 == 0, "pair-interleaved Krylov rows");
  __device__ __forceinline__ void load_vec(T* reg, const T* row) const {
    const Pair* q = reinterpret_cast<const Pair*>(row) + r;
#pragma unroll
    for (int m = 0; m < MAXM; m += 2) {
      const Pair t = q[(m / 2) * 16];
      reg[m] = t.a, reg[m + 1] = t.b;
    }
  }
  __device__ __forceinline__ void store_vec(T* row, const T* reg) const {
    Pair* q = reinterpret_cast<Pair*>(row) + r;
#pragma unroll
    for (int m = 0; m < MAXM; m += 2) q[(m / 2) * 16] = Pair{reg[m], reg[m + 1]};
  }

  // element q of this row's parameter horizon: LDS row, or (lean) the workgroup's transposed copy in HBM/L2, which the
  // coefficient phase reads 16 instances at a time.  Readers are other waves of this workgroup = same CU, same L1:
  // visible after a barrier that drains vmcnt (__syncthreads), like the parked stage tables.
  __device__ __forceinline__ void put_p(int q, T v) const {
    if constexpr (LEAN)
      pTw[size_t(q) * IPW + inst] = v;
    else
      S.p[inst * P.Pp + q] = v;
  }
  __device__ __forceinline__ T get_p(int i, int q) const { return LEAN ? pTw[size_t(q) * IPW + i] : S.p[i * P.Pp + q]; }
  // the state equation at `stage` for instance i: the built-in models ignore p; a user model compiled through the
  // plugin path may read it (Model::dxdt(ret, x, u, p), <example>/model.hpp:36)
  __device__ __forceinline__ void model_dxdt(T* f, const T* x, const T* u, T* tr, int i, int stage) const {
    if constexpr (DxdtUsesP<M, T>::value) {
      T p[M::NP > 0 ? M::NP : 1];
#pragma unroll
      for (int j = 0; j < M::NP; ++j) p[j] = get_p(i, stage * M::NP + j);
      M::dxdt_p(f, x, u, p);
    } else {
      M::dxdt(f, x, u, tr, mc);
    }
  }
  // Common prologue: U, ptau -> LDS; x -> LDS (component-major); flags cleared.  All global loads of the row are
  // issued before the first LDS store so they overlap (one HBM/L2 round trip instead of one per element).
  __device__ __forceinline__ void load_common(const T* Ug) {
    constexpr int PMAX = 8;  // ptau entries per lane held in flight (covers dim_p*(dv+1) <= 128)
    const int np_all = M::NP * (P.dv + 1);
    T urow[MAXM], preg[PMAX], xreg = T(0);
    if constexpr (!LEAN || U_IN_REGS) load_row_to_reg(urow, Ug, P.Lg);
    if (valid) {
#pragma unroll
      for (int n = 0; n < PMAX; ++n) {
        const int q = r + 16 * n;
        preg[n] = q < np_all ? P.ptau[size_t(b) * np_all + q] : T(0);
      }
      if (r < M::NX && P.x_in) xreg = P.x_in[size_t(b) * M::NX + r];
    }
    if constexpr (U_IN_REGS) {
#pragma unroll
      for (int m = 0; m < MAXM; ++m) ureg[m] = urow[m];
    }
    if (valid) {
      if constexpr (!LEAN) reg_to_lds(S.U, urow);
#pragma unroll
      for (int n = 0; n < PMAX; ++n) {
        const int q = r + 16 * n;
        if (q < np_all) put_p(q, preg[n]);
      }
      for (int q = r + 16 * PMAX; q < np_all; q += 16) put_p(q, P.ptau[size_t(b) * np_all + q]);
      if (r < M::NX) S.xs[r * IPW + inst] = xreg;
    }
    if (r == 0) {
      S.flag[inst] = 0;
      S.reason[inst] = 0;
      S.nax[inst] = 0;
      S.ksolve[inst] = 0;
      S.binst[inst] = b;
    }
  }

  // set_ptau before a tick of the fused loop (cgmres.hpp:36-39): this row's parameter horizon -> LDS
  __device__ __forceinline__ void load_ptau_tick(const T* seq_tick) {
    if (!valid) return;
    const int np_all = M::NP * (P.dv + 1);
    const T* src = seq_tick + size_t(b) * P.pseq_inst;
    for (int q = r; q < np_all; q += 16) put_p(q, src[q]);
  }

  // ---- horizon sweep (cgmres.hpp:113-162; with PERT also :168-169, with MODE the post-processing) ------------
  //   MODE F_PLAIN: out = F        F_RHS: out = (F*(1-zeta h) - Fh)/h  (:91-96)        F_AX: out = (F - Fh)/h  (:173-174)
  //   PERT: u = U + h*W.  `out` may be W itself: every (stage, instance) entry of W is read for the last time in
  //   phase 2 by the thread that then overwrites it.
  // Three phases (see the header comment).  A stage table is dv*NSTG*IPW scalars indexed [(stage*NSTG + slot)*IPW + i];
  // phase 1 writes x(s), trig(s) to `tab` — the LDS table S.R or, for the concurrent preamble sweeps, a per-workgroup
  // table in HBM — phase 2 reads `tab` and leaves the costate coefficients in S.R, phase 3 consumes S.R.

  // The Fh-in-HBM mode only exists in the long-vector instantiations (L > 160 is where LDS gets tight); the short ones
  // compile it out, so the headline kernel carries none of its branches.
  __device__ __forceinline__ bool fh_hbm() const { return LEAN || (MAXM > 10 && P.fh_hbm); }
  // value of lane `src` (a lane of this wave) through the LDS crossbar
  static __device__ __forceinline__ double row_bcast(double v, int src) {
    const int lo = __builtin_amdgcn_ds_bpermute(src * 4, __double2loint(v));
    const int hi = __builtin_amdgcn_ds_bpermute(src * 4, __double2hiint(v));
    return __hiloint2double(hi, lo);
  }
  static __device__ __forceinline__ float row_bcast(float v, int src) {
    return __int_as_float(__builtin_amdgcn_ds_bpermute(src * 4, __float_as_int(v)));
  }
  // Workgroup barrier that orders LDS only.  __syncthreads() also drains the wave's outstanding HBM traffic
  // (s_waitcnt vmcnt(0)); inside a sweep nothing another wave needs travels through HBM.
  __device__ __forceinline__ void lds_barrier() const {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
  }
  // Phases 1 and 2 are pipelined in chunks of stages (f_eval): the sweep wave publishes a chunk of the stage table
  // and goes on with the next one while the other waves turn the published chunk into costate coefficients.
  // Chunks of ~25 stages measured best (round 2, after the coefficient phase moved to stage pairs: 2 chunks at dv = 50 —
  // 1: -5 %, 3: -1.2 %, 4: -2.5 %, 5: -4 % — and 4 at dv = 100; round 1 had 4 at dv = 50): a barrier costs the sweep
  // wave ~300 cycles, and only the processing of the last chunk stays on the critical path.
  // SPLIT_TAIL (full plans with the quad sweep, 16 instances): the LAST pipeline chunk is what the costate sweep waits
  // for after the state sweep, so it is kept at SPLIT_N = 12 stages — 192 (stage, instance) items, ONE per lane of the
  // three coefficient waves (coeffs_chunked) instead of a stage pair per lane — and the first chunk takes the rest
  // (its coefficients are computed in the shadow of the sweep's last chunk either way).
  static constexpr int SPLIT_N = 12;
  __device__ __forceinline__ int chunk_len() const {  // even: stages go in pairs
    if (SPLIT_TAIL && P.dv <= 64 && P.dv >= 2 * SPLIT_N) return (P.dv - SPLIT_N + 1) & ~1;
    const int chunks = P.dv <= 64 ? 2 : 4;
    return 2 * ((P.dv + 2 * chunks - 1) / (2 * chunks));
  }

  // phase 1: state sweep, cgmres.hpp:132-140, on the lanes [lane0, lane0 + 64) of one wave; x(dv) -> xT[c*IPW + i].
  // PIPE: one lds_barrier() after every chunk (the caller's other waves run coeffs_chunked, which has the matching ones).
  // ALT (quad sweep, pipelined only): x0 / x2 are stored at the first stage of every pair only (see M::x02_step)
  template <bool PERT, bool PIPE, bool ALT = false>
  __device__ __forceinline__ void sweep_state(int lane0, const T* x0c, T dtau, T* tab, T* xT, bool only_active) {
    constexpr int NX = M::NX, NU = M::NU, NC = M::NC;
    constexpr int STEP = TAB_STEP;
    const int dv = P.dv, lt = tid - lane0, CH = chunk_len();
    if (lt < 0 || lt >= 64) return;
    if constexpr (M::HAS_QUAD_SWEEP) {
      // four lanes (one DPP quad) per instance — see PendulumDev::quad_stage
      const int qi = lt >> 2;
      int rho = lt & 3;
      // Lean plan (256 registers per wave): keep the per-lane kernel coefficients from being hoisted to the kernel
      // entry — live across the whole tick they get spilled and every chunk of every sweep starts by waiting for
      // their scratch reloads (+5 k cycles per sweep); made opaque here they are ~12 selects per sweep call.
      if constexpr (LEAN) asm volatile("" : "+v"(rho));
      const bool goq = lt < 4 * IPW && blockIdx.x * IPW + qi < P.B && (!only_active || act_q);
      typename M::QuadLane Q;
      Q.init(rho, mc);
      const T* __restrict__ U = (PERT || LEAN ? S.W : S.U) + qi * P.Lp;  // PERT: W holds U + h*direction (publish_direction)
      const T dtau1 = Q.sg * dtau;
      T x[NX], v = T(0), amax = T(0);
      if (goq) {
#pragma unroll
        for (int c = 0; c < NX; ++c) x[c] = x0c[c * IPW + qi];
        v = M::template quad_begin<false>(x, Q, mc, &amax);
      }
      // Branch-free stages, two per loop trip: all lanes store (see the slot map in models.hip.h), u0 of the next
      // stage is fetched one stage ahead (index dv*NU is the pad word of the odd-pitch row), and every table
      // pointer advances by the same constant so the second stage of a trip addresses with immediates.
      // table/control pointers and the prefetched u0 persist across chunks (only the slow redo rewinds them)
      T* pa = tab + qi;
      // the lane that holds +x1 stores it in the place of its own trig value (sin x1, which no later phase reads): one
      // select (2 instructions) instead of a second 8-byte LDS store (14.6 cycles) per stage
      const bool x1_lane = rho == M::QLANE_TRUE_X;
      T* pv = pa + Q.slot_v * IPW;
      const T* pu = U;
      T ua = T(0);
      if (goq) ua = pu[0];
      // Stage modes: 0 = trig value rotated from the previous stage (PendulumDev::quad_stage_rot), 1 = fresh evaluation
      // per stage (fast kernel), 2 = fresh evaluation with the library sin/cos.  A chunk runs at the
      // sweep's LEVEL (0: rotation, 1: fresh evaluations) and is redone one level up when an angle increment left the
      // rotation's range, in mode 2 when an argument left the fast kernel's range.  The level is kept from sweep to
      // sweep (rot_level: the perturbed sweeps of a tick and the ticks that follow see nearly the same increments —
      // without that memory a batch in fast motion paid for a failed rotation pass in EVERY sweep: 176 instead of 145
      // us/tick at tick 5500 of the seeded scenario, where the swing-up reaches 10 rad/s) and rotation is tried again
      // every ROT_HOLD sweeps.  (A third form — rotation with one more Taylor term per polynomial, good to 0.145 rad per
      // stage — was built and measured: its extra copy of the stage loop cost the slow regime 3.6 %; dropped.)
      constexpr int ROT_HOLD = 64;
      T argp = T(0);
      int zmax = 0;
      int lvl = __builtin_amdgcn_readfirstlane(rot_level);  // wave-uniform (kept in a scalar register)
      if (rot_hold == 0 && lvl > 0) lvl = 0, rot_hold = ROT_HOLD;
      auto run = [&](auto mode_tag, int n) {
        constexpr int MODE = decltype(mode_tag)::value;
        auto stage = [&](int o, T u0) {
          if (!ALT || (o & 1) == 0) {  // (o is a literal at every call site)
            pa[o * STEP + M::QSLOT_XA * IPW] = x[0];
            pa[o * STEP + M::QSLOT_XB * IPW] = x[2];
          }
          pv[o * STEP] = x1_lane ? x[1] : v;
          if constexpr (MODE == 0)
            M::quad_stage_rot(x, v, argp, u0, dtau, dtau1, Q, mc, &zmax);
          else
            M::template quad_stage<MODE == 2>(x, v, u0, dtau, dtau1, Q, mc, &amax);
        };
        int k = 0;
        if constexpr (MODE == 0) {  // rotation mode: four stages per trip (a taken branch costs ~30 cycles)
          for (; k + 4 <= n; k += 4) {
            const T ub = pu[NU];
            stage(0, ua);
            const T uc = pu[2 * NU];
            stage(1, ub);
            const T ud = pu[3 * NU];
            stage(2, uc);
            ua = pu[4 * NU];
            stage(3, ud);
            pa += 4 * STEP, pv += 4 * STEP, pu += 4 * NU;
          }
        }
        for (; k + 2 <= n; k += 2) {
          const T ub = pu[NU];
          stage(0, ua);
          ua = pu[2 * NU];
          stage(1, ub);
          pa += 2 * STEP, pv += 2 * STEP, pu += 2 * NU;
        }
        if (k < n) {  // odd tail: only the last chunk can have one (chunk_len() is even)
          stage(0, ua);
          pa += STEP, pv += STEP, pu += NU;
        }
      };
      for (int s0 = 0; s0 < dv; s0 += CH) {
        const int n = dv - s0 < CH ? dv - s0 : CH;
        // (the level decisions are taken by ALL lanes of the wave, also those without an instance: the level stays
        // wave-uniform and the stage loops never run under a divergent branch)
        T xs[NX];
#pragma unroll
        for (int c = 0; c < NX; ++c) xs[c] = x[c];
        auto rewind = [&]() {
#pragma unroll
          for (int c = 0; c < NX; ++c) x[c] = xs[c];
          pa = tab + qi + s0 * STEP, pv = pa + Q.slot_v * IPW;
          pu = U + s0 * NU;
          if (goq) ua = pu[0];
        };
        // v (and with it the rotation) is carried from the previous chunk / quad_begin; only a redo starts from a fresh
        // evaluation.  (Re-evaluating at every chunk start bounded the rotation's drift to one chunk — 26 stages at
        // dv = 50 — at ~40 instructions per chunk; over the whole horizon the drift stays five orders below the parity
        // bound, tests/test_gpu_parity.py.  126.8 -> 125.6 us/tick.)
        bool stale_v = false;
        bool done = false;
        if (lvl == 0) {
          if (goq) {
            if (stale_v) v = M::template quad_trig<false>(M::quad_arg(x, Q), Q, mc, &amax);
            argp = M::quad_arg(x, Q);
            run(std::integral_constant<int, 0>{}, n);
          }
          if (__builtin_expect(__any(goq && M::quad_rot_bad(zmax)), 0)) {  // an angle moved too far in one stage
            lvl = 1, rot_hold = ROT_HOLD, stale_v = true;
            rewind();
          } else {
            done = true;
          }
          zmax = 0;
        }
        if (__builtin_expect(!done, 0)) {
          if (goq) {
            if (stale_v) v = M::template quad_trig<false>(M::quad_arg(x, Q), Q, mc, &amax);
            run(std::integral_constant<int, 1>{}, n);
            if (n & 1) ua = pu[0];
          }
        }
        // an argument outside the fast range of the trig kernel: redo this chunk with the library sin/cos
        if (__builtin_expect(__any(goq && M::quad_arg_bad(amax)), 0)) {
          rewind();
          if (goq) {
            v = M::template quad_trig<true>(M::quad_arg(x, Q), Q, mc, &amax);
            run(std::integral_constant<int, 2>{}, n);
            if (n & 1) ua = pu[0];
          }
          amax = T(0);
        }
        if (PIPE) lds_barrier();
      }
      rot_level = lvl;
      if (rot_hold > 0) --rot_hold;
      if (goq && rho == M::QLANE_TRUE_X) {
#pragma unroll
        for (int c = 0; c < NX; ++c) xT[c * IPW + qi] = x[c];
      }
    } else {
      const int i = lt;
      const bool go = i < IPW && blockIdx.x * IPW + i < P.B && (!only_active || act_i);
      const T* __restrict__ U = (PERT || LEAN ? S.W : S.U) + i * P.Lp;  // (lean: W holds U for the unperturbed sweeps)
      T* __restrict__ R = tab + i;
      T xs[NX];
      if (go) {
#pragma unroll
        for (int c = 0; c < NX; ++c) xs[c] = x0c[c * IPW + i];
      }
      // Same loop shape as the quad sweep: two stages per trip, the controls of the next stage fetched one stage
      // ahead (the words after the last stage belong to the row's pad / the next row: in-bounds, unused), walking
      // pointers so the second stage of a trip addresses with immediates.
      struct Uc {
        T v[M::NU_DYN];
      };
      T* pr = R;
      const T* pu = U;
      auto fetch = [&](Uc& a, const T* q) {
#pragma unroll
        for (int j = 0; j < M::NU_DYN; ++j) a.v[j] = q[j];
      };
      auto stage = [&](const Uc& a, T* q, int sidx) {
        T f[NX], tr[NC > 0 ? NC : 1];
#pragma unroll
        for (int c = 0; c < NX; ++c) q[c * IPW] = xs[c];
        model_dxdt(f, xs, a.v, tr, i, sidx);
#pragma unroll
        for (int c = 0; c < NC; ++c) q[(M::TRIG_SLOT0 + c) * IPW] = tr[c];
#pragma unroll
        for (int c = 0; c < NX; ++c) xs[c] = f[c] * dtau + xs[c];
      };
      Uc ua, ub;
      if (go) fetch(ua, pu);
      for (int s0 = 0; s0 < dv; s0 += CH) {
        const int n = dv - s0 < CH ? dv - s0 : CH;
        if (go) {
          int k = 0;
          for (; k + 2 <= n; k += 2) {
            fetch(ub, pu + NU);
            stage(ua, pr, s0 + k);
            fetch(ua, pu + 2 * NU);
            stage(ub, pr + STEP, s0 + k + 1);
            pr += 2 * STEP, pu += 2 * NU;
          }
          if (k < n) {  // odd tail (last chunk only)
            stage(ua, pr, s0 + k);
            pr += STEP, pu += NU;
          }
        }
        if (PIPE) lds_barrier();
      }
      if (go) {
#pragma unroll
        for (int c = 0; c < NX; ++c) xT[c * IPW + i] = xs[c];
      }
    }
  }

  // phase 2: costate-free part of one backward stage of one instance.
  // Reads x/trig from `tab`, writes the coefficients to S.R and the costate-free part of the result to `out`.
  // Operands of an item that live in HBM/L2 — the parameter horizon in the lean plan, F(U,x+hf,t+h) with fh_hbm —
  // are fetched for a whole group of items BEFORE the group is processed (and, behind the state sweep, before the
  // barrier that releases the chunk): an item-by-item fetch exposes two dependent HBM/L2 round trips per item and
  // the coefficient waves then fall behind the sweep wave (measured: +20 % on every sweep).
  static constexpr bool HBM_OPERANDS = LEAN || MAXM > 10;  // kernels that carry the fh_hbm / lean code
  // the pipelined quad sweep stores x0 / x2 every other stage; the coefficient phase works on stage pairs
  static constexpr bool ALT_X02 = M::HAS_QUAD_SWEEP && IPW * 16 >= 128;
  static constexpr bool SPLIT_TAIL = ALT_X02 && !HBM_OPERANDS && IPW == 16;
  static_assert(!SPLIT_TAIL || IPW * 16 - 64 == 16 * 12, "SPLIT_N stages x 16 instances = the lanes of the three coefficient waves");
  static constexpr int COEFF_GROUP = sizeof(T) == 8 ? 2 : 3;
  struct CoeffPre {
    T p[M::NP > 0 ? M::NP : 1], fh[M::NU];
  };
  template <int MODE>
  __device__ __forceinline__ void coeff_prefetch(CoeffPre& c, int s, int i) const {
    if constexpr (LEAN) {
#pragma unroll
      for (int j = 0; j < M::NP; ++j) c.p[j] = pTw[size_t(s * M::NP + j) * IPW + i];
    }
    if (MODE != F_PLAIN && fh_hbm()) {
      // explicitly global: left generic, hipcc merges this pointer with an LDS one and trips over the flat
      // aperture cast ("Illegal instruction detected ... $src_shared_base", ROCm 7.2)
      typedef const T __attribute__((address_space(1))) * GPtr;
      const GPtr row = reinterpret_cast<GPtr>(reinterpret_cast<uintptr_t>(P.Fh)) + size_t(S.binst[i]) * P.Lg + s * M::NU;
#pragma unroll
      for (int j = 0; j < M::NU; ++j) c.fh[j] = row[j];
    }
  }
  // x02: x0 and x2 of this stage when the table does not hold them (second stage of a pair, ALT_X02)
  template <bool PERT, int MODE>
  __device__ __forceinline__ void coeff_item(int s, int i, T dtau, const T* tab, T* out, const CoeffPre* pre = nullptr,
                                             const T* x02 = nullptr) {
    constexpr int NX = M::NX, NU = M::NU, NP = M::NP, NC = M::NC, NBW = M::NBW;
    const T sc_phi = MODE == F_RHS ? P.one_m_zh : T(1.0);
    T x[NX], tr[NC > 0 ? NC : 1], u[NU], p[NP > 0 ? NP : 1], bw[NBW], phi[NU];
    const T* Rs = tab + tab_off(s) + i;
#pragma unroll
    for (int c = 0; c < NX; ++c) x[c] = Rs[c * IPW];
    if constexpr (ALT_X02) {
      if (x02) x[0] = x02[0], x[2] = x02[1];
    }
#pragma unroll
    for (int c = 0; c < NC; ++c) tr[c] = Rs[(M::TRIG_SLOT0 + c) * IPW];
#pragma unroll
    for (int j = 0; j < NU; ++j) u[j] = (PERT || LEAN ? S.W : S.U)[i * P.Lp + s * NU + j];
#pragma unroll
    for (int j = 0; j < NP; ++j) p[j] = (LEAN && pre) ? pre->p[j] : get_p(i, s * NP + j);
    M::stage_coeffs(bw, phi, x, u, p, tr, dtau);
    // coefficients go back pair-interleaved — pair c of instance i at [(c*IPW + i)*2, +1] of the stage's region —
    // so the costate sweep fetches them with 16-byte LDS reads (ds_read_b128: 8 cycles; ds_read2_b64: 16).  In place
    // is safe: the 16 instances of a stage are 16 adjacent lanes of ONE wave in ONE pass, and all their reads of
    // x/trig precede these writes in program order.
    static_assert(NBW % 2 == 0 && (IPW * 16) % 64 == 0, "pair-interleaved costate coefficients");
    {
      Pair* Rp = reinterpret_cast<Pair*>(S.R + tab_off(s)) + i;
#pragma unroll
      for (int c = 0; c < NBW; c += 2) Rp[(c / 2) * IPW] = Pair{bw[c], bw[c + 1]};
    }
#pragma unroll
    for (int j = 0; j < NU; ++j) {
      T rj = phi[j];
      if (MODE != F_PLAIN) {
        // two loads from two address spaces, selected by VALUE (a select of the pointers would have to go through
        // the flat aperture); with fh_hbm the LDS word read is a word of W and is not used
        T fh;
        if constexpr (LEAN) {
          fh = pre->fh[j];
        } else {
          fh = S.Fh[i * P.Lp + s * NU + j];
          if (fh_hbm()) fh = pre->fh[j];
        }
        rj = (rj * sc_phi - fh) * P.inv_h;
      }
      out[i * P.Lp + s * NU + j] = rj;
    }
  }
  // all threads, items (s, i) with i fastest
  // (i = tid mod IPW at every call site: the instance index of an item advances in steps that are multiples of IPW)
  __device__ __forceinline__ bool item_on(int i, bool only_active) const {
    return blockIdx.x * IPW + i < P.B && (!only_active || act_i);
  }
  // items q0, q0 + stride, ... (COEFF_GROUP of them) of the range [0, n_items), stage offset s0: fetch, `between`, compute
  template <bool PERT, int MODE, class Between>
  __device__ __forceinline__ void coeff_group(int q0, int stride, int n_items, int s0, T dtau, const T* tab, T* out,
                                              bool only_active, Between&& between) {
    CoeffPre pre[COEFF_GROUP];
#pragma unroll
    for (int g = 0; g < COEFF_GROUP; ++g) {
      const int q = q0 + g * stride, i = q & (IPW - 1);
      if (q < n_items && item_on(i, only_active)) coeff_prefetch<MODE>(pre[g], s0 + q / IPW, i);
    }
    between();
#pragma unroll
    for (int g = 0; g < COEFF_GROUP; ++g) {
      const int q = q0 + g * stride, i = q & (IPW - 1);
      if (q < n_items && item_on(i, only_active)) coeff_item<PERT, MODE>(s0 + q / IPW, i, dtau, tab, out, &pre[g]);
    }
  }
  // the same for items = (stage pair, instance) of the stages [s0, s_end) of the LDS table (ALT_X02): the second stage
  // of a pair takes x0 / x2 from M::x02_step instead of the table
  template <bool PERT, int MODE, class Between>
  __device__ __forceinline__ void coeff_group_pairs(int q0, int stride, int s0, int s_end, T dtau, T* out,
                                                    bool only_active, Between&& between) {
    const int n_items = ((s_end - s0 + 1) >> 1) * IPW;
    CoeffPre pre[COEFF_GROUP][2];
#pragma unroll
    for (int g = 0; g < COEFF_GROUP; ++g) {
      const int q = q0 + g * stride, i = q & (IPW - 1), s = s0 + 2 * (q / IPW);
      if (q < n_items && item_on(i, only_active)) {
        coeff_prefetch<MODE>(pre[g][0], s, i);
        if (s + 1 < s_end) coeff_prefetch<MODE>(pre[g][1], s + 1, i);
      }
    }
    between();
#pragma unroll
    for (int g = 0; g < COEFF_GROUP; ++g) {
      const int q = q0 + g * stride, i = q & (IPW - 1), s = s0 + 2 * (q / IPW);
      if (q < n_items && item_on(i, only_active)) {
        T x02[2] = {S.R[tab_off(s, M::QSLOT_XA) + i], S.R[tab_off(s, M::QSLOT_XB) + i]};
        const T u0 = (PERT || LEAN ? S.W : S.U)[i * P.Lp + s * M::NU];
        coeff_item<PERT, MODE>(s, i, dtau, S.R, out, &pre[g][0]);
        if (s + 1 < s_end) {
          M::x02_step(x02[0], x02[1], u0, dtau);
          coeff_item<PERT, MODE>(s + 1, i, dtau, S.R, out, &pre[g][1], x02);
        }
      }
    }
  }
  template <bool PERT, int MODE>
  __device__ __forceinline__ void sweep_coeffs(T dtau, const T* tab, T* out, bool only_active) {
    if constexpr (HBM_OPERANDS) {
      for (int q0 = tid; q0 < P.dv * IPW; q0 += IPW * 16 * COEFF_GROUP)
        coeff_group<PERT, MODE>(q0, IPW * 16, P.dv * IPW, 0, dtau, tab, out, only_active, [] {});
    } else {
      for (int q = tid; q < P.dv * IPW; q += IPW * 16) {
        const int i = q & (IPW - 1), s = q / IPW;
        if (!item_on(i, only_active)) continue;
        coeff_item<PERT, MODE>(s, i, dtau, tab, out);
      }
    }
  }
  // the waves other than wave 0, chunk by chunk behind sweep_state<PERT, true> (one lds_barrier() before every chunk)
  template <bool PERT, int MODE>
  __device__ __forceinline__ void coeffs_chunked(T dtau, T* out, bool only_active) {
    constexpr int NL = IPW * 16 - 64;
    const int dv = P.dv, CH = chunk_len(), lt = tid - 64;
    for (int s0 = 0; s0 < dv; s0 += CH) {
      const int n = dv - s0 < CH ? dv - s0 : CH;
      if constexpr (HBM_OPERANDS && ALT_X02) {
        coeff_group_pairs<PERT, MODE>(lt, NL, s0, s0 + n, dtau, out, only_active, [&] { lds_barrier(); });
        for (int q0 = lt + NL * COEFF_GROUP; q0 < ((n + 1) >> 1) * IPW; q0 += NL * COEFF_GROUP)
          coeff_group_pairs<PERT, MODE>(q0, NL, s0, s0 + n, dtau, out, only_active, [] {});
      } else if constexpr (HBM_OPERANDS) {
        // the first group's HBM operands are requested BEFORE the barrier, i.e. while the sweep wave still works on
        // this chunk (every thread reaches the barrier exactly once per chunk, with or without items)
        coeff_group<PERT, MODE>(lt, NL, n * IPW, s0, dtau, S.R, out, only_active, [&] { lds_barrier(); });
        for (int q0 = lt + NL * COEFF_GROUP; q0 < n * IPW; q0 += NL * COEFF_GROUP)
          coeff_group<PERT, MODE>(q0, NL, n * IPW, s0, dtau, S.R, out, only_active, [] {});
      } else if constexpr (ALT_X02) {
        lds_barrier();
        if (SPLIT_TAIL && n <= SPLIT_N) {
          // One STAGE per lane: lanes 0-31 of a wave take the first stages of two pairs, lanes 32-63 their second stages
          // (which re-derive x0 / x2 from the pair's first stage like the pair items do).  The coefficients of a stage
          // overwrite its slots in place, and a second-stage lane reads two slots of the FIRST stage: both lanes sit in
          // the same wave and every read below precedes every write in program order (straight-line code, no branch
          // between the halves; a wave's LDS operations complete in order).
          const int l = lt & 63, i = l & (IPW - 1), half = l >> 5;
          const int ps = s0 + 2 * (2 * (lt >> 6) + ((l >> 4) & 1)), st = ps + half;
          if (st < s0 + n && item_on(i, only_active)) {
            T x02[2] = {S.R[tab_off(ps, M::QSLOT_XA) + i], S.R[tab_off(ps, M::QSLOT_XB) + i]};
            const T u0 = (PERT || LEAN ? S.W : S.U)[i * P.Lp + ps * M::NU];
            T x0n = x02[0], x2n = x02[1];
            M::x02_step(x0n, x2n, u0, dtau);
            if (half) x02[0] = x0n, x02[1] = x2n;
            coeff_item<PERT, MODE>(st, i, dtau, S.R, out, nullptr, x02);
          }
          continue;
        }
        for (int q = lt; q < ((n + 1) >> 1) * IPW; q += NL) {  // items = (stage pair, instance)
          const int i = q & (IPW - 1), s = s0 + 2 * (q / IPW);
          if (!item_on(i, only_active)) continue;
          T x02[2] = {S.R[tab_off(s, M::QSLOT_XA) + i], S.R[tab_off(s, M::QSLOT_XB) + i]};
          const T u0 = (PERT || LEAN ? S.W : S.U)[i * P.Lp + s * M::NU];
          coeff_item<PERT, MODE>(s, i, dtau, S.R, out);
          if (s + 1 < s0 + n) {
            M::x02_step(x02[0], x02[1], u0, dtau);
            coeff_item<PERT, MODE>(s + 1, i, dtau, S.R, out, nullptr, x02);
          }
        }
      } else {
        lds_barrier();
        for (int q = lt; q < n * IPW; q += NL) {
          const int i = q & (IPW - 1), s = s0 + q / IPW;
          if (!item_on(i, only_active)) continue;
          coeff_item<PERT, MODE>(s, i, dtau, S.R, out);
        }
      }
    }
  }

  // phase 3: costate sweep (cgmres.hpp:145-153) + the costate part of dH/du (:156-161).  COLLECTIVE (every thread
  // calls it); the caller adds the barrier that publishes `out`.
  template <int MODE>
  __device__ __forceinline__ void sweep_costate(T dtau, const T* xT, T* out, bool only_active) {
    if constexpr (PAR_COSTATE) {
      sweep_costate_par<MODE>(dtau, xT, out, only_active);
    } else if constexpr (PAR_2PASS) {
      sweep_costate_2pass<MODE>(dtau, xT, out, only_active);
    } else {
      if (!(sweep_lane && (!only_active || act_i))) return;
      T l[M::NX];
      costate_run<MODE, false, true>(l, tid, P.dv - 1, P.dv, dtau, S.R + 2 * tid, out + tid * P.Lp,
                               [&](T* l0) { costate_terminal(l0, xT, tid); });
    }
  }
  __device__ __forceinline__ void costate_terminal(T* l, const T* xT, int i) const {
    T xs[M::NX], p[M::NP > 0 ? M::NP : 1];
#pragma unroll
    for (int c = 0; c < M::NX; ++c) xs[c] = xT[c * IPW + i];
#pragma unroll
    for (int j = 0; j < M::NP; ++j) p[j] = get_p(i, P.dv * M::NP + j);
    M::dPhidx(l, xs, p);
  }
  // `count` (workgroup-uniform) stages hi, hi-1, ... of the recurrence on this lane, from l.
  //   HOM = false: the affine stage; coefficients from `coef` (this instance's pairs of stage 0), the costate part of
  //                dH/du is added to the row `orow`: orow[s*NU + j] += sc*dF[j]
  //   HOM = true:  the bias-free stage on the first NBW_LIN coefficients; dF itself goes to orow[(s*NUL + j)*HL]
  // One wave issues one instruction per ~4.4 cycles whatever it is (tools/ubench_issue.hip), so the loop is written
  // for instruction count: moving pointers with immediate offsets, no branch inside a trip, unconditional look-ahead
  // fetches.
  //   After the `count` common stages, lanes with `more` set go on for `extra` (workgroup-uniform, 0..3) stages; the
  //   first two of those use operands the look-ahead has fetched anyway.  `init` fills l; it is called after the first
  //   operand requests have been issued so that its own LDS reads share their round trip.
  //   WRITE = false: only the end value of l is wanted (first pass of sweep_costate_2pass): no output traffic at all.
  template <int MODE, bool HOM, bool WRITE, class Init>
  __device__ __forceinline__ void costate_run(T* l, int i, int hi, int count, T dtau, const T* coef, T* orow, Init&& init,
                                              int extra = 0, bool more = false) {
    constexpr int NU = M::NU, NBW = M::NBW, NUL = M::NUL;
    constexpr int NLOAD = HOM ? M::NBW_LIN : NBW;  // coefficients fetched per stage
    constexpr int OJ = HOM ? HL : 1;               // distance between the NUL outputs of a stage
    constexpr int opitch = HOM ? NUL * HL : NU;    // and between stages
    static_assert(NLOAD % 2 == 0, "coefficients are fetched in pairs");
    const T sc = MODE == F_PLAIN ? T(1.0) : (MODE == F_RHS ? P.one_m_zh * P.inv_h : P.inv_h);
    constexpr int STEP = TAB_STEP;
    struct Ops {
      T bw[NBW], o[NUL];
    };
    auto fetch = [&](Ops& a, const T* pr, const T* po) {
      const Pair* pp = reinterpret_cast<const Pair*>(pr);
#pragma unroll
      for (int c = 0; c < NLOAD; c += 2) {
        const Pair t = pp[(c / 2) * IPW];
        a.bw[c] = t.a, a.bw[c + 1] = t.b;
      }
      if constexpr (!HOM && WRITE) {
#pragma unroll
        for (int j = 0; j < NUL; ++j) a.o[j] = po[j * OJ];
      }
    };
    auto stage = [&](const Ops& a, T* po) {
      T dF[NUL];
      M::template costate_step<HOM>(l, dF, a.bw, dtau);
      if constexpr (WRITE) {
#pragma unroll
        for (int j = 0; j < NUL; ++j) po[j * OJ] = HOM ? dF[j] : a.o[j] + dF[j] * sc;
      }
    };
    // Three register sets, three stages per trip: the operands of stage t-2 are requested while stage t computes
    // (two LDS latencies of slack).  q/o sit on stage s-4 of the trip that starts with stage s: every access is
    // pointer + non-negative immediate.  Below stage 0 the look-ahead reads words of the preceding LDS arrays
    // (at most 3 stages = 3*STEP scalars — the `post` tail below — which CtxWg::lookahead_fits guarantees to lie
    // inside the row arrays in front of the table), never used.  LDS accesses outside the workgroup's allocation
    // FAULT on this platform (aperture violation): -DCGM_DEBUG_LDS checks every address here.
#ifdef CGM_DEBUG_LDS
    auto chk = [&](const void* ptr, int what) {  // (a flag, not a printf: 40 printf sites made this build take 20 minutes)
      const long off = static_cast<const char*>(ptr) - reinterpret_cast<const char*>(LEAN ? S.xs : S.U);
      if (off < 0 || off + 16 > long(P.lds_bytes)) g_cgm_lds_oob = 0x10000 | (what << 12) | (tid & 0xfff);
    };
#else
    auto chk = [&](const void*, int) {};
#endif
    auto fetch3 = [&](Ops& a, const T* pr3, const T* po3) {
      if (NLOAD) chk(pr3, 1), chk(pr3 + (NLOAD / 2 - 1) * 2 * IPW, 2);
      if (!HOM && WRITE) chk(po3, 3);
      fetch(a, pr3, po3);
    };
    auto stage3 = [&](const Ops& a, T* po3) {
      if (WRITE) chk(po3, 4);
      stage(a, po3);
    };
    Ops A, B, C;
    int rem = count;
    const T* q = coef + (hi - 4) * STEP;  // pair-interleaved coefficients, see coeff_item
    T* o = orow + (hi - 4) * opitch;
    fetch3(A, q + 4 * STEP, o + 4 * opitch);  // stage hi
    fetch3(B, q + 3 * STEP, o + 3 * opitch);  // stage hi-1
    init(l);
    if (!HOM && WRITE && PAR_COSTATE) CGM_STAMP(*this, 18);
    for (; rem >= 3; rem -= 3) {
      fetch3(C, q + 2 * STEP, o + 2 * opitch);
      stage3(A, o + 4 * opitch);
      fetch3(A, q + STEP, o + opitch);
      stage3(B, o + 3 * opitch);
      fetch3(B, q, o);
      stage3(C, o + 2 * opitch);
      q -= 3 * STEP, o -= 3 * opitch;
    }
    if (!HOM && WRITE && PAR_COSTATE) CGM_STAMP(*this, 19);
    // the last rem (0..2) common stages, then the extra ones of the lanes with `more`: A and B hold the next two
    const int post = rem + extra;  // <= 5
    if (post > 2) fetch3(C, q + 2 * STEP, o + 2 * opitch);
    if (post >= 1 && (rem >= 1 || more)) stage3(A, o + 4 * opitch);
    if (post > 3) fetch3(A, q + STEP, o + opitch);
    if (post >= 2 && (rem >= 2 || more)) stage3(B, o + 3 * opitch);
    if (post > 4) fetch3(B, q, o);
    if (post >= 3 && more) stage3(C, o + 2 * opitch);
    if (post >= 4 && more) stage3(A, o + opitch);
    if (post >= 5 && more) stage3(B, o);
    if (!HOM && WRITE && PAR_COSTATE) CGM_STAMP(*this, 20);
  }

  // The costate recurrence is AFFINE in l once the stage coefficients are stored (l' = M_s l + b_s, dF_s = B_s l), so
  // the horizon is cut into four chunks that run SIDE BY SIDE on the four waves instead of one lane per instance
  // walking all dv stages:
  //   chunk 0 (the last stages: its start value, the terminal costate, is known) is run directly;
  //   chunks 1..3 are run from l = 0 WITH the bias (particular solution; wave 0, one lane per instance and chunk — it
  //   writes o + sc*dF_particular to `out` like the direct lane) and from the NX unit vectors WITHOUT it (wave k:
  //   column c of the chunk's transfer matrix on lane c*IPW + i, its dF per stage into the scratch Hd);
  //   then l at the chunk boundaries follows from two small matrix-vector products (wave 0) and the homogeneous part
  //   of dF, sum_c l_c(start of chunk) * Hd[stage][c], is added to `out` by all threads (one (stage, instance) each).
  // dv stage-times of one wave become dv/4 + the remainder + ~2 short parallel phases; the result differs from the
  // sequential recurrence by rounding only (~1e-16 relative to |l|; tests/test_gpu_parity.py bounds it against the oracle).
  static constexpr int HL = M::NX * IPW;  // homogeneous lanes per chunk
  // PAR is a KERNEL TEMPLATE parameter, not a run-time switch: with both forms of the sweep in one kernel (or one body
  // with run-time chunk parameters) the register allocation of the Arnoldi loop tips over — two more spills inside the
  // loop, each reload behind an s_waitcnt vmcnt(0), cost 5 % of the tick (measured).
  static constexpr bool PAR_OK = M::COSTATE_HOM && IPW == 16 && M::NX * IPW <= 64 && M::NX % 2 == 0;
  static constexpr bool PAR_COSTATE = PAR == 1 && PAR_OK && !LEAN;
  static constexpr bool PAR_2PASS = PAR == 2 && PAR_OK;
  template <int MODE>
  __device__ __forceinline__ void sweep_costate_par(T dtau, const T* xT, T* out, bool only_active) {
    constexpr int NX = M::NX, NU = M::NU, NUL = M::NUL;
    static_assert(NX % 2 == 0, "boundary records are read in pairs");
    const int dv = P.dv, n = dv >> 2, n0 = dv - 3 * n;  // chunk k >= 1: stages [(3-k)n, (4-k)n); chunk 0: [3n, dv)
    const T sc = MODE == F_PLAIN ? T(1.0) : (MODE == F_RHS ? P.one_m_zh * P.inv_h : P.inv_h);
    T* Hd = S.scan;                             // [stage < 3n][j < NUL][c*IPW + i]
    T* Rec = Hd + 3 * n * NUL * HL;             // boundary records [k < 3][i][REC], see scan_count
    constexpr int REC = Lds::SCAN_REC;
    // every address below derives from the thread index and is invariant over ticks and Arnoldi iterations: left
    // visible, the compiler hoists those computations to the top of the kernel and keeps ~20 more values alive over the
    // whole tick loop (they end up in scratch, reloaded behind s_waitcnt vmcnt(0))
    int tid_o = tid;
    asm volatile("" : "+v"(tid_o));
    const int wave = tid_o >> 6, lane = tid_o & 63;
    // --- phase A: all chunks side by side
    if (wave == 0) {
      const int i = lane & (IPW - 1), k = lane >> 4;
      if (item_on(i, only_active)) {
        T l[NX];
        const int hi = k == 0 ? dv - 1 : (4 - k) * n - 1;
        costate_run<MODE, false, true>(l, i, hi, n, dtau, S.R + 2 * i, out + i * P.Lp,
                                 [&](T* l0) {
                                   costate_terminal(l0, xT, i);
                                   if (k != 0) {
#pragma unroll
                                     for (int c = 0; c < NX; ++c) l0[c] = T(0);
                                   }
                                 },
                                 n0 - n, k == 0);
        // chunk 0: its end value = the start value of chunk 1, record 0; chunk k >= 1: particular end value, record k
        // (chunk 3's is never read and goes to the spare record)
        T* dst = Rec + (k * IPW + i) * REC + NX * NX;
#pragma unroll
        for (int c = 0; c < NX; ++c) dst[c] = l[c];
      }
    } else if (lane < HL) {
      const int i = lane & (IPW - 1), c0 = lane / IPW, k = wave;
      if (item_on(i, only_active)) {
        T l[NX];
        costate_run<MODE, true, true>(l, i, (4 - k) * n - 1, n, dtau, S.R + 2 * i, Hd + lane, [&](T* l0) {
#pragma unroll
          for (int c = 0; c < NX; ++c) l0[c] = c == c0 ? T(1) : T(0);
        });
#pragma unroll
        for (int r = 0; r < NX; ++r) Rec[(k * IPW + i) * REC + r * NX + c0] = l[r];  // column c0 of the transfer matrix
      }
    }
    CGM_STAMP(*this, 16);
    lds_barrier();
    CGM_STAMP(*this, 17);
    // --- phase B, all threads, no further barrier.  Every LDS instruction costs its issuing wave 7..16 cycles (tools/
    //     ubench_lds.hip), so the work is dealt by WAVE such that nobody reads a boundary record it does not need:
    //       wave 0: the stages of chunk 1 — start value = record 0, no product;
    //       wave 1: chunk 2 — one NX x NX product, l(start of chunk 2) = p_1 + M_1 l(start of chunk 1);
    //       waves 2, 3: chunk 3 — two products, l(start of chunk 3) = p_2 + M_2 l(start of chunk 2);
    //     (redundantly in every thread of the wave, which is cheaper than a third barrier), then
    //     out[stage] += sc * sum_c l_c(start of the chunk) * Hd[stage][c] for the thread's stages j, j + stride, ...
    {
      const int w = __builtin_amdgcn_readfirstlane(wave);  // (scalar: the branches below are wave-uniform)
      const int k = w < 2 ? w + 1 : 3;
      const int i = lane & (IPW - 1);
      const int j0 = w < 3 ? lane >> 4 : 4 + (lane >> 4), stride = w < 2 ? 4 : 8;
      const Pair* rec = reinterpret_cast<const Pair*>(Rec + i * REC);  // (REC is even: records are 16-byte aligned)
      constexpr int RP = REC / 2, KP = IPW * RP;                      // pairs per record, per chunk
      T al[NX];
#pragma unroll
      for (int c = 0; c < NX; c += 2) {
        const Pair t = rec[(NX * NX + c) / 2];
        al[c] = t.a, al[c + 1] = t.b;
      }
      auto advance = [&](int kk) {  // al <- p_kk + M_kk al
        T m[NX * NX], nx[NX];
#pragma unroll
        for (int e = 0; e < NX * NX; e += 2) {
          const Pair t = rec[kk * KP + e / 2];
          m[e] = t.a, m[e + 1] = t.b;
        }
#pragma unroll
        for (int c = 0; c < NX; c += 2) {
          const Pair t = rec[kk * KP + (NX * NX + c) / 2];
          nx[c] = t.a, nx[c + 1] = t.b;
        }
#pragma unroll
        for (int rr = 0; rr < NX; ++rr) {
#pragma unroll
          for (int c = 0; c < NX; ++c) nx[rr] = fma_t(m[rr * NX + c], al[c], nx[rr]);
        }
#pragma unroll
        for (int c = 0; c < NX; ++c) al[c] = nx[c];
      };
      if (k >= 2) advance(1);
      // one record in registers at a time: with both in flight the allocation of the Arnoldi loop tips into scratch
      __builtin_amdgcn_sched_barrier(0);
      if (k >= 3) advance(2);
      if (item_on(i, only_active)) {
        const int base = (3 - k) * n;
        for (int j = j0; j < n; j += stride) {
          const int s = base + j;
#pragma unroll
          for (int jj = 0; jj < NUL; ++jj) {
            T acc = T(0);
#pragma unroll
            for (int c = 0; c < NX; ++c) acc = fma_t(al[c], Hd[(s * NUL + jj) * HL + c * IPW + i], acc);
            T* po = out + i * P.Lp + s * NU + jj;
            *po = fma_t(acc, sc, *po);
          }
        }
      }
    }
  }

  // The chunk-parallel sweep WITHOUT per-stage scratch, for the plans whose LDS has no room for it (lean, long vectors):
  // the horizon is cut into C = P.cs_chunks (3 or 4) chunks of n = dv/C stages (chunk 0, the LAST part of the horizon,
  // takes the remainder) and walked twice —
  //   pass 1 (no output traffic at all): wave 0, lane (k, i): chunk 0 from the terminal costate — its end value is the
  //          start value of chunk 1 — and chunks 1 .. C-2 from l = 0 with the bias (particular end value p_k); waves 1..:
  //          chunks 1 .. C-2 from the NX unit vectors without the bias (transfer matrix M_k).  The last chunk of the walk
  //          (the first stages) needs neither: nobody continues from its end;
  //   one barrier;
  //   pass 2: wave 0, lane (k, i): l(start of chunk k) = p_(k-1) + M_(k-1) l(start of chunk k-1) (at most two small
  //          products, redundantly per lane), then the chunk for real: o + sc*dF to `out` like the serial sweep.
  // One wave walks 2n instead of dv stages, both times alone at the LDS pipe; the scratch is 4 + (C-2)*22 scalars per
  // instance.  Same arithmetic per stage as the serial sweep; what differs is that l enters a chunk as p + M l_start
  // instead of through the stages before it (rounding level, bounded against the oracle by the tests).
  template <int MODE>
  __device__ __forceinline__ void sweep_costate_2pass(T dtau, const T* xT, T* out, bool only_active) {
    constexpr int NX = M::NX;
    constexpr int REC = Lds::SCAN_REC;
    static_assert(NX % 2 == 0, "boundary records are read in pairs");
    const int C = P.cs_chunks, dv = P.dv, n = dv / C, n0 = dv - (C - 1) * n;  // chunk k >= 1: stages [(C-1-k)n, (C-k)n)
    T* Vec0 = S.scan;              // [i][NX]
    T* Rec = Vec0 + IPW * NX;      // [k-1][i][REC], k = 1 .. C-2
    int tid_o = tid;               // (opaque: see sweep_costate_par)
    asm volatile("" : "+v"(tid_o));
    const int wave = tid_o >> 6, lane = tid_o & 63;
    // ---- pass 1
    if (wave == 0) {
      const int i = lane & (IPW - 1), k = lane >> 4;
      if (k < C - 1 && item_on(i, only_active)) {
        T l[NX];
        costate_run<MODE, false, false>(l, i, k == 0 ? dv - 1 : (C - k) * n - 1, n, dtau, S.R + 2 * i, out + i * P.Lp,
                                        [&](T* l0) {
                                          costate_terminal(l0, xT, i);
                                          if (k != 0) {
#pragma unroll
                                            for (int c = 0; c < NX; ++c) l0[c] = T(0);
                                          }
                                        },
                                        n0 - n, k == 0);
        T* dst = k == 0 ? Vec0 + i * NX : Rec + ((k - 1) * IPW + i) * REC + NX * NX;
#pragma unroll
        for (int c = 0; c < NX; ++c) dst[c] = l[c];
      }
    } else {
      const int q = tid_o - 64, g = q / HL, ql = q - g * HL, i = ql & (IPW - 1), c0 = ql / IPW;  // chunk g + 1, column c0
      if (g < C - 2 && item_on(i, only_active)) {
        T l[NX];
        costate_run<MODE, true, false>(l, i, (C - 1 - g) * n - 1, n, dtau, S.R + 2 * i, out, [&](T* l0) {
#pragma unroll
          for (int c = 0; c < NX; ++c) l0[c] = c == c0 ? T(1) : T(0);
        });
#pragma unroll
        for (int r = 0; r < NX; ++r) Rec[(g * IPW + i) * REC + r * NX + c0] = l[r];
      }
    }
    CGM_STAMP(*this, 16);
    lds_barrier();
    CGM_STAMP(*this, 17);
    // ---- pass 2
    if (wave == 0) {
      const int i = lane & (IPW - 1), k = lane >> 4;
      if (k < C && item_on(i, only_active)) {
        T l[NX];
        costate_run<MODE, false, true>(l, i, k == 0 ? dv - 1 : (C - k) * n - 1, n, dtau, S.R + 2 * i, out + i * P.Lp,
                                       [&](T* l0) {
                                         if (k == 0) {
                                           costate_terminal(l0, xT, i);
                                         } else {
                                           const Pair* v0 = reinterpret_cast<const Pair*>(Vec0 + i * NX);
#pragma unroll
                                           for (int c = 0; c < NX; c += 2) {
                                             const Pair t = v0[c / 2];
                                             l0[c] = t.a, l0[c + 1] = t.b;
                                           }
                                           for (int kk = 1; kk < k; ++kk) {  // l0 <- p_kk + M_kk l0
                                             const Pair* rec = reinterpret_cast<const Pair*>(Rec + ((kk - 1) * IPW + i) * REC);
                                             T m[NX * NX], nx[NX];
#pragma unroll
                                             for (int e = 0; e < NX * NX; e += 2) {
                                               const Pair t = rec[e / 2];
                                               m[e] = t.a, m[e + 1] = t.b;
                                             }
#pragma unroll
                                             for (int c = 0; c < NX; c += 2) {
                                               const Pair t = rec[(NX * NX + c) / 2];
                                               nx[c] = t.a, nx[c + 1] = t.b;
                                             }
#pragma unroll
                                             for (int rr = 0; rr < NX; ++rr) {
#pragma unroll
                                               for (int c = 0; c < NX; ++c) nx[rr] = fma_t(m[rr * NX + c], l0[c], nx[rr]);
                                             }
#pragma unroll
                                             for (int c = 0; c < NX; ++c) l0[c] = nx[c];
                                           }
                                         }
                                       },
                                       n0 - n, k == 0);
      }
    }
  }

  // One complete sweep on the LDS table.  COLLECTIVE: every thread of the block must call it (two workgroup barriers
  // inside); the caller adds the barrier that publishes `out`.  x0c = initial state, component-major LDS [c*IPW + i].
  // `after_sweep` runs on the sweep wave right after its state sweep, i.e. while it would otherwise wait for the other
  // waves to finish the last coefficient chunk (gmres() requests its basis rows there); the barrier that follows
  // orders LDS only so those loads stay in flight during the costate sweep.
  // `idle_work` runs on the other waves BEFORE their first chunk barrier, i.e. while they would wait for the sweep wave
  // to finish the first chunk of stages (gmres() does the previous iteration's Hessenberg column there).
  template <bool PERT, int MODE, class After, class Idle>
  __device__ __forceinline__ void f_eval(const T* x0c, T dtau, T* out, bool only_active, After&& after_sweep,
                                         Idle&& idle_work) {
    if constexpr (IPW * 16 >= 128) {
      if (tid < 64) {
        sweep_state<PERT, true, ALT_X02>(0, x0c, dtau, S.R, S.xT, only_active);
        after_sweep();
      } else {
        idle_work();
        coeffs_chunked<PERT, MODE>(dtau, out, only_active);
      }
      CGM_STAMP(*this, 4);
      lds_barrier();
      CGM_STAMP(*this, 5);
      sweep_costate<MODE>(dtau, S.xT, out, only_active);
    } else {
      idle_work();
      f_eval<PERT, MODE>(x0c, dtau, out, only_active);
      if (tid < 64) after_sweep();
    }
  }
  template <bool PERT, int MODE>
  __device__ __forceinline__ void f_eval(const T* x0c, T dtau, T* out, bool only_active) {
    if constexpr (IPW * 16 >= 128) {
      if (tid < 64)
        sweep_state<PERT, true, ALT_X02>(0, x0c, dtau, S.R, S.xT, only_active);
      else
        coeffs_chunked<PERT, MODE>(dtau, out, only_active);
      CGM_STAMP(*this, 4);
      __syncthreads();
    } else {
      sweep_state<PERT, false>(0, x0c, dtau, S.R, S.xT, only_active);
      __syncthreads();
      CGM_STAMP(*this, 4);
      sweep_coeffs<PERT, MODE>(dtau, S.R, out, only_active);
      __syncthreads();
    }
    CGM_STAMP(*this, 5);
    sweep_costate<MODE>(dtau, S.xT, out, only_active);
  }

  // cgmres.hpp:83-85 on the sweep lanes: x_dxh = x + h*f(x, U0)
  __device__ __forceinline__ void make_xh() {
    if (!sweep_lane) return;
    constexpr int NX = M::NX;
    const int i = tid;
    T x[NX], u0[M::NU], f[NX], tr[M::NC > 0 ? M::NC : 1];
#pragma unroll
    for (int c = 0; c < NX; ++c) x[c] = S.xs[c * IPW + i];
#pragma unroll
    for (int j = 0; j < M::NU; ++j) u0[j] = (LEAN ? S.W : S.U)[i * P.Lp + j];  // (lean: call after publish_U)
    model_dxdt(f, x, u0, tr, i, 0);
#pragma unroll
    for (int c = 0; c < NX; ++c) S.xh[c * IPW + i] = f[c] * P.h + x[c];
  }
  // The three sweeps control() needs before the Arnoldi loop are independent of each other
  //     #1 Fh  = F(U, x_dxh, t+h)              cgmres.hpp:88
  //     #2 b   = (F(U, x, t)(1-zeta h) - Fh)/h   :91-96
  //     #3 Ax0 = (F(U + h dUdt, x_dxh, t+h) - Fh)/h   :99 -> gmres.hpp:33 (only when WITH_AX0; W must hold dUdt)
  // so their state sweeps — the long serial phase — run CONCURRENTLY on waves 0, 1, 2; #1 uses the LDS stage table,
  // #2/#3 park theirs in this workgroup's HBM scratch (written once, read once by the parallel coefficient phase).
  // Results: Fh in S.Fh; b and Ax0 delivered in registers (row layout) because both pass through S.W.
  // COLLECTIVE; needs at least 3 waves (falls back to sequential sweeps otherwise).
  template <bool WITH_AX0>
  __device__ __forceinline__ void preamble(T* bb, T* ax0, const T* dir = nullptr) {  // dir: registers of the W direction
    if constexpr (LEAN) {
      // Lean plan: W is the only row array.  #1 and #2 read the unperturbed U through it (state sweeps concurrently on
      // waves 0/1, then their coefficient/costate phases one after the other, U re-published in between because the
      // results pass through W), #3 follows as an ordinary pipelined sweep on W = U + h*dUdt: one serial sweep more
      // than the full plans, the price of fitting two workgroups on a CU.
      static_assert(IPW * 16 >= 128, "lean preamble: two sweep waves");
      T* tab0 = P.scr + size_t(blockIdx.x) * 2 * Lds::tab_count(P.dv);
      T* xT0 = S.xT, *xT1 = S.xT + M::NX * IPW;
      publish_U();
      __syncthreads();
      make_xh();
      __syncthreads();
      sweep_state<false, false>(0, S.xh, dtau_h, S.R, xT0, false);
      sweep_state<false, false>(64, S.xs, dtau_0, tab0, xT1, false);
      __threadfence_block();
      __syncthreads();  // drains vmcnt: the parked table is complete and visible to the other waves of this CU
      CGM_STAMP(*this, 4);
      sweep_coeffs<false, F_PLAIN>(dtau_h, S.R, S.W, false);  // in place: every entry of W is read (as u) before it is written
      __syncthreads();
      CGM_STAMP(*this, 5);
      sweep_costate<F_PLAIN>(dtau_h, xT0, S.W, false);
      __syncthreads();
      {
        T fh[MAXM];
        lds_to_reg(fh, S.W);
        reg_to_row(P.Fh, P.Lg, fh);
      }
      __threadfence_block();
      __syncthreads();  // F(U,x+hf,t+h) rows visible to the coefficient phases of this CU; W is free again
      publish_U();
      __syncthreads();
      sweep_coeffs<false, F_RHS>(dtau_0, tab0, S.W, false);
      __syncthreads();
      CGM_STAMP(*this, 5);
      sweep_costate<F_RHS>(dtau_0, xT1, S.W, false);
      __syncthreads();
      lds_to_reg(bb, S.W);
      __syncthreads();
      if (WITH_AX0) {
        publish_direction(dir);
        __syncthreads();
        f_eval<true, F_AX>(S.xh, dtau_h, S.W, false);
        __syncthreads();
        lds_to_reg(ax0, S.W);
        __syncthreads();
      }
      return;
    }
    make_xh();
    __syncthreads();
    if constexpr (IPW * 16 >= 192) {
      const size_t tab_n = Lds::tab_count(P.dv);
      T* tab0 = P.scr + size_t(blockIdx.x) * 2 * tab_n;
      T* tab1 = tab0 + tab_n;
      T* xT0 = S.xT, *xT1 = S.xT + M::NX * IPW, *xT2 = S.xT + 2 * M::NX * IPW;
      sweep_state<false, false>(0, S.xh, dtau_h, S.R, xT0, false);
      sweep_state<false, false>(64, S.xs, dtau_0, tab0, xT1, false);
      if (WITH_AX0) sweep_state<true, false>(128, S.xh, dtau_h, tab1, xT2, false);
      __threadfence_block();
      __syncthreads();  // drains vmcnt: the HBM tables are complete and visible to the other waves of this CU
      CGM_STAMP(*this, 4);
      sweep_coeffs<false, F_PLAIN>(dtau_h, S.R, S.Fh, false);  // (S.Fh is S.W itself when fh_hbm)
      __syncthreads();
      CGM_STAMP(*this, 5);
      sweep_costate<F_PLAIN>(dtau_h, xT0, S.Fh, false);
      __syncthreads();
      if (fh_hbm()) {
        // Fh went through W: move it to its HBM row and put the direction of A*x0 back (its first copy served the
        // concurrent state sweep #3; the coefficient phase of #3 reads it again)
        T fh[MAXM];
        lds_to_reg(fh, S.W);
        reg_to_row(P.Fh, P.Lg, fh);
        __threadfence_block();
        __syncthreads();  // drains vmcnt: the rows are visible to the other waves of this CU
        if (WITH_AX0) publish_direction(dir);
        __syncthreads();
      }
      if (WITH_AX0) {
        sweep_coeffs<true, F_AX>(dtau_h, tab1, S.W, false);
        __syncthreads();
        CGM_STAMP(*this, 5);
        sweep_costate<F_AX>(dtau_h, xT2, S.W, false);
        __syncthreads();
        lds_to_reg(ax0, S.W);
        __syncthreads();
      }
      sweep_coeffs<false, F_RHS>(dtau_0, tab0, S.W, false);
      __syncthreads();
      CGM_STAMP(*this, 5);
      sweep_costate<F_RHS>(dtau_0, xT1, S.W, false);
      __syncthreads();
      lds_to_reg(bb, S.W);
      __syncthreads();
    } else {
      f_eval<false, F_PLAIN>(S.xh, dtau_h, S.Fh, false);
      __syncthreads();
      if (WITH_AX0) {
        f_eval<true, F_AX>(S.xh, dtau_h, S.W, false);
        __syncthreads();
        lds_to_reg(ax0, S.W);
        __syncthreads();
      }
      f_eval<false, F_RHS>(S.xs, dtau_0, S.W, false);
      __syncthreads();
      lds_to_reg(bb, S.W);
      __syncthreads();
    }
  }
  // Ax_func in place on W (cgmres.hpp:164-175).  Collective; ends with W published.
  template <class After, class Idle>
  __device__ __forceinline__ void ax(bool only_active, After&& after_sweep, Idle&& idle_work) {
    CGM_STAMP(*this, 3);
    f_eval<true, F_AX>(S.xh, dtau_h, S.W, only_active, after_sweep, idle_work);
    __syncthreads();
    CGM_STAMP(*this, 6);
  }
  __device__ __forceinline__ void ax(bool only_active) {
    ax(only_active, [] {}, [] {});
  }

  // ---- NWT = 1: the perturbed state sweeps as Newton on the whole trajectory, row-parallel ----------------------------
  // The serial state sweep (sweep_state) keeps ONE wave busy for ~12 k cycles per mat-vec while the others mostly wait
  // (profiles/r04_wg_serial_instruction_counters.json: 23 % of the VALU issue slots used).  Here every row of 16 lanes solves its
  // own instance's recurrence cgmres.hpp:132-140 for ALL stages at once, on all four waves:
  //   * lane r of the row owns the stages 4r .. 4r+3 (dv <= 63);
  //   * x0 / x2 (model.hpp:38,40: linear, no trig) as the DIFFERENCE to the unperturbed trajectory — a scan of the
  //     control differences with powers of the constant 2 x 2 matrix, four in-row DPP steps;
  //   * x1 / x3 by Newton's method on the stage equations, started from the unperturbed trajectory (the direction is
  //     scaled by h = 1e-3..: the perturbed trajectory is close): per iteration the stage residuals, the local
  //     composition of the lane's four linearised stage maps, one in-row scan of 2 x 2 affine maps (wave_scan.hip.h),
  //     and the local expansion; sin / cos carried from iteration to iteration by rotation (fresh evaluation when an
  //     angle moved too far).  Quadratic convergence: the loop ends when a correction is below 1e-10 relative, i.e. the next one would
  //     be below rounding; measured two iterations per mat-vec.
  // The base trajectory (x0, x1, x2, sin/cos per owned stage: 28 values per lane) is taken from the stage table of the tick's
  // first unperturbed sweep (preamble), which stays the serial quad sweep.  The result differs from the serial sweep's
  // by rounding only (tests/test_gpu_parity.py bounds it against the oracle like every other mapping).
  static constexpr bool ROW_NEWTON = NWT == 1;
  // (Newton on an explicit recurrence cannot fail to terminate: iteration n makes stage n exact — its predecessor is —
  // so dv iterations reproduce the serial sweep whatever the start; the bound below is never reached by the test on the
  // correction, it only has to be >= dv.  Two iterations is what the mat-vecs of a tick take.)
  static constexpr int SPL = 4, NEWTON_MAX = 64;
  struct RowBase {
    T x0[SPL], x1[SPL], x2[SPL], sd[SPL], cd[SPL], s1[SPL], c1[SPL];
  };
  static constexpr int NBASE = 7;  // arrays of RowBase
  struct NoBase {};
  std::conditional_t<ROW_NEWTON, RowBase, NoBase> nb;
  mutable bool row_moved = true;  // the published direction changed at least one control of this row (publish_direction)
  // During the Arnoldi loop the base lives in LDS — in the stage table, which only the preamble's state sweeps use, in
  // the scratch of the serial-sweep kernel's costate scan, which this kernel does not use at all, and behind everything
  // else where those two are too small (ctx_wg places the arrays: WgParams::base_off) — as pairs [q / 2][thread] per
  // array: every lane reads back
  // exactly what it wrote (no barrier), 16 bytes per access, conflict-free.  Call after the preamble's last barrier.
  __device__ __forceinline__ Pair* base_pairs(int k, int thread) const {
    return reinterpret_cast<Pair*>(S.lds0 + P.base_off[k]) + thread;
  }
  static constexpr size_t base_array_bytes() { return size_t(SPL) * IPW * 16 * sizeof(T); }  // one of the NBASE arrays
  __device__ __forceinline__ void store_base() {
    const T* src[NBASE] = {nb.x0, nb.x1, nb.x2, nb.sd, nb.cd, nb.s1, nb.c1};
#pragma unroll
    for (int k = 0; k < NBASE; ++k) {
      Pair* d = base_pairs(k, tid);
#pragma unroll
      for (int q = 0; q < SPL; q += 2) d[(q / 2) * IPW * 16] = Pair{src[k][q], src[k][q + 1]};
    }
    inv_dtau_nb = T(1) / dtau_h;
  }
  __device__ __forceinline__ void load_base(int k, T* dst, int thread) const {
    const Pair* d = base_pairs(k, thread);
#pragma unroll
    for (int q = 0; q < SPL; q += 2) {
      const Pair t = d[(q / 2) * IPW * 16];
      dst[q] = t.a, dst[q + 1] = t.b;
    }
  }
  T inv_dtau_nb = T(0);
  // sin / cos of (base angle + dl) from the base pair, |dl| <= sqrt(rot_zmax)
  __device__ __forceinline__ void rotate(T sb, T cb, T dl, T* sn_out, T* cs_out) const {
    constexpr int NRS = decltype(mc)::NRS, NRC = decltype(mc)::NRC;
    const T z = dl * dl;
    T ps = mc.rot_sin(NRS - 1), pc = mc.rot_cos(NRC - 1);
#pragma unroll
    for (int i = NRS - 2; i >= 0; --i) ps = fma_t(z, ps, mc.rot_sin(i));
#pragma unroll
    for (int i = NRC - 2; i >= 0; --i) pc = fma_t(z, pc, mc.rot_cos(i));
    const T sn = fma_t(z * dl, ps, dl);  // sin dl
    const T cm1 = z * pc;                // cos dl - 1
    *sn_out = sb + fma_t(sb, cm1, cb * sn);
    *cs_out = cb + fma_t(cb, cm1, -(sb * sn));
  }
  // Costate recurrence (cgmres.hpp:145-153) and dH/du (:156-161) of the row's instance from the states x(s) and trig
  // values of the owned stages (x of "stage dv" = the terminal state), in registers:
  //   MODE F_PLAIN: out = F        F_RHS: out = (F*(1-zeta h) - Fh)/h  (:91-96)        F_AX: out = (F - Fh)/h  (:173-174)
  // PendulumDev::costate_step is affine in the costate with (l1, l3) closed in themselves, l0 a running sum over l3 and
  // l2 a geometric recurrence over l0 / l3 — three scans DOWN the row (partner = the lane above), each as local fold,
  // in-row scan, local expansion.  The terminal costate (cgmres.hpp:143) is the starting value of the lane that holds
  // stage dv (nothing above it but identities).  urow: the row of controls the stages were run with (U, or W = U + h d;
  // may be `out`'s row: every control is read before the first result is stored).
  template <int MODE>
  __device__ __forceinline__ void row_costate(const T* x0f, const T* y1, const T* x2f, const T* y3, const T* sd, const T* cd,
                                              const T* c1, const T* urow, T dtau, T* out, bool run, int tid_o,
                                              const T* u0_have = nullptr) {  // u0_have: the stages' first controls, if the caller holds them
    constexpr int NU = M::NU, NP = M::NP, NBW = M::NBW;
    static_assert(NU == 3 && NP == 2 && M::NUL == 1, "written for the pendulum's stage");
    const int r = tid_o & 15, inst = tid_o >> 4;
    const int dv = P.dv, s_0 = SPL * r;
    const T ee = -dtau * M::C22, a = T(1) - dtau * M::As;
    const T sc_phi = MODE == F_RHS ? P.one_m_zh : T(1.0);
    const T sc = MODE == F_PLAIN ? T(1.0) : (MODE == F_RHS ? P.one_m_zh * P.inv_h : P.inv_h);
    bool tr[SPL];
    T dq[SPL], eq[SPL];
#pragma unroll
    for (int q = 0; q < SPL; ++q) {
      tr[q] = s_0 + q < dv;
      dq[q] = tr[q] ? dtau : T(0), eq[q] = tr[q] ? ee : T(0);
    }
    const int q_term = dv - s_0;
    const bool has_term = q_term >= 0 && q_term < SPL;
    T bw[SPL][NBW], phi0[SPL];
    T lT[M::NX] = {T(0), T(0), T(0), T(0)};
    {
      T u0[SPL], u1[SPL], u2[SPL], pp[SPL][NP], fh[SPL][NU];
#pragma unroll
      for (int q = 0; q < SPL; ++q) {  // every LDS operand first (addresses clamped into the horizon: no branch, one wait)
        const int s = s_0 + q, sk = s < dv ? s : dv - 1, sp = s < dv ? s : dv;
        u0[q] = u0_have ? u0_have[q] : urow[sk * NU];
        u1[q] = urow[sk * NU + 1], u2[q] = urow[sk * NU + 2];
#pragma unroll
        for (int j = 0; j < NP; ++j) pp[q][j] = get_p(inst, sp * NP + j);
#pragma unroll
        for (int j = 0; j < NU; ++j) fh[q][j] = MODE == F_PLAIN ? T(0) : S.Fh[inst * P.Lp + sk * NU + j];
      }
#pragma unroll
      for (int q = 0; q < SPL; ++q) {
        const int s = s_0 + q;
        const T x[M::NX] = {x0f[q], y1[q], x2f[q], y3[q]};
        const T tg[3] = {sd[q], cd[q], c1[q]}, u[NU] = {u0[q], u1[q], u2[q]};
        T phi[NU], bq[NBW];
        // (with the stage's own step size: every coefficient that enters the recurrences carries it as a factor and is
        // zero beyond the horizon; bw[3], which does not, is only used where the stage exists)
        M::stage_coeffs(bq, phi, x, u, pp[q], tg, dq[q]);
#pragma unroll
        for (int cc = 0; cc < NBW; ++cc) bw[q][cc] = bq[cc];
        phi0[q] = MODE == F_PLAIN ? phi[0] : (phi[0] * sc_phi - fh[q][0]) * P.inv_h;
        if (run && tr[q]) {
          out[inst * P.Lp + s * NU + 1] = MODE == F_PLAIN ? phi[1] : (phi[1] * sc_phi - fh[q][1]) * P.inv_h;
          out[inst * P.Lp + s * NU + 2] = MODE == F_PLAIN ? phi[2] : (phi[2] * sc_phi - fh[q][2]) * P.inv_h;
        }
        if (q == q_term) M::dPhidx(lT, x, pp[q]);
      }
    }
    CGM_STAMP(*this, 24);
    T L0[SPL], L2[SPL], L3[SPL];  // costate components ENTERING the owned stages (lambda of stage s + 1)
    {
      // (l1, l3): n1 = l1 + bw1 l3 + bw5,  n3 = l3 + dq l1 + eq l3; the lane with stage dv starts from the constant map
      T D[4] = {has_term ? T(-1) : T(0), T(0), T(0), has_term ? T(-1) : T(0)};
      T c[2] = {has_term ? lT[1] : T(0), has_term ? lT[3] : T(0)};
#pragma unroll
      for (int q = SPL - 1; q >= 0; --q) {
        const T b1 = bw[q][1];
        const T n00 = fma_t(b1, D[2], D[0]);
        const T n01 = fma_t(b1, D[3], D[1] + b1);
        const T n10 = fma_t(eq[q], D[2], fma_t(dq[q], D[0], D[2] + dq[q]));
        const T n11 = fma_t(eq[q], D[3], fma_t(dq[q], D[1], D[3] + eq[q]));
        const T m0 = fma_t(b1, c[1], c[0] + bw[q][5]);
        const T m1 = fma_t(eq[q], c[1], fma_t(dq[q], c[0], c[1]));
        D[0] = n00, D[1] = n01, D[2] = n10, D[3] = n11, c[0] = m0, c[1] = m1;
      }
      aff2_step_vec<0, true>(c, D), aff2_step_mat<0, true>(D);
      aff2_step_vec<1, true>(c, D), aff2_step_mat<1, true>(D);
      aff2_step_vec<2, true>(c, D), aff2_step_mat<2, true>(D);
      aff2_step_vec<3, true>(c, D);
      // costate entering the lane's last stage (the lanes above the one with stage dv deliver zeros)
      T l1 = scan_partner<0, true>(c[0]), l3 = scan_partner<0, true>(c[1]);
      l1 = has_term ? lT[1] : l1, l3 = has_term ? lT[3] : l3;
      T l0s = has_term ? lT[0] : T(0);  // l0 relative to the lane's entry: n0 = l0 + (bw4 + bw0 l3)
#pragma unroll
      for (int q = SPL - 1; q >= 0; --q) {
        L3[q] = l3, L0[q] = l0s;
        l0s += fma_t(bw[q][0], l3, bw[q][4]);
        const T n1 = fma_t(bw[q][1], l3, l1 + bw[q][5]);
        const T n3 = fma_t(eq[q], l3, fma_t(dq[q], l1, l3));
        l1 = n1, l3 = n3;
      }
      T t0 = l0s;
      t0 += scan_partner<0, true>(t0);
      t0 += scan_partner<1, true>(t0);
      t0 += scan_partner<2, true>(t0);
      t0 += scan_partner<3, true>(t0);
      const T in0 = scan_partner<0, true>(t0);
      // l2: n2 = aq l2 + (bw2 l3 + dq l0), aq = 1 - dq As
      T e2 = has_term ? lT[2] : T(0);
#pragma unroll
      for (int q = SPL - 1; q >= 0; --q) {
        L0[q] += in0;
        e2 = fma_t(fma_t(-M::As, dq[q], T(1)), e2, fma_t(bw[q][2], L3[q], dq[q] * L0[q]));
      }
      {
        const T a2 = a * a;
        T m = a2 * a2;  // (a lane with fewer than four stages has nothing but zeros above it: its multiplier is not used)
        e2 = fma_t(m, scan_partner<0, true>(e2), e2), m = m * m;
        e2 = fma_t(m, scan_partner<1, true>(e2), e2), m = m * m;
        e2 = fma_t(m, scan_partner<2, true>(e2), e2), m = m * m;
        e2 = fma_t(m, scan_partner<3, true>(e2), e2);
      }
      T l2 = scan_partner<0, true>(e2);
      l2 = has_term ? lT[2] : l2;
#pragma unroll
      for (int q = SPL - 1; q >= 0; --q) {
        L2[q] = l2;
        l2 = fma_t(fma_t(-M::As, dq[q], T(1)), l2, fma_t(bw[q][2], L3[q], dq[q] * L0[q]));
      }
    }
    if (run) {
#pragma unroll
      for (int q = 0; q < SPL; ++q) {
        if (tr[q]) {
          const T dF = fma_t(bw[q][3], L3[q], L2[q] * M::Bs);  // B^T lambda, costate_step
          out[inst * P.Lp + (s_0 + q) * NU] = fma_t(dF, sc, phi0[q]);
        }
      }
    }
    CGM_STAMP(*this, 23);
  }

  // ---- NWT = 2: a model whose state equation is AFFINE in x for given controls (semiactive_damper/model.hpp:36-39:
  //      x0' = x1, x1' = a x0 + b u0 x1).  One evaluation of F (cgmres.hpp:113-162) is then two scans over the row and
  //      nothing else — the state recurrence :132-140 as a scan of 2 x 2 maps UP the row (lane 0 starts from the constant
  //      map "x(0)"), the costate recurrence :145-153 as one DOWN the row (the lane with stage dv starts from the terminal
  //      costate, :143), four stages per lane folded locally on either side — exact up to rounding, no iteration, no stage
  //      table, no serial sweep anywhere in the tick.
  //      MODE as in row_costate; x0c: the initial state [c*IPW + i]; urow: the row of controls (may be `out`'s row).
  template <int MODE>
  __device__ __forceinline__ void row_affine_sweep(const T* x0c, T dtau, const T* urow, T* out, bool run) {
    constexpr int NU = M::NU, NBW = M::NBW;
    static_assert(M::NX == 2 && NU == 3 && M::NP == 0 && M::NUL == 1 && M::NC == 0 && NBW == 4,
                  "written for the semi-active damper's stage");
    int tid_o = tid;  // (see row_newton_sweep)
    asm volatile("" : "+v"(tid_o));
    const int r = tid_o & 15, inst = tid_o >> 4;
    const int dv = P.dv, s_0 = SPL * r;
    const T sc_phi = MODE == F_RHS ? P.one_m_zh : T(1.0);
    const T sc = MODE == F_PLAIN ? T(1.0) : (MODE == F_RHS ? P.one_m_zh * P.inv_h : P.inv_h);
    bool tr[SPL];
    T dq[SPL], u0[SPL], u1[SPL], u2[SPL], fh[SPL][NU];
#pragma unroll
    for (int q = 0; q < SPL; ++q) {
      const int s = s_0 + q, sk = s < dv ? s : dv - 1;
      tr[q] = s < dv;
      dq[q] = tr[q] ? dtau : T(0);  // (stages beyond the horizon: identity maps)
      u0[q] = urow[sk * NU], u1[q] = urow[sk * NU + 1], u2[q] = urow[sk * NU + 2];
#pragma unroll
      for (int j = 0; j < NU; ++j) fh[q][j] = MODE == F_PLAIN ? T(0) : S.Fh[inst * P.Lp + sk * NU + j];
    }
    const int q_term = dv - s_0;
    const bool has_term = q_term >= 0 && q_term < SPL, first = r == 0;
    const T xi0 = x0c[0 * IPW + inst], xi1 = x0c[1 * IPW + inst];
    // ---- states: x(s+1) = x(s) + [[0, dq], [dq a, dq b u0]] x(s)
    T X0[SPL], X1[SPL], d2[SPL], d3[SPL];
    {
      T D[4] = {first ? T(-1) : T(0), T(0), T(0), first ? T(-1) : T(0)}, c[2] = {first ? xi0 : T(0), first ? xi1 : T(0)};
#pragma unroll
      for (int q = 0; q < SPL; ++q) {
        d2[q] = dq[q] * M::a, d3[q] = dq[q] * (M::b * u0[q]);
        const T n00 = fma_t(dq[q], D[2], D[0]);
        const T n01 = fma_t(dq[q], D[3], D[1] + dq[q]);
        const T n10 = fma_t(d3[q], D[2], fma_t(d2[q], D[0], D[2] + d2[q]));
        const T n11 = fma_t(d3[q], D[3], fma_t(d2[q], D[1], D[3] + d3[q]));
        const T m0 = fma_t(dq[q], c[1], c[0]);
        const T m1 = fma_t(d3[q], c[1], fma_t(d2[q], c[0], c[1]));
        D[0] = n00, D[1] = n01, D[2] = n10, D[3] = n11, c[0] = m0, c[1] = m1;
      }
      aff2_step_vec<0>(c, D), aff2_step_mat<0>(D);
      aff2_step_vec<1>(c, D), aff2_step_mat<1>(D);
      aff2_step_vec<2>(c, D), aff2_step_mat<2>(D);
      aff2_step_vec<3>(c, D);
      T e0 = scan_partner<0>(c[0]), e1 = scan_partner<0>(c[1]);  // the state the lane's first stage starts from
      e0 = first ? xi0 : e0, e1 = first ? xi1 : e1;
#pragma unroll
      for (int q = 0; q < SPL; ++q) {
        X0[q] = e0, X1[q] = e1;
        const T n0 = fma_t(dq[q], e1, e0);
        const T n1 = fma_t(d3[q], e1, fma_t(d2[q], e0, e1));
        e0 = n0, e1 = n1;
      }
    }
    // ---- stage coefficients, costate: n0 = l0 + bw2 + dq a l1,  n1 = l1 + bw3 + dq l0 + bw0 l1  (costate_step)
    T bw[SPL][NBW], phi0[SPL], L1[SPL];
    T lT[M::NX] = {T(0), T(0)};
#pragma unroll
    for (int q = 0; q < SPL; ++q) {
      const int s = s_0 + q;
      const T x[M::NX] = {X0[q], X1[q]}, u[NU] = {u0[q], u1[q], u2[q]};
      T phi[NU];
      M::stage_coeffs(bw[q], phi, x, u, nullptr, nullptr, dq[q]);  // (bw[1], which carries no step size, is only used where the stage exists)
      phi0[q] = MODE == F_PLAIN ? phi[0] : (phi[0] * sc_phi - fh[q][0]) * P.inv_h;
      if (run && tr[q]) {
        out[inst * P.Lp + s * NU + 1] = MODE == F_PLAIN ? phi[1] : (phi[1] * sc_phi - fh[q][1]) * P.inv_h;
        out[inst * P.Lp + s * NU + 2] = MODE == F_PLAIN ? phi[2] : (phi[2] * sc_phi - fh[q][2]) * P.inv_h;
      }
      if (q == q_term) M::dPhidx(lT, x, nullptr);
    }
    {
      T D[4] = {has_term ? T(-1) : T(0), T(0), T(0), has_term ? T(-1) : T(0)};
      T c[2] = {has_term ? lT[0] : T(0), has_term ? lT[1] : T(0)};
#pragma unroll
      for (int q = SPL - 1; q >= 0; --q) {
        const T b0 = bw[q][0];
        const T n00 = fma_t(d2[q], D[2], D[0]);
        const T n01 = fma_t(d2[q], D[3], D[1] + d2[q]);
        const T n10 = fma_t(b0, D[2], fma_t(dq[q], D[0], D[2] + dq[q]));
        const T n11 = fma_t(b0, D[3], fma_t(dq[q], D[1], D[3] + b0));
        const T m0 = fma_t(d2[q], c[1], c[0] + bw[q][2]);
        const T m1 = fma_t(b0, c[1], fma_t(dq[q], c[0], c[1] + bw[q][3]));
        D[0] = n00, D[1] = n01, D[2] = n10, D[3] = n11, c[0] = m0, c[1] = m1;
      }
      aff2_step_vec<0, true>(c, D), aff2_step_mat<0, true>(D);
      aff2_step_vec<1, true>(c, D), aff2_step_mat<1, true>(D);
      aff2_step_vec<2, true>(c, D), aff2_step_mat<2, true>(D);
      aff2_step_vec<3, true>(c, D);
      T l0 = scan_partner<0, true>(c[0]), l1 = scan_partner<0, true>(c[1]);  // costate entering the lane's last stage
      l0 = has_term ? lT[0] : l0, l1 = has_term ? lT[1] : l1;
#pragma unroll
      for (int q = SPL - 1; q >= 0; --q) {
        L1[q] = l1;
        const T n0 = fma_t(d2[q], l1, l0 + bw[q][2]);
        const T n1 = fma_t(bw[q][0], l1, fma_t(dq[q], l0, l1 + bw[q][3]));
        l0 = n0, l1 = n1;
      }
    }
    if (run) {
#pragma unroll
      for (int q = 0; q < SPL; ++q) {
        if (tr[q]) out[inst * P.Lp + (s_0 + q) * NU] = fma_t(bw[q][1] * L1[q], sc, phi0[q]);  // B^T lambda, costate_step
      }
    }
  }
  // the preamble of a tick with it: the three evaluations of cgmres.hpp:88-99 one after the other, every row for itself
  __device__ __forceinline__ void preamble_affine(T* bb, T* ax0) {
    make_xh();
    __syncthreads();  // (x + h f of every instance is formed by the sweep lanes of wave 0)
    int tid_o = tid;
    asm volatile("" : "+v"(tid_o));
    const int inst = tid_o >> 4;
    const T* Urow = S.U + inst * P.Lp;
    row_affine_sweep<F_PLAIN>(S.xh, dtau_h, Urow, S.Fh, valid);
    row_affine_sweep<F_AX>(S.xh, dtau_h, S.W + inst * P.Lp, S.W, valid);
    lds_to_reg(ax0, S.W);
    row_affine_sweep<F_RHS>(S.xs, dtau_0, Urow, S.W, valid);
    lds_to_reg(bb, S.W);
  }

  // The preamble of a tick (see preamble()) for the row-parallel kernel: the three state sweeps stay the serial quad
  // sweeps, side by side on waves 0-2 (#1 leaves its stage table in LDS, #2 / #3 park theirs in HBM); everything behind
  // them — stage coefficients, costate recurrence, dH/du — every row does for itself in registers (row_costate), one
  // sweep after the other with no workgroup barrier in between: F(U,x+hf,t+h) -> S.Fh, A*dUdt and b through the row
  // of W into registers.  The stage states of sweep #1, with freshly evaluated sin / cos, become the base of the
  // Newton sweeps (nb; moved to LDS by store_base once every lane is done with the stage table).
  __device__ __forceinline__ void preamble_rows(T* bb, T* ax0) {
    make_xh();
    __syncthreads();
    const size_t tab_n = Lds::tab_count(P.dv);
    T* tab0 = P.scr + size_t(blockIdx.x) * 2 * tab_n;
    T* tab1 = tab0 + tab_n;
    T* xT0 = S.xT, *xT1 = S.xT + M::NX * IPW, *xT2 = S.xT + 2 * M::NX * IPW;
    sweep_state<false, false>(0, S.xh, dtau_h, S.R, xT0, false);
    sweep_state<false, false>(64, S.xs, dtau_0, tab0, xT1, false);
    sweep_state<true, false>(128, S.xh, dtau_h, tab1, xT2, false);
    __threadfence_block();
    __syncthreads();  // drains vmcnt: the HBM tables are complete and visible to the other waves of this CU
    CGM_STAMP(*this, 4);
    int tid_o = tid;
    asm volatile("" : "+v"(tid_o));
    const int r = tid_o & 15, inst = tid_o >> 4, dv = P.dv;
    struct Stages {
      T x0[SPL], x1[SPL], x2[SPL], x3[SPL], sd[SPL], cd[SPL], c1[SPL];
    };
    // x and trig of the owned stages from a stage table (slot map: PendulumDev::quad_stage), x(dv) from xT
    auto read_stages = [&](Stages& g, const T* tab, const T* xT) {
#pragma unroll
      for (int q = 0; q < SPL; ++q) {
        const int s = SPL * r + q, sk = s < dv ? s : 0;
        const T* e = tab + tab_off(sk) + inst;
        g.x0[q] = e[M::QSLOT_XA * IPW], g.x1[q] = e[1 * IPW], g.x2[q] = e[M::QSLOT_XB * IPW], g.x3[q] = T(0);
        g.sd[q] = e[M::TRIG_SLOT0 * IPW], g.cd[q] = e[(M::TRIG_SLOT0 + 1) * IPW], g.c1[q] = e[(M::TRIG_SLOT0 + 2) * IPW];
      }
#pragma unroll
      for (int q = 0; q < SPL; ++q) {
        const int s = SPL * r + q;
        if (s == dv) g.x0[q] = xT[0 * IPW + inst], g.x1[q] = xT[1 * IPW + inst], g.x2[q] = xT[2 * IPW + inst], g.x3[q] = xT[3 * IPW + inst];
        if (s > dv) g.x0[q] = T(0), g.x1[q] = T(0), g.x2[q] = T(0);
      }
    };
    const T* Urow = S.U + inst * P.Lp;
    {
      Stages g;
      read_stages(g, S.R, xT0);
#pragma unroll
      for (int q = 0; q < SPL; ++q) {
        nb.x0[q] = g.x0[q], nb.x1[q] = g.x1[q], nb.x2[q] = g.x2[q];
        mc.sincos_pair(g.x0[q] - g.x1[q], g.x1[q], &nb.sd[q], &nb.cd[q], &nb.s1[q], &nb.c1[q]);
      }
      row_costate<F_PLAIN>(g.x0, g.x1, g.x2, g.x3, g.sd, g.cd, g.c1, Urow, dtau_h, S.Fh, valid, tid_o);
    }
    {
      Stages g;
      read_stages(g, tab1, xT2);
      row_costate<F_AX>(g.x0, g.x1, g.x2, g.x3, g.sd, g.cd, g.c1, S.W + inst * P.Lp, dtau_h, S.W, valid, tid_o);
      lds_to_reg(ax0, S.W);
    }
    {
      Stages g;
      read_stages(g, tab0, xT1);
      row_costate<F_RHS>(g.x0, g.x1, g.x2, g.x3, g.sd, g.cd, g.c1, Urow, dtau_0, S.W, valid, tid_o);
      lds_to_reg(bb, S.W);
    }
    __syncthreads();  // every lane is done with the stage table: the base may move in (store_base)
  }

  // F(U + h d, x + h f, t + h) - F(U, x + h f, t + h), / h, for the row's instance (W holds U + h d on entry, the
  // result on exit): state recurrence, costate recurrence and dH/du all in the registers of the row's 16 lanes — no
  // stage table, no workgroup barrier; the only LDS traffic is the row's own operands (W, U, F(U,x+hf,t+h), ptau).
  // COLLECTIVE over the wave (wave-uniform branches on __any).
  // Stages outside the horizon (s >= dv; the last lanes of the row) are IDENTITY maps by construction — their step
  // sizes and coefficients are zeroed once per sweep — so the folds and expansions below carry no per-stage selects.
  template <int MODE, class Mid>
  __device__ __forceinline__ void row_newton_sweep(T dtau, T* out, bool run, Mid&& mid) {
    constexpr int NU = M::NU;
    static_assert(MODE == F_AX, "only the mat-vec of the Arnoldi loop");
    // (everything below derives from the thread index and is invariant over sweeps and ticks: left visible, the compiler
    // hoists it to the top of the kernel and keeps dozens of values alive across the Gram-Schmidt rounds — in scratch,
    // reloaded behind s_waitcnt vmcnt(0), i.e. behind the basis rows in flight)
    int tid_o = tid;
    asm volatile("" : "+v"(tid_o));
    const int r = tid_o & 15, inst = tid_o >> 4;
    const int dv = P.dv, s_0 = SPL * r;
    const T* Wr = S.W + inst * P.Lp;
    const T* Ur = S.U + inst * P.Lp;
    const T ee = -dtau * M::C22;
    bool tr[SPL];            // stage s_0 + q has a transition (s < dv)
    T dq[SPL], eq[SPL];      // its step sizes: dtau, -dtau C22, or 0
    T u0[SPL], du[SPL];
#pragma unroll
    for (int q = 0; q < SPL; ++q) {
      tr[q] = s_0 + q < dv;
      dq[q] = tr[q] ? dtau : T(0), eq[q] = tr[q] ? ee : T(0);
      const int e = tr[q] ? (s_0 + q) * NU : 0;
      u0[q] = Wr[e];
      du[q] = tr[q] ? u0[q] - Ur[e] : T(0);
    }
    // (the base trajectory is requested with the controls: one LDS round trip for both)
    T b0[SPL], b2[SPL], y1[SPL], sd[SPL], cd[SPL], s1[SPL], c1[SPL];
    load_base(0, b0, tid_o), load_base(2, b2, tid_o), load_base(1, y1, tid_o);
    load_base(3, sd, tid_o), load_base(4, cd, tid_o), load_base(5, s1, tid_o), load_base(6, c1, tid_o);
    // ---- x0, x2: difference to the base trajectory (zero initial difference)
    const T a = T(1) - dtau * M::As, bs = dtau * M::Bs;
    T dx0[SPL], dx2[SPL];
    {
      T e0 = T(0), e2 = T(0);
#pragma unroll
      for (int q = 0; q < SPL; ++q) {
        const T n0 = fma_t(dtau, e2, e0);
        e2 = fma_t(a, e2, bs * du[q]);
        e0 = n0;
      }
      const T a2 = a * a;
      T m11 = a2 * a2, m01 = dtau * ((T(1) + a) + (a2 + a * a2));  // the lane's four stages: [[1, m01], [0, m11]]
      auto step = [&](auto tc) {
        constexpr int t = decltype(tc)::value;
        const T p0 = scan_partner<t>(e0), p2 = scan_partner<t>(e2);
        e0 = (e0 + p0) + m01 * p2;
        e2 = fma_t(m11, p2, e2);
        m01 = fma_t(m01, m11, m01);
        m11 = m11 * m11;
      };
      step(std::integral_constant<int, 0>{}), step(std::integral_constant<int, 1>{});
      step(std::integral_constant<int, 2>{}), step(std::integral_constant<int, 3>{});
      T x0 = scan_partner<0>(e0), x2 = scan_partner<0>(e2);  // the state the lane's first stage starts from
#pragma unroll
      for (int q = 0; q < SPL; ++q) {
        dx0[q] = x0, dx2[q] = x2;
        const T n0 = fma_t(dtau, x2, x0);
        x2 = fma_t(a, x2, bs * du[q]);
        x0 = n0;
      }
    }
    CGM_STAMP(*this, 21);
    // ---- x1, x3: Newton.  The trig values are carried from iteration to iteration by rotation: first by the change of
    //      x0 (y1 starts at the base value), then by the corrections of y1.
    T x0f[SPL], x2f[SPL], Pq[SPL], Qq[SPL], y3[SPL];
    {
#pragma unroll
      for (int q = 0; q < SPL; ++q) {
        x0f[q] = b0[q] + dx0[q], x2f[q] = b2[q] + dx2[q];
        Pq[q] = M::A32 * x2f[q] * x2f[q];
        Qq[q] = M::A32a * x2f[q] - M::A32b * u0[q];
      }
      // x3 is not part of the base: x3(s) = (x1(s+1) - x1(s)) / dtau of the base trajectory (a starting value only)
      const T yb = dpp_row_zero_fill<0x101>(y1[0]);  // row_shl:1
#pragma unroll
      for (int q = 0; q < SPL; ++q)
        y3[q] = tr[q] ? ((q + 1 < SPL ? y1[q + 1 < SPL ? q + 1 : 0] : yb) - y1[q]) * inv_dtau_nb : T(0);
    }
    T ad[SPL], a1[SPL];  // angle increments since the trig values were last brought up to date: of x0 - x1, of x1
#pragma unroll
    for (int q = 0; q < SPL; ++q) ad[q] = dx0[q], a1[q] = T(0);
    const T rmax = T(0.9) * sqrt_t<T>(T(decltype(mc)::rot_zmax));
    for (int it = 0; it < NEWTON_MAX; ++it) {
      T jq[SPL], c0q[SPL], c1q[SPL];
      T amax = T(0);  // (stages outside the horizon carry harmless finite values)
#pragma unroll
      for (int q = 0; q < SPL; ++q) amax = __builtin_fmax(amax, __builtin_fmax(abs_t(ad[q]), abs_t(a1[q])));
      if (__builtin_expect((P.wave_dbg & 1) || __any(run && !(amax <= rmax)), 0)) {
#pragma unroll
        for (int q = 0; q < SPL; ++q) mc.sincos_pair(x0f[q] - y1[q], y1[q], &sd[q], &cd[q], &s1[q], &c1[q]);
      } else {
#pragma unroll
        for (int q = 0; q < SPL; ++q) rotate(sd[q], cd[q], ad[q], &sd[q], &cd[q]);
        if (it > 0) {
#pragma unroll
          for (int q = 0; q < SPL; ++q) rotate(s1[q], c1[q], a1[q], &s1[q], &c1[q]);
        }
      }
      // defects of the stage equations (the stage after the lane's last one belongs to the next lane) and the local
      // composition of the linearised stage maps  D -> D + [[0, dq], [j, eq]] (I + D)  in the D-form of wave_scan.hip.h
      const T yn1 = dpp_row_zero_fill<0x101>(y1[0]), yn3 = dpp_row_zero_fill<0x101>(y3[0]);  // row_shl:1
      T D[4] = {T(0), T(0), T(0), T(0)}, c[2] = {T(0), T(0)};
#pragma unroll
      for (int q = 0; q < SPL; ++q) {
        // model.hpp:41 and its derivative in x1
        const T g = fma_t(Pq[q], sd[q], fma_t(M::A52, s1[q], fma_t(Qq[q], cd[q], M::C22 * (x2f[q] - y3[q]))));
        const T J1 = fma_t(Qq[q], sd[q], fma_t(M::A52, c1[q], -(Pq[q] * cd[q])));
        const T t1 = fma_t(dtau, y3[q], y1[q]), t3 = fma_t(dtau, g, y3[q]);
        jq[q] = dq[q] * J1;
        const T r1 = t1 - (q + 1 < SPL ? y1[q + 1 < SPL ? q + 1 : 0] : yn1);
        const T r3 = t3 - (q + 1 < SPL ? y3[q + 1 < SPL ? q + 1 : 0] : yn3);
        c0q[q] = tr[q] ? r1 : T(0), c1q[q] = tr[q] ? r3 : T(0);
        const T n00 = fma_t(dq[q], D[2], D[0]);
        const T n01 = fma_t(dq[q], D[3], D[1] + dq[q]);
        const T n10 = fma_t(eq[q], D[2], fma_t(jq[q], D[0], D[2] + jq[q]));
        const T n11 = fma_t(eq[q], D[3], fma_t(jq[q], D[1], D[3] + eq[q]));
        const T m0 = fma_t(dq[q], c[1], c[0] + c0q[q]);
        const T m1 = fma_t(eq[q], c[1], fma_t(jq[q], c[0], c[1] + c1q[q]));
        D[0] = n00, D[1] = n01, D[2] = n10, D[3] = n11, c[0] = m0, c[1] = m1;
      }
      aff2_step_vec<0>(c, D), aff2_step_mat<0>(D);
      aff2_step_vec<1>(c, D), aff2_step_mat<1>(D);
      aff2_step_vec<2>(c, D), aff2_step_mat<2>(D);
      aff2_step_vec<3>(c, D);
      T e1 = scan_partner<0>(c[0]), e3 = scan_partner<0>(c[1]);  // correction of the lane's first stage
      T viol = T(-1);
      T dl1[SPL];
      const T th = T(1e-10);
#pragma unroll
      for (int q = 0; q < SPL; ++q) {
        y1[q] += e1, y3[q] += e3;
        dl1[q] = e1, ad[q] = -e1, a1[q] = e1;
        // |e| - th (1 + |y|) > 0 ?  (the "- th" is taken off once, after the loop)
        const T vq = __builtin_fmax(fma_t(-th, abs_t(y1[q]), abs_t(e1)), fma_t(-th, abs_t(y3[q]), abs_t(e3)));
        viol = s_0 + q <= dv ? __builtin_fmax(viol, vq) : viol;  // (beyond stage dv its correction is carried on unchanged)
        const T n1 = fma_t(dq[q], e3, e1 + c0q[q]);
        const T n3 = fma_t(eq[q], e3, fma_t(jq[q], e1, e3 + c1q[q]));
        e1 = n1, e3 = n3;
      }
      CGM_STAMP(*this, 22);
      if (!__any(run && viol > th)) {
        // the correction is below the square root of the rounding level: trig values to first order, done
#pragma unroll
        for (int q = 0; q < SPL; ++q) {
          const T nsd = fma_t(cd[q], -dl1[q], sd[q]), ncd = fma_t(sd[q], dl1[q], cd[q]);
          const T ns1 = fma_t(c1[q], dl1[q], s1[q]), nc1 = fma_t(s1[q], -dl1[q], c1[q]);
          sd[q] = nsd, cd[q] = ncd, s1[q] = ns1, c1[q] = nc1;
        }
        break;
      }
    }
    mid();
    row_costate<MODE>(x0f, y1, x2f, y3, sd, cd, c1, Wr, dtau, out, run, tid_o, u0);
  }

  // Hessenberg column k of one instance: stored reflectors, new reflector, residual rotation (gmres.hpp:71-90) — scalar
  // work on the instance's small Krylov arrays in LDS.  hn = h(k+1,k).  Returns rho_e[k+1]; `writer` lanes store.
  static __device__ __forceinline__ T hess_column(T* Hi, T* gi, T* rhoi, int k, T hn, bool writer) {
    T* Hk = Hi + ((k * (k + 1)) >> 1);  // compact: column k = rows 0..k (h(k+1,k) arrives as `hn` and becomes 0)
    // The running entry stays in a register (a) and only ORIGINAL column entries / reflector words are read
    // from LDS, one step ahead: no store-to-load round trip through LDS between consecutive reflectors.
    // (the look-ahead reads one or two words past the column: the next column / array, in bounds, never used)
    T a = Hk[0];
    T g0n = gi[0], g1n = gi[1], g2n = gi[2], cn = Hk[1];
    for (int i = 0; i < k; ++i) {
      const T g0 = g0n, g1 = g1n, g2 = g2n, c = cn;
      g0n = gi[3 * i + 3], g1n = gi[3 * i + 4], g2n = gi[3 * i + 5], cn = Hk[i + 2];
      const T beta = (g0 * a + g1 * c) * g2;
      if (writer) Hk[i] = a - beta * g0;
      a = c - beta * g1;
    }
    const T c = hn;
    const T sigma = -(a < T(0.0) ? T(-1.0) : T(1.0)) * sqrt_t<T>(a * a + c * c);
    const T g0 = a - sigma, g1 = c;
    const T g2 = T(2.0) / (g0 * g0 + g1 * g1);
    const T ek = rhoi[k];
    const T beta = g0 * ek * g2;
    const T en = -beta * g1;
    if (writer) {
      gi[3 * k] = g0, gi[3 * k + 1] = g1, gi[3 * k + 2] = g2;
      Hk[k] = sigma;
      rhoi[k] = ek - beta * g0;
      rhoi[k + 1] = en;
    }
    return en;
  }

  // Gmres::gmres (gmres.hpp:28-112).  In: x (registers `xv`), b (`bb`) and A*x0 (`ax0`), all in the row layout.
  // Out: xv updated.  All threads of the block must call this (it contains workgroup barriers).
  __device__ __forceinline__ void gmres(T* xv, const T* bb, const T* ax0) {
    const int kmax = P.kmax, k1 = kmax + 1;
    T* Hi = S.H + inst * P.Hp;  // compact: column k has k+1 entries (rows 0..k) at offset k(k+1)/2; h(k+1,k): S.hsub
    auto hoff = [](int k) { return (k * (k + 1)) >> 1; };
    T* rhoi = S.rho + inst * k1;
    T* gi = S.g + inst * 3 * kmax;
    T vcur[MAXM], w[MAXM];
    // Ring of NBUF register buffers for the older basis vectors: NBUF rows are requested before the sweep starts, and
    // every buffer is refilled with row i+NBUF as soon as round i has consumed it, so each load has NBUF-1 rounds
    // (~1000 cycles) to arrive.  Static buffer indices: one fully unrolled instance per iteration count (<= KRING).
#define CGM_KCASE(f, n) \
  case n:               \
    f(std::integral_constant<int, n>{}); \
    break;
#define CGM_KCASES(f) \
  CGM_KCASE(f, 1) CGM_KCASE(f, 2) CGM_KCASE(f, 3) CGM_KCASE(f, 4) CGM_KCASE(f, 5) CGM_KCASE(f, 6) CGM_KCASE(f, 7) \
  CGM_KCASE(f, 8) CGM_KCASE(f, 9) CGM_KCASE(f, 10) CGM_KCASE(f, 11) CGM_KCASE(f, 12)
    constexpr int NBUF = MAXM <= 10 ? 3 : 2, KRING = 12;
    // row buffers of the streaming loops (lean kernels, k_max > KRING).  Three buffers were measured as well: the 256-register
    // kernels then spill inside the Arnoldi loop (fp32 N = 100: 384 spilled VGPRs, cfg 5 526 -> 609 us/tick)
    constexpr int SDEPTH = 2;
    // The streaming loops end with a row request nobody consumes (their unconditional look-ahead).  In the kernels that
    // ALSO carry the register ring (not lean: k_max picks the form at run time) the compiler must assume those registers
    // are still being written wherever it reuses them on the ring path — it reuses them in the sweeps — and, the counter
    // being in order, it put s_waitcnt vmcnt(0) at the start of wave 0's state sweep (draining the store of the new
    // basis row) and at the start of its costate sweep (waiting for the ring rows requested just before, which are not
    // needed until the Gram-Schmidt rounds).  Draining the look-ahead where the streaming loop ends keeps the ring path
    // free of both: s_waitcnt vmcnt(0), other counters untouched.
    auto drain_stream = [] {
      if constexpr (!LEAN) __builtin_amdgcn_s_waitcnt(0x0F70);
    };
    // The first NKEEP basis vectors never leave the row lanes' registers (the compiler parks them in the AGPR file): v_0
    // and v_1 are read again in every later iteration — 17 of the 55 Gram-Schmidt row reads of a k = 10 solve — and the
    // Gram-Schmidt rounds are bounded by the CU's 64 B/clk vector-memory path, not by issue.  The ring then serves the
    // rows from NKEEP on.  Pays for itself only where registers are left: the one-workgroup-per-CU kernels with short
    // vectors, with the solution vector parked in HBM for the duration of the loop like the long-vector kernels do.
    constexpr int NKEEP = (!LEAN && MAXM <= 10) ? 2 : 0;
    T vkeep[NKEEP > 0 ? NKEEP : 1][MAXM];
    // workgroup-uniform; longer bases use the plain streaming loop.  So does the lean plan: with 256 registers per wave the
    // twelve straight-line copies of the rounds push everything that lives across them (U, the sweep constants) into
    // scratch — in every block of the kernel, executed or not.
    const bool preload = !LEAN && kmax <= KRING;
    bool active = valid;
    // Long vectors: x is not touched again before the final update (gmres.hpp:110-111), and 2*MAXM more live registers
    // push the Gram-Schmidt rounds into AGPR copies and scratch (profiles/r02_isa_summary.md).  Park it in HBM
    // (own row, same thread writes and reads it back: program order) and fetch it back behind the back substitution.
    T* const park_row = P.park + size_t(blockIdx.x * IPW + inst) * P.Lv;
    constexpr bool PARK = MAXM > 10 || (LEAN && sizeof(T) == 8) || NKEEP > 0 || NWT != 0;  // (lean fp64: 256 registers per wave)
    if constexpr (PARK) store_vec(park_row, xv);
    // r0 = b - A x0 ; rho = ||r0||      gmres.hpp:33-37
    {
      T ss = 0;
#pragma unroll
      for (int m = 0; m < MAXM; ++m) {
        vcur[m] = bb[m] - ax0[m];
        ss += vcur[m] * vcur[m];
      }
      const T rho0 = sqrt_t<T>(row16_sum(ss));
      if (r == 0) rhoi[0] = rho0;
      if (active && !finite_t(rho0)) {  // CGMRES_HIP_EXIT_NONFINITE (row-uniform; see poison_nonfinite in tick_lane.hip.h)
        active = false;
        if (r == 0) S.reason[inst] = 4;
      }
      if (active && rho0 < P.tol) {  // gmres.hpp:39-41
        active = false;
        if (r == 0) S.reason[inst] = 2;
      }
      if (active) {
        const T inv = T(1.0) / rho0;  // gmres.hpp:44
#pragma unroll
        for (int m = 0; m < MAXM; ++m) vcur[m] = vcur[m] * inv;
        store_vec(vrow(0), vcur);
        if constexpr (NKEEP > 0) {
#pragma unroll
          for (int m = 0; m < MAXM; ++m) vkeep[0][m] = vcur[m];
        }
        publish_direction(vcur);
      }
      if (r == 0) S.flag[inst] = active ? 1 : 0;
    }
    // Fixed-k mode (tol = 0: no residual test can succeed, every running instance does all k_max iterations): the
    // Hessenberg column of iteration k is not needed before the final triangular solve, so it is taken off the critical
    // path — 16 lanes of wave 1 (one per instance) do column k-1 while wave 0 runs the first chunk of sweep k, where
    // the coefficient waves would otherwise wait; the last column is done in place.  With tol > 0 the column decides
    // whether the next mat-vec runs at all (gmres.hpp:93-95) and stays where the reference has it.
    // (NWT: the rows of a workgroup do not meet inside the loop at all — every row does its own column in place)
    const bool defer_hess = NWT == 0 && IPW * 16 >= 128 && P.tol == T(0);  // workgroup-uniform
    // NWT in fixed-k mode: no column is needed before the triangular solve either, and no wave has idle time to hide one
    // in — the loop only records h(k+1,k) (in the slot of the reflector's second word, which is what it becomes) and
    // the QR factorisation of the whole Hessenberg matrix follows the loop, the columns spread over the lanes of the row.
    const bool lazy_qr = NWT != 0 && P.tol == T(0);
    int k = 0;
    for (; k < kmax; ++k) {  // gmres.hpp:46
      CGM_STAMP(*this, 14);
      if constexpr (LEAN || NWT != 0 || MAXM > 10) {
        // Whatever derives from the thread index alone is invariant over the whole launch, gets hoisted to the kernel
        // entry and — there being no room in these kernels — spilled; its reloads inside the iteration sit behind
        // s_waitcnt vmcnt(0), i.e. wait for the basis rows in flight.  Opaque copies once per iteration: the few integer
        // operations are redone instead.  (Spilled VGPRs: lean pendulum fp64 91 -> 2, lean two-mass system 81 -> 0,
        // lean pendulum fp32 k = 20 86 -> 0, the row-parallel kernel 108 -> 0 and 103.6 -> 99.8 us per tick; the
        // short-vector serial-sweep kernels have next to none to lose and are 1 % slower with it: left alone.)
        asm volatile("" : "+v"(tid), "+v"(inst), "+v"(r), "+v"(b));
      }
      if constexpr (NWT != 0) {
        // Row-parallel sweeps: everything an iteration touches in LDS — the row of W, the row's small Krylov arrays —
        // is written and read by the 16 lanes of ONE row, i.e. inside one wave, whose LDS operations complete in
        // order: no workgroup barrier in the loop, and every wave leaves it when ITS four rows are done.
        if (!__any(active)) break;
      } else {
      // Workgroup barrier that orders LDS only (W and the flags are what the sweep lanes need).  __syncthreads()
      // would also drain this wave's HBM store of the new basis row (s_waitcnt vmcnt(0)) — nobody else reads it.
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
      __builtin_amdgcn_s_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
      {
        // (16-byte reads, all requested before the first is used: left as IPW scalar reads the compiler issued them as
        // eight ds_read2_b32 into ONE register pair, each behind s_waitcnt lgkmcnt(0) — eight LDS round trips in a row on
        // every wave, at the top of every iteration.  The int arrays start on a 16-byte boundary: every array in front of
        // them is a multiple of IPW scalars.)
        static_assert(IPW % 4 == 0, "flags are read four at a time");
        const int4* f4 = reinterpret_cast<const int4*>(__builtin_assume_aligned(S.flag, 16));
        int4 fl[IPW / 4];
#pragma unroll
        for (int j = 0; j < IPW / 4; ++j) fl[j] = f4[j];
        act_i = S.flag[tid & (IPW - 1)], act_q = S.flag[(tid & 63) >> 2];  // (IPW = 8: lanes 32.. read `reason` words, unused)
        int any = 0;
#pragma unroll
        for (int j = 0; j < IPW / 4; ++j) any |= (fl[j].x | fl[j].y) | (fl[j].z | fl[j].w);
        if (!any) break;
      }
      }
      CGM_STAMP(*this, 15);
      // The first basis vectors this iteration needs are requested from HBM/L2 early.  Waves 1-3 do it NOW (the rows
      // arrive while wave 0 sweeps); wave 0 — four waves' row requests take the CU's address unit ~800 cycles, which
      // would delay the sweep — does it after its state sweep, where it otherwise waits for the coefficient tail.
      T vbuf[NBUF][MAXM];
      // (vmcnt(0) first: the two call sites below write the same registers, so without it the compiler must assume the
      // other site's requests are still in flight — with an in-order counter and conditional requests that leaves it
      // only s_waitcnt vmcnt(0) in front of EVERY row: each request waited for the previous row to arrive, on wave 0
      // right after its state sweep.  What is really outstanding here is this thread's store of the newest basis row,
      // at most.  Together with drain_stream: state phase -0.8 k cycles per sweep, headline 134.2 -> 133.1 us/tick.)
      auto request_rows = [&]() {
        if (preload && active) {
          __builtin_amdgcn_s_waitcnt(0x0F70);
#pragma unroll
          for (int i = 0; i < NBUF; ++i)
            if (i + NKEEP < k) load_vec(vbuf[i], vrow(i + NKEEP));
        }
      };
      if (NWT == 0 && tid >= 64) request_rows();
      auto deferred_column = [&]() {  // column k-1 of instance j = tid-64, if iteration k-1 produced one for it
        const int j = tid - 64;
        if (defer_hess && k > 0 && j >= 0 && j < IPW && S.reason[j] == 0 && S.nax[j] == k) {
          T* Hj = S.H + j * P.Hp;
          hess_column(Hj, S.g + j * 3 * kmax, S.rho + j * k1, k - 1, S.hsub[j], true);
        }
      };
      if constexpr (NWT != 0) {
        CGM_STAMP(*this, 3);
        // (the basis rows are requested between the Newton iterations and the costate scans: early enough to arrive
        // behind the scans, late enough that their registers are not live across the iterations, where the register
        // file is fullest — requested before the sweep they sit in AGPRs and every use in the rounds below is a copy)
        if constexpr (NWT == 2) {
          request_rows();  // (a light sweep: the rows may be in flight across it)
          row_affine_sweep<F_AX>(S.xh, dtau_h, S.W + inst * P.Lp, S.W, active);
        } else {
          row_newton_sweep<F_AX>(dtau_h, S.W, active, request_rows);  // :48  W <- A v_k, in place
        }
        CGM_STAMP(*this, 6);
      } else {
        ax(true, request_rows, deferred_column);  // :48  W <- A v_k, in place
      }
      if (active) {
        lds_to_reg(w, S.W);
        if constexpr (NWT != 0) {
          if (!row_moved) {  // (see publish_direction)
#pragma unroll
            for (int m = 0; m < MAXM; ++m) w[m] = T(0);
          }
        }
        T* Hk = Hi + hoff(k);
        // modified Gram-Schmidt, gmres.hpp:52-58, in order; v_k itself is still in registers
        auto mgs_round = [&](const T* vi, int i) {
          T pa = 0, pb = 0;  // two partial sums: the fp64 FMA chain is latency-bound (8 cycles/op dependent)
#pragma unroll
          for (int m = 0; m < MAXM; m += 2) {
            pa += vi[m] * w[m];
            if (m + 1 < MAXM) pb += vi[m + 1] * w[m + 1];
          }
          const T hik = row16_sum(pa + pb);
#pragma unroll
          for (int m = 0; m < MAXM; ++m) w[m] = w[m] - vi[m] * hik;
          if (r == 0) Hk[i] = hik;
        };
        if (preload) {
          // One straight-line instance per k: with no branch between a load and its use the compiler counts the
          // outstanding loads exactly (s_waitcnt vmcnt(2*MAXM) in the steady state instead of vmcnt(0)).
          auto rounds = [&](auto kc) {
            constexpr int K = decltype(kc)::value;
#pragma unroll
            for (int i = 0; i < K; ++i) {
              if constexpr (NKEEP > 0) {
                if (i < NKEEP) {
                  mgs_round(vkeep[i < NKEEP ? i : 0], i);
                  continue;
                }
              }
              mgs_round(vbuf[(i - NKEEP) % NBUF], i);
              if (i + NBUF < K) load_vec(vbuf[(i - NKEEP) % NBUF], vrow(i + NBUF));
            }
          };
          switch (k) {
            CGM_KCASES(rounds)
            default: break;
          }
        } else {
          // (register-starved lean kernels: v_k is streamed back from its row like the older ones instead of being
          // held in registers through the whole sweep; the row was written by this same thread)
          // The next row is requested UNCONDITIONALLY (the last trip re-reads its own row): a guard would be a branch
          // between a load and its use, and the compiler then waits with vmcnt(0) — i.e. also for the row it has just
          // requested, and nothing overlaps.
          // SDEPTH row buffers with STATIC indices (no rotation by register moves: 20 v_mov per round, and a move would
          // wait for the newest request): the main loop does SDEPTH rounds per trip and refills each buffer right after
          // its round (row index clamped: the last trips re-read the last row), the remaining < SDEPTH rounds find
          // their rows already requested.
          T vq[SDEPTH][MAXM];
          const int kk = VK_IN_REGS ? k : k + 1;
          if (kk > 0) {
#pragma unroll
            for (int d = 0; d < SDEPTH; ++d) load_vec(vq[d], vrow(d < kk ? d : kk - 1));
          }
          int i = 0;
          for (; i + SDEPTH <= kk; i += SDEPTH) {
#pragma unroll
            for (int d = 0; d < SDEPTH; ++d) {
              // (keeps the requests TOGETHER and ahead of the arithmetic: under register pressure the scheduler otherwise
              // sinks every load next to its use — one exposed round trip per element)
              __builtin_amdgcn_sched_barrier(0);
              mgs_round(vq[d], i + d);
              const int nxt = i + d + SDEPTH;
              load_vec(vq[d], vrow(nxt < kk ? nxt : kk - 1));
            }
          }
#pragma unroll
          for (int d = 0; d < SDEPTH - 1; ++d)
            if (i + d < kk) mgs_round(vq[d], i + d);
          drain_stream();
        }
        if constexpr (VK_IN_REGS) mgs_round(vcur, k);
        T na = 0, nb = 0;
#pragma unroll
        for (int m = 0; m < MAXM; m += 2) {
          na += w[m] * w[m];
          if (m + 1 < MAXM) nb += w[m + 1] * w[m + 1];
        }
        CGM_STAMP(*this, 7);
        const T hn = sqrt_t<T>(row16_sum(na + nb));  // :60
        if (r == 0) {
          S.hsub[inst] = hn;
          S.nax[inst] = k + 1;
        }
        if (abs_t(hn) < T(DBL_EPSILON) || !finite_t(hn)) {  // :63-65 breakdown: x untouched; non-finite: x <- NaN below
          active = false;
          if (r == 0) {
            S.reason[inst] = finite_t(hn) ? 3 : 4;
            S.flag[inst] = 0;
          }
        } else {
          const T inv = T(1.0) / hn;  // :67
#pragma unroll
          for (int m = 0; m < MAXM; ++m) vcur[m] = w[m] * inv;
          store_vec(vrow(k + 1), vcur);
          if constexpr (NKEEP > 1) {
#pragma unroll
            for (int q = 1; q < NKEEP; ++q)
              if (k + 1 == q) {
#pragma unroll
                for (int m = 0; m < MAXM; ++m) vkeep[q][m] = vcur[m];
              }
          }
          publish_direction(vcur);
          // Hessenberg column k: stored reflectors, new reflector, residual rotation (:71-90) — scalar work.
          // Every lane of the row computes it from the same LDS words (broadcast reads; a row never straddles
          // a wave, and LDS operations of one wave complete in order), lane 0 writes back: the convergence
          // decision is therefore row-uniform.
          CGM_STAMP(*this, 8);
          // (deferred in fixed-k mode, except for the last column: no sweep follows it)
          const bool column_now = !lazy_qr && (!defer_hess || k + 1 == kmax);
          if (lazy_qr && r == 0) gi[3 * k + 1] = hn;
          const T en = column_now ? hess_column(Hi, gi, rhoi, k, hn, r == 0) : T(1);
          CGM_STAMP(*this, 9);
          if (column_now && abs_t(en) < P.tol) {  // :93-95 — converged: column k is NOT used by the solve
            active = false;
            if (r == 0) {
              S.reason[inst] = 1;
              S.ksolve[inst] = k;
              S.flag[inst] = 0;
            }
          }
        }
      }
    }
    if constexpr (NWT == 0) __syncthreads();  // (NWT: the status words of a row are its own wave's)
    if constexpr (NWT != 0) {
      if (lazy_qr) {
        // gmres.hpp:71-90 for all columns at once: lane c of the row owns column c and carries its running entry `a`
        // through the reflectors in order — step i: lane i turns (a, h(i+1,i)) into reflector i (the same expressions as
        // hess_column) and leaves it in LDS, every lane reads it back (same wave: LDS operations complete in order),
        // the lanes c > i apply it to their column, and all lanes advance the residual vector.  Per column the
        // arithmetic and its order are those of hess_column; the serial chain is k_max steps instead of k_max^2 / 2.
        const int rs = S.reason[inst], nx = S.nax[inst];
        const int n_col = !valid ? 0 : (rs == 0 ? nx : nx - 1);  // (a breakdown leaves its own column unrotated, :63-65)
        const int c = r < kmax ? r : 0;
        T* Hc = Hi + hoff(c);
        const bool mine = r < n_col;
        T a = mine ? Hc[0] : T(0);
        T ek = rhoi[0];
        for (int i = 0; i < kmax; ++i) {
          if (!__any(i < n_col)) break;
          const bool on = i < n_col;
          const T cN = gi[3 * i + 1];
          const T sigma = -(a < T(0.0) ? T(-1.0) : T(1.0)) * sqrt_t<T>(a * a + cN * cN);
          const T q0 = a - sigma;
          const T q2 = T(2.0) / (q0 * q0 + cN * cN);
          if (on && r == i) {
            gi[3 * i] = q0, gi[3 * i + 2] = q2;
            Hc[i] = sigma;
          }
          const T g0 = gi[3 * i], g1 = cN, g2 = gi[3 * i + 2];
          const T beta_r = g0 * ek * g2;
          if (on && r == 0) rhoi[i] = ek - beta_r * g0;
          ek = on ? -beta_r * g1 : ek;
          if (mine && r > i) {
            const T cn = Hc[i + 1];
            const T beta = (g0 * a + g1 * cn) * g2;
            Hc[i] = a - beta * g0;
            a = cn - beta * g1;
          }
        }
        if (r == 0 && n_col > 0) rhoi[n_col] = ek;
      }
    }
    CGM_STAMP(*this, 10);
    // natural exit: every column is used
    const int reason = S.reason[inst];
    const int ks = reason == 0 ? (valid ? kmax : 0) : (reason == 1 ? S.ksolve[inst] : 0);
    // the first basis rows of the x update are requested before the (serial) back substitution
    T vbx[NBUF][MAXM];
    if (preload && valid && reason <= 1) {
#pragma unroll
      for (int j = 0; j < NBUF; ++j)
        if (j + NKEEP < ks) load_vec(vbx[j], vrow(j + NKEEP));
    }
    if (valid && reason <= 1) {
      // back substitution (gmres.hpp:100-107), column-oriented over the lanes of the row: lane j owns e_j; step i
      // (descending) turns e_i into y_i = e_i / H_ii, broadcasts it inside the row (ds_bpermute) and every lane j < i
      // subtracts H(j,i) y_i — for a fixed j the subtractions come in the reference's order (i = ks-1 ... j+1).
      // One LDS read + one crossbar round trip per step instead of a dependent LDS chain of length ks-i on lane 0.
      if (kmax <= 15) {
        T e = r < ks ? rhoi[r] : T(0);
        const int row_lane0 = (threadIdx.x & 63) & ~15;
        for (int i = ks - 1; i >= 0; --i) {
          // (r > i + 1 reads words of the following columns / arrays: in bounds, only used where r < i)
          const T hii = Hi[hoff(i) + i], hji = Hi[hoff(i) + r];
          const T y = e / hii;                                   // meaningful in lane i
          const T yi = row_bcast(y, row_lane0 + i);
          e = r < i ? e - hji * yi : (r == i ? y : e);
        }
        if (r < ks) rhoi[r] = e;
        if (r == 0) S.ksolve[inst] = ks;
      } else if (kmax <= 31) {
        // the same with TWO unknowns per lane (e_r and e_(r+16)): k_max = 20 of the long-horizon configuration — on lane 0
        // alone the 200 dependent LDS steps of this solve were 2.7 % of that tick
        T e0 = r < ks ? rhoi[r] : T(0), e1 = r + 16 < ks ? rhoi[r + 16] : T(0);
        const int row_lane0 = (threadIdx.x & 63) & ~15;
        for (int i = ks - 1; i >= 0; --i) {
          // (rows beyond i + 1 read words of the following columns / arrays: in bounds, only used where the row is < i)
          const T hii = Hi[hoff(i) + i], h0 = Hi[hoff(i) + r], h1 = Hi[hoff(i) + r + 16];
          const bool hi_half = i >= 16;                           // (uniform) which of the lane's two unknowns e_i is
          const T y = (hi_half ? e1 : e0) / hii;                  // meaningful in lane i & 15
          const T yi = row_bcast(y, row_lane0 + (i & 15));
          e0 = r < i ? e0 - h0 * yi : (r == i ? y : e0);
          e1 = r + 16 < i ? e1 - h1 * yi : (r + 16 == i ? y : e1);
        }
        if (r < ks) rhoi[r] = e0;
        if (r + 16 < ks) rhoi[r + 16] = e1;
        if (r == 0) S.ksolve[inst] = ks;
      } else if (r == 0) {
        for (int i = ks - 1; i >= 0; --i) {
          T ei = rhoi[i];
          for (int j = ks - 1; j > i; --j) ei -= Hi[hoff(j) + i] * rhoi[j];
          rhoi[i] = ei / Hi[hoff(i) + i];
        }
        S.ksolve[inst] = ks;
      }
    }
    if constexpr (PARK) load_vec(xv, park_row);
    if constexpr (NWT == 0) __syncthreads();  // (NWT: y is written and read by the lanes of one row, i.e. one wave)
    CGM_STAMP(*this, 11);
    if (valid && reason <= 1) {
      // x += V[:,0:ks] y  (gmres.hpp:110-111), accumulated j-ascending from 0 like matrix.hpp:82-91
      T acc[MAXM];
#pragma unroll
      for (int m = 0; m < MAXM; ++m) acc[m] = T(0.0);
      if (preload) {  // same register ring as the Gram-Schmidt rounds
        auto& vbuf = vbx;
        auto rounds = [&](auto kc) {
          constexpr int K = decltype(kc)::value;
#pragma unroll
          for (int j = 0; j < K; ++j) {
            const T yj = rhoi[j];
            if constexpr (NKEEP > 0) {
              if (j < NKEEP) {
#pragma unroll
                for (int m = 0; m < MAXM; ++m) acc[m] += vkeep[j < NKEEP ? j : 0][m] * yj;
                continue;
              }
            }
#pragma unroll
            for (int m = 0; m < MAXM; ++m) acc[m] += vbuf[(j - NKEEP) % NBUF][m] * yj;
            if (j + NBUF < K) load_vec(vbuf[(j - NKEEP) % NBUF], vrow(j + NBUF));
          }
        };
        switch (ks) {
          CGM_KCASES(rounds)
          default: break;
        }
      } else {
        // streaming form, SDEPTH rows in flight (see the Gram-Schmidt loop)
        T vq[SDEPTH][MAXM];
        if (ks > 0) {
#pragma unroll
          for (int d = 0; d < SDEPTH; ++d) load_vec(vq[d], vrow(d < ks ? d : ks - 1));
        }
        auto axpy = [&](const T* vj, int j) {
          const T yj = rhoi[j];
#pragma unroll
          for (int m = 0; m < MAXM; ++m) acc[m] += vj[m] * yj;
        };
        int j = 0;
        for (; j + SDEPTH <= ks; j += SDEPTH) {
#pragma unroll
          for (int d = 0; d < SDEPTH; ++d) {
            __builtin_amdgcn_sched_barrier(0);
            axpy(vq[d], j + d);
            const int nxt = j + d + SDEPTH;
            load_vec(vq[d], vrow(nxt < ks ? nxt : ks - 1));
          }
        }
#pragma unroll
        for (int d = 0; d < SDEPTH - 1; ++d)
          if (j + d < ks) axpy(vq[d], j + d);
        drain_stream();
      }
#pragma unroll
      for (int m = 0; m < MAXM; ++m) xv[m] = xv[m] + acc[m];
    }
    if (valid && reason == 4) {  // CGMRES_HIP_EXIT_NONFINITE: what the reference's fall-through ends with
#pragma unroll
      for (int m = 0; m < MAXM; ++m) xv[m] = elem(m) < P.L ? quiet_nan<T>() : T(0);
    }
  }

  // status + small Krylov arrays of this row's instance -> HBM
  __device__ __forceinline__ void store_status() const {
    if (!valid) return;
    const int kmax = P.kmax, k1 = kmax + 1, ks_all = k1 * k1 + k1 + 3 * kmax;
    T* dst = P.kry + size_t(b) * ks_all;
    const T* Hi = S.H + inst * P.Hp;  // compact columns -> the reference's (k_max+1) x (k_max+1) column-major array
    for (int q = r; q < k1 * k1; q += 16) {
      const int col = q / k1, row = q - col * k1;
      // rows 0..col from the compact columns; the subdiagonal entry is 0 once the column has been rotated (gmres.hpp:85) —
      // a breakdown leaves at column nax-1 BEFORE its rotation (gmres.hpp:63-65): there it still holds h(k+1,k)
      T v = (col < kmax && row <= col) ? Hi[((col * (col + 1)) >> 1) + row] : T(0);
      if (row == col + 1 && col + 1 == S.nax[inst] && S.reason[inst] == 3) v = S.hsub[inst];
      dst[q] = v;
    }
    for (int q = r; q < k1; q += 16) dst[k1 * k1 + q] = S.rho[inst * k1 + q];
    for (int q = r; q < 3 * kmax; q += 16) dst[k1 * k1 + k1 + q] = S.g[inst * 3 * kmax + q];
    if (r == 0) {
      P.n_ax[b] = S.nax[inst];
      P.reason[b] = S.reason[inst];
    }
  }
};

// ---- the tick kernel: cgmres.hpp:78-110 for IPW instances ----------------------------------------
template <class M, class T, int IPW, int MAXM, bool LEAN = false, int PAR = 0, int NWT = 0>
__global__ __launch_bounds__(IPW * 16) __attribute__((amdgpu_waves_per_eu(LEAN ? 2 : 1, LEAN ? 2 : 1))) void tick_wg_kernel(
    WgParams<T> P) {
  extern __shared__ __align__(16) unsigned char smem[];
  WgCtx<M, T, IPW, MAXM, LEAN, PAR, NWT> C(P, smem);
  T du[MAXM], bb[MAXM];
  C.load_common(P.U);
  C.load_row_to_reg(du, P.dUdt, P.Lg);
  const int nt = P.n_ticks;
  for (int tk = 0; tk < nt; ++tk) {
    const bool last = tk + 1 == nt;
    C.dtau_h = P.dtau_tab[2 * tk], C.dtau_0 = P.dtau_tab[2 * tk + 1];
    // H region zeroed so the exported Hessenberg has no stale entries
    for (int q = C.r; q < P.Hp; q += 16) C.S.H[C.inst * P.Hp + q] = T(0);
    if (M::NP > 0 && P.ptau_seq) C.load_ptau_tick(P.ptau_seq + size_t(tk) * P.pseq_tick);
    __syncthreads();  // (also drains the prologue's / load_ptau_tick's HBM stores of the lean parameter table)
    CGM_STAMP(C, 0);
    if constexpr (!LEAN) C.publish_direction(du);  // direction of the first mat-vec: x0 = dUdt (warm start, cgmres.hpp:99)
    T ax0[MAXM];
    if constexpr (NWT == 2) {
      C.preamble_affine(bb, ax0);
    } else if constexpr (NWT == 1) {
      C.preamble_rows(bb, ax0);
      C.store_base();
    } else {
      C.template preamble<true>(bb, ax0, du);  // Fh in LDS (or HBM); b and A*dUdt in registers
    }
    CGM_STAMP(C, 1);
    C.gmres(du, bb, ax0);
    CGM_STAMP(C, 12);
    // U += dUdt*dt, u = U[0:dim_u]  (cgmres.hpp:102-109)
    T un[MAXM];
    if constexpr (LEAN) {
      C.get_urow(un);
    } else {
      C.lds_to_reg(un, C.S.U);
    }
#pragma unroll
    for (int m = 0; m < MAXM; ++m) un[m] = un[m] + du[m] * P.dt;
    constexpr bool U_ROW_IS_HBM = LEAN && !decltype(C)::U_IN_REGS;  // then every tick writes its U row
    if (last || U_ROW_IS_HBM) C.reg_to_row(P.U, P.Lg, un);
    if (last) {  // the rest of the controller state goes back to HBM with the last tick of the launch only
      C.reg_to_row(P.dUdt, P.Lg, du);
      if (C.valid && C.r < M::NU) P.u_out[size_t(C.b) * M::NU + C.r] = un[0];  // element e = r (m = 0), r < NU <= 16
      C.store_status();
      // x_dxh and F_dxh_h of this tick stay with the controller like the reference's members (cgmres.hpp:198-201):
      // a white-box Ax_func after control() evaluates with them (with fh_hbm the preamble has stored the row already)
      if (!C.fh_hbm()) {
        T fh[MAXM];
        C.lds_to_reg(fh, C.S.Fh);
        C.reg_to_row(P.Fh, P.Lg, fh);
      }
      if (C.valid && C.r < M::NX) P.xdxh[size_t(C.b) * M::NX + C.r] = C.S.xh[C.r * IPW + C.inst];
    }
    // the plant step and the next tick read the new U: from LDS, or (lean) from the row registers + a small u array
    if constexpr (LEAN) {
      if constexpr (decltype(C)::U_IN_REGS) {
#pragma unroll
        for (int m = 0; m < MAXM; ++m) C.ureg[m] = un[m];
      }
      if (C.valid && C.r < M::NU) C.S.u0[C.r * IPW + C.inst] = un[0];
    } else {
      C.reg_to_lds(C.S.U, un);
    }
    if (P.x_next) {  // plant step of the example main loop (<example>/main.cpp:71-73)
      __syncthreads();
      if (C.sweep_lane) {
        const int i = C.tid;
        T x[M::NX], u[M::NU], f[M::NX], tr[M::NC > 0 ? M::NC : 1];
#pragma unroll
        for (int c = 0; c < M::NX; ++c) x[c] = C.S.xs[c * IPW + i];
#pragma unroll
        for (int j = 0; j < M::NU; ++j) u[j] = LEAN ? C.S.u0[j * IPW + i] : C.S.U[i * P.Lp + j];
        C.model_dxdt(f, x, u, tr, i, 0);  // the example's plant = the model's own state equation (p of stage 0)
#pragma unroll
        for (int c = 0; c < M::NX; ++c) {
          const T xn = x[c] + f[c] * P.dt;
          if (last) P.x_next[size_t(C.bi) * M::NX + c] = xn;
          C.S.xs[c * IPW + i] = xn;
        }
      }
    }
    if (!last) {
      if (C.r == 0) C.S.flag[C.inst] = 0, C.S.reason[C.inst] = 0, C.S.nax[C.inst] = 0, C.S.ksolve[C.inst] = 0;
      __syncthreads();  // U, x and the cleared status words are in place for the next tick
    }
  }
  CGM_STAMP(C, 13);
#ifdef CGM_STAMPS
  cgm_stamp_flush();
#endif
}

// ---- white-box hooks on the same device code -------------------------------------------------------
template <class M, class T, int IPW, int MAXM>
__global__ __launch_bounds__(IPW * 16) __attribute__((amdgpu_waves_per_eu(1, 1))) void hook_wg_kernel(WgParams<T> P) {
  extern __shared__ __align__(16) unsigned char smem[];
  WgCtx<M, T, IPW, MAXM> C(P, smem);
  T a[MAXM], c2[MAXM];
  const size_t Lrow = P.L;
  auto load_im = [&](T* reg, const T* src) {  // instance-major [B][L] (no padding) -> registers
#pragma unroll
    for (int m = 0; m < MAXM; ++m) {
      const int e = C.elem(m);
      reg[m] = (C.valid && e < P.L) ? src[size_t(C.b) * Lrow + e] : T(0);
    }
  };
  auto store_im = [&](T* dst, const T* reg) {
    if (!C.valid) return;
#pragma unroll
    for (int m = 0; m < MAXM; ++m) {
      const int e = C.elem(m);
      if (e < P.L) dst[size_t(C.b) * Lrow + e] = reg[m];
    }
  };
  if (P.mode == WG_HOOK_F) {  // F_func(ret, U, x, t)
    C.load_common(P.U);
    __syncthreads();
    load_im(a, P.hook_in0);
    C.reg_to_lds(C.S.U, a);
    __syncthreads();
    C.template f_eval<false, F_PLAIN>(C.S.xs, P.hook_dtau, C.S.W, false);
    __syncthreads();
    C.lds_to_reg(a, C.S.W);
    store_im(P.hook_out, a);
    return;
  }
  if (P.mode == WG_HOOK_PREPARE) {  // cgmres.hpp:83-96
    C.load_common(P.U);
    __syncthreads();
    C.template preamble<false>(a, c2);
    if (P.hook_out) store_im(P.hook_out, a);
    if (!C.fh_hbm()) {  // (with fh_hbm the preamble has already stored the row)
      C.lds_to_reg(a, C.S.Fh);
      C.reg_to_row(P.Fh, P.Lg, a);
    }
    if (C.valid && C.r < M::NX) P.xdxh[size_t(C.b) * M::NX + C.r] = C.S.xh[C.r * IPW + C.inst];
    return;
  }
  // AX / GMRES: state left by PREPARE
  C.load_common(P.U);
  if (!C.fh_hbm()) C.load_row_to_lds(C.S.Fh, P.Fh);
  if (C.valid && C.r < M::NX) C.S.xh[C.r * IPW + C.inst] = P.xdxh[size_t(C.b) * M::NX + C.r];
  for (int q = C.r; q < P.Hp; q += 16) C.S.H[C.inst * P.Hp + q] = T(0);
  load_im(a, P.hook_in0);
  C.publish_direction(a);
  __syncthreads();
  C.ax(false);
  if (P.mode == WG_HOOK_AX) {
    C.lds_to_reg(a, C.S.W);
    store_im(P.hook_out, a);
    return;
  }
  load_im(c2, P.hook_in1);
  T ax0h[MAXM];
  C.lds_to_reg(ax0h, C.S.W);
  C.gmres(a, c2, ax0h);
  store_im(P.hook_out, a);
  C.store_status();
}

}  // namespace cgm
