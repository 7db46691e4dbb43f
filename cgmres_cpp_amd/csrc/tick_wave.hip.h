// Variant 4 ("wave"): ONE WAVEFRONT PER CONTROLLER — the latency mapping.
//
// The "wg" mapping (tick_wg.hip.h) gives an instance 16 lanes and walks the horizon recurrences of cgmres.hpp:132-153
// serially on one wave for 16 instances at a time: a tick is a chain of ~30 k instruction issues that does not get
// shorter when there are fewer instances than the GPU has lanes for (512 controllers per GPU: 116 us/tick, 224 of 256
// CUs idle).  Here a controller owns all 64 lanes of a wave and the horizon lives ACROSS the lanes:
//   * lane s carries stage s: its controls U[s*dim_u .. ], every length-L vector of the solver as dim_u elements per
//     lane (the reference's stage-major layout, cgmres.hpp:13, read 24 contiguous bytes per lane), the stage's state,
//     trig values, parameters and costate coefficients — all in registers.  No stage table, no row arrays, no barrier;
//   * the recurrences over the stage index become SCANS over the lanes (wave_scan.hip.h: six DPP steps):
//       - x0, x2 of the pendulum obey a linear recurrence driven by u0 (model.hpp:38,40): a geometric scan and a prefix sum;
//       - x1, x3 are nonlinear (model.hpp:39,41): Newton's method on the whole trajectory at once — evaluate the stage
//         function and its Jacobian at the current guess on every lane, solve the linearised recurrence for the
//         correction with a scan of 2 x 2 affine maps, repeat until the correction is below 1e-10 (quadratic
//         convergence: the next one would be < 1e-18).  The guess is the unperturbed trajectory of the same tick
//         (F(U, x+hf, t+h), cgmres.hpp:88): a forward-difference direction moves the controls by h*|v| <= 2e-3.  The
//         trig values of a guess are ROTATED from the base trajectory's through the (small) angle difference;
//       - the costate recurrence (cgmres.hpp:146-153) is linear with the stage coefficients of models.hip.h and splits
//         for the pendulum into a 2 x 2 affine scan (l1, l3), a prefix sum (l0) and a geometric scan (l2), on the
//         MIRRORED lanes (stage dv-1-m on lane m: one crossbar gather of the six coefficients, one of dF back);
//     every scan result equals the serial recurrence up to rounding (tests bound it against the oracle);
//   * the three sweeps in front of the solve (cgmres.hpp:83-96 and the residual of the warm start) start from states the
//     previous tick does not bracket, so their state sweeps run serially — concurrently on three DPP quads with the quad
//     stage of the wg mapping (PendulumDev::quad_stage) — through a small LDS table;
//   * the Krylov basis (k_max + 1 vectors x 3 elements per lane) stays in REGISTERS; Gram-Schmidt is 3 + 3 fused
//     multiply-adds around a wave-wide sum.  Every branch of the solve (gmres.hpp:39-41, 63-65, 93-95) is wave-uniform:
//     an instance that converges early simply leaves the loop (no masking, no waiting for workgroup mates).
// Per tick ~15 k instructions on the one wave instead of ~30 k on the critical wave of the wg mapping.
// Statement order per instance follows cgmres.hpp:78-175 / gmres.hpp:28-112; sums associate as scans / butterflies.
#pragma once
#include <hip/hip_runtime.h>

#include <cfloat>
#include <type_traits>

#include "tick_wg.hip.h"  // WgParams, helpers
#include "wave_scan.hip.h"

namespace cgm {

// what a model's scans need to know about the lane
struct WaveLane {
  int lane, dv, msrc;  // msrc: the mirrored lane of the costate scans (stage dv-1-m on lane m; an involution on [0, dv))
  bool in_hor;         // the lane has a stage with controls
};

// which models have the scans above, and how.  NONLINEAR: the state equation is not affine in x — Newton on the trajectory
// from a base trajectory, serial sweeps in front of the solve (the code in the kernel); otherwise the model supplies
//   state_scan  (L, u, xinit, dtau, x)            x(s) of every lane's stage from the lane's controls: ONE scan, exact
//   costate_scan(L, x, u, p, dtau, phi, dF)       costate recurrence + dH/du pieces from the lane's stage
template <class M>
struct WaveOps : std::false_type {  // (no wave scans: user models and whatever is not specialised below)
  static constexpr bool NONLINEAR = false;
  static constexpr int LDS_EXTRA = 0;
};

template <class T>
struct WaveOps<PendulumDev<T>> : std::true_type {
  using M = PendulumDev<T>;
  static constexpr bool NONLINEAR = true;
  using TickConsts = GeoPowers<T>;  // powers of 1 - dtau As (x2 and l2 are geometric recurrences)
  static constexpr int LDS_EXTRA = 0;
  static __device__ __forceinline__ void make_consts(TickConsts& G, T dtau, const WaveLane&, T*) { G.make(T(1) - dtau * M::As); }
  // sin/cos of (base angle + dl) from the base pair, |dl| <= sqrt(rot_zmax) (Taylor polynomials of the wg mapping's
  // rotation stage, MathCtx::rot_sin / rot_cos)
  template <class MC>
  static __device__ __forceinline__ void rotate(T sb, T cb, T dl, T* s, T* c, const MC& mc) {
    constexpr int NRS = MC::NRS, NRC = MC::NRC;
    const T z = dl * dl;
    T ps = mc.rot_sin(NRS - 1), pc = mc.rot_cos(NRC - 1);
#pragma unroll
    for (int i = NRS - 2; i >= 0; --i) ps = fma_t(z, ps, mc.rot_sin(i));
#pragma unroll
    for (int i = NRC - 2; i >= 0; --i) pc = fma_t(z, pc, mc.rot_cos(i));
    const T sn = fma_t(z * dl, ps, dl);  // sin dl
    const T cm1 = z * pc;                // cos dl - 1
    *s = sb + fma_t(sb, cm1, cb * sn);
    *c = cb + fma_t(cb, cm1, -(sb * sn));
  }
};

// semiactive_damper/model.hpp:36-55: x0' = x1, x1' = a x0 + b u0 x1 — affine in x for given controls: the state sweep IS
// one scan of 2 x 2 affine maps (exact up to rounding, no iteration, no serial sweep anywhere), and so is the costate
// recurrence (SemiactiveDev::costate_step).
template <class T>
struct WaveOps<SemiactiveDev<T>> : std::true_type {
  using M = SemiactiveDev<T>;
  static constexpr bool NONLINEAR = false;
  struct TickConsts {};
  static constexpr int LDS_EXTRA = 0;
  static __device__ __forceinline__ void make_consts(TickConsts&, T, const WaveLane&, T*) {}
  static __device__ __forceinline__ void state_scan(const WaveLane& L, const T* u, const T* xinit, T dtau, const TickConsts&,
                                                    T* x) {
    // y' = y + D y,  D = [[0, dtau], [dtau a, dtau b u0]]; lane 0 applies its map to the initial state
    T D[4] = {T(0), dtau, dtau * M::a, dtau * M::b * u[0]};
    const bool first = L.lane == 0;
    const T y0 = first ? xinit[0] : T(0), y1 = first ? xinit[1] : T(0);
    T c[2] = {y0 + D[1] * y1, y1 + fma_t(D[2], y0, D[3] * y1)};
    scan_aff2(D, c);
    x[0] = wave_shift_up(c[0], xinit[0]);
    x[1] = wave_shift_up(c[1], xinit[1]);
  }
  static __device__ __forceinline__ void costate_scan(const WaveLane& L, const T* x, const T* u, const T*, T dtau,
                                                      const TickConsts&, T* phi, T* dF) {
    T bw[M::NBW], trig[1] = {T(0)};
    M::stage_coeffs(bw, phi, x, u, nullptr, trig, dtau);
    T xT[M::NX], lT[M::NX];
#pragma unroll
    for (int c = 0; c < M::NX; ++c) xT[c] = wave_bcast(x[c], L.dv);
    M::dPhidx(lT, xT, nullptr);
    T mb[M::NBW];
#pragma unroll
    for (int c = 0; c < M::NBW; ++c) mb[c] = wave_gather(bw[c], L.msrc);
    const bool first = L.lane == 0;
    const T i0 = first ? lT[0] : T(0), i1 = first ? lT[1] : T(0);
    // n0 = l0 + bw2 + dtau a l1,  n1 = l1 + bw3 + dtau l0 + bw0 l1
    T D[4] = {T(0), dtau * M::a, dtau, mb[0]};
    T c[2] = {mb[2] + i0 + D[1] * i1, mb[3] + i1 + fma_t(D[2], i0, D[3] * i1)};
    scan_aff2(D, c);
    const T in1 = wave_shift_up(c[1], lT[1]);  // costate ENTERING the lane's stage
    dF[0] = wave_gather(mb[1] * in1, L.msrc);  // B^T l of the stage (model.hpp:52)
  }
};

// mass_spring_damper/model.hpp:36-64: linear and time-invariant — x' = x + dtau (A x + B u), l' = l + bw + dtau (Al l) with
// CONSTANT 4 x 4 matrices (A from dxdt :36-41, Al from dHdx :50-55 — not each other's transpose: the reference's
// k1*k2 / k1+k2 quirk).  Every composed matrix of a scan is a power of I + dtau A (resp. Al): one table of powers per
// horizon step and recurrence in LDS (power_table4, built once per tick), then a sweep is two VECTOR-only scans.
template <class T>
struct WaveOps<MsdDev<T>> : std::true_type {
  using M = MsdDev<T>;
  static constexpr bool NONLINEAR = false;
  static constexpr int TABLE = 32 * 16;              // scalars of one power table
  static constexpr int LDS_EXTRA = 4 * TABLE;        // state + costate tables for dtau_h and for dtau_0
  struct TickConsts {
    const T *st, *co;
  };
  static __device__ __forceinline__ void make_consts(TickConsts& G, T dtau, const WaveLane& L, T* lds) {
    const T A[16] = {T(0), T(0), T(1), T(0),                                                           // model.hpp:37
                     T(0), T(0), T(0), T(1),                                                           // :38
                     -(M::k1 * M::k2) / M::m1, M::k2 / M::m1, -(M::d1 + M::d2) / M::m1, M::d2 / M::m1,  // :39
                     M::k2 / M::m2, -M::k2 / M::m2, M::d2 / M::m2, -M::d2 / M::m2};                    // :40
    const T Al[16] = {T(0), T(0), -(M::k1 + M::k2) / M::m1, M::k2 / M::m2,    // :51
                      T(0), T(0), M::k2 / M::m1, -M::k2 / M::m2,              // :52
                      T(1), T(0), -(M::d1 + M::d2) / M::m1, M::d2 / M::m2,    // :53
                      T(0), T(1), M::d2 / M::m1, -M::d2 / M::m2};             // :54
    T D[16];
#pragma unroll
    for (int e = 0; e < 16; ++e) D[e] = dtau * A[e];
    power_table4(lds, D, L.lane);
#pragma unroll
    for (int e = 0; e < 16; ++e) D[e] = dtau * Al[e];
    power_table4(lds + TABLE, D, L.lane);
    G.st = lds, G.co = lds + TABLE;
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  }
  static __device__ __forceinline__ void state_scan(const WaveLane& L, const T* u, const T* xinit, T dtau, const TickConsts& G,
                                                    T* x) {
    const bool first = L.lane == 0;
    T y[4], c[4] = {T(0), T(0), L.in_hor ? dtau * (u[0] / M::m1) : T(0), L.in_hor ? dtau * (u[1] / M::m2) : T(0)};
#pragma unroll
    for (int r = 0; r < 4; ++r) y[r] = first ? xinit[r] : T(0);
#pragma unroll
    for (int r = 0; r < 4; ++r) {  // lane 0 applies its map to the initial state: c += (I + D) x(0)
      T a = c[r] + y[r];
#pragma unroll
      for (int k = 0; k < 4; ++k) a = fma_t(G.st[4 * r + k], y[k], a);
      c[r] = a;
    }
    scan_const4(c, G.st, L.lane);
#pragma unroll
    for (int r = 0; r < 4; ++r) x[r] = wave_shift_up(c[r], xinit[r]);
  }
  static __device__ __forceinline__ void costate_scan(const WaveLane& L, const T* x, const T* u, const T* p, T dtau,
                                                      const TickConsts& G, T* phi, T* dF) {
    T bw[M::NBW], trig[1] = {T(0)};
    M::stage_coeffs(bw, phi, x, u, p, trig, dtau);
    T xT[M::NX], pT[M::NP], lT[M::NX];
#pragma unroll
    for (int c = 0; c < M::NX; ++c) xT[c] = wave_bcast(x[c], L.dv);
#pragma unroll
    for (int j = 0; j < M::NP; ++j) pT[j] = wave_bcast(p[j], L.dv);
    M::dPhidx(lT, xT, pT);
    const bool first = L.lane == 0;
    T c[4], y[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) c[r] = wave_gather(bw[r], L.msrc), y[r] = first ? lT[r] : T(0);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      T a = c[r] + y[r];
#pragma unroll
      for (int k = 0; k < 4; ++k) a = fma_t(G.co[4 * r + k], y[k], a);
      c[r] = a;
    }
    scan_const4(c, G.co, L.lane);
    const T in2 = wave_shift_up(c[2], lT[2]), in3 = wave_shift_up(c[3], lT[3]);  // costate ENTERING the lane's stage
    dF[0] = wave_gather(in2 / M::m1, L.msrc);  // model.hpp:58-59
    dF[1] = wave_gather(in3 / M::m2, L.msrc);
  }
};

// LDS of one wave: the table of the serial sweeps and two control rows for them (nothing else lives in memory)
template <class M, class T>
struct WaveLds {
  static constexpr int TAB_W = 2 * M::NX;  // per (sweep, stage): one pair (state component, trig value) per quad lane
  static __host__ __device__ int row_len(int dv) { return (dv * M::NU + 2) & ~1; }
  static __host__ __device__ int tab_len(int dv) { return (dv + 1) * TAB_W; }  // one sweep's table
  // region A: the tables of sweeps #1 / #2 — dead once the solve starts — and, over them, the Krylov basis of the kernels
  // that keep it in LDS ([vector][component][stage]); then the table of sweep #3 / of the serial fall-back, the two rows
  static __host__ __device__ int region_a(int dv, int kmax) {
    const int t = 2 * tab_len(dv), v = (kmax + 1) * M::NU * dv;
    return ((t > v ? t : v) + 1) & ~1;
  }
  // (models that are affine in x have no serial sweep and keep the basis in registers: only their own tables are in LDS)
  static __host__ __device__ size_t count_T(int dv, int kmax) {
    const size_t sweeps = WaveOps<M>::NONLINEAR ? size_t(region_a(dv, kmax)) + tab_len(dv) + 2 * row_len(dv) + 2 : 0;
    return sweeps + WaveOps<M>::LDS_EXTRA + 2;
  }
  static __host__ __device__ size_t bytes(int dv, int kmax, int waves) {
    return ((count_T(dv, kmax) * sizeof(T) + 15) & ~size_t(15)) * waves;
  }
};

constexpr int WAVE_NEWTON_MAX = 8;

// f(integral_constant<int, I>) for I = LO .. HI-1 (ascending) / HI-1 .. LO (descending): loops whose index must be a constant
template <int LO, int HI, class F>
__device__ __forceinline__ void static_for(F&& f) {
  if constexpr (LO < HI) {
    f(std::integral_constant<int, LO>{});
    static_for<LO + 1, HI>(f);
  }
}
template <int LO, int HI, class F>
__device__ __forceinline__ void static_for_down(F&& f) {
  if constexpr (LO < HI) {
    f(std::integral_constant<int, HI - 1>{});
    static_for_down<LO, HI - 1>(f);
  }
}

// b in the lanes of `mask`, a elsewhere — as two v_cndmask on a scalar mask.  (Written as a ternary on the lane index the
// compiler turned the choice into exec-mask branches inside the stage loop of the serial sweeps: ~30 cycles per taken branch.)
__device__ __forceinline__ double lane_select(unsigned long long mask, double a, double b) {
  int lo, hi;
  asm("v_cndmask_b32 %0, %1, %2, %3" : "=v"(lo) : "v"(__double2loint(a)), "v"(__double2loint(b)), "s"(mask));
  asm("v_cndmask_b32 %0, %1, %2, %3" : "=v"(hi) : "v"(__double2hiint(a)), "v"(__double2hiint(b)), "s"(mask));
  return __hiloint2double(hi, lo);
}

// VLDS (experimental, NOT instantiated by the library): the Krylov basis in LDS instead of registers, compiled for two waves
// per SIMD.  Measured / estimated in DESIGN.md 4.5: capped at 256 registers the pendulum kernel spills 258 VGPRs, and two
// waves on a SIMD are bound by VALU issue (2 x 17 k instructions x 4 cycles = 58 us per round) — no gain at 4096 controllers.
template <class M, class T, int KM, int WPB, bool VLDS = false>
__global__ __launch_bounds__(64 * WPB) __attribute__((amdgpu_waves_per_eu(VLDS ? 2 : 1, VLDS ? 2 : 1))) void tick_wave_kernel(
    WgParams<T> P) {
  static_assert(WaveOps<M>::value, "model without wave scans");
  static_assert(std::is_same<T, double>::value, "the Newton thresholds are set for fp64");
  using W = WaveOps<M>;
  using Lds = WaveLds<M, T>;
  constexpr int NX = M::NX, NU = M::NU, NP = M::NP, NC = M::NC, NUL = M::NUL;
  constexpr bool NONLIN = W::NONLINEAR;
  extern __shared__ __align__(16) unsigned char smem[];
  const int lane = threadIdx.x & 63;
  const int wv = __builtin_amdgcn_readfirstlane(int(threadIdx.x) >> 6);
  const int b = blockIdx.x * WPB + wv;  // this wave's controller
  if (b >= P.B) return;                 // (no barrier anywhere in this kernel)
  const int dv = P.dv, kmax = P.kmax, k1 = kmax + 1;
  // ---- LDS of this wave
  T* const base = reinterpret_cast<T*>(smem + size_t(wv) * Lds::bytes(dv, kmax, 1));
  T* const Vl = base;                                   // [kmax+1][NU][dv] (VLDS kernels), over the tables of sweeps 0, 1
  T* const tab2 = base + Lds::region_a(dv, kmax);       // table of sweep 2
  T* const wrow = tab2 + Lds::tab_len(dv);              // [2][row_len]
  auto tab_of = [&](int q) { return q == 2 ? tab2 : base + q * Lds::tab_len(dv); };
  const int rlen = Lds::row_len(dv);
  T* const lds_extra = W::NONLINEAR ? wrow + 2 * rlen + 2 : base;  // the model's own tables (WaveOps::LDS_EXTRA scalars)
  struct alignas(2 * sizeof(T)) WPair {
    T a, b;
  };
  auto wave_fence = [] { __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); };  // compiler ordering of LDS traffic

  typename M::template MathFor<false> mc;
  mc.init();
  CGM_STAMP(0, -1);
  const bool in_hor = lane < dv;    // the lane has a stage with controls
  const bool in_traj = lane <= dv;  // ... or the terminal stage
  const int msrc = in_hor ? dv - 1 - lane : lane;  // mirrored lane of the costate scans (an involution on [0, dv))
  const WaveLane WL{lane, dv, msrc, in_hor};

  // ---- controller state of this instance
  T U[NU], du[NU], Fh[NU], p[NP > 0 ? NP : 1], xs[NX];
  {
    const T* Ug = P.U + size_t(b) * P.Lg + lane * NU;
    const T* dg = P.dUdt + size_t(b) * P.Lg + lane * NU;
#pragma unroll
    for (int j = 0; j < NU; ++j) U[j] = in_hor ? Ug[j] : T(0), du[j] = in_hor ? dg[j] : T(0);
    if constexpr (NP > 0) {
      const T* pg = P.ptau + size_t(b) * NP * (dv + 1) + lane * NP;
#pragma unroll
      for (int j = 0; j < NP; ++j) p[j] = in_traj ? pg[j] : T(0);
    } else {
      p[0] = T(0);
    }
#pragma unroll
    for (int c = 0; c < NX; ++c) xs[c] = P.x_in[size_t(b) * NX + c];
  }
  int n_ax = 0, reason = 0, ksolve = 0;
  T xh[NX];

  // ---- phase 3 of a sweep on the lanes: costate recurrence + the costate part of dH/du (cgmres.hpp:145-161) from the
  //      stage's state / trig / controls.  Returns phi (costate-free part of dH/du) and dF (to be added to component 0).
  using TickConsts = typename W::TickConsts;
  auto backward = [&](const T* x, const T* trig, const T* u, T dtau, const TickConsts& G, T* phi, T* dF) {
   if constexpr (!NONLIN) {
    W::costate_scan(WL, x, u, p, dtau, G, phi, dF);
   } else {
    T bw[M::NBW];
    M::stage_coeffs(bw, phi, x, u, p, trig, dtau);
    // terminal costate from the terminal stage (lane dv), wave-uniform
    T xT[NX], pT[NP], lT[NX];
#pragma unroll
    for (int c = 0; c < NX; ++c) xT[c] = wave_bcast(x[c], dv);
#pragma unroll
    for (int j = 0; j < NP; ++j) pT[j] = wave_bcast(p[j], dv);
    M::dPhidx(lT, xT, pT);
    T mb[M::NBW];
#pragma unroll
    for (int c = 0; c < M::NBW; ++c) mb[c] = wave_gather(bw[c], msrc);
    const bool first = lane == 0;
    const T i0 = first ? lT[0] : T(0), i1 = first ? lT[1] : T(0), i2 = first ? lT[2] : T(0), i3 = first ? lT[3] : T(0);
    // (l1, l3):  n1 = l1 + bw5 + bw1 l3,  n3 = l3 + dtau l1 - dtau C22 l3     (PendulumDev::costate_step)
    T D[4] = {T(0), mb[1], dtau, -dtau * M::C22};
    T c[2] = {mb[5] + i1 + mb[1] * i3, i3 + fma_t(D[2], i1, D[3] * i3)};
    scan_aff2(D, c);
    const T in3 = wave_shift_up(c[1], lT[3]);  // costate ENTERING the lane's stage (l1 feeds only this pair)
    // l0:  n0 = l0 + (bw4 + bw0 l3)
    const T l0 = scan_sum(fma_t(mb[0], in3, mb[4]) + i0);
    const T in0 = wave_shift_up(l0, lT[0]);
    // l2:  n2 = (1 - dtau As) l2 + (bw2 l3 + dtau l0)
    const T l2 = scan_geo(fma_t(mb[2], in3, fma_t(dtau, in0, G.pw[0] * i2)), G);
    const T in2 = wave_shift_up(l2, lT[2]);
    const T dFm = fma_t(mb[3], in3, in2 * M::Bs);  // B^T l of the stage (model.hpp:59)
    dF[0] = wave_gather(dFm, msrc);
   }
  };
  // post-processing of a sweep result (cgmres.hpp:94-96, :173-174)
  auto finish = [&](int mode, const T* phi, const T* dF, T* out) {
    const T sc_phi = mode == F_RHS ? P.one_m_zh : T(1.0);
    const T sc = mode == F_PLAIN ? T(1.0) : (mode == F_RHS ? P.one_m_zh * P.inv_h : P.inv_h);
#pragma unroll
    for (int j = 0; j < NU; ++j) {
      T rj = phi[j];
      if (mode != F_PLAIN) rj = (rj * sc_phi - Fh[j]) * P.inv_h;
      if (j < NUL) rj = fma_t(dF[j < NUL ? j : 0], sc, rj);
      out[j] = in_hor ? rj : T(0);
    }
  };

  // ---- serial state sweeps on DPP quads: quad q of [q_lo, q_hi) runs sweep q; x(s), trig(s) -> tab[q][s]
  //      q = 0: (U, x+hf, dtau_h)   q = 1: (U, x, dtau_0)   q = 2: (row 1 of wrow, x+hf, dtau_h)
  //      Stage forms as in WgCtx::sweep_state: 0 = trig value ROTATED from the previous stage's through the exact angle
  //      increment (PendulumDev::quad_stage_rot), 1 = fresh evaluation per stage, 2 = fresh with the library sin/cos.
  //      A sweep whose increments leave the rotation's range is redone one form up, and the wave then stays with fresh
  //      evaluations for ROT_HOLD sweeps (fast motion would otherwise pay for a failed rotation pass every tick).
  constexpr int ROT_HOLD = 64;
  int rot_hold = 0;
  auto serial_sweeps = [&](int q_lo, int q_hi, T dtau_h, T dtau_0) {
   if constexpr (NONLIN) {
    auto run = [&](auto mode_tag) -> int {  // 0 = done, 1 = an increment left the rotation range, 2 = argument beyond the fast trig range
      constexpr int MODE = decltype(mode_tag)::value;
      const int q = lane >> 2, rho = lane & 3;
      T amax = T(0);
      int zmax = 0;
      if (q >= q_lo && q < q_hi) {
        typename M::QuadLane Q;
        Q.init(rho, mc);
        T x[NX];
#pragma unroll
        for (int c = 0; c < NX; ++c) x[c] = q == 1 ? xs[c] : xh[c];
        const T dtau = q == 1 ? dtau_0 : dtau_h, dtau1 = Q.sg * dtau;
        const T* pu = wrow + (q == 2 ? rlen : 0);
        WPair* pt = reinterpret_cast<WPair*>(tab_of(q)) + rho;
        T v = M::template quad_begin<MODE == 2>(x, Q, mc, &amax);
        T argp = M::quad_arg(x, Q);
        T ua = pu[0];
        // What the table keeps per stage: the trig value of every quad lane, x1 (lane 2: the d-lanes hold -x1) and x3
        // (lane 3).  x0 and x2 obey a linear recurrence in u0 and are re-derived on the lanes by two scans (lin_states).
        constexpr unsigned long long LANE3 = 0x8888888888888888ull;
        auto stage = [&](int o, T u0) {
          pt[o * NX] = {lane_select(LANE3, x[1], x[3]), v};
          if constexpr (MODE == 0)
            M::quad_stage_rot(x, v, argp, u0, dtau, dtau1, Q, mc, &zmax);
          else
            M::template quad_stage<MODE == 2>(x, v, u0, dtau, dtau1, Q, mc, &amax);
        };
        int s = 0;
        for (; s + 4 <= dv; s += 4) {  // (the word after the last stage is the row's pad)
          const T ub = pu[NU];
          stage(0, ua);
          const T uc = pu[2 * NU];
          stage(1, ub);
          const T ud = pu[3 * NU];
          stage(2, uc);
          ua = pu[4 * NU];
          stage(3, ud);
          pt += 4 * NX, pu += 4 * NU;
        }
        for (; s < dv; ++s) {
          const T ub = pu[NU];
          stage(0, ua);
          ua = ub;
          pt += NX, pu += NU;
        }
        pt[0] = {lane_select(LANE3, x[1], x[3]), v};
      }
      if (MODE == 0 && __any(M::quad_rot_bad(zmax))) return 1;
      return __any(M::quad_arg_bad(amax)) ? 2 : 0;
    };
    int st = rot_hold > 0 ? 1 : run(std::integral_constant<int, 0>{});
    if (__builtin_expect(st == 1, 0)) {
      if (rot_hold == 0) rot_hold = ROT_HOLD;
      st = run(std::integral_constant<int, 1>{});
    }
    if (__builtin_expect(st == 2, 0)) run(std::integral_constant<int, 2>{});
    if (rot_hold > 0) --rot_hold;
    wave_fence();
   }
  };
  // the lane's stage of sweep q from the table: x[4] and {sin d, cos d, sin x1, cos x1}
  // x0(s), x2(s) of every lane's stage from the stage's control u0:  x2' = (1 - dtau As) x2 + dtau Bs u0,
  // x0' = x0 + dtau x2  (model.hpp:38,40) — a geometric scan and a prefix sum
  auto lin_states = [&](T u0, const T* xinit, T dtau, const TickConsts& G, T* x0, T* x2) {
   if constexpr (NONLIN) {
    const bool first = lane == 0;
    const T e2 = (in_hor ? (dtau * M::Bs) * u0 : T(0)) + (first ? G.pw[0] * xinit[2] : T(0));
    *x2 = wave_shift_up(scan_geo(e2, G), xinit[2]);
    *x0 = wave_shift_up(scan_sum(dtau * *x2 + (first ? xinit[0] : T(0))), xinit[0]);
   }
  };
  auto read_tab = [&](int q, T* x, T* tr4) {  // (x[0], x[2] are the caller's: lin_states)
   if constexpr (NONLIN) {
    const WPair* pt = reinterpret_cast<const WPair*>(tab_of(q) + (in_traj ? lane : dv) * Lds::TAB_W);
    const WPair e0 = pt[0], e1 = pt[1], e2 = pt[2], e3 = pt[3];
    x[1] = e2.a, x[3] = e3.a;
    tr4[0] = e0.b, tr4[1] = e1.b, tr4[2] = e2.b, tr4[3] = e3.b;
   }
  };

  const int nt = P.n_ticks;
  for (int tk = 0; tk < nt; ++tk) {
    const bool last = tk + 1 == nt;
    const T dtau_h = P.dtau_tab[2 * tk], dtau_0 = P.dtau_tab[2 * tk + 1];
    if constexpr (NP > 0) {
      if (P.ptau_seq) {  // set_ptau before this tick (cgmres.hpp:36-39)
        const T* src = P.ptau_seq + size_t(tk) * P.pseq_tick + size_t(b) * P.pseq_inst + lane * NP;
#pragma unroll
        for (int j = 0; j < NP; ++j) p[j] = in_traj ? src[j] : T(0);
      }
    }
    CGM_STAMP(0, 11);
    TickConsts Gh, G0;  // per-tick constants of the model's scans for the two horizon steps (pendulum: powers of 1 - dtau As)
    W::make_consts(Gh, dtau_h, WL, lds_extra);
    W::make_consts(G0, dtau_0, WL, lds_extra + W::LDS_EXTRA / 2);
    // ---- cgmres.hpp:83-85: x_dxh = x + h f(x, U_0)
    {
      T u0[NU], f[NX], tr[NC > 0 ? NC : 1];
#pragma unroll
      for (int j = 0; j < NU; ++j) u0[j] = wave_bcast(U[j], 0);
      M::dxdt(f, xs, u0, tr, mc);
#pragma unroll
      for (int c = 0; c < NX; ++c) xh[c] = f[c] * P.h + xs[c];
    }
    T xb[NX], tb[4];  // NONLINEAR: base trajectory of this tick (sweep #1) on the lanes
    T bb[NU], ax0[NU];
    if constexpr (!NONLIN) {
      // ---- the three sweeps in front of the solve, each ONE state scan + one costate scan (no serial sweep, no LDS)
      T x[NX], phi[NU], dF[NUL], uu[NU], trig[1] = {T(0)};
      W::state_scan(WL, U, xh, dtau_h, Gh, x);
      backward(x, trig, U, dtau_h, Gh, phi, dF);
      finish(F_PLAIN, phi, dF, Fh);  // cgmres.hpp:88
      W::state_scan(WL, U, xs, dtau_0, G0, x);
      backward(x, trig, U, dtau_0, G0, phi, dF);
      finish(F_RHS, phi, dF, bb);  // :91-96
#pragma unroll
      for (int j = 0; j < NU; ++j) uu[j] = du[j] * P.h + U[j];  // :168-169
      W::state_scan(WL, uu, xh, dtau_h, Gh, x);
      backward(x, trig, uu, dtau_h, Gh, phi, dF);
      finish(F_AX, phi, dF, ax0);  // :99 -> gmres.hpp:33
      (void)xb, (void)tb;
    } else {
    // ---- the three sweeps in front of the solve: #1 Fh = F(U, x+hf, t+h) (:88), #2 b (:91-96), #3 A*dUdt (:99 ->
    //      gmres.hpp:33), state sweeps side by side on three quads
    if (in_hor) {
#pragma unroll
      for (int j = 0; j < NU; ++j) {
        wrow[lane * NU + j] = U[j];
        wrow[rlen + lane * NU + j] = du[j] * P.h + U[j];  // cgmres.hpp:168-169
      }
    }
    if (lane == 0) wrow[dv * NU] = T(0), wrow[rlen + dv * NU] = T(0);
    wave_fence();
    CGM_STAMP(0, 0);
    serial_sweeps(0, 3, dtau_h, dtau_0);
    CGM_STAMP(0, 1);
    {
      read_tab(0, xb, tb);
      lin_states(U[0], xh, dtau_h, Gh, &xb[0], &xb[2]);
      T phi[NU], dF[NUL], trig[3] = {tb[0], tb[1], tb[3]};
      backward(xb, trig, U, dtau_h, Gh, phi, dF);
      finish(F_PLAIN, phi, dF, Fh);
    }
    {
      T x[NX], t4[4];
      read_tab(1, x, t4);
      lin_states(U[0], xs, dtau_0, G0, &x[0], &x[2]);
      T phi[NU], dF[NUL], trig[3] = {t4[0], t4[1], t4[3]};
      backward(x, trig, U, dtau_0, G0, phi, dF);
      finish(F_RHS, phi, dF, bb);
    }
    {
      T x[NX], t4[4], uu[NU];
      read_tab(2, x, t4);
#pragma unroll
      for (int j = 0; j < NU; ++j) uu[j] = du[j] * P.h + U[j];
      lin_states(uu[0], xh, dtau_h, Gh, &x[0], &x[2]);
      T phi[NU], dF[NUL], trig[3] = {t4[0], t4[1], t4[3]};
      backward(x, trig, uu, dtau_h, Gh, phi, dF);
      finish(F_AX, phi, dF, ax0);
    }
    }  // NONLINEAR preamble

    CGM_STAMP(0, 2);
    // ---- Ax_func (cgmres.hpp:164-175) of the direction `dir`: Newton on the trajectory, from the base trajectory
    auto ax = [&](const T* dir, T* out) {
      T u[NU];
#pragma unroll
      for (int j = 0; j < NU; ++j) u[j] = dir[j] * P.h + U[j];
      // A direction that the rounding of U + h*dir absorbs completely leaves the controls — and with them F — unchanged
      // bit for bit: A*dir = 0 exactly, the reference's breakdown case (gmres.hpp:63-65).  Newton's fixed point agrees
      // with the base trajectory only up to rounding, so that case is answered here.
      bool moved = false;
#pragma unroll
      for (int j = 0; j < NU; ++j) moved = moved || u[j] != U[j];
      if (__builtin_expect(!__any(in_hor && moved), 0)) {
#pragma unroll
        for (int j = 0; j < NU; ++j) out[j] = T(0);
        return;
      }
      if constexpr (!NONLIN) {
        T x[NX], phi[NU], dF[NUL], trig[1] = {T(0)};
        W::state_scan(WL, u, xh, dtau_h, Gh, x);
        CGM_STAMP(0, 3);
        backward(x, trig, u, dtau_h, Gh, phi, dF);
        finish(F_AX, phi, dF, out);
        CGM_STAMP(0, 5);
      } else {
      T x0, x2;
      lin_states(u[0], xh, dtau_h, Gh, &x0, &x2);
      const T Pq = M::A32 * x2 * x2, Qq = M::A32a * x2 - M::A32b * u[0];
      const T dx0 = x0 - xb[0];
      T y1 = xb[1], y3 = xb[3];
      CGM_STAMP(0, 3);
      T sd = tb[0], cd = tb[1], s1 = tb[2], c1 = tb[3];
      bool converged = false;
      const T rmax = T(0.9) * sqrt_t<T>(T(decltype(mc)::rot_zmax));
      for (int it = 0; it < ((P.wave_dbg & 2) ? 0 : WAVE_NEWTON_MAX); ++it) {
        const T d1 = y1 - xb[1], dd = dx0 - d1;
        if (__builtin_expect((P.wave_dbg & 1) || __any(in_traj && (!(abs_t(dd) <= rmax) || !(abs_t(d1) <= rmax))), 0)) {
          mc.sincos_pair(x0 - y1, y1, &sd, &cd, &s1, &c1);  // far from the base trajectory: fresh evaluation
        } else {
          W::rotate(tb[0], tb[1], dd, &sd, &cd, mc);
          W::rotate(tb[2], tb[3], d1, &s1, &c1, mc);
        }
        // model.hpp:41 and its derivative in x1
        const T g = fma_t(Pq, sd, fma_t(M::A52, s1, fma_t(Qq, cd, M::C22 * (x2 - y3))));
        const T J1 = fma_t(Qq, sd, fma_t(M::A52, c1, -(Pq * cd)));
        const T t1 = fma_t(dtau_h, y3, y1), t3 = fma_t(dtau_h, g, y3);
        // defect of the recurrence at the lane's stage, and the linearised step that leads to it
        T c[2] = {wave_shift_up(t1, xh[1]) - y1, wave_shift_up(t3, xh[3]) - y3};
        T D[4] = {T(0), dtau_h, dtau_h * wave_shift_up(J1, T(0)), -dtau_h * M::C22};
        scan_aff2(D, c);
        y1 += c[0], y3 += c[1];
        CGM_STAMP(0, 4);
        const T th = T(1e-10);
        if (!__any(in_traj && (abs_t(c[0]) > th * (T(1) + abs_t(y1)) || abs_t(c[1]) > th * (T(1) + abs_t(y3))))) {
          // the correction is below the square root of the rounding level: trig values to first order, done
          const T nsd = fma_t(cd, -c[0], sd), ncd = fma_t(sd, c[0], cd);
          const T ns1 = fma_t(c1, c[0], s1), nc1 = fma_t(s1, -c[0], c1);
          sd = nsd, cd = ncd, s1 = ns1, c1 = nc1;
          converged = true;
          break;
        }
      }
      T x[NX] = {x0, y1, x2, y3};
      T trig[3] = {sd, cd, c1};
      if (__builtin_expect(!converged, 0)) {  // (wave-uniform) Newton did not settle: the serial sweep
        if (in_hor) {
#pragma unroll
          for (int j = 0; j < NU; ++j) wrow[rlen + lane * NU + j] = u[j];
        }
        wave_fence();
        serial_sweeps(2, 3, dtau_h, dtau_0);
        T t4[4];
        read_tab(2, x, t4);  // (x[0], x[2] stay the scans' values)
        trig[0] = t4[0], trig[1] = t4[1], trig[2] = t4[3];
      }
      T phi[NU], dF[NUL];
      CGM_STAMP(0, 12);
      backward(x, trig, u, dtau_h, Gh, phi, dF);
      finish(F_AX, phi, dF, out);
      CGM_STAMP(0, 5);
      }  // NONLINEAR
    };

    // ---- Gmres::gmres (gmres.hpp:28-112): basis, Hessenberg, reflectors and residual vector all in registers.
    //      H is kept by ROWS over the lanes — lane j holds H(j, c) in Hrow[c] — and rho_e likewise (lane j holds
    //      rho_e[j] in `e`): the triangular solve at the end (gmres.hpp:100-107) then needs no memory at all.  The column
    //      of the running iteration and the reflectors are wave-uniform values.
    T V[VLDS ? 1 : KM + 1][NU];
    auto put_v = [&](auto qc, const T* v) {  // basis vector q <- v
      constexpr int q = decltype(qc)::value;
      if constexpr (VLDS) {
        if (in_hor) {
#pragma unroll
          for (int j = 0; j < NU; ++j) Vl[(q * NU + j) * dv + lane] = v[j];
        }
      } else {
#pragma unroll
        for (int j = 0; j < NU; ++j) V[q][j] = v[j];
      }
    };
    auto get_v = [&](auto qc, T* v) {
      constexpr int q = decltype(qc)::value;
      if constexpr (VLDS) {
#pragma unroll
        for (int j = 0; j < NU; ++j) v[j] = Vl[(q * NU + j) * dv + (in_hor ? lane : 0)];
        if (!in_hor) {
#pragma unroll
          for (int j = 0; j < NU; ++j) v[j] = T(0);
        }
      } else {
#pragma unroll
        for (int j = 0; j < NU; ++j) v[j] = V[q][j];
      }
    };
    T vcur[NU], w[NU];
    T Hrow[KM];
    T gr0 = T(0), gr1 = T(0), gr2 = T(0);  // reflector i = (gr0, gr1, gr2) of LANE i (gmres.hpp:81-83)
    T e = T(0), ek = T(0), hsub = T(0);
#pragma unroll
    for (int c = 0; c < KM; ++c) Hrow[c] = T(0);
    bool active = true;
    reason = 0, n_ax = 0, ksolve = 0;
    auto dot = [&](const T* a, const T* c) {
      T s = a[0] * c[0];
#pragma unroll
      for (int j = 1; j < NU; ++j) s = fma_t(a[j], c[j], s);
      return wave_sum(s);
    };
    {
#pragma unroll
      for (int j = 0; j < NU; ++j) vcur[j] = bb[j] - ax0[j];  // gmres.hpp:33-34
      const T rho0 = sqrt_t<T>(dot(vcur, vcur));               // :37
      ek = rho0;
      e = lane == 0 ? rho0 : T(0);
      // (decisions go through __any: wave-uniform by construction, and the compiler then keeps them in scalar registers)
      if (__any(!finite_t(rho0))) active = false, reason = 4;         // CGMRES_HIP_EXIT_NONFINITE
      if (active && __any(rho0 < P.tol)) active = false, reason = 2;  // :39-41
      if (active) {
        const T inv = T(1.0) / rho0;  // :44
#pragma unroll
        for (int j = 0; j < NU; ++j) vcur[j] = vcur[j] * inv;
        put_v(std::integral_constant<int, 0>{}, vcur);
      }
    }
    CGM_STAMP(0, 9);
    int nv = active ? 1 : 0;  // basis vectors stored
    // One Arnoldi iteration after the mat-vec, one static instance per k.  Returns 0 (go on), 1 (converged, :93-95),
    // 3 (breakdown, :63-65) or 4 (non-finite norm).
    auto iteration = [&](auto kc) -> int {
      constexpr int K = decltype(kc)::value;
      T hc[K + 1];
      T hraw = T(0);  // row `lane` of the column as Gram-Schmidt leaves it (what a breakdown exports)
      // modified Gram-Schmidt (:52-58), in order
      T vi[NU], vn[NU];
      get_v(std::integral_constant<int, 0>{}, vi);
      static_for<0, K + 1>([&](auto ic) {
        constexpr int i = decltype(ic)::value;
        if constexpr (i < K) get_v(std::integral_constant<int, i + 1>{}, vn);  // (LDS: requested a round ahead)
        const T hik = dot(vi, w);
#pragma unroll
        for (int j = 0; j < NU; ++j) w[j] = fma_t(-hik, vi[j], w[j]);
        hc[i] = hik;
        hraw = lane == i ? hik : hraw;
#pragma unroll
        for (int j = 0; j < NU; ++j) vi[j] = vn[j];
      });
      CGM_STAMP(0, 6);
      const T hn = sqrt_t<T>(dot(w, w));  // :60
      if (__any(abs_t(hn) < T(DBL_EPSILON) || !finite_t(hn))) {  // :63-65 breakdown: x untouched; non-finite: x <- NaN below
        hsub = hn;  // (exported as h(k+1,k) of the column that broke down, before any rotation)
        Hrow[K] = hraw;
        return __any(!finite_t(hn)) ? 4 : 3;
      }
      const T inv = T(1.0) / hn;  // :67
#pragma unroll
      for (int j = 0; j < NU; ++j) vcur[j] = w[j] * inv;
      put_v(std::integral_constant<int, K + 1>{}, vcur);
      nv = K + 2;
      CGM_STAMP(0, 7);
      // Hessenberg column K: stored reflectors, new reflector, residual rotation (:71-90)
      // (row i of the rotated column goes to lane i as it is produced)
      T a = hc[0], hrot = T(0);
#pragma unroll
      for (int i = 0; i < K; ++i) {
        const T c = hc[i + 1];
        const T q0 = wave_bcast(gr0, i), q1 = wave_bcast(gr1, i), q2 = wave_bcast(gr2, i);
        const T beta = (q0 * a + q1 * c) * q2;
        hrot = lane == i ? a - beta * q0 : hrot;
        a = c - beta * q1;
      }
      const T sigma = -(a < T(0.0) ? T(-1.0) : T(1.0)) * sqrt_t<T>(a * a + hn * hn);
      const T g0 = a - sigma, g1 = hn;
      const T g2 = T(2.0) / (g0 * g0 + g1 * g1);
      gr0 = lane == K ? g0 : gr0, gr1 = lane == K ? g1 : gr1, gr2 = lane == K ? g2 : gr2;
      hrot = lane == K ? sigma : hrot;
      const T beta = g0 * ek * g2;
      const T en = -beta * g1;
      e = lane == K ? ek - beta * g0 : (lane == K + 1 ? en : e);
      ek = en;
      Hrow[K] = hrot;
      CGM_STAMP(0, 8);
      return __any(abs_t(en) < P.tol) ? 1 : 0;
    };
    int k = 0;
    for (; active && k < kmax; ++k) {  // gmres.hpp:46
      ax(vcur, w);                     // :48
      n_ax = k + 1;
      int st = 0;
      switch (k) {
#define CGM_WCASE(n)                                                      \
  case n:                                                                 \
    if constexpr (n < KM) st = iteration(std::integral_constant<int, n>{}); \
    break;
        CGM_WCASE(0) CGM_WCASE(1) CGM_WCASE(2) CGM_WCASE(3) CGM_WCASE(4) CGM_WCASE(5) CGM_WCASE(6) CGM_WCASE(7)
        CGM_WCASE(8) CGM_WCASE(9) CGM_WCASE(10) CGM_WCASE(11) CGM_WCASE(12) CGM_WCASE(13) CGM_WCASE(14) CGM_WCASE(15)
        default: break;
#undef CGM_WCASE
      }
      if (st != 0) {
        reason = st;
        if (st == 1) ksolve = k;  // converged: column k is NOT used by the solve
        active = false;
        break;
      }
    }
    CGM_STAMP(0, 13);
    if (reason == 0) ksolve = kmax;  // natural exit: every column is used
    if (reason <= 1) {
      // back substitution (gmres.hpp:100-107), column-oriented over the lanes: step i (descending) turns e_i into
      // y_i = e_i / H_ii and every lane j < i subtracts H(j,i) y_i — for a fixed j in the reference's order
      const int ks = ksolve;
#pragma unroll
      for (int i = KM - 1; i >= 0; --i) {
        if (i < ks) {
          const T yi = wave_bcast(e, i) / wave_bcast(Hrow[i], i);
          e = lane < i ? fma_t(-Hrow[i], yi, e) : (lane == i ? yi : e);
        }
      }
      // x += V[:, 0:ks] y (gmres.hpp:110-111), accumulated j-ascending from 0
      T acc[NU];
#pragma unroll
      for (int j = 0; j < NU; ++j) acc[j] = T(0);
      static_for<0, KM>([&](auto qc) {
        constexpr int q = decltype(qc)::value;
        if (q < ks) {
          const T yq = wave_bcast(e, q);
          T vq[NU];
          get_v(qc, vq);
#pragma unroll
          for (int j = 0; j < NU; ++j) acc[j] = fma_t(vq[j], yq, acc[j]);
        }
      });
#pragma unroll
      for (int j = 0; j < NU; ++j) du[j] = du[j] + acc[j];
    }
    if (reason == 4) {
#pragma unroll
      for (int j = 0; j < NU; ++j) du[j] = in_hor ? quiet_nan<T>() : T(0);
    }
    CGM_STAMP(0, 10);
    // ---- U += dUdt*dt, u = U[0:dim_u]  (cgmres.hpp:102-109)
#pragma unroll
    for (int j = 0; j < NU; ++j) U[j] = U[j] + du[j] * P.dt;
    T unew[NU];
#pragma unroll
    for (int j = 0; j < NU; ++j) unew[j] = wave_bcast(U[j], 0);
    if (last) {
      if (in_hor) {
        T* Ug = P.U + size_t(b) * P.Lg + lane * NU;
        T* dg = P.dUdt + size_t(b) * P.Lg + lane * NU;
        T* fg = P.Fh + size_t(b) * P.Lg + lane * NU;
#pragma unroll
        for (int j = 0; j < NU; ++j) Ug[j] = U[j], dg[j] = du[j], fg[j] = Fh[j];
      }
      auto pick = [&](const T* v, int n) {  // element `lane` of a wave-uniform array
        T r = v[0];
        for (int q = 1; q < n; ++q) r = lane == q ? v[q] : r;
        return r;
      };
      if (lane < NU) P.u_out[size_t(b) * NU + lane] = pick(unew, NU);
      if (lane < NX) P.xdxh[size_t(b) * NX + lane] = pick(xh, NX);
      // status + small Krylov arrays (the layout of WgCtx::store_status) + the basis rows in the wg mapping's
      // pair-interleaved form (ctx_wg get_krylov undoes it)
      const int ks_all = k1 * k1 + k1 + 3 * kmax;
      T* dst = P.kry + size_t(b) * ks_all;
      for (int q = lane; q < k1 * k1; q += 64) {  // everything outside the upper triangle of the executed columns
        const int col = q / k1, row = q - col * k1;
        if (row > col || col >= kmax) dst[q] = (row == col + 1 && col + 1 == n_ax && reason == 3) ? hsub : T(0);
      }
#pragma unroll
      for (int c = 0; c < KM; ++c)
        if (c < kmax && lane <= c) dst[c * k1 + lane] = Hrow[c];
      if (lane < k1) dst[k1 * k1 + lane] = e;
      if (lane < kmax) {
        T* gd = dst + k1 * k1 + k1 + 3 * lane;
        gd[0] = gr0, gd[1] = gr1, gd[2] = gr2;
      }
      if (lane == 0) P.n_ax[b] = n_ax, P.reason[b] = reason;
      if (in_hor) {
        static_for<0, KM + 1>([&](auto qc) {
          constexpr int q = decltype(qc)::value;
          if (q < nv) {
            T* row = P.V + (size_t(b) * k1 + q) * P.Lv;
            T vq[NU];
            get_v(qc, vq);
#pragma unroll
            for (int j = 0; j < NU; ++j) {
              const int el = lane * NU + j, rr = el & 15, m = el >> 4;
              row[(m >> 1) * 32 + 2 * rr + (m & 1)] = vq[j];
            }
          }
        });
      }
    }
    if (P.x_next) {  // plant step of the example main loop (<example>/main.cpp:71-73)
      T f[NX], tr[NC > 0 ? NC : 1];
      M::dxdt(f, xs, unew, tr, mc);
#pragma unroll
      for (int c = 0; c < NX; ++c) xs[c] = xs[c] + f[c] * P.dt;
      if (last && lane < NX) {
        T r = xs[0];
#pragma unroll
        for (int q = 1; q < NX; ++q) r = lane == q ? xs[q] : r;
        P.x_next[size_t(b) * NX + lane] = r;
      }
    }
  }
  CGM_STAMP(0, 11);
#ifdef CGM_STAMPS
  cgm_stamp_flush();
#endif
}

}  // namespace cgm
