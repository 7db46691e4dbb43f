// Device-side problem definitions of the three models the reference ships.
// Each struct is the gfx950 counterpart of one reference `class Model`:
//   PendulumDev   — arm_type_inverted_pendulum/model.hpp:37-76 (≡ multiple_controller/model2.hpp)
//   MsdDev        — mass_spring_damper/model.hpp:36-108        (≡ multiple_controller/model1.hpp)
//   SemiactiveDev — semiactive_damper/model.hpp:36-69
// Interface (all static, all inlined into the sweep kernels, scalars in registers):
//   dxdt  (f, x, u, trig, mc)  state equation (mc = per-thread math context: pinned sin/cos constants);
//                              also fills `trig`, the per-stage values the costate
//                              sweep needs again (sin/cos of the same arguments — reusing them is
//                              bit-identical to re-evaluating, and removes 4 of 7 libm-class calls per stage)
//   dPhidx(g, x, p)            terminal costate
//   dHdx  (g, x, u, p, l, trig) costate equation
//   dHdu  (g, x, u, p, l, trig) optimality residual
//   ddHduu(m, x, u, p, l)      column-major Hessian for init_u0_newton
//   stage_coeffs / costate_step — the same backward stage regrouped for the "wg" mapping.  For every
//     Hamiltonian H = L + l^T f the costate right-hand side is affine in l:  dHdx = qx(x,p) + J(x,u)^T l,
//     and so is dHdu = phi(u) + B(x)^T l.  stage_coeffs(bw, phi, x,u,p,trig,dtau) evaluates everything that does
//     not involve l (NBW values per stage + the l-free part of dH/du) and can run for all stages at once;
//     costate_step(l, dF, bw, dtau) is what remains serial: l <- l + dtau*dHdx and dF = B^T l for the first NUL
//     components of dH/du.  Same mathematics as dHdx/dHdu, different association of the sums.
// NU_DYN = how many leading components of u the state equation reads (the forward sweep loads only those).
#pragma once
#include <hip/hip_runtime.h>

#include "dpp.hip.h"

namespace cgm {

// fp64 sin/cos for the horizon sweeps.  The device libm's sincos costs ~150 VALU instructions per call with
// its Payne-Hanek path and dominated the sweep; this one is ~35 per argument:
//   * two-constant Cody-Waite reduction by pi/2, exact for |a| < 1e5 (n*PIO2_HI has <= 50 significant bits);
//   * the classic minimax kernels on [-pi/4, pi/4] (coefficients: FreeBSD msun k_sin.c / k_cos.c, public
//     domain), cos assembled in the compensated form w + (((1-w)-hz) + z*r);
//   * Horner steps issued as 3-operand v_fma_f64 (hipcc otherwise turns fma(z,p,CONST) into v_mov + v_fmac),
//     and the chains of the two arguments / two kernels interleaved so four independent FMA chains are in
//     flight (one wave per SIMD cannot hide the dependent-issue latency otherwise).
// Accuracy (tests/test_gpu_parity.py::test_device_sincos_accuracy): <= 2 ulp + 1e-26*|a|.
// Arguments outside the fast range (or NaN) take the library path through one wave-uniform branch.
__device__ __forceinline__ double fma3(double a, double b, double c) {
  double d;
  asm("v_fma_f64 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(c));
  return d;
}

__device__ __forceinline__ double mul2(double a, double b) {  // a*b that hipcc cannot fuse into a following add
  double d;
  asm("v_mul_f64 %0, %1, %2" : "=v"(d) : "v"(a), "v"(b));
  return d;
}

__device__ __forceinline__ float fma3(float a, float b, float c) {
  float d;
  asm("v_fma_f32 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(c));
  return d;
}
__device__ __forceinline__ float mul2(float a, float b) {
  float d;
  asm("v_mul_f32 %0, %1, %2" : "=v"(d) : "v"(a), "v"(b));
  return d;
}
// type-generic spellings of the builtins the quad sweep uses
__device__ __forceinline__ double fma_t(double a, double b, double c) { return __builtin_fma(a, b, c); }
__device__ __forceinline__ float fma_t(float a, float b, float c) { return __builtin_fmaf(a, b, c); }
__device__ __forceinline__ double rint_t(double a) { return __builtin_rint(a); }
__device__ __forceinline__ float rint_t(float a) { return __builtin_rintf(a); }
__device__ __forceinline__ double maxabs_t(double m, double a) { return __builtin_fmax(m, __builtin_fabs(a)); }
__device__ __forceinline__ float maxabs_t(float m, float a) { return __builtin_fmaxf(m, __builtin_fabsf(a)); }
// ordered-as-integers view of a NON-NEGATIVE value (a running maximum without the canonicalising v_max of fmax)
__device__ __forceinline__ int nonneg_bits(double z) { return __double2hiint(z); }
__device__ __forceinline__ int nonneg_bits(float z) { return __float_as_int(z); }
__device__ __forceinline__ double xor_sign(double v, int flip) {
  return __hiloint2double(__double2hiint(v) ^ flip, __double2loint(v));
}
__device__ __forceinline__ float xor_sign(float v, int flip) { return __int_as_float(__float_as_int(v) ^ flip); }

// The 15 fp64 constants of the routine, held in VGPRs for the lifetime of a thread.  They are made opaque
// to the optimiser on purpose: with the scalar register file saturated by the kernel's pointers hipcc would
// otherwise re-materialise every constant in every stage (22 s_mov + 12 v_mov per sincos pair).
struct TrigConsts {
  double inv_pio2, pio2_hi, pio2_lo, S1, S2, S3, S4, S5, S6, C1, C2, C3, C4, C5, C6;
  // PIN = false (the 256-register lean kernels): the constants stay ordinary values the compiler may re-materialise
  // where a fresh evaluation needs them (once per chunk of stages there) instead of 30 registers live for a whole tick
  template <bool PIN = true>
  __device__ __forceinline__ void init() {
    inv_pio2 = 6.36619772367581382433e-01;
    pio2_hi = 1.57079632673412561417e+00;  // first 33 bits of pi/2
    pio2_lo = 6.07710050650619224932e-11;  // pi/2 - pio2_hi
    S1 = -1.66666666666666324348e-01, S2 = 8.33333333332248946124e-03, S3 = -1.98412698298579493134e-04;
    S4 = 2.75573137070700676789e-06, S5 = -2.50507602534068634195e-08, S6 = 1.58969099521155010221e-10;
    C1 = 4.16666666666666019037e-02, C2 = -1.38888888888741095749e-03, C3 = 2.48015872894767294178e-05;
    C4 = -2.75573143513906633035e-07, C5 = 2.08757232129817482790e-09, C6 = -1.13596475577881948265e-11;
    if constexpr (PIN) {
      asm volatile("" : "+v"(inv_pio2), "+v"(pio2_hi), "+v"(pio2_lo));
      asm volatile("" : "+v"(S1), "+v"(S2), "+v"(S3), "+v"(S4), "+v"(S5), "+v"(S6));
      asm volatile("" : "+v"(C1), "+v"(C2), "+v"(C3), "+v"(C4), "+v"(C5), "+v"(C6));
    }
  }
};
struct NoConsts {
  __device__ __forceinline__ void init() {}
};

// The library sin/cos (Payne-Hanek reduction, ~150 instructions and a few dozen live registers) is only ever reached
// through rarely-taken branches — arguments beyond the fast range.  Kept out of line: inlined into every sweep it
// raises the register pressure of blocks that never run, and the allocator then spills values that live across
// them (sweep constants, rows held in registers) everywhere, which the 256-register lean kernels pay for on the
// critical path.
// (The one-workgroup-per-CU kernels have 512 registers and keep it inline: out of line costs them 1.5 %, measured.)
__device__ __attribute__((noinline)) void sincos_library_outlined(double a, double* sn, double* cs) { ::sincos(a, sn, cs); }
template <bool OUTLINE>
__device__ __forceinline__ void sincos_library(double a, double* sn, double* cs) {
  if constexpr (OUTLINE)
    sincos_library_outlined(a, sn, cs);
  else
    ::sincos(a, sn, cs);
}

struct SinCosKernel {
  double r, z, ps, pc;
  int q;
  __device__ __forceinline__ void reduce(double a, const TrigConsts& K) {
    const double n = __builtin_rint(a * K.inv_pio2);
    r = __builtin_fma(-n, K.pio2_hi, a);
    r = __builtin_fma(-n, K.pio2_lo, r);
    q = static_cast<int>(n);
    z = r * r;
  }
  __device__ __forceinline__ void finish(double* sn, double* cs) const {
    const double ks = fma3(z * r, ps, r);
    const double hz = 0.5 * z;
    const double w = 1.0 - hz;
    const double kc = w + (((1.0 - w) - hz) + z * (z * pc));
    // quadrant: sin = {s, c, -s, -c}[q&3], cos = {c, -s, -c, s}[q&3]
    const bool odd = q & 1;
    const double ss = odd ? kc : ks;
    const double cc = odd ? ks : kc;
    const int sflip = (q & 2) << 30, cflip = ((q + 1) & 2) << 30;
    *sn = __hiloint2double(__double2hiint(ss) ^ sflip, __double2loint(ss));
    *cs = __hiloint2double(__double2hiint(cc) ^ cflip, __double2loint(cc));
  }
};

// sin/cos of two arguments at once (the pendulum needs sin/cos of x0-x1 and of x1 in every stage)
template <bool OUTLINE = false>
__device__ __forceinline__ void sincos2_f64(double a0, double a1, double* s0, double* c0, double* s1, double* c1,
                                            const TrigConsts& K) {
  const bool bad = !(__builtin_fabs(a0) < 1.0e5) || !(__builtin_fabs(a1) < 1.0e5);  // also catches NaN
  if (__builtin_expect(__any(bad), 0)) {
    sincos_library<OUTLINE>(a0, s0, c0);
    sincos_library<OUTLINE>(a1, s1, c1);
    return;
  }
  SinCosKernel k0, k1;
  k0.reduce(a0, K);
  k1.reduce(a1, K);
  k0.ps = fma3(k0.z, K.S6, K.S5), k1.ps = fma3(k1.z, K.S6, K.S5);
  k0.pc = fma3(k0.z, K.C6, K.C5), k1.pc = fma3(k1.z, K.C6, K.C5);
  k0.ps = fma3(k0.z, k0.ps, K.S4), k1.ps = fma3(k1.z, k1.ps, K.S4);
  k0.pc = fma3(k0.z, k0.pc, K.C4), k1.pc = fma3(k1.z, k1.pc, K.C4);
  k0.ps = fma3(k0.z, k0.ps, K.S3), k1.ps = fma3(k1.z, k1.ps, K.S3);
  k0.pc = fma3(k0.z, k0.pc, K.C3), k1.pc = fma3(k1.z, k1.pc, K.C3);
  k0.ps = fma3(k0.z, k0.ps, K.S2), k1.ps = fma3(k1.z, k1.ps, K.S2);
  k0.pc = fma3(k0.z, k0.pc, K.C2), k1.pc = fma3(k1.z, k1.pc, K.C2);
  k0.ps = fma3(k0.z, k0.ps, K.S1), k1.ps = fma3(k1.z, k1.ps, K.S1);
  k0.pc = fma3(k0.z, k0.pc, K.C1), k1.pc = fma3(k1.z, k1.pc, K.C1);
  k0.finish(s0, c0);
  k1.finish(s1, c1);
}

__device__ __forceinline__ void sincos_f64(double a, double* sn, double* cs) {
  TrigConsts K;
  K.init();
  double s1, c1;
  sincos2_f64(a, a, sn, cs, &s1, &c1, K);
}

// Per-scalar-type math context handed to the model functions.
template <class T, bool OUTLINE_LIB = false>  // OUTLINE_LIB: library sin/cos out of line (the 256-register lean kernels)
struct MathCtx;
// For the quad sweep both kernels are used in the common form  value = (1 + z*P(z)) * h,  P of NK coefficients:
//   sin: h = r, P = S1 + S2 z + ...            cos: h = 1, P = -1/2 + C1 z + C2 z^2 + ...
// kernel_coef(is_cos, i) is coefficient i of that P (zero-padded at the top for the shorter sin kernel);
// fast_range is the |argument| bound of the two-constant Cody-Waite reduction.
// Rotation step of the quad sweep (PendulumDev::quad_stage_rot): sin d = d (1 + z (RS[0] + z (RS[1] + ...))) and
// cos d - 1 = z (RC[0] + z (RC[1] + ...)), z = d^2, Taylor coefficients, truncation < 2e-17 relative for z <= rot_zmax.
template <bool OUTLINE_LIB>
struct MathCtx<double, OUTLINE_LIB> : TrigConsts {
  static constexpr bool OUTLINE = OUTLINE_LIB;
  static constexpr int NK = 7;
  static constexpr double fast_range = 1.0e5;
  static constexpr int NRS = 3, NRC = 4;
  static constexpr double rot_zmax = 1.6e-3;  // |d| <= 0.04: d^8/9! and d^10/10! below 2e-17
  __device__ __forceinline__ void init() { TrigConsts::template init<!OUTLINE_LIB>(); }
  __device__ __forceinline__ double rot_sin(int i) const {
    const double c[NRS] = {-1.0 / 6.0, 1.0 / 120.0, -1.0 / 5040.0};
    return c[i];
  }
  __device__ __forceinline__ double rot_cos(int i) const {
    const double c[NRC] = {-0.5, 1.0 / 24.0, -1.0 / 720.0, 1.0 / 40320.0};
    return c[i];
  }
  __device__ __forceinline__ void sincos_pair(double a0, double a1, double* s0, double* c0, double* s1,
                                              double* c1) const {
    sincos2_f64<OUTLINE_LIB>(a0, a1, s0, c0, s1, c1, *this);
  }
  __device__ __forceinline__ double kernel_coef(bool is_cos, int i) const {
    const double sk[NK] = {S1, S2, S3, S4, S5, S6, 0.0}, ck[NK] = {-0.5, C1, C2, C3, C4, C5, C6};
    return is_cos ? ck[i] : sk[i];
  }
};
// fp32: Cephes sinf/cosf minimax kernels on [-pi/4, pi/4] (public domain), FMA-based two-constant reduction
// (exact product n*PIO2_HI inside the fma), |a| < 1e4.  ~1 ulp; the fp32 parity bar is 1e-4 on u.
template <bool OUTLINE_LIB>
struct MathCtx<float, OUTLINE_LIB> {
  static constexpr bool OUTLINE = OUTLINE_LIB;
  static constexpr int NK = 4;
  static constexpr float fast_range = 1.0e4f;
  static constexpr int NRS = 2, NRC = 3;
  static constexpr float rot_zmax = 1.0e-2f;  // |d| <= 0.1: d^6/7! and d^8/8! below 3e-10
  __device__ __forceinline__ float rot_sin(int i) const {
    const float c[NRS] = {-1.0f / 6.0f, 1.0f / 120.0f};
    return c[i];
  }
  __device__ __forceinline__ float rot_cos(int i) const {
    const float c[NRC] = {-0.5f, 1.0f / 24.0f, -1.0f / 720.0f};
    return c[i];
  }
  float inv_pio2, pio2_hi, pio2_lo, S1, S2, S3, C1, C2, C3;
  __device__ __forceinline__ void init() {
    inv_pio2 = 6.36619772367581382433e-01f;
    pio2_hi = 1.57079637050628662109375f;   // float(pi/2)
    pio2_lo = -4.37113882867379262e-8f;     // pi/2 - pio2_hi
    S1 = -1.6666654611e-1f, S2 = 8.3321608736e-3f, S3 = -1.9515295891e-4f;
    C1 = 4.166664568298827e-2f, C2 = -1.388731625493765e-3f, C3 = 2.443315711809948e-5f;
    asm volatile("" : "+v"(inv_pio2), "+v"(pio2_hi), "+v"(pio2_lo));
    asm volatile("" : "+v"(S1), "+v"(S2), "+v"(S3), "+v"(C1), "+v"(C2), "+v"(C3));
  }
  __device__ __forceinline__ void sincos_pair(float a0, float a1, float* s0, float* c0, float* s1, float* c1) const {
    ::sincosf(a0, s0, c0);
    ::sincosf(a1, s1, c1);
  }
  __device__ __forceinline__ float kernel_coef(bool is_cos, int i) const {
    const float sk[NK] = {S1, S2, S3, 0.0f}, ck[NK] = {-0.5f, C1, C2, C3};
    return is_cos ? ck[i] : sk[i];
  }
};

struct ModelInfo {
  int dim_x, dim_u, dim_p, dv, k_max;
  double dt, h, zeta, Tf, alpha, tol;
};

// ------------------------------------------------------------------------------------------------
template <class T>
struct PendulumDev {
  static constexpr int NX = 4, NU = 3, NP = 2, NC = 3, NU_DYN = 1;
  static constexpr ModelInfo info() { return {4, 3, 2, 25, 5, 0.001, 0.002, 1000.0, 0.5, 0.5, 1e-6}; }  // model.hpp:8-35
  // model.hpp:80-98
  static constexpr T sf0 = T(3.0), sf1 = T(1.0), sf2 = T(0.0), sf3 = T(0.0);
  static constexpr T q0 = T(1.0), q1 = T(1.0), q2 = T(0.0), q3 = T(0.0);
  static constexpr T r0 = T(1.0), r1 = T(0.1);
  static constexpr T uc = T(0.0), ur = T(3.0);  // (umax+umin)/2, (umax-umin)/2 with umin=-3, umax=3
  static constexpr T As = T(6.25), Bs = T(15.6), A52 = T(39.1111), C22 = T(0.0407448);
  static constexpr T A32a = T(5.65635), A32 = T(0.905016), A32b = T(14.1183);

  // trig = { sin(x0-x1), cos(x0-x1), cos(x1) }
  using Math = MathCtx<T>;
  template <bool OUTLINE_LIB>
  using MathFor = MathCtx<T, OUTLINE_LIB>;
  template <class MC>
  static __device__ __forceinline__ void dxdt(T* f, const T* x, const T* u, T* trig, const MC& mc) {  // model.hpp:37-42
    T sd, cd, s1, c1;
    mc.sincos_pair(x[0] - x[1], x[1], &sd, &cd, &s1, &c1);
    trig[0] = sd;
    trig[1] = cd;
    trig[2] = c1;
    f[0] = x[2];
    f[1] = x[3];
    f[2] = -As * x[2] + Bs * u[0];
    f[3] = A32 * x[2] * x[2] * sd + A52 * s1 - A32b * cd * u[0] + A32a * cd * x[2] + C22 * (x[2] - x[3]);
  }
  static __device__ __forceinline__ void dPhidx(T* g, const T* x, const T* p) {  // model.hpp:44-49
    g[0] = (x[0] - p[0]) * sf0;
    g[1] = (x[1] - p[1]) * sf1;
    g[2] = x[2] * sf2;
    g[3] = x[3] * sf3;
  }
  static __device__ __forceinline__ void dHdx(T* g, const T* x, const T* u, const T* p, const T* l,
                                              const T* trig) {  // model.hpp:51-56
    const T sd = trig[0], cd = trig[1], c1 = trig[2];
    g[0] = (x[0] - p[0]) * q0 + l[3] * (A32 * x[2] * x[2] * cd + A32b * sd * u[0] - A32a * sd * x[2]);
    g[1] = (x[1] - p[1]) * q1 +
           l[3] * (-A32 * x[2] * x[2] * cd + A52 * c1 - A32b * sd * u[0] + A32a * sd * x[2]);
    g[2] = x[2] * q2 + l[0] - l[2] * As + l[3] * (T(2.0) * A32 * x[2] * sd + A32a * cd + C22);
    g[3] = x[3] * q3 + l[1] - l[3] * C22;
  }
  static __device__ __forceinline__ void dHdu(T* g, const T*, const T* u, const T*, const T* l,
                                              const T* trig) {  // model.hpp:58-62
    g[0] = (r0 * u[0]) + l[2] * Bs - l[3] * A32b * trig[1] + (u[2] * (T(2.0) * u[0] - T(2.0) * uc));
    g[1] = T(-0.5) * r1 + (T(2.0) * u[2] * u[1]);
    g[2] = (u[0] - uc) * (u[0] - uc) + u[1] * u[1] - ur * ur;
  }
  // --- affine-in-costate split of the backward stage (see the interface note at the top of the file) ---
  // dHdx = qx(x,p) + J(x,u)^T l and dHdu = phi(u) + B(x)^T l (model.hpp:51-62 regrouped); q2 = q3 = 0 here,
  // so the l-free parts of rows 2,3 vanish and are not stored.
  // Coefficient order: the NBW_LIN entries that multiply l first, the l-free (bias) entries last — the recurrence is
  // affine in l, so a chunk of stages can also be run from a unit vector WITHOUT the bias (costate_step<true>: one
  // column of the chunk's transfer matrix; WgCtx::sweep_costate_par) and such a lane fetches the first NBW_LIN only.
  static constexpr int NBW = 6, NUL = 1, NBW_LIN = 4;
  static constexpr bool COSTATE_HOM = true;
  static_assert(q2 == T(0.0) && q3 == T(0.0), "pendulum stage coefficients assume q2 = q3 = 0");
  static __device__ __forceinline__ void stage_coeffs(T* bw, T* phi, const T* x, const T* u, const T* p,
                                                      const T* trig, T dtau) {
    const T sd = trig[0], cd = trig[1], c1 = trig[2];
    const T S0 = A32 * x[2] * x[2] * cd + A32b * sd * u[0] - A32a * sd * x[2];                  // l3 coeff, row 0
    const T S1 = -A32 * x[2] * x[2] * cd + A52 * c1 - A32b * sd * u[0] + A32a * sd * x[2];      // l3 coeff, row 1
    const T S2 = T(2.0) * A32 * x[2] * sd + A32a * cd + C22;                                    // l3 coeff, row 2
    bw[0] = dtau * S0;
    bw[1] = dtau * S1;
    bw[2] = dtau * S2;
    bw[3] = -A32b * cd;  // l3 coeff of dHdu[0]
    bw[4] = dtau * ((x[0] - p[0]) * q0);
    bw[5] = dtau * ((x[1] - p[1]) * q1);
    phi[0] = (r0 * u[0]) + (u[2] * (T(2.0) * u[0] - T(2.0) * uc));
    phi[1] = T(-0.5) * r1 + (T(2.0) * u[2] * u[1]);
    phi[2] = (u[0] - uc) * (u[0] - uc) + u[1] * u[1] - ur * ur;
  }
  // l <- l + dtau*dHdx(l) and dF = B^T l_old, from the stored coefficients
  template <bool HOM = false>
  static __device__ __forceinline__ void costate_step(T* l, T* dF, const T* bw, T dtau) {
    dF[0] = l[2] * Bs + bw[3] * l[3];
    const T n0 = HOM ? fma_t(bw[0], l[3], l[0]) : (l[0] + bw[4]) + bw[0] * l[3];
    const T n1 = HOM ? fma_t(bw[1], l[3], l[1]) : (l[1] + bw[5]) + bw[1] * l[3];
    // (1 - dtau*As) and (1 - dtau*C22) are loop invariants of the sweep: 5 instructions for the two rows instead of 7
    // (fma_t, not __builtin_fma: the latter is the DOUBLE builtin — in the fp32 kernels it cost 8 conversion / fp64
    // instructions per stage of the costate loop, a quarter of it)
    const T n2 = fma_t(bw[2], l[3], fma_t(dtau, l[0], (T(1.0) - dtau * As) * l[2]));
    const T n3 = fma_t(dtau, l[1], (T(1.0) - dtau * C22) * l[3]);
    l[0] = n0, l[1] = n1, l[2] = n2, l[3] = n3;
  }
  // --- state sweep over a DPP quad -----------------------------------------------------------------
  // The serial state sweep (2 sincos + ~17 flops per stage, model.hpp:37-42) is the longest phase of a tick, and one
  // wave issues ONE instruction of any kind per ~4.4 cycles, dependent or not (tools/ubench_issue.hip; a taken
  // branch costs 28 more) — so what counts is the number of instructions per stage, not their latency.
  // Here the four lanes of a quad carry ONE instance and share a single instruction stream:
  //   * lane rho evaluates one of the four kernels {sin d, cos d, sin x1, cos x1} (d = x0 - x1) with per-lane
  //     coefficients in the common form  G = 1 + z*P(z),  value = G*h  (sin: P = S1..S6, h = r;  cos: P = -1/2 +
  //     z*(C1..C6), h = 1), no select inside the kernel; the sin/cos pair of an angle is exchanged with
  //     quad_perm[1,0,3,2] for the quadrant fix-up;
  //   * the d-lanes keep -x1 instead of x1, so the angle of every lane is ONE fma: arg = kap*x0 + x1c;
  //   * the trig-dependent part of dxdt[3],  A32 x2^2 sin d + (A32a x2 - A32b u0) cos d + A52 sin x1  (model.hpp:41,
  //     regrouped), is one multiply per lane + a quad sum, bit-identical in the four lanes, so x stays replicated
  //     without any broadcast.
  // Stage-table writes (all lanes, no branch), NSLOT = 6 slots per (stage, instance): x0, x1, x2 -> slots 0, 1, 2,
  // sin d, cos d, cos x1 -> slots 3, 4, 5.  x3 is not stored: stage_coeffs does not use x[3] because q3 = 0.  Per stage
  // ONE 8-byte store carries four useful values: the d-lanes their sin d / cos d, the cos-x1 lane its value, and the
  // sin-x1 lane — whose own value no later phase reads — x1 instead (WgCtx::sweep_state: one select instead of a
  // second LDS store, which costs the issuing wave 14.6 cycles).  x0 and x2 (the same value in every lane) are
  // written by all lanes, every other stage (see x02_step).  (TAB_PAD: spare stages behind the table — none needed since
  // every store lands in its own stage.)
  static constexpr bool HAS_QUAD_SWEEP = true;
  static constexpr int NSLOT = 6, TRIG_SLOT0 = 3, TAB_PAD = 0;
  // x0 and x2 obey a recurrence of their own — x0' = x0 + dtau x2, x2' = x2 + dtau (-As x2 + Bs u0) (model.hpp:38,40) —
  // that needs no trig value.  In the pipelined sweeps the sweep wave therefore writes them to the stage table at every
  // OTHER stage only (an LDS store occupies the CU's LDS pipe for its 26 cycles whichever wave issues it), and the
  // coefficient phase, which handles the two stages of a pair in one thread, re-derives the odd stage's values with
  // the very same operations (bit-identical).
  static __device__ __forceinline__ void x02_step(T& x0, T& x2, T u0, T dtau) {
    const T f2 = fma_t(-As, x2, Bs * u0);
    x0 = fma_t(dtau, x2, x0);
    x2 = fma_t(dtau, f2, x2);
  }
  static_assert(NBW <= NSLOT, "the coefficients of a stage overwrite its state/trig slots in place");
  static constexpr int QSLOT_XA = 0, QSLOT_XB = 2, QLANE_TRUE_X = 2;  // write2 slots; a lane whose x[1] is +x1
  struct QuadLane {
    static constexpr int NK = Math::NK;
    T k[NK];                       // P(z) = k[0] + z*(k[1] + ... + z*k[NK-1])
    T hs, hc;                      // h = hs*r + hc
    T mp, mq, mr, ms;              // weight (mp*x2 + mq)*x2 + (mr*u0 + ms) of this lane's trig value in dxdt[3]
    T kap, sg;                     // arg = kap*x0 + x1c, x1c = sg*x1
    static constexpr int NRS = Math::NRS, NRC = Math::NRC;
    T rs[NRS], rs1, rc[NRC];       // rotation step: sgn*sin d = d (rs1 + z (rs[0] + ...)), cos d - 1 = z (rc[0] + ...)
    bool is_cos;
    int slot_v;
    template <class MC>
    __device__ __forceinline__ void init(int rho, const MC& mc) {
      is_cos = rho & 1;
#pragma unroll
      for (int i = 0; i < NK; ++i) k[i] = mc.kernel_coef(is_cos, i);
      hs = is_cos ? T(0) : T(1), hc = is_cos ? T(1) : T(0);
      mp = rho == 0 ? A32 : T(0), mq = rho == 1 ? A32a : T(0), mr = rho == 1 ? -A32b : T(0), ms = rho == 2 ? A52 : T(0);
      kap = rho < 2 ? T(1) : T(0), sg = rho < 2 ? T(-1) : T(1);
      // sin(a+d) = sin a cos d + cos a sin d on a sin lane, cos(a+d) = cos a cos d - sin a sin d on a cos lane
      rs1 = is_cos ? T(-1) : T(1);
#pragma unroll
      for (int i = 0; i < NRS; ++i) rs[i] = rs1 * mc.rot_sin(i);
#pragma unroll
      for (int i = 0; i < NRC; ++i) rc[i] = mc.rot_cos(i);
#pragma unroll
      for (int i = 0; i < NRC; ++i)
        if constexpr (!MC::OUTLINE) asm volatile("" : "+v"(rc[i]));  // kept in registers like the kernel constants
      slot_v = rho == 0 ? 3 : (rho == 1 ? 4 : (rho == 2 ? 1 : 5));  // (the sin-x1 lane stores x1 in slot 1 instead)
    }
  };
  // this lane's trig value of `arg`; *amax accumulates max|arg| (arguments outside the fast range make the caller
  // redo the sweep with SLOW = true, the library sin/cos)
  template <bool SLOW, class MC>
  static __device__ __forceinline__ T quad_trig(T arg, const QuadLane& Q, const MC& mc, T* amax) {
    if constexpr (SLOW) {
      double sn, cs;
      sincos_library<MC::OUTLINE>(double(arg), &sn, &cs);
      return Q.is_cos ? T(cs) : T(sn);
    } else {
      constexpr int NK = QuadLane::NK;
      *amax = maxabs_t(*amax, arg);
      const T n = rint_t(arg * mc.inv_pio2);
      T r = fma_t(-n, mc.pio2_hi, arg);
      r = fma_t(-n, mc.pio2_lo, r);
      const int q = static_cast<int>(n);
      const T z = r * r;
      T P = fma3(z, Q.k[NK - 1], Q.k[NK - 2]);
#pragma unroll
      for (int i = NK - 3; i >= 0; --i) P = fma3(z, P, Q.k[i]);
      const T G = fma_t(z, P, T(1.0));
      const T h = fma3(Q.hs, r, Q.hc);
      const T mine = mul2(G, h);
      const T other = dpp_move<DPP_QUAD_SWAP1>(mine);  // the cos kernel of my angle if I am the sin lane, and v.v.
      const T pick = (q & 1) ? other : mine;            // sin = {s,c,-s,-c}[q&3], cos = {c,-s,-c,s}[q&3]
      const int flip = ((q + (Q.is_cos ? 1 : 0)) & 2) << 30;
      return xor_sign(pick, flip);
    }
  }
  static __device__ __forceinline__ bool quad_arg_bad(T amax) { return !(amax < T(Math::fast_range)); }
  // lane-local form of the state (x[1] <- sg*x1) and the first trig value
  static __device__ __forceinline__ T quad_arg(const T* x, const QuadLane& Q) { return fma_t(Q.kap, x[0], x[1]); }
  template <bool SLOW, class MC>
  static __device__ __forceinline__ T quad_begin(T* x, const QuadLane& Q, const MC& mc, T* amax) {
    x[1] = Q.sg * x[1];
    return quad_trig<SLOW>(quad_arg(x, Q), Q, mc, amax);
  }
  // One stage: x(s), v = trig(x(s)) -> x(s+1), v = trig(x(s+1)).  dtau1 = sg*dtau.
  template <bool SLOW, class MC>
  static __device__ __forceinline__ void quad_stage(T* x, T& v, T u0, T dtau, T dtau1, const QuadLane& Q,
                                                    const MC& mc, T* amax) {
    const T m = fma_t(fma_t(Q.mp, x[2], Q.mq), x[2], fma_t(Q.mr, u0, Q.ms));
    const T trig_sum = quad_sum(mul2(m, v));  // mul2: the same rounded product in every lane of the quad
    const T f3 = fma_t(C22, x[2] - x[3], trig_sum);
    const T f2 = fma_t(-As, x[2], Bs * u0);
    x[0] = fma_t(dtau, x[2], x[0]);
    x[1] = fma_t(dtau1, x[3], x[1]);
    x[2] = fma_t(dtau, f2, x[2]);
    x[3] = fma_t(dtau, f3, x[3]);
    v = quad_trig<SLOW>(quad_arg(x, Q), Q, mc, amax);
  }
  // The same stage with the trig value obtained by ROTATING the previous stage's value through the angle increment
  // d = arg(s+1) - arg(s) (exact: both are within a factor two of each other) instead of a fresh evaluation:
  //   v' = v + (v (cos d - 1) + v_partner (+-sin d)),   v_partner = the other kernel of this lane's angle (quad swap)
  // 17 instructions instead of 27 (no argument reduction, no quadrant logic, two short Taylor polynomials).  Each step
  // adds ~2 ulp of rounding; the sweep restarts from a fresh evaluation at every chunk of stages (~25, WgCtx::chunk_len),
  // so the accumulated error stays below ~50 ulp worst case, ~8 ulp typical — 4 orders of magnitude inside the parity
  // tolerance on u.  *zmax accumulates max d^2 (integer view): a chunk whose increments leave |d| <= sqrt(rot_zmax)
  // is redone with fresh evaluations per stage (quad_stage<false>).
  template <class MC>
  static __device__ __forceinline__ void quad_stage_rot(T* x, T& v, T& argp, T u0, T dtau, T dtau1, const QuadLane& Q,
                                                        const MC& mc, int* zmax) {
    constexpr int NRS = QuadLane::NRS, NRC = QuadLane::NRC;
    static_assert(NRC == NRS + 1, "the two Taylor polynomials are advanced in lock-step");
    // Two independent dependency chains per stage — A: weight, product, quad sum, f3 -> x3';  B: new angle, increment,
    // the two Taylor polynomials, partner exchange -> v' — and on this hardware a fp64 operation that consumes the result
    // of the instruction issued right before it costs an extra issue slot (the compiler pads with s_nop: 4 per stage
    // when it schedules the chains one after the other, as it did for every second stage of the unrolled loop).
    // The groups below alternate A and B; a sched_barrier between groups keeps the scheduler from re-serialising them.
#define CGM_SB() __builtin_amdgcn_sched_barrier(0)
    // (a barrier after EVERY statement: the order below is the schedule.  A DPP move must not follow the instruction
    // that produces its source within two issue slots, a fp64 operation not its producer within one.)
    const T ma = fma_t(Q.mp, x[2], Q.mq);           CGM_SB();  // A
    const T x1n = fma_t(dtau1, x[3], x[1]);         CGM_SB();  // B  (old x3)
    const T mb = fma_t(Q.mr, u0, Q.ms);             CGM_SB();  // A
    const T x0n = fma_t(dtau, x[2], x[0]);          CGM_SB();  // B  (old x2)
    const T m = fma_t(ma, x[2], mb);                CGM_SB();  // A
    const T arg = fma_t(Q.kap, x0n, x1n);           CGM_SB();  // B
    T t = mul2(m, v);                               CGM_SB();  // A  (mul2: the same rounded product in every lane of the quad)
    const T d = arg - argp;                         CGM_SB();  // B
    const T f2a = Bs * u0;                          CGM_SB();  // C
    const T z = mul2(d, d);                         CGM_SB();  // B
    T tp = dpp_move<DPP_QUAD_SWAP1>(t);             CGM_SB();  // A  (two slots after t)
    T ps = fma3(z, Q.rs[NRS - 1], Q.rs[NRS - 2]);   CGM_SB();  // B
    T pc = fma3(z, Q.rc[NRC - 1], Q.rc[NRC - 2]);   CGM_SB();  // B
    t = t + tp;                                     CGM_SB();  // A
#pragma unroll
    for (int i = NRS - 3; i >= 0; --i) {            // B (fp64: one round)
      ps = fma3(z, ps, Q.rs[i]);                    CGM_SB();
      pc = fma3(z, pc, Q.rc[i + 1]);                CGM_SB();
    }
    const T f2 = fma_t(-As, x[2], f2a);             CGM_SB();  // C
    tp = dpp_move<DPP_QUAD_SWAP2>(t);               CGM_SB();  // A
    ps = fma3(z, ps, Q.rs1);                        CGM_SB();  // B
    pc = fma3(z, pc, Q.rc[0]);                      CGM_SB();  // B
    const T df = x[2] - x[3];                       CGM_SB();  // A
    const T trig_sum = t + tp;                      CGM_SB();  // A  (bit-identical in the four lanes)
    const T sd = mul2(d, ps);                       CGM_SB();  // B  +-sin d
    const T pv = dpp_move<DPP_QUAD_SWAP1>(v);       CGM_SB();  // B
    const T f3 = fma_t(C22, df, trig_sum);          CGM_SB();  // A
    const T cm1 = mul2(z, pc);                      CGM_SB();  // B  cos d - 1
    T w = mul2(pv, sd);                             CGM_SB();  // B
    x[3] = fma_t(dtau, f3, x[3]);                   CGM_SB();  // A
    x[2] = fma_t(dtau, f2, x[2]);                   CGM_SB();  // C
    w = fma_t(v, cm1, w);                           CGM_SB();  // B
    const int zb = nonneg_bits(z);
    *zmax = zb > *zmax ? zb : *zmax;                CGM_SB();
    x[0] = x0n, x[1] = x1n, argp = arg;
    v = v + w;                                      CGM_SB();  // B
#undef CGM_SB
  }
  static __device__ __forceinline__ bool quad_rot_bad(int zmax) { return zmax > nonneg_bits(T(Math::rot_zmax)); }
  static __device__ __forceinline__ void ddHduu(T* m, const T*, const T* u, const T*, const T*) {  // :64-76
    m[0] = r0 + 2 * u[2];
    m[1] = 0;
    m[2] = 2 * u[0] - 2 * uc;
    m[3] = 0;
    m[4] = 2 * u[2];
    m[5] = 2 * u[1];
    m[6] = 2 * u[0] - 2 * uc;
    m[7] = 2 * u[1];
    m[8] = 0;
  }
};

// ------------------------------------------------------------------------------------------------
template <class T>
struct MsdDev {
  static constexpr int NX = 4, NU = 6, NP = 2, NC = 0, NU_DYN = 2;
  static constexpr ModelInfo info() { return {4, 6, 2, 50, 5, 0.001, 0.002, 1000.0, 1.0, 0.5, 1e-6}; }  // model.hpp:7-34
  static constexpr T sf0 = T(10.0), sf1 = T(10.0), sf2 = T(1.0), sf3 = T(1.0);  // model.hpp:112-114
  static constexpr T q0 = T(1.0), q1 = T(1.0), q2 = T(10.0), q3 = T(10.0);
  static constexpr T r0 = T(0.1), r1 = T(0.1), r2 = T(0.01), r3 = T(0.01);
  static constexpr T uc = T(0.0), ur = T(10.0);  // umin=-10, umax=10 (model.hpp:117-120)
  static constexpr T m1 = T(1.0), m2 = T(1.0), d1 = T(1.0), d2 = T(1.0), k1 = T(1.0), k2 = T(1.0);

  using Math = NoConsts;
  template <bool>
  using MathFor = NoConsts;
  static __device__ __forceinline__ void dxdt(T* f, const T* x, const T* u, T*, const Math&) {  // model.hpp:36-41
    f[0] = x[2];
    f[1] = x[3];
    // dxdt uses (k1*k2), dHdx below uses (k1+k2): the reference's own inconsistency, kept
    f[2] = -(k1 * k2) / m1 * x[0] + k2 / m1 * x[1] - (d1 + d2) / m1 * x[2] + d2 / m1 * x[3] + u[0] / m1;
    f[3] = k2 / m2 * x[0] - k2 / m2 * x[1] + d2 / m2 * x[2] - d2 / m2 * x[3] + u[1] / m2;
  }
  static __device__ __forceinline__ void dPhidx(T* g, const T* x, const T* p) {  // model.hpp:43-48
    g[0] = -(p[0] - x[0]) * sf0;
    g[1] = -(p[1] - x[1]) * sf1;
    g[2] = x[2] * sf2;
    g[3] = x[3] * sf3;
  }
  static __device__ __forceinline__ void dHdx(T* g, const T* x, const T*, const T* p, const T* l,
                                              const T*) {  // model.hpp:50-55
    g[0] = -(p[0] - x[0]) * q0 - (k1 + k2) / m1 * l[2] + k2 / m2 * l[3];
    g[1] = -(p[1] - x[1]) * q1 + k2 / m1 * l[2] - k2 / m2 * l[3];
    g[2] = x[2] * q2 + l[0] - (d1 + d2) / m1 * l[2] + d2 / m2 * l[3];
    g[3] = x[3] * q3 + l[1] + d2 / m1 * l[2] - d2 / m2 * l[3];
  }
  static __device__ __forceinline__ void dHdu(T* g, const T*, const T* u, const T*, const T* l,
                                              const T*) {  // model.hpp:57-64
    g[0] = r0 * u[0] + l[2] / m1 + T(2.0) * u[4] * (u[0] - uc);
    g[1] = r1 * u[1] + l[3] / m2 + T(2.0) * u[5] * (u[1] - uc);
    g[2] = -r2 + T(2.0) * u[4] * u[2];
    g[3] = -r3 + T(2.0) * u[5] * u[3];
    g[4] = (u[0] - uc) * (u[0] - uc) + u[2] * u[2] - ur * ur;
    g[5] = (u[1] - uc) * (u[1] - uc) + u[3] * u[3] - ur * ur;
  }
  static constexpr bool HAS_QUAD_SWEEP = false;  // no transcendental in the state equation: nothing to spread
  static constexpr int NSLOT = NX + NC, TRIG_SLOT0 = NX, TAB_PAD = 0;
  // affine-in-costate split (model.hpp:50-64 regrouped): the Jacobian is constant, only qx depends on the stage
  static constexpr int NBW = 4, NUL = 2, NBW_LIN = 0;  // (all four are bias entries, see PendulumDev::NBW_LIN)
  static constexpr bool COSTATE_HOM = true;
  static __device__ __forceinline__ void stage_coeffs(T* bw, T* phi, const T* x, const T* u, const T* p, const T*,
                                                      T dtau) {
    bw[0] = dtau * (-(p[0] - x[0]) * q0);
    bw[1] = dtau * (-(p[1] - x[1]) * q1);
    bw[2] = dtau * (x[2] * q2);
    bw[3] = dtau * (x[3] * q3);
    phi[0] = r0 * u[0] + T(2.0) * u[4] * (u[0] - uc);
    phi[1] = r1 * u[1] + T(2.0) * u[5] * (u[1] - uc);
    phi[2] = -r2 + T(2.0) * u[4] * u[2];
    phi[3] = -r3 + T(2.0) * u[5] * u[3];
    phi[4] = (u[0] - uc) * (u[0] - uc) + u[2] * u[2] - ur * ur;
    phi[5] = (u[1] - uc) * (u[1] - uc) + u[3] * u[3] - ur * ur;
  }
  template <bool HOM = false>
  static __device__ __forceinline__ void costate_step(T* l, T* dF, const T* bw, T dtau) {
    dF[0] = l[2] / m1;
    dF[1] = l[3] / m2;
    const T n0 = (HOM ? l[0] : l[0] + bw[0]) + dtau * (-(k1 + k2) / m1 * l[2] + k2 / m2 * l[3]);
    const T n1 = (HOM ? l[1] : l[1] + bw[1]) + dtau * (k2 / m1 * l[2] - k2 / m2 * l[3]);
    const T n2 = (HOM ? l[2] : l[2] + bw[2]) + dtau * (l[0] - (d1 + d2) / m1 * l[2] + d2 / m2 * l[3]);
    const T n3 = (HOM ? l[3] : l[3] + bw[3]) + dtau * (l[1] + d2 / m1 * l[2] - d2 / m2 * l[3]);
    l[0] = n0, l[1] = n1, l[2] = n2, l[3] = n3;
  }
  static __device__ __forceinline__ void ddHduu(T* m, const T*, const T* u, const T*, const T*) {  // :66-108
#pragma unroll
    for (int i = 0; i < 36; ++i) m[i] = 0;
    m[0] = r0 + 2 * u[4];
    m[4] = 2 * (u[0] - uc);
    m[7] = r1 + 2 * u[5];
    m[11] = 2 * (u[1] - uc);
    m[14] = 2 * u[4];
    m[16] = 2 * u[2];
    m[21] = 2 * u[5];
    m[23] = 2 * u[3];
    m[24] = 2 * (u[0] - uc);
    m[26] = 2 * u[2];
    m[31] = 2 * (u[1] - uc);
    m[33] = 2 * u[3];
  }
};

// ------------------------------------------------------------------------------------------------
template <class T>
struct SemiactiveDev {
  static constexpr int NX = 2, NU = 3, NP = 0, NC = 0, NU_DYN = 1;
  static constexpr ModelInfo info() { return {2, 3, 0, 50, 5, 0.001, 0.002, 1000.0, 1.0, 0.5, 1e-6}; }  // model.hpp:7-34
  static constexpr T sf0 = T(1.0), sf1 = T(10.0), q0 = T(1.0), q1 = T(10.0);  // model.hpp:73-76
  static constexpr T r0 = T(1.0), r1 = T(0.01);
  static constexpr T uc = T(0.5), ur = T(0.5);  // umin=0, umax=1 (model.hpp:79-82)
  static constexpr T a = T(-1.0), b = T(-1.0);  // model.hpp:85-86

  using Math = NoConsts;
  template <bool>
  using MathFor = NoConsts;
  static __device__ __forceinline__ void dxdt(T* f, const T* x, const T* u, T*, const Math&) {  // model.hpp:36-39
    f[0] = x[1];
    f[1] = a * x[0] + b * u[0] * x[1];
  }
  static __device__ __forceinline__ void dPhidx(T* g, const T* x, const T*) {  // model.hpp:41-44
    g[0] = x[0] * sf0;
    g[1] = x[1] * sf1;
  }
  static __device__ __forceinline__ void dHdx(T* g, const T* x, const T* u, const T*, const T* l,
                                              const T*) {  // model.hpp:46-49
    g[0] = x[0] * q0 + a * l[1];
    g[1] = x[1] * q1 + l[0] + b * u[0] * l[1];
  }
  static __device__ __forceinline__ void dHdu(T* g, const T* x, const T* u, const T*, const T* l,
                                              const T*) {  // model.hpp:51-55
    g[0] = r0 * u[0] + b * x[1] * l[1] + 2 * u[2] * (u[0] - uc);
    g[1] = -r1 + 2 * u[1] * u[2];
    g[2] = (u[0] - uc) * (u[0] - uc) + u[1] * u[1] - ur * ur;
  }
  static constexpr bool HAS_QUAD_SWEEP = false;
  static constexpr bool ROW_AFFINE = true;  // dxdt is affine in x for given u: WgCtx::row_affine_sweep (tick_wg.hip.h, NWT = 2)
  static constexpr int NSLOT = NX + NC, TRIG_SLOT0 = NX, TAB_PAD = 0;
  // affine-in-costate split (model.hpp:46-55 regrouped)
  static constexpr int NBW = 4, NUL = 1, NBW_LIN = 2;  // (see PendulumDev::NBW_LIN)
  static constexpr bool COSTATE_HOM = true;
  static __device__ __forceinline__ void stage_coeffs(T* bw, T* phi, const T* x, const T* u, const T*, const T*,
                                                      T dtau) {
    bw[0] = dtau * (b * u[0]);  // l1 coeff, row 1
    bw[1] = b * x[1];           // l1 coeff of dHdu[0]
    bw[2] = dtau * (x[0] * q0);
    bw[3] = dtau * (x[1] * q1);
    phi[0] = r0 * u[0] + 2 * u[2] * (u[0] - uc);
    phi[1] = -r1 + 2 * u[1] * u[2];
    phi[2] = (u[0] - uc) * (u[0] - uc) + u[1] * u[1] - ur * ur;
  }
  template <bool HOM = false>
  static __device__ __forceinline__ void costate_step(T* l, T* dF, const T* bw, T dtau) {
    dF[0] = bw[1] * l[1];
    const T n0 = (HOM ? l[0] : l[0] + bw[2]) + dtau * (a * l[1]);
    const T n1 = (HOM ? l[1] : l[1] + bw[3]) + (dtau * l[0] + bw[0] * l[1]);
    l[0] = n0, l[1] = n1;
  }
  static __device__ __forceinline__ void ddHduu(T* m, const T*, const T* u, const T*, const T*) {  // :57-69
    m[0] = r0 + 2 * u[2];
    m[1] = 0;
    m[2] = 2 * (u[0] - uc);
    m[3] = 0;
    m[4] = 2 * u[2];
    m[5] = 2 * u[1];
    m[6] = 2 * (u[0] - uc);
    m[7] = 2 * u[1];
    m[8] = 0;
  }
};

}  // namespace cgm
