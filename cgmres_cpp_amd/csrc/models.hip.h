// Device-side problem definitions of the three models the reference ships.
// Each struct is the gfx950 counterpart of one reference `class Model`:
//   PendulumDev   — arm_type_inverted_pendulum/model.hpp:37-76 (≡ multiple_controller/model2.hpp)
//   MsdDev        — mass_spring_damper/model.hpp:36-108        (≡ multiple_controller/model1.hpp)
//   SemiactiveDev — semiactive_damper/model.hpp:36-69
// Interface (all static, all inlined into the sweep kernels, scalars in registers):
//   dxdt  (f, x, u, trig)      state equation; also fills `trig`, the per-stage values the costate
//                              sweep needs again (sin/cos of the same arguments — reusing them is
//                              bit-identical to re-evaluating, and removes 4 of 7 libm-class calls per stage)
//   dPhidx(g, x, p)            terminal costate
//   dHdx  (g, x, u, p, l, trig) costate equation
//   dHdu  (g, x, u, p, l, trig) optimality residual
//   ddHduu(m, x, u, p, l)      column-major Hessian for init_u0_newton
// NU_DYN = how many leading components of u the state equation reads (the forward sweep loads only those).
#pragma once
#include <hip/hip_runtime.h>

namespace cgm {

// sin and cos of one fp64 argument in ~35 VALU instructions (the device libm's sincos costs ~150 with its
// Payne-Hanek path and dominated the horizon sweep).  Two-constant Cody-Waite reduction by pi/2 — exact for
// |a| < 1e5 because n*PIO2_HI has <= 50 significant bits — followed by the classic minimax kernels on
// [-pi/4, pi/4] (coefficients: FreeBSD msun k_sin.c / k_cos.c, public domain; cos assembled in the
// compensated form w + (((1-w)-hz) + z*r)).  Measured against the host libm in tests: <= 1.5 ulp.
// Larger arguments take the library path.
__device__ __forceinline__ void sincos_f64(double a, double* sn, double* cs) {
  if (__builtin_expect(!(__builtin_fabs(a) < 1.0e5), 0)) {  // also catches NaN
    ::sincos(a, sn, cs);
    return;
  }
  constexpr double INV_PIO2 = 6.36619772367581382433e-01;
  constexpr double PIO2_HI = 1.57079632673412561417e+00;  // first 33 bits of pi/2
  constexpr double PIO2_LO = 6.07710050650619224932e-11;  // pi/2 - PIO2_HI
  const double n = __builtin_rint(a * INV_PIO2);
  double r = __builtin_fma(-n, PIO2_HI, a);
  r = __builtin_fma(-n, PIO2_LO, r);
  const int q = static_cast<int>(n);
  const double z = r * r;
  // sin kernel
  constexpr double S1 = -1.66666666666666324348e-01, S2 = 8.33333333332248946124e-03,
                   S3 = -1.98412698298579493134e-04, S4 = 2.75573137070700676789e-06,
                   S5 = -2.50507602534068634195e-08, S6 = 1.58969099521155010221e-10;
  double ps = __builtin_fma(z, S6, S5);
  ps = __builtin_fma(z, ps, S4);
  ps = __builtin_fma(z, ps, S3);
  ps = __builtin_fma(z, ps, S2);
  ps = __builtin_fma(z, ps, S1);
  const double ks = __builtin_fma(z * r, ps, r);
  // cos kernel
  constexpr double C1 = 4.16666666666666019037e-02, C2 = -1.38888888888741095749e-03,
                   C3 = 2.48015872894767294178e-05, C4 = -2.75573143513906633035e-07,
                   C5 = 2.08757232129817482790e-09, C6 = -1.13596475577881948265e-11;
  double pc = __builtin_fma(z, C6, C5);
  pc = __builtin_fma(z, pc, C4);
  pc = __builtin_fma(z, pc, C3);
  pc = __builtin_fma(z, pc, C2);
  pc = __builtin_fma(z, pc, C1);
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

template <class T>
__device__ __forceinline__ void sincos_t(T a, T* s, T* c);
template <>
__device__ __forceinline__ void sincos_t<double>(double a, double* s, double* c) {
  sincos_f64(a, s, c);
}
template <>
__device__ __forceinline__ void sincos_t<float>(float a, float* s, float* c) {
  ::sincosf(a, s, c);
}

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
  static __device__ __forceinline__ void dxdt(T* f, const T* x, const T* u, T* trig) {  // model.hpp:37-42
    T sd, cd, s1, c1;
    sincos_t<T>(x[0] - x[1], &sd, &cd);
    sincos_t<T>(x[1], &s1, &c1);
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

  static __device__ __forceinline__ void dxdt(T* f, const T* x, const T* u, T*) {  // model.hpp:36-41
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

  static __device__ __forceinline__ void dxdt(T* f, const T* x, const T* u, T*) {  // model.hpp:36-39
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
