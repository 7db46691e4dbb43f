// ORACLE — TEST INFRASTRUCTURE ONLY. Never included by the product (cgmres_cpp_amd/, include/).
//
// CPU restatement of the three problem definitions the reference ships
// (arm_type_inverted_pendulum/model.hpp, mass_spring_damper/model.hpp,
// semiactive_damper/model.hpp; multiple_controller/model1.hpp ≡ MSD, model2.hpp ≡ pendulum).
// Every expression keeps the reference's left-to-right evaluation order so that an
// fp64 build without FMA contraction reproduces the reference bit for bit (checked by
// tests/test_oracle_vs_ref.py against oracle/_ref).
//
// The horizon sizes (dv, k_max) and tol are run-time values here: the reference fixes them
// at compile time, BASELINE.json's configs override them (SURVEY.md §8 size table).
#pragma once
#include <cmath>

namespace oracle {

enum ModelId { kPendulum = 0, kMassSpringDamper = 1, kSemiactiveDamper = 2 };

// Tuning constants common to a model: reference */model.hpp "Sampling period" … "tol" block.
struct Tuning {
  double dt, h, zeta, Tf, alpha;
};

// ---------------------------------------------------------------------------------------
// Arm-type inverted pendulum — arm_type_inverted_pendulum/model.hpp:37-62 (+ :64-76 Hessian)
// x = (th_arm, th_pend, w_arm, w_pend), u = (torque cmd, slack, multiplier), p = (target0, target1)
// ---------------------------------------------------------------------------------------
template <class T>
struct Pendulum {
  static constexpr int dim_x = 4, dim_u = 3, dim_p = 2;
  static constexpr int shipped_dv = 25, shipped_kmax = 5;  // model.hpp:27,35
  static constexpr Tuning tuning() { return {0.001, 0.002, 1000.0, 0.5, 0.5}; }  // model.hpp:21-31
  static constexpr double tol = 1e-6;                                             // model.hpp:33

  // weights / limits / plant constants: model.hpp:80-98
  static constexpr T sf0 = T(3.0), sf1 = T(1.0), sf2 = T(0.0), sf3 = T(0.0);
  static constexpr T q0 = T(1.0), q1 = T(1.0), q2 = T(0.0), q3 = T(0.0);
  static constexpr T r0 = T(1.0), r1 = T(0.1);
  static constexpr T umin = T(-3.0), umax = T(3.0);
  static constexpr T uc = (umax + umin) / T(2.0), ur = (umax - umin) / T(2.0);
  static constexpr T As = T(6.25), Bs = T(15.6), A52 = T(39.1111), C22 = T(0.0407448);
  static constexpr T A32a = T(5.65635), A32 = T(0.905016), A32b = T(14.1183);

  static void dxdt(T* f, const T* x, const T* u, const T*) {  // model.hpp:37-42
    using std::cos;
    using std::sin;
    f[0] = x[2];
    f[1] = x[3];
    f[2] = -As * x[2] + Bs * u[0];
    f[3] = A32 * x[2] * x[2] * sin(x[0] - x[1]) + A52 * sin(x[1]) - A32b * cos(x[0] - x[1]) * u[0] +
           A32a * cos(x[0] - x[1]) * x[2] + C22 * (x[2] - x[3]);
  }
  static void dPhidx(T* g, const T* x, const T* p) {  // model.hpp:44-49
    g[0] = (x[0] - p[0]) * sf0;
    g[1] = (x[1] - p[1]) * sf1;
    g[2] = x[2] * sf2;
    g[3] = x[3] * sf3;
  }
  static void dHdx(T* g, const T* x, const T* u, const T* p, const T* l) {  // model.hpp:51-56
    using std::cos;
    using std::sin;
    g[0] = (x[0] - p[0]) * q0 +
           l[3] * (A32 * x[2] * x[2] * cos(x[0] - x[1]) + A32b * sin(x[0] - x[1]) * u[0] -
                   A32a * sin(x[0] - x[1]) * x[2]);
    g[1] = (x[1] - p[1]) * q1 +
           l[3] * (-A32 * x[2] * x[2] * cos(x[0] - x[1]) + A52 * cos(x[1]) -
                   A32b * sin(x[0] - x[1]) * u[0] + A32a * sin(x[0] - x[1]) * x[2]);
    g[2] = x[2] * q2 + l[0] - l[2] * As +
           l[3] * (T(0.2e1) * A32 * x[2] * sin(x[0] - x[1]) + A32a * cos(x[0] - x[1]) + C22);
    g[3] = x[3] * q3 + l[1] - l[3] * C22;
  }
  static void dHdu(T* g, const T* x, const T* u, const T*, const T* l) {  // model.hpp:58-62
    using std::cos;
    g[0] = (r0 * u[0]) + l[2] * Bs - l[3] * A32b * cos(x[0] - x[1]) + (u[2] * (T(2.0) * u[0] - T(2.0) * uc));
    g[1] = T(-0.5) * r1 + (T(2.0) * u[2] * u[1]);
    g[2] = (u[0] - uc) * (u[0] - uc) + u[1] * u[1] - ur * ur;
  }
  // column-major dim_u x dim_u (entry [dim_u*col+row]); model.hpp:64-76
  static void ddHduu(T* m, const T*, const T* u, const T*, const T*) {
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
  // plant used by the closed-loop harness: arm_type_inverted_pendulum/simulator.hpp:14-19 (same f)
  static void plant(T* f, const T* x, const T* u) { dxdt(f, x, u, nullptr); }
};

// ---------------------------------------------------------------------------------------
// Two-mass spring damper — mass_spring_damper/model.hpp:36-64 (+ :66-108 Hessian)
// All physical constants are 1.0 (model.hpp:123); the coefficient expressions are kept as
// written, including the (k1*k2) vs (k1+k2) mismatch between dxdt (:39) and dHdx (:51).
// ---------------------------------------------------------------------------------------
template <class T>
struct MassSpringDamper {
  static constexpr int dim_x = 4, dim_u = 6, dim_p = 2;
  static constexpr int shipped_dv = 50, shipped_kmax = 5;  // model.hpp:26,34
  static constexpr Tuning tuning() { return {0.001, 0.002, 1000.0, 1.0, 0.5}; }  // model.hpp:20-30
  static constexpr double tol = 1e-6;

  static constexpr T sf0 = T(10.0), sf1 = T(10.0), sf2 = T(1.0), sf3 = T(1.0);  // model.hpp:112-114
  static constexpr T q0 = T(1.0), q1 = T(1.0), q2 = T(10.0), q3 = T(10.0);
  static constexpr T r0 = T(0.1), r1 = T(0.1), r2 = T(0.01), r3 = T(0.01);
  static constexpr T umin = T(-10.0), umax = T(10.0);  // model.hpp:117-120
  static constexpr T uc = (umax + umin) / T(2.0), ur = (umax - umin) / T(2.0);
  static constexpr T m1 = T(1.0), m2 = T(1.0), d1 = T(1.0), d2 = T(1.0), k1 = T(1.0), k2 = T(1.0);

  static void dxdt(T* f, const T* x, const T* u, const T*) {  // model.hpp:36-41
    f[0] = x[2];
    f[1] = x[3];
    f[2] = -(k1 * k2) / m1 * x[0] + k2 / m1 * x[1] - (d1 + d2) / m1 * x[2] + d2 / m1 * x[3] + u[0] / m1;
    f[3] = k2 / m2 * x[0] - k2 / m2 * x[1] + d2 / m2 * x[2] - d2 / m2 * x[3] + u[1] / m2;
  }
  static void dPhidx(T* g, const T* x, const T* p) {  // model.hpp:43-48
    g[0] = -(p[0] - x[0]) * sf0;
    g[1] = -(p[1] - x[1]) * sf1;
    g[2] = x[2] * sf2;
    g[3] = x[3] * sf3;
  }
  static void dHdx(T* g, const T* x, const T*, const T* p, const T* l) {  // model.hpp:50-55
    g[0] = -(p[0] - x[0]) * q0 - (k1 + k2) / m1 * l[2] + k2 / m2 * l[3];
    g[1] = -(p[1] - x[1]) * q1 + k2 / m1 * l[2] - k2 / m2 * l[3];
    g[2] = x[2] * q2 + l[0] - (d1 + d2) / m1 * l[2] + d2 / m2 * l[3];
    g[3] = x[3] * q3 + l[1] + d2 / m1 * l[2] - d2 / m2 * l[3];
  }
  static void dHdu(T* g, const T*, const T* u, const T*, const T* l) {  // model.hpp:57-64
    g[0] = r0 * u[0] + l[2] / m1 + T(2.0) * u[4] * (u[0] - uc);
    g[1] = r1 * u[1] + l[3] / m2 + T(2.0) * u[5] * (u[1] - uc);
    g[2] = -r2 + T(2.0) * u[4] * u[2];
    g[3] = -r3 + T(2.0) * u[5] * u[3];
    g[4] = (u[0] - uc) * (u[0] - uc) + u[2] * u[2] - ur * ur;
    g[5] = (u[1] - uc) * (u[1] - uc) + u[3] * u[3] - ur * ur;
  }
  static void ddHduu(T* m, const T*, const T* u, const T*, const T*) {  // model.hpp:66-108
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
  static void plant(T* f, const T* x, const T* u) { dxdt(f, x, u, nullptr); }  // simulator.hpp:14-19
};

// ---------------------------------------------------------------------------------------
// Semi-active damper — semiactive_damper/model.hpp:36-55 (+ :57-69 Hessian); dim_p = 0
// ---------------------------------------------------------------------------------------
template <class T>
struct SemiactiveDamper {
  static constexpr int dim_x = 2, dim_u = 3, dim_p = 0;
  static constexpr int shipped_dv = 50, shipped_kmax = 5;  // model.hpp:26,34
  static constexpr Tuning tuning() { return {0.001, 0.002, 1000.0, 1.0, 0.5}; }  // model.hpp:20-30
  static constexpr double tol = 1e-6;

  static constexpr T sf0 = T(1.0), sf1 = T(10.0);  // model.hpp:73-76
  static constexpr T q0 = T(1.0), q1 = T(10.0);
  static constexpr T r0 = T(1.0), r1 = T(0.01);
  static constexpr T umin = T(0.0), umax = T(1.0);  // model.hpp:79-82
  static constexpr T uc = (umax + umin) / T(2.0), ur = (umax - umin) / T(2.0);
  static constexpr T a = T(-1.0), b = T(-1.0);  // model.hpp:85-86

  static void dxdt(T* f, const T* x, const T* u, const T*) {  // model.hpp:36-39
    f[0] = x[1];
    f[1] = a * x[0] + b * u[0] * x[1];
  }
  static void dPhidx(T* g, const T* x, const T*) {  // model.hpp:41-44
    g[0] = x[0] * sf0;
    g[1] = x[1] * sf1;
  }
  static void dHdx(T* g, const T* x, const T* u, const T*, const T* l) {  // model.hpp:46-49
    g[0] = x[0] * q0 + a * l[1];
    g[1] = x[1] * q1 + l[0] + b * u[0] * l[1];
  }
  static void dHdu(T* g, const T* x, const T* u, const T*, const T* l) {  // model.hpp:51-55
    g[0] = r0 * u[0] + b * x[1] * l[1] + 2 * u[2] * (u[0] - uc);
    g[1] = -r1 + 2 * u[1] * u[2];
    g[2] = (u[0] - uc) * (u[0] - uc) + u[1] * u[1] - ur * ur;
  }
  static void ddHduu(T* m, const T*, const T* u, const T*, const T*) {  // model.hpp:57-69
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
  static void plant(T* f, const T* x, const T* u) { dxdt(f, x, u, nullptr); }  // simulator.hpp:14-17
};

}  // namespace oracle
