// Host-side problem definitions for the example programs, written against the Model concept that
// Cgmres<Model> expects (reference include/cgmres.hpp:11,64,68-69,83,137,145,150,160,179-188):
//   static constexpr uint16_t dim_x, dim_u, dim_p, dv, k_max;  static constexpr double dt, h, zeta, Tf, alpha, tol;
//   static void dxdt(ret,x,u,p); dPhidx(ret,x,p); dHdx(ret,x,u,p,lmd); dHdu(ret,x,u,p,lmd); ddHduu(ret,x,u,p,lmd);
// The three plants are the ones the reference ships as examples (arm_type_inverted_pendulum/, mass_spring_damper/,
// semiactive_damper/ : each model.hpp + simulator.hpp); the horizon sizes are template parameters here so one
// definition serves the shipped sizes and the BASELINE.json sizes (N = 50, k_max = 10).
// On the GPU these classes are only used for the registry fingerprint (cgmres.hpp, identify_model): the tick itself
// runs the device implementation in cgmres_cpp_amd/csrc/models.hip.h.
#pragma once
#include <math.h>
#include <stdint.h>

namespace examples {

// ---- arm-type inverted pendulum: 4 states, u = (torque, slack, multiplier), p = target angles ------------
struct PendulumPlant {
  static constexpr double As = 6.25, Bs = 15.6, A52 = 39.1111, C22 = 0.0407448;
  static constexpr double A32a = 5.65635, A32 = 0.905016, A32b = 14.1183;
  static void rhs(double* f, const double* x, const double* u) {
    const double d = x[0] - x[1];
    f[0] = x[2];
    f[1] = x[3];
    f[2] = -As * x[2] + Bs * u[0];
    f[3] = A32 * x[2] * x[2] * sin(d) + A52 * sin(x[1]) - A32b * cos(d) * u[0] + A32a * cos(d) * x[2] +
           C22 * (x[2] - x[3]);
  }
};

template <uint16_t DV = 25, uint16_t KMAX = 5>
struct PendulumModel : PendulumPlant {
  static constexpr uint16_t dim_x = 4, dim_u = 3, dim_p = 2, dv = DV, k_max = KMAX;
  static constexpr double dt = 0.001, h = 0.002, zeta = 1000.0, Tf = 0.5, alpha = 0.5, tol = 1e-6;
  static constexpr double sf[4] = {3.0, 1.0, 0.0, 0.0}, q[4] = {1.0, 1.0, 0.0, 0.0};
  static constexpr double r0 = 1.0, r1 = 0.1, uc = 0.0, ur = 3.0;  // |u| <= 3 through the slack variable

  static void dxdt(double* f, const double* x, const double* u, const double*) { rhs(f, x, u); }
  static void dPhidx(double* g, const double* x, const double* p) {
    g[0] = (x[0] - p[0]) * sf[0];
    g[1] = (x[1] - p[1]) * sf[1];
    g[2] = x[2] * sf[2];
    g[3] = x[3] * sf[3];
  }
  static void dHdx(double* g, const double* x, const double* u, const double* p, const double* l) {
    const double d = x[0] - x[1], sd = sin(d), cd = cos(d);
    g[0] = (x[0] - p[0]) * q[0] + l[3] * (A32 * x[2] * x[2] * cd + A32b * sd * u[0] - A32a * sd * x[2]);
    g[1] = (x[1] - p[1]) * q[1] +
           l[3] * (-A32 * x[2] * x[2] * cd + A52 * cos(x[1]) - A32b * sd * u[0] + A32a * sd * x[2]);
    g[2] = x[2] * q[2] + l[0] - l[2] * As + l[3] * (2.0 * A32 * x[2] * sd + A32a * cd + C22);
    g[3] = x[3] * q[3] + l[1] - l[3] * C22;
  }
  static void dHdu(double* g, const double* x, const double* u, const double*, const double* l) {
    g[0] = r0 * u[0] + l[2] * Bs - l[3] * A32b * cos(x[0] - x[1]) + u[2] * (2.0 * u[0] - 2.0 * uc);
    g[1] = -0.5 * r1 + 2.0 * u[2] * u[1];
    g[2] = (u[0] - uc) * (u[0] - uc) + u[1] * u[1] - ur * ur;
  }
  static void ddHduu(double* m, const double*, const double* u, const double*, const double*) {
    const double v[9] = {r0 + 2 * u[2], 0, 2 * u[0] - 2 * uc, 0, 2 * u[2], 2 * u[1], 2 * u[0] - 2 * uc, 2 * u[1], 0};
    for (int i = 0; i < 9; ++i) m[i] = v[i];
  }
};

// ---- two-mass spring damper: 4 states, two bounded forces (+2 slacks, +2 multipliers), p = position targets ---
struct MsdPlant {
  static constexpr double m1 = 1.0, m2 = 1.0, d1 = 1.0, d2 = 1.0, k1 = 1.0, k2 = 1.0;
  static void rhs(double* f, const double* x, const double* u) {
    f[0] = x[2];
    f[1] = x[3];
    f[2] = -(k1 * k2) / m1 * x[0] + k2 / m1 * x[1] - (d1 + d2) / m1 * x[2] + d2 / m1 * x[3] + u[0] / m1;
    f[3] = k2 / m2 * x[0] - k2 / m2 * x[1] + d2 / m2 * x[2] - d2 / m2 * x[3] + u[1] / m2;
  }
};

template <uint16_t DV = 50, uint16_t KMAX = 5>
struct MsdModel : MsdPlant {
  static constexpr uint16_t dim_x = 4, dim_u = 6, dim_p = 2, dv = DV, k_max = KMAX;
  static constexpr double dt = 0.001, h = 0.002, zeta = 1000.0, Tf = 1.0, alpha = 0.5, tol = 1e-6;
  static constexpr double sf[4] = {10.0, 10.0, 1.0, 1.0}, q[4] = {1.0, 1.0, 10.0, 10.0};
  static constexpr double r[4] = {0.1, 0.1, 0.01, 0.01}, uc = 0.0, ur = 10.0;

  static void dxdt(double* f, const double* x, const double* u, const double*) { rhs(f, x, u); }
  static void dPhidx(double* g, const double* x, const double* p) {
    g[0] = -(p[0] - x[0]) * sf[0];
    g[1] = -(p[1] - x[1]) * sf[1];
    g[2] = x[2] * sf[2];
    g[3] = x[3] * sf[3];
  }
  static void dHdx(double* g, const double* x, const double*, const double* p, const double* l) {
    g[0] = -(p[0] - x[0]) * q[0] - (k1 + k2) / m1 * l[2] + k2 / m2 * l[3];  // (k1+k2) here, (k1*k2) in rhs: as shipped
    g[1] = -(p[1] - x[1]) * q[1] + k2 / m1 * l[2] - k2 / m2 * l[3];
    g[2] = x[2] * q[2] + l[0] - (d1 + d2) / m1 * l[2] + d2 / m2 * l[3];
    g[3] = x[3] * q[3] + l[1] + d2 / m1 * l[2] - d2 / m2 * l[3];
  }
  static void dHdu(double* g, const double*, const double* u, const double*, const double* l) {
    g[0] = r[0] * u[0] + l[2] / m1 + 2.0 * u[4] * (u[0] - uc);
    g[1] = r[1] * u[1] + l[3] / m2 + 2.0 * u[5] * (u[1] - uc);
    g[2] = -r[2] + 2.0 * u[4] * u[2];
    g[3] = -r[3] + 2.0 * u[5] * u[3];
    g[4] = (u[0] - uc) * (u[0] - uc) + u[2] * u[2] - ur * ur;
    g[5] = (u[1] - uc) * (u[1] - uc) + u[3] * u[3] - ur * ur;
  }
  static void ddHduu(double* m, const double*, const double* u, const double*, const double*) {
    for (int i = 0; i < 36; ++i) m[i] = 0;
    m[0] = r[0] + 2 * u[4], m[4] = 2 * (u[0] - uc), m[7] = r[1] + 2 * u[5], m[11] = 2 * (u[1] - uc);
    m[14] = 2 * u[4], m[16] = 2 * u[2], m[21] = 2 * u[5], m[23] = 2 * u[3];
    m[24] = 2 * (u[0] - uc), m[26] = 2 * u[2], m[31] = 2 * (u[1] - uc), m[33] = 2 * u[3];
  }
};

// ---- semi-active damper: 2 states, damping coefficient in [0,1], no reference parameters ---------------------
struct SemiactivePlant {
  static constexpr double a = -1.0, b = -1.0;
  static void rhs(double* f, const double* x, const double* u) {
    f[0] = x[1];
    f[1] = a * x[0] + b * u[0] * x[1];
  }
};

template <uint16_t DV = 50, uint16_t KMAX = 5>
struct SemiactiveModel : SemiactivePlant {
  static constexpr uint16_t dim_x = 2, dim_u = 3, dim_p = 0, dv = DV, k_max = KMAX;
  static constexpr double dt = 0.001, h = 0.002, zeta = 1000.0, Tf = 1.0, alpha = 0.5, tol = 1e-6;
  static constexpr double sf[2] = {1.0, 10.0}, q[2] = {1.0, 10.0}, r0 = 1.0, r1 = 0.01, uc = 0.5, ur = 0.5;

  static void dxdt(double* f, const double* x, const double* u, const double*) { rhs(f, x, u); }
  static void dPhidx(double* g, const double* x, const double*) {
    g[0] = x[0] * sf[0];
    g[1] = x[1] * sf[1];
  }
  static void dHdx(double* g, const double* x, const double* u, const double*, const double* l) {
    g[0] = x[0] * q[0] + a * l[1];
    g[1] = x[1] * q[1] + l[0] + b * u[0] * l[1];
  }
  static void dHdu(double* g, const double* x, const double* u, const double*, const double* l) {
    g[0] = r0 * u[0] + b * x[1] * l[1] + 2 * u[2] * (u[0] - uc);
    g[1] = -r1 + 2 * u[1] * u[2];
    g[2] = (u[0] - uc) * (u[0] - uc) + u[1] * u[1] - ur * ur;
  }
  static void ddHduu(double* m, const double*, const double* u, const double*, const double*) {
    const double v[9] = {r0 + 2 * u[2], 0, 2 * (u[0] - uc), 0, 2 * u[2], 2 * u[1], 2 * (u[0] - uc), 2 * u[1], 0};
    for (int i = 0; i < 9; ++i) m[i] = v[i];
  }
};

}  // namespace examples
