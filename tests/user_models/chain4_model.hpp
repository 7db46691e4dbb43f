// A second USER model in the reference's Model concept (<example>/model.hpp:7-76), written for this repository's tests:
// two masses on a chain of springs, the outer one hardening (cubic), ONE unconstrained input on the first mass, a
// reference p[0] for the first position.  dim_x = 4 with dim_u = 1 is the shape that stresses the wg mapping's LDS plan
// for plugins: 4*4 + 1*4 + 4 = 24 stage coefficients per (stage, instance) against a row of only dv controls
// (CtxWg::lookahead_fits), so short horizons must fall back to the lane mapping.
// Chain4T<1> is the same plant with STIFF cost weights (1e5 on the position error, 1e-2 on the input): the
// costate-free part q of dH/dx is then five orders of magnitude above the Jacobian entries the plugin glue derives by
// differencing (csrc/user_model.hip.h: scaled probe costates).
#pragma once
#include <cmath>
#include <cstdint>

template <int STIFF>
class Chain4T {
 public:
  static constexpr uint16_t dim_x = 4;  // q1, q2, v1, v2
  static constexpr uint16_t dim_u = 1;
  static constexpr uint16_t dim_p = 1;  // reference for q1
  static constexpr double dt = 0.001;
  static constexpr double h = 0.002;
  static constexpr double zeta = 1000.0;
  static constexpr uint16_t dv = 25;
  static constexpr double Tf = 0.5;
  static constexpr double alpha = 0.5;
  static constexpr double tol = 1e-6;
  static constexpr uint16_t k_max = 5;

  static void dxdt(double* ret, const double* x, const double* u, const double* p) {
    ret[0] = x[2];
    ret[1] = x[3];
    ret[2] = -k1 * x[0] + k2 * (x[1] - x[0]) - c * x[2] + u[0];
    ret[3] = -k2 * (x[1] - x[0]) - c * x[3] - beta * x[1] * x[1] * x[1];
  }
  static void dPhidx(double* ret, const double* x, const double* p) {
    ret[0] = (x[0] - p[0]) * sf0;
    ret[1] = x[1] * sf1;
    ret[2] = x[2] * sf2;
    ret[3] = x[3] * sf2;
  }
  static void dHdx(double* ret, const double* x, const double* u, const double* p, const double* lmd) {
    ret[0] = (x[0] - p[0]) * q0 + lmd[2] * (-k1 - k2) + lmd[3] * k2;
    ret[1] = x[1] * q1 + lmd[2] * k2 + lmd[3] * (-k2 - 3.0 * beta * x[1] * x[1]);
    ret[2] = x[2] * q2 + lmd[0] - c * lmd[2];
    ret[3] = x[3] * q2 + lmd[1] - c * lmd[3];
  }
  static void dHdu(double* ret, const double* x, const double* u, const double* p, const double* lmd) {
    ret[0] = r0 * u[0] + lmd[2];
  }
  static void ddHduu(double* ret, const double* x, const double* u, const double* p, const double* lmd) { ret[0] = r0; }

 private:
  static constexpr double k1 = 1.0, k2 = 2.0, c = 0.3, beta = 0.5;
  static constexpr double q0 = STIFF ? 1.0e5 : 2.0, q1 = STIFF ? 3.0e4 : 1.0, q2 = STIFF ? 10.0 : 0.1;
  static constexpr double sf0 = STIFF ? 1.0e5 : 3.0, sf1 = STIFF ? 1.0e4 : 1.0, sf2 = 0.1;
  static constexpr double r0 = STIFF ? 1.0e-2 : 0.5;
};
using Chain4Model = Chain4T<0>;
using Chain4Stiff = Chain4T<1>;
