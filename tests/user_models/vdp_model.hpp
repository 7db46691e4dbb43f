// A USER model in the reference's Model concept (<example>/model.hpp:7-76): forced Van der Pol oscillator with a
// saturated input (slack + multiplier like the shipped examples) and a time-varying parameter p = (reference
// for x0, additive disturbance) that the state equation READS — none of the shipped models does that.
// Used by tests/test_user_model_plugin.py: compiled for the GPU through cgmres_cpp_amd/plugin.py and for the CPU
// into the oracle's generic Controller (tests/user_models/vdp_oracle.cpp).  Written for this repository.
#pragma once
#include <cmath>
#include <cstdint>

class VdpModel {
 public:
  static constexpr uint16_t dim_x = 2;
  static constexpr uint16_t dim_u = 3;  // input, slack, multiplier
  static constexpr uint16_t dim_p = 2;  // x0 reference, disturbance
  static constexpr double dt = 0.001;
  static constexpr double h = 0.002;
  static constexpr double zeta = 1000.0;
  static constexpr uint16_t dv = 30;
  static constexpr double Tf = 1.0;
  static constexpr double alpha = 0.5;
  static constexpr double tol = 1e-6;
  static constexpr uint16_t k_max = 6;

  static void dxdt(double* ret, const double* x, const double* u, const double* p) {
    ret[0] = x[1];
    ret[1] = mu * (1.0 - x[0] * x[0]) * x[1] - x[0] + u[0] + p[1];
  }
  static void dPhidx(double* ret, const double* x, const double* p) {
    ret[0] = (x[0] - p[0]) * sf0;
    ret[1] = x[1] * sf1;
  }
  static void dHdx(double* ret, const double* x, const double* u, const double* p, const double* lmd) {
    ret[0] = (x[0] - p[0]) * q0 + lmd[1] * (-2.0 * mu * x[0] * x[1] - 1.0);
    ret[1] = x[1] * q1 + lmd[0] + lmd[1] * mu * (1.0 - x[0] * x[0]);
  }
  static void dHdu(double* ret, const double* x, const double* u, const double* p, const double* lmd) {
    ret[0] = r0 * u[0] + lmd[1] + 2.0 * u[2] * (u[0] - uc);
    ret[1] = -r1 + 2.0 * u[2] * u[1];
    ret[2] = (u[0] - uc) * (u[0] - uc) + u[1] * u[1] - ur * ur;
  }
  // column-major like the reference's linsolve expects (mat[dim_u * col + row]); symmetric here
  static void ddHduu(double* ret, const double* x, const double* u, const double* p, const double* lmd) {
    ret[0] = r0 + 2.0 * u[2];
    ret[1] = 0.0;
    ret[2] = 2.0 * (u[0] - uc);
    ret[3] = 0.0;
    ret[4] = 2.0 * u[2];
    ret[5] = 2.0 * u[1];
    ret[6] = 2.0 * (u[0] - uc);
    ret[7] = 2.0 * u[1];
    ret[8] = 0.0;
  }

 private:
  static constexpr double mu = 1.0;
  static constexpr double sf0 = 2.0, sf1 = 1.0;
  static constexpr double q0 = 1.0, q1 = 1.0;
  static constexpr double r0 = 1.0, r1 = 0.1;
  static constexpr double uc = 0.0, ur = 2.0;
};
