// Two OPERATORS for the stand-alone Gmres of the reference (include/gmres.hpp: an abstract solver whose subclass
// supplies Ax_func), written for this repository's tests.  Each struct is at once
//   * the body of a host subclass's Ax_func (oracle/gmres_ref.cpp drives the UNMODIFIED reference solver with it), and
//   * the device operator of csrc/user_operator.hip.h (compiled for gfx950 by cgmres_cpp_amd/plugin.py).
// params: per-instance scalars (a shift / a convection strength), so a batch solves different systems.
#pragma once

struct SpdTridiagOp {  // shifted 1-D Laplacian: tridiag(-1, 2 + p0, -1), symmetric positive definite for p0 > 0
  static constexpr int len = 24, n_params = 1;
  static void Ax(double* Ax, const double* x, const double* p) {
    for (int i = 0; i < len; ++i) {
      double a = (2.0 + p[0]) * x[i];
      if (i > 0) a = a - x[i - 1];
      if (i + 1 < len) a = a - x[i + 1];
      Ax[i] = a;
    }
  }
};

struct ConvDiffOp {  // nonsymmetric: convection-diffusion stencil with a varying diagonal and one far coupling per row
  static constexpr int len = 40, n_params = 2;
  static void Ax(double* Ax, const double* x, const double* p) {
    for (int i = 0; i < len; ++i) {
      double a = (2.0 + p[0] + 0.01 * i) * x[i];
      if (i > 0) a = a - (1.0 + p[1]) * x[i - 1];
      if (i + 1 < len) a = a - (1.0 - p[1]) * x[i + 1];
      a = a + 0.05 * x[(i * 7 + 3) % len];
      Ax[i] = a;
    }
  }
};

// The same stencil at the vector lengths of the controller's own solves (dim_u * dv = 150 for the pendulum at N = 50, 300
// for the two-mass system): what the one-wave-per-system solver is sized for.
template <int N>
struct ConvDiffOpN {
  static constexpr int len = N, n_params = 2;
  static void Ax(double* Ax, const double* x, const double* p) {
    for (int i = 0; i < len; ++i) {
      double a = (2.0 + p[0] + 0.01 * i) * x[i];
      if (i > 0) a = a - (1.0 + p[1]) * x[i - 1];
      if (i + 1 < len) a = a - (1.0 - p[1]) * x[i + 1];
      a = a + 0.05 * x[(i * 7 + 3) % len];
      Ax[i] = a;
    }
  }
};
using ConvDiffOp150 = ConvDiffOpN<150>;
using ConvDiffOp300 = ConvDiffOpN<300>;
