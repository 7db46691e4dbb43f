// TEST INFRASTRUCTURE: the oracle's generic controller (oracle/cgmres_oracle.hpp, pinned against the reference for
// the shipped models) instantiated for the user model of tests/user_models/vdp_model.hpp, closed loop with the
// model's own state equation as the plant.  Prints one line per tick: u[0..2] x[0..1] in %.17g.
//   g++ -O2 -std=c++17 -ffp-contract=off -I<repo> tests/user_models/vdp_oracle.cpp -o vdp_oracle && ./vdp_oracle B ticks
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "oracle/cgmres_oracle.hpp"
#include "tests/user_models/vdp_model.hpp"

struct M : VdpModel {
  static constexpr oracle::Tuning tuning() { return {dt, h, zeta, Tf, alpha}; }
};

int main(int argc, char** argv) {
  const int B = argc > 1 ? atoi(argv[1]) : 4, ticks = argc > 2 ? atoi(argv[2]) : 20;
  for (int b = 0; b < B; ++b) {
    // the same per-instance scenario as the test: deterministic, no RNG
    double x[2] = {1.0 + 0.1 * b, -0.5 + 0.05 * b}, p[2] = {0.2 * b, 0.05 * (b % 3)}, u0[3] = {0.1, 1.9, 0.03};
    oracle::Controller<M, double> c(M::dv, M::k_max, M::tol);
    c.set_ptau_repeat(p);
    c.init_u0(u0);
    c.init_u0_newton(u0, x, p, 10);
    for (int t = 0; t < ticks; ++t) {
      double u[3], f[2];
      c.control(u, x);
      printf("%d %d %.17g %.17g %.17g %.17g %.17g\n", b, t, u[0], u[1], u[2], x[0], x[1]);
      M::dxdt(f, x, u, p);
      for (int i = 0; i < 2; ++i) x[i] = x[i] + f[i] * M::dt;
    }
  }
  return 0;
}
