// TEST INFRASTRUCTURE: the oracle's generic controller (oracle/cgmres_oracle.hpp, pinned against the reference for the
// shipped models) instantiated for a user model header, closed loop with the model's own state equation as the plant.
//   g++ -O2 -std=c++17 -ffp-contract=off -I<repo> -DMODEL_HEADER='"tests/user_models/chain4_model.hpp"' -DMODEL_CLASS=Chain4Model
//       tests/user_models/user_oracle.cpp -o user_oracle && ./user_oracle B ticks dv k_max tol
// Scenario of instance b (deterministic, no RNG): x_i = 0.3 + 0.05 b - 0.11 i, p_j = 0.2 + 0.03 b + 0.01 j, u0 = 0.1.
// Prints one line per tick: b t k u[0..NU) x[0..NX) in %.17g; with a sixth argument "state" the line continues with the
// controller state the tick STARTED from — t, U[0..L), dUdt[0..L) — for teacher-forced comparisons.
#include <cstdio>
#include <cstdlib>

#include "oracle/cgmres_oracle.hpp"
#include MODEL_HEADER

struct M : MODEL_CLASS {
  static constexpr oracle::Tuning tuning() { return {dt, h, zeta, Tf, alpha}; }
};

int main(int argc, char** argv) {
  const int B = argc > 1 ? atoi(argv[1]) : 4, ticks = argc > 2 ? atoi(argv[2]) : 20;
  const int dv = argc > 3 ? atoi(argv[3]) : M::dv, kmax = argc > 4 ? atoi(argv[4]) : M::k_max;
  const double tol = argc > 5 ? atof(argv[5]) : M::tol;
  const bool with_state = argc > 6;
  constexpr int NX = M::dim_x, NU = M::dim_u, NP = M::dim_p;
  for (int b = 0; b < B; ++b) {
    double x[NX], p[NP > 0 ? NP : 1], u0[NU];
    for (int i = 0; i < NX; ++i) x[i] = 0.3 + 0.05 * b - 0.11 * i;
    for (int j = 0; j < NP; ++j) p[j] = 0.2 + 0.03 * b + 0.01 * j;
    for (int j = 0; j < NU; ++j) u0[j] = 0.1;
    oracle::Controller<M, double> c(dv, kmax, tol);
    c.set_ptau_repeat(p);
    c.init_u0(u0);
    c.init_u0_newton(u0, x, p, 10);
    for (int t = 0; t < ticks; ++t) {
      double u[NU], f[NX];
      const double t_before = c.t();
      const std::vector<double> U_before = c.U(), d_before = c.dUdt();
      c.control(u, x);
      printf("%d %d %d", b, t, c.n_ax());
      for (int j = 0; j < NU; ++j) printf(" %.17g", u[j]);
      for (int i = 0; i < NX; ++i) printf(" %.17g", x[i]);
      if (with_state) {
        printf(" %.17g", t_before);
        for (double v : U_before) printf(" %.17g", v);
        for (double v : d_before) printf(" %.17g", v);
      }
      printf("\n");
      M::dxdt(f, x, u, p);
      for (int i = 0; i < NX; ++i) x[i] = x[i] + f[i] * M::dt;
    }
  }
  return 0;
}
