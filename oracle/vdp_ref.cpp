// ORACLE — TEST INFRASTRUCTURE ONLY, buildable only where /root/reference is mounted.
// The user model of tests/user_models/vdp_model.hpp driven by the UNMODIFIED reference controller
// (`Cgmres<VdpModel>` from /root/reference/include/cgmres.hpp, included where it lies): this is what generates
// tests/golden/user_vdp_closed_loop.txt, so the user-model fixture is pinned to the reference itself, not to the
// repo's restatement.  Same scenario and output format as tests/user_models/vdp_oracle.cpp.
//   g++ -O3 -std=c++17 -ffp-contract=off -I/root/reference/include -I<repo> oracle/vdp_ref.cpp -o oracle/_ref/vdp_ref
// The reference leaves dUdt uninitialised (cgmres.hpp:14) and reads it on the first tick: a zero-filling
// operator new[] makes the first tick deterministic.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>

void* operator new[](std::size_t n) {
  void* p = std::calloc(n ? n : 1, 1);
  if (!p) throw std::bad_alloc();
  return p;
}
void operator delete[](void* p) noexcept { std::free(p); }
void operator delete[](void* p, std::size_t) noexcept { std::free(p); }

#include "cgmres.hpp"  // the reference's (via -I/root/reference/include)
#include "tests/user_models/vdp_model.hpp"

int main(int argc, char** argv) {
  const int B = argc > 1 ? atoi(argv[1]) : 4, ticks = argc > 2 ? atoi(argv[2]) : 20;
  for (int b = 0; b < B; ++b) {
    double x[2] = {1.0 + 0.1 * b, -0.5 + 0.05 * b}, p[2] = {0.2 * b, 0.05 * (b % 3)}, u0[3] = {0.1, 1.9, 0.03};
    Cgmres<VdpModel> c;
    c.set_ptau_repeat(p);
    c.init_u0(u0);
    c.init_u0_newton(u0, x, p, 10);
    for (int t = 0; t < ticks; ++t) {
      double u[3], f[2];
      c.control(u, x);
      printf("%d %d %.17g %.17g %.17g %.17g %.17g\n", b, t, u[0], u[1], u[2], x[0], x[1]);
      VdpModel::dxdt(f, x, u, p);
      for (int i = 0; i < 2; ++i) x[i] = x[i] + f[i] * VdpModel::dt;
    }
  }
  return 0;
}
