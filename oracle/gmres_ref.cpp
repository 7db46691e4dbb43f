// ORACLE — TEST INFRASTRUCTURE ONLY, buildable only where /root/reference is mounted.
// The UNMODIFIED reference solver `class Gmres` (/root/reference/include/gmres.hpp:8-129, included where it lies) with a
// subclass whose Ax_func is one of the operators of tests/user_models/gmres_ops.hpp.  Generates the fixtures
// tests/golden/user_gmres_<op>.txt: the stand-alone solver of the device library (cgmres_hip_gmres_user) is pinned to
// the reference itself.
//   g++ -O3 -std=c++17 -ffp-contract=off -I/root/reference/include -I<repo> oracle/gmres_ref.cpp -o oracle/_ref/gmres_ref
//   oracle/_ref/gmres_ref spd|convdiff|convdiff150|convdiff300  ->  one line per instance: b k_max tol | x[0..len) in %.17g
// Scenario of instance i (deterministic, shared with the tests): p_j = 0.3 + 0.11 i + 0.05 j (SPD) or
// (0.4 + 0.07 i, 0.35 - 0.02 i) (conv-diff); b_e = sin(0.3 e + 0.5 i) + 0.1 e; x0_e = 0.01 (e - i); k_max and tol per
// case below (early exits included).
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>

void* operator new[](std::size_t n) {  // zero-filled heap: the reference reads v_mat / h_mat before writing parts of them
  void* p = std::calloc(n ? n : 1, 1);
  if (!p) throw std::bad_alloc();
  return p;
}
void operator delete[](void* p) noexcept { std::free(p); }
void operator delete[](void* p, std::size_t) noexcept { std::free(p); }

#include "gmres.hpp"  // the reference's (via -I/root/reference/include)
#include "tests/user_models/gmres_ops.hpp"

template <class Op>
class Solver : public Gmres {
 public:
  Solver(uint16_t k_max, double tol, const double* p) : Gmres(Op::len, k_max, tol), p_(p) {}
  void solve(double* x, const double* b) { gmres(x, b); }

 private:
  void Ax_func(double* Ax, const double* x) override { Op::Ax(Ax, x, p_); }
  const double* p_;
};

template <class Op>
void run(bool spd, int n_inst = 12, int k_short = 5) {
  constexpr int L = Op::len;
  const int kmaxs[3] = {spd ? 12 : 20, 30, k_short};
  const double tols[3] = {1e-9, 1e-6, 0.0};
  for (int c = 0; c < 3; ++c)
    for (int i = 0; i < n_inst; ++i) {
      double p[2] = {spd ? 0.3 + 0.11 * i : 0.4 + 0.07 * i, spd ? 0.0 : 0.35 - 0.02 * i};
      if (spd) p[0] += 0.05 * 0;
      double x[L], b[L];
      for (int e = 0; e < L; ++e) b[e] = std::sin(0.3 * e + 0.5 * i) + 0.1 * e, x[e] = 0.01 * (e - i);
      Solver<Op> s(kmaxs[c], tols[c], p);
      s.solve(x, b);
      printf("%d %d %.17g", i, kmaxs[c], tols[c]);
      for (int e = 0; e < L; ++e) printf(" %.17g", x[e]);
      printf("\n");
    }
}

int main(int argc, char** argv) {
  if (argc > 1 && !strcmp(argv[1], "spd"))
    run<SpdTridiagOp>(true);
  else if (argc > 1 && !strcmp(argv[1], "convdiff150"))
    run<ConvDiffOp150>(false, 4, 10);
  else if (argc > 1 && !strcmp(argv[1], "convdiff300"))
    run<ConvDiffOp300>(false, 4, 10);
  else
    run<ConvDiffOp>(false);
  return 0;
}
