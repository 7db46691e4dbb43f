// A main written against the reference's class surface: a class derived from `Gmres` (include/gmres.hpp) with an Ax_func
// of its own, calling the protected gmres(x, b) — here with the operator ConvDiffOp of gmres_ops.hpp.  Compiles against
// the reference's include/ (host solver) and against this repository's include/ (device solver) alike; the only added
// line is use_device_operator(), guarded so the reference build does not see it.
//   ./gmres_main [plugin.so]  ->  the same lines as oracle/_ref/gmres_ref convdiff
#include <cmath>
#include <cstdio>
#include <cstdlib>

#include "gmres.hpp"
#include "gmres_ops.hpp"

class Solver : public Gmres {
 public:
  Solver(uint16_t k_max, double tol, const double* p, const char* plugin) : Gmres(ConvDiffOp::len, k_max, tol), p_(p) {
#ifdef CGMRES_HIP_H_
    if (plugin) use_device_operator(plugin, p);
#endif
  }
  void solve(double* x, const double* b) { gmres(x, b); }

 private:
  void Ax_func(double* Ax, const double* x) override { ConvDiffOp::Ax(Ax, x, p_); }
  const double* p_;
};

int main(int argc, char** argv) {
  constexpr int L = ConvDiffOp::len;
  const int kmaxs[3] = {20, 30, 5};
  const double tols[3] = {1e-9, 1e-6, 0.0};
  for (int c = 0; c < 3; ++c)
    for (int i = 0; i < 12; ++i) {
      double p[2] = {0.4 + 0.07 * i, 0.35 - 0.02 * i}, x[L], b[L];
      for (int e = 0; e < L; ++e) b[e] = std::sin(0.3 * e + 0.5 * i) + 0.1 * e, x[e] = 0.01 * (e - i);
      Solver s(kmaxs[c], tols[c], p, argc > 1 ? argv[1] : nullptr);
      s.solve(x, b);
      printf("%d %d %.17g", i, kmaxs[c], tols[c]);
      for (int e = 0; e < L; ++e) printf(" %.17g", x[e]);
      printf("\n");
    }
  return 0;
}
