// ORACLE — TEST INFRASTRUCTURE ONLY.
// Abstract controller behind oracle/orc_api.h, shared by liboracle.so (restatement) and
// _ref/libref.so (unmodified reference).  A library defines `OrcBase* orc_factory(...)` and
// includes this header with ORC_DEFINE_CAPI to get the extern "C" entry points.
#pragma once
#include <chrono>
#include <cstring>
#include <thread>
#include <vector>

#include "orc_api.h"

struct OrcBase {
  virtual ~OrcBase() {}
  int dims[7];
  double tun[5];
  virtual void set_ptau(const double*) = 0;
  virtual void init_u0(const double*) = 0;
  virtual void init_u0_newton(double*, const double*, const double*, int) = 0;
  virtual void control(double*, const double*) = 0;
  virtual void get_state(double*, double*, double*) = 0;
  virtual void set_state(double, const double*, const double*) = 0;
  virtual void F(double*, const double*, const double*, double) = 0;
  virtual void prepare(double*, const double*) = 0;
  virtual void Ax(double*, const double*) = 0;
  virtual void gmres(double*, const double*) = 0;
  virtual void get_krylov(double*, double*, double*, double*) = 0;
  virtual void last_solve(int*) = 0;
  virtual void plant(double*, const double*, const double*) = 0;
};

OrcBase* orc_factory(int model, int dv, int kmax, double tol, int dtype);

#ifdef ORC_DEFINE_CAPI
extern "C" {
void* orc_create(int model, int dv, int kmax, double tol, int dtype) {
  if (dv < 1 || kmax < 1) return nullptr;
  return orc_factory(model, dv, kmax, tol, dtype);
}
void orc_destroy(void* c) { delete static_cast<OrcBase*>(c); }
void orc_dims(void* c, int* o) { std::memcpy(o, static_cast<OrcBase*>(c)->dims, 7 * sizeof(int)); }
void orc_tuning(void* c, double* o) { std::memcpy(o, static_cast<OrcBase*>(c)->tun, 5 * sizeof(double)); }
void orc_set_ptau(void* c, const double* p) { static_cast<OrcBase*>(c)->set_ptau(p); }
void orc_init_u0(void* c, const double* u) { static_cast<OrcBase*>(c)->init_u0(u); }
void orc_init_u0_newton(void* c, double* u0, const double* x0, const double* p0, int n) {
  static_cast<OrcBase*>(c)->init_u0_newton(u0, x0, p0, n);
}
void orc_control(void* c, double* u, const double* x) { static_cast<OrcBase*>(c)->control(u, x); }
void orc_get_state(void* c, double* t, double* U, double* d) { static_cast<OrcBase*>(c)->get_state(t, U, d); }
void orc_set_state(void* c, double t, const double* U, const double* d) {
  static_cast<OrcBase*>(c)->set_state(t, U, d);
}
void orc_F(void* c, double* r, const double* U, const double* x, double t) {
  static_cast<OrcBase*>(c)->F(r, U, x, t);
}
void orc_prepare(void* c, double* b, const double* x) { static_cast<OrcBase*>(c)->prepare(b, x); }
void orc_Ax(void* c, double* o, const double* v) { static_cast<OrcBase*>(c)->Ax(o, v); }
void orc_gmres(void* c, double* x, const double* b) { static_cast<OrcBase*>(c)->gmres(x, b); }
void orc_get_krylov(void* c, double* V, double* H, double* rho, double* g) {
  static_cast<OrcBase*>(c)->get_krylov(V, H, rho, g);
}
void orc_last_solve(void* c, int* o) { static_cast<OrcBase*>(c)->last_solve(o); }
void orc_plant(void* c, double* f, const double* x, const double* u) { static_cast<OrcBase*>(c)->plant(f, x, u); }
double orc_run_closed_loop(void** ctrls, int n, double* x, double* u, int ticks, int nthreads) {
  if (nthreads < 1) nthreads = 1;
  if (nthreads > n) nthreads = n;
  auto work = [&](int lo, int hi) {
    double f[16];
    for (int i = lo; i < hi; ++i) {
      OrcBase* c = static_cast<OrcBase*>(ctrls[i]);
      const int nx = c->dims[0], nu = c->dims[1];
      const double dt = c->tun[0];
      double* xi = x + size_t(i) * nx;
      double* ui = u + size_t(i) * nu;
      for (int t = 0; t < ticks; ++t) {
        c->control(ui, xi);
        c->plant(f, xi, ui);
        for (int k = 0; k < nx; ++k) xi[k] = xi[k] + f[k] * dt;
      }
    }
  };
  const auto t0 = std::chrono::steady_clock::now();
  std::vector<std::thread> th;
  for (int w = 0; w < nthreads; ++w) th.emplace_back(work, int(long(n) * w / nthreads), int(long(n) * (w + 1) / nthreads));
  for (auto& t : th) t.join();
  return std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
}
}
#endif
