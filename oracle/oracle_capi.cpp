// ORACLE — TEST INFRASTRUCTURE ONLY. C ABI (oracle/orc_api.h) over the CPU restatement.
// Build: see oracle/Makefile (g++ -O3 -ffp-contract=off, the reference's own -O3 without FMA).
#include <cstring>
#include <memory>
#include <vector>

#include "cgmres_oracle.hpp"
#define ORC_DEFINE_CAPI
#include "orc_base.hpp"

namespace {

template <class T>
std::vector<T> in(const double* p, size_t n) {
  std::vector<T> v(n ? n : 1);
  for (size_t i = 0; i < n; ++i) v[i] = T(p[i]);
  return v;
}
template <class T>
void out(double* dst, const T* src, size_t n) {
  if (!dst) return;
  for (size_t i = 0; i < n; ++i) dst[i] = double(src[i]);
}

template <class Model, class T>
struct Impl : OrcBase {
  oracle::Controller<Model, T> c;
  static constexpr int nx = Model::dim_x, nu = Model::dim_u, np = Model::dim_p;
  Impl(int dv, int kmax, double tol, int dtype) : c(dv, kmax, tol) {
    int d[7] = {nx, nu, np, dv, kmax, nu * dv, dtype};
    std::memcpy(dims, d, sizeof d);
    auto tu = Model::tuning();
    double q[5] = {tu.dt, tu.h, tu.zeta, tu.Tf, tu.alpha};
    std::memcpy(tun, q, sizeof q);
  }
  void set_ptau(const double* p) override {
    auto v = in<T>(p, np * (c.dv() + 1));
    c.set_ptau(v.data());
  }
  void init_u0(const double* u0) override {
    auto v = in<T>(u0, nu);
    c.init_u0(v.data());
  }
  void init_u0_newton(double* u0, const double* x0, const double* p0, int n) override {
    auto u = in<T>(u0, nu);
    auto x = in<T>(x0, nx);
    auto p = in<T>(p0, np);
    c.init_u0_newton(u.data(), x.data(), p.data(), n);
    out(u0, u.data(), nu);
  }
  void control(double* u, const double* x) override {
    auto xv = in<T>(x, nx);
    T uo[nu];
    c.control(uo, xv.data());
    out(u, uo, nu);
  }
  void get_state(double* t, double* U, double* dUdt) override {
    if (t) *t = double(c.t());
    out(U, c.U().data(), c.len());
    out(dUdt, c.dUdt().data(), c.len());
  }
  void set_state(double t, const double* U, const double* dUdt) override {
    c.t() = T(t);
    for (int i = 0; i < c.len(); ++i) {
      c.U()[i] = T(U[i]);
      c.dUdt()[i] = T(dUdt[i]);
    }
  }
  void F(double* ret, const double* U, const double* x, double t) override {
    auto Uv = in<T>(U, c.len());
    auto xv = in<T>(x, nx);
    std::vector<T> r(c.len());
    c.F(r.data(), Uv.data(), xv.data(), T(t));
    out(ret, r.data(), c.len());
  }
  // cgmres.hpp:83-96 without the solve
  void prepare(double* b, const double* x) override {
    auto xv = in<T>(x, nx);
    const auto tu = Model::tuning();
    const T h = T(tu.h), zeta = T(tu.zeta);
    T* xh = c.xh().data();
    Model::dxdt(xh, xv.data(), &c.U()[0], &c.ptau()[0]);
    for (int i = 0; i < nx; ++i) xh[i] = xh[i] * h;
    for (int i = 0; i < nx; ++i) xh[i] = xh[i] + xv[i];
    c.F(c.Fh().data(), c.U().data(), xh, c.t() + h);
    std::vector<T> bv(c.len());
    c.F(bv.data(), c.U().data(), xv.data(), c.t());
    const T cc = (1 - zeta * h);
    for (auto& e : bv) e = e * cc;
    for (int i = 0; i < c.len(); ++i) bv[i] = bv[i] - c.Fh()[i];
    const T inv_h = T(1.0) / h;
    for (auto& e : bv) e = e * inv_h;
    out(b, bv.data(), c.len());
  }
  void Ax(double* o, const double* v) override {
    auto vv = in<T>(v, c.len());
    std::vector<T> r(c.len());
    c.Ax(r.data(), vv.data());
    out(o, r.data(), c.len());
  }
  void gmres(double* x, const double* b) override {
    auto xv = in<T>(x, c.len());
    auto bv = in<T>(b, c.len());
    c.gmres(xv.data(), bv.data());
    out(x, xv.data(), c.len());
  }
  void get_krylov(double* V, double* H, double* rho, double* g) override {
    out(V, c.V().data(), c.V().size());
    out(H, c.H().data(), c.H().size());
    out(rho, c.rho().data(), c.rho().size());
    out(g, c.g().data(), c.g().size());
  }
  void last_solve(int* o) override {
    o[0] = c.n_ax();
    o[1] = c.k_used();
    o[2] = c.exit_reason();
  }
  void plant(double* f, const double* x, const double* u) override {
    auto xv = in<T>(x, nx);
    auto uv = in<T>(u, nu);
    T r[nx];
    Model::plant(r, xv.data(), uv.data());
    out(f, r, nx);
  }
};

template <class T>
OrcBase* make(int model, int dv, int kmax, double tol, int dtype) {
  switch (model) {
    case oracle::kPendulum:
      return new Impl<oracle::Pendulum<T>, T>(dv, kmax, tol < 0 ? oracle::Pendulum<T>::tol : tol, dtype);
    case oracle::kMassSpringDamper:
      return new Impl<oracle::MassSpringDamper<T>, T>(dv, kmax, tol < 0 ? oracle::MassSpringDamper<T>::tol : tol,
                                                      dtype);
    case oracle::kSemiactiveDamper:
      return new Impl<oracle::SemiactiveDamper<T>, T>(dv, kmax, tol < 0 ? oracle::SemiactiveDamper<T>::tol : tol,
                                                      dtype);
  }
  return nullptr;
}

}  // namespace

OrcBase* orc_factory(int model, int dv, int kmax, double tol, int dtype) {
  return dtype == 1 ? make<float>(model, dv, kmax, tol, 1) : make<double>(model, dv, kmax, tol, 0);
}
