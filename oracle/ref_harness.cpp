// ORACLE — TEST INFRASTRUCTURE ONLY, buildable only where /root/reference is mounted.
//
// Compiles the UNMODIFIED reference headers where they lie (-I/root/reference/include and the
// example directories, see oracle/Makefile) into oracle/_ref/libref.so behind oracle/orc_api.h.
// No reference source is copied: this file only #includes it.  Techniques:
//   * white-box access: libc headers first, then `#define private public` around the includes;
//   * BASELINE sizes: a subclass of the example's Model shadows dv / k_max / tol;
//   * Arnoldi count: the subclass wraps Model::dHdu and counts calls (one F-eval = dv calls);
//   * deterministic first tick: dUdt (uninitialised in the reference ctor, cgmres.hpp:14) is zeroed;
//   * fp32 build (-DREF_F32): `#define double float` around the reference includes.
// This translation unit is built twice (fp64 / fp32); oracle/ref_capi.cpp dispatches.
#include <float.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <cmath>
#include <vector>

#include "orc_base.hpp"

#ifdef REF_F32
#define REF_NS ref32
#define double float
#else
#define REF_NS ref64
#endif

namespace REF_NS {
#define private public
#define protected public
#include "include/cgmres.hpp"  // pulls gmres.hpp, matrix.hpp (relative to -I/root/reference)
#undef private
#undef protected
namespace pend {
#include "arm_type_inverted_pendulum/model.hpp"
#include "arm_type_inverted_pendulum/simulator.hpp"
}  // namespace pend
namespace msd {
#include "mass_spring_damper/model.hpp"
#include "mass_spring_damper/simulator.hpp"
}  // namespace msd
namespace semi {
#include "semiactive_damper/model.hpp"
#include "semiactive_damper/simulator.hpp"
}  // namespace semi

typedef double real;  // float in the REF_F32 build (macro still active here)
}  // namespace REF_NS

#ifdef REF_F32
#undef double
#endif

namespace REF_NS {

// Size/tol override + dHdu call counter, without editing the example's Model.
template <class M, int DV, int KM, int ZERO_TOL>
struct Sized : M {
  static constexpr uint16_t dv = DV;
  static constexpr uint16_t k_max = KM;
  static constexpr real tol = ZERO_TOL ? real(0.0) : real(1e-6);
  static inline long calls = 0;
  static void dHdu(real* r, const real* x, const real* u, const real* p, const real* l) {
    ++calls;
    M::dHdu(r, x, u, p, l);
  }
};

template <class SM, class Sim>
struct RefImpl : OrcBase {
  Cgmres<SM> c;
  static constexpr int nx = SM::dim_x, nu = SM::dim_u, np = SM::dim_p, dv = SM::dv, km = SM::k_max, len = nu * dv;
  int n_ax = 0;
  RefImpl(int dtype) {
    int d[7] = {nx, nu, np, dv, km, len, dtype};
    memcpy(dims, d, sizeof d);
    double q[5] = {double(SM::dt), double(SM::h), double(SM::zeta), double(SM::Tf), double(SM::alpha)};
    memcpy(tun, q, sizeof q);
    for (int i = 0; i < len; ++i) c.dUdt[i] = 0, c.U[i] = 0, c.F_dxh_h[i] = 0;
    for (int i = 0; i < np * (dv + 1); ++i) c.ptau[i] = 0;
    for (int i = 0; i < len * (km + 1); ++i) c.v_mat[i] = 0;
    for (int i = 0; i < (km + 1) * (km + 1); ++i) c.h_mat[i] = 0;
    for (int i = 0; i < km + 1; ++i) c.rho_e_vec[i] = 0;
    for (int i = 0; i < 3 * km; ++i) c.g_vec[i] = 0;
  }
  static std::vector<real> in(const double* p, size_t n) {
    std::vector<real> v(n ? n : 1);
    for (size_t i = 0; i < n; ++i) v[i] = real(p[i]);
    return v;
  }
  static void out(double* d, const real* s, size_t n) {
    if (d)
      for (size_t i = 0; i < n; ++i) d[i] = double(s[i]);
  }
  void set_ptau(const double* p) override {
    auto v = in(p, np * (dv + 1));
    if (np) c.set_ptau(v.data());
  }
  void init_u0(const double* u) override {
    auto v = in(u, nu);
    c.init_u0(v.data());
  }
  void init_u0_newton(double* u0, const double* x0, const double* p0, int n) override {
    auto u = in(u0, nu);
    auto x = in(x0, nx);
    auto p = in(p0, np);
    c.init_u0_newton(u.data(), x.data(), p.data(), uint16_t(n));
    out(u0, u.data(), nu);
  }
  void control(double* u, const double* x) override {
    auto xv = in(x, nx);
    real uo[nu];
    SM::calls = 0;
    c.control(uo, xv.data());
    n_ax = int(SM::calls / dv) - 3;
    out(u, uo, nu);
  }
  void get_state(double* t, double* U, double* d) override {
    if (t) *t = double(c.t);
    out(U, c.U, len);
    out(d, c.dUdt, len);
  }
  void set_state(double t, const double* U, const double* d) override {
    c.t = real(t);
    for (int i = 0; i < len; ++i) c.U[i] = real(U[i]), c.dUdt[i] = real(d[i]);
  }
  void F(double* r, const double* U, const double* x, double t) override {
    auto Uv = in(U, len);
    auto xv = in(x, nx);
    std::vector<real> o(len);
    c.F_func(o.data(), Uv.data(), xv.data(), real(t));
    out(r, o.data(), len);
  }
  // the statements of Cgmres::control up to the solve, issued through the reference's own helpers
  void prepare(double* b, const double* x) override {
    auto xv = in(x, nx);
    std::vector<real> bv(len);
    SM::dxdt(c.x_dxh, xv.data(), &c.U[0], &c.ptau[0]);
    mul(c.x_dxh, c.x_dxh, c.h, nx);
    add(c.x_dxh, c.x_dxh, xv.data(), nx);
    c.F_func(c.F_dxh_h, c.U, c.x_dxh, c.t + c.h);
    c.F_func(bv.data(), c.U, xv.data(), c.t);
    mul(bv.data(), bv.data(), (1 - c.zeta * c.h), len);
    sub(bv.data(), bv.data(), c.F_dxh_h, len);
    div(bv.data(), bv.data(), c.h, len);
    out(b, bv.data(), len);
  }
  void Ax(double* o, const double* v) override {
    auto vv = in(v, len);
    std::vector<real> r(len);
    c.Ax_func(r.data(), vv.data());
    out(o, r.data(), len);
  }
  void gmres(double* x, const double* b) override {
    auto xv = in(x, len);
    auto bv = in(b, len);
    SM::calls = 0;
    c.gmres(xv.data(), bv.data());
    n_ax = int(SM::calls / dv) - 1;
    out(x, xv.data(), len);
  }
  void get_krylov(double* V, double* H, double* rho, double* g) override {
    out(V, c.v_mat, size_t(len) * (km + 1));
    out(H, c.h_mat, size_t(km + 1) * (km + 1));
    out(rho, c.rho_e_vec, km + 1);
    out(g, c.g_vec, 3 * km);
  }
  void last_solve(int* o) override {
    o[0] = n_ax;
    o[1] = -1;
    o[2] = -1;
  }
  void plant(double* f, const double* x, const double* u) override {
    auto xv = in(x, nx);
    auto uv = in(u, nu);
    real r[nx];
    Sim::dxdt(r, xv.data(), uv.data());
    out(f, r, nx);
  }
};

template <class M, class Sim, int DV, int KM>
OrcBase* pick_tol(double tol, int dtype) {
  if (tol < 0 || tol == 1e-6) return new RefImpl<Sized<M, DV, KM, 0>, Sim>(dtype);
  if (tol == 0.0) return new RefImpl<Sized<M, DV, KM, 1>, Sim>(dtype);
  return nullptr;
}

// Instantiation table: shipped sizes, BASELINE.json sizes and one tiny size.
template <class M, class Sim>
OrcBase* pick_size(int dv, int km, double tol, int dtype) {
#define REF_CASE(DV, KM) \
  if (dv == DV && km == KM) return pick_tol<M, Sim, DV, KM>(tol, dtype)
  REF_CASE(8, 3);
  REF_CASE(20, 5);
  REF_CASE(25, 5);
  REF_CASE(50, 5);
  REF_CASE(50, 10);
  REF_CASE(100, 20);
#undef REF_CASE
  return nullptr;
}

OrcBase* make(int model, int dv, int km, double tol, int dtype) {
  switch (model) {
    case 0:
      return pick_size<pend::Model, pend::Simulator>(dv, km, tol, dtype);
    case 1:
      return pick_size<msd::Model, msd::Simulator>(dv, km, tol, dtype);
    case 2:
      return pick_size<semi::Model, semi::Simulator>(dv, km, tol, dtype);
  }
  return nullptr;
}

}  // namespace REF_NS

#ifdef REF_F32
OrcBase* ref_make_f32(int model, int dv, int km, double tol) { return ref32::make(model, dv, km, tol, 1); }
#else
OrcBase* ref_make_f64(int model, int dv, int km, double tol) { return ref64::make(model, dv, km, tol, 0); }
#endif
