// cgmres.hpp — drop-in façade for the reference's `template <class Model> class Cgmres : public Gmres`
// (reference include/cgmres.hpp:8-207) on top of the C ABI of libcgmres_hip.so (include/cgmres_hip.h).
//
// Same public surface, same argument meaning:
//   Cgmres()                                  :11      -> cgmres_hip_create (batch = 1)
//   double get_dtau(double t) const           :32-34   (host formula, identical)
//   void set_ptau(const double*)              :36-39   -> cgmres_hip_set_ptau
//   void set_ptau_repeat(const double*)       :41-49   -> cgmres_hip_set_ptau_repeat
//   void init_u0(const double*)               :51-59   -> cgmres_hip_init_u0
//   void init_u0_newton(u0, x0, p0, n_loop)   :61-76   -> cgmres_hip_init_u0_newton (u0 updated in place)
//   void control(double* u, const double* x)  :78-110  -> cgmres_hip_control
//   static constexpr dim_x, dim_u, dim_p, dt, h, zeta, dv, Tf, alpha   :179-188
// so the example programs (<example>/main.cpp) compile and link unchanged with `-I include -I <example>`
// and `-lcgmres_hip`.  A single Cgmres object is a batch of one controller on the GPU; to run thousands of
// controllers use CgmresBatch<Model> (cgmres_batch.hpp) — that is where the device earns its keep.
//
// `Model` is the user's host class (static constexpr sizes + static dxdt/dPhidx/dHdx/dHdu/ddHduu).  Device code
// cannot call it, so the constructor identifies it in the library's registry: same dim_x/dim_u/dim_p and the same
// values of dxdt/dPhidx/dHdx/dHdu at two probe points as the device implementation (cgmres_hip_model_probe).
// An unknown Model aborts with a message — there is no CPU fallback.
//
// Behavioural notes: dUdt(0) = 0 (the reference reads uninitialised memory on the first tick);
// errors from the library are fatal (the reference's methods return void and cannot report).
#pragma once
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "cgmres_hip.h"
#include "gmres.hpp"
#include "matrix.hpp"

namespace cgmres_detail {

inline void check(int rc, const char* what) {
  if (rc != 0) {
    fprintf(stderr, "cgmres_hip: %s failed (%d): %s\n", what, rc, cgmres_hip_last_error());
    exit(-1);
  }
}

// Does registry entry `id` compute the user's Model?  Same dimensions and the same dxdt/dPhidx/dHdx/dHdu at two
// probe points (host class vs device build, 1e-12 relative).
template <class Model>
inline bool model_matches(int32_t id, int32_t device) {
  constexpr int nx = Model::dim_x, nu = Model::dim_u, np = Model::dim_p;
  int32_t dims[5];
  double tuning[6];
  if (cgmres_hip_model_info(id, dims, tuning) != 0) return false;
  if (dims[0] != nx || dims[1] != nu || dims[2] != np) return false;
  for (int probe = 0; probe < 2; ++probe) {
    double x[nx], u[nu], p[np + 1], l[nx], host[3 * nx + nu], dev[3 * nx + nu];
    for (int i = 0; i < nx; ++i) x[i] = 0.3 + 0.37 * i + 0.21 * probe, l[i] = -0.4 + 0.29 * i - 0.13 * probe;
    for (int j = 0; j < nu; ++j) u[j] = 0.11 + 0.23 * j + 0.17 * probe;
    for (int j = 0; j < np + 1; ++j) p[j] = 0.7 - 0.45 * j + 0.05 * probe;
    Model::dxdt(&host[0], x, u, p);
    Model::dPhidx(&host[nx], x, p);
    Model::dHdx(&host[2 * nx], x, u, p, l);
    Model::dHdu(&host[3 * nx], x, u, p, l);
    if (cgmres_hip_model_probe(id, device, x, u, p, l, dev) != 0) return false;
    for (int k = 0; k < 3 * nx + nu; ++k) {
      const double scale = fabs(host[k]) > 1.0 ? fabs(host[k]) : 1.0;
      if (!(fabs(host[k] - dev[k]) <= 1e-12 * scale)) return false;
    }
  }
  return true;
}

// Registry lookup by numeric fingerprint of the user's Model (see file header): the built-in models first, then the
// plugins named in the environment variable CGMRES_HIP_MODEL_PLUGINS (':'-separated shared objects generated from
// the user's own model.hpp by `python -m cgmres_cpp_amd.plugin`, registered through cgmres_hip_register_model).
template <class Model>
inline int32_t identify_model(int32_t device) {
  for (int32_t id = 0; id < CGMRES_HIP_MODEL_COUNT; ++id)
    if (model_matches<Model>(id, device)) return id;
  if (const char* env = getenv("CGMRES_HIP_MODEL_PLUGINS")) {
    const char* a = env;
    while (*a) {
      const char* e = strchr(a, ':');
      const size_t n = e ? size_t(e - a) : strlen(a);
      char path[4096];
      if (n > 0 && n < sizeof path) {
        memcpy(path, a, n);
        path[n] = 0;
        int32_t id = -1;
        if (cgmres_hip_register_model(path, &id) == 0 && model_matches<Model>(id, device)) return id;
      }
      a += n + (e ? 1 : 0);
    }
  }
  return -1;
}

template <class Model>
inline cgmres_hip_config config_for(int32_t batch, int32_t device) {
  const int32_t id = identify_model<Model>(device);
  if (id < 0) {
    fprintf(stderr,
            "Cgmres<Model>: this Model (dim_x=%d dim_u=%d dim_p=%d) is not in the libcgmres_hip registry, or no "
            "MI355X is usable: %s\n",
            int(Model::dim_x), int(Model::dim_u), int(Model::dim_p), cgmres_hip_last_error());
    exit(-1);
  }
  cgmres_hip_config cfg;
  check(cgmres_hip_default_config(id, &cfg), "default_config");
  cfg.batch = batch;
  cfg.device = device;
  cfg.dv = Model::dv;  // run-time counterparts of the Model's static constexpr block
  cfg.k_max = Model::k_max;
  cfg.tol = Model::tol;
  cfg.dt = Model::dt;
  cfg.h = Model::h;
  cfg.zeta = Model::zeta;
  cfg.Tf = Model::Tf;
  cfg.alpha = Model::alpha;
  return cfg;
}

}  // namespace cgmres_detail

template <class Model>
class Cgmres : public Gmres {
 public:
  Cgmres(void) : Gmres(dim_u * dv, Model::k_max, Model::tol), handle(nullptr) {
    const cgmres_hip_config cfg = cgmres_detail::config_for<Model>(1, 0);
    cgmres_detail::check(cgmres_hip_create(&cfg, &handle), "create");
  }
  ~Cgmres(void) {
    if (handle) cgmres_hip_destroy(handle);
  }

  double get_dtau(const double t) const { return Tf * (1 - exp(-alpha * t)) / (double)dv; }

  // ptau = [ p(t), p(t + dtau), ..., p(t + dv * dtau) ]
  void set_ptau(const double* ptau_buf) {
    if (dim_p) cgmres_detail::check(cgmres_hip_set_ptau(handle, ptau_buf, 0), "set_ptau");
  }
  // ptau = [ p(t), p(t), ..., p(t) ]
  void set_ptau_repeat(const double* p_buf) {
    if (dim_p) cgmres_detail::check(cgmres_hip_set_ptau_repeat(handle, p_buf, 0), "set_ptau_repeat");
  }
  // U(i) = u0 for every stage
  void init_u0(const double* u0) { cgmres_detail::check(cgmres_hip_init_u0(handle, u0, 0), "init_u0"); }
  // Newton iterations on dH/du(x0, u0, p0, dPhi/dx(x0)) = 0; u0 is updated in place, then replicated
  void init_u0_newton(double* u0, const double* x0, const double* p0, const uint16_t n_loop) {
    cgmres_detail::check(cgmres_hip_init_u0_newton(handle, u0, x0, p0, n_loop), "init_u0_newton");
  }
  // one control tick: u = U[0:dim_u] after the FDGMRES update
  void control(double* u, const double* x) { cgmres_detail::check(cgmres_hip_control(handle, u, x), "control"); }

  // The C handle, for callers that want the batched / device-pointer entry points of cgmres_hip.h.
  cgmres_hip_handle native_handle() const { return handle; }

 protected:
  // Forward-difference Jacobian-vector product (reference cgmres.hpp:164-175), evaluated on the GPU with the
  // state left by the last control()/prepare.  Kept because it is Gmres' pure virtual.
  void Ax_func(double* Ax, const double* v) override {
    cgmres_detail::check(cgmres_hip_Ax_func(handle, Ax, v), "Ax_func");
  }

 public:
  static constexpr uint16_t dim_x = Model::dim_x;
  static constexpr uint16_t dim_u = Model::dim_u;
  static constexpr uint16_t dim_p = Model::dim_p;

  static constexpr double dt = Model::dt;
  static constexpr double h = Model::h;
  static constexpr double zeta = Model::zeta;
  static constexpr uint16_t dv = Model::dv;
  static constexpr double Tf = Model::Tf;
  static constexpr double alpha = Model::alpha;

 private:
  cgmres_hip_handle handle;
  Cgmres(const Cgmres&);
  Cgmres& operator=(const Cgmres&);
};
