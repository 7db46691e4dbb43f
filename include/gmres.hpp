// gmres.hpp — drop-in façade for the reference's `class Gmres` (reference include/gmres.hpp:8-129): the abstract
// matrix-free GMRES(k_max) whose subclass supplies `Ax_func` (:26) and calls the protected `gmres(x, b)` (:28-112).
//
// In the reference its one user is Cgmres<Model>, which calls gmres() from control() (cgmres.hpp:99); here that whole
// tick — Ax_func included — runs inside libcgmres_hip.so (cgmres_hip_control), and Cgmres<Model> (cgmres.hpp of this
// directory) never calls gmres().  A class that derives from Gmres DIRECTLY, with an operator of its own, is served as
// well — on the GPU, like everything else here: a host callback cannot run inside a device solve, so the subclass names
// the device build of its operator,
//     use_device_operator("libcgmres_op_<name>.so", params);   // in its constructor
// a shared object generated from the header that holds the operator (`struct Op { static constexpr int len, n_params;
// static void Ax(double* Ax, const double* x, const double* params); }` — the same function its host Ax_func calls) by
// `python -m cgmres_cpp_amd.plugin --operator <header> --cls Op`; gmres(x, b) then is cgmres_hip_gmres_user for a batch of
// one.  There is NO host implementation behind gmres(): without a registered operator the call ends the program with a
// message (the reference's own failure convention, DEBUG_MODE's exit(-1)) — never a silent CPU fallback.
// Thousands of systems at once: cgmres_hip_gmres_user directly (include/cgmres_hip.h).
#pragma once
#include <float.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#include "cgmres_hip.h"
#include "matrix.hpp"

class Gmres {
 protected:
  Gmres(const uint16_t len, const uint16_t k_max, const double tol) : len(len), k_max(k_max), tol(tol) {}
  virtual ~Gmres() {}

  // A*x of the linear system (gmres.hpp:26); Cgmres<Model> overrides it with the forward-difference product
  // (cgmres.hpp:164-175).  Kept pure virtual: the type hierarchy of the reference compiles unchanged.
  virtual void Ax_func(double* Ax, const double* x) = 0;

  // No counterpart in the reference: names the device build of this object's operator (see the header comment) and the
  // per-instance scalars it reads (copied).  plugin_path: what plugin.build_operator returned.
  void use_device_operator(const char* plugin_path, const double* params = nullptr, int32_t device = 0) {
    int32_t dims[2] = {0, 0};
    if (cgmres_hip_register_operator(plugin_path, &op_id_) != 0 || cgmres_hip_operator_info(op_id_, dims) != 0) {
      fprintf(stderr, "Gmres::use_device_operator: %s\n", cgmres_hip_last_error());
      exit(-1);
    }
    if (dims[0] != len || dims[1] > kMaxParams || (dims[1] > 0 && !params)) {
      fprintf(stderr, "Gmres::use_device_operator: operator has len %d / %d params, this solver len %d\n", dims[0], dims[1], len);
      exit(-1);
    }
    for (int32_t j = 0; j < dims[1]; ++j) params_[j] = params[j];
    device_ = device;
  }

  // gmres.hpp:28-112: x in/out (warm start), b the right-hand side; every exit path of the reference (residual below tol
  // :39-41, breakdown :63-65 with its "Breakdown" line, convergence :93-95).
  void gmres(double* x, const double* b_vec) {
    if (op_id_ < 0) {
      fprintf(stderr,
              "Gmres::gmres: no device operator registered (use_device_operator): a host Ax_func cannot run inside the "
              "device solve and this library has no CPU fallback\n");
      exit(-1);
    }
    int32_t n_ax = 0, reason = 0;
    if (cgmres_hip_gmres_user(op_id_, device_, 1, k_max, tol, params_, x, b_vec, &n_ax, &reason) != 0) {
      fprintf(stderr, "Gmres::gmres: %s\n", cgmres_hip_last_error());
      exit(-1);
    }
    if (reason == CGMRES_HIP_EXIT_BREAKDOWN) printf("Breakdown\n");  // gmres.hpp:64
    last_n_ax_ = n_ax, last_reason_ = reason;
  }

  const uint16_t len;
  const uint16_t k_max;
  const double tol;
  int32_t last_n_ax_ = 0, last_reason_ = 0;  // Arnoldi products / CGMRES_HIP_EXIT_* of the last gmres() (the reference drops them)

 private:
  static constexpr int32_t kMaxParams = 16;
  int32_t op_id_ = -1, device_ = 0;
  double params_[kMaxParams] = {0};
  Gmres(const Gmres&);
  Gmres& operator=(const Gmres&);
};
