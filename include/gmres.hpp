// gmres.hpp — class surface of the reference's include/gmres.hpp (class Gmres, :8-129) for the drop-in façade.
//
// In the reference Gmres is the CPU solver: protected constructor (len, k_max, tol) (:10), pure virtual
// Ax_func (:26), protected gmres(x, b) (:28-112).  Its only user is Cgmres<Model>, which calls gmres() from
// control() (cgmres.hpp:99).  Here the whole solve — Ax_func included — runs inside libcgmres_hip.so on the GPU
// (cgmres_hip_control), so this class only keeps the type hierarchy and the protected names alive for code that
// mentions them.  There is deliberately NO host implementation behind gmres(): a class that derives from Gmres with
// a host-side Ax_func of its own is outside the accelerated path (a host callback cannot run inside the device solve)
// and is told so AT COMPILE TIME — calling Gmres::gmres is a hard error with the message below — instead of being
// served by a silent CPU fallback or aborting at run time.  The solver itself, with the controller's forward-difference
// operator, is reachable as cgmres_hip_gmres (include/cgmres_hip.h).
#pragma once
#include <float.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#include "matrix.hpp"

class Gmres {
 protected:
  Gmres(const uint16_t len, const uint16_t k_max, const double tol) : len(len), k_max(k_max), tol(tol) {}
  virtual ~Gmres() {}

  // A*x of the linear system; Cgmres<Model> overrides it with the forward-difference product (cgmres.hpp:164-175)
  virtual void Ax_func(double* Ax, const double* x) = 0;

  // gmres.hpp:28-112.  Cgmres<Model>::control never calls this: the batched solve is cgmres_hip_control().
#define CGMRES_HIP_NO_HOST_GMRES                                                                              \
  "Gmres::gmres: the stand-alone host solver is not part of the MI355X path (a host Ax_func cannot run in the " \
  "device solve and there is no CPU fallback); use Cgmres<Model>::control or cgmres_hip_gmres"
#if defined(__clang__)
  void gmres(double* x, const double* b_vec) __attribute__((unavailable(CGMRES_HIP_NO_HOST_GMRES)));
#elif defined(__GNUC__)
  void gmres(double* x, const double* b_vec) __attribute__((error(CGMRES_HIP_NO_HOST_GMRES)));
#else
  void gmres(double* x, const double* b_vec) = delete;
#endif

  const uint16_t len;
  const uint16_t k_max;
  const double tol;

 private:
  Gmres(const Gmres&);
  Gmres& operator=(const Gmres&);
};
