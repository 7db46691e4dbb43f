// ORACLE — TEST INFRASTRUCTURE ONLY. extern "C" entry points of oracle/_ref/libref.so:
// dispatches orc_create to the fp64 / fp32 builds of oracle/ref_harness.cpp.
#define ORC_DEFINE_CAPI
#include "orc_base.hpp"

OrcBase* ref_make_f64(int model, int dv, int km, double tol);
OrcBase* ref_make_f32(int model, int dv, int km, double tol);

OrcBase* orc_factory(int model, int dv, int kmax, double tol, int dtype) {
  return dtype == 1 ? ref_make_f32(model, dv, kmax, tol) : ref_make_f64(model, dv, kmax, tol);
}
