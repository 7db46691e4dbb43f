// Plugin path for USER models: an arbitrary reference-style `Model` class
// (static constexpr dim_x/dim_u/dim_p/dv/k_max/dt/h/zeta/Tf/alpha/tol + static dxdt/dPhidx/dHdx/dHdu/ddHduu with
// the signatures of <example>/model.hpp:36-76) is compiled for the device by including its header between
//     #pragma clang force_cuda_host_device begin / end
// and wrapped by UserDev<Model> into the interface the "lane" mapping consumes (tick_lane.hip.h: one lane per
// instance, reference statement order — the mapping that needs nothing but the four model functions).  The "wg"
// mapping needs the affine costate split of a model (stage_coeffs / costate_step) and is therefore reserved to the
// built-in models.  cgmres_cpp_amd/plugin.py generates the translation unit; cgmres_hip_register_model() loads the
// resulting shared object into the registry of libcgmres_hip.so.  fp64 only (the reference Model concept is double).
#pragma once
#include "ctx_lane.hip.h"

namespace cgm {

template <class Model>
struct UserDev {
  static constexpr int NX = Model::dim_x, NU = Model::dim_u, NP = Model::dim_p, NC = 0, NU_DYN = Model::dim_u;
  static constexpr bool DXDT_USES_P = true;
  using Math = NoConsts;
  static constexpr ModelInfo info() {
    return {NX, NU, NP, Model::dv, Model::k_max, Model::dt, Model::h, Model::zeta, Model::Tf, Model::alpha, Model::tol};
  }
  // the 5-argument form is what the built-in models expose (it also returns the trig values of x); here it is a
  // no-op kept for the call sites that only want those (Newton initialisation): NC = 0
  static __device__ __forceinline__ void dxdt(double*, const double*, const double*, double*, const Math&) {}
  static __device__ __forceinline__ void dxdt_p(double* f, const double* x, const double* u, const double* p) {
    Model::dxdt(f, x, u, p);
  }
  static __device__ __forceinline__ void dPhidx(double* g, const double* x, const double* p) { Model::dPhidx(g, x, p); }
  static __device__ __forceinline__ void dHdx(double* g, const double* x, const double* u, const double* p,
                                              const double* l, const double*) {
    Model::dHdx(g, x, u, p, l);
  }
  static __device__ __forceinline__ void dHdu(double* g, const double* x, const double* u, const double* p,
                                              const double* l, const double*) {
    Model::dHdu(g, x, u, p, l);
  }
  static __device__ __forceinline__ void ddHduu(double* m, const double* x, const double* u, const double* p,
                                                const double* l) {
    Model::ddHduu(m, x, u, p, l);
  }
};

template <class Model>
__global__ void user_probe_kernel(const double* x, const double* u, const double* p, const double* l, double* out) {
  constexpr int NX = Model::dim_x, NU = Model::dim_u;
  double f[NX], g[NX], hx[NX], hu[NU];
  Model::dxdt(f, x, u, p);
  Model::dPhidx(g, x, p);
  Model::dHdx(hx, x, u, p, l);
  Model::dHdu(hu, x, u, p, l);
  for (int i = 0; i < NX; ++i) out[i] = f[i], out[NX + i] = g[i], out[2 * NX + i] = hx[i];
  for (int j = 0; j < NU; ++j) out[3 * NX + j] = hu[j];
}

}  // namespace cgm

// The four entry points of a model plugin (bound by cgmres_hip_register_model in capi.hip).
#define CGMRES_HIP_DEFINE_PLUGIN(MODEL)                                                                          \
  extern "C" {                                                                                                   \
  int32_t cgmres_hip_plugin_abi(void) { return CGMRES_HIP_ABI_VERSION; }                                         \
  void cgmres_hip_plugin_info(int32_t dims[5], double tuning[6]) {                                               \
    constexpr cgm::ModelInfo mi = cgm::UserDev<MODEL>::info();                                                   \
    dims[0] = mi.dim_x, dims[1] = mi.dim_u, dims[2] = mi.dim_p, dims[3] = mi.dv, dims[4] = mi.k_max;             \
    tuning[0] = mi.dt, tuning[1] = mi.h, tuning[2] = mi.zeta, tuning[3] = mi.Tf, tuning[4] = mi.alpha;           \
    tuning[5] = mi.tol;                                                                                          \
  }                                                                                                              \
  /* a CtxLane for this model, not yet initialised; nullptr for anything but fp64 */                             \
  cgmres_hip_ctx* cgmres_hip_plugin_make(const cgmres_hip_config* cfg) {                                         \
    if (cfg->dtype != CGMRES_HIP_F64) return nullptr;                                                            \
    return new cgm::CtxLane<cgm::UserDev<MODEL>, double>();                                                      \
  }                                                                                                              \
  /* [dxdt | dPhidx | dHdx | dHdu] of the DEVICE build at one point; device pointers, one thread */              \
  int cgmres_hip_plugin_probe(const double* x, const double* u, const double* p, const double* l, double* out,   \
                              void* stream) {                                                                    \
    cgm::user_probe_kernel<MODEL><<<1, 1, 0, static_cast<hipStream_t>(stream)>>>(x, u, p, l, out);               \
    return hipGetLastError() == hipSuccess ? 0 : -1;                                                             \
  }                                                                                                              \
  const char* cgmres_hip_plugin_last_error(void) { return cgm::g_err.c_str(); }                                  \
  }
