// Plugin path for USER models: an arbitrary reference-style `Model` class
// (static constexpr dim_x/dim_u/dim_p/dv/k_max/dt/h/zeta/Tf/alpha/tol + static dxdt/dPhidx/dHdx/dHdu/ddHduu with
// the signatures of <example>/model.hpp:36-76) is compiled for the device by including its header between
//     #pragma clang force_cuda_host_device begin / end
// and wrapped by UserDev<Model> into the interface BOTH kernel mappings consume:
//   * "lane" (tick_lane.hip.h: one lane per instance, reference statement order) needs nothing but the model functions;
//   * "wg" (tick_wg.hip.h) needs the backward stage split into a costate-free part and a part that is affine in the
//     costate (stage_coeffs / costate_step).  For ANY Hamiltonian H = L + lambda^T f both dH/dx and dH/du are affine in
//     lambda, so the split is generated here by PROBING the user's own functions: dHdx(lambda = 0) = q,
//     (dHdx(s e_c) - q)/s = column c of J^T, and the same for dH/du (phi, B^T) — NX + 1 evaluations of each per (stage,
//     instance), done by the stage-parallel coefficient phase; what stays serial is the dense affine recurrence
//     lambda <- (lambda + dtau q) + (dtau J^T) lambda and dF = B^T lambda.  Same mathematics as the reference's loop
//     (cgmres.hpp:145-161), different association of the sums.
//     The probe costate is s e_c with s = the power of two at or above max(1, |q|_inf, |phi|_inf): the difference
//     g(s e_c) - q then loses eps*max(|q|, s|J|)/s = eps*max(1, |J|)-ish per entry instead of eps*|q| — with stiff cost
//     weights (|q| ~ 1e6) the plain unit probe would put an ABSOLUTE error of 1e-10 on every Jacobian entry, which the
//     recurrence multiplies by a costate of the size of q.  Scaling by a power of two is exact.
//     Whether dHdx / dHdu really ARE affine in the costate is checked once per process on the device
//     (user_affinity_kernel: g(2 e_c) - g(0) against 2 (g(e_c) - g(0)) at two probe points); a model that fails runs on
//     the lane mapping, which calls the user's functions as they are.
// cgmres_cpp_amd/plugin.py generates the translation unit; cgmres_hip_register_model() loads the resulting shared
// object into the registry of libcgmres_hip.so.  fp64 only (the reference Model concept is double).
#pragma once
#include "factory_impl.hip.h"

namespace cgm {

template <class Model>
struct UserDev {
  static constexpr int NX = Model::dim_x, NU = Model::dim_u, NP = Model::dim_p, NC = 0, NU_DYN = Model::dim_u;
  static constexpr bool DXDT_USES_P = true;
  using Math = NoConsts;
  template <bool>
  using MathFor = NoConsts;
  // ---- what the wg mapping needs (tick_wg.hip.h) ----
  static constexpr bool HAS_QUAD_SWEEP = false;
  static constexpr int NSLOT = NX, TRIG_SLOT0 = NX, TAB_PAD = 0;
  static constexpr int NUL = NU;                               // every component of dH/du may depend on the costate
  // coefficients of a stage: dtau*J^T (row-major) | B^T (row-major) | [pad] | dtau*q | [pad] — the entries that multiply
  // lambda first (NBW_LIN of them, an even count: they are fetched in pairs), the bias last (PendulumDev::NBW_LIN)
  static constexpr int NLIN_RAW = NX * NX + NU * NX;
  static constexpr int NBW_LIN = NLIN_RAW + (NLIN_RAW & 1);
  static constexpr int NBW_RAW = NBW_LIN + NX;
  static constexpr int NBW = NBW_RAW + (NBW_RAW & 1);
  static constexpr bool COSTATE_HOM = true;  // costate_step<true> exists: chunk-parallel costate sweep where it fits
  static __device__ __forceinline__ void stage_coeffs(double* bw, double* phi, const double* x, const double* u,
                                                      const double* p, const double*, double dtau) {
    double l[NX], q[NX], g[NX], hu[NU];
#pragma unroll
    for (int c = 0; c < NX; ++c) l[c] = 0.0;
    Model::dHdx(q, x, u, p, l);    // costate-free parts
    Model::dHdu(phi, x, u, p, l);
    double big = 1.0;
#pragma unroll
    for (int r = 0; r < NX; ++r) {
      bw[NBW_LIN + r] = dtau * q[r];
      big = __builtin_fmax(big, __builtin_fabs(q[r]));
    }
#pragma unroll
    for (int j = 0; j < NU; ++j) big = __builtin_fmax(big, __builtin_fabs(phi[j]));
    // s = 2^e >= big (a NaN/Inf `big` gives some finite s: the NaN then comes through q itself)
    int e = 0;
    (void)__builtin_frexp(big < 1e300 ? big : 1.0, &e);
    const double sc = __builtin_ldexp(1.0, e), inv_sc = __builtin_ldexp(1.0, -e);
#pragma unroll
    for (int c = 0; c < NX; ++c) {  // scaled unit costates: column c of J^T and of B^T
      l[c] = sc;
      Model::dHdx(g, x, u, p, l);
      Model::dHdu(hu, x, u, p, l);
      l[c] = 0.0;
#pragma unroll
      for (int r = 0; r < NX; ++r) bw[r * NX + c] = dtau * ((g[r] - q[r]) * inv_sc);
#pragma unroll
      for (int j = 0; j < NU; ++j) bw[NX * NX + j * NX + c] = (hu[j] - phi[j]) * inv_sc;
    }
    if (NBW_LIN != NLIN_RAW) bw[NBW_LIN - 1] = 0.0;
    if (NBW != NBW_RAW) bw[NBW - 1] = 0.0;
  }
  // l <- l + dtau*dHdx(l) and dF = B^T l_old, from the stored coefficients; HOM: without the bias dtau*q (one column of
  // a chunk's transfer matrix, WgCtx::sweep_costate_par)
  template <bool HOM = false>
  static __device__ __forceinline__ void costate_step(double* l, double* dF, const double* bw, double) {
    double n[NX];
#pragma unroll
    for (int j = 0; j < NU; ++j) {
      double a = 0.0;
#pragma unroll
      for (int c = 0; c < NX; ++c) a += bw[NX * NX + j * NX + c] * l[c];
      dF[j] = a;
    }
#pragma unroll
    for (int r = 0; r < NX; ++r) {
      double a = HOM ? l[r] : l[r] + bw[NBW_LIN + r];
#pragma unroll
      for (int c = 0; c < NX; ++c) a += bw[r * NX + c] * l[c];
      n[r] = a;
    }
#pragma unroll
    for (int r = 0; r < NX; ++r) l[r] = n[r];
  }
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

// max over two probe points and all unit costates of |g(2 e_c) - g(0) - 2 (g(e_c) - g(0))| / max(1, |g|) for g = dHdx
// and dHdu: zero (up to rounding) when both are affine in the costate, as they are for every H = L + lambda^T f
template <class Model>
__global__ void user_affinity_kernel(double* out) {
  constexpr int NX = Model::dim_x, NU = Model::dim_u, NP = Model::dim_p;
  double worst = 0.0;
  for (int pt = 0; pt < 2; ++pt) {
    double x[NX], u[NU], p[NP > 0 ? NP : 1], l[NX], g0[NX], g1[NX], g2[NX], h0[NU], h1[NU], h2[NU];
    // fixed, irrational-looking probe values of moderate size (no structure a model could be singular on by design)
    for (int i = 0; i < NX; ++i) x[i] = 0.37 + 0.211 * i - 0.53 * pt, l[i] = 0.0;
    for (int j = 0; j < NU; ++j) u[j] = 0.29 - 0.173 * j + 0.41 * pt;
    for (int j = 0; j < NP; ++j) p[j] = 0.13 + 0.07 * j;
    Model::dHdx(g0, x, u, p, l);
    Model::dHdu(h0, x, u, p, l);
    for (int c = 0; c < NX; ++c) {
      l[c] = 1.0;
      Model::dHdx(g1, x, u, p, l);
      Model::dHdu(h1, x, u, p, l);
      l[c] = 2.0;
      Model::dHdx(g2, x, u, p, l);
      Model::dHdu(h2, x, u, p, l);
      l[c] = 0.0;
      for (int r = 0; r < NX; ++r) {
        const double v = fabs((g2[r] - g0[r]) - 2.0 * (g1[r] - g0[r])) / fmax(1.0, fabs(g2[r]));
        worst = (v > worst || v != v) ? v : worst;  // (a NaN sticks)
      }
      for (int j = 0; j < NU; ++j) {
        const double v = fabs((h2[j] - h0[j]) - 2.0 * (h1[j] - h0[j])) / fmax(1.0, fabs(h2[j]));
        worst = (v > worst || v != v) ? v : worst;
      }
    }
  }
  *out = worst;
}

// host: true when the device build of the model passed user_affinity_kernel (evaluated once per process)
template <class Model>
inline bool user_model_is_affine_in_costate(int device) {
  static int cached = -1;
  if (cached >= 0) return cached != 0;
  double* d = nullptr;
  double v = -1.0;
  if (hipSetDevice(device) != hipSuccess || hipMalloc(reinterpret_cast<void**>(&d), sizeof(double)) != hipSuccess) return false;
  user_affinity_kernel<Model><<<1, 1>>>(d);
  const bool ok = hipGetLastError() == hipSuccess && hipDeviceSynchronize() == hipSuccess &&
                  hipMemcpy(&v, d, sizeof(double), hipMemcpyDeviceToHost) == hipSuccess;
  (void)hipFree(d);
  cached = (ok && v >= 0.0 && v <= 1e-10) ? 1 : 0;
  return cached != 0;
}

}  // namespace cgm

#ifdef CGM_DEBUG_LDS
// diagnostic plugin builds only: the out-of-allocation flag of the costate sweep (0 = every address was inside), cleared
extern "C" int cgmres_hip_plugin_debug_lds_oob(void) {
  int v = -1, zero = 0;
  if (hipDeviceSynchronize() != hipSuccess || hipMemcpyFromSymbol(&v, HIP_SYMBOL(cgm::g_cgm_lds_oob), sizeof v) != hipSuccess) return -1;
  (void)hipMemcpyToSymbol(HIP_SYMBOL(cgm::g_cgm_lds_oob), &zero, sizeof zero);
  return v;
}
#endif

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
  /* a controller batch for this model on the requested / default mapping (wg when the sizes fit its LDS plan, */ \
  /* lane otherwise), not yet initialised; nullptr for anything but fp64 or an unsupported explicit mapping      */ \
  cgmres_hip_ctx* cgmres_hip_plugin_make(const cgmres_hip_config* cfg) {                                         \
    if (cfg->dtype != CGMRES_HIP_F64) return nullptr;                                                            \
    int resolved = 0;                                                                                            \
    cgmres_hip_config c = *cfg;                                                                                  \
    /* the wg mapping's generated split needs dHdx / dHdu affine in the costate: checked on the device */        \
    if (c.variant != 1 && !cgm::user_model_is_affine_in_costate<MODEL>(c.device)) {                              \
      if (c.variant != 0) {                                                                                      \
        cgm::fail(CGMRES_HIP_EINVAL, "user model: dHdx/dHdu are not affine in the costate: lane mapping only");  \
        return nullptr;                                                                                          \
      }                                                                                                          \
      c.variant = 1;                                                                                             \
    }                                                                                                            \
    return cgm::make_variant<cgm::UserDev<MODEL>, double>(c, &resolved);                                         \
  }                                                                                                              \
  /* [dxdt | dPhidx | dHdx | dHdu] of the DEVICE build at one point; device pointers, one thread */              \
  int cgmres_hip_plugin_probe(const double* x, const double* u, const double* p, const double* l, double* out,   \
                              void* stream) {                                                                    \
    cgm::user_probe_kernel<MODEL><<<1, 1, 0, static_cast<hipStream_t>(stream)>>>(x, u, p, l, out);               \
    return hipGetLastError() == hipSuccess ? 0 : -1;                                                             \
  }                                                                                                              \
  const char* cgmres_hip_plugin_last_error(void) { return cgm::g_err.c_str(); }                                  \
  }
