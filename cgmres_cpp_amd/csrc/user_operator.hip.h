// Plugin path for a stand-alone Gmres with a CALLER-SUPPLIED operator — the one member of the reference's class
// surface that has no Cgmres around it: `class Gmres` (include/gmres.hpp:8-129) is an abstract solver whose subclass
// provides `Ax_func(double* Ax, const double* x)` (:26) and calls the protected `gmres(x, b)` (:28-112).
// A host callback cannot run inside a device solve, so the operator is compiled FOR the device: a header with
//     struct MyOp {
//       static constexpr int len = ...;        // length of the vectors (the `len` of Gmres(len, k_max, tol), :10)
//       static constexpr int n_params = ...;   // scalars the operator reads per instance (0 allowed)
//       static void Ax(double* Ax, const double* x, const double* params);   // contiguous vectors, like Ax_func
//     };
// is included between `#pragma clang force_cuda_host_device begin/end` (the same function is what the host subclass's
// Ax_func calls) and CGMRES_HIP_DEFINE_OPERATOR(MyOp) turns it into a shared object that
// cgmres_hip_register_operator() loads.  cgmres_hip_gmres_user() then solves `batch` independent systems, one per lane,
// in the reference's statement order (gmres_lane_core: sequential dots, modified Gram-Schmidt in order, 2-vector
// Householder QR, every exit path of gmres.hpp:39-41 / :63-65 / :93-95).  fp64, like the reference.
#pragma once
#include <vector>

#include "ctx_common.hip.h"
#include "tick_lane.hip.h"

namespace cgm {

template <class Op>
__global__ __launch_bounds__(64) void gmres_op_kernel(int B, int ldb, int kmax, double tol, const double* __restrict__ params,
                                                      double* __restrict__ x, const double* __restrict__ bv,
                                                      double* __restrict__ V, double* __restrict__ H,
                                                      double* __restrict__ rho, double* __restrict__ g,
                                                      int* __restrict__ n_ax, int* __restrict__ reason) {
  constexpr int L = Op::len, NP = Op::n_params;
  const int b = blockIdx.x * 64 + threadIdx.x;
  if (b >= B) return;
  const size_t ld = ldb;
  double p[NP > 0 ? NP : 1];
#pragma unroll
  for (int j = 0; j < NP; ++j) p[j] = params[size_t(b) * NP + j];
  int n = 0;
  const int why = gmres_lane_core<double>(
      L, kmax, tol, ld, V + b, H + b, rho + b, g + b, x + b, bv + b, &n, [&](double* out, const double* v) {
        double vin[L], vout[L];  // the user's operator works on contiguous vectors, like the reference's Ax_func
        for (int e = 0; e < L; ++e) vin[e] = v[size_t(e) * ld];
        Op::Ax(vout, vin, p);
        for (int e = 0; e < L; ++e) out[size_t(e) * ld] = vout[e];
      });
  n_ax[b] = n;
  reason[b] = why;
}

// host side of one solve: instance-major host arrays in, element-major device arrays inside
template <class Op>
int gmres_op_solve(int device, int batch, int kmax, double tol, const double* params, double* x, const double* bvec,
                   int32_t* n_ax, int32_t* reason) {
  constexpr int L = Op::len, NP = Op::n_params;
  if (batch < 1 || kmax < 1 || !(tol >= 0) || !x || !bvec || (NP && !params))
    return fail(CGMRES_HIP_EINVAL, "gmres_user: bad argument");
  if (long(L) * (kmax + 1) >= 65536) return fail(CGMRES_HIP_EINVAL, "gmres_user: len*(k_max+1) beyond the reference's 16-bit index range");
  HIP_TRY(hipSetDevice(device));
  const int ldb = (batch + 63) / 64 * 64, k1 = kmax + 1;
  const size_t nV = size_t(L) * k1 * ldb, nH = size_t(k1) * k1 * ldb, nr = size_t(k1) * ldb, ng = size_t(3) * kmax * ldb;
  const size_t nx = size_t(L) * ldb, total = nV + nH + nr + ng + 2 * nx + size_t(NP ? NP : 1) * batch;
  double* d = nullptr;
  int* di = nullptr;
  HIP_TRY(hipMalloc(reinterpret_cast<void**>(&d), total * sizeof(double)));
  if (hipMalloc(reinterpret_cast<void**>(&di), size_t(2) * ldb * sizeof(int)) != hipSuccess) {
    (void)hipFree(d);
    return fail(CGMRES_HIP_ENOMEM, "gmres_user: out of device memory");
  }
  double *dV = d, *dH = dV + nV, *dr = dH + nH, *dg = dr + nr, *dx = dg + ng, *db = dx + nx, *dp = db + nx;
  std::vector<double> hx(nx, 0.0), hb(nx, 0.0);
  for (int i = 0; i < batch; ++i)
    for (int e = 0; e < L; ++e) hx[size_t(e) * ldb + i] = x[size_t(i) * L + e], hb[size_t(e) * ldb + i] = bvec[size_t(i) * L + e];
  hipError_t e = hipMemset(d, 0, (nV + nH + nr + ng) * sizeof(double));
  if (e == hipSuccess) e = hipMemcpy(dx, hx.data(), nx * sizeof(double), hipMemcpyHostToDevice);
  if (e == hipSuccess) e = hipMemcpy(db, hb.data(), nx * sizeof(double), hipMemcpyHostToDevice);
  if (e == hipSuccess && NP) e = hipMemcpy(dp, params, size_t(NP) * batch * sizeof(double), hipMemcpyHostToDevice);
  if (e == hipSuccess) {
    gmres_op_kernel<Op><<<ldb / 64, 64>>>(batch, ldb, kmax, tol, dp, dx, db, dV, dH, dr, dg, di, di + ldb);
    e = hipGetLastError();
  }
  if (e == hipSuccess) e = hipDeviceSynchronize();
  if (e == hipSuccess) e = hipMemcpy(hx.data(), dx, nx * sizeof(double), hipMemcpyDeviceToHost);
  std::vector<int> hi(size_t(2) * ldb);
  if (e == hipSuccess) e = hipMemcpy(hi.data(), di, hi.size() * sizeof(int), hipMemcpyDeviceToHost);
  (void)hipFree(d);
  (void)hipFree(di);
  if (e != hipSuccess) return fail(CGMRES_HIP_ERUNTIME, "gmres_user: %s", hipGetErrorString(e));
  for (int i = 0; i < batch; ++i) {
    for (int el = 0; el < L; ++el) x[size_t(i) * L + el] = hx[size_t(el) * ldb + i];
    if (n_ax) n_ax[i] = hi[i];
    if (reason) reason[i] = hi[ldb + i];
  }
  return 0;
}

}  // namespace cgm

// The entry points of an operator plugin (bound by cgmres_hip_register_operator in capi.hip).
#define CGMRES_HIP_DEFINE_OPERATOR(OP)                                                                             \
  extern "C" {                                                                                                     \
  int32_t cgmres_hip_opplugin_abi(void) { return CGMRES_HIP_ABI_VERSION; }                                         \
  void cgmres_hip_opplugin_info(int32_t dims[2]) { dims[0] = OP::len, dims[1] = OP::n_params; }                    \
  int cgmres_hip_opplugin_solve(int32_t device, int32_t batch, int32_t k_max, double tol, const double* params,    \
                                double* x, const double* b, int32_t* n_ax, int32_t* reason) {                      \
    return cgm::gmres_op_solve<OP>(device, batch, k_max, tol, params, x, b, n_ax, reason);                         \
  }                                                                                                                \
  const char* cgmres_hip_opplugin_last_error(void) { return cgm::g_err.c_str(); }                                  \
  }
