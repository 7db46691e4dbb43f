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
// cgmres_hip_register_operator() loads.  cgmres_hip_gmres_user() then solves `batch` independent systems: one WAVEFRONT
// per system when a solve fits one wave's LDS (gmres_wave_kernel: vectors over the lanes, wave-wide sums), one lane per
// system otherwise (gmres_op_kernel, the reference's statement order: gmres_lane_core) — modified Gram-Schmidt in order,
// 2-vector Householder QR, every exit path of gmres.hpp:39-41 / :63-65 / :93-95 either way.  fp64, like the reference.
#pragma once
#include <mutex>
#include <vector>

#include "ctx_common.hip.h"
#include "tick_lane.hip.h"
#include "wave_scan.hip.h"

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

// ---- one WAVEFRONT per system -----------------------------------------------------------------------------------------
// The mapping above walks every vector of a solve on one lane (strided element-major accesses, private copies of the
// operator's operands): fine for the statement order, slow for anything but tiny systems.  Here a system owns a wave:
//   * element e of every vector lives on lane e mod 64 (registers for the work vector, an LDS row per basis vector), so
//     the Gram-Schmidt dots are (len/64) multiply-adds per lane + one wave-wide sum (wave_scan.hip.h) and the updates
//     are (len/64) multiply-adds — the solver of the wave mapping (tick_wave.hip.h) around ANY operator;
//   * the operator itself is the caller's serial function: lane 0 runs Op::Ax on two contiguous LDS rows (the basis
//     vector it reads IS its operand: no gather), the other lanes wait;
//   * Hessenberg / reflectors / residual vector: the small LDS arrays and the in-register column pass of the wg mapping.
// Everything of a solve stays in LDS (len (k_max + 3) + ~k_max^2/2 scalars): taken when that fits, the lane kernel serves
// the rest.  Sums associate as per-lane partial sums + a reduction instead of index-ascending (rounding level).
template <class T>
__device__ __forceinline__ T hess_column_lds(T* Hi, T* gi, T* rhoi, int k, T hn, bool writer) {  // gmres.hpp:71-90
  T* Hk = Hi + ((k * (k + 1)) >> 1);  // compact: column k = rows 0..k (h(k+1,k) arrives as `hn` and becomes 0)
  T a = Hk[0];
  for (int i = 0; i < k; ++i) {
    const T g0 = gi[3 * i], g1 = gi[3 * i + 1], g2 = gi[3 * i + 2], c = Hk[i + 1];
    const T beta = (g0 * a + g1 * c) * g2;
    if (writer) Hk[i] = a - beta * g0;
    a = c - beta * g1;
  }
  const T c = hn;
  const T sigma = -(a < T(0.0) ? T(-1.0) : T(1.0)) * sqrt_t<T>(a * a + c * c);
  const T g0 = a - sigma, g1 = c;
  const T g2 = T(2.0) / (g0 * g0 + g1 * g1);
  const T ek = rhoi[k];
  const T beta = g0 * ek * g2;
  const T en = -beta * g1;
  if (writer) {
    gi[3 * k] = g0, gi[3 * k + 1] = g1, gi[3 * k + 2] = g2;
    Hk[k] = sigma;
    rhoi[k] = ek - beta * g0;
    rhoi[k + 1] = en;
  }
  return en;
}

struct GmresWaveLds {
  static __host__ __device__ int pitch_H(int kmax) { return ((kmax * (kmax + 1)) / 2 + 2) & ~1; }
  static __host__ __device__ size_t count(int L, int kmax) {
    return size_t(kmax + 3) * L + pitch_H(kmax) + (kmax + 2) + 3 * kmax + 2;
  }
};

template <class Op>
__global__ __launch_bounds__(64) void gmres_wave_kernel(int B, int kmax, double tol, const double* __restrict__ params,
                                                        double* __restrict__ x, const double* __restrict__ bv,
                                                        int* __restrict__ n_ax_out, int* __restrict__ reason_out) {
  constexpr int L = Op::len, NP = Op::n_params, M = (L + 63) / 64;
  extern __shared__ __align__(16) double sm[];
  const int b = blockIdx.x, lane = threadIdx.x;
  if (b >= B) return;
  double* const vin = sm;                      // [L] operand row of the warm-start product
  double* const vout = vin + L;                // [L] result row of the operator
  double* const V = vout + L;                  // [kmax + 1][L]
  double* const Hi = V + size_t(kmax + 1) * L;  // compact Hessenberg
  double* const rhoi = Hi + GmresWaveLds::pitch_H(kmax);
  double* const gi = rhoi + (kmax + 2);
  auto fence = [] { __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); };
  double p[NP > 0 ? NP : 1];
#pragma unroll
  for (int j = 0; j < NP; ++j) p[j] = params[size_t(b) * NP + j];
  auto apply = [&](const double* src) {  // vout <- A src (gmres.hpp:26: the caller's Ax_func), on one lane
    fence();
    if (lane == 0) Op::Ax(vout, src, p);
    fence();
  };
  auto dot = [&](const double* a, const double* c) {
    double s = 0.0;
#pragma unroll
    for (int m = 0; m < M; ++m) s = __builtin_fma(a[m], c[m], s);
    return wave_sum(s);
  };
  auto row = [&](const double* r, double* reg) {  // LDS row -> this lane's elements (zero beyond len)
#pragma unroll
    for (int m = 0; m < M; ++m) reg[m] = lane + 64 * m < L ? r[lane + 64 * m] : 0.0;
  };
  auto put = [&](double* r, const double* reg) {
#pragma unroll
    for (int m = 0; m < M; ++m)
      if (lane + 64 * m < L) r[lane + 64 * m] = reg[m];
  };
  double xr[M], w[M], vi[M];
#pragma unroll
  for (int m = 0; m < M; ++m) {
    const int e = lane + 64 * m;
    xr[m] = e < L ? x[size_t(b) * L + e] : 0.0;
    w[m] = e < L ? bv[size_t(b) * L + e] : 0.0;
  }
  for (int q = lane; q < GmresWaveLds::pitch_H(kmax); q += 64) Hi[q] = 0.0;
  put(vin, xr);
  apply(vin);  // gmres.hpp:33
  row(vout, vi);
#pragma unroll
  for (int m = 0; m < M; ++m) w[m] = w[m] - vi[m];  // :34
  const double rho0 = ::sqrt(dot(w, w));            // :37
  if (lane == 0) rhoi[0] = rho0;
  int reason = 0, n_ax = 0, ksolve = 0;
  bool active = true;
  if (__any(!finite_t(rho0))) active = false, reason = 4;
  if (active && __any(rho0 < tol)) active = false, reason = 2;  // :39-41
  if (active) {
    const double inv = 1.0 / rho0;  // :44
#pragma unroll
    for (int m = 0; m < M; ++m) w[m] = w[m] * inv;
    put(V, w);
  }
  int k = 0;
  for (; active && k < kmax; ++k) {  // :46
    apply(V + size_t(k) * L);        // :48
    row(vout, w);
    n_ax = k + 1;
    double* Hk = Hi + ((k * (k + 1)) >> 1);
    for (int i = 0; i <= k; ++i) {  // :52-58 modified Gram-Schmidt, in order
      row(V + size_t(i) * L, vi);
      const double hik = dot(vi, w);
#pragma unroll
      for (int m = 0; m < M; ++m) w[m] = __builtin_fma(-hik, vi[m], w[m]);
      if (lane == 0) Hk[i] = hik;
    }
    const double hn = ::sqrt(dot(w, w));  // :60
    if (__any(abs_t(hn) < DBL_EPSILON || !finite_t(hn))) {  // :63-65
      reason = __any(!finite_t(hn)) ? 4 : 3;
      active = false;
      break;
    }
    const double inv = 1.0 / hn;  // :67
#pragma unroll
    for (int m = 0; m < M; ++m) w[m] = w[m] * inv;
    put(V + size_t(k + 1) * L, w);
    fence();
    const double en = hess_column_lds(Hi, gi, rhoi, k, hn, lane == 0);  // :71-90
    fence();
    if (__any(abs_t(en) < tol)) {  // :93-95 — column k is NOT used by the solve
      reason = 1, ksolve = k;
      active = false;
      break;
    }
  }
  if (reason == 0) ksolve = kmax;
  if (reason <= 1) {
    const int ks = ksolve;
    if (lane == 0) {  // :100-107 (a k x k triangle: serial)
      for (int i = ks - 1; i >= 0; --i) {
        double ei = rhoi[i];
        for (int j = ks - 1; j > i; --j) ei -= Hi[((j * (j + 1)) >> 1) + i] * rhoi[j];
        rhoi[i] = ei / Hi[((i * (i + 1)) >> 1) + i];
      }
    }
    fence();
    double acc[M];
#pragma unroll
    for (int m = 0; m < M; ++m) acc[m] = 0.0;
    for (int j = 0; j < ks; ++j) {  // :110-111, accumulated j-ascending from 0
      const double yj = rhoi[j];
      row(V + size_t(j) * L, vi);
#pragma unroll
      for (int m = 0; m < M; ++m) acc[m] = __builtin_fma(vi[m], yj, acc[m]);
    }
#pragma unroll
    for (int m = 0; m < M; ++m) xr[m] = xr[m] + acc[m];
  }
  if (reason == 4) {
#pragma unroll
    for (int m = 0; m < M; ++m) xr[m] = quiet_nan<double>();
  }
#pragma unroll
  for (int m = 0; m < M; ++m)
    if (lane + 64 * m < L) x[size_t(b) * L + lane + 64 * m] = xr[m];
  if (lane == 0) n_ax_out[b] = n_ax, reason_out[b] = reason;
}

// Device workspace of an operator plugin: grown on demand, reused by every solve, released when the plugin is unloaded.
struct OpWorkspace {
  std::mutex mu;
  int device = -1;
  size_t cap_d = 0, cap_i = 0;
  double* d = nullptr;
  int* di = nullptr;
  hipStream_t stream = nullptr;
  ~OpWorkspace() { release(); }
  void release() {
    if (device < 0) return;
    if (hipSetDevice(device) != hipSuccess) return;  // (the runtime may already be gone at process exit)
    if (d) (void)hipFree(d);
    if (di) (void)hipFree(di);
    if (stream) (void)hipStreamDestroy(stream);
    d = nullptr, di = nullptr, stream = nullptr, cap_d = cap_i = 0, device = -1;
  }
  int ensure(int dev, size_t need_d, size_t need_i) {
    if (dev != device) release();
    HIP_TRY(hipSetDevice(dev));
    device = dev;
    if (!stream) HIP_TRY(hipStreamCreateWithFlags(&stream, hipStreamNonBlocking));
    if (need_d > cap_d) {
      if (d) (void)hipFree(d);
      d = nullptr, cap_d = 0;
      HIP_TRY(hipMalloc(reinterpret_cast<void**>(&d), need_d * sizeof(double)));
      cap_d = need_d;
    }
    if (need_i > cap_i) {
      if (di) (void)hipFree(di);
      di = nullptr, cap_i = 0;
      HIP_TRY(hipMalloc(reinterpret_cast<void**>(&di), need_i * sizeof(int)));
      cap_i = need_i;
    }
    return 0;
  }
};
template <class Op>
inline OpWorkspace& op_workspace() {
  static OpWorkspace ws;
  return ws;
}

constexpr size_t kGmresWaveLdsLimit = 150 * 1024;

// host side of one solve: instance-major host arrays in and out; the plugin's own stream and workspace
template <class Op>
int gmres_op_solve(int device, int batch, int kmax, double tol, const double* params, double* x, const double* bvec,
                   int32_t* n_ax, int32_t* reason) {
  constexpr int L = Op::len, NP = Op::n_params;
  if (batch < 1 || kmax < 1 || !(tol >= 0) || !x || !bvec || (NP && !params))
    return fail(CGMRES_HIP_EINVAL, "gmres_user: bad argument");
  if (long(L) * (kmax + 1) >= 65536) return fail(CGMRES_HIP_EINVAL, "gmres_user: len*(k_max+1) beyond the reference's 16-bit index range");
  OpWorkspace& ws = op_workspace<Op>();
  std::lock_guard<std::mutex> lock(ws.mu);
  const size_t lds = GmresWaveLds::count(L, kmax) * sizeof(double);
  if (lds <= kGmresWaveLdsLimit) {
    // wave per system: instance-major on the device as well (no transposes)
    const size_t nx = size_t(L) * batch, np = size_t(NP ? NP : 1) * batch;
    if (int rc = ws.ensure(device, 2 * nx + np, size_t(2) * batch)) return rc;
    double *dx = ws.d, *db = dx + nx, *dp = db + nx;
    int* di = ws.di;
    HIP_TRY(hipMemcpyAsync(dx, x, nx * sizeof(double), hipMemcpyHostToDevice, ws.stream));
    HIP_TRY(hipMemcpyAsync(db, bvec, nx * sizeof(double), hipMemcpyHostToDevice, ws.stream));
    if (NP) HIP_TRY(hipMemcpyAsync(dp, params, size_t(NP) * batch * sizeof(double), hipMemcpyHostToDevice, ws.stream));
    HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(gmres_wave_kernel<Op>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, int(lds)));
    gmres_wave_kernel<Op><<<batch, 64, lds, ws.stream>>>(batch, kmax, tol, dp, dx, db, di, di + batch);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpyAsync(x, dx, nx * sizeof(double), hipMemcpyDeviceToHost, ws.stream));
    std::vector<int> hi(size_t(2) * batch);
    HIP_TRY(hipMemcpyAsync(hi.data(), di, hi.size() * sizeof(int), hipMemcpyDeviceToHost, ws.stream));
    HIP_TRY(hipStreamSynchronize(ws.stream));
    for (int i = 0; i < batch; ++i) {
      if (n_ax) n_ax[i] = hi[i];
      if (reason) reason[i] = hi[batch + i];
    }
    return 0;
  }
  // beyond the LDS of one wave: one lane per system, element-major device arrays (transposed on the host)
  const int ldb = (batch + 63) / 64 * 64, k1 = kmax + 1;
  const size_t nV = size_t(L) * k1 * ldb, nH = size_t(k1) * k1 * ldb, nr = size_t(k1) * ldb, ng = size_t(3) * kmax * ldb;
  const size_t nx = size_t(L) * ldb, total = nV + nH + nr + ng + 2 * nx + size_t(NP ? NP : 1) * batch;
  if (int rc = ws.ensure(device, total, size_t(2) * ldb)) return rc;
  double* d = ws.d;
  int* di = ws.di;
  double *dV = d, *dH = dV + nV, *dr = dH + nH, *dg = dr + nr, *dx = dg + ng, *db = dx + nx, *dp = db + nx;
  std::vector<double> hx(nx, 0.0), hb(nx, 0.0);
  for (int i = 0; i < batch; ++i)
    for (int e = 0; e < L; ++e) hx[size_t(e) * ldb + i] = x[size_t(i) * L + e], hb[size_t(e) * ldb + i] = bvec[size_t(i) * L + e];
  HIP_TRY(hipMemsetAsync(d, 0, (nV + nH + nr + ng) * sizeof(double), ws.stream));
  HIP_TRY(hipMemcpyAsync(dx, hx.data(), nx * sizeof(double), hipMemcpyHostToDevice, ws.stream));
  HIP_TRY(hipMemcpyAsync(db, hb.data(), nx * sizeof(double), hipMemcpyHostToDevice, ws.stream));
  if (NP) HIP_TRY(hipMemcpyAsync(dp, params, size_t(NP) * batch * sizeof(double), hipMemcpyHostToDevice, ws.stream));
  gmres_op_kernel<Op><<<ldb / 64, 64, 0, ws.stream>>>(batch, ldb, kmax, tol, dp, dx, db, dV, dH, dr, dg, di, di + ldb);
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipMemcpyAsync(hx.data(), dx, nx * sizeof(double), hipMemcpyDeviceToHost, ws.stream));
  std::vector<int> hi(size_t(2) * ldb);
  HIP_TRY(hipMemcpyAsync(hi.data(), di, hi.size() * sizeof(int), hipMemcpyDeviceToHost, ws.stream));
  HIP_TRY(hipStreamSynchronize(ws.stream));
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
