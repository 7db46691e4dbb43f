// libcgmres_hip.so — implementation of include/cgmres_hip.h.
// Host side of the batched C/GMRES tick: owns the HBM-resident controller state of `batch` instances
// (the private members of the reference's Cgmres/Gmres objects, cgmres.hpp:195-202 / gmres.hpp:120-124,
// laid out element-major), computes the batch-wide scalars of a tick (t, dtau) and launches the kernels.
// There is no CPU implementation of the path in this library.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/cgmres_hip.h"
#include "models.hip.h"
#include "tick_lane.hip.h"
#include "util_kernels.hip.h"

namespace {

thread_local std::string g_err;

int fail(int code, const char* fmt, ...) {
  char buf[512];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof buf, fmt, ap);
  va_end(ap);
  g_err = buf;
  return code;
}

#define HIP_TRY(expr)                                                                                   \
  do {                                                                                                  \
    hipError_t e_ = (expr);                                                                             \
    if (e_ != hipSuccess)                                                                               \
      return fail(e_ == hipErrorOutOfMemory ? CGMRES_HIP_ENOMEM : CGMRES_HIP_ERUNTIME, "%s: %s (%s:%d)", #expr, \
                  hipGetErrorString(e_), __FILE__, __LINE__);                                           \
  } while (0)

cgm::ModelInfo model_info(int id, bool* ok) {
  *ok = true;
  switch (id) {
    case CGMRES_HIP_MODEL_PENDULUM:
      return cgm::PendulumDev<double>::info();
    case CGMRES_HIP_MODEL_MSD:
      return cgm::MsdDev<double>::info();
    case CGMRES_HIP_MODEL_SEMIACTIVE:
      return cgm::SemiactiveDev<double>::info();
  }
  *ok = false;
  return {};
}

int check_device(int device) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess || n <= 0)
    return fail(CGMRES_HIP_ENODEV, "no HIP device visible: this library has no CPU path");
  if (device < 0 || device >= n) return fail(CGMRES_HIP_ENODEV, "device %d out of range (%d visible)", device, n);
  hipDeviceProp_t prop;
  HIP_TRY(hipGetDeviceProperties(&prop, device));
  if (std::strncmp(prop.gcnArchName, "gfx950", 6) != 0)
    return fail(CGMRES_HIP_ENODEV, "device %d is %s; the kernels are built for gfx950 only", device, prop.gcnArchName);
  return 0;
}

}  // namespace

// Type-erased controller batch.
struct cgmres_hip_ctx {
  cgmres_hip_config cfg{};
  int nx = 0, nu = 0, np = 0, L = 0, ldb = 0;
  hipStream_t stream = nullptr;
  bool own_stream = false;
  hipEvent_t ev0 = nullptr, ev1 = nullptr;
  virtual ~cgmres_hip_ctx() {}
  virtual int init() = 0;
  virtual int set_ptau(const void*, int per_instance, bool repeat) = 0;
  virtual int init_u0(const void*, int per_instance) = 0;
  virtual int init_u0_newton(void*, const void*, const void*, int) = 0;
  virtual int control_host(void*, const void*) = 0;
  virtual int control_device(void*, const void*, void* x_next) = 0;
  virtual int closed_loop(void*, void*, int) = 0;
  virtual double time() const = 0;
  virtual int get_state(double*, void*, void*) = 0;
  virtual int set_state(double, const void*, const void*) = 0;
  virtual int get_status(int32_t*, int32_t*) = 0;
  virtual int get_krylov(void*, void*, void*, void*) = 0;
  virtual int hook_F(void*, const void*, const void*, double) = 0;
  virtual int hook_prepare(void*, const void*) = 0;
  virtual int hook_Ax(void*, const void*) = 0;
  virtual int hook_gmres(void*, const void*) = 0;
};

namespace {

template <class M, class T>
struct Ctx final : cgmres_hip_ctx {
  cgm::TickParams<T> P{};
  T t = T(0);
  std::vector<void*> owned;
  T *stage_a = nullptr, *stage_b = nullptr;  // instance-major device staging, grown on demand
  size_t stage_a_n = 0, stage_b_n = 0;
  T *x_dev = nullptr, *u_dev = nullptr;      // [B][nx], [B][nu] staging for the host-pointer entry points
  int* status_host_tmp = nullptr;

  ~Ctx() override {
    (void)hipSetDevice(cfg.device);
    if (stream) (void)hipStreamSynchronize(stream);
    for (void* p : owned) (void)hipFree(p);
    if (stage_a) (void)hipFree(stage_a);
    if (stage_b) (void)hipFree(stage_b);
    if (ev0) (void)hipEventDestroy(ev0);
    if (ev1) (void)hipEventDestroy(ev1);
    if (own_stream && stream) (void)hipStreamDestroy(stream);
  }

  template <class Q>
  int dalloc(Q** p, size_t n) {
    void* q = nullptr;
    HIP_TRY(hipMalloc(&q, (n ? n : 1) * sizeof(Q)));
    HIP_TRY(hipMemsetAsync(q, 0, (n ? n : 1) * sizeof(Q), stream));
    owned.push_back(q);
    *p = static_cast<Q*>(q);
    return 0;
  }
  int grow(T** buf, size_t* have, size_t need) {
    if (*have >= need) return 0;
    if (*buf) {
      HIP_TRY(hipStreamSynchronize(stream));
      HIP_TRY(hipFree(*buf));
      *buf = nullptr;
      *have = 0;
    }
    void* q = nullptr;
    HIP_TRY(hipMalloc(&q, need * sizeof(T)));
    *buf = static_cast<T*>(q);
    *have = need;
    return 0;
  }

  int init() override {
    HIP_TRY(hipSetDevice(cfg.device));
    if (cfg.stream) {
      stream = static_cast<hipStream_t>(cfg.stream);
    } else {
      HIP_TRY(hipStreamCreateWithFlags(&stream, hipStreamNonBlocking));
      own_stream = true;
    }
    HIP_TRY(hipEventCreate(&ev0));
    HIP_TRY(hipEventCreate(&ev1));
    nx = M::NX, nu = M::NU, np = M::NP;
    L = nu * cfg.dv;
    ldb = (cfg.batch + 63) / 64 * 64;
    const size_t ld = ldb, k1 = cfg.k_max + 1;
    P.B = cfg.batch, P.ldb = ldb, P.dv = cfg.dv, P.kmax = cfg.k_max, P.L = L;
    P.h = T(cfg.h), P.dt = T(cfg.dt), P.tol = T(cfg.tol);
    P.inv_h = T(1.0) / P.h;
    P.one_m_zh = (1 - T(cfg.zeta) * P.h);
    int rc = 0;
    if ((rc = dalloc(&P.U, L * ld)) || (rc = dalloc(&P.dUdt, L * ld)) || (rc = dalloc(&P.Fh, L * ld)) ||
        (rc = dalloc(&P.bvec, L * ld)) || (rc = dalloc(&P.V, L * k1 * ld)) || (rc = dalloc(&P.H, k1 * k1 * ld)) ||
        (rc = dalloc(&P.g, 3 * cfg.k_max * ld)) || (rc = dalloc(&P.rho, k1 * ld)) || (rc = dalloc(&P.xdxh, nx * ld)) ||
        (rc = dalloc(&P.ptau, size_t(np) * (cfg.dv + 1) * ld)) || (rc = dalloc(&P.traj, size_t(nx) * cfg.dv * ld)) ||
        (rc = dalloc(&P.trig, size_t(M::NC) * cfg.dv * ld)) || (rc = dalloc(&P.n_ax, ld)) ||
        (rc = dalloc(&P.reason, ld)) || (rc = dalloc(&x_dev, size_t(cfg.batch) * nx)) ||
        (rc = dalloc(&u_dev, size_t(cfg.batch) * nu)))
      return rc;
    HIP_TRY(hipStreamSynchronize(stream));
    return 0;
  }

  dim3 lane_grid() const { return dim3((cfg.batch + 63) / 64); }
  T dtau_of(T tt) const {  // cgmres.hpp:32-34, evaluated once per tick on the host for the whole batch
    return T(cfg.Tf) * (1 - std::exp(-T(cfg.alpha) * tt)) / T(cfg.dv);
  }

  // host instance-major [B or 1][n] -> device element-major [n*rep][ldb]
  int upload(T* dst, const void* src, int n, int per_instance, int stages_rep) {
    const size_t cnt = size_t(per_instance ? cfg.batch : 1) * n;
    if (int rc = grow(&stage_a, &stage_a_n, cnt)) return rc;
    HIP_TRY(hipMemcpyAsync(stage_a, src, cnt * sizeof(T), hipMemcpyHostToDevice, stream));
    dim3 grid((cfg.batch + 255) / 256, n * (stages_rep ? stages_rep : 1));
    if (stages_rep)
      cgm::replicate_stages<T><<<grid, 256, 0, stream>>>(dst, stage_a, cfg.batch, ldb, n, stages_rep, !per_instance);
    else
      cgm::to_element_major<T><<<grid, 256, 0, stream>>>(dst, stage_a, cfg.batch, ldb, n, !per_instance);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipStreamSynchronize(stream));  // the host buffer may be reused by the caller right away
    return 0;
  }
  int download(void* dst, const T* src, int n) {
    if (!dst) return 0;
    const size_t cnt = size_t(cfg.batch) * n;
    if (int rc = grow(&stage_a, &stage_a_n, cnt)) return rc;
    dim3 grid((cfg.batch + 255) / 256, n);
    cgm::to_instance_major<T><<<grid, 256, 0, stream>>>(stage_a, src, cfg.batch, ldb, n);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpyAsync(dst, stage_a, cnt * sizeof(T), hipMemcpyDeviceToHost, stream));
    HIP_TRY(hipStreamSynchronize(stream));
    return 0;
  }

  int set_ptau(const void* p, int per_instance, bool repeat) override {
    HIP_TRY(hipSetDevice(cfg.device));
    if (np == 0) return 0;  // semiactive: ptau is a zero-length array (semiactive_damper/main.cpp:45-47)
    if (!p) return fail(CGMRES_HIP_EINVAL, "set_ptau: null pointer");
    return repeat ? upload(P.ptau, p, np, per_instance, cfg.dv + 1) : upload(P.ptau, p, np * (cfg.dv + 1), per_instance, 0);
  }
  int init_u0(const void* u0, int per_instance) override {
    HIP_TRY(hipSetDevice(cfg.device));
    if (!u0) return fail(CGMRES_HIP_EINVAL, "init_u0: null pointer");
    return upload(P.U, u0, nu, per_instance, cfg.dv);
  }
  int init_u0_newton(void* u0, const void* x0, const void* p0, int n_loop) override {
    HIP_TRY(hipSetDevice(cfg.device));
    if (!u0 || !x0 || (np && !p0)) return fail(CGMRES_HIP_EINVAL, "init_u0_newton: null pointer");
    if (n_loop < 0) return fail(CGMRES_HIP_EINVAL, "init_u0_newton: n_loop < 0");
    const size_t B = cfg.batch;
    if (int rc = grow(&stage_b, &stage_b_n, B * (nu + nx + np))) return rc;
    T *du = stage_b, *dx = du + B * nu, *dp = dx + B * nx;
    HIP_TRY(hipMemcpyAsync(du, u0, B * nu * sizeof(T), hipMemcpyHostToDevice, stream));
    HIP_TRY(hipMemcpyAsync(dx, x0, B * nx * sizeof(T), hipMemcpyHostToDevice, stream));
    if (np) HIP_TRY(hipMemcpyAsync(dp, p0, B * np * sizeof(T), hipMemcpyHostToDevice, stream));
    cgm::newton_u0_kernel<M, T><<<dim3((B + 63) / 64), 64, 0, stream>>>(du, dx, dp, cfg.batch, n_loop);
    HIP_TRY(hipGetLastError());
    dim3 grid((cfg.batch + 255) / 256, nu * cfg.dv);
    cgm::replicate_stages<T><<<grid, 256, 0, stream>>>(P.U, du, cfg.batch, ldb, nu, cfg.dv, 0);  // cgmres.hpp:75
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpyAsync(u0, du, B * nu * sizeof(T), hipMemcpyDeviceToHost, stream));
    HIP_TRY(hipStreamSynchronize(stream));
    return 0;
  }

  int launch_tick(T* u_out, const T* x_in, T* x_next) {
    P.x_in = x_in, P.u_out = u_out, P.x_next = x_next;
    P.dtau_h = dtau_of(t + P.h);  // cgmres.hpp:88
    P.dtau_0 = dtau_of(t);        // cgmres.hpp:91
    cgm::tick_lane_kernel<M, T><<<lane_grid(), 64, 0, stream>>>(P);
    HIP_TRY(hipGetLastError());
    t = t + P.dt;  // cgmres.hpp:107
    return 0;
  }
  int control_device(void* u, const void* x, void* x_next) override {
    HIP_TRY(hipSetDevice(cfg.device));
    if (!u || !x) return fail(CGMRES_HIP_EINVAL, "control: null pointer");
    return launch_tick(static_cast<T*>(u), static_cast<const T*>(x), static_cast<T*>(x_next));
  }
  int control_host(void* u, const void* x) override {
    HIP_TRY(hipSetDevice(cfg.device));
    if (!u || !x) return fail(CGMRES_HIP_EINVAL, "control: null pointer");
    HIP_TRY(hipMemcpyAsync(x_dev, x, size_t(cfg.batch) * nx * sizeof(T), hipMemcpyHostToDevice, stream));
    if (int rc = launch_tick(u_dev, x_dev, nullptr)) return rc;
    HIP_TRY(hipMemcpyAsync(u, u_dev, size_t(cfg.batch) * nu * sizeof(T), hipMemcpyDeviceToHost, stream));
    HIP_TRY(hipStreamSynchronize(stream));
    return 0;
  }
  int closed_loop(void* x, void* u, int n_ticks) override {
    HIP_TRY(hipSetDevice(cfg.device));
    if (!u || !x) return fail(CGMRES_HIP_EINVAL, "closed_loop: null pointer");
    for (int i = 0; i < n_ticks; ++i)
      if (int rc = launch_tick(static_cast<T*>(u), static_cast<const T*>(x), static_cast<T*>(x))) return rc;
    return 0;
  }

  double time() const override { return double(t); }
  int get_state(double* tt, void* U, void* dUdt) override {
    HIP_TRY(hipSetDevice(cfg.device));
    if (tt) *tt = double(t);
    if (int rc = download(U, P.U, L)) return rc;
    return download(dUdt, P.dUdt, L);
  }
  int set_state(double tt, const void* U, const void* dUdt) override {
    HIP_TRY(hipSetDevice(cfg.device));
    t = T(tt);
    if (U)
      if (int rc = upload(P.U, U, L, 1, 0)) return rc;
    if (dUdt)
      if (int rc = upload(P.dUdt, dUdt, L, 1, 0)) return rc;
    return 0;
  }
  int get_status(int32_t* n_ax, int32_t* reason) override {
    HIP_TRY(hipSetDevice(cfg.device));
    if (n_ax) HIP_TRY(hipMemcpyAsync(n_ax, P.n_ax, size_t(cfg.batch) * 4, hipMemcpyDeviceToHost, stream));
    if (reason) HIP_TRY(hipMemcpyAsync(reason, P.reason, size_t(cfg.batch) * 4, hipMemcpyDeviceToHost, stream));
    HIP_TRY(hipStreamSynchronize(stream));
    return 0;
  }
  int get_krylov(void* V, void* H, void* rho, void* g) override {
    HIP_TRY(hipSetDevice(cfg.device));
    const int k1 = cfg.k_max + 1;
    int rc;
    if ((rc = download(V, P.V, L * k1)) || (rc = download(H, P.H, k1 * k1)) || (rc = download(rho, P.rho, k1)) ||
        (rc = download(g, P.g, 3 * cfg.k_max)))
      return rc;
    return 0;
  }

  // ---- white-box hooks -------------------------------------------------------------------------
  int hook_F(void* ret, const void* U, const void* x, double tt) override {
    HIP_TRY(hipSetDevice(cfg.device));
    const size_t ld = ldb;
    if (int rc = grow(&stage_b, &stage_b_n, 2 * size_t(L) * ld)) return rc;
    T *Uem = stage_b, *Rem = stage_b + size_t(L) * ld;
    if (int rc = upload(Uem, U, L, 1, 0)) return rc;
    HIP_TRY(hipMemcpyAsync(x_dev, x, size_t(cfg.batch) * nx * sizeof(T), hipMemcpyHostToDevice, stream));
    P.x_in = x_dev;
    cgm::hook_F_kernel<M, T><<<lane_grid(), 64, 0, stream>>>(P, Uem, Rem, dtau_of(T(tt)));
    HIP_TRY(hipGetLastError());
    return download(ret, Rem, L);
  }
  int hook_prepare(void* b, const void* x) override {
    HIP_TRY(hipSetDevice(cfg.device));
    HIP_TRY(hipMemcpyAsync(x_dev, x, size_t(cfg.batch) * nx * sizeof(T), hipMemcpyHostToDevice, stream));
    P.x_in = x_dev;
    P.dtau_h = dtau_of(t + P.h);
    P.dtau_0 = dtau_of(t);
    cgm::hook_prepare_kernel<M, T><<<lane_grid(), 64, 0, stream>>>(P);
    HIP_TRY(hipGetLastError());
    return download(b, P.bvec, L);
  }
  int hook_Ax(void* out, const void* v) override {
    HIP_TRY(hipSetDevice(cfg.device));
    const size_t ld = ldb;
    if (int rc = grow(&stage_b, &stage_b_n, 2 * size_t(L) * ld)) return rc;
    T *Vem = stage_b, *Oem = stage_b + size_t(L) * ld;
    if (int rc = upload(Vem, v, L, 1, 0)) return rc;
    P.dtau_h = dtau_of(t + P.h);
    cgm::hook_Ax_kernel<M, T><<<lane_grid(), 64, 0, stream>>>(P, Vem, Oem);
    HIP_TRY(hipGetLastError());
    return download(out, Oem, L);
  }
  int hook_gmres(void* x, const void* b) override {
    HIP_TRY(hipSetDevice(cfg.device));
    const size_t ld = ldb;
    if (int rc = grow(&stage_b, &stage_b_n, 2 * size_t(L) * ld)) return rc;
    T *Xem = stage_b, *Bem = stage_b + size_t(L) * ld;
    if (int rc = upload(Xem, x, L, 1, 0)) return rc;
    if (int rc = upload(Bem, b, L, 1, 0)) return rc;
    P.dtau_h = dtau_of(t + P.h);
    cgm::hook_gmres_kernel<M, T><<<lane_grid(), 64, 0, stream>>>(P, Xem, Bem);
    HIP_TRY(hipGetLastError());
    return download(x, Xem, L);
  }
};

template <class T>
cgmres_hip_ctx* make_ctx(int model) {
  switch (model) {
    case CGMRES_HIP_MODEL_PENDULUM:
      return new Ctx<cgm::PendulumDev<T>, T>();
    case CGMRES_HIP_MODEL_MSD:
      return new Ctx<cgm::MsdDev<T>, T>();
    case CGMRES_HIP_MODEL_SEMIACTIVE:
      return new Ctx<cgm::SemiactiveDev<T>, T>();
  }
  return nullptr;
}

}  // namespace

#define NEED(h) \
  if (!(h)) return fail(CGMRES_HIP_EINVAL, "%s: null handle", __func__)

extern "C" {

const char* cgmres_hip_last_error(void) { return g_err.c_str(); }

int cgmres_hip_device_count(void) {
  int n = 0;
  return hipGetDeviceCount(&n) == hipSuccess ? n : 0;
}

int cgmres_hip_model_info(int32_t model_id, int32_t dims[5], double tuning[6]) {
  bool ok;
  const cgm::ModelInfo mi = model_info(model_id, &ok);
  if (!ok) return fail(CGMRES_HIP_EINVAL, "unknown model_id %d", model_id);
  if (dims) dims[0] = mi.dim_x, dims[1] = mi.dim_u, dims[2] = mi.dim_p, dims[3] = mi.dv, dims[4] = mi.k_max;
  if (tuning)
    tuning[0] = mi.dt, tuning[1] = mi.h, tuning[2] = mi.zeta, tuning[3] = mi.Tf, tuning[4] = mi.alpha, tuning[5] = mi.tol;
  return 0;
}

int cgmres_hip_default_config(int32_t model_id, cgmres_hip_config* cfg) {
  bool ok;
  const cgm::ModelInfo mi = model_info(model_id, &ok);
  if (!ok) return fail(CGMRES_HIP_EINVAL, "unknown model_id %d", model_id);
  if (!cfg) return fail(CGMRES_HIP_EINVAL, "null cfg");
  std::memset(cfg, 0, sizeof *cfg);
  cfg->abi_version = CGMRES_HIP_ABI_VERSION;
  cfg->model_id = model_id, cfg->dtype = CGMRES_HIP_F64, cfg->batch = 1, cfg->dv = mi.dv, cfg->k_max = mi.k_max;
  cfg->tol = mi.tol, cfg->dt = mi.dt, cfg->h = mi.h, cfg->zeta = mi.zeta, cfg->Tf = mi.Tf, cfg->alpha = mi.alpha;
  return 0;
}

int cgmres_hip_model_probe(int32_t model_id, int32_t device, const double* x, const double* u, const double* p,
                           const double* lmd, double* out) {
  bool ok;
  const cgm::ModelInfo mi = model_info(model_id, &ok);
  if (!ok) return fail(CGMRES_HIP_EINVAL, "unknown model_id %d", model_id);
  if (!x || !u || !lmd || !out || (mi.dim_p && !p)) return fail(CGMRES_HIP_EINVAL, "model_probe: null pointer");
  if (int rc = check_device(device)) return rc;
  HIP_TRY(hipSetDevice(device));
  const int n_in = mi.dim_x + mi.dim_u + mi.dim_p + mi.dim_x, n_out = 3 * mi.dim_x + mi.dim_u;
  double* d = nullptr;
  HIP_TRY(hipMalloc(reinterpret_cast<void**>(&d), (n_in + n_out + 1) * sizeof(double)));
  double *dx = d, *du = dx + mi.dim_x, *dp = du + mi.dim_u, *dl = dp + mi.dim_p, *dout = dl + mi.dim_x;
  (void)hipMemcpy(dx, x, mi.dim_x * 8, hipMemcpyHostToDevice);
  (void)hipMemcpy(du, u, mi.dim_u * 8, hipMemcpyHostToDevice);
  if (mi.dim_p) (void)hipMemcpy(dp, p, mi.dim_p * 8, hipMemcpyHostToDevice);
  (void)hipMemcpy(dl, lmd, mi.dim_x * 8, hipMemcpyHostToDevice);
  switch (model_id) {
    case CGMRES_HIP_MODEL_PENDULUM:
      cgm::probe_kernel<cgm::PendulumDev<double>><<<1, 1>>>(dx, du, dp, dl, dout);
      break;
    case CGMRES_HIP_MODEL_MSD:
      cgm::probe_kernel<cgm::MsdDev<double>><<<1, 1>>>(dx, du, dp, dl, dout);
      break;
    default:
      cgm::probe_kernel<cgm::SemiactiveDev<double>><<<1, 1>>>(dx, du, dp, dl, dout);
  }
  hipError_t e = hipMemcpy(out, dout, n_out * 8, hipMemcpyDeviceToHost);
  (void)hipFree(d);
  if (e != hipSuccess) return fail(CGMRES_HIP_ERUNTIME, "model_probe: %s", hipGetErrorString(e));
  return 0;
}

int cgmres_hip_create(const cgmres_hip_config* cfg, cgmres_hip_handle* out) {
  if (!cfg || !out) return fail(CGMRES_HIP_EINVAL, "create: null argument");
  *out = nullptr;
  if (cfg->abi_version != CGMRES_HIP_ABI_VERSION)
    return fail(CGMRES_HIP_EINVAL, "ABI version %d, library has %d", cfg->abi_version, CGMRES_HIP_ABI_VERSION);
  bool ok;
  const cgm::ModelInfo mi = model_info(cfg->model_id, &ok);
  if (!ok) return fail(CGMRES_HIP_EINVAL, "unknown model_id %d", cfg->model_id);
  if (cfg->dtype != CGMRES_HIP_F64 && cfg->dtype != CGMRES_HIP_F32) return fail(CGMRES_HIP_EINVAL, "bad dtype %d", cfg->dtype);
  if (cfg->batch < 1 || cfg->dv < 1 || cfg->k_max < 1)
    return fail(CGMRES_HIP_EINVAL, "batch/dv/k_max must be >= 1 (got %d/%d/%d)", cfg->batch, cfg->dv, cfg->k_max);
  // the reference indexes with uint16_t/int16_t (gmres.hpp:29, matrix.hpp:10): same limits here
  const long len = long(mi.dim_u) * cfg->dv;
  if (len >= 32768 || len * (cfg->k_max + 1) >= 65536)
    return fail(CGMRES_HIP_EINVAL, "dim_u*dv = %ld with k_max = %d exceeds the reference's 16-bit index range", len, cfg->k_max);
  if (!(cfg->h > 0) || !(cfg->dt > 0) || !(cfg->tol >= 0)) return fail(CGMRES_HIP_EINVAL, "h, dt must be > 0 and tol >= 0");
  if (cfg->variant < 0 || cfg->variant > 1) return fail(CGMRES_HIP_EINVAL, "unknown variant %d", cfg->variant);
  if (int rc = check_device(cfg->device)) return rc;
  cgmres_hip_ctx* c = cfg->dtype == CGMRES_HIP_F32 ? make_ctx<float>(cfg->model_id) : make_ctx<double>(cfg->model_id);
  if (!c) return fail(CGMRES_HIP_ENOMEM, "create: allocation failed");
  c->cfg = *cfg;
  if (int rc = c->init()) {
    delete c;
    return rc;
  }
  *out = c;
  return 0;
}

int cgmres_hip_destroy(cgmres_hip_handle h) {
  NEED(h);
  delete h;
  return 0;
}
int cgmres_hip_get_config(cgmres_hip_handle h, cgmres_hip_config* cfg) {
  NEED(h);
  if (!cfg) return fail(CGMRES_HIP_EINVAL, "null cfg");
  *cfg = h->cfg;
  return 0;
}
int cgmres_hip_set_ptau(cgmres_hip_handle h, const void* p, int per_instance) {
  NEED(h);
  return h->set_ptau(p, per_instance, false);
}
int cgmres_hip_set_ptau_repeat(cgmres_hip_handle h, const void* p, int per_instance) {
  NEED(h);
  return h->set_ptau(p, per_instance, true);
}
int cgmres_hip_init_u0(cgmres_hip_handle h, const void* u0, int per_instance) {
  NEED(h);
  return h->init_u0(u0, per_instance);
}
int cgmres_hip_init_u0_newton(cgmres_hip_handle h, void* u0, const void* x0, const void* p0, int32_t n_loop) {
  NEED(h);
  return h->init_u0_newton(u0, x0, p0, n_loop);
}
int cgmres_hip_control(cgmres_hip_handle h, void* u, const void* x) {
  NEED(h);
  return h->control_host(u, x);
}
int cgmres_hip_control_device(cgmres_hip_handle h, void* u, const void* x) {
  NEED(h);
  return h->control_device(u, x, nullptr);
}
int cgmres_hip_closed_loop_device(cgmres_hip_handle h, void* x, void* u, int32_t n_ticks) {
  NEED(h);
  if (n_ticks < 0) return fail(CGMRES_HIP_EINVAL, "n_ticks < 0");
  return h->closed_loop(x, u, n_ticks);
}
int cgmres_hip_synchronize(cgmres_hip_handle h) {
  NEED(h);
  HIP_TRY(hipSetDevice(h->cfg.device));
  HIP_TRY(hipStreamSynchronize(h->stream));
  return 0;
}
int cgmres_hip_get_time(cgmres_hip_handle h, double* t) {
  NEED(h);
  if (t) *t = h->time();
  return 0;
}
int cgmres_hip_get_state(cgmres_hip_handle h, double* t, void* U, void* dUdt) {
  NEED(h);
  return h->get_state(t, U, dUdt);
}
int cgmres_hip_set_state(cgmres_hip_handle h, double t, const void* U, const void* dUdt) {
  NEED(h);
  return h->set_state(t, U, dUdt);
}
int cgmres_hip_get_status(cgmres_hip_handle h, int32_t* n_ax, int32_t* reason) {
  NEED(h);
  return h->get_status(n_ax, reason);
}
int cgmres_hip_get_krylov(cgmres_hip_handle h, void* V, void* H, void* rho, void* g) {
  NEED(h);
  return h->get_krylov(V, H, rho, g);
}
int cgmres_hip_F_func(cgmres_hip_handle h, void* ret, const void* U, const void* x, double t) {
  NEED(h);
  if (!ret || !U || !x) return fail(CGMRES_HIP_EINVAL, "F_func: null pointer");
  return h->hook_F(ret, U, x, t);
}
int cgmres_hip_prepare(cgmres_hip_handle h, void* b, const void* x) {
  NEED(h);
  if (!x) return fail(CGMRES_HIP_EINVAL, "prepare: null pointer");
  return h->hook_prepare(b, x);
}
int cgmres_hip_Ax_func(cgmres_hip_handle h, void* out, const void* v) {
  NEED(h);
  if (!out || !v) return fail(CGMRES_HIP_EINVAL, "Ax_func: null pointer");
  return h->hook_Ax(out, v);
}
int cgmres_hip_gmres(cgmres_hip_handle h, void* x, const void* b) {
  NEED(h);
  if (!x || !b) return fail(CGMRES_HIP_EINVAL, "gmres: null pointer");
  return h->hook_gmres(x, b);
}
int cgmres_hip_timer_start(cgmres_hip_handle h) {
  NEED(h);
  HIP_TRY(hipSetDevice(h->cfg.device));
  HIP_TRY(hipEventRecord(h->ev0, h->stream));
  return 0;
}
int cgmres_hip_timer_stop(cgmres_hip_handle h, float* ms) {
  NEED(h);
  HIP_TRY(hipSetDevice(h->cfg.device));
  HIP_TRY(hipEventRecord(h->ev1, h->stream));
  HIP_TRY(hipEventSynchronize(h->ev1));
  float v = 0;
  HIP_TRY(hipEventElapsedTime(&v, h->ev0, h->ev1));
  if (ms) *ms = v;
  return 0;
}
int cgmres_hip_malloc(cgmres_hip_handle h, void** p, uint64_t bytes) {
  NEED(h);
  if (!p) return fail(CGMRES_HIP_EINVAL, "malloc: null pointer");
  HIP_TRY(hipSetDevice(h->cfg.device));
  HIP_TRY(hipMalloc(p, bytes ? bytes : 1));
  return 0;
}
int cgmres_hip_free(cgmres_hip_handle h, void* p) {
  NEED(h);
  HIP_TRY(hipSetDevice(h->cfg.device));
  HIP_TRY(hipStreamSynchronize(h->stream));
  HIP_TRY(hipFree(p));
  return 0;
}
int cgmres_hip_memcpy_h2d(cgmres_hip_handle h, void* dst, const void* src, uint64_t bytes) {
  NEED(h);
  HIP_TRY(hipSetDevice(h->cfg.device));
  HIP_TRY(hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, h->stream));
  HIP_TRY(hipStreamSynchronize(h->stream));
  return 0;
}
int cgmres_hip_memcpy_d2h(cgmres_hip_handle h, void* dst, const void* src, uint64_t bytes) {
  NEED(h);
  HIP_TRY(hipSetDevice(h->cfg.device));
  HIP_TRY(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, h->stream));
  HIP_TRY(hipStreamSynchronize(h->stream));
  return 0;
}

}  // extern "C"
