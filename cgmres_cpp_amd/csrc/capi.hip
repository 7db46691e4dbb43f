// libcgmres_hip.so — extern "C" entry points of include/cgmres_hip.h.
// Host side of the batched C/GMRES tick: a handle owns the HBM-resident controller state of `batch`
// instances (the private members of the reference's Cgmres/Gmres objects, cgmres.hpp:195-202 /
// gmres.hpp:120-124) behind one of three kernel mappings:
//   variant 2 "wg"      (tick_wg.hip.h)   one workgroup per 16 (or 8) instances, LDS-staged sweeps, DPP row
//                                         reductions — the default whenever the problem fits its LDS budget
//   variant 3 "wg-lean" (tick_wg.hip.h)   the same on half the LDS: two workgroups per CU — the default when a batch
//                                         needs more workgroups than the GPU has CUs
//   variant 1 "lane"    (tick_lane.hip.h) one lane per instance, reference statement order, any size
// There is no CPU implementation of the path in this library.
#include <dlfcn.h>

#include <deque>
#include <mutex>

#include "ctx_common.hip.h"
#include "factory.hip.h"
#include "util_kernels.hip.h"

namespace {

using cgm::fail;

// User models registered at run time (cgmres_hip_register_model): model ids CGMRES_HIP_MODEL_USER_BASE + n.
struct Plugin {
  void* dl;
  std::string path;
  cgm::ModelInfo info;
  cgmres_hip_ctx* (*make)(const cgmres_hip_config*);
  int (*probe)(const double*, const double*, const double*, const double*, double*, void*);
  const char* (*last_error)(void);
};
std::deque<Plugin>& plugins() {  // deque: find_plugin hands out pointers that must survive later push_backs
  static std::deque<Plugin> v;
  return v;
}
std::mutex& plugins_mutex() {
  static std::mutex m;
  return m;
}
const Plugin* find_plugin(int id) {
  std::lock_guard<std::mutex> g(plugins_mutex());
  const int n = id - CGMRES_HIP_MODEL_USER_BASE;
  return n >= 0 && n < int(plugins().size()) ? &plugins()[n] : nullptr;
}

// Device operators registered at run time (cgmres_hip_register_operator): ids 0, 1, ...
struct OpPlugin {
  void* dl;
  std::string path;
  int32_t len, n_params;
  int (*solve)(int32_t, int32_t, int32_t, double, const double*, double*, const double*, int32_t*, int32_t*);
  const char* (*last_error)(void);
};
std::deque<OpPlugin>& op_plugins() {
  static std::deque<OpPlugin> v;
  return v;
}
const OpPlugin* find_op(int id) {
  std::lock_guard<std::mutex> g(plugins_mutex());
  return id >= 0 && id < int(op_plugins().size()) ? &op_plugins()[id] : nullptr;
}

cgm::ModelInfo model_info(int id, bool* ok) {
  *ok = true;
  if (const Plugin* pl = find_plugin(id)) return pl->info;
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

cgmres_hip_ctx* make_ctx(const cgmres_hip_config& cfg, int* resolved) {
  const bool f32 = cfg.dtype == CGMRES_HIP_F32;
  if (const Plugin* pl = find_plugin(cfg.model_id)) {  // user models: the plugin picks the mapping (user_model.hip.h)
    *resolved = cfg.variant;
    return pl->make(&cfg);
  }
  switch (cfg.model_id) {
    case CGMRES_HIP_MODEL_PENDULUM:
      return f32 ? cgm::make_pendulum_f32(cfg, resolved) : cgm::make_pendulum_f64(cfg, resolved);
    case CGMRES_HIP_MODEL_MSD:
      return f32 ? cgm::make_msd_f32(cfg, resolved) : cgm::make_msd_f64(cfg, resolved);
    case CGMRES_HIP_MODEL_SEMIACTIVE:
      return f32 ? cgm::make_semiactive_f32(cfg, resolved) : cgm::make_semiactive_f64(cfg, resolved);
  }
  return nullptr;
}

__global__ void sincos_selftest_kernel(const double* a, double* s, double* c, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) cgm::sincos_f64(a[i], &s[i], &c[i]);
}

}  // namespace

#define NEED(h) \
  if (!(h)) return fail(CGMRES_HIP_EINVAL, "%s: null handle", __func__)

extern "C" {

const char* cgmres_hip_last_error(void) { return cgm::g_err.c_str(); }

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
  // every step is checked: the facade's model fingerprint (include/cgmres.hpp) trusts `out` to pick a device model
  hipError_t e = hipMemcpy(dx, x, mi.dim_x * 8, hipMemcpyHostToDevice);
  if (e == hipSuccess) e = hipMemcpy(du, u, mi.dim_u * 8, hipMemcpyHostToDevice);
  if (e == hipSuccess && mi.dim_p) e = hipMemcpy(dp, p, mi.dim_p * 8, hipMemcpyHostToDevice);
  if (e == hipSuccess) e = hipMemcpy(dl, lmd, mi.dim_x * 8, hipMemcpyHostToDevice);
  if (e == hipSuccess) e = hipMemset(dout, 0xff, n_out * 8);  // (NaN pattern: a kernel that did not run cannot pass for a match)
  const char* what = "copy in";
  if (e == hipSuccess) {
    what = "kernel";
    if (const Plugin* pl = find_plugin(model_id)) {
      if (pl->probe(dx, du, dp, dl, dout, nullptr) != 0) e = hipErrorLaunchFailure;
    } else {
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
      e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipDeviceSynchronize();  // (execution errors of the probe kernel surface here)
  }
  if (e == hipSuccess) {
    what = "copy out";
    e = hipMemcpy(out, dout, n_out * 8, hipMemcpyDeviceToHost);
  }
  (void)hipFree(d);
  if (e != hipSuccess) return fail(CGMRES_HIP_ERUNTIME, "model_probe (%s): %s", what, hipGetErrorString(e));
  return 0;
}

int cgmres_hip_selftest_sincos(int32_t device, const double* a, int32_t n, double* s, double* c) {
  if (!a || !s || !c || n < 0) return fail(CGMRES_HIP_EINVAL, "selftest_sincos: bad argument");
  if (int rc = check_device(device)) return rc;
  HIP_TRY(hipSetDevice(device));
  double* d = nullptr;
  HIP_TRY(hipMalloc(reinterpret_cast<void**>(&d), size_t(3) * (n ? n : 1) * sizeof(double)));
  hipError_t e = hipMemcpy(d, a, size_t(n) * 8, hipMemcpyHostToDevice);
  if (e == hipSuccess && n > 0) {
    sincos_selftest_kernel<<<(n + 255) / 256, 256>>>(d, d + n, d + 2 * size_t(n), n);
    e = hipGetLastError();
    if (e == hipSuccess) e = hipDeviceSynchronize();
  }
  if (e == hipSuccess) e = hipMemcpy(s, d + n, size_t(n) * 8, hipMemcpyDeviceToHost);
  if (e == hipSuccess) e = hipMemcpy(c, d + 2 * size_t(n), size_t(n) * 8, hipMemcpyDeviceToHost);
  (void)hipFree(d);
  if (e != hipSuccess) return fail(CGMRES_HIP_ERUNTIME, "selftest_sincos: %s", hipGetErrorString(e));
  return 0;
}

int cgmres_hip_register_model(const char* plugin_path, int32_t* model_id) {
  if (!plugin_path || !model_id) return fail(CGMRES_HIP_EINVAL, "register_model: null argument");
  {
    std::lock_guard<std::mutex> g(plugins_mutex());
    for (size_t n = 0; n < plugins().size(); ++n)
      if (plugins()[n].path == plugin_path) {
        *model_id = CGMRES_HIP_MODEL_USER_BASE + int(n);
        return 0;
      }
  }
  void* dl = dlopen(plugin_path, RTLD_NOW | RTLD_LOCAL);
  if (!dl) return fail(CGMRES_HIP_EINVAL, "register_model: %s", dlerror());
  auto abi = reinterpret_cast<int32_t (*)(void)>(dlsym(dl, "cgmres_hip_plugin_abi"));
  auto info = reinterpret_cast<void (*)(int32_t*, double*)>(dlsym(dl, "cgmres_hip_plugin_info"));
  Plugin pl{};
  pl.dl = dl, pl.path = plugin_path;
  pl.make = reinterpret_cast<decltype(pl.make)>(dlsym(dl, "cgmres_hip_plugin_make"));
  pl.probe = reinterpret_cast<decltype(pl.probe)>(dlsym(dl, "cgmres_hip_plugin_probe"));
  pl.last_error = reinterpret_cast<decltype(pl.last_error)>(dlsym(dl, "cgmres_hip_plugin_last_error"));
  if (!abi || !info || !pl.make || !pl.probe || !pl.last_error) {
    dlclose(dl);
    return fail(CGMRES_HIP_EINVAL, "register_model: %s is not a cgmres_hip model plugin", plugin_path);
  }
  if (abi() != CGMRES_HIP_ABI_VERSION) {
    dlclose(dl);
    return fail(CGMRES_HIP_EINVAL, "register_model: plugin ABI %d, library has %d", abi(), CGMRES_HIP_ABI_VERSION);
  }
  int32_t dims[5];
  double tun[6];
  info(dims, tun);
  pl.info = {dims[0], dims[1], dims[2], dims[3], dims[4], tun[0], tun[1], tun[2], tun[3], tun[4], tun[5]};
  if (dims[0] < 1 || dims[1] < 1 || dims[2] < 0) {
    dlclose(dl);
    return fail(CGMRES_HIP_EINVAL, "register_model: bad dimensions %d/%d/%d", dims[0], dims[1], dims[2]);
  }
  std::lock_guard<std::mutex> g(plugins_mutex());
  plugins().push_back(pl);
  *model_id = CGMRES_HIP_MODEL_USER_BASE + int(plugins().size()) - 1;
  return 0;
}

int cgmres_hip_register_operator(const char* plugin_path, int32_t* op_id) {
  if (!plugin_path || !op_id) return fail(CGMRES_HIP_EINVAL, "register_operator: null argument");
  {
    std::lock_guard<std::mutex> g(plugins_mutex());
    for (size_t n = 0; n < op_plugins().size(); ++n)
      if (op_plugins()[n].path == plugin_path) {
        *op_id = int(n);
        return 0;
      }
  }
  void* dl = dlopen(plugin_path, RTLD_NOW | RTLD_LOCAL);
  if (!dl) return fail(CGMRES_HIP_EINVAL, "register_operator: %s", dlerror());
  auto abi = reinterpret_cast<int32_t (*)(void)>(dlsym(dl, "cgmres_hip_opplugin_abi"));
  auto info = reinterpret_cast<void (*)(int32_t*)>(dlsym(dl, "cgmres_hip_opplugin_info"));
  OpPlugin pl{};
  pl.dl = dl, pl.path = plugin_path;
  pl.solve = reinterpret_cast<decltype(pl.solve)>(dlsym(dl, "cgmres_hip_opplugin_solve"));
  pl.last_error = reinterpret_cast<decltype(pl.last_error)>(dlsym(dl, "cgmres_hip_opplugin_last_error"));
  if (!abi || !info || !pl.solve || !pl.last_error) {
    dlclose(dl);
    return fail(CGMRES_HIP_EINVAL, "register_operator: %s is not a cgmres_hip operator plugin", plugin_path);
  }
  if (abi() != CGMRES_HIP_ABI_VERSION) {
    dlclose(dl);
    return fail(CGMRES_HIP_EINVAL, "register_operator: plugin ABI %d, library has %d", abi(), CGMRES_HIP_ABI_VERSION);
  }
  int32_t dims[2];
  info(dims);
  pl.len = dims[0], pl.n_params = dims[1];
  if (pl.len < 1 || pl.n_params < 0) {
    dlclose(dl);
    return fail(CGMRES_HIP_EINVAL, "register_operator: bad dimensions %d/%d", dims[0], dims[1]);
  }
  std::lock_guard<std::mutex> g(plugins_mutex());
  op_plugins().push_back(pl);
  *op_id = int(op_plugins().size()) - 1;
  return 0;
}

int cgmres_hip_operator_info(int32_t op_id, int32_t dims[2]) {
  const OpPlugin* pl = find_op(op_id);
  if (!pl || !dims) return fail(CGMRES_HIP_EINVAL, "operator_info: unknown operator %d", op_id);
  dims[0] = pl->len, dims[1] = pl->n_params;
  return 0;
}

int cgmres_hip_gmres_user(int32_t op_id, int32_t device, int32_t batch, int32_t k_max, double tol, const double* params,
                          double* x, const double* b, int32_t* n_ax, int32_t* reason) {
  const OpPlugin* pl = find_op(op_id);
  if (!pl) return fail(CGMRES_HIP_EINVAL, "gmres_user: unknown operator %d", op_id);
  if (int rc = check_device(device)) return rc;
  const int rc = pl->solve(device, batch, k_max, tol, params, x, b, n_ax, reason);
  if (rc) cgm::g_err = pl->last_error();  // (the plugin has its own copy of the error string)
  return rc;
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
  if (cfg->variant < 0 || cfg->variant > 4) return fail(CGMRES_HIP_EINVAL, "unknown variant %d", cfg->variant);
  if (cfg->flags & ~(CGMRES_HIP_FLAG_SERIAL_COSTATE | CGMRES_HIP_FLAG_IPW8 | CGMRES_HIP_FLAG_NO_BINNING | CGMRES_HIP_FLAG_TWO_PASS_COSTATE |
                     CGMRES_HIP_FLAG_NO_WAVE | CGMRES_HIP_FLAG_WAVE_FRESH_TRIG | CGMRES_HIP_FLAG_WAVE_SERIAL_SWEEPS |
                     CGMRES_HIP_FLAG_SERIAL_STATE_SWEEP))
    return fail(CGMRES_HIP_EINVAL, "unknown flags 0x%x", cfg->flags);
  if (cfg->reserved != 0) return fail(CGMRES_HIP_EINVAL, "cgmres_hip_config.reserved must be 0 (got %d)", cfg->reserved);
  if (int rc = check_device(cfg->device)) return rc;
  int resolved = 0;
  cgmres_hip_ctx* c = make_ctx(*cfg, &resolved);
  if (!c) return fail(CGMRES_HIP_EINVAL, "variant %d does not support model %d with dv = %d, k_max = %d", resolved,
                      cfg->model_id, cfg->dv, cfg->k_max);
  c->cfg = *cfg;  // init() replaces cfg.variant (the request, 0 = library's choice) by the resolved mapping
  if (int rc = c->init()) {
    if (const Plugin* pl = find_plugin(cfg->model_id)) cgm::g_err = pl->last_error();  // the plugin has its own copy
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
const char* cgmres_hip_variant_name(cgmres_hip_handle h) { return h ? h->variant_name() : nullptr; }
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
// The reference's DEBUG_MODE builds exit(-1) when an output argument aliases an input (cgmres.hpp:119-124,
// matrix.hpp:76-81); at the ABI the same calls are refused.
int cgmres_hip_control(cgmres_hip_handle h, void* u, const void* x) {
  NEED(h);
  if (u && u == x) return fail(CGMRES_HIP_EINVAL, "control: u and x are the same buffer (cgmres.hpp DEBUG_MODE alias guard)");
  return h->control_host(u, x);
}
int cgmres_hip_control_device(cgmres_hip_handle h, void* u, const void* x) {
  NEED(h);
  if (u && u == x) return fail(CGMRES_HIP_EINVAL, "control_device: u and x are the same buffer (cgmres.hpp DEBUG_MODE alias guard)");
  return h->control_device(u, x, nullptr);
}
int cgmres_hip_closed_loop_device(cgmres_hip_handle h, void* x, void* u, int32_t n_ticks) {
  NEED(h);
  if (n_ticks < 0) return fail(CGMRES_HIP_EINVAL, "n_ticks < 0");
  return h->closed_loop(x, u, n_ticks);
}
int cgmres_hip_closed_loop_device_ptau(cgmres_hip_handle h, void* x, void* u, int32_t n_ticks, const void* ptau_seq,
                                       int per_instance) {
  NEED(h);
  if (n_ticks < 0) return fail(CGMRES_HIP_EINVAL, "n_ticks < 0");
  if (h->np && !ptau_seq) return fail(CGMRES_HIP_EINVAL, "closed_loop_device_ptau: null ptau sequence");
  return h->closed_loop(x, u, n_ticks, h->np ? ptau_seq : nullptr, per_instance);
}
int cgmres_hip_shard_bounds(int32_t n, int32_t world, int32_t rank, int32_t* lo, int32_t* hi) {
  if (n < 0 || world < 1 || rank < 0 || rank >= world || !lo || !hi)
    return fail(CGMRES_HIP_EINVAL, "shard_bounds: n = %d, world = %d, rank = %d", n, world, rank);
  const int32_t base = n / world, extra = n % world;
  *lo = rank * base + (rank < extra ? rank : extra);
  *hi = *lo + base + (rank < extra ? 1 : 0);
  return 0;
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
  if (ret == U || ret == x) return fail(CGMRES_HIP_EINVAL, "F_func: ret aliases an input (cgmres.hpp:119-124)");
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
  if (out == v) return fail(CGMRES_HIP_EINVAL, "Ax_func: Ax aliases v (cgmres.hpp:119-124)");
  return h->hook_Ax(out, v);
}
int cgmres_hip_gmres(cgmres_hip_handle h, void* x, const void* b) {
  NEED(h);
  if (!x || !b) return fail(CGMRES_HIP_EINVAL, "gmres: null pointer");
  if (x == b) return fail(CGMRES_HIP_EINVAL, "gmres: x aliases b (matrix.hpp:76-81)");
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

#ifdef CGM_STAMPS
/* diagnostic build only (tools/phase_stamps.py): copies and clears the 64 stamp words of the pendulum/f64 wg context */
int cgmres_hip_debug_stamps(long long* out) {
  long long* p = cgm::debug_stamps_ptr();
  if (!p || !out) return fail(CGMRES_HIP_EINVAL, "no stamp buffer");
  HIP_TRY(hipDeviceSynchronize());
  HIP_TRY(hipMemcpy(out, p, 64 * sizeof(long long), hipMemcpyDeviceToHost));
  HIP_TRY(hipMemset(p, 0, 64 * sizeof(long long)));
  return 0;
}
#endif

}  // extern "C"
