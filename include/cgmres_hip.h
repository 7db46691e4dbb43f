/* cgmres_hip.h — C ABI of libcgmres_hip.so: batched C/GMRES control ticks on MI355X (gfx950).
 *
 * This is the drop-in boundary for ONE path of blockahead/CGMRES_cpp: the per-tick FDGMRES solve
 *     Cgmres<Model>::control()      reference include/cgmres.hpp:78-110
 *       F_func / Ax_func            reference include/cgmres.hpp:113-175
 *       Gmres::gmres()              reference include/gmres.hpp:28-112
 * batched over `batch` independent controller instances that advance in lock-step (one shared t).
 * The reference has no FFI; its boundary is the C++ class surface, which include/cgmres.hpp (façade)
 * re-creates on top of the entry points below.  INTEGRATION.md shows the binding.
 *
 * Conventions
 *   - every function returns 0 on success, a negative CGMRES_HIP_E* code otherwise; the message of the
 *     last failure on the calling thread is cgmres_hip_last_error().  No exceptions cross the ABI.
 *   - `void*` vectors hold `double` (dtype F64) or `float` (dtype F32) scalars.
 *   - all vectors are INSTANCE-MAJOR, instance b first, then the reference's own layout:
 *       x[b][dim_x], u[b][dim_u], U[b][dim_u*stage + j] (cgmres.hpp:13,57),
 *       ptau[b][dim_p*stage + j], stage = 0..dv (cgmres.hpp:17,37-38).
 *     Host entry points take host pointers; *_device entry points take device pointers with the same
 *     layout (HBM resident: nothing crosses PCIe in them).
 *   - one handle = one GPU + one HIP stream; a handle is not re-entrant, distinct handles are independent
 *     (same threading contract as distinct Cgmres objects).
 *   - there is NO CPU fallback: without a usable gfx950 device cgmres_hip_create fails.
 */
#ifndef CGMRES_HIP_H_
#define CGMRES_HIP_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* 2: cgmres_hip_config gained `flags`; plugin contract (the model plugin picks the mapping, closed loop with a reference
 *    sequence); exit reason CGMRES_HIP_EXIT_NONFINITE.  A plugin or binding built against version 1 is rejected. */
#define CGMRES_HIP_ABI_VERSION 2

/* problem definitions compiled into the library (reference <example>/model.hpp) */
enum {
  CGMRES_HIP_MODEL_PENDULUM = 0,   /* arm_type_inverted_pendulum/model.hpp == multiple_controller/model2.hpp */
  CGMRES_HIP_MODEL_MSD = 1,        /* mass_spring_damper/model.hpp        == multiple_controller/model1.hpp */
  CGMRES_HIP_MODEL_SEMIACTIVE = 2, /* semiactive_damper/model.hpp */
  CGMRES_HIP_MODEL_COUNT = 3,
  CGMRES_HIP_MODEL_USER_BASE = 1000 /* ids of user models registered with cgmres_hip_register_model */
};
enum { CGMRES_HIP_F64 = 0, CGMRES_HIP_F32 = 1 };

/* error codes */
enum {
  CGMRES_HIP_OK = 0,
  CGMRES_HIP_EINVAL = -1,   /* bad argument / unsupported configuration */
  CGMRES_HIP_ENODEV = -2,   /* no usable gfx950 device */
  CGMRES_HIP_ERUNTIME = -3, /* HIP runtime error */
  CGMRES_HIP_ENOMEM = -4
};

/* per-instance exit reason of the last solve (the reference only prints "Breakdown", gmres.hpp:64) */
enum {
  CGMRES_HIP_EXIT_NATURAL = 0,        /* all k_max Arnoldi iterations ran      gmres.hpp:46 */
  CGMRES_HIP_EXIT_CONVERGED = 1,      /* |rho_e[k+1]| < tol                     gmres.hpp:93-95 */
  CGMRES_HIP_EXIT_SMALL_RESIDUAL = 2, /* ||r0|| < tol, dUdt untouched           gmres.hpp:39-41 */
  CGMRES_HIP_EXIT_BREAKDOWN = 3,      /* |h(k+1,k)| < DBL_EPSILON, dUdt untouched  gmres.hpp:63-65 */
  /* ||r0|| or h(k+1,k) is NaN/Inf: the reference has no test for it — every comparison with a NaN is false, the solve
   * runs on and NaNs propagate into dUdt, U and u.  Here the solve of that instance stops at the first
   * non-finite norm, its dUdt is set to NaN (what the reference ends up with), and the status says why. */
  CGMRES_HIP_EXIT_NONFINITE = 4
};

/* cgmres_hip_config.flags: choices inside a mapping that are normally the library's (measurement / A-B use) */
enum {
  CGMRES_HIP_FLAG_SERIAL_COSTATE = 1, /* wg mapping: keep the one-lane-per-instance costate sweep (no chunk-parallel scan) */
  CGMRES_HIP_FLAG_IPW8 = 2,           /* wg mapping: 8 instead of 16 instances per workgroup where that fits */
  CGMRES_HIP_FLAG_NO_BINNING = 4,     /* closed loop: keep instances in caller order inside the workgroups (no k-binning) */
  CGMRES_HIP_FLAG_TWO_PASS_COSTATE = 8, /* wg mapping: the two-pass chunk-parallel costate sweep also where the LDS-scratch form fits */
  CGMRES_HIP_FLAG_NO_WAVE = 16,        /* library's choice of mapping: never the wave mapping (small batches stay on wg) */
  CGMRES_HIP_FLAG_WAVE_FRESH_TRIG = 32, /* wave mapping and the row-parallel wg kernel of the pendulum: every Newton iteration
                                          evaluates sin/cos afresh (the path taken when an angle moves by more than the
                                          rotation range between iterations) */
  CGMRES_HIP_FLAG_WAVE_SERIAL_SWEEPS = 64, /* wave mapping: every mat-vec takes the serial state sweep (the fall-back of a Newton
                                          iteration that does not settle) */
  CGMRES_HIP_FLAG_SERIAL_STATE_SWEEP = 128 /* wg mapping, fp64: keep the serial state sweep in the Arnoldi loop instead of the
                                          row-parallel sweeps (tick_wg.hip.h: NWT), which the library takes where they apply
                                          (full LDS plan, k_max <= 12: pendulum 33 <= dv <= 53, semi-active damper dv <= 53) */
};

/* cgmres_hip_closed_loop_device advances up to this many consecutive ticks per kernel launch (the controller
 * state stays on chip between them); a tool that divides kernel time by ticks needs the number. */
#define CGMRES_HIP_TICKS_PER_LAUNCH 10

/* Run-time counterpart of the `static constexpr` block of a reference Model (e.g.
 * arm_type_inverted_pendulum/model.hpp:7-35), which Cgmres re-exports (cgmres.hpp:179-188). */
typedef struct cgmres_hip_config {
  int32_t abi_version; /* CGMRES_HIP_ABI_VERSION */
  int32_t model_id;    /* CGMRES_HIP_MODEL_* */
  int32_t dtype;       /* CGMRES_HIP_F64 / _F32 */
  int32_t batch;       /* number of controller instances B >= 1 */
  int32_t dv;          /* Model::dv    horizon stages */
  int32_t k_max;       /* Model::k_max GMRES iterations */
  int32_t device;      /* HIP device ordinal */
  int32_t variant;     /* kernel mapping: 0 = library's choice, 1 = "lane", 2 = "wg" (one workgroup per CU: everything
                          of 16 instances in LDS), 3 = "wg-lean" (half the LDS, two workgroups per CU),
                          4 = "wave" (one wavefront per controller, horizon recurrences as wave scans: the latency
                          mapping for batches smaller than the GPU); get_config returns the resolved value */
  int32_t flags;       /* CGMRES_HIP_FLAG_* (0 = library defaults) */
  int32_t reserved;    /* 0 */
  double tol;          /* Model::tol */
  double dt;           /* Model::dt   sampling period */
  double h;            /* Model::h    forward-difference step */
  double zeta;         /* Model::zeta */
  double Tf;           /* Model::Tf */
  double alpha;        /* Model::alpha */
  void* stream;        /* hipStream_t to launch on; NULL = the library creates its own */
} cgmres_hip_config;

typedef struct cgmres_hip_ctx* cgmres_hip_handle;

/* ---- registry -------------------------------------------------------------------------------- */
/* dims[0..4] = dim_x, dim_u, dim_p, shipped dv, shipped k_max; tuning[0..5] = dt, h, zeta, Tf, alpha, tol */
int cgmres_hip_model_info(int32_t model_id, int32_t dims[5], double tuning[6]);
/* Fills every field of *cfg from the registry (shipped dv / k_max / tuning), batch = 1, device = 0. */
int cgmres_hip_default_config(int32_t model_id, cgmres_hip_config* cfg);
/* Evaluates the DEVICE model functions at one probe point (fp64) so a host binding can fingerprint a
 * user Model class against the registry: out = [dxdt(dim_x) | dPhidx(dim_x) | dHdx(dim_x) | dHdu(dim_u)]. */
int cgmres_hip_model_probe(int32_t model_id, int32_t device, const double* x, const double* u, const double* p,
                           const double* lmd, double* out);
/* Registers a USER model: `plugin_path` is a shared object generated from a reference-style Model header
 * (static constexpr dim_x/.../tol and static dxdt/dPhidx/dHdx/dHdu/ddHduu, <example>/model.hpp:7-76) by
 * cgmres_cpp_amd/plugin.py (hipcc, gfx950).  *model_id receives an id >= CGMRES_HIP_MODEL_USER_BASE that every
 * other entry point accepts; such models run in fp64, on the "wg" mapping when their sizes fit its LDS plan
 * (the affine costate split is generated from the user's own dHdx / dHdu, csrc/user_model.hip.h) and on the "lane"
 * mapping otherwise.  Registering the same path twice returns the same id.  This is what a `Cgmres<Model>` facade binds for a Model that is not in the registry. */
int cgmres_hip_register_model(const char* plugin_path, int32_t* model_id);
/* ---- stand-alone Gmres with a caller-supplied operator: class Gmres, reference include/gmres.hpp:8-129 ------------ */
/* Registers a device OPERATOR: `plugin_path` is a shared object generated by cgmres_cpp_amd/plugin.py
 * (build_operator) from a header with `struct Op { static constexpr int len, n_params; static void Ax(double* Ax,
 * const double* x, const double* params); }` — the device build of a subclass's `Ax_func` (gmres.hpp:26).
 * *op_id receives an id for cgmres_hip_gmres_user; the same path twice returns the same id. */
int cgmres_hip_register_operator(const char* plugin_path, int32_t* op_id);
/* dims[0] = len, dims[1] = n_params of a registered operator */
int cgmres_hip_operator_info(int32_t op_id, int32_t dims[2]);
/* Gmres::gmres(x, b) (gmres.hpp:28-112) for `batch` independent systems A(params_i) x_i = b_i, all with the same
 * k_max and tol, on the GPU: x [batch][len] in/out (the warm start, :33), b [batch][len], params [batch][n_params]
 * (NULL when n_params = 0); n_ax / reason [batch] as in cgmres_hip_get_status (NULL skips).  Host pointers; blocks. */
int cgmres_hip_gmres_user(int32_t op_id, int32_t device, int32_t batch, int32_t k_max, double tol, const double* params,
                          double* x, const double* b, int32_t* n_ax, int32_t* reason);
/* Diagnostic: the device sin/cos the horizon sweeps use (fp64), evaluated at n host-supplied arguments. */
int cgmres_hip_selftest_sincos(int32_t device, const double* a, int32_t n, double* s, double* c);
const char* cgmres_hip_last_error(void);
int cgmres_hip_device_count(void);

/* ---- lifetime: Cgmres() / ~Cgmres(), cgmres.hpp:11-30 ------------------------------------------- */
/* State after create: t = 0, U = 0, dUdt = 0 (the reference leaves dUdt uninitialised, cgmres.hpp:14). */
int cgmres_hip_create(const cgmres_hip_config* cfg, cgmres_hip_handle* out);
int cgmres_hip_destroy(cgmres_hip_handle h);
int cgmres_hip_get_config(cgmres_hip_handle h, cgmres_hip_config* cfg);
/* Name of the mapping / kernel family the handle resolved to: "lane", "wave", "wg", "wg-lean", "wg+row-newton", "wg+row-scan",
 * "wg+parallel-costate", "wg+two-pass-costate", "wg-lean+two-pass-costate" (static
 * string; NULL on an invalid handle).  No counterpart in the reference: for logs and for tests that must know which
 * instantiation they exercised. */
const char* cgmres_hip_variant_name(cgmres_hip_handle h);

/* ---- setup: cgmres.hpp:36-76 --------------------------------------------------------------------- */
/* per_instance = 0: one vector broadcast to every instance; 1: [batch][...] */
int cgmres_hip_set_ptau(cgmres_hip_handle h, const void* ptau, int per_instance);         /* cgmres.hpp:36-39 */
int cgmres_hip_set_ptau_repeat(cgmres_hip_handle h, const void* p, int per_instance);     /* cgmres.hpp:41-49 */
int cgmres_hip_init_u0(cgmres_hip_handle h, const void* u0, int per_instance);            /* cgmres.hpp:51-59 */
/* Batched Newton on dH/du = 0 with the 3x3 / 6x6 partial-pivot solve (matrix.hpp:166-224), on the device.
 * u0 [batch][dim_u] in/out, x0 [batch][dim_x], p0 [batch][dim_p] (ignored when dim_p = 0).  cgmres.hpp:61-76 */
int cgmres_hip_init_u0_newton(cgmres_hip_handle h, void* u0, const void* x0, const void* p0, int32_t n_loop);

/* ---- the hot path: Cgmres::control, cgmres.hpp:78-110 -------------------------------------------- */
/* One tick for every instance: u [batch][dim_u] out, x [batch][dim_x] in.  Host pointers; blocks. */
int cgmres_hip_control(cgmres_hip_handle h, void* u, const void* x);
/* Same, device pointers, asynchronous on the handle's stream. */
int cgmres_hip_control_device(cgmres_hip_handle h, void* u_dev, const void* x_dev);
/* n_ticks of the example main loop (<example>/main.cpp:63-73) without leaving the GPU: control, then the
 * simulator's forward-Euler plant step x += dxdt(x,u)*dt.  x_dev in/out, u_dev = last tick's u.
 * Asynchronous on the handle's stream. */
int cgmres_hip_closed_loop_device(cgmres_hip_handle h, void* x_dev, void* u_dev, int32_t n_ticks);
/* The same loop with a TIME-VARYING reference: before tick k (k = 0 .. n_ticks-1) the parameter horizon of every
 * instance is replaced by ptau_seq[k], i.e. what a caller of the reference does by calling set_ptau (cgmres.hpp:36-39)
 * before every control().  ptau_seq_dev is a device pointer: [n_ticks][batch][dim_p*(dv+1)] when per_instance = 1,
 * [n_ticks][dim_p*(dv+1)] (one horizon broadcast to all instances) when per_instance = 0.  The ticks stay fused
 * (CGMRES_HIP_TICKS_PER_LAUNCH per launch); after the call the handle keeps the last tick's ptau, as set_ptau would.
 * With dim_p = 0 the sequence is ignored.  Asynchronous on the handle's stream. */
int cgmres_hip_closed_loop_device_ptau(cgmres_hip_handle h, void* x_dev, void* u_dev, int32_t n_ticks,
                                       const void* ptau_seq_dev, int per_instance);
int cgmres_hip_synchronize(cgmres_hip_handle h);

/* ---- state: the private members of Cgmres (cgmres.hpp:195-202) and Gmres (gmres.hpp:120-124) ------ */
int cgmres_hip_get_time(cgmres_hip_handle h, double* t);
int cgmres_hip_get_state(cgmres_hip_handle h, double* t, void* U, void* dUdt); /* [batch][dim_u*dv]; NULL skips */
int cgmres_hip_set_state(cgmres_hip_handle h, double t, const void* U, const void* dUdt);
/* n_ax[b] = Arnoldi mat-vecs executed inside the k loop of the last solve, reason[b] = CGMRES_HIP_EXIT_* */
int cgmres_hip_get_status(cgmres_hip_handle h, int32_t* n_ax, int32_t* reason);
/* H [batch][(k_max+1)*(k_max+1)] column-major ld k_max+1 (gmres.hpp:12,54), rho [batch][k_max+1],
 * g [batch][3*k_max] (gmres.hpp:14), V [batch][(k_max+1)*len] column-major (gmres.hpp:11); NULL skips */
int cgmres_hip_get_krylov(cgmres_hip_handle h, void* V, void* H, void* rho, void* g);

/* ---- white-box hooks used by the parity tests (private methods of the reference) ------------------ */
/* F_func(ret, U, x, t), cgmres.hpp:113-162, with the handle's ptau.  All [batch][...], host pointers. */
int cgmres_hip_F_func(cgmres_hip_handle h, void* ret, const void* U, const void* x, double t);
/* The statements of control() before the solve (cgmres.hpp:83-96): leaves x_dxh, F_dxh_h in the handle and
 * returns the GMRES right-hand side b [batch][len]. */
int cgmres_hip_prepare(cgmres_hip_handle h, void* b, const void* x);
/* Ax_func(out, v), cgmres.hpp:164-175; needs cgmres_hip_prepare first. */
int cgmres_hip_Ax_func(cgmres_hip_handle h, void* out, const void* v);
/* Gmres::gmres(x, b), gmres.hpp:28-112; needs cgmres_hip_prepare first. x [batch][len] in/out. */
int cgmres_hip_gmres(cgmres_hip_handle h, void* x, const void* b);

/* ---- sharding a batch over the GPUs of one node ------------------------------------- */
/* Contiguous, balanced slice [*lo, *hi) of `n` controller instances for shard `rank` of `world` (the first n % world
 * shards get one more) — the rule cgmres_cpp_amd/sharding.py and include/cgmres_batch.hpp (CgmresBatchSharded) share.
 * The reference's multiple_controller/main.cpp:89-110 owns its controllers one by one; this is the bookkeeping of
 * owning `n` of them on `world` devices.  Pure host arithmetic: needs no GPU. */
int cgmres_hip_shard_bounds(int32_t n, int32_t world, int32_t rank, int32_t* lo, int32_t* hi);

/* ---- measurement ---------------------------------------------------------------------------------- */
/* HIP events on the handle's stream around whatever is enqueued between the two calls. */
int cgmres_hip_timer_start(cgmres_hip_handle h);
int cgmres_hip_timer_stop(cgmres_hip_handle h, float* elapsed_ms); /* synchronises the stream */
/* Device allocator for callers without a HIP runtime of their own (tests, bench via ctypes). */
int cgmres_hip_malloc(cgmres_hip_handle h, void** dev_ptr, uint64_t bytes);
int cgmres_hip_free(cgmres_hip_handle h, void* dev_ptr);
int cgmres_hip_memcpy_h2d(cgmres_hip_handle h, void* dev_dst, const void* host_src, uint64_t bytes);
int cgmres_hip_memcpy_d2h(cgmres_hip_handle h, void* host_dst, const void* dev_src, uint64_t bytes);

#ifdef __cplusplus
}
#endif
#endif /* CGMRES_HIP_H_ */
