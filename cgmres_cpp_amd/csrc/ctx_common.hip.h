// Host-side plumbing shared by the kernel mappings: error reporting, the type-erased controller batch
// behind a cgmres_hip_handle, stream/event ownership and device allocations.
#pragma once
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/cgmres_hip.h"
#include "models.hip.h"

namespace cgm {

inline thread_local std::string g_err;

inline int fail(int code, const char* fmt, ...) {
  char buf[512];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof buf, fmt, ap);
  va_end(ap);
  g_err = buf;
  return code;
}

#define HIP_TRY(expr)                                                                                        \
  do {                                                                                                       \
    hipError_t e_ = (expr);                                                                                  \
    if (e_ != hipSuccess)                                                                                    \
      return ::cgm::fail(e_ == hipErrorOutOfMemory ? CGMRES_HIP_ENOMEM : CGMRES_HIP_ERUNTIME, "%s: %s (%s:%d)", \
                         #expr, hipGetErrorString(e_), __FILE__, __LINE__);                                  \
  } while (0)

}  // namespace cgm

// Type-erased controller batch: what a cgmres_hip_handle points to.
struct cgmres_hip_ctx {
  cgmres_hip_config cfg{};
  int nx = 0, nu = 0, np = 0, L = 0;
  hipStream_t stream = nullptr;
  bool own_stream = false;
  hipEvent_t ev0 = nullptr, ev1 = nullptr;
  std::vector<void*> owned;

  virtual ~cgmres_hip_ctx() {
    (void)hipSetDevice(cfg.device);
    if (stream) (void)hipStreamSynchronize(stream);
    for (void* p : owned) (void)hipFree(p);
    if (ev0) (void)hipEventDestroy(ev0);
    if (ev1) (void)hipEventDestroy(ev1);
    if (own_stream && stream) (void)hipStreamDestroy(stream);
  }
  int init_common() {
    HIP_TRY(hipSetDevice(cfg.device));
    if (cfg.stream) {
      stream = static_cast<hipStream_t>(cfg.stream);
    } else {
      HIP_TRY(hipStreamCreateWithFlags(&stream, hipStreamNonBlocking));
      own_stream = true;
    }
    HIP_TRY(hipEventCreate(&ev0));
    HIP_TRY(hipEventCreate(&ev1));
    return 0;
  }
  template <class Q>
  int dalloc(Q** p, size_t n) {  // zero-filled, freed with the context
    void* q = nullptr;
    HIP_TRY(hipMalloc(&q, (n ? n : 1) * sizeof(Q)));
    HIP_TRY(hipMemsetAsync(q, 0, (n ? n : 1) * sizeof(Q), stream));
    owned.push_back(q);
    *p = static_cast<Q*>(q);
    return 0;
  }
  // staging buffer that only grows; kept in `owned`
  template <class Q>
  int grow(Q** buf, size_t* have, size_t need) {
    if (*have >= need) return 0;
    HIP_TRY(hipStreamSynchronize(stream));
    void* q = nullptr;
    HIP_TRY(hipMalloc(&q, need * sizeof(Q)));
    if (*buf) {
      for (auto& o : owned)
        if (o == *buf) o = q;
      HIP_TRY(hipFree(*buf));
    } else {
      owned.push_back(q);
    }
    *buf = static_cast<Q*>(q);
    *have = need;
    return 0;
  }

  virtual const char* variant_name() const = 0;
  virtual int init() = 0;
  virtual int set_ptau(const void*, int per_instance, bool repeat) = 0;
  virtual int init_u0(const void*, int per_instance) = 0;
  virtual int init_u0_newton(void*, const void*, const void*, int) = 0;
  virtual int control_host(void*, const void*) = 0;
  virtual int control_device(void*, const void*, void* x_next) = 0;
  virtual int closed_loop(void*, void*, int, const void* ptau_seq = nullptr, int per_instance = 0) = 0;
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
