// Host side of the "lane" mapping (tick_lane.hip.h): element-major HBM state, one kernel per tick.
#pragma once
#include "ctx_common.hip.h"
#include "tick_lane.hip.h"
#include "util_kernels.hip.h"

namespace cgm {

template <class M, class T>
struct CtxLane final : cgmres_hip_ctx {
  TickParams<T> P{};
  T t = T(0);
  int ldb = 0;
  T *stage_a = nullptr, *stage_b = nullptr;  // device staging, grown on demand
  size_t stage_a_n = 0, stage_b_n = 0;
  T *x_dev = nullptr, *u_dev = nullptr;  // [B][nx], [B][nu] staging for the host-pointer entry points

  const char* variant_name() const override { return "lane"; }

  int init() override {
    if (int rc = init_common()) return rc;
    cfg.variant = 1;
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
      replicate_stages<T><<<grid, 256, 0, stream>>>(dst, stage_a, cfg.batch, ldb, n, stages_rep, !per_instance);
    else
      to_element_major<T><<<grid, 256, 0, stream>>>(dst, stage_a, cfg.batch, ldb, n, !per_instance);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipStreamSynchronize(stream));  // the host buffer may be reused by the caller right away
    return 0;
  }
  int download(void* dst, const T* src, int n) {
    if (!dst) return 0;
    const size_t cnt = size_t(cfg.batch) * n;
    if (int rc = grow(&stage_a, &stage_a_n, cnt)) return rc;
    dim3 grid((cfg.batch + 255) / 256, n);
    to_instance_major<T><<<grid, 256, 0, stream>>>(stage_a, src, cfg.batch, ldb, n);
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
    newton_u0_kernel<M, T><<<dim3((B + 63) / 64), 64, 0, stream>>>(du, dx, dp, cfg.batch, n_loop);
    HIP_TRY(hipGetLastError());
    dim3 grid((cfg.batch + 255) / 256, nu * cfg.dv);
    replicate_stages<T><<<grid, 256, 0, stream>>>(P.U, du, cfg.batch, ldb, nu, cfg.dv, 0);  // cgmres.hpp:75
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpyAsync(u0, du, B * nu * sizeof(T), hipMemcpyDeviceToHost, stream));
    HIP_TRY(hipStreamSynchronize(stream));
    return 0;
  }

  int launch_tick(T* u_out, const T* x_in, T* x_next) {
    P.x_in = x_in, P.u_out = u_out, P.x_next = x_next;
    P.dtau_h = dtau_of(t + P.h);  // cgmres.hpp:88
    P.dtau_0 = dtau_of(t);        // cgmres.hpp:91
    tick_lane_kernel<M, T><<<lane_grid(), 64, 0, stream>>>(P);
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
  int closed_loop(void* x, void* u, int n_ticks, const void* ptau_seq, int per_instance) override {
    HIP_TRY(hipSetDevice(cfg.device));
    if (!u || !x) return fail(CGMRES_HIP_EINVAL, "closed_loop: null pointer");
    const int all = np * (cfg.dv + 1);
    const size_t per_tick = size_t(per_instance ? cfg.batch : 1) * all;
    for (int i = 0; i < n_ticks; ++i) {
      if (ptau_seq && all) {  // set_ptau before this tick (cgmres.hpp:36-39), device to device
        dim3 grid((cfg.batch + 255) / 256, all);
        to_element_major<T><<<grid, 256, 0, stream>>>(P.ptau, static_cast<const T*>(ptau_seq) + i * per_tick, cfg.batch,
                                                       ldb, all, !per_instance);
        HIP_TRY(hipGetLastError());
      }
      if (int rc = launch_tick(static_cast<T*>(u), static_cast<const T*>(x), static_cast<T*>(x))) return rc;
    }
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
    hook_F_kernel<M, T><<<lane_grid(), 64, 0, stream>>>(P, Uem, Rem, dtau_of(T(tt)));
    HIP_TRY(hipGetLastError());
    return download(ret, Rem, L);
  }
  int hook_prepare(void* b, const void* x) override {
    HIP_TRY(hipSetDevice(cfg.device));
    HIP_TRY(hipMemcpyAsync(x_dev, x, size_t(cfg.batch) * nx * sizeof(T), hipMemcpyHostToDevice, stream));
    P.x_in = x_dev;
    P.dtau_h = dtau_of(t + P.h);
    P.dtau_0 = dtau_of(t);
    hook_prepare_kernel<M, T><<<lane_grid(), 64, 0, stream>>>(P);
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
    hook_Ax_kernel<M, T><<<lane_grid(), 64, 0, stream>>>(P, Vem, Oem);
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
    hook_gmres_kernel<M, T><<<lane_grid(), 64, 0, stream>>>(P, Xem, Bem);
    HIP_TRY(hipGetLastError());
    return download(x, Xem, L);
  }
};

}  // namespace cgm
