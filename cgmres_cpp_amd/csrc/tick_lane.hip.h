// Variant 1 ("lane"): one wavefront lane per controller instance, whole tick in one kernel.
//
// Every per-instance vector lives in HBM as [element][ldb] with the batch index innermost, so a
// wave-level load of "element e of 64 consecutive instances" is one coalesced 512-byte (fp64) access.
// Each lane walks its own instance through exactly the reference's statement order
// (cgmres.hpp:78-175, gmres.hpp:28-112): sequential dot products, modified Gram-Schmidt in order,
// per-lane early exit — so this variant is the closest GPU rendering of the reference's rounding,
// and the baseline the faster mappings are checked against.
//
// Per-stage temporaries: the state trajectory x(0..dv-1) and the trig values the costate sweep reuses
// go to an HBM scratch [stage][component][ldb] (L2 resident: 4096 x 2.8 KB = 11.6 MB); the costate
// itself never leaves registers because dH/du(i) is evaluated inside the backward loop.
#pragma once
#include <type_traits>
#include <hip/hip_runtime.h>

#include <cfloat>

#include "models.hip.h"

namespace cgm {

template <class T>
struct TickParams {
  int B, ldb, dv, kmax, L;
  T h, dt, tol, inv_h, one_m_zh;  // inv_h = 1/h, one_m_zh = 1 - zeta*h (cgmres.hpp:94-96)
  T dtau_h, dtau_0;               // get_dtau(t+h), get_dtau(t) (cgmres.hpp:32-34), computed on the host
  // element-major state, [n][ldb]
  T *U, *dUdt, *Fh, *bvec, *V, *H, *g, *rho, *xdxh, *ptau, *traj, *trig;
  int *n_ax, *reason;
  // instance-major I/O, [B][dim]
  const T* x_in;
  T* u_out;
  T* x_next;  // non-null: also write the plant's forward-Euler step x + dxdt(x,u)*dt here
};

enum FOut { F_PLAIN = 0, F_RHS = 1, F_AX = 2 };

// Optimality residual F(U [+ h v], x, t) — cgmres.hpp:113-162 (+ :168-174 when PERTURB / F_AX).
//   F_PLAIN: out = F                                   (cgmres.hpp:88)
//   F_RHS  : out = (F*(1-zeta*h) - Fh) * (1/h)         (cgmres.hpp:91-96)
//   F_AX   : out = (F - Fh) * (1/h)                    (cgmres.hpp:173-174)
// The state equation of stage `stage`.  The built-in models do not read the time-varying parameter p in dxdt; a
// user model compiled through the plugin path (user_model.hip.h) may (Model::dxdt(ret, x, u, p), */model.hpp:36).
template <class M, class T, class = void>
struct DxdtUsesP : std::false_type {};
template <class M, class T>
struct DxdtUsesP<M, T, std::enable_if_t<M::DXDT_USES_P>> : std::true_type {};
template <class M, class T>
__device__ __forceinline__ void state_eq(T* f, const T* x, const T* u, T* tr, const typename M::Math& mc,
                                         const T* ptau, size_t ld, int stage) {
  if constexpr (DxdtUsesP<M, T>::value) {
    T p[M::NP > 0 ? M::NP : 1];
#pragma unroll
    for (int j = 0; j < M::NP; ++j) p[j] = ptau[size_t(stage * M::NP + j) * ld];
    M::dxdt_p(f, x, u, p);
  } else {
    M::dxdt(f, x, u, tr, mc);
  }
}

template <class M, class T, bool PERTURB, int MODE>
__device__ __forceinline__ void f_eval_lane(const TickParams<T>& P, size_t ld, const T* __restrict__ U,
                                            const T* __restrict__ v, const T* x0, T dtau,
                                            T* __restrict__ out, const T* __restrict__ Fh,
                                            T* __restrict__ traj, T* __restrict__ trig,
                                            const T* __restrict__ ptau) {
  constexpr int NX = M::NX, NU = M::NU, NP = M::NP, NC = M::NC;
  typename M::Math mc;
  mc.init();
  T xs[NX];
#pragma unroll
  for (int i = 0; i < NX; ++i) xs[i] = x0[i];
  const int dv = P.dv;
  // state sweep, cgmres.hpp:132-140
  for (int s = 0; s < dv; ++s) {
    T u[M::NU_DYN], f[NX], tr[NC > 0 ? NC : 1];
#pragma unroll
    for (int j = 0; j < M::NU_DYN; ++j) {
      T uj = U[size_t(s * NU + j) * ld];
      if (PERTURB) uj = v[size_t(s * NU + j) * ld] * P.h + uj;  // cgmres.hpp:168-169
      u[j] = uj;
    }
#pragma unroll
    for (int i = 0; i < NX; ++i) traj[size_t(s * NX + i) * ld] = xs[i];
    state_eq<M, T>(f, xs, u, tr, mc, ptau, ld, s);
#pragma unroll
    for (int c = 0; c < NC; ++c) trig[size_t(s * NC + c) * ld] = tr[c];
#pragma unroll
    for (int i = 0; i < NX; ++i) xs[i] = f[i] * dtau + xs[i];
  }
  // terminal costate, cgmres.hpp:145
  T l[NX], p[NP > 0 ? NP : 1];
#pragma unroll
  for (int j = 0; j < NP; ++j) p[j] = ptau[size_t(dv * NP + j) * ld];
  M::dPhidx(l, xs, p);
  // costate sweep fused with dH/du, cgmres.hpp:146-161
  for (int s = dv - 1; s >= 0; --s) {
    T u[NU], tr[NC > 0 ? NC : 1], Fs[NU], gx[NX];
#pragma unroll
    for (int i = 0; i < NX; ++i) xs[i] = traj[size_t(s * NX + i) * ld];
#pragma unroll
    for (int c = 0; c < NC; ++c) tr[c] = trig[size_t(s * NC + c) * ld];
#pragma unroll
    for (int j = 0; j < NU; ++j) {
      T uj = U[size_t(s * NU + j) * ld];
      if (PERTURB) uj = v[size_t(s * NU + j) * ld] * P.h + uj;
      u[j] = uj;
    }
#pragma unroll
    for (int j = 0; j < NP; ++j) p[j] = ptau[size_t(s * NP + j) * ld];
    M::dHdu(Fs, xs, u, p, l, tr);
#pragma unroll
    for (int j = 0; j < NU; ++j) {
      const size_t e = size_t(s * NU + j) * ld;
      T r = Fs[j];
      if (MODE == F_RHS) r = (r * P.one_m_zh - Fh[e]) * P.inv_h;
      if (MODE == F_AX) r = (r - Fh[e]) * P.inv_h;
      out[e] = r;
    }
    M::dHdx(gx, xs, u, p, l, tr);
#pragma unroll
    for (int i = 0; i < NX; ++i) l[i] = gx[i] * dtau + l[i];
  }
}

// Lane-local views of the element-major arrays of one instance.
template <class T>
struct LaneView {
  T *U, *dUdt, *Fh, *bvec, *V, *H, *g, *rho, *xdxh, *ptau, *traj, *trig;
  __device__ LaneView(const TickParams<T>& P, int b)
      : U(P.U + b), dUdt(P.dUdt + b), Fh(P.Fh + b), bvec(P.bvec + b), V(P.V + b), H(P.H + b), g(P.g + b),
        rho(P.rho + b), xdxh(P.xdxh + b), ptau(P.ptau + b), traj(P.traj + b), trig(P.trig + b) {}
};

// control() up to the solve — cgmres.hpp:83-96.  Leaves x_dxh, F_dxh_h and b in HBM.
template <class M, class T>
__device__ __forceinline__ void prepare_lane(const TickParams<T>& P, size_t ld, const LaneView<T>& A, const T* x) {
  constexpr int NX = M::NX, NU = M::NU;
  T u0[NU], f[NX], xh[NX], tr[M::NC > 0 ? M::NC : 1];
  typename M::Math mc;
  mc.init();
#pragma unroll
  for (int j = 0; j < NU; ++j) u0[j] = A.U[size_t(j) * ld];
  state_eq<M, T>(f, x, u0, tr, mc, A.ptau, ld, 0);  // cgmres.hpp:83
#pragma unroll
  for (int i = 0; i < NX; ++i) {
    xh[i] = f[i] * P.h + x[i];  // :84-85
    A.xdxh[size_t(i) * ld] = xh[i];
  }
  f_eval_lane<M, T, false, F_PLAIN>(P, ld, A.U, nullptr, xh, P.dtau_h, A.Fh, nullptr, A.traj, A.trig, A.ptau);  // :88
  f_eval_lane<M, T, false, F_RHS>(P, ld, A.U, nullptr, x, P.dtau_0, A.bvec, A.Fh, A.traj, A.trig, A.ptau);    // :91-96
}

// Ax_func — cgmres.hpp:164-175
template <class M, class T>
__device__ __forceinline__ void ax_lane(const TickParams<T>& P, size_t ld, const LaneView<T>& A, T* out,
                                        const T* v) {
  T xh[M::NX];
#pragma unroll
  for (int i = 0; i < M::NX; ++i) xh[i] = A.xdxh[size_t(i) * ld];
  f_eval_lane<M, T, true, F_AX>(P, ld, A.U, v, xh, P.dtau_h, out, A.Fh, A.traj, A.trig, A.ptau);
}

template <class T>
__device__ __forceinline__ T sqrt_t(T a);
template <>
__device__ __forceinline__ double sqrt_t<double>(double a) {
  return ::sqrt(a);
}
template <>
__device__ __forceinline__ float sqrt_t<float>(float a) {
  return ::sqrtf(a);
}
template <class T>
__device__ __forceinline__ T abs_t(T a) {
  return a < T(0) ? -a : a;
}
// neither NaN nor +-Inf (one v_cmp_class on the hardware)
__device__ __forceinline__ bool finite_t(double a) { return __builtin_isfinite(a); }
__device__ __forceinline__ bool finite_t(float a) { return __builtin_isfinite(a); }
template <class T>
__device__ __forceinline__ T quiet_nan();
template <>
__device__ __forceinline__ double quiet_nan<double>() { return __builtin_nan(""); }
template <>
__device__ __forceinline__ float quiet_nan<float>() { return __builtin_nanf(""); }
// CGMRES_HIP_EXIT_NONFINITE: a non-finite residual / Arnoldi norm.  The reference tests for neither; every comparison
// with a NaN is false there, the remaining iterations run on NaNs and the solution vector ends up NaN (include/gmres.hpp:
// 39-41, 63-65, 93-95 all fall through).  Here the instance stops at once and its solution vector is set to NaN.
template <class T>
__device__ __forceinline__ int poison_nonfinite(T* x, int L, size_t ld) {
  for (int e = 0; e < L; ++e) x[size_t(e) * ld] = quiet_nan<T>();
  return 4;
}

// Gmres::gmres — gmres.hpp:28-112, per lane, sequential reductions in index order, around ANY operator:
// `ax(out, v)` is the reference's pure virtual Ax_func (gmres.hpp:26) — the controller's forward-difference product
// (gmres_lane below) or a caller-supplied device functor (user_operator.hip.h).  Element-major vectors: element e of this
// lane's instance at [e*ld]; Krylov arrays V[(kmax+1)*L], H[(kmax+1)^2] column-major ld kmax+1, rho[kmax+1], g[3*kmax].
// Returns the exit reason; n_ax = Arnoldi mat-vecs executed inside the k loop.
template <class T, class AxF>
__device__ __forceinline__ int gmres_lane_core(int L, int kmax, T tol, size_t ld, T* V, T* Hm, T* rho, T* g, T* x,
                                               const T* b, int* n_ax_out, AxF&& ax) {
  const int ldh = kmax + 1;
  *n_ax_out = 0;
  // r0 = b - A x0, rho = ||r0||   gmres.hpp:33-37
  ax(V, x);
  T ss = 0;
  for (int e = 0; e < L; ++e) {
    const T r = b[size_t(e) * ld] - V[size_t(e) * ld];
    V[size_t(e) * ld] = r;
    ss += r * r;
  }
  const T rho0 = sqrt_t<T>(ss);
  rho[0] = rho0;
  if (!finite_t(rho0)) return poison_nonfinite(x, L, ld);  // (no such test in the reference: NaNs propagate there)
  if (rho0 < tol) return 2;  // gmres.hpp:39-41
  {
    const T inv = T(1.0) / rho0;  // gmres.hpp:44 via matrix.hpp:122-128
    for (int e = 0; e < L; ++e) V[size_t(e) * ld] = V[size_t(e) * ld] * inv;
  }
  int k, reason = 0;
  for (k = 0; k < kmax; ++k) {  // gmres.hpp:46
    T* w = V + size_t(L) * (k + 1) * ld;
    ax(w, V + size_t(L) * k * ld);  // :48
    *n_ax_out = k + 1;
    T* Hk = Hm + size_t(ldh) * k * ld;
    for (int i = 0; i <= k; ++i) {  // :52-58 modified Gram-Schmidt
      const T* vi = V + size_t(L) * i * ld;
      T hik = 0;
      for (int e = 0; e < L; ++e) hik += vi[size_t(e) * ld] * w[size_t(e) * ld];
      Hk[size_t(i) * ld] = hik;
      for (int e = 0; e < L; ++e) w[size_t(e) * ld] = w[size_t(e) * ld] - vi[size_t(e) * ld] * hik;
    }
    T nn = 0;
    for (int e = 0; e < L; ++e) nn += w[size_t(e) * ld] * w[size_t(e) * ld];
    const T hn = sqrt_t<T>(nn);  // :60
    Hk[size_t(k + 1) * ld] = hn;
    if (!finite_t(hn)) return poison_nonfinite(x, L, ld);
    if (abs_t(hn) < T(DBL_EPSILON)) return 3;  // :63-65
    {
      const T inv = T(1.0) / hn;  // :67
      for (int e = 0; e < L; ++e) w[size_t(e) * ld] = w[size_t(e) * ld] * inv;
    }
    for (int i = 0; i < k; ++i) {  // :71-77 stored reflectors
      const T g0 = g[size_t(3 * i) * ld], g1 = g[size_t(3 * i + 1) * ld], g2 = g[size_t(3 * i + 2) * ld];
      const T a = Hk[size_t(i) * ld], c = Hk[size_t(i + 1) * ld];
      const T beta = (g0 * a + g1 * c) * g2;
      Hk[size_t(i) * ld] = a - beta * g0;
      Hk[size_t(i + 1) * ld] = c - beta * g1;
    }
    {  // :78-90 new reflector and residual rotation
      const T a = Hk[size_t(k) * ld], c = Hk[size_t(k + 1) * ld];
      const T sigma = -(a < T(0.0) ? T(-1.0) : T(1.0)) * sqrt_t<T>(a * a + c * c);
      const T g0 = a - sigma, g1 = c;
      const T g2 = T(2.0) / (g0 * g0 + g1 * g1);
      g[size_t(3 * k) * ld] = g0;
      g[size_t(3 * k + 1) * ld] = g1;
      g[size_t(3 * k + 2) * ld] = g2;
      Hk[size_t(k) * ld] = sigma;
      Hk[size_t(k + 1) * ld] = T(0.0);
      const T ek = rho[size_t(k) * ld];
      const T beta = g0 * ek * g2;
      rho[size_t(k) * ld] = ek - beta * g0;
      const T en = -beta * g1;
      rho[size_t(k + 1) * ld] = en;
      if (abs_t(en) < tol) {  // :93-95, k not incremented
        reason = 1;
        break;
      }
    }
  }
  // back substitution on the leading k x k block, gmres.hpp:100-107
  for (int i = k - 1; i >= 0; --i) {
    T ei = rho[size_t(i) * ld];
    for (int j = k - 1; j > i; --j) ei -= Hm[size_t(ldh * j + i) * ld] * rho[size_t(j) * ld];
    rho[size_t(i) * ld] = ei / Hm[size_t(ldh * i + i) * ld];
  }
  // x += V[:,0:k] y, gmres.hpp:110-111 (accumulated j-ascending from 0 like matrix.hpp:82-91)
  for (int e = 0; e < L; ++e) {
    T acc = T(0.0);
    for (int j = 0; j < k; ++j) acc += V[(size_t(L) * j + e) * ld] * rho[size_t(j) * ld];
    x[size_t(e) * ld] = x[size_t(e) * ld] + acc;
  }
  return reason;
}

// ... with the controller's forward-difference operator (cgmres.hpp:164-175)
template <class M, class T>
__device__ __forceinline__ int gmres_lane(const TickParams<T>& P, size_t ld, const LaneView<T>& A, T* x,
                                          const T* b, int* n_ax_out) {
  return gmres_lane_core<T>(P.L, P.kmax, P.tol, ld, A.V, A.H, A.rho, A.g, x, b, n_ax_out,
                            [&](T* out, const T* v) { ax_lane<M, T>(P, ld, A, out, v); });
}

// ---- kernels ------------------------------------------------------------------------------------
template <class M, class T>
__global__ __launch_bounds__(64) void tick_lane_kernel(TickParams<T> P) {
  const int b = blockIdx.x * 64 + threadIdx.x;
  if (b >= P.B) return;
  const size_t ld = P.ldb;
  const LaneView<T> A(P, b);
  T x[M::NX];
#pragma unroll
  for (int i = 0; i < M::NX; ++i) x[i] = P.x_in[size_t(b) * M::NX + i];
  prepare_lane<M, T>(P, ld, A, x);
  int n_ax = 0;
  const int reason = gmres_lane<M, T>(P, ld, A, A.dUdt, A.bvec, &n_ax);  // cgmres.hpp:99
  P.n_ax[b] = n_ax;
  P.reason[b] = reason;
  for (int e = 0; e < P.L; ++e) A.U[size_t(e) * ld] = A.U[size_t(e) * ld] + A.dUdt[size_t(e) * ld] * P.dt;  // :102-103
  T u[M::NU];
#pragma unroll
  for (int j = 0; j < M::NU; ++j) {
    u[j] = A.U[size_t(j) * ld];  // :109
    P.u_out[size_t(b) * M::NU + j] = u[j];
  }
  if (P.x_next) {  // */main.cpp:71-73 plant step (Simulator::dxdt has the Model's state equation)
    T f[M::NX], tr[M::NC > 0 ? M::NC : 1];
    typename M::Math mc;
    mc.init();
    state_eq<M, T>(f, x, u, tr, mc, A.ptau, ld, 0);
#pragma unroll
    for (int i = 0; i < M::NX; ++i) P.x_next[size_t(b) * M::NX + i] = x[i] + f[i] * P.dt;
  }
}

// White-box hooks (tests): F_func, the pre-solve part of control, Ax_func, gmres on their own.
template <class M, class T>
__global__ __launch_bounds__(64) void hook_F_kernel(TickParams<T> P, const T* Uin, T* ret, T dtau) {
  const int b = blockIdx.x * 64 + threadIdx.x;
  if (b >= P.B) return;
  const LaneView<T> A(P, b);
  T x[M::NX];
#pragma unroll
  for (int i = 0; i < M::NX; ++i) x[i] = P.x_in[size_t(b) * M::NX + i];
  f_eval_lane<M, T, false, F_PLAIN>(P, P.ldb, Uin + b, nullptr, x, dtau, ret + b, nullptr, A.traj, A.trig, A.ptau);
}
template <class M, class T>
__global__ __launch_bounds__(64) void hook_prepare_kernel(TickParams<T> P) {
  const int b = blockIdx.x * 64 + threadIdx.x;
  if (b >= P.B) return;
  const LaneView<T> A(P, b);
  T x[M::NX];
#pragma unroll
  for (int i = 0; i < M::NX; ++i) x[i] = P.x_in[size_t(b) * M::NX + i];
  prepare_lane<M, T>(P, P.ldb, A, x);
}
template <class M, class T>
__global__ __launch_bounds__(64) void hook_Ax_kernel(TickParams<T> P, const T* v, T* out) {
  const int b = blockIdx.x * 64 + threadIdx.x;
  if (b >= P.B) return;
  const LaneView<T> A(P, b);
  ax_lane<M, T>(P, P.ldb, A, out + b, v + b);
}
template <class M, class T>
__global__ __launch_bounds__(64) void hook_gmres_kernel(TickParams<T> P, T* x, const T* bv) {
  const int b = blockIdx.x * 64 + threadIdx.x;
  if (b >= P.B) return;
  const LaneView<T> A(P, b);
  int n_ax = 0;
  P.reason[b] = gmres_lane<M, T>(P, P.ldb, A, x + b, bv + b, &n_ax);
  P.n_ax[b] = n_ax;
}

}  // namespace cgm
