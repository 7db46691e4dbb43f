// Variant 4 ("wave"): ONE WAVEFRONT PER CONTROLLER — the latency mapping.
//
// The "wg" mapping (tick_wg.hip.h) gives an instance 16 lanes and walks the horizon recurrences of cgmres.hpp:132-153
// serially on one wave for 16 instances at a time: a tick is a chain of ~30 k instruction issues that does not get
// shorter when there are fewer instances than the GPU has lanes for (512 controllers per GPU: 116 us/tick, 224 of 256
// CUs idle).  Here a controller owns all 64 lanes of a wave and the horizon lives ACROSS the lanes:
//   * lane s carries stage s: its controls U[s*dim_u .. ], every length-L vector of the solver as dim_u elements per
//     lane (the reference's stage-major layout, cgmres.hpp:13, read 24 contiguous bytes per lane), the stage's state,
//     trig values, parameters and costate coefficients — all in registers.  No stage table, no row arrays, no barrier;
//   * the recurrences over the stage index become SCANS over the lanes (wave_scan.hip.h: six DPP steps):
//       - x0, x2 of the pendulum obey a linear recurrence driven by u0 (model.hpp:38,40): a geometric scan and a prefix sum;
//       - x1, x3 are nonlinear (model.hpp:39,41): Newton's method on the whole trajectory at once — evaluate the stage
//         function and its Jacobian at the current guess on every lane, solve the linearised recurrence for the
//         correction with a scan of 2 x 2 affine maps, repeat until the correction is below 1e-10 (quadratic
//         convergence: the next one would be < 1e-18).  The guess is the unperturbed trajectory of the same tick
//         (F(U, x+hf, t+h), cgmres.hpp:88): a forward-difference direction moves the controls by h*|v| <= 2e-3.  The
//         trig values of a guess are ROTATED from the base trajectory's through the (small) angle difference;
//       - the costate recurrence (cgmres.hpp:146-153) is linear with the stage coefficients of models.hip.h and splits
//         for the pendulum into a 2 x 2 affine scan (l1, l3), a prefix sum (l0) and a geometric scan (l2), on the
//         MIRRORED lanes (stage dv-1-m on lane m: one crossbar gather of the six coefficients, one of dF back);
//     every scan result equals the serial recurrence up to rounding (tests bound it against the oracle);
//   * the three sweeps in front of the solve (cgmres.hpp:83-96 and the residual of the warm start) start from states the
//     previous tick does not bracket, so their state sweeps run serially — concurrently on three DPP quads with the quad
//     stage of the wg mapping (PendulumDev::quad_stage) — through a small LDS table;
//   * the Krylov basis (k_max + 1 vectors x 3 elements per lane) stays in REGISTERS; Gram-Schmidt is 3 + 3 fused
//     multiply-adds around a wave-wide sum.  Every branch of the solve (gmres.hpp:39-41, 63-65, 93-95) is wave-uniform:
//     an instance that converges early simply leaves the loop (no masking, no waiting for workgroup mates).
// Per tick ~15 k instructions on the one wave instead of ~30 k on the critical wave of the wg mapping.
// Statement order per instance follows cgmres.hpp:78-175 / gmres.hpp:28-112; sums associate as scans / butterflies.
#pragma once
#include <hip/hip_runtime.h>

#include <cfloat>
#include <type_traits>

#include "tick_wg.hip.h"  // WgParams, helpers
#include "wave_scan.hip.h"

namespace cgm {

// which models have the scans above
template <class M>
struct WaveOps : std::false_type {};

template <class T>
struct WaveOps<PendulumDev<T>> : std::true_type {
  using M = PendulumDev<T>;
  // sin/cos of (base angle + dl) from the base pair, |dl| <= sqrt(rot_zmax) (Taylor polynomials of the wg mapping's
  // rotation stage, MathCtx::rot_sin / rot_cos)
  template <class MC>
  static __device__ __forceinline__ void rotate(T sb, T cb, T dl, T* s, T* c, const MC& mc) {
    constexpr int NRS = MC::NRS, NRC = MC::NRC;
    const T z = dl * dl;
    T ps = mc.rot_sin(NRS - 1), pc = mc.rot_cos(NRC - 1);
#pragma unroll
    for (int i = NRS - 2; i >= 0; --i) ps = fma_t(z, ps, mc.rot_sin(i));
#pragma unroll
    for (int i = NRC - 2; i >= 0; --i) pc = fma_t(z, pc, mc.rot_cos(i));
    const T sn = fma_t(z * dl, ps, dl);  // sin dl
    const T cm1 = z * pc;                // cos dl - 1
    *s = sb + fma_t(sb, cm1, cb * sn);
    *c = cb + fma_t(cb, cm1, -(sb * sn));
  }
};

// LDS of one wave: the table of the serial sweeps, two control rows for them, the small Krylov arrays
template <class M, class T>
struct WaveLds {
  static constexpr int TAB_W = 2 * M::NX;  // per (sweep, stage): NX pairs (state component, trig value)
  static __host__ __device__ int row_len(int dv) { return (dv * M::NU + 2) & ~1; }
  static __host__ __device__ int pitch_H(int kmax) { return ((kmax * (kmax + 1)) / 2 + 3) & ~1; }
  static __host__ __device__ size_t count_T(int dv, int kmax) {
    return size_t(3) * (dv + 1) * TAB_W + 2 * row_len(dv) + pitch_H(kmax) + (kmax + 2) + 3 * kmax + 6;
  }
  static __host__ __device__ size_t bytes(int dv, int kmax, int waves) {
    return ((count_T(dv, kmax) * sizeof(T) + 15) & ~size_t(15)) * waves;
  }
};

constexpr int WAVE_NEWTON_MAX = 8;

template <class M, class T, int KM, int WPB>
__global__ __launch_bounds__(64 * WPB) void tick_wave_kernel(WgParams<T> P) {
  static_assert(WaveOps<M>::value, "model without wave scans");
  static_assert(std::is_same<T, double>::value, "the Newton thresholds are set for fp64");
  using W = WaveOps<M>;
  using Lds = WaveLds<M, T>;
  constexpr int NX = M::NX, NU = M::NU, NP = M::NP;
  static_assert(NX == 4 && NU == 3 && NP == 2, "pendulum shapes");
  extern __shared__ __align__(16) unsigned char smem[];
  const int lane = threadIdx.x & 63;
  const int wv = __builtin_amdgcn_readfirstlane(int(threadIdx.x) >> 6);
  const int b = blockIdx.x * WPB + wv;  // this wave's controller
  if (b >= P.B) return;                 // (no barrier anywhere in this kernel)
  const int dv = P.dv, kmax = P.kmax, k1 = kmax + 1;
  // ---- LDS of this wave
  T* const base = reinterpret_cast<T*>(smem + size_t(wv) * Lds::bytes(dv, kmax, 1));
  T* const tab = base;                                  // [3][dv+1][TAB_W]
  T* const wrow = tab + 3 * (dv + 1) * Lds::TAB_W;      // [2][row_len]
  const int rlen = Lds::row_len(dv);
  T* const Hi = wrow + 2 * rlen;                        // compact Hessenberg: column k = rows 0..k at k(k+1)/2
  T* const rhoi = Hi + Lds::pitch_H(kmax);
  T* const gi = rhoi + (kmax + 2);
  auto hoff = [](int k) { return (k * (k + 1)) >> 1; };
  struct alignas(2 * sizeof(T)) WPair {
    T a, b;
  };
  auto wave_fence = [] { __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); };  // compiler ordering of LDS traffic

  typename M::template MathFor<false> mc;
  mc.init();
  const bool in_hor = lane < dv;    // the lane has a stage with controls
  const bool in_traj = lane <= dv;  // ... or the terminal stage
  const int msrc = in_hor ? dv - 1 - lane : lane;  // mirrored lane of the costate scans (an involution on [0, dv))

  // ---- controller state of this instance
  T U[NU], du[NU], Fh[NU], p[NP], xs[NX];
  {
    const T* Ug = P.U + size_t(b) * P.Lg + lane * NU;
    const T* dg = P.dUdt + size_t(b) * P.Lg + lane * NU;
#pragma unroll
    for (int j = 0; j < NU; ++j) U[j] = in_hor ? Ug[j] : T(0), du[j] = in_hor ? dg[j] : T(0);
    const T* pg = P.ptau + size_t(b) * NP * (dv + 1) + lane * NP;
#pragma unroll
    for (int j = 0; j < NP; ++j) p[j] = in_traj ? pg[j] : T(0);
#pragma unroll
    for (int c = 0; c < NX; ++c) xs[c] = P.x_in[size_t(b) * NX + c];
  }
  int n_ax = 0, reason = 0, ksolve = 0;
  T xh[NX];

  // ---- phase 3 of a sweep on the lanes: costate recurrence + the costate part of dH/du (cgmres.hpp:145-161) from the
  //      stage's state / trig / controls.  Returns phi (costate-free part of dH/du) and dF (to be added to component 0).
  auto backward = [&](const T* x, const T* trig, const T* u, T dtau, const GeoPowers<T>& G, T* phi, T* dF) {
    T bw[M::NBW];
    M::stage_coeffs(bw, phi, x, u, p, trig, dtau);
    // terminal costate from the terminal stage (lane dv), wave-uniform
    T xT[NX], pT[NP], lT[NX];
#pragma unroll
    for (int c = 0; c < NX; ++c) xT[c] = wave_bcast(x[c], dv);
#pragma unroll
    for (int j = 0; j < NP; ++j) pT[j] = wave_bcast(p[j], dv);
    M::dPhidx(lT, xT, pT);
    T mb[M::NBW];
#pragma unroll
    for (int c = 0; c < M::NBW; ++c) mb[c] = wave_gather(bw[c], msrc);
    const bool first = lane == 0;
    const T i0 = first ? lT[0] : T(0), i1 = first ? lT[1] : T(0), i2 = first ? lT[2] : T(0), i3 = first ? lT[3] : T(0);
    // (l1, l3):  n1 = l1 + bw5 + bw1 l3,  n3 = l3 + dtau l1 - dtau C22 l3     (PendulumDev::costate_step)
    T D[4] = {T(0), mb[1], dtau, -dtau * M::C22};
    T c[2] = {mb[5] + i1 + mb[1] * i3, i3 + fma_t(D[2], i1, D[3] * i3)};
    scan_aff2(D, c);
    const T in3 = wave_shift_up(c[1], lT[3]);  // costate ENTERING the lane's stage (l1 feeds only this pair)
    // l0:  n0 = l0 + (bw4 + bw0 l3)
    const T l0 = scan_sum(fma_t(mb[0], in3, mb[4]) + i0);
    const T in0 = wave_shift_up(l0, lT[0]);
    // l2:  n2 = (1 - dtau As) l2 + (bw2 l3 + dtau l0)
    const T l2 = scan_geo(fma_t(mb[2], in3, fma_t(dtau, in0, G.pw[0] * i2)), G);
    const T in2 = wave_shift_up(l2, lT[2]);
    const T dFm = fma_t(mb[3], in3, in2 * M::Bs);  // B^T l of the stage (model.hpp:59)
    *dF = wave_gather(dFm, msrc);
  };
  // post-processing of a sweep result (cgmres.hpp:94-96, :173-174)
  auto finish = [&](int mode, const T* phi, T dF, T* out) {
    const T sc_phi = mode == F_RHS ? P.one_m_zh : T(1.0);
    const T sc = mode == F_PLAIN ? T(1.0) : (mode == F_RHS ? P.one_m_zh * P.inv_h : P.inv_h);
#pragma unroll
    for (int j = 0; j < NU; ++j) {
      T rj = phi[j];
      if (mode != F_PLAIN) rj = (rj * sc_phi - Fh[j]) * P.inv_h;
      if (j == 0) rj = fma_t(dF, sc, rj);
      out[j] = in_hor ? rj : T(0);
    }
  };

  // ---- serial state sweeps on DPP quads: quad q of `nq` runs sweep q; x(s), trig(s) -> tab[q][s]
  //      q = 0: (U, x+hf, dtau_h)   q = 1: (U, x, dtau_0)   q = 2: (row 1 of wrow, x+hf, dtau_h)
  auto serial_sweeps = [&](int q_lo, int q_hi, T dtau_h, T dtau_0) {
    auto run = [&](auto slow_tag) -> bool {
      constexpr bool SLOW = decltype(slow_tag)::value;
      const int q = lane >> 2, rho = lane & 3;
      T amax = T(0);
      if (q >= q_lo && q < q_hi) {
        typename M::QuadLane Q;
        Q.init(rho, mc);
        T x[NX];
#pragma unroll
        for (int c = 0; c < NX; ++c) x[c] = q == 1 ? xs[c] : xh[c];
        const T dtau = q == 1 ? dtau_0 : dtau_h, dtau1 = Q.sg * dtau;
        const T* urow = wrow + (q == 2 ? rlen : 0);
        WPair* pt = reinterpret_cast<WPair*>(tab + (q * (dv + 1)) * Lds::TAB_W) + rho;
        T v = M::template quad_begin<SLOW>(x, Q, mc, &amax);
        T ua = urow[0];
        for (int s = 0; s < dv; ++s) {
          // lane rho keeps component {0, 2, 1, 3}[rho] (x1 is sign-flipped on the d-lanes) next to its trig value
          const T xc = rho == 0 ? x[0] : (rho == 1 ? x[2] : (rho == 2 ? x[1] : x[3]));
          pt[s * NX] = {xc, v};
          const T ub = urow[(s + 1) * NU];  // (the word after the last stage is the row's pad)
          M::template quad_stage<SLOW>(x, v, ua, dtau, dtau1, Q, mc, &amax);
          ua = ub;
        }
        const T xc = rho == 0 ? x[0] : (rho == 1 ? x[2] : (rho == 2 ? x[1] : x[3]));
        pt[dv * NX] = {xc, v};
      }
      return __any(M::quad_arg_bad(amax));
    };
    if (__builtin_expect(run(std::false_type{}), 0)) run(std::true_type{});  // arguments beyond the fast trig range
    wave_fence();
  };
  // the lane's stage of sweep q from the table: x[4] and {sin d, cos d, sin x1, cos x1}
  auto read_tab = [&](int q, T* x, T* tr4) {
    const WPair* pt = reinterpret_cast<const WPair*>(tab + (q * (dv + 1) + (in_traj ? lane : dv)) * Lds::TAB_W);
    const WPair e0 = pt[0], e1 = pt[1], e2 = pt[2], e3 = pt[3];
    x[0] = e0.a, x[2] = e1.a, x[1] = e2.a, x[3] = e3.a;
    tr4[0] = e0.b, tr4[1] = e1.b, tr4[2] = e2.b, tr4[3] = e3.b;
  };

  const int nt = P.n_ticks;
  for (int tk = 0; tk < nt; ++tk) {
    const bool last = tk + 1 == nt;
    const T dtau_h = P.dtau_tab[2 * tk], dtau_0 = P.dtau_tab[2 * tk + 1];
    if (P.ptau_seq) {  // set_ptau before this tick (cgmres.hpp:36-39)
      const T* src = P.ptau_seq + size_t(tk) * P.pseq_tick + size_t(b) * P.pseq_inst + lane * NP;
#pragma unroll
      for (int j = 0; j < NP; ++j) p[j] = in_traj ? src[j] : T(0);
    }
    GeoPowers<T> Gh, G0;  // powers of 1 - dtau As for the geometric scans of this tick
    Gh.make(T(1) - dtau_h * M::As);
    G0.make(T(1) - dtau_0 * M::As);
    for (int q = lane; q < Lds::pitch_H(kmax); q += 64) Hi[q] = T(0);  // the exported Hessenberg has no stale entries
    // ---- cgmres.hpp:83-85: x_dxh = x + h f(x, U_0)
    {
      T u0[NU], f[NX], tr[M::NC];
#pragma unroll
      for (int j = 0; j < NU; ++j) u0[j] = wave_bcast(U[j], 0);
      M::dxdt(f, xs, u0, tr, mc);
#pragma unroll
      for (int c = 0; c < NX; ++c) xh[c] = f[c] * P.h + xs[c];
    }
    // ---- the three sweeps in front of the solve: #1 Fh = F(U, x+hf, t+h) (:88), #2 b (:91-96), #3 A*dUdt (:99 ->
    //      gmres.hpp:33), state sweeps side by side on three quads
    if (in_hor) {
#pragma unroll
      for (int j = 0; j < NU; ++j) {
        wrow[lane * NU + j] = U[j];
        wrow[rlen + lane * NU + j] = du[j] * P.h + U[j];  // cgmres.hpp:168-169
      }
    }
    if (lane == 0) wrow[dv * NU] = T(0), wrow[rlen + dv * NU] = T(0);
    wave_fence();
    serial_sweeps(0, 3, dtau_h, dtau_0);
    T xb[NX], tb[4];  // base trajectory of this tick (sweep #1) on the lanes
    T bb[NU], ax0[NU];
    {
      read_tab(0, xb, tb);
      T phi[NU], dF, trig[3] = {tb[0], tb[1], tb[3]};
      backward(xb, trig, U, dtau_h, Gh, phi, &dF);
      finish(F_PLAIN, phi, dF, Fh);
    }
    {
      T x[NX], t4[4];
      read_tab(1, x, t4);
      T phi[NU], dF, trig[3] = {t4[0], t4[1], t4[3]};
      backward(x, trig, U, dtau_0, G0, phi, &dF);
      finish(F_RHS, phi, dF, bb);
    }
    {
      T x[NX], t4[4], uu[NU];
      read_tab(2, x, t4);
#pragma unroll
      for (int j = 0; j < NU; ++j) uu[j] = du[j] * P.h + U[j];
      T phi[NU], dF, trig[3] = {t4[0], t4[1], t4[3]};
      backward(x, trig, uu, dtau_h, Gh, phi, &dF);
      finish(F_AX, phi, dF, ax0);
    }

    // ---- Ax_func (cgmres.hpp:164-175) of the direction `dir`: Newton on the trajectory, from the base trajectory
    auto ax = [&](const T* dir, T* out) {
      T u[NU];
#pragma unroll
      for (int j = 0; j < NU; ++j) u[j] = dir[j] * P.h + U[j];
      // x2' = (1 - dtau As) x2 + dtau Bs u0;  x0' = x0 + dtau x2      (model.hpp:38,40)
      const bool first = lane == 0;
      const T e2 = (in_hor ? (dtau_h * M::Bs) * u[0] : T(0)) + (first ? Gh.pw[0] * xh[2] : T(0));
      const T x2 = wave_shift_up(scan_geo(e2, Gh), xh[2]);
      const T x0 = wave_shift_up(scan_sum(dtau_h * x2 + (first ? xh[0] : T(0))), xh[0]);
      const T Pq = M::A32 * x2 * x2, Qq = M::A32a * x2 - M::A32b * u[0];
      const T dx0 = x0 - xb[0];
      T y1 = xb[1], y3 = xb[3];
      T sd = tb[0], cd = tb[1], s1 = tb[2], c1 = tb[3];
      bool converged = false;
      const T rmax = T(0.9) * sqrt_t<T>(T(decltype(mc)::rot_zmax));
      for (int it = 0; it < WAVE_NEWTON_MAX; ++it) {
        const T d1 = y1 - xb[1], dd = dx0 - d1;
        if (__builtin_expect(__any(in_traj && (!(abs_t(dd) <= rmax) || !(abs_t(d1) <= rmax))), 0)) {
          mc.sincos_pair(x0 - y1, y1, &sd, &cd, &s1, &c1);  // far from the base trajectory: fresh evaluation
        } else {
          W::rotate(tb[0], tb[1], dd, &sd, &cd, mc);
          W::rotate(tb[2], tb[3], d1, &s1, &c1, mc);
        }
        // model.hpp:41 and its derivative in x1
        const T g = fma_t(Pq, sd, fma_t(M::A52, s1, fma_t(Qq, cd, M::C22 * (x2 - y3))));
        const T J1 = fma_t(Qq, sd, fma_t(M::A52, c1, -(Pq * cd)));
        const T t1 = fma_t(dtau_h, y3, y1), t3 = fma_t(dtau_h, g, y3);
        // defect of the recurrence at the lane's stage, and the linearised step that leads to it
        T c[2] = {wave_shift_up(t1, xh[1]) - y1, wave_shift_up(t3, xh[3]) - y3};
        T D[4] = {T(0), dtau_h, dtau_h * wave_shift_up(J1, T(0)), -dtau_h * M::C22};
        scan_aff2(D, c);
        y1 += c[0], y3 += c[1];
        const T th = T(1e-10);
        if (!__any(in_traj && (abs_t(c[0]) > th * (T(1) + abs_t(y1)) || abs_t(c[1]) > th * (T(1) + abs_t(y3))))) {
          // the correction is below the square root of the rounding level: trig values to first order, done
          const T nsd = fma_t(cd, -c[0], sd), ncd = fma_t(sd, c[0], cd);
          const T ns1 = fma_t(c1, c[0], s1), nc1 = fma_t(s1, -c[0], c1);
          sd = nsd, cd = ncd, s1 = ns1, c1 = nc1;
          converged = true;
          break;
        }
      }
      T x[NX] = {x0, y1, x2, y3};
      T trig[3] = {sd, cd, c1};
      if (__builtin_expect(!converged, 0)) {  // (wave-uniform) Newton did not settle: the serial sweep
        if (in_hor) {
#pragma unroll
          for (int j = 0; j < NU; ++j) wrow[rlen + lane * NU + j] = u[j];
        }
        wave_fence();
        serial_sweeps(2, 3, dtau_h, dtau_0);
        T t4[4];
        read_tab(2, x, t4);
        trig[0] = t4[0], trig[1] = t4[1], trig[2] = t4[3];
      }
      T phi[NU], dF;
      backward(x, trig, u, dtau_h, Gh, phi, &dF);
      finish(F_AX, phi, dF, out);
    };

    // ---- Gmres::gmres (gmres.hpp:28-112), basis in registers
    T V[KM + 1][NU];
    T vcur[NU], w[NU];
    bool active = true;
    reason = 0, n_ax = 0, ksolve = 0;
    auto dot = [&](const T* a, const T* c) {
      T s = a[0] * c[0];
#pragma unroll
      for (int j = 1; j < NU; ++j) s = fma_t(a[j], c[j], s);
      return wave_sum(s);
    };
    {
#pragma unroll
      for (int j = 0; j < NU; ++j) vcur[j] = bb[j] - ax0[j];  // gmres.hpp:33-34
      const T rho0 = sqrt_t<T>(dot(vcur, vcur));               // :37
      if (lane == 0) rhoi[0] = rho0;
      // (decisions go through __any: wave-uniform by construction, and the compiler then keeps them in scalar registers)
      if (__any(!finite_t(rho0))) active = false, reason = 4;         // CGMRES_HIP_EXIT_NONFINITE
      if (active && __any(rho0 < P.tol)) active = false, reason = 2;  // :39-41
      if (active) {
        const T inv = T(1.0) / rho0;  // :44
#pragma unroll
        for (int j = 0; j < NU; ++j) vcur[j] = vcur[j] * inv, V[0][j] = vcur[j];
      }
    }
    int nv = active ? 1 : 0;  // basis vectors stored
    int k = 0;
    for (; active && k < kmax; ++k) {  // gmres.hpp:46
      ax(vcur, w);                     // :48
      n_ax = k + 1;
      T* Hk = Hi + hoff(k);
      // modified Gram-Schmidt (:52-58), in order, one static instance per k (register-resident basis)
      auto rounds = [&](auto kc) {
        constexpr int K = decltype(kc)::value;
#pragma unroll
        for (int i = 0; i <= K; ++i) {
          const T hik = dot(V[i], w);
#pragma unroll
          for (int j = 0; j < NU; ++j) w[j] = fma_t(-hik, V[i][j], w[j]);
          if (lane == 0) Hk[i] = hik;
        }
      };
      switch (k) {
#define CGM_WCASE(n)                                    \
  case n:                                               \
    if constexpr (n < KM) rounds(std::integral_constant<int, n>{}); \
    break;
        CGM_WCASE(0) CGM_WCASE(1) CGM_WCASE(2) CGM_WCASE(3) CGM_WCASE(4) CGM_WCASE(5) CGM_WCASE(6) CGM_WCASE(7)
        CGM_WCASE(8) CGM_WCASE(9) CGM_WCASE(10) CGM_WCASE(11) CGM_WCASE(12) CGM_WCASE(13) CGM_WCASE(14) CGM_WCASE(15)
        default: break;
      }
      const T hn = sqrt_t<T>(dot(w, w));  // :60
      if (__any(abs_t(hn) < T(DBL_EPSILON) || !finite_t(hn))) {  // :63-65 breakdown: x untouched; non-finite: x <- NaN below
        reason = __any(!finite_t(hn)) ? 4 : 3;
        if (lane == 0) rhoi[kmax + 1] = hn;  // (exported as h(k+1,k) of the column that broke down)
        active = false;
        break;
      }
      const T inv = T(1.0) / hn;  // :67
#pragma unroll
      for (int j = 0; j < NU; ++j) vcur[j] = w[j] * inv;
      switch (k) {
#define CGM_WSTORE(n)                                  \
  case n:                                              \
    if constexpr (n < KM) {                            \
      for (int j = 0; j < NU; ++j) V[n + 1][j] = vcur[j]; \
    }                                                  \
    break;
        CGM_WSTORE(0) CGM_WSTORE(1) CGM_WSTORE(2) CGM_WSTORE(3) CGM_WSTORE(4) CGM_WSTORE(5) CGM_WSTORE(6) CGM_WSTORE(7)
        CGM_WSTORE(8) CGM_WSTORE(9) CGM_WSTORE(10) CGM_WSTORE(11) CGM_WSTORE(12) CGM_WSTORE(13) CGM_WSTORE(14)
        CGM_WSTORE(15)
        default: break;
      }
      nv = k + 2;
      // Hessenberg column k: stored reflectors, new reflector, residual rotation (:71-90) — wave-uniform scalar work on
      // the small arrays in LDS (every lane computes, lane 0 stores)
      wave_fence();
      const T en = WgCtx<M, T, 16, 10>::hess_column(Hi, gi, rhoi, k, hn, lane == 0);
      wave_fence();
      if (__any(abs_t(en) < P.tol)) {  // :93-95 — converged: column k is NOT used by the solve
        reason = 1, ksolve = k;
        active = false;
        break;
      }
    }
    if (reason == 0) ksolve = kmax;  // natural exit: every column is used
    if (reason <= 1) {
      // back substitution (gmres.hpp:100-107) over the lanes: lane j owns e_j (see WgCtx::gmres)
      const int ks = ksolve;
      T e = lane < ks ? rhoi[lane] : T(0);
      for (int i = ks - 1; i >= 0; --i) {
        const T hii = Hi[hoff(i) + i], hji = Hi[hoff(i) + (lane < i ? lane : i)];
        const T y = e / hii;  // meaningful in lane i
        const T yi = wave_bcast(y, i);
        e = lane < i ? e - hji * yi : (lane == i ? y : e);
      }
      if (lane < ks) rhoi[lane] = e;
      // x += V[:, 0:ks] y (gmres.hpp:110-111), accumulated j-ascending from 0
      T acc[NU];
#pragma unroll
      for (int j = 0; j < NU; ++j) acc[j] = T(0);
#pragma unroll
      for (int q = 0; q < KM; ++q) {
        if (q < ks) {
          const T yq = wave_bcast(e, q);
#pragma unroll
          for (int j = 0; j < NU; ++j) acc[j] = fma_t(V[q][j], yq, acc[j]);
        }
      }
#pragma unroll
      for (int j = 0; j < NU; ++j) du[j] = du[j] + acc[j];
    }
    if (reason == 4) {
#pragma unroll
      for (int j = 0; j < NU; ++j) du[j] = in_hor ? quiet_nan<T>() : T(0);
    }
    // ---- U += dUdt*dt, u = U[0:dim_u]  (cgmres.hpp:102-109)
#pragma unroll
    for (int j = 0; j < NU; ++j) U[j] = U[j] + du[j] * P.dt;
    T unew[NU];
#pragma unroll
    for (int j = 0; j < NU; ++j) unew[j] = wave_bcast(U[j], 0);
    if (last) {
      if (in_hor) {
        T* Ug = P.U + size_t(b) * P.Lg + lane * NU;
        T* dg = P.dUdt + size_t(b) * P.Lg + lane * NU;
        T* fg = P.Fh + size_t(b) * P.Lg + lane * NU;
#pragma unroll
        for (int j = 0; j < NU; ++j) Ug[j] = U[j], dg[j] = du[j], fg[j] = Fh[j];
      }
      if (lane < NU) P.u_out[size_t(b) * NU + lane] = lane == 0 ? unew[0] : (lane == 1 ? unew[1] : unew[2]);
      if (lane < NX) P.xdxh[size_t(b) * NX + lane] = lane == 0 ? xh[0] : (lane == 1 ? xh[1] : (lane == 2 ? xh[2] : xh[3]));
      // status + small Krylov arrays (the layout of WgCtx::store_status) + the basis rows in the wg mapping's
      // pair-interleaved form (ctx_wg get_krylov undoes it)
      wave_fence();
      const int ks_all = k1 * k1 + k1 + 3 * kmax;
      T* dst = P.kry + size_t(b) * ks_all;
      for (int q = lane; q < k1 * k1; q += 64) {
        const int col = q / k1, row = q - col * k1;
        T v = (col < kmax && row <= col) ? Hi[hoff(col) + row] : T(0);
        if (row == col + 1 && col + 1 == n_ax && reason == 3) v = rhoi[kmax + 1];
        dst[q] = v;
      }
      for (int q = lane; q < k1; q += 64) dst[k1 * k1 + q] = rhoi[q];
      for (int q = lane; q < 3 * kmax; q += 64) dst[k1 * k1 + k1 + q] = gi[q];
      if (lane == 0) P.n_ax[b] = n_ax, P.reason[b] = reason;
      if (in_hor) {
#pragma unroll
        for (int q = 0; q <= KM; ++q) {
          if (q < nv) {
            T* row = P.V + (size_t(b) * k1 + q) * P.Lv;
#pragma unroll
            for (int j = 0; j < NU; ++j) {
              const int e = lane * NU + j, rr = e & 15, m = e >> 4;
              row[(m >> 1) * 32 + 2 * rr + (m & 1)] = V[q][j];
            }
          }
        }
      }
    }
    if (P.x_next) {  // plant step of the example main loop (<example>/main.cpp:71-73)
      T f[NX], tr[M::NC];
      M::dxdt(f, xs, unew, tr, mc);
#pragma unroll
      for (int c = 0; c < NX; ++c) xs[c] = xs[c] + f[c] * P.dt;
      if (last && lane < NX)
        P.x_next[size_t(b) * NX + lane] = lane == 0 ? xs[0] : (lane == 1 ? xs[1] : (lane == 2 ? xs[2] : xs[3]));
    }
  }
}

}  // namespace cgm
