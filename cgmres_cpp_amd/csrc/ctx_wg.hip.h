// Host side of the "wg" mapping (tick_wg.hip.h): owns the instance-major HBM state of the batch,
// computes the batch-wide scalars of each tick (t, dtau) and launches one kernel per control tick.
#pragma once
#include <vector>
#include "ctx_common.hip.h"
#include "tick_wg.hip.h"
#include "tick_wave.hip.h"
#include "util_kernels.hip.h"

namespace cgm {

constexpr size_t kLdsLimit = 160 * 1024 - 1024;  // gfx950: 160 KiB per workgroup, minus the kernels' small static LDS
// Two workgroups share a CU when each allocates at most half of the 160 KiB (measured, tools/ubench_hwid.hip: 81408
// bytes of dynamic LDS co-reside, 81920 do not).
constexpr size_t kLdsLimitLean = 80 * 1024 - 512;

template <class T>
__global__ void replicate_rows_im(T* __restrict__ dst, size_t dst_pitch, const T* __restrict__ src, int B, int n,
                                  int reps, int bcast) {
  // dst[b][rep*n + j] = src[(bcast ? 0 : b)*n + j]
  const int b = blockIdx.y;
  for (int q = blockIdx.x * blockDim.x + threadIdx.x; q < n * reps; q += gridDim.x * blockDim.x)
    if (b < B) dst[size_t(b) * dst_pitch + q] = src[size_t(bcast ? 0 : b) * n + (q % n)];
}

template <class M, class T>
struct CtxWg final : cgmres_hip_ctx {
  WgParams<T> P{};
  T t = T(0);
  int ipw = 0, maxm = 0, ks_all = 0, plan = PLAN_FULL;
  size_t lds_bytes = 0, lds_bytes_hook = 0;
  void (*k_tick)(WgParams<T>) = nullptr;
  void (*k_hook)(WgParams<T>) = nullptr;
  T *stage = nullptr, *stage2 = nullptr;
  size_t stage_n = 0, stage2_n = 0;
  T *x_dev = nullptr, *u_dev = nullptr;
  // control() through host pointers on a small batch (a single Cgmres<Model> object: batch 1): x and u travel through
  // host-mapped pinned buffers the kernel reads / writes directly — one launch + one stream synchronisation per tick
  // instead of copy, launch, copy, synchronise (each hipMemcpyAsync costs ~8 us of host time)
  T *pin_x = nullptr, *pin_u = nullptr, *pin_x_dev = nullptr, *pin_u_dev = nullptr;
  static constexpr int kPinnedIoMaxBatch = 1024;
  ~CtxWg() override {
    (void)hipSetDevice(cfg.device);
    if (stream) (void)hipStreamSynchronize(stream);
    if (pin_x) (void)hipHostFree(pin_x);
    if (pin_u) (void)hipHostFree(pin_u);
  }
  int fh_hbm_for_hooks = 0;
  int* perm_dev = nullptr;   // placement of the next fused launch (bin_by_count_kernel)
  bool binning = false;      // closed loop: bin the instances by their last Arnoldi count before every launch
  bool have_counts = false;  // n_ax holds the counts of a finished tick

  int par_costate = 0;  // 0 serial, 1 chunk-parallel with LDS scratch, 2 two-pass (WgCtx::PAR)
  // variant 4 ("wave", tick_wave.hip.h): one wavefront per controller; the tick kernel changes, the HBM state, the
  // white-box hooks and everything else of this context stay those of the wg mapping
  bool wave = false;
  bool row_newton = false;  // wg kernel with WgCtx::NWT = 1
  // A Newton sweep costs the same whatever the horizon (four stages per lane, lanes beyond the horizon idle), the serial
  // sweep is proportional to it: measured at 4096 controllers, k_max = 10 — dv = 30: 103.6 vs 100.2 us per tick (serial
  // wins), 36: 105.3 vs 109.3, 40: 106.8 vs 114.6, 44: 99.6 vs 115.1, 50: 102.0 vs 125.1; dv = 25, k_max = 5: 59.7 vs 52.3.
  static constexpr int kRowNewtonMinDv = 33;
  bool row_scan = false;    // wg kernel with WgCtx::NWT = 2
  template <class MM, class = void>
  struct RowAffine : std::false_type {};
  template <class MM>
  struct RowAffine<MM, std::void_t<decltype(MM::ROW_AFFINE)>> : std::integral_constant<bool, MM::ROW_AFFINE> {};
  static constexpr int kWaveKmax = 10, kWaveWpb = 1;
  size_t lds_bytes_tick() const { return wave ? WaveLds<M, T>::bytes(cfg.dv, cfg.k_max, kWaveWpb) : lds_bytes; }
  static bool wave_supported(const cgmres_hip_config& c) {
    if constexpr (WaveOps<M>::value && std::is_same<T, double>::value)
      return c.dv >= 1 && c.dv <= 63 && c.k_max >= 1 && c.k_max <= kWaveKmax;
    return false;
  }
  const char* variant_name() const override {
    if (wave) return "wave";
    if (row_newton) return "wg+row-newton";
    if (row_scan) return "wg+row-scan";
    static const char* const names[2][3] = {{"wg", "wg+parallel-costate", "wg+two-pass-costate"},
                                            {"wg-lean", "wg-lean", "wg-lean+two-pass-costate"}};
    return names[plan == PLAN_LEAN][par_costate];
  }

  // (IPW, MAXM) instantiations: 16 or 8 instances per workgroup, vectors up to 160 or 320 elements; the lean LDS plan
  // (two workgroups per CU) exists for 16 instances per workgroup
  template <int IPW, int MAXM>
  void pick(bool lean, int par) {
    k_tick = tick_wg_kernel<M, T, IPW, MAXM>;
    if constexpr (IPW == 16) {
      if (lean) k_tick = tick_wg_kernel<M, T, IPW, MAXM, true>;
      if constexpr (kParCostate<MAXM>) {
        if (par == 1 && !lean) k_tick = tick_wg_kernel<M, T, IPW, MAXM, false, 1>;
      }
      if constexpr (kPar2) {
        if (par == 2) k_tick = lean ? tick_wg_kernel<M, T, IPW, MAXM, true, 2> : tick_wg_kernel<M, T, IPW, MAXM, false, 2>;
      }
    }
    k_hook = hook_wg_kernel<M, T, IPW, MAXM>;
    ipw = IPW, maxm = MAXM;
  }
  // kernels with the chunk-parallel costate sweep: the form with per-stage LDS scratch (WgCtx::sweep_costate_par) exists for
  // the short-vector instantiations, the two-pass form (sweep_costate_2pass) for every 16-instance kernel
  static constexpr bool kPar2 = M::COSTATE_HOM && M::NX * 16 <= 64 && M::NX % 2 == 0;
  template <int MAXM>
  static constexpr bool kParCostate = kPar2 && MAXM == 10;
  static int pitch_H(int k_max) { return ((k_max * (k_max + 1)) / 2 + 2) | 1; }  // (+2: hess_column's look-ahead past the last column)
  // The costate sweep's look-ahead (WgCtx::costate_run) requests the coefficients of up to THREE stages below the first
  // stage of its range (the tail of the chunk-parallel form: `post` = 5) and the output words of those stages; the
  // values are never used, but the addresses must stay inside the workgroup's LDS allocation (an access outside it is
  // an aperture violation on this platform).  Below the stage table sit `rows` row arrays of pitch Lp (+ `front` small
  // words per instance in the lean plan): they must cover 3 stages of the table (3*NSTG words per instance, + the pair
  // offset), and the arrays in front of the first `out` row must cover 3*NU words.  Built-in models (NSTG <= 6) pass
  // from dv = 5 (lean) / any dv (full plans); a user model with many stage coefficients and a short horizon
  // (NX = 4, NU = 1: NSTG = 24, Lp = dv|1) does not — it then runs on the lane mapping.
  static bool lookahead_fits(int rows, int Lp, int front_words_per_inst) {
    constexpr int NSTG = WgLds<M, T, 16>::NSTG;
    return rows * Lp + front_words_per_inst >= 3 * NSTG + 2 && (rows - 1) * Lp + front_words_per_inst >= 3 * M::NU;
  }
  // lean plan: 16 instances per workgroup in at most half a CU's LDS; the white-box hooks keep running on the full
  // (or fh_hbm) plan of the same sizes, so that one must fit as well
  static bool lean_supported(const cgmres_hip_config& c, size_t* bytes_out) {
    const int L = M::NU * c.dv;
    if (L > 320 || DxdtUsesP<M, T>::value) return false;  // (a state equation that reads p wants the horizon in LDS)
    const int Lp = L | 1, Pp = (M::NP * (c.dv + 1)) | 1, Hp = pitch_H(c.k_max);
    if (!lookahead_fits(1, Lp, 4 * M::NX + M::NU)) return false;  // (lean: W is the only row array, see WgLds)
    const size_t bl = WgLds<M, T, 16>::bytes(c.dv, c.k_max, Lp, Pp, Hp, PLAN_LEAN);
    int ipw_full;
    size_t b_full;
    if (bl > kLdsLimitLean || !supported(c, &ipw_full, &b_full) || ipw_full != 16) return false;
    *bytes_out = bl;
    return true;
  }
  // Preference: 16 instances per workgroup with everything in LDS; 16 with F(U,x+hf,t+h) in HBM (WgLds::count_T);
  // 8 instances per workgroup.
  static bool supported(const cgmres_hip_config& c, int* ipw_out, size_t* bytes_out, int* fh_hbm_out = nullptr) {
    const int L = M::NU * c.dv;
    if (L > 320) return false;
    const int Lp = L | 1, Pp = (M::NP * (c.dv + 1)) | 1, Hp = pitch_H(c.k_max);
    const size_t b16 = WgLds<M, T, 16>::bytes(c.dv, c.k_max, Lp, Pp, Hp);
    const size_t b16h = WgLds<M, T, 16>::bytes(c.dv, c.k_max, Lp, Pp, Hp, PLAN_FH_HBM);
    const size_t b8 = WgLds<M, T, 8>::bytes(c.dv, c.k_max, Lp, Pp, Hp);
    if (fh_hbm_out) *fh_hbm_out = 0;
    const bool full_ok = lookahead_fits(3, Lp, 0), fh_ok = lookahead_fits(2, Lp, 0);  // row arrays in front of the table
    if ((c.flags & CGMRES_HIP_FLAG_IPW8) && b8 <= kLdsLimit && full_ok) {
      *ipw_out = 8, *bytes_out = b8;
      return true;
    }
    if (b16 <= kLdsLimit && full_ok) {
      *ipw_out = 16, *bytes_out = b16;
      return true;
    }
    if (L > 160 && b16h <= kLdsLimit && fh_ok) {  // the long-vector kernels (MAXM = 20) are the ones that carry this mode
      *ipw_out = 16, *bytes_out = b16h;
      if (fh_hbm_out) *fh_hbm_out = 1;
      return true;
    }
    if (b8 <= kLdsLimit && full_ok) {
      *ipw_out = 8, *bytes_out = b8;
      return true;
    }
    return false;
  }

  int init() override {
    if (int rc = init_common()) return rc;
    nx = M::NX, nu = M::NU, np = M::NP;
    L = nu * cfg.dv;
    int want = 0;
    int fh_hbm = 0;
    if (!supported(cfg, &want, &lds_bytes, &fh_hbm))
      return fail(CGMRES_HIP_EINVAL, "wg mapping: dim_u*dv = %d / LDS footprint not supported", L);
    // Plan.  variant 3 asks for the lean plan; the default takes it when the batch needs more 16-instance workgroups than
    // the GPU has CUs (two workgroups per CU then run their serial phases side by side instead of in two rounds).
    lds_bytes_hook = lds_bytes;
    size_t lean_bytes = 0;
    const bool lean_ok = want == 16 && lean_supported(cfg, &lean_bytes);
    int cus = 0;
    HIP_TRY(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, cfg.device));
    bool lean = false;
    if (cfg.variant == 3) {
      if (!lean_ok) return fail(CGMRES_HIP_EINVAL, "wg-lean mapping: LDS footprint of dim_u*dv = %d, k_max = %d not supported", L, cfg.k_max);
      lean = true;
    } else if (cfg.variant == 0 && lean_ok && (cfg.batch + 15) / 16 > cus) {
      lean = true;
    }
    // the latency mapping: asked for, or (library's choice) up to two controllers per SIMD.  One wave per SIMD runs a
    // tick in ~43 us (wg: ~117 us whatever the batch); the kernel takes all 512 registers, so a batch beyond one
    // controller per SIMD runs in rounds: two rounds (~88 us) still beat the wg mapping, three do not.
    if (cfg.variant == 4) {
      if (!wave_supported(cfg)) return fail(CGMRES_HIP_EINVAL, "wave mapping: model / dtype / dv = %d / k_max = %d not supported", cfg.dv, cfg.k_max);
      wave = true;
    } else if (cfg.variant == 0 && wave_supported(cfg) && cfg.batch <= 8 * cus && !(cfg.flags & CGMRES_HIP_FLAG_NO_WAVE)) {
      wave = true;
    }
    if (lean) plan = PLAN_LEAN, fh_hbm = 0, lds_bytes = lean_bytes;
    else plan = fh_hbm ? PLAN_FH_HBM : PLAN_FULL;
    cfg.variant = wave ? 4 : (lean ? 3 : 2);
    const int fh_hbm_hook = lean ? [&] { int i, f = 0; size_t bb; supported(cfg, &i, &bb, &f); return f; }() : fh_hbm;
    const bool big = L > 160;
    // chunk-parallel costate sweep (WgCtx::sweep_costate_par): its own kernel instantiation on the full plan, taken when
    // its scratch fits as well (the white-box hooks keep the serial sweep)
    int par = 0;
    const bool serial = cfg.flags & CGMRES_HIP_FLAG_SERIAL_COSTATE, two_pass = cfg.flags & CGMRES_HIP_FLAG_TWO_PASS_COSTATE;
    if (kParCostate<10> && want == 16 && !big && !lean && cfg.dv >= 4 && !serial && !two_pass) {
      const size_t extra = WgLds<M, T, 16>::scan_count(cfg.dv) * sizeof(T) + 16;
      if (lds_bytes + extra <= kLdsLimit) par = 1, lds_bytes += extra;
    }
    if (kPar2 && want == 16 && par == 0 && !serial) {
      // two-pass form: 4 chunks where their boundary records fit behind the plan's arrays, 3 otherwise (the lean plans)
      const size_t limit = lean ? kLdsLimitLean : kLdsLimit;
      for (int chunks = 4; chunks >= 3 && par == 0; --chunks) {
        const size_t extra = WgLds<M, T, 16>::scan2_count(chunks) * sizeof(T) + 16;
        if (cfg.dv >= 2 * chunks && lds_bytes + extra <= limit) par = 2, lds_bytes += extra, P.cs_chunks = chunks;
      }
    }
    par_costate = par;
    if (want == 16 && !big) pick<16, 10>(lean, par);
    if (want == 16 && big) pick<16, 20>(lean, par);
    if (want == 8 && !big) pick<8, 10>(false, 0);
    if (want == 8 && big) pick<8, 20>(false, 0);
    if constexpr (WaveOps<M>::value && std::is_same<T, double>::value) {
      if (wave) k_tick = tick_wave_kernel<M, T, kWaveKmax, kWaveWpb>;
    }
    // row-parallel scans for a state equation that is affine in x (WgCtx::NWT = 2): the full plan's 16-instance kernel
    if constexpr (RowAffine<M>::value && std::is_same<T, double>::value) {
      // (a flag that asks for a particular costate sweep asks for the kernel that has one)
      if (!(cfg.flags & CGMRES_HIP_FLAG_SERIAL_STATE_SWEEP) && !serial && !two_pass && !wave && want == 16 && !big && !lean &&
          !fh_hbm && cfg.dv <= 63 && cfg.k_max <= 12) {
        row_scan = true;
        k_tick = tick_wg_kernel<M, T, 16, 10, false, 0, 2>;
      }
    }
    // row-parallel Newton state sweeps (WgCtx::NWT = 1): the full plan's 16-instance kernel.  Its base trajectory (NBASE
    // arrays of 8 KB) goes where LDS is idle during the Arnoldi loop — the stage table, the scratch of the costate scan —
    // and behind everything else for the rest; the kernel is taken when all of that fits.
    if constexpr (M::HAS_QUAD_SWEEP && std::is_same<T, double>::value) {
      using Ctx = WgCtx<M, T, 16, 10, false, 1, 1>;
      using LdsX = WgLds<M, T, 16, NWT_TABX>;
      const int Lp = L | 1, Pp = (np * (cfg.dv + 1)) | 1, Hp = pitch_H(cfg.k_max);
      const size_t arr = Ctx::base_array_bytes();
      auto up16 = [](size_t v) { return (v + 15) & ~size_t(15); };
      const size_t tab_off = size_t(3) * 16 * Lp * sizeof(T), tab_cap = LdsX::tab_count(cfg.dv) * sizeof(T);
      const size_t scan_off = up16(LdsX::count_T(cfg.dv, cfg.k_max, Lp, Pp, Hp) * sizeof(T) + 5 * 16 * sizeof(int));
      const size_t scan_cap = par == 1 ? WgLds<M, T, 16>::scan_count(cfg.dv) * sizeof(T)
                                       : (par == 2 ? WgLds<M, T, 16>::scan2_count(P.cs_chunks) * sizeof(T) : 0);
      const size_t with_pad = lds_bytes + size_t(cfg.dv) * NWT_TABX * sizeof(T);
      size_t end = up16(with_pad), off[Ctx::NBASE];
      size_t in_tab = 0, in_scan = 0;
      for (int k = 0; k < Ctx::NBASE; ++k) {
        if ((in_tab + 1) * arr <= tab_cap) off[k] = tab_off + in_tab++ * arr;
        else if ((in_scan + 1) * arr <= scan_cap) off[k] = scan_off + in_scan++ * arr;
        else off[k] = end, end += arr;
      }
      if (!(cfg.flags & CGMRES_HIP_FLAG_SERIAL_STATE_SWEEP) && !serial && !two_pass && !wave && want == 16 && !big && !lean &&
          !fh_hbm && cfg.dv >= kRowNewtonMinDv && cfg.dv <= 63 && cfg.k_max <= 12 && end <= kLdsLimit) {
        row_newton = true, lds_bytes = end;
        for (int k = 0; k < Ctx::NBASE; ++k) P.base_off[k] = int(off[k]);
        k_tick = tick_wg_kernel<M, T, 16, 10, false, 1, 1>;
      }
    }
    HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(k_tick), hipFuncAttributeMaxDynamicSharedMemorySize,
                                int(lds_bytes_tick())));
    HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(k_hook), hipFuncAttributeMaxDynamicSharedMemorySize,
                                int(lds_bytes_hook)));
    fh_hbm_for_hooks = fh_hbm_hook;
    const int k1 = cfg.k_max + 1;
    ks_all = k1 * k1 + k1 + 3 * cfg.k_max;
    P.B = cfg.batch, P.dv = cfg.dv, P.kmax = cfg.k_max, P.L = L, P.fh_hbm = fh_hbm, P.lds_bytes = int(lds_bytes);
    P.Lp = L | 1, P.Lg = (L + 15) / 16 * 16, P.Lv = 16 * maxm, P.Pp = (np * (cfg.dv + 1)) | 1, P.Hp = pitch_H(cfg.k_max);
    P.h = T(cfg.h), P.dt = T(cfg.dt), P.tol = T(cfg.tol);
    P.wave_dbg = ((cfg.flags & CGMRES_HIP_FLAG_WAVE_FRESH_TRIG) ? 1 : 0) | ((cfg.flags & CGMRES_HIP_FLAG_WAVE_SERIAL_SWEEPS) ? 2 : 0);
    P.inv_h = T(1.0) / P.h;
    P.one_m_zh = (1 - T(cfg.zeta) * P.h);
    const size_t B = cfg.batch, Lg = P.Lg;
    int rc = 0;
    if ((rc = dalloc(&P.U, B * Lg)) || (rc = dalloc(&P.dUdt, B * Lg)) || (rc = dalloc(&P.Fh, B * Lg)) ||
        (rc = dalloc(&P.V, B * k1 * size_t(P.Lv))) || (rc = dalloc(&P.xdxh, B * nx)) ||
        (rc = dalloc(&P.ptau, B * size_t(np) * (cfg.dv + 1))) || (rc = dalloc(&P.kry, B * ks_all)) ||
        (rc = dalloc(&P.scr, size_t((cfg.batch + ipw - 1) / ipw) * 2 * (cfg.dv + WgLds<M, T, 16>::TAB_PAD) * (WgLds<M, T, 16>::NSTG * ipw + NWT_TABX))) ||
        (rc = dalloc(&P.pT, lean ? size_t((cfg.batch + ipw - 1) / ipw) * (cfg.dv + 1) * (np ? np : 1) * ipw : 1)) ||
        (rc = dalloc(&P.park, size_t((cfg.batch + ipw - 1) / ipw) * ipw * P.Lv)) ||  // (every kernel family parks the solution vector now)
        (rc = dalloc(&P.n_ax, B)) || (rc = dalloc(&P.reason, B)) || (rc = dalloc(&x_dev, B * nx)) ||
        (rc = dalloc(&u_dev, B * nu)) || (rc = dalloc(&perm_dev, B)))
      return rc;
    // Binning pays only when the batch needs more workgroups than the GPU holds at once (then the device works through
    // a queue of workgroups and the sum of their times counts); with every workgroup resident the launch lasts as long
    // as its slowest workgroup wherever the instances sit.  Early exits need tol > 0.
    binning = !wave && cfg.tol > 0 && !(cfg.flags & CGMRES_HIP_FLAG_NO_BINNING) &&
              (cfg.batch + ipw - 1) / ipw > cus * (lean ? 2 : 1);
    HIP_TRY(hipStreamSynchronize(stream));
    return 0;
  }

  dim3 grid() const { return dim3((cfg.batch + ipw - 1) / ipw); }
  dim3 block() const { return dim3(ipw * 16); }
  T dtau_of(T tt) const {  // cgmres.hpp:32-34, once per tick on the host for the whole batch
    return T(cfg.Tf) * (1 - std::exp(-T(cfg.alpha) * tt)) / T(cfg.dv);
  }

  // host [B][n] <-> device rows with pitch
  int rows_h2d(T* dst, size_t pitch, const void* src, int n) {
    HIP_TRY(hipMemcpy2DAsync(dst, pitch * sizeof(T), src, size_t(n) * sizeof(T), size_t(n) * sizeof(T), cfg.batch,
                             hipMemcpyHostToDevice, stream));
    HIP_TRY(hipStreamSynchronize(stream));
    return 0;
  }
  int rows_d2h(void* dst, const T* src, size_t pitch, int n, size_t rows) {
    if (!dst) return 0;
    HIP_TRY(hipMemcpy2DAsync(dst, size_t(n) * sizeof(T), src, pitch * sizeof(T), size_t(n) * sizeof(T), rows,
                             hipMemcpyDeviceToHost, stream));
    HIP_TRY(hipStreamSynchronize(stream));
    return 0;
  }
  int replicate(T* dst, size_t pitch, const void* src, int n, int reps, int per_instance) {
    const size_t cnt = size_t(per_instance ? cfg.batch : 1) * n;
    if (int rc = grow(&stage, &stage_n, cnt)) return rc;
    HIP_TRY(hipMemcpyAsync(stage, src, cnt * sizeof(T), hipMemcpyHostToDevice, stream));
    replicate_rows_im<T><<<dim3(4, cfg.batch), 256, 0, stream>>>(dst, pitch, stage, cfg.batch, n, reps, !per_instance);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipStreamSynchronize(stream));
    return 0;
  }

  int set_ptau(const void* p, int per_instance, bool repeat) override {
    HIP_TRY(hipSetDevice(cfg.device));
    if (np == 0) return 0;
    if (!p) return fail(CGMRES_HIP_EINVAL, "set_ptau: null pointer");
    const int all = np * (cfg.dv + 1);
    return repeat ? replicate(P.ptau, all, p, np, cfg.dv + 1, per_instance) : replicate(P.ptau, all, p, all, 1, per_instance);
  }
  int init_u0(const void* u0, int per_instance) override {
    HIP_TRY(hipSetDevice(cfg.device));
    if (!u0) return fail(CGMRES_HIP_EINVAL, "init_u0: null pointer");
    return replicate(P.U, P.Lg, u0, nu, cfg.dv, per_instance);
  }
  int init_u0_newton(void* u0, const void* x0, const void* p0, int n_loop) override {
    HIP_TRY(hipSetDevice(cfg.device));
    if (!u0 || !x0 || (np && !p0)) return fail(CGMRES_HIP_EINVAL, "init_u0_newton: null pointer");
    if (n_loop < 0) return fail(CGMRES_HIP_EINVAL, "init_u0_newton: n_loop < 0");
    const size_t B = cfg.batch;
    if (int rc = grow(&stage2, &stage2_n, B * (nu + nx + np))) return rc;
    T *du = stage2, *dx = du + B * nu, *dp = dx + B * nx;
    HIP_TRY(hipMemcpyAsync(du, u0, B * nu * sizeof(T), hipMemcpyHostToDevice, stream));
    HIP_TRY(hipMemcpyAsync(dx, x0, B * nx * sizeof(T), hipMemcpyHostToDevice, stream));
    if (np) HIP_TRY(hipMemcpyAsync(dp, p0, B * np * sizeof(T), hipMemcpyHostToDevice, stream));
    newton_u0_kernel<M, T><<<dim3((B + 63) / 64), 64, 0, stream>>>(du, dx, dp, cfg.batch, n_loop);
    HIP_TRY(hipGetLastError());
    replicate_rows_im<T><<<dim3(4, cfg.batch), 256, 0, stream>>>(P.U, P.Lg, du, cfg.batch, nu, cfg.dv, 0);  // cgmres.hpp:75
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpyAsync(u0, du, B * nu * sizeof(T), hipMemcpyDeviceToHost, stream));
    HIP_TRY(hipStreamSynchronize(stream));
    return 0;
  }

  // n consecutive ticks in one launch (n > 1 needs the on-device plant: x_next != nullptr)
  int launch_ticks(T* u_out, const T* x_in, T* x_next, int n) {
    P.mode = WG_TICK;
    P.x_in = x_in, P.u_out = u_out, P.x_next = x_next;
    P.n_ticks = n;
    for (int k = 0; k < n; ++k) {
      P.dtau_tab[2 * k] = dtau_of(t + P.h);  // cgmres.hpp:88
      P.dtau_tab[2 * k + 1] = dtau_of(t);    // cgmres.hpp:91
      t = t + P.dt;                          // cgmres.hpp:107
    }
    P.dtau_h = P.dtau_tab[0], P.dtau_0 = P.dtau_tab[1];
    if (wave)
      k_tick<<<dim3((cfg.batch + kWaveWpb - 1) / kWaveWpb), dim3(64 * kWaveWpb), lds_bytes_tick(), stream>>>(P);
    else
      k_tick<<<grid(), block(), lds_bytes, stream>>>(P);
    HIP_TRY(hipGetLastError());
    return 0;
  }
  int launch_tick(T* u_out, const T* x_in, T* x_next) { return launch_ticks(u_out, x_in, x_next, 1); }
  int control_device(void* u, const void* x, void* x_next) override {
    HIP_TRY(hipSetDevice(cfg.device));
    if (!u || !x) return fail(CGMRES_HIP_EINVAL, "control: null pointer");
    return launch_tick(static_cast<T*>(u), static_cast<const T*>(x), static_cast<T*>(x_next));
  }
  int control_host(void* u, const void* x) override {
    HIP_TRY(hipSetDevice(cfg.device));
    if (!u || !x) return fail(CGMRES_HIP_EINVAL, "control: null pointer");
    if (cfg.batch <= kPinnedIoMaxBatch) {
      const size_t bx = size_t(cfg.batch) * nx * sizeof(T), bu = size_t(cfg.batch) * nu * sizeof(T);
      if (!pin_x) {
        HIP_TRY(hipHostMalloc(reinterpret_cast<void**>(&pin_x), bx, hipHostMallocMapped));
        HIP_TRY(hipHostMalloc(reinterpret_cast<void**>(&pin_u), bu, hipHostMallocMapped));
        HIP_TRY(hipHostGetDevicePointer(reinterpret_cast<void**>(&pin_x_dev), pin_x, 0));
        HIP_TRY(hipHostGetDevicePointer(reinterpret_cast<void**>(&pin_u_dev), pin_u, 0));
      }
      std::memcpy(pin_x, x, bx);
      if (int rc = launch_tick(pin_u_dev, pin_x_dev, nullptr)) return rc;
      HIP_TRY(hipStreamSynchronize(stream));
      std::memcpy(u, pin_u, bu);
      return 0;
    }
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
    const T* seq = all ? static_cast<const T*>(ptau_seq) : nullptr;
    const size_t per_tick = size_t(per_instance ? cfg.batch : 1) * all;
    P.pseq_tick = per_tick, P.pseq_inst = per_instance ? all : 0;
    int rc = 0;
    for (int i = 0; i < n_ticks && !rc; i += CGM_FUSE_MAX) {
      const int n = n_ticks - i < CGM_FUSE_MAX ? n_ticks - i : CGM_FUSE_MAX;
      P.ptau_seq = seq ? seq + size_t(i) * per_tick : nullptr;  // the kernel reloads ptau at the top of every tick
      if (binning && have_counts) {
        bin_by_count_kernel<0><<<1, 1024, 0, stream>>>(perm_dev, P.n_ax, cfg.batch, cfg.k_max);
        HIP_TRY(hipGetLastError());
        P.perm = perm_dev;
      }
      rc = launch_ticks(static_cast<T*>(u), static_cast<const T*>(x), static_cast<T*>(x), n);
      P.perm = nullptr;
      have_counts = true;
    }
    P.ptau_seq = nullptr;
    if (!rc && seq && n_ticks > 0) {  // the handle keeps the last tick's ptau, as set_ptau would (cgmres.hpp:36-39)
      replicate_rows_im<T><<<dim3(4, cfg.batch), 256, 0, stream>>>(P.ptau, all, seq + size_t(n_ticks - 1) * per_tick,
                                                                    cfg.batch, all, 1, !per_instance);
      HIP_TRY(hipGetLastError());
    }
    return rc;
  }

  double time() const override { return double(t); }
  int get_state(double* tt, void* U, void* dUdt) override {
    HIP_TRY(hipSetDevice(cfg.device));
    if (tt) *tt = double(t);
    if (int rc = rows_d2h(U, P.U, P.Lg, L, cfg.batch)) return rc;
    return rows_d2h(dUdt, P.dUdt, P.Lg, L, cfg.batch);
  }
  int set_state(double tt, const void* U, const void* dUdt) override {
    HIP_TRY(hipSetDevice(cfg.device));
    t = T(tt);
    if (U)
      if (int rc = rows_h2d(P.U, P.Lg, U, L)) return rc;
    if (dUdt)
      if (int rc = rows_h2d(P.dUdt, P.Lg, dUdt, L)) return rc;
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
    if (V) {  // rows are pair-interleaved on the device (WgCtx::load_vec): element r + 16 m sits at (m/2)*32 + 2r + (m&1)
      const size_t rows = size_t(cfg.batch) * k1;
      std::vector<T> tmp(rows * P.Lv);
      HIP_TRY(hipMemcpyAsync(tmp.data(), P.V, tmp.size() * sizeof(T), hipMemcpyDeviceToHost, stream));
      HIP_TRY(hipStreamSynchronize(stream));
      T* dst = static_cast<T*>(V);
      for (size_t q = 0; q < rows; ++q)
        for (int e = 0; e < L; ++e) {
          const int rr = e & 15, m = e >> 4;
          dst[q * L + e] = tmp[q * P.Lv + (m >> 1) * 32 + 2 * rr + (m & 1)];
        }
    }
    if (H)
      if (int rc = rows_d2h(H, P.kry, ks_all, k1 * k1, cfg.batch)) return rc;
    if (rho)
      if (int rc = rows_d2h(rho, P.kry + k1 * k1, ks_all, k1, cfg.batch)) return rc;
    if (g)
      if (int rc = rows_d2h(g, P.kry + k1 * k1 + k1, ks_all, 3 * cfg.k_max, cfg.batch)) return rc;
    return 0;
  }

  // ---- white-box hooks: same device functions, selected by P.mode ---------------------------------
  int run_hook(int mode, const void* in0, const void* in1, void* out, const void* x, T dtau) {
    const size_t n = size_t(cfg.batch) * L;
    if (int rc = grow(&stage2, &stage2_n, 3 * n)) return rc;
    T *d0 = stage2, *d1 = stage2 + n, *dout = stage2 + 2 * n;
    if (in0) HIP_TRY(hipMemcpyAsync(d0, in0, n * sizeof(T), hipMemcpyHostToDevice, stream));
    if (in1) HIP_TRY(hipMemcpyAsync(d1, in1, n * sizeof(T), hipMemcpyHostToDevice, stream));
    if (x) HIP_TRY(hipMemcpyAsync(x_dev, x, size_t(cfg.batch) * nx * sizeof(T), hipMemcpyHostToDevice, stream));
    P.mode = mode;
    P.hook_in0 = d0, P.hook_in1 = d1, P.hook_out = out ? dout : nullptr, P.hook_dtau = dtau;
    P.x_in = x ? x_dev : nullptr, P.u_out = nullptr, P.x_next = nullptr;
    P.dtau_h = dtau_of(t + P.h);
    P.dtau_0 = dtau_of(t);
    const int keep = P.fh_hbm;
    P.fh_hbm = fh_hbm_for_hooks, P.lds_bytes = int(lds_bytes_hook);  // the hook kernels use the full / fh_hbm plan
    k_hook<<<grid(), block(), lds_bytes_hook, stream>>>(P);
    P.fh_hbm = keep, P.lds_bytes = int(lds_bytes);
    HIP_TRY(hipGetLastError());
    if (out) HIP_TRY(hipMemcpyAsync(out, dout, n * sizeof(T), hipMemcpyDeviceToHost, stream));
    HIP_TRY(hipStreamSynchronize(stream));
    return 0;
  }
  int hook_F(void* ret, const void* U, const void* x, double tt) override {
    HIP_TRY(hipSetDevice(cfg.device));
    return run_hook(WG_HOOK_F, U, nullptr, ret, x, dtau_of(T(tt)));
  }
  int hook_prepare(void* b, const void* x) override {
    HIP_TRY(hipSetDevice(cfg.device));
    return run_hook(WG_HOOK_PREPARE, nullptr, nullptr, b, x, T(0));
  }
  int hook_Ax(void* out, const void* v) override {
    HIP_TRY(hipSetDevice(cfg.device));
    return run_hook(WG_HOOK_AX, v, nullptr, out, nullptr, T(0));
  }
  int hook_gmres(void* x, const void* b) override {
    HIP_TRY(hipSetDevice(cfg.device));
    return run_hook(WG_HOOK_GMRES, x, b, x, nullptr, T(0));
  }
};

}  // namespace cgm
