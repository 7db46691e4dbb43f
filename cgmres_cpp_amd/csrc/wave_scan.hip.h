// Inclusive scans over the 64 lanes of one wavefront with DPP moves only (no LDS, no barrier) — the building block of the
// "wave" mapping (tick_wave.hip.h), where lane s of a wave carries stage s of ONE controller's horizon and the serial
// recurrences of cgmres.hpp:132-153 become scans of affine maps.
//
// Element i (lane i) is the map that takes the value of stage i to stage i + 1; after the scan lane i holds the
// composition  a_i o a_(i-1) o ... o a_0.  Kogge-Stone in six steps: row_shr:1,2,4,8 inside the 16-lane DPP rows, then
// row_bcast:15 into rows 1 and 3 and row_bcast:31 into rows 2 and 3.  A lane without a partner in a step reads ZEROS
// (bound_ctrl / a zeroed destination under the row mask); the maps are therefore carried as  x -> x + D x + c  (D = M - I),
// for which the all-zero element is the identity:
//     (D, c) o (Dp, cp) = (D + Dp + D Dp,  c + cp + D cp).
// Information only travels from lower to higher lanes: lanes above the last stage may hold anything.
#pragma once
#include <hip/hip_runtime.h>

#include "models.hip.h"  // fma_t

namespace cgm {

constexpr int DPP_WAVE_SHR1 = 0x138;  // lane i <- lane i-1 across the whole wave
constexpr int DPP_WAVE_SHL1 = 0x130;  // lane i <- lane i+1
constexpr int DPP_ROW_BCAST15 = 0x142, DPP_ROW_BCAST31 = 0x143;

template <int CTRL, int ROW_MASK>
__device__ __forceinline__ double dpp_zero_fill(double x) {
  int lo = __double2loint(x), hi = __double2hiint(x);
  lo = __builtin_amdgcn_update_dpp(0, lo, CTRL, ROW_MASK, 0xf, true);
  hi = __builtin_amdgcn_update_dpp(0, hi, CTRL, ROW_MASK, 0xf, true);
  return __hiloint2double(hi, lo);
}
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float dpp_zero_fill(float x) {
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), CTRL, ROW_MASK, 0xf, true));
}
// The same for a control that writes EVERY lane (row mask 0xf: the in-row shifts): the move needs no previous value of
// its destination, so none is materialised (update_dpp's `old` operand costs a v_mov_b32 per half).
template <int CTRL>
__device__ __forceinline__ double dpp_row_zero_fill(double x) {
  const int lo = __builtin_amdgcn_mov_dpp(__double2loint(x), CTRL, 0xf, 0xf, true);
  const int hi = __builtin_amdgcn_mov_dpp(__double2hiint(x), CTRL, 0xf, 0xf, true);
  return __hiloint2double(hi, lo);
}
template <int CTRL>
__device__ __forceinline__ float dpp_row_zero_fill(float x) {
  return __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(x), CTRL, 0xf, 0xf, true));
}
// the partner's value in step STEP (0..5) of the scan, 0 where the lane has none
// REV (in-row steps only): the partner is the lane ABOVE — a scan from the highest lane of a 16-lane row downwards
template <int STEP, bool REV = false, class T>
__device__ __forceinline__ T scan_partner(T x) {
  static_assert(STEP >= 0 && STEP < 6, "six steps cover 64 lanes");
  static_assert(!REV || STEP < 4, "reverse scans stay inside a row");
  if constexpr (REV)
    return dpp_row_zero_fill<0x100 + (1 << STEP)>(x);  // row_shl:1,2,4,8
  else if constexpr (STEP < 4)
    return dpp_row_zero_fill<0x110 + (1 << STEP)>(x);  // row_shr:1,2,4,8
  else if constexpr (STEP == 4)
    return dpp_zero_fill<DPP_ROW_BCAST15, 0xA>(x);
  else
    return dpp_zero_fill<DPP_ROW_BCAST31, 0xC>(x);
}

// lane i <- lane i-1; lane 0 <- first
template <class T>
__device__ __forceinline__ T wave_shift_up(T x, T first);
template <>
__device__ __forceinline__ double wave_shift_up<double>(double x, double first) {
  int lo = __double2loint(x), hi = __double2hiint(x);
  lo = __builtin_amdgcn_update_dpp(__double2loint(first), lo, DPP_WAVE_SHR1, 0xf, 0xf, false);
  hi = __builtin_amdgcn_update_dpp(__double2hiint(first), hi, DPP_WAVE_SHR1, 0xf, 0xf, false);
  return __hiloint2double(hi, lo);
}
template <>
__device__ __forceinline__ float wave_shift_up<float>(float x, float first) {
  return __int_as_float(
      __builtin_amdgcn_update_dpp(__float_as_int(first), __float_as_int(x), DPP_WAVE_SHR1, 0xf, 0xf, false));
}

// value of lane `src` (wave-uniform index) in every lane
__device__ __forceinline__ double wave_bcast(double v, int src) {
  const int lo = __builtin_amdgcn_readlane(__double2loint(v), src);
  const int hi = __builtin_amdgcn_readlane(__double2hiint(v), src);
  return __hiloint2double(hi, lo);
}
__device__ __forceinline__ float wave_bcast(float v, int src) {
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), src));
}
// value of lane `src_lane` (per-lane index) through the LDS crossbar
__device__ __forceinline__ double wave_gather(double v, int src_lane) {
  const int lo = __builtin_amdgcn_ds_bpermute(src_lane * 4, __double2loint(v));
  const int hi = __builtin_amdgcn_ds_bpermute(src_lane * 4, __double2hiint(v));
  return __hiloint2double(hi, lo);
}
__device__ __forceinline__ float wave_gather(float v, int src_lane) {
  return __int_as_float(__builtin_amdgcn_ds_bpermute(src_lane * 4, __float_as_int(v)));
}

// Sum over all 64 lanes as a wave-uniform value: a reduction INTO lane 63 — row_shr:1,2,4,8 bring every row's sum to its
// lane 15, row_bcast:15 / row_bcast:31 (unmasked: only lane 63 has to be right) carry the row sums across — and one
// readlane pair.  18 + 2 instructions, and the result lives in scalar registers (a free operand of the lanes' FMAs).
template <class T>
__device__ __forceinline__ T wave_sum(T x) {
  x += dpp_zero_fill<0x111, 0xf>(x);
  x += dpp_zero_fill<0x112, 0xf>(x);
  x += dpp_zero_fill<0x114, 0xf>(x);
  x += dpp_zero_fill<0x118, 0xf>(x);
  x += dpp_zero_fill<DPP_ROW_BCAST15, 0xf>(x);
  x += dpp_zero_fill<DPP_ROW_BCAST31, 0xf>(x);
  return wave_bcast(x, 63);
}

// ---- scans ---------------------------------------------------------------------------------------------------------
// prefix sum: lane i <- c_0 + ... + c_i
template <class T>
__device__ __forceinline__ T scan_sum(T c) {
  c += scan_partner<0>(c);
  c += scan_partner<1>(c);
  c += scan_partner<2>(c);
  c += scan_partner<3>(c);
  c += scan_partner<4>(c);
  c += scan_partner<5>(c);
  return c;
}
// x' = a x + c_i with the SAME a in every element: lane i <- sum_j a^(i-j) c_j.  pw[t] = a^(length of the lane's own
// segment in step t) = a, a^2, a^4, a^8 (uniform), a^((i & 15) + 1), a^((i & 31) + 1): GeoPowers::make.
template <class T>
struct GeoPowers {
  T pw[6];
  __device__ __forceinline__ void make(T a) {
    // the multiplier part of the scan of identical elements (m = a): after steps 0..3 lane i holds a^((i & 15) + 1),
    // after step 4 a^((i & 31) + 1) — exactly the per-lane multipliers of the last two steps
    const T a2 = a * a, a4 = a2 * a2, a8 = a4 * a4;
    pw[0] = a, pw[1] = a2, pw[2] = a4, pw[3] = a8;
    T m = a;  // as deviation from one (zero-filled partners are then the identity): m = 1 + d
    T d = a - T(1);
    auto step = [&](T dp) { d = d + dp + d * dp; };
    step(scan_partner<0>(d));
    step(scan_partner<1>(d));
    step(scan_partner<2>(d));
    step(scan_partner<3>(d));
    pw[4] = T(1) + d;
    step(scan_partner<4>(d));
    pw[5] = T(1) + d;
    (void)m;
  }
};
template <class T>
__device__ __forceinline__ T scan_geo(T c, const GeoPowers<T>& G) {
  c = fma_t(G.pw[0], scan_partner<0>(c), c);
  c = fma_t(G.pw[1], scan_partner<1>(c), c);
  c = fma_t(G.pw[2], scan_partner<2>(c), c);
  c = fma_t(G.pw[3], scan_partner<3>(c), c);
  c = fma_t(G.pw[4], scan_partner<4>(c), c);
  c = fma_t(G.pw[5], scan_partner<5>(c), c);
  return c;
}

// 2 x 2 affine maps  y -> y + D y + c,  D = {d00, d01, d10, d11} row-major.  After the call c holds the composed offset
// of the lane's prefix (= the value of the recurrence started from 0); D is destroyed.
// LEVELS (optional): the lane's own D before every step, for scan_aff2_vec with the same matrices.
template <class T>
struct Aff2Levels {
  T d[6][4];
};
template <int STEP, bool REV = false, class T>
__device__ __forceinline__ void aff2_step_vec(T* c, const T* D) {
  const T p0 = scan_partner<STEP, REV>(c[0]), p1 = scan_partner<STEP, REV>(c[1]);
  const T n0 = fma_t(D[1], p1, fma_t(D[0], p0, c[0] + p0));
  const T n1 = fma_t(D[3], p1, fma_t(D[2], p0, c[1] + p1));
  c[0] = n0, c[1] = n1;
}
template <int STEP, bool REV = false, class T>
__device__ __forceinline__ void aff2_step_mat(T* D) {
  const T q0 = scan_partner<STEP, REV>(D[0]), q1 = scan_partner<STEP, REV>(D[1]), q2 = scan_partner<STEP, REV>(D[2]),
          q3 = scan_partner<STEP, REV>(D[3]);
  const T n0 = fma_t(D[1], q2, fma_t(D[0], q0, D[0] + q0));
  const T n1 = fma_t(D[1], q3, fma_t(D[0], q1, D[1] + q1));
  const T n2 = fma_t(D[3], q2, fma_t(D[2], q0, D[2] + q2));
  const T n3 = fma_t(D[3], q3, fma_t(D[2], q1, D[3] + q3));
  D[0] = n0, D[1] = n1, D[2] = n2, D[3] = n3;
}
template <int STEP, class T>
__device__ __forceinline__ void aff2_keep(Aff2Levels<T>* lv, const T* D) {
  if (lv) {
#pragma unroll
    for (int e = 0; e < 4; ++e) lv->d[STEP][e] = D[e];
  }
}
template <class T>
__device__ __forceinline__ void scan_aff2(T* D, T* c, Aff2Levels<T>* lv = nullptr) {
  aff2_keep<0>(lv, D), aff2_step_vec<0>(c, D), aff2_step_mat<0>(D);
  aff2_keep<1>(lv, D), aff2_step_vec<1>(c, D), aff2_step_mat<1>(D);
  aff2_keep<2>(lv, D), aff2_step_vec<2>(c, D), aff2_step_mat<2>(D);
  aff2_keep<3>(lv, D), aff2_step_vec<3>(c, D), aff2_step_mat<3>(D);
  aff2_keep<4>(lv, D), aff2_step_vec<4>(c, D), aff2_step_mat<4>(D);
  aff2_keep<5>(lv, D), aff2_step_vec<5>(c, D);  // (the last step needs no composed matrix)
}
// the same recurrence with the matrices of an earlier scan_aff2 (their levels), new offsets
template <class T>
__device__ __forceinline__ void scan_aff2_vec(T* c, const Aff2Levels<T>& lv) {
  aff2_step_vec<0>(c, lv.d[0]);
  aff2_step_vec<1>(c, lv.d[1]);
  aff2_step_vec<2>(c, lv.d[2]);
  aff2_step_vec<3>(c, lv.d[3]);
  aff2_step_vec<4>(c, lv.d[4]);
  aff2_step_vec<5>(c, lv.d[5]);
}

// ---- 4 x 4 affine maps with the SAME matrix in every element (linear time-invariant state / costate equations) ---------
// y' = y + D y + c_i.  The scan of such elements composes only matrices that are POWERS of M = I + D: step t of an in-row
// step uses M^(2^t), the two cross-row steps M^((i & 15) + 1) and M^((i & 31) + 1) — all found in one table
//     pw[n - 1][16] = M^n - I,  n = 1 .. 32   (row-major, this wave's LDS)
// built once per tick by a matrix-only scan of identical elements over the lanes 0..31 (power_table4), after which every
// sweep is a VECTOR-only scan: 4 partner values + 16 multiply-adds per step, the matrix read from the table.
template <class T>
__device__ __forceinline__ void compose4(T* D, const T* Q) {  // D <- D + Q + D Q
  T n[16];
#pragma unroll
  for (int r = 0; r < 4; ++r) {
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      T a = D[4 * r + c] + Q[4 * r + c];
#pragma unroll
      for (int k = 0; k < 4; ++k) a = fma_t(D[4 * r + k], Q[4 * k + c], a);
      n[4 * r + c] = a;
    }
  }
#pragma unroll
  for (int e = 0; e < 16; ++e) D[e] = n[e];
}
template <int STEP, class T>
__device__ __forceinline__ void power_step4(T* D) {
  T Q[16];
#pragma unroll
  for (int e = 0; e < 16; ++e) Q[e] = scan_partner<STEP>(D[e]);
  compose4(D, Q);
}
// pw <- powers of I + D0 (D0 wave-uniform); every lane takes part, lanes 0..31 store.  The caller fences before reading.
template <class T>
__device__ __forceinline__ void power_table4(T* pw, const T* D0, int lane) {
  T D[16];
#pragma unroll
  for (int e = 0; e < 16; ++e) D[e] = D0[e];
  power_step4<0>(D), power_step4<1>(D), power_step4<2>(D), power_step4<3>(D), power_step4<4>(D);
  if (lane < 32) {
#pragma unroll
    for (int e = 0; e < 16; ++e) pw[lane * 16 + e] = D[e];
  }
}
template <int STEP, class T>
__device__ __forceinline__ void const4_step(T* c, const T* D) {
  T p[4], n[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) p[r] = scan_partner<STEP>(c[r]);
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    T a = c[r] + p[r];
#pragma unroll
    for (int k = 0; k < 4; ++k) a = fma_t(D[4 * r + k], p[k], a);
    n[r] = a;
  }
#pragma unroll
  for (int r = 0; r < 4; ++r) c[r] = n[r];
}
// c <- the recurrence started from 0 with offsets c (lane i: value after element i), matrices from the table
template <class T>
__device__ __forceinline__ void scan_const4(T* c, const T* pw, int lane) {
  const4_step<0>(c, pw + 0 * 16);          // M^1
  const4_step<1>(c, pw + 1 * 16);          // M^2
  const4_step<2>(c, pw + 3 * 16);          // M^4
  const4_step<3>(c, pw + 7 * 16);          // M^8
  const4_step<4>(c, pw + (lane & 15) * 16);  // M^((i & 15) + 1)
  const4_step<5>(c, pw + (lane & 31) * 16);  // M^((i & 31) + 1)
}

}  // namespace cgm
