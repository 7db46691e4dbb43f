// Cross-lane moves inside a 16-lane DPP row (gfx9 DPP controls), for fp64 (two 32-bit moves) and fp32.
#pragma once
#include <hip/hip_runtime.h>

namespace cgm {

constexpr int DPP_QUAD_SWAP1 = 0xB1;      // quad_perm [1,0,3,2]: lane ^ 1
constexpr int DPP_QUAD_SWAP2 = 0x4E;      // quad_perm [2,3,0,1]: lane ^ 2
constexpr int DPP_QUAD_BCAST3 = 0xFF;     // quad_perm [3,3,3,3]
constexpr int DPP_ROW_HALF_MIRROR = 0x141;
constexpr int DPP_ROW_MIRROR = 0x140;

template <int CTRL>
__device__ __forceinline__ double dpp_move(double x) {
  int lo = __double2loint(x), hi = __double2hiint(x);
  // all lanes of the row are active wherever this is used and every control below reads a valid lane, so the
  // "old" operand is irrelevant: passing 0 with bound_ctrl lets hipcc emit the DPP move without a pre-copy
  lo = __builtin_amdgcn_update_dpp(0, lo, CTRL, 0xf, 0xf, true);
  hi = __builtin_amdgcn_update_dpp(0, hi, CTRL, 0xf, 0xf, true);
  return __hiloint2double(hi, lo);
}
template <int CTRL>
__device__ __forceinline__ float dpp_move(float x) {
  int v = __float_as_int(x);
  v = __builtin_amdgcn_update_dpp(0, v, CTRL, 0xf, 0xf, true);
  return __int_as_float(v);
}

// Sum over the 4 lanes of a quad / the 16 lanes of a row.  Every lane receives the same bits: each step adds
// the two operands of a commutative pair.
template <class T>
__device__ __forceinline__ T quad_sum(T x) {
  x += dpp_move<DPP_QUAD_SWAP1>(x);
  x += dpp_move<DPP_QUAD_SWAP2>(x);
  return x;
}
template <class T>
__device__ __forceinline__ T row16_sum(T x) {
  x = quad_sum(x);
  x += dpp_move<DPP_ROW_HALF_MIRROR>(x);
  x += dpp_move<DPP_ROW_MIRROR>(x);
  return x;
}

}  // namespace cgm
