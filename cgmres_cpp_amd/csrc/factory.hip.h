// One factory per (model, scalar type): each lives in its own translation unit (inst_*.hip) so the kernel
// instantiations compile in parallel.  Returns nullptr when the requested variant cannot serve the sizes;
// *resolved receives the variant actually chosen (1 = lane, 2 = wg).
#pragma once
#include "ctx_common.hip.h"

namespace cgm {
cgmres_hip_ctx* make_pendulum_f64(const cgmres_hip_config& cfg, int* resolved);
cgmres_hip_ctx* make_pendulum_f32(const cgmres_hip_config& cfg, int* resolved);
cgmres_hip_ctx* make_msd_f64(const cgmres_hip_config& cfg, int* resolved);
cgmres_hip_ctx* make_msd_f32(const cgmres_hip_config& cfg, int* resolved);
cgmres_hip_ctx* make_semiactive_f64(const cgmres_hip_config& cfg, int* resolved);
cgmres_hip_ctx* make_semiactive_f32(const cgmres_hip_config& cfg, int* resolved);
#ifdef CGM_STAMPS
long long* debug_stamps_ptr();  // diagnostic build only: stamp buffer of the last pendulum/f64 wg context
#endif
}  // namespace cgm
