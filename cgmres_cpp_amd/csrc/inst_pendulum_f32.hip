// Kernel instantiations of the pendulum model in float precision (both mappings).
#include "factory_impl.hip.h"

namespace cgm {
cgmres_hip_ctx* make_pendulum_f32(const cgmres_hip_config& cfg, int* resolved) {
  return make_variant<PendulumDev<float>, float>(cfg, resolved);
}
#if defined(CGM_STAMPS) && defined(CGM_STAMPS_MODEL) && CGM_STAMPS_MODEL == 2
long long* debug_stamps_ptr() {  // diagnostic build: tools/phase_stamps.py --model=pendulum32
  void* p = nullptr;
  return hipGetSymbolAddress(&p, HIP_SYMBOL(g_cgm_stamps)) == hipSuccess ? static_cast<long long*>(p) : nullptr;
}
#endif
}  // namespace cgm
