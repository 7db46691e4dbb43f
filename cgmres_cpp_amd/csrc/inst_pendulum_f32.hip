// Kernel instantiations of the pendulum model in float precision (both mappings).
#include "factory_impl.hip.h"

namespace cgm {
cgmres_hip_ctx* make_pendulum_f32(const cgmres_hip_config& cfg, int* resolved) {
  return make_variant<PendulumDev<float>, float>(cfg, resolved);
}
}  // namespace cgm
