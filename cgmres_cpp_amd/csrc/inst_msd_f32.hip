// Kernel instantiations of the msd model in float precision (both mappings).
#include "factory_impl.hip.h"

namespace cgm {
cgmres_hip_ctx* make_msd_f32(const cgmres_hip_config& cfg, int* resolved) {
  return make_variant<MsdDev<float>, float>(cfg, resolved);
}
}  // namespace cgm
