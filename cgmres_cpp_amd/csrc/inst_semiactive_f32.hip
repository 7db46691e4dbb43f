// Kernel instantiations of the semiactive model in float precision (both mappings).
#include "factory_impl.hip.h"

namespace cgm {
cgmres_hip_ctx* make_semiactive_f32(const cgmres_hip_config& cfg, int* resolved) {
  return make_variant<SemiactiveDev<float>, float>(cfg, resolved);
}
}  // namespace cgm
