// Kernel instantiations of the semiactive model in double precision (both mappings).
#include "factory_impl.hip.h"

namespace cgm {
cgmres_hip_ctx* make_semiactive_f64(const cgmres_hip_config& cfg, int* resolved) {
  return make_variant<SemiactiveDev<double>, double>(cfg, resolved);
}
}  // namespace cgm
