// Kernel instantiations of the msd model in double precision (both mappings).
#include "factory_impl.hip.h"

namespace cgm {
cgmres_hip_ctx* make_msd_f64(const cgmres_hip_config& cfg, int* resolved) {
  return make_variant<MsdDev<double>, double>(cfg, resolved);
}
}  // namespace cgm
