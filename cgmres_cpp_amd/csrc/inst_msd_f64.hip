// Kernel instantiations of the msd model in double precision (both mappings).
#include "factory_impl.hip.h"

namespace cgm {
cgmres_hip_ctx* make_msd_f64(const cgmres_hip_config& cfg, int* resolved) {
  return make_variant<MsdDev<double>, double>(cfg, resolved);
}
#if defined(CGM_STAMPS) && defined(CGM_STAMPS_MODEL) && CGM_STAMPS_MODEL == 1
long long* debug_stamps_ptr() {  // diagnostic build: tools/phase_stamps.py --model msd
  void* p = nullptr;
  return hipGetSymbolAddress(&p, HIP_SYMBOL(g_cgm_stamps)) == hipSuccess ? static_cast<long long*>(p) : nullptr;
}
#endif
}  // namespace cgm
