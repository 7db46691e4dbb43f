// Included by the inst_*.hip translation units only.
#pragma once
#include "ctx_lane.hip.h"
#include "ctx_wg.hip.h"
#include "factory.hip.h"

namespace cgm {
template <class M, class T>
cgmres_hip_ctx* make_variant(const cgmres_hip_config& cfg, int* resolved) {
  int ipw;
  size_t bytes;
  const bool wg_ok = CtxWg<M, T>::supported(cfg, &ipw, &bytes);
  const int v = cfg.variant == 0 ? (wg_ok ? 2 : 1) : cfg.variant;
  *resolved = v;
  if (v == 2 || v == 3 || v == 4) return wg_ok ? new CtxWg<M, T>() : nullptr;  // (CtxWg::init picks / checks the LDS plan: 2 or 3)
  return new CtxLane<M, T>();
}
}  // namespace cgm
