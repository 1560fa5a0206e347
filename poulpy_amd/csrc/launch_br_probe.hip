// launch_br_probe.hip — the rounding-margin instantiations of the one-kernel blind rotation (k_br_fused<.., PROBE = true>, device_br.hpp):
// every form br_forms.hpp lists, with the probe compiled into the carry phase.  Dispatched only while pz_module_set_margin_probe is on.
#include <hip/hip_runtime.h>

#include "internal.hpp"
#include "br_forms.hpp"

namespace pz {

int br_fused_launch_probe(pz_module* M, const BrFusedArgs& g, const BrFusedPlan& pl) { return br_fused_launch<true>(M, g, pl); }

}  // namespace pz
