// launch_tail_probe.hip — the rounding-margin instantiations of the fused tail (k_inv_tail<.., PROBE = true>, device_fft.hpp): every form
// tail_forms.hpp lists, with the probe block compiled in.  Dispatched only while pz_module_set_margin_probe is on.
#include <hip/hip_runtime.h>

#include "internal.hpp"
#include "tail_forms.hpp"

namespace pz {

int tail_launch_form_probe(pz_module* M, const TailArgs& g, int blocks, const TailForm& f) { return tail_launch_form<true>(M, g, blocks, f); }

}  // namespace pz
