// mmvq_fused_q8_0.hip — the persistent grouped mat-vec kernels (mmvq_fused.h) for the weight format(s) T_Q8_0 / T_Q8_0:
// one translation unit per format so that the families compile in parallel.
#include "mmvq_fused.h"

namespace mi355x {

MI_DEFINE_FUSED_LAUNCHER(launch_fused_q8_0, T_Q8_0, T_Q8_0, true)

} // namespace mi355x
