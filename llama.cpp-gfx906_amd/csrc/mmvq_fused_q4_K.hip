// mmvq_fused_q4_K.hip — the persistent grouped mat-vec kernels (mmvq_fused.h) for the weight format(s) T_Q4_K / T_Q4_K:
// one translation unit per format so that the families compile in parallel.
#include "mmvq_fused.h"

namespace mi355x {

MI_DEFINE_FUSED_LAUNCHER(launch_fused_q4_K, T_Q4_K, T_Q4_K, true)

} // namespace mi355x
