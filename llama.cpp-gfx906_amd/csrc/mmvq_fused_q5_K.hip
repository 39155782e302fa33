// mmvq_fused_q5_K.hip — the persistent grouped mat-vec kernels (mmvq_fused.h) for the weight format(s) T_Q5_K / T_Q5_K:
// one translation unit per format so that the families compile in parallel.
#include "mmvq_fused.h"

namespace mi355x {

MI_DEFINE_FUSED_LAUNCHER(launch_fused_q5_K, T_Q5_K, T_Q5_K, true)

} // namespace mi355x
