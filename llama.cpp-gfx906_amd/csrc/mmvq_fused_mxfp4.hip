// mmvq_fused_mxfp4.hip — the persistent grouped mat-vec kernels (mmvq_fused.h) for the weight format(s) T_MXFP4 / T_MXFP4:
// one translation unit per format so that the families compile in parallel.
#include "mmvq_fused.h"

namespace mi355x {

MI_DEFINE_FUSED_LAUNCHER(launch_fused_mxfp4, T_MXFP4, T_MXFP4, true)

} // namespace mi355x
