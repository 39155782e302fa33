// mmvq_fused_q8_0_q4_K.hip — the persistent grouped mat-vec kernels (mmvq_fused.h) for launches that mix T_Q8_0 and T_Q4_K groups
// (Mixtral-8x7B Q4_K_M: attn_q Q4_K with attn_k / attn_v bumped to Q8_0, src/llama-quant.cpp:262-271) — the two formats quantize the
// activation differently (Q8_0 / Q8_K blocks); each workgroup builds the image of its own group's format.
#include "mmvq_fused.h"

namespace mi355x {

MI_DEFINE_FUSED_LAUNCHER(launch_fused_q8_0_q4_K, T_Q8_0, T_Q4_K, false)

} // namespace mi355x
