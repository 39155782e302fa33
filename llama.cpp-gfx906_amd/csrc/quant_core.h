// quant_core.h — f32 -> Q8_0 / Q8_K activation quantization of ONE 4-float fragment per lane, shared by
// quantize_act.hip (stand-alone kernel) and decode_fused.hip (fused into the producer / the mat-vec prologue).
// Arithmetic: see quantize_act.hip's header (bit-exact restatement of the reference quantizers).
#pragma once

#include "dev_common.h"

namespace mi355x {

static __device__ __forceinline__ uint32_t pack4_i8(int q0, int q1, int q2, int q3) {
    return (uint32_t)(q0 & 0xFF) | ((uint32_t)(q1 & 0xFF) << 8) | ((uint32_t)(q2 & 0xFF) << 16) | ((uint32_t)(q3 & 0xFF) << 24);
}

// Q8_0: 8 consecutive lanes hold one 32-element block (4 floats each). All 8 lanes of a group must call this.
// Returns the packed quants; d_out/bsum_out are valid in every lane of the group.
static __device__ __forceinline__ uint32_t quant_frag_q8_0(float4v v, float & d_out, int & bsum_out) {
    float amax = fmaxf(fmaxf(fabsf(v.x), fabsf(v.y)), fmaxf(fabsf(v.z), fabsf(v.w)));
    amax = group8_max(amax);
    const float dd = amax / 127.0f;
    const float id = dd != 0.0f ? 1.0f/dd : 0.0f;
    const int q0 = (int) roundf(v.x*id), q1 = (int) roundf(v.y*id), q2 = (int) roundf(v.z*id), q3 = (int) roundf(v.w*id);
    bsum_out = group8_sum_i(q0 + q1 + q2 + q3);
    d_out = f16_bits_to_f32(f32_to_f16_bits(dd));   // the CPU path stores d as f16 and reads it back
    return pack4_i8(q0, q1, q2, q3);
}

// Q8_K: the 64 lanes of a wave hold one 256-element block (4 floats each, lane order = element order).
// Returns the packed quants; d_out is wave-uniform; bsum16_out is the sum of this lane's 16-element group (valid in all 4 lanes of the quad).
static __device__ __forceinline__ uint32_t quant_frag_q8_K(float4v v, float & d_out, int & bsum16_out) {
    float amax = fabsf(v.x), mx = v.x;   // first element of largest magnitude (strict > keeps the first)
    if (fabsf(v.y) > amax) { amax = fabsf(v.y); mx = v.y; }
    if (fabsf(v.z) > amax) { amax = fabsf(v.z); mx = v.z; }
    if (fabsf(v.w) > amax) { amax = fabsf(v.w); mx = v.w; }
    const float wmax = wave_max(amax);
    // branch-free (an all-zero block takes the same instructions with a stand-in scale and is zeroed at the end): several chunks
    // quantized back to back can then be interleaved by the scheduler instead of running as dependent chains one after the other
    const bool zero = wmax == 0.0f;
    const unsigned long long ball = __ballot(amax == wmax);
    const int first = __builtin_ctzll(ball);
    const float maxv = zero ? 1.0f : readlane_f(mx, first);
    const float iscale = -127.0f/maxv;
    int q0 = __float2int_rn(iscale*v.x), q1 = __float2int_rn(iscale*v.y), q2 = __float2int_rn(iscale*v.z), q3 = __float2int_rn(iscale*v.w);
    q0 = min(127, q0); q1 = min(127, q1); q2 = min(127, q2); q3 = min(127, q3);
    int s = q0 + q1 + q2 + q3;
    s += dpp_i<0xB1>(s);
    s += dpp_i<0x4E>(s);
    bsum16_out = zero ? 0 : s;
    d_out = zero ? 0.0f : 1.0f/iscale;
    return zero ? 0u : pack4_i8(q0, q1, q2, q3);
}

// the two halves of quant_store_chunk256, so that a caller can quantize several chunks first and store them afterwards
template <int ACT>
static __device__ __forceinline__ uint32_t quant_chunk256(float4v v, float & dd, int & bsum) {
    return ACT == T_Q8_0 ? quant_frag_q8_0(v, dd, bsum) : quant_frag_q8_K(v, dd, bsum);
}
template <int ACT>
static __device__ __forceinline__ void store_chunk256(uint32_t p, float dd, int bsum, int c, int lane, int8_t * qs, float * d, int16_t * bs) {
    *(uint32_t *) (qs + c*256 + lane*4) = p;
    if (ACT == T_Q8_0) {
        if ((lane & 7) == 0) { d[c*8 + (lane >> 3)] = dd; bs[c*8 + (lane >> 3)] = (int16_t) bsum; }
    } else {
        if ((lane & 3) == 0) bs[c*16 + (lane >> 2)] = (int16_t) bsum;
        if (lane == 0) d[c] = dd;
    }
}

// quantize the 256-element chunk `c` of one row held as 4 floats per lane and store it
//   qs: int8 [k] row base, d: scales row base, bs: bsums row base (global or LDS)
template <int ACT>
static __device__ __forceinline__ void quant_store_chunk256(float4v v, int c, int lane, int8_t * qs, float * d, int16_t * bs) {
    float dd; int bsum;
    if (ACT == T_Q8_0) {
        const uint32_t p = quant_frag_q8_0(v, dd, bsum);
        *(uint32_t *) (qs + c*256 + lane*4) = p;
        if ((lane & 7) == 0) { d[c*8 + (lane >> 3)] = dd; bs[c*8 + (lane >> 3)] = (int16_t) bsum; }
    } else {
        const uint32_t p = quant_frag_q8_K(v, dd, bsum);
        *(uint32_t *) (qs + c*256 + lane*4) = p;
        if ((lane & 3) == 0) bs[c*16 + (lane >> 2)] = (int16_t) bsum;
        if (lane == 0) d[c] = dd;
    }
}

} // namespace mi355x
