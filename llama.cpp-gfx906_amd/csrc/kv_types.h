// kv_types.h — the KV cache's element types as the attention kernels read them (FLASH_ATTN_EXT's type_KV: F16, BF16, Q8_0, Q4_0 —
// tests/test-backend-ops.cpp:6081-6087; -ctk / -ctv of the host, src/llama-kv-cache-unified.cpp). A head's row of HD elements is HD/32
// blocks (or HD 2-byte floats); a thread takes the 8 consecutive elements [8 e8, 8 e8 + 8): kv_raw8 is the load (issued in batches,
// nothing touched), kv_cvt8 the conversion to f32 (the reference's dequantize_row_*: d*q for Q8_0, d*(nibble - 8) for Q4_0 whose
// elements 0..15 are the low and 16..31 the high nibbles of its 16 bytes — ggml/src/ggml-quants.c, restated in oracle/ggml_oracle.c).
#pragma once

#include "dev_common.h"
#include "kernels.h"

namespace mi355x {

template <int TY> static __device__ __forceinline__ int4v kv_raw8(const char * row, int e8) {
    if constexpr (TY == T_F16 || TY == T_BF16) return *(const int4v *) (row + (size_t) e8*16);
    else if constexpr (TY == T_Q8_0) {
        const char * b = row + (size_t)(e8 >> 2)*34;
        const int2v qq = ld_b64(b + 2 + (e8 & 3)*8);
        return int4v{ qq.x, qq.y, (int) ld_u16(b), 0 };
    } else {      // Q4_0
        const char * b = row + (size_t)(e8 >> 2)*18;
        const int2v qq = ld_b64(b + 2 + (e8 & 1)*8);
        return int4v{ qq.x, qq.y, (int) ld_u16(b), 0 };
    }
}
template <int TY> static __device__ __forceinline__ void kv_cvt8(const int4v r, int e8, float (&f)[8]) {
    if constexpr (TY == T_F16) {
        const uint32_t w[4] = { (uint32_t) r.x, (uint32_t) r.y, (uint32_t) r.z, (uint32_t) r.w };
#pragma unroll
        for (int i = 0; i < 4; i++) { f[2*i] = f16_bits_to_f32((uint16_t) w[i]); f[2*i + 1] = f16_bits_to_f32((uint16_t)(w[i] >> 16)); }
    } else if constexpr (TY == T_BF16) {
        const uint32_t w[4] = { (uint32_t) r.x, (uint32_t) r.y, (uint32_t) r.z, (uint32_t) r.w };
#pragma unroll
        for (int i = 0; i < 4; i++) { f[2*i] = __builtin_bit_cast(float, w[i] << 16); f[2*i + 1] = __builtin_bit_cast(float, w[i] & 0xFFFF0000u); }
    } else if constexpr (TY == T_Q8_0) {
        const float d = f16_bits_to_f32((uint16_t) r.z);
        const uint32_t w[2] = { (uint32_t) r.x, (uint32_t) r.y };
#pragma unroll
        for (int i = 0; i < 8; i++) f[i] = d*(float)(int8_t)(w[i >> 2] >> (8*(i & 3)));
    } else {
        const float d = f16_bits_to_f32((uint16_t) r.z);
        const int sh = (e8 & 2) ? 4 : 0;      // elements 16..31 of the block: the high nibbles
        const uint32_t w[2] = { (uint32_t) r.x >> sh, (uint32_t) r.y >> sh };
#pragma unroll
        for (int i = 0; i < 8; i++) f[i] = d*(float)((int)((w[i >> 2] >> (8*(i & 3))) & 0xF) - 8);
    }
}

} // namespace mi355x
