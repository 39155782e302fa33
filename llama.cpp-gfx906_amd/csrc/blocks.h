// blocks.h — GGUF block formats as the backend sees them in HBM (raw GGUF bytes,
// unmodified: src/llama-model-loader.cpp:1060 uploads them as-is).
// Layouts: gguf-py/gguf/quants.py (class per type, cited below); sizes:
// gguf-py/gguf/constants.py:2839-2872. Shared by host and device code.
#pragma once

#include <stdint.h>

#define MI_QK_K 256
#define MI_K_SCALE_SIZE 12

#pragma pack(push, 1)
// Q4_0: 32 el / 18 B — quants.py:241-251. el j<16 = low nibble of qs[j], j>=16 = high nibble; x = d*(q-8)
struct block_q4_0 { uint16_t d; uint8_t qs[16]; };
// Q8_0: 32 el / 34 B — quants.py:396-401. x = d*q
struct block_q8_0 { uint16_t d; int8_t qs[32]; };
// Q4_K: 256 el / 144 B — quants.py:504-522. 8 sub-blocks of 32; qs in 4 groups of 32 B
// (low nibble = sub-block 2g, high = 2g+1); x = d*sc*q - dmin*m, sc/m 6-bit (quants.py:479-501)
struct block_q4_K { uint16_t d; uint16_t dmin; uint8_t scales[MI_K_SCALE_SIZE]; uint8_t qs[MI_QK_K/2]; };
// Q5_K: 256 el / 176 B — quants.py:527-549. as Q4_K plus qh: bit i of qh[j] = 5th bit of sub-block i, el j
struct block_q5_K { uint16_t d; uint16_t dmin; uint8_t scales[MI_K_SCALE_SIZE]; uint8_t qh[MI_QK_K/8]; uint8_t qs[MI_QK_K/2]; };
// Q6_K: 256 el / 210 B — quants.py:554-572. ql 128 B, qh 64 B, 16 int8 scales, f16 d LAST; x = d*sc*(q-32)
struct block_q6_K { uint8_t ql[MI_QK_K/2]; uint8_t qh[MI_QK_K/4]; int8_t scales[MI_QK_K/16]; uint16_t d; };
// MXFP4: 32 el / 17 B — quants.py:656-700. e8m0 scale byte, nibbles split like Q4_0; x = e8m0_half(e)*kvalues[q]
struct block_mxfp4 { uint8_t e; uint8_t qs[16]; };
#pragma pack(pop)

static_assert(sizeof(block_q4_0)  == 18,  "Q4_0 block size (constants.py:2842)");
static_assert(sizeof(block_q8_0)  == 34,  "Q8_0 block size (constants.py:2846)");
static_assert(sizeof(block_q4_K)  == 144, "Q4_K block size (constants.py:2850)");
static_assert(sizeof(block_q5_K)  == 176, "Q5_K block size (constants.py:2851)");
static_assert(sizeof(block_q6_K)  == 210, "Q6_K block size (constants.py:2852)");
static_assert(sizeof(block_mxfp4) == 17,  "MXFP4 block size (constants.py:2871)");
