// mmq.hip — quantized mat-mul for n > MMVQ_MAX_N activation columns (prefill, llama-bench pp512).
//
// Here MUL_MAT is a real dense contraction (arithmetic intensity ~ 2*n/0.56 FLOP per weight byte), so it goes to the
// matrix cores: each workgroup dequantizes a [BM x BK] tile of packed weights to bf16 in LDS ONCE and multiplies it with a
// [BN x BK] tile of bf16 activations by v_mfma_f32_32x32x16_bf16, accumulating in f32.
// Roofline: MFMA (bf16 dense peak ~2.5 PFLOP/s). Algorithmic FLOPs per launch = 2*m*n*k.
//
// Numerics: weights are dequantized exactly as the reference does (oracle/ggml_oracle.c dequantize_row_*) and rounded to
// bf16 (8-bit mantissa, relative error 2^-9 — far below the 4-6 bit quantization step); activations are rounded to bf16
// (the CPU path instead rounds them to int8 per block). Measured NMSE against the exact product is ~1e-5..1e-6
// (gate 5e-4, tests/test-backend-ops.cpp:3106-3108).
//
// Tiling: BM = BN = 128, BK = 64, 4 waves as 2 x 2, each wave 64 x 64 = 2 x 2 MFMA tiles (64 accumulator VGPRs).
// A operand = activations (rows = tokens), B operand = weights (cols = weight rows), so that D has the weight row on the
// lane and consecutive lanes store consecutive floats of dst. LDS rows are padded by 16 bytes (144-byte stride) to keep
// the ds_read_b128 operand reads conflict-free (cdna_hip_programming.md Guideline 4).
#include "blocks.h"
#include "dev_common.h"
#include "kernels.h"
#include "rope_dev.h"

#include <type_traits>

namespace mi355x {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float  f32x16 __attribute__((ext_vector_type(16)));

static __device__ __forceinline__ uint32_t bf16_rne(float f) {   // finite inputs only
    uint32_t u = __builtin_bit_cast(uint32_t, f);
    return (u + 0x7FFFu + ((u >> 16) & 1u)) >> 16;
}

// ---- f32 -> bf16 / f16 activation pre-pass: strided f32 rows -> dense [batch][n][kp] 16-bit, kp = k rounded up to the 64-wide k-step
//      and the tail filled with zeros (the mat-mul kernel then needs no k-tail handling for its activations) ----
static __host__ __device__ __forceinline__ int64_t mmq_kp(int64_t k) { return (k + 63) & ~(int64_t) 63; }
struct act16_args { const char * x; size_t nb1, nb2, nb3; int64_t k, n, ne2; uint16_t * y; };
template <bool F16>
__global__ void __launch_bounds__(256) k_act_to_16(const act16_args p) {
    const int64_t row = blockIdx.y, bz = blockIdx.z;
    const int64_t i2 = bz % p.ne2, i3 = bz / p.ne2;
    const int64_t i0 = ((int64_t) blockIdx.x*256 + threadIdx.x)*4;
    const int64_t kp = mmq_kp(p.k);
    if (i0 >= kp) return;
    const float4v v = i0 < p.k ? __builtin_bit_cast(float4v, ld_b128(p.x + row*p.nb1 + i2*p.nb2 + i3*p.nb3 + i0*4)) : float4v{ 0.0f, 0.0f, 0.0f, 0.0f };
    uint2 o;
    if (F16) { o.x = (uint32_t) f32_to_f16_bits(v.x) | ((uint32_t) f32_to_f16_bits(v.y) << 16); o.y = (uint32_t) f32_to_f16_bits(v.z) | ((uint32_t) f32_to_f16_bits(v.w) << 16); }
    else     { o.x = pack_bf16(v.x, v.y); o.y = pack_bf16(v.z, v.w); }
    *(uint2 *) (p.y + (bz*p.n + row)*kp + i0) = o;
}

// ---- 32 consecutive elements [c32*32, c32*32 + 32) of one weight row, in two steps so that the loads of the NEXT k-step can
//      be in flight while the matrix cores work on the current one: load_raw32 (global loads only), decode32 (reference dequant) ----
struct raw32 { int4v v[5]; uint32_t s; };

static __device__ __forceinline__ void k4_sc_m(const uint32_t (&hw)[4], int j, float & sc, float & m) {   // quants.py:479-501
    // the 12 scale bytes are hw[1..3]. Branch-free (a branch here would split the block the MFMAs and the decode are interleaved in):
    // with a, b, c = byte (j & 3) of hw[1], hw[2], hw[3]: j < 4: sc = a & 63, m = b & 63; else sc = (c & 15) | (a >> 6) << 4, m = (c >> 4) | (b >> 6) << 4
    const int sh = 8*(j & 3);
    const uint32_t a = (hw[1] >> sh) & 0xFF, b = (hw[2] >> sh) & 0xFF, c = (hw[3] >> sh) & 0xFF;
    const uint32_t hi = 0u - (uint32_t)((j >> 2) & 1);      // all ones for j >= 4; bit-select, not ?: (which the compiler turns back into a branch)
    const uint32_t s6 = ((a & 63) & ~hi) | (((c & 0xF) | ((a >> 6) << 4)) & hi);
    const uint32_t m6 = ((b & 63) & ~hi) | (((c >> 4)  | ((b >> 6) << 4)) & hi);
    sc = (float) s6; m = (float) m6;
}

// dq_head = what is common to the 32 elements (scales), decode4 = elements [4*wi, 4*wi + 4) of the chunk, wi = 0..7 a compile-time
// constant: the kernel spreads the eight pieces of a chunk between its MFMAs
struct dq_head { float a, b; };
template <int TYPE> static __device__ __forceinline__ raw32 load_raw32(const char * row, int c32);
template <int TYPE> static __device__ __forceinline__ dq_head decode_head(const raw32 & r, int c32);
// ws = the word slot piece wi's words sit in (a constant: it indexes registers); wi itself may be a run-time value (arithmetic only).
// k_mmq loads whole chunks: ws = wi; k_mmq16 loads a quarter chunk into slots 0, 1
template <int TYPE> static __device__ __forceinline__ void decode4(const raw32 & r, const dq_head & h, int c32, int ws, int wi, float (&o)[4]);

static __device__ __forceinline__ uint32_t raw_word(const raw32 & r, int i) {      // 32-bit word i of the 16-byte vectors, i a constant
    const int4v v = r.v[i >> 2];
    return (uint32_t) ((i & 3) == 0 ? v.x : (i & 3) == 1 ? v.y : (i & 3) == 2 ? v.z : v.w);
}

// Q4_0 — quants.py:241-251: element j < 16 = low nibble of byte j, element 16 + j = high nibble
template <> __device__ __forceinline__ raw32 load_raw32<T_Q4_0>(const char * row, int c32) {
    raw32 r; const char * b = row + (size_t) c32*18; r.s = ld_u16(b); r.v[0] = ld_b128(b + 2); return r;
}
template <> __device__ __forceinline__ dq_head decode_head<T_Q4_0>(const raw32 & r, int) { return { f16_bits_to_f32((uint16_t) r.s), 0.0f }; }
template <> __device__ __forceinline__ void decode4<T_Q4_0>(const raw32 & r, const dq_head & h, int, int ws, int wi, float (&o)[4]) {
    const uint32_t q4 = (raw_word(r, ws & 3) >> (4*(wi >> 2))) & 0x0F0F0F0Fu;
#pragma unroll
    for (int b = 0; b < 4; b++) o[b] = (float)((int)((q4 >> (8*b)) & 0xFF) - 8)*h.a;
}
// Q8_0 — quants.py:396-401
template <> __device__ __forceinline__ raw32 load_raw32<T_Q8_0>(const char * row, int c32) {
    raw32 r; const char * b = row + (size_t) c32*34; r.s = ld_u16(b); r.v[0] = ld_b128(b + 2); r.v[1] = ld_b128(b + 18); return r;
}
template <> __device__ __forceinline__ dq_head decode_head<T_Q8_0>(const raw32 & r, int) { return { f16_bits_to_f32((uint16_t) r.s), 0.0f }; }
template <> __device__ __forceinline__ void decode4<T_Q8_0>(const raw32 & r, const dq_head & h, int, int ws, int, float (&o)[4]) {
    const uint32_t w = raw_word(r, ws);
#pragma unroll
    for (int b = 0; b < 4; b++) o[b] = (float)(int8_t)((w >> (8*b)) & 0xFF)*h.a;
}
// MXFP4 — quants.py:656-700: nibble order as Q4_0; kvalues (quants.py:659): magnitude by the low 3 bits, sign by bit 3
template <> __device__ __forceinline__ raw32 load_raw32<T_MXFP4>(const char * row, int c32) {
    raw32 r; const char * b = row + (size_t) c32*17; r.s = *(const uint8_t *) b; r.v[0] = ld_b128(b + 1); return r;
}
template <> __device__ __forceinline__ dq_head decode_head<T_MXFP4>(const raw32 & r, int) { return { e8m0_to_f32_half(r.s), 0.0f }; }
template <> __device__ __forceinline__ void decode4<T_MXFP4>(const raw32 & r, const dq_head & h, int, int ws, int wi, float (&o)[4]) {
    const uint32_t q4 = (raw_word(r, ws & 3) >> (4*(wi >> 2))) & 0x0F0F0F0Fu;
    const uint64_t mag = 0x0C08060403020100ull;
#pragma unroll
    for (int b = 0; b < 4; b++) {
        const uint32_t q = (q4 >> (8*b)) & 0xFF;
        const float v = (float)((mag >> (8*(q & 7))) & 0xFF)*h.a;
        o[b] = __builtin_bit_cast(float, __builtin_bit_cast(uint32_t, v) ^ ((q & 8) << 28));      // -v for the upper half of the table
    }
}
// MXFP4 on gfx950's FP4 conversion (round 3): the 8 consecutive weights [8 g, 8 g + 8) of a 32-block as bf16, ready for the LDS tile. An MXFP4 element IS an E2M1 float
// times 2^(e - 127) (the reference's integer table is 2 x E2M1, its scale 2^(e - 127) / 2): v_cvt_scalef32_pk_bf16_fp4 turns a byte's two nibbles into two bf16 with the scale
// applied — exact, as the float path is — and one v_perm_b32 per pair puts neighbours side by side (a byte holds elements i and i + 16): 12 instructions per 8 weights
// where the table lookup + int -> float -> multiply -> sign -> pack took ~60.
typedef __bf16 mq_bf2 __attribute__((ext_vector_type(2)));
static __device__ __forceinline__ int4v decode8_bf16_mxfp4(const raw32 & r, const dq_head & h, int g) {
    const float scale = 2.0f*h.a;
    const uint32_t sel = (g & 2) ? 0x07060302u : 0x05040100u;      // the high-nibble elements (16..31) or the low-nibble ones of two converted bytes
    uint32_t out[4];
#pragma unroll
    for (int d = 0; d < 2; d++) {
        const uint32_t w = raw_word(r, (2*g + d) & 3);
        const uint32_t c0 = __builtin_bit_cast(uint32_t, __builtin_amdgcn_cvt_scalef32_pk_bf16_fp4(w, scale, 0)), c1 = __builtin_bit_cast(uint32_t, __builtin_amdgcn_cvt_scalef32_pk_bf16_fp4(w, scale, 1));
        const uint32_t c2 = __builtin_bit_cast(uint32_t, __builtin_amdgcn_cvt_scalef32_pk_bf16_fp4(w, scale, 2)), c3 = __builtin_bit_cast(uint32_t, __builtin_amdgcn_cvt_scalef32_pk_bf16_fp4(w, scale, 3));
        out[2*d]     = __builtin_amdgcn_perm(c1, c0, sel);
        out[2*d + 1] = __builtin_amdgcn_perm(c3, c2, sel);
    }
    return int4v{ (int) out[0], (int) out[1], (int) out[2], (int) out[3] };
}
// Q4_K — quants.py:504-522
template <> __device__ __forceinline__ raw32 load_raw32<T_Q4_K>(const char * row, int c32) {
    raw32 r; const int sb = c32 & 7; const char * b = row + (size_t)(c32 >> 3)*144;
    r.v[0] = *(const int4v *) b; r.v[1] = *(const int4v *) (b + 16 + 32*(sb >> 1)); r.v[2] = *(const int4v *) (b + 32 + 32*(sb >> 1)); r.s = 0; return r;
}
static __device__ __forceinline__ dq_head k45_head(const raw32 & r, int c32) {
    const uint32_t hw[4] = { (uint32_t) r.v[0].x, (uint32_t) r.v[0].y, (uint32_t) r.v[0].z, (uint32_t) r.v[0].w };
    const float d = f16_bits_to_f32((uint16_t)(hw[0] & 0xFFFF)), dmin = f16_bits_to_f32((uint16_t)(hw[0] >> 16));
    float sc, m; k4_sc_m(hw, c32 & 7, sc, m);
    return { d*sc, dmin*m };
}
template <> __device__ __forceinline__ dq_head decode_head<T_Q4_K>(const raw32 & r, int c32) { return k45_head(r, c32); }
template <> __device__ __forceinline__ void decode4<T_Q4_K>(const raw32 & r, const dq_head & h, int c32, int ws, int, float (&o)[4]) {
    // word-wise: one shift + mask per 4 nibbles, then v_cvt_f32_ubyte{0..3} straight from the masked word
    const uint32_t q4 = (raw_word(r, 4 + ws) >> ((c32 & 1)*4)) & 0x0F0F0F0Fu;
#pragma unroll
    for (int b = 0; b < 4; b++) o[b] = h.a*(float)((q4 >> (8*b)) & 0xFF) - h.b;
}
// Q5_K — quants.py:527-549
template <> __device__ __forceinline__ raw32 load_raw32<T_Q5_K>(const char * row, int c32) {
    raw32 r; const int sb = c32 & 7; const char * b = row + (size_t)(c32 >> 3)*176;
    r.v[0] = *(const int4v *) b; r.v[1] = *(const int4v *) (b + 16); r.v[2] = *(const int4v *) (b + 32);
    r.v[3] = *(const int4v *) (b + 48 + 32*(sb >> 1)); r.v[4] = *(const int4v *) (b + 64 + 32*(sb >> 1)); r.s = 0; return r;
}
template <> __device__ __forceinline__ dq_head decode_head<T_Q5_K>(const raw32 & r, int c32) { return k45_head(r, c32); }
template <> __device__ __forceinline__ void decode4<T_Q5_K>(const raw32 & r, const dq_head & h, int c32, int ws, int, float (&o)[4]) {
    const int sb = c32 & 7;      // word-wise: 4 quants per word = low nibbles | (bit sb of the qh bytes) << 4
    const uint32_t q5 = ((raw_word(r, 12 + ws) >> ((sb & 1)*4)) & 0x0F0F0F0Fu) | (((raw_word(r, 4 + ws) >> sb) & 0x01010101u) << 4);
#pragma unroll
    for (int b = 0; b < 4; b++) o[b] = h.a*(float)((q5 >> (8*b)) & 0xFF) - h.b;
}
// Q6_K — quants.py:554-572. chunk c of a 256-superblock: half n = c>>2, quarter pq = c&3: elements 128n + 32pq + l
template <> __device__ __forceinline__ raw32 load_raw32<T_Q6_K>(const char * row, int c32) {
    raw32 r; const int c = c32 & 7, n = c >> 2, pq = c & 3; const char * b = row + (size_t)(c32 >> 3)*210;   // 2-byte aligned only
    const char * ql = b + 64*n + 32*(pq & 1);
    r.v[0] = ld_b128(ql); r.v[1] = ld_b128(ql + 16); r.v[2] = ld_b128(b + 128 + 32*n); r.v[3] = ld_b128(b + 128 + 32*n + 16);
    r.s = (uint32_t) ld_u16(b + 208) | ((uint32_t) ld_u16(b + 192 + 8*n + 2*pq) << 16);
    return r;
}
template <> __device__ __forceinline__ dq_head decode_head<T_Q6_K>(const raw32 & r, int) {
    const float d = f16_bits_to_f32((uint16_t)(r.s & 0xFFFF));
    return { d*(float)(int8_t)((r.s >> 16) & 0xFF), d*(float)(int8_t)(r.s >> 24) };
}
template <> __device__ __forceinline__ void decode4<T_Q6_K>(const raw32 & r, const dq_head & h, int c32, int ws, int wi, float (&o)[4]) {
    const int pq = c32 & 3;      // word-wise: 4 quants per word = low nibbles | (2 bits of the qh bytes) << 4, each 0..63
    const uint32_t q6 = ((raw_word(r, ws) >> ((pq >> 1)*4)) & 0x0F0F0F0Fu) | (((raw_word(r, 8 + ws) >> (2*pq)) & 0x03030303u) << 4);
    const float sc = wi < 4 ? h.a : h.b;
#pragma unroll
    for (int b = 0; b < 4; b++) o[b] = sc*(float)((int)((q6 >> (8*b)) & 0xFF) - 32);
}

// ---- the tiled kernel ----
constexpr int MQ_BM = 128, MQ_BN = 128, MQ_BK = 64, MQ_LD = MQ_BK*2 + 16;   // LDS row stride in bytes (padded)

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

// several weight tensors against the same activations in one launch (wq / wk / wv): the m-tiles of the segments are laid end to end
struct mmq_seg { const char * W; char * dst; size_t w_row_stride, dst_nb1; int m, tile0, col0, type2, rope; const float * bias; };   // bias: NULL, or m floats added to every token's row (no k split)   // col0: first column in a split-k plane; type2: decode
                                               // as TYPE2; rope: the NORM rotary embedding (mmq_args::rope) is applied to this segment's rows of heads
struct mmq_args {
    const char * W; size_t w_row_stride, w_nb2, w_nb3; int m, k;
    const uint16_t * X; int n;                 // dense [batch][n][k] 16-bit
    char * dst; size_t dst_nb1, dst_nb2, dst_nb3;
    int ne12, r2, r3;                          // batch = blockIdx.z = i13*ne12 + i12; weights broadcast: i02 = i12/r2, i03 = i13/r3
    int ksplit, mtiles;                        // ksplit = 2 | 4: blockIdx.y = part*mtiles + m-tile; part i stores its partial product into plane i
    char * dst2;                               //   of dst2 (dense [n][m] f32 each); k_combine adds the planes in a fixed order: deterministic, nothing to clear
    const char * res; size_t res_nb1;          // ksplit == 1: f32 rows added to the product in the epilogue (the residual of build_attn / build_ffn), or NULL
    uint16_t * y16;                            // DUAL kernel: != NULL: the SwiGLU result as bf16 rows of m elements (dst may then be NULL)
    const char * W2;                           // DUAL kernel: the second weight tensor (dst = silu(W.x) * (W2.x), build_ffn's gate / up + swiglu)
    // MUL_MAT_ID (grouped by expert): blockIdx.x walks the tile table k_moe_sort wrote; a tile = up to 128 (token, slot) pairs of ONE expert
    const int * moe;                           // NULL, or [0] = n_tiles, then {expert, first, count}[max_tiles], then sorted pair ids
    int moe_max_tiles, n_used, n_b;            // pair = token*n_used + slot; X row of a pair = token*n_b + slot % n_b
    fused_rope rope;                           // for segments with .rope (wq, wk): RESHAPE -> ROPE of build_attn folded into the epilogue / the combine pass
    int nseg; mmq_seg seg[3];                  // nseg > 0: W / m / dst / strides per segment; p.m = the summed rows (the width of a split-k plane)
    // MUL_MAT_ID epilogues (MOE kernels): per-expert bias rows (ADD_ID, src/llama-graph.cpp:927,940,985) for W / W2, the SwiGLU flavour of the dual kernel,
    // and (single-tensor kernel) the routing weight of the pair — MUL(experts, weights), :990 — applied after the bias
    const float * bias; const float * bias2; size_t bias_stride;   // floats between two experts' bias rows
    const float * scale; size_t scale_nb0, scale_nb1;       // weight of (token, slot) at scale + slot*nb0 + token*nb1 (bytes)
    int glu_oai; float glu_alpha, glu_limit;
    const char * glu_up; size_t glu_up_nb1, glu_up_nb2;     // single-tensor MOE kernel: != NULL: this launch is the GATE product and the up product (+ bias) is already in memory
                                                            // at glu_up + slot*nb1 + token*nb2: the epilogue evaluates the GLU and writes y16 / dst like the dual kernel
    int dbg;                                   // -DMI_MMQ_DBG builds only (a runtime branch in this loop costs 10 % of pp512): GGML_MI355X_MMQ_DBG ablations for timing, wrong results: 1 = no weight decode, 2 = no MFMAs, 4 = no global loads inside the loop, 8 = no LDS commits
};

// TYPE = a block format (bf16 MFMA on dequantized weights) or T_F16 (f16 MFMA, weights copied as they are: the attention
// products K.q and V.kq of build_attn_mha, src/llama-graph.cpp:1285,1320, when more than 8 tokens are in flight)
// BN = tokens per workgroup tile: 128 (4 waves) or 256 (8 waves; the weight tile is dequantized by waves 0-3 only and reused by twice
// as many MFMAs — with 128 tokens the dequantization VALU work, not the matrix cores, sets the pace: an ablation without any global
// load still ran at 22 % of the bf16 peak)
// DUAL (BN = 256 only): two weight tensors against the same activations and their SwiGLU in the epilogue — waves 0-3 dequantize the
// gate tile, waves 4-7 the up tile (every wave has dequantization work now), every wave multiplies its 64 x 64 token / row tile with
// both: 32 MFMAs per k-step and wave against one 64-element dequantization, one result tensor instead of two plus a GLU kernel
// MOE: MUL_MAT_ID tile-table mode (a tile = up to BN (token, slot) pairs of one expert). With 256-pair tiles a wave whose 64 pairs are past the tile's count skips
// its matrix-core work altogether and only stages (experts of a prompt pass hold ~64 - 128 pairs each: the kernel is bound by the weight decode).
// What counts is how many waves per CU DECODE: the 128-pair kernel runs two workgroups of four decoding waves per CU, the single-tensor 256-pair kernel one workgroup
// in which only waves 0-3 decode — at ~128 pairs per expert it decodes each expert once instead of 1.5 times and is still 30 % slower. The dual 256-pair kernel has
// eight decoding waves (four per tensor) and one activation tile for both tensors: that one pays.
template <int TYPE, int BN = MQ_BN, bool DUAL = false, int TYPE2 = TYPE, bool MOE = false>
__global__ void __launch_bounds__(BN*2) __attribute__((amdgpu_waves_per_eu(2, 2))) k_mmq(const mmq_args p) {   // LDS allows 8 waves per CU anyway; without the
    // occupancy pin the scheduler reverts the interleaved order below to keep a third wave's worth of registers free
    extern __shared__ __attribute__((aligned(16))) char lds[];      // 2 x (W tile [| W2 tile] | X tile)
    constexpr int WTILE = MQ_BM*MQ_LD, XTILE = BN*MQ_LD, WT = DUAL ? 2 : 1, STAGE = WT*WTILE + XTILE;     // bytes
    static_assert(!DUAL || BN == 256, "the dual kernel has 8 waves");
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int khalf = p.ksplit > 1 ? (int) blockIdx.y / p.mtiles : 0;      // which part of k
    int mt = (int) blockIdx.y - khalf*p.mtiles;
    const int n0 = blockIdx.x*BN;                        // the n-tiles of one weight tile are dispatched together
    const int wm = wave & 1, wn = wave >> 1;             // wave tile: weight rows wm*64.., tokens wn*64..
    const int n = p.n, k = p.k;
    int m = p.m, col0 = 0; bool use2 = false;
    const int i12 = blockIdx.z % p.ne12, i13 = blockIdx.z / p.ne12;
    const char * W = p.W + (size_t)(i12/p.r2)*p.w_nb2 + (size_t)(i13/p.r3)*p.w_nb3;
    size_t w_row_stride = p.w_row_stride, seg_dst_nb1 = p.dst_nb1; char * seg_dst = p.dst; const float * seg_bias = nullptr;
    if (p.nseg) {                              // workgroup-uniform
        const int si = (mt >= p.seg[1].tile0 ? 1 : 0) + (p.nseg > 2 && mt >= p.seg[2].tile0 ? 1 : 0);
        W = p.seg[si].W; m = p.seg[si].m; w_row_stride = p.seg[si].w_row_stride; seg_dst = p.seg[si].dst; seg_dst_nb1 = p.seg[si].dst_nb1;
        col0 = p.seg[si].col0; use2 = p.seg[si].type2 != 0; mt -= p.seg[si].tile0; seg_bias = p.seg[si].bias;
    }
    const int m0 = mt*MQ_BM;
    const int kp = (k + MQ_BK - 1) & ~(MQ_BK - 1);               // row length of the activation copy (zero-padded)
    const uint16_t * X = p.X + (size_t) blockIdx.z*n*kp;
    char * dst = p.ksplit > 1 ? p.dst2 + (size_t) khalf*p.m*p.n*4 + (size_t) col0*4 : seg_dst + (size_t) i12*p.dst_nb2 + (size_t) i13*p.dst_nb3;
    const size_t dst_nb1 = p.ksplit > 1 ? (size_t) p.m*4 : seg_dst_nb1;
    int moe_first = 0, moe_cnt = 0, moe_e = 0;
    const int * moe_pairs = nullptr;
    if (MOE) {                                 // workgroup-uniform
        if ((int) blockIdx.x >= p.moe[0]) return;
        const int * t = p.moe + 1 + 3*blockIdx.x;
        moe_e = t[0]; W = p.W + (size_t) moe_e*p.w_nb2; moe_first = t[1]; moe_cnt = t[2];
        moe_pairs = p.moe + 1 + 3*p.moe_max_tiles;
    }

    f32x16 acc[2][2], acc2[DUAL ? 2 : 1][DUAL ? 2 : 1];
#pragma unroll
    for (int i = 0; i < 2; i++)
#pragma unroll
        for (int j = 0; j < 2; j++)
#pragma unroll
            for (int r = 0; r < 16; r++) { acc[i][j][r] = 0.0f; if (DUAL) acc2[DUAL ? i : 0][DUAL ? j : 0][r] = 0.0f; }

    // staging roles: thread -> (row = tid/2, half = tid&1): 32 of the 64 k of that row — every thread an activation row, the first 256
    // threads (waves 0-3) also a weight row
    const int srow = tid >> 1, shalf = tid & 1;
    const bool w_role = DUAL || BN == MQ_BM || tid < 2*MQ_BM;        // wave-uniform
    const int wt = DUAL ? tid >> 8 : 0;                              // DUAL: which weight tile this thread dequantizes
    const char * wrow_p = (DUAL && wt ? (MOE ? p.W2 + (size_t) moe_e*p.w_nb2 : p.W2) : W) + (size_t) min(m0 + (srow & (MQ_BM - 1)), m - 1)*w_row_stride;
    const uint16_t * xrow_p = X + (size_t) min(n0 + srow, n - 1)*kp;
    if (MOE) {
        const int pair = moe_pairs[moe_first + min(srow, moe_cnt - 1)];
        xrow_p = p.X + (size_t)((pair/p.n_used)*p.n_b + (pair % p.n_used) % p.n_b)*kp;
    }
    // split-K: each half walks k/2 (a multiple of 256, so block boundaries stay aligned); steps are counted from step0
    const int nsteps_all = (k + MQ_BK - 1)/MQ_BK;
    const int nsteps = p.ksplit > 1 ? nsteps_all/p.ksplit : nsteps_all;
    const int step0 = khalf*nsteps;

    // Software pipeline, two register stages: iteration s issues the global loads of step s+2, runs the MFMAs of step s from one LDS
    // buffer and — interleaved with them instruction by instruction (sched_group_barrier) — decodes step s+1 (loaded one iteration
    // ago, so nothing waits on memory) into the other buffer. Lock-step phases (all waves decode, then all waves multiply) left the
    // matrix cores idle during the decode and the VALU idle during the MFMAs: ~5400 clocks per k-step against 1024 of MFMA work.
    // Every iteration is the same straight-line block: steps past the end are fetched from clamped addresses and committed to a
    // buffer nobody reads; in a k tail (k % 64 == 32) the activation copy holds zeros, so the clamped (finite) weights drop out.
    struct stage_regs { raw32 rw; int4v xv[4]; };
    auto run = [&](auto w_role_tag, auto type_tag, auto mf_tag) {
        constexpr bool WR = decltype(w_role_tag)::value;
        constexpr bool MF = decltype(mf_tag)::value;       // false (MOE only): this wave's pairs are all past the tile's count — stage, do not multiply
        constexpr int TY = decltype(type_tag)::value;      // the block format this workgroup's segment is decoded as
        auto fetch = [&](stage_regs & r, int step) {
#ifdef MI_MMQ_DBG
            if ((p.dbg & 4) && step > step0 + 1) return;
#endif
            const int kc = step*MQ_BK + 32*shalf, kcl = min(kc, k - 32);
            if (WR) {
                if (TY == T_F16) {
#pragma unroll
                    for (int i = 0; i < 4; i++) r.rw.v[i] = ld_b128(wrow_p + (size_t) kcl*2 + 16*i);
                } else {
                    r.rw = load_raw32<TY == T_F16 ? T_Q8_0 : TY>(wrow_p, kcl >> 5);
                }
            }
#pragma unroll
            for (int i = 0; i < 4; i++) r.xv[i] = ld_b128((const char *) (xrow_p + min(kc, kp - 32)) + 16*i);
        };
        // one eighth of a thread's staging work for step `step`: 16 bytes of activations into LDS, 8 weights decoded, packed and stored
        auto commit_piece = [&](const stage_regs & r, const dq_head & h, int step, int buf, int g) {
#ifdef MI_MMQ_DBG
            if (p.dbg & 8) return;
#endif
            const int kcl = min(step*MQ_BK + 32*shalf, k - 32);
            char * wp = lds + buf*STAGE + wt*WTILE + (srow & (MQ_BM - 1))*MQ_LD + shalf*64;
            char * xp = lds + buf*STAGE + WT*WTILE + srow*MQ_LD + shalf*64;
            *(int4v *) (xp + 16*g) = r.xv[g];
            if (WR) {
                int4v wpk;
#ifdef MI_MMQ_DBG
                if (TY == T_F16 || (p.dbg & 1)) {
#else
                if (TY == T_F16) {
#endif
                    wpk = r.rw.v[g & 3];
                } else if (TY == T_MXFP4) {
                    wpk = decode8_bf16_mxfp4(r.rw, h, g);
                } else {
                    float lo[4], hi[4];
                    decode4<TY == T_F16 ? T_Q8_0 : TY>(r.rw, h, kcl >> 5, 2*g, 2*g, lo);
                    decode4<TY == T_F16 ? T_Q8_0 : TY>(r.rw, h, kcl >> 5, 2*g + 1, 2*g + 1, hi);
                    wpk.x = (int) pack_bf16(lo[0], lo[1]); wpk.y = (int) pack_bf16(lo[2], lo[3]);
                    wpk.z = (int) pack_bf16(hi[0], hi[1]); wpk.w = (int) pack_bf16(hi[2], hi[3]);
                }
                *(int4v *) (wp + 16*g) = wpk;
            }
        };
        auto head_of = [&](const stage_regs & r, int step) -> dq_head {
            if (!WR || TY == T_F16) return { 0.0f, 0.0f };
            return decode_head<TY == T_F16 ? T_Q8_0 : TY>(r.rw, min(step*MQ_BK + 32*shalf, k - 32) >> 5);
        };
        struct frags { int4v a[2], b[2], b2[DUAL ? 2 : 1]; };
        // A = activations (rows = tokens), B = weights (cols = weight rows)
        auto read_frags = [&](frags & f, int buf, int kk) {
            const char * lw = lds + buf*STAGE, * lx = lw + WT*WTILE;
#pragma unroll
            for (int i = 0; i < 2; i++) {
                f.a[i] = *(const int4v *) (lx + (wn*64 + i*32 + (lane & 31))*MQ_LD + kk*32 + (lane >> 5)*16);
                f.b[i] = *(const int4v *) (lw + (wm*64 + i*32 + (lane & 31))*MQ_LD + kk*32 + (lane >> 5)*16);
                if (DUAL) f.b2[DUAL ? i : 0] = *(const int4v *) (lw + WTILE + (wm*64 + i*32 + (lane & 31))*MQ_LD + kk*32 + (lane >> 5)*16);
            }
        };
        auto mfma_row = [&](const frags & f, int i) {       // the MFMAs of token sub-tile i: 2 (4 with the second weight tile)
#ifdef MI_MMQ_DBG
            if (p.dbg & 2) return;
#endif
#pragma unroll
            for (int j = 0; j < 2; j++) {
                if (TY == T_F16) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, f.a[i]), __builtin_bit_cast(f16x8, f.b[j]), acc[i][j], 0, 0, 0);
                else               acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, f.a[i]), __builtin_bit_cast(bf16x8, f.b[j]), acc[i][j], 0, 0, 0);
                if (DUAL) acc2[DUAL ? i : 0][DUAL ? j : 0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, f.a[i]), __builtin_bit_cast(bf16x8, f.b2[DUAL ? j : 0]), acc2[DUAL ? i : 0][DUAL ? j : 0], 0, 0, 0);
            }
        };
        // The order is pinned by hand (sched_barrier(0): nothing crosses): the scheduler left to itself — and sched_group_barrier
        // pipelines were not honoured here — puts all MFMAs first and the whole decode after them.
        auto iteration = [&](stage_regs & cur, stage_regs & nxt, int s, int buf) {
            fetch(nxt, step0 + s + 2);
            const dq_head h = head_of(cur, step0 + s + 1);
            if constexpr (!MF) {
#pragma unroll
                for (int kk = 0; kk < MQ_BK/16; kk++) commit_piece(cur, h, step0 + s + 1, buf ^ 1, kk);
            } else {
            frags f0, f1;       // operand fragments, read one k-slice ahead (the dual kernel has no registers left for that)
            if (!DUAL) read_frags(f0, buf, 0);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int kk = 0; kk < MQ_BK/16; kk++) {
                frags & f = (!DUAL && (kk & 1)) ? f1 : f0; frags & fn = (kk & 1) ? f0 : f1;
                if (DUAL) read_frags(f, buf, kk);
                else if (kk + 1 < MQ_BK/16) read_frags(fn, buf, kk + 1);
                mfma_row(f, 0);
                // (dual kernel: running the decode piece BEFORE the slice's first MFMAs, under the operand reads' latency, measured 3 % slower)
                if (!DUAL) { __builtin_amdgcn_sched_barrier(0); mfma_row(f, 1); commit_piece(cur, h, step0 + s + 1, buf ^ 1, kk); }
                else       { commit_piece(cur, h, step0 + s + 1, buf ^ 1, kk); __builtin_amdgcn_sched_barrier(0); mfma_row(f, 1); }
                __builtin_amdgcn_sched_barrier(0);
            }
            }
            __syncthreads();
            __builtin_amdgcn_sched_barrier(0);
        };
        stage_regs p0, p1;
        fetch(p0, step0);
        fetch(p1, step0 + 1);
        {
            const dq_head h = head_of(p0, step0);
#pragma unroll
            for (int g = 0; g < 4; g++) commit_piece(p0, h, step0, 0, g);
        }
        __syncthreads();
        __builtin_amdgcn_sched_barrier(0);
        int s = 0;
        for (; s + 1 < nsteps; s += 2) {
            iteration(p1, p0, s, 0);
            iteration(p0, p1, s + 1, 1);
        }
        if (s < nsteps) iteration(p1, p0, s, 0);
    };
    if constexpr (TYPE2 != TYPE) {
        if (use2) { if (w_role) run(std::true_type{}, std::integral_constant<int, TYPE2>{}, std::true_type{}); else run(std::false_type{}, std::integral_constant<int, TYPE2>{}, std::true_type{}); }
        else      { if (w_role) run(std::true_type{}, std::integral_constant<int, TYPE>{}, std::true_type{});  else run(std::false_type{}, std::integral_constant<int, TYPE>{}, std::true_type{}); }
    } else if constexpr (MOE && BN == 256) {
        const bool live = wn*64 < moe_cnt;             // wave-uniform
        if (live) { if (w_role) run(std::true_type{}, std::integral_constant<int, TYPE>{}, std::true_type{});  else run(std::false_type{}, std::integral_constant<int, TYPE>{}, std::true_type{}); }
        else      { if (w_role) run(std::true_type{}, std::integral_constant<int, TYPE>{}, std::false_type{}); else run(std::false_type{}, std::integral_constant<int, TYPE>{}, std::false_type{}); }
    } else {
        if (w_role) run(std::true_type{}, std::integral_constant<int, TYPE>{}, std::true_type{}); else run(std::false_type{}, std::integral_constant<int, TYPE>{}, std::true_type{});
    }
    // ---- store: D[row = token][col = weight row]; col = lane & 31, row = (r & 3) + 8*(r >> 2) + 4*(lane >> 5) ----
#pragma unroll
    for (int i = 0; i < 2; i++) {
#pragma unroll
        for (int j = 0; j < 2; j++) {
            const int col = m0 + wm*64 + j*32 + (lane & 31);
#pragma unroll
            for (int r = 0; r < 16; r++) {
                const int row = n0 + wn*64 + i*32 + (r & 3) + 8*(r >> 2) + 4*(lane >> 5);
                if (MOE) {
                    const int rt = MOE ? wn*64 + i*32 + (r & 3) + 8*(r >> 2) + 4*(lane >> 5) : 0;
                    if (col < m && rt < moe_cnt) {
                        const int pair = moe_pairs[moe_first + rt];
                        const int tok = pair/p.n_used, slot = pair - tok*p.n_used;
                        float v = acc[i][j][r];
                        if (p.bias) v += p.bias[(size_t) moe_e*p.bias_stride + col];
                        if (DUAL || p.glu_up) {
                            float u;
                            if (DUAL) { u = acc2[DUAL ? i : 0][DUAL ? j : 0][r]; if (p.bias2) u += p.bias2[(size_t) moe_e*p.bias_stride + col]; }
                            else u = *(const float *) (p.glu_up + (size_t) tok*p.glu_up_nb2 + (size_t) slot*p.glu_up_nb1 + (size_t) col*4);
                            float y;
                            if (p.glu_oai) {          // elem.hip k_glu, GGML_GLU_OP_SWIGLU_OAI
                                const float xc = fminf(v, p.glu_limit), gc = fmaxf(fminf(u, p.glu_limit), -p.glu_limit);
                                y = (xc/(1.0f + expf(-xc*p.glu_alpha)))*(gc + 1.0f);
                            } else y = (v/(1.0f + expf(-v)))*u;
                            if (p.dst) *(float *) (p.dst + (size_t) tok*p.dst_nb2 + (size_t) slot*p.dst_nb1 + (size_t) col*4) = y;
                            if (p.y16) { const uint32_t pk = pack_bf16(y, y); p.y16[(size_t) pair*m + col] = (uint16_t) pk; }
                        } else {
                            if (p.scale) v *= *(const float *) ((const char *) p.scale + (size_t) slot*p.scale_nb0 + (size_t) tok*p.scale_nb1);
                            *(float *) (p.dst + (size_t) tok*p.dst_nb2 + (size_t) slot*p.dst_nb1 + (size_t) col*4) = v;
                        }
                    }
                } else if (col < m && row < n) {
                    float * o = (float *) (dst + (size_t) row*dst_nb1 + (size_t) col*4);
                    if (DUAL) {         // silu(gate)*up, as elem.hip k_glu
                        const float g = acc[i][j][r], y = (g/(1.0f + expf(-g)))*acc2[DUAL ? i : 0][DUAL ? j : 0][r];
                        if (p.dst) *o = y;
                        if (p.y16) { const uint32_t pk = pack_bf16(y, y); p.y16[(size_t) row*m + col] = (uint16_t) pk; }
                    }
                    else if (p.res && p.ksplit == 1) *o = acc[i][j][r] + *(const float *) (p.res + (size_t) row*p.res_nb1 + (size_t) col*4);
                    else if (seg_bias && p.ksplit == 1) *o = acc[i][j][r] + seg_bias[col];
                    else *o = acc[i][j][r];
                }
            }
        }
    }
}

// ---- the same tile (128 weight rows x 256 tokens x 64 k) on 16 waves -------------------------------------------------------------
// k_mmq's 8 waves run at ~170 clocks per MFMA and wave whatever the tile shape — a wave's own chain of MFMA, decode VALU, LDS store
// and waits — so 2 waves per SIMD keep the matrix cores 36 % busy. Here a wave owns 64 tokens x 32 rows (32 accumulator registers,
// 8 MFMAs per k-step), decodes 8 weights and stages 16 activations per k-step: 4 waves per SIMD at <= 128 VGPRs.
// Quarter-chunk loads: thread -> (row, 32-chunk, quarter q): pieces 2q, 2q + 1; their words land in word slots 0, 1 of each vector.
template <int TYPE> static __device__ __forceinline__ raw32 load_raw8(const char * row, int c32, int q);
template <> __device__ __forceinline__ raw32 load_raw8<T_Q4_0>(const char * row, int c32, int q) {
    raw32 r; const char * b = row + (size_t) c32*18; r.s = ld_u16(b); const int2v t = ld_b64(b + 2 + 8*(q & 1)); r.v[0].x = t.x; r.v[0].y = t.y; return r;
}
template <> __device__ __forceinline__ raw32 load_raw8<T_MXFP4>(const char * row, int c32, int q) {
    raw32 r; const char * b = row + (size_t) c32*17; r.s = *(const uint8_t *) b; const int2v t = ld_b64(b + 1 + 8*(q & 1)); r.v[0].x = t.x; r.v[0].y = t.y; return r;
}
template <> __device__ __forceinline__ raw32 load_raw8<T_Q8_0>(const char * row, int c32, int q) {
    raw32 r; const char * b = row + (size_t) c32*34; r.s = ld_u16(b); const int2v t = ld_b64(b + 2 + 8*q); r.v[0].x = t.x; r.v[0].y = t.y; return r;
}
template <> __device__ __forceinline__ raw32 load_raw8<T_Q4_K>(const char * row, int c32, int q) {
    raw32 r; const int sb = c32 & 7; const char * b = row + (size_t)(c32 >> 3)*144;
    r.v[0] = *(const int4v *) b; r.s = 0;
    const int2v t = *(const int2v *) (b + 16 + 32*(sb >> 1) + 8*q); r.v[1].x = t.x; r.v[1].y = t.y; return r;
}
template <> __device__ __forceinline__ raw32 load_raw8<T_Q5_K>(const char * row, int c32, int q) {
    raw32 r; const int sb = c32 & 7; const char * b = row + (size_t)(c32 >> 3)*176;
    r.v[0] = *(const int4v *) b; r.s = 0;
    const int2v h = *(const int2v *) (b + 16 + 8*q), t = *(const int2v *) (b + 48 + 32*(sb >> 1) + 8*q);
    r.v[1].x = h.x; r.v[1].y = h.y; r.v[3].x = t.x; r.v[3].y = t.y; return r;
}
template <> __device__ __forceinline__ raw32 load_raw8<T_Q6_K>(const char * row, int c32, int q) {
    raw32 r; const int c = c32 & 7, n = c >> 2, pq = c & 3; const char * b = row + (size_t)(c32 >> 3)*210;   // 2-byte aligned only
    const int2v l = ld_b64(b + 64*n + 32*(pq & 1) + 8*q), h = ld_b64(b + 128 + 32*n + 8*q);
    r.v[0].x = l.x; r.v[0].y = l.y; r.v[2].x = h.x; r.v[2].y = h.y;
    r.s = (uint32_t) ld_u16(b + 208) | ((uint32_t) ld_u16(b + 192 + 8*n + 2*pq) << 16);
    return r;
}

// MOE: the tile-table mode of k_mmq (a tile = up to 256 (token, slot) pairs of one expert) with all 16 waves decoding; a wave whose 64 pairs are past the tile's
// count stages and decodes but skips its matrix-core work
template <int TYPE, int TYPE2 = TYPE, bool MOE = false>
__global__ void __launch_bounds__(1024) k_mmq16(const mmq_args p) {
    extern __shared__ __attribute__((aligned(16))) char lds[];      // 2 x (W tile | X tile)
    constexpr int BN = 256, WTILE = MQ_BM*MQ_LD, XTILE = BN*MQ_LD, STAGE = WTILE + XTILE;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int khalf = p.ksplit > 1 ? (int) blockIdx.y / p.mtiles : 0;
    int mt = (int) blockIdx.y - khalf*p.mtiles;
    const int n0 = blockIdx.x*BN;
    const int wm = wave & 3, wn = wave >> 2;             // wave tile: weight rows wm*32.., tokens wn*64..
    const int n = p.n, k = p.k;
    int m = p.m, col0 = 0; bool use2 = false, do_rope = false;
    const char * W = p.W;
    size_t w_row_stride = p.w_row_stride, seg_dst_nb1 = p.dst_nb1; char * seg_dst = p.dst; const float * seg_bias = nullptr;
    if (p.nseg) {                              // workgroup-uniform
        const int si = (mt >= p.seg[1].tile0 ? 1 : 0) + (p.nseg > 2 && mt >= p.seg[2].tile0 ? 1 : 0);
        W = p.seg[si].W; m = p.seg[si].m; w_row_stride = p.seg[si].w_row_stride; seg_dst = p.seg[si].dst; seg_dst_nb1 = p.seg[si].dst_nb1;
        col0 = p.seg[si].col0; use2 = p.seg[si].type2 != 0; mt -= p.seg[si].tile0; seg_bias = p.seg[si].bias;
        do_rope = p.seg[si].rope != 0 && p.ksplit == 1;           // with a k split the combine pass rotates
    }
    const int m0 = mt*MQ_BM;
    const int kp = (k + MQ_BK - 1) & ~(MQ_BK - 1);
    char * dst = p.ksplit > 1 ? p.dst2 + (size_t) khalf*p.m*p.n*4 + (size_t) col0*4 : seg_dst;
    const size_t dst_nb1 = p.ksplit > 1 ? (size_t) p.m*4 : seg_dst_nb1;
    int moe_first = 0, moe_cnt = 0, moe_e = 0;
    const int * moe_pairs = nullptr;
    if (MOE) {                                 // workgroup-uniform
        if ((int) blockIdx.x >= p.moe[0]) return;
        const int * t = p.moe + 1 + 3*blockIdx.x;
        moe_e = t[0]; W = p.W + (size_t) moe_e*p.w_nb2; moe_first = t[1]; moe_cnt = t[2];
        moe_pairs = p.moe + 1 + 3*p.moe_max_tiles;
    }

    f32x16 acc[2];
#pragma unroll
    for (int i = 0; i < 2; i++)
#pragma unroll
        for (int r = 0; r < 16; r++) acc[i][r] = 0.0f;

    // staging roles: weights: row = tid/8, 32-chunk = (tid/4)&1, quarter = tid&3 (8 weights); activations: row = tid/4, 16 of the 64 k
    const int wrow = tid >> 3, wchunk = (tid >> 2) & 1, wq = tid & 3;
    const int xrow = tid >> 2, xq = tid & 3;
    const char * wrow_p = W + (size_t) min(m0 + wrow, m - 1)*w_row_stride;
    const uint16_t * xrow_p = p.X + (size_t) min(n0 + xrow, n - 1)*kp + 16*xq;
    if (MOE) {
        const int pair = moe_pairs[moe_first + min(xrow, moe_cnt - 1)];
        xrow_p = p.X + (size_t)((pair/p.n_used)*p.n_b + (pair % p.n_used) % p.n_b)*kp + 16*xq;
    }
    const int nsteps_all = (k + MQ_BK - 1)/MQ_BK;
    const int nsteps = p.ksplit > 1 ? nsteps_all/p.ksplit : nsteps_all;
    const int step0 = khalf*nsteps;

    struct stage_regs { raw32 rw; int4v xv[2]; };
    auto run = [&](auto type_tag, auto mf_tag) {
        constexpr int TY = decltype(type_tag)::value;
        constexpr bool MF = decltype(mf_tag)::value;
        auto fetch = [&](stage_regs & r, int step) {
            const int kcl = min(step*MQ_BK + 32*wchunk, k - 32);
            r.rw = load_raw8<TY>(wrow_p, kcl >> 5, wq);
            const char * xs = (const char *) (xrow_p + min(step*MQ_BK, kp - MQ_BK));
            r.xv[0] = ld_b128(xs); r.xv[1] = ld_b128(xs + 16);
        };
        auto wpos = [&](int buf) -> char * { return lds + buf*STAGE + wrow*MQ_LD + wchunk*64 + wq*16; };
        auto xpos = [&](int buf) -> char * { return lds + buf*STAGE + WTILE + xrow*MQ_LD + xq*32; };
        auto c32_of = [&](int step) -> int { return min(step*MQ_BK + 32*wchunk, k - 32) >> 5; };
        struct frags { int4v a[2], b; };
        auto read_frags = [&](frags & f, int buf, int kk) {
            const char * lw = lds + buf*STAGE, * lx = lw + WTILE;
#pragma unroll
            for (int i = 0; i < 2; i++) f.a[i] = *(const int4v *) (lx + (wn*64 + i*32 + (lane & 31))*MQ_LD + kk*32 + (lane >> 5)*16);
            f.b = *(const int4v *) (lw + (wm*32 + (lane & 31))*MQ_LD + kk*32 + (lane >> 5)*16);
        };
        auto mfma = [&](const frags & f, int i) {
            acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, f.a[i]), __builtin_bit_cast(bf16x8, f.b), acc[i], 0, 0, 0);
        };
        auto iteration = [&](stage_regs & cur, stage_regs & nxt, int s, int buf) {
            fetch(nxt, step0 + s + 2);
            const int c32 = c32_of(step0 + s + 1);
            const dq_head h = decode_head<TY>(cur.rw, c32);
            if constexpr (!MF) {
                float lo[4], hi[4];
                *(int4v *) (xpos(buf ^ 1)) = cur.xv[0]; *(int4v *) (xpos(buf ^ 1) + 16) = cur.xv[1];
                decode4<TY>(cur.rw, h, c32, 0, 2*wq, lo); decode4<TY>(cur.rw, h, c32, 1, 2*wq + 1, hi);
                int4v wpk;
                wpk.x = (int) pack_bf16(lo[0], lo[1]); wpk.y = (int) pack_bf16(lo[2], lo[3]);
                wpk.z = (int) pack_bf16(hi[0], hi[1]); wpk.w = (int) pack_bf16(hi[2], hi[3]);
                *(int4v *) (wpos(buf ^ 1)) = wpk;
                __syncthreads();
                __builtin_amdgcn_sched_barrier(0);
                return;
            }
            frags f0, f1;
            read_frags(f0, buf, 0);
            __builtin_amdgcn_sched_barrier(0);
            float lo[4], hi[4];
            // kk = 0: activations, first half
            read_frags(f1, buf, 1);
            mfma(f0, 0);
            *(int4v *) (xpos(buf ^ 1)) = cur.xv[0];
            __builtin_amdgcn_sched_barrier(0);
            mfma(f0, 1);
            decode4<TY>(cur.rw, h, c32, 0, 2*wq, lo);
            __builtin_amdgcn_sched_barrier(0);
            // kk = 1
            read_frags(f0, buf, 2);
            mfma(f1, 0);
            decode4<TY>(cur.rw, h, c32, 1, 2*wq + 1, hi);
            __builtin_amdgcn_sched_barrier(0);
            mfma(f1, 1);
            {
                int4v wpk;
                wpk.x = (int) pack_bf16(lo[0], lo[1]); wpk.y = (int) pack_bf16(lo[2], lo[3]);
                wpk.z = (int) pack_bf16(hi[0], hi[1]); wpk.w = (int) pack_bf16(hi[2], hi[3]);
                *(int4v *) (wpos(buf ^ 1)) = wpk;
            }
            __builtin_amdgcn_sched_barrier(0);
            // kk = 2
            read_frags(f1, buf, 3);
            mfma(f0, 0);
            *(int4v *) (xpos(buf ^ 1) + 16) = cur.xv[1];
            __builtin_amdgcn_sched_barrier(0);
            mfma(f0, 1);
            __builtin_amdgcn_sched_barrier(0);
            // kk = 3
            mfma(f1, 0);
            mfma(f1, 1);
            __syncthreads();
            __builtin_amdgcn_sched_barrier(0);
        };
        stage_regs p0, p1;
        fetch(p0, step0);
        fetch(p1, step0 + 1);
        {
            const int c32 = c32_of(step0);
            const dq_head h = decode_head<TY>(p0.rw, c32);
            float lo[4], hi[4];
            decode4<TY>(p0.rw, h, c32, 0, 2*wq, lo); decode4<TY>(p0.rw, h, c32, 1, 2*wq + 1, hi);
            int4v wpk;
            wpk.x = (int) pack_bf16(lo[0], lo[1]); wpk.y = (int) pack_bf16(lo[2], lo[3]);
            wpk.z = (int) pack_bf16(hi[0], hi[1]); wpk.w = (int) pack_bf16(hi[2], hi[3]);
            *(int4v *) (wpos(0)) = wpk;
            *(int4v *) (xpos(0)) = p0.xv[0]; *(int4v *) (xpos(0) + 16) = p0.xv[1];
        }
        __syncthreads();
        __builtin_amdgcn_sched_barrier(0);
        int s = 0;
        for (; s + 1 < nsteps; s += 2) {
            iteration(p1, p0, s, 0);
            iteration(p0, p1, s + 1, 1);
        }
        if (s < nsteps) iteration(p1, p0, s, 0);
    };
    if constexpr (TYPE2 != TYPE) {
        if (use2) run(std::integral_constant<int, TYPE2>{}, std::true_type{}); else run(std::integral_constant<int, TYPE>{}, std::true_type{});
    } else if constexpr (MOE) {
        if (wn*64 < moe_cnt) run(std::integral_constant<int, TYPE>{}, std::true_type{}); else run(std::integral_constant<int, TYPE>{}, std::false_type{});     // wave-uniform
    } else {
        run(std::integral_constant<int, TYPE>{}, std::true_type{});
    }
    // ---- store: D[row = token][col = weight row] ----
#pragma unroll
    for (int i = 0; i < 2; i++) {
        const int col = m0 + wm*32 + (lane & 31);
#pragma unroll
        for (int r = 0; r < 16; r++) {
            const int row = n0 + wn*64 + i*32 + (r & 3) + 8*(r >> 2) + 4*(lane >> 5);
            float v = acc[i][r];
            if (MOE) {         // as k_mmq's MOE epilogue: bias, then the GLU against a finished up product, or the routing weight
                const int rt = wn*64 + i*32 + (r & 3) + 8*(r >> 2) + 4*(lane >> 5);
                if (col < m && rt < moe_cnt) {
                    const int pair = moe_pairs[moe_first + rt];
                    const int tok = pair/p.n_used, slot = pair - tok*p.n_used;
                    if (p.bias) v += p.bias[(size_t) moe_e*p.bias_stride + col];
                    if (p.glu_up) {
                        const float u = *(const float *) (p.glu_up + (size_t) tok*p.glu_up_nb2 + (size_t) slot*p.glu_up_nb1 + (size_t) col*4);
                        float y;
                        if (p.glu_oai) {
                            const float xc = fminf(v, p.glu_limit), gc = fmaxf(fminf(u, p.glu_limit), -p.glu_limit);
                            y = (xc/(1.0f + expf(-xc*p.glu_alpha)))*(gc + 1.0f);
                        } else y = (v/(1.0f + expf(-v)))*u;
                        if (p.dst) *(float *) (p.dst + (size_t) tok*p.dst_nb2 + (size_t) slot*p.dst_nb1 + (size_t) col*4) = y;
                        if (p.y16) { const uint32_t pk = pack_bf16(y, y); p.y16[(size_t) pair*m + col] = (uint16_t) pk; }
                    } else {
                        if (p.scale) v *= *(const float *) ((const char *) p.scale + (size_t) slot*p.scale_nb0 + (size_t) tok*p.scale_nb1);
                        *(float *) (p.dst + (size_t) tok*p.dst_nb2 + (size_t) slot*p.dst_nb1 + (size_t) col*4) = v;
                    }
                }
                continue;
            }
            if (do_rope) {                     // workgroup-uniform: the pair (2i, 2i + 1) of a head sits in two neighbouring lanes (m is even, heads start at even columns)
                const float other = __shfl_xor(v, 1);
                float x0 = (col & 1) ? other : v, x1 = (col & 1) ? v : other;
                rope_pair(p.rope, p.rope.pos[min(row, n - 1)], (col % p.rope.head_dim) & ~1, x0, x1);
                v = (col & 1) ? x1 : x0;
            }
            if (col < m && row < n) {
                float * o = (float *) (dst + (size_t) row*dst_nb1 + (size_t) col*4);
                if (p.res && p.ksplit == 1) *o = v + *(const float *) (p.res + (size_t) row*p.res_nb1 + (size_t) col*4);
                else if (seg_bias && p.ksplit == 1) *o = v + seg_bias[col];
                else *o = v;
            }
        }
    }
}


// ---- wave-specialized form (round 4) ------------------------------------------------------------------------------------------------
// k_mmq / k_mmq16 make every wave do everything (load, decode, LDS store, operand reads, MFMA) in lock step between barriers: the parts of a k-step ADD UP
// (tools/glu_ablation.py: 62 + 48 + 50 + 38 us over a 51 us floor) and the matrix cores sit at ~36 %. Here the roles are split:
//   * waves 0-7 (two per SIMD) only read operand fragments and issue MFMAs: a wave owns 64 weight rows x 128 tokens (128 accumulator registers; ROWS = 128:
//     32 rows x 128 tokens), 6 ds_read_b128 per 8 MFMAs; the SIMD's two MFMA waves cover each other's LDS latency;
//   * waves 8-11 (one per SIMD) are PRODUCERS: global loads two k-steps ahead (registers), the reference dequantization to bf16, the LDS commits of
//     the next step's weight tile and activation tile — their VALU work runs beside the MFMA waves' matrix instructions (separate pipes).
// One barrier per k-step, two LDS stages. Tile: ROWS = 256 weight rows (DUAL: 128 rows of the gate tensor + the same 128 rows of the up tensor, SwiGLU in the
// epilogue through an LDS exchange between the wave pairs) x 256 tokens x 64 k: a weight is decoded once per 256 tokens, and per k-step a CU stages 72 KB for
// 256 MFMAs (the 128 x 256 tile: 54 KB for 128). 3 waves per SIMD: <= 168 registers.
// The activation tile (256 tokens x 64 k, rows padded to 144 bytes like the weight tile) comes by LDS-DMA issued by the MFMA waves: 37 pieces of 64 16-byte
// chunks = 7 rows of 9 chunks + the first chunk of the next row (written again, with the same bytes, by the next piece), so that every piece has the SAME
// per-lane source offset and a wave-uniform base — no registers, no ds_write, and the waves that issue them have no other memory traffic to wait for.
constexpr int MQ_WS_XPIECES = 37, MQ_WS_XTILE = ((MQ_WS_XPIECES*63 + 1)*16 + 127) & ~127;
template <int TYPE, int ROWS, bool DUAL, int TYPE2 = TYPE>
__global__ void __launch_bounds__(768) __attribute__((amdgpu_waves_per_eu(3, 3))) k_mmq_ws(const mmq_args p) {
    extern __shared__ __attribute__((aligned(16))) char lds[];      // 2 x (W tile | X tile)
    constexpr int BN = 256, WTILE = ROWS*MQ_LD, XTILE = MQ_WS_XTILE, STAGE = WTILE + XTILE;
    constexpr int NR = ROWS/128;                             // 32-row sub-tiles per MFMA wave
    constexpr int TROWS = DUAL ? 128 : ROWS;                 // rows of one tensor per tile
    static_assert(!DUAL || ROWS == 256, "the dual tile is 2 x 128 rows");
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int khalf = p.ksplit > 1 ? (int) blockIdx.y / p.mtiles : 0;
    int mt = (int) blockIdx.y - khalf*p.mtiles;
    const int n0 = blockIdx.x*BN;
    const int n = p.n, k = p.k;
    int m = p.m, col0 = 0; bool use2 = false;
    const char * W = p.W;
    size_t w_row_stride = p.w_row_stride, seg_dst_nb1 = p.dst_nb1; char * seg_dst = p.dst;
    if (!DUAL && p.nseg) {                     // workgroup-uniform (tile0 counts TROWS-row tiles here)
        const int si = (mt >= p.seg[1].tile0 ? 1 : 0) + (p.nseg > 2 && mt >= p.seg[2].tile0 ? 1 : 0);
        W = p.seg[si].W; m = p.seg[si].m; w_row_stride = p.seg[si].w_row_stride; seg_dst = p.seg[si].dst; seg_dst_nb1 = p.seg[si].dst_nb1;
        col0 = p.seg[si].col0; use2 = p.seg[si].type2 != 0; mt -= p.seg[si].tile0;
    }
    const int m0 = mt*TROWS;
    const int kp = (k + MQ_BK - 1) & ~(MQ_BK - 1);
    char * dst = p.ksplit > 1 ? p.dst2 + (size_t) khalf*p.m*p.n*4 + (size_t) col0*4 : seg_dst;
    const size_t dst_nb1 = p.ksplit > 1 ? (size_t) p.m*4 : seg_dst_nb1;
    const int nsteps_all = (k + MQ_BK - 1)/MQ_BK;
    const int nsteps = p.ksplit > 1 ? nsteps_all/p.ksplit : nsteps_all;
    const int step0 = khalf*nsteps;

    if (wave >= 8) {
        // ================= producers =================
        const int ptid = tid - 512, prow = ptid >> 1, shalf = ptid & 1;
        constexpr int NW = ROWS/128;             // weight chunks (32 k of one row) per thread and step
        const char * wrow_p[NW];
#pragma unroll
        for (int c = 0; c < NW; c++) {
            const int lr = prow + 128*c;         // row of the LDS tile; DUAL: rows 128.. are the second tensor's rows 0..
            const int tr = DUAL ? prow : lr;
            wrow_p[c] = ((DUAL && c) ? p.W2 : W) + (size_t) min(m0 + tr, m - 1)*w_row_stride;
        }
        // registers: the weights' raw blocks two k-steps ahead (two stages: HBM latency)
        struct w_regs { raw32 rw[NW]; };
        auto run = [&](auto type_tag) {
            constexpr int TY = decltype(type_tag)::value;
            auto fetch_w = [&](w_regs & r, int step) {
                const int kcl = min(step*MQ_BK + 32*shalf, k - 32);
#pragma unroll
                for (int c = 0; c < NW; c++) r.rw[c] = load_raw32<TY>(wrow_p[c], kcl >> 5);
            };
            auto commit_w = [&](const w_regs & r, int step, int buf) {
#ifdef MI_MMQ_DBG
                if (p.dbg & 8) return;
                if (p.dbg & 1) {
#pragma unroll
                    for (int c = 0; c < NW; c++)
#pragma unroll
                        for (int g = 0; g < 4; g++) *(int4v *) (lds + buf*STAGE + (prow + 128*c)*MQ_LD + shalf*64 + 16*g) = r.rw[c].v[g & 1];
                    return;
                }
#endif
                const int c32 = min(step*MQ_BK + 32*shalf, k - 32) >> 5;
#pragma unroll
                for (int c = 0; c < NW; c++) {
                    const dq_head h = decode_head<TY>(r.rw[c], c32);
                    char * wp = lds + buf*STAGE + (prow + 128*c)*MQ_LD + shalf*64;
#pragma unroll
                    for (int g = 0; g < 4; g++) {
                        int4v wpk;
                        if (TY == T_MXFP4) wpk = decode8_bf16_mxfp4(r.rw[c], h, g);
                        else {
                            float lo[4], hi[4];
                            decode4<TY>(r.rw[c], h, c32, 2*g, 2*g, lo);
                            decode4<TY>(r.rw[c], h, c32, 2*g + 1, 2*g + 1, hi);
                            wpk.x = (int) pack_bf16(lo[0], lo[1]); wpk.y = (int) pack_bf16(lo[2], lo[3]);
                            wpk.z = (int) pack_bf16(hi[0], hi[1]); wpk.w = (int) pack_bf16(hi[2], hi[3]);
                        }
                        *(int4v *) (wp + 16*g) = wpk;
                    }
                }
            };
            // four register stages: the loads of step s + 3 are issued while step s + 1 is decoded — a k-step lasts ~1 us, an HBM load under this traffic ~2 us
            // (with two stages the whole kernel ran at the latency of one load per step: 76 us with everything else switched off)
            w_regs w0, w1, w2, w3;
            fetch_w(w0, step0); fetch_w(w1, step0 + 1); fetch_w(w2, step0 + 2);
            commit_w(w0, step0, 0);
            __syncthreads();
            int s = 0;
            for (; s + 3 < nsteps; s += 4) {
                fetch_w(w3, step0 + s + 3); commit_w(w1, step0 + s + 1, 1); __syncthreads();
                fetch_w(w0, step0 + s + 4); commit_w(w2, step0 + s + 2, 0); __syncthreads();
                fetch_w(w1, step0 + s + 5); commit_w(w3, step0 + s + 3, 1); __syncthreads();
                fetch_w(w2, step0 + s + 6); commit_w(w0, step0 + s + 4, 0); __syncthreads();
            }
            // the last 0..3 steps (what is committed past the end goes to the buffer nobody reads; w1, w2, w3 then w0 hold steps s + 1, s + 2, s + 3, s + 4)
            if (s < nsteps)     { fetch_w(w3, step0 + s + 3); commit_w(w1, step0 + s + 1, 1); __syncthreads(); }
            if (s + 1 < nsteps) { commit_w(w2, step0 + s + 2, 0); __syncthreads(); }
            if (s + 2 < nsteps) { commit_w(w3, step0 + s + 3, 1); __syncthreads(); }
        };
        if constexpr (TYPE2 != TYPE) { if (use2) run(std::integral_constant<int, TYPE2>{}); else run(std::integral_constant<int, TYPE>{}); }
        else run(std::integral_constant<int, TYPE>{});
        if (DUAL) { __syncthreads(); if (p.y16) __syncthreads(); }      // (the epilogue's barriers: exchange, then — with a bf16 output — staging)
        return;
    }

    // ================= MFMA waves =================
    // v_mfma_f32_16x16x32_bf16: the chip holds a higher clock on this shape than on 32x32x16 at the same cycles per FLOP (MI355X_MICROARCH.md, DVFS item 7).
    // A (tokens): lane l holds token l & 15, k = 8 (l >> 4) .. + 8 of a 32-deep slice; B (weight rows) likewise; D: col (weight row) = l & 15, row (token) = 4 (l >> 4) + reg.
    typedef float f32x4 __attribute__((ext_vector_type(4)));
    const int wr = wave & 3, wn = wave >> 2;                 // LDS-tile rows wr*(ROWS/4).., tokens wn*128..
    constexpr int NT = 8, NB = 2*NR;                          // 16-token / 16-row sub-tiles per wave
    f32x4 acc[NT][NB];
#pragma unroll
    for (int i = 0; i < NT; i++)
#pragma unroll
        for (int j = 0; j < NB; j++) acc[i][j] = f32x4{ 0.0f, 0.0f, 0.0f, 0.0f };
    const int foff = (lane & 15)*MQ_LD + (lane >> 4)*16;
    // the activation tile of a step: this wave's pieces P = wave, wave + 8, ... (lane l of a piece: chunk l % 9 of row 7 P + l / 9; chunk 8 is the row's padding — it
    // re-reads chunk 0 — and lane 63 is chunk 0 of row 7 P + 7)
    const uint32_t x_voff = (uint32_t)(lane/9)*(uint32_t) kp*2u + (uint32_t)((lane % 9) & 7)*16u;
    const char * const x_tile = (const char *) (p.X + (size_t) n0*kp);
    auto dma_x = [&](int step, int buf) {
        const char * gstep = x_tile + (size_t) min(step*MQ_BK, kp - MQ_BK)*2;
        const uint32_t lbase = (uint32_t)(size_t)(const char __attribute__((address_space(3))) *) (lds + buf*STAGE + WTILE);
#pragma unroll
        for (int q = 0; q < 5; q++) {
            const int P = wave + 8*q;
            if (P < MQ_WS_XPIECES) {
                // (uniform by construction; said again so that the compiler keeps them in scalar registers)
                const uint64_t ga = (uint64_t)(uintptr_t)(gstep + (size_t)(7*P)*kp*2);
                const char * g = (const char *)(uintptr_t)(((uint64_t)(uint32_t) __builtin_amdgcn_readfirstlane((int)(ga >> 32)) << 32) | (uint32_t) __builtin_amdgcn_readfirstlane((int)(uint32_t) ga));
                const uint32_t ldst = (uint32_t) __builtin_amdgcn_readfirstlane((int)(lbase + (uint32_t) P*(63*16)));
                asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %2" :: "v"(x_voff), "s"(ldst), "s"(g) : "memory");
            }
        }
    };
    dma_x(step0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();                                         // step 0 is in buffer 0
    for (int s = 0; s < nsteps; s++) {
#ifdef MI_MMQ_DBG
        if (!(p.dbg & 32))
#endif
        dma_x(step0 + s + 1, (s + 1) & 1);                   // (past the end: a clamped step into the buffer nobody reads)
        const char * lw = lds + (s & 1)*STAGE + wr*(ROWS/4)*MQ_LD + foff;
        const char * lx = lds + (s & 1)*STAGE + WTILE + wn*128*MQ_LD + foff;
        // A k-step is 2 slices of 32 k x 8 slots (token sub-tile i) of NB MFMAs each. Operand fragments are read ONE SLOT AHEAD into the other register set and the
        // order is pinned: left alone under the 168-register cap, the compiler funnels every fragment through one register set (ds_read, s_waitcnt lgkmcnt(0),
        // MFMAs, again) and every read shows its whole latency.
        int4v af[2], bf[NB];      // (the weight fragments of a slice in ONE register set, re-read at the slice boundary: a second set does not fit 168 registers)
        af[0] = *(const int4v *) lx;
#pragma unroll
        for (int j = 0; j < NB; j++) bf[j] = *(const int4v *) (lw + j*16*MQ_LD);
#ifdef MI_MMQ_DBG
#define MI_WS_RD(c_) if (!(p.dbg & 16)) { c_ }
#define MI_WS_MM(t_) if (p.dbg & 2) { asm volatile("" :: "v"(af[(t_) & 1]), "v"(bf[0])); } else
#else
#define MI_WS_RD(c_) { c_ }
#define MI_WS_MM(t_)
#endif
#define MI_WS_SLOT(t_) { \
            MI_WS_RD( if ((t_) + 1 < 16) af[((t_) + 1) & 1] = *(const int4v *) (lx + (((t_) + 1) & 7)*16*MQ_LD + (((t_) + 1) >> 3)*64); ) \
            MI_WS_MM(t_) { _Pragma("unroll") for (int j = 0; j < NB; j++) \
                acc[(t_) & 7][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, af[(t_) & 1]), __builtin_bit_cast(bf16x8, bf[j]), acc[(t_) & 7][j], 0, 0, 0); } \
            __builtin_amdgcn_sched_group_barrier(0x100, ((t_) + 1 < 16 ? 1 : 0), 0); \
            __builtin_amdgcn_sched_group_barrier(0x008, NB, 0); \
            if ((t_) == 7) { MI_WS_RD( _Pragma("unroll") for (int j = 0; j < NB; j++) bf[j] = *(const int4v *) (lw + j*16*MQ_LD + 64); ) __builtin_amdgcn_sched_group_barrier(0x100, NB, 0); } }
        MI_WS_SLOT(0) MI_WS_SLOT(1) MI_WS_SLOT(2) MI_WS_SLOT(3) MI_WS_SLOT(4) MI_WS_SLOT(5) MI_WS_SLOT(6) MI_WS_SLOT(7)
        MI_WS_SLOT(8) MI_WS_SLOT(9) MI_WS_SLOT(10) MI_WS_SLOT(11) MI_WS_SLOT(12) MI_WS_SLOT(13) MI_WS_SLOT(14) MI_WS_SLOT(15)
#undef MI_WS_SLOT
#undef MI_WS_RD
#undef MI_WS_MM
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // this wave's pieces of the next step's activation tile have landed
        __syncthreads();
    }

    // ---- store: D[row = token][col = weight row]; col = lane & 15, row = 4*(lane >> 4) + r ----
    if constexpr (DUAL) {
        // waves wr = 0, 1 hold the gate product of tile rows wr*64.., waves wr = 2, 3 the up product of the same rows: each pair exchanges half of its
        // accumulators through LDS (the stages are free now) and finishes half of the 64 x 128 elements: token sub-tiles 0..3 by the gate wave, 4..7 by the up wave
        const bool is_up = wr >= 2;                                          // wave-uniform
        float * xch = (float *) lds + (size_t) wave*(4*NB*4*64);           // this wave's outgoing half: [sub-tile][j][r][lane]
        // (accumulator indices are compile-time constants in every branch: a run-time index sends the whole array to scratch)
        auto send = [&](auto base_tag) {
            constexpr int B = decltype(base_tag)::value;
#pragma unroll
            for (int ii = 0; ii < 4; ii++)
#pragma unroll
                for (int j = 0; j < NB; j++)
#pragma unroll
                    for (int r = 0; r < 4; r++) xch[((ii*NB + j)*4 + r)*64 + lane] = acc[B + ii][j][r];
        };
        if (is_up) send(std::integral_constant<int, 0>{}); else send(std::integral_constant<int, 4>{});      // the sub-tiles the partner finishes
        __syncthreads();
        const float * pin = (const float *) lds + (size_t)(wave ^ 2)*(4*NB*4*64);
        // the finished 64 tokens x 64 rows of this wave: f32 straight to dst if asked for; the bf16 copy the next mat-mul reads goes through LDS so that it leaves as
        // whole 128-byte rows (a lane holds ONE column of four tokens: stored directly that is 2-byte elements in 32-byte runs, and the launch ran 6 % slower in the model)
        constexpr int YLD = 64*2 + 16;                                       // bytes per staged token row (padded)
        auto finish = [&](auto base_tag, auto up_tag) {
            constexpr int B = decltype(base_tag)::value; constexpr bool UP = decltype(up_tag)::value;
            float y[4][NB][4];
#pragma unroll
            for (int ii = 0; ii < 4; ii++)
#pragma unroll
                for (int j = 0; j < NB; j++)
#pragma unroll
                    for (int r = 0; r < 4; r++) {
                        const float other = pin[((ii*NB + j)*4 + r)*64 + lane];
                        const float g = UP ? other : acc[B + ii][j][r], u = UP ? acc[B + ii][j][r] : other;
                        y[ii][j][r] = (g/(1.0f + expf(-g)))*u;              // silu(gate)*up, as elem.hip k_glu
                    }
            if (p.dst) {
#pragma unroll
                for (int ii = 0; ii < 4; ii++)
#pragma unroll
                    for (int j = 0; j < NB; j++) {
                        const int col = m0 + (wr & 1)*64 + j*16 + (lane & 15);
#pragma unroll
                        for (int r = 0; r < 4; r++) {
                            const int row = n0 + wn*128 + (B + ii)*16 + 4*(lane >> 4) + r;
                            if (col < m && row < n) *(float *) (dst + (size_t) row*dst_nb1 + (size_t) col*4) = y[ii][j][r];
                        }
                    }
            }
            if (p.y16) {
                __syncthreads();                                             // every wave has read its partner's half: the exchange area is free
                char * stg = lds + (size_t) wave*(64*YLD);
#pragma unroll
                for (int ii = 0; ii < 4; ii++)
#pragma unroll
                    for (int j = 0; j < NB; j++)
#pragma unroll
                        for (int r = 0; r < 4; r++)
                            *(uint16_t *) (stg + (ii*16 + 4*(lane >> 4) + r)*YLD + (j*16 + (lane & 15))*2) = (uint16_t) pack_bf16(y[ii][j][r], y[ii][j][r]);
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");         // (only this wave reads what it staged)
                const int colb = m0 + (wr & 1)*64;
#pragma unroll
                for (int q = 0; q < 8; q++) {
                    const int tk = q*8 + (lane >> 3), ch = lane & 7;         // token row of the staged tile, 16-byte chunk (8 rows of the weight tile)
                    const int row = n0 + wn*128 + B*16 + tk, col = colb + ch*8;
                    const int4v v = *(const int4v *) (stg + tk*YLD + ch*16);
                    if (row < n && (m & 7) == 0 && col + 8 <= m) *(int4v *) (p.y16 + (size_t) row*m + col) = v;
                    else if (row < n) { for (int t = 0; t < 8; t++) if (col + t < m) p.y16[(size_t) row*m + col + t] = *(const uint16_t *) (stg + tk*YLD + ch*16 + 2*t); }
                }
            }
        };
        if (is_up) finish(std::integral_constant<int, 4>{}, std::true_type{}); else finish(std::integral_constant<int, 0>{}, std::false_type{});
    } else {
#pragma unroll
        for (int i = 0; i < NT; i++)
#pragma unroll
            for (int j = 0; j < NB; j++) {
                const int col = m0 + wr*(ROWS/4) + j*16 + (lane & 15);
#pragma unroll
                for (int r = 0; r < 4; r++) {
                    const int row = n0 + wn*128 + i*16 + 4*(lane >> 4) + r;
                    if (col < m && row < n) {
                        float * o = (float *) (dst + (size_t) row*dst_nb1 + (size_t) col*4);
                        if (p.res && p.ksplit == 1) *o = acc[i][j][r] + *(const float *) (p.res + (size_t) row*p.res_nb1 + (size_t) col*4);
                        else *o = acc[i][j][r];
                    }
                }
            }
    }
}
constexpr size_t MQ_LDS_BYTES_WS256 = 2*((size_t) 256*MQ_LD + MQ_WS_XTILE);
static bool mmq_ws_on() { static const bool on = !getenv("GGML_MI355X_MMQ_WS") || atoi(getenv("GGML_MI355X_MMQ_WS")) != 0; return on; }

constexpr size_t MQ_LDS_BYTES = 4*(size_t) MQ_BM*MQ_LD;
constexpr size_t MQ_LDS_BYTES_256 = 2*(size_t)(MQ_BM + 256)*MQ_LD;
constexpr size_t MQ_LDS_BYTES_DUAL = 2*(size_t)(2*MQ_BM + 256)*MQ_LD;

// 108 KB of dynamic LDS: more than the 64 KB a kernel gets without asking
template <int T_>
static void launch_mmq_wide(dim3 grid, const mmq_args & a, hipStream_t stream) {
    MI_LDS_LIMIT_OR_DIE(MQ_LDS_BYTES_256, k_mmq<T_, 256>);
    static const bool w16 = !getenv("GGML_MI355X_MMQ16") || atoi(getenv("GGML_MI355X_MMQ16")) != 0;
    if (w16 && !a.moe && a.ne12 == 1 && a.r2 == 1 && a.r3 == 1) {
        MI_LDS_LIMIT_OR_DIE(MQ_LDS_BYTES_256, k_mmq16<T_>);
        hipLaunchKernelGGL((k_mmq16<T_>), grid, dim3(1024), MQ_LDS_BYTES_256, stream, a);
        return;
    }
    hipLaunchKernelGGL((k_mmq<T_, 256>), grid, dim3(512), MQ_LDS_BYTES_256, stream, a);
}

// (room for whole 256-token tiles + 8 rows behind the last one: k_mmq_ws copies an activation tile HBM/L2 -> LDS in 1 KiB pieces of 7 rows + one chunk; rows past n are never stored)
static size_t mmq_x_bytes(int64_t k, int64_t n) { return ((size_t)((n + 255)/256*256 + 8)*mmq_kp(k)*2 + 255) & ~(size_t) 255; }
size_t mul_mat_q_x_bytes(int64_t k, int64_t n) { return mmq_x_bytes(k, n); }
size_t mul_mat_q_scratch_bytes(int64_t k, int64_t n, int64_t m) { return mmq_x_bytes(k, n) + (size_t) 4*m*n*4 + 512; }     // bf16 copy of x | up to 4 split-k planes

// dst = plane 0 + plane 1 [+ plane 2 + plane 3] [+ residual], always in this order; 4 consecutive weight rows of one token per thread
template <int NP>
__global__ void __launch_bounds__(256) k_combine(char * dst, size_t dst_nb1, const float * planes, const char * res, size_t res_nb1, int64_t m4, int64_t n) {
    const int64_t i = (int64_t) blockIdx.x*256 + threadIdx.x;
    if (i >= m4*n) return;
    const int64_t row = i / m4, c4 = i - row*m4;
    float4v a = ((const float4v *) planes)[i];
#pragma unroll
    for (int pl = 1; pl < NP; pl++) { const float4v b = ((const float4v *) planes)[(int64_t) pl*m4*n + i]; a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w; }
    if (res) { const float4v b = *(const float4v *) (res + (size_t) row*res_nb1 + (size_t) c4*16); a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w; }
    *(float4v *) (dst + (size_t) row*dst_nb1 + (size_t) c4*16) = a;
}

void mul_mat_q(int type_a, const void * W, size_t w_row_stride, int64_t m, int64_t k,
               const float * x, size_t x_row_stride, int64_t n, void * scratch, bool scratch_ready, float * dst, size_t dst_col_stride_bytes,
               const float * res, size_t res_row_stride, hipStream_t stream, mmq_deferred * defer) {
    if (defer) { defer->np = 0; defer->planes = nullptr; }
    if (m == 0 || n == 0) return;
    uint16_t * xb = (uint16_t *) scratch;
    if (!scratch_ready) {
        act16_args pa = { (const char *) x, x_row_stride, 0, 0, k, n, 1, xb };
        hipLaunchKernelGGL((k_act_to_16<false>), dim3((unsigned)((mmq_kp(k) + 1023)/1024), (unsigned) n, 1), dim3(256), 0, stream, pa);
    }
    mmq_args a = { (const char *) W, w_row_stride, 0, 0, (int) m, (int) k, xb, (int) n, (char *) dst, dst_col_stride_bytes, 0, 0, 1, 1, 1, 1, 0, nullptr,
                   (const char *) res, res_row_stride, nullptr, nullptr, nullptr, 0, 0, 0 };
    const int mtiles = (int)((m + MQ_BM - 1)/MQ_BM);
    a.mtiles = mtiles;
    // 256-token tiles when they still fill the chip (m = 14336, n = 512: 224 workgroups); else 128-token tiles, and a grid that would
    // leave the chip half empty (m = 4096, n = 512: 128 tiles on 256 CUs) is split in two along k
    const int64_t wtiles = (int64_t) mtiles*((n + 255)/256);
    const bool vec_ok = m % 4 == 0 && dst_col_stride_bytes % 16 == 0 && ((uintptr_t) dst % 16) == 0 && (!res || (res_row_stride % 16 == 0 && ((uintptr_t) res % 16) == 0));
    // few rows (wo: m = 4096, k = 4096; ffn_down: k = 14336): 256-token tiles and k in four parts (with the 16-wave kernel this beats
    // 128-token tiles split in two also at k = 4096: wo 49 -> 42 us)
    static const int64_t wide4_min_k = getenv("GGML_MI355X_WIDE4_MINK") ? atoll(getenv("GGML_MI355X_WIDE4_MINK")) : 4096;
    const bool wide4 = n >= 256 && wtiles < 160 && wtiles*4 >= 160 && k % 1024 == 0 && k >= wide4_min_k && vec_ok;
    const bool wide = (n >= 256 && wtiles >= 160) || wide4;
    const int ntiles = wide ? (int)((n + 255)/256) : (int)((n + MQ_BN - 1)/MQ_BN);
    float * planes = (float *) ((char *) scratch + mmq_x_bytes(k, n));
    if (wide4) a.ksplit = 4;
    else if (!wide && (int64_t) ntiles*mtiles <= 160 && k % 512 == 0 && k >= 2048 && vec_ok) a.ksplit = 2;
    if (a.ksplit > 1) a.dst2 = (char *) planes;
    const dim3 grid((unsigned) ntiles, (unsigned)(mtiles*a.ksplit), 1);
#define MI_MMQ(T_) do { if (wide) launch_mmq_wide<T_>(grid, a, stream); \
                        else      hipLaunchKernelGGL((k_mmq<T_, 128>), grid, dim3(256), MQ_LDS_BYTES, stream, a); } while (0)
    switch (type_a) {
        case T_Q4_0:  MI_MMQ(T_Q4_0);  break;
        case T_Q8_0:  MI_MMQ(T_Q8_0);  break;
        case T_Q4_K:  MI_MMQ(T_Q4_K);  break;
        case T_Q5_K:  MI_MMQ(T_Q5_K);  break;
        case T_Q6_K:  MI_MMQ(T_Q6_K);  break;
        case T_MXFP4: MI_MMQ(T_MXFP4); break;
        default: fprintf(stderr, "mmq: unsupported type %d\n", type_a); abort();
    }
#undef MI_MMQ
    if (defer && a.ksplit > 1) { defer->np = a.ksplit; defer->planes = planes; return; }
    const unsigned cgrid = (unsigned)((m/4*n + 255)/256);
    if (a.ksplit == 4)      hipLaunchKernelGGL(k_combine<4>, dim3(cgrid), dim3(256), 0, stream, (char *) dst, dst_col_stride_bytes, planes, (const char *) res, res_row_stride, m/4, n);
    else if (a.ksplit == 2) hipLaunchKernelGGL(k_combine<2>, dim3(cgrid), dim3(256), 0, stream, (char *) dst, dst_col_stride_bytes, planes, (const char *) res, res_row_stride, m/4, n);
}

// ---- several mat-muls on the same activations as one launch (wq / wk / wv of build_attn, src/llama-model.cpp:6017-6040) ----
// st_*: the KV-cache writes that follow wk / wv (SET_ROWS, src/llama-kv-cache-unified.cpp:1123,1157-1167) done by the same pass — mode 1: f16 row
// idx[token] of the K cache (st_row_elems apart); mode 2: element scatter into the transposed V cache, st16[idx[token*m_seg + col]]
struct combine_seg_args { const float * planes; int64_t m4_tot, n; int nseg; char * dst[3]; size_t dst_nb1[3]; int col4[3]; int rope_seg[3]; fused_rope rope;
                          uint16_t * st16[3]; const int64_t * st_idx[3]; int64_t st_row_elems[3]; int st_mode[3]; int m_seg[3]; };
template <int NP>
__global__ void __launch_bounds__(256) k_combine_seg(const combine_seg_args p) {
    const int64_t i = (int64_t) blockIdx.x*256 + threadIdx.x;
    if (i >= p.m4_tot*p.n) return;
    const int64_t row = i / p.m4_tot; const int c4 = (int)(i - row*p.m4_tot);
    float4v a = ((const float4v *) p.planes)[i];
#pragma unroll
    for (int pl = 1; pl < NP; pl++) { const float4v b = ((const float4v *) p.planes)[(int64_t) pl*p.m4_tot*p.n + i]; a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w; }
    const int si = (c4 >= p.col4[1] ? 1 : 0) + (p.nseg > 2 && c4 >= p.col4[2] ? 1 : 0);
    if (p.rope_seg[si]) {                      // 4 consecutive columns = 2 pairs of one head
        const int cih = ((c4 - p.col4[si])*4) % p.rope.head_dim, pos = p.rope.pos[row];
        float x0 = a.x, x1 = a.y, x2 = a.z, x3 = a.w;
        rope_pair(p.rope, pos, cih, x0, x1);
        rope_pair(p.rope, pos, cih + 2, x2, x3);
        a = float4v{ x0, x1, x2, x3 };
    }
    *(float4v *) (p.dst[si] + (size_t) row*p.dst_nb1[si] + (size_t)(c4 - p.col4[si])*16) = a;
    if (p.st_mode[si] == 1) {
        uint16_t * q = p.st16[si] + p.st_idx[si][row]*p.st_row_elems[si] + (size_t)(c4 - p.col4[si])*4;
        *(uint2 *) q = uint2{ (uint32_t) f32_to_f16_bits(a.x) | ((uint32_t) f32_to_f16_bits(a.y) << 16), (uint32_t) f32_to_f16_bits(a.z) | ((uint32_t) f32_to_f16_bits(a.w) << 16) };
    } else if (p.st_mode[si] == 2) {
        const int64_t * ix = p.st_idx[si] + row*p.m_seg[si] + (size_t)(c4 - p.col4[si])*4;
        p.st16[si][ix[0]] = f32_to_f16_bits(a.x); p.st16[si][ix[1]] = f32_to_f16_bits(a.y); p.st16[si][ix[2]] = f32_to_f16_bits(a.z); p.st16[si][ix[3]] = f32_to_f16_bits(a.w);
    }
}

template <int T_, int T2_>
static void launch_mmq_multi(dim3 grid, const mmq_args & a, hipStream_t stream) {
    MI_LDS_LIMIT_OR_DIE(MQ_LDS_BYTES_256, k_mmq<T_, 256, false, T2_>);
    static const bool w16 = !getenv("GGML_MI355X_MMQ16") || atoi(getenv("GGML_MI355X_MMQ16")) != 0;
    if (w16) {
        MI_LDS_LIMIT_OR_DIE(MQ_LDS_BYTES_256, k_mmq16<T_, T2_>);
        hipLaunchKernelGGL((k_mmq16<T_, T2_>), grid, dim3(1024), MQ_LDS_BYTES_256, stream, a);
        return;
    }
    hipLaunchKernelGGL((k_mmq<T_, 256, false, T2_>), grid, dim3(512), MQ_LDS_BYTES_256, stream, a);
}

// false: not done (too few tiles for 256-token tiles, an unsupported mix of types, scratch too small) — the caller runs them one by one
bool mul_mat_q_multi(int nseg, const int * types, const void * const * W, const size_t * w_row_stride, const int64_t * m, float * const * dst, const size_t * dst_stride,
                     int64_t k, const float * x, size_t x_row_stride, int64_t n, void * scratch, size_t scratch_size, bool scratch_ready,
                     const mmvq_rope * rope, const int * seg_rope, const mmq_kv_store * kvs, hipStream_t stream, const float * const * seg_bias) {
    if (nseg < 2 || nseg > 3 || n < 256) return false;
    static const bool w16 = !getenv("GGML_MI355X_MMQ16") || atoi(getenv("GGML_MI355X_MMQ16")) != 0;
    if (rope) {        // epilogue / combine-pass ROPE: 16-wave kernel only, heads of an even size that start at column multiples of 4
        if (!w16 || rope->head_dim % 4 != 0 || rope->p.n_dims % 2 != 0) return false;
        for (int s = 0; s < nseg; s++) if (seg_rope[s] && m[s] % rope->head_dim != 0) return false;
    }
    int t1 = types[0], t2 = types[0];
    for (int s = 1; s < nseg; s++) if (types[s] != t1) { if (t2 != t1 && types[s] != t2) return false; t2 = types[s]; }
    if (t1 != t2) {
        if (t1 == T_Q6_K) { const int t = t1; t1 = t2; t2 = t; }
        if (!(t2 == T_Q6_K && (t1 == T_Q4_K || t1 == T_Q5_K))) return false;       // the mixed pairs of the K-quant mixes (attn_v one step up)
    }
    mmq_args a = { nullptr, 0, 0, 0, 0, (int) k, (const uint16_t *) scratch, (int) n, nullptr, 0, 0, 0, 1, 1, 1, 1, 0, nullptr, nullptr, 0, nullptr, nullptr, nullptr, 0, 0, 0 };
    a.nseg = nseg;
    int64_t m_tot = 0; int tiles = 0; bool vec_ok = true;
    for (int s = 0; s < nseg; s++) {
        a.seg[s] = { (const char *) W[s], (char *) dst[s], w_row_stride[s], dst_stride[s], (int) m[s], tiles, (int) m_tot, types[s] != t1 ? 1 : 0, rope && seg_rope[s] ? 1 : 0,
                     seg_bias ? seg_bias[s] : nullptr };
        vec_ok = vec_ok && m[s] % 4 == 0 && dst_stride[s] % 16 == 0 && ((uintptr_t) dst[s] % 16) == 0;
        m_tot += m[s]; tiles += (int)((m[s] + MQ_BM - 1)/MQ_BM);
    }
    if (nseg < 3) a.seg[2] = a.seg[1];
    if (rope) a.rope = make_fused_rope(*rope);
    if (m_tot >= (1ll << 30)) return false;
    a.m = (int) m_tot; a.mtiles = tiles;
    const int ntiles = (int)((n + 255)/256);
    const int64_t wt = (int64_t) tiles*ntiles;
    if (wt >= 160) a.ksplit = 1;
    else if (wt*2 >= 160 && k % 512 == 0 && k >= 2048 && vec_ok) a.ksplit = 2;
    else if (wt*4 >= 160 && k % 1024 == 0 && k >= 4096 && vec_ok) a.ksplit = 4;
    else if (wt >= 64) a.ksplit = 1;        // k does not split (gpt-oss: 2880): still one launch of 80 tiles instead of three of 64 + 8 + 8 on 256 CUs
    else return false;
    // the rotation rides on the combine pass (one sincos per pair, no lane exchange); in the mat-mul epilogue both lanes of a pair would
    // evaluate it (measured: pp2048 -2.6 % against separate ROPE kernels), so without a k split the caller keeps its ROPE launches
    if ((rope || kvs) && a.ksplit == 1) return false;
    if (seg_bias && a.ksplit != 1) return false;          // (the bias rides on the epilogue only)
    if (mmq_x_bytes(k, n) + (a.ksplit > 1 ? (size_t) a.ksplit*m_tot*n*4 : 0) + 512 > scratch_size) return false;
    if (!scratch_ready) {
        act16_args pa = { (const char *) x, x_row_stride, 0, 0, k, n, 1, (uint16_t *) scratch };
        hipLaunchKernelGGL((k_act_to_16<false>), dim3((unsigned)((mmq_kp(k) + 1023)/1024), (unsigned) n, 1), dim3(256), 0, stream, pa);
    }
    float * planes = (float *) ((char *) scratch + mmq_x_bytes(k, n));
    if (a.ksplit > 1) a.dst2 = (char *) planes;
    const dim3 grid((unsigned) ntiles, (unsigned)(tiles*a.ksplit), 1);
    if (t1 != t2) {
        if (t1 == T_Q4_K) launch_mmq_multi<T_Q4_K, T_Q6_K>(grid, a, stream); else launch_mmq_multi<T_Q5_K, T_Q6_K>(grid, a, stream);
    } else switch (t1) {
        case T_Q4_0:  launch_mmq_wide<T_Q4_0>(grid, a, stream);  break;
        case T_Q8_0:  launch_mmq_wide<T_Q8_0>(grid, a, stream);  break;
        case T_Q4_K:  launch_mmq_wide<T_Q4_K>(grid, a, stream);  break;
        case T_Q5_K:  launch_mmq_wide<T_Q5_K>(grid, a, stream);  break;
        case T_Q6_K:  launch_mmq_wide<T_Q6_K>(grid, a, stream);  break;
        case T_MXFP4: launch_mmq_wide<T_MXFP4>(grid, a, stream); break;
        default: fprintf(stderr, "mmq_multi: unsupported type %d\n", t1); abort();
    }
    if (a.ksplit > 1) {
        combine_seg_args ca = { planes, m_tot/4, n, nseg, { nullptr, nullptr, nullptr }, { 0, 0, 0 }, { 0, 0, 0 }, { 0, 0, 0 }, a.rope,
                                { nullptr, nullptr, nullptr }, { nullptr, nullptr, nullptr }, { 0, 0, 0 }, { 0, 0, 0 }, { 0, 0, 0 } };
        if (kvs) for (int s = 0; s < nseg; s++) { ca.st16[s] = kvs->st16[s]; ca.st_idx[s] = kvs->st_idx[s]; ca.st_row_elems[s] = kvs->st_row_elems[s]; ca.st_mode[s] = kvs->st_mode[s]; ca.m_seg[s] = (int) m[s]; }
        for (int s = 0; s < 3; s++) { const int q = s < nseg ? s : nseg - 1; ca.dst[s] = a.seg[q].dst; ca.dst_nb1[s] = a.seg[q].dst_nb1; ca.col4[s] = a.seg[q].col0/4; ca.rope_seg[s] = a.seg[q].rope; }
        const unsigned cgrid = (unsigned)((m_tot/4*n + 255)/256);
        if (a.ksplit == 4) hipLaunchKernelGGL(k_combine_seg<4>, dim3(cgrid), dim3(256), 0, stream, ca);
        else               hipLaunchKernelGGL(k_combine_seg<2>, dim3(cgrid), dim3(256), 0, stream, ca);
    }
    return true;
}

// gate / up + SwiGLU of build_ffn (src/llama-graph.cpp:632-774) for many tokens: dst[n][m] = silu(Wg.x) * (Wu.x)
template <int T_>
static void launch_mmq_dual(dim3 grid, const mmq_args & a, hipStream_t stream) {
    // the wave-specialized kernel where its producers' four register stages fit (Q4_K, MXFP4: 168 registers, no scratch; the other formats' raw blocks are larger and spill)
    if constexpr (T_ == T_Q4_K || T_ == T_MXFP4) if (mmq_ws_on()) {
        MI_LDS_LIMIT_OR_DIE(MQ_LDS_BYTES_WS256, k_mmq_ws<T_, 256, true>);
        hipLaunchKernelGGL((k_mmq_ws<T_, 256, true>), grid, dim3(768), MQ_LDS_BYTES_WS256, stream, a);
        return;
    }
    MI_LDS_LIMIT_OR_DIE(MQ_LDS_BYTES_DUAL, k_mmq<T_, 256, true>);
    hipLaunchKernelGGL((k_mmq<T_, 256, true>), grid, dim3(512), MQ_LDS_BYTES_DUAL, stream, a);
}
bool mul_mat_q_glu_supported(int64_t m, int64_t n) { return n >= 256 && ((m + MQ_BM - 1)/MQ_BM)*((n + 255)/256) >= 160; }
void mul_mat_q_glu(int type_a, const void * Wg, const void * Wu, size_t w_row_stride, int64_t m, int64_t k,
                   const float * x, size_t x_row_stride, int64_t n, void * scratch, bool scratch_ready, float * dst, size_t dst_col_stride_bytes, hipStream_t stream,
                   uint16_t * y16) {
    if (m == 0 || n == 0) return;
    uint16_t * xb = (uint16_t *) scratch;
    if (!scratch_ready) {
        act16_args pa = { (const char *) x, x_row_stride, 0, 0, k, n, 1, xb };
        hipLaunchKernelGGL((k_act_to_16<false>), dim3((unsigned)((mmq_kp(k) + 1023)/1024), (unsigned) n, 1), dim3(256), 0, stream, pa);
    }
    mmq_args a = { (const char *) Wg, w_row_stride, 0, 0, (int) m, (int) k, xb, (int) n, (char *) dst, dst_col_stride_bytes, 0, 0, 1, 1, 1, 1, 0, nullptr, nullptr, 0, y16, (const char *) Wu, nullptr, 0, 0, 0 };
    a.mtiles = (int)((m + MQ_BM - 1)/MQ_BM);
    { static const int dbg = getenv("GGML_MI355X_MMQ_DBG") ? atoi(getenv("GGML_MI355X_MMQ_DBG")) : 0; a.dbg = dbg; }
    const dim3 grid((unsigned)((n + 255)/256), (unsigned) a.mtiles, 1);
    switch (type_a) {
        case T_Q4_0:  launch_mmq_dual<T_Q4_0>(grid, a, stream);  break;
        case T_Q8_0:  launch_mmq_dual<T_Q8_0>(grid, a, stream);  break;
        case T_Q4_K:  launch_mmq_dual<T_Q4_K>(grid, a, stream);  break;
        case T_Q5_K:  launch_mmq_dual<T_Q5_K>(grid, a, stream);  break;
        case T_Q6_K:  launch_mmq_dual<T_Q6_K>(grid, a, stream);  break;
        case T_MXFP4: launch_mmq_dual<T_MXFP4>(grid, a, stream); break;
        default: fprintf(stderr, "mmq_glu: unsupported type %d\n", type_a); abort();
    }
}

// ---- MUL_MAT_ID for many tokens: pairs (token, slot) sorted by expert, then the tiled kernel above per (expert, 128 pairs) ----
// one workgroup: counts per expert, tile table, counting sort of the pair ids. Which position a pair gets inside its expert's
// range is not deterministic (LDS atomics) and does not matter: every pair's result row is computed independently.
struct moe_sort_args { const char * ids; size_t ids_nb0, ids_nb1; int n_used, n_tokens, n_expert, max_tiles; int * out; int tile; };
__global__ void __launch_bounds__(256) k_moe_sort(const moe_sort_args p) {
    __shared__ int cnt[256], off[256], cur[256];
    const int tid = threadIdx.x, n_pairs = p.n_used*p.n_tokens;
    if (tid < 256) { cnt[tid] = 0; cur[tid] = 0; }
    __syncthreads();
    for (int i = tid; i < n_pairs; i += 256) {
        const int e = *(const int32_t *) (p.ids + (size_t)(i % p.n_used)*p.ids_nb0 + (size_t)(i/p.n_used)*p.ids_nb1);
        atomicAdd(&cnt[e], 1);
    }
    __syncthreads();
    if (tid == 0) {
        int o = 0, nt = 0;
        int * tiles = p.out + 1;
        for (int e = 0; e < p.n_expert; e++) {
            off[e] = o;
            for (int r = 0; r < cnt[e]; r += p.tile) { tiles[3*nt] = e; tiles[3*nt + 1] = o + r; tiles[3*nt + 2] = min(p.tile, cnt[e] - r); nt++; }
            o += cnt[e];
        }
        p.out[0] = nt;
    }
    __syncthreads();
    int * pairs = p.out + 1 + 3*p.max_tiles;
    for (int i = tid; i < n_pairs; i += 256) {
        const int e = *(const int32_t *) (p.ids + (size_t)(i % p.n_used)*p.ids_nb0 + (size_t)(i/p.n_used)*p.ids_nb1);
        pairs[off[e] + atomicAdd(&cur[e], 1)] = i;
    }
}

static int moe_max_tiles(int64_t n_pairs, int64_t n_expert) { return (int)(n_pairs/MQ_BN + n_expert); }      // (an upper bound for 128- and 256-pair tiles)
static size_t moe_table_bytes(int64_t n_pairs, int64_t n_expert) { return ((size_t)(1 + 3*moe_max_tiles(n_pairs, n_expert) + n_pairs)*4 + 255) & ~(size_t) 255; }
bool mul_mat_q_id_supported(int64_t n_expert, int64_t n_used, int64_t n_tokens) { return n_expert <= 256 && n_used*n_tokens < (1 << 24); }
size_t mul_mat_q_id_scratch_bytes(int64_t k, int64_t n_b, int64_t n_tokens, int64_t n_used, int64_t n_expert) {
    return mmq_x_bytes(k, n_b*n_tokens) + moe_table_bytes(n_used*n_tokens, n_expert) + 256;
}
// the fused expert chain (gate / up / GLU -> down): bf16 copy of the layer input | tile table | bf16 GLU result, one row per (token, slot) pair
mmq_moe_plan mul_mat_q_id_plan(void * scratch, int64_t k, int64_t n_tokens, int64_t n_used, int64_t n_expert, int64_t m_glu) {
    mmq_moe_plan pl;
    pl.xb = (uint16_t *) scratch;
    pl.table = (int *) ((char *) scratch + mmq_x_bytes(k, n_tokens));
    pl.table2 = (int *) ((char *) pl.table + moe_table_bytes(n_used*n_tokens, n_expert));
    pl.y16 = (uint16_t *) ((char *) pl.table2 + moe_table_bytes(n_used*n_tokens, n_expert));
    pl.bytes = mmq_x_bytes(k, n_tokens) + 2*moe_table_bytes(n_used*n_tokens, n_expert) + (((size_t) n_used*n_tokens*mmq_kp(m_glu)*2 + 255) & ~(size_t) 255) + 256;
    return pl;
}
int mul_mat_q_id_tile(int64_t n_used, int64_t n_tokens, int64_t n_expert) {
    static const int forced = getenv("GGML_MI355X_MOE_TILE") ? atoi(getenv("GGML_MI355X_MOE_TILE")) : 0;
    if (forced == 128 || forced == 256) return forced;
    // measured at 512 tokens: Mixtral (~128 pairs per expert, every wave of a tile multiplies): 128-pair tiles 9.85k tok/s, 256-pair tiles on the 16-wave kernel 7.9k
    // (on k_mmq<256> 6.97k); gpt-oss (64 pairs per expert: a quarter of the 16-wave kernel's waves multiply, all sixteen decode): 256-pair tiles 22.85k, 128-pair 21.7k
    return n_used*n_tokens <= 96*n_expert ? 256 : 128;
}
void mul_mat_q_id_sort(const int32_t * ids, size_t ids_nb0, size_t ids_nb1, int64_t n_used, int64_t n_tokens, int64_t n_expert, int tile, int * table, hipStream_t stream) {
    moe_sort_args ps = { (const char *) ids, ids_nb0, ids_nb1, (int) n_used, (int) n_tokens, (int) n_expert, moe_max_tiles(n_used*n_tokens, n_expert), table, tile };
    hipLaunchKernelGGL(k_moe_sort, dim3(1), dim3(256), 0, stream, ps);
}
void mul_mat_q_id_act16(const float * b, size_t b_nb1, size_t b_nb2, int64_t k, int64_t n_b, int64_t n_tokens, uint16_t * xb, hipStream_t stream) {
    act16_args pa = { (const char *) b, b_nb1, b_nb2, 0, k, n_b, n_tokens, xb };     // dense [token][n_b][k] bf16
    hipLaunchKernelGGL((k_act_to_16<false>), dim3((unsigned)((mmq_kp(k) + 1023)/1024), (unsigned) n_b, (unsigned) n_tokens), dim3(256), 0, stream, pa);
}

template <int T_>
static void launch_mmq_moe(dim3 grid, const mmq_args & a, int tile, bool dual, hipStream_t stream) {
    if (dual) {
        MI_LDS_LIMIT_OR_DIE(MQ_LDS_BYTES_DUAL, k_mmq<T_, 256, true, T_, true>);
        hipLaunchKernelGGL((k_mmq<T_, 256, true, T_, true>), grid, dim3(512), MQ_LDS_BYTES_DUAL, stream, a);
    } else if (tile == 256) {
        MI_LDS_LIMIT_OR_DIE(MQ_LDS_BYTES_256, k_mmq16<T_, T_, true>);
        hipLaunchKernelGGL((k_mmq16<T_, T_, true>), grid, dim3(1024), MQ_LDS_BYTES_256, stream, a);
    } else {
        MI_LDS_LIMIT_OR_DIE(MQ_LDS_BYTES, k_mmq<T_, 128, false, T_, true>);
        hipLaunchKernelGGL((k_mmq<T_, 128, false, T_, true>), grid, dim3(256), MQ_LDS_BYTES, stream, a);
    }
}

// the tiled kernel over a sorted pair table. xb: bf16 rows [token][n_b][kp(k)] (n_b == 1: every slot of a token reads the token's row; n_b == n_used: a row per pair);
// W2 != NULL: the dual kernel (tile must be 256): GLU(W.x + bias, W2.x + bias2) -> dst (f32, may be NULL) and / or y16 (bf16 rows of m per pair)
void mul_mat_q_id_tiles(int type_a, const void * W, const void * W2, size_t w_row_stride, size_t w_expert_stride, int64_t m, int64_t k,
                        const uint16_t * xb, int64_t n_b, const int * table, int tile, int64_t n_used, int64_t n_tokens, int64_t n_expert,
                        const mmq_moe_epi & epi, float * dst, size_t dst_nb1, size_t dst_nb2, uint16_t * y16, hipStream_t stream) {
    if (m == 0 || n_used*n_tokens == 0) return;
    if (W2 && tile != 256) { fprintf(stderr, "mmq_id: the dual kernel runs 256-pair tiles\n"); abort(); }
    const int max_tiles = moe_max_tiles(n_used*n_tokens, n_expert);
    mmq_args a = { (const char *) W, w_row_stride, w_expert_stride, 0, (int) m, (int) k, xb, (int)(n_used*n_tokens), (char *) dst, dst_nb1, dst_nb2, 0, 1, 1, 1, 1, 0, nullptr, nullptr, 0, y16, (const char *) W2,
                   table, max_tiles, (int) n_used, (int) n_b };
    a.mtiles = (int)((m + MQ_BM - 1)/MQ_BM);
    a.bias = epi.bias; a.bias2 = epi.bias2; a.bias_stride = epi.bias_stride; a.scale = epi.scale; a.scale_nb0 = epi.scale_nb0; a.scale_nb1 = epi.scale_nb1;
    a.glu_oai = epi.oai; a.glu_alpha = epi.alpha; a.glu_limit = epi.limit;
    a.glu_up = (const char *) epi.glu_up; a.glu_up_nb1 = epi.glu_up_nb1; a.glu_up_nb2 = epi.glu_up_nb2;
    const dim3 grid((unsigned) max_tiles, (unsigned) a.mtiles, 1);
#define MI_MMQ(T_) launch_mmq_moe<T_>(grid, a, tile, W2 != nullptr, stream)
    switch (type_a) {
        case T_Q4_0:  MI_MMQ(T_Q4_0);  break;
        case T_Q8_0:  MI_MMQ(T_Q8_0);  break;
        case T_Q4_K:  MI_MMQ(T_Q4_K);  break;
        case T_Q5_K:  MI_MMQ(T_Q5_K);  break;
        case T_Q6_K:  MI_MMQ(T_Q6_K);  break;
        case T_MXFP4: MI_MMQ(T_MXFP4); break;
        default: fprintf(stderr, "mmq_id: unsupported type %d\n", type_a); abort();
    }
#undef MI_MMQ
}

void mul_mat_q_id(int type_a, const void * W, size_t w_row_stride, size_t w_expert_stride, int64_t m, int64_t k,
                  const float * b, size_t b_nb1, size_t b_nb2, int64_t n_b,
                  const int32_t * ids, size_t ids_nb0, size_t ids_nb1, int64_t n_used, int64_t n_tokens, int64_t n_expert,
                  void * scratch, float * dst, size_t dst_nb1, size_t dst_nb2, hipStream_t stream) {
    if (m == 0 || n_used*n_tokens == 0) return;
    uint16_t * xb = (uint16_t *) scratch;
    int * table = (int *) ((char *) scratch + mmq_x_bytes(k, n_b*n_tokens));
    const int tile = mul_mat_q_id_tile(n_used, n_tokens, n_expert);
    mul_mat_q_id_act16(b, b_nb1, b_nb2, k, n_b, n_tokens, xb, stream);
    mul_mat_q_id_sort(ids, ids_nb0, ids_nb1, n_used, n_tokens, n_expert, tile, table, stream);
    mul_mat_q_id_tiles(type_a, W, nullptr, w_row_stride, w_expert_stride, m, k, xb, n_b, table, tile, n_used, n_tokens, n_expert, mmq_moe_epi{}, dst, dst_nb1, dst_nb2, nullptr, stream);
}

// f16 x f32 with ggml broadcast on the matrix cores (n > 8 columns): a rows contiguous f16, b rows contiguous f32
bool mul_mat_dense_mfma_supported(const mm_dense_args & p) {
    return p.type_a == T_F16 && p.type_b == T_F32 && p.nb00 == 2 && p.nb10 == 4 && p.ne11 > MMVQ_MAX_N && p.ne00 % 32 == 0 &&
           p.ne12*p.ne13 <= 65535 && p.ne01 < (1ll << 30) && p.ne11 < (1ll << 30);
}
size_t mul_mat_dense_mfma_scratch_bytes(const mm_dense_args & p) { return mmq_x_bytes(p.ne10, p.ne11*p.ne12*p.ne13) + 256; }

void mul_mat_dense_mfma(const mm_dense_args & p, void * scratch, hipStream_t stream) {
    if (p.ne01 == 0 || p.ne11 == 0 || p.ne12*p.ne13 == 0) return;
    uint16_t * xb = (uint16_t *) scratch;
    const int64_t nbatch = p.ne12*p.ne13;
    act16_args pa = { (const char *) p.b, p.nb11, p.nb12, p.nb13, p.ne10, p.ne11, p.ne12, xb };
    hipLaunchKernelGGL((k_act_to_16<true>), dim3((unsigned)((mmq_kp(p.ne10) + 1023)/1024), (unsigned) p.ne11, (unsigned) nbatch), dim3(256), 0, stream, pa);
    mmq_args a = { (const char *) p.a, p.nb01, p.nb02, p.nb03, (int) p.ne01, (int) p.ne00, xb, (int) p.ne11, (char *) p.dst, p.nb1, p.nb2, p.nb3,
                   (int) p.ne12, (int)(p.ne12/p.ne02), (int)(p.ne13/p.ne03), 1, 0, nullptr, nullptr, 0, nullptr, nullptr, nullptr, 0, 0, 0 };
    const dim3 grid((unsigned)((p.ne11 + MQ_BN - 1)/MQ_BN), (unsigned)((p.ne01 + MQ_BM - 1)/MQ_BM), (unsigned) nbatch);
    hipLaunchKernelGGL((k_mmq<T_F16>), grid, dim3(256), MQ_LDS_BYTES, stream, a);
}

} // namespace mi355x
