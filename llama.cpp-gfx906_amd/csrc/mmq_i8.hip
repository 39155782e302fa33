// mmq_i8.hip — prefill mat-mul (n > 8 tokens) of Q4_K weights on the INT8 matrix cores, in the CPU path's own arithmetic: activations
// quantized to Q8_K (int8 + one f32 scale per 256 block, quantize_act.hip), per 32-element sub-block one v_mfma_i32_32x32x32_i8 whose result
// is weighted by the sub-block's 6-bit scale in integers (v_mad_i32_i24), per 256 block one f32 FMA with d_w * d_a, and the mins through ONE
// f16 matrix instruction per two blocks: sum_j (dmin*m_j) * (d_a*bsum_j) (v_mfma_f32_32x32x16_f16: k = 2 blocks x 8 sub-blocks).
// Against the bf16 kernel (mmq.hip): no dequantization to bf16 — the weight blocks go to LDS as raw bytes (coalesced 16-byte copies; a first version
// with every lane loading its row's nibbles from global memory spent its time in the address unit: 32 cache lines per load instruction) and a
// lane reads its row's nibbles from there straight into the B operand — and twice the matrix rate; the price is 16 integer multiply-adds
// per matrix instruction.
//
// Tiling: a wave owns 64 weight rows (two 32-row tiles) x 64 tokens (two 32-token tiles): weights are the B operand so that a lane's 16 results
// of a tile belong to ONE weight row (its scale is a per-lane value) and 16 tokens; a workgroup = 4 waves = 256 rows x 64 tokens; the tokens'
// int8 rows of the current 256-block sit in LDS (padded rows: conflict-free ds_read_b128), double-buffered, filled through registers.
#include "kernels.h"
#include "dev_common.h"
#include "blocks.h"

#include <algorithm>

namespace mi355x {

typedef int      i32x16 __attribute__((ext_vector_type(16)));
typedef float    f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8  __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2  __attribute__((ext_vector_type(2)));

struct mmq_i8_args {
    const uint8_t * W; size_t w_stride;      // Q4_K rows
    const int8_t * qs; const float * d; const int16_t * bs;     // Q8_K image of the n tokens
    float * dst; size_t ldd;                  // dst[token*ldd + row]
    int m, k, n;
    int dbg;                                   // GGML_MI355X_MMQ_I8_DBG (ablations, wrong results): 1 = no global loads after the first block, 2 = no compute
};

constexpr int I8_TOK = 64;                    // tokens per workgroup
constexpr int I8_PITCH = 256 + 16;            // LDS bytes per token row of one 256-block
constexpr int I8_ACT = I8_TOK*I8_PITCH + I8_TOK*4;   // + the block's 64 activation scales
// a buffer = I8_ACT + the workgroup's weight rows of the block as raw 144-byte Q4_K blocks (row pitch 144 B = 36 banks: the 16 rows of a
// ds_read_b128 group land on 16 distinct multiples of 4 banks — conflict-free as it is)
constexpr int i8_buf_bytes(int rows) { return I8_ACT + rows*144; }

static __device__ __forceinline__ i32x16 mfma_i8(int4v a, int4v b, i32x16 c) {
    return __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b, c, 0, 0, 0);
}


// (round 3: the first variant of this kernel — one integer multiply-add per matrix result and sub-block, GGML_MI355X_MMQ_I8_VARIANT=0 — is removed: it was
// 20 % slower than the split-scale kernel below on every shape measured in round 2 and nothing was built on it)


// ---- variant "split scale": no integer work on the matrix results inside a block. The 6-bit sub-block scale is folded into the B operand instead:
// sc = 8*hi + lo with hi, lo in 0..7, so q*hi and q*lo (<= 105) are int8 again — four of them per register by ONE v_pk_mul_lo_u16 (a 16-bit lane
// holds two bytes, (q_a + 256 q_b)*hi carries nothing across) — and two matrix instructions per sub-block accumulate sum_j hi_j*dot_j and
// sum_j lo_j*dot_j over the whole 256 block in their C operands. Per block and result: isum = (acc_hi << 3) + acc_lo, then the same f32 FMA. The
// operand arithmetic (16 instructions per sub-block) is paid once per weight fragment and shared by the wave's token tiles, where the
// multiply-adds of the first variant were paid per result. A wave owns 32 rows x 64 tokens; a workgroup 128 rows x 64 tokens.
typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));
static __device__ __forceinline__ int pkmul(int a, uint32_t b2) {
    return __builtin_bit_cast(int, (u16x2)(__builtin_bit_cast(u16x2, a)*__builtin_bit_cast(u16x2, b2)));
}

// NW = 4: a wave owns 32 rows x 64 tokens (2 waves per SIMD); NW = 8: 32 rows x 32 tokens, waves 2i / 2i + 1 share a row group (4 waves per SIMD at <= 128 registers)
// TOK = 128 (NW = 4): a wave owns 32 rows x 128 tokens — the operand arithmetic of a weight fragment is shared by four token tiles (8 matrix instructions per
// 18 vector ones); 12 accumulator sets = 192 registers, so one wave per SIMD
template <int NW, int TOK>
__global__ void __launch_bounds__(NW*64, TOK == 128 ? 1 : 2) k_mmq_i8s_q4_K(const mmq_i8_args p) {
    constexpr int ROWS = 128, ACT = TOK*I8_PITCH + TOK*4, I8_BUF = ACT + ROWS*144, NT = NW == 4 ? TOK/32 : 1, NTH = NW*64;
    constexpr int I8_TOK = TOK, I8_ACT = ACT;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int col = lane & 31, kh = lane >> 5;
    const int nb = p.k >> 8;
    const int ntok = (p.n + I8_TOK - 1)/I8_TOK;
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    const int rblk = (slot/ntok)*8 + xcd;
    if (rblk*ROWS >= p.m) return;
    const int tok0 = (slot % ntok)*I8_TOK;
    const int rgrp = NW == 4 ? wave : wave >> 1, tw = NW == 4 ? 0 : (wave & 1)*32;
    const int row0 = rblk*ROWS + rgrp*32;
    const bool active = row0 < p.m;

    constexpr int AST = TOK*16/NTH;                                 // activation chunks per thread
    const int s_chunk = tid & 15, s_tok = tid >> 4;
    int4v stage[AST];
    float stage_d = 0.0f;
    constexpr int WCH = ROWS*9, WST = (WCH + NTH - 1)/NTH;
    int4v wstage[WST];
    // per-thread source pointers, advanced by one 256-block per step (the address arithmetic of the copy is otherwise a fifth of the loop's instructions)
    const int8_t * sp[AST]; const float * sdp = p.d + (size_t) min(tok0 + (tid & (I8_TOK - 1)), p.n - 1)*nb; const uint8_t * wp[WST];
#pragma unroll
    for (int i = 0; i < AST; i++) sp[i] = p.qs + (size_t) min(tok0 + s_tok + (NTH/16)*i, p.n - 1)*p.k + s_chunk*16;
#pragma unroll
    for (int i = 0; i < WST; i++) { const int ch = min(tid + NTH*i, WCH - 1); wp[i] = p.W + (size_t) min(rblk*ROWS + ch/9, p.m - 1)*p.w_stride + (ch % 9)*16; }
    auto stage_load = [&](int) {
#pragma unroll
        for (int i = 0; i < AST; i++) { stage[i] = *(const int4v *) sp[i]; sp[i] += 256; }
        if (tid < I8_TOK) { stage_d = *sdp; sdp++; }
#pragma unroll
        for (int i = 0; i < WST; i++) { if (tid + NTH*i < WCH) wstage[i] = ld_b128(wp[i]); wp[i] += 144; }
    };
    auto stage_store = [&](int buf) {
        char * b = smem + buf*I8_BUF;
#pragma unroll
        for (int i = 0; i < AST; i++) *(int4v *) (b + (s_tok + (NTH/16)*i)*I8_PITCH + s_chunk*16) = stage[i];
        if (tid < I8_TOK) *(float *) (b + I8_TOK*I8_PITCH + tid*4) = stage_d;
#pragma unroll
        for (int i = 0; i < WST; i++) { const int ch = tid + NTH*i; if (ch < WCH) *(int4v *) (b + I8_ACT + ch*16) = wstage[i]; }
    };

    f32x16 accf[NT];
#pragma unroll
    for (int t = 0; t < NT; t++)
#pragma unroll
        for (int r = 0; r < 16; r++) accf[t][r] = 0.0f;
    f16x8 minw;
#pragma unroll
    for (int j = 0; j < 8; j++) minw[j] = (_Float16) 0.0f;

    stage_load(0);
    stage_store(0);
    __syncthreads();

    for (int kb = 0; kb < nb; kb++) {
        const int buf = kb & 1;
        const char * lb = smem + buf*I8_BUF;
        if (kb + 1 < nb && !(p.dbg & 1)) stage_load(kb + 1);
        if (active && !(p.dbg & 2)) {
            const char * wl = lb + I8_ACT + (rgrp*32 + col)*144;
            const int4v hd = *(const int4v *) wl;
            const uint32_t dd = (uint32_t) hd.x, s0 = (uint32_t) hd.y, s1 = (uint32_t) hd.z, s2 = (uint32_t) hd.w;
            const uint32_t scp[2] = { s0 & 0x3F3F3F3Fu, (s2 & 0x0F0F0F0Fu) | ((s0 >> 2) & 0x30303030u) };
            const uint32_t hip[2] = { (scp[0] >> 3) & 0x07070707u, (scp[1] >> 3) & 0x07070707u };
            const uint32_t lop[2] = { scp[0] & 0x07070707u, scp[1] & 0x07070707u };
            {
                const uint32_t m0 = s1 & 0x3F3F3F3Fu, m1 = ((s2 >> 4) & 0x0F0F0F0Fu) | ((s1 >> 2) & 0x30303030u);
                const float ndmin = -f16_bits_to_f32((uint16_t)(dd >> 16));
                f16x8 mw;
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    mw[j]     = (_Float16)(ndmin*(float)((m0 >> (8*j)) & 0xFF));
                    mw[j + 4] = (_Float16)(ndmin*(float)((m1 >> (8*j)) & 0xFF));
                }
                if ((kb & 1) == kh) minw = mw;
            }
            i32x16 ahi[NT], alo[NT], zero;
#pragma unroll
            for (int r = 0; r < 16; r++) zero[r] = 0;
#pragma unroll
            for (int g = 0; g < 4; g++) {
                const int4v raw = *(const int4v *) (wl + 16 + g*32 + kh*16);
#pragma unroll
                for (int h = 0; h < 2; h++) {
                    const int sub = 2*g + h;
                    const int4v nib = h ? ((raw >> 4) & 0x0F0F0F0F) : (raw & 0x0F0F0F0F);
                    const uint32_t hi = (hip[sub >> 2] >> (8*(sub & 3))) & 0xFF, lo = (lop[sub >> 2] >> (8*(sub & 3))) & 0xFF;
                    const uint32_t hi2 = hi | (hi << 16), lo2 = lo | (lo << 16);
                    const int4v bhi = { pkmul(nib.x, hi2), pkmul(nib.y, hi2), pkmul(nib.z, hi2), pkmul(nib.w, hi2) };
                    const int4v blo = { pkmul(nib.x, lo2), pkmul(nib.y, lo2), pkmul(nib.z, lo2), pkmul(nib.w, lo2) };
#pragma unroll
                    for (int t = 0; t < NT; t++) {
                        const int4v aop = *(const int4v *) (lb + (tw + t*32 + col)*I8_PITCH + sub*32 + kh*16);
                        ahi[t] = mfma_i8(aop, bhi, sub == 0 ? zero : ahi[t]);
                        alo[t] = mfma_i8(aop, blo, sub == 0 ? zero : alo[t]);
                    }
                }
            }
            // ---- block end: isum = 8*hi-sum + lo-sum; acc += isum * d_w * d_a(token)
            const float * dl = (const float *) (lb + I8_TOK*I8_PITCH);
            const float dw = f16_bits_to_f32((uint16_t)(dd & 0xFFFF));
#pragma unroll
            for (int t = 0; t < NT; t++) {
#pragma unroll
                for (int q4 = 0; q4 < 4; q4++) {
                    const float4v v = *(const float4v *) (dl + tw + t*32 + q4*8 + kh*4);
                    const float da[4] = { v.x, v.y, v.z, v.w };
#pragma unroll
                    for (int e = 0; e < 4; e++) {
                        const int r = q4*4 + e;
                        accf[t][r] = fmaf((float)((ahi[t][r] << 3) + alo[t][r]), dw*da[e], accf[t][r]);
                    }
                }
            }
            if ((kb & 1) || kb == nb - 1) {
                const int blk = (kb & ~1) + kh;
#pragma unroll
                for (int t = 0; t < NT; t++) {
                    const int tok = min(tok0 + tw + t*32 + col, p.n - 1);
                    f16x8 ab;
                    if (blk < nb) {
                        const float dtok = p.d[(size_t) tok*nb + blk];
                        const int4v b0 = *(const int4v *) (p.bs + (size_t) tok*(p.k >> 4) + blk*16);
                        const int4v b1 = *(const int4v *) (p.bs + (size_t) tok*(p.k >> 4) + blk*16 + 8);
                        const int w[8] = { b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w };
#pragma unroll
                        for (int j = 0; j < 8; j++) ab[j] = (_Float16)(dtok*(float)((int)(short)(w[j] & 0xFFFF) + (w[j] >> 16)));
                    } else {
#pragma unroll
                        for (int j = 0; j < 8; j++) ab[j] = (_Float16) 0.0f;
                    }
                    f16x8 bw = minw;
                    if ((kb & 1) == 0 && kh == 1) {
#pragma unroll
                        for (int j = 0; j < 8; j++) bw[j] = (_Float16) 0.0f;
                    }
                    accf[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ab, bw, accf[t], 0, 0, 0);
                }
            }
        }
        if (kb + 1 < nb) stage_store(buf ^ 1);
        __syncthreads();
    }
    if (!active) return;
#pragma unroll
    for (int t = 0; t < NT; t++)
#pragma unroll
        for (int r = 0; r < 16; r++) {
            const int tok = tok0 + tw + t*32 + (r >> 2)*8 + kh*4 + (r & 3);
            if (tok < p.n) p.dst[(size_t) tok*p.ldd + row0 + col] = accf[t][r];
        }
}

bool mul_mat_q_i8_supported(int type_a, int64_t m, int64_t k, int64_t n) {
    return type_a == T_Q4_K && m % 64 == 0 && k % 256 == 0 && n > 8 && m < (1ll << 30) && k < (1ll << 24);
}

// dst[token*ldd + row] = W[row, :] . x[token, :] with x given as its Q8_K image (quantize_act); false = shape not served
bool mul_mat_q_i8(int type_a, const void * W, size_t w_row_stride, int64_t m, int64_t k, const act_q8 & act, int64_t n,
                  float * dst, size_t dst_col_stride_bytes, hipStream_t stream) {
    if (!mul_mat_q_i8_supported(type_a, m, k, n) || act.kind != T_Q8_K) return false;
    mmq_i8_args a;
    a.W = (const uint8_t *) W; a.w_stride = w_row_stride; a.qs = act.qs; a.d = act.d; a.bs = act.bsums;
    a.dst = dst; a.ldd = dst_col_stride_bytes/4; a.m = (int) m; a.k = (int) k; a.n = (int) n;
    static const int dbg = getenv("GGML_MI355X_MMQ_I8_DBG") ? atoi(getenv("GGML_MI355X_MMQ_I8_DBG")) : 0;
    a.dbg = dbg;
    const int64_t ntok = (n + I8_TOK - 1)/I8_TOK;
    {
        const dim3 gs((unsigned)((((m + 127)/128 + 7)/8)*8*ntok));
        static const int nw = getenv("GGML_MI355X_MMQ_I8_WAVES") ? atoi(getenv("GGML_MI355X_MMQ_I8_WAVES")) : 8;      // 8 | 4 | 1 (= 4 waves, 128-token tiles)
        if (nw == 1) {
            const int64_t ntok128 = (n + 127)/128;
            const dim3 g1((unsigned)((((m + 127)/128 + 7)/8)*8*ntok128));
            constexpr int lds = 2*(128*I8_PITCH + 128*4 + 128*144);
            MI_LDS_LIMIT_OR_DIE(lds, k_mmq_i8s_q4_K<4, 128>); hipLaunchKernelGGL((k_mmq_i8s_q4_K<4, 128>), g1, dim3(256), lds, stream, a);
        }
        else if (nw == 4) { MI_LDS_LIMIT_OR_DIE(2*i8_buf_bytes(128), k_mmq_i8s_q4_K<4, 64>); hipLaunchKernelGGL((k_mmq_i8s_q4_K<4, 64>), gs, dim3(256), 2*i8_buf_bytes(128), stream, a); }
        else              { MI_LDS_LIMIT_OR_DIE(2*i8_buf_bytes(128), k_mmq_i8s_q4_K<8, 64>); hipLaunchKernelGGL((k_mmq_i8s_q4_K<8, 64>), gs, dim3(512), 2*i8_buf_bytes(128), stream, a); }
    }
    return true;
}

} // namespace mi355x
