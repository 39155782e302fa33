// attn_prefill.hip — K.q -> soft_max(scale, mask, sinks) -> V.kq -> [hd*n_head, T] for MANY tokens (T > 8) as one kernel on the matrix
// cores, with an online softmax: the n_kv x T score matrix never goes to memory.
//
// Replaces, for the node group build_attn_mha emits without flash attention (src/llama-graph.cpp:1283-1330):
//   MUL_MAT(k, q) [prec F32] -> SOFT_MAX(mask, scale) [+ sinks] -> MUL_MAT(v, kq) -> PERMUTE -> CONT
// with the unified KV cache's layouts (src/llama-kv-cache-unified.cpp:1056-1106): K rows [hd] per cell, V TRANSPOSED (v_trans: one
// row of cells per head dimension). Arithmetic as the node-by-node path: q and the probabilities rounded to f16 (the vec_dot type
// of an F16 matrix), f32 accumulation, soft_max in f32.
//
// One workgroup of 8 waves per (head, 32 queries); wave w takes the cell blocks w, w + 8, ... (a block whose mask is -inf for all of the
// 32 x 32 pairs — the causal future — is skipped before any arithmetic) and the partial (max, sum, O) states are merged through
// LDS at the end. Everything is computed TRANSPOSED so that a lane owns ONE query column:
//   S^T[32 cells x 32 queries] = K[32 x hd] . Q^T      (v_mfma_f32_32x32x16_f16, A = K rows straight from the cache, B = Q^T)
//   per-lane online softmax over the 16 cells a lane holds (+ one exchange with lane ^ 32: the other 16 cells of the same query)
//   O^T[hd x 32 queries] += V^T[hd x 32 cells] . P^T   (A = rows of the transposed V cache, B = P^T straight from the registers of S^T)
// The C-layout of S^T (cell = (r & 3) + 8 (r >> 2) + 4 (lane >> 5)) IS a valid B-operand k-slot order as long as V^T's k-slots use
// the same cell permutation — a sum over cells does not care about their order — so the probabilities never leave the registers.
#include "dev_common.h"
#include "kernels.h"

namespace mi355x {

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef float    f32x16 __attribute__((ext_vector_type(16)));

struct attn_pf_args {
    const char * q; size_t q_nb1, q_nb2;                // q [hd, T, n_head] f32 (permuted view): nb1 = token stride, nb2 = head stride
    const char * k; size_t k_nb1, k_nb2;                // k [hd, n_kv, n_head_kv] f16: nb1 = cell stride, nb2 = head stride
    const char * v; size_t v_nb1, v_nb2;                // v [n_kv, hd, n_head_kv] f16 (transposed): nb1 = dim stride, nb2 = head stride
    const char * mask; size_t m_nb1; int mask_f16;      // mask [n_kv, T_pad]
    const float * sinks;
    float * dst; size_t dst_nb1;                        // [hd*n_head, T]
    int n_kv, n_head, n_head_kv, T;
    float scale;
};

static __device__ __forceinline__ float xhalf(float v, int lane) {      // the value lane ^ 32 holds
    return __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute((lane ^ 32) << 2, __builtin_bit_cast(int, v)));
}

// VT: transposed V cache (rows over cells). !VT: V rows are cells (FLASH_ATTN_EXT): the 8 cells of a k-slot group are then 8 two-byte
// gathers per operand — correct, not fast; a transposing LDS read is the next step for that layout.
constexpr int APF_NW = 8;       // 2 waves per SIMD at 256 registers each: one workgroup per CU
// MASK: 0 = none, 1 = f32, 2 = f16 — a template parameter, not a branch: every load of a block (mask, K, V) is issued before the first
// use, so a block costs one memory round trip (as run-time branches the loads sat in blocks of their own, each waiting for its data)
template <int HD, bool VT, int MASK>
__global__ void __launch_bounds__(64*APF_NW) k_attn_prefill(const attn_pf_args p) {
    constexpr int NC = HD/16, NDT = HD/32;
    extern __shared__ float part[];                           // [(APF_NW - 1)*(16*NDT + 2)*64]: the other waves' o[NDT][16], m, l per lane
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, ql = lane & 31, hf = lane >> 5;
    const int h = blockIdx.y, hk = h/(p.n_head/p.n_head_kv);
    const int q0 = blockIdx.x*32;
    const int t = min(q0 + ql, p.T - 1);

    // Q^T as B operand: chunk c holds head dims 16c + 8 hf .. + 7 of query t, rounded to f16
    f16x8 qb[NC];
    {
        const char * qrow = p.q + (size_t) t*p.q_nb1 + (size_t) h*p.q_nb2;
#pragma unroll
        for (int c = 0; c < NC; c++) {
            const float4v a = *(const float4v *) (qrow + (size_t)(16*c + 8*hf)*4), b = *(const float4v *) (qrow + (size_t)(16*c + 8*hf + 4)*4);
            qb[c] = f16x8{ (_Float16) a.x, (_Float16) a.y, (_Float16) a.z, (_Float16) a.w, (_Float16) b.x, (_Float16) b.y, (_Float16) b.z, (_Float16) b.w };
        }
    }
    f32x16 o[NDT];
#pragma unroll
    for (int d = 0; d < NDT; d++)
#pragma unroll
        for (int r = 0; r < 16; r++) o[d][r] = 0.0f;
    float m = -INFINITY, l = 0.0f;     // running maximum (common to both halves) and this half's share of the denominator

    const char * kbase = p.k + (size_t) hk*p.k_nb2 + (size_t)(8*hf)*2;
    const char * vbase = VT ? p.v + (size_t) hk*p.v_nb2 + (size_t) ql*p.v_nb1 + (size_t)(4*hf)*2
                            : p.v + (size_t) hk*p.v_nb2 + (size_t) ql*2 + (size_t)(4*hf)*p.v_nb1;
    const char * mrow = p.mask ? p.mask + (size_t) t*p.m_nb1 : nullptr;

    for (int kv0 = 32*wave; kv0 < p.n_kv; kv0 += 32*APF_NW) {
        // ---- the mask of this lane's 16 cells; a block nobody may look at is skipped ----
        float mk[16];
        if (MASK == 2) {
            int2v raw[4];
#pragma unroll
            for (int j = 0; j < 4; j++) raw[j] = ld_b64(mrow + (size_t)(kv0 + 8*j + 4*hf)*2);
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const f16x4 hv = __builtin_bit_cast(f16x4, raw[j]);
                mk[4*j] = (float) hv[0]; mk[4*j + 1] = (float) hv[1]; mk[4*j + 2] = (float) hv[2]; mk[4*j + 3] = (float) hv[3];
            }
        } else if (MASK == 1) {
            float4v fv[4];
#pragma unroll
            for (int j = 0; j < 4; j++) fv[j] = __builtin_bit_cast(float4v, ld_b128(mrow + (size_t)(kv0 + 8*j + 4*hf)*4));
#pragma unroll
            for (int j = 0; j < 4; j++) { mk[4*j] = fv[j].x; mk[4*j + 1] = fv[j].y; mk[4*j + 2] = fv[j].z; mk[4*j + 3] = fv[j].w; }
        } else {
#pragma unroll
            for (int r = 0; r < 16; r++) mk[r] = 0.0f;
        }
        // K rows and the V operands of the whole block: in flight together with the mask
        const char * krow = kbase + (size_t)(kv0 + ql)*p.k_nb1;
        int4v kraw[NC];
#pragma unroll
        for (int c = 0; c < NC; c++) kraw[c] = *(const int4v *) (krow + (size_t) c*32);
        int4v va[2][NDT];
#pragma unroll
        for (int c2 = 0; c2 < 2; c2++)
#pragma unroll
            for (int d = 0; d < NDT; d++) {
                if (VT) {
                    const char * vp = vbase + (size_t)(32*d)*p.v_nb1 + (size_t)(kv0 + 16*c2)*2;
                    const int2v lo = ld_b64(vp), hi = ld_b64(vp + 16);
                    va[c2][d] = int4v{ lo.x, lo.y, hi.x, hi.y };
                } else {
                    const char * vp = vbase + (size_t)(32*d)*2 + (size_t)(kv0 + 16*c2)*p.v_nb1;      // cell kv0 + 16 c2 + 4 hf, dim 32 d + ql
                    uint32_t hv[8];
#pragma unroll
                    for (int sl = 0; sl < 8; sl++) hv[sl] = ld_u16(vp + (size_t)(8*(sl >> 2) + (sl & 3))*p.v_nb1);
                    va[c2][d] = int4v{ (int)(hv[0] | (hv[1] << 16)), (int)(hv[2] | (hv[3] << 16)), (int)(hv[4] | (hv[5] << 16)), (int)(hv[6] | (hv[7] << 16)) };
                }
            }
        if (MASK) {
            float mx = -INFINITY;
#pragma unroll
            for (int r = 0; r < 16; r++) mx = fmaxf(mx, mk[r]);
            if (__builtin_amdgcn_ballot_w64(mx != -INFINITY) == 0) continue;       // wave-uniform
        }
        // ---- S^T = K . Q^T ----
        f32x16 s;
#pragma unroll
        for (int r = 0; r < 16; r++) s[r] = 0.0f;
#pragma unroll
        for (int c = 0; c < NC; c++) s = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, kraw[c]), qb[c], s, 0, 0, 0);
        // ---- scale + mask; s[r] belongs to cell kv0 + (r & 3) + 8 (r >> 2) + 4 hf of query t ----
        float bm = -INFINITY;
#pragma unroll
        for (int r = 0; r < 16; r++) { s[r] = s[r]*p.scale + mk[r]; bm = fmaxf(bm, s[r]); }
        bm = fmaxf(bm, xhalf(bm, lane));
        const float m_new = fmaxf(m, bm);
        const float alpha = m == -INFINITY ? 0.0f : expf(m - m_new);     // m_new == -inf only while every cell so far was masked: p = 0
        float psum = 0.0f;
        float pr[16];
#pragma unroll
        for (int r = 0; r < 16; r++) { pr[r] = s[r] == -INFINITY ? 0.0f : expf(s[r] - m_new); psum += pr[r]; }
        l = l*alpha + psum;
        m = m_new;
#pragma unroll
        for (int d = 0; d < NDT; d++)
#pragma unroll
            for (int r = 0; r < 16; r++) o[d][r] *= alpha;
        // ---- O^T += V^T . P^T: the k-slots of chunk c2 are the cells kv0 + 16 c2 + 8 (s >> 2) + 4 hf + (s & 3), s = 0..7 ----
#pragma unroll
        for (int c2 = 0; c2 < 2; c2++) {
            const f16x8 pb = { (_Float16) pr[8*c2 + 0], (_Float16) pr[8*c2 + 1], (_Float16) pr[8*c2 + 2], (_Float16) pr[8*c2 + 3],
                               (_Float16) pr[8*c2 + 4], (_Float16) pr[8*c2 + 5], (_Float16) pr[8*c2 + 6], (_Float16) pr[8*c2 + 7] };
#pragma unroll
            for (int d = 0; d < NDT; d++) {
                o[d] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, va[c2][d]), pb, o[d], 0, 0, 0);
            }
        }
    }
    // ---- merge the four waves' states (same lane = same query and the same cells-within-block pattern) ----
    constexpr int PS = 16*NDT + 2;
    if (wave > 0) {
        float * pp = part + (size_t)(wave - 1)*PS*64 + lane;
#pragma unroll
        for (int d = 0; d < NDT; d++)
#pragma unroll
            for (int r = 0; r < 16; r++) pp[(16*d + r)*64] = o[d][r];
        pp[(16*NDT)*64] = m; pp[(16*NDT + 1)*64] = l;
    }
    __syncthreads();
    if (wave > 0) return;
#pragma unroll 1
    for (int w = 1; w < APF_NW; w++) {
        const float * pp = part + (size_t)(w - 1)*PS*64 + lane;
        const float mw = pp[(16*NDT)*64], lw = pp[(16*NDT + 1)*64];
        const float m_new = fmaxf(m, mw);
        const float a = m == -INFINITY ? 0.0f : expf(m - m_new), b = mw == -INFINITY ? 0.0f : expf(mw - m_new);
        l = l*a + lw*b; m = m_new;
#pragma unroll
        for (int d = 0; d < NDT; d++)
#pragma unroll
            for (int r = 0; r < 16; r++) o[d][r] = o[d][r]*a + pp[(16*d + r)*64]*b;
    }
    // ---- finish: both halves' denominators, the sink logit (src/llama-graph.cpp:1313), normalise, store ----
    float lt = l + xhalf(l, lane);
    float fin = 1.0f;
    if (p.sinks) {
        const float sk = p.sinks[h];
        const float mf = fmaxf(m, sk);
        fin = m == -INFINITY ? 0.0f : expf(m - mf);
        lt = lt*fin + expf(sk - mf);
    }
    const float inv = fin/lt;
    if (q0 + ql < p.T) {
        float * orow = (float *) ((char *) p.dst + (size_t) t*p.dst_nb1) + (size_t) h*HD;
#pragma unroll
        for (int d = 0; d < NDT; d++)
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const float4v ov = { o[d][4*j]*inv, o[d][4*j + 1]*inv, o[d][4*j + 2]*inv, o[d][4*j + 3]*inv };
                *(float4v *) (orow + 32*d + 8*j + 4*hf) = ov;
            }
    }
}

bool attn_prefill_supported(int64_t head_dim, int64_t n_kv) { return (head_dim == 128 || head_dim == 64) && n_kv % 32 == 0 && n_kv > 0; }

void attn_prefill(const void * q, size_t q_nb1, size_t q_nb2, const void * k, size_t k_nb1, size_t k_nb2, const void * v, size_t v_nb1, size_t v_nb2,
                  const void * mask, size_t m_nb1, bool mask_f16, const float * sinks, float * dst, size_t dst_nb1,
                  int64_t head_dim, int64_t n_kv, int64_t n_head, int64_t n_head_kv, int64_t T, float scale, hipStream_t stream, bool v_trans) {
    attn_pf_args a = { (const char *) q, q_nb1, q_nb2, (const char *) k, k_nb1, k_nb2, (const char *) v, v_nb1, v_nb2,
                       (const char *) mask, m_nb1, mask_f16 ? 1 : 0, sinks, dst, dst_nb1, (int) n_kv, (int) n_head, (int) n_head_kv, (int) T, scale };
    const dim3 grid((unsigned)((T + 31)/32), (unsigned) n_head);
    const int mk = !mask ? 0 : (mask_f16 ? 2 : 1);
#define MI_APF1(HD_, VT_, MK_) do { \
        constexpr size_t lds_ = (size_t)(APF_NW - 1)*(16*(HD_/32) + 2)*64*4; \
        static const bool once_ = [] { MI_HIP_CHECK(hipFuncSetAttribute((const void *) k_attn_prefill<HD_, VT_, MK_>, hipFuncAttributeMaxDynamicSharedMemorySize, (int) lds_)); return true; }(); \
        (void) once_; \
        hipLaunchKernelGGL((k_attn_prefill<HD_, VT_, MK_>), grid, dim3(64*APF_NW), lds_, stream, a); } while (0)
#define MI_APF(HD_, VT_) do { if (mk == 0) MI_APF1(HD_, VT_, 0); else if (mk == 1) MI_APF1(HD_, VT_, 1); else MI_APF1(HD_, VT_, 2); } while (0)
    if (!v_trans) { if (head_dim == 128) MI_APF(128, false); else MI_APF(64, false); }
    else          { if (head_dim == 128) MI_APF(128, true);  else MI_APF(64, true); }
#undef MI_APF
#undef MI_APF1
}

} // namespace mi355x
