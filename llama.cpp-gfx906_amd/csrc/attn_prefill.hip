// attn_prefill.hip — K.q -> soft_max(scale, mask, sinks) -> V.kq -> [hd*n_head, T] for MANY tokens (T > 8) as one kernel on the matrix
// cores, with an online softmax: the n_kv x T score matrix never goes to memory.
//
// Replaces, for the node group build_attn_mha emits without flash attention (src/llama-graph.cpp:1283-1330):
//   MUL_MAT(k, q) [prec F32] -> SOFT_MAX(mask, scale) [+ sinks] -> MUL_MAT(v, kq) -> PERMUTE -> CONT
// with the unified KV cache's layouts (src/llama-kv-cache-unified.cpp:1056-1106): K rows [hd] per cell, V TRANSPOSED (v_trans: one
// row of cells per head dimension). Arithmetic as the node-by-node path: q and the probabilities rounded to f16 (the vec_dot type
// of an F16 matrix), f32 accumulation, soft_max in f32.
//
// One wave per (head, 32 queries); 8 waves (a GQA group x query tiles) share the K / V^T blocks through LDS (see the kernel). Everything is
// computed TRANSPOSED so that a lane owns ONE query column:
//   S^T[32 cells x 32 queries] = K[32 x hd] . Q^T      (v_mfma_f32_32x32x16_f16, A = K rows straight from the cache, B = Q^T)
//   per-lane online softmax over the 16 cells a lane holds (+ one exchange with lane ^ 32: the other 16 cells of the same query)
//   O^T[hd x 32 queries] += V^T[hd x 32 cells] . P^T   (A = rows of the transposed V cache, B = P^T straight from the registers of S^T)
// The C-layout of S^T (cell = (r & 3) + 8 (r >> 2) + 4 (lane >> 5)) IS a valid B-operand k-slot order as long as V^T's k-slots use
// the same cell permutation — a sum over cells does not care about their order — so the probabilities never leave the registers.
#include "dev_common.h"
#include "kv_types.h"
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
    float * dst; size_t dst_nb1;                        // [hd*n_head, T] (NULL: only the bf16 copy is wanted)
    uint16_t * y16; int kp16;                           // != NULL: the result also as bf16 rows of kp16 elements — the activation copy the wo mat-mul reads (mmq.hip)
    int n_kv, n_head, n_head_kv, T;
    float scale;
    float softcap, max_bias, m0, m1; int n_head_log2;      // logit_softcap (scale already divided by it) and ALiBi, as in decode_fused.hip
};

static __device__ __forceinline__ float xhalf(float v, int lane) {      // the value lane ^ 32 holds
    return __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute((lane ^ 32) << 2, __builtin_bit_cast(int, v)));
}

// VT: transposed V cache (rows over cells). !VT: V rows are cells (FLASH_ATTN_EXT): the 8 cells of a k-slot group are then 8 two-byte
// gathers per operand — correct, not fast; a transposing LDS read is the next step for that layout.
constexpr int APF_NW = 8;
constexpr int APF_KLD = 16;     // LDS row padding (bytes): K rows HD*2 + 16 (ds_read_b128 conflict-free), V^T rows 64 + 8 (ds_read_b64 conflict-free)
// One workgroup of 8 waves = hpw heads of ONE kv head (the GQA group, or a power-of-two part of it) x 8/hpw tiles of 32 queries. The waves
// walk the cell blocks together: a block's K rows and V^T rows are staged ONCE into LDS with coalesced loads (double-buffered, the next
// block's loads in flight during the arithmetic) and every wave takes its MFMA operands from there. Before this, every wave gathered its
// own K rows (2 KB apart) and V^T rows (n_ctx*2 bytes apart) 8-16 bytes at a time: 768 sector requests per block and wave, 42 us per
// layer at 512 tokens and 399 us at 2048 however the blocks were spread over waves.
// MASK: 0 = none, 1 = f32, 2 = f16 (a template parameter: the loads stay in one basic block). A wave skips the arithmetic of a block
// whose mask is -inf for all its 32 x 32 pairs (the causal future); staging and barriers go on.
// KS = 2 (few workgroups, i.e. short prompts: the cell loop is the critical path): an iteration stages TWO cell blocks, the waves of a
// (head, query tile) come in pairs — one per block — and the pair's states are merged through LDS at the end: half the iterations.
template <int HD, bool VT, int MASK, int KS>
__global__ void __launch_bounds__(64*APF_NW) k_attn_prefill(const attn_pf_args p, int hpw) {
    constexpr int NC = HD/16, NDT = HD/32, CPR = HD/8;                 // CPR: 16-byte chunks per K row
    constexpr int KROW = HD*2 + APF_KLD, VROW = 64 + 8, KBLK = 32*KROW, VBLK = HD*VROW;
    extern __shared__ __attribute__((aligned(16))) char lds_all[];     // K: [2][KS][KBLK] | V^T: [2][KS][VBLK]
    char * const ldsk_base = lds_all, * const ldsv_base = lds_all + 2*KS*KBLK;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, ql = lane & 31, hf = lane >> 5;
    const int h = blockIdx.y*hpw + wave % hpw, hk = h/(p.n_head/p.n_head_kv);
    const int ks = KS > 1 ? (wave/hpw) & (KS - 1) : 0;                  // which of the iteration's blocks this wave multiplies
    const int q0 = (blockIdx.x*(APF_NW/(hpw*KS)) + wave/(hpw*KS))*32;
    const bool active = q0 < p.T;                        // wave-uniform; an idle wave still stages and meets the barriers
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
    float m = -INFINITY, l = 0.0f;     // running maximum (common to both halves; base-2 domain) and this half's share of the denominator
    constexpr float LOG2E = 1.4426950408889634f;
    const float scale2 = p.scale*LOG2E;
    const float slope = p.max_bias > 0.0f ? (h < p.n_head_log2 ? powf(p.m0, (float)(h + 1)) : powf(p.m1, (float)(2*(h - p.n_head_log2) + 1))) : 1.0f;

    // staging roles. K (and a row-major V): thread -> (cell = tid / CPR, 16-byte chunk); transposed V: thread -> (dim row = tid / 4, 8 cells)
    const bool k_role = tid < 32*CPR, v_role = VT ? tid < HD*4 : k_role;
    const int kcell = tid / CPR, kch = tid % CPR, vrow = tid >> 2, vch = tid & 3;
    const char * kg = p.k + (size_t) hk*p.k_nb2 + (size_t) kch*16;
    const char * vg = VT ? p.v + (size_t) hk*p.v_nb2 + (size_t) min(vrow, HD - 1)*p.v_nb1 + (size_t) vch*16
                         : p.v + (size_t) hk*p.v_nb2 + (size_t) kch*16;
    const char * mrow = MASK ? p.mask + (size_t) t*p.m_nb1 : nullptr;

    int4v kst[KS], vst[KS];            // the next iteration's K / V pieces
    int4v mst[4];                      // the next block's mask values of this lane (f16: .xy used)
    auto fetch = [&](int kv0) {        // kv0 = first cell of the iteration (KS blocks)
#pragma unroll
        for (int b = 0; b < KS; b++) {
            const int kvb = min(kv0 + 32*b, p.n_kv - 32);
            if (k_role) kst[b] = *(const int4v *) (kg + (size_t)(kvb + min(kcell, 31))*p.k_nb1);
            if (v_role) vst[b] = VT ? *(const int4v *) (vg + (size_t) kvb*2) : *(const int4v *) (vg + (size_t)(kvb + min(kcell, 31))*p.v_nb1);
        }
        const int kvc = min(kv0 + 32*ks, p.n_kv - 32);
        if (MASK == 2) {
#pragma unroll
            for (int j = 0; j < 4; j++) { const int2v raw = ld_b64(mrow + (size_t)(kvc + 8*j + 4*hf)*2); mst[j].x = raw.x; mst[j].y = raw.y; }
        } else if (MASK == 1) {
#pragma unroll
            for (int j = 0; j < 4; j++) mst[j] = ld_b128(mrow + (size_t)(kvc + 8*j + 4*hf)*4);
        }
    };
    auto stage = [&](int buf) {
#pragma unroll
        for (int b = 0; b < KS; b++) {
            char * lk = ldsk_base + (buf*KS + b)*KBLK, * lv = ldsv_base + (buf*KS + b)*VBLK;
            if (k_role) *(int4v *) (lk + kcell*KROW + kch*16) = kst[b];
            if (v_role) {
                if (VT) {
                    *(int2v *) (lv + vrow*VROW + vch*16)     = int2v{ vst[b].x, vst[b].y };
                    *(int2v *) (lv + vrow*VROW + vch*16 + 8) = int2v{ vst[b].z, vst[b].w };
                } else {        // transpose on the way in: 8 dims of one cell -> 8 rows of V^T
                    const uint32_t w[4] = { (uint32_t) vst[b].x, (uint32_t) vst[b].y, (uint32_t) vst[b].z, (uint32_t) vst[b].w };
#pragma unroll
                    for (int e = 0; e < 8; e++) *(uint16_t *) (lv + (kch*8 + e)*VROW + kcell*2) = (uint16_t)(w[e >> 1] >> (16*(e & 1)));
                }
            }
        }
    };

    fetch(0);
    stage(0);
    float mk[16];
    auto take_mask = [&]() {
        if (MASK == 2) {
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const f16x4 hv = __builtin_bit_cast(f16x4, int2v{ mst[j].x, mst[j].y });
                mk[4*j] = (float) hv[0]; mk[4*j + 1] = (float) hv[1]; mk[4*j + 2] = (float) hv[2]; mk[4*j + 3] = (float) hv[3];
            }
        } else if (MASK == 1) {
#pragma unroll
            for (int j = 0; j < 4; j++) { const float4v fv = __builtin_bit_cast(float4v, mst[j]); mk[4*j] = fv.x; mk[4*j + 1] = fv.y; mk[4*j + 2] = fv.z; mk[4*j + 3] = fv.w; }
        } else {
#pragma unroll
            for (int r = 0; r < 16; r++) mk[r] = 0.0f;
        }
    };
    take_mask();
    __syncthreads();

    for (int kv0 = 0, buf = 0; kv0 < p.n_kv; kv0 += 32*KS, buf ^= 1) {
        fetch(kv0 + 32*KS);            // clamped past the end: staged into a buffer nobody reads
        bool need = active && kv0 + 32*ks < p.n_kv;
        if (MASK && need) {
            float mx = -INFINITY;
#pragma unroll
            for (int r = 0; r < 16; r++) mx = fmaxf(mx, mk[r]);
            need = __builtin_amdgcn_ballot_w64(mx != -INFINITY) != 0;       // wave-uniform
        }
        if (need) {
            // ---- S^T = K . Q^T ----
            f32x16 s;
#pragma unroll
            for (int r = 0; r < 16; r++) s[r] = 0.0f;
            const char * kl = ldsk_base + (buf*KS + ks)*KBLK + ql*KROW + hf*16;
#pragma unroll
            for (int c = 0; c < NC; c++) s = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, *(const int4v *) (kl + c*32)), qb[c], s, 0, 0, 0);
            // ---- scale + mask; s[r] belongs to cell kv0 + (r & 3) + 8 (r >> 2) + 4 hf of query t. The softmax runs in the base-2 domain
            //      (scores times log2 e, v_exp_f32 is 2^x): one instruction per exponential instead of expf's six ----
            float bm = -INFINITY;
            if (p.softcap != 0.0f || p.max_bias > 0.0f) {
#pragma unroll
                for (int r = 0; r < 16; r++) { float x = s[r]*p.scale; if (p.softcap != 0.0f) x = p.softcap*tanhf(x); s[r] = (x + slope*mk[r])*LOG2E; bm = fmaxf(bm, s[r]); }
            } else
#pragma unroll
            for (int r = 0; r < 16; r++) { s[r] = s[r]*scale2 + mk[r]*LOG2E; bm = fmaxf(bm, s[r]); }
            bm = fmaxf(bm, xhalf(bm, lane));
            const float m_new = fmaxf(m, bm);
            const float alpha = m == -INFINITY ? 0.0f : __builtin_amdgcn_exp2f(m - m_new);     // m_new == -inf only while every cell so far was masked: p = 0
            float psum = 0.0f;
            float pr[16];
#pragma unroll
            for (int r = 0; r < 16; r++) { pr[r] = s[r] == -INFINITY ? 0.0f : __builtin_amdgcn_exp2f(s[r] - m_new); psum += pr[r]; }
            l = l*alpha + psum;
            m = m_new;
            if (__builtin_amdgcn_ballot_w64(alpha != 1.0f) != 0) {      // the running maximum settles early: most blocks rescale nothing
#pragma unroll
                for (int d = 0; d < NDT; d++)
#pragma unroll
                    for (int r = 0; r < 16; r++) o[d][r] *= alpha;
            }
            // ---- O^T += V^T . P^T: the k-slots of chunk c2 are the cells kv0 + 16 c2 + 8 (s >> 2) + 4 hf + (s & 3), s = 0..7 ----
#pragma unroll
            for (int c2 = 0; c2 < 2; c2++) {
                const f16x8 pb = { (_Float16) pr[8*c2 + 0], (_Float16) pr[8*c2 + 1], (_Float16) pr[8*c2 + 2], (_Float16) pr[8*c2 + 3],
                                   (_Float16) pr[8*c2 + 4], (_Float16) pr[8*c2 + 5], (_Float16) pr[8*c2 + 6], (_Float16) pr[8*c2 + 7] };
#pragma unroll
                for (int d = 0; d < NDT; d++) {
                    const char * vp = ldsv_base + (buf*KS + ks)*VBLK + (32*d + ql)*VROW + (16*c2 + 4*hf)*2;
                    const int2v lo = *(const int2v *) vp, hi = *(const int2v *) (vp + 16);
                    o[d] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, int4v{ lo.x, lo.y, hi.x, hi.y }), pb, o[d], 0, 0, 0);
                }
            }
        }
        stage(buf ^ 1);                // last read before the previous barrier
        take_mask();
        __syncthreads();
    }
    if (KS > 1) {        // the waves of a (head, query tile) merge their states pairwise (KS = 4: 1 -> 0 and 3 -> 2, then 2 -> 0); the staging buffers are free: the loop's last barrier is behind us
        constexpr int PS = 16*NDT + 2;
        bool gone = false;
#pragma unroll
        for (int step = 1; step < KS; step <<= 1) {
            const bool give = !gone && (ks & (2*step - 1)) == step, take = !gone && (ks & (2*step - 1)) == 0;
            // the slot of the RECEIVING wave: (query tile of the workgroup, head, receiver's ks)
            const int rk = give ? ks - step : ks;
            float * part = (float *) lds_all + (size_t)((wave/(KS*hpw)*hpw + wave % hpw)*(KS/2) + rk/2)*PS*64 + lane;      // (receivers have even ks: KS/2 slots per (tile, head))
            if (give) {
#pragma unroll
                for (int d = 0; d < NDT; d++)
#pragma unroll
                    for (int r = 0; r < 16; r++) part[(16*d + r)*64] = o[d][r];
                part[(16*NDT)*64] = m; part[(16*NDT + 1)*64] = l;
                gone = true;
            }
            __syncthreads();
            if (take) {
                const float mw = part[(16*NDT)*64], lw = part[(16*NDT + 1)*64];
                const float m_new = fmaxf(m, mw);
                const float a = m == -INFINITY ? 0.0f : __builtin_amdgcn_exp2f(m - m_new), b = mw == -INFINITY ? 0.0f : __builtin_amdgcn_exp2f(mw - m_new);
                l = l*a + lw*b; m = m_new;
#pragma unroll
                for (int d = 0; d < NDT; d++)
#pragma unroll
                    for (int r = 0; r < 16; r++) o[d][r] = o[d][r]*a + part[(16*d + r)*64]*b;
            }
            if (step*2 < KS) __syncthreads();      // (the slots are written again in the next round)
        }
        if (gone) return;
    }
    if (!active) return;
    // ---- finish: both halves' denominators, the sink logit (src/llama-graph.cpp:1313), normalise, store ----
    float lt = l + xhalf(l, lane);
    float fin = 1.0f;
    if (p.sinks) {
        const float sk = p.sinks[h]*LOG2E;
        const float mf = fmaxf(m, sk);
        fin = m == -INFINITY ? 0.0f : __builtin_amdgcn_exp2f(m - mf);
        lt = lt*fin + __builtin_amdgcn_exp2f(sk - mf);
    }
    const float inv = fin/lt;
    if (q0 + ql < p.T) {
        float * orow = p.dst ? (float *) ((char *) p.dst + (size_t) t*p.dst_nb1) + (size_t) h*HD : nullptr;
        uint16_t * yrow = p.y16 ? p.y16 + (size_t) t*p.kp16 + (size_t) h*HD : nullptr;
#pragma unroll
        for (int d = 0; d < NDT; d++)
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const float4v ov = { o[d][4*j]*inv, o[d][4*j + 1]*inv, o[d][4*j + 2]*inv, o[d][4*j + 3]*inv };
                if (orow) *(float4v *) (orow + 32*d + 8*j + 4*hf) = ov;
                if (yrow) *(uint2 *) (yrow + 32*d + 8*j + 4*hf) = uint2{ pack_bf16(ov.x, ov.y), pack_bf16(ov.z, ov.w) };
            }
    }
}

bool attn_prefill_supported(int64_t head_dim, int64_t n_kv) { return (head_dim == 128 || head_dim == 64) && n_kv % 32 == 0 && n_kv > 0; }

void attn_prefill(const void * q, size_t q_nb1, size_t q_nb2, const void * k, size_t k_nb1, size_t k_nb2, const void * v, size_t v_nb1, size_t v_nb2,
                  const void * mask, size_t m_nb1, bool mask_f16, const float * sinks, float * dst, size_t dst_nb1,
                  int64_t head_dim, int64_t n_kv, int64_t n_head, int64_t n_head_kv, int64_t T, float scale, hipStream_t stream, bool v_trans, uint16_t * y16, const attn_extra * ex) {
    attn_pf_args a = { (const char *) q, q_nb1, q_nb2, (const char *) k, k_nb1, k_nb2, (const char *) v, v_nb1, v_nb2,
                       (const char *) mask, m_nb1, mask_f16 ? 1 : 0, sinks, dst, dst_nb1, y16, (int)((head_dim*n_head + 63) & ~(int64_t) 63), (int) n_kv, (int) n_head, (int) n_head_kv, (int) T, scale,
                       0.0f, 0.0f, 1.0f, 1.0f, 1 };
    if (ex) {
        attn_alibi(ex->max_bias, n_head, a.m0, a.m1, a.n_head_log2);
        a.softcap = ex->softcap; a.max_bias = ex->max_bias;
        if (ex->softcap != 0.0f) a.scale = scale/ex->softcap;
    }
    const int64_t R = n_head/n_head_kv;
    int hpw = R % 8 == 0 ? 8 : R % 4 == 0 ? 4 : R % 2 == 0 ? 2 : 1;          // heads of one kv head per workgroup
    // few workgroups (short prompts): the waves of a (head, query tile) split the cell blocks of an iteration two or four ways, shortening the loop that is the critical path
    const int64_t wgs1 = ((T + 32*(APF_NW/hpw) - 1)/(32*(APF_NW/hpw)))*(n_head/hpw);
    int ksp = (hpw <= 4 && wgs1 < 160 && n_kv >= 128) ? 2 : 1;
    static const bool ks4_on = !getenv("GGML_MI355X_ATTN_PF_KS4") || atoi(getenv("GGML_MI355X_ATTN_PF_KS4")) != 0;
    if (ks4_on && ksp == 2 && R % 2 == 0 && n_kv >= 256 && head_dim == 128 && v_trans) {
        // ... four ways with two heads per workgroup when even the pairs leave half the chip idle (pp512 of a 32-head model: 128 workgroups -> 256)
        const int64_t wgs2 = ((T + 32*(APF_NW/(hpw*2)) - 1)/(32*(APF_NW/(hpw*2))))*(n_head/hpw);
        if (wgs2 < 200) { hpw = 2; ksp = 4; }
    }
    const int qpw = APF_NW/(hpw*ksp);                                                // query tiles per workgroup
    const dim3 grid((unsigned)((T + 32*qpw - 1)/(32*qpw)), (unsigned)(n_head/hpw));
    const int mk = !mask ? 0 : (mask_f16 ? 2 : 1);
#define MI_APF2(HD_, VT_, MK_, KS_) do { \
        constexpr size_t lds_ = (size_t) 2*KS_*(32*(HD_*2 + APF_KLD) + HD_*(64 + 8)); \
        MI_LDS_LIMIT_OR_DIE(lds_, k_attn_prefill<HD_, VT_, MK_, KS_>); \
        hipLaunchKernelGGL((k_attn_prefill<HD_, VT_, MK_, KS_>), grid, dim3(64*APF_NW), lds_, stream, a, hpw); } while (0)
#define MI_APF1(HD_, VT_, MK_) do { if (ksp == 4) { if constexpr (HD_ == 128 && VT_) MI_APF2(HD_, VT_, MK_, 4); } else if (ksp == 2) MI_APF2(HD_, VT_, MK_, 2); else MI_APF2(HD_, VT_, MK_, 1); } while (0)
#define MI_APF(HD_, VT_) do { if (mk == 0) MI_APF1(HD_, VT_, 0); else if (mk == 1) MI_APF1(HD_, VT_, 1); else MI_APF1(HD_, VT_, 2); } while (0)
    if (!v_trans) { if (head_dim == 128) MI_APF(128, false); else MI_APF(64, false); }
    else          { if (head_dim == 128) MI_APF(128, true);  else MI_APF(64, true); }
#undef MI_APF
#undef MI_APF1
#undef MI_APF2
}

// ---- the cache as the kernels above want it: f16, cells as rows or transposed (kv_types.h has the element types) ----
struct kvc_args { const char * src; size_t nb1, nb2; int hd, n_kv, n_head_kv; uint16_t * dst; };
template <int TY>
__global__ void __launch_bounds__(256) k_kv_rows_f16(const kvc_args p) {
    const int c8 = p.hd >> 3;
    const long long e = (long long) blockIdx.x*256 + threadIdx.x, tot = (long long) p.n_head_kv*p.n_kv*c8;
    if (e >= tot) return;
    const int sub = (int)(e % c8); const long long cj = e / c8; const int j = (int)(cj % p.n_kv), hk = (int)(cj / p.n_kv);
    float f[8];
    kv_cvt8<TY>(kv_raw8<TY>(p.src + (size_t) hk*p.nb2 + (size_t) j*p.nb1, sub), sub, f);
    uint32_t w[4];
#pragma unroll
    for (int i = 0; i < 4; i++) w[i] = (uint32_t) f32_to_f16_bits(f[2*i]) | ((uint32_t) f32_to_f16_bits(f[2*i + 1]) << 16);
    *(int4v *) (p.dst + ((size_t) hk*p.n_kv + j)*p.hd + sub*8) = int4v{ (int) w[0], (int) w[1], (int) w[2], (int) w[3] };
}
// one workgroup = 64 cells of one kv head: cells are read as rows (coalesced), turned in LDS, written as 128-byte pieces of the rows over cells
template <int TY>
__global__ void __launch_bounds__(256) k_kv_trans_f16(const kvc_args p) {
    constexpr int LROW = 64 + 8;       // halves per LDS row (16-byte aligned rows)
    __shared__ __attribute__((aligned(16))) uint16_t tile[128*LROW];
    const int hk = blockIdx.y, j0 = blockIdx.x*64, c8 = p.hd >> 3;
    for (int e = threadIdx.x; e < 64*c8; e += 256) {
        const int cell = e / c8, sub = e % c8, j = min(j0 + cell, p.n_kv - 1);
        float f[8];
        kv_cvt8<TY>(kv_raw8<TY>(p.src + (size_t) hk*p.nb2 + (size_t) j*p.nb1, sub), sub, f);
#pragma unroll
        for (int i = 0; i < 8; i++) tile[(sub*8 + i)*LROW + cell] = f32_to_f16_bits(f[i]);
    }
    __syncthreads();
    for (int e = threadIdx.x; e < p.hd*8; e += 256) {
        const int d = e >> 3, g8 = e & 7;
        if (j0 + g8*8 < p.n_kv) *(int4v *) (p.dst + ((size_t) hk*p.hd + d)*p.n_kv + j0 + g8*8) = *(const int4v *) (tile + d*LROW + g8*8);
    }
}
void kv_to_f16(int type, const void * src, size_t nb1, size_t nb2, int64_t hd, int64_t n_kv, int64_t n_head_kv, uint16_t * dst, bool transpose, hipStream_t stream) {
    if ((hd != 64 && hd != 128) || n_kv % 8 || n_kv < 8) { fprintf(stderr, "kv_to_f16: unsupported shape (hd %lld, n_kv %lld)\n", (long long) hd, (long long) n_kv); abort(); }
    const kvc_args a = { (const char *) src, nb1, nb2, (int) hd, (int) n_kv, (int) n_head_kv, dst };
    const long long tot = (long long) n_head_kv*n_kv*(hd/8);
#define MI_KVC(TY_) do { if (transpose) hipLaunchKernelGGL((k_kv_trans_f16<TY_>), dim3((unsigned)((n_kv + 63)/64), (unsigned) n_head_kv), dim3(256), 0, stream, a); \
                         else hipLaunchKernelGGL((k_kv_rows_f16<TY_>), dim3((unsigned)((tot + 255)/256)), dim3(256), 0, stream, a); } while (0)
    if (type == T_F16) MI_KVC(T_F16); else if (type == T_BF16) MI_KVC(T_BF16); else if (type == T_Q8_0) MI_KVC(T_Q8_0); else if (type == T_Q4_0) MI_KVC(T_Q4_0);
    else { fprintf(stderr, "kv_to_f16: cache type %d is not supported\n", type); abort(); }
#undef MI_KVC
}

} // namespace mi355x
