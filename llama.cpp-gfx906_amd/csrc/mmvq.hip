// mmvq.hip — quantized mat-vec for the decode step (n <= 8 activation columns), gfx950.
//
// Replaces, for this path, the CPU backend's {quantize src1 row; vec_dot per (row, col)} loop
//  for GGML_OP_MUL_MAT nodes emitted by build_lora_mm (src/llama-graph.cpp:543-567)
// and, in its _id form, GGML_OP_MUL_MAT_ID from build_lora_mm_id (src/llama-graph.cpp:569-595).
// The integer arithmetic restates ggml's generic vec_dot_*_q8_* (oracle/ggml_oracle.c); the only
// difference from the CPU result is the order of the final f32 additions.
//
// Roofline: HBM. Algorithmic bytes per launch = m * (k/blck) * type_size (weights, read once)
// + n*k*~1.1 (int8 activations, L2-resident) + m*n*4 (dst).
//
// Mapping (wave64, no cross-wave communication):
//   * a workgroup = 4 waves; each wave owns R consecutive weight rows and walks the whole K.
//   * inside a row, LPB lanes cooperate on one block so that every lane issues ONE 16-byte (Q4_K,
//     Q5_K, Q8_0, Q4_0, MXFP4) or 3 x 8-byte (Q6_K) load of packed quants per row per step and a
//     wave-wide load instruction covers 64/LPB consecutive blocks (>= 1 KiB of contiguous HBM).
//     Loads go straight to VGPRs (cdna_hip_programming.md "GEMV / M <= 16 decode weights": LDS
//     round trip is pure overhead), nontemporal where the address is 16-byte aligned.
//   * the int8 activation vector (shared by every row) is staged ONCE per workgroup into LDS for
//     n == 1 (k + k/16 bytes), or read through L1/L2 for 2 <= n <= 8.
//   * per-lane partial sums are reduced with DPP row operations + v_readlane (dev_common.h).
#include "blocks.h"
#include "dev_common.h"
#include "kernels.h"

namespace mi355x {

// activation view handed to the fragment loaders (pointers may be LDS or global; everything is
// force-inlined so the address space is resolved at compile time)
struct act_view {
    const int8_t  * qs;     // [k]
    const float   * d;      // [k/32] or [k/256]
    const int16_t * bs;     // [k/32] or [k/16]
};

static __device__ __forceinline__ int4v lds_or_global_b128(const int8_t * p) { return *(const int4v *) p; }
static __device__ __forceinline__ int2v lds_or_global_b64 (const int8_t * p) { return *(const int2v *) p; }

static __device__ __forceinline__ int dot16(const int4v & w, const int4v & a) {
    return dot4(w.x, a.x, dot4(w.y, a.y, dot4(w.z, a.z, dot4(w.w, a.w, 0))));
}

// 6-bit scale/min pair for sub-blocks (2g, 2g+1) of a K-quant header — gguf-py/gguf/quants.py:479-501.
// s0,s1,s2 = the 12 scale bytes as 3 little-endian dwords. Returns sc packed as (sc[2g] | sc[2g+1]<<8), same for m.
static __device__ __forceinline__ void k4_scales(uint32_t s0, uint32_t s1, uint32_t s2, int g, uint32_t & sc2, uint32_t & m2) {
    const int sh = (g & 1)*16;
    const uint32_t a0 = (s0 >> sh) & 0xFFFF, a1 = (s1 >> sh) & 0xFFFF, a2 = (s2 >> sh) & 0xFFFF;
    if (g < 2) {
        sc2 = a0 & 0x3F3F;
        m2  = a1 & 0x3F3F;
    } else {
        sc2 = (a2 & 0x0F0F)        | ((a0 >> 2) & 0x3030);
        m2  = ((a2 >> 4) & 0x0F0F) | ((a1 >> 2) & 0x3030);
    }
}

// ------------------------------------------------------------------------------------------------
// per-type fragments. LPB = lanes per block; `slot` = lane % LPB; `ib` = block index inside the row
// ------------------------------------------------------------------------------------------------
template <int TYPE> struct mmvq_t;

// ---- Q4_K ---------------------------------------------------------------------------------------
template <> struct mmvq_t<T_Q4_K> {
    static constexpr int LPB = 8, BLOCK_BYTES = 144, QK = 256, ACT = T_Q8_K;
    struct afrag { int4v lo, hi; float d8; int bs_lo, bs_hi; };
    struct wfrag { int4v hdr, qs; };
    static __device__ __forceinline__ afrag load_a(const act_view & a, int64_t ib, int slot) {
        const int g = slot >> 1, h = slot & 1;
        afrag f;
        const int8_t * p = a.qs + ib*256 + 64*g + 16*h;
        f.lo = lds_or_global_b128(p);
        f.hi = lds_or_global_b128(p + 32);
        f.d8 = a.d[ib];
        f.bs_lo = a.bs[ib*16 + 4*g + h];
        f.bs_hi = a.bs[ib*16 + 4*g + 2 + h];
        return f;
    }
    static __device__ __forceinline__ wfrag load_w(const char * row, int64_t ib, int slot) {
        const char * b = row + ib*BLOCK_BYTES;
        wfrag w;
        w.hdr = *(const int4v *) b;                       // d, dmin, 12 scale bytes: shared by the 8 lanes (one 16-byte line)
        w.qs  = ld_b128_nt(b + 16 + 16*slot);
        return w;
    }
    static __device__ __forceinline__ float dot(const wfrag & w, const afrag & a, int slot) {
        const int g = slot >> 1;
        uint32_t sc2, m2;
        k4_scales((uint32_t) w.hdr.y, (uint32_t) w.hdr.z, (uint32_t) w.hdr.w, g, sc2, m2);
        int4v lo, hi;
        lo.x = w.qs.x & 0x0F0F0F0F; hi.x = (w.qs.x >> 4) & 0x0F0F0F0F;
        lo.y = w.qs.y & 0x0F0F0F0F; hi.y = (w.qs.y >> 4) & 0x0F0F0F0F;
        lo.z = w.qs.z & 0x0F0F0F0F; hi.z = (w.qs.z >> 4) & 0x0F0F0F0F;
        lo.w = w.qs.w & 0x0F0F0F0F; hi.w = (w.qs.w >> 4) & 0x0F0F0F0F;
        const int isum = (int)(sc2 & 0xFF)*dot16(lo, a.lo) + (int)(sc2 >> 8)*dot16(hi, a.hi);
        const int msum = (int)(m2  & 0xFF)*a.bs_lo        + (int)(m2  >> 8)*a.bs_hi;
        const float d    = f16_bits_to_f32((uint16_t)((uint32_t) w.hdr.x & 0xFFFF));
        const float dmin = f16_bits_to_f32((uint16_t)((uint32_t) w.hdr.x >> 16));
        return (d*a.d8)*(float) isum - (dmin*a.d8)*(float) msum;
    }
};

// ---- Q5_K ---------------------------------------------------------------------------------------
template <> struct mmvq_t<T_Q5_K> {
    static constexpr int LPB = 8, BLOCK_BYTES = 176, QK = 256, ACT = T_Q8_K;
    typedef mmvq_t<T_Q4_K>::afrag afrag;
    struct wfrag { int4v hdr, qh, qs; };
    static __device__ __forceinline__ afrag load_a(const act_view & a, int64_t ib, int slot) { return mmvq_t<T_Q4_K>::load_a(a, ib, slot); }
    static __device__ __forceinline__ wfrag load_w(const char * row, int64_t ib, int slot) {
        const char * b = row + ib*BLOCK_BYTES;
        wfrag w;
        w.hdr = *(const int4v *) b;
        w.qh  = *(const int4v *) (b + 16 + 16*(slot & 1));  // high bits for byte positions 16h..16h+15, all 8 sub-blocks
        w.qs  = ld_b128_nt(b + 48 + 16*slot);
        return w;
    }
    static __device__ __forceinline__ float dot(const wfrag & w, const afrag & a, int slot) {
        const int g = slot >> 1;
        uint32_t sc2, m2;
        k4_scales((uint32_t) w.hdr.y, (uint32_t) w.hdr.z, (uint32_t) w.hdr.w, g, sc2, m2);
        const int b0 = 2*g, b1 = 2*g + 1;   // bit of qh holding the 5th bit of sub-blocks 2g / 2g+1
        int4v lo, hi;
#define MI_Q5(c) \
        lo.c = (w.qs.c & 0x0F0F0F0F)        | ((((uint32_t) w.qh.c >> b0) & 0x01010101) << 4); \
        hi.c = ((w.qs.c >> 4) & 0x0F0F0F0F) | ((((uint32_t) w.qh.c >> b1) & 0x01010101) << 4);
        MI_Q5(x) MI_Q5(y) MI_Q5(z) MI_Q5(w)
#undef MI_Q5
        const int isum = (int)(sc2 & 0xFF)*dot16(lo, a.lo) + (int)(sc2 >> 8)*dot16(hi, a.hi);
        const int msum = (int)(m2  & 0xFF)*a.bs_lo        + (int)(m2  >> 8)*a.bs_hi;
        const float d    = f16_bits_to_f32((uint16_t)((uint32_t) w.hdr.x & 0xFFFF));
        const float dmin = f16_bits_to_f32((uint16_t)((uint32_t) w.hdr.x >> 16));
        return (d*a.d8)*(float) isum - (dmin*a.d8)*(float) msum;
    }
};

// ---- Q6_K ---------------------------------------------------------------------------------------
// lane slot j: half n = j>>2 (128 elements each), l0 = 8*(j&3): the lane owns l = l0..l0+7 of that half, i.e.
// elements 128n + {0,32,64,96} + l (quants.py:554-572): ql[64n+l] lo/hi nibble, ql[64n+32+l] lo/hi nibble,
// qh[32n+l] 2-bit fields. 24 bytes of quants per lane.
template <> struct mmvq_t<T_Q6_K> {
    static constexpr int LPB = 8, BLOCK_BYTES = 210, QK = 256, ACT = T_Q8_K;
    struct afrag { int2v a[4]; int s[4]; float d8; };   // s[i] = sum of the 8 int8 of a[i] (for the -32 offset)
    struct wfrag { int2v qla, qlb, qh; int2v sc; uint32_t d; };
    static __device__ __forceinline__ afrag load_a(const act_view & a, int64_t ib, int slot) {
        const int n = slot >> 2, l0 = 8*(slot & 3);
        afrag f;
        const int8_t * p = a.qs + ib*256 + 128*n + l0;
#pragma unroll
        for (int i = 0; i < 4; i++) {
            f.a[i] = lds_or_global_b64(p + 32*i);
            f.s[i] = dot4(0x01010101, f.a[i].x, dot4(0x01010101, f.a[i].y, 0));
        }
        f.d8 = a.d[ib];
        return f;
    }
    static __device__ __forceinline__ wfrag load_w(const char * row, int64_t ib, int slot) {
        const int n = slot >> 2, l0 = 8*(slot & 3);
        const char * b = row + ib*BLOCK_BYTES;   // only 2-byte aligned: unaligned-mode global loads
        wfrag w;
        w.qla = ld_b64(b + 64*n + l0);
        w.qlb = ld_b64(b + 64*n + 32 + l0);
        w.qh  = ld_b64(b + 128 + 32*n + l0);
        w.sc  = ld_b64(b + 192 + 8*n);           // scales[8n .. 8n+7]
        w.d   = ld_u16(b + 208);
        return w;
    }
    static __device__ __forceinline__ float dot(const wfrag & w, const afrag & a, int slot) {
        const int is = (slot & 3) >> 1;          // l0/16
        // the four scales this lane needs are bytes is, is+2, is+4, is+6 of sc
        const uint32_t sx = (uint32_t) w.sc.x >> (8*is), sy = (uint32_t) w.sc.y >> (8*is);
        const int sc0 = (int8_t)(sx & 0xFF), sc1 = (int8_t)((sx >> 16) & 0xFF), sc2 = (int8_t)(sy & 0xFF), sc3 = (int8_t)((sy >> 16) & 0xFF);
        int acc[4];
#define MI_Q6(c, A0, A1, A2, A3) { \
        const uint32_t qa = (uint32_t) w.qla.c, qb = (uint32_t) w.qlb.c, qh = (uint32_t) w.qh.c; \
        const int v0 = (int)((qa & 0x0F0F0F0F)        | ((qh << 4) & 0x30303030)); \
        const int v1 = (int)((qb & 0x0F0F0F0F)        | ((qh << 2) & 0x30303030)); \
        const int v2 = (int)(((qa >> 4) & 0x0F0F0F0F) | ( qh       & 0x30303030)); \
        const int v3 = (int)(((qb >> 4) & 0x0F0F0F0F) | ((qh >> 2) & 0x30303030)); \
        A0 = dot4(v0, a.a[0].c, A0); A1 = dot4(v1, a.a[1].c, A1); A2 = dot4(v2, a.a[2].c, A2); A3 = dot4(v3, a.a[3].c, A3); }
        acc[0] = acc[1] = acc[2] = acc[3] = 0;
        MI_Q6(x, acc[0], acc[1], acc[2], acc[3])
        MI_Q6(y, acc[0], acc[1], acc[2], acc[3])
#undef MI_Q6
        // sum (q-32)*a = sum q*a - 32*sum a
        const int isum = sc0*(acc[0] - 32*a.s[0]) + sc1*(acc[1] - 32*a.s[1]) + sc2*(acc[2] - 32*a.s[2]) + sc3*(acc[3] - 32*a.s[3]);
        return (f16_bits_to_f32((uint16_t) w.d)*a.d8)*(float) isum;
    }
};

// ---- Q8_0 ---------------------------------------------------------------------------------------
template <> struct mmvq_t<T_Q8_0> {
    static constexpr int LPB = 2, BLOCK_BYTES = 34, QK = 32, ACT = T_Q8_0;
    struct afrag { int4v a; float d8; };
    struct wfrag { int4v qs; uint32_t d; };
    static __device__ __forceinline__ afrag load_a(const act_view & a, int64_t ib, int slot) {
        afrag f;
        f.a = lds_or_global_b128(a.qs + ib*32 + 16*slot);
        f.d8 = a.d[ib];
        return f;
    }
    static __device__ __forceinline__ wfrag load_w(const char * row, int64_t ib, int slot) {
        const char * b = row + ib*BLOCK_BYTES;
        wfrag w;
        w.qs = ld_b128(b + 2 + 16*slot);
        w.d  = ld_u16(b);
        return w;
    }
    static __device__ __forceinline__ float dot(const wfrag & w, const afrag & a, int) {
        return (float) dot16(w.qs, a.a)*(f16_bits_to_f32((uint16_t) w.d)*a.d8);
    }
};

// ---- Q4_0 ---------------------------------------------------------------------------------------
template <> struct mmvq_t<T_Q4_0> {
    static constexpr int LPB = 1, BLOCK_BYTES = 18, QK = 32, ACT = T_Q8_0;
    struct afrag { int4v lo, hi; float d8; int bs; };
    struct wfrag { int4v qs; uint32_t d; };
    static __device__ __forceinline__ afrag load_a(const act_view & a, int64_t ib, int) {
        afrag f;
        f.lo = lds_or_global_b128(a.qs + ib*32);
        f.hi = lds_or_global_b128(a.qs + ib*32 + 16);
        f.d8 = a.d[ib];
        f.bs = a.bs[ib];
        return f;
    }
    static __device__ __forceinline__ wfrag load_w(const char * row, int64_t ib, int) {
        const char * b = row + ib*BLOCK_BYTES;
        wfrag w;
        w.qs = ld_b128(b + 2);
        w.d  = ld_u16(b);
        return w;
    }
    static __device__ __forceinline__ float dot(const wfrag & w, const afrag & a, int) {
        int4v lo, hi;
        lo.x = w.qs.x & 0x0F0F0F0F; hi.x = (w.qs.x >> 4) & 0x0F0F0F0F;
        lo.y = w.qs.y & 0x0F0F0F0F; hi.y = (w.qs.y >> 4) & 0x0F0F0F0F;
        lo.z = w.qs.z & 0x0F0F0F0F; hi.z = (w.qs.z >> 4) & 0x0F0F0F0F;
        lo.w = w.qs.w & 0x0F0F0F0F; hi.w = (w.qs.w >> 4) & 0x0F0F0F0F;
        const int sumi = dot16(lo, a.lo) + dot16(hi, a.hi) - 8*a.bs;      // sum (q-8)*a
        return ((float) sumi*f16_bits_to_f32((uint16_t) w.d))*a.d8;
    }
};

// ---- MXFP4 --------------------------------------------------------------------------------------
// 16-entry int8 lookup (quants.py:659) for 4 packed 4-bit indices with two v_perm_b32 + a bit-select
static __device__ __forceinline__ int mxfp4_lut4(uint32_t idx) {
    // kvalues = 0,1,2,3,4,6,8,12 | 0,-1,-2,-3,-4,-6,-8,-12
    const uint32_t pos_lo = 0x03020100u, pos_hi = 0x0C080604u, neg_lo = 0xFDFEFF00u, neg_hi = 0xF4F8FAFCu;
    const uint32_t sel = idx & 0x07070707u;
    const uint32_t p = __builtin_amdgcn_perm(pos_hi, pos_lo, sel);
    const uint32_t n = __builtin_amdgcn_perm(neg_hi, neg_lo, sel);
    const uint32_t m = ((idx >> 3) & 0x01010101u)*0xFFu;   // 0xFF in bytes whose index has bit 3 set
    return (int)((p & ~m) | (n & m));
}

template <> struct mmvq_t<T_MXFP4> {
    static constexpr int LPB = 1, BLOCK_BYTES = 17, QK = 32, ACT = T_Q8_0;
    struct afrag { int4v lo, hi; float d8; };
    struct wfrag { int4v qs; uint32_t e; };
    static __device__ __forceinline__ afrag load_a(const act_view & a, int64_t ib, int) {
        afrag f;
        f.lo = lds_or_global_b128(a.qs + ib*32);
        f.hi = lds_or_global_b128(a.qs + ib*32 + 16);
        f.d8 = a.d[ib];
        return f;
    }
    static __device__ __forceinline__ wfrag load_w(const char * row, int64_t ib, int) {
        const char * b = row + ib*BLOCK_BYTES;
        wfrag w;
        w.qs = ld_b128(b + 1);
        w.e  = *(const uint8_t *) b;
        return w;
    }
    static __device__ __forceinline__ float dot(const wfrag & w, const afrag & a, int) {
        int4v lo, hi;
        lo.x = mxfp4_lut4((uint32_t) w.qs.x); hi.x = mxfp4_lut4((uint32_t) w.qs.x >> 4);
        lo.y = mxfp4_lut4((uint32_t) w.qs.y); hi.y = mxfp4_lut4((uint32_t) w.qs.y >> 4);
        lo.z = mxfp4_lut4((uint32_t) w.qs.z); hi.z = mxfp4_lut4((uint32_t) w.qs.z >> 4);
        lo.w = mxfp4_lut4((uint32_t) w.qs.w); hi.w = mxfp4_lut4((uint32_t) w.qs.w >> 4);
        const int sumi = dot16(lo, a.lo) + dot16(hi, a.hi);
        return (a.d8*e8m0_to_f32_half(w.e))*(float) sumi;
    }
};

// ------------------------------------------------------------------------------------------------
// the kernel
// ------------------------------------------------------------------------------------------------
struct mmvq_args {
    const char * W; size_t w_row_stride; size_t w_expert_stride;
    int64_t m, k;
    const int8_t * a_qs; const float * a_d; const int16_t * a_bs;   // [n][...]
    float * dst; size_t dst_col_stride;                              // bytes
    // MUL_MAT_ID form (ids != nullptr): blockIdx.y = pair p = t*n_used + u
    const char * ids; size_t ids_nb0, ids_nb1; int n_used; int n_b; size_t dst_nb1, dst_nb2;
};

template <int TYPE, int NCOLS, int R, bool LDS_ACT, bool IDS>
__global__ void __launch_bounds__(256) k_mmvq(const mmvq_args p) {
    typedef mmvq_t<TYPE> T;
    constexpr int LPB = T::LPB;
    constexpr int BPW = 64/LPB;                         // blocks per wave step
    constexpr int ND  = T::ACT == T_Q8_0 ? 32 : 256;    // elements per activation scale
    constexpr int NBS = T::ACT == T_Q8_0 ? 32 : 16;     // elements per activation bsum

    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int slot = lane % LPB;
    const int64_t nb = p.k / T::QK;
    const int64_t row0 = ((int64_t) blockIdx.x*4 + wave)*R;

    const char * W = p.W;
    const int8_t * a_qs = p.a_qs; const float * a_d = p.a_d; const int16_t * a_bs = p.a_bs;
    float * dst = p.dst;
    if (IDS) {
        const int pr = blockIdx.y, t = pr / p.n_used, u = pr % p.n_used;
        const int e = *(const int32_t *) (p.ids + (size_t) t*p.ids_nb1 + (size_t) u*p.ids_nb0);
        W += (size_t) e*p.w_expert_stride;
        const int64_t arow = (int64_t) t*p.n_b + (u % p.n_b);
        a_qs += arow*p.k; a_d += arow*(p.k/ND); a_bs += arow*(p.k/NBS);
        dst = (float *) ((char *) dst + (size_t) t*p.dst_nb2 + (size_t) u*p.dst_nb1);
    }

    act_view av[NCOLS];
    if (LDS_ACT) {
        // stage the (single) activation column: qs | d | bsums, each padded to 16 bytes
        static_assert(!LDS_ACT || NCOLS == 1, "LDS staging is for n == 1");
        const int64_t qs_b = (p.k + 15) & ~15, d_b = ((p.k/ND)*4 + 15) & ~15, bs_b = ((p.k/NBS)*2 + 15) & ~15;
        for (int64_t i = threadIdx.x*16; i < qs_b; i += 256*16) *(int4v *) (smem + i) = *(const int4v *) ((const char *) a_qs + i);
        for (int64_t i = threadIdx.x*16; i < d_b;  i += 256*16) *(int4v *) (smem + qs_b + i) = *(const int4v *) ((const char *) a_d + i);
        for (int64_t i = threadIdx.x*16; i < bs_b; i += 256*16) *(int4v *) (smem + qs_b + d_b + i) = *(const int4v *) ((const char *) a_bs + i);
        __syncthreads();
        av[0].qs = (const int8_t *) smem; av[0].d = (const float *) (smem + qs_b); av[0].bs = (const int16_t *) (smem + qs_b + d_b);
    } else {
#pragma unroll
        for (int c = 0; c < NCOLS; c++) { av[c].qs = a_qs + c*p.k; av[c].d = a_d + c*(p.k/ND); av[c].bs = a_bs + c*(p.k/NBS); }
    }

    if (row0 >= p.m) return;   // wave-uniform; after the barrier

    float acc[NCOLS][R];
#pragma unroll
    for (int c = 0; c < NCOLS; c++)
#pragma unroll
        for (int r = 0; r < R; r++) acc[c][r] = 0.0f;

    const char * rows[R];
#pragma unroll
    for (int r = 0; r < R; r++) rows[r] = W + (size_t) min(row0 + r, p.m - 1)*p.w_row_stride;   // clamp: tail rows recompute the last row, never stored

    for (int64_t ib0 = 0; ib0 < nb; ib0 += BPW) {
        const int64_t ib = ib0 + lane/LPB;
        if (ib < nb) {
            typename T::wfrag w[R];
#pragma unroll
            for (int r = 0; r < R; r++) w[r] = T::load_w(rows[r], ib, slot);
#pragma unroll
            for (int c = 0; c < NCOLS; c++) {
                const typename T::afrag a = T::load_a(av[c], ib, slot);
#pragma unroll
                for (int r = 0; r < R; r++) acc[c][r] += T::dot(w[r], a, slot);
            }
        }
    }

#pragma unroll
    for (int c = 0; c < NCOLS; c++) {
#pragma unroll
        for (int r = 0; r < R; r++) {
            const float s = wave_sum(acc[c][r]);
            if (lane == 0 && row0 + r < p.m) {
                *(float *) ((char *) dst + (size_t) c*p.dst_col_stride + (size_t)(row0 + r)*4) = s;
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------------
bool mul_mat_vec_q_supported(int type_a) {
    switch (type_a) {
        case T_Q4_0: case T_Q8_0: case T_Q4_K: case T_Q5_K: case T_Q6_K: case T_MXFP4: return true;
        default: return false;
    }
}

static size_t lds_bytes_for(int act_kind, int64_t k) {
    const int nd = act_kind == T_Q8_0 ? 32 : 256, nbs = act_kind == T_Q8_0 ? 32 : 16;
    return ((k + 15) & ~15) + (((k/nd)*4 + 15) & ~15) + (((k/nbs)*2 + 15) & ~15);
}

template <int TYPE, int NCOLS, bool IDS>
static void launch_mmvq_n(const mmvq_args & a, int act_kind, int64_t n_pairs, hipStream_t stream) {
    constexpr int R = 2;
    const dim3 grid((unsigned)((a.m + 4*R - 1)/(4*R)), (unsigned) n_pairs);
    if (NCOLS == 1) {
        const size_t lds = lds_bytes_for(act_kind, a.k);
        if (lds <= 64*1024) {
            hipLaunchKernelGGL((k_mmvq<TYPE, 1, R, true, IDS>), grid, dim3(256), lds, stream, a);
            return;
        }
    }
    hipLaunchKernelGGL((k_mmvq<TYPE, NCOLS, R, false, IDS>), grid, dim3(256), 0, stream, a);
}

template <int TYPE, bool IDS>
static void launch_mmvq_t(const mmvq_args & a, int act_kind, int64_t n, int64_t n_pairs, hipStream_t stream) {
    switch (n) {
        case 1: launch_mmvq_n<TYPE, 1, IDS>(a, act_kind, n_pairs, stream); break;
        case 2: launch_mmvq_n<TYPE, 2, IDS>(a, act_kind, n_pairs, stream); break;
        case 3: launch_mmvq_n<TYPE, 3, IDS>(a, act_kind, n_pairs, stream); break;
        case 4: launch_mmvq_n<TYPE, 4, IDS>(a, act_kind, n_pairs, stream); break;
        case 5: launch_mmvq_n<TYPE, 5, IDS>(a, act_kind, n_pairs, stream); break;
        case 6: launch_mmvq_n<TYPE, 6, IDS>(a, act_kind, n_pairs, stream); break;
        case 7: launch_mmvq_n<TYPE, 7, IDS>(a, act_kind, n_pairs, stream); break;
        case 8: launch_mmvq_n<TYPE, 8, IDS>(a, act_kind, n_pairs, stream); break;
        default: fprintf(stderr, "mmvq: n=%lld out of range\n", (long long) n); abort();
    }
}

template <bool IDS>
static void launch_mmvq(int type_a, const mmvq_args & a, int act_kind, int64_t n, int64_t n_pairs, hipStream_t stream) {
    switch (type_a) {
        case T_Q4_0:  launch_mmvq_t<T_Q4_0,  IDS>(a, act_kind, n, n_pairs, stream); break;
        case T_Q8_0:  launch_mmvq_t<T_Q8_0,  IDS>(a, act_kind, n, n_pairs, stream); break;
        case T_Q4_K:  launch_mmvq_t<T_Q4_K,  IDS>(a, act_kind, n, n_pairs, stream); break;
        case T_Q5_K:  launch_mmvq_t<T_Q5_K,  IDS>(a, act_kind, n, n_pairs, stream); break;
        case T_Q6_K:  launch_mmvq_t<T_Q6_K,  IDS>(a, act_kind, n, n_pairs, stream); break;
        case T_MXFP4: launch_mmvq_t<T_MXFP4, IDS>(a, act_kind, n, n_pairs, stream); break;
        default: fprintf(stderr, "mmvq: unsupported type %d\n", type_a); abort();
    }
}

void mul_mat_vec_q(int type_a, const void * W, size_t w_row_stride, int64_t m, int64_t k,
                   const act_q8 & act, int64_t n, float * dst, size_t dst_col_stride_bytes, hipStream_t stream) {
    if (m == 0 || n == 0) return;
    mmvq_args a = {};
    a.W = (const char *) W; a.w_row_stride = w_row_stride; a.m = m; a.k = k;
    a.a_qs = act.qs; a.a_d = act.d; a.a_bs = act.bsums;
    a.dst = dst; a.dst_col_stride = dst_col_stride_bytes;
    launch_mmvq<false>(type_a, a, act.kind, n, 1, stream);
}

void mul_mat_vec_q_id(int type_a, const void * W, size_t w_row_stride, size_t w_expert_stride, int64_t m, int64_t k,
                      const act_q8 & act, const int32_t * ids, size_t ids_nb0, size_t ids_nb1,
                      int64_t n_used, int64_t n_tokens, int64_t n_b,
                      float * dst, size_t dst_nb1, size_t dst_nb2, hipStream_t stream) {
    if (m == 0 || n_used == 0 || n_tokens == 0) return;
    mmvq_args a = {};
    a.W = (const char *) W; a.w_row_stride = w_row_stride; a.w_expert_stride = w_expert_stride; a.m = m; a.k = k;
    a.a_qs = act.qs; a.a_d = act.d; a.a_bs = act.bsums;
    a.dst = dst; a.dst_col_stride = 0;
    a.ids = (const char *) ids; a.ids_nb0 = ids_nb0; a.ids_nb1 = ids_nb1; a.n_used = (int) n_used; a.n_b = (int) n_b;
    a.dst_nb1 = dst_nb1; a.dst_nb2 = dst_nb2;
    launch_mmvq<true>(type_a, a, act.kind, 1, n_used*n_tokens, stream);
}

} // namespace mi355x
