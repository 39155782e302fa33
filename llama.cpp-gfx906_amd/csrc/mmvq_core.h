// mmvq_core.h — per-format fragments of the quantized mat-vec (shared by mmvq.hip and decode_fused.hip).
// For every block format: how LPB lanes split one block (`load_w`), which int8 activations each lane needs
// (`load_a`) and the integer dot that restates ggml's generic vec_dot_*_q8_* (oracle/ggml_oracle.c).
#pragma once

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
// Four lanes per 210-byte block (round 2; MI_Q6K_LPB8 keeps round 1's eight): lane slot j: half n = j>>1 (128 elements each), l0 = 16*(j&1):
// the lane owns l = l0..l0+15 of that half, i.e. elements 128n + {0,32,64,96} + l (quants.py:554-572): ql[64n+l] lo/hi nibble,
// ql[64n+32+l] lo/hi nibble, qh[32n+l] 2-bit fields — three 16-byte loads + the scales + d per lane, half the load instructions per byte
// of the eight-lane split (whose 8-byte loads kept the CU's address unit busy: the workgroups of a Q6_K group ran every phase of the
// norm+QKV launch ~1.7x slower than their Q4_K neighbours, tools/stamp_timeline.py).
#ifndef MI_Q6K_LPB8
template <> struct mmvq_t<T_Q6_K> {
    static constexpr int LPB = 4, BLOCK_BYTES = 210, QK = 256, ACT = T_Q8_K;
    struct afrag { int4v a[4]; int s[4]; float d8; };   // s[i] = sum of the 16 int8 of a[i] (for the -32 offset)
    struct wfrag { int4v qla, qlb, qh; int2v sc; uint32_t d; };
    static __device__ __forceinline__ afrag load_a(const act_view & a, int64_t ib, int slot) {
        const int n = slot >> 1, l0 = 16*(slot & 1);
        afrag f;
        const int8_t * p = a.qs + ib*256 + 128*n + l0;
#pragma unroll
        for (int i = 0; i < 4; i++) {
            f.a[i] = lds_or_global_b128(p + 32*i);
            f.s[i] = dot4(0x01010101, f.a[i].x, dot4(0x01010101, f.a[i].y, dot4(0x01010101, f.a[i].z, dot4(0x01010101, f.a[i].w, 0))));
        }
        f.d8 = a.d[ib];
        return f;
    }
    static __device__ __forceinline__ wfrag load_w(const char * row, int64_t ib, int slot) {
        const int n = slot >> 1, l0 = 16*(slot & 1);
        const char * b = row + ib*BLOCK_BYTES;   // only 2-byte aligned: unaligned-mode global loads
        wfrag w;
#ifndef MI_Q6K_A2
        w.qla = ld_b128(b + 64*n + l0);
        w.qlb = ld_b128(b + 64*n + 32 + l0);
        w.qh  = ld_b128(b + 128 + 32*n + l0);
        w.sc  = ld_b64(b + 192 + 8*n);           // scales[8n .. 8n+7]
#else
        w.qla = ld_b128_a2(b + 64*n + l0);
        w.qlb = ld_b128_a2(b + 64*n + 32 + l0);
        w.qh  = ld_b128_a2(b + 128 + 32*n + l0);
        w.sc  = ld_b64_a2(b + 192 + 8*n);        // scales[8n .. 8n+7]
#endif
        w.d   = ld_u16(b + 208);
        return w;
    }
    static __device__ __forceinline__ float dot(const wfrag & w, const afrag & a, int slot) {
        const int is = slot & 1;                 // l0/16
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
        MI_Q6(z, acc[0], acc[1], acc[2], acc[3])
        MI_Q6(w, acc[0], acc[1], acc[2], acc[3])
#undef MI_Q6
        // sum (q-32)*a = sum q*a - 32*sum a
        const int isum = sc0*(acc[0] - 32*a.s[0]) + sc1*(acc[1] - 32*a.s[1]) + sc2*(acc[2] - 32*a.s[2]) + sc3*(acc[3] - 32*a.s[3]);
        return (f16_bits_to_f32((uint16_t) w.d)*a.d8)*(float) isum;
    }
};
#else
// -DMI_Q6K_LPB8 (round 2, measured and NOT the default: correct, but a pure Q6_K model decodes 4.5 % slower with it — 418 vs 438 tok/s — and
// Q4_K_M the same: the slow start of Q6_K launches is not the number of memory requests per byte either).
// Eight lanes per block, each load instruction CONTIGUOUS within the block (the four-lane split above touches the block in 32-byte
// pieces: 4x the memory requests per byte of a Q4_K load): lane j loads ql[16j .. 16j+15] (the block's 128 ql bytes = one run) and
// qh[32n + 16(m&1) ..+15], n = j>>2, m = j&3. Its 16 ql bytes are, for m < 2: l = 16m + t -> elements 128n + l (low nibbles, qh bits 0-1)
// and 128n + 64 + l (high nibbles, qh bits 4-5); for m >= 2: l = 16(m-2) + t -> elements 128n + 32 + l (low, bits 2-3) and 128n + 96 + l
// (high, bits 6-7) (quants.py:554-572). Each 16-element group is exactly one scale and one Q8_K bsum.
template <> struct mmvq_t<T_Q6_K> {
    static constexpr int LPB = 8, BLOCK_BYTES = 210, QK = 256, ACT = T_Q8_K;
    struct afrag { int4v lo, hi; int s_lo, s_hi; float d8; };
    struct wfrag { int4v ql, qh; int2v sc; uint32_t d; };
    static __device__ __forceinline__ afrag load_a(const act_view & a, int64_t ib, int slot) {
        const int n = slot >> 2, m = slot & 3;
        const int g = 8*n + 2*(m >> 1) + (m & 1);         // 16-element group of the low-nibble elements; the high-nibble ones are group g + 4
        afrag f;
        const int8_t * p = a.qs + ib*256 + 16*g;
        f.lo = lds_or_global_b128(p); f.hi = lds_or_global_b128(p + 64);
        f.s_lo = a.bs[ib*16 + g]; f.s_hi = a.bs[ib*16 + g + 4];
        f.d8 = a.d[ib];
        return f;
    }
    static __device__ __forceinline__ wfrag load_w(const char * row, int64_t ib, int slot) {
        const int n = slot >> 2, m = slot & 3;
        const char * b = row + ib*BLOCK_BYTES;   // only 2-byte aligned: unaligned-mode global loads
        wfrag w;
        w.ql = ld_b128(b + 16*slot);
        w.qh = ld_b128(b + 128 + 32*n + 16*(m & 1));
        w.sc = ld_b64(b + 192 + 8*n);            // scales[8n .. 8n+7]
        w.d  = ld_u16(b + 208);
        return w;
    }
    static __device__ __forceinline__ float dot(const wfrag & w, const afrag & a, int slot) {
        const int m = slot & 3;
        const int idx = 2*(m >> 1) + (m & 1);    // scale bytes idx (low-nibble group) and idx + 4 (high-nibble group) of sc
        const uint32_t sw_lo = (uint32_t) w.sc.x >> (8*idx), sw_hi = (uint32_t) w.sc.y >> (8*idx);
        const int sc_lo = (int8_t)(sw_lo & 0xFF), sc_hi = (int8_t)(sw_hi & 0xFF);
        const int sh = 2*(m >> 1);
        int acc_lo = 0, acc_hi = 0;
#define MI_Q6C(c) { \
        const uint32_t ql = (uint32_t) w.ql.c, qh = (uint32_t) w.qh.c; \
        const int v_lo = (int)((ql & 0x0F0F0F0Fu)        | (((qh >> sh) & 0x03030303u) << 4)); \
        const int v_hi = (int)(((ql >> 4) & 0x0F0F0F0Fu) | (((qh >> (sh + 4)) & 0x03030303u) << 4)); \
        acc_lo = dot4(v_lo, a.lo.c, acc_lo); acc_hi = dot4(v_hi, a.hi.c, acc_hi); }
        MI_Q6C(x) MI_Q6C(y) MI_Q6C(z) MI_Q6C(w)
#undef MI_Q6C
        // sum (q-32)*a = sum q*a - 32*sum a
        const int isum = sc_lo*(acc_lo - 32*a.s_lo) + sc_hi*(acc_hi - 32*a.s_hi);
        return (f16_bits_to_f32((uint16_t) w.d)*a.d8)*(float) isum;
    }
};
#endif

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


// R rows x NCOLS columns of dot products for one wave: rows[r] are the weight row pointers, av[c] the activation columns.
// Every lane returns its PARTIAL sums in acc; reduce with wave_sum().
template <int TYPE, int NCOLS, int R>
static __device__ __forceinline__ void mmvq_wave_partial(const char * const (&rows)[R], const act_view (&av)[NCOLS], int64_t nb, int lane,
                                                         float (&acc)[NCOLS][R]) {
    typedef mmvq_t<TYPE> T;
    constexpr int LPB = T::LPB, BPW = 64/LPB;
    const int slot = lane % LPB;
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
}

// bytes of LDS needed to stage one activation column: qs | d | bsums, each padded to 16 bytes
static __host__ __device__ inline size_t act_lds_bytes(int act_kind, int64_t k) {
    const int nd = act_kind == T_Q8_0 ? 32 : 256, nbs = act_kind == T_Q8_0 ? 32 : 16;
    return ((k + 15) & ~15) + (((k/nd)*4 + 15) & ~15) + (((k/nbs)*2 + 15) & ~15);
}

// cooperative copy of one quantized activation column from global memory into LDS (all threads of the block)
template <int ACT>
static __device__ __forceinline__ act_view stage_act_lds(char * smem, const int8_t * a_qs, const float * a_d, const int16_t * a_bs, int64_t k) {
    constexpr int ND = ACT == T_Q8_0 ? 32 : 256, NBS = ACT == T_Q8_0 ? 32 : 16;
    const int64_t qs_b = (k + 15) & ~15, d_b = ((k/ND)*4 + 15) & ~15, bs_b = ((k/NBS)*2 + 15) & ~15;
    const int nt = blockDim.x;
    for (int64_t i = threadIdx.x*16; i < qs_b; i += nt*16) *(int4v *) (smem + i) = ld_b128((const char *) a_qs + i);
    for (int64_t i = threadIdx.x*16; i < d_b;  i += nt*16) *(int4v *) (smem + qs_b + i) = ld_b128((const char *) a_d + i);
    for (int64_t i = threadIdx.x*16; i < bs_b; i += nt*16) *(int4v *) (smem + qs_b + d_b + i) = ld_b128((const char *) a_bs + i);
    act_view v;
    v.qs = (const int8_t *) smem; v.d = (const float *) (smem + qs_b); v.bs = (const int16_t *) (smem + qs_b + d_b);
    return v;
}

} // namespace mi355x
