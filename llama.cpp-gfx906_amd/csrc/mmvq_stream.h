// mmvq_stream.h — the STREAMED quantized mat-vec (n = 1): device code (host side: mmvq_stream.hip).
//
// What bounds a decode step is how many bytes of packed weights a CU pulls out of HBM per microsecond, and round 2's kernels lost twice:
// their weight path lived in registers (two k-steps ahead: ~1.3 us of stream, so the 3-5 us prologue — load x, norm, quantize — starved it
// at the start of every launch), and they spent ~540 vector instructions per 2.3 KB of weights (8 lanes per block, each repeating the
// scale unpack and the float math, a 64-lane reduction per row pair): instruction-bound at 17-22 GB/s per CU where HBM offers 25.
// This kernel is built the other way round (tools/stream_probe.hip has the measurements that shaped it):
//   * ONE loader wave per workgroup (one workgroup per CU) copies the workgroup's rows — one contiguous byte range per weight tensor —
//     HBM -> LDS with global_load_lds_dwordx4 (1 KiB per instruction, coalesced, nontemporal, no VGPRs) into a ring of slots; it starts
//     right after the consumers have requested their activation, never waits for the prologue, keeps 4-6 slots in flight and blocks only
//     on the memory pipeline's own queue (8 waves that each filled a private ring spent 4-6 us blocked in their own issue, prologue undone).
//   * EIGHT consumer waves build the activation image (the CPU backend's int8 blocks: quant_core.h) while the ring fills, then each takes
//     every 8th slot. A slot = 64 units of 256 weights; a lane owns ONE unit: it reads its block from the slot (a unit's 16-byte chunks are
//     an odd number of chunks apart from its neighbour's: conflict-free) and that block's activation, unpacks the scales once, runs the
//     integer dots of the CPU's vec_dot (same integer sub-sums, oracle/ggml_oracle.c) and one float multiply-add per block.
//   * Nothing is reduced across the wave except 16 lanes (one DPP row) when a row's block count is a multiple of 16; partial sums land in
//     LDS, and after the last slot the consumers add each row's partials in a fixed order and run the epilogue with a lane per ROW (or
//     rotation pair): residual, rotation, SwiGLU, cache stores — no single-live-lane epilogues.
// Synchronisation inside the workgroup is by words in LDS (landed / done counters, polled with s_sleep): the loader cannot stand at an
// s_barrier while it is issuing, so after the first barrier (activation loads are queued before any weight) nobody uses one.
#pragma once

#include "mmvq_core.h"
#include "quant_core.h"
#include "rope_dev.h"

namespace mi355x {

constexpr int ST_NC = 8;                 // consumer waves
constexpr int ST_THREADS = (ST_NC + 1)*64;
constexpr int ST_SYNC_WORDS = 64;        // [0] slots landed, [2] image parts ready, [3] consumers finished, [4] norm partials ready, [16 + s] done[s]
constexpr int ST_MAX_RING = 48;          // slots
// the activation of one 256-block as the consumers read it: 16 chunks of int8; one chunk with the eight 32-element sums split into (h, l)
// signed bytes, sum = 128 h + l (so that sum_j m_j * bsum_j is 4-byte dots); two chunks with the sixteen 16-element sums split the same way
// (Q6_K's -32 offset) — 19 chunks = 304 bytes per block, an odd number of 16-byte slots: lanes that hold consecutive blocks read
// conflict-free. The block's Q8_K scale sits in a float array of its own.
constexpr int ST_ACT_STRIDE = 304;

struct st_group {
    const char * W; const char * W2;          // W2: the second tensor of EPI_GLU
    float * dst;
    const float * res; const float * res2;    // EPI_ADD addends; EPI_ROPE: res = a bias added before the rotation
    uint16_t * st16; const int64_t * st_idx; long long st_row_elems;
    int m, type, epi, st_mode;
    int ralign;                               // rows are dealt to workgroups in multiples of this (2: rotation pairs, head size: NEOX pairs)
    int npart_max;                            // floats of partial sums the largest workgroup of this group needs (the LDS carve is the same in all of them)
    float glu_alpha, glu_limit;
};
struct st_args {
    int n_groups, k, nb, mode;
    int block_end[MMVQ_MAX_GROUPS];
    uint32_t magic; int S; float eps; int pad0;
    const float * x; const float * norm_w;
    const int8_t * a_qs; const float * a_d; const int16_t * a_bs;      // PRO_Q8: the n = 1 image act_q8_carve lays out
    fused_rope rope;
    st_group g[MMVQ_MAX_GROUPS];
    unsigned long long * stamps;              // diagnostic builds (-DMI_STAMPS): [workgroup][wave][8]
};

// LDS-DMA, 1 KiB per instruction: lane l's 16 bytes at gbase + OFF + 16*l -> LDS M0 + OFF + 16*l (the instruction's offset field advances
// BOTH addresses). gbase and the LDS address are wave-uniform (scalar registers), voff = 16*lane: nothing per piece is vector work — a
// loader that did 64-bit vector address arithmetic per piece was instruction-bound at 13 GB/s per CU. M0 is compiler-reserved, but the
// compiler sets it before each of its own uses, so it is not saved here. Nontemporal (NT): these bytes are read once per token
// (measured: 6.4-6.5 TB/s against 5.8 with the default policy on a 295 MB stream).
template <bool NT, int N>      // N <= 4 pieces: source gbase .. gbase + N KiB -> LDS lds_dst .. lds_dst + N KiB
static __device__ __forceinline__ void st_dma_4(const char * gbase, uint32_t voff, uint32_t lds_dst) {
#define MI_DMA(OFF_) "global_load_lds_dwordx4 %0, %2 offset:" #OFF_
    if (NT) {
        if (N == 1) asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\t" MI_DMA(0) " nt" :: "v"(voff), "s"(lds_dst), "s"(gbase) : "memory");
        if (N == 2) asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\t" MI_DMA(0) " nt\n\t" MI_DMA(1024) " nt" :: "v"(voff), "s"(lds_dst), "s"(gbase) : "memory");
        if (N == 3) asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\t" MI_DMA(0) " nt\n\t" MI_DMA(1024) " nt\n\t" MI_DMA(2048) " nt" :: "v"(voff), "s"(lds_dst), "s"(gbase) : "memory");
        if (N == 4) asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\t" MI_DMA(0) " nt\n\t" MI_DMA(1024) " nt\n\t" MI_DMA(2048) " nt\n\t" MI_DMA(3072) " nt" :: "v"(voff), "s"(lds_dst), "s"(gbase) : "memory");
    } else {
        if (N == 1) asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\t" MI_DMA(0) :: "v"(voff), "s"(lds_dst), "s"(gbase) : "memory");
        if (N == 2) asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\t" MI_DMA(0) "\n\t" MI_DMA(1024) :: "v"(voff), "s"(lds_dst), "s"(gbase) : "memory");
        if (N == 3) asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\t" MI_DMA(0) "\n\t" MI_DMA(1024) "\n\t" MI_DMA(2048) :: "v"(voff), "s"(lds_dst), "s"(gbase) : "memory");
        if (N == 4) asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\t" MI_DMA(0) "\n\t" MI_DMA(1024) "\n\t" MI_DMA(2048) "\n\t" MI_DMA(3072) :: "v"(voff), "s"(lds_dst), "s"(gbase) : "memory");
    }
#undef MI_DMA
}
// the PPS pieces of one slot (source bytes gbase .. gbase + PPS KiB, contiguous)
template <bool NT, int PPS>
static __device__ __forceinline__ void st_dma_slot(const char * gbase, uint32_t voff, uint32_t lds_dst) {
    static_assert(PPS >= 1 && PPS <= 20, "a slot is at most 20 KiB");
#define MI_G(q4_) if constexpr (PPS > 4*(q4_)) st_dma_4<NT, (PPS - 4*(q4_) >= 4 ? 4 : PPS - 4*(q4_))>(gbase + (q4_)*4096, voff, lds_dst + (q4_)*4096);
    MI_G(0) MI_G(1) MI_G(2) MI_G(3) MI_G(4)
#undef MI_G
}

// the last slot of a tensor's stream: lanes whose 16 bytes would lie past `lim` (the tensor's last 16 bytes, relative to gbase) re-read those
template <bool NT, int PPS>
static __device__ __forceinline__ void st_dma_slot_clamped(const char * gbase, uint32_t voff, uint32_t lds_dst, uint32_t lim) {
#pragma unroll
    for (int q = 0; q < PPS; q++) st_dma_4<NT, 1>(gbase, min(voff + q*1024u, lim), lds_dst + q*1024);
}

typedef const char __attribute__((address_space(3))) * st_lptr;
static __device__ __forceinline__ uint32_t st_lds_addr(const void * p) { return (uint32_t)(size_t)(st_lptr) p; }
static __device__ __forceinline__ uint32_t st_poll_ld(const uint32_t * w) {
    return __hip_atomic_load((const uint32_t __attribute__((address_space(3))) *)(uintptr_t) st_lds_addr(w), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
static __device__ __forceinline__ void st_flag_st(uint32_t * w, uint32_t v) {
    __hip_atomic_store((uint32_t __attribute__((address_space(3))) *)(uintptr_t) st_lds_addr(w), v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
static __device__ __forceinline__ void st_flag_add(uint32_t * w, uint32_t v) {
    __hip_atomic_fetch_add((uint32_t __attribute__((address_space(3))) *)(uintptr_t) st_lds_addr(w), v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
// wait until *w >= target (all lanes read the same word: one broadcast read per trip)
static __device__ __forceinline__ void st_wait_ge(const uint32_t * w, uint32_t target) {
    while (st_poll_ld(w) < target) __builtin_amdgcn_s_sleep(1);
    asm volatile("" ::: "memory");      // nothing that reads what the word guards moves above the wait
}
// every consumer wave arrives once at counter w (its LDS writes drained first), then waits for all of them
static __device__ __forceinline__ void st_consumers_meet(uint32_t * w, int lane) {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    if (lane == 0) st_flag_add(w, 1u);
    st_wait_ge(w, ST_NC);
}

// ---- per-format unit: UB bytes of packed weights = 256 weights; one lane consumes one unit ----
//   load(a): the unit's bytes from LDS byte address a into registers;  dot(w, ab, d8): ab = the block's activation (ST_ACT_STRIDE bytes), d8 its scale
template <int TYPE> struct st_unit;

static __device__ __forceinline__ int4v st_ld16(uint32_t a) { return *(const int4v __attribute__((address_space(3))) *)(uintptr_t) a; }
static __device__ __forceinline__ uint32_t st_ld4(uint32_t a) { return *(const uint32_t __attribute__((address_space(3))) *)(uintptr_t) a; }

// the 6-bit scales / mins of a Q4_K / Q5_K header as 4 bytes per word (quants.py:479-501)
static __device__ __forceinline__ void st_k4_scales(const int4v hdr, uint32_t & sc_lo, uint32_t & sc_hi, uint32_t & m_lo, uint32_t & m_hi) {
    const uint32_t s0 = (uint32_t) hdr.y, s1 = (uint32_t) hdr.z, s2 = (uint32_t) hdr.w;
    sc_lo = s0 & 0x3F3F3F3Fu; m_lo = s1 & 0x3F3F3F3Fu;
    sc_hi = (s2 & 0x0F0F0F0Fu) | ((s0 >> 2) & 0x30303030u);
    m_hi  = ((s2 >> 4) & 0x0F0F0F0Fu) | ((s1 >> 2) & 0x30303030u);
}
// sum_j m_j * bsum32_j from the block's sums chunk (h0..7 | l0..7)
static __device__ __forceinline__ int st_k4_mins(uint32_t m_lo, uint32_t m_hi, const int4v HL) {
    return (dot4((int) m_lo, HL.x, dot4((int) m_hi, HL.y, 0)) << 7) + dot4((int) m_lo, HL.z, dot4((int) m_hi, HL.w, 0));
}

template <> struct st_unit<T_Q4_K> {
    static constexpr int UB = 144;
    struct wfrag { int4v c[9]; };
    static __device__ __forceinline__ wfrag load(uint32_t a) {
        wfrag w;
#pragma unroll
        for (int j = 0; j < 9; j++) w.c[j] = st_ld16(a + 16*j);
        return w;
    }
    static __device__ __forceinline__ float dot(const wfrag & w, const char * ab, float d8) {
        uint32_t sc_lo, sc_hi, m_lo, m_hi;
        st_k4_scales(w.c[0], sc_lo, sc_hi, m_lo, m_hi);
        int isum = 0;
#pragma unroll
        for (int g = 0; g < 4; g++) {
            // 32 bytes of qs: low nibbles = sub-block 2g, high nibbles = sub-block 2g + 1 (quants.py:504-522)
            const int4v qa = w.c[1 + 2*g], qb = w.c[2 + 2*g];
            const int4v A0 = *(const int4v *) (ab + 64*g), A1 = *(const int4v *) (ab + 64*g + 16), A2 = *(const int4v *) (ab + 64*g + 32), A3 = *(const int4v *) (ab + 64*g + 48);
            int dlo = 0, dhi = 0;
#define MI_N(q_, a_, b_) { const uint32_t q = (uint32_t)(q_); dlo = dot4((int)(q & 0x0F0F0F0Fu), a_, dlo); dhi = dot4((int)((q >> 4) & 0x0F0F0F0Fu), b_, dhi); }
            MI_N(qa.x, A0.x, A2.x) MI_N(qa.y, A0.y, A2.y) MI_N(qa.z, A0.z, A2.z) MI_N(qa.w, A0.w, A2.w)
            MI_N(qb.x, A1.x, A3.x) MI_N(qb.y, A1.y, A3.y) MI_N(qb.z, A1.z, A3.z) MI_N(qb.w, A1.w, A3.w)
#undef MI_N
            const uint32_t scw = (g < 2 ? sc_lo : sc_hi) >> (16*(g & 1));
            isum += __mul24((int)(scw & 0xFF), dlo) + __mul24((int)((scw >> 8) & 0xFF), dhi);
        }
        const int msum = st_k4_mins(m_lo, m_hi, *(const int4v *) (ab + 256));
        const float d = f16_bits_to_f32((uint16_t)((uint32_t) w.c[0].x & 0xFFFF)), dmin = f16_bits_to_f32((uint16_t)((uint32_t) w.c[0].x >> 16));
        return (d*d8)*(float) isum - (dmin*d8)*(float) msum;
    }
};

template <> struct st_unit<T_Q5_K> {
    static constexpr int UB = 176;
    struct wfrag { int4v c[11]; };      // header | qh (2 chunks) | qs (8 chunks)
    static __device__ __forceinline__ wfrag load(uint32_t a) {
        wfrag w;
#pragma unroll
        for (int j = 0; j < 11; j++) w.c[j] = st_ld16(a + 16*j);
        return w;
    }
    static __device__ __forceinline__ float dot(const wfrag & w, const char * ab, float d8) {
        uint32_t sc_lo, sc_hi, m_lo, m_hi;
        st_k4_scales(w.c[0], sc_lo, sc_hi, m_lo, m_hi);
        int isum = 0;
#pragma unroll
        for (int g = 0; g < 4; g++) {
            // as Q4_K, plus bit 2g (2g + 1) of qh[l] = the fifth bit of element l of sub-block 2g (2g + 1) (quants.py:527-549)
            const int4v qa = w.c[3 + 2*g], qb = w.c[4 + 2*g], ha = w.c[1], hb = w.c[2];
            const int4v A0 = *(const int4v *) (ab + 64*g), A1 = *(const int4v *) (ab + 64*g + 16), A2 = *(const int4v *) (ab + 64*g + 32), A3 = *(const int4v *) (ab + 64*g + 48);
            int dlo = 0, dhi = 0;
#define MI_N(q_, h_, a_, b_) { const uint32_t q = (uint32_t)(q_), h = (uint32_t)(h_); \
            dlo = dot4((int)((q & 0x0F0F0F0Fu) | (((h >> (2*g)) & 0x01010101u) << 4)), a_, dlo); \
            dhi = dot4((int)(((q >> 4) & 0x0F0F0F0Fu) | (((h >> (2*g + 1)) & 0x01010101u) << 4)), b_, dhi); }
            MI_N(qa.x, ha.x, A0.x, A2.x) MI_N(qa.y, ha.y, A0.y, A2.y) MI_N(qa.z, ha.z, A0.z, A2.z) MI_N(qa.w, ha.w, A0.w, A2.w)
            MI_N(qb.x, hb.x, A1.x, A3.x) MI_N(qb.y, hb.y, A1.y, A3.y) MI_N(qb.z, hb.z, A1.z, A3.z) MI_N(qb.w, hb.w, A1.w, A3.w)
#undef MI_N
            const uint32_t scw = (g < 2 ? sc_lo : sc_hi) >> (16*(g & 1));
            isum += __mul24((int)(scw & 0xFF), dlo) + __mul24((int)((scw >> 8) & 0xFF), dhi);
        }
        const int msum = st_k4_mins(m_lo, m_hi, *(const int4v *) (ab + 256));
        const float d = f16_bits_to_f32((uint16_t)((uint32_t) w.c[0].x & 0xFFFF)), dmin = f16_bits_to_f32((uint16_t)((uint32_t) w.c[0].x >> 16));
        return (d*d8)*(float) isum - (dmin*d8)*(float) msum;
    }
};

// Q6_K: 210-byte blocks — a unit starts on a 2-byte boundary in the slot. A lane reads the 53 aligned dwords that cover its block and
// realigns them with one v_alignbit each (shift 0 or 16, the same instruction on every lane).
template <> struct st_unit<T_Q6_K> {
    static constexpr int UB = 210;
    struct wfrag { uint32_t d[53]; uint32_t sh; };
    static __device__ __forceinline__ wfrag load(uint32_t a) {
        wfrag w;
        const uint32_t a4 = a & ~3u;
        w.sh = (a & 2u)*8;
#pragma unroll
        for (int j = 0; j < 53; j++) w.d[j] = st_ld4(a4 + 4*j);
        return w;
    }
    static __device__ __forceinline__ uint32_t dw(const wfrag & w, int i) {      // dword i of the block
        return __builtin_amdgcn_alignbit(w.d[i < 52 ? i + 1 : 52], w.d[i], w.sh);
    }
    static __device__ __forceinline__ float dot(const wfrag & w, const char * ab, float d8) {
        // ql 128 B (dwords 0..31) | qh 64 B (32..47) | 16 int8 scales (48..51) | d (52, low half) — quants.py:554-572
        int isum = 0;
#pragma unroll
        for (int n = 0; n < 2; n++) {
            int acc[8];      // the eight 16-element groups of this half: group 8n + 2i + (t >> 2)
#pragma unroll
            for (int i = 0; i < 8; i++) acc[i] = 0;
#pragma unroll
            for (int t = 0; t < 8; t++) {
                const uint32_t qa = dw(w, 16*n + t), qb = dw(w, 16*n + 8 + t), qh = dw(w, 32 + 8*n + t);
                const int v0 = (int)((qa & 0x0F0F0F0Fu)        | ((qh << 4) & 0x30303030u));
                const int v1 = (int)((qb & 0x0F0F0F0Fu)        | ((qh << 2) & 0x30303030u));
                const int v2 = (int)(((qa >> 4) & 0x0F0F0F0Fu) | ( qh       & 0x30303030u));
                const int v3 = (int)(((qb >> 4) & 0x0F0F0F0Fu) | ((qh >> 2) & 0x30303030u));
                const int * ap = (const int *) (ab + 128*n + 4*t);
                acc[0 + (t >> 2)] = dot4(v0, ap[0],  acc[0 + (t >> 2)]);
                acc[2 + (t >> 2)] = dot4(v1, ap[8],  acc[2 + (t >> 2)]);
                acc[4 + (t >> 2)] = dot4(v2, ap[16], acc[4 + (t >> 2)]);
                acc[6 + (t >> 2)] = dot4(v3, ap[24], acc[6 + (t >> 2)]);
            }
            const uint32_t sA = dw(w, 48 + 2*n), sB = dw(w, 48 + 2*n + 1);      // scales 8n .. 8n + 7
#pragma unroll
            for (int i = 0; i < 8; i++) {
                const int sc = (int)(int8_t)(((i < 4 ? sA : sB) >> (8*(i & 3))) & 0xFF);
                isum += __mul24(sc, acc[i]);
            }
        }
        // - 32 * sum_j sc_j * bsum16_j: the sixteen 16-element sums as (h, l) bytes in chunks 17 / 18 of the activation block
        const int4v H = *(const int4v *) (ab + 272), L = *(const int4v *) (ab + 288);
        const int s0 = (int) dw(w, 48), s1 = (int) dw(w, 49), s2 = (int) dw(w, 50), s3 = (int) dw(w, 51);
        const int bs = (dot4(s0, H.x, dot4(s1, H.y, dot4(s2, H.z, dot4(s3, H.w, 0)))) << 7) + dot4(s0, L.x, dot4(s1, L.y, dot4(s2, L.z, dot4(s3, L.w, 0))));
        isum -= 32*bs;
        const float d = f16_bits_to_f32((uint16_t)(dw(w, 52) & 0xFFFF));
        return (d*d8)*(float) isum;
    }
};

#ifdef MI_STAMPS
#define ST_STAMP(i_) do { if (p.stamps && lane == 0) p.stamps[((size_t) blockIdx.x*(ST_NC + 1) + wave)*8 + (i_)] = __builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define ST_STAMP(i_) do { } while (0)
#endif

// (h, l) bytes of a block sum: s = 128 h + l, l in [-64, 63]
static __device__ __forceinline__ void st_hl(int s, int & h, int & l) { h = (s + 64) >> 7; l = s - (h << 7); }

// the sum over a row's partials (fixed order)
static __device__ __forceinline__ float st_row_sum(const float * part, int row, int npr) {
    float s = 0.0f;
    for (int q = 0; q < npr; q++) s += part[row*npr + q];
    return s;
}

template <int TYPE, bool NT>
static __device__ __forceinline__ void st_body(const st_args & p, const st_group & g, const int wg, const int nwg, char * lds, const int lane, const int wave) {
    typedef st_unit<TYPE> U;
    constexpr int PPS = (64*U::UB + 1023)/1024, SLOT = PPS*1024;
    constexpr int DEPTH = 63/PPS < 6 ? 63/PPS : 6;      // slots the loader keeps in flight (a wave counts at most 63 outstanding loads)
    const int nb = p.nb;
    const bool GLU = g.epi == EPI_GLU;
    // rows of this workgroup: [r0, r0 + R), dealt in multiples of ralign
    // (ralign also keeps every workgroup's first byte 16-byte aligned — Q6_K rows are 210 nb bytes; the rows past the last whole unit belong to the last workgroup)
    const int nru = g.m/g.ralign;
    const int r0 = (int)((long long) wg*nru/nwg)*g.ralign, R = (wg == nwg - 1 ? g.m : (int)((long long)(wg + 1)*nru/nwg)*g.ralign) - r0;
    const int n1 = R*nb, ns1 = (n1 + 63) >> 6, nslots = GLU ? 2*ns1 : ns1;      // units / slots of one stream; slots of the workgroup
    const int S = p.S;
    const bool row16 = (nb & 15) == 0;               // a DPP row of 16 lanes = 16 units of ONE weight row
    const int npr = row16 ? nb >> 4 : nb;            // partials per row
    uint32_t * sync = (uint32_t *) lds;
    char * act = lds + ST_SYNC_WORDS*4;
    float * dd = (float *) (act + (size_t) nb*ST_ACT_STRIDE);
    float * red = dd + ((nb + 3) & ~3);              // [ST_NC] sums of squares (PRO_NORM)
    float * part = red + 16;
    char * ring = (char *) (((size_t)(part + g.npart_max) + 15) & ~(size_t) 15);
    ST_STAMP(0);
    if (threadIdx.x < ST_SYNC_WORDS) sync[threadIdx.x] = 0;

    // ---- consumers: request the activation before any weight is requested (a CU returns loads in request order) ----
    const int mode = p.mode;
    const int nchunk = nb;                            // 256-element chunks of the activation; consumer wave w owns chunks w, w + 8, ...
    float4v xv[8], wv[8];
    int4v areg[2], breg[2]; float dreg = 0.0f;
    if (wave < ST_NC) {
        if (mode == PRO_Q8) {
            const int nq = p.k >> 4;
#pragma unroll
            for (int i = 0; i < 2; i++) { const int q = min((int) threadIdx.x + i*ST_NC*64, nq - 1); areg[i] = *(const int4v *) (p.a_qs + (size_t) q*16); }
            const int ibl = min((int) threadIdx.x, nb - 1);
            breg[0] = *(const int4v *) (p.a_bs + (size_t) ibl*16); breg[1] = *(const int4v *) (p.a_bs + (size_t) ibl*16 + 8);
            dreg = p.a_d[ibl];
        } else {
#pragma unroll
            for (int i = 0; i < 8; i++) {
                const int c = wave + ST_NC*i;
                if (c < nchunk) {      // wave-uniform
                    xv[i] = *(const float4v *) (p.x + (size_t) c*256 + lane*4);
                    if (mode == PRO_NORM) wv[i] = *(const float4v *) (p.norm_w + (size_t) c*256 + lane*4);
                }
            }
        }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();

    if (wave == ST_NC) {
        // ================= the loader =================
        const uint32_t ring_a = st_lds_addr(ring);
        const uint32_t voff = lane*16;
        int landed = 0;
        for (int i = 0; i < nslots; i++) {
            const int si = GLU ? (i >= ns1) : 0, il = i - si*ns1;
            const char * gb = (si ? g.W2 : g.W) + (long long) r0*nb*U::UB + (long long) il*64*U::UB;
            if (i >= S) {
                // the slot must have been consumed; publish what is in flight first so that nobody waits for us meanwhile
                if (st_poll_ld(&sync[16 + i % S]) < (uint32_t)(i - S + 1)) {
                    if (landed < i) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); landed = i; if (lane == 0) st_flag_st(&sync[0], (uint32_t) landed); }
                    st_wait_ge(&sync[16 + i % S], (uint32_t)(i - S + 1));
                }
            }
            if (il == ns1 - 1) {       // the stream's last slot may reach past the end of the tensor
                const long long lim = (long long) g.m*nb*U::UB - 16 - ((long long) r0*nb*U::UB + (long long) il*64*U::UB);
                st_dma_slot_clamped<NT, PPS>(gb, voff, ring_a + (uint32_t)(i % S)*SLOT, (uint32_t)(lim < 0x7FFFFFFF ? lim : 0x7FFFFFFF));
            } else st_dma_slot<NT, PPS>(gb, voff, ring_a + (uint32_t)(i % S)*SLOT);
            if (i >= DEPTH - 1) {      // all but the youngest DEPTH - 1 slots have landed
                asm volatile("s_waitcnt vmcnt(%0)" :: "n"((DEPTH - 1)*PPS) : "memory");
                if (landed < i - (DEPTH - 2)) { landed = i - (DEPTH - 2); if (lane == 0) st_flag_st(&sync[0], (uint32_t) landed); }
            }
        }
#define MI_DRAIN(d_) if (DEPTH - 2 >= (d_)) { asm volatile("s_waitcnt vmcnt(%0)" :: "n"((d_)*PPS) : "memory"); \
            if (nslots - (d_) > landed) { landed = nslots - (d_); if (lane == 0) st_flag_st(&sync[0], (uint32_t) landed); } }
        MI_DRAIN(4) MI_DRAIN(3) MI_DRAIN(2) MI_DRAIN(1) MI_DRAIN(0)
#undef MI_DRAIN
        ST_STAMP(1);
        return;
    }

    // ================= consumers =================
    // ---- the activation image ----
    ST_STAMP(1);
    if (mode == PRO_Q8) {
#pragma unroll
        for (int i = 0; i < 2; i++) {
            const int q = threadIdx.x + i*ST_NC*64;
            if (q < (p.k >> 4)) { const int ib = q >> 4, c = q & 15; *(int4v *) (act + (size_t) ib*ST_ACT_STRIDE + c*16) = areg[i]; }
        }
        if ((int) threadIdx.x < nb) {
            const int ib = threadIdx.x;
            uint32_t h32[2] = { 0, 0 }, l32[2] = { 0, 0 }, h16[4] = { 0, 0, 0, 0 }, l16[4] = { 0, 0, 0, 0 };
#pragma unroll
            for (int j = 0; j < 8; j++) {
                const uint32_t wsum = (uint32_t)(j < 4 ? breg[0][j] : breg[1][j - 4]);
                const int sa = (int)(int16_t)(wsum & 0xFFFF), sb = (int)(int16_t)(wsum >> 16);
                int h, l;
                st_hl(sa + sb, h, l); h32[j >> 2] |= (uint32_t)(h & 0xFF) << (8*(j & 3)); l32[j >> 2] |= (uint32_t)(l & 0xFF) << (8*(j & 3));
                st_hl(sa, h, l); h16[j >> 1] |= (uint32_t)(h & 0xFF) << (8*((2*j) & 3)); l16[j >> 1] |= (uint32_t)(l & 0xFF) << (8*((2*j) & 3));
                st_hl(sb, h, l); h16[j >> 1] |= (uint32_t)(h & 0xFF) << (8*((2*j + 1) & 3)); l16[j >> 1] |= (uint32_t)(l & 0xFF) << (8*((2*j + 1) & 3));
            }
            char * ab = act + (size_t) ib*ST_ACT_STRIDE;
            *(int4v *) (ab + 256) = int4v{ (int) h32[0], (int) h32[1], (int) l32[0], (int) l32[1] };
            *(int4v *) (ab + 272) = int4v{ (int) h16[0], (int) h16[1], (int) h16[2], (int) h16[3] };
            *(int4v *) (ab + 288) = int4v{ (int) l16[0], (int) l16[1], (int) l16[2], (int) l16[3] };
            dd[ib] = dreg;
        }
    } else {
        float scale = 1.0f;
        if (mode == PRO_NORM) {
            float ss = 0.0f;
#pragma unroll
            for (int i = 0; i < 8; i++) if (wave + ST_NC*i < nchunk) ss += xv[i].x*xv[i].x + xv[i].y*xv[i].y + xv[i].z*xv[i].z + xv[i].w*xv[i].w;
            ss = wave_sum(ss);
            if (lane == 0) red[wave] = ss;
            st_consumers_meet(&sync[4], lane);
            ss = ((red[0] + red[1]) + (red[2] + red[3])) + ((red[4] + red[5]) + (red[6] + red[7]));
            scale = 1.0f/sqrtf(ss/(float) p.k + p.eps);
        }
#pragma unroll
        for (int i = 0; i < 8; i++) {
            const int c = wave + ST_NC*i;
            if (c < nchunk) {
                float4v v = xv[i];
                if (mode == PRO_NORM) { v.x = (v.x*scale)*wv[i].x; v.y = (v.y*scale)*wv[i].y; v.z = (v.z*scale)*wv[i].z; v.w = (v.w*scale)*wv[i].w; }
                float d8; int bs16;
                const uint32_t q4 = quant_frag_q8_K(v, d8, bs16);
                char * ab = act + (size_t) c*ST_ACT_STRIDE;
                *(uint32_t *) (ab + lane*4) = q4;
                // the 16-element sum of quad q = lane >> 2 (valid in its four lanes); the 32-element sum j at lane 8j + 4 (row_shr:4 brings lane 8j's)
                const int bs32 = bs16 + dpp_i<0x114>(bs16);
                int h, l;
                st_hl(bs16, h, l);
                if ((lane & 3) == 0) { ab[272 + (lane >> 2)] = (char) h; ab[288 + (lane >> 2)] = (char) l; }
                st_hl(bs32, h, l);
                if ((lane & 7) == 4) { ab[256 + (lane >> 3)] = (char) h; ab[264 + (lane >> 3)] = (char) l; }
                if (lane == 0) dd[c] = d8;
            }
        }
    }
    st_consumers_meet(&sync[2], lane);
    ST_STAMP(2);

    // ---- the epilogue's operands of this thread's first row / pair: requested now, needed after the last slot ----
    float e_r0 = 0.0f, e_r1 = 0.0f, e_q0 = 0.0f, e_c = 1.0f, e_s = 0.0f; long long e_i0 = 0, e_i1 = 0;
    const long long idx0 = g.st_mode == 1 ? g.st_idx[0] : 0;
    if (g.epi == EPI_ROPE) {
        const fused_rope & rp = p.rope;
        const int pr = threadIdx.x, hd = rp.head_dim;
        if (pr < (R >> 1)) {
            int ra, rb, ip;
            if (rp.neox) { const int hh = pr/(hd >> 1), i = pr - hh*(hd >> 1); ra = hh*hd + i; rb = ra + (hd >> 1); ip = i; }
            else         { ra = 2*pr; rb = ra + 1; ip = ((r0 + ra) % hd) >> 1; }
            if (g.res) { e_r0 = g.res[r0 + ra]; e_r1 = g.res[r0 + rb]; }
            if (ip < (rp.n_dims >> 1)) { e_c = rp.tab[2*ip]; e_s = rp.tab[2*ip + 1]; }
            if (g.st_mode == 2) { e_i0 = g.st_idx[r0 + ra]; e_i1 = g.st_idx[r0 + rb]; }
        }
    } else if ((int) threadIdx.x < R) {
        const int row = r0 + threadIdx.x;
        if (g.epi == EPI_ADD) { e_r0 = g.res[row]; if (g.res2) e_q0 = g.res2[row]; }
        if (g.st_mode == 2) e_i0 = g.st_idx[row];
    }

    // ---- the stream ----
    const uint32_t ring_a = st_lds_addr(ring);
    const uint32_t magic = p.magic;
    bool first = true;
    for (int i = wave; i < nslots; i += ST_NC) {
        const int si = GLU ? (i >= ns1) : 0, il = i - si*ns1;
        const int u = il*64 + lane;                              // unit inside the stream
        const bool live = u < n1;
        const int uc = live ? u : n1 - 1;
        const int ib = nb == 1 ? 0 : uc - (int) __umulhi((uint32_t) uc, magic)*nb;      // (the magic number of nb = 1 does not fit 32 bits)
        const char * ab = act + (size_t) ib*ST_ACT_STRIDE;
        const float d8 = dd[ib];
        st_wait_ge(&sync[0], (uint32_t)(i + 1));
        if (first) { ST_STAMP(3); first = false; }
        const typename U::wfrag w = U::load(ring_a + (uint32_t)(i % S)*SLOT + (uint32_t)(live ? lane : 0)*U::UB);
        if (nslots > S) {      // the slot is free as soon as its bytes are in registers
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            if (lane == 0) st_flag_st(&sync[16 + i % S], (uint32_t)(i + 1));
        }
        float res = U::dot(w, ab, d8);
        if (!live) res = 0.0f;
        if (row16) {
            res = row16_sum(res);
            if ((lane & 15) == 0 && live) part[(size_t) si*R*npr + (u >> 4)] = res;
        } else if (live) part[(size_t) si*n1 + u] = res;
    }
    ST_STAMP(4);
    st_consumers_meet(&sync[3], lane);
    ST_STAMP(5);

    // ---- rows: partials added in a fixed order, epilogue with a lane per row (or rotation pair) ----
    const float * part2 = part + (size_t) R*npr;
    if (g.epi == EPI_ROPE) {
        const fused_rope & rp = p.rope;
        const int hd = rp.head_dim, half = rp.n_dims >> 1;
        for (int pr = threadIdx.x; pr < (R >> 1); pr += ST_NC*64) {
            // the two rows of pair pr (local): NORM (2 pr, 2 pr + 1); NEOX: i and i + hd/2 inside one head (ralign = hd)
            int ra, rb, ip;
            if (rp.neox) { const int hh = pr/(hd >> 1), i = pr - hh*(hd >> 1); ra = hh*hd + i; rb = ra + (hd >> 1); ip = i; }
            else         { ra = 2*pr; rb = ra + 1; ip = ((r0 + ra) % hd) >> 1; }
            if (pr != (int) threadIdx.x) {       // (not the prefetched pair: a workgroup with more than 1024 rotated rows)
                e_r0 = e_r1 = 0.0f; e_c = 1.0f; e_s = 0.0f;
                if (g.res) { e_r0 = g.res[r0 + ra]; e_r1 = g.res[r0 + rb]; }
                if (ip < half) { e_c = rp.tab[2*ip]; e_s = rp.tab[2*ip + 1]; }
                if (g.st_mode == 2) { e_i0 = g.st_idx[r0 + ra]; e_i1 = g.st_idx[r0 + rb]; }
            }
            float s0 = st_row_sum(part, ra, npr), s1 = st_row_sum(part, rb, npr);
            if (g.res) { s0 += e_r0; s1 += e_r1; }              // bias first, then the rotation
            if (ip < half) { const float a = s0, b = s1; s0 = a*e_c - b*e_s; s1 = a*e_s + b*e_c; }
            g.dst[r0 + ra] = s0; g.dst[r0 + rb] = s1;
            if (g.st_mode == 1) { uint16_t * q = g.st16 + idx0*g.st_row_elems; q[r0 + ra] = f32_to_f16_bits(s0); q[r0 + rb] = f32_to_f16_bits(s1); }
            else if (g.st_mode == 2) { g.st16[e_i0] = f32_to_f16_bits(s0); g.st16[e_i1] = f32_to_f16_bits(s1); }
        }
    } else {
        for (int rr = threadIdx.x; rr < R; rr += ST_NC*64) {
            float s0 = st_row_sum(part, rr, npr);
            const int row = r0 + rr;
            if (rr != (int) threadIdx.x) {
                if (g.epi == EPI_ADD) { e_r0 = g.res[row]; e_q0 = g.res2 ? g.res2[row] : 0.0f; }
                if (g.st_mode == 2) e_i0 = g.st_idx[row];
            }
            if (GLU) {
                const float up_s = st_row_sum(part2, rr, npr);
                if (g.glu_alpha != 0.0f) {      // swiglu_oai, as elem.hip k_glu
                    const float xc = fminf(s0, g.glu_limit), gc = fmaxf(fminf(up_s, g.glu_limit), -g.glu_limit);
                    s0 = (xc/(1.0f + expf(-xc*g.glu_alpha)))*(gc + 1.0f);
                } else {
                    s0 = (s0/(1.0f + expf(-s0)))*up_s;      // silu(gate)*up, as elem.hip k_glu
                }
            } else if (g.epi == EPI_ADD) {
                s0 += e_r0;
                if (g.res2) s0 += e_q0;
            }
            g.dst[row] = s0;
            if (g.st_mode == 1) g.st16[idx0*g.st_row_elems + row] = f32_to_f16_bits(s0);
            else if (g.st_mode == 2) g.st16[e_i0] = f32_to_f16_bits(s0);
        }
    }
    ST_STAMP(6);
}

template <int TA, int TB, bool NT>
__global__ void __launch_bounds__(ST_THREADS, 3) k_mmvq_stream(const st_args p) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int b = blockIdx.x;
    int gi = 0;
#pragma unroll
    for (int q = 0; q < MMVQ_MAX_GROUPS - 1; q++) if (b >= p.block_end[q]) gi = q + 1;
    const int first = gi ? p.block_end[gi - 1] : 0, nwg = p.block_end[gi] - first;
    const st_group & g = p.g[gi];
    if (TA == TB || g.type == TA) st_body<TA, NT>(p, g, b - first, nwg, lds, lane, wave);
    else                          st_body<TB, NT>(p, g, b - first, nwg, lds, lane, wave);
}

} // namespace mi355x
