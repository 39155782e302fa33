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
// Rows whose length is a multiple of 32 but not of 256 (gpt-oss: k = 2880 = 90 blocks of 32): a unit is TEN 32-element blocks (320 weights) of a Q8_0 or
// MXFP4 row — 9 units per row at k = 2880 — against the CPU path's Q8_0 activation: 320 int8 + the ten f16-rounded block scales as floats = 360 bytes,
// held 368 apart (23 chunks: odd). Pseudo type ids for the unit templates:
constexpr int ST_Q8_0_B10 = 1008, ST_MXFP4_B10 = 1039;
constexpr int ST_ACT_STRIDE_B10 = 368;
// MXFP4 weights are E2M1 floats (x 2 = the reference's integer table): gfx950 converts a byte's two of them to f16 in ONE instruction
// (v_cvt_scalef32_pk_f16_fp4) and v_dot2_f32_f16 multiplies the pair with two activations — 2 instructions per 2 weights where the table lookup
// through v_perm_b32 took ~6. The sums stay exact (multiples of 1/2 below 2^15 in f32), so 2 x sum is the reference's integer sumi bit for bit.
// For that the activation of an MXFP4 unit is held as f16 PAIRS (a_i, a_{i+16}) — a byte of a block holds elements i (low nibble) and i + 16 —
// 64 bytes per block, 640 + 40 (the ten block scales) per unit, 688 apart (43 chunks).
constexpr int ST_ACT_STRIDE_FP4 = 688;
typedef _Float16 st_h2 __attribute__((ext_vector_type(2)));

struct st_group {
    const char * W; const char * W2;          // W2: the second tensor of EPI_GLU
    float * dst;
    const float * res; const float * res2;    // EPI_ADD addends; EPI_ROPE: res = a bias added before the rotation
    uint16_t * st16; const int64_t * st_idx; long long st_row_elems;
    int m, type, epi, st_mode;
    // MUL_MAT_ID for one token (src/llama-graph.cpp:569-595): the group's matrices are expert eid[0] of a stack (a device value: W += eid[0]*estride,
    // W2 likewise); x_off: this group's activation starts x_off floats into the launch's x (the down projection reads one vector per used expert)
    const int32_t * eid; long long estride; int x_off;
    // gpt-oss's expert FFN (src/llama-graph.cpp:927-983): per-expert biases added to the two products before swiglu_oai (rows eid[0] of [m, n_expert] tables),
    // and res_eid != 0: EPI_ADD's res is such a table too (MUL_MAT_ID -> ADD_ID)
    const float * b_gate; const float * b_up; int res_eid;
    int ralign;                               // rows are dealt to workgroups in multiples of this (2: rotation pairs, head size: NEOX pairs)
    // NEOX rotation as TWO row streams (neox2 > 0): workgroup wg owns pairs [wg*neox2, (wg + 1)*neox2) of the group's m/2 pairs — rows h*hd + i .. of a head's first
    // half through W and their partners h*hd + hd/2 + i .. through W2 = W + hd/2 rows, exactly the gate / up arrangement — so that a head is spread over
    // hd/2/neox2 workgroups instead of held by one (gpt-oss's norm + QKV launch ran on ~100 of 256 CUs); neox_hh = hd/2
    int neox2, neox_hh;
    int npart_max;                            // floats of partial sums the largest workgroup of this group needs (the LDS carve is the same in all of them)
    float glu_alpha, glu_limit;
};
struct st_args {
    int n_groups, k, nb, mode;                // nb: units per row (k / 256, or k / 320 for the ten-block units)
    int nchunk, act_stride;                   // 256-element pieces of the activation vector ((k + 255) / 256); bytes between the image's units
    int block_end[MMVQ_MAX_GROUPS];
    uint32_t magic; int S; float eps; int early;     // early: slots of the first phase the loader issues BEFORE it meets the consumers at the first barrier (0: none)
    const float * x; const float * norm_w;
    const int8_t * a_qs; const float * a_d; const int16_t * a_bs;      // PRO_Q8: the n = 1 image act_q8_carve lays out
    fused_rope rope;
    st_group g[MMVQ_MAX_GROUPS];
    // PRO_NORM, at most 16 blocks, planes != NULL: the pending MoE combine (kernels.h mmvq_input) — the vector is sum_u w_u * plane u (+ x, the residual);
    // workgroup 0 also stores it to x_out
    const float * planes; int n_planes, plane_stride; float * x_out;
    const float * pl_probs; const int32_t * pl_ids; int pl_mode;
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
    static_assert(PPS >= 1 && PPS <= 24, "a slot is at most 24 KiB");
#define MI_G(q4_) if constexpr (PPS > 4*(q4_)) st_dma_4<NT, (PPS - 4*(q4_) >= 4 ? 4 : PPS - 4*(q4_))>(gbase + (q4_)*4096, voff, lds_dst + (q4_)*4096);
    MI_G(0) MI_G(1) MI_G(2) MI_G(3) MI_G(4) MI_G(5)
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
// (the counters are cumulative: the n-th meeting at w, n = 0, 1, ..., waits for (n + 1)*ST_NC arrivals)
static __device__ __forceinline__ void st_consumers_meet(uint32_t * w, int lane, int n) {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    if (lane == 0) st_flag_add(w, 1u);
    st_wait_ge(w, (uint32_t)(n + 1)*ST_NC);
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

// Q8_0: 34-byte blocks of 32 weights (f16 d | 32 int8) — a unit is EIGHT of them (272 bytes = 17 chunks: odd, conflict-free, and a unit
// starts on a 16-byte boundary). Block j's quants start 34 j + 2 bytes in: dword-aligned for odd j, two bytes off for even j (one
// v_alignbit per dword, the shift known at compile time). The activation is the CPU path's Q8_0 (vec_dot_type of Q8_0,
// ggml/src/ggml-cpu/ggml-cpu.c type traits): the unit's 256 int8 at ab, its eight f16-rounded scales as floats at ab + 256; per block
// sumi * (d_w * d_a), added block after block (ggml_vec_dot_q8_0_q8_0, ggml/src/ggml-cpu/quants.c).
template <> struct st_unit<T_Q8_0> {
    static constexpr int UB = 272;
    struct wfrag { int4v c[17]; };
    static __device__ __forceinline__ wfrag load(uint32_t a) {
        wfrag w;
#pragma unroll
        for (int j = 0; j < 17; j++) w.c[j] = st_ld16(a + 16*j);
        return w;
    }
    static __device__ __forceinline__ uint32_t dw(const wfrag & w, int i) { return (uint32_t) w.c[i >> 2][i & 3]; }
    static __device__ __forceinline__ float dot(const wfrag & w, const char * ab, float) {
        float acc = 0.0f;
        const float4v da0 = *(const float4v *) (ab + 256), da1 = *(const float4v *) (ab + 272);
#pragma unroll
        for (int j = 0; j < 8; j++) {
            const int B = 34*j, q = (B + 2) >> 2;
            const uint32_t dbits = (B & 2) ? dw(w, B >> 2) >> 16 : dw(w, B >> 2) & 0xFFFF;
            const int4v A0 = *(const int4v *) (ab + 32*j), A1 = *(const int4v *) (ab + 32*j + 16);
            int isum = 0;
#pragma unroll
            for (int i = 0; i < 8; i++) {
                const uint32_t qv = ((B + 2) & 2) ? __builtin_amdgcn_alignbit(dw(w, q + i + 1), dw(w, q + i), 16) : dw(w, q + i);
                isum = dot4((int) qv, i < 4 ? A0[i] : A1[i - 4], isum);
            }
            const float da = j < 4 ? da0[j] : da1[j - 4];
            acc += (float) isum*(f16_bits_to_f32((uint16_t) dbits)*da);
        }
        return acc;
    }
};

// Q4_0: 18-byte blocks of 32 weights (f16 d | 16 bytes of nibbles: elements 0..15 low, 16..31 high; value = nibble - 8) — a unit is eight of them
// (144 bytes = 9 chunks). Against the Q8_0 activation image (as for Q8_0 weights) plus the eight 32-element sums of its quants as int16 at ab + 288:
// sum (nib - 8) q = sum nib q - 8 sum q, the integers of ggml_vec_dot_q4_0_q8_0; per block sumi * (d_w * d_a), block after block.
template <> struct st_unit<T_Q4_0> {
    static constexpr int UB = 144;
    struct wfrag { int4v c[9]; };
    static __device__ __forceinline__ wfrag load(uint32_t a) {
        wfrag w;
#pragma unroll
        for (int j = 0; j < 9; j++) w.c[j] = st_ld16(a + 16*j);
        return w;
    }
    static __device__ __forceinline__ uint32_t dw(const wfrag & w, int i) { return (uint32_t) w.c[i >> 2][i & 3]; }
    static __device__ __forceinline__ float dot(const wfrag & w, const char * ab, float) {
        float acc = 0.0f;
        const float4v da0 = *(const float4v *) (ab + 256), da1 = *(const float4v *) (ab + 272);
        const int4v bsv = *(const int4v *) (ab + 288);
#pragma unroll
        for (int j = 0; j < 8; j++) {
            const int B = 18*j, q = (B + 2) >> 2;
            const uint32_t dbits = (B & 2) ? dw(w, B >> 2) >> 16 : dw(w, B >> 2) & 0xFFFF;
            const int4v A0 = *(const int4v *) (ab + 32*j), A1 = *(const int4v *) (ab + 32*j + 16);
            int isum = 0;
#pragma unroll
            for (int i = 0; i < 4; i++) {
                const uint32_t qv = ((B + 2) & 2) ? __builtin_amdgcn_alignbit(dw(w, q + i + 1), dw(w, q + i), 16) : dw(w, q + i);
                isum = dot4((int)(qv & 0x0F0F0F0Fu), A0[i], isum);
                isum = dot4((int)((qv >> 4) & 0x0F0F0F0Fu), A1[i], isum);
            }
            const uint32_t bw = (uint32_t) bsv[j >> 1];
            const int bsum = (int)(int16_t)((j & 1) ? bw >> 16 : bw & 0xFFFF);
            const float da = j < 4 ? da0[j] : da1[j - 4];
            acc += (float)(isum - 8*bsum)*(f16_bits_to_f32((uint16_t) dbits)*da);
        }
        return acc;
    }
};

// ---- ten-block units (k % 256 != 0) ----
// Q8_0: 340 bytes = 85 dwords, dword-aligned in the slot (odd number of dwords: conflict-free ds_read_b32). Block j: d at byte 34 j, quants at 34 j + 2.
template <> struct st_unit<ST_Q8_0_B10> {
    static constexpr int UB = 340;
    struct wfrag { uint32_t d[85]; };
    static __device__ __forceinline__ wfrag load(uint32_t a) {
        wfrag w;
#pragma unroll
        for (int j = 0; j < 85; j++) w.d[j] = st_ld4(a + 4*j);
        return w;
    }
    static __device__ __forceinline__ float dot(const wfrag & w, const char * ab, float) {
        float acc = 0.0f;
#pragma unroll
        for (int j = 0; j < 10; j++) {
            const int B = 34*j, q = (B + 2) >> 2;
            const uint32_t dbits = (B & 2) ? w.d[B >> 2] >> 16 : w.d[B >> 2] & 0xFFFF;
            const int4v A0 = *(const int4v *) (ab + 32*j), A1 = *(const int4v *) (ab + 32*j + 16);
            int isum = 0;
#pragma unroll
            for (int i = 0; i < 8; i++) {
                const uint32_t qv = ((B + 2) & 2) ? __builtin_amdgcn_alignbit(w.d[q + i + 1], w.d[q + i], 16) : w.d[q + i];
                isum = dot4((int) qv, i < 4 ? A0[i] : A1[i - 4], isum);
            }
            acc += (float) isum*(f16_bits_to_f32((uint16_t) dbits)*((const float *) (ab + 320))[j]);
        }
        return acc;
    }
};
// MXFP4: 170 bytes (ten blocks of {e8m0 scale, 16 bytes of nibbles: elements 0..15 low, 16..31 high}) — a unit starts on a 2-byte boundary: the 44 aligned
// dwords that cover it are read and realigned once (v_alignbit, shift 0 or 16 as for Q6_K); block j's bytes then sit at the compile-time offset 17 j.
// Values through the 16-entry table (mxfp4_lut4, mmvq_core.h); per block sumi * (d_a * 2^(e - 127) / 2) as ggml_vec_dot_mxfp4_q8_0 has it.
template <> struct st_unit<ST_MXFP4_B10> {
    static constexpr int UB = 170;
    struct wfrag { uint32_t d[44]; };
    static __device__ __forceinline__ wfrag load(uint32_t a) {
        wfrag w; uint32_t r[45];
        const uint32_t a4 = a & ~3u, sh = (a & 2u)*8;
#pragma unroll
        for (int j = 0; j < 45; j++) r[j] = st_ld4(a4 + 4*j);
#pragma unroll
        for (int j = 0; j < 44; j++) w.d[j] = __builtin_amdgcn_alignbit(r[j + 1], r[j], sh);
        return w;
    }
    static __device__ __forceinline__ float dot(const wfrag & w, const char * ab, float) {
        float acc = 0.0f;
#pragma unroll
        for (int j = 0; j < 10; j++) {
            const int o = 17*j;
            const uint32_t e = (w.d[o >> 2] >> (8*(o & 3))) & 0xFF;
            const int q = (o + 1) >> 2, bs = (o + 1) & 3;
            float sf = 0.0f;
#pragma unroll
            for (int i = 0; i < 4; i++) {
                const uint32_t qs = bs ? __builtin_amdgcn_alignbyte(w.d[q + i + 1], w.d[q + i], bs) : w.d[q + i];
                const int4v A = *(const int4v *) (ab + 64*j + 16*i);      // the pairs (a_e, a_{e+16}) of elements e = 4 i .. 4 i + 3
                // (scalars first: __builtin_bit_cast applied directly to an element of an ext-vector reads element 0 with this compiler, ROCm 7.2 clang)
                const int a0 = A.x, a1 = A.y, a2 = A.z, a3 = A.w;
                sf = __builtin_amdgcn_fdot2(__builtin_amdgcn_cvt_scalef32_pk_f16_fp4(qs, 1.0f, 0), __builtin_bit_cast(st_h2, a0), sf, false);
                sf = __builtin_amdgcn_fdot2(__builtin_amdgcn_cvt_scalef32_pk_f16_fp4(qs, 1.0f, 1), __builtin_bit_cast(st_h2, a1), sf, false);
                sf = __builtin_amdgcn_fdot2(__builtin_amdgcn_cvt_scalef32_pk_f16_fp4(qs, 1.0f, 2), __builtin_bit_cast(st_h2, a2), sf, false);
                sf = __builtin_amdgcn_fdot2(__builtin_amdgcn_cvt_scalef32_pk_f16_fp4(qs, 1.0f, 3), __builtin_bit_cast(st_h2, a3), sf, false);
            }
            acc += (((const float *) (ab + 640))[j]*e8m0_to_f32_half(e))*(2.0f*sf);      // 2 sf = sumi of ggml_vec_dot_mxfp4_q8_0, exactly
        }
        return acc;
    }
};

constexpr int ST_NSTAMP = 16;      // stamps per wave (diagnostic builds)
#ifdef MI_STAMPS
#define ST_STAMP(i_) do { if (stamps && lane == 0) __hip_atomic_store(&stamps[((size_t) blockIdx.x*(ST_NC + 1) + wave)*ST_NSTAMP + (i_)], (unsigned long long) __builtin_amdgcn_s_memrealtime(), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); } while (0)
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


// where a workgroup's LDS regions are
struct st_lds {
    uint32_t * sync; char * act; float * dd; float * red; float * part; uint32_t ring_a; int slot_stride, S;
};
// rows of workgroup wg of nwg in group g: [r0, r0 + R), dealt in multiples of ralign (which also keeps every workgroup's first byte 16-byte
// aligned — Q6_K rows are 210 nb bytes); the rows past the last whole unit belong to the last workgroup
static __device__ __forceinline__ void st_rows(const st_group & g, int wg, int nwg, int & r0, int & R) {
    if (g.neox2) { const int p0 = wg*g.neox2; r0 = (p0/g.neox_hh)*(2*g.neox_hh) + p0 % g.neox_hh; R = g.neox2; return; }      // first-half rows; the partners are neox_hh rows further
    // (32-bit: nwg * nru < 2^31 is checked on the host. As 64-bit divisions these two lines were ~240 scalar instructions in front of the loader's first request
    // and of every consumer's activation loads)
    const uint32_t nru = (uint32_t) g.m/(uint32_t) g.ralign;
    r0 = (int)(((uint32_t) wg*nru/(uint32_t) nwg)*(uint32_t) g.ralign);
    R = (wg == nwg - 1 ? g.m : (int)((((uint32_t) wg + 1u)*nru/(uint32_t) nwg)*(uint32_t) g.ralign)) - r0;
}

// ================= the loader's share of one phase: slots slot0 .. slot0 + nslots of the workgroup's slot sequence =================
// In flight: at most ST_INFLIGHT pieces behind the slot being issued (a wave counts at most 63 outstanding loads; a slot is at most 14).
// Which slots have LANDED follows from the piece count alone — after `s_waitcnt vmcnt(N)` all but the youngest N pieces are in LDS — and
// the pieces a slot ends at are kept per slot (cum[]: phases of a chain have slots of different sizes).
constexpr int ST_INFLIGHT = 36;
// Everything here is wave-uniform and must stay in scalar registers and out of memory: the loader is the critical path of the whole kernel
// (a version that kept a per-slot table — in LDS, then in an array the compiler moved to vector registers — and re-read the group's
// pointers from the kernel arguments every slot lost a sixth of the stream rate). So: slots of ONE phase have one size, and which of them
// have landed is arithmetic on the piece count; of the phase before, the same with its own size; anything older has landed for sure
// because a phase that issued fewer than ST_INFLIGHT pieces is followed by a full drain.
struct st_loader_state {
    int landed, pieces;                       // slots published; pieces issued so far
    int c_slot0, c_base, p_slot0, p_base, p_magic, c_pps;      // the current / the previous phase: its first slot, the pieces issued before it; 65536 / pieces-per-slot (rounded up) of the previous one
};
template <int PPS>
static __device__ __forceinline__ void st_loader_publish(const st_lds & L, st_loader_state & s, int issued_slots, int outstanding, int lane) {
    const int done = s.pieces - outstanding;
    int l;
    if (done >= s.c_base) l = s.c_slot0 + (done - s.c_base)/PPS;
    else if (done >= s.p_base) l = s.p_slot0 + (((done - s.p_base)*s.p_magic) >> 16);
    else l = s.p_slot0;
    l = min(l, issued_slots);
    if (l > s.landed) { s.landed = l; if (lane == 0) st_flag_st(&L.sync[0], (uint32_t) l); }
}
template <int TYPE, bool NT>
static __device__ __forceinline__ void st_loader_phase(const st_args & p, const st_group & g, int wg, int nwg, const st_lds & L, int slot0, st_loader_state & ls, int lane, int expert, int barrier_after = 0) {
    typedef st_unit<TYPE> U;
    constexpr int PPS = (64*U::UB + 1023)/1024;
    const int nb = p.nb;
    const bool GLU = g.epi == EPI_GLU || g.neox2 > 0;      // two row streams
    int r0, R; st_rows(g, wg, nwg, r0, R);
    const int n1 = R*nb, ns1 = (n1 + 63) >> 6, nslots = GLU ? 2*ns1 : ns1;
    const int S = L.S;
    const uint32_t voff = lane*16;
    // a new phase: the one before becomes "previous"; if it was short, nothing of it (or of anything older) may stay unpublished
    if (ls.pieces - ls.c_base < ST_INFLIGHT && ls.pieces > 0) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (slot0 > ls.landed) { ls.landed = slot0; if (lane == 0) st_flag_st(&L.sync[0], (uint32_t) slot0); }
    }
    ls.p_slot0 = ls.c_slot0; ls.p_base = ls.c_base; ls.c_slot0 = slot0; ls.c_base = ls.pieces; ls.c_pps = PPS;
    const long long row_off = (long long) r0*nb*U::UB;
    const long long e_off = g.eid ? (long long) expert*g.estride : 0;      // (the expert the router picked: read on the device — k_mmvq_stream requests it before the first barrier)
    const char * const w0 = g.W + e_off + row_off; const char * const w1 = GLU ? g.W2 + e_off + row_off : w0;
    // the tensor's last 16-byte chunk (the grid of chunks starts at this workgroup's first byte, which is 16-byte aligned), relative to that byte. A tensor
    // whose size is not a multiple of 16 ends inside that chunk: it is read in place — up to 15 bytes of the zeroed padding every quantized tensor of this
    // backend's buffers carries (MI_TENSOR_PAD) — and only chunks that start past the end are redirected to it
    const long long lim_all = (((long long) g.m*nb*U::UB - row_off - 1) & ~15ll);
    int ring_i = slot0 % S;
    for (int i = 0; i < nslots; i++) {
        const int gi = slot0 + i;                            // slot number in the workgroup's sequence
        const int si = GLU ? (i >= ns1) : 0, il = i - si*ns1;
        const char * gb = (si ? w1 : w0) + (long long) il*(64*U::UB);
        if (gi >= S) {
            // the ring slot must have been consumed; publish what is in flight first so that nobody waits for us meanwhile
            if (st_poll_ld(&L.sync[16 + ring_i]) < (uint32_t)(gi - S + 1)) {
                if (ls.landed < gi) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); ls.landed = gi; if (lane == 0) st_flag_st(&L.sync[0], (uint32_t) gi); }
                st_wait_ge(&L.sync[16 + ring_i], (uint32_t)(gi - S + 1));
            }
        }
        const uint32_t dst = L.ring_a + (uint32_t) ring_i*L.slot_stride;
        if (il == ns1 - 1) {       // the stream's last slot may reach past the end of the tensor
            const long long lim = lim_all - (long long) il*(64*U::UB) - ((si && g.neox2) ? (long long) g.neox_hh*nb*U::UB : 0);      // (the partner stream starts neox_hh rows further in)
            st_dma_slot_clamped<NT, PPS>(gb, voff, dst, (uint32_t)(lim < 0x7FFFFFFF ? lim : 0x7FFFFFFF));
        } else st_dma_slot<NT, PPS>(gb, voff, dst);
        ls.pieces += PPS;
        if (++ring_i == S) ring_i = 0;
        // (barrier_after > 0: the workgroup's first barrier — behind which the consumers' activation loads are queued — is met after this many slots are on their way)
        if (barrier_after > 0 && i + 1 == barrier_after) { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); __builtin_amdgcn_s_barrier(); }
        asm volatile("s_waitcnt vmcnt(%0)" :: "n"(ST_INFLIGHT) : "memory");
        if (ls.pieces > ST_INFLIGHT) st_loader_publish<PPS>(L, ls, gi + 1, ST_INFLIGHT, lane);
    }
    if (barrier_after > nslots) { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); __builtin_amdgcn_s_barrier(); }      // (fewer slots than that: the barrier is still met once)
    ls.p_magic = (65536 + PPS - 1)/PPS;       // (for the phase after this one)
}
// after the last phase: everything lands
static __device__ __forceinline__ void st_loader_drain(const st_lds & L, int nslots_total, st_loader_state & ls, int lane) {
    asm volatile("s_waitcnt vmcnt(18)" ::: "memory");
    { const int done = ls.pieces - 18; const int l = min(nslots_total, done >= ls.c_base ? ls.c_slot0 + (done - ls.c_base)/max(ls.c_pps, 1) : ls.c_slot0);
      if (l > ls.landed) { ls.landed = l; if (lane == 0) st_flag_st(&L.sync[0], (uint32_t) l); } }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (nslots_total > ls.landed) { ls.landed = nslots_total; if (lane == 0) st_flag_st(&L.sync[0], (uint32_t) nslots_total); }
}

// ---- the activation image of a phase ----
// PRO_Q8: a Q8_K image some earlier launch made (act_q8_carve layout), re-laid for the consumers
template <bool FIRST>
static __device__ __forceinline__ void st_prologue_q8(const st_args & p, const st_lds & L, int ctid) {
    const int nb = p.nb, nq = p.k >> 4;
    int4v areg[2], breg[2];
#pragma unroll
    for (int i = 0; i < 2; i++) { const int q = min(ctid + i*ST_NC*64, nq - 1); areg[i] = *(const int4v *) (p.a_qs + (size_t) q*16); }
    const int ibl = min(ctid, nb - 1);
    breg[0] = *(const int4v *) (p.a_bs + (size_t) ibl*16); breg[1] = *(const int4v *) (p.a_bs + (size_t) ibl*16 + 8);
    const float dreg = p.a_d[ibl];
    if (FIRST) { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); __builtin_amdgcn_s_barrier(); }
#pragma unroll
    for (int i = 0; i < 2; i++) {
        const int q = ctid + i*ST_NC*64;
        if (q < nq) { const int ib = q >> 4, c = q & 15; *(int4v *) (L.act + (size_t) ib*ST_ACT_STRIDE + c*16) = areg[i]; }
    }
    if (ctid < nb) {
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
        char * ab = L.act + (size_t) ctid*ST_ACT_STRIDE;
        *(int4v *) (ab + 256) = int4v{ (int) h32[0], (int) h32[1], (int) l32[0], (int) l32[1] };
        *(int4v *) (ab + 272) = int4v{ (int) h16[0], (int) h16[1], (int) h16[2], (int) h16[3] };
        *(int4v *) (ab + 288) = int4v{ (int) l16[0], (int) l16[1], (int) l16[2], (int) l16[3] };
        L.dd[ctid] = dreg;
    }
}
// PRO_QUANT / PRO_NORM: x (f32) -> [RMS_NORM * w ->] Q8_K blocks, quant_core.h's arithmetic. Consumer wave w owns the 256-element chunks
// w, w + 8, ...: NA of them, all quantized in straight-line code (a chunk past the end is a clamped duplicate that is not stored) so that
// the dependent chains of the wave-wide maxima and sums interleave — chunk after chunk behind a branch each cost ~0.5 us per chunk
// IMG: 0 = Q8_K blocks (K-quant weights); 1 = the workgroup's weights are Q8_0 in 256-weight units — the image is the CPU path's Q8_0 instead (32-element
// blocks, f16-rounded scales: quant_core.h); 2 = the same Q8_0 activation for the ten-block units (k % 256 != 0: the last 256-piece is partial)
template <int NA, bool FIRST, int IMG, bool NORM>
static __device__ __forceinline__ void st_prologue_f32(const st_args & p, const st_lds & L, int x_off, int seq, int & n_norm, int lane, int wave) {
    constexpr bool Q80 = IMG != 0;
    const int nchunk = p.nchunk;
    constexpr bool norm = NORM;        // (compile-time: see st_prologue_q8k16)
    float4v xv[NA], wv[NA];
#pragma unroll
    for (int i = 0; i < NA; i++) {
        const int c = min(wave + ST_NC*i, nchunk - 1);
        const bool in_k = IMG < 2 || c*256 + lane*4 < p.k;
        xv[i] = in_k && p.x ? *(const float4v *) (p.x + x_off + c*256 + lane*4) : float4v{ 0.0f, 0.0f, 0.0f, 0.0f };
        wv[i] = norm && in_k ? *(const float4v *) (p.norm_w + (size_t) c*256 + lane*4) : float4v{ 1.0f, 1.0f, 1.0f, 1.0f };
    }
    if constexpr (NA <= 2) {
        if (p.planes) {
            float4v pv[NA][8];
#pragma unroll
            for (int i = 0; i < NA; i++) {
                const int c = min(wave + ST_NC*i, nchunk - 1);
                const bool in_k = IMG < 2 || c*256 + lane*4 < p.k;
#pragma unroll
                for (int pl = 0; pl < 8; pl++) pv[i][pl] = in_k ? *(const float4v *) (p.planes + (size_t) min(pl, p.n_planes - 1)*p.plane_stride + (size_t) c*256 + lane*4) : float4v{ 0.0f, 0.0f, 0.0f, 0.0f };
            }
            // the planes' weights (MoE combine): every thread derives the same few numbers — probs[ids[u]], then the normalisation of k_moe_combine (elem.hip)
            float w[8];
            {
                // the weights' loads are REQUESTED here and the workgroup's first barrier follows at once: the loader stands at that barrier, and with the arithmetic
                // below (a wait for the cold values, eight expf, a division) in front of it the whole weight stream of the launch started ~2 us late
                float pr[8]; int idv[8];
#pragma unroll
                for (int u = 0; u < 8; u++) {
                    pr[u] = 0.0f; idv[u] = 0;
                    if (u < p.n_planes) { if (p.pl_ids) idv[u] = p.pl_ids[u]; else pr[u] = p.pl_probs[u]; }      // (pl_ids == NULL: the router's values in slot order)
                }
                if (FIRST) { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); __builtin_amdgcn_s_barrier(); }
                if (p.pl_ids) {
#pragma unroll
                    for (int u = 0; u < 8; u++) if (u < p.n_planes) pr[u] = p.pl_probs[idv[u]];
                }
                if (p.pl_mode == 0) {
                    float sum = 0.0f;
#pragma unroll
                    for (int u = 0; u < 8; u++) if (u < p.n_planes) sum += pr[u];
#pragma unroll
                    for (int u = 0; u < 8; u++) w[u] = pr[u]/sum;
                } else {
                    float mx = -INFINITY;
#pragma unroll
                    for (int u = 0; u < 8; u++) if (u < p.n_planes) mx = fmaxf(mx, pr[u]);
                    float sum = 0.0f;
#pragma unroll
                    for (int u = 0; u < 8; u++) { w[u] = u < p.n_planes ? expf(pr[u] - mx) : 0.0f; sum += w[u]; }
                    const float inv = 1.0f/sum;
#pragma unroll
                    for (int u = 0; u < 8; u++) w[u] *= inv;
                }
            }
#pragma unroll
            for (int i = 0; i < NA; i++) {
                {
                    float4v acc = pv[i][0];
                    acc.x *= w[0]; acc.y *= w[0]; acc.z *= w[0]; acc.w *= w[0];
#pragma unroll
                    for (int pl = 1; pl < 8; pl++) if (pl < p.n_planes) { acc.x += pv[i][pl].x*w[pl]; acc.y += pv[i][pl].y*w[pl]; acc.z += pv[i][pl].z*w[pl]; acc.w += pv[i][pl].w*w[pl]; }
                    if (p.x) { acc.x += xv[i].x; acc.y += xv[i].y; acc.z += xv[i].z; acc.w += xv[i].w; }
                    xv[i] = acc;
                }
                const int c = wave + ST_NC*i;
                if (blockIdx.x == 0 && c < nchunk && (IMG < 2 || c*256 + lane*4 < p.k)) *(float4v *) (p.x_out + (size_t) c*256 + lane*4) = xv[i];
            }
        } else if (FIRST) { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); __builtin_amdgcn_s_barrier(); }
    } else
    if (FIRST) { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); __builtin_amdgcn_s_barrier(); }
#ifdef MI_STAMPS
    {   // diagnostic builds: when are the activation loads back? (stamp 7 of the wave; the stamp waits for them, which the product does not do here)
        unsigned long long * stamps = p.stamps;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (FIRST) { ST_STAMP(7); }
    }
#endif
    float scale = 1.0f;
    if (norm) {
        float ss = 0.0f;
#pragma unroll
        for (int i = 0; i < NA; i++) if (wave + ST_NC*i < nchunk) ss += xv[i].x*xv[i].x + xv[i].y*xv[i].y + xv[i].z*xv[i].z + xv[i].w*xv[i].w;
        ss = wave_sum(ss);
        if (lane == 0) L.red[wave] = ss;      // (everybody has read the previous phase's sums: three meetings lie in between)
        st_consumers_meet(&L.sync[4], lane, n_norm++);
        const float * red = L.red;
        ss = ((red[0] + red[1]) + (red[2] + red[3])) + ((red[4] + red[5]) + (red[6] + red[7]));
        scale = 1.0f/sqrtf(ss/(float) p.k + p.eps);
    }
    uint32_t q4[NA]; float d8[NA]; int bs16[NA];
#pragma unroll
    for (int i = 0; i < NA; i++) {
        float4v v = xv[i];
        if (norm) { v.x = (v.x*scale)*wv[i].x; v.y = (v.y*scale)*wv[i].y; v.z = (v.z*scale)*wv[i].z; v.w = (v.w*scale)*wv[i].w; }
        q4[i] = Q80 ? quant_frag_q8_0(v, d8[i], bs16[i]) : quant_frag_q8_K(v, d8[i], bs16[i]);
    }
#pragma unroll
    for (int i = 0; i < NA; i++) {
        const int c = wave + ST_NC*i;
        if (c < nchunk) {
            if (IMG == 3) {
                // f16 pairs (a_i, a_{i+16}): the lane 4 further holds the elements 16 further of the same 32-block
                const uint32_t other = (uint32_t) __builtin_amdgcn_ds_bpermute((lane ^ 4) << 2, (int) q4[i]);
                const int e = c*256 + lane*4;
                if (e < p.k) {
                    const int un = e/320, off = e - un*320, blk = off >> 5, e32 = off & 31;
                    char * ab = L.act + (size_t) un*ST_ACT_STRIDE_FP4;
                    if (e32 < 16) {
                        uint32_t w4[4];
#pragma unroll
                        for (int t = 0; t < 4; t++) {
                            const st_h2 pr = { (_Float16)(int)(int8_t)(q4[i] >> (8*t)), (_Float16)(int)(int8_t)(other >> (8*t)) };
                            w4[t] = __builtin_bit_cast(uint32_t, pr);
                        }
                        *(int4v *) (ab + 64*blk + 4*e32) = int4v{ (int) w4[0], (int) w4[1], (int) w4[2], (int) w4[3] };
                    }
                    if ((lane & 7) == 0) ((float *) (ab + 640))[blk] = d8[i];
                }
                continue;
            }
            if (IMG == 2) {
                const int e = c*256 + lane*4;
                if (e < p.k) {
                    const int un = e/320, off = e - un*320;
                    char * ab = L.act + (size_t) un*ST_ACT_STRIDE_B10;
                    *(uint32_t *) (ab + off) = q4[i];
                    if ((lane & 7) == 0) ((float *) (ab + 320))[off >> 5] = d8[i];
                }
                continue;
            }
            char * ab = L.act + (size_t) c*ST_ACT_STRIDE;
            *(uint32_t *) (ab + lane*4) = q4[i];
            if (Q80) { if ((lane & 7) == 0) { ((float *) (ab + 256))[lane >> 3] = d8[i]; ((int16_t *) (ab + 288))[lane >> 3] = (int16_t) bs16[i]; } continue; }      // (the sums: Q4_0's - 8 offset)
            // the 16-element sum of quad q = lane >> 2 (valid in its four lanes); the 32-element sum j at lane 8j + 4 (row_shr:4 brings lane 8j's)
            const int bs32 = bs16[i] + dpp_i<0x114>(bs16[i]);
            int h, l;
            st_hl(bs16[i], h, l);
            if ((lane & 3) == 0) { ab[272 + (lane >> 2)] = (char) h; ab[288 + (lane >> 2)] = (char) l; }
            st_hl(bs32, h, l);
            if ((lane & 7) == 4) { ab[256 + (lane >> 3)] = (char) h; ab[264 + (lane >> 3)] = (char) l; }
            if (lane == 0) L.dd[c] = d8[i];
        }
    }
}

// PRO_QUANT / PRO_NORM into Q8_K blocks, FOUR blocks per wave-instruction (round 4). st_prologue_f32 gives a wave one 256-element block at a time, four floats per
// lane: per block a full-wave maximum (DPP row steps + readlanes), a ballot, a readlane, two correctly rounded divisions, the quad sums and seven masked byte
// stores serve just four elements per lane — ~100 vector instructions per block, two consumer waves per SIMD: the image of a 14336-long vector was ready 4.7 us
// into the launch with the loads back after 2.0 (stamps build), i.e. the quantizer, not memory, set the start of every launch's multiply phase.
// Here a DPP ROW (16 lanes) owns a block and a lane 16 consecutive elements: the maximum is four row steps, the 16-element sums are per lane, a lane stores its
// 16 quants as one 16-byte write, and every overhead instruction works for four blocks at once. Same arithmetic per element as quant_core.h's quant_frag_q8_K
// (first element of largest magnitude, iscale = -127 / max, round to nearest even, min 127, d = 1 / iscale): the same bytes.
//   quad q = blocks 4q .. 4q + 3; wave w owns quads w, w + 8, ...: NQ of them
template <int NQ, bool FIRST, bool NORM>
static __device__ __forceinline__ void st_prologue_q8k16(const st_args & p, const st_lds & L, int x_off, int seq, int & n_norm, int lane, int wave) {
    const int nchunk = p.nchunk;
    constexpr bool norm = NORM;      // (compile-time: as a run-time flag both forms were computed and selected between, square root and all)
    const int r = lane >> 4, l16 = lane & 15;
    float4v xv[NQ][4], wv[NQ][4];
#pragma unroll
    for (int i = 0; i < NQ; i++) {
        const int c = min(4*(wave + ST_NC*i) + r, nchunk - 1);       // (a block past the end: a clamped duplicate that is not stored)
        const float * xs = p.x + x_off + c*256 + l16*16;
#pragma unroll
        for (int j = 0; j < 4; j++) {
            xv[i][j] = *(const float4v *) (xs + 4*j);
            wv[i][j] = norm ? *(const float4v *) (p.norm_w + (size_t) c*256 + l16*16 + 4*j) : float4v{ 1.0f, 1.0f, 1.0f, 1.0f };
        }
    }
    if (FIRST) { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); __builtin_amdgcn_s_barrier(); }
#ifdef MI_STAMPS
    {   unsigned long long * stamps = p.stamps;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (FIRST) { ST_STAMP(7); }
    }
#endif
    float scale = 1.0f;
    if (norm) {
        float ss = 0.0f;
#pragma unroll
        for (int i = 0; i < NQ; i++) {
            if (4*(wave + ST_NC*i) + r < nchunk) {
#pragma unroll
                for (int j = 0; j < 4; j++) ss += (xv[i][j].x*xv[i][j].x + xv[i][j].y*xv[i][j].y) + (xv[i][j].z*xv[i][j].z + xv[i][j].w*xv[i][j].w);
            }
        }
        ss = wave_sum(ss);
        if (lane == 0) L.red[wave] = ss;
        st_consumers_meet(&L.sync[4], lane, n_norm++);
        const float * red = L.red;
        ss = ((red[0] + red[1]) + (red[2] + red[3])) + ((red[4] + red[5]) + (red[6] + red[7]));
        scale = 1.0f/sqrtf(ss/(float) p.k + p.eps);
    }
#pragma unroll
    for (int i = 0; i < NQ; i++) {
        const int c = 4*(wave + ST_NC*i) + r;
        float v[16];
#pragma unroll
        for (int j = 0; j < 4; j++) {
            v[4*j] = xv[i][j].x; v[4*j + 1] = xv[i][j].y; v[4*j + 2] = xv[i][j].z; v[4*j + 3] = xv[i][j].w;
            if (norm) { v[4*j] = (v[4*j]*scale)*wv[i][j].x; v[4*j + 1] = (v[4*j + 1]*scale)*wv[i][j].y; v[4*j + 2] = (v[4*j + 2]*scale)*wv[i][j].z; v[4*j + 3] = (v[4*j + 3]*scale)*wv[i][j].w; }
        }
        // the first element of largest magnitude (the reference's strict > keeps the first) is +max or -max: only its SIGN has to be found. A lane's largest and
        // smallest element (two max trees) say whether +max, -max or both occur among its 16; both in one lane (never with real data, but exact all the same)
        // takes the scan — lowest index wins — behind a wave-uniform branch
        float pmax = fmaxf(fmaxf(fmaxf(v[0], v[1]), fmaxf(v[2], v[3])), fmaxf(fmaxf(v[4], v[5]), fmaxf(v[6], v[7])));
        pmax = fmaxf(pmax, fmaxf(fmaxf(fmaxf(v[8], v[9]), fmaxf(v[10], v[11])), fmaxf(fmaxf(v[12], v[13]), fmaxf(v[14], v[15]))));
        float nmin = fminf(fminf(fminf(v[0], v[1]), fminf(v[2], v[3])), fminf(fminf(v[4], v[5]), fminf(v[6], v[7])));
        nmin = fminf(nmin, fminf(fminf(fminf(v[8], v[9]), fminf(v[10], v[11])), fminf(fminf(v[12], v[13]), fminf(v[14], v[15]))));
        const float amax = fmaxf(pmax, -nmin);
        const float rmax = row16_max(amax);
        const bool has_pos = pmax == rmax, has_neg = -nmin == rmax;
        float mx = has_neg ? -rmax : rmax;          // (a lane without the maximum: unused)
        if (__ballot(has_pos && has_neg)) {
            mx = v[15];
#pragma unroll
            for (int e = 14; e >= 0; e--) mx = fabsf(v[e]) == rmax ? v[e] : mx;
        }
        const bool zero = rmax == 0.0f;
        const unsigned long long ball = __ballot(amax == rmax);
        const uint32_t mine = (uint32_t)(ball >> (lane & 48)) & 0xFFFFu;       // the row's lanes that hold the maximum: the lowest one holds the first such element
        const int first = (lane & 48) + (int) __builtin_ctz(mine | 0x10000u);
        const float got = __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(first << 2, __builtin_bit_cast(int, mx)));
        const float maxv = zero ? 1.0f : got;
#ifdef MI_STAMPS
        { unsigned long long * stamps = p.stamps; if (FIRST && i == 0) { asm volatile("" :: "v"(maxv)); ST_STAMP(8); } }
#endif
        const float iscale = -127.0f/maxv;
        // round to nearest even by the reference's own device (nearest_int: add 1.5 * 2^23, the integer sits in the low mantissa bits): the quant is the low BYTE of
        // the sum's bit pattern, so four of them are packed by byte selects without a conversion, and their sum is one signed dot4 against 1,1,1,1. (The reference's
        // MIN(127, .) never binds: |iscale * x| <= 127 (1 + 2^-24)^2, which rounds to 127.)
        uint32_t tb[16];
#pragma unroll
        for (int e = 0; e < 16; e++) { const float t = iscale*v[e] + 12582912.0f; tb[e] = __builtin_bit_cast(uint32_t, t); }
        int4v pk; int sum = 0;
#pragma unroll
        for (int j = 0; j < 4; j++) {
            uint32_t w = __builtin_amdgcn_perm(tb[4*j + 1], tb[4*j], 0x0c0c0400u);          // { t0.b0, t1.b0, 0, 0 }
            w = __builtin_amdgcn_perm(tb[4*j + 2], w, 0x0c040100u);                          // { .., .., t2.b0, 0 }
            w = __builtin_amdgcn_perm(tb[4*j + 3], w, 0x04020100u);                          // { .., .., .., t3.b0 }
            pk[j] = (int) w;
            sum = __builtin_amdgcn_sdot4((int) w, 0x01010101, sum, false);
        }
        if (zero) { pk = int4v{ 0, 0, 0, 0 }; sum = 0; }
        const float d = zero ? 0.0f : 1.0f/iscale;
        const int sum32 = sum + dpp_i<0xB1>(sum);       // (quad_perm [1,0,3,2]: the neighbour's 16-element sum)
#ifdef MI_STAMPS
        { unsigned long long * stamps = p.stamps; if (FIRST && i == 0) { asm volatile("" :: "v"(pk.x), "v"(sum32), "v"(d)); ST_STAMP(9); } }
#endif
        if (c < nchunk) {
            char * ab = L.act + (size_t) c*ST_ACT_STRIDE;
            *(int4v *) (ab + l16*16) = pk;
            int h, l;
            st_hl(sum, h, l);
            ab[272 + l16] = (char) h; ab[288 + l16] = (char) l;
            st_hl(sum32, h, l);
            if ((l16 & 1) == 0) { ab[256 + (l16 >> 1)] = (char) h; ab[264 + (l16 >> 1)] = (char) l; }
            if (l16 == 0) L.dd[c] = d;
        }
    }
}

// ================= the consumers' share of one phase =================
//   seq: how many phases this workgroup has run before (its LDS counters are cumulative); FIRST: the launch's first phase — the activation
//   loads are queued before the loader starts (the barrier every wave of the workgroup takes exactly once)
template <int TYPE, bool FIRST>
static __device__ __forceinline__ void st_consumer_phase(const st_args & p, const st_group & g, int wg, int nwg, const st_lds & L, int slot0, int seq, int & n_norm,
                                                         int lane, int wave, unsigned long long * stamps, int xt0 = 0, int xt1 = 0) {
    typedef st_unit<TYPE> U;
    const int nb = p.nb;
    const bool GLU = g.epi == EPI_GLU || g.neox2 > 0;      // two row streams
    int r0, R; st_rows(g, wg, nwg, r0, R);
    const int n1 = R*nb, ns1 = (n1 + 63) >> 6, nslots = GLU ? 2*ns1 : ns1;
    const int S = L.S;
    const bool row16 = (nb & 15) == 0;               // a DPP row of 16 lanes = 16 units of ONE weight row
    const int npr = row16 ? nb >> 4 : nb;            // partials per row
    uint32_t * sync = L.sync; char * act = L.act; float * dd = L.dd; float * part = L.part;
    const int ctid = threadIdx.x;                    // consumer thread id (the consumers are waves 0 .. ST_NC - 1)

    // ---- the activation image (FIRST: the loads are requested before any weight is — a CU returns loads in request order) ----
    const int mode = p.mode;
    constexpr int ST_IMG = TYPE == ST_MXFP4_B10 ? 3 : TYPE == ST_Q8_0_B10 ? 2 : (TYPE == T_Q8_0 || TYPE == T_Q4_0) ? 1 : 0;
    if (mode == PRO_Q8) st_prologue_q8<FIRST>(p, L, ctid);
    else if (ST_IMG == 0 && !p.planes && p.x && !(p.early & 0x100) && p.nchunk <= 128) {      // (bit 8 of `early`: GGML_MI355X_STREAM_Q16=0, the one-block-per-wave quantizer)
        if (mode == PRO_NORM) {
            if (p.nchunk <= 32)      st_prologue_q8k16<1, FIRST, true>(p, L, g.x_off, seq, n_norm, lane, wave);
            else if (p.nchunk <= 64) st_prologue_q8k16<2, FIRST, true>(p, L, g.x_off, seq, n_norm, lane, wave);
            else                     st_prologue_q8k16<4, FIRST, true>(p, L, g.x_off, seq, n_norm, lane, wave);
        } else {
            if (p.nchunk <= 32)      st_prologue_q8k16<1, FIRST, false>(p, L, g.x_off, seq, n_norm, lane, wave);
            else if (p.nchunk <= 64) st_prologue_q8k16<2, FIRST, false>(p, L, g.x_off, seq, n_norm, lane, wave);
            else                     st_prologue_q8k16<4, FIRST, false>(p, L, g.x_off, seq, n_norm, lane, wave);
        }
    }
    else if (mode == PRO_NORM) {
        if (p.nchunk <= 8)        st_prologue_f32<1, FIRST, ST_IMG, true>(p, L, g.x_off, seq, n_norm, lane, wave);
        else if (p.nchunk <= 16)  st_prologue_f32<2, FIRST, ST_IMG, true>(p, L, g.x_off, seq, n_norm, lane, wave);
        else if (p.nchunk <= 32)  st_prologue_f32<4, FIRST, ST_IMG, true>(p, L, g.x_off, seq, n_norm, lane, wave);
        else                      st_prologue_f32<8, FIRST, ST_IMG, true>(p, L, g.x_off, seq, n_norm, lane, wave);
    } else {
        if (p.nchunk <= 8)        st_prologue_f32<1, FIRST, ST_IMG, false>(p, L, g.x_off, seq, n_norm, lane, wave);
        else if (p.nchunk <= 16)  st_prologue_f32<2, FIRST, ST_IMG, false>(p, L, g.x_off, seq, n_norm, lane, wave);
        else if (p.nchunk <= 32)  st_prologue_f32<4, FIRST, ST_IMG, false>(p, L, g.x_off, seq, n_norm, lane, wave);
        else                      st_prologue_f32<8, FIRST, ST_IMG, false>(p, L, g.x_off, seq, n_norm, lane, wave);
    }
    ST_STAMP(1);
    asm volatile("s_waitcnt vmcnt(0)" :: "v"(xt0), "v"(xt1) : "memory");      // (the entry touches' registers are free from here on; nothing of this wave's is in flight any more)
    st_consumers_meet(&sync[2], lane, seq);
    ST_STAMP(2);

    // ---- the epilogue's operands of this thread's first row / pair: requested now, needed after the last slot ----
    float e_r0 = 0.0f, e_r1 = 0.0f, e_q0 = 0.0f, e_c = 1.0f, e_s = 0.0f; long long e_i0 = 0, e_i1 = 0;
    // row offset into the [m, n_expert] bias tables of gpt-oss's experts (the expert index is a device value)
    const int e_tab = (g.eid && (g.res_eid || g.b_gate)) ? __builtin_amdgcn_readfirstlane(g.eid[0])*g.m : 0;
    const long long idx0 = g.st_mode == 1 ? g.st_idx[0] : 0;
    if (g.epi == EPI_ROPE) {
        const fused_rope & rp = p.rope;
        const int pr = ctid, hd = rp.head_dim;
        if (pr < (g.neox2 ? R : (R >> 1))) {
            int ra, rb, ip;
            if (g.neox2)      { ra = pr; rb = pr + g.neox_hh; ip = (r0 + pr) % hd; }      // (rows relative to r0; the partner was streamed through W2)
            else if (rp.neox) { const int hh = pr/(hd >> 1), i = pr - hh*(hd >> 1); ra = hh*hd + i; rb = ra + (hd >> 1); ip = i; }
            else              { ra = 2*pr; rb = ra + 1; ip = ((r0 + ra) % hd) >> 1; }
            if (g.res) { e_r0 = g.res[r0 + ra]; e_r1 = g.res[r0 + rb]; }
            if (ip < (rp.n_dims >> 1)) { e_c = rp.tab[2*ip]; e_s = rp.tab[2*ip + 1]; }
            if (g.st_mode == 2) { e_i0 = g.st_idx[r0 + ra]; e_i1 = g.st_idx[r0 + rb]; }
        }
    } else if (ctid < R) {
        const int row = r0 + ctid;
        if (g.epi == EPI_ADD) { e_r0 = g.res[e_tab + row]; if (g.res2) e_q0 = g.res2[row]; }
        if (g.epi == EPI_GLU && g.b_gate) { e_r0 = g.b_gate[e_tab + row]; e_q0 = g.b_up[e_tab + row]; }
        if (g.st_mode == 2) e_i0 = g.st_idx[row];
    }

    // ---- the stream ----
    const uint32_t magic = p.magic;
    bool first = true;
    int ring_i = (slot0 + wave) % S;
    for (int i = wave; i < nslots; i += ST_NC) {
        const int gi = slot0 + i;
        const int si = GLU ? (i >= ns1) : 0, il = i - si*ns1;
        const int u = il*64 + lane;                              // unit inside the stream
        const bool live = u < n1;
        const int uc = live ? u : n1 - 1;
        const int ib = nb == 1 ? 0 : uc - (int) __umulhi((uint32_t) uc, magic)*nb;      // (the magic number of nb = 1 does not fit 32 bits)
        const char * ab = act + (size_t) ib*(ST_IMG == 3 ? ST_ACT_STRIDE_FP4 : ST_IMG == 2 ? ST_ACT_STRIDE_B10 : ST_ACT_STRIDE);
        const float d8 = dd[ib];
        st_wait_ge(&sync[0], (uint32_t)(gi + 1));
        if (first) { ST_STAMP(3); first = false; }
        const typename U::wfrag w = U::load(L.ring_a + (uint32_t) ring_i*L.slot_stride + (uint32_t)(live ? lane : 0)*U::UB);
        // the slot is free as soon as its bytes are in registers
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if (lane == 0) st_flag_st(&sync[16 + ring_i], (uint32_t)(gi + 1));
        ring_i += ST_NC; while (ring_i >= S) ring_i -= S;
        float res = U::dot(w, ab, d8);
        if (!live) res = 0.0f;
        if (row16) {
            res = row16_sum(res);
            if ((lane & 15) == 0 && live) part[(size_t) si*R*npr + (u >> 4)] = res;
        } else if (live) part[(size_t) si*n1 + u] = res;
    }
    ST_STAMP(4);
    st_consumers_meet(&sync[3], lane, 2*seq);
    ST_STAMP(5);

    // ---- rows: partials added in a fixed order, epilogue with a lane per row (or rotation pair) ----
    const float * part2 = part + (size_t) R*npr;
    if (g.epi == EPI_ROPE) {
        const fused_rope & rp = p.rope;
        const int hd = rp.head_dim, half = rp.n_dims >> 1;
        for (int pr = ctid; pr < (g.neox2 ? R : (R >> 1)); pr += ST_NC*64) {
            // the two rows of pair pr (local): NORM (2 pr, 2 pr + 1); NEOX: i and i + hd/2 inside one head (ralign = hd), or — two row streams — row pr of each stream
            int ra, rb, ip;
            if (g.neox2)      { ra = pr; rb = pr + g.neox_hh; ip = (r0 + pr) % hd; }
            else if (rp.neox) { const int hh = pr/(hd >> 1), i = pr - hh*(hd >> 1); ra = hh*hd + i; rb = ra + (hd >> 1); ip = i; }
            else              { ra = 2*pr; rb = ra + 1; ip = ((r0 + ra) % hd) >> 1; }
            if (pr != ctid) {       // (not the prefetched pair: a workgroup with more than 1024 rotated rows)
                e_r0 = e_r1 = 0.0f; e_c = 1.0f; e_s = 0.0f;
                if (g.res) { e_r0 = g.res[r0 + ra]; e_r1 = g.res[r0 + rb]; }
                if (ip < half) { e_c = rp.tab[2*ip]; e_s = rp.tab[2*ip + 1]; }
                if (g.st_mode == 2) { e_i0 = g.st_idx[r0 + ra]; e_i1 = g.st_idx[r0 + rb]; }
            }
            float s0 = st_row_sum(part, ra, npr), s1 = g.neox2 ? st_row_sum(part2, pr, npr) : st_row_sum(part, rb, npr);
            if (g.res) { s0 += e_r0; s1 += e_r1; }              // bias first, then the rotation
            if (ip < half) { const float a = s0, b = s1; s0 = a*e_c - b*e_s; s1 = a*e_s + b*e_c; }
            g.dst[r0 + ra] = s0; g.dst[r0 + rb] = s1;
            if (g.st_mode == 1) { uint16_t * q = g.st16 + idx0*g.st_row_elems; q[r0 + ra] = f32_to_f16_bits(s0); q[r0 + rb] = f32_to_f16_bits(s1); }
            else if (g.st_mode == 2) { g.st16[e_i0] = f32_to_f16_bits(s0); g.st16[e_i1] = f32_to_f16_bits(s1); }
        }
    } else {
        for (int rr = ctid; rr < R; rr += ST_NC*64) {
            float s0 = st_row_sum(part, rr, npr);
            const int row = r0 + rr;
            if (rr != ctid) {
                if (g.epi == EPI_ADD) { e_r0 = g.res[e_tab + row]; e_q0 = g.res2 ? g.res2[row] : 0.0f; }
                if (g.epi == EPI_GLU && g.b_gate) { e_r0 = g.b_gate[e_tab + row]; e_q0 = g.b_up[e_tab + row]; }
                if (g.st_mode == 2) e_i0 = g.st_idx[row];
            }
            if (GLU) {
                float up_s = st_row_sum(part2, rr, npr);
                if (g.b_gate) { s0 += e_r0; up_s += e_q0; }      // ADD_ID on both products, then the activation
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

// which group of the phase a workgroup belongs to
static __device__ __forceinline__ int st_group_of(const st_args & p, int b, int & first, int & nwg) {
    int gi = 0;
#pragma unroll
    for (int q = 0; q < MMVQ_MAX_GROUPS - 1; q++) if (b >= p.block_end[q]) gi = q + 1;
    first = gi ? p.block_end[gi - 1] : 0; nwg = p.block_end[gi] - first;
    return gi;
}
static __device__ __forceinline__ st_lds st_carve(char * lds, int nb_max, int npart_max, int slot_stride, int S, int act_stride = ST_ACT_STRIDE) {
    st_lds L;
    L.sync = (uint32_t *) lds;
    L.act = lds + 2*ST_SYNC_WORDS*4;
    L.dd = (float *) (L.act + (size_t) nb_max*act_stride);
    L.red = L.dd + ((nb_max + 3) & ~3);              // [2][ST_NC] sums of squares (PRO_NORM)
    L.part = L.red + 16;
    L.ring_a = st_lds_addr((char *) (((size_t)(L.part + npart_max) + 15) & ~(size_t) 15));
    L.slot_stride = slot_stride; L.S = S;
    return L;
}

static __device__ __forceinline__ int st_phase_slots(const st_args & a, const st_group & g, int wg, int nwg) {
    int r0, R; st_rows(g, wg, nwg, r0, R);
    const int ns1 = (R*a.nb + 63) >> 6;
    return (g.epi == EPI_GLU || g.neox2 > 0) ? 2*ns1 : ns1;
}

// ---- one grouped launch ----
// GLU (compile-time, single-group launches only): the gate/up/SwiGLU launch — the dominant kernel of a decode step — is an instantiation
// of its own, so that kernel traces (rocprofv3) tell it from the other mat-vecs of the same weight format
template <int TA, int TB, bool NT, bool GLU>
__global__ void __launch_bounds__(ST_THREADS, 3) k_mmvq_stream(const st_args p) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    // The activation vector (and the norm weights) are what every launch waits for first — written by the launch before on other XCDs, back ~1.5 us after they are
    // asked for — and the loads that ask for them sit behind this kernel's whole set-up (group lookup, LDS carve, row ranges: several hundred instructions and a
    // chain of scalar loads). One dword per 128-byte line, asked for here from two kernel-argument fields, starts the fetch; the real loads find the lines on their way.
    // (the destination registers stay reserved until the image is built — in-order return: the real loads' data is back only after these — or the compiler would
    // hand them to something else while the loads are still to write them: the first version of this hung the GPU box that way)
    int xt0 = 0, xt1 = 0;
    if (!(p.early & 0x200) && p.mode != PRO_Q8 && p.x) {
        const uint32_t off = threadIdx.x*128u;
        if (off < (uint32_t) p.k*4u) {
            asm volatile("global_load_dword %0, %1, off" : "+v"(xt0) : "v"((const char *) p.x + off) : "memory");
            if (p.mode == PRO_NORM) asm volatile("global_load_dword %0, %1, off" : "+v"(xt1) : "v"((const char *) p.norm_w + off) : "memory");
        }
    }
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    int first, nwg;
    const int gi = st_group_of(p, (int) blockIdx.x, first, nwg);
    const st_group & g = p.g[gi];
    if (GLU) __builtin_assume(g.epi == EPI_GLU);
    const int wg = (int) blockIdx.x - first;
    const bool is_a = TA == TB || g.type == TA;
    const st_lds L = st_carve(lds, p.nb, g.npart_max, ((64*(is_a ? st_unit<TA>::UB : st_unit<TB>::UB) + 1023)/1024)*1024, p.S, p.act_stride);
    unsigned long long * stamps = p.stamps;
    ST_STAMP(0);
    if (threadIdx.x < ST_SYNC_WORDS) L.sync[threadIdx.x] = 0;
    if (wave == ST_NC) {
        // an expert stack: which expert — a value the router launch has just written, i.e. a load that misses this CU's caches — is requested BEFORE the barrier
        // the loader shares with the consumers' activation loads and collected after it (the first DMA used to wait ~1 us for it behind that barrier)
        int e_raw = 0;
        if (g.eid) asm volatile("global_load_dword %0, %1, off" : "=v"(e_raw) : "v"(g.eid) : "memory");
        // weights first (p.early slots, plain groups only): a CU returns its loads in request order, so with the activations in front the first slot lands ~1 us after
        // THEY do (2 - 4 us into the launch: they are cold in another XCD's L2); a slot in front of them delays them by its ~0.3 us and is there when the image is
        const int early = g.eid ? 0 : (p.early & 0xFF);
        if (!early) { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); __builtin_amdgcn_s_barrier(); }
        if (g.eid) { asm volatile("s_waitcnt vmcnt(0)" : "+v"(e_raw) :: "memory"); __builtin_amdgcn_sched_barrier(0); }
        const int expert = __builtin_amdgcn_readfirstlane(e_raw);
        ST_STAMP(3);        // (loader: the expert index is here — or nothing was waited for)
        st_loader_state ls = { 0, 0, 0, 0, 0, 0, 0, 0 };
        if (is_a) st_loader_phase<TA, NT>(p, g, wg, nwg, L, 0, ls, lane, expert, early);
        else      st_loader_phase<TB, NT>(p, g, wg, nwg, L, 0, ls, lane, expert, early);
        st_loader_drain(L, st_phase_slots(p, g, wg, nwg), ls, lane);
        ST_STAMP(1);
        return;
    }
    int n_norm = 0;
    if (is_a) st_consumer_phase<TA, true>(p, g, wg, nwg, L, 0, 0, n_norm, lane, wave, stamps, xt0, xt1);
    else      st_consumer_phase<TB, true>(p, g, wg, nwg, L, 0, 0, n_norm, lane, wave, stamps, xt0, xt1);
}

} // namespace mi355x
