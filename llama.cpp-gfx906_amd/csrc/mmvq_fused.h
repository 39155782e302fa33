// mmvq_fused.h — the PERSISTENT grouped quantized mat-vec (n = 1) with prologues and epilogues: device code shared by the
// per-format translation units (mmvq_fused_<type>.hip, one each so that they compile in parallel) and the host side (mmvq_fused.hip).
//
// A launch has ONE workgroup of 8 (or 16) waves per CU; each workgroup belongs to one group (weight tensor) and its waves walk that
// tensor's row pairs (single rows for the dual GLU stream) with a grid stride, so that
//   * the activation is prepared ONCE per workgroup (copy / quantize / rms-norm + quantize into LDS) instead of once per 8 rows,
//   * the packed-weight stream never stops: loads run D steps ahead across row boundaries in a STATIC ring of register sets (the
//     loop is unrolled D times; a rotating copy w0 = w1 makes the compiler wait for every outstanding load at the top of each step,
//     measured: tools/stamp_timeline.py), and the DPP reduction + epilogue of one row pair overlaps the loads of the next.
// Order of issue: (1) activation loads by every wave, workgroup barrier (a CU returns loads in request order: nothing HBM-bound
// may be queued in front of them), (2) norm weights, then the D steps of weight loads one at a time BETWEEN the phases of
// (3) the prologue into LDS (a wave that cannot queue a load cannot do its share of the prologue either) + barrier,
// (4) integer dots, (5) reduction + epilogue per row pair.
// Every load is unconditional (clamped address) so that the number of outstanding loads is the same on every path.
//
// Head latency (round 2, tools/stamp_timeline.py + the ISA): the first version reached its activation loads through ~10 DEPENDENT
// scalar loads of the kernel-argument block (group lookup loop, per-group descriptor, type switch, eid / pos / st_idx pointer chases,
// each behind its own s_waitcnt) — 1.3-1.5 us before the first vector load. Now everything the head needs sits in a flat header that
// arrives with the first scalar-load batch (group of this workgroup = compares + selects, no memory), the per-group descriptor and the
// pointer chases overlap the activation round trip, and the operands of a row pair's epilogue (residual, KV-cache indices, rope
// frequency factor) are requested when the pair STARTS instead of after its last dot product (each was a dependent global load
// at the tail of the launch: the V-cache element scatter alone held the norm+QKV launch 3.5 us after every other workgroup had finished).
#pragma once
#include <hip/hip_ext.h>

#include "mmvq_core.h"
#include "quant_core.h"
#include "rope_dev.h"

#include <math.h>

namespace mi355x {

// The header occupies the first four 64-byte lines of the kernel-argument block; the kernel fetches them with ONE batch of scalar
// loads (k_mmvq_fused) and picks its group's entries with compares and selects.
struct fused_mmvq_args {
    // line 0
    int block_end[MMVQ_MAX_GROUPS];       // cumulative workgroup counts; unused entries = INT_MAX
    int x_off[MMVQ_MAX_GROUPS];           // = g[i].x_off
    int gtype[MMVQ_MAX_GROUPS];           // = g[i].type
    int gm[MMVQ_MAX_GROUPS];              // = g[i].m
    // line 1
    const char * gW[MMVQ_MAX_GROUPS];     // = g[i].W, g[i].W2
    const char * gW2[MMVQ_MAX_GROUPS];
    // line 2
    const int32_t * geid[MMVQ_MAX_GROUPS];// = g[i].eid (NULL: a plain weight tensor)
    const int64_t * kidx[MMVQ_MAX_GROUPS];// = g[i].st_idx where st_mode == 1 (the K-cache row index), else NULL
    // line 3
    uint32_t grow_stride[MMVQ_MAX_GROUPS];// = g[i].row_stride
    uint32_t gestride[MMVQ_MAX_GROUPS];   // = g[i].estride
    // PRO_Q8: quantized activation column as ONE contiguous image in global memory (qs | d | bsums at the offsets
    // act_q8_carve gives for n = 1), staged verbatim into LDS. PRO_QUANT / PRO_NORM build the same image in LDS from x.
    const char * act;
    const float * x; const float * norm_w;
    const int32_t * pos;                  // = rope.pos when a group has EPI_ROPE, else NULL
    // line 4 (first half)
    int n_groups;
    int k;
    int act_kind;
    int act_chunks;                       // 16-byte chunks
    int off_d, off_bs;                    // byte offsets of d / bsums inside the image
    float eps; int pad0;
    // ---- end of the header ----
    fused_rope rope;
    mmvq_fin fin;                         // GLU launches: the producer quantizes its own output for the mat-vec that reads it next (kind == 0: off)
    mmvq_group g[MMVQ_MAX_GROUPS];
#ifdef MI_STAMPS
    unsigned long long * stamps;          // [workgroup][MI_STAMP_N] stamps (tools/stamp_timeline.py), NULL = off
#endif
};
static_assert(MMVQ_MAX_GROUPS == 4, "the header layout is written for four groups");
static_assert(offsetof(fused_mmvq_args, gW) == 64 && offsetof(fused_mmvq_args, geid) == 128 && offsetof(fused_mmvq_args, grow_stride) == 192 &&
              offsetof(fused_mmvq_args, n_groups) == 256 && offsetof(fused_mmvq_args, k) == 260 && offsetof(fused_mmvq_args, act_chunks) == 268 &&
              offsetof(fused_mmvq_args, eps) == 280, "fused_mmvq_args: the header must be four 64-byte lines");

// what a workgroup picks out of the header
struct fused_sel {
    int gi, wg_in_group, nwg_group, x_off, type, m;
    int neox_hl;                          // 0: a unit is two adjacent rows; h + 1: NEOX rotation partners, rows i and i + 2^h of a 2^(h+1)-row head
    const char * W; const char * W2; const int32_t * eid; const int64_t * kidx;
    uint32_t row_stride, estride;
    const char * act; const float * x; const float * norm_w; const int32_t * pos;
    int k, act_chunks, off_d, off_bs; float eps;
    const void * dummy;                   // a readable address for the loads of absent operands
};

typedef int int16v __attribute__((ext_vector_type(16)));
typedef int int8v  __attribute__((ext_vector_type(8)));

// a pointer from two header dwords: made in the GLOBAL address space so that the loads through it stay global_load (a pointer made
// from integers is generic: every load became flat_load, which also counts on lgkmcnt and returns out of order)
template <typename P>
static __device__ __forceinline__ P mk_ptr(int lo, int hi) {
    typedef const char __attribute__((address_space(1))) * gptr;
    return (P) (const char *) (gptr) (((unsigned long long)(unsigned) hi << 32) | (unsigned long long)(unsigned) lo);
}
static __device__ __forceinline__ int pick4(int a0, int a1, int a2, int a3, int i) { return i == 0 ? a0 : (i == 1 ? a1 : (i == 2 ? a2 : a3)); }

// ONE batch of scalar loads for the whole header (the compiler's own loads of by-value struct fields came out as a chain of
// dependent loads, each behind its own wait: ISA of round 1's kernel), then compares and selects
static __device__ __forceinline__ fused_sel load_header(int b) {
    const void * ka = (const void *) __builtin_amdgcn_kernarg_segment_ptr();
    int16v h0, h1, h2, h3; int8v h4;
    asm volatile("s_nop 4\n\ts_load_dwordx16 %0, %5, 0x0\n\ts_load_dwordx16 %1, %5, 0x40\n\ts_load_dwordx16 %2, %5, 0x80\n\t"
                 "s_load_dwordx16 %3, %5, 0xc0\n\ts_load_dwordx8 %4, %5, 0x100\n\ts_waitcnt lgkmcnt(0)"
                 : "=&s"(h0), "=&s"(h1), "=&s"(h2), "=&s"(h3), "=&s"(h4) : "s"(ka) : "memory");
    fused_sel s;
    const int be0 = h0[0], be1 = h0[1], be2 = h0[2], be3 = h0[3];
    const int gi = (b >= be0 ? 1 : 0) + (b >= be1 ? 1 : 0) + (b >= be2 ? 1 : 0);
    const int first = gi == 0 ? 0 : (gi == 1 ? be0 : (gi == 2 ? be1 : be2));
    const int last  = pick4(be0, be1, be2, be3, gi);
    s.gi = gi; s.wg_in_group = b - first; s.nwg_group = last - first;
    s.x_off = pick4(h0[4], h0[5], h0[6], h0[7], gi);
    { const int gt = pick4(h0[8], h0[9], h0[10], h0[11], gi); s.type = gt & 0xFFFF; s.neox_hl = gt >> 16; }
    s.m     = pick4(h0[12], h0[13], h0[14], h0[15], gi);
    s.W   = mk_ptr<const char *>(pick4(h1[0], h1[2], h1[4], h1[6], gi), pick4(h1[1], h1[3], h1[5], h1[7], gi));
    s.W2  = mk_ptr<const char *>(pick4(h1[8], h1[10], h1[12], h1[14], gi), pick4(h1[9], h1[11], h1[13], h1[15], gi));
    s.eid = mk_ptr<const int32_t *>(pick4(h2[0], h2[2], h2[4], h2[6], gi), pick4(h2[1], h2[3], h2[5], h2[7], gi));
    s.kidx = mk_ptr<const int64_t *>(pick4(h2[8], h2[10], h2[12], h2[14], gi), pick4(h2[9], h2[11], h2[13], h2[15], gi));
    s.row_stride = (uint32_t) pick4(h3[0], h3[1], h3[2], h3[3], gi);
    s.estride    = (uint32_t) pick4(h3[4], h3[5], h3[6], h3[7], gi);
    s.act = mk_ptr<const char *>(h3[8], h3[9]);
    s.x = mk_ptr<const float *>(h3[10], h3[11]);
    s.norm_w = mk_ptr<const float *>(h3[12], h3[13]);
    s.pos = mk_ptr<const int32_t *>(h3[14], h3[15]);
    s.k = h4[1]; s.act_chunks = h4[3]; s.off_d = h4[4]; s.off_bs = h4[5]; s.eps = __builtin_bit_cast(float, (int) h4[6]);
    s.dummy = (const void *) ka;
    return s;
}

// the three device values a workgroup needs besides its descriptor — the expert index of a MUL_MAT_ID group, the K-cache row, the
// rope position — in ONE batch of scalar loads (absent ones read the kernel-argument block and are ignored)
static __device__ __forceinline__ void load_chased(const fused_sel & s, int & eid0, long long & idx0, int & pos0) {
    const void * pe = s.eid ? (const void *) s.eid : s.dummy;
    const void * pk = s.kidx ? (const void *) s.kidx : s.dummy;
    const void * pp = s.pos ? (const void *) s.pos : s.dummy;
    int e, q; long long i;
    asm volatile("s_nop 4\n\ts_load_dword %0, %3, 0x0\n\ts_load_dwordx2 %1, %4, 0x0\n\ts_load_dword %2, %5, 0x0\n\ts_waitcnt lgkmcnt(0)"
                 : "=&s"(e), "=&s"(i), "=&s"(q) : "s"(pe), "s"(pk), "s"(pp) : "memory");
    eid0 = s.eid ? e : 0; idx0 = s.kidx ? i : 0; pos0 = s.pos ? q : 0;
}

#ifdef MI_STAMPS
#define MI_STAMP_N 16
#define MI_STAMP(i_) do { if (p.stamps && threadIdx.x == 0) p.stamps[blockIdx.x*MI_STAMP_N + (i_)] = wall_clock64(); } while (0)
#define MI_STAMP_CYC(i_) do { if (p.stamps && threadIdx.x == 0) p.stamps[blockIdx.x*MI_STAMP_N + (i_)] = clock64(); } while (0)
#else
#define MI_STAMP(i_) do { } while (0)
#define MI_STAMP_CYC(i_) do { } while (0)
#endif

// ---- bytes handed from one workgroup to another INSIDE a launch (MI355X guide, inter-workgroup visibility): write-through (`sc1`)
// stores, drained by every storing wave before its workgroup signals; every load of them an `sc1` load (served by L2, never by the
// reading CU's L1, which another CU's stores do not refresh) ----
static __device__ __forceinline__ void st_f32_sc1(float * p, float v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
static __device__ __forceinline__ float4v ld_f4_sc1(const float * p) {      // the wait is part of the statement: hipcc does not count asm loads
    float4v v;
    asm volatile("global_load_dwordx4 %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=&v"(v) : "v"(p) : "memory");
    return v;
}

// operands of one row pair's epilogue, requested when the pair starts (all lanes load the same addresses: one line, broadcast)
// EXT (template): the extended epilogue — a bias before the rotation and NEOX rotation pairs (EPI_ROPE), a second addend and per-expert
// bias rows (EPI_ADD): gpt-oss's graphs. Compiled only into the instantiations a launch with such a group takes (fused_launch.ext): carried
// in every kernel it cost the Llama-3-8B decode 2.7 % (528 -> 514 tok/s: code size and scalar registers of launches that last 4-8 us)
struct pair_pre { float r0, r1; long long i0, i1; long long rcs; float q0, q1; };      // rcs: the pair's rotation, (cos, sin)*mscale as two f32 (split only after pair_wait)

// The addresses are workgroup... wave-uniform (the rows of a unit are), so these are SCALAR loads (s_load, issued here by inline asm, waited
// for in pair_wait just before the pair's finish): round 2 first had them as vector loads — seven 4-byte loads per row pair, most of
// them of an absent operand (read from `dummy`, a readable address, and ignored: loads inside branches made the compiler wait for every
// outstanding load where the branches join) — and two more for the second addend cost the Llama-3-8B decode 3.5 % (530 -> 511 tok/s).
// The scalar cache is invalidated at kernel start; nothing read here is written inside the launch.
// row % head_dim without the integer-division sequence when the head size is a power of two (it is: 64 / 128)
static __device__ __forceinline__ int mi_row_in_head(const fused_rope & r, int row) {
    return (r.head_dim & (r.head_dim - 1)) == 0 ? (row & (r.head_dim - 1)) : row % r.head_dim;
}
// a wave-uniform address into scalar registers (the compiler keeps some of these in VGPRs, depending on the instantiation)
template <typename P> static __device__ __forceinline__ P mi_uni(P q) {
    const unsigned long long a = (unsigned long long) q;
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned) a), hi = __builtin_amdgcn_readfirstlane((unsigned)(a >> 32));
    return (P) (((unsigned long long) hi << 32) | lo);
}
template <bool GLU, bool EXT>
static __device__ __forceinline__ pair_pre pair_prefetch(const mmvq_group & g, const fused_rope & rope, const char * dummy, int m, int row0, int row1, int eid0) {
    pair_pre e = { 0.0f, 0.0f, 0, 0, 0, 0.0f, 0.0f };
    if (GLU) return e;
    const int ra = __builtin_amdgcn_readfirstlane(min(row0, m - 1)), rb = __builtin_amdgcn_readfirstlane(min(row1, m - 1));   // wave-uniform by construction
    // res: the residual of EPI_ADD, or the bias added before the rotation (EPI_ROPE: gpt-oss's wq / wk, src/llama-model.cpp:17636-17652)
    const bool has_res = g.res != nullptr && (g.epi == EPI_ADD || (EXT && g.epi == EPI_ROPE)), has_rot = g.epi == EPI_ROPE, has_idx = g.st_mode == 2;
    // res_eid: res is a [m, n_expert] bias table and this group is expert eid0 (ADD_ID after a one-token MUL_MAT_ID: gpt-oss's ffn_down_exps.bias)
    const float * rp = has_res ? g.res + (EXT && g.res_eid ? (size_t) eid0*m : 0) : (const float *) dummy;
    // res2: a second addend after the first (wo.x + bias, then + the residual stream: two ADD nodes in the graph)
    const bool has_res2 = EXT && g.res2 != nullptr && g.epi == EPI_ADD;
    const float * rq = has_res2 ? g.res2 : (const float *) dummy;
    const float * fp = has_rot ? rope.tab : (const float *) dummy;       // the pair's (cos, sin): 8 bytes at tab + 2*ip
    const int64_t * ip = has_idx ? g.st_idx : (const int64_t *) dummy;
    const float * a_r0 = rp + (has_res ? ra : 0), * a_r1 = rp + (has_res ? rb : 0), * a_q0 = rq + (has_res2 ? ra : 0), * a_q1 = rq + (has_res2 ? rb : 0);
    const int rih = mi_row_in_head(rope, ra);
    const float * a_ff = fp + (has_rot ? 2*(rope.neox ? min(rih, (rope.n_dims >> 1) - 1) : (min(rih, rope.n_dims - 1) >> 1)) : 0);
    const int64_t * a_i0 = ip + (has_idx ? ra : 0), * a_i1 = ip + (has_idx ? rb : 0);
    asm volatile("s_load_dword %0, %7, 0x0\n\ts_load_dword %1, %8, 0x0\n\ts_load_dword %2, %9, 0x0\n\ts_load_dword %3, %10, 0x0\n\t"
                 "s_load_dwordx2 %4, %11, 0x0\n\ts_load_dwordx2 %5, %12, 0x0\n\ts_load_dwordx2 %6, %13, 0x0"
                 : "=&s"(e.r0), "=&s"(e.r1), "=&s"(e.q0), "=&s"(e.q1), "=&s"(e.rcs), "=&s"(e.i0), "=&s"(e.i1)
                 : "s"(mi_uni(a_r0)), "s"(mi_uni(a_r1)), "s"(mi_uni(a_q0)), "s"(mi_uni(a_q1)), "s"(mi_uni(a_ff)), "s"(mi_uni(a_i0)), "s"(mi_uni(a_i1)));      // no memory clobber: it would fence the weight stream's scheduling
    return e;
}
// the scalar loads of pair_prefetch have landed (ties the registers to the wait so that no use can move above it)
static __device__ __forceinline__ void pair_wait(pair_pre & e) {
    asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(e.r0), "+s"(e.r1), "+s"(e.q0), "+s"(e.q1), "+s"(e.rcs), "+s"(e.i0), "+s"(e.i1));
}

// rope on one rotation pair — NORM (2i, 2i+1) or NEOX (i, i + n_dims/2), row_in_head = the first of the two — with the pair's (cos, sin)*mscale
// from the token's rotation table (k_rope_table: the formulas of rope_pair / elem.hip k_rope, evaluated once per token instead of once per
// row pair inside every launch, where the one live lane of the finish cost as many issue cycles as 64)
static __device__ __forceinline__ void rope_pair_cs(const fused_rope & r, int row_in_head, long long rcs, float & x0, float & x1) {
    if (row_in_head >= r.n_dims) return;
    const float c = __builtin_bit_cast(float, (int)(rcs & 0xFFFFFFFFll)), s = __builtin_bit_cast(float, (int)((unsigned long long) rcs >> 32));
    const float a = x0, b = x1;
    x0 = a*c - b*s;
    x1 = a*s + b*c;
}

// what lane 0 does with the two finished rows of a pair (inlined: a call would spill the in-flight prefetch registers)
template <bool EXT>
static __device__ __forceinline__ void finish_pair(const mmvq_group & g, const fused_rope & rope, float s0, float s1, int row0, int row1, int pos0, long long idx0,
                                                   const pair_pre & e, const int g_m, const int rows) {
    const int m = rows > 1 ? g_m : row0 + 1;      // rows == 1: the unit has no second row
    if (g.epi == EPI_ADD) {
        s0 += e.r0;
        if (row1 < m) s1 += e.r1;
        if (EXT && g.res2) { s0 += e.q0; if (row1 < m) s1 += e.q1; }
    } else if (g.epi == EPI_ROPE) {
        if (EXT && g.res) { s0 += e.r0; s1 += e.r1; }                          // bias first, then the rotation
        rope_pair_cs(rope, mi_row_in_head(rope, row0), e.rcs, s0, s1);   // m is a multiple of the head size on this path
    }
    g.dst[row0] = s0;
    if (row1 < m) g.dst[row1] = s1;
    if (g.st_mode == 1) {
        uint16_t * q = g.st16 + idx0*g.st_row_elems;
        q[row0] = f32_to_f16_bits(s0);
        if (row1 < m) q[row1] = f32_to_f16_bits(s1);
    } else if (g.st_mode == 2) {
        g.st16[e.i0] = f32_to_f16_bits(s0);
        if (row1 < m) g.st16[e.i1] = f32_to_f16_bits(s1);
    }
}

// ---- batched finish of the dual GLU stream (round 2). The kernels are bound by instruction issue, not by HBM (ISA count: ~690 instructions per
// k-step in the gate/up loop; 22 GB/s per CU is what four SIMDs issue), and for k = 4096 almost half of them were the per-row finish — two
// full-wave reductions, expf, a division, address arithmetic — repeated for each of a wave's 7 rows. Now a wave keeps the lane-partial sums of
// its finished rows (8 gate + 8 up registers), and reduces all 16 together: v_permlane32_swap / v_permlane16_swap fold two (then four) rows
// into one register, so that after 40 instructions four registers hold the 16 totals, four rows (16 lanes each) per register; silu(gate)*up
// then runs once per register instead of once per row, and the first lane of each 16-lane row stores its row.
// Summation order per row: lane i + lane i+32, then + the other 16-lane half, then the 16 lanes (DPP) — every partial sum is still f32.
typedef unsigned mi_u2 __attribute__((ext_vector_type(2)));
static __device__ __forceinline__ float mi_fold32(float a, float b) {      // lanes 0-31: a summed over its halves; lanes 32-63: b likewise
    const mi_u2 r = __builtin_amdgcn_permlane32_swap(__builtin_bit_cast(unsigned, a), __builtin_bit_cast(unsigned, b), false, false);
    const unsigned r0 = r.x, r1 = r.y;
    return __builtin_bit_cast(float, r0) + __builtin_bit_cast(float, r1);
}
static __device__ __forceinline__ float mi_fold16(float a, float b) {      // 16-lane rows 0..3: a.row0+a.row1, b.row0+b.row1, a.row2+a.row3, b.row2+b.row3
    const mi_u2 r = __builtin_amdgcn_permlane16_swap(__builtin_bit_cast(unsigned, a), __builtin_bit_cast(unsigned, b), false, false);
    const unsigned r0 = r.x, r1 = r.y;
    return __builtin_bit_cast(float, r0) + __builtin_bit_cast(float, r1);
}
// hg / hu: lane-partial sums of the last `nh` (<= 8) finished rows, entry 0 the most recent = row p_last, entry e = row p_last - e*u_step
static __device__ __forceinline__ void glu_flush8(const float (&hg)[8], const float (&hu)[8], int nh, int p_last, int u_step, const mmvq_group & g,
                                                  int g_m, int eid0, bool sc1_store, int lane) {
    float tg[2], tu[2];
#pragma unroll
    for (int m = 0; m < 2; m++) {
        // register m: 16-lane rows 0..3 = entries 4m + {0, 2, 1, 3}
        tg[m] = row16_sum(mi_fold16(mi_fold32(hg[4*m], hg[4*m + 1]), mi_fold32(hg[4*m + 2], hg[4*m + 3])));
        tu[m] = row16_sum(mi_fold16(mi_fold32(hu[4*m], hu[4*m + 1]), mi_fold32(hu[4*m + 2], hu[4*m + 3])));
    }
    const int r = lane >> 4, perm = (r == 1) ? 2 : ((r == 2) ? 1 : r);
#pragma unroll
    for (int m = 0; m < 2; m++) {
        const int e = 4*m + perm;
        const int row = p_last - e*u_step;
        const bool live = e < nh && (lane & 15) == 0;
        float s0 = tg[m], up_s = tu[m];
        if (g.b_gate) {     // + bias rows of this group's expert (ADD_ID)
            const size_t brow = (size_t) eid0*g_m + (live ? row : 0);
            s0 += g.b_gate[brow]; up_s += g.b_up[brow];
        }
        if (g.glu_alpha != 0.0f) {      // swiglu_oai, as elem.hip k_glu
            const float xc = fminf(s0, g.glu_limit), gc = fmaxf(fminf(up_s, g.glu_limit), -g.glu_limit);
            s0 = (xc/(1.0f + expf(-xc*g.glu_alpha)))*(gc + 1.0f);
        } else {
            s0 = (s0/(1.0f + expf(-s0)))*up_s;      // silu(gate)*up, as elem.hip k_glu
        }
        if (live) { if (sc1_store) st_f32_sc1(g.dst + row, s0); else g.dst[row] = s0; }
    }
}

//   PRO  : where the activation comes from (mmvq_prologue)
//   NA   : PRO_Q8: 16-byte image chunks per thread; PRO_QUANT/PRO_NORM: 256-element chunks per wave (k <= NA*256*waves)
//   D    : ring depth (2; 4 for the one-row GLU units and for long single-tensor streams)
//   FWT  : waves per workgroup
template <int TYPE, bool GLU, int PRO, int NA, int D, int FWT, bool EXT>
static __device__ __forceinline__ void fused_body(const fused_mmvq_args & p, const fused_sel & sel, char * smem, int lane, int wave) {
    typedef mmvq_t<TYPE> T;
    // rows per unit of work: a pair for single-tensor groups; ONE row (of gate and of up) for the dual GLU stream, so that n_ff = 14336
    // rows split evenly over 2048 waves (7 each; as pairs it was 4 for half the waves and 3 for the rest — tools/stamp_timeline.py)
    constexpr int R = GLU ? 1 : 2, LPB = T::LPB, BPW = 64/LPB, ACT = T::ACT;
    const int nb = sel.k / T::QK;
    const int iters = (nb + BPW - 1)/BPW;
    const int slot = lane % LPB, ibl = lane / LPB;

    MI_STAMP(0); MI_STAMP_CYC(8);
    const float * gx = sel.x + sel.x_off;
    int4v areg[PRO == PRO_Q8 ? NA : 1];
    float4v xv[PRO != PRO_Q8 ? NA : 1], wv[PRO == PRO_NORM ? NA : 1];
    const int nchunk = (sel.k + 255) >> 8;     // the last chunk may be partial (k % 32 == 0 with Q8_0 activations: gpt-oss's 2880)
    // element offset of this lane's 4 floats in chunk slot i: clamped into the vector; `live` tells whether they exist
#define MI_XOFF(i_) min(min(wave + FWT*(i_), nchunk - 1)*256 + lane*4, sel.k - 4)
#define MI_XLIVE(i_) ((wave + FWT*(i_))*256 + lane*4 < sel.k)

    // ---- (1) activation loads: their addresses come from the header alone ----
    if (PRO == PRO_Q8) {
#pragma unroll
        for (int i = 0; i < NA; i++) {
            const int idx = min((int) threadIdx.x + i*(FWT*64), sel.act_chunks - 1);
            areg[i] = *(const int4v *) (sel.act + (size_t) idx*16);
        }
    } else {
#pragma unroll
        for (int i = 0; i < NA; i++) { xv[i] = *(const float4v *) (gx + MI_XOFF(i)); if (!MI_XLIVE(i)) xv[i] = float4v{ 0.0f, 0.0f, 0.0f, 0.0f }; }
    }
    // A CU's L1 returns data in request order across all its waves: a load that hits L2 (the activation, just written) queued
    // behind one that goes to HBM (weights, norm weights) of ANY wave comes back with HBM latency — 1-4 us instead of ~0.5 us,
    // and the whole prologue hangs on it (measured, tools/stamp_timeline.py: the second workgroup on a CU saw its activation 2 us
    // after the first). So: every wave issues its activation loads, the workgroup meets at a barrier (issue order = request
    // order), and only then are norm weights and the weight stream requested. The asm statements are compiler barriers too.
    asm volatile("" ::: "memory");
    // the device values behind pointers (expert index, K-cache row, rope position): one scalar-load batch beside the activation round trip
    int eid0, pos0; long long idx0;
    load_chased(sel, eid0, idx0, pos0);
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");

    const mmvq_group & g = p.g[sel.gi];                  // the cold part of the descriptor (epilogue operands): loaded when first used
    const int g_m = sel.m;
    const size_t g_row_stride = sel.row_stride;
    const int P = (g_m + R - 1)/R;                       // units (row pairs; rows of the dual GLU stream) in this group
    // which units this wave owns: unit j of the wave is u_base + j*u_step, j < n_mine. Grid-strided by default; a GLU launch that
    // finalises its output (p.fin) gives every wave a CONTIGUOUS run of rows instead, so that a 256-row chunk of the output — the unit
    // the activation quantizer works on — comes from five or six workgroups, not from all of them
    const bool fin_on = GLU && p.fin.kind != 0;
    const int fin_rpw = (P + sel.nwg_group*FWT - 1)/(sel.nwg_group*FWT);       // rows per wave (contiguous ownership)
    const int u_step = fin_on ? 1 : sel.nwg_group*FWT;
    // 16-wave workgroups: unit = wave*nwg + wg, so that a group may spread over MORE workgroups than units/16 (its high waves then own
    // nothing): the Q6_K group of a norm+QKV launch needs ~3 us to queue its first weight loads at 16 pairs per CU (every Q6_K launch
    // does: tools/stamp_timeline.py on a Q6_K model), 64 CUs of that launch were idle
    const int u_base = fin_on ? (sel.wg_in_group*FWT + wave)*fin_rpw : (FWT == 16 ? wave*sel.nwg_group + sel.wg_in_group : sel.wg_in_group*FWT + wave);
    const int n_mine = fin_on ? max(0, min(fin_rpw, P - u_base)) : (u_base < P ? (P - 1 - u_base)/u_step + 1 : 0);
    int p_cur = u_base;
    // the two rows of unit pp (single-tensor groups): adjacent, or the NEOX rotation partners i and i + half of one head
    const int nhl = (GLU || !EXT) ? 0 : sel.neox_hl;
#define MI_ROW_A(pp_) (nhl ? ((((pp_) >> (nhl - 1)) << nhl) + ((pp_) & ((1 << (nhl - 1)) - 1))) : (pp_)*R)
#define MI_ROW_B(pp_) (MI_ROW_A(pp_) + (nhl ? (1 << (nhl - 1)) : 1))
    // an expert of a stack (MUL_MAT_ID, one token): the index is a device value, workgroup-uniform
    const size_t eoff = (size_t) eid0*sel.estride;
    const char * gW = sel.W + eoff; const char * gW2 = GLU ? sel.W2 + eoff : nullptr;
    if (PRO == PRO_NORM) {
#pragma unroll
        for (int i = 0; i < NA; i++) wv[i] = *(const float4v *) (sel.norm_w + min(wave + FWT*i, nchunk - 1)*256 + lane*4);
    }
    // the stream is the sequence of (row pair, k-step) this wave will consume; (p_pf, it_pf) is the next step to fetch.
    // Past the end of the stream the loads go to the wave's own first block (an L1 hit), not to a line every wave would share.
    int j_pf = 0, it_pf = 0;
    typename T::wfrag w[D][R], u[GLU ? D : 1][R];
#define MI_FETCH(d_) { \
        const bool live = j_pf < n_mine; \
        const int pp = live ? u_base + j_pf*u_step : min(u_base, P - 1); \
        const int ibf = live ? min(it_pf*BPW + ibl, nb - 1) : 0; \
        _Pragma("unroll") for (int r = 0; r < R; r++) { \
            const size_t off = (size_t) min(GLU ? pp : (r ? MI_ROW_B(pp) : MI_ROW_A(pp)), g_m - 1)*g_row_stride; \
            w[d_][r] = T::load_w(gW + off, ibf, slot); \
            if (GLU) u[GLU ? d_ : 0][r] = T::load_w(gW2 + off, ibf, slot); \
        } \
        if (++it_pf == iters) { it_pf = 0; j_pf++; } }
#define MI_FENCE asm volatile("" ::: "memory")

    // ---- (2) weight prefetch: the first D steps of this wave's stream ----
    // A wave blocks at a load it cannot queue (the CU's request queue is finite) and then cannot run its share of the prologue
    // either, so the D steps are not issued in one burst: one step now, the others between the phases of the prologue (FENCE keeps
    // the compiler from hoisting them back up). HBM then has work from the first 0.2 us on and the prologue math starts as soon as
    // the activation is there.
    { MI_FETCH(0) }
    MI_FENCE;
    pair_pre epre = pair_prefetch<GLU, EXT>(g, p.rope, sel.W, g_m, MI_ROW_A(p_cur), R > 1 ? MI_ROW_B(p_cur) : MI_ROW_A(p_cur), eid0);      // behind the first weight step: needed only after the pair's last dot
    MI_FENCE;

    // ---- (3) prologue: build the quantized activation image in LDS ----
    // the image this workgroup BUILDS (PRO_QUANT / PRO_NORM) is laid out for its own group's activation format — a launch may mix K-quant
    // groups (Q8_K image) with Q8_0 groups (Mixtral's wq Q4_K + wk / wv Q8_0); a copied image (PRO_Q8) has the one layout the host gave
    const int off_bs_l = PRO == PRO_Q8 ? sel.off_bs : sel.off_d + ((((sel.k >> (ACT == T_Q8_0 ? 5 : 8))*4) + 255) & ~255);
    int8_t * l_qs = (int8_t *) smem; float * l_d = (float *) (smem + sel.off_d); int16_t * l_bs = (int16_t *) (smem + off_bs_l);
    if (PRO == PRO_Q8) {
#pragma unroll
        for (int i = 0; i < NA; i++) {
            const int idx = threadIdx.x + i*(FWT*64);
            if (idx < sel.act_chunks) *(int4v *) (smem + (size_t) idx*16) = areg[i];
        }
    } else {
        float scale = 1.0f;
        if (PRO == PRO_NORM) {
            float * red = (float *) (smem + off_bs_l + (((sel.k >> (ACT == T_Q8_0 ? 5 : 4))*2 + 15) & ~15));   // FWT floats after the image
            float ss = 0.0f;
#pragma unroll
            for (int i = 0; i < NA; i++) if (wave + FWT*i < nchunk) ss += xv[i].x*xv[i].x + xv[i].y*xv[i].y + xv[i].z*xv[i].z + xv[i].w*xv[i].w;
            ss = wave_sum(ss);
            MI_STAMP(4);
            if (lane == 0) red[wave] = ss;
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            ss = ((red[0] + red[1]) + (red[2] + red[3])) + ((red[4] + red[5]) + (red[6] + red[7]));
            if (FWT == 16) ss += ((red[8] + red[9]) + (red[10] + red[11])) + ((red[12] + red[13]) + (red[14] + red[15]));
            scale = 1.0f/sqrtf(ss/(float) sel.k + sel.eps);
            MI_STAMP(7);
            MI_FENCE;
            if constexpr (D > 1) { MI_FETCH(1) }
            MI_FENCE;
        }
        // all chunks of this wave are quantized first (independent chains the scheduler can interleave; a chunk past the end is
        // quantized too — its lanes hold a clamped duplicate — and simply not stored), then stored
        uint32_t qp[NA]; float qd[NA]; int qb[NA];
#pragma unroll
        for (int i = 0; i < NA; i++) {
            float4v v = xv[i];
            if (PRO == PRO_NORM) { v.x = (v.x*scale)*wv[i].x; v.y = (v.y*scale)*wv[i].y; v.z = (v.z*scale)*wv[i].z; v.w = (v.w*scale)*wv[i].w; }
            qp[i] = quant_chunk256<ACT>(v, qd[i], qb[i]);
        }
        MI_STAMP(5);
        MI_FENCE;
        // the steps must be fetched in ring order: set d holds stream step d
        if constexpr (PRO == PRO_QUANT && D > 1) { MI_FETCH(1) }
        else if constexpr (D > 2) { MI_FETCH(2) }
        MI_FENCE;
#pragma unroll
        for (int i = 0; i < NA; i++) {
            const int c = wave + FWT*i;
            if (c < nchunk) store_chunk256<ACT>(qp[i], qd[i], qb[i], c, lane, l_qs, l_d, l_bs);
        }
    }
    MI_FENCE;
    if constexpr (PRO == PRO_Q8 && D > 1) { MI_FETCH(1) }
    if constexpr (D > 2 && PRO != PRO_NORM) { MI_FETCH(2) }
    if constexpr (D > 3) { MI_FETCH(3) }
    MI_FENCE;
    MI_STAMP(6);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    act_view av;
    av.qs = l_qs; av.d = l_d; av.bs = l_bs;
    MI_STAMP(1);

    // ---- (4)+(5) stream: D steps per trip, each consuming one register set and refilling it for D steps later ----
    const int total = n_mine*iters;
    int it = 0;
    float acc[2] = { 0.0f, 0.0f }, acu[2] = { 0.0f, 0.0f };
    float hg[GLU ? 8 : 1], hu[GLU ? 8 : 1]; int nh = 0;      // the dual GLU stream: lane-partial sums of finished rows, reduced eight at a time (glu_flush8)
#pragma unroll
    for (int q = 0; q < (GLU ? 8 : 1); q++) { hg[q] = 0.0f; hu[q] = 0.0f; }
#ifdef MI_STAMPS
    bool first_pair = true;
#endif
    for (int s = 0; s < total; s += D) {
#pragma unroll
        for (int d = 0; d < D; d++) {
            if (s + d < total) {        // wave-uniform
                const int ib = it*BPW + ibl;
                if (ib < nb) {
                    const typename T::afrag a = T::load_a(av, ib, slot);
#pragma unroll
                    for (int r = 0; r < R; r++) { acc[r] += T::dot(w[d][r], a, slot); if (GLU) acu[r] += T::dot(u[GLU ? d : 0][r], a, slot); }
                }
                MI_FETCH(d)
                if (++it == iters) {
#ifdef MI_STAMPS
                    if (first_pair) { MI_STAMP(2); first_pair = false; }
#endif
                    if (GLU) {
#pragma unroll
                        for (int q = 7; q > 0; q--) { hg[GLU ? q : 0] = hg[GLU ? q - 1 : 0]; hu[GLU ? q : 0] = hu[GLU ? q - 1 : 0]; }
                        hg[0] = acc[0]; hu[0] = acu[0];
                        if (++nh == 8) { glu_flush8((const float (&)[8]) hg, (const float (&)[8]) hu, 8, p_cur, u_step, g, g_m, eid0, fin_on, lane); nh = 0; }
                    } else {
                        const float s0 = wave_sum(acc[0]), s1 = R > 1 ? wave_sum(acc[1]) : 0.0f;
                        pair_wait(epre);
                        if (lane == 0) finish_pair<EXT>(g, p.rope, s0, s1, MI_ROW_A(p_cur), MI_ROW_B(p_cur), pos0, idx0, epre, g_m, R);
                    }
                    it = 0; p_cur += u_step;
                    acc[0] = acc[1] = 0.0f; acu[0] = acu[1] = 0.0f;
                    if (!GLU && s + d + 1 < total) epre = pair_prefetch<GLU, EXT>(g, p.rope, sel.W, g_m, MI_ROW_A(p_cur), MI_ROW_B(p_cur), eid0);     // the next pair's epilogue operands
                }
            }
        }
    }
    if (GLU && nh > 0) glu_flush8((const float (&)[8]) hg, (const float (&)[8]) hu, nh, p_cur - u_step, u_step, g, g_m, eid0, fin_on, lane);
    if (GLU && fin_on) {
        // ---- producer-side activation quantization (round 2): the mat-vec that reads this launch's output next needs it as int8 blocks
        // (Q8_K / Q8_0, 256-element chunks). Instead of every one of its 256 workgroups quantizing all of it again in its prologue
        // (5.5 us of a 15 us ffn_down launch, instruction-bound), the workgroup that completes a chunk here quantizes that chunk once.
        // Per chunk: an arrival counter; each workgroup whose rows touch the chunk adds 1 after its rows are stored (sc1) and drained;
        // the one whose add came last (by the returned value) loads the chunk (sc1), quantizes it and stores the image piece.
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        int * fin_list = (int *) (smem + off_bs_l + (((sel.k >> (ACT == T_Q8_0 ? 5 : 4))*2 + 15) & ~15) + 64);      // [0] = count, [1..] = chunks (after the RMS scratch)
        const int RW = FWT*fin_rpw, ra = sel.wg_in_group*RW, rb = min(ra + RW, g_m);
        if (wave == 0) {
            // one lane per chunk this workgroup's rows touch (at most 7: mul_mat_vec_q_fused_fin_supported): the counter round trips run side by
            // side instead of one after the other in a single thread's loop (a workgroup whose run of rows spans two chunks paid two of them)
            const int c = (ra >> 8) + lane;
            const bool mine = ra < g_m && lane < 8 && c <= ((rb - 1) >> 8);
            bool last = false;
            if (mine) {
                const int expect = min((c << 8) + 255, g_m - 1)/RW - (c << 8)/RW + 1;
                const unsigned old = __hip_atomic_fetch_add(p.fin.counters + c, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                last = (int) old == expect - 1;
                if (last) __hip_atomic_store(p.fin.counters + c, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);     // re-armed for the next launch
            }
            const unsigned long long m = __ballot(last);
            if (last) fin_list[1 + __popcll(m & ((1ull << lane) - 1))] = c;
            if (lane == 0) fin_list[0] = __popcll(m);
        }
        __syncthreads();
        const int nfin = fin_list[0];
        for (int j = wave; j < nfin; j += FWT) {
            const int c = fin_list[1 + j];
            const float4v v = ld_f4_sc1(g.dst + c*256 + lane*4);
            if (p.fin.kind == T_Q8_0) quant_store_chunk256<T_Q8_0>(v, c, lane, p.fin.qs, p.fin.d, p.fin.bs);
            else                      quant_store_chunk256<T_Q8_K>(v, c, lane, p.fin.qs, p.fin.d, p.fin.bs);
        }
    }
    MI_STAMP(3); MI_STAMP_CYC(9);
#undef MI_FETCH
#undef MI_ROW_A
#undef MI_ROW_B
#undef MI_FENCE
#undef MI_XOFF
#undef MI_XLIVE
}

// One instantiation per {weight type or pair of types} x {GLU} x {prologue} x {activation size class}: a single kernel switching
// over all six formats at run time allocates registers for the fattest path (227 VGPRs -> 2 waves/SIMD), which starves the HBM stream.
// FWT = waves per workgroup: 8, or 16 for launches with more row pairs than 8 waves x CUs but no more than 16 x CUs (norm + QKV:
// 3072 pairs) — ONE 1024-thread workgroup per CU shares one prologue (two 8-wave workgroups on a CU ran the second one's prologue
// ~2x slower), every wave owns a single pair, and the prologue has one 256-chunk per wave instead of two
template <int TA, int TB, bool GLU, int PRO, int NA, int D, int FWT = 8, bool EXT = false>
__global__ void __launch_bounds__(FWT*64, FWT == 8 ? 2 : 1) k_mmvq_fused(const fused_mmvq_args p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    // the wave index as a SCALAR: everything derived from it (the units a wave owns, its trip count, row offsets, the epilogue's addresses)
    // then lives in SGPRs and its branches are scalar branches
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const fused_sel sel = load_header((int) blockIdx.x);
    // 16 waves: every wave owns ONE unit of k <= 4096; a format whose wave covers 16 blocks per step (Q6_K) finishes it in one step, and a
    // second register set would only hold a dead fetch (the Q4_K + Q6_K norm+QKV kernel spilled 3 dwords at its 128-VGPR limit, and the
    // reloads queued behind the weight stream: its Q6_K workgroups left the prologue 3.3 us late, tools/stamp_timeline.py)
    constexpr int DA = (FWT == 16 && mmvq_t<TA>::QK == 256 && mmvq_t<TA>::LPB <= 4) ? 1 : D;
    constexpr int DB = (FWT == 16 && mmvq_t<TB>::QK == 256 && mmvq_t<TB>::LPB <= 4) ? 1 : D;
    if (TA == TB || sel.type == TA) fused_body<TA, GLU, PRO, NA, DA, FWT, EXT>(p, sel, smem, lane, wave);
    else                            fused_body<TB, GLU, PRO, NA, DB, FWT, EXT>(p, sel, smem, lane, wave);
}

// Timing hook: when the host set an event pair (option "profile"), the launch carries it as the DISPATCH's own start / stop events
// (hipExtLaunchKernelGGL): hipEventElapsedTime(ev0, ev1) is then the kernel's execution time by the same packet timestamps rocprofv3's
// kernel trace reports — not "launch call to completion" as a pair of hipEventRecord around the launch measures
extern hipEvent_t mi355x_fused_ev0, mi355x_fused_ev1;
extern const char * mi355x_fused_last_kernel;       // the instantiation the last launch used, spelled as rocprofv3's kernel trace spells it
template <int TA, int TB, bool GLU, int PRO, int NA, int D, int FWT = 8, bool EXT = false> static const char * fused_kname() {
    static char name[96] = "";
    if (!name[0]) snprintf(name, sizeof(name), "k_mmvq_fused<%d, %d, %s, %d, %d, %d, %d, %s>", TA, TB, GLU ? "true" : "false", PRO, NA, D, FWT, EXT ? "true" : "false");
    return name;
}
#define MI_UNP(...) __VA_ARGS__
// the single-tensor instantiations exist twice: with and without the extended epilogue
#define MI_FLX(TARGS6_, FW_, grid_, block_, lds_, stream_, a_) do { \
    if (L.ext) MI_FL((MI_UNP TARGS6_, FW_, true), grid_, block_, lds_, stream_, a_); \
    else       MI_FL((MI_UNP TARGS6_, FW_, false), grid_, block_, lds_, stream_, a_); } while (0)
#define MI_FL(TARGS_, grid_, block_, lds_, stream_, a_) do { \
    mi355x_fused_last_kernel = fused_kname<MI_UNP TARGS_>(); \
    if (mi355x_fused_ev0) { hipExtLaunchKernelGGL((k_mmvq_fused<MI_UNP TARGS_>), grid_, block_, lds_, stream_, mi355x_fused_ev0, mi355x_fused_ev1, 0, a_); mi355x_fused_ev0 = nullptr; mi355x_fused_ev1 = nullptr; } \
    else hipLaunchKernelGGL((k_mmvq_fused<MI_UNP TARGS_>), grid_, block_, lds_, stream_, a_); } while (0)

// a grouped launch, prepared on the host
struct fused_launch { fused_mmvq_args a; int blocks; size_t lds; int ta, tb; bool glu; int mode, na; bool deep; int64_t k; uint64_t wbytes; int fw; bool ext; };

// the launcher of one {TA, TB} kernel family: picks the instantiation for L's prologue / activation size class / ring depth
#define MI_DEFINE_FUSED_LAUNCHER(NAME_, TA_, TB_, HAS_GLU_) \
void NAME_(const fused_launch & L, hipStream_t stream) { \
    const fused_mmvq_args & a = L.a; \
    const dim3 grid((unsigned) L.blocks); \
    const size_t lds = L.lds; \
    const int mode = L.mode, na = L.na; \
    const bool deep = L.deep; \
    constexpr int FW = 8; \
    if (L.fw == 16) { MI_FLX((TA_, TB_, false, PRO_NORM, 1, 2), 16, grid, dim3(1024), lds, stream, a); return; } \
    if (HAS_GLU_ && L.glu) {     /* one-row units: 4 steps = the bytes 2 steps of pairs held */ \
        if (mode == PRO_Q8) { \
            if (na == 1)      MI_FL((TA_, TB_, HAS_GLU_, PRO_Q8, 1, 4), grid, dim3(FW*64), lds, stream, a); \
            else if (na == 2) MI_FL((TA_, TB_, HAS_GLU_, PRO_Q8, 2, 4), grid, dim3(FW*64), lds, stream, a); \
            else              MI_FL((TA_, TB_, HAS_GLU_, PRO_Q8, 4, 4), grid, dim3(FW*64), lds, stream, a); \
        } else if (mode == PRO_NORM) { \
            if (na == 2) MI_FL((TA_, TB_, HAS_GLU_, PRO_NORM, 2, 4), grid, dim3(FW*64), lds, stream, a); \
            else         MI_FL((TA_, TB_, HAS_GLU_, PRO_NORM, 8, 4), grid, dim3(FW*64), lds, stream, a); \
        } else { \
            if (na == 2) MI_FL((TA_, TB_, HAS_GLU_, PRO_QUANT, 2, 4), grid, dim3(FW*64), lds, stream, a); \
            else         MI_FL((TA_, TB_, HAS_GLU_, PRO_QUANT, 8, 4), grid, dim3(FW*64), lds, stream, a); \
        } \
        return; \
    } \
    if (deep) { \
        if (mode == PRO_Q8) { \
            if (na == 1)      MI_FLX((TA_, TB_, false, PRO_Q8, 1, 4), 8, grid, dim3(FW*64), lds, stream, a); \
            else if (na == 2) MI_FLX((TA_, TB_, false, PRO_Q8, 2, 4), 8, grid, dim3(FW*64), lds, stream, a); \
            else              MI_FLX((TA_, TB_, false, PRO_Q8, 4, 4), 8, grid, dim3(FW*64), lds, stream, a); \
        } else if (mode == PRO_NORM) { \
            if (na == 2) MI_FLX((TA_, TB_, false, PRO_NORM, 2, 4), 8, grid, dim3(FW*64), lds, stream, a); \
            else         MI_FLX((TA_, TB_, false, PRO_NORM, 8, 4), 8, grid, dim3(FW*64), lds, stream, a); \
        } else { \
            if (na == 2) MI_FLX((TA_, TB_, false, PRO_QUANT, 2, 4), 8, grid, dim3(FW*64), lds, stream, a); \
            else         MI_FLX((TA_, TB_, false, PRO_QUANT, 8, 4), 8, grid, dim3(FW*64), lds, stream, a); \
        } \
        return; \
    } \
    if (mode == PRO_Q8) { \
        if (na == 1)      MI_FLX((TA_, TB_, false, PRO_Q8, 1, 2), 8, grid, dim3(FW*64), lds, stream, a); \
        else if (na == 2) MI_FLX((TA_, TB_, false, PRO_Q8, 2, 2), 8, grid, dim3(FW*64), lds, stream, a); \
        else              MI_FLX((TA_, TB_, false, PRO_Q8, 4, 2), 8, grid, dim3(FW*64), lds, stream, a); \
    } else if (mode == PRO_NORM) { \
        if (na == 2) MI_FLX((TA_, TB_, false, PRO_NORM, 2, 2), 8, grid, dim3(FW*64), lds, stream, a); \
        else         MI_FLX((TA_, TB_, false, PRO_NORM, 8, 2), 8, grid, dim3(FW*64), lds, stream, a); \
    } else { \
        if (na == 2) MI_FLX((TA_, TB_, false, PRO_QUANT, 2, 2), 8, grid, dim3(FW*64), lds, stream, a); \
        else         MI_FLX((TA_, TB_, false, PRO_QUANT, 8, 2), 8, grid, dim3(FW*64), lds, stream, a); \
    } \
}

void launch_fused_q4_K(const fused_launch & L, hipStream_t stream);
void launch_fused_q5_K(const fused_launch & L, hipStream_t stream);
void launch_fused_q6_K(const fused_launch & L, hipStream_t stream);
void launch_fused_q8_0(const fused_launch & L, hipStream_t stream);
void launch_fused_q4_0(const fused_launch & L, hipStream_t stream);
void launch_fused_mxfp4(const fused_launch & L, hipStream_t stream);
void launch_fused_q4_K_q5_K(const fused_launch & L, hipStream_t stream);
void launch_fused_q4_K_q6_K(const fused_launch & L, hipStream_t stream);
void launch_fused_q8_0_q4_K(const fused_launch & L, hipStream_t stream);
void launch_fused_q5_K_q6_K(const fused_launch & L, hipStream_t stream);

} // namespace mi355x
