// mmvq_stream_cols.hip — K-quant weights x 2..8 quantized activation columns on the STREAMED mat-vec's weight path (mmvq_stream.h).
//
// What bounded the columns so far was the weight stream, not the arithmetic: the matrix-core kernel (mmvq_cols_mfma.hip) pulls its 16-row
// tiles with per-lane 16-byte loads 64 contiguous bytes at a time and reached 3.5 TB/s (Q4_K 4096 x 14336, n = 8: 16-22 us = 1.5-2.0 TB/s with
// its image copy). Here the weights come the way the n = 1 kernel gets them — one loader wave per workgroup, LDS-DMA of the workgroup's
// contiguous row range, 1 KiB per instruction into a ring of slots — and the eight consumer waves multiply every unit (a lane = one
// 256-weight block, its bytes read from the slot ONCE) with all the columns: the CPU's vec_dot per (block, column), same integer sub-sums
// (st_unit<T>::dot), n times. That is ~90 vector instructions per block and column: 8 columns at 2.5 TB/s keep the four SIMDs of a CU a
// third busy, so the matrix cores are not needed below ~5 TB/s.
//
// LDS: the n column images (int8 blocks + the eight 32-element sums as (h, l) bytes [+ the sixteen 16-element sums for Q6_K]: 272 / 304
// bytes per block — odd numbers of 16-byte chunks) + the ring. At k = 14336 eight Q4_K columns leave three 9 KiB slots; what does not
// leave two slots is not taken (the caller falls back to the matrix-core kernel).
// Reduction: 16 lanes (one DPP row) when the row's block count is a multiple of 16, else 8 lanes (a multiple of 8 is required);
// partial sums in LDS, added per (row, column) in a fixed order after the last slot.
// Roofline: HBM; algorithmic bytes = m * row_size (the weights once, whatever n).
#include <hip/hip_ext.h>

#include "mmvq_stream.h"

#include <limits.h>
#include <algorithm>

namespace mi355x {

struct st_cols {
    int n;                      // real columns (<= N)
    int act_stride;             // bytes per block in a column image: 272 (Q4_K / Q5_K: no 16-element sums) or 304
    int col_stride;             // bytes between column images
    int nbp;                    // floats per column in the scale array
    int npc;                    // partial sums per column
    const int8_t * a_qs; const float * a_d; const int16_t * a_bs;      // act_q8 (Q8_K), n columns
    float * dst; size_t dst_col_stride;
};

template <int TYPE, int N, bool NT>
__global__ void __launch_bounds__(ST_THREADS, 1) k_mmvq_stream_cols(const st_args p, const st_cols c) {
    typedef st_unit<TYPE> U;
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const st_group & g = p.g[0];
    const int wg = blockIdx.x, nwg = gridDim.x, nb = p.nb;
    // carve: sync | images | scales | partials | ring
    st_lds L;
    L.sync = (uint32_t *) lds;
    L.act = lds + 2*ST_SYNC_WORDS*4;
    L.dd = (float *) (L.act + (size_t) c.n*c.col_stride);
    L.red = nullptr;
    L.part = L.dd + (size_t) c.n*c.nbp;
    L.ring_a = st_lds_addr((char *) (((size_t)(L.part + (size_t) N*c.npc) + 15) & ~(size_t) 15));
    L.slot_stride = ((64*U::UB + 1023)/1024)*1024; L.S = p.S;
    if (threadIdx.x < ST_SYNC_WORDS) L.sync[threadIdx.x] = 0;
    int r0, R; st_rows(g, wg, nwg, r0, R);
    const int n1 = R*nb, nslots = (n1 + 63) >> 6;
    if (wave == ST_NC) {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        st_loader_state ls = { 0, 0, 0, 0, 0, 0, 0 };
        st_loader_phase<TYPE, NT>(p, g, wg, nwg, L, 0, ls, lane, 0);
        st_loader_drain(L, nslots, ls, lane);
        return;
    }
    const int ctid = threadIdx.x;
    // ---- the column images: every 16-byte chunk of the n columns' quants, eight in flight per thread; the sums and scales per (column, block) ----
    {
        const int qpc = p.k >> 4, tot = c.n*qpc;             // chunks per column, in all
        constexpr int B = 8;
        int4v t[B];
        const int nbat = (tot + B*ST_NC*64 - 1)/(B*ST_NC*64);
        // sums / scales of (column, block) pairs: one pair per thread and trip
        const int npair = c.n*nb;
        for (int b = 0; b < nbat; b++) {
            const int i0 = ctid + b*B*ST_NC*64;
#pragma unroll
            for (int u = 0; u < B; u++) {
                const int i = min(i0 + u*ST_NC*64, tot - 1), col = i/qpc, q = i - col*qpc;
                t[u] = *(const int4v *) (c.a_qs + (size_t) col*p.k + (size_t) q*16);
            }
            if (b == 0) { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); __builtin_amdgcn_s_barrier(); }      // (the loader starts behind the first batch of requests)
#pragma unroll
            for (int u = 0; u < B; u++) {
                const int i = i0 + u*ST_NC*64;
                if (i < tot) { const int col = i/qpc, q = i - col*qpc; *(int4v *) (L.act + (size_t) col*c.col_stride + (size_t)(q >> 4)*c.act_stride + (q & 15)*16) = t[u]; }
            }
        }
        for (int pi = ctid; pi < npair; pi += ST_NC*64) {
            const int col = pi/nb, ib = pi - col*nb;
            const int16_t * bs = c.a_bs + ((size_t) col*nb + ib)*16;
            const int4v b0 = *(const int4v *) bs, b1 = *(const int4v *) (bs + 8);
            const float dv = c.a_d[(size_t) col*nb + ib];
            uint32_t h32[2] = { 0, 0 }, l32[2] = { 0, 0 }, h16[4] = { 0, 0, 0, 0 }, l16[4] = { 0, 0, 0, 0 };
#pragma unroll
            for (int j = 0; j < 8; j++) {
                const uint32_t wsum = (uint32_t)(j < 4 ? b0[j] : b1[j - 4]);
                const int sa = (int)(int16_t)(wsum & 0xFFFF), sb = (int)(int16_t)(wsum >> 16);
                int h, l;
                st_hl(sa + sb, h, l); h32[j >> 2] |= (uint32_t)(h & 0xFF) << (8*(j & 3)); l32[j >> 2] |= (uint32_t)(l & 0xFF) << (8*(j & 3));
                st_hl(sa, h, l); h16[j >> 1] |= (uint32_t)(h & 0xFF) << (8*((2*j) & 3)); l16[j >> 1] |= (uint32_t)(l & 0xFF) << (8*((2*j) & 3));
                st_hl(sb, h, l); h16[j >> 1] |= (uint32_t)(h & 0xFF) << (8*((2*j + 1) & 3)); l16[j >> 1] |= (uint32_t)(l & 0xFF) << (8*((2*j + 1) & 3));
            }
            char * ab = L.act + (size_t) col*c.col_stride + (size_t) ib*c.act_stride;
            *(int4v *) (ab + 256) = int4v{ (int) h32[0], (int) h32[1], (int) l32[0], (int) l32[1] };
            if (TYPE == T_Q6_K) {
                *(int4v *) (ab + 272) = int4v{ (int) h16[0], (int) h16[1], (int) h16[2], (int) h16[3] };
                *(int4v *) (ab + 288) = int4v{ (int) l16[0], (int) l16[1], (int) l16[2], (int) l16[3] };
            }
            L.dd[(size_t) col*c.nbp + ib] = dv;
        }
    }
    st_consumers_meet(&L.sync[2], lane, 0);

    // ---- the stream: a lane owns one unit of the slot and multiplies it with every column ----
    const bool row16 = (nb & 15) == 0;
    const int gsh = row16 ? 4 : 3;                           // partial sums per 16 or 8 units
    const int npr = nb >> gsh;
    const uint32_t magic = p.magic;
    const int S = L.S;
    int ring_i = wave % S;
    for (int i = wave; i < nslots; i += ST_NC) {
        const int u = i*64 + lane;
        const bool live = u < n1;
        const int uc = live ? u : n1 - 1;
        const int ib = nb == 1 ? 0 : uc - (int) __umulhi((uint32_t) uc, magic)*nb;
        st_wait_ge(&L.sync[0], (uint32_t)(i + 1));
        const typename U::wfrag w = U::load(L.ring_a + (uint32_t) ring_i*L.slot_stride + (uint32_t)(live ? lane : 0)*U::UB);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if (lane == 0) st_flag_st(&L.sync[16 + ring_i], (uint32_t)(i + 1));
        ring_i += ST_NC; while (ring_i >= S) ring_i -= S;
        const char * ab = L.act + (size_t) ib*c.act_stride;
#pragma unroll
        for (int col = 0; col < N; col++) {
            const int cc = min(col, c.n - 1);
            float res = U::dot(w, ab + (size_t) cc*c.col_stride, L.dd[(size_t) cc*c.nbp + ib]);
            if (!live) res = 0.0f;
            if (row16) res = row16_sum(res);
            else { res += dpp_f<0xB1>(res); res += dpp_f<0x4E>(res); res += dpp_f<0x141>(res); }      // 8 lanes: quad swaps, then row_half_mirror
            if ((lane & ((1 << gsh) - 1)) == 0 && live) L.part[(size_t) col*c.npc + (u >> gsh)] = res;
        }
    }
    st_consumers_meet(&L.sync[3], lane, 0);

    // ---- (row, column): partials added in a fixed order ----
    for (int e = ctid; e < R*c.n; e += ST_NC*64) {
        const int col = e/R, rr = e - col*R;
        *(float *) ((char *) c.dst + (size_t) col*c.dst_col_stride + (size_t)(r0 + rr)*4) = st_row_sum(L.part + (size_t) col*c.npc, rr, npr);
    }
}

static int stc_cu_count() {
    static int n_cu = 0;
    if (n_cu == 0) {
        int dev = 0; hipDeviceProp_t prop;
        if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) n_cu = prop.multiProcessorCount;
        if (n_cu <= 0) n_cu = 256;
    }
    return n_cu;
}

template <int TYPE, int N>
static void stc_launch(const st_args & a, const st_cols & c, int blocks, size_t lds, hipStream_t stream) {
    MI_LDS_LIMIT_OR_DIE(163840, k_mmvq_stream_cols<TYPE, N, true>);
    hipLaunchKernelGGL((k_mmvq_stream_cols<TYPE, N, true>), dim3((unsigned) blocks), dim3(ST_THREADS), lds, stream, a, c);
}

// false = not taken (format, shape, LDS): the caller runs another kernel
bool mul_mat_vec_q_stream_cols(int type_a, const void * W, size_t w_row_stride, int64_t m, int64_t k,
                               const act_q8 & act, int64_t n, float * dst, size_t dst_col_stride_bytes, hipStream_t stream) {
    static int on = -1;
    if (on < 0) { const char * e = getenv("GGML_MI355X_STREAM_COLS"); on = e ? atoi(e) : 1; }
    if (!on || !mul_mat_vec_q_stream_enabled() || n < 2 || n > 8 || act.kind != T_Q8_K) return false;
    // Q6_K: built and tested (GGML_MI355X_STREAM_COLS=2 routes it here), not routed by default — its unit costs twice the vector instructions of
    // Q4_K's per column (6-bit unpack, 16-element scale groups) and the kernel is instruction-bound from 3 columns on: 23.7 / 24.1 / 39.9 us at
    // n = 3 / 4 / 5 (4096 x 14336) against 17.7 / 19.6 / 20.7 for the kernels it would replace
    if (type_a == T_Q6_K && on < 2) return false;
    const int ub = type_a == T_Q4_K ? 144 : type_a == T_Q5_K ? 176 : type_a == T_Q6_K ? 210 : 0;
    if (!ub || k % 2048 != 0 || k > 16384 || m < 1 || m >= (1ll << 23)) return false;       // (a row's blocks in groups of 8 or 16 lanes)
    const int nb = (int)(k/256);
    if (w_row_stride != (size_t) nb*ub || ((uintptr_t) W % 16) || ((uintptr_t) act.qs % 16) || ((uintptr_t) act.bsums % 16) || ((uintptr_t) dst % 4) || dst_col_stride_bytes % 4) return false;
    st_args a = st_args{};
    a.n_groups = 1; a.k = (int) k; a.nb = nb; a.mode = PRO_Q8;
    a.magic = nb == 1 ? 0u : (uint32_t)((0x100000000ull + nb - 1)/nb);
    for (int i = 0; i < MMVQ_MAX_GROUPS; i++) a.block_end[i] = INT_MAX;
    st_group & s = a.g[0];
    s.W = (const char *) W; s.m = (int) m; s.type = type_a; s.epi = EPI_NONE;
    s.ralign = 1;
    while (((int64_t) s.ralign*nb*ub) % 16 != 0) s.ralign *= 2;
    const int nru = std::max(1, (int)(m/s.ralign));
    const int blocks = std::min(stc_cu_count(), nru);
    a.block_end[0] = blocks;
    const int Rmax = ((nru + blocks - 1)/blocks)*s.ralign + (int)(m - (m/s.ralign)*s.ralign);
    st_cols c = {};
    c.n = (int) n;
    c.act_stride = type_a == T_Q6_K ? 304 : 272;
    c.col_stride = nb*c.act_stride;
    c.nbp = (nb + 3) & ~3;
    const int gsh = (nb & 15) == 0 ? 4 : 3;
    c.npc = Rmax*(nb >> gsh);
    c.a_qs = act.qs; c.a_d = act.d; c.a_bs = act.bsums; c.dst = dst; c.dst_col_stride = dst_col_stride_bytes;
    const int N = n <= 2 ? 2 : n <= 4 ? 4 : 8;
    const size_t fixed = 2*ST_SYNC_WORDS*4 + (size_t) n*c.col_stride + (size_t) n*c.nbp*4 + (size_t) N*c.npc*4 + 16;
    const int slot = ((64*ub + 1023)/1024)*1024;
    const int nslots_max = (int)(((int64_t) Rmax*nb + 63)/64);
    if (fixed + (size_t) std::min(nslots_max, 2)*slot > 163840) return false;
    int S = (int)((163840 - (int64_t) fixed)/slot);
    if (S > nslots_max) S = nslots_max;
    if (S > ST_MAX_RING) S = ST_MAX_RING;
    a.S = S;
    const size_t lds = fixed + (size_t) S*slot;
#define MI_STC(T_) do { if (N == 2) stc_launch<T_, 2>(a, c, blocks, lds, stream); else if (N == 4) stc_launch<T_, 4>(a, c, blocks, lds, stream); else stc_launch<T_, 8>(a, c, blocks, lds, stream); } while (0)
    if (type_a == T_Q4_K) MI_STC(T_Q4_K); else if (type_a == T_Q5_K) MI_STC(T_Q5_K); else MI_STC(T_Q6_K);
#undef MI_STC
    return true;
}

} // namespace mi355x
