// mmvq_cols_mfma.hip — K-quant weights x 2..8 quantized activation columns on the int8 matrix cores (gfx950).
//
// For n = 1 the dot products of a row are VALU work hidden under the HBM stream (mmvq_fused.h). With n columns that work grows n-fold
// while the weight bytes stay: the round-1 kernels (mmvq.hip) were VALU-bound from n = 3 on (Q4_K, 4096 x 14336: 2.7 TB/s at n = 1, 1.1 TB/s
// at n = 8, profiles/r01_n_op_perf_reference_cases.json). Here the integer dots of a 32-element sub-block go to
// v_mfma_i32_16x16x32_i8: one instruction multiplies the sub-block of 16 weight rows with 16 activation columns (n <= 8 of them real),
// and the VALU keeps only what the CPU's vec_dot does per sub-block on integers: isum += scale*dot, msum += min*bsum, and per 256-block
// the two f32 multiply-adds (ggml_vec_dot_q4_K_q8_K, ggml-cpu/quants.c: sumf += d*isum - dmin*msum) — same arithmetic, same integers.
//
// Operands. MFMA "A" (16 x 32) = activations: lane l supplies column l & 15 (clamped to n - 1), bytes 8*(l >> 4) .. +7 of the sub-block,
// read from the LDS image; "B" (32 x 16) = weights: lane l supplies weight row l & 15 of the tile, the same 8 k-positions, unpacked from the
// bytes the lane itself loaded (no cross-lane movement: in every K-quant a 32-element sub-block is 32 contiguous nibbles / bytes).
// D: lane l holds weight row l & 15 x columns 4*(l >> 4) .. +3 — the lane's scales are those of the row whose bytes it loaded.
//
// Work split: 16-row tiles, grid-strided over one 8-wave workgroup per CU; the 8 waves of a workgroup split the k blocks of the tile
// (block b -> wave b % 8), their f32 partial sums meet in LDS. Weight loads run two blocks ahead in a static ring of register sets; the
// stream of (tile, block) steps of a wave continues across tiles. Roofline: HBM (weight bytes once); algorithmic bytes = m * row_size.
#include "blocks.h"
#include "dev_common.h"
#include "kernels.h"

namespace mi355x {

typedef int   i32x4 __attribute__((ext_vector_type(4)));
typedef short i16x2 __attribute__((ext_vector_type(2)));

struct colmf_args {
    const char * W; size_t w_row_stride; int m, k, n;
    const int8_t * a_qs; const float * a_d; const int16_t * a_bs;
    float * dst; size_t dst_col_stride;     // bytes
    int col_stride;                          // bytes between the columns' images in LDS (= 32 mod 256: conflict-free operand reads)
    int off_d, off_bs;                       // within a column image
    int n_tiles;
    int dbg;                                 // timing experiments (GGML_MI355X_COLS_DBG): 1 no math, 2 no weight loads inside the loop, 4 no image copy
};

static __device__ __forceinline__ i32x4 mfma_i8(int2v a, int2v b, i32x4 c = i32x4{ 0, 0, 0, 0 }) {
    return __builtin_amdgcn_mfma_i32_16x16x32_i8(__builtin_bit_cast(long, a), __builtin_bit_cast(long, b), c, 0, 0, 0);
}
static __device__ __forceinline__ int mad24(int a, int b, int c) { return __mul24(a, b) + c; }
// a.lo*b.lo + a.hi*b.hi + c on packed int16 pairs. The operands arrive as SCALAR parameters: __builtin_bit_cast applied directly to an
// element of an ext-vector (bit_cast(i16x2, v.y)) reads element 0 with this compiler (ROCm 7.2 clang)
static __device__ __forceinline__ int dot2_i16(int a, int b, int c) {
    return __builtin_amdgcn_sdot2(__builtin_bit_cast(i16x2, a), __builtin_bit_cast(i16x2, b), c, false);
}

template <int TYPE> struct colmf_t;

// ---- Q4_K: [d f16][dmin f16][scales 12][qs 128]; sub-block pair P: qs[32P .. 32P+31], low nibbles = sub-block 2P, high = 2P+1.
//      16-byte loads: load j (0, 1) covers qs[64j .. 64j+63] = pairs 2j, 2j+1; lane group g takes bytes 16g .. 16g+15 of it, i.e. elements
//      16(g&1) .. +15 of pair 2j + (g>>1): 64 contiguous bytes per row and instruction (8-byte loads, 32 contiguous bytes per row, streamed
//      at 3 TB/s). An MFMA's four k-groups then carry two different pairs: two MFMAs, each with the other pair's activations zeroed ----
template <> struct colmf_t<T_Q4_K> {
    static constexpr int BLOCK_BYTES = 144;
    struct wreg { int4v hdr; int4v q[2]; };
    static __device__ __forceinline__ wreg load(const char * b, int g) {
        wreg w;
        w.hdr = ld_b128(b);
        w.q[0] = ld_b128(b + 16 + 16*g); w.q[1] = ld_b128(b + 80 + 16*g);
        return w;
    }
};
// ---- Q5_K: [d][dmin][scales 12][qh 32][qs 128]; bit s of qh[e] = 5th bit of element e of sub-block s ----
template <> struct colmf_t<T_Q5_K> {
    static constexpr int BLOCK_BYTES = 176;
    struct wreg { int4v hdr; int4v qh; int4v q[2]; };
    static __device__ __forceinline__ wreg load(const char * b, int g) {
        wreg w;
        w.hdr = ld_b128(b);
        w.qh = ld_b128(b + 16 + 16*(g & 1));
        w.q[0] = ld_b128(b + 48 + 16*g); w.q[1] = ld_b128(b + 112 + 16*g);
        return w;
    }
};
// ---- Q6_K: [ql 128][qh 64][scales 16 x int8][d f16]; half h (128 elements): for l < 32: element l = ql[64h+l] & 15 | (qh[32h+l] & 3) << 4,
//      l+32 = ql[64h+32+l] & 15 | (qh >> 2 & 3) << 4, l+64 = ql[64h+l] >> 4 | (qh >> 4 & 3) << 4, l+96 = ql[64h+32+l] >> 4 | (qh >> 6) << 4;
//      one int8 scale per 16 elements. The lane of k-group g takes l = 8g .. 8g+7: groups 0, 1 fall into the first scale of each
//      32-element run, groups 2, 3 into the second ----
template <> struct colmf_t<T_Q6_K> {
    static constexpr int BLOCK_BYTES = 210;
    struct wreg { int2v qla[2], qlb[2], qh[2]; int4v sc; uint32_t d; };
    static __device__ __forceinline__ wreg load(const char * b, int g) {
        wreg w;
#pragma unroll
        for (int h = 0; h < 2; h++) {
            w.qla[h] = ld_b64(b + 64*h + 8*g);
            w.qlb[h] = ld_b64(b + 64*h + 32 + 8*g);
            w.qh[h]  = ld_b64(b + 128 + 32*h + 8*g);
        }
        w.sc = ld_b128(b + 192);
        w.d  = ld_u16(b + 208);
        return w;
    }
};

// scales and mins of a Q4_K / Q5_K block as 2 x 4 packed bytes each (get_scale_min_k4, ggml-quants.c)
static __device__ __forceinline__ void k4_unpack(const int4v & hdr, uint32_t (&sc)[2], uint32_t (&mn)[2]) {
    const uint32_t s0 = (uint32_t) hdr.y, s1 = (uint32_t) hdr.z, s2 = (uint32_t) hdr.w;
    sc[0] = s0 & 0x3F3F3F3Fu; mn[0] = s1 & 0x3F3F3F3Fu;
    sc[1] = (s2 & 0x0F0F0F0Fu) | ((s0 >> 2) & 0x30303030u);
    mn[1] = ((s2 >> 4) & 0x0F0F0F0Fu) | ((s1 >> 2) & 0x30303030u);
}

template <int TYPE>
__global__ void __launch_bounds__(512, 1) k_mmvq_cols_mfma(const colmf_args p) {
    typedef colmf_t<TYPE> T;
    constexpr int D = TYPE == T_Q6_K ? 3 : 4, NW = 8;      // ring depth: a tile is only k/256/8 blocks per wave (7 at k = 14336): two in flight left the HBM pipe half empty
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int r16 = lane & 15, g = lane >> 4;
    const int nb = p.k >> 8;
    const int bpw = (nb + NW - 1)/NW;                         // blocks per wave per tile
    const int my_tiles = (int) blockIdx.x < p.n_tiles ? (p.n_tiles - 1 - (int) blockIdx.x)/(int) gridDim.x + 1 : 0;
    const int total = my_tiles*bpw;

    // ---- weight prefetch first (HBM), then the activation images (L2) ----
    typename T::wreg w[D];
    int j_pf = 0, i_pf = 0;
#define CM_FETCH(d_) { \
        const bool live = j_pf < my_tiles; \
        const int tile = live ? (int) blockIdx.x + j_pf*(int) gridDim.x : (int) blockIdx.x; \
        const int blk = live ? min(wave + NW*i_pf, nb - 1) : 0; \
        const int row = min(tile*16 + r16, p.m - 1); \
        w[d_] = T::load(p.W + (size_t) row*p.w_row_stride + (size_t) blk*T::BLOCK_BYTES, g); \
        if (++i_pf == bpw) { i_pf = 0; j_pf++; } }
    {   // column images: [qs k][d k/256 f32][bsums k/16 i16], each region 16-byte padded, contiguous in LDS; columns col_stride apart.
        // One flat list of 16-byte chunks over all columns, eight loads in flight per thread before the first LDS store (a
        // load-wait-store loop per region and column paid 3 L2 round trips per column: 13 us of a 27 us launch at n = 8); the weight
        // ring is requested right after the first batch of image loads
        const int qs_c = (int)((((size_t) p.k + 15) & ~(size_t) 15) >> 4), d_c = (nb*4 + 15) >> 4, bs_c = ((p.k >> 3) + 15) >> 4, cpc = qs_c + d_c + bs_c;
        int tot = (p.dbg & 4) ? 0 : p.n*cpc;
        if (p.dbg & 8) {      // the plain copy (debug)
            tot = 0;
            const size_t qs_b = (size_t) qs_c*16, d_b = (size_t) d_c*16, bs_b = (size_t) bs_c*16;
            for (int c = 0; c < p.n; c++) {
                char * base = smem + (size_t) c*p.col_stride;
                const char * g_qs = (const char *) (p.a_qs + (size_t) c*p.k); const char * g_d = (const char *) (p.a_d + (size_t) c*nb);
                const char * g_bs = (const char *) (p.a_bs + (size_t) c*(p.k/16));
                for (size_t i = (size_t) threadIdx.x*16; i < qs_b; i += 512*16) *(int4v *) (base + i) = ld_b128(g_qs + i);
                for (size_t i = (size_t) threadIdx.x*4;  i < d_b;  i += 512*4)  *(uint32_t *) (base + p.off_d + i) = ld_u32(g_d + i);
                for (size_t i = (size_t) threadIdx.x*4;  i < bs_b; i += 512*4)  *(uint32_t *) (base + p.off_bs + i) = ld_u32(g_bs + i);
            }
        }
        auto src_of = [&](int idx) -> const char * {
            const int c = idx/cpc, jj = idx - c*cpc;
            return jj < qs_c ? (const char *) p.a_qs + (size_t) c*p.k + (size_t) jj*16
                 : jj < qs_c + d_c ? (const char *) p.a_d + (size_t) c*nb*4 + (size_t)(jj - qs_c)*16
                 : (const char *) p.a_bs + (size_t) c*(p.k >> 3) + (size_t)(jj - qs_c - d_c)*16;
        };
        auto dst_of = [&](int idx) -> char * { const int c = idx/cpc, jj = idx - c*cpc; return smem + (size_t) c*p.col_stride + (size_t) jj*16; };
        const int last = max(tot - 1, 0);
        // software-pipelined in batches of 8 chunks per thread: batch b + 1 is requested before batch b is stored
        // batches of 8 chunks per thread (load all, then store all); the weight ring is requested right after the first batch's loads.
        // Measured on Q4_K 4096 x 14336 (rocprofv3, n = 2 / 4 / 8): this order 13.9 / 14.1 / 16.2 us; both of n = 8's batches requested
        // before the ring 15.3 / 17.0 / 20.0 (the ring starts late); the second batch requested before the first is stored, behind the
        // ring, 14.2 / 14.8 / 18.6 (it comes back with the ring's HBM latency: loads return in request order)
        int4v ta[8];
        const int nbat = (tot + 8*512 - 1)/(8*512);          // workgroup-uniform: no slice is requested that nobody stores
        for (int b = 0; b < nbat || b == 0; b++) {
            const int i0 = (int) threadIdx.x + b*8*512;
#pragma unroll
            for (int u = 0; u < 8; u++) if ((b*8 + u)*512 < tot) ta[u] = ld_b128(src_of(min(i0 + u*512, last)));
            if (b == 0) {
                asm volatile("" ::: "memory");
#pragma unroll
                for (int d = 0; d < D; d++) CM_FETCH(d)
                asm volatile("" ::: "memory");
            }
#pragma unroll
            for (int u = 0; u < 8; u++) if (i0 + u*512 < tot) *(int4v *) dst_of(i0 + u*512) = ta[u];
            asm volatile("" ::: "memory");
        }
    }
    __syncthreads();

    const char * a_col  = smem + (size_t) min(r16, p.n - 1)*p.col_stride;          // this lane's MFMA-A column
    const char * o_col[4];                                                           // the four columns of this lane's outputs
#pragma unroll
    for (int i = 0; i < 4; i++) o_col[i] = smem + (size_t) min(4*g + i, p.n - 1)*p.col_stride;
    float * red = (float *) (smem + (size_t) p.n*p.col_stride);                      // [2][NW][64][4] f32 partial sums

    float acc[4] = { 0.0f, 0.0f, 0.0f, 0.0f };
    int it = 0, j_cur = 0;
    for (int s = 0; s < total; s += D) {
#pragma unroll
        for (int d = 0; d < D; d++) {
            if (s + d < total) {        // wave-uniform
                const int blk = wave + NW*it;
                if (p.dbg & 1) { acc[0] += __builtin_bit_cast(float, ((const int *) &w[d])[0] ^ ((const int *) &w[d])[sizeof(typename T::wreg)/4 - 1]); }
                else if (blk < nb) {
                    const typename T::wreg & x = w[d];
                    const char * aq = a_col + (size_t) blk*256 + 8*g;
                    int isum[4] = { 0, 0, 0, 0 }, msum[4] = { 0, 0, 0, 0 };
                    float dw, dm = 0.0f;
                    if constexpr (TYPE == T_Q4_K || TYPE == T_Q5_K) {
                        uint32_t sc[2], mn[2];
                        k4_unpack(x.hdr, sc, mn);
                        dw = f16_bits_to_f32((uint16_t)((uint32_t) x.hdr.x & 0xFFFF)); dm = f16_bits_to_f32((uint16_t)((uint32_t) x.hdr.x >> 16));
                        int4v bsv[4][2];
#pragma unroll
                        for (int i = 0; i < 4; i++) { bsv[i][0] = *(const int4v *) (o_col[i] + p.off_bs + blk*32); bsv[i][1] = *(const int4v *) (o_col[i] + p.off_bs + blk*32 + 16); }
                        const int4v zz = { 0, 0, 0, 0 };
                        const bool gA = g < 2;
#pragma unroll
                        for (int j2 = 0; j2 < 2; j2++) {
                            const int4v qv = x.q[j2];
                            int4v lo = { qv.x & 0x0F0F0F0F, qv.y & 0x0F0F0F0F, qv.z & 0x0F0F0F0F, qv.w & 0x0F0F0F0F };
                            int4v hi = { (qv.x >> 4) & 0x0F0F0F0F, (qv.y >> 4) & 0x0F0F0F0F, (qv.z >> 4) & 0x0F0F0F0F, (qv.w >> 4) & 0x0F0F0F0F };
                            const int P = 2*j2 + (g >> 1);                 // this lane's sub-block pair
                            if constexpr (TYPE == T_Q5_K) {
                                const uint32_t sl = 2*P, sh = 2*P + 1;
                                lo.x |= (int)((((uint32_t) x.qh.x >> sl) & 0x01010101u) << 4); lo.y |= (int)((((uint32_t) x.qh.y >> sl) & 0x01010101u) << 4);
                                lo.z |= (int)((((uint32_t) x.qh.z >> sl) & 0x01010101u) << 4); lo.w |= (int)((((uint32_t) x.qh.w >> sl) & 0x01010101u) << 4);
                                hi.x |= (int)((((uint32_t) x.qh.x >> sh) & 0x01010101u) << 4); hi.y |= (int)((((uint32_t) x.qh.y >> sh) & 0x01010101u) << 4);
                                hi.z |= (int)((((uint32_t) x.qh.z >> sh) & 0x01010101u) << 4); hi.w |= (int)((((uint32_t) x.qh.w >> sh) & 0x01010101u) << 4);
                            }
                            // this lane's activations: elements 16(g&1) .. +15 of sub-blocks 2P (with the low nibbles) and 2P + 1 (high)
                            const char * ap = a_col + (size_t) blk*256 + 64*P + 16*(g & 1);
                            const int4v al = *(const int4v *) ap, ah = *(const int4v *) (ap + 32);
                            const int4v alA = gA ? al : zz, alB = gA ? zz : al, ahA = gA ? ah : zz, ahB = gA ? zz : ah;
                            // pair 2*j2 (k-groups 0, 1) and pair 2*j2 + 1 (k-groups 2, 3): two 8-element k-chunks per lane each
                            i32x4 dAl = mfma_i8(int2v{ alA.x, alA.y }, int2v{ lo.x, lo.y }); dAl = mfma_i8(int2v{ alA.z, alA.w }, int2v{ lo.z, lo.w }, dAl);
                            i32x4 dAh = mfma_i8(int2v{ ahA.x, ahA.y }, int2v{ hi.x, hi.y }); dAh = mfma_i8(int2v{ ahA.z, ahA.w }, int2v{ hi.z, hi.w }, dAh);
                            i32x4 dBl = mfma_i8(int2v{ alB.x, alB.y }, int2v{ lo.x, lo.y }); dBl = mfma_i8(int2v{ alB.z, alB.w }, int2v{ lo.z, lo.w }, dBl);
                            i32x4 dBh = mfma_i8(int2v{ ahB.x, ahB.y }, int2v{ hi.x, hi.y }); dBh = mfma_i8(int2v{ ahB.z, ahB.w }, int2v{ hi.z, hi.w }, dBh);
                            // sub-blocks 4*j2 .. 4*j2 + 3 = bytes 0..3 of sc[j2] / mn[j2]
                            const uint32_t scw = sc[j2], mnw = mn[j2];
                            const int s0 = (int)(scw & 0xFF), s1 = (int)((scw >> 8) & 0xFF), s2 = (int)((scw >> 16) & 0xFF), s3 = (int)(scw >> 24);
                            const int m0 = (int)(mnw & 0xFF), m1 = (int)((mnw >> 8) & 0xFF), m2 = (int)((mnw >> 16) & 0xFF), m3 = (int)(mnw >> 24);
                            const int mm0 = m0 | (m0 << 16), mm1 = m1 | (m1 << 16), mm2 = m2 | (m2 << 16), mm3 = m3 | (m3 << 16);
#pragma unroll
                            for (int i = 0; i < 4; i++) {
                                isum[i] = mad24(dAl[i], s0, isum[i]); isum[i] = mad24(dAh[i], s1, isum[i]);
                                isum[i] = mad24(dBl[i], s2, isum[i]); isum[i] = mad24(dBh[i], s3, isum[i]);
                                // bsums of sub-block sb = the two int16 of dword sb of the block's 16 bsums: m*(lo + hi) as a 2-way i16 dot
                                const int4v & bv = bsv[i][j2];
                                msum[i] = dot2_i16(bv.x, mm0, msum[i]); msum[i] = dot2_i16(bv.y, mm1, msum[i]);
                                msum[i] = dot2_i16(bv.z, mm2, msum[i]); msum[i] = dot2_i16(bv.w, mm3, msum[i]);
                            }
                        }
                    } else {      // Q6_K
                        dw = f16_bits_to_f32((uint16_t) x.d);
                        const bool first = g < 2;      // k-groups 0, 1: first 16 elements of a 32-element run (scale 2e), groups 2, 3: the second (scale 2e + 1)
#pragma unroll
                        for (int h = 0; h < 2; h++) {
                            const uint32_t qa0 = (uint32_t) x.qla[h].x, qa1 = (uint32_t) x.qla[h].y, qb0 = (uint32_t) x.qlb[h].x, qb1 = (uint32_t) x.qlb[h].y;
                            const uint32_t qh0 = (uint32_t) x.qh[h].x, qh1 = (uint32_t) x.qh[h].y;
                            int2v e[4];
                            e[0].x = (int)((qa0 & 0x0F0F0F0Fu) | ((qh0 << 4) & 0x30303030u)); e[0].y = (int)((qa1 & 0x0F0F0F0Fu) | ((qh1 << 4) & 0x30303030u));
                            e[1].x = (int)((qb0 & 0x0F0F0F0Fu) | ((qh0 << 2) & 0x30303030u)); e[1].y = (int)((qb1 & 0x0F0F0F0Fu) | ((qh1 << 2) & 0x30303030u));
                            e[2].x = (int)(((qa0 >> 4) & 0x0F0F0F0Fu) | (qh0 & 0x30303030u)); e[2].y = (int)(((qa1 >> 4) & 0x0F0F0F0Fu) | (qh1 & 0x30303030u));
                            e[3].x = (int)(((qb0 >> 4) & 0x0F0F0F0Fu) | ((qh0 >> 2) & 0x30303030u)); e[3].y = (int)(((qb1 >> 4) & 0x0F0F0F0Fu) | ((qh1 >> 2) & 0x30303030u));
                            const uint32_t scw0 = (uint32_t)(h ? x.sc.z : x.sc.x), scw1 = (uint32_t)(h ? x.sc.w : x.sc.y);   // scales 8h .. 8h+7
#pragma unroll
                            for (int r = 0; r < 4; r++) {        // 32-element run r of the half: elements 128h + 32r + (0..31), scales 8h + 2r, 8h + 2r + 1
                                const int2v a = *(const int2v *) (aq + 128*h + 32*r);
                                const int2v z = { 0, 0 };
                                // two products, each with the other scale's k-groups zeroed on the activation side
                                const i32x4 dA = mfma_i8(first ? a : z, e[r]), dB = mfma_i8(first ? z : a, e[r]);
                                const uint32_t sw = r < 2 ? scw0 : scw1;
                                const int sA = (int)(int8_t)((sw >> (16*(r & 1))) & 0xFF), sB = (int)(int8_t)((sw >> (16*(r & 1) + 8)) & 0xFF);
#pragma unroll
                                for (int i = 0; i < 4; i++) {
                                    // (q - 32) . a = q . a - 32 * bsum16
                                    const int16_t * bs = (const int16_t *) (o_col[i] + p.off_bs) + blk*16 + 8*h + 2*r;
                                    isum[i] = mad24(dA[i] - 32*(int) bs[0], sA, isum[i]);
                                    isum[i] = mad24(dB[i] - 32*(int) bs[1], sB, isum[i]);
                                }
                            }
                        }
                    }
#pragma unroll
                    for (int i = 0; i < 4; i++) {
                        const float da = ((const float *) (o_col[i] + p.off_d))[blk];
                        acc[i] += (dw*da)*(float) isum[i];
                        if constexpr (TYPE != T_Q6_K) acc[i] -= (dm*da)*(float) msum[i];
                    }
                }
                if (!(p.dbg & 2)) CM_FETCH(d)
                if (++it == bpw) {
                    // ---- the tile is complete in this wave: partial sums meet in LDS, wave 0 adds them in wave order and stores ----
                    float * rb = red + (size_t)(j_cur & 1)*NW*256;
                    *(float4v *) (rb + ((size_t) wave*64 + lane)*4) = float4v{ acc[0], acc[1], acc[2], acc[3] };
                    __syncthreads();
                    if (wave == 0) {
                        float4v t = *(const float4v *) (rb + (size_t) lane*4);
#pragma unroll
                        for (int ww = 1; ww < NW; ww++) { const float4v u = *(const float4v *) (rb + ((size_t) ww*64 + lane)*4); t.x += u.x; t.y += u.y; t.z += u.z; t.w += u.w; }
                        const int tile = (int) blockIdx.x + j_cur*(int) gridDim.x;
                        const int row = tile*16 + r16;
                        if (row < p.m) {
                            const float tv[4] = { t.x, t.y, t.z, t.w };
#pragma unroll
                            for (int i = 0; i < 4; i++) if (4*g + i < p.n) *(float *) ((char *) p.dst + (size_t)(4*g + i)*p.dst_col_stride + (size_t) row*4) = tv[i];
                        }
                    }
                    it = 0; j_cur++;
                    acc[0] = acc[1] = acc[2] = acc[3] = 0.0f;
                }
            }
        }
    }
#undef CM_FETCH
}

template <int TYPE>
static bool launch_cols_mfma(colmf_args & a, size_t lds, hipStream_t stream) {
    if (!MI_LDS_LIMIT(156*1024, k_mmvq_cols_mfma<TYPE>)) return false;
    static int n_cu = 0;
    if (n_cu == 0) {
        int dev = 0; hipDeviceProp_t prop;
        n_cu = hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess ? prop.multiProcessorCount : 256;
    }
    const int blocks = std::min(n_cu, a.n_tiles);
    hipLaunchKernelGGL((k_mmvq_cols_mfma<TYPE>), dim3((unsigned) blocks), dim3(512), lds, stream, a);
    return true;
}

// false = not taken (format, shape or LDS size): the caller runs its own kernel
bool mul_mat_vec_q_cols_mfma(int type_a, const void * W, size_t w_row_stride, int64_t m, int64_t k,
                             const act_q8 & act, int64_t n, float * dst, size_t dst_col_stride_bytes, hipStream_t stream) {
    static int on = -1;
    if (on < 0) { const char * e = getenv("GGML_MI355X_MMVQ_COLS_MFMA"); on = e ? atoi(e) : 1; }
    if (!on || n < 2 || n > 8 || act.kind != T_Q8_K || k % 256 != 0 || m < 16 || m >= (1ll << 27) || k >= (1ll << 24)) return false;
    if (type_a != T_Q4_K && type_a != T_Q5_K && type_a != T_Q6_K) return false;
    // Where it pays (rocprofv3 kernel times, 4096 x 14336, profiles/r02_*cols*): this kernel costs ~14 us + 0.4 us per column (its 16-row
    // tiles stream the weights at ~3.5 TB/s: 64 contiguous bytes per row and load instruction), the one-row-pair-per-wave kernel of
    // mmvq.hip ~6 us + 2.5 us per column (VALU). From 5 columns on this one is ahead for Q4_K / Q5_K. Q6_K (16-element scale groups:
    // 8-byte loads, 32 quarter-filled MFMAs per block) is no faster than the old kernel at any n: only with GGML_MI355X_MMVQ_COLS_MFMA=2,
    // which also takes every n >= 2 (tests/ run both settings)
    if (on < 2 && (n < 5 || type_a == T_Q6_K)) return false;
    colmf_args a = {};
    const size_t qs_b = ((size_t) k + 15) & ~(size_t) 15, d_b = (((size_t) k/256)*4 + 15) & ~(size_t) 15, bs_b = (((size_t) k/16)*2 + 15) & ~(size_t) 15;
    const size_t img = qs_b + d_b + bs_b;
    a.col_stride = (int)(((img + 255) & ~(size_t) 255) + 32);
    a.off_d = (int) qs_b; a.off_bs = (int)(qs_b + d_b);
    const size_t lds = (size_t) n*a.col_stride + 2*8*64*4*sizeof(float);
    if (lds > 156*1024) return false;
    a.W = (const char *) W; a.w_row_stride = w_row_stride; a.m = (int) m; a.k = (int) k; a.n = (int) n;
    a.a_qs = act.qs; a.a_d = act.d; a.a_bs = act.bsums;
    a.dst = dst; a.dst_col_stride = dst_col_stride_bytes;
    a.n_tiles = (int)((m + 15)/16);
    static int dbg = -1;
    if (dbg < 0) { const char * e = getenv("GGML_MI355X_COLS_DBG"); dbg = e ? atoi(e) : 0; }
    a.dbg = dbg;
    switch (type_a) {
        case T_Q4_K: return launch_cols_mfma<T_Q4_K>(a, lds, stream);
        case T_Q5_K: return launch_cols_mfma<T_Q5_K>(a, lds, stream);
        default:     return launch_cols_mfma<T_Q6_K>(a, lds, stream);
    }
}

} // namespace mi355x
