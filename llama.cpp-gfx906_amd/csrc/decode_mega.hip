// decode_mega.hip — the persistent single-token decode kernel (program format and hand-off protocol: decode_mega.h).
//
// Why (round 2 measurements, tools/stamp_timeline.py): as five launches per layer a decoded token spent 58 us per layer against
// 21 us of weight streaming. The rest was per-launch: ~2.5 us of boundary, a head whose scalar-load chains and ~600 redundant
// prologue instructions per wave ran 2.5-5.5 us before the first dot product, and a tail of 2-4 us while the last workgroups
// finished. Inside ONE launch a phase's weights are requested BEFORE its input is waited for (the hand-off runs while they arrive),
// and workgroups that finish a phase early move on to the next phase's weights instead of idling until the launch ends.
// One workgroup per CU and the whole grid resident, or a wait could never be satisfied (the host sizes the grid by the CU count);
// every wait is bounded and reports through `err`.
#include "decode_mega.h"
#include "mmvq_core.h"
#include "quant_core.h"

#include <limits.h>
#include <math.h>

namespace mi355x {

// ---- granules: {32-bit payload, tag} as ONE aligned 8-byte write-through store; 16-byte `sc1` loads fetch two of them ----
static __device__ __forceinline__ void mg_st_gran(unsigned long long * p, uint32_t bits, uint32_t tag) {
    __hip_atomic_store(p, ((unsigned long long) tag << 32) | (unsigned long long) bits, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
static __device__ __forceinline__ uint32_t mg_ld_gran_val(const unsigned long long * p) {      // a granule that is known to be complete
    return (uint32_t) __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// 16 bytes per lane with `sc1` as ONE instruction the compiler counts: a raw buffer load (aux 16 = sc1; the MI355X guide's R1 form) over
// a descriptor of the address space above `base` (wave-uniform), lane offset in bytes
typedef __amdgpu_buffer_rsrc_t mg_rsrc;
static __device__ __forceinline__ mg_rsrc mg_make_rsrc(const void * base) {
    return __builtin_amdgcn_make_buffer_rsrc((void *) base, (short) 0, (int) 0x7FFFFFFF, (int) 0x00020000);
}
static __device__ __forceinline__ int4v mg_ld_b128(const mg_rsrc r, unsigned off) { return __builtin_bit_cast(int4v, __builtin_amdgcn_raw_buffer_load_b128(r, (int) off, 0, 16)); }

// (clang / ROCm 7.2: __builtin_bit_cast(float, vec.z) on an ext-vector ELEMENT reads element 0 — DESIGN.md section 3; the element is passed
// through a scalar parameter first)
static __device__ __forceinline__ float mg_f(int v) { return __builtin_bit_cast(float, v); }

constexpr int MEGA_SPIN_LIMIT = 1 << 20;
constexpr int MEGA_SWEEP_LIMIT = 1 << 16;

#ifdef MI_STAMPS
// debug build: wall-clock stamps per (phase, workgroup): 0 phase entry, 1 ring issued, 2 input hinted, 3 image in LDS, 4 rows done, 5 phase end
#define MG_STAMP(i_) do { if (stamps && threadIdx.x == 0) stamps[((size_t) stamp_phase*gridDim.x + blockIdx.x)*8 + (i_)] = wall_clock64(); } while (0)
#define MG_STAMP_ARGS , unsigned long long * stamps, int stamp_phase
#define MG_STAMP_PASS , stamps, i
#define MG_STAMP_PASS_INNER , stamps, stamp_phase
#else
#define MG_STAMP(i_) do { } while (0)
#define MG_STAMP_ARGS
#define MG_STAMP_PASS
#define MG_STAMP_PASS_INNER
#endif

static __device__ __forceinline__ bool mega_gave_up(unsigned * err) { return __hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) != 0; }
static __device__ __forceinline__ void mega_give_up(unsigned * err, unsigned code) { __hip_atomic_store(err, code, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); }

// the HINT: a counter in 8 shards (64 bytes apart). ONE wave waits until their sum reaches `target`; what it then reads is still checked by tag.
static __device__ __forceinline__ void mega_hint_add(unsigned * hint) {
    if (hint) __hip_atomic_fetch_add(hint + ((unsigned) blockIdx.x & 7u)*16u, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// Four polls are kept in flight, issued ~0.15 us apart: a poll's round trip is ~1 us beside streaming CUs, and one poll at a time sees a
// change 1.5 round trips late on average (2-3 us per hop measured, tools/mega_stamps.py).
static __device__ __forceinline__ unsigned mega_hint_ld(const unsigned * ptr, int lane) {
    return lane < 8 ? __hip_atomic_load(ptr + lane*16, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0u;
}
static __device__ __forceinline__ bool mega_hint_done(unsigned v, unsigned target) {
    return (int)((unsigned) __builtin_amdgcn_readfirstlane(group8_sum_i((int) v)) - target) >= 0;
}
static __device__ __forceinline__ void mega_hint_wait(const unsigned * ptr, unsigned target, unsigned * err, int lane) {
    if (!ptr) return;
    unsigned v0 = mega_hint_ld(ptr, lane); __builtin_amdgcn_s_sleep(6);
    unsigned v1 = mega_hint_ld(ptr, lane); __builtin_amdgcn_s_sleep(6);
    unsigned v2 = mega_hint_ld(ptr, lane); __builtin_amdgcn_s_sleep(6);
    unsigned v3 = mega_hint_ld(ptr, lane);
    for (int spins = 0;; spins++) {
        if (mega_hint_done(v0, target)) return;
        v0 = mega_hint_ld(ptr, lane); __builtin_amdgcn_s_sleep(6);
        if (mega_hint_done(v1, target)) return;
        v1 = mega_hint_ld(ptr, lane); __builtin_amdgcn_s_sleep(6);
        if (mega_hint_done(v2, target)) return;
        v2 = mega_hint_ld(ptr, lane); __builtin_amdgcn_s_sleep(6);
        if (mega_hint_done(v3, target)) return;
        v3 = mega_hint_ld(ptr, lane); __builtin_amdgcn_s_sleep(6);
        if ((spins & 255) == 255) {
            if (mega_gave_up(err)) return;      // somebody gave up: do not add a second timeout to it
            if (spins >= (MEGA_SPIN_LIMIT >> 2)) { mega_give_up(err, 1u); return; }
        }
    }
}

// the piece of one 256-element chunk held as 4 floats per lane by ONE wave: quantize (Q8_K) and publish as MEGA_PIECE_WORDS granules
static __device__ __forceinline__ void mega_publish_piece(float4v v, int c, int lane, unsigned long long * pieces, uint32_t tag) {
    float dd; int bsum;
    const uint32_t p = quant_chunk256<T_Q8_K>(v, dd, bsum);
    unsigned long long * pc = pieces + (size_t) c*MEGA_PIECE_WORDS;
    mg_st_gran(pc + lane, p, tag);
    // bsums: lanes 0, 4, .., 60 hold the 16 sums; word j = sum[2j] | sum[2j+1] << 16 is stored by lane 8j
    const int other = __shfl(bsum, lane + 4);
    if ((lane & 7) == 0) mg_st_gran(pc + 65 + (lane >> 3), ((uint32_t) bsum & 0xFFFFu) | ((uint32_t) other << 16), tag);
    if (lane == 0) mg_st_gran(pc + 64, __builtin_bit_cast(uint32_t, dd), tag);
}

// NORM rope on the pair (2i, 2i+1) — same formulas as rope_pair / elem.hip k_rope<false>, frequency factor passed in
static __device__ __forceinline__ void mega_rope_pair(const fused_rope & r, int pos, int row_in_head, float ff, float & x0, float & x1) {
    if (row_in_head >= r.n_dims) return;
    const int ip = row_in_head >> 1;
    const float theta_base = (float) pos*powf(r.theta_scale, (float) ip);
    const float theta_extrap = theta_base/(r.ff ? ff : 1.0f);
    float theta_interp = r.freq_scale*theta_extrap, theta = theta_interp, mscale = r.attn_factor;
    if (r.ext_factor != 0.0f) {
        const float y = ((float) ip - r.corr_lo)/fmaxf(0.001f, r.corr_hi - r.corr_lo);
        const float ramp_mix = (1.0f - fminf(1.0f, fmaxf(0.0f, y)))*r.ext_factor;
        theta = theta_interp*(1.0f - ramp_mix) + theta_extrap*ramp_mix;
        mscale *= 1.0f + 0.1f*logf(1.0f/r.freq_scale);
    }
    const float c = cosf(theta)*mscale, s = sinf(theta)*mscale;
    const float a = x0, b = x1;
    x0 = a*c - b*s;
    x1 = a*s + b*c;
}

struct mega_pre { float r0, r1; long long i0, i1; float ff; };

// ---- chunk owner (wave 0 of workgroup j < n_own, before the phase's own wait): chunk j of the previous phase's output -> piece ----
static __device__ __forceinline__ void mega_own_chunk(const mega_phase & ph, uint32_t tag_base, int phase_index, int lane, unsigned * err MG_STAMP_ARGS) {
    const int c = (int) blockIdx.x;
    MG_STAMP(6);        // (no hint hop: an owner's sweep is 2 KB; it polls by the data itself)
    const uint32_t src_tag = tag_base + (uint32_t) ph.own_src_tag_phase + 1u;
    const mg_rsrc rs = mg_make_rsrc(ph.own_src);
    const unsigned off = (unsigned)(c*256 + lane*4)*8u;
    float4v v;
    for (int sweeps = 0;; sweeps++) {
        const int4v a = mg_ld_b128(rs, off), b = mg_ld_b128(rs, off + 16u);
        const bool ok = (uint32_t) a.y == src_tag && (uint32_t) a.w == src_tag && (uint32_t) b.y == src_tag && (uint32_t) b.w == src_tag;
        v = float4v{ mg_f(a.x), mg_f(a.z), mg_f(b.x), mg_f(b.z) };
        if (__all(ok)) break;
        if ((sweeps & 63) == 63 && (mega_gave_up(err) || sweeps >= MEGA_SWEEP_LIMIT)) { if (sweeps >= MEGA_SWEEP_LIMIT) mega_give_up(err, 2u); break; }
        __builtin_amdgcn_s_sleep(1);
    }
    mega_publish_piece(v, c, lane, ph.own_pieces, tag_base + (uint32_t) phase_index + 1u);
    if (lane == 0) mega_hint_add(ph.hint2);
    MG_STAMP(7);
}

// ---- a mat-vec phase for one workgroup: the group's row pairs (rows of the dual GLU stream), grid-strided over the group's workgroups ----
template <int TYPE, bool GLU, int D>
static __device__ __forceinline__ void mega_mm(const mega_phase & ph, const mega_group & g, int wg_in_group, int nwg_group, char * smem,
                                               int lane, int wave, uint32_t tag_base, int phase_index, unsigned * err MG_STAMP_ARGS) {
    typedef mmvq_t<TYPE> T;
    constexpr int R = GLU ? 1 : 2, LPB = T::LPB, BPW = 64/LPB, FWT = 8, ACT = T::ACT;
    static_assert(ACT == T_Q8_K, "the persistent kernel serves the K-quants");
    MG_STAMP(0);
    const int k = ph.k, nb = k / T::QK, iters = (nb + BPW - 1)/BPW;
    const int slot = lane % LPB, ibl = lane / LPB;
    const int g_m = g.m;
    const size_t g_row_stride = g.row_stride;
    const char * gW = g.W; const char * gW2 = GLU ? g.W2 : nullptr;
    const int P = (g_m + R - 1)/R;
    const int u_step = nwg_group*FWT, u_base = wg_in_group*FWT + wave;
    const int n_mine = u_base < P ? (P - 1 - u_base)/u_step + 1 : 0;
    const uint32_t my_tag = tag_base + (uint32_t) phase_index + 1u;
    int p_cur = u_base;

    // ---- (1) the weight ring: the first D steps of this wave's stream, requested BEFORE the input is waited for ----
    int j_pf = 0, it_pf = 0;
    typename T::wfrag w[D][R], u[GLU ? D : 1][R];
#define MG_FETCH(d_) { \
        const bool live = j_pf < n_mine; \
        const int pp = live ? u_base + j_pf*u_step : min(u_base, P - 1); \
        const int ibf = live ? min(it_pf*BPW + ibl, nb - 1) : 0; \
        _Pragma("unroll") for (int r = 0; r < R; r++) { \
            const size_t off = (size_t) min(pp*R + r, g_m - 1)*g_row_stride; \
            w[d_][r] = T::load_w(gW + off, ibf, slot); \
            if (GLU) u[GLU ? d_ : 0][r] = T::load_w(gW2 + off, ibf, slot); \
        } \
        if (++it_pf == iters) { it_pf = 0; j_pf++; } }
    // wave 0 waits for the hint FIRST and requests its own steps afterwards: a poll queued behind the wave's ring loads returns with them
    // (a wave's memory operations complete in order), i.e. 2-3 us late — measured (tools/mega_stamps.py)
    if (wave != 0) {
#pragma unroll
        for (int d = 0; d < D; d++) MG_FETCH(d)
    }
    asm volatile("" ::: "memory");
    MG_STAMP(1);

    // epilogue operands of the first pair (branch-free; absent ones read the weights and are ignored). The residual was completed two
    // hand-offs ago (the phase that produced this phase's input could only start when it was complete), so it may be read now.
    const bool has_resg = !GLU && g.epi == EPI_ADD && g.res_gran != nullptr, has_resp = !GLU && g.epi == EPI_ADD && g.res_gran == nullptr;
    const bool has_ff = !GLU && g.epi == EPI_ROPE && ph.rope.ff != nullptr, has_idx = !GLU && g.st_mode == 2;
    const int pos0 = (!GLU && g.epi == EPI_ROPE) ? ph.pos[0] : 0;
    const long long idx0 = (!GLU && g.st_mode == 1) ? (long long) g.st_idx[0] : 0;
    mega_pre e = { 0.0f, 0.0f, 0, 0, 1.0f };
#define MG_PRE(row0_) { \
        const int ra = min((row0_), g_m - 1), rb = min((row0_) + 1, g_m - 1); \
        const unsigned long long * rg = has_resg ? g.res_gran : (const unsigned long long *) gW; \
        const float * rp = has_resp ? g.res : (const float *) gW; \
        const uint32_t ga = mg_ld_gran_val(rg + (has_resg ? ra : 0)), gb = mg_ld_gran_val(rg + (has_resg ? rb : 0)); \
        const float pa = rp[has_resp ? ra : 0], pb = rp[has_resp ? rb : 0]; \
        e.r0 = has_resg ? __builtin_bit_cast(float, ga) : pa; e.r1 = has_resg ? __builtin_bit_cast(float, gb) : pb; \
        const float * fp = has_ff ? ph.rope.ff : (const float *) gW; \
        e.ff = fp[has_ff ? (min(ra % ph.rope.head_dim, ph.rope.n_dims - 1) >> 1) : 0]; \
        const int64_t * ip = has_idx ? g.st_idx : (const int64_t *) gW; \
        e.i0 = ip[has_idx ? ra : 0]; e.i1 = ip[has_idx ? rb : 0]; }
    if (!GLU) MG_PRE(p_cur*R)

    // ---- (2) the input: wait for the hint, then build the activation image in LDS ----
    if (wave == 0) {
        if (ph.n_own > 0 && (int) blockIdx.x < ph.n_own) mega_own_chunk(ph, tag_base, phase_index, lane, err MG_STAMP_PASS_INNER);
        mega_hint_wait(ph.wait, ph.wait_target, err, lane);
#pragma unroll
        for (int d = 0; d < D; d++) MG_FETCH(d)
    }
    asm volatile("" ::: "memory");
    __syncthreads();
    MG_STAMP(2);
    int8_t * l_qs = (int8_t *) smem; float * l_d = (float *) (smem + ph.off_d); int16_t * l_bs = (int16_t *) (smem + ph.off_bs);
    const int in_mode = ph.in_mode;
    if (in_mode == MIN_IMAGE) {
        const int nch = ph.act_chunks;
        for (int i = threadIdx.x; i < nch; i += MEGA_WG_THREADS) *(int4v *) (smem + (size_t) i*16) = *(const int4v *) (ph.act + (size_t) i*16);
    } else if (in_mode == MIN_PIECES) {
        // every thread fetches pairs of piece granules (16 bytes) and scatters the two words to their places in the image; a wave
        // repeats its share until all the tags it saw were this launch's
        const uint32_t tag = tag_base + (uint32_t) ph.pieces_tag_phase + 1u;
        const mg_rsrc rp = mg_make_rsrc(ph.pieces);
        const int npairs = ph.n_pieces*(MEGA_PIECE_WORDS/2);
        const int off_d = ph.off_d, off_bs = ph.off_bs;
        for (int sweeps = 0;; sweeps++) {
            bool ok = true;
            for (int i0 = 0; i0 < npairs; i0 += 4*MEGA_WG_THREADS) {
                int4v t[4];
#pragma unroll
                for (int q = 0; q < 4; q++) t[q] = mg_ld_b128(rp, (unsigned) min(i0 + q*MEGA_WG_THREADS + (int) threadIdx.x, npairs - 1)*16u);
#pragma unroll
                for (int q = 0; q < 4; q++) {
                    const int i = i0 + q*MEGA_WG_THREADS + (int) threadIdx.x;
                    if (i < npairs) {
                        const int c = i/(MEGA_PIECE_WORDS/2), wd = (i - c*(MEGA_PIECE_WORDS/2))*2;      // words wd, wd + 1 of chunk c
                        if (wd < 73) {
                            ok = ok && (uint32_t) t[q].y == tag && (wd + 1 >= 73 || (uint32_t) t[q].w == tag);
                            if (wd < 64)       *(int2v *) (smem + c*256 + wd*4) = int2v{ t[q].x, t[q].z };      // (wd even: both are quant words)
                            else if (wd == 64) { *(int *) (smem + off_d + c*4) = t[q].x; *(int *) (smem + off_bs + c*32) = t[q].z; }
                            else { *(int *) (smem + off_bs + c*32 + (wd - 65)*4) = t[q].x; if (wd + 1 < 73) *(int *) (smem + off_bs + c*32 + (wd - 64)*4) = t[q].z; }
                        }
                    }
                }
            }
            if (__all(ok)) break;
            if ((sweeps & 63) == 63 && (mega_gave_up(err) || sweeps >= MEGA_SWEEP_LIMIT)) { if (sweeps >= MEGA_SWEEP_LIMIT) mega_give_up(err, 3u); break; }
            __builtin_amdgcn_s_sleep(1);
        }
    } else {
        // RMS_NORM(x) * w, quantized (build_norm, src/llama-graph.cpp:597-630, + the mat-vecs' activation quantizer) by every consumer workgroup,
        // with the launch path's arithmetic and summation order (mmvq_fused.h, PRO_NORM at 8 waves)
        constexpr int NAF = 4;                      // k <= 8192: at most 4 chunks of 256 per wave
        const int nchunk = k >> 8;
        float4v xv[NAF], wv[NAF];
#pragma unroll
        for (int i = 0; i < NAF; i++) wv[i] = *(const float4v *) ((in_mode == MIN_QUANT_GRAN ? (const float *) gW : ph.norm_w) + min(wave + FWT*i, nchunk - 1)*256 + lane*4);
        const bool do_norm = in_mode != MIN_QUANT_GRAN;
        if (in_mode == MIN_NORM_PLAIN) {
#pragma unroll
            for (int i = 0; i < NAF; i++) xv[i] = *(const float4v *) (ph.x + min(wave + FWT*i, nchunk - 1)*256 + lane*4);
        } else {
            const uint32_t tag = tag_base + (uint32_t) ph.in_tag_phase + 1u;
            const mg_rsrc rx = mg_make_rsrc(ph.x_gran);
            for (int sweeps = 0;; sweeps++) {
                bool ok = true;
                int4v a[NAF], b[NAF];
#pragma unroll
                for (int i = 0; i < NAF; i++) {
                    const unsigned off = (unsigned)(min(wave + FWT*i, nchunk - 1)*256 + lane*4)*8u;
                    a[i] = mg_ld_b128(rx, off); b[i] = mg_ld_b128(rx, off + 16u);
                }
#pragma unroll
                for (int i = 0; i < NAF; i++) {
                    ok = ok && (uint32_t) a[i].y == tag && (uint32_t) a[i].w == tag && (uint32_t) b[i].y == tag && (uint32_t) b[i].w == tag;
                    xv[i] = float4v{ mg_f(a[i].x), mg_f(a[i].z), mg_f(b[i].x), mg_f(b[i].z) };
                }
#ifdef MI_MEGA_TRACE
                if (__all(ok) && blockIdx.x == 0 && threadIdx.x < 2) printf("[mega] phase %d lane %d gran: %g/%08x %g/%08x %g/%08x %g/%08x tag %08x sweeps %d\n", phase_index, (int) threadIdx.x,
                    xv[0].x, a[0].y, xv[0].y, a[0].w, xv[0].z, b[0].y, xv[0].w, b[0].w, tag, sweeps);
#endif
                if (__all(ok)) break;
                if ((sweeps & 63) == 63 && (mega_gave_up(err) || sweeps >= MEGA_SWEEP_LIMIT)) { if (sweeps >= MEGA_SWEEP_LIMIT) mega_give_up(err, 4u); break; }
                __builtin_amdgcn_s_sleep(1);
            }
        }
        float * red = (float *) (smem + ph.off_bs + (((k >> 4)*2 + 15) & ~15));   // 8 floats after the image
        float ss = 0.0f;
#pragma unroll
        for (int i = 0; i < NAF; i++) if (wave + FWT*i < nchunk) ss += xv[i].x*xv[i].x + xv[i].y*xv[i].y + xv[i].z*xv[i].z + xv[i].w*xv[i].w;
        ss = wave_sum(ss);
        if (lane == 0) red[wave] = ss;
        __syncthreads();
        ss = ((red[0] + red[1]) + (red[2] + red[3])) + ((red[4] + red[5]) + (red[6] + red[7]));
        const float scale = 1.0f/sqrtf(ss/(float) k + ph.eps);
#pragma unroll
        for (int i = 0; i < NAF; i++) {
            const int c = wave + FWT*i;
            if (c < nchunk) {       // wave-uniform
                float4v v = xv[i];
                if (do_norm) v.x = (v.x*scale)*wv[i].x; if (do_norm) { v.y = (v.y*scale)*wv[i].y; v.z = (v.z*scale)*wv[i].z; v.w = (v.w*scale)*wv[i].w; }   // RMS_NORM then MUL: two roundings, as unfused
                if (ph.norm_out && blockIdx.x == 0) *(float4v *) (ph.norm_out + c*256 + lane*4) = v;      // the tensor itself, once
                quant_store_chunk256<ACT>(v, c, lane, l_qs, l_d, l_bs);
            }
        }
    }
    __syncthreads();
#ifdef MI_MEGA_TRACE
    if (blockIdx.x == 0 && threadIdx.x == 0) printf("[mega] phase %d in_mode %d k %d m %d glu %d: d[0]=%g qs[0]=%08x bs[0]=%d n_own %d wait_target %u\n", phase_index, in_mode, k, g_m, (int) GLU,
                                                    l_d[0], *(const unsigned *) l_qs, (int) l_bs[0], ph.n_own, ph.wait_target);
#endif
    MG_STAMP(3);
    act_view av;
    av.qs = l_qs; av.d = l_d; av.bs = l_bs;

    // ---- (3) stream ----
    const int total = n_mine*iters;
    int it = 0;
    float acc[2] = { 0.0f, 0.0f }, acu[2] = { 0.0f, 0.0f };
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
                MG_FETCH(d)
                if (++it == iters) {
                    float s0 = wave_sum(acc[0]), s1 = R > 1 ? wave_sum(acc[1]) : 0.0f;
                    const int row0 = p_cur*R;
#ifdef MI_MEGA_TRACE
                    if (row0 == 0 && lane == 0) printf("[mega] phase %d row0: raw s0 %g s1 %g\n", phase_index, s0, s1);
#endif
                    if (GLU) {
                        const float up_s = wave_sum(acu[0]);
                        s0 = (s0/(1.0f + expf(-s0)))*up_s;      // silu(gate)*up, as elem.hip k_glu
                        if (lane == 0) { g.dst[row0] = s0; if (g.gran) mg_st_gran(g.gran + row0, __builtin_bit_cast(uint32_t, s0), my_tag); }
#ifdef MI_MEGA_TRACE
                        if (lane == 0 && row0 < 8) printf("[mega] phase %d GLU row %d = %g gran %p\n", phase_index, row0, s0, (void *) (g.gran + row0));
#endif
                    } else if (lane == 0) {
                        const int m = g_m;
                        if (g.epi == EPI_ADD) { s0 += e.r0; if (row0 + 1 < m) s1 += e.r1; }
                        else if (g.epi == EPI_ROPE) mega_rope_pair(ph.rope, pos0, row0 % ph.rope.head_dim, e.ff, s0, s1);
                        g.dst[row0] = s0;
                        if (row0 + 1 < m) g.dst[row0 + 1] = s1;
                        if (g.gran) {
                            mg_st_gran(g.gran + row0, __builtin_bit_cast(uint32_t, s0), my_tag);
                            if (row0 + 1 < m) mg_st_gran(g.gran + row0 + 1, __builtin_bit_cast(uint32_t, s1), my_tag);
                        }
                        if (g.st_mode == 1) {
                            uint16_t * q = g.st16 + idx0*g.st_row_elems + row0;
                            q[0] = f32_to_f16_bits(s0);
                            if (row0 + 1 < m) q[1] = f32_to_f16_bits(s1);
                        } else if (g.st_mode == 2) {
                            g.st16[e.i0] = f32_to_f16_bits(s0);
                            if (row0 + 1 < m) g.st16[e.i1] = f32_to_f16_bits(s1);
                        }
                    }
                    it = 0; p_cur += u_step;
                    acc[0] = acc[1] = 0.0f; acu[0] = acu[1] = 0.0f;
                    if (!GLU && s + d + 1 < total) MG_PRE(p_cur*R)
                }
            }
        }
    }
#undef MG_FETCH
#undef MG_PRE
    MG_STAMP(4);
    __syncthreads();                    // (also: the LDS image is dead)
#ifdef MI_MEGA_TRACE
    if (blockIdx.x == 0 && threadIdx.x == 0 && g.gran) {
        __builtin_amdgcn_s_sleep(127); __builtin_amdgcn_s_sleep(127); __builtin_amdgcn_s_sleep(127); __builtin_amdgcn_s_sleep(127);
        for (int q = 0; q < 8; q++) { const unsigned long long v = __hip_atomic_load(g.gran + q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            printf("[mega] phase %d readback gran[%d] = %g / %08x\n", phase_index, q, __builtin_bit_cast(float, (unsigned) v), (unsigned)(v >> 32)); }
    }
#endif
    if (threadIdx.x == 0) mega_hint_add(ph.hint);
    MG_STAMP(5);
}

// ---- attention for ONE token, two heads per workgroup (4 waves each): kq = K.q ; p = softmax(kq*scale + mask) ; out = V^T.p
// (build_attn_mha without flash attention, src/llama-graph.cpp:1283-1330; same arithmetic as decode_fused.hip k_attn_decode<HD, true>).
// q and the NEW cell's k / v come as granules of the QKV phase (the cell's cache entry, written by the same phase for later tokens, is
// not read here); older cells come from the f16 cache, which no earlier phase of this launch has written. The two heads' 2*HD outputs are
// one 256-element chunk: written as f32 and as the piece wo reads. ----
static __device__ __forceinline__ float mg_dot8(const int4v kv, const float4v a, const float4v b) {
    const uint32_t k0 = (uint32_t) kv.x, k1 = (uint32_t) kv.y, k2 = (uint32_t) kv.z, k3 = (uint32_t) kv.w;
    float acc;
    acc  = f16_bits_to_f32((uint16_t) k0)*a.x + f16_bits_to_f32((uint16_t)(k0 >> 16))*a.y;
    acc += f16_bits_to_f32((uint16_t) k1)*a.z + f16_bits_to_f32((uint16_t)(k1 >> 16))*a.w;
    acc += f16_bits_to_f32((uint16_t) k2)*b.x + f16_bits_to_f32((uint16_t)(k2 >> 16))*b.y;
    acc += f16_bits_to_f32((uint16_t) k3)*b.z + f16_bits_to_f32((uint16_t)(k3 >> 16))*b.w;
    return acc;
}

template <int HD>
static __device__ __forceinline__ void mega_attn(const mega_phase & ph, char * smem, uint32_t tag_base, int phase_index, unsigned * err MG_STAMP_ARGS) {
    static_assert(HD == 128, "two heads of 128 make one 256-element chunk");
    MG_STAMP(0);
    const int tid = threadIdx.x, hl = tid >> 8, t = tid & 255, lane = t & 63, wave = t >> 6;
    const int n_kv = ph.n_kv;
    const int gqa = ph.n_head/ph.n_head_kv;
    const int h = (int) blockIdx.x*2 + hl, hk = h/gqa;
    const int s_words = (n_kv + 3) & ~3;
    float * s = (float *) smem + hl*s_words;                       // [n_kv] scores -> probabilities, per head
    float * sh = (float *) smem + 2*s_words + hl*8;                // reduction scratch, per head
    float * obuf = (float *) smem + 2*s_words + 16;                // [256] the chunk
    float * qn = obuf + 256;                                       // [2][128] q of the two heads
    float * kn = qn + 256;                                         // [2][128] k of the new cell (f16-rounded) for each local head's kv head
    float * vn = kn + 256;                                         // [2][128] v of the new cell
    constexpr int LPC = HD/8, CPW = 64/LPC, U = 4;
    const int sub = lane % LPC, cw = lane / LPC;

    MG_STAMP(2);        // (no hint hop: 16 workgroups sweep 4 KB each; they poll by the data itself)

    // ---- q (2 x 128) and the new cell's k, v (128 each per local head) ----
    {
        const uint32_t tag = tag_base + (uint32_t) ph.in_tag_phase + 1u;
        const mg_rsrc rq = mg_make_rsrc(ph.q_gran), rk = mg_make_rsrc(ph.k_gran), rv = mg_make_rsrc(ph.v_gran);
        const int hk0 = ((int) blockIdx.x*2)/gqa, hk1 = ((int) blockIdx.x*2 + 1)/gqa;
        // 128 pairs of q granules, 2 x 64 pairs of k, 2 x 64 pairs of v: threads 0..127 q, 128..255 k, 256..383 v
        for (int sweeps = 0;; sweeps++) {
            bool ok = true;
            if (tid < 128) {
                const int4v a = mg_ld_b128(rq, (unsigned)((int) blockIdx.x*256 + tid*2)*8u);
                ok = (uint32_t) a.y == tag && (uint32_t) a.w == tag;
                qn[tid*2] = mg_f(a.x); qn[tid*2 + 1] = mg_f(a.z);
            } else if (tid < 384) {
                const bool isv = tid >= 256; const int tt = (tid - 128) & 127, lh = tt >> 6, pi = tt & 63;
                const int4v a = mg_ld_b128(isv ? rv : rk, (unsigned)((lh ? hk1 : hk0)*HD + pi*2)*8u);
                ok = (uint32_t) a.y == tag && (uint32_t) a.w == tag;
                float * dstp = (isv ? vn : kn) + lh*HD + pi*2;
                dstp[0] = f16_bits_to_f32(f32_to_f16_bits(mg_f(a.x)));      // what the cache holds for this cell
                dstp[1] = f16_bits_to_f32(f32_to_f16_bits(mg_f(a.z)));
            }
            if (__all(ok)) break;
            if ((sweeps & 63) == 63 && (mega_gave_up(err) || sweeps >= MEGA_SWEEP_LIMIT)) { if (sweeps >= MEGA_SWEEP_LIMIT) mega_give_up(err, 5u); break; }
            __builtin_amdgcn_s_sleep(1);
        }
    }
    __syncthreads();
    const int cur = (int) ph.cell_idx[0];

    // ---- scores of the cached cells ----
    const float * qp = qn + hl*HD + sub*8;
    const float4v q0 = *(const float4v *) qp, q1 = *(const float4v *) (qp + 4);
    const char * kbase = ph.kc + (size_t) hk*ph.k_nb2 + sub*16;
    const char * mrow = ph.mask;
    float mx = -INFINITY;
    for (int j0 = wave*CPW + cw; j0 < n_kv; j0 += 4*CPW*U) {
        int4v kreg[U]; float mreg[U];
#pragma unroll
        for (int uu = 0; uu < U; uu++) {
            const int j = min(j0 + uu*4*CPW, n_kv - 1);
            kreg[uu] = *(const int4v *) (kbase + (size_t) j*ph.k_nb1);
            mreg[uu] = 0.0f;
            if (mrow) mreg[uu] = ph.mask_f16 ? f16_bits_to_f32(*(const uint16_t *) (mrow + (size_t) j*2)) : *(const float *) (mrow + (size_t) j*4);
        }
#pragma unroll
        for (int uu = 0; uu < U; uu++) {
            const int j = j0 + uu*4*CPW;
            float acc = mg_dot8(kreg[uu], q0, q1);
            acc += dpp_f<0xB1>(acc); acc += dpp_f<0x4E>(acc); acc += dpp_f<0x141>(acc); acc += dpp_f<0x140>(acc);
            if (j < n_kv && j != cur) {
                const float v = acc*ph.scale + mreg[uu];
                if (sub == 0) s[j] = v;
                mx = fmaxf(mx, v);
            }
        }
    }
    // ---- the new cell: k from the granules, summed exactly as a cached row is (8 elements per lane, then the 16 lanes of its group) ----
    if (wave == 0 && cur < n_kv) {
        const float * kp = kn + hl*HD + sub*8;
        float acc;
        acc  = kp[0]*q0.x + kp[1]*q0.y;
        acc += kp[2]*q0.z + kp[3]*q0.w;
        acc += kp[4]*q1.x + kp[5]*q1.y;
        acc += kp[6]*q1.z + kp[7]*q1.w;
        acc += dpp_f<0xB1>(acc); acc += dpp_f<0x4E>(acc); acc += dpp_f<0x141>(acc); acc += dpp_f<0x140>(acc);
        float mv = 0.0f;
        if (mrow) mv = ph.mask_f16 ? f16_bits_to_f32(*(const uint16_t *) (mrow + (size_t) cur*2)) : *(const float *) (mrow + (size_t) cur*4);
        const float v = acc*ph.scale + mv;
        if (lane == 0) s[cur] = v;
        mx = fmaxf(mx, v);
    }
    mx = wave_max(mx);
    if (lane == 0) sh[wave] = mx;
    __syncthreads();
    mx = fmaxf(fmaxf(sh[0], sh[1]), fmaxf(sh[2], sh[3]));
    __syncthreads();
    float sum = 0.0f;
    const float mxs = mx == -INFINITY ? 0.0f : mx;
    for (int j = t; j < n_kv; j += 256) { const float ev = expf(s[j] - mxs); s[j] = ev; sum += ev; }
    sum = wave_sum(sum);
    if (lane == 0) sh[wave] = sum;
    __syncthreads();
    sum = (sh[0] + sh[1]) + (sh[2] + sh[3]);
    const float inv = sum > 0.0f ? 1.0f/sum : 0.0f;
    for (int j = t; j < n_kv; j += 256) s[j] *= inv;   // the unfused SOFT_MAX normalises before V.p
    __syncthreads();

    // ---- out[d] = sum_j V[d][j]*p[j]: 16 lanes per V row, 4 rows per wave, HD/16 row groups. Column `cur` of the cache may not hold this
    // launch's value yet: its f16 is replaced by the granule's (same bits as the cache will hold) ----
    constexpr int NG = HD/16;
    const int l16 = lane & 15, rw = lane >> 4;
    const char * vbase = ph.vc + (size_t) hk*ph.v_nb2 + (size_t)(wave*4 + rw)*ph.v_nb1;
    float acc[NG];
#pragma unroll
    for (int gq = 0; gq < NG; gq++) acc[gq] = 0.0f;
    const int nchunk = n_kv >> 3;                        // n_kv % 8 == 0
    const int cchunk = cur >> 3, cword = (cur & 7) >> 1, chalf = cur & 1;
    for (int c = l16; c < nchunk; c += 16) {
        int4v vreg[NG];
#pragma unroll
        for (int gq = 0; gq < NG; gq++) vreg[gq] = *(const int4v *) (vbase + (size_t)(gq*16)*ph.v_nb1 + (size_t) c*16);
        const float4v p0 = *(const float4v *) (s + c*8), p1 = *(const float4v *) (s + c*8 + 4);
        if (c == cchunk) {
#pragma unroll
            for (int gq = 0; gq < NG; gq++) {
                const uint32_t nv = f32_to_f16_bits(vn[hl*HD + gq*16 + wave*4 + rw]);
                uint32_t wds[4] = { (uint32_t) vreg[gq].x, (uint32_t) vreg[gq].y, (uint32_t) vreg[gq].z, (uint32_t) vreg[gq].w };
#pragma unroll
                for (int q = 0; q < 4; q++) if (q == cword) wds[q] = chalf ? ((wds[q] & 0x0000FFFFu) | (nv << 16)) : ((wds[q] & 0xFFFF0000u) | nv);
                vreg[gq] = int4v{ (int) wds[0], (int) wds[1], (int) wds[2], (int) wds[3] };
            }
        }
#pragma unroll
        for (int gq = 0; gq < NG; gq++) acc[gq] += mg_dot8(vreg[gq], p0, p1);
    }
#pragma unroll
    for (int gq = 0; gq < NG; gq++) {
        const float r = row16_sum(acc[gq]);
        if (l16 == 0) obuf[hl*HD + gq*16 + wave*4 + rw] = r;
    }
    __syncthreads();
    if (tid < 64) {
        const int c = (int) blockIdx.x;
        const float4v v = *(const float4v *) (obuf + tid*4);
        *(float4v *) (ph.attn_dst + c*256 + tid*4) = v;
        mega_publish_piece(v, c, tid, ph.own_pieces, tag_base + (uint32_t) phase_index + 1u);
        if (tid == 0) mega_hint_add(ph.hint);
    }
    MG_STAMP(5);
}

// ---- the kernel: K-quant weight formats (Q4_K / Q5_K / Q6_K in any mixture: llama_tensor_get_type's Q4_K_M, Q5_K_M, Q6_K files) ----
__global__ void __launch_bounds__(MEGA_WG_THREADS, 2) k_mega_kquants(const mega_phase * __restrict__ prog, int n_phases, unsigned * epoch, unsigned * err MG_STAMP_ARGS) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int b = (int) blockIdx.x;
    // tags of this launch: (launch count + 1) << 12 | phase + 1. The word is bumped by workgroup 0 at the very end (every workgroup has
    // read it long before: they all start together, the whole grid is resident)
    const uint32_t tag_base = (__hip_atomic_load(epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1u) << 12;
    for (int i = 0; i < n_phases; i++) {
        const mega_phase & ph = prog[i];
        const int kind = ph.kind;
        if (kind == MEGA_ATTN) {
            if (b < ph.n_active) mega_attn<128>(ph, smem, tag_base, i, err MG_STAMP_PASS);
        } else if (kind == MEGA_MM) {
            if (b < ph.n_active) {
                const int be0 = ph.block_end[0], be1 = ph.block_end[1];
                const int gi = (b >= be0 ? 1 : 0) + (b >= be1 ? 1 : 0);
                const int first = gi == 0 ? 0 : (gi == 1 ? be0 : be1);
                const int last = gi == 0 ? be0 : (gi == 1 ? be1 : ph.block_end[2]);
                const mega_group & g = ph.g[gi];
                const int type = g.type;
                if (ph.glu) {
                    if (type == T_Q4_K)      mega_mm<T_Q4_K, true, 4>(ph, g, b - first, last - first, smem, lane, wave, tag_base, i, err MG_STAMP_PASS);
                    else if (type == T_Q5_K) mega_mm<T_Q5_K, true, 4>(ph, g, b - first, last - first, smem, lane, wave, tag_base, i, err MG_STAMP_PASS);
                    else                     mega_mm<T_Q6_K, true, 4>(ph, g, b - first, last - first, smem, lane, wave, tag_base, i, err MG_STAMP_PASS);
                } else {
                    if (type == T_Q4_K)      mega_mm<T_Q4_K, false, 4>(ph, g, b - first, last - first, smem, lane, wave, tag_base, i, err MG_STAMP_PASS);
                    else if (type == T_Q5_K) mega_mm<T_Q5_K, false, 4>(ph, g, b - first, last - first, smem, lane, wave, tag_base, i, err MG_STAMP_PASS);
                    else                     mega_mm<T_Q6_K, false, 4>(ph, g, b - first, last - first, smem, lane, wave, tag_base, i, err MG_STAMP_PASS);
                }
            }
        }
        __syncthreads();        // the LDS words are reused by the next phase
    }
    if (b == 0 && threadIdx.x == 0) __hip_atomic_fetch_add(epoch, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

bool mega_supported_types(const int * types, int n) {
    for (int i = 0; i < n; i++) if (types[i] != T_Q4_K && types[i] != T_Q5_K && types[i] != T_Q6_K) return false;
    return true;
}

int mega_max_workgroups(void) {
    int dev = 0; hipDeviceProp_t prop;
    if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) { (void) hipGetLastError(); return 0; }
    return prop.multiProcessorCount;
}

#ifdef MI_STAMPS
static unsigned long long * g_mega_stamp_buf = nullptr; static int g_mega_stamp_phases = 0, g_mega_stamp_last_n = 0, g_mega_stamp_last_wg = 0;
extern "C" int mi355x_mega_stamps_enable(int max_phases, int n_wg) {
    if (g_mega_stamp_buf) { (void) hipFree(g_mega_stamp_buf); g_mega_stamp_buf = nullptr; }
    g_mega_stamp_phases = max_phases;
    if (max_phases <= 0) return 0;
    if (hipMalloc(&g_mega_stamp_buf, (size_t) max_phases*n_wg*8*8) != hipSuccess) return -1;
    (void) hipMemset(g_mega_stamp_buf, 0, (size_t) max_phases*n_wg*8*8);
    return 0;
}
// stamps of the most recent launch with the largest phase count: [phase][wg][8]
extern "C" int mi355x_mega_stamps_read(unsigned long long * out, int * n_phases, int * n_wg) {
    if (!g_mega_stamp_buf) return -1;
    *n_phases = g_mega_stamp_last_n; *n_wg = g_mega_stamp_last_wg;
    (void) hipMemcpy(out, g_mega_stamp_buf, (size_t) g_mega_stamp_last_n*g_mega_stamp_last_wg*8*8, hipMemcpyDeviceToHost);
    return 0;
}
#endif

void mega_launch(const mega_phase * prog_dev, int n_phases, int n_wg, unsigned * epoch, unsigned * err, size_t lds_bytes, hipStream_t stream) {
#ifdef MI_STAMPS
    unsigned long long * st = (g_mega_stamp_buf && n_phases <= g_mega_stamp_phases && n_phases >= g_mega_stamp_last_n) ? g_mega_stamp_buf : nullptr;     // the long program of a token
    if (st) { g_mega_stamp_last_n = n_phases; g_mega_stamp_last_wg = n_wg; }
    hipLaunchKernelGGL(k_mega_kquants, dim3((unsigned) n_wg), dim3(MEGA_WG_THREADS), lds_bytes, stream, prog_dev, n_phases, epoch, err, st, 0);
#else
    hipLaunchKernelGGL(k_mega_kquants, dim3((unsigned) n_wg), dim3(MEGA_WG_THREADS), lds_bytes, stream, prog_dev, n_phases, epoch, err);
#endif
}

} // namespace mi355x
