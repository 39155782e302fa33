// decode_mega.hip — the persistent single-token decode kernel (program format: decode_mega.h).
//
// Why (round 2 measurements, tools/stamp_timeline.py): as five launches per layer a decoded token spent 58 us per layer against
// 21 us of weight streaming. The rest was per-launch: ~2.5 us of boundary, a head whose scalar-load chains and ~600 redundant
// prologue instructions per wave (RMS norm + activation quantization repeated by every one of 256 workgroups) ran 2.5-5.5 us
// before the first dot product, and a tail of 2-4 us while the last workgroups finished. Inside ONE launch
//   * a phase's weights are requested BEFORE the wait for its input (the wait, the finaliser and the image copy run while they
//     arrive: all of wo's and ffn_down's weights and more than half of gate/up's fit the register ring),
//   * the activation vector is normalised and quantized ONCE, by the workgroup that completes it (the last arriver), and every
//     consumer copies the finished int8 image (4.6 - 17 KB) into LDS,
//   * workgroups that finish a phase early move on to the next phase's weights instead of idling until the launch ends.
// Hand-offs follow the MI355X guide's measured-valid form: payload stored write-through (`sc1`), every storing wave drains
// (`s_waitcnt vmcnt(0)`), workgroup barrier, ONE lane adds to an agent-scope counter; consumers poll the counter with `sc1` loads
// from one wave, then a workgroup barrier, then `sc1` loads of the payload. Every wait is bounded and reports through `err`.
// One workgroup per CU and the whole grid resident, or a wait could never be satisfied (the host sizes the grid by the CU count).
#include "decode_mega.h"
#include "mmvq_core.h"
#include "quant_core.h"

#include <limits.h>
#include <math.h>

namespace mi355x {

// ---- write-through stores / L1-bypassing loads (agent scope, relaxed: `sc1`) ----
static __device__ __forceinline__ void mg_st_f32(float * p, float v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
static __device__ __forceinline__ void mg_st_u32(uint32_t * p, uint32_t v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
static __device__ __forceinline__ void mg_st_u16(uint16_t * p, uint16_t v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
static __device__ __forceinline__ float mg_ld_f32(const float * p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
// 16 bytes per lane with `sc1` as ONE instruction the compiler counts: a raw buffer load (aux 16 = sc1; the MI355X guide's R1 form) over
// a descriptor of the whole address space above `base` (wave-uniform), lane offset in bytes
typedef __amdgpu_buffer_rsrc_t mg_rsrc;
static __device__ __forceinline__ mg_rsrc mg_make_rsrc(const void * base) {
    return __builtin_amdgcn_make_buffer_rsrc((void *) base, (short) 0, (int) 0x7FFFFFFF, (int) 0x00020000);
}
static __device__ __forceinline__ int4v mg_ld_b128(const mg_rsrc r, unsigned off) { return __builtin_bit_cast(int4v, __builtin_amdgcn_raw_buffer_load_b128(r, (int) off, 0, 16)); }
static __device__ __forceinline__ float4v mg_ld_f4(const mg_rsrc r, unsigned off) { return __builtin_bit_cast(float4v, __builtin_amdgcn_raw_buffer_load_b128(r, (int) off, 0, 16)); }

constexpr int MEGA_SPIN_LIMIT = 1 << 21;

#ifdef MI_STAMPS
// debug build: wall-clock stamps per (phase, workgroup): 0 phase entry, 1 ring issued, 2 input signalled, 3 image in LDS, 4 rows done, 5 phase end
#define MG_STAMP(i_) do { if (stamps && threadIdx.x == 0) stamps[((size_t) stamp_phase*gridDim.x + blockIdx.x)*8 + (i_)] = wall_clock64(); } while (0)
#define MG_STAMP_ARGS , unsigned long long * stamps, int stamp_phase
#define MG_STAMP_PASS , stamps, i
#else
#define MG_STAMP(i_) do { } while (0)
#define MG_STAMP_ARGS
#define MG_STAMP_PASS
#endif

// ONE wave calls this (all its lanes poll the same word: one request); the caller puts a workgroup barrier behind it
static __device__ __forceinline__ void mega_wait(const unsigned * ptr, unsigned target, unsigned * err) {
    if (!ptr) return;
    int spins = 0;
    while ((int)(__hip_atomic_load(ptr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - target) < 0) {
        __builtin_amdgcn_s_sleep(4);
        if ((++spins & 1023) == 0) {
            if (__hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) != 0) return;      // somebody gave up: do not add a second timeout to it
            if (spins >= MEGA_SPIN_LIMIT) { __hip_atomic_store(err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); return; }
        }
    }
}

static __device__ __forceinline__ void mega_drain() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }

// ---- quantized image pieces: chunk c of a vector held as 4 floats per lane, stored write-through ----
template <int ACT>
static __device__ __forceinline__ void mega_quant_store(float4v v, int c, int lane, char * img, int off_d, int off_bs) {
    float dd; int bsum;
    const uint32_t p = quant_chunk256<ACT>(v, dd, bsum);
    mg_st_u32((uint32_t *) (img + c*256 + lane*4), p);
    float * d = (float *) (img + off_d); uint16_t * bs = (uint16_t *) (img + off_bs);
    if (ACT == T_Q8_0) {
        if ((lane & 7) == 0) { mg_st_f32(d + c*8 + (lane >> 3), dd); mg_st_u16(bs + c*8 + (lane >> 3), (uint16_t)(int16_t) bsum); }
    } else {
        if ((lane & 3) == 0) mg_st_u16(bs + c*16 + (lane >> 2), (uint16_t)(int16_t) bsum);
        if (lane == 0) mg_st_f32(d + c, dd);
    }
}

// ---- MFIN_NORM: RMS_NORM(x) * w -> f32 tensor + quantized image, by the 8 waves of ONE workgroup (build_norm, src/llama-graph.cpp:597-630,
// + the consumer mat-vecs' activation quantizer). Same arithmetic and summation order as the launch path's in-prologue norm (mmvq_fused.h). ----
template <int ACT>
static __device__ __forceinline__ void mega_fin_norm_t(const mega_phase & ph, char * smem, int lane, int wave) {
    constexpr int NAF = 4;                      // k <= 8192: at most 4 chunks of 256 per wave
    const int k = ph.fin_k, nchunk = k >> 8;
    float * red = (float *) smem;
    float4v xv[NAF], wv[NAF];
    const mg_rsrc rx = mg_make_rsrc(ph.fin_x);
#pragma unroll
    for (int i = 0; i < NAF; i++) {
        const int c = min(wave + 8*i, nchunk - 1);
        xv[i] = mg_ld_f4(rx, (unsigned)(c*256 + lane*4)*4u);
        wv[i] = *(const float4v *) (ph.fin_norm_w + c*256 + lane*4);
    }
    float ss = 0.0f;
#pragma unroll
    for (int i = 0; i < NAF; i++) if (wave + 8*i < nchunk) ss += xv[i].x*xv[i].x + xv[i].y*xv[i].y + xv[i].z*xv[i].z + xv[i].w*xv[i].w;
    ss = wave_sum(ss);
    __syncthreads();                            // the LDS words below may still be read as the previous phase's image
    if (lane == 0) red[wave] = ss;
    __syncthreads();
    ss = ((red[0] + red[1]) + (red[2] + red[3])) + ((red[4] + red[5]) + (red[6] + red[7]));
    const float scale = 1.0f/sqrtf(ss/(float) k + ph.fin_eps);
#pragma unroll
    for (int i = 0; i < NAF; i++) {
        const int c = wave + 8*i;
        if (c < nchunk) {       // wave-uniform
            float4v v = xv[i];
            v.x = (v.x*scale)*wv[i].x; v.y = (v.y*scale)*wv[i].y; v.z = (v.z*scale)*wv[i].z; v.w = (v.w*scale)*wv[i].w;   // RMS_NORM then MUL: two roundings, as unfused
            if (ph.fin_norm_out) *(float4v *) (ph.fin_norm_out + c*256 + lane*4) = v;
            mega_quant_store<ACT>(v, c, lane, ph.fin_img, ph.fin_off_d, ph.fin_off_bs);
        }
    }
    mega_drain();
    __syncthreads();
    if (threadIdx.x == 0 && ph.signal) __hip_atomic_fetch_add(ph.signal, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
static __device__ __forceinline__ void mega_fin_norm(const mega_phase & ph, char * smem, int lane, int wave) {
    if (ph.fin_kind == T_Q8_0) mega_fin_norm_t<T_Q8_0>(ph, smem, lane, wave); else mega_fin_norm_t<T_Q8_K>(ph, smem, lane, wave);
}

// ---- the end of a mat-vec phase for one workgroup: its rows are stored; signal / finalise as the phase says ----
static __device__ __forceinline__ void mega_phase_end(const mega_phase & ph, const mega_group & g, int wg_in_group, int nwg_group, char * smem, int lane, int wave) {
    mega_drain();                               // every storing wave: its write-through stores have left
    __syncthreads();
    int * list = (int *) smem;                  // the image in LDS is dead now
    if (ph.fin_mode == MFIN_NONE) {
        if (threadIdx.x == 0 && ph.signal) __hip_atomic_fetch_add(ph.signal, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    } else if (ph.fin_mode == MFIN_NORM) {
        if (threadIdx.x == 0) list[0] = __hip_atomic_fetch_add(ph.arrive, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == (unsigned)(ph.n_active - 1);
        __syncthreads();
        const bool last = list[0] != 0;
        if (last) mega_fin_norm(ph, smem, lane, wave);
    } else {    // MFIN_CHUNK (the dual GLU stream, rows grid-strided: round i of this workgroup = rows wg*8 + stride*i .. +7, all in one chunk)
        const int stride = nwg_group*8, rounds = (g.m - wg_in_group*8 + stride - 1)/stride;
        const int expect = 256/8;               // workgroups per chunk (stride % 256 == 0, m % 256 == 0: checked by the host)
        if ((int) threadIdx.x < 64) {
            bool mine = false; int chunk = 0;
            if ((int) threadIdx.x < rounds) {
                chunk = (wg_in_group*8 + stride*(int) threadIdx.x) >> 8;
                mine = __hip_atomic_fetch_add(ph.arrive + chunk, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == (unsigned)(expect - 1);
            }
            const unsigned long long bal = __ballot(mine);
            if (mine) list[1 + __popcll(bal & ((1ull << threadIdx.x) - 1))] = chunk;
            if (threadIdx.x == 0) list[0] = __popcll(bal);
        }
        __syncthreads();
        const int nfin = list[0];
        for (int j = wave; j < nfin; j += 8) {
            const int c = list[1 + j];
            const float4v v = mg_ld_f4(mg_make_rsrc(g.dst), (unsigned)(c*256 + lane*4)*4u);
            if (ph.fin_kind == T_Q8_0) mega_quant_store<T_Q8_0>(v, c, lane, ph.fin_img, ph.fin_off_d, ph.fin_off_bs);
            else                       mega_quant_store<T_Q8_K>(v, c, lane, ph.fin_img, ph.fin_off_d, ph.fin_off_bs);
            mega_drain();                       // this wave stored the piece itself and signals for itself
            if (lane == 0 && ph.signal) __hip_atomic_fetch_add(ph.signal, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
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

// ---- a mat-vec phase for one workgroup: the group's row pairs (rows of the dual GLU stream), grid-strided over the group's workgroups ----
template <int TYPE, bool GLU, int D>
static __device__ __forceinline__ void mega_mm(const mega_phase & ph, const mega_group & g, int wg_in_group, int nwg_group, char * smem,
                                               int lane, int wave, unsigned * err MG_STAMP_ARGS) {
    MG_STAMP(0);
    typedef mmvq_t<TYPE> T;
    constexpr int R = GLU ? 1 : 2, LPB = T::LPB, BPW = 64/LPB, FWT = 8;
    const int k = ph.k, nb = k / T::QK, iters = (nb + BPW - 1)/BPW;
    const int slot = lane % LPB, ibl = lane / LPB;
    const int g_m = g.m;
    const size_t g_row_stride = g.row_stride;
    const char * gW = g.W; const char * gW2 = GLU ? g.W2 : nullptr;
    const int P = (g_m + R - 1)/R;
    const int u_step = nwg_group*FWT, u_base = wg_in_group*FWT + wave;
    const int n_mine = u_base < P ? (P - 1 - u_base)/u_step + 1 : 0;
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
    // wave 0 polls for the input FIRST and requests its own steps afterwards: a poll queued behind the wave's ring loads returns with them
    // (a wave's memory operations complete in order), i.e. 2-3 us late — measured (tools/mega_stamps.py)
    if (wave != 0) {
#pragma unroll
        for (int d = 0; d < D; d++) MG_FETCH(d)
    }
    asm volatile("" ::: "memory");
    MG_STAMP(1);

    // epilogue operands of the first pair (branch-free; absent ones read the weights and are ignored). The residual was completed two
    // hand-offs ago (its own finaliser ran before the phase that produced this phase's input started), so it may be read now.
    const bool has_res = !GLU && g.epi == EPI_ADD, has_ff = !GLU && g.epi == EPI_ROPE && ph.rope.ff != nullptr, has_idx = !GLU && g.st_mode == 2;
    const int pos0 = (!GLU && g.epi == EPI_ROPE) ? ph.pos[0] : 0;
    const long long idx0 = (!GLU && g.st_mode == 1) ? (long long) g.st_idx[0] : 0;
    mega_pre e = { 0.0f, 0.0f, 0, 0, 1.0f };
#define MG_PRE(row0_) { \
        const int ra = min((row0_), g_m - 1), rb = min((row0_) + 1, g_m - 1); \
        const float * rp = has_res ? g.res : (const float *) gW; \
        e.r0 = mg_ld_f32(rp + (has_res ? ra : 0)); e.r1 = mg_ld_f32(rp + (has_res ? rb : 0)); \
        const float * fp = has_ff ? ph.rope.ff : (const float *) gW; \
        e.ff = fp[has_ff ? (min(ra % ph.rope.head_dim, ph.rope.n_dims - 1) >> 1) : 0]; \
        const int64_t * ip = has_idx ? g.st_idx : (const int64_t *) gW; \
        e.i0 = ip[has_idx ? ra : 0]; e.i1 = ip[has_idx ? rb : 0]; }
    if (!GLU) MG_PRE(p_cur*R)

    // ---- (2) wait for the input image, copy it into LDS ----
    if (wave == 0) {
        mega_wait(ph.wait, ph.wait_target, err);
#pragma unroll
        for (int d = 0; d < D; d++) MG_FETCH(d)
    }
    asm volatile("" ::: "memory");
    __syncthreads();
    MG_STAMP(2);
    {
        const int nch = ph.act_chunks;
        const mg_rsrc ra = mg_make_rsrc(ph.act);
        for (int i0 = 0; i0 < nch; i0 += 4*MEGA_WG_THREADS) {
            int4v t[4];
#pragma unroll
            for (int q = 0; q < 4; q++) t[q] = mg_ld_b128(ra, (unsigned) min(i0 + q*MEGA_WG_THREADS + (int) threadIdx.x, nch - 1)*16u);
#pragma unroll
            for (int q = 0; q < 4; q++) { const int i = i0 + q*MEGA_WG_THREADS + (int) threadIdx.x; if (i < nch) *(int4v *) (smem + (size_t) i*16) = t[q]; }
        }
    }
    __syncthreads();
    MG_STAMP(3);
    act_view av;
    av.qs = (const int8_t *) smem; av.d = (const float *) (smem + ph.off_d); av.bs = (const int16_t *) (smem + ph.off_bs);

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
                    if (GLU) {
                        const float up_s = wave_sum(acu[0]);
                        s0 = (s0/(1.0f + expf(-s0)))*up_s;      // silu(gate)*up, as elem.hip k_glu
                        if (lane == 0) mg_st_f32(g.dst + row0, s0);
                    } else if (lane == 0) {
                        const int m = g_m;
                        if (g.epi == EPI_ADD) { s0 += e.r0; if (row0 + 1 < m) s1 += e.r1; }
                        else if (g.epi == EPI_ROPE) mega_rope_pair(ph.rope, pos0, row0 % ph.rope.head_dim, e.ff, s0, s1);
                        mg_st_f32(g.dst + row0, s0);
                        if (row0 + 1 < m) mg_st_f32(g.dst + row0 + 1, s1);
                        if (g.st_mode == 1) {
                            uint16_t * q = g.st16 + idx0*g.st_row_elems + row0;
                            mg_st_u16(q, f32_to_f16_bits(s0));
                            if (row0 + 1 < m) mg_st_u16(q + 1, f32_to_f16_bits(s1));
                        } else if (g.st_mode == 2) {
                            mg_st_u16(g.st16 + e.i0, f32_to_f16_bits(s0));
                            if (row0 + 1 < m) mg_st_u16(g.st16 + e.i1, f32_to_f16_bits(s1));
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
    mega_phase_end(ph, g, wg_in_group, nwg_group, smem, lane, wave);
    MG_STAMP(5);
}

// ---- attention for ONE token over the f16 KV cache, two heads per workgroup (4 waves each): kq = K.q ; p = softmax(kq*scale + mask) ;
// out = V^T.p (build_attn_mha without flash attention, src/llama-graph.cpp:1283-1330; same arithmetic as decode_fused.hip k_attn_decode<HD, true>).
// The two heads' 2*HD outputs are one 256-element chunk: written as f32 and as a piece of the quantized image wo reads. ----
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
static __device__ __forceinline__ void mega_attn(const mega_phase & ph, char * smem, unsigned * err MG_STAMP_ARGS) {
    MG_STAMP(0);
    static_assert(HD == 128, "two heads of 128 make one 256-element chunk");
    const int tid = threadIdx.x, hl = tid >> 8, t = tid & 255, lane = t & 63, wave = t >> 6;
    const int n_kv = ph.n_kv;
    const int h = (int) blockIdx.x*2 + hl, hk = h/(ph.n_head/ph.n_head_kv);
    const int s_words = (n_kv + 3) & ~3;
    float * s = (float *) smem + hl*s_words;                       // [n_kv] scores -> probabilities, per head
    float * sh = (float *) smem + 2*s_words + hl*8;                // reduction scratch, per head
    float * obuf = (float *) smem + 2*s_words + 16;                // [256] the chunk
    constexpr int LPC = HD/8, CPW = 64/LPC, U = 4;
    const int sub = lane % LPC, cw = lane / LPC;

    if (tid < 64) mega_wait(ph.wait, ph.wait_target, err);
    __syncthreads();
    MG_STAMP(2);

    // ---- scores ----
    const mg_rsrc rq = mg_make_rsrc(ph.q), rk = mg_make_rsrc(ph.kc), rv = mg_make_rsrc(ph.vc);
    const unsigned qoff = (unsigned)((size_t) h*ph.q_nb2) + sub*32u;
    const float4v q0 = mg_ld_f4(rq, qoff), q1 = mg_ld_f4(rq, qoff + 16u);
    const unsigned kbase = (unsigned)((size_t) hk*ph.k_nb2) + sub*16u;
    const char * mrow = ph.mask;
    float mx = -INFINITY;
    for (int j0 = wave*CPW + cw; j0 < n_kv; j0 += 4*CPW*U) {
        int4v kreg[U]; float mreg[U];
#pragma unroll
        for (int uu = 0; uu < U; uu++) {
            const int j = min(j0 + uu*4*CPW, n_kv - 1);
            kreg[uu] = mg_ld_b128(rk, kbase + (unsigned) j*(unsigned) ph.k_nb1);
            mreg[uu] = 0.0f;
            if (mrow) mreg[uu] = ph.mask_f16 ? f16_bits_to_f32(*(const uint16_t *) (mrow + (size_t) j*2)) : *(const float *) (mrow + (size_t) j*4);
        }
#pragma unroll
        for (int uu = 0; uu < U; uu++) {
            const int j = j0 + uu*4*CPW;
            float acc = mg_dot8(kreg[uu], q0, q1);
            acc += dpp_f<0xB1>(acc); acc += dpp_f<0x4E>(acc); acc += dpp_f<0x141>(acc); acc += dpp_f<0x140>(acc);
            if (j < n_kv) {
                const float v = acc*ph.scale + mreg[uu];
                if (sub == 0) s[j] = v;
                mx = fmaxf(mx, v);
            }
        }
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

    // ---- out[d] = sum_j V[d][j]*p[j]: 16 lanes per V row, 4 rows per wave, HD/16 row groups ----
    constexpr int NG = HD/16;
    const int l16 = lane & 15, rw = lane >> 4;
    const unsigned vbase = (unsigned)((size_t) hk*ph.v_nb2 + (size_t)(wave*4 + rw)*ph.v_nb1);
    float acc[NG];
#pragma unroll
    for (int gq = 0; gq < NG; gq++) acc[gq] = 0.0f;
    const int nchunk = n_kv >> 3;                        // n_kv % 8 == 0
    for (int c = l16; c < nchunk; c += 16) {
        int4v vreg[NG];
#pragma unroll
        for (int gq = 0; gq < NG; gq++) vreg[gq] = mg_ld_b128(rv, vbase + (unsigned)(gq*16)*(unsigned) ph.v_nb1 + (unsigned) c*16u);
        const float4v p0 = *(const float4v *) (s + c*8), p1 = *(const float4v *) (s + c*8 + 4);
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
        if (ph.fin_kind == T_Q8_0) mega_quant_store<T_Q8_0>(v, c, tid, ph.fin_img, ph.fin_off_d, ph.fin_off_bs);
        else                       mega_quant_store<T_Q8_K>(v, c, tid, ph.fin_img, ph.fin_off_d, ph.fin_off_bs);
        mega_drain();
        if (tid == 0 && ph.signal) __hip_atomic_fetch_add(ph.signal, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    MG_STAMP(5);
}

// ---- the kernel: K-quant weight formats (Q4_K / Q5_K / Q6_K in any mixture: llama_tensor_get_type's Q4_K_M, Q5_K_M, Q6_K files) ----
__global__ void __launch_bounds__(MEGA_WG_THREADS, 2) k_mega_kquants(const mega_phase * __restrict__ prog, int n_phases, unsigned * err MG_STAMP_ARGS) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int b = (int) blockIdx.x;
    for (int i = 0; i < n_phases; i++) {
        const mega_phase & ph = prog[i];
        const int kind = ph.kind;
        if (kind == MEGA_FIN) {
            if (b == 0) mega_fin_norm(ph, smem, lane, wave);
        } else if (kind == MEGA_ATTN) {
            if (b < ph.n_active) mega_attn<128>(ph, smem, err MG_STAMP_PASS);
        } else if (kind == MEGA_MM) {
            if (b < ph.n_active) {
                const int be0 = ph.block_end[0], be1 = ph.block_end[1];
                const int gi = (b >= be0 ? 1 : 0) + (b >= be1 ? 1 : 0);
                const int first = gi == 0 ? 0 : (gi == 1 ? be0 : be1);
                const int last = gi == 0 ? be0 : (gi == 1 ? be1 : ph.block_end[2]);
                const mega_group & g = ph.g[gi];
                const int type = g.type;
                if (ph.glu) {
                    if (type == T_Q4_K)      mega_mm<T_Q4_K, true, 4>(ph, g, b - first, last - first, smem, lane, wave, err MG_STAMP_PASS);
                    else if (type == T_Q5_K) mega_mm<T_Q5_K, true, 4>(ph, g, b - first, last - first, smem, lane, wave, err MG_STAMP_PASS);
                    else                     mega_mm<T_Q6_K, true, 4>(ph, g, b - first, last - first, smem, lane, wave, err MG_STAMP_PASS);
                } else {
                    if (type == T_Q4_K)      mega_mm<T_Q4_K, false, 4>(ph, g, b - first, last - first, smem, lane, wave, err MG_STAMP_PASS);
                    else if (type == T_Q5_K) mega_mm<T_Q5_K, false, 4>(ph, g, b - first, last - first, smem, lane, wave, err MG_STAMP_PASS);
                    else                     mega_mm<T_Q6_K, false, 4>(ph, g, b - first, last - first, smem, lane, wave, err MG_STAMP_PASS);
                }
            }
        }
        __syncthreads();        // the LDS words are reused by the next phase
    }
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

void mega_launch(const mega_phase * prog_dev, int n_phases, int n_wg, unsigned * err, size_t lds_bytes, hipStream_t stream) {
#ifdef MI_STAMPS
    unsigned long long * st = (g_mega_stamp_buf && n_phases <= g_mega_stamp_phases && n_phases >= g_mega_stamp_last_n) ? g_mega_stamp_buf : nullptr;     // the long program of a token
    if (st) { g_mega_stamp_last_n = n_phases; g_mega_stamp_last_wg = n_wg; }
    hipLaunchKernelGGL(k_mega_kquants, dim3((unsigned) n_wg), dim3(MEGA_WG_THREADS), lds_bytes, stream, prog_dev, n_phases, err, st, 0);
#else
    hipLaunchKernelGGL(k_mega_kquants, dim3((unsigned) n_wg), dim3(MEGA_WG_THREADS), lds_bytes, stream, prog_dev, n_phases, err);
#endif
}

} // namespace mi355x
