// attn_wo.hip — one token: attention AND the output projection as ONE launch with no wait inside.
//
// Before: [attention: one workgroup per head] -> boundary -> [wo mat-vec: every workgroup needs ALL heads' output] -> ... : a full-vector
// dependency, i.e. a launch boundary (~1.8 us) on either side of a 2 us kernel. Here the dependency is removed by redundancy and locality:
//   * workgroup b belongs to KV head kg = b % n_head_kv (with round-robin dispatch every XCD then works on one KV head: its K / V rows are
//     read from that XCD's L2) and owns rows [j R, (j + 1) R) of wo, j = b / n_head_kv, R = m / (workgroups per KV head);
//   * it computes the attention of ITS KV head's query heads itself (every workgroup of the group repeats it: for n_kv <= 256 that is
//     <= 128 KB of K / V from L2 per workgroup) — the gqa * head_dim values it gets are exactly `nbs` whole 256-blocks of wo's input vector,
//     so they are quantized to Q8_K blocks locally (no maximum or norm across workgroups);
//   * it multiplies them with its rows' k-slice of wo (nbs blocks of 144 / 176 bytes per row, requested at the very top of the kernel: the
//     loads fly while the attention runs) with the CPU's integer vec_dot (st_unit<T>::dot) and stores ONE OF n_head_kv PARTIAL PLANES:
//     plane[kg][row] = the slice's sum. The launch that consumes wo's result (norm + gate/up: mmvq_stream.h's prologue) adds the residual and
//     the planes in a fixed order — deterministic — and writes the sum where the graph expects the ADD's result.
// Arithmetic of the attention: as k_attn_decode<128, VT = true> (decode_fused.hip): q and p rounded to f16 like the CPU backend's F16
// mat-muls, f32 sums. The mat-vec: the same integer sub-sums as everywhere; only the order of the f32 additions over a row's blocks differs.
// Roofline: launch-latency-bound (9.4 MB of weights per layer at Llama-3-8B = 1.5 us of stream); what it was to buy is one launch and two
// boundaries per layer.
// MEASURED (round 3, Llama-3-8B Q4_K_M tg128, same box, back to back): 546 tok/s with this launch against 581 without — OPT-IN
// (GGML_MI355X_ATTN_WO=1), kept correct by tests/test_gpu_llama_graph.py. rocprofv3: the launch itself ~8 us (two attention rounds of two heads
// + the slice) in place of 2.2 (attention) + 6.9 (wo) + two boundaries — but the consumer pays more than that back: the norm + gate/up launch's
// prologue reads nine vectors instead of one, all cold in every XCD's L2 after the boundary (the planes were written through eight different
// L2s): 14.1 -> 17.1 us per launch, and the launches after it run 0.5-1 us slower as well. Removing a full-vector dependency by redundancy
// moves its cost into whoever has to add the pieces up.
#include "mmvq_stream.h"

#include <stdio.h>
#include <stdlib.h>

namespace mi355x {

struct aw_args {
    const char * q; size_t q_nb2;                  // q [128, 1, n_head] f32: head stride
    const char * k; size_t k_nb1, k_nb2;            // K cache view [128, n_kv, n_head_kv] f16: cell stride, head stride
    const char * v; size_t v_nb1, v_nb2;            // transposed V cache view [n_kv, 128, n_head_kv] f16: dim stride, head stride
    const char * mask; int mask_f16;                // [n_kv] (token 0's row)
    const float * sinks;
    int n_kv, n_head, n_head_kv; float scale;
    const char * W; size_t w_row_stride; int m, R, nbs;      // rows per workgroup; 256-blocks per k-slice (= gqa / 2)
    float * planes; int plane_stride;               // [n_head_kv][plane_stride] f32
};

static __device__ __forceinline__ float aw_dot8_f16_f32(const int4v kv, const float4v a, const float4v b) {      // = decode_fused.hip dot8_f16_f32
    const uint32_t k0 = (uint32_t) kv.x, k1 = (uint32_t) kv.y, k2 = (uint32_t) kv.z, k3 = (uint32_t) kv.w;
    float acc;
    acc  = f16_bits_to_f32((uint16_t) k0)*a.x + f16_bits_to_f32((uint16_t)(k0 >> 16))*a.y;
    acc += f16_bits_to_f32((uint16_t) k1)*a.z + f16_bits_to_f32((uint16_t)(k1 >> 16))*a.w;
    acc += f16_bits_to_f32((uint16_t) k2)*b.x + f16_bits_to_f32((uint16_t)(k2 >> 16))*b.y;
    acc += f16_bits_to_f32((uint16_t) k3)*b.z + f16_bits_to_f32((uint16_t)(k3 >> 16))*b.w;
    return acc;
}

template <int TYPE>
__global__ void __launch_bounds__(512) k_attn_wo(const aw_args p) {
    typedef st_unit<TYPE> U;
    constexpr int HD = 128;
    __shared__ float s_all[2][256];                 // scores -> probabilities of the two heads of a round
    __shared__ float sh_all[2][4];
    __shared__ __attribute__((aligned(16))) float out[8*HD];       // the KV head's query heads' outputs = the slice of wo's input
    __shared__ __attribute__((aligned(16))) char img[4*ST_ACT_STRIDE];
    __shared__ float dd[4];
    const int tid = threadIdx.x, lane = tid & 63;
    const int kg = blockIdx.x % p.n_head_kv, jb = blockIdx.x / p.n_head_kv;
    const int gqa = p.n_head/p.n_head_kv, nbs = p.nbs;

    // ---- this thread's unit of the k-slice: requested now, needed after the attention ----
    const int nunit = p.R*nbs;
    const int u = min(tid, nunit - 1);
    const int row = jb*p.R + u/nbs, blk = u - (u/nbs)*nbs;
    typename U::wfrag w;
    {
        const char * wp = p.W + (size_t) row*p.w_row_stride + (size_t)(kg*nbs + blk)*U::UB;
#pragma unroll
        for (int c = 0; c < U::UB/16; c++) w.c[c] = *(const int4v *) (wp + 16*c);
    }

    // ---- attention: two heads per round, four waves each (the transposed-V kernel of decode_fused.hip, one token) ----
    const int half = tid >> 8, wave = (tid >> 6) & 3, t256 = tid & 255;
    float * s = s_all[half]; float * sh = sh_all[half];
    constexpr int LPC = HD/8, CPW = 64/LPC, UQ = 4, NG = HD/16;
    const int sub = lane % LPC, cw = lane / LPC;
    const int n_kv = p.n_kv;
    for (int round = 0; round < nbs; round++) {
        const int hl = 2*round + half, h = kg*gqa + hl;
        const float * qp = (const float *) (p.q + (size_t) h*p.q_nb2) + sub*8;
        float4v q0 = *(const float4v *) qp, q1 = *(const float4v *) (qp + 4);
#define MI_R16(x_) x_ = f16_bits_to_f32(f32_to_f16_bits(x_))
        MI_R16(q0.x); MI_R16(q0.y); MI_R16(q0.z); MI_R16(q0.w); MI_R16(q1.x); MI_R16(q1.y); MI_R16(q1.z); MI_R16(q1.w);
        const char * kbase = p.k + (size_t) kg*p.k_nb2 + sub*16;
        const char * mrow = p.mask;
        int4v vpre[NG];
        {
            const int l16p = lane & 15, rwp = lane >> 4;
            const char * vb0 = p.v + (size_t) kg*p.v_nb2 + (size_t)(wave*4 + rwp)*p.v_nb1;
            const int c0 = min(l16p, max((n_kv >> 3) - 1, 0));
#pragma unroll
            for (int g = 0; g < NG; g++) vpre[g] = ld_b128(vb0 + (size_t)(g*16)*p.v_nb1 + (size_t) c0*16);
        }
        float mx = p.sinks ? p.sinks[h] : -INFINITY;
        for (int j0 = wave*CPW + cw; j0 < n_kv; j0 += 4*CPW*UQ) {
            int4v kreg[UQ]; float mreg[UQ];
#pragma unroll
            for (int uu = 0; uu < UQ; uu++) {
                const int j = min(j0 + uu*4*CPW, n_kv - 1);
                kreg[uu] = *(const int4v *) (kbase + (size_t) j*p.k_nb1);
                mreg[uu] = 0.0f;
                if (mrow) mreg[uu] = p.mask_f16 ? f16_bits_to_f32(*(const uint16_t *) (mrow + (size_t) j*2)) : *(const float *) (mrow + (size_t) j*4);
            }
#pragma unroll
            for (int uu = 0; uu < UQ; uu++) {
                const int j = j0 + uu*4*CPW;
                float acc = aw_dot8_f16_f32(kreg[uu], q0, q1);
                acc += dpp_f<0xB1>(acc); acc += dpp_f<0x4E>(acc); acc += dpp_f<0x141>(acc); acc += dpp_f<0x140>(acc);
                if (j < n_kv) {
                    const float v = acc*p.scale + mreg[uu];
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
        for (int j = t256; j < n_kv; j += 256) { const float e = expf(s[j] - mxs); s[j] = e; sum += e; }
        sum = wave_sum(sum);
        if (lane == 0) sh[wave] = sum;
        __syncthreads();
        sum = (sh[0] + sh[1]) + (sh[2] + sh[3]);      // (block_sum4's order)
        if (p.sinks) sum += expf(p.sinks[h] - mx);
        for (int j = t256; j < n_kv; j += 256) { float pj = sum > 0.0f ? s[j]/sum : 0.0f; MI_R16(pj); s[j] = pj; }
#undef MI_R16
        __syncthreads();
        const int l16 = lane & 15, rw = lane >> 4;
        const char * vbase = p.v + (size_t) kg*p.v_nb2 + (size_t)(wave*4 + rw)*p.v_nb1;
        float acc[NG];
#pragma unroll
        for (int g = 0; g < NG; g++) acc[g] = 0.0f;
        const int nchunk = n_kv >> 3;
        for (int c = l16; c < nchunk; c += 16) {
            int4v vreg[NG];
#pragma unroll
            for (int g = 0; g < NG; g++) vreg[g] = c == l16 ? vpre[g] : ld_b128(vbase + (size_t)(g*16)*p.v_nb1 + (size_t) c*16);
            const float4v p0 = *(const float4v *) (s + c*8), p1 = *(const float4v *) (s + c*8 + 4);
#pragma unroll
            for (int g = 0; g < NG; g++) acc[g] += aw_dot8_f16_f32(vreg[g], p0, p1);
        }
#pragma unroll
        for (int g = 0; g < NG; g++) {
            const float r = row16_sum(acc[g]);
            if (l16 == 0) out[hl*HD + g*16 + wave*4 + rw] = r;
        }
        __syncthreads();      // (s and sh are reused by the next round; out is read below)
    }

    // ---- the slice as Q8_K blocks (quant_core.h), laid out as the streamed kernel's units read it (mmvq_stream.h) ----
    if ((tid >> 6) < nbs) {
        const int b = tid >> 6;
        float d8; int bs16;
        const uint32_t q4 = quant_frag_q8_K(*(const float4v *) (out + b*256 + lane*4), d8, bs16);
        char * ab = img + (size_t) b*ST_ACT_STRIDE;
        *(uint32_t *) (ab + lane*4) = q4;
        const int bs32 = bs16 + dpp_i<0x114>(bs16);
        int hh, ll;
        st_hl(bs16, hh, ll);
        if ((lane & 3) == 0) { ab[272 + (lane >> 2)] = (char) hh; ab[288 + (lane >> 2)] = (char) ll; }
        st_hl(bs32, hh, ll);
        if ((lane & 7) == 4) { ab[256 + (lane >> 3)] = (char) hh; ab[264 + (lane >> 3)] = (char) ll; }
        if (lane == 0) dd[b] = d8;
    }
    __syncthreads();

    // ---- rows x slice ----
    float res = U::dot(w, img + (size_t) blk*ST_ACT_STRIDE, dd[blk]);
    if (nbs >= 2) res += dpp_f<0xB1>(res);
    if (nbs >= 4) res += dpp_f<0x4E>(res);
    if (tid < nunit && blk == 0) p.planes[(size_t) kg*p.plane_stride + row] = res;
}

// x_out[i] = res[i] + plane 0 [i] + plane 1 [i] + ... (the order the consuming launch's prologue uses): the stand-alone form, for a consumer that is not
// the streamed kernel
__global__ void __launch_bounds__(256) k_planes_sum(const float * res, const float * planes, int n_planes, int plane_stride, float * x_out, int m) {
    const int i = blockIdx.x*256 + threadIdx.x;
    if (i >= m) return;
    float x = res[i];
    for (int pl = 0; pl < n_planes; pl++) x += planes[(size_t) pl*plane_stride + i];
    x_out[i] = x;
}
void planes_sum(const float * res, const float * planes, int n_planes, int plane_stride, float * x_out, int64_t m, hipStream_t stream) {
    hipLaunchKernelGGL(k_planes_sum, dim3((unsigned)((m + 255)/256)), dim3(256), 0, stream, res, planes, n_planes, plane_stride, x_out, (int) m);
}

static int aw_cu_count() {
    static int n_cu = 0;
    if (n_cu == 0) {
        int dev = 0; hipDeviceProp_t prop;
        if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) n_cu = prop.multiProcessorCount;
        if (n_cu <= 0) n_cu = 256;
    }
    return n_cu;
}
// workgroups per KV head for this shape (0: the shape is not taken)
static int aw_groups(int type, int64_t m, int64_t k, int64_t hd, int64_t n_kv, int64_t n_head, int64_t n_head_kv) {
    if ((type != T_Q4_K && type != T_Q5_K) || hd != 128 || n_kv < 8 || n_kv > 256 || n_kv % 8 || n_head_kv < 1 || n_head % n_head_kv || k != n_head*hd) return 0;
    const int64_t gqa = n_head/n_head_kv;
    if (gqa != 2 && gqa != 4 && gqa != 8) return 0;
    const int64_t nbs = gqa/2;
    int64_t G = aw_cu_count()/n_head_kv;
    while (G > 1 && (m % G != 0 || (m/G)*nbs > 512)) G--;
    if (G < 1 || m % G != 0 || (m/G)*nbs > 512) return 0;
    // (fewer than half the CUs busy is not worth the redundancy)
    if (G*n_head_kv*2 < aw_cu_count()) return 0;
    return (int) G;
}
bool attn_wo_supported(int type, int64_t m, int64_t k, int64_t hd, int64_t n_kv, int64_t n_head, int64_t n_head_kv) {
    static int on = -1;
    if (on < 0) { const char * e = getenv("GGML_MI355X_ATTN_WO"); on = e ? atoi(e) : 0; }      // opt-in: measured slower (see the header)
    return on && aw_groups(type, m, k, hd, n_kv, n_head, n_head_kv) > 0;
}
void attn_wo(const void * q, size_t q_nb2, const void * k, size_t k_nb1, size_t k_nb2, const void * v, size_t v_nb1, size_t v_nb2,
             const void * mask, bool mask_f16, const float * sinks, int64_t hd, int64_t n_kv, int64_t n_head, int64_t n_head_kv, float scale,
             int type, const void * W, size_t w_row_stride, int64_t m, float * planes, int64_t plane_stride, hipStream_t stream) {
    const int G = aw_groups(type, m, n_head*hd, hd, n_kv, n_head, n_head_kv);
    if (G <= 0) { fprintf(stderr, "attn_wo: unsupported shape\n"); abort(); }
    aw_args a = { (const char *) q, q_nb2, (const char *) k, k_nb1, k_nb2, (const char *) v, v_nb1, v_nb2, (const char *) mask, mask_f16 ? 1 : 0, sinks,
                  (int) n_kv, (int) n_head, (int) n_head_kv, scale, (const char *) W, w_row_stride, (int) m, (int)(m/G), (int)(n_head/n_head_kv/2), planes, (int) plane_stride };
    const dim3 grid((unsigned)(G*n_head_kv));
    if (type == T_Q4_K) hipLaunchKernelGGL((k_attn_wo<T_Q4_K>), grid, dim3(512), 0, stream, a);
    else                hipLaunchKernelGGL((k_attn_wo<T_Q5_K>), grid, dim3(512), 0, stream, a);
}

} // namespace mi355x
