// mm_dense.hip — f16/f32 x f32 -> f32 mat-mul with ggml's dim-2/3 broadcast, for the attention
// products of build_attn_mha (src/llama-graph.cpp:1285 kq = mul_mat(k, q); :1320 kqv = mul_mat(v, kq)):
// `a` is an F16 view of the KV cache (K: [head_dim, n_kv, n_head_kv], V transposed: [n_kv, head_dim, n_head_kv]),
// `b` F32 (possibly a permuted view), GQA broadcast r2 = ne12/ne02. Pinned by
// tests/test-backend-ops.cpp:5791-5813 (f16 x f32, permuted / strided, nr = [4,1]).
//
// At decode these are a few hundred KB of L2-resident data per layer: launch-bound, so one simple
// wave-per-output-row kernel (DPP reduction) serves; K/V bytes are read once per column tile of 8.
// dst[i0=i01, i1=i11, i2=i12, i3=i13] = sum_k a[k, i01, i12/r2, i13/r3] * b[k, i11, i12, i13]
#include "dev_common.h"
#include "kernels.h"

namespace mi355x {

static __device__ __forceinline__ float ld_a(const char * p, int type) {
    return type == T_F16 ? f16_bits_to_f32(*(const uint16_t *) p)
         : type == T_BF16 ? __builtin_bit_cast(float, (uint32_t)(*(const uint16_t *) p) << 16)
         : *(const float *) p;
}

template <int NC, int FAST>   // FAST 1: a contiguous f16 along k, b contiguous f32 along k; FAST 2: both contiguous f32, 16-byte aligned rows
                               // (the MoE router: F32 [n_embd, n_expert] x the normed activation, src/llama-graph.cpp:838)
__global__ void __launch_bounds__(256) k_mm_dense(const mm_dense_args p) {
    const int lane = threadIdx.x & 63;
    const int64_t i01 = (int64_t) blockIdx.x*4 + (threadIdx.x >> 6);
    if (i01 >= p.ne01) return;
    const int64_t c0 = (int64_t) blockIdx.y*NC;
    const int64_t i12 = blockIdx.z % p.ne12, i13 = blockIdx.z / p.ne12;
    const int64_t i02 = i12/(p.ne12/p.ne02), i03 = i13/(p.ne13/p.ne03);
    const char * a = (const char *) p.a + i01*p.nb01 + i02*p.nb02 + i03*p.nb03;
    const char * b = (const char *) p.b + i12*p.nb12 + i13*p.nb13;
    const int64_t K = p.ne00;

    float acc[NC];
#pragma unroll
    for (int c = 0; c < NC; c++) acc[c] = 0.0f;

    if (FAST == 2) {
        for (int k = lane*4; k < (int) K; k += 256) {   // K % 4 == 0; independent 16-byte loads, no type switch, 32-bit indices
            const float4v av = *(const float4v *) (a + (size_t) k*4);
#pragma unroll
            for (int c = 0; c < NC; c++) {
                if (c0 + c < p.ne11) {
                    const float4v bv = *(const float4v *) (b + (c0 + c)*p.nb11 + (size_t) k*4);
                    acc[c] += (av.x*bv.x + av.y*bv.y) + (av.z*bv.z + av.w*bv.w);
                }
            }
        }
    } else if (FAST == 1) {
        for (int64_t k = lane*2; k < K; k += 128) {   // K is even on this path
            const uint32_t av = ld_u32(a + k*2);
            const float a0 = f16_bits_to_f32((uint16_t)(av & 0xFFFF)), a1 = f16_bits_to_f32((uint16_t)(av >> 16));
#pragma unroll
            for (int c = 0; c < NC; c++) {
                if (c0 + c < p.ne11) {
                    const int2v bv = ld_b64(b + (c0 + c)*p.nb11 + k*4);
                    // NB: __builtin_bit_cast on an ext-vector ELEMENT (bv.y) reads element 0 with this toolchain (ROCm 7.2 clang) — go through scalars
                    const int bx = bv.x, by = bv.y;
                    // src1 is converted to src0's vec_dot_type first (F16), as the CPU backend's mat-mul does (tests/test-quantize-fns.cpp:82-99):
                    // the fused decode attention (decode_fused.hip) follows the same rule, so fused and node-by-node execution agree
                    acc[c] += a0*f16_bits_to_f32(f32_to_f16_bits(__int_as_float(bx))) + a1*f16_bits_to_f32(f32_to_f16_bits(__int_as_float(by)));
                }
            }
        }
    } else {
        for (int64_t k = lane; k < K; k += 64) {
            const float av = ld_a(a + k*p.nb00, p.type_a);
#pragma unroll
            for (int c = 0; c < NC; c++) {
                if (c0 + c < p.ne11) {
                    float bv = ld_a(b + (c0 + c)*p.nb11 + k*p.nb10, p.type_b);
                    if (p.type_a == T_F16 && p.type_b == T_F32) bv = f16_bits_to_f32(f32_to_f16_bits(bv));
                    acc[c] += av*bv;
                }
            }
        }
    }
#pragma unroll
    for (int c = 0; c < NC; c++) {
        const float s = wave_sum(acc[c]);
        if (lane == 0 && c0 + c < p.ne11) {
            *(float *) ((char *) p.dst + i01*4 + (c0 + c)*p.nb1 + i12*p.nb2 + i13*p.nb3) = s;
        }
    }
}

// The FAST == 2 case with many columns (the MoE router of a prompt pass: 32 x 2880 f32 against 512 tokens): a wave owns NR weight rows x NC columns, so a k-chunk
// is NR + NC loads for NR*NC*4 multiply-adds instead of 1 + NC for NC*4 — the same per-lane k assignment, expression and reduction as k_mm_dense<NC, 2>, hence the
// same bits (the router's logits decide the routing: nothing is rounded differently from the one-token path)
template <int NC, int NR>
__global__ void __launch_bounds__(256) k_mm_dense_rows(const mm_dense_args p) {
    const int lane = threadIdx.x & 63;
    const int64_t r0 = ((int64_t) blockIdx.x*4 + (threadIdx.x >> 6))*NR;
    if (r0 >= p.ne01) return;
    const int64_t c0 = (int64_t) blockIdx.y*NC;
    const int64_t i12 = blockIdx.z % p.ne12, i13 = blockIdx.z / p.ne12;
    const int64_t i02 = i12/(p.ne12/p.ne02), i03 = i13/(p.ne13/p.ne03);
    const char * a = (const char *) p.a + i02*p.nb02 + i03*p.nb03;
    const char * b = (const char *) p.b + i12*p.nb12 + i13*p.nb13;
    const int K = (int) p.ne00;
    float acc[NR][NC];
#pragma unroll
    for (int r = 0; r < NR; r++)
#pragma unroll
        for (int c = 0; c < NC; c++) acc[r][c] = 0.0f;
    for (int k = lane*4; k < K; k += 256) {
        float4v av[NR], bv[NC];
#pragma unroll
        for (int r = 0; r < NR; r++) av[r] = *(const float4v *) (a + (size_t) min(r0 + r, p.ne01 - 1)*p.nb01 + (size_t) k*4);
#pragma unroll
        for (int c = 0; c < NC; c++) bv[c] = *(const float4v *) (b + (size_t) min(c0 + c, p.ne11 - 1)*p.nb11 + (size_t) k*4);
#pragma unroll
        for (int r = 0; r < NR; r++)
#pragma unroll
            for (int c = 0; c < NC; c++) acc[r][c] += (av[r].x*bv[c].x + av[r].y*bv[c].y) + (av[r].z*bv[c].z + av[r].w*bv[c].w);
    }
#pragma unroll
    for (int r = 0; r < NR; r++)
#pragma unroll
        for (int c = 0; c < NC; c++) {
            const float s = wave_sum(acc[r][c]);
            if (lane == 0 && r0 + r < p.ne01 && c0 + c < p.ne11) *(float *) ((char *) p.dst + (r0 + r)*4 + (c0 + c)*p.nb1 + i12*p.nb2 + i13*p.nb3) = s;
        }
}

void mul_mat_dense(const mm_dense_args & p, hipStream_t stream) {
    if (p.ne01 == 0 || p.ne11 == 0 || p.ne12*p.ne13 == 0) return;
    const bool fast = p.type_a == T_F16 && p.type_b == T_F32 && p.nb00 == 2 && p.nb10 == 4 && (p.ne00 % 2 == 0);
    constexpr int NC = 8;
    const dim3 grid((unsigned)((p.ne01 + 3)/4), (unsigned)((p.ne11 + NC - 1)/NC), (unsigned)(p.ne12*p.ne13));
    const bool fast32 = p.type_a == T_F32 && p.type_b == T_F32 && p.nb00 == 4 && p.nb10 == 4 && p.ne00 % 4 == 0 && p.ne00 < (1ll << 30) &&
                        ((uintptr_t) p.a % 16) == 0 && ((uintptr_t) p.b % 16) == 0 && p.nb01 % 16 == 0 && p.nb02 % 16 == 0 && p.nb03 % 16 == 0 &&
                        p.nb11 % 16 == 0 && p.nb12 % 16 == 0 && p.nb13 % 16 == 0;
    if (fast)        hipLaunchKernelGGL((k_mm_dense<NC, 1>), grid, dim3(256), 0, stream, p);
    else if (fast32 && p.ne11 >= 64 && p.ne01 >= 32) {
        const dim3 g4((unsigned)((p.ne01 + 15)/16), (unsigned)((p.ne11 + NC - 1)/NC), (unsigned)(p.ne12*p.ne13));
        hipLaunchKernelGGL((k_mm_dense_rows<NC, 4>), g4, dim3(256), 0, stream, p);
    } else if (fast32 && p.ne11 >= 64 && p.ne01 >= 8) {      // (few experts: two rows per wave keep enough waves in flight)
        const dim3 g2((unsigned)((p.ne01 + 7)/8), (unsigned)((p.ne11 + NC - 1)/NC), (unsigned)(p.ne12*p.ne13));
        hipLaunchKernelGGL((k_mm_dense_rows<NC, 2>), g2, dim3(256), 0, stream, p);
    }
    else if (fast32) hipLaunchKernelGGL((k_mm_dense<NC, 2>), grid, dim3(256), 0, stream, p);
    else             hipLaunchKernelGGL((k_mm_dense<NC, 0>), grid, dim3(256), 0, stream, p);
}

// ---- MUL_MAT_ID over an F16 / BF16 / F32 expert stack (tests/test-backend-ops.cpp:5821-5824 with base_types F32 / F16; an unquantized MoE checkpoint) ----
// dst[:, slot, token] = as[:, :, ids[slot, token]] . b[:, slot % ne11, token]: one wave per (output row, pair), the expert index read on the device.
// Correctness path: a wave streams its row with whatever stride the stack has; F16 weights round src1 to F16 first, as ggml-cpu's F16 vec_dot does.
struct mm_id_dense_args { const char * as; int type_a; int64_t k, m; size_t nb00, nb01, nb02; const char * b; size_t nb10, nb11, nb12; int64_t n_b;
                          const char * ids; size_t ids_nb0, ids_nb1; int64_t n_used, n_tokens, n_expert; char * dst; size_t nb1, nb2; };
__global__ void __launch_bounds__(256) k_mm_id_dense(const mm_id_dense_args p) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t) blockIdx.x*4 + (threadIdx.x >> 6);
    if (row >= p.m) return;
    const int64_t slot = blockIdx.y, tok = blockIdx.z;
    const int e = *(const int32_t *) (p.ids + slot*p.ids_nb0 + tok*p.ids_nb1);
    float acc = 0.0f;
    if (e >= 0 && e < p.n_expert) {
        const char * a = p.as + (size_t) e*p.nb02 + row*p.nb01;
        const char * b = p.b + (slot % p.n_b)*p.nb11 + tok*p.nb12;
        for (int64_t k = lane; k < p.k; k += 64) {
            float bv = *(const float *) (b + k*p.nb10);
            if (p.type_a == T_F16) bv = f16_bits_to_f32(f32_to_f16_bits(bv));
            acc += ld_a(a + k*p.nb00, p.type_a)*bv;
        }
    }
    acc = wave_sum(acc);
    if (lane == 0) *(float *) (p.dst + row*4 + slot*p.nb1 + tok*p.nb2) = acc;
}
void mul_mat_id_dense(int type_a, const void * as, size_t nb00, size_t nb01, size_t nb02, int64_t m, int64_t k, const void * b, size_t nb10, size_t nb11, size_t nb12, int64_t n_b,
                      const void * ids, size_t ids_nb0, size_t ids_nb1, int64_t n_used, int64_t n_tokens, int64_t n_expert, float * dst, size_t nb1, size_t nb2, hipStream_t stream) {
    if (m == 0 || n_used*n_tokens == 0) return;
    const mm_id_dense_args p = { (const char *) as, type_a, k, m, nb00, nb01, nb02, (const char *) b, nb10, nb11, nb12, n_b, (const char *) ids, ids_nb0, ids_nb1, n_used, n_tokens, n_expert, (char *) dst, nb1, nb2 };
    hipLaunchKernelGGL(k_mm_id_dense, dim3((unsigned)((m + 3)/4), (unsigned) n_used, (unsigned) n_tokens), dim3(256), 0, stream, p);
}

} // namespace mi355x
