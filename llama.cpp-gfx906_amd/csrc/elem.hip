// elem.hip — the element kernels a Llama-family decode graph needs besides the mat-muls
// (SURVEY.md Appendix A). Each launcher names the reference call site that emits the op and the
// reference test that pins it. All are HBM/launch-bound: coalesced loads along ne0, wave64 DPP
// reductions, no LDS beyond a 4-float cross-wave exchange.
#include "dev_common.h"
#include "kernels.h"

#include <math.h>

namespace mi355x {

struct td {   // device-side copy of tensor_desc
    char * data; int type; int64_t ne0, ne1, ne2, ne3; size_t nb0, nb1, nb2, nb3;
};
static td mk(const tensor_desc & t) {
    return td{ (char *) t.data, t.type, t.ne[0], t.ne[1], t.ne[2], t.ne[3], t.nb[0], t.nb[1], t.nb[2], t.nb[3] };
}

static __device__ __forceinline__ float ld_elem(const char * p, int type) {
    switch (type) {
        case T_F32: return *(const float *) p;
        case T_F16: return f16_bits_to_f32(*(const uint16_t *) p);
        case T_BF16: return __builtin_bit_cast(float, (uint32_t)(*(const uint16_t *) p) << 16);
        case T_I32: return (float) *(const int32_t *) p;
        default: return 0.0f;
    }
}
static __device__ __forceinline__ void st_elem(char * p, int type, float v) {
    switch (type) {
        case T_F32: *(float *) p = v; break;
        case T_F16: *(uint16_t *) p = f32_to_f16_bits(v); break;
        case T_BF16: {
            // round-to-nearest-even, NaN stays NaN (ggml_compute_fp32_to_bf16 semantics)
            uint32_t u = __builtin_bit_cast(uint32_t, v);
            uint16_t h;
            if ((u & 0x7fffffff) > 0x7f800000) h = (uint16_t)((u >> 16) | 64);
            else h = (uint16_t)((u + (0x7fff + ((u >> 16) & 1))) >> 16);
            *(uint16_t *) p = h;
        } break;
        case T_I32: *(int32_t *) p = (int32_t) v; break;
        default: break;
    }
}

// linear index -> 4-D coordinates over (e0, e1, e2, *). 64-bit division is ~10x the cost of 32-bit on the VALU and these kernels
// are tiny, so the 32-bit path is taken whenever the element count allows (always, on the decode/prefill path).
struct idx4 { int64_t i0, i1, i2, i3; };
static __device__ __forceinline__ idx4 unravel(int64_t i, int64_t e0, int64_t e1, int64_t e2) {
    idx4 r;
    if ((e0 | e1 | e2 | i) < (1ll << 31)) {
        uint32_t u = (uint32_t) i; const uint32_t a = (uint32_t) e0, b = (uint32_t) e1, c = (uint32_t) e2;
        r.i0 = u % a; u /= a; r.i1 = u % b; u /= b; r.i2 = u % c; r.i3 = u / c;
    } else {
        r.i0 = i % e0; r.i1 = (i / e0) % e1; r.i2 = (i / (e0*e1)) % e2; r.i3 = i / (e0*e1*e2);
    }
    return r;
}
static __device__ __forceinline__ int64_t wrap(int64_t i, int64_t e) { return e == 1 ? 0 : (i < e ? i : ((i | e) < (1ll << 31) ? (int64_t)((uint32_t) i % (uint32_t) e) : i % e)); }

// block-wide sum / max for 256-thread workgroups (4 waves)
static __device__ __forceinline__ float block_sum(float v, float * sh) {
    v = wave_sum(v);
    const int w = threadIdx.x >> 6, nw = blockDim.x >> 6;
    if (nw == 1) return v;
    __syncthreads();
    if ((threadIdx.x & 63) == 0) sh[w] = v;
    __syncthreads();
    float r = sh[0];
    for (int i = 1; i < nw; i++) r += sh[i];
    return r;
}
static __device__ __forceinline__ float block_max(float v, float * sh) {
    v = wave_max(v);
    const int w = threadIdx.x >> 6, nw = blockDim.x >> 6;
    if (nw == 1) return v;
    __syncthreads();
    if ((threadIdx.x & 63) == 0) sh[w] = v;
    __syncthreads();
    float r = sh[0];
    for (int i = 1; i < nw; i++) r = fmaxf(r, sh[i]);
    return r;
}

// ---- RMS_NORM (+MUL (+ADD)) — src/llama-graph.cpp:597-630; tests/test-backend-ops.cpp:2773,2856 ------
// y = x / sqrt(mean(x^2) + eps) [* w] [+ add]; one workgroup per row.
template <bool HAS_W, bool HAS_ADD>
__global__ void __launch_bounds__(256) k_rms_norm(const td src, const td w, const td add, const td dst, float eps, uint16_t * y16, int kp16) {
    // y16 != NULL (the host has checked the conditions of the 16-byte path below): also the bf16 copy of the result the prefill mat-mul reads,
    // rows of kp16 elements with a zero tail (mmq.hip k_act_to_16's layout) — saves that pass
    __shared__ float sh[4];
    const int64_t row = blockIdx.x;
    const idx4 rx = unravel(row, src.ne1, src.ne2, 1ll << 30); const int64_t i1 = rx.i0, i2 = rx.i1, i3 = rx.i2;
    const char * x = src.data + i1*src.nb1 + i2*src.nb2 + i3*src.nb3;
    char * y = dst.data + i1*dst.nb1 + i2*dst.nb2 + i3*dst.nb3;
    float ss = 0.0f;
    const bool vec = (src.ne0 % 4 == 0) && (((uintptr_t) x | (uintptr_t) y) % 16 == 0);
    if (vec) {
        for (int64_t i = threadIdx.x*4; i < src.ne0; i += blockDim.x*4) {
            const float4v v = *(const float4v *) (x + i*4);
            ss += v.x*v.x + v.y*v.y + v.z*v.z + v.w*v.w;
        }
    } else {
        for (int64_t i = threadIdx.x; i < src.ne0; i += blockDim.x) { const float v = *(const float *) (x + i*4); ss += v*v; }
    }
    ss = block_sum(ss, sh);
    const float mean = ss / (float) src.ne0;
    const float scale = 1.0f/sqrtf(mean + eps);
    const char * wp = nullptr; const char * ap = nullptr;
    if (HAS_W)   wp = w.data   + wrap(i1, w.ne1)*w.nb1     + wrap(i2, w.ne2)*w.nb2     + wrap(i3, w.ne3)*w.nb3;
    if (HAS_ADD) ap = add.data + wrap(i1, add.ne1)*add.nb1 + wrap(i2, add.ne2)*add.nb2 + wrap(i3, add.ne3)*add.nb3;
    const bool w_full = !HAS_W || w.ne0 == src.ne0, a_full = !HAS_ADD || add.ne0 == src.ne0;   // no wrap needed (the model case)
    if (vec && w_full && a_full && (!HAS_W || (uintptr_t) wp % 16 == 0) && (!HAS_ADD || (uintptr_t) ap % 16 == 0) && src.ne0 < (1ll << 30)) {
        // the model case: 16-byte loads and stores, 32-bit indices (the scalar loop below cost 11.7 us for one 4096-float row)
        const int n0 = (int) src.ne0;
        for (int i = threadIdx.x*4; i < n0; i += blockDim.x*4) {
            float4v v = *(const float4v *) (x + (size_t) i*4);
            v.x *= scale; v.y *= scale; v.z *= scale; v.w *= scale;
            if (HAS_W)   { const float4v t = *(const float4v *) (wp + (size_t) i*4); v.x *= t.x; v.y *= t.y; v.z *= t.z; v.w *= t.w; }
            if (HAS_ADD) { const float4v t = *(const float4v *) (ap + (size_t) i*4); v.x += t.x; v.y += t.y; v.z += t.z; v.w += t.w; }
            *(float4v *) (y + (size_t) i*4) = v;
            if (y16) *(uint2 *) (y16 + (size_t) row*kp16 + i) = uint2{ pack_bf16(v.x, v.y), pack_bf16(v.z, v.w) };
        }
        if (y16) for (int i = n0 + threadIdx.x*4; i < kp16; i += blockDim.x*4) *(uint2 *) (y16 + (size_t) row*kp16 + i) = uint2{ 0u, 0u };
        return;
    }
    for (int64_t i = threadIdx.x; i < src.ne0; i += blockDim.x) {
        float v = *(const float *) (x + i*4) * scale;
        if (HAS_W)   v *= *(const float *) (wp + (w_full ? i : wrap(i, w.ne0))*4);
        if (HAS_ADD) v += *(const float *) (ap + (a_full ? i : wrap(i, add.ne0))*4);
        *(float *) (y + i*4) = v;
    }
}

static int rows_block(int64_t ne0) { return ne0 >= 1024 ? 256 : (ne0 >= 256 ? 128 : 64); }

void rms_norm(const tensor_desc & src, const tensor_desc & dst, float eps, hipStream_t stream) {
    const int64_t nrows = src.ne[1]*src.ne[2]*src.ne[3];
    if (nrows == 0) return;
    hipLaunchKernelGGL((k_rms_norm<false, false>), dim3((unsigned) nrows), dim3(rows_block(src.ne[0])), 0, stream, mk(src), td{}, td{}, mk(dst), eps, (uint16_t *) nullptr, 0);
}
bool rms_norm_mul_bf16_supported(const tensor_desc & src, const tensor_desc & w, const tensor_desc & dst) {   // the kernel's 16-byte path, 2-d rows
    return src.ne[0] % 4 == 0 && src.ne[0] < (1ll << 30) && src.ne[2] == 1 && src.ne[3] == 1 && w.ne[0] == src.ne[0] && w.ne[1] == 1 && w.ne[2] == 1 && w.ne[3] == 1 &&
           ((uintptr_t) src.data % 16) == 0 && src.nb[1] % 16 == 0 && ((uintptr_t) dst.data % 16) == 0 && dst.nb[1] % 16 == 0 && ((uintptr_t) w.data % 16) == 0 &&
           src.nb[0] == 4 && dst.nb[0] == 4 && w.nb[0] == 4;
}
void rms_norm_mul(const tensor_desc & src, const tensor_desc & w, const tensor_desc * add, const tensor_desc & dst, float eps, hipStream_t stream, uint16_t * y16) {
    const int64_t nrows = src.ne[1]*src.ne[2]*src.ne[3];
    if (nrows == 0) return;
    const dim3 g((unsigned) nrows), b(rows_block(src.ne[0]));
    const int kp16 = (int)((src.ne[0] + 63) & ~(int64_t) 63);
    if (add) hipLaunchKernelGGL((k_rms_norm<true, true>),  g, b, 0, stream, mk(src), mk(w), mk(*add), mk(dst), eps, (uint16_t *) nullptr, 0);
    else     hipLaunchKernelGGL((k_rms_norm<true, false>), g, b, 0, stream, mk(src), mk(w), td{},     mk(dst), eps, y16, kp16);
}

// ---- the pass that adds a split-k mat-mul's planes (+ residual), fused with the RMS_NORM * w that reads the sum next (prefill: wo -> ADD ->
//      ffn_norm, ffn_down -> ADD -> next layer's attn_norm): x = plane 0 + plane 1 [+ ...] + res in mmq.hip k_combine's order; x is written
//      (the residual stream), y = x / sqrt(mean(x^2) + eps) * w, and y's bf16 copy for the mat-mul after the norm ----
template <int NP>
__global__ void __launch_bounds__(256) k_combine_rms_norm(const float * planes, int64_t plane_elems, const char * res, size_t res_nb1, char * sum_out, size_t sum_nb1,
                                                          const float * w, char * y, size_t y_nb1, uint16_t * y16, int kp16, int m, float eps) {
    __shared__ float sh[4];
    const int64_t row = blockIdx.x;
    const float * pl = planes + row*m;
    const char * rr = res + row*res_nb1; char * so = sum_out + row*sum_nb1; char * yo = y + row*y_nb1;
    float ss = 0.0f;
    for (int i = threadIdx.x*4; i < m; i += 256*4) {
        float4v a = *(const float4v *) (pl + i);
#pragma unroll
        for (int p = 1; p < NP; p++) { const float4v b = *(const float4v *) (pl + (int64_t) p*plane_elems + i); a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w; }
        { const float4v b = *(const float4v *) (rr + (size_t) i*4); a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w; }
        *(float4v *) (so + (size_t) i*4) = a;
        ss += a.x*a.x + a.y*a.y + a.z*a.z + a.w*a.w;
    }
    ss = block_sum(ss, sh);
    const float scale = 1.0f/sqrtf(ss/(float) m + eps);
    for (int i = threadIdx.x*4; i < m; i += 256*4) {
        float4v v = *(const float4v *) (so + (size_t) i*4);          // written by this thread above
        const float4v t = *(const float4v *) (w + i);
        v.x *= scale; v.y *= scale; v.z *= scale; v.w *= scale;
        v.x *= t.x; v.y *= t.y; v.z *= t.z; v.w *= t.w;
        *(float4v *) (yo + (size_t) i*4) = v;
        if (y16) *(uint2 *) (y16 + (size_t) row*kp16 + i) = uint2{ pack_bf16(v.x, v.y), pack_bf16(v.z, v.w) };
    }
    if (y16) for (int i = m + threadIdx.x*4; i < kp16; i += 256*4) *(uint2 *) (y16 + (size_t) row*kp16 + i) = uint2{ 0u, 0u };
}
void combine_rms_norm(const float * planes, int np, int64_t m, int64_t n, const float * res, size_t res_nb1, float * sum_out, size_t sum_nb1,
                      const float * w, float * y, size_t y_nb1, uint16_t * y16, float eps, hipStream_t stream) {
    if (m == 0 || n == 0) return;
    const int kp16 = (int)((m + 63) & ~(int64_t) 63);
#define MI_CRN(NP_) hipLaunchKernelGGL((k_combine_rms_norm<NP_>), dim3((unsigned) n), dim3(256), 0, stream, planes, m*n, (const char *) res, res_nb1, (char *) sum_out, sum_nb1, \
                                       w, (char *) y, y_nb1, y16, kp16, (int) m, eps)
    if (np == 2) MI_CRN(2); else if (np == 4) MI_CRN(4); else MI_CRN(8);
#undef MI_CRN
}

// ---- ADD / MUL / DIV / SUB with ggml repeat-broadcast — tests/test-backend-ops.cpp:2469 ----------------
template <int OP>
__global__ void __launch_bounds__(256) k_bin_bcast(const td a, const td b, const td dst, int64_t n) {
    const int64_t i = (int64_t) blockIdx.x*256 + threadIdx.x;
    if (i >= n) return;
    const idx4 ix = unravel(i, dst.ne0, dst.ne1, dst.ne2); const int64_t i0 = ix.i0, i1 = ix.i1, i2 = ix.i2, i3 = ix.i3;
    const float x = ld_elem(a.data + i0*a.nb0 + i1*a.nb1 + i2*a.nb2 + i3*a.nb3, a.type);
    const float y = ld_elem(b.data + wrap(i0, b.ne0)*b.nb0 + wrap(i1, b.ne1)*b.nb1 + wrap(i2, b.ne2)*b.nb2 + wrap(i3, b.ne3)*b.nb3, b.type);
    float r;
    if (OP == BIN_ADD) r = x + y; else if (OP == BIN_MUL) r = x*y; else if (OP == BIN_DIV) r = x/y; else r = x - y;
    st_elem(dst.data + i0*dst.nb0 + i1*dst.nb1 + i2*dst.nb2 + i3*dst.nb3, dst.type, r);
}
void bin_bcast(int op, const tensor_desc & a, const tensor_desc & b, const tensor_desc & dst, hipStream_t stream) {
    const int64_t n = dst.ne[0]*dst.ne[1]*dst.ne[2]*dst.ne[3];
    if (n == 0) return;
    const dim3 g((unsigned)((n + 255)/256));
    switch (op) {
        case BIN_ADD: hipLaunchKernelGGL((k_bin_bcast<BIN_ADD>), g, dim3(256), 0, stream, mk(a), mk(b), mk(dst), n); break;
        case BIN_MUL: hipLaunchKernelGGL((k_bin_bcast<BIN_MUL>), g, dim3(256), 0, stream, mk(a), mk(b), mk(dst), n); break;
        case BIN_DIV: hipLaunchKernelGGL((k_bin_bcast<BIN_DIV>), g, dim3(256), 0, stream, mk(a), mk(b), mk(dst), n); break;
        default:      hipLaunchKernelGGL((k_bin_bcast<BIN_SUB>), g, dim3(256), 0, stream, mk(a), mk(b), mk(dst), n); break;
    }
}

// ---- ADD_ID — src/llama-graph.cpp:927,940,985; tests/test-backend-ops.cpp:2548 ---------------------------
// out[:, iu, it] = a[:, iu, it] + bias[:, ids[iu, it]]
__global__ void __launch_bounds__(256) k_add_id(const td a, const td bias, const td ids, const td dst, int64_t n) {
    const int64_t i = (int64_t) blockIdx.x*256 + threadIdx.x;
    if (i >= n) return;
    const idx4 ix = unravel(i, dst.ne0, dst.ne1, 1ll << 30); const int64_t i0 = ix.i0, iu = ix.i1, it = ix.i2;
    const int e = *(const int32_t *) (ids.data + iu*ids.nb0 + it*ids.nb1);
    const float x = *(const float *) (a.data + i0*a.nb0 + iu*a.nb1 + it*a.nb2);
    const float y = *(const float *) (bias.data + i0*bias.nb0 + (int64_t) e*bias.nb1);
    *(float *) (dst.data + i0*dst.nb0 + iu*dst.nb1 + it*dst.nb2) = x + y;
}
void add_id(const tensor_desc & a, const tensor_desc & bias, const tensor_desc & ids, const tensor_desc & dst, hipStream_t stream) {
    const int64_t n = dst.ne[0]*dst.ne[1]*dst.ne[2];
    if (n == 0) return;
    hipLaunchKernelGGL(k_add_id, dim3((unsigned)((n + 255)/256)), dim3(256), 0, stream, mk(a), mk(bias), mk(ids), mk(dst), n);
}

// ---- SCALE — tests/test-backend-ops.cpp:2643 ----------------------------------------------------------------
__global__ void __launch_bounds__(256) k_scale(const td src, const td dst, float s, float b, int64_t n) {
    const int64_t i = (int64_t) blockIdx.x*256 + threadIdx.x;
    if (i >= n) return;
    const idx4 ix = unravel(i, dst.ne0, dst.ne1, dst.ne2); const int64_t i0 = ix.i0, i1 = ix.i1, i2 = ix.i2, i3 = ix.i3;
    const float x = *(const float *) (src.data + i0*src.nb0 + i1*src.nb1 + i2*src.nb2 + i3*src.nb3);
    *(float *) (dst.data + i0*dst.nb0 + i1*dst.nb1 + i2*dst.nb2 + i3*dst.nb3) = x*s + b;
}
void scale(const tensor_desc & src, const tensor_desc & dst, float s, float b, hipStream_t stream) {
    const int64_t n = dst.ne[0]*dst.ne[1]*dst.ne[2]*dst.ne[3];
    if (n == 0) return;
    hipLaunchKernelGGL(k_scale, dim3((unsigned)((n + 255)/256)), dim3(256), 0, stream, mk(src), mk(dst), s, b, n);
}

// ---- CPY / CONT / DUP — src/llama-graph.cpp:1317,1330; tests/test-backend-ops.cpp:2383,2438 ---------------
// element i (in src order) is written to element i of dst (in dst order); shapes may differ, nelements equal
__global__ void __launch_bounds__(256) k_cpy(const td src, const td dst, int64_t n) {
    const int64_t i = (int64_t) blockIdx.x*256 + threadIdx.x;
    if (i >= n) return;
    const idx4 sx = unravel(i, src.ne0, src.ne1, src.ne2), dx = unravel(i, dst.ne0, dst.ne1, dst.ne2);
    const int64_t s0 = sx.i0, s1 = sx.i1, s2 = sx.i2, s3 = sx.i3, d0 = dx.i0, d1 = dx.i1, d2 = dx.i2, d3 = dx.i3;
    const char * sp = src.data + s0*src.nb0 + s1*src.nb1 + s2*src.nb2 + s3*src.nb3;
    char * dp = dst.data + d0*dst.nb0 + d1*dst.nb1 + d2*dst.nb2 + d3*dst.nb3;
    if (src.type == dst.type) {
        if (src.type == T_F32 || src.type == T_I32) *(uint32_t *) dp = *(const uint32_t *) sp;
        else *(uint16_t *) dp = *(const uint16_t *) sp;
    } else {
        st_elem(dp, dst.type, ld_elem(sp, src.type));
    }
}
void cpy(const tensor_desc & src, const tensor_desc & dst, hipStream_t stream) {
    const int64_t n = src.ne[0]*src.ne[1]*src.ne[2]*src.ne[3];
    if (n == 0) return;
    hipLaunchKernelGGL(k_cpy, dim3((unsigned)((n + 255)/256)), dim3(256), 0, stream, mk(src), mk(dst), n);
}

// ---- SET_ROWS — src/llama-kv-cache-unified.cpp:1123,1154,1167; tests/test-backend-ops.cpp:2060-2127 -------
// dst[idx[i1, i2 % idx.ne1, i3 % idx.ne2], i2, i3][:] = convert(src[:, i1, i2, i3]); idx is I64
__global__ void __launch_bounds__(256) k_set_rows(const td src, const td idx, const td dst, int64_t n) {
    const int64_t i = (int64_t) blockIdx.x*256 + threadIdx.x;
    if (i >= n) return;
    const idx4 ix = unravel(i, src.ne0, src.ne1, src.ne2); const int64_t i0 = ix.i0, i1 = ix.i1, i2 = ix.i2, i3 = ix.i3;
    const int64_t r = *(const int64_t *) (idx.data + i1*idx.nb0 + wrap(i2, idx.ne1)*idx.nb1 + wrap(i3, idx.ne2)*idx.nb2);
    const float v = *(const float *) (src.data + i0*src.nb0 + i1*src.nb1 + i2*src.nb2 + i3*src.nb3);
    st_elem(dst.data + i0*dst.nb0 + r*dst.nb1 + i2*dst.nb2 + i3*dst.nb3, dst.type, v);
}
// quantized destination (a quantized KV cache: -ctk q8_0 / q4_0; tests/test-backend-ops.cpp:5333-5343): one thread per 32-element
// block, the reference row quantizers restated operation for operation (gguf-py/gguf/quants.py:222-238 Q4_0, :381-393 Q8_0 —
// oracle/ggml_oracle.c quantize_row_q4_0_ref / quantize_row_q8_0_ref, pinned bit-exactly by tests/golden/quant_*.npz)
template <int TYPE>
__global__ void __launch_bounds__(256) k_set_rows_q(const td src, const td idx, const td dst, int64_t n_blocks) {
    const int64_t i = (int64_t) blockIdx.x*256 + threadIdx.x;
    if (i >= n_blocks) return;
    const int64_t nb0 = src.ne0/32;
    const idx4 ix = unravel(i, nb0, src.ne1, src.ne2); const int64_t ib = ix.i0, i1 = ix.i1, i2 = ix.i2, i3 = ix.i3;
    const int64_t r = *(const int64_t *) (idx.data + i1*idx.nb0 + wrap(i2, idx.ne1)*idx.nb1 + wrap(i3, idx.ne2)*idx.nb2);
    const char * x = src.data + (ib*32)*src.nb0 + i1*src.nb1 + i2*src.nb2 + i3*src.nb3;
    float v[32];
#pragma unroll
    for (int j = 0; j < 32; j++) v[j] = *(const float *) (x + j*src.nb0);
    uint8_t * out = (uint8_t *) (dst.data + r*dst.nb1 + i2*dst.nb2 + i3*dst.nb3) + ib*(TYPE == T_Q8_0 ? 34 : 18);
    if (TYPE == T_Q8_0) {
        float amax = 0.0f;
#pragma unroll
        for (int j = 0; j < 32; j++) { const float a = fabsf(v[j]); if (a > amax) amax = a; }
        const float d = amax/127.0f;
        const float id = d != 0.0f ? 1.0f/d : 0.0f;
        const uint16_t dh = f32_to_f16_bits(d);
        out[0] = (uint8_t) dh; out[1] = (uint8_t)(dh >> 8);
#pragma unroll
        for (int j = 0; j < 32; j++) out[2 + j] = (uint8_t)(int8_t) roundf(v[j]*id);
    } else {
        float amax = 0.0f, mx = 0.0f;
#pragma unroll
        for (int j = 0; j < 32; j++) { if (amax < fabsf(v[j])) { amax = fabsf(v[j]); mx = v[j]; } }
        float d = mx/-8.0f;
        // keep the product in a register of its own: fused into the f16 conversion (v_fma_mixlo_f16 with a +0 addend) a zero block's
        // d = 0 / -8 = -0.0 came out as +0.0, one bit off the reference's bytes
        asm volatile("" : "+v"(d));
        const float id = d != 0.0f ? 1.0f/d : 0.0f;
        const uint16_t dh = f32_to_f16_bits(d);
        out[0] = (uint8_t) dh; out[1] = (uint8_t)(dh >> 8);
#pragma unroll
        for (int j = 0; j < 16; j++) {
            const int v0 = (int)(v[j]*id + 8.5f), v1 = (int)(v[16 + j]*id + 8.5f);
            out[2 + j] = (uint8_t)((v0 < 15 ? v0 : 15) | ((v1 < 15 ? v1 : 15) << 4));
        }
    }
}
void set_rows(const tensor_desc & src, const tensor_desc & idx, const tensor_desc & dst, hipStream_t stream) {
    const int64_t n = src.ne[0]*src.ne[1]*src.ne[2]*src.ne[3];
    if (n == 0) return;
    if (dst.type == T_Q8_0 || dst.type == T_Q4_0) {
        const int64_t nblk = n/32;
        if (dst.type == T_Q8_0) hipLaunchKernelGGL((k_set_rows_q<T_Q8_0>), dim3((unsigned)((nblk + 255)/256)), dim3(256), 0, stream, mk(src), mk(idx), mk(dst), nblk);
        else                    hipLaunchKernelGGL((k_set_rows_q<T_Q4_0>), dim3((unsigned)((nblk + 255)/256)), dim3(256), 0, stream, mk(src), mk(idx), mk(dst), nblk);
        return;
    }
    hipLaunchKernelGGL(k_set_rows, dim3((unsigned)((n + 255)/256)), dim3(256), 0, stream, mk(src), mk(idx), mk(dst), n);
}

// ---- GET_ROWS — src/llama-graph.cpp:887, src/llama-model.cpp:6053; tests/test-backend-ops.cpp:1951 --------
// dst[:, i10, i11, i12] = src[:, idx[i10, i11, i12], i11, i12]
__global__ void __launch_bounds__(256) k_get_rows(const td src, const td idx, const td dst, int64_t n) {
    const int64_t i = (int64_t) blockIdx.x*256 + threadIdx.x;
    if (i >= n) return;
    const idx4 ix = unravel(i, dst.ne0, dst.ne1, dst.ne2); const int64_t i0 = ix.i0, i10 = ix.i1, i11 = ix.i2, i12 = ix.i3;
    const int64_t r = *(const int32_t *) (idx.data + i10*idx.nb0 + i11*idx.nb1 + i12*idx.nb2);
    const float v = ld_elem(src.data + i0*src.nb0 + r*src.nb1 + i11*src.nb2 + i12*src.nb3, src.type);
    st_elem(dst.data + i0*dst.nb0 + i10*dst.nb1 + i11*dst.nb2 + i12*dst.nb3, dst.type, v);
}
void get_rows(const tensor_desc & src, const tensor_desc & idx, const tensor_desc & dst, hipStream_t stream) {
    const int64_t n = dst.ne[0]*dst.ne[1]*dst.ne[2]*dst.ne[3];
    if (n == 0) return;
    hipLaunchKernelGGL(k_get_rows, dim3((unsigned)((n + 255)/256)), dim3(256), 0, stream, mk(src), mk(idx), mk(dst), n);
}

// ---- SUM_ROWS — src/llama-graph.cpp:901; tests/test-backend-ops.cpp:4203 ----------------------------------
__global__ void __launch_bounds__(64) k_sum_rows(const td src, const td dst) {
    const int64_t row = blockIdx.x;
    const int64_t i1 = row % src.ne1, i2 = (row / src.ne1) % src.ne2, i3 = row / (src.ne1*src.ne2);
    const char * x = src.data + i1*src.nb1 + i2*src.nb2 + i3*src.nb3;
    float s = 0.0f;
    for (int64_t i = threadIdx.x; i < src.ne0; i += 64) s += *(const float *) (x + i*src.nb0);
    s = wave_sum(s);
    if (threadIdx.x == 0) *(float *) (dst.data + i1*dst.nb1 + i2*dst.nb2 + i3*dst.nb3) = s;
}
void sum_rows(const tensor_desc & src, const tensor_desc & dst, hipStream_t stream) {
    const int64_t nrows = src.ne[1]*src.ne[2]*src.ne[3];
    if (nrows == 0) return;
    hipLaunchKernelGGL(k_sum_rows, dim3((unsigned) nrows), dim3(64), 0, stream, mk(src), mk(dst));
}

// ---- ARGSORT (top_k = view of it) — src/llama-graph.cpp:883; tests/test-backend-ops.cpp:4120 -------------
// bitonic sort of one row in LDS, ne0 <= 1024 (n_expert <= 128 on the path)
__global__ void __launch_bounds__(1024) k_argsort(const td src, const td dst, int order, int npad) {
    extern __shared__ int idxs[];
    const int64_t row = blockIdx.x;
    const int64_t i1 = row % src.ne1, i2 = (row / src.ne1) % src.ne2, i3 = row / (src.ne1*src.ne2);
    const char * x = src.data + i1*src.nb1 + i2*src.nb2 + i3*src.nb3;
    const int col = threadIdx.x;
    const int ncols = (int) src.ne0;
    if (col < npad) idxs[col] = col;
    __syncthreads();
    for (int k = 2; k <= npad; k *= 2) {
        for (int j = k/2; j > 0; j /= 2) {
            const int ixj = col ^ j;
            if (col < npad && ixj > col) {
                const int a = idxs[col], b = idxs[ixj];
                bool swap;
                if ((col & k) == 0) {
                    swap = a >= ncols || (b < ncols && (order == 0 ? *(const float *) (x + a*src.nb0) > *(const float *) (x + b*src.nb0)
                                                                      : *(const float *) (x + a*src.nb0) < *(const float *) (x + b*src.nb0)));
                } else {
                    swap = b >= ncols || (a < ncols && (order == 0 ? *(const float *) (x + a*src.nb0) < *(const float *) (x + b*src.nb0)
                                                                      : *(const float *) (x + a*src.nb0) > *(const float *) (x + b*src.nb0)));
                }
                if (swap) { idxs[col] = b; idxs[ixj] = a; }
            }
            __syncthreads();
        }
    }
    if (col < ncols) *(int32_t *) (dst.data + col*dst.nb0 + i1*dst.nb1 + i2*dst.nb2 + i3*dst.nb3) = idxs[col];
}
void argsort(const tensor_desc & src, const tensor_desc & dst, int order, hipStream_t stream) {
    const int64_t nrows = src.ne[1]*src.ne[2]*src.ne[3];
    if (nrows == 0) return;
    int npad = 1; while (npad < src.ne[0]) npad *= 2;
    const int threads = npad < 64 ? 64 : npad;
    hipLaunchKernelGGL(k_argsort, dim3((unsigned) nrows), dim3(threads), npad*sizeof(int), stream, mk(src), mk(dst), order, npad);
}

// ---- UNARY — used by routers (sigmoid) and tests ----------------------------------------------------------
enum { U_ABS, U_SGN, U_NEG, U_STEP, U_TANH, U_ELU, U_RELU, U_SIGMOID, U_GELU, U_GELU_QUICK, U_SILU, U_HARDSWISH, U_HARDSIGMOID, U_EXP, U_GELU_ERF };
static __device__ __forceinline__ float gelu_f(float x) {
    const float c = 0.044715f, s = 0.79788456080286535587989211986876f;
    return 0.5f*x*(1.0f + tanhf(s*x*(1.0f + c*x*x)));
}
static __device__ __forceinline__ float silu_f(float x) { return x/(1.0f + expf(-x)); }
__global__ void __launch_bounds__(256) k_unary(int op, const td src, const td dst, int64_t n) {
    const int64_t i = (int64_t) blockIdx.x*256 + threadIdx.x;
    if (i >= n) return;
    const idx4 ix = unravel(i, dst.ne0, dst.ne1, dst.ne2); const int64_t i0 = ix.i0, i1 = ix.i1, i2 = ix.i2, i3 = ix.i3;
    const float x = ld_elem(src.data + i0*src.nb0 + i1*src.nb1 + i2*src.nb2 + i3*src.nb3, src.type);
    float r;
    switch (op) {
        case U_ABS: r = fabsf(x); break;
        case U_SGN: r = x > 0.f ? 1.f : (x < 0.f ? -1.f : 0.f); break;
        case U_NEG: r = -x; break;
        case U_STEP: r = x > 0.f ? 1.f : 0.f; break;
        case U_TANH: r = tanhf(x); break;
        case U_ELU: r = x > 0.f ? x : expm1f(x); break;
        case U_RELU: r = fmaxf(x, 0.f); break;
        case U_SIGMOID: r = 1.0f/(1.0f + expf(-x)); break;
        case U_GELU: r = gelu_f(x); break;
        case U_GELU_QUICK: r = x*(1.0f/(1.0f + expf(-1.702f*x))); break;
        case U_SILU: r = silu_f(x); break;
        case U_HARDSWISH: r = x*fminf(1.0f, fmaxf(0.0f, (x + 3.0f)/6.0f)); break;
        case U_HARDSIGMOID: r = fminf(1.0f, fmaxf(0.0f, (x + 3.0f)/6.0f)); break;
        case U_EXP: r = expf(x); break;
        case U_GELU_ERF: r = 0.5f*x*(1.0f + erff(x*0.70710678118654752440f)); break;
        default: r = x; break;
    }
    st_elem(dst.data + i0*dst.nb0 + i1*dst.nb1 + i2*dst.nb2 + i3*dst.nb3, dst.type, r);
}
void unary(int op, const tensor_desc & src, const tensor_desc & dst, hipStream_t stream) {
    const int64_t n = dst.ne[0]*dst.ne[1]*dst.ne[2]*dst.ne[3];
    if (n == 0) return;
    hipLaunchKernelGGL(k_unary, dim3((unsigned)((n + 255)/256)), dim3(256), 0, stream, op, mk(src), mk(dst), n);
}

// ---- GLU — src/llama-graph.cpp:691,947,961-968; tests/test-backend-ops.cpp:1832-1949 ---------------------
// split form: out = act(a) * b; single-tensor form (b == null): a holds [x | g] halves along ne0
enum { G_REGLU, G_GEGLU, G_SWIGLU, G_SWIGLU_OAI, G_GEGLU_ERF, G_GEGLU_QUICK };
__global__ void __launch_bounds__(256) k_glu(int op, const td a, const td b, const td dst, int64_t n, float alpha, float limit) {
    const int64_t i = (int64_t) blockIdx.x*256 + threadIdx.x;
    if (i >= n) return;
    const idx4 ix = unravel(i, dst.ne0, dst.ne1, dst.ne2); const int64_t i0 = ix.i0, i1 = ix.i1, i2 = ix.i2, i3 = ix.i3;
    const float x = ld_elem(a.data + i0*a.nb0 + i1*a.nb1 + i2*a.nb2 + i3*a.nb3, a.type);
    const float g = ld_elem(b.data + i0*b.nb0 + i1*b.nb1 + i2*b.nb2 + i3*b.nb3, b.type);
    float r;
    switch (op) {
        case G_REGLU: r = fmaxf(x, 0.f)*g; break;
        case G_GEGLU: r = gelu_f(x)*g; break;
        case G_SWIGLU: r = silu_f(x)*g; break;
        case G_SWIGLU_OAI: {
            const float xc = fminf(x, limit);
            const float gc = fmaxf(fminf(g, limit), -limit);
            r = (xc/(1.0f + expf(-xc*alpha)))*(gc + 1.0f);
        } break;
        case G_GEGLU_ERF: r = 0.5f*x*(1.0f + erff(x*0.70710678118654752440f))*g; break;
        case G_GEGLU_QUICK: r = x*(1.0f/(1.0f + expf(-1.702f*x)))*g; break;
        default: r = 0.f; break;
    }
    st_elem(dst.data + i0*dst.nb0 + i1*dst.nb1 + i2*dst.nb2 + i3*dst.nb3, dst.type, r);
}
void glu(int glu_op, bool swapped, const tensor_desc & a, const tensor_desc * b, const tensor_desc & dst, float alpha, float limit, hipStream_t stream) {
    const int64_t n = dst.ne[0]*dst.ne[1]*dst.ne[2]*dst.ne[3];
    if (n == 0) return;
    td ta = mk(a), tb;
    if (b) {
        tb = mk(*b);
        if (swapped) { td t = ta; ta = tb; tb = t; }
    } else {
        // halves of one tensor: x = first half, gate = second half (swapped: the other way round)
        tb = ta;
        const size_t half = (size_t)(a.ne[0]/2)*a.nb[0];
        if (swapped) ta.data += half; else tb.data += half;
    }
    hipLaunchKernelGGL(k_glu, dim3((unsigned)((n + 255)/256)), dim3(256), 0, stream, glu_op, ta, tb, mk(dst), n, alpha, limit);
}

// ---- ROPE — src/llama-model.cpp:6030-6040; tests/test-backend-ops.cpp:3660-3782 (max asymmetry 1e-3) ------
struct rope_corr { float lo, hi; };
static __device__ __forceinline__ void rope_yarn(float theta_extrap, float freq_scale, rope_corr cd, int64_t i0, float ext_factor, float mscale,
                                                  float & c, float & s) {
    float theta_interp = freq_scale*theta_extrap;
    float theta = theta_interp;
    if (ext_factor != 0.0f) {
        const float y = ((float)(i0/2) - cd.lo)/fmaxf(0.001f, cd.hi - cd.lo);
        const float ramp_mix = (1.0f - fminf(1.0f, fmaxf(0.0f, y)))*ext_factor;
        theta = theta_interp*(1.0f - ramp_mix) + theta_extrap*ramp_mix;
        mscale *= 1.0f + 0.1f*logf(1.0f/freq_scale);
    }
    c = cosf(theta)*mscale;
    s = sinf(theta)*mscale;
}

// one thread per rotated pair; src/dst [ne0 = head dim, ne1 = heads, ne2 = tokens, ne3]
template <bool NEOX>
__global__ void __launch_bounds__(256) k_rope(const td src, const int32_t * pos, const float * ff, const td dst, rope_params p, rope_corr cd,
                                              float theta_scale, int64_t npairs) {
    const int64_t i = (int64_t) blockIdx.x*256 + threadIdx.x;
    if (i >= npairs) return;
    const int64_t half = src.ne0/2;
    const idx4 ix = unravel(i, half, src.ne1, src.ne2); const int64_t ip = ix.i0, i1 = ix.i1, i2 = ix.i2, i3 = ix.i3;
    const char * x = src.data + i1*src.nb1 + i2*src.nb2 + i3*src.nb3;
    char * y = dst.data + i1*dst.nb1 + i2*dst.nb2 + i3*dst.nb3;
    const int64_t i0 = 2*ip;
    if (i0 >= p.n_dims) {   // pass-through for dims beyond n_dims
        st_elem(y + i0*dst.nb0, dst.type, ld_elem(x + i0*src.nb0, src.type));
        st_elem(y + (i0 + 1)*dst.nb0, dst.type, ld_elem(x + (i0 + 1)*src.nb0, src.type));
        return;
    }
    const float theta_base = (float) pos[i2]*powf(theta_scale, (float) ip);
    const float freq_factor = ff ? ff[ip] : 1.0f;
    float c, s;
    rope_yarn(theta_base/freq_factor, p.freq_scale, cd, i0, p.ext_factor, p.attn_factor, c, s);
    const int64_t ia = NEOX ? ip : i0, ib = NEOX ? ip + p.n_dims/2 : i0 + 1;
    const float x0 = ld_elem(x + ia*src.nb0, src.type), x1 = ld_elem(x + ib*src.nb0, src.type);
    st_elem(y + ia*dst.nb0, dst.type, x0*c - x1*s);
    st_elem(y + ib*dst.nb0, dst.type, x0*s + x1*c);
}

static float rope_corr_dim(int n_dims, int n_ctx_orig, float n_rot, float base) {
    return n_dims*logf(n_ctx_orig/(n_rot*2*(float) M_PI))/(2*logf(base));
}
void rope(const tensor_desc & src, const int32_t * pos, const float * freq_factors, const tensor_desc & dst, const rope_params & p, hipStream_t stream) {
    const int64_t npairs = (src.ne[0]/2)*src.ne[1]*src.ne[2]*src.ne[3];
    if (npairs == 0) return;
    const float theta_scale = powf(p.freq_base, -2.0f/p.n_dims);
    rope_corr cd;
    const float start = floorf(rope_corr_dim(p.n_dims, p.n_ctx_orig, p.beta_fast, p.freq_base));
    const float end   = ceilf (rope_corr_dim(p.n_dims, p.n_ctx_orig, p.beta_slow, p.freq_base));
    cd.lo = fmaxf(0.0f, start);
    cd.hi = fminf((float)(p.n_dims - 1), end);
    const dim3 g((unsigned)((npairs + 255)/256));
    if (p.mode & 2) hipLaunchKernelGGL((k_rope<true>),  g, dim3(256), 0, stream, mk(src), pos, freq_factors, mk(dst), p, cd, theta_scale, npairs);
    else            hipLaunchKernelGGL((k_rope<false>), g, dim3(256), 0, stream, mk(src), pos, freq_factors, mk(dst), p, cd, theta_scale, npairs);
}

// ---- SOFT_MAX ext — src/llama-graph.cpp:1312-1313; tests/test-backend-ops.cpp:3569-3626 (NMSE 1e-6) -------
// y = softmax(x*scale + slope*mask) along ne0, optional per-head sink logit (in max and denominator only)
__global__ void __launch_bounds__(256) k_soft_max(const td src, const td mask, const float * sinks, const td dst, float scale, float max_bias,
                                                   float m0, float m1, int n_head_log2, bool has_mask) {
    __shared__ float sh[4];
    const int64_t row = blockIdx.x;
    const idx4 rx = unravel(row, src.ne1, src.ne2, 1ll << 30); const int64_t i1 = rx.i0, i2 = rx.i1, i3 = rx.i2;
    const char * x = src.data + i1*src.nb1 + i2*src.nb2 + i3*src.nb3;
    char * y = dst.data + i1*dst.nb1 + i2*dst.nb2 + i3*dst.nb3;
    const char * mp = has_mask ? mask.data + i1*mask.nb1 + wrap(i2, mask.ne2)*mask.nb2 + wrap(i3, mask.ne3)*mask.nb3 : nullptr;
    float slope = 1.0f;
    if (max_bias > 0.0f) {
        const int h = (int) i2;
        slope = h < n_head_log2 ? powf(m0, (float)(h + 1)) : powf(m1, (float)(2*(h - n_head_log2) + 1));
    }
    float mx = sinks ? sinks[i2] : -INFINITY;
    for (int64_t i = threadIdx.x; i < src.ne0; i += blockDim.x) {
        float v = *(const float *) (x + i*4)*scale;
        if (has_mask) v += slope*ld_elem(mp + i*mask.nb0, mask.type);
        mx = fmaxf(mx, v);
    }
    mx = block_max(mx, sh);
    float sum = 0.0f;
    for (int64_t i = threadIdx.x; i < src.ne0; i += blockDim.x) {
        float v = *(const float *) (x + i*4)*scale;
        if (has_mask) v += slope*ld_elem(mp + i*mask.nb0, mask.type);
        const float e = expf(v - mx);
        sum += e;
        *(float *) (y + i*4) = e;
    }
    sum = block_sum(sum, sh);
    if (sinks) sum += expf(sinks[i2] - mx);
    const float inv = 1.0f/sum;
    for (int64_t i = threadIdx.x; i < src.ne0; i += blockDim.x) *(float *) (y + i*4) *= inv;
}
void soft_max(const tensor_desc & src, const tensor_desc * mask, const float * sinks, const tensor_desc & dst,
              float scale, float max_bias, hipStream_t stream) {
    const int64_t nrows = src.ne[1]*src.ne[2]*src.ne[3];
    if (nrows == 0) return;
    const int n_head = (int) src.ne[2];
    const int n_head_log2 = 1u << (uint32_t) floorf(log2f((float) n_head));
    const float m0 = powf(2.0f, -(max_bias)/n_head_log2), m1 = powf(2.0f, -(max_bias/2.0f)/n_head_log2);
    hipLaunchKernelGGL(k_soft_max, dim3((unsigned) nrows), dim3(rows_block(src.ne[0])), 0, stream,
                       mk(src), mask ? mk(*mask) : td{}, sinks, mk(dst), scale, max_bias, m0, m1, n_head_log2, mask != nullptr);
}

// ---- MoE router, one token (decode): logits = W_r . x (+ bias) -> [soft_max] -> argsort descending, one workgroup
// (src/llama-graph.cpp:838-883: build_lora_mm(gate_inp), ggml_add(gate_inp_b), ggml_soft_max, ggml_top_k = argsort + view)
struct moe_route_args { const float * w; size_t w_nb1; const float * x; const float * bias; int k, n_expert, softmax; float * logits; float * probs; int32_t * sorted;
                        // norm_w != NULL (the wide kernel only): x is the RAW residual stream and the router's input is y = (x * rsqrt(mean(x^2) + eps)) * norm_w
                        // (build_norm's RMS_NORM -> MUL folded in); workgroup 0 also writes y to y_out, where the expert mat-vecs read it
                        const float * norm_w; float eps; float * y_out;
                        unsigned * err;         // host-mapped error words (may be NULL): [1] = the ranking workgroup gave up waiting for a logit
                        float * topv; };        // may be NULL: the eight best values (probabilities, or logits without soft_max) in rank order — what GET_ROWS(probs, selected) gathers, handed
                                                // to the combine directly so that its reader does not chase ids -> probs through two cold loads
__global__ void __launch_bounds__(1024) k_moe_route(const moe_route_args p) {
    __shared__ float v[256];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;      // 16 waves: the rows of 32 experts are 2 per wave
    for (int e = wave; e < p.n_expert; e += 16) {
        const char * row = (const char *) p.w + (size_t) e*p.w_nb1;
        float acc = 0.0f, acc2 = 0.0f;
        int i = lane*4;
        for (; i + 256 < p.k; i += 512) {       // two independent 16-byte loads per operand in flight
            const float4v a = *(const float4v *) (row + (size_t) i*4), b = *(const float4v *) (p.x + i);
            const float4v a2 = *(const float4v *) (row + (size_t)(i + 256)*4), b2 = *(const float4v *) (p.x + i + 256);
            acc  += (a.x*b.x + a.y*b.y) + (a.z*b.z + a.w*b.w);
            acc2 += (a2.x*b2.x + a2.y*b2.y) + (a2.z*b2.z + a2.w*b2.w);
        }
        if (i < p.k) {
            const float4v a = *(const float4v *) (row + (size_t) i*4), b = *(const float4v *) (p.x + i);
            acc += (a.x*b.x + a.y*b.y) + (a.z*b.z + a.w*b.w);
        }
        acc += acc2;
        acc = wave_sum(acc);
        if (lane == 0) { if (p.bias) acc += p.bias[e]; v[e] = acc; if (p.logits) p.logits[e] = acc; }
    }
    __syncthreads();
    const int e = threadIdx.x;
    if (p.softmax) {
        float mx = -INFINITY, sum = 0.0f;
        for (int j = 0; j < p.n_expert; j++) mx = fmaxf(mx, v[j]);
        for (int j = 0; j < p.n_expert; j++) sum += expf(v[j] - mx);
        const float pe = e < p.n_expert ? expf(v[e] - mx)*(1.0f/sum) : 0.0f;
        __syncthreads();
        if (e < p.n_expert) { v[e] = pe; p.probs[e] = pe; }
        __syncthreads();
    }
    if (e < p.n_expert) {      // rank = how many values sort before this one (descending, index breaks ties)
        int rank = 0;
        const float me = v[e];
        for (int j = 0; j < p.n_expert; j++) rank += (v[j] > me) || (v[j] == me && j < e);
        p.sorted[rank] = e;
        if (p.topv && rank < 8) p.topv[rank] = me;
    }
}
// One workgroup streams the whole router matrix at ONE CU's rate (gpt-oss: 32 rows of 2880 floats = 368 KB: 13 us; Mixtral: 8 x 4096: 8 us). Here one
// expert row per wave, MR_EPW rows per workgroup; every workgroup publishes its logits (release fence + counter), and the one that arrives last (acquire) does the soft_max / ranking
// and re-arms the counter. ws = [256 floats of logits | 1 int counter], zero-initialised once.
constexpr int MR_EPW = 2;
__global__ void __launch_bounds__(256) k_moe_route_wide(const moe_route_args p, float * ws) {
    __shared__ float v[256];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int e0 = blockIdx.x*MR_EPW + wave;
    // ws: [0, 256) unused floats of the ticket version | word 257: launch counter | from byte 2048: 256 hand-off granules of 8 bytes
    unsigned * epoch = (unsigned *) ws + 257;
    unsigned long long * gran = (unsigned long long *) ((char *) ws + 2048);
    const unsigned tag = __hip_atomic_load(epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1u;
    // the expert's row of router weights (the HBM part of this kernel) is requested first, whole (k <= 4096: 16 x 16 bytes per lane), before the
    // norm below and before any x is needed: the dot loop used to wait for two loads per trip, 6-8 dependent round trips (4.5 us of a 9 us kernel)
    constexpr int RT = 8;
    const bool pre = p.k <= RT*512;
    const bool mine = wave < MR_EPW && e0 < p.n_expert;
    float4v ra[RT], rb[RT];
    if (pre && mine) {
        const char * row = (const char *) p.w + (size_t) e0*p.w_nb1;
#pragma unroll
        for (int t = 0; t < RT; t++) {
            ra[t] = *(const float4v *) (row + (size_t) min(lane*4 + 512*t, p.k - 4)*4);
            rb[t] = *(const float4v *) (row + (size_t) min(lane*4 + 512*t + 256, p.k - 4)*4);
        }
    }
    float scale = 1.0f;
    // k <= 4096 with the norm folded in: a wave's 64 lanes cover the whole vector at exactly the positions its dot needs (lane*4 + 512t and + 256), so the wave
    // takes x ONCE, sums the squares over its own lanes (no LDS, no barrier, no second trip to L2 for the dot) and normalises what it holds
    const bool wave_norm = p.norm_w && pre;
    float4v xa[RT], xb[RT];
    if (wave_norm) {
        float ss = 0.0f;
#pragma unroll
        for (int t = 0; t < RT; t++) { xa[t] = *(const float4v *) (p.x + min(lane*4 + 512*t, p.k - 4)); xb[t] = *(const float4v *) (p.x + min(lane*4 + 512*t + 256, p.k - 4)); }
        float4v na[RT], nb[RT];
#pragma unroll
        for (int t = 0; t < RT; t++) { na[t] = *(const float4v *) (p.norm_w + min(lane*4 + 512*t, p.k - 4)); nb[t] = *(const float4v *) (p.norm_w + min(lane*4 + 512*t + 256, p.k - 4)); }
#pragma unroll
        for (int t = 0; t < RT; t++) {
            const int i = lane*4 + 512*t;
            if (i < p.k)       ss += (xa[t].x*xa[t].x + xa[t].y*xa[t].y) + (xa[t].z*xa[t].z + xa[t].w*xa[t].w);
            if (i + 256 < p.k) ss += (xb[t].x*xb[t].x + xb[t].y*xb[t].y) + (xb[t].z*xb[t].z + xb[t].w*xb[t].w);
        }
        ss = wave_sum(ss);
        scale = 1.0f/sqrtf(ss/(float) p.k + p.eps);
#pragma unroll
        for (int t = 0; t < RT; t++) {
            xa[t] = float4v{ (xa[t].x*scale)*na[t].x, (xa[t].y*scale)*na[t].y, (xa[t].z*scale)*na[t].z, (xa[t].w*scale)*na[t].w };
            xb[t] = float4v{ (xb[t].x*scale)*nb[t].x, (xb[t].y*scale)*nb[t].y, (xb[t].z*scale)*nb[t].z, (xb[t].w*scale)*nb[t].w };
        }
        if (blockIdx.x == 0 && wave == 0 && p.y_out) {
#pragma unroll
            for (int t = 0; t < RT; t++) {
                const int i = lane*4 + 512*t;
                if (i < p.k)       *(float4v *) (p.y_out + i) = xa[t];
                if (i + 256 < p.k) *(float4v *) (p.y_out + i + 256) = xb[t];
            }
        }
    } else
    if (p.norm_w) {        // every workgroup normalises the whole vector for itself (k floats from L2: 11-16 KB)
        __shared__ float ssw[4];
        float ss = 0.0f;
        for (int i0 = threadIdx.x*4; i0 < p.k; i0 += 4*1024) {      // four loads in flight per thread (k = 4096: one trip)
            float4v a[4];
#pragma unroll
            for (int u = 0; u < 4; u++) a[u] = *(const float4v *) (p.x + min(i0 + u*1024, p.k - 4));
#pragma unroll
            for (int u = 0; u < 4; u++) if (i0 + u*1024 < p.k) ss += (a[u].x*a[u].x + a[u].y*a[u].y) + (a[u].z*a[u].z + a[u].w*a[u].w);
        }
        ss = wave_sum(ss);
        if (lane == 0) ssw[wave] = ss;
        __syncthreads();
        ss = (ssw[0] + ssw[1]) + (ssw[2] + ssw[3]);
        scale = 1.0f/sqrtf(ss/(float) p.k + p.eps);
        if (blockIdx.x == 0 && p.y_out) {
            for (int i = threadIdx.x*4; i < p.k; i += 1024) {
                const float4v a = *(const float4v *) (p.x + i), nw = *(const float4v *) (p.norm_w + i);
                *(float4v *) (p.y_out + i) = float4v{ (a.x*scale)*nw.x, (a.y*scale)*nw.y, (a.z*scale)*nw.z, (a.w*scale)*nw.w };
            }
        }
    }
    auto xin = [&](int i) -> float4v {
        const float4v a = *(const float4v *) (p.x + i);
        if (!p.norm_w) return a;
        const float4v nw = *(const float4v *) (p.norm_w + i);
        return float4v{ (a.x*scale)*nw.x, (a.y*scale)*nw.y, (a.z*scale)*nw.z, (a.w*scale)*nw.w };
    };
    if (mine) {
        const char * row = (const char *) p.w + (size_t) e0*p.w_nb1;
        float acc = 0.0f, acc2 = 0.0f;
        if (pre) {
            // the same terms in the same order as the loop below: acc takes i = lane*4 + 512t (< k), acc2 takes i + 256 (< k)
            if (!wave_norm) {
#pragma unroll
                for (int t = 0; t < RT; t++) { xa[t] = xin(min(lane*4 + 512*t, p.k - 4)); xb[t] = xin(min(lane*4 + 512*t + 256, p.k - 4)); }
            }
#pragma unroll
            for (int t = 0; t < RT; t++) {
                const int i = lane*4 + 512*t;
                if (i < p.k)       acc  += (ra[t].x*xa[t].x + ra[t].y*xa[t].y) + (ra[t].z*xa[t].z + ra[t].w*xa[t].w);
                if (i + 256 < p.k) acc2 += (rb[t].x*xb[t].x + rb[t].y*xb[t].y) + (rb[t].z*xb[t].z + rb[t].w*xb[t].w);
            }
        } else {
        int i = lane*4;
        for (; i + 256 < p.k; i += 512) {
            const float4v a = *(const float4v *) (row + (size_t) i*4), b = xin(i);
            const float4v a2 = *(const float4v *) (row + (size_t)(i + 256)*4), b2 = xin(i + 256);
            acc  += (a.x*b.x + a.y*b.y) + (a.z*b.z + a.w*b.w);
            acc2 += (a2.x*b2.x + a2.y*b2.y) + (a2.z*b2.z + a2.w*b2.w);
        }
        if (i < p.k) {
            const float4v a = *(const float4v *) (row + (size_t) i*4), b = xin(i);
            acc += (a.x*b.x + a.y*b.y) + (a.z*b.z + a.w*b.w);
        }
        }
        acc += acc2;                        // the same summation order as k_moe_route: identical logits
        acc = wave_sum(acc);
        // The hand-off to the ranking workgroup: the logit travels WITH its flag — one 8-byte agent-scope store {logit, launch tag} — and workgroup 0 polls
        // the granules ("the data is the flag"). Before: write-through store, drain it, take a ticket, the last arriver reloads = three dependent trips to
        // L2 (~4 us of a 7 us kernel); now the ranking starts one poll after the slowest logit lands. The tag is the launch counter kept in ws
        // (workgroup 0 bumps it when it has consumed every granule; launches of one stream do not overlap), so a stale granule never matches.
        if (lane == 0) {
            if (p.bias) acc += p.bias[e0];
            const unsigned long long gv = ((unsigned long long) tag << 32) | (unsigned long long) __builtin_bit_cast(unsigned, acc);
            __hip_atomic_store(gran + e0, gv, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (p.logits) p.logits[e0] = acc;
        }
    }
    if (blockIdx.x != 0) return;
    const int e = threadIdx.x;
    if (e < p.n_expert) {
        unsigned long long gv = 0; int spins = 0;
        do { gv = __hip_atomic_load(gran + e, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); } while ((unsigned)(gv >> 32) != tag && ++spins < (1 << 22));   // bounded: a lost
        v[e] = __builtin_bit_cast(float, (unsigned) gv);                                                // workgroup must not hang the device ...
        if ((unsigned)(gv >> 32) != tag && p.err) __hip_atomic_store(p.err + 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);   // ... and a stale logit must not be ranked silently: the host aborts at its next synchronize
    }
    __syncthreads();
    if (threadIdx.x == 0) __hip_atomic_store(epoch, tag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (p.softmax) {
        float mx = -INFINITY, sum = 0.0f;
        for (int j = 0; j < p.n_expert; j++) mx = fmaxf(mx, v[j]);
        for (int j = 0; j < p.n_expert; j++) sum += expf(v[j] - mx);
        const float pe = e < p.n_expert ? expf(v[e] - mx)*(1.0f/sum) : 0.0f;
        __syncthreads();
        if (e < p.n_expert) { v[e] = pe; p.probs[e] = pe; }
        __syncthreads();
    }
    if (e < p.n_expert) {      // rank = how many values sort before this one (descending, index breaks ties)
        int rank = 0;
        const float me = v[e];
        for (int j = 0; j < p.n_expert; j++) rank += (v[j] > me) || (v[j] == me && j < e);
        p.sorted[rank] = e;
        if (p.topv && rank < 8) p.topv[rank] = me;
    }
}
bool moe_route_norm_supported(int64_t k, int64_t n_expert, const float * ws) {
    static const bool wide_on = !getenv("GGML_MI355X_MOE_ROUTE_WIDE") || atoi(getenv("GGML_MI355X_MOE_ROUTE_WIDE")) != 0;
    return ws && wide_on && n_expert >= 4 && n_expert <= 256 && k % 4 == 0;
}
void moe_route(const float * w, size_t w_nb1, const float * x, const float * bias, int64_t k, int64_t n_expert, bool softmax,
               float * logits, float * probs, int32_t * sorted, hipStream_t stream, float * ws, const float * norm_w, float eps, float * y_out, unsigned * err, float * topv) {
    moe_route_args a = { w, w_nb1, x, bias, (int) k, (int) n_expert, softmax ? 1 : 0, logits, probs, sorted, norm_w, eps, y_out, err, topv };
    static const bool wide_on = !getenv("GGML_MI355X_MOE_ROUTE_WIDE") || atoi(getenv("GGML_MI355X_MOE_ROUTE_WIDE")) != 0;
    if (norm_w && !moe_route_norm_supported(k, n_expert, ws)) { fprintf(stderr, "moe_route: the norm is folded into the multi-workgroup kernel only\n"); abort(); }
    if (ws && wide_on && n_expert >= 4 && n_expert <= 256) hipLaunchKernelGGL(k_moe_route_wide, dim3((unsigned)((n_expert + MR_EPW - 1)/MR_EPW)), dim3(256), 0, stream, a, ws);
    else hipLaunchKernelGGL(k_moe_route, dim3(1), dim3(1024), 0, stream, a);
}

// ---- MoE combine, one token (decode): the tail of build_moe_ffn as ONE kernel (src/llama-graph.cpp:887-1012) ----
//   weights = get_rows(probs, selected) -> (sum_rows, div | soft_max) ; experts * weights ; sum over the used experts ; + residual
// mode 0: w_u = p_u / sum(p) (norm_w, llm_build_llama's MoE branch); mode 1: w = soft_max(selected logits) (SOFTMAX_WEIGHT, gpt-oss)
struct moe_combine_args { const float * probs; const int32_t * ids; int n_used, mode; const char * experts; size_t e_nb1; int n_embd; const float * res; float * dst; };
__global__ void __launch_bounds__(256) k_moe_combine(const moe_combine_args p) {
    float w[8];
    float pv[8];
#pragma unroll
    for (int u = 0; u < 8; u++) pv[u] = u < p.n_used ? (p.ids ? p.probs[p.ids[u]] : p.probs[u]) : 0.0f;      // ids == NULL: probs holds the selected values in slot order (the router's topv)
    if (p.mode == 0) {
        float sum = 0.0f;
#pragma unroll
        for (int u = 0; u < 8; u++) if (u < p.n_used) sum += pv[u];
#pragma unroll
        for (int u = 0; u < 8; u++) w[u] = pv[u]/sum;
    } else {
        float mx = -INFINITY;
#pragma unroll
        for (int u = 0; u < 8; u++) if (u < p.n_used) mx = fmaxf(mx, pv[u]);
        float sum = 0.0f;
#pragma unroll
        for (int u = 0; u < 8; u++) { w[u] = u < p.n_used ? expf(pv[u] - mx) : 0.0f; sum += w[u]; }
        const float inv = 1.0f/sum;
#pragma unroll
        for (int u = 0; u < 8; u++) w[u] *= inv;
    }
    const int i = (blockIdx.x*256 + threadIdx.x)*4;
    if (i >= p.n_embd) return;
    float4v acc = *(const float4v *) (p.experts + (size_t) i*4);
    acc.x *= w[0]; acc.y *= w[0]; acc.z *= w[0]; acc.w *= w[0];
#pragma unroll
    for (int u = 1; u < 8; u++) {
        if (u < p.n_used) {
            const float4v e = *(const float4v *) (p.experts + (size_t) u*p.e_nb1 + (size_t) i*4);
            acc.x += e.x*w[u]; acc.y += e.y*w[u]; acc.z += e.z*w[u]; acc.w += e.w*w[u];
        }
    }
    if (p.res) { const float4v r = *(const float4v *) (p.res + i); acc.x += r.x; acc.y += r.y; acc.z += r.z; acc.w += r.w; }
    *(float4v *) (p.dst + i) = acc;
}
void moe_combine(const float * probs, const int32_t * ids, int n_used, int mode, const void * experts, size_t e_nb1, int64_t n_embd,
                 const float * res, float * dst, hipStream_t stream) {
    moe_combine_args a = { probs, ids, n_used, mode, (const char *) experts, e_nb1, (int) n_embd, res, dst };
    hipLaunchKernelGGL(k_moe_combine, dim3((unsigned)((n_embd/4 + 255)/256)), dim3(256), 0, stream, a);
}

// ---- small host -> device uploads of one graph's inputs (positions, mask, cache indices, one embedding row ...) as ONE launch: the sources sit in a
// pinned, device-mapped staging area (backend.cpp: be_set_tensor_async), every item is copied by its own workgroups straight over PCIe ----
__global__ void __launch_bounds__(256) k_upload_batch(const upload_batch b) {
    int it = 0, blk = (int) blockIdx.x;
#pragma unroll
    for (int i = 0; i < UPLOAD_BATCH_MAX; i++) if (i < b.n && it == i && blk >= b.blocks[i]) { blk -= b.blocks[i]; it = i + 1; }
    if (it >= b.n) return;
    const char * src = (const char *) b.src[it]; char * dst = (char *) b.dst[it];
    const uint32_t bytes = b.bytes[it];
    const uint32_t o = (uint32_t) blk*4096u + threadIdx.x*16u;
    if (o >= bytes) return;
    if ((((uintptr_t) src | (uintptr_t) dst) & 15) == 0 && o + 16 <= bytes) { *(int4v *) (dst + o) = *(const int4v *) (src + o); return; }
    for (uint32_t j = o; j < o + 16 && j < bytes; j++) dst[j] = src[j];
}
void upload_batch_launch(const upload_batch & b, hipStream_t stream) {
    int blocks = 0;
    for (int i = 0; i < b.n; i++) blocks += b.blocks[i];
    if (blocks > 0) hipLaunchKernelGGL(k_upload_batch, dim3((unsigned) blocks), dim3(256), 0, stream, b);
}

// ---- HBM probe ----------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) k_hbm_read(const int4v * p, size_t n16, unsigned * sink) {
    int acc = 0;
    const size_t stride = (size_t) gridDim.x*256;
    for (size_t i = (size_t) blockIdx.x*256 + threadIdx.x; i < n16; i += stride) {
        const int4v v = __builtin_nontemporal_load(p + i);
        acc ^= v.x ^ v.y ^ v.z ^ v.w;
    }
    if (acc == 0x12345678) *sink = acc;   // never true in practice; keeps the loads alive
}
// the same bytes with the Q4_K mat-vec's load shape and no arithmetic: 8 lanes per 144-byte block, each loading the block's 16-byte
// header (the same 16 bytes for the 8 lanes) and its own 16 bytes of nibbles; 2 rows x 2 k-steps in flight per wave; rows of 2304 bytes
// (k = 4096) walked with a grid stride — what the streaming phase of the decode kernels can reach at most
__global__ void __launch_bounds__(512) k_hbm_read_q4k(const char * p, int n_pairs, unsigned * sink) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, slot = lane & 7, ibl = lane >> 3;
    const int stride = gridDim.x*8;
    int acc = 0;
    for (int pr = blockIdx.x*8 + wave; pr < n_pairs; pr += stride) {
        int4v h[2][2], q[2][2];
#pragma unroll
        for (int it = 0; it < 2; it++)
#pragma unroll
            for (int r = 0; r < 2; r++) {
                const char * b = p + (size_t)(pr*2 + r)*2304 + (size_t)(it*8 + ibl)*144;
                h[it][r] = *(const int4v *) b;
                q[it][r] = *(const int4v *) (b + 16 + 16*slot);
            }
#pragma unroll
        for (int it = 0; it < 2; it++)
#pragma unroll
            for (int r = 0; r < 2; r++) acc ^= h[it][r].x ^ h[it][r].w ^ q[it][r].x ^ q[it][r].y ^ q[it][r].z ^ q[it][r].w;
    }
    if (acc == 0x12345678) *sink = acc;
}
void hbm_read_probe(const void * p, size_t bytes, unsigned * sink, hipStream_t stream) {
    static int mode = -1;
    if (mode < 0) { const char * e = getenv("GGML_MI355X_PROBE_MODE"); mode = e ? atoi(e) : 0; }
    if (mode == 1) hipLaunchKernelGGL(k_hbm_read_q4k, dim3(256), dim3(512), 0, stream, (const char *) p, (int)(bytes/4608), sink);
    else           hipLaunchKernelGGL(k_hbm_read, dim3(256*8), dim3(256), 0, stream, (const int4v *) p, bytes/16, sink);
}

// ---- many tokens: the expert-slot sum of build_moe_ffn (src/llama-graph.cpp:996-1012): ((slot0 + slot1) + slot2) + ... [+ residual], in ggml's order ----
struct slot_sum_args { const char * ex; size_t nb1, nb2; int n_used; int64_t m4, n_tokens; const char * res; size_t res_nb1; char * dst; size_t dst_nb1; };
__global__ void __launch_bounds__(256) k_slot_sum(const slot_sum_args p) {
    const int64_t i = (int64_t) blockIdx.x*256 + threadIdx.x;
    if (i >= p.m4*p.n_tokens) return;
    const int64_t t = i/p.m4, c4 = i - t*p.m4;
    const char * e = p.ex + (size_t) t*p.nb2 + (size_t) c4*16;
    float4v a = *(const float4v *) e;
    for (int u = 1; u < p.n_used; u++) { const float4v b = *(const float4v *) (e + (size_t) u*p.nb1); a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w; }
    if (p.res) { const float4v b = *(const float4v *) (p.res + (size_t) t*p.res_nb1 + (size_t) c4*16); a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w; }
    *(float4v *) (p.dst + (size_t) t*p.dst_nb1 + (size_t) c4*16) = a;
}
void moe_slot_sum(const void * experts, size_t nb1, size_t nb2, int n_used, int64_t m, int64_t n_tokens, const float * res, size_t res_nb1, float * dst, size_t dst_nb1, hipStream_t stream) {
    if (m == 0 || n_tokens == 0) return;
    const slot_sum_args p = { (const char *) experts, nb1, nb2, n_used, m/4, n_tokens, (const char *) res, res_nb1, (char *) dst, dst_nb1 };
    hipLaunchKernelGGL(k_slot_sum, dim3((unsigned)((m/4*n_tokens + 255)/256)), dim3(256), 0, stream, p);
}

} // namespace mi355x
