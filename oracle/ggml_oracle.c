// oracle/ggml_oracle.c — CPU restatement of the reference's quantized mat-mul
// arithmetic. TEST INFRASTRUCTURE ONLY: only tests/, __graft_entry__.smoke() and
// bench.py's cpu_baseline leg may load this; the product (libggml-mi355x.so)
// never links or calls it.
//
// Pinning status:
//   * dequantize_row_{q4_0,q8_0,q4_K,q5_K,q6_K,mxfp4} and
//     quantize_row_{q4_0,q8_0,mxfp4}_ref: PINNED bit-exactly against the
//     reference's own Python definition gguf-py/gguf/quants.py (which
//     gguf-py/tests/test_quants.py:116-141 declares bit-exact to the C code),
//     through the golden vectors in tests/golden/ (generator:
//     tests/golden/make_golden.py, run in the build container only).
//   * quantize_row_q8_K and the integer vec_dot_*_q8_* routines: the C they
//     restate lives in the un-vendored submodule ggml (.gitmodules:1-3; anchor
//     scripts/sync-ggml.last:1 = b141fc226b68e4af383101c39da90b54ede98850), so
//     they follow ggml's published generic (scalar) algorithm from
//     [UPSTREAM-KNOWLEDGE] and are "parity unpinned" bit-wise; they are
//     tolerance-pinned by the reference's own gates
//     (tests/test-quantize-fns.cpp:17-23,82-99: |dot - ref|/n <= 0.02;
//      tests/test-backend-ops.cpp:3106-3108: MUL_MAT NMSE <= 5e-4).
//
// Build: see oracle/Makefile (-O2 -ffp-contract=off: fused multiply-adds would
// break bit-exactness with the numpy reference).
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define QK_K 256
#define K_SCALE_SIZE 12

enum { T_F32 = 0, T_F16 = 1, T_Q4_0 = 2, T_Q8_0 = 8, T_Q4_K = 12, T_Q5_K = 13, T_Q6_K = 14, T_Q8_K = 15, T_MXFP4 = 39 };

#pragma pack(push, 1)
// gguf-py/gguf/constants.py:2842 (32, 2+16); layout gguf-py/gguf/quants.py:241-251
typedef struct { uint16_t d; uint8_t qs[16]; } block_q4_0;
// constants.py:2846 (32, 2+32); quants.py:396-401
typedef struct { uint16_t d; int8_t qs[32]; } block_q8_0;
// constants.py:2850 (256, 2+2+128+12); quants.py:504-522
typedef struct { uint16_t d; uint16_t dmin; uint8_t scales[K_SCALE_SIZE]; uint8_t qs[QK_K/2]; } block_q4_K;
// constants.py:2851 (256, 2+2+128+32+12); quants.py:527-549
typedef struct { uint16_t d; uint16_t dmin; uint8_t scales[K_SCALE_SIZE]; uint8_t qh[QK_K/8]; uint8_t qs[QK_K/2]; } block_q5_K;
// constants.py:2852 (256, 2+128+64+16); quants.py:554-572 — d is LAST
typedef struct { uint8_t ql[QK_K/2]; uint8_t qh[QK_K/4]; int8_t scales[QK_K/16]; uint16_t d; } block_q6_K;
// constants.py:2853 (256, 4+256+32)
typedef struct { float d; int8_t qs[QK_K]; int16_t bsums[QK_K/16]; } block_q8_K;
// constants.py:2871 (32, 1+16); quants.py:656-700
typedef struct { uint8_t e; uint8_t qs[16]; } block_mxfp4;
#pragma pack(pop)

_Static_assert(sizeof(block_q4_0) == 18, "q4_0");
_Static_assert(sizeof(block_q8_0) == 34, "q8_0");
_Static_assert(sizeof(block_q4_K) == 144, "q4_K");
_Static_assert(sizeof(block_q5_K) == 176, "q5_K");
_Static_assert(sizeof(block_q6_K) == 210, "q6_K");
_Static_assert(sizeof(block_q8_K) == 292, "q8_K");
_Static_assert(sizeof(block_mxfp4) == 17, "mxfp4");

// gguf-py/gguf/quants.py:659
static const int8_t kvalues_mxfp4[16] = { 0, 1, 2, 3, 4, 6, 8, 12, 0, -1, -2, -3, -4, -6, -8, -12 };

// ---- fp16 -------------------------------------------------------------------
static inline float fp16_to_fp32(uint16_t h) {
    const uint32_t sign = (uint32_t)(h & 0x8000) << 16;
    const uint32_t exp = (h >> 10) & 0x1F, man = h & 0x3FF;
    uint32_t bits;
    if (exp == 0) {
        if (man == 0) bits = sign;
        else {
            int e = -1; uint32_t m = man;
            do { m <<= 1; e++; } while ((m & 0x400) == 0);
            bits = sign | ((uint32_t)(127 - 15 - e) << 23) | ((m & 0x3FF) << 13);
        }
    } else if (exp == 31) bits = sign | 0x7F800000u | (man << 13);
    else bits = sign | ((exp + 127 - 15) << 23) | (man << 13);
    float f; memcpy(&f, &bits, 4); return f;
}

static inline uint16_t fp32_to_fp16(float f) {
    uint32_t x; memcpy(&x, &f, 4);
    const uint32_t sign = (x >> 16) & 0x8000, ax = x & 0x7FFFFFFFu;
    if (ax >= 0x7F800000u) return (uint16_t)(sign | 0x7C00 | ((ax > 0x7F800000u) ? (0x200 | ((ax >> 13) & 0x3FF)) : 0));
    if (ax >= 0x477FF000u) return (uint16_t)(sign | 0x7C00);
    if (ax < 0x33000001u) return (uint16_t) sign;
    const int32_t e = (int32_t)(ax >> 23) - 127;
    const uint32_t m = (ax & 0x7FFFFF) | 0x800000;
    uint32_t shift, hexp;
    if (e < -14) { shift = (uint32_t)(13 + (-14 - e)); hexp = 0; } else { shift = 13; hexp = (uint32_t)(e + 15); }
    uint32_t hm = m >> shift;
    const uint32_t rem = m & ((1u << shift) - 1), half = 1u << (shift - 1);
    if (rem > half || (rem == half && (hm & 1))) hm++;
    const uint32_t h = hexp == 0 ? hm : ((hexp - 1) << 10) + hm;
    return (uint16_t)(sign | h);
}

// gguf-py/gguf/quants.py:663-665 (e8m0_to_fp32_half)
static inline float e8m0_to_fp32_half(uint8_t x) {
    const uint32_t bits = x < 2 ? (0x00200000u << x) : ((uint32_t)(x - 1) << 23);
    float f; memcpy(&f, &bits, 4); return f;
}

void orc_fp16_to_fp32_row(const uint16_t * x, float * y, int64_t n) { for (int64_t i = 0; i < n; i++) y[i] = fp16_to_fp32(x[i]); }
void orc_fp32_to_fp16_row(const float * x, uint16_t * y, int64_t n) { for (int64_t i = 0; i < n; i++) y[i] = fp32_to_fp16(x[i]); }

// ---- K-quant 6-bit scale/min unpack: gguf-py/gguf/quants.py:479-501 -----------
static inline void get_scale_min_k4(int j, const uint8_t * q, uint8_t * d, uint8_t * m) {
    if (j < 4) {
        *d = q[j] & 63; *m = q[j + 4] & 63;
    } else {
        *d = (q[j+4] & 0xF) | ((q[j-4] >> 6) << 4);
        *m = (q[j+4] >>  4) | ((q[j-0] >> 6) << 4);
    }
}

// ---- dequantize_row_* (PINNED by tests/golden/dequant_*.npz) -----------------
static void dequantize_row_q4_0(const block_q4_0 * x, float * y, int64_t k) {
    const int64_t nb = k / 32;
    for (int64_t i = 0; i < nb; i++) {
        const float d = fp16_to_fp32(x[i].d);
        for (int j = 0; j < 16; ++j) {
            const int x0 = (x[i].qs[j] & 0x0F) - 8;
            const int x1 = (x[i].qs[j] >>   4) - 8;
            y[i*32 + j +  0] = x0*d;
            y[i*32 + j + 16] = x1*d;
        }
    }
}

static void dequantize_row_q8_0(const block_q8_0 * x, float * y, int64_t k) {
    const int64_t nb = k / 32;
    for (int64_t i = 0; i < nb; i++) {
        const float d = fp16_to_fp32(x[i].d);
        for (int j = 0; j < 32; ++j) y[i*32 + j] = x[i].qs[j]*d;
    }
}

static void dequantize_row_q4_K(const block_q4_K * x, float * y, int64_t k) {
    const int64_t nb = k / QK_K;
    for (int64_t i = 0; i < nb; i++) {
        const uint8_t * q = x[i].qs;
        const float d = fp16_to_fp32(x[i].d), min = fp16_to_fp32(x[i].dmin);
        int is = 0; uint8_t sc, m;
        for (int j = 0; j < QK_K; j += 64) {
            get_scale_min_k4(is + 0, x[i].scales, &sc, &m);
            const float d1 = d * sc; const float m1 = min * m;
            get_scale_min_k4(is + 1, x[i].scales, &sc, &m);
            const float d2 = d * sc; const float m2 = min * m;
            for (int l = 0; l < 32; ++l) *y++ = d1 * (q[l] & 0xF) - m1;
            for (int l = 0; l < 32; ++l) *y++ = d2 * (q[l]  >> 4) - m2;
            q += 32; is += 2;
        }
    }
}

static void dequantize_row_q5_K(const block_q5_K * x, float * y, int64_t k) {
    const int64_t nb = k / QK_K;
    for (int64_t i = 0; i < nb; i++) {
        const uint8_t * ql = x[i].qs;
        const uint8_t * qh = x[i].qh;
        const float d = fp16_to_fp32(x[i].d), min = fp16_to_fp32(x[i].dmin);
        int is = 0; uint8_t sc, m; uint8_t u1 = 1, u2 = 2;
        for (int j = 0; j < QK_K; j += 64) {
            get_scale_min_k4(is + 0, x[i].scales, &sc, &m);
            const float d1 = d * sc; const float m1 = min * m;
            get_scale_min_k4(is + 1, x[i].scales, &sc, &m);
            const float d2 = d * sc; const float m2 = min * m;
            for (int l = 0; l < 32; ++l) *y++ = d1 * ((ql[l] & 0xF) + (qh[l] & u1 ? 16 : 0)) - m1;
            for (int l = 0; l < 32; ++l) *y++ = d2 * ((ql[l]  >> 4) + (qh[l] & u2 ? 16 : 0)) - m2;
            ql += 32; is += 2; u1 <<= 2; u2 <<= 2;
        }
    }
}

static void dequantize_row_q6_K(const block_q6_K * x, float * y, int64_t k) {
    const int64_t nb = k / QK_K;
    for (int64_t i = 0; i < nb; i++) {
        const float d = fp16_to_fp32(x[i].d);
        const uint8_t * ql = x[i].ql;
        const uint8_t * qh = x[i].qh;
        const int8_t  * sc = x[i].scales;
        for (int n = 0; n < QK_K; n += 128) {
            for (int l = 0; l < 32; ++l) {
                const int is = l/16;
                const int8_t q1 = (int8_t)((ql[l +  0] & 0xF) | (((qh[l] >> 0) & 3) << 4)) - 32;
                const int8_t q2 = (int8_t)((ql[l + 32] & 0xF) | (((qh[l] >> 2) & 3) << 4)) - 32;
                const int8_t q3 = (int8_t)((ql[l +  0]  >> 4) | (((qh[l] >> 4) & 3) << 4)) - 32;
                const int8_t q4 = (int8_t)((ql[l + 32]  >> 4) | (((qh[l] >> 6) & 3) << 4)) - 32;
                y[l +  0] = d * sc[is + 0] * q1;
                y[l + 32] = d * sc[is + 2] * q2;
                y[l + 64] = d * sc[is + 4] * q3;
                y[l + 96] = d * sc[is + 6] * q4;
            }
            y += 128; ql += 64; qh += 32; sc += 8;
        }
    }
}

static void dequantize_row_mxfp4(const block_mxfp4 * x, float * y, int64_t k) {
    const int64_t nb = k / 32;
    for (int64_t i = 0; i < nb; i++) {
        const float d = e8m0_to_fp32_half(x[i].e);
        for (int j = 0; j < 16; ++j) {
            y[i*32 + j +  0] = kvalues_mxfp4[x[i].qs[j] & 0x0F]*d;
            y[i*32 + j + 16] = kvalues_mxfp4[x[i].qs[j] >>   4]*d;
        }
    }
}

static int64_t blck_size(int type) {
    switch (type) {
        case T_F32: case T_F16: return 1;
        case T_Q4_0: case T_Q8_0: case T_MXFP4: return 32;
        case T_Q4_K: case T_Q5_K: case T_Q6_K: case T_Q8_K: return QK_K;
    }
    return 0;
}
static size_t type_size(int type) {
    switch (type) {
        case T_F32: return 4; case T_F16: return 2;
        case T_Q4_0: return 18; case T_Q8_0: return 34; case T_MXFP4: return 17;
        case T_Q4_K: return 144; case T_Q5_K: return 176; case T_Q6_K: return 210; case T_Q8_K: return 292;
    }
    return 0;
}
int64_t orc_blck_size(int type) { return blck_size(type); }
int64_t orc_type_size(int type) { return (int64_t) type_size(type); }
int64_t orc_row_size(int type, int64_t k) { return (int64_t)(type_size(type) * (k / blck_size(type))); }

// to_float of the type traits (tests/test-backend-ops.cpp:166, tests/test-quantize-fns.cpp:53)
int orc_dequantize_row(int type, const void * x, float * y, int64_t k) {
    switch (type) {
        case T_F32:   memcpy(y, x, k*4); return 0;
        case T_F16:   orc_fp16_to_fp32_row((const uint16_t *) x, y, k); return 0;
        case T_Q4_0:  dequantize_row_q4_0 ((const block_q4_0  *) x, y, k); return 0;
        case T_Q8_0:  dequantize_row_q8_0 ((const block_q8_0  *) x, y, k); return 0;
        case T_Q4_K:  dequantize_row_q4_K ((const block_q4_K  *) x, y, k); return 0;
        case T_Q5_K:  dequantize_row_q5_K ((const block_q5_K  *) x, y, k); return 0;
        case T_Q6_K:  dequantize_row_q6_K ((const block_q6_K  *) x, y, k); return 0;
        case T_MXFP4: dequantize_row_mxfp4((const block_mxfp4 *) x, y, k); return 0;
    }
    return -1;
}

// ---- quantize_row_*_ref (PINNED by tests/golden/quant_*.npz for q4_0/q8_0/mxfp4)
// gguf-py/gguf/quants.py:222-238
static void quantize_row_q4_0_ref(const float * x, block_q4_0 * y, int64_t k) {
    const int64_t nb = k / 32;
    for (int64_t i = 0; i < nb; i++) {
        float amax = 0.0f, max = 0.0f;
        for (int j = 0; j < 32; j++) {
            const float v = x[i*32 + j];
            if (amax < fabsf(v)) { amax = fabsf(v); max = v; }
        }
        const float d  = max / -8;
        const float id = d ? 1.0f/d : 0.0f;
        y[i].d = fp32_to_fp16(d);
        for (int j = 0; j < 16; ++j) {
            const float x0 = x[i*32 + 0  + j]*id;
            const float x1 = x[i*32 + 16 + j]*id;
            int v0 = (int8_t)(x0 + 8.5f), v1 = (int8_t)(x1 + 8.5f);
            const uint8_t xi0 = v0 < 15 ? v0 : 15;
            const uint8_t xi1 = v1 < 15 ? v1 : 15;
            y[i].qs[j] = xi0 | (xi1 << 4);
        }
    }
}

// gguf-py/gguf/quants.py:381-393 ("bit-exact same results as reference implementation in ggml-quants.c")
static void quantize_row_q8_0_ref(const float * x, block_q8_0 * y, int64_t k) {
    const int64_t nb = k / 32;
    for (int64_t i = 0; i < nb; i++) {
        float amax = 0.0f;
        for (int j = 0; j < 32; j++) { const float v = fabsf(x[i*32 + j]); if (v > amax) amax = v; }
        const float d = amax / ((1 << 7) - 1);
        const float id = d ? 1.0f/d : 0.0f;
        y[i].d = fp32_to_fp16(d);
        for (int j = 0; j < 32; ++j) y[i].qs[j] = (int8_t) roundf(x[i*32 + j]*id);
    }
}

// gguf-py/gguf/quants.py:668-688
static void quantize_row_mxfp4_ref(const float * x, block_mxfp4 * y, int64_t k) {
    const int64_t nb = k / 32;
    for (int64_t i = 0; i < nb; i++) {
        float amax = 0.0f;
        for (int j = 0; j < 32; j++) { const float v = fabsf(x[i*32 + j]); if (amax < v) amax = v; }
        const uint8_t e = amax > 0.0f ? (uint8_t)(floorf(log2f(amax)) - 2 + 127) : 0;
        const float d = e8m0_to_fp32_half(e);
        y[i].e = e;
        for (int j = 0; j < 16; ++j) {
            uint8_t best[2];
            for (int h = 0; h < 2; h++) {
                const float v = x[i*32 + h*16 + j];
                int bi = 0; float be = fabsf(kvalues_mxfp4[0]*d - v);
                for (int c = 1; c < 16; c++) {
                    const float err = fabsf(kvalues_mxfp4[c]*d - v);
                    if (err < be) { bi = c; be = err; }
                }
                best[h] = (uint8_t) bi;
            }
            y[i].qs[j] = best[0] | (best[1] << 4);
        }
    }
}

// round-half-to-even via the float magic number ([UPSTREAM-KNOWLEDGE] ggml-quants.c nearest_int)
static inline int nearest_int(float fval) {
    float val = fval + 12582912.f;
    int i; memcpy(&i, &val, sizeof(int));
    return (i & 0x007fffff) - 0x00400000;
}

// [UPSTREAM-KNOWLEDGE] quantize_row_q8_K_ref — activation quantizer for all K-quant dot products
static void quantize_row_q8_K_ref(const float * x, block_q8_K * y, int64_t k) {
    const int64_t nb = k / QK_K;
    for (int64_t i = 0; i < nb; i++) {
        float max = 0, amax = 0;
        for (int j = 0; j < QK_K; ++j) {
            const float ax = fabsf(x[j]);
            if (ax > amax) { amax = ax; max = x[j]; }
        }
        if (!amax) {
            y[i].d = 0;
            memset(y[i].qs, 0, QK_K);
            memset(y[i].bsums, 0, sizeof(y[i].bsums));
            x += QK_K;
            continue;
        }
        const float iscale = -127.f/max;
        for (int j = 0; j < QK_K; ++j) {
            const int v = nearest_int(iscale*x[j]);
            y[i].qs[j] = v < 127 ? v : 127;
        }
        for (int j = 0; j < QK_K/16; ++j) {
            int sum = 0;
            for (int ii = 0; ii < 16; ++ii) sum += y[i].qs[j*16 + ii];
            y[i].bsums[j] = (int16_t) sum;
        }
        y[i].d = 1/iscale;
        x += QK_K;
    }
}

int orc_quantize_row(int type, const float * x, void * y, int64_t k) {
    switch (type) {
        case T_Q4_0:  quantize_row_q4_0_ref (x, (block_q4_0  *) y, k); return 0;
        case T_Q8_0:  quantize_row_q8_0_ref (x, (block_q8_0  *) y, k); return 0;
        case T_MXFP4: quantize_row_mxfp4_ref(x, (block_mxfp4 *) y, k); return 0;
        case T_Q8_K:  quantize_row_q8_K_ref (x, (block_q8_K  *) y, k); return 0;
        case T_F16:   orc_fp32_to_fp16_row(x, (uint16_t *) y, k); return 0;
        case T_F32:   memcpy(y, x, k*4); return 0;
    }
    return -1;
}

// vec_dot_type of each weight type ([UPSTREAM-KNOWLEDGE] ggml-cpu type traits)
int orc_vec_dot_type(int type) {
    switch (type) {
        case T_Q4_0: case T_Q8_0: case T_MXFP4: return T_Q8_0;
        case T_Q4_K: case T_Q5_K: case T_Q6_K:  return T_Q8_K;
        case T_F16: return T_F16;
        case T_F32: return T_F32;
    }
    return -1;
}

// ---- integer vec_dot, ggml's generic (scalar) form [UPSTREAM-KNOWLEDGE] -------
static float vec_dot_q4_0_q8_0(int64_t n, const block_q4_0 * x, const block_q8_0 * y) {
    const int64_t nb = n / 32;
    float sumf = 0;
    for (int64_t ib = 0; ib < nb; ++ib) {
        int sumi0 = 0, sumi1 = 0;
        for (int j = 0; j < 16; ++j) {
            const int v0 = (x[ib].qs[j] & 0x0F) - 8;
            const int v1 = (x[ib].qs[j] >>   4) - 8;
            sumi0 += v0 * y[ib].qs[j];
            sumi1 += v1 * y[ib].qs[j + 16];
        }
        const int sumi = sumi0 + sumi1;
        sumf += sumi*fp16_to_fp32(x[ib].d)*fp16_to_fp32(y[ib].d);
    }
    return sumf;
}

static float vec_dot_q8_0_q8_0(int64_t n, const block_q8_0 * x, const block_q8_0 * y) {
    const int64_t nb = n / 32;
    float sumf = 0;
    for (int64_t ib = 0; ib < nb; ++ib) {
        int sumi = 0;
        for (int j = 0; j < 32; j++) sumi += x[ib].qs[j]*y[ib].qs[j];
        sumf += sumi*(fp16_to_fp32(x[ib].d)*fp16_to_fp32(y[ib].d));
    }
    return sumf;
}

static float vec_dot_mxfp4_q8_0(int64_t n, const block_mxfp4 * x, const block_q8_0 * y) {
    const int64_t nb = n / 32;
    float sumf = 0;
    for (int64_t ib = 0; ib < nb; ++ib) {
        const float d = fp16_to_fp32(y[ib].d)*e8m0_to_fp32_half(x[ib].e);
        int sumi1 = 0, sumi2 = 0;
        for (int j = 0; j < 16; ++j) {
            sumi1 += y[ib].qs[j +  0] * kvalues_mxfp4[x[ib].qs[j] & 0xf];
            sumi2 += y[ib].qs[j + 16] * kvalues_mxfp4[x[ib].qs[j] >>  4];
        }
        sumf += d * (sumi1 + sumi2);
    }
    return sumf;
}

static float vec_dot_q4_K_q8_K(int64_t n, const block_q4_K * x, const block_q8_K * y) {
    const int64_t nb = n / QK_K;
    int8_t  aux8[QK_K];
    float   sums[8] = {0};
    float sumf = 0;
    for (int64_t i = 0; i < nb; ++i) {
        const uint8_t * q4 = x[i].qs;
        const int8_t  * q8 = y[i].qs;
        int32_t aux32[8] = {0};
        int8_t * a = aux8;
        for (int j = 0; j < QK_K/64; ++j) {
            for (int l = 0; l < 32; ++l) a[l] = (int8_t)(q4[l] & 0xF);
            a += 32;
            for (int l = 0; l < 32; ++l) a[l] = (int8_t)(q4[l]  >> 4);
            a += 32; q4 += 32;
        }
        uint8_t scales[8], mins[8];
        for (int j = 0; j < 8; j++) get_scale_min_k4(j, x[i].scales, &scales[j], &mins[j]);
        int sumi = 0;
        for (int j = 0; j < QK_K/16; ++j) sumi += y[i].bsums[j] * mins[j/2];
        a = aux8;
        for (int j = 0; j < QK_K/32; ++j) {
            const int32_t scale = scales[j];
            for (int g = 0; g < 4; g++) {
                for (int l = 0; l < 8; ++l) aux32[l] += scale * (int16_t)(q8[l] * a[l]);
                q8 += 8; a += 8;
            }
        }
        const float d = fp16_to_fp32(x[i].d) * y[i].d;
        for (int l = 0; l < 8; ++l) sums[l] += d * aux32[l];
        const float dmin = fp16_to_fp32(x[i].dmin) * y[i].d;
        sumf -= dmin * sumi;
    }
    for (int l = 0; l < 8; ++l) sumf += sums[l];
    return sumf;
}

static float vec_dot_q5_K_q8_K(int64_t n, const block_q5_K * x, const block_q8_K * y) {
    const int64_t nb = n / QK_K;
    int8_t  aux8[QK_K];
    float   sums[8] = {0};
    float sumf = 0;
    for (int64_t i = 0; i < nb; ++i) {
        const uint8_t * q4 = x[i].qs;
        const uint8_t * hm = x[i].qh;
        const int8_t  * q8 = y[i].qs;
        int32_t aux32[8] = {0};
        int8_t * a = aux8;
        uint8_t m = 1;
        for (int j = 0; j < QK_K/64; ++j) {
            for (int l = 0; l < 32; ++l) a[l] = (int8_t)(q4[l] & 0xF);
            for (int l = 0; l < 32; ++l) a[l] += (hm[l] & m ? 16 : 0);
            a += 32; m <<= 1;
            for (int l = 0; l < 32; ++l) a[l] = (int8_t)(q4[l]  >> 4);
            for (int l = 0; l < 32; ++l) a[l] += (hm[l] & m ? 16 : 0);
            a += 32; m <<= 1;
            q4 += 32;
        }
        uint8_t scales[8], mins[8];
        for (int j = 0; j < 8; j++) get_scale_min_k4(j, x[i].scales, &scales[j], &mins[j]);
        int sumi = 0;
        for (int j = 0; j < QK_K/16; ++j) sumi += y[i].bsums[j] * mins[j/2];
        a = aux8;
        for (int j = 0; j < QK_K/32; ++j) {
            const int32_t scale = scales[j];
            for (int g = 0; g < 4; g++) {
                for (int l = 0; l < 8; ++l) aux32[l] += scale * (int16_t)(q8[l] * a[l]);
                q8 += 8; a += 8;
            }
        }
        const float d = fp16_to_fp32(x[i].d) * y[i].d;
        for (int l = 0; l < 8; ++l) sums[l] += d * aux32[l];
        const float dmin = fp16_to_fp32(x[i].dmin) * y[i].d;
        sumf -= dmin * sumi;
    }
    for (int l = 0; l < 8; ++l) sumf += sums[l];
    return sumf;
}

static float vec_dot_q6_K_q8_K(int64_t n, const block_q6_K * x, const block_q8_K * y) {
    const int64_t nb = n / QK_K;
    int8_t  aux8[QK_K];
    float   sums[8] = {0};
    for (int64_t i = 0; i < nb; ++i) {
        const uint8_t * q4 = x[i].ql;
        const uint8_t * qh = x[i].qh;
        const int8_t  * q8 = y[i].qs;
        int32_t aux32[8] = {0};
        int8_t * a = aux8;
        for (int j = 0; j < QK_K; j += 128) {
            for (int l = 0; l < 32; ++l) {
                a[l +  0] = (int8_t)((q4[l +  0] & 0xF) | (((qh[l] >> 0) & 3) << 4)) - 32;
                a[l + 32] = (int8_t)((q4[l + 32] & 0xF) | (((qh[l] >> 2) & 3) << 4)) - 32;
                a[l + 64] = (int8_t)((q4[l +  0] >>  4) | (((qh[l] >> 4) & 3) << 4)) - 32;
                a[l + 96] = (int8_t)((q4[l + 32] >>  4) | (((qh[l] >> 6) & 3) << 4)) - 32;
            }
            a += 128; q4 += 64; qh += 32;
        }
        a = aux8;
        int is = 0;
        for (int j = 0; j < QK_K/16; ++j) {
            const int scale = x[i].scales[is++];
            for (int g = 0; g < 2; g++) {
                for (int l = 0; l < 8; ++l) aux32[l] += scale * (int16_t)(q8[l] * a[l]);
                q8 += 8; a += 8;
            }
        }
        const float d = fp16_to_fp32(x[i].d) * y[i].d;
        for (int l = 0; l < 8; ++l) sums[l] += d * aux32[l];
    }
    float sumf = 0;
    for (int l = 0; l < 8; ++l) sumf += sums[l];
    return sumf;
}

// ---- AVX2 forms of the two K-quant dots a Q4_K_M model spends its time in (bench.py's cpu_baseline: the scalar loops above run at ~0.3 GB/s of
// weights per core, an order of magnitude under what a vectorised CPU backend streams). Same arithmetic as the scalar restatement, lane for lane:
// the eight int32 lanes aux32[l] (elements with index = l mod 8) and the eight float lanes sums[l] are kept as they are, so the result is
// BIT-IDENTICAL to the scalar function (tests/test_oracle_golden.py checks that) — only the 32 int16 products of a sub-block are formed at once.
#if defined(__x86_64__)
#include <immintrin.h>
static int g_simd = -1;   // -1: ask the CPU once; 0: scalar; 1: AVX2
void orc_set_simd(int on) { g_simd = on ? (__builtin_cpu_supports("avx2") ? 1 : 0) : 0; }
static inline int use_avx2(void) { if (g_simd < 0) g_simd = __builtin_cpu_supports("avx2") ? 1 : 0; return g_simd; }

// lane l of the result = sum over g of (int16)(q8[8g + l] * a[8g + l]) for the 32 (n16 = 2) or the first / second 16 (n16 = 1) elements
__attribute__((target("avx2"))) static inline void lane_sums32(__m256i a8, __m256i q8, __m256i * first16, __m256i * second16) {
    const __m256i a_lo = _mm256_cvtepi8_epi16(_mm256_castsi256_si128(a8)), a_hi = _mm256_cvtepi8_epi16(_mm256_extracti128_si256(a8, 1));
    const __m256i q_lo = _mm256_cvtepi8_epi16(_mm256_castsi256_si128(q8)), q_hi = _mm256_cvtepi8_epi16(_mm256_extracti128_si256(q8, 1));
    const __m256i p_lo = _mm256_mullo_epi16(a_lo, q_lo), p_hi = _mm256_mullo_epi16(a_hi, q_hi);       // elements 0..15, 16..31 (each product fits int16)
    *first16  = _mm256_add_epi32(_mm256_cvtepi16_epi32(_mm256_castsi256_si128(p_lo)), _mm256_cvtepi16_epi32(_mm256_extracti128_si256(p_lo, 1)));
    *second16 = _mm256_add_epi32(_mm256_cvtepi16_epi32(_mm256_castsi256_si128(p_hi)), _mm256_cvtepi16_epi32(_mm256_extracti128_si256(p_hi, 1)));
}

__attribute__((target("avx2"))) static float vec_dot_q4_K_q8_K_avx2(int64_t n, const block_q4_K * x, const block_q8_K * y) {
    const int64_t nb = n / QK_K;
    float sums[8] = {0};
    float sumf = 0;
    const __m256i m4 = _mm256_set1_epi8(0xF);
    for (int64_t i = 0; i < nb; ++i) {
        uint8_t scales[8], mins[8];
        for (int j = 0; j < 8; j++) get_scale_min_k4(j, x[i].scales, &scales[j], &mins[j]);
        int sumi = 0;
        for (int j = 0; j < QK_K/16; ++j) sumi += y[i].bsums[j] * mins[j/2];
        __m256i acc = _mm256_setzero_si256();
        for (int j = 0; j < QK_K/64; ++j) {
            const __m256i q4 = _mm256_loadu_si256((const __m256i *) (x[i].qs + 32*j));
            const __m256i lo = _mm256_and_si256(q4, m4), hi = _mm256_and_si256(_mm256_srli_epi16(q4, 4), m4);
            __m256i f, s2;
            lane_sums32(lo, _mm256_loadu_si256((const __m256i *) (y[i].qs + 64*j)), &f, &s2);
            acc = _mm256_add_epi32(acc, _mm256_mullo_epi32(_mm256_set1_epi32(scales[2*j]), _mm256_add_epi32(f, s2)));
            lane_sums32(hi, _mm256_loadu_si256((const __m256i *) (y[i].qs + 64*j + 32)), &f, &s2);
            acc = _mm256_add_epi32(acc, _mm256_mullo_epi32(_mm256_set1_epi32(scales[2*j + 1]), _mm256_add_epi32(f, s2)));
        }
        int32_t aux32[8];
        _mm256_storeu_si256((__m256i *) aux32, acc);
        const float d = fp16_to_fp32(x[i].d) * y[i].d;
        for (int l = 0; l < 8; ++l) sums[l] += d * aux32[l];
        const float dmin = fp16_to_fp32(x[i].dmin) * y[i].d;
        sumf -= dmin * sumi;
    }
    for (int l = 0; l < 8; ++l) sumf += sums[l];
    return sumf;
}

__attribute__((target("avx2"))) static float vec_dot_q6_K_q8_K_avx2(int64_t n, const block_q6_K * x, const block_q8_K * y) {
    const int64_t nb = n / QK_K;
    float sums[8] = {0};
    const __m256i m4 = _mm256_set1_epi8(0xF), m2 = _mm256_set1_epi8(3), off = _mm256_set1_epi8(32);
    for (int64_t i = 0; i < nb; ++i) {
        __m256i acc = _mm256_setzero_si256();
        for (int h = 0; h < 2; h++) {                       // 128 elements: ql 64 bytes, qh 32 bytes, 8 scales
            const __m256i ql0 = _mm256_loadu_si256((const __m256i *) (x[i].ql + 64*h)), ql1 = _mm256_loadu_si256((const __m256i *) (x[i].ql + 64*h + 32));
            const __m256i qh = _mm256_loadu_si256((const __m256i *) (x[i].qh + 32*h));
            __m256i a[4];
            a[0] = _mm256_or_si256(_mm256_and_si256(ql0, m4), _mm256_slli_epi16(_mm256_and_si256(qh, m2), 4));
            a[1] = _mm256_or_si256(_mm256_and_si256(ql1, m4), _mm256_slli_epi16(_mm256_and_si256(_mm256_srli_epi16(qh, 2), m2), 4));
            a[2] = _mm256_or_si256(_mm256_and_si256(_mm256_srli_epi16(ql0, 4), m4), _mm256_slli_epi16(_mm256_and_si256(_mm256_srli_epi16(qh, 4), m2), 4));
            a[3] = _mm256_or_si256(_mm256_and_si256(_mm256_srli_epi16(ql1, 4), m4), _mm256_slli_epi16(_mm256_and_si256(_mm256_srli_epi16(qh, 6), m2), 4));
            for (int v = 0; v < 4; v++) {
                __m256i f, s2;
                lane_sums32(_mm256_sub_epi8(a[v], off), _mm256_loadu_si256((const __m256i *) (y[i].qs + 128*h + 32*v)), &f, &s2);
                const int is = 8*h + 2*v;
                acc = _mm256_add_epi32(acc, _mm256_mullo_epi32(_mm256_set1_epi32(x[i].scales[is]), f));
                acc = _mm256_add_epi32(acc, _mm256_mullo_epi32(_mm256_set1_epi32(x[i].scales[is + 1]), s2));
            }
        }
        int32_t aux32[8];
        _mm256_storeu_si256((__m256i *) aux32, acc);
        const float d = fp16_to_fp32(x[i].d) * y[i].d;
        for (int l = 0; l < 8; ++l) sums[l] += d * aux32[l];
    }
    float sumf = 0;
    for (int l = 0; l < 8; ++l) sumf += sums[l];
    return sumf;
}
#else
void orc_set_simd(int on) { (void) on; }
static inline int use_avx2(void) { return 0; }
#endif

// vec_dot(n, &s, x (weight row, type), y (row already in vec_dot_type)) — tests/test-quantize-fns.cpp:93
float orc_vec_dot(int type, int64_t n, const void * x, const void * y) {
#if defined(__x86_64__)
    if (use_avx2()) {
        if (type == T_Q4_K) return vec_dot_q4_K_q8_K_avx2(n, (const block_q4_K *) x, (const block_q8_K *) y);
        if (type == T_Q6_K) return vec_dot_q6_K_q8_K_avx2(n, (const block_q6_K *) x, (const block_q8_K *) y);
    }
#endif
    switch (type) {
        case T_Q4_0:  return vec_dot_q4_0_q8_0 (n, (const block_q4_0  *) x, (const block_q8_0 *) y);
        case T_Q8_0:  return vec_dot_q8_0_q8_0 (n, (const block_q8_0  *) x, (const block_q8_0 *) y);
        case T_MXFP4: return vec_dot_mxfp4_q8_0(n, (const block_mxfp4 *) x, (const block_q8_0 *) y);
        case T_Q4_K:  return vec_dot_q4_K_q8_K (n, (const block_q4_K  *) x, (const block_q8_K *) y);
        case T_Q5_K:  return vec_dot_q5_K_q8_K (n, (const block_q5_K  *) x, (const block_q8_K *) y);
        case T_Q6_K:  return vec_dot_q6_K_q8_K (n, (const block_q6_K  *) x, (const block_q8_K *) y);
    }
    return NAN;
}

// ---- MUL_MAT (2-D core; broadcast over dims 2/3 is done by the Python wrapper) --
// dst[i1*m + i0] = sum_k a[i0][k] * b[i1][k]   (tests/test-backend-ops.cpp:3128)
// mode 0: exact — dequantize a row to f32, accumulate in f64 (the "true" value)
// mode 1: CPU-backend style — quantize each b row to vec_dot_type, integer vec_dot
int orc_mul_mat(int type, const void * a, const float * b, float * dst, int64_t m, int64_t n, int64_t k, int mode) {
    const size_t rs = type_size(type) * (k / blck_size(type));
    if (mode == 0 || type == T_F32 || type == T_F16) {
        #pragma omp parallel
        {
            float * row = (float *) malloc(k*sizeof(float));
            #pragma omp for schedule(static)
            for (int64_t i0 = 0; i0 < m; i0++) {
                orc_dequantize_row(type, (const char *) a + i0*rs, row, k);
                for (int64_t i1 = 0; i1 < n; i1++) {
                    double acc = 0;
                    const float * bb = b + i1*k;
                    for (int64_t kk = 0; kk < k; kk++) acc += (double) row[kk] * (double) bb[kk];
                    dst[i1*m + i0] = (float) acc;
                }
            }
            free(row);
        }
        return 0;
    }
    const int vt = orc_vec_dot_type(type);
    if (vt != T_Q8_0 && vt != T_Q8_K) return -1;
    const size_t qrs = type_size(vt) * (k / blck_size(vt));
    char * bq = (char *) malloc(qrs * n);
    for (int64_t i1 = 0; i1 < n; i1++) orc_quantize_row(vt, b + i1*k, bq + i1*qrs, k);
    #pragma omp parallel for schedule(static)
    for (int64_t i0 = 0; i0 < m; i0++) {
        for (int64_t i1 = 0; i1 < n; i1++) {
            dst[i1*m + i0] = orc_vec_dot(type, k, (const char *) a + i0*rs, bq + i1*qrs);
        }
    }
    free(bq);
    return 0;
}

// same, b already quantized to vec_dot_type (lets bench.py time the dot products alone)
int orc_mul_mat_q(int type, const void * a, const void * bq, float * dst, int64_t m, int64_t n, int64_t k) {
    const int vt = orc_vec_dot_type(type);
    const size_t rs = type_size(type) * (k / blck_size(type));
    const size_t qrs = type_size(vt) * (k / blck_size(vt));
    #pragma omp parallel for schedule(static)
    for (int64_t i0 = 0; i0 < m; i0++) {
        for (int64_t i1 = 0; i1 < n; i1++) {
            dst[i1*m + i0] = orc_vec_dot(type, k, (const char *) a + i0*rs, (const char *) bq + i1*qrs);
        }
    }
    return 0;
}
