// ggml-compat.cpp — minimal stand-in for libggml-base, HARNESS ONLY.
//
// The reference's ggml/ submodule is empty (/root/reference/.gitmodules:1-3), so
// there is no libggml-base to load our backend. This file restates just enough
// of ggml's host side for the backend to be driven exactly the way the
// reference drives it (tests/test-backend-ops.cpp:1082-1240 for single ops,
// src/llama-context.cpp:714-776 for graphs): tensor/graph construction, the
// type-traits table, the public ggml_backend_* wrappers that dispatch through
// the vtables, a sequential buffer allocator and a dlopen-based registry.
// There is NO compute here: a graph can only be evaluated by a backend.
// With a real ggml checkout this file is not built.
#include "ggml.h"
#include "ggml-backend.h"
#include "ggml-backend-impl.h"
#include "ggml-impl.h"

#include <dlfcn.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <string>
#include <vector>

extern "C" {

void ggml_abort(const char * file, int line, const char * fmt, ...) {
    fflush(stdout);
    fprintf(stderr, "%s:%d: ", file, line);
    va_list args;
    va_start(args, fmt);
    vfprintf(stderr, fmt, args);
    va_end(args);
    fprintf(stderr, "\n");
    abort();
}

bool ggml_guid_matches(ggml_guid_t a, ggml_guid_t b) { return memcmp(a, b, sizeof(ggml_guid)) == 0; }

// ---------------------------------------------------------------------------
// type traits — sizes from gguf-py/gguf/constants.py:2839-2872
// ---------------------------------------------------------------------------
struct type_traits { const char * name; int64_t blck; size_t size; bool quant; };

static const type_traits * traits(enum ggml_type t) {
    static type_traits tt[GGML_TYPE_COUNT];
    static bool init = false;
    if (!init) {
        for (auto & e : tt) e = { nullptr, 0, 0, false };
        tt[GGML_TYPE_F32]   = { "f32",   1,   4,   false };
        tt[GGML_TYPE_F16]   = { "f16",   1,   2,   false };
        tt[GGML_TYPE_BF16]  = { "bf16",  1,   2,   false };
        tt[GGML_TYPE_F64]   = { "f64",   1,   8,   false };
        tt[GGML_TYPE_I8]    = { "i8",    1,   1,   false };
        tt[GGML_TYPE_I16]   = { "i16",   1,   2,   false };
        tt[GGML_TYPE_I32]   = { "i32",   1,   4,   false };
        tt[GGML_TYPE_I64]   = { "i64",   1,   8,   false };
        tt[GGML_TYPE_Q4_0]  = { "q4_0",  32,  18,  true  };
        tt[GGML_TYPE_Q4_1]  = { "q4_1",  32,  20,  true  };
        tt[GGML_TYPE_Q5_0]  = { "q5_0",  32,  22,  true  };
        tt[GGML_TYPE_Q5_1]  = { "q5_1",  32,  24,  true  };
        tt[GGML_TYPE_Q8_0]  = { "q8_0",  32,  34,  true  };
        tt[GGML_TYPE_Q8_1]  = { "q8_1",  32,  36,  true  };
        tt[GGML_TYPE_Q2_K]  = { "q2_K",  256, 84,  true  };
        tt[GGML_TYPE_Q3_K]  = { "q3_K",  256, 110, true  };
        tt[GGML_TYPE_Q4_K]  = { "q4_K",  256, 144, true  };
        tt[GGML_TYPE_Q5_K]  = { "q5_K",  256, 176, true  };
        tt[GGML_TYPE_Q6_K]  = { "q6_K",  256, 210, true  };
        tt[GGML_TYPE_Q8_K]  = { "q8_K",  256, 292, true  };
        tt[GGML_TYPE_MXFP4] = { "mxfp4", 32,  17,  true  };
        init = true;
    }
    GGML_ASSERT(t >= 0 && t < GGML_TYPE_COUNT);
    return &tt[t];
}

int64_t      ggml_blck_size(enum ggml_type t) { return traits(t)->blck; }
size_t       ggml_type_size(enum ggml_type t) { return traits(t)->size; }
const char * ggml_type_name(enum ggml_type t) { return t < GGML_TYPE_COUNT && traits(t)->name ? traits(t)->name : "NONE"; }
bool         ggml_is_quantized(enum ggml_type t) { return traits(t)->quant; }
size_t       ggml_row_size(enum ggml_type t, int64_t ne) {
    GGML_ASSERT(traits(t)->blck > 0 && ne % traits(t)->blck == 0);
    return traits(t)->size * ne / traits(t)->blck;
}

const char * ggml_op_name(enum ggml_op op) {
    switch (op) {
        case GGML_OP_NONE: return "NONE";           case GGML_OP_DUP: return "DUP";
        case GGML_OP_ADD: return "ADD";             case GGML_OP_ADD_ID: return "ADD_ID";
        case GGML_OP_SUB: return "SUB";             case GGML_OP_MUL: return "MUL";
        case GGML_OP_DIV: return "DIV";             case GGML_OP_SUM_ROWS: return "SUM_ROWS";
        case GGML_OP_RMS_NORM: return "RMS_NORM";   case GGML_OP_MUL_MAT: return "MUL_MAT";
        case GGML_OP_MUL_MAT_ID: return "MUL_MAT_ID"; case GGML_OP_SCALE: return "SCALE";
        case GGML_OP_CPY: return "CPY";             case GGML_OP_CONT: return "CONT";
        case GGML_OP_RESHAPE: return "RESHAPE";     case GGML_OP_VIEW: return "VIEW";
        case GGML_OP_PERMUTE: return "PERMUTE";     case GGML_OP_TRANSPOSE: return "TRANSPOSE";
        case GGML_OP_GET_ROWS: return "GET_ROWS";   case GGML_OP_SET_ROWS: return "SET_ROWS";
        case GGML_OP_SOFT_MAX: return "SOFT_MAX";   case GGML_OP_ROPE: return "ROPE";
        case GGML_OP_ARGSORT: return "ARGSORT";     case GGML_OP_FLASH_ATTN_EXT: return "FLASH_ATTN_EXT";
        case GGML_OP_UNARY: return "UNARY";         case GGML_OP_GLU: return "GLU";
        default: return "OTHER";
    }
}
const char * ggml_op_desc(const struct ggml_tensor * t) { return ggml_op_name(t->op); }

const char * ggml_status_to_string(enum ggml_status s) {
    switch (s) {
        case GGML_STATUS_ALLOC_FAILED: return "GGML status: error (failed to allocate memory)";
        case GGML_STATUS_FAILED:       return "GGML status: error (operation failed)";
        case GGML_STATUS_SUCCESS:      return "GGML status: success";
        case GGML_STATUS_ABORTED:      return "GGML status: warning (operation aborted)";
    }
    return "GGML status: unknown";
}

int64_t ggml_nelements(const struct ggml_tensor * t) { return t->ne[0]*t->ne[1]*t->ne[2]*t->ne[3]; }
int64_t ggml_nrows(const struct ggml_tensor * t)     { return t->ne[1]*t->ne[2]*t->ne[3]; }
size_t  ggml_element_size(const struct ggml_tensor * t) { return ggml_type_size(t->type); }

size_t ggml_nbytes(const struct ggml_tensor * t) {
    for (int i = 0; i < GGML_MAX_DIMS; ++i) {
        if (t->ne[i] <= 0) return 0;
    }
    size_t nbytes;
    const size_t blck = ggml_blck_size(t->type);
    if (blck == 1) {
        nbytes = ggml_type_size(t->type);
        for (int i = 0; i < GGML_MAX_DIMS; ++i) nbytes += (t->ne[i] - 1)*t->nb[i];
    } else {
        nbytes = t->ne[0]*t->nb[0]/blck;
        for (int i = 1; i < GGML_MAX_DIMS; ++i) nbytes += (t->ne[i] - 1)*t->nb[i];
    }
    return nbytes;
}

int ggml_n_dims(const struct ggml_tensor * t) {
    for (int i = GGML_MAX_DIMS - 1; i >= 1; --i) {
        if (t->ne[i] > 1) return i + 1;
    }
    return 1;
}

bool ggml_is_transposed(const struct ggml_tensor * t) { return t->nb[0] > t->nb[1]; }
bool ggml_is_permuted(const struct ggml_tensor * t) {
    return t->nb[0] > t->nb[1] || t->nb[1] > t->nb[2] || t->nb[2] > t->nb[3];
}
bool ggml_is_empty(const struct ggml_tensor * t) {
    for (int i = 0; i < GGML_MAX_DIMS; ++i) if (t->ne[i] == 0) return true;
    return false;
}

static bool is_contiguous_n(const struct ggml_tensor * t, int n) {
    size_t next_nb = ggml_type_size(t->type);
    if (t->ne[0] != ggml_blck_size(t->type) && t->nb[0] != next_nb) return false;
    next_nb *= t->ne[0]/ggml_blck_size(t->type);
    for (int i = 1; i < GGML_MAX_DIMS; i++) {
        if (t->ne[i] != 1) {
            if (i > n) {
                if (t->nb[i] != next_nb) return false;
                next_nb *= t->ne[i];
            } else {
                // this dimension does not need to be contiguous
                next_nb = t->ne[i]*t->nb[i];
            }
        }
    }
    return true;
}
bool ggml_is_contiguous  (const struct ggml_tensor * t) { return is_contiguous_n(t, 0); }
bool ggml_is_contiguous_0(const struct ggml_tensor * t) { return is_contiguous_n(t, 0); }
bool ggml_is_contiguous_1(const struct ggml_tensor * t) { return is_contiguous_n(t, 1); }
bool ggml_is_contiguous_2(const struct ggml_tensor * t) { return is_contiguous_n(t, 2); }
bool ggml_is_contiguously_allocated(const struct ggml_tensor * t) {
    return ggml_nbytes(t) == (size_t) ggml_nelements(t) * ggml_type_size(t->type)/ggml_blck_size(t->type);
}
bool ggml_is_contiguous_rows(const struct ggml_tensor * t) {
    return t->ne[0] == ggml_blck_size(t->type) || t->nb[0] == ggml_type_size(t->type);
}
bool ggml_are_same_shape(const struct ggml_tensor * a, const struct ggml_tensor * b) {
    return a->ne[0] == b->ne[0] && a->ne[1] == b->ne[1] && a->ne[2] == b->ne[2] && a->ne[3] == b->ne[3];
}
bool ggml_are_same_stride(const struct ggml_tensor * a, const struct ggml_tensor * b) {
    return a->nb[0] == b->nb[0] && a->nb[1] == b->nb[1] && a->nb[2] == b->nb[2] && a->nb[3] == b->nb[3];
}
// can t0 be repeated to the shape of t1
bool ggml_can_repeat(const struct ggml_tensor * t0, const struct ggml_tensor * t1) {
    return ggml_is_empty(t0) ? ggml_is_empty(t1) :
        (t1->ne[0]%t0->ne[0] == 0) && (t1->ne[1]%t0->ne[1] == 0) &&
        (t1->ne[2]%t0->ne[2] == 0) && (t1->ne[3]%t0->ne[3] == 0);
}

enum ggml_unary_op ggml_get_unary_op(const struct ggml_tensor * t) {
    GGML_ASSERT(t->op == GGML_OP_UNARY);
    return (enum ggml_unary_op) t->op_params[0];
}
enum ggml_glu_op ggml_get_glu_op(const struct ggml_tensor * t) {
    GGML_ASSERT(t->op == GGML_OP_GLU);
    return (enum ggml_glu_op) t->op_params[0];
}

// IEEE binary16 <-> binary32, round-to-nearest-even
float ggml_fp16_to_fp32(ggml_fp16_t h) {
    const uint32_t sign = (uint32_t)(h & 0x8000) << 16;
    const uint32_t exp  = (h >> 10) & 0x1F;
    const uint32_t man  = h & 0x3FF;
    uint32_t bits;
    if (exp == 0) {
        if (man == 0) {
            bits = sign;
        } else {
            // subnormal: normalise
            int e = -1;
            uint32_t m = man;
            do { m <<= 1; e++; } while ((m & 0x400) == 0);
            bits = sign | ((uint32_t)(127 - 15 - e) << 23) | ((m & 0x3FF) << 13);
        }
    } else if (exp == 31) {
        bits = sign | 0x7F800000u | (man << 13);
    } else {
        bits = sign | ((exp + 127 - 15) << 23) | (man << 13);
    }
    float f;
    memcpy(&f, &bits, 4);
    return f;
}

ggml_fp16_t ggml_fp32_to_fp16(float f) {
    uint32_t x;
    memcpy(&x, &f, 4);
    const uint32_t sign = (x >> 16) & 0x8000;
    const uint32_t ax = x & 0x7FFFFFFFu;
    if (ax >= 0x7F800000u) { // inf / nan
        return (ggml_fp16_t)(sign | 0x7C00 | ((ax > 0x7F800000u) ? (0x200 | ((ax >> 13) & 0x3FF)) : 0));
    }
    if (ax >= 0x477FF000u) { // rounds to >= 65520 -> inf
        return (ggml_fp16_t)(sign | 0x7C00);
    }
    if (ax < 0x33000001u) { // < 2^-25 (or == 2^-25 which ties to even = 0)
        return (ggml_fp16_t) sign;
    }
    int32_t e = (int32_t)(ax >> 23) - 127;
    uint32_t m = (ax & 0x7FFFFF) | 0x800000;
    uint32_t shift;
    uint32_t hexp;
    if (e < -14) { // subnormal half
        shift = (uint32_t)(13 + (-14 - e));
        hexp = 0;
    } else {
        shift = 13;
        hexp = (uint32_t)(e + 15);
    }
    uint32_t hm = m >> shift;
    const uint32_t rem = m & ((1u << shift) - 1);
    const uint32_t half = 1u << (shift - 1);
    if (rem > half || (rem == half && (hm & 1))) hm++;
    uint32_t h;
    if (hexp == 0) {
        h = hm; // may carry into exponent bit 10 -> smallest normal, which is correct
    } else {
        h = ((hexp - 1) << 10) + hm; // hm has the implicit bit (0x400); carries propagate
    }
    return (ggml_fp16_t)(sign | h);
}

// ---------------------------------------------------------------------------
// context
// ---------------------------------------------------------------------------
struct ggml_context {
    bool no_alloc;
    std::vector<ggml_tensor *> tensors;
    std::vector<void *>        blobs;   // graphs, tensor data when !no_alloc
};

size_t ggml_tensor_overhead(void) { return sizeof(ggml_tensor) + 32; }
size_t ggml_graph_overhead_custom(size_t size, bool) { return sizeof(ggml_cgraph) + size*6*sizeof(void *); }

struct ggml_context * ggml_init(struct ggml_init_params params) {
    ggml_context * ctx = new ggml_context;
    ctx->no_alloc = params.no_alloc;
    return ctx;
}

void ggml_free(struct ggml_context * ctx) {
    if (!ctx) return;
    for (auto * t : ctx->tensors) free(t);
    for (auto * b : ctx->blobs) free(b);
    delete ctx;
}

struct ggml_tensor * ggml_get_first_tensor(const struct ggml_context * ctx) {
    return ctx->tensors.empty() ? nullptr : ctx->tensors[0];
}
struct ggml_tensor * ggml_get_next_tensor(const struct ggml_context * ctx, struct ggml_tensor * tensor) {
    // tensors carry their index in `extra`-free space: linear scan is fine for harness sizes,
    // but graphs with thousands of nodes make this quadratic -> keep an index in padding
    size_t idx;
    memcpy(&idx, tensor->padding, sizeof(idx));
    return idx + 1 < ctx->tensors.size() ? ctx->tensors[idx + 1] : nullptr;
}

static struct ggml_tensor * new_tensor_impl(struct ggml_context * ctx, enum ggml_type type, int n_dims, const int64_t * ne,
                                            struct ggml_tensor * view_src, size_t view_offs) {
    GGML_ASSERT(type >= 0 && type < GGML_TYPE_COUNT && traits(type)->blck > 0);
    GGML_ASSERT(n_dims >= 1 && n_dims <= GGML_MAX_DIMS);

    if (view_src != NULL && view_src->view_src != NULL) {
        view_offs += view_src->view_offs;
        view_src   = view_src->view_src;
    }

    ggml_tensor * t = (ggml_tensor *) calloc(1, sizeof(ggml_tensor));
    t->type = type;
    for (int i = 0; i < GGML_MAX_DIMS; i++) t->ne[i] = i < n_dims ? ne[i] : 1;
    t->nb[0] = ggml_type_size(type);
    t->nb[1] = t->nb[0]*(t->ne[0]/ggml_blck_size(type));
    for (int i = 2; i < GGML_MAX_DIMS; i++) t->nb[i] = t->nb[i - 1]*t->ne[i - 1];
    t->op = GGML_OP_NONE;
    t->view_src = view_src;
    t->view_offs = view_offs;
    if (view_src != NULL) {
        t->data = view_src->data ? (char *) view_src->data + view_offs : NULL;
    } else if (!ctx->no_alloc) {
        void * d = NULL;
        GGML_ASSERT(posix_memalign(&d, 64, GGML_PAD(ggml_nbytes(t) + 64, 64)) == 0);
        ctx->blobs.push_back(d);
        t->data = d;
    }
    size_t idx = ctx->tensors.size();
    memcpy(t->padding, &idx, sizeof(idx));
    ctx->tensors.push_back(t);
    return t;
}

struct ggml_tensor * ggml_new_tensor(struct ggml_context * ctx, enum ggml_type type, int n_dims, const int64_t * ne) {
    return new_tensor_impl(ctx, type, n_dims, ne, NULL, 0);
}
struct ggml_tensor * ggml_new_tensor_1d(struct ggml_context * ctx, enum ggml_type type, int64_t ne0) {
    return ggml_new_tensor(ctx, type, 1, &ne0);
}
struct ggml_tensor * ggml_new_tensor_2d(struct ggml_context * ctx, enum ggml_type type, int64_t ne0, int64_t ne1) {
    const int64_t ne[2] = { ne0, ne1 };
    return ggml_new_tensor(ctx, type, 2, ne);
}
struct ggml_tensor * ggml_new_tensor_3d(struct ggml_context * ctx, enum ggml_type type, int64_t ne0, int64_t ne1, int64_t ne2) {
    const int64_t ne[3] = { ne0, ne1, ne2 };
    return ggml_new_tensor(ctx, type, 3, ne);
}
struct ggml_tensor * ggml_new_tensor_4d(struct ggml_context * ctx, enum ggml_type type, int64_t ne0, int64_t ne1, int64_t ne2, int64_t ne3) {
    const int64_t ne[4] = { ne0, ne1, ne2, ne3 };
    return ggml_new_tensor(ctx, type, 4, ne);
}
struct ggml_tensor * ggml_dup_tensor(struct ggml_context * ctx, const struct ggml_tensor * src) {
    return ggml_new_tensor(ctx, src->type, GGML_MAX_DIMS, src->ne);
}
struct ggml_tensor * ggml_view_tensor(struct ggml_context * ctx, struct ggml_tensor * src) {
    ggml_tensor * r = new_tensor_impl(ctx, src->type, GGML_MAX_DIMS, src->ne, src, 0);
    snprintf(r->name, sizeof(r->name), "%.50s (view)", src->name);
    for (int i = 0; i < GGML_MAX_DIMS; i++) r->nb[i] = src->nb[i];
    return r;
}

const char * ggml_get_name(const struct ggml_tensor * t) { return t->name; }
struct ggml_tensor * ggml_set_name(struct ggml_tensor * t, const char * name) {
    snprintf(t->name, sizeof(t->name), "%s", name);
    return t;
}
void ggml_set_input (struct ggml_tensor * t) { t->flags |= GGML_TENSOR_FLAG_INPUT; }
void ggml_set_output(struct ggml_tensor * t) { t->flags |= GGML_TENSOR_FLAG_OUTPUT; }

static void set_f32(struct ggml_tensor * t, int i, float v) { memcpy(&t->op_params[i], &v, 4); }

// ---------------------------------------------------------------------------
// op constructors (shape rules restated from how the reference calls/tests each op)
// ---------------------------------------------------------------------------
static struct ggml_tensor * binop(struct ggml_context * ctx, enum ggml_op op, struct ggml_tensor * a, struct ggml_tensor * b) {
    GGML_ASSERT(ggml_can_repeat(b, a));
    ggml_tensor * r = ggml_dup_tensor(ctx, a);
    r->op = op; r->src[0] = a; r->src[1] = b;
    return r;
}
struct ggml_tensor * ggml_add(struct ggml_context * ctx, struct ggml_tensor * a, struct ggml_tensor * b) { return binop(ctx, GGML_OP_ADD, a, b); }
struct ggml_tensor * ggml_mul(struct ggml_context * ctx, struct ggml_tensor * a, struct ggml_tensor * b) { return binop(ctx, GGML_OP_MUL, a, b); }
struct ggml_tensor * ggml_div(struct ggml_context * ctx, struct ggml_tensor * a, struct ggml_tensor * b) { return binop(ctx, GGML_OP_DIV, a, b); }

// tests/test-backend-ops.cpp:2548 test_add_id
struct ggml_tensor * ggml_add_id(struct ggml_context * ctx, struct ggml_tensor * a, struct ggml_tensor * b, struct ggml_tensor * ids) {
    GGML_ASSERT(a->ne[0] == b->ne[0]);
    GGML_ASSERT(a->ne[1] == ids->ne[0]);
    GGML_ASSERT(a->ne[2] == ids->ne[1]);
    GGML_ASSERT(ids->type == GGML_TYPE_I32);
    ggml_tensor * r = ggml_dup_tensor(ctx, a);
    r->op = GGML_OP_ADD_ID; r->src[0] = a; r->src[1] = b; r->src[2] = ids;
    return r;
}

struct ggml_tensor * ggml_sum_rows(struct ggml_context * ctx, struct ggml_tensor * a) {
    int64_t ne[GGML_MAX_DIMS] = { 1, a->ne[1], a->ne[2], a->ne[3] };
    ggml_tensor * r = ggml_new_tensor(ctx, a->type, GGML_MAX_DIMS, ne);
    r->op = GGML_OP_SUM_ROWS; r->src[0] = a;
    return r;
}

struct ggml_tensor * ggml_scale_bias(struct ggml_context * ctx, struct ggml_tensor * a, float s, float b) {
    ggml_tensor * r = ggml_dup_tensor(ctx, a);
    set_f32(r, 0, s); set_f32(r, 1, b);
    r->op = GGML_OP_SCALE; r->src[0] = a;
    return r;
}
struct ggml_tensor * ggml_scale(struct ggml_context * ctx, struct ggml_tensor * a, float s) { return ggml_scale_bias(ctx, a, s, 0.0f); }

struct ggml_tensor * ggml_rms_norm(struct ggml_context * ctx, struct ggml_tensor * a, float eps) {
    ggml_tensor * r = ggml_dup_tensor(ctx, a);
    set_f32(r, 0, eps);
    r->op = GGML_OP_RMS_NORM; r->src[0] = a;
    return r;
}

// tests/test-backend-ops.cpp:3128 "C^T = A * B^T: (k, m) * (k, n) => (m, n)"
struct ggml_tensor * ggml_mul_mat(struct ggml_context * ctx, struct ggml_tensor * a, struct ggml_tensor * b) {
    GGML_ASSERT(a->ne[0] == b->ne[0] && b->ne[2] % a->ne[2] == 0 && b->ne[3] % a->ne[3] == 0);
    GGML_ASSERT(!ggml_is_transposed(a));
    const int64_t ne[4] = { a->ne[1], b->ne[1], b->ne[2], b->ne[3] };
    ggml_tensor * r = ggml_new_tensor(ctx, GGML_TYPE_F32, 4, ne);
    r->op = GGML_OP_MUL_MAT; r->src[0] = a; r->src[1] = b;
    return r;
}
void ggml_mul_mat_set_prec(struct ggml_tensor * a, enum ggml_prec prec) {
    GGML_ASSERT(a->op == GGML_OP_MUL_MAT);
    a->op_params[0] = (int32_t) prec;
}

// tests/test-backend-ops.cpp:3226-3245
struct ggml_tensor * ggml_mul_mat_id(struct ggml_context * ctx, struct ggml_tensor * as, struct ggml_tensor * b, struct ggml_tensor * ids) {
    GGML_ASSERT(!ggml_is_transposed(as));
    GGML_ASSERT(ids->type == GGML_TYPE_I32);
    GGML_ASSERT(as->ne[3] == 1);                    // as is 3d (one matrix per expert)
    GGML_ASSERT(b->ne[3] == 1);                     // b is 3d
    GGML_ASSERT(ids->ne[2] == 1 && ids->ne[3] == 1); // ids is 2d
    GGML_ASSERT(ids->ne[1] == b->ne[2]);            // must have an expert list per b row
    GGML_ASSERT(as->ne[0] == b->ne[0]);             // can_mul_mat
    GGML_ASSERT(ids->ne[0] % b->ne[1] == 0);        // can broadcast
    const int64_t ne[4] = { as->ne[1], ids->ne[0], b->ne[2], 1 };
    ggml_tensor * r = ggml_new_tensor(ctx, GGML_TYPE_F32, 4, ne);
    r->op = GGML_OP_MUL_MAT_ID; r->src[0] = as; r->src[1] = b; r->src[2] = ids;
    return r;
}

struct ggml_tensor * ggml_cpy(struct ggml_context * ctx, struct ggml_tensor * a, struct ggml_tensor * b) {
    GGML_ASSERT(ggml_nelements(a) == ggml_nelements(b));
    ggml_tensor * r = ggml_view_tensor(ctx, b);
    r->op = GGML_OP_CPY; r->src[0] = a; r->src[1] = b;
    return r;
}
struct ggml_tensor * ggml_cast(struct ggml_context * ctx, struct ggml_tensor * a, enum ggml_type type) {
    ggml_tensor * r = ggml_new_tensor(ctx, type, GGML_MAX_DIMS, a->ne);
    r->op = GGML_OP_CPY; r->src[0] = a; r->src[1] = r;
    return r;
}
struct ggml_tensor * ggml_flash_attn_ext(struct ggml_context * ctx, struct ggml_tensor * q, struct ggml_tensor * k, struct ggml_tensor * v,
                                         struct ggml_tensor * mask, float scale, float max_bias, float logit_softcap) {
    GGML_ASSERT(q->ne[0] == k->ne[0] && k->ne[1] == v->ne[1] && k->ne[2] == v->ne[2] && q->ne[2] % k->ne[2] == 0);
    if (mask) {
        GGML_ASSERT(ggml_is_contiguous(mask));
        GGML_ASSERT(mask->ne[0] == k->ne[1] && mask->ne[1] >= q->ne[1]);
    }
    if (max_bias > 0.0f) GGML_ASSERT(mask);
    const int64_t ne[4] = { v->ne[0], q->ne[2], q->ne[1], q->ne[3] };
    ggml_tensor * r = ggml_new_tensor(ctx, GGML_TYPE_F32, 4, ne);
    set_f32(r, 0, scale); set_f32(r, 1, max_bias); set_f32(r, 2, logit_softcap);
    r->op = GGML_OP_FLASH_ATTN_EXT; r->src[0] = q; r->src[1] = k; r->src[2] = v; r->src[3] = mask;
    return r;
}
void ggml_flash_attn_ext_set_prec(struct ggml_tensor * a, enum ggml_prec prec) {
    GGML_ASSERT(a->op == GGML_OP_FLASH_ATTN_EXT);
    a->op_params[3] = (int32_t) prec;
}
void ggml_flash_attn_ext_add_sinks(struct ggml_tensor * a, struct ggml_tensor * sinks) {
    if (!sinks) { a->src[4] = NULL; return; }
    GGML_ASSERT(a->op == GGML_OP_FLASH_ATTN_EXT && a->src[4] == NULL && a->src[0]->ne[2] == sinks->ne[0] && sinks->type == GGML_TYPE_F32);
    a->src[4] = sinks;
}
struct ggml_tensor * ggml_cont(struct ggml_context * ctx, struct ggml_tensor * a) {
    ggml_tensor * r = ggml_dup_tensor(ctx, a);
    r->op = GGML_OP_CONT; r->src[0] = a;
    return r;
}
struct ggml_tensor * ggml_cont_2d(struct ggml_context * ctx, struct ggml_tensor * a, int64_t ne0, int64_t ne1) {
    GGML_ASSERT(ggml_nelements(a) == ne0*ne1);
    ggml_tensor * r = ggml_new_tensor_2d(ctx, a->type, ne0, ne1);
    r->op = GGML_OP_CONT; r->src[0] = a;
    return r;
}

static struct ggml_tensor * reshape_impl(struct ggml_context * ctx, struct ggml_tensor * a, int n, const int64_t * ne) {
    GGML_ASSERT(ggml_is_contiguous(a));
    int64_t nel = 1; for (int i = 0; i < n; i++) nel *= ne[i];
    GGML_ASSERT(ggml_nelements(a) == nel);
    ggml_tensor * r = new_tensor_impl(ctx, a->type, n, ne, a, 0);
    snprintf(r->name, sizeof(r->name), "%.46s (reshaped)", a->name);
    r->op = GGML_OP_RESHAPE; r->src[0] = a;
    return r;
}
struct ggml_tensor * ggml_reshape_2d(struct ggml_context * ctx, struct ggml_tensor * a, int64_t ne0, int64_t ne1) {
    const int64_t ne[2] = { ne0, ne1 }; return reshape_impl(ctx, a, 2, ne);
}
struct ggml_tensor * ggml_reshape_3d(struct ggml_context * ctx, struct ggml_tensor * a, int64_t ne0, int64_t ne1, int64_t ne2) {
    const int64_t ne[3] = { ne0, ne1, ne2 }; return reshape_impl(ctx, a, 3, ne);
}
struct ggml_tensor * ggml_reshape_4d(struct ggml_context * ctx, struct ggml_tensor * a, int64_t ne0, int64_t ne1, int64_t ne2, int64_t ne3) {
    const int64_t ne[4] = { ne0, ne1, ne2, ne3 }; return reshape_impl(ctx, a, 4, ne);
}

static struct ggml_tensor * view_impl(struct ggml_context * ctx, struct ggml_tensor * a, int n, const int64_t * ne, size_t offset) {
    ggml_tensor * r = new_tensor_impl(ctx, a->type, n, ne, a, offset);
    snprintf(r->name, sizeof(r->name), "%.50s (view)", a->name);
    memcpy(r->op_params, &offset, sizeof(offset));
    r->op = GGML_OP_VIEW; r->src[0] = a;
    return r;
}
struct ggml_tensor * ggml_view_1d(struct ggml_context * ctx, struct ggml_tensor * a, int64_t ne0, size_t offset) {
    return view_impl(ctx, a, 1, &ne0, offset);
}
struct ggml_tensor * ggml_view_2d(struct ggml_context * ctx, struct ggml_tensor * a, int64_t ne0, int64_t ne1, size_t nb1, size_t offset) {
    const int64_t ne[2] = { ne0, ne1 };
    ggml_tensor * r = view_impl(ctx, a, 2, ne, offset);
    r->nb[1] = nb1; r->nb[2] = r->nb[1]*ne1; r->nb[3] = r->nb[2];
    return r;
}
struct ggml_tensor * ggml_view_3d(struct ggml_context * ctx, struct ggml_tensor * a, int64_t ne0, int64_t ne1, int64_t ne2, size_t nb1, size_t nb2, size_t offset) {
    const int64_t ne[3] = { ne0, ne1, ne2 };
    ggml_tensor * r = view_impl(ctx, a, 3, ne, offset);
    r->nb[1] = nb1; r->nb[2] = nb2; r->nb[3] = r->nb[2]*ne2;
    return r;
}
struct ggml_tensor * ggml_view_4d(struct ggml_context * ctx, struct ggml_tensor * a, int64_t ne0, int64_t ne1, int64_t ne2, int64_t ne3, size_t nb1, size_t nb2, size_t nb3, size_t offset) {
    const int64_t ne[4] = { ne0, ne1, ne2, ne3 };
    ggml_tensor * r = view_impl(ctx, a, 4, ne, offset);
    r->nb[1] = nb1; r->nb[2] = nb2; r->nb[3] = nb3;
    return r;
}

struct ggml_tensor * ggml_permute(struct ggml_context * ctx, struct ggml_tensor * a, int axis0, int axis1, int axis2, int axis3) {
    GGML_ASSERT(axis0 >= 0 && axis0 < 4 && axis1 >= 0 && axis1 < 4 && axis2 >= 0 && axis2 < 4 && axis3 >= 0 && axis3 < 4);
    GGML_ASSERT(axis0 != axis1 && axis0 != axis2 && axis0 != axis3 && axis1 != axis2 && axis1 != axis3 && axis2 != axis3);
    ggml_tensor * r = ggml_view_tensor(ctx, a);
    snprintf(r->name, sizeof(r->name), "%.46s (permuted)", a->name);
    int64_t ne[4]; size_t nb[4];
    ne[axis0] = a->ne[0]; ne[axis1] = a->ne[1]; ne[axis2] = a->ne[2]; ne[axis3] = a->ne[3];
    nb[axis0] = a->nb[0]; nb[axis1] = a->nb[1]; nb[axis2] = a->nb[2]; nb[axis3] = a->nb[3];
    for (int i = 0; i < 4; i++) { r->ne[i] = ne[i]; r->nb[i] = nb[i]; }
    r->op = GGML_OP_PERMUTE; r->src[0] = a;
    r->op_params[0] = axis0; r->op_params[1] = axis1; r->op_params[2] = axis2; r->op_params[3] = axis3;
    return r;
}
struct ggml_tensor * ggml_transpose(struct ggml_context * ctx, struct ggml_tensor * a) {
    ggml_tensor * r = ggml_view_tensor(ctx, a);
    snprintf(r->name, sizeof(r->name), "%.44s (transposed)", a->name);
    r->ne[0] = a->ne[1]; r->ne[1] = a->ne[0];
    r->nb[0] = a->nb[1]; r->nb[1] = a->nb[0];
    r->op = GGML_OP_TRANSPOSE; r->src[0] = a;
    return r;
}

// tests/test-backend-ops.cpp:1951 test_get_rows
struct ggml_tensor * ggml_get_rows(struct ggml_context * ctx, struct ggml_tensor * a, struct ggml_tensor * b) {
    GGML_ASSERT(a->ne[2] == b->ne[1]);
    GGML_ASSERT(b->ne[3] == 1);
    GGML_ASSERT(b->type == GGML_TYPE_I32);
    enum ggml_type type = a->type == GGML_TYPE_I32 ? GGML_TYPE_I32 : GGML_TYPE_F32;
    ggml_tensor * r = ggml_new_tensor_4d(ctx, type, a->ne[0], b->ne[0], b->ne[1], b->ne[2]);
    r->op = GGML_OP_GET_ROWS; r->src[0] = a; r->src[1] = b;
    return r;
}

// tests/test-backend-ops.cpp:2060-2127 test_set_rows: dst `a`, source rows `b` (F32), row indices `c` (I64)
struct ggml_tensor * ggml_set_rows(struct ggml_context * ctx, struct ggml_tensor * a, struct ggml_tensor * b, struct ggml_tensor * c) {
    GGML_ASSERT(a->ne[0] == b->ne[0]);
    GGML_ASSERT(a->ne[2] == b->ne[2]);
    GGML_ASSERT(a->ne[3] == b->ne[3]);
    GGML_ASSERT(b->ne[1] == c->ne[0]);
    GGML_ASSERT(b->ne[2] % c->ne[1] == 0);
    GGML_ASSERT(b->ne[3] % c->ne[2] == 0);
    GGML_ASSERT(c->ne[3] == 1);
    GGML_ASSERT(b->type == GGML_TYPE_F32);
    GGML_ASSERT(c->type == GGML_TYPE_I64);
    GGML_ASSERT(ggml_is_contiguous_rows(a));
    GGML_ASSERT(ggml_is_contiguous_rows(b));
    ggml_tensor * r = ggml_view_tensor(ctx, a);
    r->op = GGML_OP_SET_ROWS; r->src[0] = b; r->src[1] = c;
    return r;
}

struct ggml_tensor * ggml_soft_max_ext(struct ggml_context * ctx, struct ggml_tensor * a, struct ggml_tensor * mask, float scale, float max_bias) {
    GGML_ASSERT(ggml_is_contiguous(a));
    if (mask) {
        GGML_ASSERT(mask->type == GGML_TYPE_F16 || mask->type == GGML_TYPE_F32);
        GGML_ASSERT(ggml_is_contiguous(mask));
        GGML_ASSERT(mask->ne[0] == a->ne[0]);
        GGML_ASSERT(mask->ne[1] >= a->ne[1]);
        GGML_ASSERT(a->ne[2] % mask->ne[2] == 0);
        GGML_ASSERT(a->ne[3] % mask->ne[3] == 0);
    }
    if (max_bias > 0.0f) GGML_ASSERT(mask);
    ggml_tensor * r = ggml_dup_tensor(ctx, a);
    set_f32(r, 0, scale); set_f32(r, 1, max_bias);
    r->op = GGML_OP_SOFT_MAX; r->src[0] = a; r->src[1] = mask;
    return r;
}
struct ggml_tensor * ggml_soft_max(struct ggml_context * ctx, struct ggml_tensor * a) { return ggml_soft_max_ext(ctx, a, NULL, 1.0f, 0.0f); }
void ggml_soft_max_add_sinks(struct ggml_tensor * a, struct ggml_tensor * sinks) {
    if (!sinks) { a->src[2] = NULL; return; }
    GGML_ASSERT(a->op == GGML_OP_SOFT_MAX);
    GGML_ASSERT(a->src[2] == NULL);
    GGML_ASSERT(a->src[0]->ne[2] == sinks->ne[0]);
    GGML_ASSERT(sinks->type == GGML_TYPE_F32);
    a->src[2] = sinks;
}

struct ggml_tensor * ggml_rope_ext(struct ggml_context * ctx, struct ggml_tensor * a, struct ggml_tensor * b, struct ggml_tensor * c,
        int n_dims, int mode, int n_ctx_orig, float freq_base, float freq_scale, float ext_factor, float attn_factor, float beta_fast, float beta_slow) {
    GGML_ASSERT((mode & 1) == 0 && "mode & 1 == 1 is no longer supported");
    GGML_ASSERT(b->ne[1] == 1 && b->ne[2] == 1 && b->ne[3] == 1); // ggml_is_vector
    GGML_ASSERT(b->type == GGML_TYPE_I32);
    GGML_ASSERT(a->ne[2] == b->ne[0]);
    if (c) {
        GGML_ASSERT(c->type == GGML_TYPE_F32);
        GGML_ASSERT(c->ne[0] >= n_dims / 2);
    }
    ggml_tensor * r = ggml_dup_tensor(ctx, a);
    r->op_params[0] = 0; /* n_past */ r->op_params[1] = n_dims; r->op_params[2] = mode; r->op_params[3] = 0; /* n_ctx */
    r->op_params[4] = n_ctx_orig;
    set_f32(r, 5, freq_base); set_f32(r, 6, freq_scale); set_f32(r, 7, ext_factor);
    set_f32(r, 8, attn_factor); set_f32(r, 9, beta_fast); set_f32(r, 10, beta_slow);
    r->op = GGML_OP_ROPE; r->src[0] = a; r->src[1] = b; r->src[2] = c;
    return r;
}

struct ggml_tensor * ggml_argsort(struct ggml_context * ctx, struct ggml_tensor * a, enum ggml_sort_order order) {
    ggml_tensor * r = ggml_new_tensor(ctx, GGML_TYPE_I32, GGML_MAX_DIMS, a->ne);
    r->op_params[0] = (int32_t) order;
    r->op = GGML_OP_ARGSORT; r->src[0] = a;
    return r;
}
struct ggml_tensor * ggml_top_k(struct ggml_context * ctx, struct ggml_tensor * a, int k) {
    GGML_ASSERT(a->ne[0] >= k);
    ggml_tensor * r = ggml_argsort(ctx, a, GGML_SORT_ORDER_DESC);
    return ggml_view_4d(ctx, r, k, r->ne[1], r->ne[2], r->ne[3], r->nb[1], r->nb[2], r->nb[3], 0);
}

struct ggml_tensor * ggml_glu_split(struct ggml_context * ctx, struct ggml_tensor * a, struct ggml_tensor * b, enum ggml_glu_op op) {
    GGML_ASSERT(ggml_is_contiguous_1(a) && ggml_is_contiguous_1(b));
    GGML_ASSERT(ggml_are_same_shape(a, b) && a->type == b->type);
    ggml_tensor * r = ggml_new_tensor(ctx, a->type, GGML_MAX_DIMS, a->ne);
    r->op_params[0] = (int32_t) op; r->op_params[1] = 0; /* swapped */
    r->op = GGML_OP_GLU; r->src[0] = a; r->src[1] = b;
    return r;
}
struct ggml_tensor * ggml_swiglu_split(struct ggml_context * ctx, struct ggml_tensor * a, struct ggml_tensor * b) {
    return ggml_glu_split(ctx, a, b, GGML_GLU_OP_SWIGLU);
}
struct ggml_tensor * ggml_swiglu_oai(struct ggml_context * ctx, struct ggml_tensor * a, struct ggml_tensor * b, float alpha, float limit) {
    ggml_tensor * r = ggml_glu_split(ctx, a, b, GGML_GLU_OP_SWIGLU_OAI);
    set_f32(r, 2, alpha); set_f32(r, 3, limit);
    return r;
}

struct ggml_tensor * ggml_unary(struct ggml_context * ctx, struct ggml_tensor * a, enum ggml_unary_op op) {
    GGML_ASSERT(ggml_is_contiguous_1(a));
    ggml_tensor * r = ggml_dup_tensor(ctx, a);
    r->op_params[0] = (int32_t) op;
    r->op = GGML_OP_UNARY; r->src[0] = a;
    return r;
}
struct ggml_tensor * ggml_silu   (struct ggml_context * ctx, struct ggml_tensor * a) { return ggml_unary(ctx, a, GGML_UNARY_OP_SILU); }
struct ggml_tensor * ggml_sigmoid(struct ggml_context * ctx, struct ggml_tensor * a) { return ggml_unary(ctx, a, GGML_UNARY_OP_SIGMOID); }

// ---------------------------------------------------------------------------
// graph
// ---------------------------------------------------------------------------
static size_t hash_ptr(const void * p, size_t size) { return (size_t)(((uintptr_t) p) >> 4) % size; }

static bool hash_insert(struct ggml_hash_set * hs, struct ggml_tensor * key) {
    size_t h = hash_ptr(key, hs->size), i = h;
    while (hs->used[i/32] & (1u << (i%32))) {
        if (hs->keys[i] == key) return false; // already present
        i = (i + 1) % hs->size;
        GGML_ASSERT(i != h && "graph hash set full");
    }
    hs->used[i/32] |= 1u << (i%32);
    hs->keys[i] = key;
    return true;
}

struct ggml_cgraph * ggml_new_graph_custom(struct ggml_context * ctx, size_t size, bool grads) {
    GGML_ASSERT(!grads);
    ggml_cgraph * g = (ggml_cgraph *) calloc(1, sizeof(ggml_cgraph));
    g->size = (int) size;
    g->nodes = (ggml_tensor **) calloc(size, sizeof(void *));
    g->leafs = (ggml_tensor **) calloc(size, sizeof(void *));
    g->visited_hash_set.size = size*2 + 1;
    g->visited_hash_set.used = (uint32_t *) calloc(g->visited_hash_set.size/32 + 1, 4);
    g->visited_hash_set.keys = (ggml_tensor **) calloc(g->visited_hash_set.size, sizeof(void *));
    g->use_counts = (int32_t *) calloc(g->visited_hash_set.size, 4);
    g->order = GGML_CGRAPH_EVAL_ORDER_LEFT_TO_RIGHT;
    ctx->blobs.push_back(g);
    ctx->blobs.push_back(g->nodes);
    ctx->blobs.push_back(g->leafs);
    ctx->blobs.push_back(g->visited_hash_set.used);
    ctx->blobs.push_back(g->visited_hash_set.keys);
    ctx->blobs.push_back(g->use_counts);
    return g;
}
struct ggml_cgraph * ggml_new_graph(struct ggml_context * ctx) { return ggml_new_graph_custom(ctx, GGML_DEFAULT_GRAPH_SIZE, false); }

static void visit_parents(struct ggml_cgraph * g, struct ggml_tensor * node) {
    if (!hash_insert(&g->visited_hash_set, node)) return;
    for (int i = 0; i < GGML_MAX_SRC; ++i) {
        if (node->src[i]) visit_parents(g, node->src[i]);
    }
    if (node->op == GGML_OP_NONE && !(node->flags & GGML_TENSOR_FLAG_PARAM)) {
        GGML_ASSERT(g->n_leafs < g->size);
        g->leafs[g->n_leafs++] = node;
    } else {
        GGML_ASSERT(g->n_nodes < g->size);
        g->nodes[g->n_nodes++] = node;
    }
}
void ggml_build_forward_expand(struct ggml_cgraph * g, struct ggml_tensor * tensor) { visit_parents(g, tensor); }
int  ggml_graph_n_nodes(struct ggml_cgraph * g) { return g->n_nodes; }
struct ggml_tensor * ggml_graph_node(struct ggml_cgraph * g, int i) {
    if (i < 0) { GGML_ASSERT(g->n_nodes + i >= 0); return g->nodes[g->n_nodes + i]; }
    GGML_ASSERT(i < g->n_nodes);
    return g->nodes[i];
}
void ggml_graph_clear(struct ggml_cgraph * g) {
    g->n_nodes = 0; g->n_leafs = 0;
    memset(g->visited_hash_set.used, 0, (g->visited_hash_set.size/32 + 1)*4);
}

// ---------------------------------------------------------------------------
// ggml-backend public wrappers (dispatch through the vtables)
// ---------------------------------------------------------------------------
const char * ggml_backend_buft_name(ggml_backend_buffer_type_t buft) { return buft->iface.get_name(buft); }
ggml_backend_buffer_t ggml_backend_buft_alloc_buffer(ggml_backend_buffer_type_t buft, size_t size) {
    if (size == 0) {
        // return a dummy buffer for zero-sized allocations (src/llama-model.cpp:280 relies on this)
        struct ggml_backend_buffer_i none = {};
        return ggml_backend_buffer_init(buft, none, NULL, 0);
    }
    return buft->iface.alloc_buffer(buft, size);
}
size_t ggml_backend_buft_get_alignment(ggml_backend_buffer_type_t buft) { return buft->iface.get_alignment(buft); }
size_t ggml_backend_buft_get_max_size(ggml_backend_buffer_type_t buft) {
    return buft->iface.get_max_size ? buft->iface.get_max_size(buft) : SIZE_MAX;
}
size_t ggml_backend_buft_get_alloc_size(ggml_backend_buffer_type_t buft, const struct ggml_tensor * tensor) {
    if (buft->iface.get_alloc_size) {
        size_t size = buft->iface.get_alloc_size(buft, tensor);
        GGML_ASSERT(size >= ggml_nbytes(tensor));
        return size;
    }
    return ggml_nbytes(tensor);
}
bool ggml_backend_buft_is_host(ggml_backend_buffer_type_t buft) { return buft->iface.is_host ? buft->iface.is_host(buft) : false; }
ggml_backend_dev_t ggml_backend_buft_get_device(ggml_backend_buffer_type_t buft) { return buft->device; }

ggml_backend_buffer_t ggml_backend_buffer_init(ggml_backend_buffer_type_t buft, struct ggml_backend_buffer_i iface, void * context, size_t size) {
    ggml_backend_buffer_t buffer = new ggml_backend_buffer{ iface, buft, context, size, GGML_BACKEND_BUFFER_USAGE_ANY };
    return buffer;
}
const char * ggml_backend_buffer_name(ggml_backend_buffer_t buffer) { return ggml_backend_buft_name(buffer->buft); }
void ggml_backend_buffer_free(ggml_backend_buffer_t buffer) {
    if (!buffer) return;
    if (buffer->iface.free_buffer) buffer->iface.free_buffer(buffer);
    delete buffer;
}
size_t ggml_backend_buffer_get_size(ggml_backend_buffer_t buffer) { return buffer->size; }
void * ggml_backend_buffer_get_base(ggml_backend_buffer_t buffer) {
    if (buffer->size == 0) return NULL;
    void * base = buffer->iface.get_base(buffer);
    GGML_ASSERT(base != NULL && "backend buffer base cannot be NULL");
    return base;
}
enum ggml_status ggml_backend_buffer_init_tensor(ggml_backend_buffer_t buffer, struct ggml_tensor * tensor) {
    if (buffer->iface.init_tensor) return buffer->iface.init_tensor(buffer, tensor);
    return GGML_STATUS_SUCCESS;
}
void ggml_backend_buffer_clear(ggml_backend_buffer_t buffer, uint8_t value) {
    if (buffer->size == 0) return;
    buffer->iface.clear(buffer, value);
}
size_t ggml_backend_buffer_get_alignment(ggml_backend_buffer_t buffer) { return ggml_backend_buft_get_alignment(buffer->buft); }
size_t ggml_backend_buffer_get_alloc_size(ggml_backend_buffer_t buffer, const struct ggml_tensor * tensor) {
    return ggml_backend_buft_get_alloc_size(buffer->buft, tensor);
}
bool ggml_backend_buffer_is_host(ggml_backend_buffer_t buffer) { return ggml_backend_buft_is_host(buffer->buft); }
void ggml_backend_buffer_set_usage(ggml_backend_buffer_t buffer, enum ggml_backend_buffer_usage usage) { buffer->usage = usage; }
enum ggml_backend_buffer_usage ggml_backend_buffer_get_usage(ggml_backend_buffer_t buffer) { return buffer->usage; }
ggml_backend_buffer_type_t ggml_backend_buffer_get_type(ggml_backend_buffer_t buffer) { return buffer->buft; }
void ggml_backend_buffer_reset(ggml_backend_buffer_t buffer) { if (buffer->iface.reset) buffer->iface.reset(buffer); }

bool ggml_backend_buffer_copy_tensor(const struct ggml_tensor * src, struct ggml_tensor * dst) {
    ggml_backend_buffer_t dst_buf = dst->view_src ? dst->view_src->buffer : dst->buffer;
    if (dst_buf->iface.cpy_tensor) return dst_buf->iface.cpy_tensor(dst_buf, src, dst);
    return false;
}

enum ggml_status ggml_backend_tensor_alloc(ggml_backend_buffer_t buffer, struct ggml_tensor * tensor, void * addr) {
    GGML_ASSERT(tensor->buffer == NULL && tensor->data == NULL && tensor->view_src == NULL);
    GGML_ASSERT(addr >= ggml_backend_buffer_get_base(buffer));
    GGML_ASSERT((char *) addr + ggml_backend_buffer_get_alloc_size(buffer, tensor) <=
                (char *) ggml_backend_buffer_get_base(buffer) + ggml_backend_buffer_get_size(buffer));
    tensor->buffer = buffer;
    tensor->data = addr;
    return ggml_backend_buffer_init_tensor(buffer, tensor);
}
enum ggml_status ggml_backend_view_init(struct ggml_tensor * tensor) {
    GGML_ASSERT(tensor->buffer == NULL && tensor->view_src != NULL);
    GGML_ASSERT(tensor->view_src->buffer != NULL && tensor->view_src->data != NULL);
    tensor->buffer = tensor->view_src->buffer;
    tensor->data = (char *) tensor->view_src->data + tensor->view_offs;
    return ggml_backend_buffer_init_tensor(tensor->buffer, tensor);
}

ggml_guid_t  ggml_backend_guid(ggml_backend_t backend) { return backend ? backend->guid : NULL; }
const char * ggml_backend_name(ggml_backend_t backend) { return backend ? backend->iface.get_name(backend) : "NULL"; }
void         ggml_backend_free(ggml_backend_t backend) { if (backend) backend->iface.free(backend); }
ggml_backend_buffer_type_t ggml_backend_get_default_buffer_type(ggml_backend_t backend) { return ggml_backend_dev_buffer_type(backend->device); }
ggml_backend_buffer_t ggml_backend_alloc_buffer(ggml_backend_t backend, size_t size) {
    return ggml_backend_buft_alloc_buffer(ggml_backend_get_default_buffer_type(backend), size);
}
size_t ggml_backend_get_alignment(ggml_backend_t backend) { return ggml_backend_buft_get_alignment(ggml_backend_get_default_buffer_type(backend)); }
size_t ggml_backend_get_max_size(ggml_backend_t backend) { return ggml_backend_buft_get_max_size(ggml_backend_get_default_buffer_type(backend)); }

void ggml_backend_tensor_set_async(ggml_backend_t backend, struct ggml_tensor * tensor, const void * data, size_t offset, size_t size) {
    GGML_ASSERT(tensor->data != NULL && "tensor not allocated");
    GGML_ASSERT(offset + size <= ggml_nbytes(tensor) && "tensor write out of bounds");
    if (backend->iface.set_tensor_async == NULL) ggml_backend_tensor_set(tensor, data, offset, size);
    else backend->iface.set_tensor_async(backend, tensor, data, offset, size);
}
void ggml_backend_tensor_get_async(ggml_backend_t backend, const struct ggml_tensor * tensor, void * data, size_t offset, size_t size) {
    GGML_ASSERT(tensor->data != NULL && "tensor not allocated");
    GGML_ASSERT(offset + size <= ggml_nbytes(tensor) && "tensor read out of bounds");
    if (backend->iface.get_tensor_async == NULL) ggml_backend_tensor_get(tensor, data, offset, size);
    else backend->iface.get_tensor_async(backend, tensor, data, offset, size);
}
void ggml_backend_tensor_set(struct ggml_tensor * tensor, const void * data, size_t offset, size_t size) {
    ggml_backend_buffer_t buf = tensor->view_src ? tensor->view_src->buffer : tensor->buffer;
    if (size == 0) return;
    GGML_ASSERT(buf != NULL && "tensor buffer not set");
    GGML_ASSERT(tensor->data != NULL && "tensor not allocated");
    GGML_ASSERT(offset + size <= ggml_nbytes(tensor) && "tensor write out of bounds");
    buf->iface.set_tensor(buf, tensor, data, offset, size);
}
void ggml_backend_tensor_get(const struct ggml_tensor * tensor, void * data, size_t offset, size_t size) {
    ggml_backend_buffer_t buf = tensor->view_src ? tensor->view_src->buffer : tensor->buffer;
    if (size == 0) return;
    GGML_ASSERT(buf != NULL && "tensor buffer not set");
    GGML_ASSERT(tensor->data != NULL && "tensor not allocated");
    GGML_ASSERT(offset + size <= ggml_nbytes(tensor) && "tensor read out of bounds");
    buf->iface.get_tensor(buf, tensor, data, offset, size);
}
void ggml_backend_tensor_memset(struct ggml_tensor * tensor, uint8_t value, size_t offset, size_t size) {
    ggml_backend_buffer_t buf = tensor->view_src ? tensor->view_src->buffer : tensor->buffer;
    if (size == 0) return;
    GGML_ASSERT(buf != NULL && tensor->data != NULL);
    GGML_ASSERT(offset + size <= ggml_nbytes(tensor) && "tensor write out of bounds");
    GGML_ASSERT(buf->iface.memset_tensor != NULL && "memset not implemented by backend buffer");
    buf->iface.memset_tensor(buf, tensor, value, offset, size);
}

void ggml_backend_synchronize(ggml_backend_t backend) { if (backend->iface.synchronize) backend->iface.synchronize(backend); }
enum ggml_status ggml_backend_graph_compute_async(ggml_backend_t backend, struct ggml_cgraph * cgraph) {
    return backend->iface.graph_compute(backend, cgraph);
}
enum ggml_status ggml_backend_graph_compute(ggml_backend_t backend, struct ggml_cgraph * cgraph) {
    enum ggml_status err = ggml_backend_graph_compute_async(backend, cgraph);
    ggml_backend_synchronize(backend);
    return err;
}
bool ggml_backend_supports_op  (ggml_backend_t backend, const struct ggml_tensor * op) { return ggml_backend_dev_supports_op(backend->device, op); }
bool ggml_backend_supports_buft(ggml_backend_t backend, ggml_backend_buffer_type_t buft) { return ggml_backend_dev_supports_buft(backend->device, buft); }
bool ggml_backend_offload_op   (ggml_backend_t backend, const struct ggml_tensor * op) { return ggml_backend_dev_offload_op(backend->device, op); }
ggml_backend_dev_t ggml_backend_get_device(ggml_backend_t backend) { return backend->device; }

void ggml_backend_tensor_copy(struct ggml_tensor * src, struct ggml_tensor * dst) {
    GGML_ASSERT(ggml_nbytes(src) == ggml_nbytes(dst) && "cannot copy tensors with different layouts");
    if (src == dst) return;
    if (ggml_backend_buffer_is_host(src->buffer)) {
        ggml_backend_tensor_set(dst, src->data, 0, ggml_nbytes(src));
    } else if (ggml_backend_buffer_is_host(dst->buffer)) {
        ggml_backend_tensor_get(src, dst->data, 0, ggml_nbytes(src));
    } else if (!ggml_backend_buffer_copy_tensor(src, dst)) {
        size_t nbytes = ggml_nbytes(src);
        void * data = malloc(nbytes);
        ggml_backend_tensor_get(src, data, 0, nbytes);
        ggml_backend_tensor_set(dst, data, 0, nbytes);
        free(data);
    }
}
void ggml_backend_tensor_copy_async(ggml_backend_t backend_src, ggml_backend_t backend_dst, struct ggml_tensor * src, struct ggml_tensor * dst) {
    GGML_ASSERT(ggml_nbytes(src) == ggml_nbytes(dst) && "cannot copy tensors with different layouts");
    if (src == dst) return;
    if (backend_dst->iface.cpy_tensor_async != NULL) {
        if (backend_dst->iface.cpy_tensor_async(backend_src, backend_dst, src, dst)) return;
    }
    // an async copy would normally happen after all the queued operations on both backends are completed
    ggml_backend_synchronize(backend_src);
    ggml_backend_synchronize(backend_dst);
    ggml_backend_tensor_copy(src, dst);
}

ggml_backend_event_t ggml_backend_event_new(ggml_backend_dev_t device) {
    if (device->iface.event_new == NULL) return NULL;
    return device->iface.event_new(device);
}
void ggml_backend_event_free(ggml_backend_event_t event) { if (event) event->device->iface.event_free(event->device, event); }
void ggml_backend_event_record(ggml_backend_event_t event, ggml_backend_t backend) {
    GGML_ASSERT(backend->iface.event_record != NULL);
    backend->iface.event_record(backend, event);
}
void ggml_backend_event_synchronize(ggml_backend_event_t event) {
    GGML_ASSERT(event->device->iface.event_synchronize);
    event->device->iface.event_synchronize(event->device, event);
}
void ggml_backend_event_wait(ggml_backend_t backend, ggml_backend_event_t event) {
    GGML_ASSERT(backend->iface.event_wait != NULL);
    backend->iface.event_wait(backend, event);
}

const char * ggml_backend_dev_name(ggml_backend_dev_t d) { return d->iface.get_name(d); }
const char * ggml_backend_dev_description(ggml_backend_dev_t d) { return d->iface.get_description(d); }
void ggml_backend_dev_memory(ggml_backend_dev_t d, size_t * free, size_t * total) { d->iface.get_memory(d, free, total); }
enum ggml_backend_dev_type ggml_backend_dev_type(ggml_backend_dev_t d) { return d->iface.get_type(d); }
void ggml_backend_dev_get_props(ggml_backend_dev_t d, struct ggml_backend_dev_props * props) {
    memset(props, 0, sizeof(*props));
    d->iface.get_props(d, props);
}
ggml_backend_reg_t ggml_backend_dev_backend_reg(ggml_backend_dev_t d) { return d->reg; }
ggml_backend_t ggml_backend_dev_init(ggml_backend_dev_t d, const char * params) { return d->iface.init_backend(d, params); }
ggml_backend_buffer_type_t ggml_backend_dev_buffer_type(ggml_backend_dev_t d) { return d->iface.get_buffer_type(d); }
ggml_backend_buffer_type_t ggml_backend_dev_host_buffer_type(ggml_backend_dev_t d) {
    return d->iface.get_host_buffer_type ? d->iface.get_host_buffer_type(d) : NULL;
}
ggml_backend_buffer_t ggml_backend_dev_buffer_from_host_ptr(ggml_backend_dev_t d, void * ptr, size_t size, size_t max_tensor_size) {
    return d->iface.buffer_from_host_ptr ? d->iface.buffer_from_host_ptr(d, ptr, size, max_tensor_size) : NULL;
}
bool ggml_backend_dev_supports_op(ggml_backend_dev_t d, const struct ggml_tensor * op) { return d->iface.supports_op(d, op); }
bool ggml_backend_dev_supports_buft(ggml_backend_dev_t d, ggml_backend_buffer_type_t buft) { return d->iface.supports_buft(d, buft); }
bool ggml_backend_dev_offload_op(ggml_backend_dev_t d, const struct ggml_tensor * op) {
    return d->iface.offload_op ? d->iface.offload_op(d, op) : false;
}

const char * ggml_backend_reg_name(ggml_backend_reg_t reg) { return reg->iface.get_name(reg); }
size_t ggml_backend_reg_dev_count(ggml_backend_reg_t reg) { return reg->iface.get_device_count(reg); }
ggml_backend_dev_t ggml_backend_reg_dev_get(ggml_backend_reg_t reg, size_t index) { return reg->iface.get_device(reg, index); }
void * ggml_backend_reg_get_proc_address(ggml_backend_reg_t reg, const char * name) {
    return reg->iface.get_proc_address ? reg->iface.get_proc_address(reg, name) : NULL;
}

// ---- registry ---------------------------------------------------------------
static std::vector<ggml_backend_reg_t> & regs() { static std::vector<ggml_backend_reg_t> r; return r; }
static std::vector<ggml_backend_dev_t> & devs() { static std::vector<ggml_backend_dev_t> d; return d; }

void ggml_backend_register(ggml_backend_reg_t reg) {
    if (!reg) return;
    for (auto * r : regs()) if (r == reg) return;
    regs().push_back(reg);
    for (size_t i = 0; i < ggml_backend_reg_dev_count(reg); i++) devs().push_back(ggml_backend_reg_dev_get(reg, i));
}
size_t ggml_backend_reg_count(void) { return regs().size(); }
ggml_backend_reg_t ggml_backend_reg_get(size_t i) { GGML_ASSERT(i < regs().size()); return regs()[i]; }
ggml_backend_reg_t ggml_backend_reg_by_name(const char * name) {
    for (auto * r : regs()) if (strcasecmp(ggml_backend_reg_name(r), name) == 0) return r;
    return NULL;
}
size_t ggml_backend_dev_count(void) { return devs().size(); }
ggml_backend_dev_t ggml_backend_dev_get(size_t i) { GGML_ASSERT(i < devs().size()); return devs()[i]; }
ggml_backend_dev_t ggml_backend_dev_by_name(const char * name) {
    for (auto * d : devs()) if (strcasecmp(ggml_backend_dev_name(d), name) == 0) return d;
    return NULL;
}

// the dynamic-backend contract (docs/build.md:613): dlopen, optional ggml_backend_score, ggml_backend_init
ggml_backend_reg_t ggml_backend_load(const char * path) {
    void * h = dlopen(path, RTLD_NOW | RTLD_LOCAL);
    if (!h) { fprintf(stderr, "ggml_backend_load: dlopen(%s) failed: %s\n", path, dlerror()); return NULL; }
    ggml_backend_score_t score_fn = (ggml_backend_score_t) dlsym(h, "ggml_backend_score");
    if (score_fn && score_fn() == 0) {
        fprintf(stderr, "ggml_backend_load: backend %s is not supported on this system\n", path);
        dlclose(h);
        return NULL;
    }
    ggml_backend_init_t init_fn = (ggml_backend_init_t) dlsym(h, "ggml_backend_init");
    if (!init_fn) { fprintf(stderr, "ggml_backend_load: %s does not export ggml_backend_init\n", path); dlclose(h); return NULL; }
    ggml_backend_reg_t reg = init_fn();
    if (!reg || reg->api_version != GGML_BACKEND_API_VERSION) {
        fprintf(stderr, "ggml_backend_load: %s: api version mismatch (%d vs %d)\n", path, reg ? reg->api_version : -1, GGML_BACKEND_API_VERSION);
        dlclose(h);
        return NULL;
    }
    ggml_backend_register(reg);
    return reg;
}

// ---- sequential allocator (stands in for ggml-alloc's ctx allocation) --------
struct ggml_backend_buffer * ggml_backend_alloc_ctx_tensors_from_buft(struct ggml_context * ctx, ggml_backend_buffer_type_t buft) {
    const size_t align = ggml_backend_buft_get_alignment(buft);
    size_t total = 0;
    for (auto * t : ctx->tensors) {
        if (t->data == NULL && t->view_src == NULL) {
            total += GGML_PAD(ggml_backend_buft_get_alloc_size(buft, t), align);
        }
    }
    if (total == 0) return NULL;
    ggml_backend_buffer_t buffer = ggml_backend_buft_alloc_buffer(buft, total);
    if (!buffer) return NULL;
    char * base = (char *) ggml_backend_buffer_get_base(buffer);
    size_t off = 0;
    for (auto * t : ctx->tensors) {
        if (t->data == NULL) {
            if (t->view_src == NULL) {
                ggml_backend_tensor_alloc(buffer, t, base + off);
                off += GGML_PAD(ggml_backend_buft_get_alloc_size(buft, t), align);
            } else if (t->buffer == NULL) {
                ggml_backend_view_init(t);
            }
        } else if (t->view_src != NULL && t->buffer == NULL) {
            ggml_backend_view_init(t);
        }
    }
    return buffer;
}
struct ggml_backend_buffer * ggml_backend_alloc_ctx_tensors(struct ggml_context * ctx, ggml_backend_t backend) {
    return ggml_backend_alloc_ctx_tensors_from_buft(ctx, ggml_backend_get_default_buffer_type(backend));
}

} // extern "C"
