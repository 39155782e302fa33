// ggml-compat/ggml.h — restatement of the slice of upstream ggml's public header (ggml/include/ggml.h) that the MI355X backend's hot
// path touches.
//
// ATTRIBUTION: the declarations below restate the public interface of ggml (https://github.com/ggml-org/ggml, the tensor library of
// llama.cpp; Copyright (c) 2023-2025 The ggml authors, MIT License). Struct field order, enum values, function names and signatures — and
// several field comments — are upstream ggml's, as an ABI restatement requires; they are reproduced here from knowledge of that interface at
// the sync point named below, because the reference tree carries no copy of it. MIT License notice: "Permission is hereby granted, free of
// charge, to any person obtaining a copy of this software and associated documentation files, to deal in the Software without restriction
// ... The above copyright notice and this permission notice shall be included in all copies or substantial portions of the Software."
//
// WHY THIS FILE EXISTS: the reference tree's `ggml/` is an empty, un-vendored
// submodule (/root/reference/.gitmodules:1-3), so the real ggml.h is absent.
// Every declaration below is restated from usage in the reference
// (tests/test-backend-ops.cpp, src/llama-graph.cpp, src/llama-model.cpp, …)
// plus [UPSTREAM-KNOWLEDGE] of ggml at the sync point
// scripts/sync-ggml.last:1 (llama.cpp build 6174). Field ORDER and enum VALUES
// are ABI: when a real ggml checkout is available, build the backend with
// -I<ggml>/include -I<ggml>/src instead of -Iinclude/ggml-compat and this
// directory drops out.
//
// Type ids are pinned by gguf-py/gguf/constants.py:2698-2730 (citeable);
// block/type sizes by gguf-py/gguf/constants.py:2839-2872.
#pragma once

#include <stdbool.h>
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#if defined(__GNUC__)
#    define GGML_API __attribute__((visibility("default")))
#else
#    define GGML_API
#endif

#define GGML_MAX_DIMS       4
#define GGML_MAX_SRC        10
#define GGML_MAX_NAME       64
#define GGML_MAX_OP_PARAMS  64
#define GGML_DEFAULT_GRAPH_SIZE 2048
#define GGML_MEM_ALIGN      16

#define GGML_KQ_MASK_PAD    64
#define QK_K                256

#define GGML_UNUSED(x) (void)(x)
#define GGML_PAD(x, n) (((x) + (n) - 1) & ~((n) - 1))

GGML_API void ggml_abort(const char * file, int line, const char * fmt, ...);
#define GGML_ABORT(...) ggml_abort(__FILE__, __LINE__, __VA_ARGS__)
#define GGML_ASSERT(x) do { if (!(x)) GGML_ABORT("GGML_ASSERT(%s) failed", #x); } while (0)

enum ggml_status {
    GGML_STATUS_ALLOC_FAILED = -2,
    GGML_STATUS_FAILED       = -1,
    GGML_STATUS_SUCCESS      = 0,
    GGML_STATUS_ABORTED      = 1,
};

typedef uint16_t ggml_fp16_t;
typedef struct { uint16_t bits; } ggml_bf16_t;

// gguf-py/gguf/constants.py:2698-2730
enum ggml_type {
    GGML_TYPE_F32     = 0,
    GGML_TYPE_F16     = 1,
    GGML_TYPE_Q4_0    = 2,
    GGML_TYPE_Q4_1    = 3,
    GGML_TYPE_Q5_0    = 6,
    GGML_TYPE_Q5_1    = 7,
    GGML_TYPE_Q8_0    = 8,
    GGML_TYPE_Q8_1    = 9,
    GGML_TYPE_Q2_K    = 10,
    GGML_TYPE_Q3_K    = 11,
    GGML_TYPE_Q4_K    = 12,
    GGML_TYPE_Q5_K    = 13,
    GGML_TYPE_Q6_K    = 14,
    GGML_TYPE_Q8_K    = 15,
    GGML_TYPE_IQ2_XXS = 16,
    GGML_TYPE_IQ2_XS  = 17,
    GGML_TYPE_IQ3_XXS = 18,
    GGML_TYPE_IQ1_S   = 19,
    GGML_TYPE_IQ4_NL  = 20,
    GGML_TYPE_IQ3_S   = 21,
    GGML_TYPE_IQ2_S   = 22,
    GGML_TYPE_IQ4_XS  = 23,
    GGML_TYPE_I8      = 24,
    GGML_TYPE_I16     = 25,
    GGML_TYPE_I32     = 26,
    GGML_TYPE_I64     = 27,
    GGML_TYPE_F64     = 28,
    GGML_TYPE_IQ1_M   = 29,
    GGML_TYPE_BF16    = 30,
    GGML_TYPE_TQ1_0   = 34,
    GGML_TYPE_TQ2_0   = 35,
    GGML_TYPE_MXFP4   = 39,
    GGML_TYPE_COUNT   = 40,
};

enum ggml_prec {
    GGML_PREC_DEFAULT = 0,
    GGML_PREC_F32     = 10,
};

// [UPSTREAM-KNOWLEDGE] op order at the sync point. Presence of every op named
// here is corroborated by tests/test-backend-ops.cpp; the ORDER is not
// citeable in-tree and must be checked against a real ggml.h.
enum ggml_op {
    GGML_OP_NONE = 0,

    GGML_OP_DUP,
    GGML_OP_ADD,
    GGML_OP_ADD_ID,
    GGML_OP_ADD1,
    GGML_OP_ACC,
    GGML_OP_SUB,
    GGML_OP_MUL,
    GGML_OP_DIV,
    GGML_OP_SQR,
    GGML_OP_SQRT,
    GGML_OP_LOG,
    GGML_OP_SIN,
    GGML_OP_COS,
    GGML_OP_SUM,
    GGML_OP_SUM_ROWS,
    GGML_OP_MEAN,
    GGML_OP_ARGMAX,
    GGML_OP_COUNT_EQUAL,
    GGML_OP_REPEAT,
    GGML_OP_REPEAT_BACK,
    GGML_OP_CONCAT,
    GGML_OP_SILU_BACK,
    GGML_OP_NORM,
    GGML_OP_RMS_NORM,
    GGML_OP_RMS_NORM_BACK,
    GGML_OP_GROUP_NORM,
    GGML_OP_L2_NORM,

    GGML_OP_MUL_MAT,
    GGML_OP_MUL_MAT_ID,
    GGML_OP_OUT_PROD,

    GGML_OP_SCALE,
    GGML_OP_SET,
    GGML_OP_CPY,
    GGML_OP_CONT,
    GGML_OP_RESHAPE,
    GGML_OP_VIEW,
    GGML_OP_PERMUTE,
    GGML_OP_TRANSPOSE,
    GGML_OP_GET_ROWS,
    GGML_OP_GET_ROWS_BACK,
    GGML_OP_SET_ROWS,
    GGML_OP_DIAG,
    GGML_OP_DIAG_MASK_INF,
    GGML_OP_DIAG_MASK_ZERO,
    GGML_OP_SOFT_MAX,
    GGML_OP_SOFT_MAX_BACK,
    GGML_OP_ROPE,
    GGML_OP_ROPE_BACK,
    GGML_OP_CLAMP,
    GGML_OP_CONV_TRANSPOSE_1D,
    GGML_OP_IM2COL,
    GGML_OP_IM2COL_BACK,
    GGML_OP_CONV_2D,
    GGML_OP_CONV_2D_DW,
    GGML_OP_CONV_TRANSPOSE_2D,
    GGML_OP_POOL_1D,
    GGML_OP_POOL_2D,
    GGML_OP_POOL_2D_BACK,
    GGML_OP_UPSCALE,
    GGML_OP_PAD,
    GGML_OP_PAD_REFLECT_1D,
    GGML_OP_ROLL,
    GGML_OP_ARANGE,
    GGML_OP_TIMESTEP_EMBEDDING,
    GGML_OP_ARGSORT,
    GGML_OP_LEAKY_RELU,

    GGML_OP_FLASH_ATTN_EXT,
    GGML_OP_FLASH_ATTN_BACK,
    GGML_OP_SSM_CONV,
    GGML_OP_SSM_SCAN,
    GGML_OP_WIN_PART,
    GGML_OP_WIN_UNPART,
    GGML_OP_GET_REL_POS,
    GGML_OP_ADD_REL_POS,
    GGML_OP_RWKV_WKV6,
    GGML_OP_GATED_LINEAR_ATTN,
    GGML_OP_RWKV_WKV7,

    GGML_OP_UNARY,

    GGML_OP_MAP_CUSTOM1,
    GGML_OP_MAP_CUSTOM2,
    GGML_OP_MAP_CUSTOM3,

    GGML_OP_CUSTOM,

    GGML_OP_CROSS_ENTROPY_LOSS,
    GGML_OP_CROSS_ENTROPY_LOSS_BACK,
    GGML_OP_OPT_STEP_ADAMW,
    GGML_OP_OPT_STEP_SGD,

    GGML_OP_GLU,

    GGML_OP_COUNT,
};

enum ggml_unary_op {
    GGML_UNARY_OP_ABS,
    GGML_UNARY_OP_SGN,
    GGML_UNARY_OP_NEG,
    GGML_UNARY_OP_STEP,
    GGML_UNARY_OP_TANH,
    GGML_UNARY_OP_ELU,
    GGML_UNARY_OP_RELU,
    GGML_UNARY_OP_SIGMOID,
    GGML_UNARY_OP_GELU,
    GGML_UNARY_OP_GELU_QUICK,
    GGML_UNARY_OP_SILU,
    GGML_UNARY_OP_HARDSWISH,
    GGML_UNARY_OP_HARDSIGMOID,
    GGML_UNARY_OP_EXP,
    GGML_UNARY_OP_GELU_ERF,

    GGML_UNARY_OP_COUNT,
};

enum ggml_glu_op {
    GGML_GLU_OP_REGLU,
    GGML_GLU_OP_GEGLU,
    GGML_GLU_OP_SWIGLU,
    GGML_GLU_OP_SWIGLU_OAI,
    GGML_GLU_OP_GEGLU_ERF,
    GGML_GLU_OP_GEGLU_QUICK,

    GGML_GLU_OP_COUNT,
};

enum ggml_sort_order {
    GGML_SORT_ORDER_ASC,
    GGML_SORT_ORDER_DESC,
};

enum ggml_tensor_flag {
    GGML_TENSOR_FLAG_INPUT  =  1,
    GGML_TENSOR_FLAG_OUTPUT =  2,
    GGML_TENSOR_FLAG_PARAM  =  4,
    GGML_TENSOR_FLAG_LOSS   =  8,
};

#define GGML_ROPE_TYPE_NEOX   2
#define GGML_ROPE_TYPE_MROPE  8
#define GGML_ROPE_TYPE_VISION 24

struct ggml_backend_buffer;
struct ggml_context;
struct ggml_cgraph;

// field order = ABI
struct ggml_tensor {
    enum ggml_type type;

    struct ggml_backend_buffer * buffer;

    int64_t ne[GGML_MAX_DIMS]; // number of elements
    size_t  nb[GGML_MAX_DIMS]; // stride in bytes:
                               // nb[0] = ggml_type_size(type)
                               // nb[1] = nb[0]   * (ne[0] / ggml_blck_size(type)) + padding
                               // nb[i] = nb[i-1] * ne[i-1]

    enum ggml_op op;
    int32_t op_params[GGML_MAX_OP_PARAMS / sizeof(int32_t)];

    int32_t flags;

    struct ggml_tensor * src[GGML_MAX_SRC];

    struct ggml_tensor * view_src;
    size_t               view_offs;

    void * data;

    char name[GGML_MAX_NAME];

    void * extra;

    char padding[8];
};

struct ggml_init_params {
    size_t mem_size;
    void * mem_buffer;
    bool   no_alloc;
};

typedef uint8_t ggml_guid[16];
typedef ggml_guid * ggml_guid_t;
GGML_API bool ggml_guid_matches(ggml_guid_t guid_a, ggml_guid_t guid_b);

// ---- type traits -----------------------------------------------------------
GGML_API int64_t      ggml_blck_size(enum ggml_type type);
GGML_API size_t       ggml_type_size(enum ggml_type type);
GGML_API size_t       ggml_row_size (enum ggml_type type, int64_t ne);
GGML_API const char * ggml_type_name(enum ggml_type type);
GGML_API bool         ggml_is_quantized(enum ggml_type type);
GGML_API const char * ggml_op_name  (enum ggml_op op);
GGML_API const char * ggml_op_desc  (const struct ggml_tensor * t);
GGML_API const char * ggml_status_to_string(enum ggml_status status);

GGML_API int64_t ggml_nelements(const struct ggml_tensor * tensor);
GGML_API int64_t ggml_nrows    (const struct ggml_tensor * tensor);
GGML_API size_t  ggml_nbytes   (const struct ggml_tensor * tensor);
GGML_API size_t  ggml_element_size(const struct ggml_tensor * tensor);
GGML_API int     ggml_n_dims   (const struct ggml_tensor * tensor);

GGML_API bool ggml_is_transposed (const struct ggml_tensor * tensor);
GGML_API bool ggml_is_permuted   (const struct ggml_tensor * tensor);
GGML_API bool ggml_is_empty      (const struct ggml_tensor * tensor);
GGML_API bool ggml_is_contiguous (const struct ggml_tensor * tensor);
GGML_API bool ggml_is_contiguous_0(const struct ggml_tensor * tensor);
GGML_API bool ggml_is_contiguous_1(const struct ggml_tensor * tensor); // contiguous for dims >= 1
GGML_API bool ggml_is_contiguous_2(const struct ggml_tensor * tensor); // contiguous for dims >= 2
GGML_API bool ggml_is_contiguously_allocated(const struct ggml_tensor * tensor);
GGML_API bool ggml_is_contiguous_rows(const struct ggml_tensor * tensor);
GGML_API bool ggml_are_same_shape (const struct ggml_tensor * t0, const struct ggml_tensor * t1);
GGML_API bool ggml_are_same_stride(const struct ggml_tensor * t0, const struct ggml_tensor * t1);
GGML_API bool ggml_can_repeat(const struct ggml_tensor * t0, const struct ggml_tensor * t1);

GGML_API enum ggml_unary_op ggml_get_unary_op(const struct ggml_tensor * tensor);
GGML_API enum ggml_glu_op   ggml_get_glu_op  (const struct ggml_tensor * tensor);

GGML_API float       ggml_fp16_to_fp32(ggml_fp16_t);
GGML_API ggml_fp16_t ggml_fp32_to_fp16(float);

// ---- context / tensor construction (harness side; mirrors ggml.h names) -----
GGML_API size_t ggml_tensor_overhead(void);
GGML_API size_t ggml_graph_overhead_custom(size_t size, bool grads);
GGML_API struct ggml_context * ggml_init(struct ggml_init_params params);
GGML_API void                  ggml_free(struct ggml_context * ctx);
GGML_API struct ggml_tensor *  ggml_get_first_tensor(const struct ggml_context * ctx);
GGML_API struct ggml_tensor *  ggml_get_next_tensor (const struct ggml_context * ctx, struct ggml_tensor * tensor);

GGML_API struct ggml_tensor * ggml_new_tensor   (struct ggml_context * ctx, enum ggml_type type, int n_dims, const int64_t * ne);
GGML_API struct ggml_tensor * ggml_new_tensor_1d(struct ggml_context * ctx, enum ggml_type type, int64_t ne0);
GGML_API struct ggml_tensor * ggml_new_tensor_2d(struct ggml_context * ctx, enum ggml_type type, int64_t ne0, int64_t ne1);
GGML_API struct ggml_tensor * ggml_new_tensor_3d(struct ggml_context * ctx, enum ggml_type type, int64_t ne0, int64_t ne1, int64_t ne2);
GGML_API struct ggml_tensor * ggml_new_tensor_4d(struct ggml_context * ctx, enum ggml_type type, int64_t ne0, int64_t ne1, int64_t ne2, int64_t ne3);
GGML_API struct ggml_tensor * ggml_dup_tensor   (struct ggml_context * ctx, const struct ggml_tensor * src);
GGML_API struct ggml_tensor * ggml_view_tensor  (struct ggml_context * ctx, struct ggml_tensor * src);

GGML_API const char *         ggml_get_name(const struct ggml_tensor * tensor);
GGML_API struct ggml_tensor * ggml_set_name(struct ggml_tensor * tensor, const char * name);
GGML_API void ggml_set_input (struct ggml_tensor * tensor);
GGML_API void ggml_set_output(struct ggml_tensor * tensor);

// ops on the hot path (each: the reference call site that emits it)
GGML_API struct ggml_tensor * ggml_add     (struct ggml_context * ctx, struct ggml_tensor * a, struct ggml_tensor * b); // src/llama-model.cpp:6057
GGML_API struct ggml_tensor * ggml_add_id  (struct ggml_context * ctx, struct ggml_tensor * a, struct ggml_tensor * b, struct ggml_tensor * ids); // src/llama-graph.cpp:927
GGML_API struct ggml_tensor * ggml_mul     (struct ggml_context * ctx, struct ggml_tensor * a, struct ggml_tensor * b); // src/llama-graph.cpp:619
GGML_API struct ggml_tensor * ggml_div     (struct ggml_context * ctx, struct ggml_tensor * a, struct ggml_tensor * b); // src/llama-graph.cpp:910
GGML_API struct ggml_tensor * ggml_sum_rows(struct ggml_context * ctx, struct ggml_tensor * a);                         // src/llama-graph.cpp:901
GGML_API struct ggml_tensor * ggml_scale   (struct ggml_context * ctx, struct ggml_tensor * a, float s);
GGML_API struct ggml_tensor * ggml_scale_bias(struct ggml_context * ctx, struct ggml_tensor * a, float s, float b);
GGML_API struct ggml_tensor * ggml_rms_norm(struct ggml_context * ctx, struct ggml_tensor * a, float eps);              // src/llama-graph.cpp:605
GGML_API struct ggml_tensor * ggml_mul_mat (struct ggml_context * ctx, struct ggml_tensor * a, struct ggml_tensor * b); // src/llama-graph.cpp:546
GGML_API void                 ggml_mul_mat_set_prec(struct ggml_tensor * a, enum ggml_prec prec);                       // src/llama-graph.cpp:1289
GGML_API struct ggml_tensor * ggml_mul_mat_id(struct ggml_context * ctx, struct ggml_tensor * as, struct ggml_tensor * b, struct ggml_tensor * ids); // src/llama-graph.cpp:573
GGML_API struct ggml_tensor * ggml_cpy     (struct ggml_context * ctx, struct ggml_tensor * a, struct ggml_tensor * b);
GGML_API struct ggml_tensor * ggml_cont    (struct ggml_context * ctx, struct ggml_tensor * a);
GGML_API struct ggml_tensor * ggml_cont_2d (struct ggml_context * ctx, struct ggml_tensor * a, int64_t ne0, int64_t ne1); // src/llama-graph.cpp:1330
GGML_API struct ggml_tensor * ggml_reshape_2d(struct ggml_context * ctx, struct ggml_tensor * a, int64_t ne0, int64_t ne1);
GGML_API struct ggml_tensor * ggml_reshape_3d(struct ggml_context * ctx, struct ggml_tensor * a, int64_t ne0, int64_t ne1, int64_t ne2);
GGML_API struct ggml_tensor * ggml_reshape_4d(struct ggml_context * ctx, struct ggml_tensor * a, int64_t ne0, int64_t ne1, int64_t ne2, int64_t ne3);
GGML_API struct ggml_tensor * ggml_view_1d (struct ggml_context * ctx, struct ggml_tensor * a, int64_t ne0, size_t offset);
GGML_API struct ggml_tensor * ggml_view_2d (struct ggml_context * ctx, struct ggml_tensor * a, int64_t ne0, int64_t ne1, size_t nb1, size_t offset);
GGML_API struct ggml_tensor * ggml_view_3d (struct ggml_context * ctx, struct ggml_tensor * a, int64_t ne0, int64_t ne1, int64_t ne2, size_t nb1, size_t nb2, size_t offset);
GGML_API struct ggml_tensor * ggml_view_4d (struct ggml_context * ctx, struct ggml_tensor * a, int64_t ne0, int64_t ne1, int64_t ne2, int64_t ne3, size_t nb1, size_t nb2, size_t nb3, size_t offset);
GGML_API struct ggml_tensor * ggml_permute (struct ggml_context * ctx, struct ggml_tensor * a, int axis0, int axis1, int axis2, int axis3);
GGML_API struct ggml_tensor * ggml_transpose(struct ggml_context * ctx, struct ggml_tensor * a);
GGML_API struct ggml_tensor * ggml_get_rows(struct ggml_context * ctx, struct ggml_tensor * a, struct ggml_tensor * b); // src/llama-model.cpp:6053
GGML_API struct ggml_tensor * ggml_set_rows(struct ggml_context * ctx, struct ggml_tensor * a, struct ggml_tensor * b, struct ggml_tensor * c); // src/llama-kv-cache-unified.cpp:1123
GGML_API struct ggml_tensor * ggml_soft_max(struct ggml_context * ctx, struct ggml_tensor * a);
GGML_API struct ggml_tensor * ggml_soft_max_ext(struct ggml_context * ctx, struct ggml_tensor * a, struct ggml_tensor * mask, float scale, float max_bias); // src/llama-graph.cpp:1312
GGML_API void                 ggml_soft_max_add_sinks(struct ggml_tensor * a, struct ggml_tensor * sinks);
// q [hd, n_batch, n_head, ne3] (F32), k [hd, n_kv, n_head_kv, ne3], v [hd_v, n_kv, n_head_kv, ne3] (F16, NOT transposed), mask [n_kv, n_batch_pad] F16
// -> [hd_v, n_head, n_batch, ne3] F32 (src/llama-graph.cpp:1261-1265; tests/test-backend-ops.cpp:4559)
GGML_API struct ggml_tensor * ggml_flash_attn_ext(struct ggml_context * ctx, struct ggml_tensor * q, struct ggml_tensor * k, struct ggml_tensor * v,
                                                   struct ggml_tensor * mask, float scale, float max_bias, float logit_softcap);
GGML_API void                 ggml_flash_attn_ext_set_prec(struct ggml_tensor * a, enum ggml_prec prec);
GGML_API void                 ggml_flash_attn_ext_add_sinks(struct ggml_tensor * a, struct ggml_tensor * sinks);
GGML_API struct ggml_tensor * ggml_cast(struct ggml_context * ctx, struct ggml_tensor * a, enum ggml_type type);              // src/llama-graph.cpp:1313
GGML_API struct ggml_tensor * ggml_rope_ext(struct ggml_context * ctx, struct ggml_tensor * a, struct ggml_tensor * b, struct ggml_tensor * c,
        int n_dims, int mode, int n_ctx_orig, float freq_base, float freq_scale, float ext_factor, float attn_factor, float beta_fast, float beta_slow); // src/llama-model.cpp:6030
GGML_API struct ggml_tensor * ggml_argsort (struct ggml_context * ctx, struct ggml_tensor * a, enum ggml_sort_order order);
GGML_API struct ggml_tensor * ggml_top_k   (struct ggml_context * ctx, struct ggml_tensor * a, int k);                  // src/llama-graph.cpp:883
GGML_API struct ggml_tensor * ggml_swiglu_split(struct ggml_context * ctx, struct ggml_tensor * a, struct ggml_tensor * b); // src/llama-graph.cpp:691
GGML_API struct ggml_tensor * ggml_swiglu_oai(struct ggml_context * ctx, struct ggml_tensor * a, struct ggml_tensor * b, float alpha, float limit); // src/llama-graph.cpp:961
GGML_API struct ggml_tensor * ggml_glu_split(struct ggml_context * ctx, struct ggml_tensor * a, struct ggml_tensor * b, enum ggml_glu_op op);
GGML_API struct ggml_tensor * ggml_unary   (struct ggml_context * ctx, struct ggml_tensor * a, enum ggml_unary_op op);
GGML_API struct ggml_tensor * ggml_silu    (struct ggml_context * ctx, struct ggml_tensor * a);
GGML_API struct ggml_tensor * ggml_sigmoid (struct ggml_context * ctx, struct ggml_tensor * a);

// graph
GGML_API struct ggml_cgraph * ggml_new_graph       (struct ggml_context * ctx);
GGML_API struct ggml_cgraph * ggml_new_graph_custom(struct ggml_context * ctx, size_t size, bool grads);
GGML_API void                 ggml_build_forward_expand(struct ggml_cgraph * cgraph, struct ggml_tensor * tensor);
GGML_API int                  ggml_graph_n_nodes(struct ggml_cgraph * cgraph);
GGML_API struct ggml_tensor * ggml_graph_node   (struct ggml_cgraph * cgraph, int i);
GGML_API void                 ggml_graph_clear  (struct ggml_cgraph * cgraph);

#ifdef __cplusplus
}
#endif
