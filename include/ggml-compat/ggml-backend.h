// ggml-compat/ggml-backend.h — restatement of upstream ggml's backend API (ggml/include/ggml-backend.h; The ggml authors, MIT License —
// see the attribution in ggml.h beside this file): the surface that libllama / llama-bench / test-backend-ops consume
// (SURVEY.md §8b table "In-tree evidence of the surface"). Each group cites the
// reference call sites that rely on it. The real header is absent from the
// reference tree (.gitmodules:1-3); names and signatures are
// [UPSTREAM-KNOWLEDGE] corroborated by those call sites.
#pragma once

#include "ggml.h"

#ifdef __cplusplus
extern "C" {
#endif

#define GGML_BACKEND_API GGML_API

typedef struct ggml_backend_buffer_type * ggml_backend_buffer_type_t;
typedef struct ggml_backend_buffer *      ggml_backend_buffer_t;
typedef struct ggml_backend_event *       ggml_backend_event_t;
typedef struct ggml_backend *             ggml_backend_t;
typedef void *                            ggml_backend_graph_plan_t;
typedef struct ggml_backend_reg *         ggml_backend_reg_t;
typedef struct ggml_backend_device *      ggml_backend_dev_t;

// ---- buffer type (src/llama-model.cpp:280,2143,5576; src/llama-model-loader.cpp:951-959)
GGML_API const char *          ggml_backend_buft_name          (ggml_backend_buffer_type_t buft);
GGML_API ggml_backend_buffer_t ggml_backend_buft_alloc_buffer  (ggml_backend_buffer_type_t buft, size_t size);
GGML_API size_t                ggml_backend_buft_get_alignment (ggml_backend_buffer_type_t buft);
GGML_API size_t                ggml_backend_buft_get_max_size  (ggml_backend_buffer_type_t buft);
GGML_API size_t                ggml_backend_buft_get_alloc_size(ggml_backend_buffer_type_t buft, const struct ggml_tensor * tensor);
GGML_API bool                  ggml_backend_buft_is_host       (ggml_backend_buffer_type_t buft);
GGML_API ggml_backend_dev_t    ggml_backend_buft_get_device    (ggml_backend_buffer_type_t buft);

// ---- buffer (src/llama-kv-cache-unified.cpp:175-182,658; src/llama-model.cpp:5633)
enum ggml_backend_buffer_usage {
    GGML_BACKEND_BUFFER_USAGE_ANY     = 0,
    GGML_BACKEND_BUFFER_USAGE_WEIGHTS = 1,
    GGML_BACKEND_BUFFER_USAGE_COMPUTE = 2,
};

GGML_API const char *               ggml_backend_buffer_name          (ggml_backend_buffer_t buffer);
GGML_API void                       ggml_backend_buffer_free          (ggml_backend_buffer_t buffer);
GGML_API void *                     ggml_backend_buffer_get_base      (ggml_backend_buffer_t buffer);
GGML_API size_t                     ggml_backend_buffer_get_size      (ggml_backend_buffer_t buffer);
GGML_API enum ggml_status           ggml_backend_buffer_init_tensor   (ggml_backend_buffer_t buffer, struct ggml_tensor * tensor);
GGML_API size_t                     ggml_backend_buffer_get_alignment (ggml_backend_buffer_t buffer);
GGML_API size_t                     ggml_backend_buffer_get_alloc_size(ggml_backend_buffer_t buffer, const struct ggml_tensor * tensor);
GGML_API void                       ggml_backend_buffer_clear         (ggml_backend_buffer_t buffer, uint8_t value);
GGML_API bool                       ggml_backend_buffer_is_host       (ggml_backend_buffer_t buffer);
GGML_API void                       ggml_backend_buffer_set_usage     (ggml_backend_buffer_t buffer, enum ggml_backend_buffer_usage usage);
GGML_API enum ggml_backend_buffer_usage ggml_backend_buffer_get_usage (ggml_backend_buffer_t buffer);
GGML_API ggml_backend_buffer_type_t ggml_backend_buffer_get_type      (ggml_backend_buffer_t buffer);
GGML_API void                       ggml_backend_buffer_reset         (ggml_backend_buffer_t buffer);

GGML_API enum ggml_status ggml_backend_tensor_alloc(ggml_backend_buffer_t buffer, struct ggml_tensor * tensor, void * addr);
GGML_API enum ggml_status ggml_backend_view_init(struct ggml_tensor * tensor);

// ---- backend / stream (src/llama-context.cpp:156,228,1132,1482; tests/test-backend-ops.cpp:1293,1348)
GGML_API ggml_guid_t  ggml_backend_guid(ggml_backend_t backend);
GGML_API const char * ggml_backend_name(ggml_backend_t backend);
GGML_API void         ggml_backend_free(ggml_backend_t backend);

GGML_API ggml_backend_buffer_type_t ggml_backend_get_default_buffer_type(ggml_backend_t backend);
GGML_API ggml_backend_buffer_t      ggml_backend_alloc_buffer(ggml_backend_t backend, size_t size);
GGML_API size_t                     ggml_backend_get_alignment(ggml_backend_t backend);
GGML_API size_t                     ggml_backend_get_max_size(ggml_backend_t backend);

GGML_API void ggml_backend_tensor_set_async(ggml_backend_t backend,       struct ggml_tensor * tensor, const void * data, size_t offset, size_t size);
GGML_API void ggml_backend_tensor_get_async(ggml_backend_t backend, const struct ggml_tensor * tensor,       void * data, size_t offset, size_t size);

// "tensor" must be allocated in a buffer
GGML_API void ggml_backend_tensor_set   (      struct ggml_tensor * tensor, const void * data, size_t offset, size_t size);
GGML_API void ggml_backend_tensor_get   (const struct ggml_tensor * tensor,       void * data, size_t offset, size_t size);
GGML_API void ggml_backend_tensor_memset(      struct ggml_tensor * tensor,     uint8_t value, size_t offset, size_t size);

GGML_API void ggml_backend_synchronize(ggml_backend_t backend);

GGML_API enum ggml_status ggml_backend_graph_compute      (ggml_backend_t backend, struct ggml_cgraph * cgraph);
GGML_API enum ggml_status ggml_backend_graph_compute_async(ggml_backend_t backend, struct ggml_cgraph * cgraph);

GGML_API bool ggml_backend_supports_op  (ggml_backend_t backend, const struct ggml_tensor * op);
GGML_API bool ggml_backend_supports_buft(ggml_backend_t backend, ggml_backend_buffer_type_t buft);
GGML_API bool ggml_backend_offload_op   (ggml_backend_t backend, const struct ggml_tensor * op);

// tensor copy between different backends
GGML_API void ggml_backend_tensor_copy(struct ggml_tensor * src, struct ggml_tensor * dst);
// asynchronous copy; the copy is performed after all queued work on backend_src and starts on backend_dst
GGML_API void ggml_backend_tensor_copy_async(ggml_backend_t backend_src, ggml_backend_t backend_dst, struct ggml_tensor * src, struct ggml_tensor * dst);

GGML_API ggml_backend_dev_t ggml_backend_get_device(ggml_backend_t backend);

// ---- events (src/llama-model-loader.cpp:965-1002,1084-1085)
GGML_API ggml_backend_event_t ggml_backend_event_new(ggml_backend_dev_t device);
GGML_API void                 ggml_backend_event_free(ggml_backend_event_t event);
GGML_API void                 ggml_backend_event_record(ggml_backend_event_t event, ggml_backend_t backend);
GGML_API void                 ggml_backend_event_synchronize(ggml_backend_event_t event);
GGML_API void                 ggml_backend_event_wait(ggml_backend_t backend, ggml_backend_event_t event);

// ---- device (src/llama.cpp:176-218; src/llama-model.cpp:5584-5601; tests/test-backend-ops.cpp:6402-6433)
enum ggml_backend_dev_type {
    GGML_BACKEND_DEVICE_TYPE_CPU,
    GGML_BACKEND_DEVICE_TYPE_GPU,
    GGML_BACKEND_DEVICE_TYPE_ACCEL,
};

struct ggml_backend_dev_caps {
    bool async;                // asynchronous operations
    bool host_buffer;          // pinned host buffer
    bool buffer_from_host_ptr; // creating buffers from host ptr
    bool events;               // event synchronization
};

struct ggml_backend_dev_props {
    const char * name;
    const char * description;
    size_t memory_free;
    size_t memory_total;
    enum ggml_backend_dev_type type;
    struct ggml_backend_dev_caps caps;
};

GGML_API const char *                  ggml_backend_dev_name(ggml_backend_dev_t device);
GGML_API const char *                  ggml_backend_dev_description(ggml_backend_dev_t device);
GGML_API void                          ggml_backend_dev_memory(ggml_backend_dev_t device, size_t * free, size_t * total);
GGML_API enum ggml_backend_dev_type    ggml_backend_dev_type(ggml_backend_dev_t device);
GGML_API void                          ggml_backend_dev_get_props(ggml_backend_dev_t device, struct ggml_backend_dev_props * props);
GGML_API ggml_backend_reg_t            ggml_backend_dev_backend_reg(ggml_backend_dev_t device);
GGML_API ggml_backend_t                ggml_backend_dev_init(ggml_backend_dev_t device, const char * params);
GGML_API ggml_backend_buffer_type_t    ggml_backend_dev_buffer_type(ggml_backend_dev_t device);
GGML_API ggml_backend_buffer_type_t    ggml_backend_dev_host_buffer_type(ggml_backend_dev_t device);
GGML_API ggml_backend_buffer_t         ggml_backend_dev_buffer_from_host_ptr(ggml_backend_dev_t device, void * ptr, size_t size, size_t max_tensor_size);
GGML_API bool                          ggml_backend_dev_supports_op(ggml_backend_dev_t device, const struct ggml_tensor * op);
GGML_API bool                          ggml_backend_dev_supports_buft(ggml_backend_dev_t device, ggml_backend_buffer_type_t buft);
GGML_API bool                          ggml_backend_dev_offload_op(ggml_backend_dev_t device, const struct ggml_tensor * op);

// ---- registry (src/llama.cpp:54,143,340-345; src/llama-model.cpp:371-384)
GGML_API const char *       ggml_backend_reg_name(ggml_backend_reg_t reg);
GGML_API size_t             ggml_backend_reg_dev_count(ggml_backend_reg_t reg);
GGML_API ggml_backend_dev_t ggml_backend_reg_dev_get(ggml_backend_reg_t reg, size_t index);
GGML_API void *             ggml_backend_reg_get_proc_address(ggml_backend_reg_t reg, const char * name);

// string-named optional procs (src/llama-context.cpp:187; src/llama-model.cpp:373,344; src/llama.cpp:342)
typedef ggml_backend_buffer_type_t   (*ggml_backend_split_buffer_type_t)(int main_device, const float * tensor_split);
typedef void                         (*ggml_backend_set_n_threads_t)(ggml_backend_t backend, int n_threads);
typedef ggml_backend_buffer_type_t * (*ggml_backend_dev_get_extra_bufts_t)(ggml_backend_dev_t device);
struct ggml_backend_feature {
    const char * name;
    const char * value;
};
typedef struct ggml_backend_feature * (*ggml_backend_get_features_t)(ggml_backend_reg_t reg);

// registry of loaded backends (tools/llama-bench/llama-bench.cpp:1853; tests/test-backend-ops.cpp:6390-6405)
GGML_API void               ggml_backend_register(ggml_backend_reg_t reg);
GGML_API size_t             ggml_backend_reg_count(void);
GGML_API ggml_backend_reg_t ggml_backend_reg_get(size_t index);
GGML_API ggml_backend_reg_t ggml_backend_reg_by_name(const char * name);
GGML_API size_t             ggml_backend_dev_count(void);
GGML_API ggml_backend_dev_t ggml_backend_dev_get(size_t index);
GGML_API ggml_backend_dev_t ggml_backend_dev_by_name(const char * name);
GGML_API ggml_backend_reg_t ggml_backend_load(const char * path);

// ---- allocation helpers (ggml-alloc.h in the real tree; tests/test-backend-ops.cpp:1134)
GGML_API struct ggml_backend_buffer * ggml_backend_alloc_ctx_tensors_from_buft(struct ggml_context * ctx, ggml_backend_buffer_type_t buft);
GGML_API struct ggml_backend_buffer * ggml_backend_alloc_ctx_tensors(struct ggml_context * ctx, ggml_backend_t backend);

#ifdef __cplusplus
}
#endif
