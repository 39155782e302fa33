// ggml-mi355x.h — public C-ABI of the MI355X (gfx950 / CDNA4) ggml backend.
//
// This is the drop-in boundary: everything the reference's host side
// binds for the quantized MUL_MAT / MUL_MAT_ID path is reached through these symbols
// and the vtables they return. Each entry names the reference interface it replaces.
// Plain C: pointers and sizes only.
//
// Build-time ABI note: the vtable/tensor layouts come from include/ggml-compat/*.h
// (a restatement of upstream ggml's interface — The ggml authors, MIT License, attributed there) because the reference's ggml/ submodule is empty
// (/root/reference/.gitmodules:1-3). Against a real ggml checkout, build with its
// headers instead — the symbols below do not change.
#pragma once

#include "ggml.h"
#include "ggml-backend.h"

#ifdef __cplusplus
extern "C" {
#endif

#define GGML_MI355X_NAME        "MI355X"
#define GGML_MI355X_MAX_DEVICES 16

// --- dynamic-backend entry points ---------------------------------------------------------
// replaces: the `ggml_backend_init` / `ggml_backend_score` exports every libggml-<name>.so
// carries under -DGGML_BACKEND_DL=ON (docs/build.md:613); found by ggml_backend_load_all()
// (tools/llama-bench/llama-bench.cpp:1853, tests/test-backend-ops.cpp:6390).
GGML_BACKEND_API ggml_backend_reg_t ggml_backend_init(void);
// > 0 only when a gfx950 device is visible; 0 = "do not load me on this system"
GGML_BACKEND_API int                ggml_backend_score(void);

// --- static-registry entry points -----------------------------------------------------------
// replaces: ggml_backend_cuda_reg() in the static registry (the slot the gfx906 fork's HIP build
// fills; docs/build-gfx906.md:31-32). Registry name "MI355X"; device names "MI355X0", "MI355X1", …
// (what -dev / test-backend-ops -b match: common/arg.cpp:1186, tests/test-backend-ops.cpp:6405).
GGML_BACKEND_API ggml_backend_reg_t ggml_backend_mi355x_reg(void);

// replaces: ggml_backend_cuda_init(device) — one backend = one ordered HIP stream on `device`
// (src/llama-context.cpp:155-161 reaches it through ggml_backend_dev_init)
GGML_BACKEND_API ggml_backend_t ggml_backend_mi355x_init(int device);
GGML_BACKEND_API bool           ggml_backend_is_mi355x(ggml_backend_t backend);
GGML_BACKEND_API int            ggml_backend_mi355x_get_device_count(void);
GGML_BACKEND_API void           ggml_backend_mi355x_get_device_description(int device, char * description, size_t description_size);
GGML_BACKEND_API void           ggml_backend_mi355x_get_device_memory(int device, size_t * free, size_t * total);

// replaces: ggml_backend_cuda_buffer_type(device) — device (HBM) buffers, 128-byte alignment
// (src/llama-model.cpp:5589-5624 allocates weights through it; src/llama-kv-cache-unified.cpp:175 the KV cache)
GGML_BACKEND_API ggml_backend_buffer_type_t ggml_backend_mi355x_buffer_type(int device);
// replaces: ggml_backend_cuda_host_buffer_type() — pinned host staging (src/llama-model-loader.cpp:951-959)
GGML_BACKEND_API ggml_backend_buffer_type_t ggml_backend_mi355x_host_buffer_type(void);

// --- optional procs, looked up BY NAME through reg->iface.get_proc_address ----------------------
// (src/llama-context.cpp:187, src/llama-model.cpp:344,373, src/llama.cpp:342). Exposed here too:
//   "ggml_backend_get_features"        -> ggml_backend_get_features_t
//   "ggml_backend_mi355x_get_stream"   -> void * (*)(ggml_backend_t)        : the backend's hipStream_t
//   "ggml_backend_mi355x_get_counters" -> see struct below
//   "ggml_backend_mi355x_set_option"   -> int (*)(ggml_backend_t, const char * key, int value)
//   "ggml_backend_split_buffer_type"   -> ggml_backend_split_buffer_type_t  : -sm row (EXPERIMENTAL: functional, eager fork / join per mat-mul,
//                                         measured 117 vs 564 tok/s on virtual devices; every device that receives rows must be peer-accessible
//                                         from the main device — checked when the buffer type is made)
// not provided (returns NULL): "ggml_backend_set_n_threads" (GPU backend), "ggml_backend_dev_get_extra_bufts".

// per-backend counters for the measurement leg (SURVEY.md §5 "expose per-kernel bytes/time counters")
struct ggml_backend_mi355x_counters {
    uint64_t graphs_computed;
    uint64_t nodes_computed;
    uint64_t kernels_launched;
    uint64_t graph_replays;        // graph_compute calls served by a cached hipGraph
    uint64_t graph_captures;
    uint64_t mmvq_launches;        // decode mat-vec launches
    uint64_t mmq_launches;         // prefill MFMA mat-mul launches
    uint64_t weight_bytes;         // algorithmic weight bytes streamed by mmvq/mmq launches
    uint64_t act_quant_launches;
    uint64_t act_quant_reused;     // MUL_MATs that reused the previous node's quantized activations
    uint64_t split_mul_mats;       // MUL_MATs on row-split weights (one launch per device that holds rows)
};
// Row-split weights (-sm row). Replaces ggml_backend_cuda_split_buffer_type, which the host finds through the registry proc
// "ggml_backend_split_buffer_type" (src/llama-model.cpp:368-387; typedef ggml_backend_split_buffer_type_t, ggml-backend.h): main_device is
// the index of the device whose backend runs the graph, tensor_split the per-device proportions (llama_model_params.tensor_split; all zero =
// equal shares). NULL when a device that would hold rows has no peer mapping to the main device.
GGML_BACKEND_API ggml_backend_buffer_type_t ggml_backend_mi355x_split_buffer_type(int main_device, const float * tensor_split);
GGML_BACKEND_API void * ggml_backend_mi355x_get_stream(ggml_backend_t backend);
// Layer-split hand-off through buffers that are not ggml tensors (one process per GPU: the RCCL send / recv buffers of bench.py --gpus N; inside ONE
// process the reference's scheduler uses backend_i.cpy_tensor_async, src/llama-context.cpp:255-285): device-to-device copies between a tensor and a raw
// device pointer, ordered on the backend's stream. (set_tensor_async / get_tensor_async take HOST memory, as in ggml-backend.h.) Also procs by name.
GGML_BACKEND_API void   ggml_backend_mi355x_tensor_set_from_device_async(ggml_backend_t backend, struct ggml_tensor * tensor, const void * dev_src, size_t offset, size_t size);
GGML_BACKEND_API void   ggml_backend_mi355x_tensor_get_to_device_async(ggml_backend_t backend, const struct ggml_tensor * tensor, void * dev_dst, size_t offset, size_t size);
GGML_BACKEND_API void   ggml_backend_mi355x_get_counters(ggml_backend_t backend, struct ggml_backend_mi355x_counters * out);
GGML_BACKEND_API void   ggml_backend_mi355x_reset_counters(ggml_backend_t backend);
// options: "graphs" (0/1 hipGraph capture+replay), "fusion" (0/1 node fusion),
//          "profile" (0/1: eager execution with every quantized mat-mul launch bracketed by a hipEvent pair)
GGML_BACKEND_API int    ggml_backend_mi355x_set_option(ggml_backend_t backend, const char * key, int value);


// measurement leg: durations of the quantized mat-mul launches recorded while option "profile" was on
struct ggml_backend_mi355x_prof_entry {
    int32_t  type;              // ggml_type of the weights
    int32_t  n;                 // activation columns
    int64_t  m, k;              // weight rows, row length
    uint64_t launches;
    double   total_ms;          // sum over launches of (stop event - start event); for the grouped mat-vec launches the pair is the
                                // dispatch's own start / stop (hipExtLaunchKernelGGL): the interval rocprofv3's kernel trace reports
    uint64_t bytes_per_launch;  // algorithmic weight bytes: m * row_size(type, k)
    char     kernel[96];        // grouped launches: the instantiation as rocprofv3 spells it ("k_mmvq_fused<12, 12, true, 2, 2, 4, 8>"), else ""
};
GGML_BACKEND_API int ggml_backend_mi355x_get_profile(ggml_backend_t backend, struct ggml_backend_mi355x_prof_entry * out, int cap);

#ifdef __cplusplus
}
#endif
