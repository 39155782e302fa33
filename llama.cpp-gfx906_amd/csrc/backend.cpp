// backend.cpp — the ggml-facing half of the MI355X backend: registry, device, buffer types,
// buffers, the stream backend and graph_compute. This is the ONLY translation
// unit that depends on the ggml struct layouts (include/ggml-compat/*.h, or a real ggml's
// headers). Everything below the op switch is plain-pointer HIP code in kernels.h.
//
// Conventions kept from the reference's call sites:
//  * alloc failure -> NULL (src/llama-model.cpp:5602,5611); compute -> enum ggml_status
//    (src/llama-context.cpp:1101-1106); unsupported op -> supports_op says no, graph_compute never
//    meets one; supports_op never dereferences tensor->data (dummy tensors, src/llama-model.cpp:278-283).
//  * one ggml_backend_t = one ordered HIP stream driven by one host thread at a time
//    (tests/test-thread-safety.cpp:1-4); graph_compute is asynchronous (src/llama-context.cpp:1449).
//  * device type GPU so that it is auto-selected for offload (src/llama.cpp:183-190);
//    caps.async && caps.events so that pipeline parallelism can turn on (src/llama-context.cpp:262-279).
#include "ggml-mi355x.h"
#include "ggml-backend-impl.h"
#include "ggml-impl.h"

#include "kernels.h"

#include <hip/hip_runtime.h>

#include <stddef.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <chrono>
#include <mutex>
#include <string>
#include <unordered_map>
#include <vector>

using namespace mi355x;

#define MI_CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { \
    GGML_ABORT("MI355X backend: HIP error %s (%s) at %s:%d", hipGetErrorName(e_), hipGetErrorString(e_), __FILE__, __LINE__); } } while (0)

#define MI_LOG(...) do { fprintf(stderr, "ggml-mi355x: " __VA_ARGS__); } while (0)

// On the graph path (be_graph_compute and what it calls) a HIP error is not fatal: it unwinds to be_graph_compute, which ends a capture in
// progress and returns GGML_STATUS_FAILED — llama_context::decode maps that and rolls the KV cells of the batch back
// (src/llama-context.cpp:1078-1107). Everywhere else (buffer and device entry points, which have no status to return) MI_CHECK aborts.
struct mi_graph_error { hipError_t err; const char * file; int line; };
#define MI_CHECK_G(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) throw mi_graph_error{ e_, __FILE__, __LINE__ }; } while (0)
#define MI_REQUIRE_G(cond) do { if (!(cond)) throw mi_graph_error{ hipErrorInvalidValue, __FILE__, __LINE__ }; } while (0)

static constexpr size_t MI_BUFFER_ALIGN = 128;
static constexpr size_t MI_TENSOR_PAD   = 256;   // readable slack after quantized tensors (wave-wide 16 B loads may run past the last block)

// ---------------------------------------------------------------------------------------------
// device table
// ---------------------------------------------------------------------------------------------
struct mi_device {
    int id;
    std::string name;         // "MI355X0"
    std::string description;
    size_t total_mem;
    struct ggml_backend_device dev;
    struct ggml_backend_buffer_type buft;
};

struct mi_globals {
    int n_devices = 0;
    mi_device devices[GGML_MI355X_MAX_DEVICES];
    struct ggml_backend_reg reg;
    struct ggml_backend_buffer_type host_buft;
    bool initialised = false;
};

static mi_globals & G();

extern "C" {
int    ggml_backend_mi355x_test_quantize(ggml_backend_t backend, const float * x, int64_t k, int64_t n, int kind, int8_t * qs, float * d, int16_t * bsums);
double ggml_backend_mi355x_test_hbm_read_gbps(ggml_backend_t backend, size_t bytes, int iters);
}

static void set_device(int id) { MI_CHECK(hipSetDevice(id)); }

// ---------------------------------------------------------------------------------------------
// device buffer
// ---------------------------------------------------------------------------------------------
struct mi_buffer_ctx {
    int device;
    void * base;
};

static void buf_free(ggml_backend_buffer_t buffer) {
    mi_buffer_ctx * c = (mi_buffer_ctx *) buffer->context;
    set_device(c->device);
    MI_CHECK(hipFree(c->base));
    delete c;
}
static void * buf_get_base(ggml_backend_buffer_t buffer) { return ((mi_buffer_ctx *) buffer->context)->base; }

static enum ggml_status buf_init_tensor(ggml_backend_buffer_t buffer, struct ggml_tensor * tensor) {
    mi_buffer_ctx * c = (mi_buffer_ctx *) buffer->context;
    if (tensor->view_src != NULL) return GGML_STATUS_SUCCESS;
    if (ggml_is_quantized(tensor->type) && tensor->view_src == NULL) {
        // zero the padding after the last block so over-reads see defined bytes
        const size_t size = ggml_nbytes(tensor);
        const size_t alloc = ggml_backend_buft_get_alloc_size(buffer->buft, tensor);
        if (alloc > size) {
            set_device(c->device);
            MI_CHECK(hipMemset((char *) tensor->data + size, 0, alloc - size));
        }
    }
    return GGML_STATUS_SUCCESS;
}
static void buf_memset_tensor(ggml_backend_buffer_t buffer, struct ggml_tensor * tensor, uint8_t value, size_t offset, size_t size) {
    mi_buffer_ctx * c = (mi_buffer_ctx *) buffer->context;
    set_device(c->device);
    MI_CHECK(hipMemsetAsync((char *) tensor->data + offset, value, size, hipStreamPerThread));
    MI_CHECK(hipStreamSynchronize(hipStreamPerThread));
}
static void buf_set_tensor(ggml_backend_buffer_t buffer, struct ggml_tensor * tensor, const void * data, size_t offset, size_t size) {
    mi_buffer_ctx * c = (mi_buffer_ctx *) buffer->context;
    set_device(c->device);
    MI_CHECK(hipMemcpyAsync((char *) tensor->data + offset, data, size, hipMemcpyHostToDevice, hipStreamPerThread));
    MI_CHECK(hipStreamSynchronize(hipStreamPerThread));
}
static void buf_get_tensor(ggml_backend_buffer_t buffer, const struct ggml_tensor * tensor, void * data, size_t offset, size_t size) {
    mi_buffer_ctx * c = (mi_buffer_ctx *) buffer->context;
    set_device(c->device);
    MI_CHECK(hipMemcpyAsync(data, (const char *) tensor->data + offset, size, hipMemcpyDeviceToHost, hipStreamPerThread));
    MI_CHECK(hipStreamSynchronize(hipStreamPerThread));
}

static bool buffer_is_mi355x(ggml_backend_buffer_t buffer);

static bool buf_cpy_tensor(ggml_backend_buffer_t buffer, const struct ggml_tensor * src, struct ggml_tensor * dst) {
    ggml_backend_buffer_t sbuf = src->view_src ? src->view_src->buffer : src->buffer;
    if (!sbuf || !buffer_is_mi355x(sbuf)) return false;
    mi_buffer_ctx * sc = (mi_buffer_ctx *) sbuf->context;
    mi_buffer_ctx * dc = (mi_buffer_ctx *) buffer->context;
    set_device(dc->device);
    if (sc->device == dc->device) {
        MI_CHECK(hipMemcpyAsync(dst->data, src->data, ggml_nbytes(src), hipMemcpyDeviceToDevice, hipStreamPerThread));
    } else {
        MI_CHECK(hipMemcpyPeerAsync(dst->data, dc->device, src->data, sc->device, ggml_nbytes(src), hipStreamPerThread));   // xGMI hop
    }
    MI_CHECK(hipStreamSynchronize(hipStreamPerThread));
    return true;
}
static void buf_clear(ggml_backend_buffer_t buffer, uint8_t value) {
    mi_buffer_ctx * c = (mi_buffer_ctx *) buffer->context;
    set_device(c->device);
    MI_CHECK(hipMemsetAsync(c->base, value, buffer->size, hipStreamPerThread));
    MI_CHECK(hipStreamSynchronize(hipStreamPerThread));
}

static const struct ggml_backend_buffer_i mi_buffer_iface = {
    /* .free_buffer   = */ buf_free,
    /* .get_base      = */ buf_get_base,
    /* .init_tensor   = */ buf_init_tensor,
    /* .memset_tensor = */ buf_memset_tensor,
    /* .set_tensor    = */ buf_set_tensor,
    /* .get_tensor    = */ buf_get_tensor,
    /* .cpy_tensor    = */ buf_cpy_tensor,
    /* .clear         = */ buf_clear,
    /* .reset         = */ NULL,
};

static bool buffer_is_mi355x(ggml_backend_buffer_t buffer) { return buffer->iface.free_buffer == buf_free; }

// ---------------------------------------------------------------------------------------------
// device buffer type
// ---------------------------------------------------------------------------------------------
static const char * buft_get_name(ggml_backend_buffer_type_t buft) { return ((mi_device *) buft->context)->name.c_str(); }

static ggml_backend_buffer_t buft_alloc_buffer(ggml_backend_buffer_type_t buft, size_t size) {
    mi_device * d = (mi_device *) buft->context;
    set_device(d->id);
    void * p = NULL;
    hipError_t err = hipMalloc(&p, size + MI_TENSOR_PAD);
    if (err != hipSuccess) {
        (void) hipGetLastError();
        MI_LOG("allocating %.2f MiB on device %d failed: %s\n", size/1024.0/1024.0, d->id, hipGetErrorString(err));
        return NULL;
    }
    mi_buffer_ctx * c = new mi_buffer_ctx{ d->id, p };
    return ggml_backend_buffer_init(buft, mi_buffer_iface, c, size);
}
static size_t buft_get_alignment(ggml_backend_buffer_type_t) { return MI_BUFFER_ALIGN; }
static size_t buft_get_alloc_size(ggml_backend_buffer_type_t, const struct ggml_tensor * tensor) {
    size_t size = ggml_nbytes(tensor);
    if (ggml_is_quantized(tensor->type)) size += MI_TENSOR_PAD;
    return size;
}
static bool buft_is_host_no(ggml_backend_buffer_type_t) { return false; }

static const struct ggml_backend_buffer_type_i mi_buft_iface = {
    /* .get_name       = */ buft_get_name,
    /* .alloc_buffer   = */ buft_alloc_buffer,
    /* .get_alignment  = */ buft_get_alignment,
    /* .get_max_size   = */ NULL,
    /* .get_alloc_size = */ buft_get_alloc_size,
    /* .is_host        = */ buft_is_host_no,
};

// ---------------------------------------------------------------------------------------------
// row-split buffer type (-sm row; the host binds it through the "ggml_backend_split_buffer_type" proc, src/llama-model.cpp:368-387).
// EXPERIMENTAL until it has run on two real GPUs: so far only under GGML_MI355X_VIRTUAL_DEVICES (both "devices" the same HIP device, no peer traffic).
// Peer kernels store their row ranges straight into the main device's compute buffer through the peer mapping, ordered by events only; that the main device
// sees those stores without stale L2 lines is unproven here (the reference computes into a per-device dst and copies: ggml-cuda.cu ggml_cuda_op_mul_mat).
// a weight matrix's ROWS are spread over the devices in the proportions of tensor_split; every device holds its slice in its own HBM.
// A MUL_MAT on such a weight runs one launch per device, each on its own stream: the devices read the (small) activations straight from the
// main device's memory and write their rows of the result straight into the main device's dst — both through the peer mappings over xGMI, no
// staging copies and no collective (the result is a concatenation of row ranges, not a sum). Only 2-D quantized weights of MUL_MAT can live here
// (supports_op refuses everything else, so the host's weight_buft_supported probe, src/llama-model.cpp:152-286, sends norms, biases and
// expert stacks to the device's ordinary buffer type).
// ---------------------------------------------------------------------------------------------
struct mi_split_buft_ctx {
    int main_device = 0;                                   // index into G().devices
    float cum[GGML_MI355X_MAX_DEVICES + 1] = {};           // cum[i] = fraction of the rows before device i; cum[n_devices] = 1
    std::string name;
    struct ggml_backend_buffer_type buft;
};
struct mi_split_tensor {                                   // tensor->extra of a tensor in a split buffer
    void *  data[GGML_MI355X_MAX_DEVICES] = {};            // the slice in device i's memory (NULL: no rows there)
    int64_t row_lo[GGML_MI355X_MAX_DEVICES + 1] = {};      // device i holds rows [row_lo[i], row_lo[i+1])
};
struct mi_split_buffer_ctx {
    mi_split_buft_ctx * bt;
    std::vector<mi_split_tensor *> tensors;
};
static const int64_t MI_SPLIT_ROW_ROUND = 64;              // slice boundaries at multiples of this many rows (the MFMA tile height of the prefill kernel)

static void split_rows(const mi_split_buft_ctx * bt, int64_t nrows, int64_t * row_lo) {
    const int n = G().n_devices;
    for (int i = 0; i <= n; i++) {
        int64_t r = i == n ? nrows : (int64_t)(nrows*(double) bt->cum[i]);
        if (i > 0 && i < n) r -= r % MI_SPLIT_ROW_ROUND;
        row_lo[i] = i == 0 ? 0 : std::max(r, row_lo[i - 1]);
    }
}
static void split_buf_free(ggml_backend_buffer_t buffer) {
    mi_split_buffer_ctx * c = (mi_split_buffer_ctx *) buffer->context;
    for (mi_split_tensor * e : c->tensors) {
        for (int i = 0; i < G().n_devices; i++) if (e->data[i]) { set_device(G().devices[i].id); MI_CHECK(hipFree(e->data[i])); }
        delete e;
    }
    delete c;
}
static void * split_buf_get_base(ggml_backend_buffer_t) { return (void *) 0x1000; }     // never dereferenced: the slices hang off tensor->extra
static enum ggml_status split_buf_init_tensor(ggml_backend_buffer_t buffer, struct ggml_tensor * tensor) {
    mi_split_buffer_ctx * c = (mi_split_buffer_ctx *) buffer->context;
    if (tensor->view_src != NULL || !ggml_is_contiguous(tensor) || tensor->ne[2] != 1 || tensor->ne[3] != 1) return GGML_STATUS_FAILED;   // whole 2-D matrices only
    mi_split_tensor * e = new mi_split_tensor;
    split_rows(c->bt, tensor->ne[1], e->row_lo);
    for (int i = 0; i < G().n_devices; i++) {
        const int64_t rows = e->row_lo[i + 1] - e->row_lo[i];
        if (rows == 0) continue;
        const size_t bytes = (size_t) rows*tensor->nb[1];
        set_device(G().devices[i].id);
        if (hipMalloc(&e->data[i], bytes + MI_TENSOR_PAD) != hipSuccess) {
            (void) hipGetLastError();
            for (int j = 0; j < i; j++) if (e->data[j]) { set_device(G().devices[j].id); (void) hipFree(e->data[j]); }
            delete e;
            return GGML_STATUS_ALLOC_FAILED;
        }
        MI_CHECK(hipMemset((char *) e->data[i] + bytes, 0, MI_TENSOR_PAD));       // wave-wide loads may run past the last block
    }
    c->tensors.push_back(e);
    tensor->extra = e;
    return GGML_STATUS_SUCCESS;
}
static void split_buf_set_tensor(ggml_backend_buffer_t, struct ggml_tensor * tensor, const void * data, size_t offset, size_t size) {
    GGML_ASSERT(offset == 0 && size == ggml_nbytes(tensor) && "split tensors are written whole");
    const mi_split_tensor * e = (const mi_split_tensor *) tensor->extra;
    for (int i = 0; i < G().n_devices; i++) {
        if (!e->data[i]) continue;
        set_device(G().devices[i].id);
        MI_CHECK(hipMemcpyAsync(e->data[i], (const char *) data + e->row_lo[i]*tensor->nb[1], (size_t)(e->row_lo[i + 1] - e->row_lo[i])*tensor->nb[1], hipMemcpyHostToDevice, hipStreamPerThread));
    }
    for (int i = 0; i < G().n_devices; i++) if (e->data[i]) { set_device(G().devices[i].id); MI_CHECK(hipStreamSynchronize(hipStreamPerThread)); }
}
static void split_buf_get_tensor(ggml_backend_buffer_t, const struct ggml_tensor * tensor, void * data, size_t offset, size_t size) {
    GGML_ASSERT(offset == 0 && size == ggml_nbytes(tensor) && "split tensors are read whole");
    const mi_split_tensor * e = (const mi_split_tensor *) tensor->extra;
    for (int i = 0; i < G().n_devices; i++) {
        if (!e->data[i]) continue;
        set_device(G().devices[i].id);
        MI_CHECK(hipMemcpyAsync((char *) data + e->row_lo[i]*tensor->nb[1], e->data[i], (size_t)(e->row_lo[i + 1] - e->row_lo[i])*tensor->nb[1], hipMemcpyDeviceToHost, hipStreamPerThread));
    }
    for (int i = 0; i < G().n_devices; i++) if (e->data[i]) { set_device(G().devices[i].id); MI_CHECK(hipStreamSynchronize(hipStreamPerThread)); }
}
static void split_buf_clear(ggml_backend_buffer_t, uint8_t) {}
static const struct ggml_backend_buffer_i mi_split_buffer_iface = {
    /* .free_buffer   = */ split_buf_free,
    /* .get_base      = */ split_buf_get_base,
    /* .init_tensor   = */ split_buf_init_tensor,
    /* .memset_tensor = */ NULL,
    /* .set_tensor    = */ split_buf_set_tensor,
    /* .get_tensor    = */ split_buf_get_tensor,
    /* .cpy_tensor    = */ NULL,
    /* .clear         = */ split_buf_clear,
    /* .reset         = */ NULL,
};
static const char * split_buft_get_name(ggml_backend_buffer_type_t buft) { return ((mi_split_buft_ctx *) buft->context)->name.c_str(); }
static ggml_backend_buffer_t split_buft_alloc_buffer(ggml_backend_buffer_type_t buft, size_t size) {
    // the slices are allocated per tensor in init_tensor: the address range handed to the host's allocator is only bookkeeping
    return ggml_backend_buffer_init(buft, mi_split_buffer_iface, new mi_split_buffer_ctx{ (mi_split_buft_ctx *) buft->context, {} }, size);
}
static size_t split_buft_get_alloc_size(ggml_backend_buffer_type_t buft, const struct ggml_tensor * tensor) {
    int64_t row_lo[GGML_MI355X_MAX_DEVICES + 1];
    split_rows((const mi_split_buft_ctx *) buft->context, tensor->ne[1], row_lo);
    size_t total = 0;
    for (int i = 0; i < G().n_devices; i++) if (row_lo[i + 1] > row_lo[i]) total += (size_t)(row_lo[i + 1] - row_lo[i])*tensor->nb[1] + MI_TENSOR_PAD;
    return total;
}
static const struct ggml_backend_buffer_type_i mi_split_buft_iface = {
    /* .get_name       = */ split_buft_get_name,
    /* .alloc_buffer   = */ split_buft_alloc_buffer,
    /* .get_alignment  = */ buft_get_alignment,
    /* .get_max_size   = */ NULL,
    /* .get_alloc_size = */ split_buft_get_alloc_size,
    /* .is_host        = */ buft_is_host_no,
};
static bool buft_is_split(ggml_backend_buffer_type_t buft) { return buft && buft->iface.get_name == split_buft_get_name; }
static bool tensor_is_split(const struct ggml_tensor * t) { return t && t->buffer && buft_is_split(t->buffer->buft); }

// ---------------------------------------------------------------------------------------------
// pinned host buffer type (src/llama-model-loader.cpp:951-959 uses it for the 4 x 1 MiB upload ring)
// ---------------------------------------------------------------------------------------------
static void host_buf_free(ggml_backend_buffer_t buffer) { MI_CHECK(hipHostFree(buffer->context)); }
static void * host_buf_get_base(ggml_backend_buffer_t buffer) { return buffer->context; }
static void host_buf_memset_tensor(ggml_backend_buffer_t, struct ggml_tensor * t, uint8_t v, size_t off, size_t size) { memset((char *) t->data + off, v, size); }
static void host_buf_set_tensor(ggml_backend_buffer_t, struct ggml_tensor * t, const void * data, size_t off, size_t size) { memcpy((char *) t->data + off, data, size); }
static void host_buf_get_tensor(ggml_backend_buffer_t, const struct ggml_tensor * t, void * data, size_t off, size_t size) { memcpy(data, (const char *) t->data + off, size); }
static void host_buf_clear(ggml_backend_buffer_t buffer, uint8_t value) { memset(buffer->context, value, buffer->size); }

static const struct ggml_backend_buffer_i mi_host_buffer_iface = {
    host_buf_free, host_buf_get_base, NULL, host_buf_memset_tensor, host_buf_set_tensor, host_buf_get_tensor, NULL, host_buf_clear, NULL,
};

static const char * host_buft_get_name(ggml_backend_buffer_type_t) { return GGML_MI355X_NAME "_Host"; }
static ggml_backend_buffer_t host_buft_alloc_buffer(ggml_backend_buffer_type_t buft, size_t size) {
    void * p = NULL;
    hipError_t err = hipHostMalloc(&p, size, hipHostMallocDefault);
    if (err != hipSuccess) {
        (void) hipGetLastError();
        MI_LOG("failed to allocate %.2f MiB of pinned memory: %s\n", size/1024.0/1024.0, hipGetErrorString(err));
        return NULL;
    }
    return ggml_backend_buffer_init(buft, mi_host_buffer_iface, p, size);
}
static size_t host_buft_get_alignment(ggml_backend_buffer_type_t) { return 64; }
static bool host_buft_is_host(ggml_backend_buffer_type_t) { return true; }

static const struct ggml_backend_buffer_type_i mi_host_buft_iface = {
    host_buft_get_name, host_buft_alloc_buffer, host_buft_get_alignment, NULL, NULL, host_buft_is_host,
};

// ---------------------------------------------------------------------------------------------
// backend (stream)
// ---------------------------------------------------------------------------------------------
// What must be unchanged for a captured hipGraph to be replayed, per node (the checks the reference's GPU
// backends make before reusing a captured graph): the node itself, its op/type/shape/strides/data pointer,
// a hash of op_params, and the data pointers + leading shape of its sources (leaf tensors are not nodes and
// split graphs carry no leaf list, so their placement is pinned through the nodes that read them).
struct node_sig {
    const void * node; const void * data;
    const void * src[GGML_MAX_SRC];                                            // the source TENSORS: the graph's topology (checked for the whole graph before any segment is launched)
    int32_t op, type;
    int64_t ne[4]; size_t nb[3];
    const void * src_data[GGML_MAX_SRC]; uint32_t src_hash[GGML_MAX_SRC];      // every source: placement + a hash of its type, shape and strides
    uint32_t params_hash; uint32_t flags;
};

static_assert(sizeof(node_sig) == 16 + 8*GGML_MAX_SRC + 8 + 32 + 24 + 8*GGML_MAX_SRC + 4*GGML_MAX_SRC + 8, "node_sig has no padding bytes: signatures are compared with memcmp");
struct graph_seg { int begin, end; hipGraphExec_t exec; };      // nodes [begin, end) as one executable graph (exec == NULL: view ops only, nothing to launch)
struct graph_entry {
    std::vector<node_sig> sig; uint64_t digest = 0;      // digest: a cheap hash of the signature — only the entry that shares it is compared in full
    std::vector<graph_seg> segs;                          // empty: not captured (yet)
    uint64_t last_use = 0;
    int seen = 0;              // identical submissions observed (capture on the 2nd)
};
static constexpr int MI_MAX_GRAPHS = 24;

struct mi_backend_ctx {
    int device;                    // HIP device id
    int dev_index = 0;             // index into G().devices
    std::string name;
    hipStream_t stream = nullptr;

    void * scratch = nullptr;      // quantized activations
    size_t scratch_size = 0;
    int moe_dual = -1;                  // option "moe_dual"
    float * moe_ws = nullptr;                                   // logits + arrival counter of the multi-workgroup router kernel (moe_route)
    // the last router launch of this graph pass: it ranked `vals` (probabilities, or logits) into `sorted` and left the eight best values in rank order at moe_ws + 64
    // (floats the kernel does not use otherwise): a combine that gathers vals[sorted[u]] takes them from there — one load instead of two dependent cold ones
    struct { bool valid = false; const void * vals = nullptr; const void * sorted = nullptr; } rt;
    // producer-side activation quantization (mmvq_fin): the image a GLU launch writes for the mat-vec that follows + its arrival counters
    void * fin_img = nullptr; unsigned * fin_cnt = nullptr;
    // the rotation table of the token being decoded ((cos, sin) per pair index; mmvq_rope::table): filled by one small launch when a graph's
    // first rotated mat-vec launch comes up (and again if a launch asks for other rope parameters), read by every such launch after it
    float * rope_tab = nullptr; mmvq_rope rope_tab_key = {}; bool rope_tab_valid = false;
    static constexpr size_t FIN_IMG_BYTES = 64*1024; static constexpr int FIN_COUNTERS = 256;
    void * cvt = nullptr; size_t cvt_size = 0;                  // the F32 copy of an F16 src1 of a quantized MUL_MAT
    void * kv16 = nullptr; size_t kv16_size = 0;                // FLASH_ATTN_EXT: dense f16 copies of a quantized / bf16 cache view, and the transposed V of the prefill kernel (kv_to_f16)
    // a MoE combine left pending (try_fused_moe_combine): the vector x_out = res + sum_u w_u * plane u does not exist until the launch that reads it (the next
    // norm + mat-vec launch, streamed kernel) has evaluated it in its prologue, or pp_flush has
    struct { bool active = false; const float * res = nullptr; float * x_out = nullptr; int n_planes = 0; int64_t m = 0;
             const float * planes = nullptr; int stride = 0;                                     // where the planes are (floats between them)
             const float * probs = nullptr; size_t probs_bytes = 0; const int32_t * ids = nullptr; int mode = 0; } pp;      // planes = the used experts' outputs, weights from probs[ids[u]]
    float * attn_part = nullptr; size_t attn_part_bytes = 0;   // partial results of the decode attention's cell ranges at long contexts (attn_decode)

    // activation-quantisation reuse inside one graph_compute
    struct { const void * data; int64_t k, n_inner, n_outer; size_t s_inner, s_outer; int kind; act_q8 q; bool valid; size_t span;
             size_t off; } aq = {};   // off: where in the scratch the bf16 copy sits (a producer kernel that still reads offset 0 writes its result's copy higher up)
    bool aq_fresh = false;             // set by a fusion matcher whose own kernel produced the cached copy (its node's write must not invalidate it)

    // hipGraph cache (one entry: llama.cpp re-submits the same decode graph, src/llama-context.cpp:728)
    bool use_graphs = true;
    bool use_fusion = true;
    std::unordered_map<const struct ggml_tensor *, int> uses;   // consumers per tensor in the graph being run (fusion legality)
    std::vector<graph_entry> graphs;     // small LRU: decode graphs differ only in n_kv (one per 32 tokens of context)
    std::vector<node_sig> cur_sig; uint64_t cur_digest = 0;
    uint64_t graph_tick = 0;

    struct ggml_backend_mi355x_counters cnt = {};

    bool capturing = false;
    graph_entry * cap_entry = nullptr;               // the cache entry a capture in progress fills (dropped if the capture fails)
    // small uploads (set_tensor_async of <= UP_SMALL_MAX bytes: a decode step's inputs) are staged in a pinned, device-mapped ring and go out as ONE
    // launch in front of whatever the stream is asked to do next (uploads_flush), instead of one copy each
    struct { char * host = nullptr; char * dev = nullptr; int cur = 0; size_t off = 0; hipEvent_t ev[2] = { nullptr, nullptr }; bool ev_pending[2] = { false, false };
             int n = 0; upload_batch b = {}; bool failed = false; } up;
    static constexpr size_t UP_HALF = 1u << 20, UP_SMALL_MAX = 64u << 10;
    unsigned * err_host = nullptr;                   // host-mapped words: a bounded wait inside a kernel gave up ([1] the MoE router)
    unsigned * err_dev = nullptr;                    // ... their device address

    // "profile" option: every quantized mat-mul launch is bracketed by a hipEvent pair on this stream (eager mode)
    struct prof_rec { int type; int64_t m, k, n; uint64_t bytes; hipEvent_t e0, e1; const char * kernel = nullptr; };
    bool profiling = false;
    bool prof_in_graph = false;      // option "profile" = 2: the event pairs are captured into the hipGraphs as event-record nodes
    bool prof_suspend = false;       //   ... and eager passes (before a graph's capture) are not recorded
    std::vector<prof_rec> prof;
    std::vector<hipEvent_t> ev_pool;

    // row-split mat-muls: a stream, a completion event and (for the many-token kernel) a scratch area per peer device, made on first use
    struct split_peer { hipStream_t stream = nullptr; hipEvent_t done = nullptr; void * scratch = nullptr; size_t scratch_size = 0; };
    split_peer peers[GGML_MI355X_MAX_DEVICES];
    hipEvent_t split_ready = nullptr;
    hipEvent_t cpy_ev = nullptr;     // cpy_tensor_async to another backend: the destination stream waits on it
    bool split_graph = false;        // the graph being run reads row-split weights: eager execution (several devices' streams take part)
};

static hipEvent_t prof_event(mi_backend_ctx * c) {
    if (!c->ev_pool.empty()) { hipEvent_t e = c->ev_pool.back(); c->ev_pool.pop_back(); return e; }
    hipEvent_t e; MI_CHECK(hipEventCreate(&e)); return e;
}
static void prof_begin(mi_backend_ctx * c, int type, int64_t m, int64_t k, int64_t n, uint64_t bytes) {
    if (!c->profiling || c->prof_suspend) return;
    mi_backend_ctx::prof_rec r = { type, m, k, n, bytes, prof_event(c), prof_event(c) };
    MI_CHECK(hipEventRecord(r.e0, c->stream));
    c->prof.push_back(r);
}
static void prof_end(mi_backend_ctx * c) {
    if (!c->profiling || c->prof_suspend) return;
    MI_CHECK(hipEventRecord(c->prof.back().e1, c->stream));
}
// the grouped mat-vec module launches (possibly several graph nodes later, possibly several launches merged into one): it calls back
static void prof_hook_pre(void * ctx, int type, uint64_t wbytes, int n_merged, int64_t k) {
    // m < 0: grouped launch, |m| = KiB of weights; n = launches merged into this kernel
    mi_backend_ctx * c = (mi_backend_ctx *) ctx;
    if (!c->profiling || c->prof_suspend) return;
    if (c->prof_in_graph) { prof_begin(c, type, -(int64_t)(wbytes/1024), k, n_merged, wbytes); return; }
    // eager: the pair rides on the dispatch itself (kernel start -> kernel end, the interval rocprofv3's kernel trace reports)
    mi_backend_ctx::prof_rec r = { type, -(int64_t)(wbytes/1024), k, n_merged, wbytes, prof_event(c), prof_event(c) };
    mul_mat_vec_q_fused_set_launch_events(r.e0, r.e1);
    c->prof.push_back(r);
}
static void prof_hook_post(void * ctx, int, uint64_t, int, int64_t) {
    mi_backend_ctx * c = (mi_backend_ctx *) ctx;
    if (!c->profiling || c->prof_suspend) return;
    if (c->prof_in_graph) prof_end(c);
    c->prof.back().kernel = mul_mat_vec_q_fused_last_kernel();
}


static ggml_guid_t mi_guid(void) {
    static ggml_guid guid = { 0x4d, 0x49, 0x33, 0x35, 0x35, 0x58, 0x67, 0x66, 0x78, 0x39, 0x35, 0x30, 0x63, 0x64, 0x6e, 0x34 };
    return &guid;
}

static const char * be_get_name(ggml_backend_t backend) { return ((mi_backend_ctx *) backend->context)->name.c_str(); }

static void be_free(ggml_backend_t backend) {
    mi_backend_ctx * c = (mi_backend_ctx *) backend->context;
    set_device(c->device);
    if (c->stream) (void) hipStreamSynchronize(c->stream);
    for (auto & e : c->graphs) for (auto & sg : e.segs) if (sg.exec) (void) hipGraphExecDestroy(sg.exec);
    if (c->scratch) (void) hipFree(c->scratch);
    if (c->attn_part) (void) hipFree(c->attn_part);
    if (c->moe_ws) (void) hipFree(c->moe_ws);
    if (c->kv16) (void) hipFree(c->kv16);
    if (c->cvt) (void) hipFree(c->cvt);
    if (c->err_host) (void) hipHostFree(c->err_host);
    if (c->up.host) { (void) hipHostFree(c->up.host); (void) hipEventDestroy(c->up.ev[0]); (void) hipEventDestroy(c->up.ev[1]); }
    if (c->fin_img) (void) hipFree(c->fin_img);
    if (c->fin_cnt) (void) hipFree(c->fin_cnt);
    if (c->rope_tab) (void) hipFree(c->rope_tab);
    for (int i = 0; i < G().n_devices; i++) {
        mi_backend_ctx::split_peer & pr = c->peers[i];
        if (!pr.stream) continue;
        set_device(G().devices[i].id);
        (void) hipStreamSynchronize(pr.stream);
        if (pr.scratch) (void) hipFree(pr.scratch);
        (void) hipEventDestroy(pr.done); (void) hipStreamDestroy(pr.stream);
    }
    set_device(c->device);
    if (c->split_ready) (void) hipEventDestroy(c->split_ready);
    if (c->cpy_ev) (void) hipEventDestroy(c->cpy_ev);
    if (c->stream) (void) hipStreamDestroy(c->stream);
    delete c;
    delete backend;
}

// ---- small uploads, batched -------------------------------------------------------------------------------------
static void uploads_flush(mi_backend_ctx * c) {
    if (c->up.n == 0) return;
    upload_batch_launch(c->up.b, c->stream);
    c->cnt.kernels_launched++;
    c->up.n = 0;
}
static bool uploads_stage(mi_backend_ctx * c, void * dst, const void * data, size_t size) {
    auto & u = c->up;
    if (u.failed) return false;
    if (!u.host) {
        if (hipHostMalloc((void **) &u.host, 2*mi_backend_ctx::UP_HALF, hipHostMallocMapped) != hipSuccess || hipHostGetDevicePointer((void **) &u.dev, u.host, 0) != hipSuccess ||
            hipEventCreateWithFlags(&u.ev[0], hipEventDisableTiming) != hipSuccess || hipEventCreateWithFlags(&u.ev[1], hipEventDisableTiming) != hipSuccess) {
            (void) hipGetLastError(); u.failed = true; return false;
        }
    }
    const size_t need = (size + 255) & ~(size_t) 255;
    if (u.off + need > mi_backend_ctx::UP_HALF) {
        // this half is full: everything that reads it is on the stream once the pending batch is; the other half may be reused when ITS event has passed
        uploads_flush(c);
        MI_CHECK(hipEventRecord(u.ev[u.cur], c->stream)); u.ev_pending[u.cur] = true;
        u.cur ^= 1; u.off = 0;
        if (u.ev_pending[u.cur]) { MI_CHECK(hipEventSynchronize(u.ev[u.cur])); u.ev_pending[u.cur] = false; }
    }
    if (u.n == UPLOAD_BATCH_MAX) uploads_flush(c);
    const size_t o = (size_t) u.cur*mi_backend_ctx::UP_HALF + u.off;
    memcpy(u.host + o, data, size);
    u.b.src[u.n] = u.dev + o; u.b.dst[u.n] = dst; u.b.bytes[u.n] = (uint32_t) size; u.b.blocks[u.n] = (int)((size + 4095)/4096);
    u.n++; u.b.n = u.n;
    u.off += need;
    return true;
}

// `data` is HOST memory (pageable or pinned), as the interface says; device-to-device copies have entry points of their own below
static void be_set_tensor_async(ggml_backend_t backend, struct ggml_tensor * tensor, const void * data, size_t offset, size_t size) {
    mi_backend_ctx * c = (mi_backend_ctx *) backend->context;
    set_device(c->device);
    if (size == 0) return;
    static const bool batch = !getenv("GGML_MI355X_UPLOAD_BATCH") || atoi(getenv("GGML_MI355X_UPLOAD_BATCH")) != 0;
    if (batch && size <= mi_backend_ctx::UP_SMALL_MAX && uploads_stage(c, (char *) tensor->data + offset, data, size)) return;
    uploads_flush(c);       // stream order: what was set before this goes first
    MI_CHECK(hipMemcpyAsync((char *) tensor->data + offset, data, size, hipMemcpyHostToDevice, c->stream));
}
static void be_get_tensor_async(ggml_backend_t backend, const struct ggml_tensor * tensor, void * data, size_t offset, size_t size) {
    mi_backend_ctx * c = (mi_backend_ctx *) backend->context;
    set_device(c->device);
    uploads_flush(c);
    MI_CHECK(hipMemcpyAsync(data, (const char *) tensor->data + offset, size, hipMemcpyDeviceToHost, c->stream));
}

// layer-split hand-off: a point-to-point copy of [n_embd, n_tokens] F32 over one xGMI link,
// ordered after the producer stream's work and before the consumer stream's next op, no host sync.
static bool be_cpy_tensor_async(ggml_backend_t backend_src, ggml_backend_t backend_dst, const struct ggml_tensor * src, struct ggml_tensor * dst) {
    if (!ggml_backend_is_mi355x(backend_src) || !ggml_backend_is_mi355x(backend_dst)) return false;
    ggml_backend_buffer_t sbuf = src->view_src ? src->view_src->buffer : src->buffer;
    ggml_backend_buffer_t dbuf = dst->view_src ? dst->view_src->buffer : dst->buffer;
    if (!buffer_is_mi355x(sbuf) || !buffer_is_mi355x(dbuf)) return false;
    mi_backend_ctx * cs = (mi_backend_ctx *) backend_src->context;
    mi_backend_ctx * cd = (mi_backend_ctx *) backend_dst->context;
    mi_buffer_ctx * bs = (mi_buffer_ctx *) sbuf->context;
    mi_buffer_ctx * bd = (mi_buffer_ctx *) dbuf->context;
    if (cs->device != bs->device || cd->device != bd->device) return false;
    set_device(cs->device); uploads_flush(cs);
    if (backend_src != backend_dst) { set_device(cd->device); uploads_flush(cd); }
    if (backend_src != backend_dst) {
        set_device(cs->device);
        if (cs->device == cd->device) {
            MI_CHECK(hipMemcpyAsync(dst->data, src->data, ggml_nbytes(dst), hipMemcpyDeviceToDevice, cs->stream));
        } else {
            MI_CHECK(hipMemcpyPeerAsync(dst->data, cd->device, src->data, cs->device, ggml_nbytes(dst), cs->stream));
        }
        // make the destination stream wait for the copy: one event per source backend, re-recorded per copy (a wait captures the record that precedes it,
        // so re-recording for the next copy does not disturb a wait already queued)
        if (!cs->cpy_ev) MI_CHECK(hipEventCreateWithFlags(&cs->cpy_ev, hipEventDisableTiming));
        MI_CHECK(hipEventRecord(cs->cpy_ev, cs->stream));
        set_device(cd->device);
        MI_CHECK(hipStreamWaitEvent(cd->stream, cs->cpy_ev, 0));
    } else {
        set_device(cs->device);
        MI_CHECK(hipMemcpyAsync(dst->data, src->data, ggml_nbytes(dst), hipMemcpyDeviceToDevice, cs->stream));
    }
    return true;
}

static void be_synchronize(ggml_backend_t backend) {
    mi_backend_ctx * c = (mi_backend_ctx *) backend->context;
    set_device(c->device);
    uploads_flush(c);
    MI_CHECK(hipStreamSynchronize(c->stream));
    if (c->err_host && c->err_host[1] != 0) {
        GGML_ABORT("MI355X backend: the MoE router kernel timed out waiting for an expert's logit (GGML_MI355X_MOE_ROUTE_WIDE=0 selects the one-workgroup router)");
    }
}

// ---- op support -----------------------------------------------------------------------------------
static bool is_view_op(enum ggml_op op) {
    return op == GGML_OP_NONE || op == GGML_OP_RESHAPE || op == GGML_OP_VIEW || op == GGML_OP_PERMUTE || op == GGML_OP_TRANSPOSE;
}
static bool float_type(enum ggml_type t) { return t == GGML_TYPE_F32 || t == GGML_TYPE_F16 || t == GGML_TYPE_BF16; }
static bool fa_kv_type(enum ggml_type t) { return t == GGML_TYPE_F16 || t == GGML_TYPE_BF16 || t == GGML_TYPE_Q8_0 || t == GGML_TYPE_Q4_0; }
// FLASH_ATTN_EXT: the dense f16 copies its kernels need — the V cache transposed for the matrix-core prefill kernel (always: rows over cells are what it
// stages with coalesced loads), K when it is not f16; for a few tokens only what the decode kernel does not read directly (attn_decode_kv_types_fused)
static void fa_kv16_plan(const struct ggml_tensor * n, size_t & k_bytes, size_t & v_bytes) {
    const struct ggml_tensor * q = n->src[0]; const struct ggml_tensor * k = n->src[1]; const struct ggml_tensor * v = n->src[2];
    const size_t dense = (((size_t) k->ne[0]*k->ne[1]*k->ne[2]*2) + 255) & ~(size_t) 255;
    if (q->ne[1] > 8) { k_bytes = k->type != GGML_TYPE_F16 ? dense : 0; v_bytes = dense; }
    else if (attn_decode_kv_types_fused((int) k->type, (int) v->type)) { k_bytes = v_bytes = 0; }
    else { k_bytes = k->type != GGML_TYPE_F16 ? dense : 0; v_bytes = v->type != GGML_TYPE_F16 ? dense : 0; }
}

static float op_f32(const struct ggml_tensor * t, int i) { float f; memcpy(&f, &t->op_params[i], 4); return f; }

static bool mi_supports_op(const struct ggml_tensor * op) {
    const struct ggml_tensor * s0 = op->src[0];
    const struct ggml_tensor * s1 = op->src[1];
    // row-split weights: MUL_MAT's 2-D quantized src0 only
    for (int j = 0; j < GGML_MAX_SRC; j++) {
        if (!tensor_is_split(op->src[j])) continue;
        if (op->op != GGML_OP_MUL_MAT || j != 0 || !ggml_is_quantized(s0->type) || s0->ne[2] != 1 || s0->ne[3] != 1 || s1->ne[2] != 1 || s1->ne[3] != 1) return false;
    }
    switch (op->op) {
        case GGML_OP_NONE: case GGML_OP_RESHAPE: case GGML_OP_VIEW: case GGML_OP_PERMUTE: case GGML_OP_TRANSPOSE:
            return true;
        case GGML_OP_MUL_MAT: {
            if (op->type != GGML_TYPE_F32) return false;
            if (ggml_is_quantized(s0->type)) {
                // quantized weights x F32 activations: the hot path
                if (!mul_mat_vec_q_supported((int) s0->type)) return false;
                if (s1->type != GGML_TYPE_F32 && s1->type != GGML_TYPE_F16) return false;      // (F16 activations: converted to F32 first — tests/test-backend-ops.cpp:5715-5716)
                if (s0->ne[0] % ggml_blck_size(s0->type) != 0) return false;
                if (s1->nb[0] != ggml_type_size(s1->type)) return false;
                if (s0->nb[0] != ggml_type_size(s0->type)) return false;
                return true;
            }
            if (s0->type == GGML_TYPE_F16 || s0->type == GGML_TYPE_F32 || s0->type == GGML_TYPE_BF16) {
                return s1->type == GGML_TYPE_F32 || s1->type == GGML_TYPE_F16;
            }
            return false;
        }
        case GGML_OP_MUL_MAT_ID: {
            if (op->type != GGML_TYPE_F32 || s1->type != GGML_TYPE_F32) return false;
            if (s0->type == GGML_TYPE_F16 || s0->type == GGML_TYPE_F32 || s0->type == GGML_TYPE_BF16)      // an unquantized expert stack (tests/test-backend-ops.cpp:5821-5824)
                return op->src[2]->type == GGML_TYPE_I32 && op->nb[0] == 4;
            if (!ggml_is_quantized(s0->type) || !mul_mat_vec_q_supported((int) s0->type)) return false;
            if (s0->ne[0] % ggml_blck_size(s0->type) != 0) return false;
            if (s1->nb[0] != sizeof(float)) return false;
            return op->src[2]->type == GGML_TYPE_I32;
        }
        case GGML_OP_FLASH_ATTN_EXT: {     // head size 64 / 128; K / V cache types F16, BF16, Q8_0, Q4_0; ALiBi and logit soft-cap (src/llama-graph.cpp:1245-1265)
            const struct ggml_tensor * k = op->src[1]; const struct ggml_tensor * v = op->src[2]; const struct ggml_tensor * mask = op->src[3];
            if (s0->type != GGML_TYPE_F32 || !fa_kv_type(k->type) || !fa_kv_type(v->type) || op->type != GGML_TYPE_F32) return false;
            if (s0->ne[3] != 1 || k->ne[3] != 1 || v->ne[3] != 1 || v->ne[0] != k->ne[0] || s0->ne[2] % k->ne[2] != 0) return false;
            if (op_f32(op, 1) < 0.0f) return false;
            if (mask && (mask->type != GGML_TYPE_F16 || mask->ne[2] != 1 || mask->ne[3] != 1 || mask->ne[0] != k->ne[1] || mask->ne[1] < s0->ne[1] || mask->nb[1] % 16)) return false;
            if (op->nb[1] != (size_t) k->ne[0]*4) return false;
            // few tokens: the whole score row in LDS, or the cell ranges through the fixed partial-result buffer (8 tokens x 128 heads x 32 ranges)
            if (s0->ne[1] <= 8 && !attn_decode_supported(k->ne[0], k->ne[1]) &&
                !(attn_decode_supported_split(k->ne[0], k->ne[1]) && attn_decode_part_bytes(k->ne[0], k->ne[1], s0->ne[2], s0->ne[1]) <= (size_t) 8*128*32*130*4)) return false;
            if (s0->nb[0] != 4 || k->nb[0] != ggml_type_size(k->type) || v->nb[0] != ggml_type_size(v->type) || s0->nb[1] % 16 || s0->nb[2] % 16) return false;
            // (f16 rows are read 16 bytes at a time; block rows 2-byte aligned)
            if (k->nb[1] % (k->type == GGML_TYPE_F16 || k->type == GGML_TYPE_BF16 ? 16 : 2) || k->nb[2] % (k->type == GGML_TYPE_F16 || k->type == GGML_TYPE_BF16 ? 16 : 2)) return false;
            if (v->nb[1] % (v->type == GGML_TYPE_F16 || v->type == GGML_TYPE_BF16 ? 16 : 2) || v->nb[2] % (v->type == GGML_TYPE_F16 || v->type == GGML_TYPE_BF16 ? 16 : 2)) return false;
            return s0->ne[1] <= 8 ? (attn_decode_supported(k->ne[0], k->ne[1]) || attn_decode_supported_split(k->ne[0], k->ne[1])) : attn_prefill_supported(k->ne[0], k->ne[1]);
        }
        case GGML_OP_RMS_NORM:
            return s0->type == GGML_TYPE_F32 && op->type == GGML_TYPE_F32 && s0->nb[0] == sizeof(float);
        case GGML_OP_ADD: case GGML_OP_MUL: case GGML_OP_DIV: case GGML_OP_SUB:
            return float_type(s0->type) && float_type(s1->type) && float_type(op->type);
        case GGML_OP_ADD_ID:
            return s0->type == GGML_TYPE_F32 && s1->type == GGML_TYPE_F32 && op->src[2]->type == GGML_TYPE_I32;
        case GGML_OP_SCALE:
            return s0->type == GGML_TYPE_F32;
        case GGML_OP_CPY: case GGML_OP_CONT: case GGML_OP_DUP: {
            const enum ggml_type st = s0->type, dt = op->op == GGML_OP_CPY ? s1->type : op->type;
            return (float_type(st) && float_type(dt)) || (st == GGML_TYPE_I32 && dt == GGML_TYPE_I32);
        }
        case GGML_OP_SET_ROWS:
            if (s0->type != GGML_TYPE_F32 || s1->type != GGML_TYPE_I64) return false;
            if (op->type == GGML_TYPE_Q8_0 || op->type == GGML_TYPE_Q4_0) return s0->ne[0] % 32 == 0 && op->nb[0] == ggml_type_size(op->type);   // quantized KV cache rows
            return float_type(op->type);
        case GGML_OP_GET_ROWS:
            return (float_type(s0->type) || s0->type == GGML_TYPE_I32) && s1->type == GGML_TYPE_I32 && s0->nb[0] == ggml_type_size(s0->type);
        case GGML_OP_SUM_ROWS:
            return s0->type == GGML_TYPE_F32;
        case GGML_OP_ARGSORT:
            return s0->type == GGML_TYPE_F32 && s0->ne[0] <= 1024;
        case GGML_OP_UNARY:
            return float_type(s0->type) && (int) ggml_get_unary_op(op) <= (int) GGML_UNARY_OP_GELU_ERF;
        case GGML_OP_GLU:
            return float_type(s0->type) && (int) ggml_get_glu_op(op) < (int) GGML_GLU_OP_COUNT;
        case GGML_OP_ROPE: {
            const int mode = op->op_params[2];
            if (mode & (GGML_ROPE_TYPE_MROPE | 16)) return false;    // mrope / vision: not on the path
            return float_type(s0->type) && s0->type == op->type;
        }
        case GGML_OP_SOFT_MAX:
            return s0->type == GGML_TYPE_F32 && s0->nb[0] == sizeof(float) && (s1 == NULL || s1->type == GGML_TYPE_F32 || s1->type == GGML_TYPE_F16);
        default:
            return false;
    }
}

// ---- helpers ----------------------------------------------------------------------------------------
static tensor_desc desc(const struct ggml_tensor * t) {
    tensor_desc d;
    d.data = t->data; d.type = (int) t->type;
    for (int i = 0; i < 4; i++) { d.ne[i] = t->ne[i]; d.nb[i] = t->nb[i]; }
    return d;
}

static bool ranges_overlap(const void * a, size_t na, const void * b, size_t nb) {
    const char * pa = (const char *) a; const char * pb = (const char *) b;
    return pa < pb + nb && pb < pa + na;
}

// scratch needed by the largest quantized mat-mul in the graph
static size_t graph_scratch_need(const struct ggml_cgraph * g) {
    size_t need = 0;
    for (int i = 0; i < g->n_nodes; i++) {
        const struct ggml_tensor * n = g->nodes[i];
        if ((n->op == GGML_OP_MUL_MAT || n->op == GGML_OP_MUL_MAT_ID) && ggml_is_quantized(n->src[0]->type)) {
            const int kind = act_kind_for((int) n->src[0]->type);
            if (kind < 0) continue;
            const struct ggml_tensor * b = n->src[1];
            const int64_t rows = n->op == GGML_OP_MUL_MAT ? b->ne[1] : b->ne[1]*b->ne[2];
            size_t s = act_q8_bytes(kind, b->ne[0], n->op == GGML_OP_MUL_MAT && rows <= MMVQ_MAX_N ? rows*b->ne[2] : rows);    // (few columns: all batches quantized at once)
            if (n->op == GGML_OP_MUL_MAT && rows > MMVQ_MAX_N) s = mul_mat_q_scratch_bytes(b->ne[0], rows, n->src[0]->ne[1]) + mul_mat_q_x_bytes(b->ne[0], rows);   // + room for a producer's copy above its own
            if (n->op == GGML_OP_MUL_MAT_ID) {
                const struct ggml_tensor * ids = n->src[2];
                size_t sg = mul_mat_q_id_scratch_bytes(b->ne[0], b->ne[1], ids->ne[1], ids->ne[0], n->src[0]->ne[2]);
                if (sg > s) s = sg;
                if (b->ne[1] == 1 && ids->ne[0]*ids->ne[1] > 4*MMVQ_MAX_N) {      // the fused expert chain of a prompt pass keeps the GLU result (rows of this tensor's m) beside the input copy
                    sg = mul_mat_q_id_plan(nullptr, b->ne[0], ids->ne[1], ids->ne[0], n->src[0]->ne[2], n->src[0]->ne[1]).bytes;
                    if (sg > s) s = sg;
                }
            }
            if (s > need) need = s;
        } else if (n->op == GGML_OP_MUL_MAT && n->src[0]->type == GGML_TYPE_F16 && n->src[1]->type == GGML_TYPE_F32 && n->src[1]->ne[1] > MMVQ_MAX_N) {
            const size_t s = (size_t)((n->src[1]->ne[0] + 63) & ~(int64_t) 63)*(size_t) ggml_nrows(n->src[1])*2 + 1024;     // f16 copy of src1 (rows padded to 64) for the matrix-core attention products
            if (s > need) need = s;
        }
    }
    return need;
}

// quantize (or reuse) the activations of a quantized mat-mul: rows r -> x + (r % n_inner)*s_inner + (r / n_inner)*s_outer
static act_q8 get_act(mi_backend_ctx * c, const void * x, int64_t k, int64_t n_inner, int64_t n_outer, size_t s_inner, size_t s_outer, int kind) {
    if (c->aq.valid && c->aq.data == x && c->aq.k == k && c->aq.n_inner == n_inner && c->aq.n_outer == n_outer &&
        c->aq.s_inner == s_inner && c->aq.s_outer == s_outer && c->aq.kind == kind) {
        c->cnt.act_quant_reused++;
        return c->aq.q;
    }
    act_q8 q = act_q8_carve(c->scratch, kind, k, n_inner*n_outer);
    quantize_act((const float *) x, n_inner, s_inner, s_outer, q, c->stream);
    c->cnt.act_quant_launches++; c->cnt.kernels_launched++;
    c->aq = { x, k, n_inner, n_outer, s_inner, s_outer, kind, q, true,
              (size_t)(n_outer - 1)*s_outer + (size_t)(n_inner - 1)*s_inner + (size_t) k*4 };
    return q;
}

static constexpr int ACT_KIND_BF16 = -16;   // aq.kind of the dense bf16 copy the MFMA prefill kernel reads

// out/res: the prefill residual fusion (try_fused_prefill_add) writes W.x + res into `out` instead of W.x into dst
static void op_mul_mat(mi_backend_ctx * c, struct ggml_tensor * dst, struct ggml_tensor * out = nullptr, const struct ggml_tensor * res = nullptr, mmq_deferred * defer = nullptr) {
    const struct ggml_tensor * a = dst->src[0];
    const struct ggml_tensor * b = dst->src[1];
    struct ggml_tensor b32;
    if (ggml_is_quantized(a->type) && b->type == GGML_TYPE_F16) {
        // F16 activations (tests/test-backend-ops.cpp:5715-5716): a dense F32 copy first (the CPU backend converts src1 to the weights' vec_dot type row by row as well),
        // then the F32 path
        b32 = *b; b32.type = GGML_TYPE_F32; b32.data = c->cvt;
        b32.nb[0] = 4; b32.nb[1] = (size_t) b->ne[0]*4; b32.nb[2] = b32.nb[1]*b->ne[1]; b32.nb[3] = b32.nb[2]*b->ne[2];
        cpy(desc(b), desc(&b32), c->stream);
        c->cnt.kernels_launched++;
        c->aq.valid = false;
        b = &b32;
    }
    if (ggml_is_quantized(a->type)) {
        const int kind = act_kind_for((int) a->type);
        const int64_t K = a->ne[0], M = a->ne[1], N = b->ne[1];
        const int64_t r2 = b->ne[2]/a->ne[2], r3 = b->ne[3]/a->ne[3];
        // few columns x several batches (the K.q product over a quantized K cache: one batch per head): ONE launch, blockIdx.y = batch
        if (N <= MMVQ_MAX_N && b->ne[2] > 1 && b->ne[3] == 1 && a->ne[3] == 1 && act_q8_bytes(kind, K, N*b->ne[2]) <= c->scratch_size) {
            const act_q8 q = get_act(c, b->data, K, N, b->ne[2], b->nb[1], b->nb[2], kind);
            prof_begin(c, (int) a->type, M, K, N, (uint64_t) a->ne[2]*M*ggml_row_size(a->type, K));
            mul_mat_vec_q_batched((int) a->type, a->data, a->nb[1], a->nb[2], (int) r2, M, K, q, N, b->ne[2], (float *) dst->data, dst->nb[1], dst->nb[2], c->stream);
            prof_end(c);
            c->cnt.mmvq_launches++; c->cnt.kernels_launched++;
            c->cnt.weight_bytes += (uint64_t) a->ne[2]*M*ggml_row_size(a->type, K);
            return;
        }
        for (int64_t i13 = 0; i13 < b->ne[3]; i13++) {
            for (int64_t i12 = 0; i12 < b->ne[2]; i12++) {
                const char * bp = (const char *) b->data + i12*b->nb[2] + i13*b->nb[3];
                const char * W = (const char *) a->data + (i12/r2)*a->nb[2] + (i13/r3)*a->nb[3];
                float * d = (float *) ((char *) dst->data + i12*dst->nb[2] + i13*dst->nb[3]);
                prof_begin(c, (int) a->type, M, K, N, (uint64_t) M*ggml_row_size(a->type, K));
                if (N <= MMVQ_MAX_N) {
                    const act_q8 q = get_act(c, bp, K, N, 1, b->nb[1], 0, kind);
                    mul_mat_vec_q((int) a->type, W, a->nb[1], M, K, q, N, d, dst->nb[1], c->stream);
                    c->cnt.mmvq_launches++;
                } else {
                    // the scratch holds the bf16 copy of the activations; wq/wk/wv and gate/up read the same ones: convert once
                    const bool ready = c->aq.valid && c->aq.kind == ACT_KIND_BF16 && c->aq.data == bp && c->aq.k == K && c->aq.n_inner == N &&
                                       c->aq.s_inner == b->nb[1];
                    void * scr = (char *) c->scratch + (ready ? c->aq.off : 0);
                    if (out) mul_mat_q((int) a->type, W, a->nb[1], M, K, (const float *) bp, b->nb[1], N, scr, ready, (float *) out->data, out->nb[1],
                                       (const float *) res->data, res->ne[1] == 1 && N > 1 ? 0 : res->nb[1], c->stream, defer);     // (a bias row: the same addend for every token)
                    else     mul_mat_q((int) a->type, W, a->nb[1], M, K, (const float *) bp, b->nb[1], N, scr, ready, d, dst->nb[1], nullptr, 0, c->stream);
                    if (ready) c->cnt.act_quant_reused++;
                    else c->aq = { bp, K, N, 1, b->nb[1], 0, ACT_KIND_BF16, act_q8{}, true, (size_t)(N - 1)*b->nb[1] + (size_t) K*4 };
                    c->cnt.mmq_launches++; c->cnt.kernels_launched += ready ? 0 : 1;
                }
                prof_end(c);
                c->cnt.kernels_launched++;
                c->cnt.weight_bytes += (uint64_t) M*ggml_row_size(a->type, K);
            }
        }
        return;
    }
    mm_dense_args p;
    p.a = a->data; p.type_a = (int) a->type;
    p.ne00 = a->ne[0]; p.ne01 = a->ne[1]; p.ne02 = a->ne[2]; p.ne03 = a->ne[3];
    p.nb00 = a->nb[0]; p.nb01 = a->nb[1]; p.nb02 = a->nb[2]; p.nb03 = a->nb[3];
    p.b = b->data; p.type_b = (int) b->type;
    p.ne10 = b->ne[0]; p.ne11 = b->ne[1]; p.ne12 = b->ne[2]; p.ne13 = b->ne[3];
    p.nb10 = b->nb[0]; p.nb11 = b->nb[1]; p.nb12 = b->nb[2]; p.nb13 = b->nb[3];
    p.dst = (float *) dst->data; p.nb1 = dst->nb[1]; p.nb2 = dst->nb[2]; p.nb3 = dst->nb[3];
    if (mul_mat_dense_mfma_supported(p) && mul_mat_dense_mfma_scratch_bytes(p) <= c->scratch_size) {
        c->aq.valid = false;   // scratch reused
        mul_mat_dense_mfma(p, c->scratch, c->stream);
        c->cnt.kernels_launched += 2;
        return;
    }
    mul_mat_dense(p, c->stream);
    c->cnt.kernels_launched++;
}

// MUL_MAT on a row-split weight: the main device quantizes the activations once (few tokens) and runs its own slice; every other device that holds
// rows runs the same kernel on its slice on its own stream, reading the activations from and writing its rows of dst into the main device's
// memory through the peer mapping. Order: peers start after everything the main stream has queued so far (split_ready), the main stream
// continues after every peer's launch has finished (peer.done) — so dst is complete, and the scratch the peers read is not reused early.
static void op_mul_mat_split(mi_backend_ctx * c, struct ggml_tensor * dst) {
    const struct ggml_tensor * a = dst->src[0];
    const struct ggml_tensor * b = dst->src[1];
    const mi_split_tensor * e = (const mi_split_tensor *) a->extra;
    const int kind = act_kind_for((int) a->type);
    const int64_t K = a->ne[0], N = b->ne[1];
    const bool few = N <= MMVQ_MAX_N;
    act_q8 q = {};
    if (few) q = get_act(c, b->data, K, N, 1, b->nb[1], 0, kind);
    else c->aq.valid = false;
    if (!c->split_ready) MI_CHECK_G(hipEventCreateWithFlags(&c->split_ready, hipEventDisableTiming));
    MI_CHECK_G(hipEventRecord(c->split_ready, c->stream));
    const size_t need = c->scratch_size;
    bool waited[GGML_MI355X_MAX_DEVICES] = {};
    for (int i = 0; i < G().n_devices; i++) {
        const int64_t rows = e->row_lo[i + 1] - e->row_lo[i];
        if (rows == 0) continue;
        float * d = (float *) dst->data + e->row_lo[i];
        const bool local = i == c->dev_index;
        hipStream_t st = c->stream; void * scr = c->scratch;
        if (!local) {
            mi_backend_ctx::split_peer & pr = c->peers[i];
            set_device(G().devices[i].id);
            if (!pr.stream) { MI_CHECK_G(hipStreamCreateWithFlags(&pr.stream, hipStreamNonBlocking)); MI_CHECK_G(hipEventCreateWithFlags(&pr.done, hipEventDisableTiming)); }
            if (!few && pr.scratch_size < need) {
                MI_CHECK_G(hipStreamSynchronize(pr.stream));
                if (pr.scratch) MI_CHECK_G(hipFree(pr.scratch));
                pr.scratch = nullptr; pr.scratch_size = 0;
                MI_CHECK_G(hipMalloc(&pr.scratch, need)); pr.scratch_size = need;
            }
            MI_CHECK_G(hipStreamWaitEvent(pr.stream, c->split_ready, 0));
            st = pr.stream; scr = pr.scratch;
        }
        if (few) mul_mat_vec_q((int) a->type, e->data[i], a->nb[1], rows, K, q, N, d, dst->nb[1], st);
        else     mul_mat_q((int) a->type, e->data[i], a->nb[1], rows, K, (const float *) b->data, b->nb[1], N, scr, false, d, dst->nb[1], nullptr, 0, st);
        c->cnt.kernels_launched++; c->cnt.weight_bytes += (uint64_t) rows*a->nb[1];
        if (few) c->cnt.mmvq_launches++; else c->cnt.mmq_launches++;
        if (!local) { MI_CHECK_G(hipEventRecord(c->peers[i].done, st)); waited[i] = true; }
    }
    set_device(c->device);
    for (int i = 0; i < G().n_devices; i++) if (waited[i]) MI_CHECK_G(hipStreamWaitEvent(c->stream, c->peers[i].done, 0));
    c->cnt.split_mul_mats++;
}

static void op_mul_mat_id(mi_backend_ctx * c, struct ggml_tensor * dst) {
    const struct ggml_tensor * as  = dst->src[0];
    const struct ggml_tensor * b   = dst->src[1];
    const struct ggml_tensor * ids = dst->src[2];
    const int64_t K = as->ne[0], M = as->ne[1];
    const int64_t n_used = ids->ne[0], n_tokens = ids->ne[1], n_b = b->ne[1];
    if (!ggml_is_quantized(as->type)) {      // F16 / BF16 / F32 expert stack
        mul_mat_id_dense((int) as->type, as->data, as->nb[0], as->nb[1], as->nb[2], M, K, b->data, b->nb[0], b->nb[1], b->nb[2], n_b,
                         ids->data, ids->nb[0], ids->nb[1], n_used, n_tokens, as->ne[2], (float *) dst->data, dst->nb[1], dst->nb[2], c->stream);
        c->cnt.kernels_launched++;
        return;
    }
    const int kind = act_kind_for((int) as->type);
    // many tokens: sort the (token, slot) pairs by expert on the device and run the MFMA tile kernel per expert; the mat-vec kernel
    // below would read every expert matrix once per pair
    if (n_used*n_tokens > 4*MMVQ_MAX_N && mul_mat_q_id_supported(as->ne[2], n_used, n_tokens) && dst->nb[0] == sizeof(float)) {
        c->aq.valid = false;   // the scratch is reused
        mul_mat_q_id((int) as->type, as->data, as->nb[1], as->nb[2], M, K, (const float *) b->data, b->nb[1], b->nb[2], n_b,
                     (const int32_t *) ids->data, ids->nb[0], ids->nb[1], n_used, n_tokens, as->ne[2],
                     c->scratch, (float *) dst->data, dst->nb[1], dst->nb[2], c->stream);
        c->cnt.mmq_launches++; c->cnt.kernels_launched += 3;
        c->cnt.weight_bytes += (uint64_t) as->ne[2]*M*ggml_row_size(as->type, K);
        return;
    }
    const act_q8 q = get_act(c, b->data, K, n_b, n_tokens, b->nb[1], b->nb[2], kind);
    mul_mat_vec_q_id((int) as->type, as->data, as->nb[1], as->nb[2], M, K, q,
                     (const int32_t *) ids->data, ids->nb[0], ids->nb[1], n_used, n_tokens, n_b,
                     (float *) dst->data, dst->nb[1], dst->nb[2], c->stream);
    c->cnt.mmvq_launches++; c->cnt.kernels_launched++;
    c->cnt.weight_bytes += (uint64_t) n_used*n_tokens*M*ggml_row_size(as->type, K);
}

// the pending MoE combine is needed as a tensor after all (its reader is not the launch that would have evaluated it itself)
static void pp_flush(mi_backend_ctx * c) {
    if (!c->pp.active) return;
    moe_combine(c->pp.probs, c->pp.ids, c->pp.n_planes, c->pp.mode, c->pp.planes, (size_t) c->pp.stride*4, c->pp.m, c->pp.res, c->pp.x_out, c->stream);
    c->cnt.kernels_launched++;
    c->pp.active = false;
}

static void emit_mmv(mi_backend_ctx * c, const mmvq_group * grp, int nc, int64_t K, const mmvq_input & in, const mmvq_rope * rope, const mmvq_fin * fin) {
    mul_mat_vec_q_fused(grp, nc, K, in, rope, c->stream, fin);
    c->cnt.kernels_launched++;
}

// ---------------------------------------------------------------------------------------------------------------
// decode fusions (decode_fused.hip). Each matcher checks op, type, shape, layout AND that every skipped intermediate
// has exactly one consumer and is not a graph output; anything else falls back to node-by-node execution.
// ---------------------------------------------------------------------------------------------------------------
static int next_real(const struct ggml_cgraph * g, int i) {
    for (int j = i + 1; j < g->n_nodes; j++) {
        if (!is_view_op(g->nodes[j]->op) && !ggml_is_empty(g->nodes[j])) return j;
    }
    return -1;
}
static int n_uses(mi_backend_ctx * c, const struct ggml_tensor * t) {
    auto it = c->uses.find(t);
    return it == c->uses.end() ? 0 : it->second;
}
static bool is_internal(mi_backend_ctx * c, const struct ggml_tensor * t) {   // safe to leave unwritten
    return n_uses(c, t) == 1 && !(t->flags & GGML_TENSOR_FLAG_OUTPUT);
}
static bool is_row_vec_f32(const struct ggml_tensor * t) {   // [ne0, 1, 1, 1] contiguous f32, 16-byte aligned
    return t->type == GGML_TYPE_F32 && t->ne[1] == 1 && t->ne[2] == 1 && t->ne[3] == 1 && t->nb[0] == 4 && ((uintptr_t) t->data % 16) == 0;
}
static bool fusable_mmv(const struct ggml_tensor * n) {       // quantized weights x one f32 column
    if (n->op != GGML_OP_MUL_MAT) return false;
    const struct ggml_tensor * a = n->src[0]; const struct ggml_tensor * b = n->src[1];
    if (!ggml_is_quantized(a->type) || !mul_mat_vec_q_supported((int) a->type) || tensor_is_split(a)) return false;
    if (a->ne[2] != 1 || a->ne[3] != 1 || !is_row_vec_f32(b)) return false;
    if (a->nb[0] != ggml_type_size(a->type)) return false;
    return mul_mat_vec_q_fused_supported(a->ne[0], act_kind_for((int) a->type)) && a->ne[1] < (1 << 30);
}

struct mmv_chain { mmvq_group grp; int last; bool has_rope; mmvq_rope rope; const void * out_ptr; size_t out_bytes; };

// the chain of nodes that starts at mat-vec node i and can run as one group of the fused launch
static mmv_chain match_mmv_chain(mi_backend_ctx * c, const struct ggml_cgraph * g, int i) {
    struct ggml_tensor * n = g->nodes[i];
    const struct ggml_tensor * a = n->src[0];
    mmv_chain ch = {};
    ch.grp = { (const char *) a->data, nullptr, a->nb[1], (int) a->ne[1], (int) a->type, (float *) n->data, EPI_NONE, nullptr, nullptr, nullptr, 0, 0 };
    ch.last = i; ch.has_rope = false; ch.out_ptr = n->data; ch.out_bytes = ggml_nbytes(n);

    // MUL_MAT [-> ADD(bias)] -> RESHAPE -> ROPE: NORM pairs (src/llama-model.cpp:6017-6040), or NEOX pairs over a whole power-of-two head, with
    // gpt-oss's bias in between (src/llama-model.cpp:17636-17660)
    if (i + 2 < g->n_nodes && is_internal(c, n)) {
        int ir = i + 1;
        const struct ggml_tensor * bias = nullptr; const struct ggml_tensor * pre = n;
        if (g->nodes[ir]->op == GGML_OP_ADD && ir + 2 < g->n_nodes) {
            const struct ggml_tensor * ad = g->nodes[ir];
            const struct ggml_tensor * other = ad->src[0] == n ? ad->src[1] : (ad->src[1] == n ? ad->src[0] : nullptr);
            if (other && other != n && other->type == GGML_TYPE_F32 && ad->type == GGML_TYPE_F32 && ggml_are_same_shape(other, n) && ggml_are_same_shape(ad, n) &&
                ggml_is_contiguous(other) && ggml_is_contiguous(ad) && is_internal(c, ad)) { bias = other; pre = ad; ir++; }
        }
        struct ggml_tensor * rs = g->nodes[ir]; struct ggml_tensor * rp = g->nodes[ir + 1];
        const int rmode = rp->op == GGML_OP_ROPE ? rp->op_params[2] : -1;
        const bool neox_ok = rmode == 2 && rp->op_params[1] == rp->ne[0] && (rp->ne[0] & (rp->ne[0] - 1)) == 0 && rp->ne[0] >= 4 && a->ne[1] % rp->ne[0] == 0;
        if (rs->op == GGML_OP_RESHAPE && rs->src[0] == pre && is_internal(c, rs) && rp->op == GGML_OP_ROPE && rp->src[0] == rs &&
            rp->type == GGML_TYPE_F32 && ggml_is_contiguous(rp) && (rmode == 0 || neox_ok) && rp->ne[2] == 1 && rp->ne[3] == 1 &&
            rp->op_params[1] % 2 == 0 && rp->ne[0] % 2 == 0 && a->ne[1] % 2 == 0 && rp->src[1]->type == GGML_TYPE_I32) {
            ch.grp.dst = (float *) rp->data; ch.grp.epi = EPI_ROPE; ch.last = ir + 1; ch.has_rope = true;
            ch.grp.res = bias ? (const float *) bias->data : nullptr;
            ch.rope.pos = (const int32_t *) rp->src[1]->data;
            ch.rope.freq_factors = rp->src[2] ? (const float *) rp->src[2]->data : nullptr;
            ch.rope.head_dim = (int) rp->ne[0];
            ch.rope.p.n_dims = rp->op_params[1]; ch.rope.p.mode = rp->op_params[2]; ch.rope.p.n_ctx_orig = rp->op_params[4];
            ch.rope.p.freq_base = op_f32(rp, 5); ch.rope.p.freq_scale = op_f32(rp, 6); ch.rope.p.ext_factor = op_f32(rp, 7);
            ch.rope.p.attn_factor = op_f32(rp, 8); ch.rope.p.beta_fast = op_f32(rp, 9); ch.rope.p.beta_slow = op_f32(rp, 10);
            ch.out_ptr = rp->data; ch.out_bytes = ggml_nbytes(rp);
            return ch;
        }
    }
    const int j = next_real(g, i);
    if (j < 0) return ch;
    struct ggml_tensor * nx = g->nodes[j];
    // MUL_MAT -> ADD(residual): src/llama-model.cpp:6057,6096
    if (nx->op == GGML_OP_ADD && is_internal(c, n) && (nx->src[0] == n || nx->src[1] == n) && nx->type == GGML_TYPE_F32) {
        const struct ggml_tensor * other = nx->src[0] == n ? nx->src[1] : nx->src[0];
        if (other->type == GGML_TYPE_F32 && ggml_are_same_shape(other, n) && ggml_is_contiguous(other) && ggml_is_contiguous(nx) && ggml_are_same_shape(nx, n)) {
            ch.grp.dst = (float *) nx->data; ch.grp.epi = EPI_ADD; ch.grp.res = (const float *) other->data; ch.last = j;
            ch.out_ptr = nx->data; ch.out_bytes = ggml_nbytes(nx);
            // ... -> ADD again (gpt-oss: wo.x + bias, then + the residual stream, src/llama-graph.cpp:1479-1481 + llama-model.cpp:17676)
            const int j2 = next_real(g, j);
            if (j2 > 0 && is_internal(c, nx)) {
                struct ggml_tensor * n2 = g->nodes[j2];
                if (n2->op == GGML_OP_ADD && (n2->src[0] == nx || n2->src[1] == nx) && n2->src[0] != n2->src[1] && n2->type == GGML_TYPE_F32) {
                    const struct ggml_tensor * o2 = n2->src[0] == nx ? n2->src[1] : n2->src[0];
                    if (o2->type == GGML_TYPE_F32 && ggml_are_same_shape(o2, n) && ggml_is_contiguous(o2) && ggml_is_contiguous(n2) && ggml_are_same_shape(n2, n)) {
                        ch.grp.dst = (float *) n2->data; ch.grp.res2 = (const float *) o2->data; ch.last = j2;
                        ch.out_ptr = n2->data; ch.out_bytes = ggml_nbytes(n2);
                    }
                }
            }
            return ch;
        }
    }
    // MUL_MAT(gate) ; MUL_MAT(up) -> GLU(swiglu, gate, up), in either graph order (the DFS of ggml_build_forward_expand puts
    // GLU's src[0] = gate first): src/llama-graph.cpp:646-691
    if (fusable_mmv(nx) && nx->src[1] == n->src[1] && nx->src[0]->type == a->type && ggml_are_same_shape(nx->src[0], a) &&
        nx->src[0]->nb[1] == a->nb[1] && is_internal(c, n) && is_internal(c, nx)) {
        const int j2 = next_real(g, j);
        if (j2 > 0) {
            struct ggml_tensor * gl = g->nodes[j2];
            if (gl->op == GGML_OP_GLU && ggml_get_glu_op(gl) == GGML_GLU_OP_SWIGLU && gl->op_params[1] == 0 && gl->src[1] != NULL &&
                ((gl->src[0] == nx && gl->src[1] == n) || (gl->src[0] == n && gl->src[1] == nx)) && gl->type == GGML_TYPE_F32 && ggml_is_contiguous(gl)) {
                ch.grp.W  = (const char *) gl->src[0]->src[0]->data;   // gate weights (silu side)
                ch.grp.W2 = (const char *) gl->src[1]->src[0]->data;   // up weights
                ch.grp.dst = (float *) gl->data; ch.grp.epi = EPI_GLU; ch.last = j2;
                ch.out_ptr = gl->data; ch.out_bytes = ggml_nbytes(gl);
                return ch;
            }
        }
    }
    return ch;
}

static const struct ggml_tensor * base_of(const struct ggml_tensor * t) {   // strip view ops
    while (t && is_view_op(t->op) && t->op != GGML_OP_NONE && t->src[0]) t = t->src[0];
    return t;
}

// Run the mat-vec at node i together with the mat-vecs that directly follow it on the same activation.
//   norm/normw != NULL: the activation is MUL(RMS_NORM(norm->src[0]), normw) and the caller has checked that every consumer of
//   that product is a mat-vec; if they all fit into this launch the norm is computed in the prologue (PRO_NORM).
// Returns the index of the last node consumed (-1 = not fused).
static int compute_node(mi_backend_ctx * c, struct ggml_cgraph * g, int i);
static int try_fused_mmv(mi_backend_ctx * c, struct ggml_cgraph * g, int i, const struct ggml_tensor * norm, const struct ggml_tensor * normw) {
    struct ggml_tensor * n = g->nodes[i];
    if (!fusable_mmv(n)) return -1;
    int deferred[MMVQ_MAX_GROUPS]; int n_def = 0;      // ROPE nodes between the grouped mat-vecs that the epilogue cannot do (NEOX): run after the launch
    const struct ggml_tensor * b = n->src[1];
    const int kind = act_kind_for((int) n->src[0]->type);
    mmv_chain chains[MMVQ_MAX_GROUPS];
    int nc = 0, last = i, n_mm = 0;
    chains[nc++] = match_mmv_chain(c, g, i);
    last = chains[0].last;
    n_mm += chains[0].grp.epi == EPI_GLU ? 2 : 1;
    while (nc < MMVQ_MAX_GROUPS && chains[0].grp.epi != EPI_GLU) {
        int j = next_real(g, last);
        if (j < 0) break;
        struct ggml_tensor * m = g->nodes[j];
        int def_j = -1;
        if (m->op == GGML_OP_ROPE && !chains[nc - 1].has_rope && m->src[0] && base_of(m->src[0])->data == chains[nc - 1].out_ptr && n_def < MMVQ_MAX_GROUPS) {
            // gpt-oss: wq -> + bias -> RESHAPE -> ROPE(NEOX) ; wk -> ... ; wv (src/llama-model.cpp:17636-17660): the rotation of the chain just matched
            // only needs that chain — if the next mat-vec on the same activation follows it, it joins the launch and the ROPE runs afterwards
            const int j2 = next_real(g, j);
            if (j2 < 0 || !fusable_mmv(g->nodes[j2]) || g->nodes[j2]->src[1] != b) break;
            bool okr = !ranges_overlap(m->data, ggml_nbytes(m), b->data, ggml_nbytes(b));
            for (int q = 0; q < nc && okr; q++) okr = m->data == chains[q].out_ptr ? q == nc - 1 : !ranges_overlap(m->data, ggml_nbytes(m), chains[q].out_ptr, chains[q].out_bytes);
            if (!okr) break;
            def_j = j; j = j2; m = g->nodes[j];
        }
        if (!fusable_mmv(m) || m->src[1] != b) break;
        if (act_kind_for((int) m->src[0]->type) != kind) {
            // another activation format (Mixtral: wq Q4_K, wk / wv Q8_0) joins only a launch in which every workgroup quantizes the activation
            // itself: the norm prologue, or the plain quantizing one when no image of b is cached
            const bool cached_b = c->aq.valid && c->aq.data == b->data;
            if (!(norm || !cached_b) || !mul_mat_vec_q_fused_can_group_mixed(chains[0].grp.type, (int) m->src[0]->type) ||
                !mul_mat_vec_q_fused_prologue_supported(n->src[0]->ne[0], kind) || !mul_mat_vec_q_fused_prologue_supported(n->src[0]->ne[0], act_kind_for((int) m->src[0]->type))) break;
        }
        mmv_chain ch = match_mmv_chain(c, g, j);
        if (ch.grp.epi == EPI_GLU) break;   // the dual (GLU) kernel runs alone
        {   // at most two distinct weight types per launch, and only pairs that have a kernel
            int t2 = -1; bool ok_t = true;
            for (int q = 0; q < nc; q++) if (chains[q].grp.type != chains[0].grp.type) t2 = chains[q].grp.type;
            if (ch.grp.type != chains[0].grp.type) { if (t2 >= 0 && t2 != ch.grp.type) ok_t = false; else ok_t = mul_mat_vec_q_fused_can_group(chains[0].grp.type, ch.grp.type) || mul_mat_vec_q_fused_can_group_mixed(chains[0].grp.type, ch.grp.type); }
            if (!ok_t) break;
        }
        if (ch.has_rope && chains[0].has_rope && memcmp(&ch.rope, &chains[0].rope, sizeof(ch.rope)) != 0) break;   // one rope descriptor per launch
        // groups run concurrently: no output may alias another group's output, residual, or the shared activation
        bool ok = !ranges_overlap(ch.out_ptr, ch.out_bytes, b->data, ggml_nbytes(b));
        for (int q = 0; q < nc && ok; q++) {
            ok = !ranges_overlap(ch.out_ptr, ch.out_bytes, chains[q].out_ptr, chains[q].out_bytes);
            if (ok && chains[q].grp.res) ok = !ranges_overlap(ch.out_ptr, ch.out_bytes, chains[q].grp.res, (size_t) chains[q].grp.m*4);
            if (ok && ch.grp.res)       ok = !ranges_overlap(chains[q].out_ptr, chains[q].out_bytes, ch.grp.res, (size_t) ch.grp.m*4);
            if (ok && chains[q].grp.res2) ok = !ranges_overlap(ch.out_ptr, ch.out_bytes, chains[q].grp.res2, (size_t) chains[q].grp.m*4);
            if (ok && ch.grp.res2)        ok = !ranges_overlap(chains[q].out_ptr, chains[q].out_bytes, ch.grp.res2, (size_t) ch.grp.m*4);
        }
        // ... nor what a ROPE that now runs after the launch still has to write (its own chain's buffer excepted: in-place rotation)
        for (int r = 0; r < n_def + (def_j >= 0 ? 1 : 0) && ok; r++) {
            const struct ggml_tensor * rp = g->nodes[r < n_def ? deferred[r] : def_j];
            ok = !ranges_overlap(ch.out_ptr, ch.out_bytes, rp->data, ggml_nbytes(rp));
        }
        if (!ok) break;
        if (def_j >= 0) deferred[n_def++] = def_j;
        chains[nc++] = ch;
        last = ch.last;
        n_mm++;
    }

    // the KV-cache writes that follow wk / wv: SET_ROWS(k_cache, rope(k)) ; SET_ROWS(v_view[1,N], v[1,N]) — src/llama-kv-cache-unified.cpp:1123,1157-1167
    {
        const int j1 = next_real(g, last);
        const int j2 = j1 > 0 ? next_real(g, j1) : -1;
        if (j2 > 0 && g->nodes[j1]->op == GGML_OP_SET_ROWS && g->nodes[j2]->op == GGML_OP_SET_ROWS) {
            struct ggml_tensor * sk = g->nodes[j1]; struct ggml_tensor * sv = g->nodes[j2];
            const struct ggml_tensor * ks = sk->src[0]; const struct ggml_tensor * ki = sk->src[1];
            const struct ggml_tensor * vs = sv->src[0]; const struct ggml_tensor * vi = sv->src[1];
            int qk = -1, qv = -1;
            for (int q = 0; q < nc; q++) {
                if (base_of(ks)->data == chains[q].out_ptr && ks->data == chains[q].out_ptr) qk = q;
                if (base_of(vs)->data == chains[q].out_ptr && vs->data == chains[q].out_ptr) qv = q;
            }
            // a quantized K cache (-ctk q8_0 / q4_0): the row quantizer needs whole 32-element blocks, which no wave of this launch holds — K's SET_ROWS runs as
            // its own launch right after this one (deferred), V's store is still taken into the launch
            const bool k_quant = sk->type == GGML_TYPE_Q8_0 || sk->type == GGML_TYPE_Q4_0;
            const bool ok = qk >= 0 && qv >= 0 && qk != qv && (sk->type == GGML_TYPE_F16 || (k_quant && n_def < MMVQ_MAX_GROUPS)) && sv->type == GGML_TYPE_F16 &&
                ks->type == GGML_TYPE_F32 && vs->type == GGML_TYPE_F32 && ki->type == GGML_TYPE_I64 && vi->type == GGML_TYPE_I64 &&
                ggml_is_contiguous(ki) && ggml_is_contiguous(vi) && ggml_is_contiguous(ks) && ggml_is_contiguous(vs) &&
                ks->ne[1] == 1 && ks->ne[2] == 1 && ks->ne[3] == 1 && ks->ne[0] == chains[qk].grp.m && (k_quant || (sk->nb[0] == 2 && sk->nb[1] % 2 == 0)) &&
                vs->ne[2] == 1 && vs->ne[3] == 1 && qv >= 0 && (chains[qv].grp.epi == EPI_NONE || chains[qv].grp.epi == EPI_ADD) &&
                // V: element scatter on the transposed cache's [1, N] view (v_trans), or — with flash attention — a row like K (:1154)
                ((vs->ne[0] == 1 && vs->ne[1] == chains[qv].grp.m && sv->ne[0] == 1 && sv->nb[1] == 2 && ggml_nelements(vi) == chains[qv].grp.m) ||
                 (vs->ne[0] == chains[qv].grp.m && vs->ne[1] == 1 && sv->nb[0] == 2 && sv->nb[1] % 2 == 0 && ggml_nelements(vi) == 1)) &&
                // the cache rows written must not be read by anything inside the launch (they are not: only weights and the activation are)
                !ranges_overlap(sk->data, ggml_nbytes(sk), b->data, ggml_nbytes(b)) && !ranges_overlap(sv->data, ggml_nbytes(sv), b->data, ggml_nbytes(b));
            if (ok) {
                if (k_quant) deferred[n_def++] = j1;
                else {
                    chains[qk].grp.st16 = (uint16_t *) sk->data; chains[qk].grp.st_idx = (const int64_t *) ki->data;
                    chains[qk].grp.st_row_elems = (int64_t)(sk->nb[1]/2); chains[qk].grp.st_mode = 1;
                }
                chains[qv].grp.st16 = (uint16_t *) sv->data; chains[qv].grp.st_idx = (const int64_t *) vi->data;
                if (vs->ne[0] == 1) { chains[qv].grp.st_row_elems = 0; chains[qv].grp.st_mode = 2; }
                else                { chains[qv].grp.st_row_elems = (int64_t)(sv->nb[1]/2); chains[qv].grp.st_mode = 1; }
                last = j2;
            }
        }
    }

    mmvq_group grp[MMVQ_MAX_GROUPS];
    const mmvq_rope * rope = nullptr;
    uint64_t wbytes = 0;
    for (int q = 0; q < nc; q++) {
        grp[q] = chains[q].grp;
        if (chains[q].has_rope) rope = &chains[q].rope;
        wbytes += (uint64_t) grp[q].m*grp[q].row_stride*(grp[q].epi == EPI_GLU ? 2 : 1);
    }
    const int64_t K = n->src[0]->ne[0];
    mmvq_input in = {};
    in.act_kind = kind;
    const bool cached = c->aq.valid && c->aq.data == b->data && c->aq.k == K && c->aq.n_inner == 1 && c->aq.n_outer == 1 && c->aq.kind == kind;
    // The norm's product is left unwritten only where nobody else can look at it: not when the view handed to graph_compute ENDS at one of
    // its consumers (the scheduler's eval-callback cuts the graph there and the callback — tools/imatrix/imatrix.cpp:223-247 — reads the mat-mul's
    // src1), and not for result_norm, which llama_context reads back as the embeddings without an OUTPUT flag (src/llama-context.cpp:1137-1151)
    bool view_ends_here = false;
    for (int q = 0; q < g->n_nodes; q++) if (g->nodes[q]->op == GGML_OP_MUL_MAT && g->nodes[q]->src[1] == b && q == g->n_nodes - 1) view_ends_here = true;
    const bool keep_norm = norm && (view_ends_here || strncmp(b->name, "result_norm", 11) == 0 || strncmp(b->name, "result_embd", 11) == 0);
    if (norm && !keep_norm && n_mm == n_uses(c, b) && mul_mat_vec_q_fused_prologue_supported(K, kind) && ((uintptr_t) normw->data % 16) == 0) {
        in.mode = PRO_NORM; in.x = (const float *) norm->src[0]->data; in.norm_w = (const float *) normw->data; in.eps = op_f32(norm, 0);
    } else if (norm) {
        return -1;      // the caller runs the norm (+ quantization) as its own kernel, then comes back without `norm`
    } else if (cached) {
        in.mode = PRO_Q8; in.act = c->aq.q; c->cnt.act_quant_reused++;
    } else if (mul_mat_vec_q_fused_prologue_supported(K, kind)) {
        in.mode = PRO_QUANT; in.x = (const float *) b->data;
    } else {
        in.mode = PRO_Q8; in.act = get_act(c, b->data, K, 1, 1, b->nb[1], 0, kind);
    }
    // gate/up/SwiGLU whose output the down projection reads next (build_ffn, src/llama-graph.cpp:691-748): the launch also writes the quantized
    // image of its output (mmvq_fin) — the f32 tensor is written as always, so any other reader still finds it
    mmvq_fin fin = {}; act_q8 fin_q = {}; const struct ggml_tensor * fin_t = nullptr;
    static const bool fin_env = !getenv("GGML_MI355X_FIN") || atoi(getenv("GGML_MI355X_FIN")) != 0;
    if (fin_env && nc == 1 && grp[0].epi == EPI_GLU && !grp[0].eid && c->fin_img && !mul_mat_vec_q_stream_takes(grp, nc, K, in, rope)) {
        const int jn = next_real(g, last);
        const struct ggml_tensor * gl = g->nodes[last];
        const struct ggml_tensor * mm = jn > 0 ? g->nodes[jn] : nullptr;
        if (mm && fusable_mmv(mm) && mm->src[1] == gl && (void *) grp[0].dst == gl->data) {
            const int kind2 = act_kind_for((int) mm->src[0]->type);
            const int64_t M = grp[0].m;
            if (kind2 > 0 && mul_mat_vec_q_fused_fin_supported(M, K) && mul_mat_vec_q_fused_supported(M, kind2) && M/256 <= mi_backend_ctx::FIN_COUNTERS &&
                act_q8_bytes(kind2, M, 1) <= mi_backend_ctx::FIN_IMG_BYTES) {
                fin_q = act_q8_carve(c->fin_img, kind2, M, 1);
                fin = { kind2, 0, fin_q.qs, fin_q.d, fin_q.bsums, c->fin_cnt };
                fin_t = gl;
            }
        }
    }
    mmvq_rope rope_l;
    if (rope) {
        if (!c->rope_tab || rope->p.n_dims/2 > 512) return -1;
        if (!c->rope_tab_valid || memcmp(&c->rope_tab_key, rope, sizeof(*rope)) != 0) {
            mul_mat_vec_q_fused_rope_table(*rope, c->rope_tab, c->stream);
            c->cnt.kernels_launched++;
            c->rope_tab_key = *rope; c->rope_tab_valid = true;
        }
        rope_l = *rope; rope_l.table = c->rope_tab; rope = &rope_l;
    }
    if (c->pp.active && in.mode == PRO_NORM && (const void *) in.x == (const void *) c->pp.x_out && K == c->pp.m) {
        // the norm's input is the MoE combine left pending: this launch evaluates it in its prologue (and stores the sum). EVERY workgroup's prologue reads the
        // residual, the experts' outputs, the router's probabilities and the ids while other workgroups already store their rows — and by graph order all four are
        // dead after the combine, so an allocator that reuses memory (ggml-alloc) may have put this launch's outputs there: then the combine runs as
        // its own kernel first
        bool clash = false;
        const size_t planes_bytes = ((size_t)(c->pp.n_planes - 1)*c->pp.stride + (size_t) c->pp.m)*4;
        for (int q = 0; q < nc && !clash; q++) {
            const void * d = chains[q].out_ptr; const size_t nb = chains[q].out_bytes;
            clash = (c->pp.res && ranges_overlap(d, nb, c->pp.res, (size_t) c->pp.m*4)) || ranges_overlap(d, nb, c->pp.planes, planes_bytes) ||
                    ranges_overlap(d, nb, c->pp.probs, c->pp.probs_bytes) || (c->pp.ids && ranges_overlap(d, nb, c->pp.ids, (size_t) c->pp.n_planes*4));
        }
        if (clash) pp_flush(c);
        else {
            in.x = c->pp.res; in.planes = c->pp.planes; in.n_planes = c->pp.n_planes; in.plane_stride = c->pp.stride; in.x_out = c->pp.x_out;
            in.pl_probs = c->pp.probs; in.pl_ids = c->pp.ids; in.pl_mode = c->pp.mode;
            c->pp.active = false;
        }
    }
    emit_mmv(c, grp, nc, K, in, rope, fin_t ? &fin : nullptr);
    if (fin_t) {
        c->aq = { fin_t->data, grp[0].m, 1, 1, fin_t->nb[1], 0, fin.kind, fin_q, true, (size_t) grp[0].m*4, 0 };
        c->aq_fresh = true;
    }
    c->cnt.mmvq_launches++; c->cnt.weight_bytes += wbytes;
    for (int r = 0; r < n_def; r++) compute_node(c, g, deferred[r]);       // (flushes a held-back launch first)
    return last;
}

// the quantized many-token mat-mul that is the ONLY reader of t and directly follows node `last`, or NULL: its producer may then emit the
// bf16 activation copy itself (and skip the f32 tensor)
static const struct ggml_tensor * prefill_mm_consumer(mi_backend_ctx * c, const struct ggml_cgraph * g, int last, const struct ggml_tensor * t) {
    static const bool on = !getenv("GGML_MI355X_PREFILL_BF16_OUT") || atoi(getenv("GGML_MI355X_PREFILL_BF16_OUT")) != 0;
    const int j = next_real(g, last);
    if (!on || j < 0 || j == g->n_nodes - 1) return nullptr;      // (a view that ends at the consumer: the eval-callback case — its src1 must exist as f32)
    const struct ggml_tensor * n = g->nodes[j];
    if (n->op != GGML_OP_MUL_MAT || n->src[1] != t || !ggml_is_quantized(n->src[0]->type) || tensor_is_split(n->src[0]) || t->type != GGML_TYPE_F32 || t->ne[1] <= MMVQ_MAX_N || t->ne[2] != 1 || t->ne[3] != 1 ||
        n->src[0]->ne[2] != 1 || n->src[0]->ne[3] != 1 || t->nb[0] != 4 || t->ne[0] % 64 != 0 || !is_internal(c, t)) return nullptr;
    return n;
}

// SET_ROWS(k) immediately followed by SET_ROWS(v as [1, N] element scatter), f32 -> f16 (src/llama-kv-cache-unified.cpp:1123,1157-1167)
static int try_fused_kv_store(mi_backend_ctx * c, struct ggml_cgraph * g, int i) {
    struct ggml_tensor * sk = g->nodes[i];
    const int j = next_real(g, i);
    if (j < 0) return 0;
    struct ggml_tensor * sv = g->nodes[j];
    if (sv->op != GGML_OP_SET_ROWS) return 0;
    const struct ggml_tensor * ks = sk->src[0]; const struct ggml_tensor * ki = sk->src[1];
    const struct ggml_tensor * vs = sv->src[0]; const struct ggml_tensor * vi = sv->src[1];
    if (sk->type != GGML_TYPE_F16 || sv->type != GGML_TYPE_F16 || ks->type != GGML_TYPE_F32 || vs->type != GGML_TYPE_F32) return 0;
    if (ki->type != GGML_TYPE_I64 || vi->type != GGML_TYPE_I64 || !ggml_is_contiguous(ki) || !ggml_is_contiguous(vi)) return 0;
    if (ks->ne[2] != 1 || ks->ne[3] != 1 || vs->ne[2] != 1 || vs->ne[3] != 1 || ks->nb[0] != 4 || sk->nb[0] != 2) return 0;
    if (vs->ne[0] != 1 || vs->nb[1] != 4 || sv->ne[0] != 1 || sv->nb[1] != 2) return 0;
    if (ks->ne[0]*ks->ne[1] + vs->ne[1] >= (1ll << 31)) return 0;
    kv_store_f16((const float *) ks->data, ks->nb[1], (const int64_t *) ki->data, sk->data, sk->nb[1], ks->ne[0], ks->ne[1],
                 (const float *) vs->data, (const int64_t *) vi->data, sv->data, vs->ne[1], c->stream);
    c->cnt.kernels_launched++;
    return j - i + 1;
}

// MUL_MAT(k, q) -> SOFT_MAX -> MUL_MAT(v, kq) -> PERMUTE -> CONT (src/llama-graph.cpp:1283-1330)
static int try_fused_attn(mi_backend_ctx * c, struct ggml_cgraph * g, int i) {
    struct ggml_tensor * kq = g->nodes[i];
    const struct ggml_tensor * k = kq->src[0]; const struct ggml_tensor * q = kq->src[1];
    const bool kq8 = k->type == GGML_TYPE_Q8_0;       // -ctk q8_0: the decode kernel reads the blocks itself (<= 8 tokens); prompts keep the generic ops
    if ((k->type != GGML_TYPE_F16 && !kq8) || q->type != GGML_TYPE_F32 || k->ne[3] != 1 || q->ne[3] != 1) return 0;
    const int64_t hd = k->ne[0], n_kv = k->ne[1], n_head_kv = k->ne[2], T = q->ne[1], n_head = q->ne[2];
    const bool prefill = T > 8;      // many tokens: the matrix-core kernel with online softmax (attn_prefill.hip)
    if (kq8 && (prefill || k->nb[0] != 34 || k->nb[2] != (size_t) hd/32*34 || k->nb[1] % 2 || (uintptr_t) k->data % 2)) return 0;
    if (n_head % n_head_kv != 0 || !(prefill ? attn_prefill_supported(hd, n_kv) : (attn_decode_supported(hd, n_kv) || (c->attn_part && attn_decode_supported_split(hd, n_kv) && attn_decode_part_bytes(hd, n_kv, n_head, T) <= c->attn_part_bytes)))) return 0;
    if ((!kq8 && (k->nb[0] != 2 || k->nb[1] % 16 || k->nb[2] % 16 || (uintptr_t) k->data % 16)) || q->nb[0] != 4 || q->nb[1] % 16 || q->nb[2] % 16 || (uintptr_t) q->data % 16) return 0;
    const int j1 = next_real(g, i); if (j1 < 0) return 0;
    struct ggml_tensor * sm = g->nodes[j1];
    if (sm->op != GGML_OP_SOFT_MAX || sm->src[0] != kq || op_f32(sm, 1) != 0.0f || !is_internal(c, kq)) return 0;
    const struct ggml_tensor * mask = sm->src[1];
    if (mask && (mask->ne[0] != n_kv || mask->ne[2] != 1 || mask->ne[3] != 1 || !(mask->type == GGML_TYPE_F32 || mask->type == GGML_TYPE_F16))) return 0;
    const int j2 = next_real(g, j1); if (j2 < 0) return 0;
    struct ggml_tensor * kqv = g->nodes[j2];
    if (kqv->op != GGML_OP_MUL_MAT || kqv->src[1] != sm || !is_internal(c, sm)) return 0;
    const struct ggml_tensor * v = kqv->src[0];
    if (v->type != GGML_TYPE_F16 || v->nb[0] != 2 || v->ne[0] != n_kv || v->ne[1] != hd || v->ne[2] != n_head_kv || v->ne[3] != 1) return 0;
    // PERMUTE(0,2,1,3) view then CONT into [hd*n_head, T]
    const int j3 = next_real(g, j2); if (j3 < 0) return 0;
    struct ggml_tensor * ct = g->nodes[j3];
    if (ct->op != GGML_OP_CONT || !is_internal(c, kqv)) return 0;
    const struct ggml_tensor * pm = ct->src[0];
    if (pm->op != GGML_OP_PERMUTE || pm->src[0] != kqv || !is_internal(c, pm)) return 0;
    if (pm->ne[0] != hd || pm->ne[1] != n_head || pm->ne[2] != T || pm->ne[3] != 1) return 0;
    if (ct->type != GGML_TYPE_F32 || !ggml_is_contiguous(ct) || ggml_nelements(ct) != hd*n_head*T) return 0;
    if (prefill) {
        if (mask && (mask->ne[1] < T || ((uintptr_t) mask->data % 16) || mask->nb[1] % 16)) return 0;
        if (v->nb[1] % 8 || v->nb[2] % 8 || ((uintptr_t) v->data % 8)) return 0;
        // wo reads the result next: hand it the bf16 copy directly (nothing reads the scratch's offset 0 during this kernel)
        const struct ggml_tensor * wo = prefill_mm_consumer(c, g, j3, ct);
        const bool y16 = wo && mul_mat_q_scratch_bytes(hd*n_head, T, wo->src[0]->ne[1]) <= c->scratch_size;
        attn_prefill(q->data, q->nb[1], q->nb[2], k->data, k->nb[1], k->nb[2], v->data, v->nb[1], v->nb[2],
                     mask ? mask->data : nullptr, mask ? mask->nb[1] : 0, mask && mask->type == GGML_TYPE_F16,
                     sm->src[2] ? (const float *) sm->src[2]->data : nullptr, y16 ? nullptr : (float *) ct->data, (size_t) hd*n_head*4,
                     hd, n_kv, n_head, n_head_kv, T, op_f32(sm, 0), c->stream, true, y16 ? (uint16_t *) c->scratch : nullptr);
        if (y16) {
            c->aq = { ct->data, hd*n_head, T, 1, ct->nb[1], 0, ACT_KIND_BF16, act_q8{}, true, (size_t)(T - 1)*ct->nb[1] + (size_t) hd*n_head*4, 0 };
            c->aq_fresh = true;
        }
        c->cnt.kernels_launched++;
        return j3 - i + 1;
    }
    if (kq8) {
        const bool use_part = c->attn_part && attn_decode_part_bytes(hd, n_kv, n_head, T) <= c->attn_part_bytes;
        attn_decode(q->data, q->nb[1], q->nb[2], k->data, k->nb[1], k->nb[2], v->data, v->nb[1], v->nb[2], mask ? mask->data : nullptr, mask ? mask->nb[1] : 0,
                    mask && mask->type == GGML_TYPE_F16, sm->src[2] ? (const float *) sm->src[2]->data : nullptr, (float *) ct->data, (size_t) hd*n_head*4,
                    hd, n_kv, n_head, n_head_kv, T, op_f32(sm, 0), c->stream, true, use_part ? c->attn_part : nullptr, use_part ? c->attn_part_bytes : 0, true);
        c->cnt.kernels_launched++;
        return j3 - i + 1;
    }
    {
        const bool use_part = c->attn_part && attn_decode_part_bytes(hd, n_kv, n_head, T) <= c->attn_part_bytes;
        attn_decode(q->data, q->nb[1], q->nb[2], k->data, k->nb[1], k->nb[2], v->data, v->nb[1], v->nb[2], mask ? mask->data : nullptr, mask ? mask->nb[1] : 0,
                    mask && mask->type == GGML_TYPE_F16, sm->src[2] ? (const float *) sm->src[2]->data : nullptr, (float *) ct->data, (size_t) hd*n_head*4,
                    hd, n_kv, n_head, n_head_kv, T, op_f32(sm, 0), c->stream, true, use_part ? c->attn_part : nullptr, use_part ? c->attn_part_bytes : 0);
        c->cnt.kernels_launched++;
    }
    return j3 - i + 1;
}

// GET_ROWS(probs, selected) -> [SUM_ROWS -> DIV | SOFT_MAX] -> MUL(experts, weights) -> ADD ... ADD (sum over the used experts)
// [-> ADD residual], one token: the tail of build_moe_ffn (src/llama-graph.cpp:887-1012) as one kernel
static int try_fused_moe_combine(mi_backend_ctx * c, struct ggml_cgraph * g, int i) {
    struct ggml_tensor * gr = g->nodes[i];
    const struct ggml_tensor * pr = gr->src[0]; const struct ggml_tensor * ids = gr->src[1];
    if (gr->type != GGML_TYPE_F32 || pr->type != GGML_TYPE_F32 || ids->type != GGML_TYPE_I32) return 0;
    const int64_t n_used = ids->ne[0];
    if (gr->ne[0] != 1 || gr->ne[1] != n_used || gr->ne[2] != 1 || gr->ne[3] != 1 || n_used < 2 || n_used > 8) return 0;
    if (pr->ne[0] != 1 || pr->ne[2] != 1 || pr->ne[3] != 1 || pr->nb[1] != 4 || ids->ne[1] != 1 || ids->nb[0] != 4) return 0;
    if (!is_internal(c, gr)) return 0;
    int j = next_real(g, i); if (j < 0) return 0;
    int mode; const struct ggml_tensor * wsrc;      // wsrc: the tensor whose reshape is the MUL's weight operand
    struct ggml_tensor * n1 = g->nodes[j];
    if (n1->op == GGML_OP_SUM_ROWS && base_of(n1->src[0]) == gr && n1->src[0]->ne[0] == n_used && is_internal(c, n1)) {
        const int j2 = next_real(g, j); if (j2 < 0) return 0;
        struct ggml_tensor * dv = g->nodes[j2];
        if (dv->op != GGML_OP_DIV || base_of(dv->src[0]) != gr || dv->src[1] != n1 || dv->ne[0] != n_used || ggml_nelements(dv) != n_used) return 0;
        mode = 0; wsrc = dv; j = j2;
    } else if (n1->op == GGML_OP_SOFT_MAX && base_of(n1->src[0]) == gr && !n1->src[1] && !n1->src[2] && op_f32(n1, 0) == 1.0f && op_f32(n1, 1) == 0.0f &&
               n1->ne[0] == n_used && ggml_nelements(n1) == n_used) {
        mode = 1; wsrc = n1;
    } else return 0;
    if (n_uses(c, wsrc) != 1 || (wsrc->flags & GGML_TENSOR_FLAG_OUTPUT)) return 0;
    const int jm = next_real(g, j); if (jm < 0) return 0;
    struct ggml_tensor * mul = g->nodes[jm];
    if (mul->op != GGML_OP_MUL || mul->type != GGML_TYPE_F32) return 0;
    const struct ggml_tensor * ex = base_of(mul->src[1]) == wsrc ? mul->src[0] : (base_of(mul->src[0]) == wsrc ? mul->src[1] : nullptr);
    if (!ex || ex->type != GGML_TYPE_F32 || ex->ne[1] != n_used || ex->ne[2] != 1 || ex->ne[3] != 1 || ex->nb[0] != 4 || ex->ne[0] % 4 || ex->nb[1] % 16 ||
        ((uintptr_t) ex->data % 16) || !ggml_are_same_shape(ex, mul) || (mul->flags & GGML_TENSOR_FLAG_OUTPUT) || n_uses(c, mul) != (int) n_used) return 0;
    const int64_t n_embd = ex->ne[0];
    // the adds: ((view0 + view1) + view2) + ...
    int jl = jm; const struct ggml_tensor * prev = nullptr;
    for (int64_t u = 1; u < n_used; u++) {
        const int ja = next_real(g, jl); if (ja < 0) return 0;
        struct ggml_tensor * ad = g->nodes[ja];
        if (ad->op != GGML_OP_ADD || ad->type != GGML_TYPE_F32 || ad->ne[0] != n_embd || ggml_nelements(ad) != n_embd) return 0;
        const struct ggml_tensor * a0 = ad->src[0]; const struct ggml_tensor * a1 = ad->src[1];
        if (u == 1) { if (base_of(a0) != mul || a0->data != mul->data) return 0; }
        else if (a0 != prev) return 0;
        if (base_of(a1) != mul || (const char *) a1->data != (const char *) mul->data + u*mul->nb[1]) return 0;
        if (u + 1 < n_used && !is_internal(c, ad)) return 0;
        prev = ad; jl = ja;
    }
    struct ggml_tensor * out = g->nodes[jl];
    const float * res = nullptr;
    // + residual (src/llama-model.cpp:6096)
    if (is_internal(c, out)) {
        const int jr = next_real(g, jl);
        if (jr > 0) {
            struct ggml_tensor * ra = g->nodes[jr];
            if (ra->op == GGML_OP_ADD && ra->type == GGML_TYPE_F32 && ggml_nelements(ra) == n_embd && ggml_is_contiguous(ra) && (ra->src[0] == out || ra->src[1] == out)) {
                const struct ggml_tensor * r = ra->src[0] == out ? ra->src[1] : ra->src[0];
                if (r->type == GGML_TYPE_F32 && ggml_nelements(r) == n_embd && ggml_is_contiguous(r) && ((uintptr_t) r->data % 16) == 0 && ((uintptr_t) ra->data % 16) == 0) {
                    res = (const float *) r->data; out = ra; jl = jr;
                }
            }
        }
    }
    if (!ggml_is_contiguous(out) || ((uintptr_t) out->data % 16)) return 0;
    // the sum's first reader is the norm of a grouped mat-vec launch (the next layer's norm + QKV, or the final norm + lm_head): that launch's prologue
    // evaluates the combine itself (weighted planes, mmvq_stream.h) — this kernel and its boundary go. Anything else: pp_flush runs it after all.
    static const bool defer_on = !getenv("GGML_MI355X_MOE_COMBINE_DEFER") || atoi(getenv("GGML_MI355X_MOE_COMBINE_DEFER")) != 0;
    // the values this combine gathers are the ones the router just ranked (same tensors): its top-8 list is probs[ids[u]] already
    const bool direct = c->rt.valid && c->moe_ws && pr->data == c->rt.vals && ids->data == c->rt.sorted && ids->nb[0] == 4;
    {
        const int jn = next_real(g, jl);
        const int jm2 = jn > 0 ? next_real(g, jn) : -1;
        const int jq = jm2 > 0 ? next_real(g, jm2) : -1;
        // (every workgroup of that launch reads the residual and the experts' outputs while one of them stores the sum: the sum's tensor must not share memory
        // with either — an allocator that made the ADD in place, as ggml-alloc does when the residual has no later reader, keeps the stand-alone kernel)
        const bool aliased = (res && ranges_overlap(out->data, ggml_nbytes(out), res, (size_t) n_embd*4)) || ranges_overlap(out->data, ggml_nbytes(out), ex->data, ggml_nbytes(ex)) ||
                             ranges_overlap(out->data, ggml_nbytes(out), pr->data, ggml_nbytes(pr)) || ranges_overlap(out->data, ggml_nbytes(out), ids->data, ggml_nbytes(ids));
        if (defer_on && !aliased && jq > 0 && c->use_fusion && mul_mat_vec_q_stream_enabled() && n_embd <= 4096 && ex->nb[1] % 16 == 0 &&
            g->nodes[jn]->op == GGML_OP_RMS_NORM && g->nodes[jn]->src[0] == out && is_row_vec_f32(out) &&
            g->nodes[jm2]->op == GGML_OP_MUL && (g->nodes[jm2]->src[0] == g->nodes[jn] || g->nodes[jm2]->src[1] == g->nodes[jn]) &&
            fusable_mmv(g->nodes[jq]) && g->nodes[jq]->src[1] == g->nodes[jm2] && g->nodes[jq]->src[0]->ne[0] == n_embd &&
            (n_embd % 256 == 0 ? (g->nodes[jq]->src[0]->type == GGML_TYPE_Q4_K || g->nodes[jq]->src[0]->type == GGML_TYPE_Q5_K || g->nodes[jq]->src[0]->type == GGML_TYPE_Q6_K ||
                                  g->nodes[jq]->src[0]->type == GGML_TYPE_Q8_0 || g->nodes[jq]->src[0]->type == GGML_TYPE_Q4_0)
                               : (n_embd % 320 == 0 && (g->nodes[jq]->src[0]->type == GGML_TYPE_Q8_0 || g->nodes[jq]->src[0]->type == GGML_TYPE_MXFP4)))) {
            pp_flush(c);
            c->pp.active = true; c->pp.res = res; c->pp.x_out = (float *) out->data; c->pp.n_planes = (int) n_used; c->pp.m = n_embd;
            c->pp.planes = (const float *) ex->data; c->pp.stride = (int)(ex->nb[1]/4);
            c->pp.probs = direct ? c->moe_ws + 64 : (const float *) pr->data; c->pp.probs_bytes = direct ? 32 : ggml_nbytes(pr); c->pp.ids = direct ? nullptr : (const int32_t *) ids->data; c->pp.mode = mode;
            return jl - i + 1;
        }
    }
    moe_combine(direct ? c->moe_ws + 64 : (const float *) pr->data, direct ? nullptr : (const int32_t *) ids->data, (int) n_used, mode, ex->data, ex->nb[1], n_embd, res, (float *) out->data, c->stream);
    c->cnt.kernels_launched++;
    return jl - i + 1;
}

// MUL_MAT_ID for ONE token as grouped launches of the persistent mat-vec kernel, one group per used expert (the expert index is read
// on the device): up + gate + SwiGLU of build_moe_ffn in one launch (src/llama-graph.cpp:923-947), any other one-token MUL_MAT_ID
// (the down projection: one activation vector per expert, :981) in one launch with the f32 -> int8 quantization in its prologue
static bool moe_mmv_ok(const struct ggml_tensor * n) {
    if (n->op != GGML_OP_MUL_MAT_ID) return false;
    const struct ggml_tensor * as = n->src[0]; const struct ggml_tensor * b = n->src[1]; const struct ggml_tensor * ids = n->src[2];
    if (!ggml_is_quantized(as->type) || !mul_mat_vec_q_supported((int) as->type) || as->nb[0] != ggml_type_size(as->type) || as->ne[3] != 1) return false;
    if (ids->type != GGML_TYPE_I32 || ids->ne[1] != 1 || ids->ne[2] != 1 || ids->nb[0] != 4 || ids->ne[0] < 1 || ids->ne[0] > MMVQ_MAX_GROUPS) return false;
    if (b->type != GGML_TYPE_F32 || b->ne[2] != 1 || b->ne[3] != 1 || b->nb[0] != 4 || !(b->ne[1] == 1 || b->ne[1] == ids->ne[0])) return false;
    if (((uintptr_t) b->data % 16) || (b->ne[1] > 1 && b->nb[1] % 16) || n->type != GGML_TYPE_F32 || n->nb[0] != 4) return false;
    const int64_t K = as->ne[0];
    return mul_mat_vec_q_fused_supported(K, act_kind_for((int) as->type)) && mul_mat_vec_q_fused_prologue_supported(K, act_kind_for((int) as->type)) && as->ne[1] < (1 << 30);
}
static int try_fused_moe_experts(mi_backend_ctx * c, struct ggml_cgraph * g, int i) {
    struct ggml_tensor * up = g->nodes[i];
    if (!moe_mmv_ok(up)) return 0;
    const struct ggml_tensor * as = up->src[0]; const struct ggml_tensor * b = up->src[1]; const struct ggml_tensor * ids = up->src[2];
    const int n_used = (int) ids->ne[0];
    const int64_t K = as->ne[0], M = as->ne[1];
    mmvq_input in = {};
    in.act_kind = act_kind_for((int) as->type); in.mode = PRO_QUANT; in.x = (const float *) b->data;
    mmvq_group grp[MMVQ_MAX_GROUPS];
    // up, gate (in either order: the graph lists a node's sources depth-first), each optionally followed by its ADD_ID bias, then
    // swiglu_split(gate, up) or swiglu_oai(gate, up) (src/llama-graph.cpp:923-968)
    if (b->ne[1] == 1 && is_internal(c, up)) {
        auto bias_of = [&](struct ggml_tensor * mm, int & j) -> struct ggml_tensor * {      // the ADD_ID that consumes mm, if it is next
            const int jn = next_real(g, j);
            if (jn < 0) return nullptr;
            struct ggml_tensor * ad = g->nodes[jn];
            if (ad->op != GGML_OP_ADD_ID || ad->src[0] != mm || ad->src[2] != ids || ad->type != GGML_TYPE_F32 || !is_internal(c, ad)) return nullptr;
            const struct ggml_tensor * bt = ad->src[1];
            if (bt->type != GGML_TYPE_F32 || bt->ne[0] != M || bt->nb[0] != 4 || bt->nb[1] != (size_t) M*4 || bt->ne[1] != as->ne[2]) return nullptr;
            j = jn;
            return ad;
        };
        int j = i;
        struct ggml_tensor * a_b = bias_of(up, j);
        const int jo = next_real(g, j);
        struct ggml_tensor * other = jo > 0 ? g->nodes[jo] : nullptr;
        if (other && moe_mmv_ok(other) && other->src[1] == b && other->src[2] == ids && other->src[0]->type == as->type && other->src[0]->ne[0] == K &&
            other->src[0]->ne[1] == M && other->src[0]->nb[1] == as->nb[1] && other->src[0]->nb[2] == as->nb[2] && is_internal(c, other)) {
            j = jo;
            struct ggml_tensor * o_b = bias_of(other, j);
            const int jg = next_real(g, j);
            struct ggml_tensor * gl = jg > 0 ? g->nodes[jg] : nullptr;
            struct ggml_tensor * a_out = a_b ? a_b : up; struct ggml_tensor * o_out = o_b ? o_b : other;
            if (gl && gl->op == GGML_OP_GLU && gl->op_params[1] == 0 && (!a_b) == (!o_b) &&
                (ggml_get_glu_op(gl) == GGML_GLU_OP_SWIGLU || ggml_get_glu_op(gl) == GGML_GLU_OP_SWIGLU_OAI) &&
                ((gl->src[0] == o_out && gl->src[1] == a_out) || (gl->src[0] == a_out && gl->src[1] == o_out)) &&
                gl->type == GGML_TYPE_F32 && gl->nb[0] == 4 && gl->ne[0] == M && gl->ne[1] == n_used && ggml_nelements(gl) == M*n_used &&
                !ranges_overlap(gl->data, ggml_nbytes(gl), b->data, ggml_nbytes(b))) {
                // out = act(src0) * src1: src0 is the gate side
                const bool a_is_gate = gl->src[0] == a_out;
                const struct ggml_tensor * mm_gate = a_is_gate ? up : other; const struct ggml_tensor * mm_up = a_is_gate ? other : up;
                const struct ggml_tensor * bg = a_is_gate ? a_b : o_b; const struct ggml_tensor * bu = a_is_gate ? o_b : a_b;
                const bool oai = ggml_get_glu_op(gl) == GGML_GLU_OP_SWIGLU_OAI;
                for (int u = 0; u < n_used; u++) {
                    grp[u] = { (const char *) mm_gate->src[0]->data, (const char *) mm_up->src[0]->data, as->nb[1], (int) M, (int) as->type,
                               (float *) ((char *) gl->data + (size_t) u*gl->nb[1]), EPI_GLU, nullptr, nullptr, nullptr, 0, 0,
                               (const int32_t *) ids->data + u, as->nb[2], 0,
                               bg ? (const float *) bg->src[1]->data : nullptr, bu ? (const float *) bu->src[1]->data : nullptr,
                               oai ? op_f32(gl, 2) : 0.0f, oai ? op_f32(gl, 3) : 0.0f };
                }
                mul_mat_vec_q_fused(grp, n_used, K, in, nullptr, c->stream);
                c->cnt.mmvq_launches++; c->cnt.kernels_launched++;
                c->cnt.weight_bytes += (uint64_t) 2*n_used*M*ggml_row_size(as->type, K);
                return jg - i + 1;
            }
        }
    }
    // a single one-token MUL_MAT_ID, optionally followed by its ADD_ID bias (gpt-oss's ffn_down_exps.bias, src/llama-graph.cpp:979-983)
    if (ranges_overlap(up->data, ggml_nbytes(up), b->data, ggml_nbytes(b))) return 0;
    struct ggml_tensor * out = up; const struct ggml_tensor * bias = nullptr; int consumed = 1;
    {
        const int jn = next_real(g, i);
        struct ggml_tensor * ad = jn > 0 ? g->nodes[jn] : nullptr;
        if (ad && ad->op == GGML_OP_ADD_ID && ad->src[0] == up && ad->src[2] == ids && ad->type == GGML_TYPE_F32 && is_internal(c, up) && ggml_are_same_shape(ad, up) &&
            ad->nb[0] == 4 && ad->nb[1] == up->nb[1] && ad->src[1]->type == GGML_TYPE_F32 && ad->src[1]->ne[0] == M && ad->src[1]->nb[0] == 4 &&
            ad->src[1]->nb[1] == (size_t) M*4 && ad->src[1]->ne[1] == as->ne[2] && !ranges_overlap(ad->data, ggml_nbytes(ad), b->data, ggml_nbytes(b))) {
            out = ad; bias = ad->src[1]; consumed = jn - i + 1;
        }
    }
    for (int u = 0; u < n_used; u++) {
        grp[u] = { (const char *) as->data, nullptr, as->nb[1], (int) M, (int) as->type, (float *) ((char *) out->data + (size_t) u*out->nb[1]), bias ? EPI_ADD : EPI_NONE,
                   bias ? (const float *) bias->data : nullptr, nullptr, nullptr, 0, 0, (const int32_t *) ids->data + u, as->nb[2], b->ne[1] > 1 ? (int)((size_t) u*b->nb[1]/4) : 0 };
        grp[u].res_eid = bias ? 1 : 0;
    }
    mul_mat_vec_q_fused(grp, n_used, K, in, nullptr, c->stream);
    c->cnt.mmvq_launches++; c->cnt.kernels_launched++;
    c->cnt.weight_bytes += (uint64_t) n_used*M*ggml_row_size(as->type, K);
    return consumed;
}

// Prompt pass: the slot sum that ends build_moe_ffn (src/llama-graph.cpp:996-1012) — ADD(view 0, view 1), ADD(., view 2), ... over the n_used slot views of the
// weighted experts' outputs — and the residual ADD behind it (src/llama-model.cpp:6096) as one pass instead of n_used launches
static int try_fused_slot_sum(mi_backend_ctx * c, struct ggml_cgraph * g, int i) {
    struct ggml_tensor * ad = g->nodes[i];
    const struct ggml_tensor * v0 = ad->src[0]; const struct ggml_tensor * v1 = ad->src[1];
    if (ad->type != GGML_TYPE_F32 || !v0 || !v1 || !v0->view_src || v0->view_src != v1->view_src) return 0;
    const struct ggml_tensor * ex = v0->view_src;
    const int64_t M = ex->ne[0], n_used = ex->ne[1], T = ex->ne[2];
    if (ex->type != GGML_TYPE_F32 || ex->ne[3] != 1 || n_used < 2 || n_used > 8 || T < 2 || M % 4 || ex->nb[0] != 4 || ex->nb[1] % 16 || ex->nb[2] % 16 || ((uintptr_t) ex->data % 16)) return 0;
    auto is_slot = [&](const struct ggml_tensor * v, int64_t u) {
        return v->view_src == ex && v->type == GGML_TYPE_F32 && v->ne[0] == M && v->ne[1] == T && v->ne[2] == 1 && v->ne[3] == 1 && v->nb[0] == 4 && v->nb[1] == ex->nb[2] &&
               (const char *) v->data == (const char *) ex->data + u*ex->nb[1];
    };
    if (!is_slot(v0, 0) || !is_slot(v1, 1)) return 0;
    int jl = i; const struct ggml_tensor * prev = ad;
    for (int64_t u = 2; u < n_used; u++) {
        if (!is_internal(c, prev)) return 0;
        const int ja = next_real(g, jl); if (ja < 0) return 0;
        const struct ggml_tensor * a2 = g->nodes[ja];
        if (a2->op != GGML_OP_ADD || a2->type != GGML_TYPE_F32 || a2->src[0] != prev || !is_slot(a2->src[1], u)) return 0;
        prev = a2; jl = ja;
    }
    struct ggml_tensor * out = g->nodes[jl];
    const struct ggml_tensor * res = nullptr;
    if (is_internal(c, out)) {
        const int jr = next_real(g, jl);
        struct ggml_tensor * ra = jr > 0 ? g->nodes[jr] : nullptr;
        if (ra && ra->op == GGML_OP_ADD && ra->type == GGML_TYPE_F32 && (ra->src[0] == out || ra->src[1] == out)) {
            const struct ggml_tensor * r = ra->src[0] == out ? ra->src[1] : ra->src[0];
            if (r != out && r->type == GGML_TYPE_F32 && ggml_are_same_shape(r, out) && ggml_are_same_shape(ra, out) && r->nb[0] == 4 && r->nb[1] % 16 == 0 && ((uintptr_t) r->data % 16) == 0) {
                res = r; out = ra; jl = jr;
            }
        }
    }
    if (out->ne[0] != M || out->ne[1] != T || out->ne[2] != 1 || out->ne[3] != 1 || out->nb[0] != 4 || out->nb[1] % 16 || ((uintptr_t) out->data % 16)) return 0;
    // every element is read and written by the same lane, so `out` may be the residual's or slot 0's own memory (an in-place ADD) — but not a shifted overlap
    if ((out->data != ex->data || out->nb[1] != ex->nb[2]) && ranges_overlap(out->data, ggml_nbytes(out), ex->data, ggml_nbytes(ex))) return 0;
    if (res && (out->data != res->data || out->nb[1] != res->nb[1]) && ranges_overlap(out->data, ggml_nbytes(out), res->data, ggml_nbytes(res))) return 0;
    moe_slot_sum(ex->data, ex->nb[1], ex->nb[2], (int) n_used, M, T, res ? (const float *) res->data : nullptr, res ? res->nb[1] : 0, (float *) out->data, out->nb[1], c->stream);
    c->cnt.kernels_launched++;
    return jl - i + 1;
}

// Prompt pass (many tokens) through the experts of build_moe_ffn (src/llama-graph.cpp:914-990):
//   MUL_MAT_ID(up) [ADD_ID] ; MUL_MAT_ID(gate) [ADD_ID] ; GLU (swiglu | swiglu_oai) ; MUL_MAT_ID(down) [ADD_ID] [MUL weights]
// as: ONE bf16 copy of the layer input, ONE sort of the (token, slot) pairs by expert, ONE dual tile launch (both products of an expert's pairs against one
// activation tile, biases and the GLU in its epilogue, result as bf16 rows per pair) and ONE tile launch for the down projection reading those rows (bias and routing
// weight in its epilogue). Node by node the same work is 3 copies + 3 sorts + 3 tile launches + up to 5 element kernels, and every expert's gate and up tiles stage the same
// activation rows twice. The down half is taken only if it follows directly; intermediates are skipped only if nobody else reads them.
static int compute_node(mi_backend_ctx * c, struct ggml_cgraph * g, int i);
static bool moe_mmq_ok(const struct ggml_tensor * n) {
    if (n->op != GGML_OP_MUL_MAT_ID) return false;
    const struct ggml_tensor * as = n->src[0]; const struct ggml_tensor * b = n->src[1]; const struct ggml_tensor * ids = n->src[2];
    if (!ggml_is_quantized(as->type) || act_kind_for((int) as->type) < 0 || as->nb[0] != ggml_type_size(as->type) || as->ne[3] != 1) return false;
    if (ids->type != GGML_TYPE_I32 || ids->ne[2] != 1 || ids->ne[3] != 1) return false;
    const int64_t n_used = ids->ne[0], n_tokens = ids->ne[1];
    if (n_used*n_tokens <= 4*MMVQ_MAX_N || !mul_mat_q_id_supported(as->ne[2], n_used, n_tokens)) return false;
    if (b->type != GGML_TYPE_F32 || b->nb[0] != 4 || b->ne[2] != n_tokens || b->ne[3] != 1 || !(b->ne[1] == 1 || b->ne[1] == n_used)) return false;
    return n->type == GGML_TYPE_F32 && n->nb[0] == 4 && n->ne[0] == as->ne[1] && n->ne[1] == n_used && n->ne[2] == n_tokens && n->ne[3] == 1;
}
static int try_fused_prefill_moe(mi_backend_ctx * c, struct ggml_cgraph * g, int i) {
    static const bool on = !getenv("GGML_MI355X_PREFILL_MOE") || atoi(getenv("GGML_MI355X_PREFILL_MOE")) != 0;
    if (!on) return 0;
    struct ggml_tensor * up = g->nodes[i];
    if (!moe_mmq_ok(up) || up->src[1]->ne[1] != 1 || !is_internal(c, up)) return 0;
    const struct ggml_tensor * as = up->src[0]; const struct ggml_tensor * b = up->src[1]; const struct ggml_tensor * ids = up->src[2];
    const int64_t K = as->ne[0], M = as->ne[1], E = as->ne[2], n_used = ids->ne[0], n_tokens = ids->ne[1];
    if (M % 64 != 0) return 0;                  // the GLU rows are the down projection's bf16 activation rows: no padding between them
    auto bias_of = [&](struct ggml_tensor * mm, const struct ggml_tensor * w, int & j) -> struct ggml_tensor * {      // the ADD_ID that consumes mm, if it is next
        const int jn = next_real(g, j);
        if (jn < 0) return nullptr;
        struct ggml_tensor * ad = g->nodes[jn];
        if (ad->op != GGML_OP_ADD_ID || ad->src[0] != mm || ad->src[2] != ids || ad->type != GGML_TYPE_F32 || !ggml_are_same_shape(ad, mm) || ad->nb[0] != 4) return nullptr;
        const struct ggml_tensor * bt = ad->src[1];
        if (bt->type != GGML_TYPE_F32 || bt->ne[0] != w->ne[1] || bt->nb[0] != 4 || bt->nb[1] % 4 || bt->ne[1] != w->ne[2] || !is_internal(c, mm)) return nullptr;
        j = jn;
        return ad;
    };
    int j = i;
    struct ggml_tensor * a_b = bias_of(up, as, j);
    if (a_b && !is_internal(c, a_b)) return 0;
    const int jo = next_real(g, j);
    struct ggml_tensor * other = jo > 0 ? g->nodes[jo] : nullptr;
    if (!other || !moe_mmq_ok(other) || other->src[1] != b || other->src[2] != ids || other->src[0]->type != as->type || other->src[0]->ne[0] != K || other->src[0]->ne[1] != M ||
        other->src[0]->ne[2] != E || other->src[0]->nb[1] != as->nb[1] || other->src[0]->nb[2] != as->nb[2] || !is_internal(c, other)) return 0;
    j = jo;
    struct ggml_tensor * o_b = bias_of(other, other->src[0], j);
    if ((o_b && !is_internal(c, o_b)) || (!a_b) != (!o_b)) return 0;
    if (a_b && a_b->src[1]->nb[1] != o_b->src[1]->nb[1]) return 0;
    const int jg = next_real(g, j);
    struct ggml_tensor * gl = jg > 0 ? g->nodes[jg] : nullptr;
    struct ggml_tensor * a_out = a_b ? a_b : up; struct ggml_tensor * o_out = o_b ? o_b : other;
    if (!gl || gl->op != GGML_OP_GLU || gl->op_params[1] != 0 || (ggml_get_glu_op(gl) != GGML_GLU_OP_SWIGLU && ggml_get_glu_op(gl) != GGML_GLU_OP_SWIGLU_OAI) ||
        !((gl->src[0] == o_out && gl->src[1] == a_out) || (gl->src[0] == a_out && gl->src[1] == o_out)) || gl->type != GGML_TYPE_F32 || gl->nb[0] != 4 ||
        gl->ne[0] != M || gl->ne[1] != n_used || gl->ne[2] != n_tokens || gl->ne[3] != 1) return 0;
    const bool a_is_gate = gl->src[0] == a_out;           // out = act(src0) * src1: src0 is the gate side
    const struct ggml_tensor * w_gate = (a_is_gate ? up : other)->src[0]; const struct ggml_tensor * w_up = (a_is_gate ? other : up)->src[0];
    const struct ggml_tensor * bg = a_is_gate ? a_b : o_b; const struct ggml_tensor * bu = a_is_gate ? o_b : a_b;
    const bool oai = ggml_get_glu_op(gl) == GGML_GLU_OP_SWIGLU_OAI;

    // the down half: MUL_MAT_ID(down, gl, ids) [ADD_ID] [MUL weights], directly behind the GLU
    int consumed_to = jg, between_from = 0, between_to = 0;       // [between_from, between_to): the routing weights' nodes between the down product and its MUL
    struct ggml_tensor * down = nullptr; struct ggml_tensor * d_out = nullptr; const struct ggml_tensor * d_bias = nullptr; const struct ggml_tensor * d_scale = nullptr;
    {
        const int jd = next_real(g, jg);
        struct ggml_tensor * dn = jd > 0 ? g->nodes[jd] : nullptr;
        if (dn && moe_mmq_ok(dn) && dn->src[1] == gl && dn->src[2] == ids && dn->src[0]->ne[0] == M && dn->src[0]->ne[2] == E && dn->nb[0] == 4) {
            down = dn; d_out = dn; consumed_to = jd;
            int jj = jd;
            struct ggml_tensor * ad = bias_of(dn, dn->src[0], jj);
            if (ad) { d_out = ad; d_bias = ad->src[1]; consumed_to = jj; }
            // MUL(experts, weights): the graph lists a node's sources depth-first, so the routing weights' own nodes (GET_ROWS of the probabilities, their
            // normalisation: src/llama-graph.cpp:887-908) sit BETWEEN the down product and the MUL. They do not depend on anything of this chain: they are
            // computed first (node by node), then the down launch applies the weight in its epilogue.
            int jm = jj; bool between_ok = true;
            for (int hop = 0; hop < 12; hop++) {
                jm = next_real(g, jm);
                if (jm < 0) break;
                const struct ggml_tensor * t = g->nodes[jm];
                if (t->op == GGML_OP_MUL && t->src[0] == d_out) break;
                const bool light = t->op == GGML_OP_GET_ROWS || t->op == GGML_OP_SOFT_MAX || t->op == GGML_OP_SUM_ROWS || t->op == GGML_OP_DIV || t->op == GGML_OP_SCALE;
                bool reads_chain = false;
                for (int q = 0; q < GGML_MAX_SRC; q++) {
                    const struct ggml_tensor * sq = t->src[q];
                    while (sq) { if (sq == up || sq == other || sq == a_out || sq == o_out || sq == gl || sq == dn || sq == d_out) reads_chain = true; sq = sq->view_src; }
                }
                if (!light || reads_chain) { between_ok = false; break; }
            }
            struct ggml_tensor * ml = jm > 0 && between_ok ? g->nodes[jm] : nullptr;
            if (ml && ml->op == GGML_OP_MUL && ml->src[0] == d_out && is_internal(c, d_out) && ml->type == GGML_TYPE_F32 && ggml_are_same_shape(ml, d_out) && ml->nb[0] == 4) {
                const struct ggml_tensor * wt = ml->src[1];
                if (wt->type == GGML_TYPE_F32 && wt->ne[0] == 1 && wt->ne[1] == n_used && wt->ne[2] == n_tokens && wt->ne[3] == 1 &&
                    !ranges_overlap(ml->data, ggml_nbytes(ml), wt->data, ggml_nbytes(wt))) { d_scale = wt; d_out = ml; between_from = jj + 1; between_to = jm; consumed_to = jm; }
            }
        }
    }
    const mmq_moe_plan pl = mul_mat_q_id_plan(c->scratch, K, n_tokens, n_used, E, M);
    if (pl.bytes > c->scratch_size) return 0;
    const bool gl_f32 = !down || !is_internal(c, gl);      // somebody else reads the GLU result as a tensor
    c->aq.valid = false;     // the scratch is reused
    // Few pairs per expert (gpt-oss at 512 tokens: 64): the dual kernel — one 256-pair tile per expert, eight decoding waves, both tensors against one activation tile
    // (pp512 18.7k -> 22.2k tok/s). Many (Mixtral: ~128): the dual kernel's single workgroup per CU decodes at half the rate of two 128-pair workgroups (926 us against
    // 2 x 350 - 390 + 62 for the GLU kernel), so the two products stay separate launches of the 128-pair kernel: the up launch writes its tensor, the gate launch reads
    // it back in its epilogue and evaluates the GLU there.
    static const int dual_env = getenv("GGML_MI355X_MOE_DUAL") ? atoi(getenv("GGML_MI355X_MOE_DUAL")) : -1;
    const int dual_opt = c->moe_dual >= 0 ? c->moe_dual : dual_env;
    const bool dual = dual_opt >= 0 ? dual_opt != 0 : n_used*n_tokens <= 96*E;
    const int tile = dual ? 256 : mul_mat_q_id_tile(n_used, n_tokens, E);
    if (!dual && gl_f32) {      // the gate launch reads the up tensor while it writes the GLU tensor: the same element by the same lane if they are one buffer, a race if they overlap otherwise
        const struct ggml_tensor * u_chk = a_is_gate ? o_out : a_out;
        if (gl->data != u_chk->data && ranges_overlap(gl->data, ggml_nbytes(gl), u_chk->data, ggml_nbytes(u_chk))) return 0;
    }
    mul_mat_q_id_act16((const float *) b->data, b->nb[1], b->nb[2], K, 1, n_tokens, pl.xb, c->stream);
    mul_mat_q_id_sort((const int32_t *) ids->data, ids->nb[0], ids->nb[1], n_used, n_tokens, E, tile, pl.table, c->stream);
    mmq_moe_epi eg;
    eg.oai = oai ? 1 : 0; eg.alpha = oai ? op_f32(gl, 2) : 0.0f; eg.limit = oai ? op_f32(gl, 3) : 0.0f;
    if (dual) {
        if (bg) { eg.bias = (const float *) bg->src[1]->data; eg.bias2 = (const float *) bu->src[1]->data; eg.bias_stride = bg->src[1]->nb[1]/4; }
        mul_mat_q_id_tiles((int) as->type, w_gate->data, w_up->data, as->nb[1], as->nb[2], M, K, pl.xb, 1, pl.table, tile, n_used, n_tokens, E, eg,
                           gl_f32 ? (float *) gl->data : nullptr, gl->nb[1], gl->nb[2], down ? pl.y16 : nullptr, c->stream);
        c->cnt.mmq_launches++; c->cnt.kernels_launched += 3;
    } else {
        struct ggml_tensor * u_out = a_is_gate ? o_out : a_out;       // the up side's last tensor: its memory holds the up product (+ bias) between the two launches
        mmq_moe_epi eu;
        if (bu) { eu.bias = (const float *) bu->src[1]->data; eu.bias_stride = bu->src[1]->nb[1]/4; }
        mul_mat_q_id_tiles((int) as->type, w_up->data, nullptr, as->nb[1], as->nb[2], M, K, pl.xb, 1, pl.table, tile, n_used, n_tokens, E, eu,
                           (float *) u_out->data, u_out->nb[1], u_out->nb[2], nullptr, c->stream);
        if (bg) { eg.bias = (const float *) bg->src[1]->data; eg.bias_stride = bg->src[1]->nb[1]/4; }
        eg.glu_up = (const float *) u_out->data; eg.glu_up_nb1 = u_out->nb[1]; eg.glu_up_nb2 = u_out->nb[2];
        mul_mat_q_id_tiles((int) as->type, w_gate->data, nullptr, as->nb[1], as->nb[2], M, K, pl.xb, 1, pl.table, tile, n_used, n_tokens, E, eg,
                           gl_f32 ? (float *) gl->data : nullptr, gl->nb[1], gl->nb[2], down ? pl.y16 : nullptr, c->stream);
        c->cnt.mmq_launches += 2; c->cnt.kernels_launched += 4;
    }
    c->cnt.weight_bytes += (uint64_t) 2*E*M*ggml_row_size(as->type, K);
    if (down) {
        if (between_to > between_from) {
            const bool uf = c->use_fusion; c->use_fusion = false;
            for (int q = between_from; q < between_to; ) q += compute_node(c, g, q);
            c->use_fusion = uf;
        }
        const struct ggml_tensor * ad = down->src[0];
        mmq_moe_epi ed;
        if (d_bias) { ed.bias = (const float *) d_bias->data; ed.bias_stride = d_bias->nb[1]/4; }
        if (d_scale) { ed.scale = (const float *) d_scale->data; ed.scale_nb0 = d_scale->nb[1]; ed.scale_nb1 = d_scale->nb[2]; }
        const int tile_d = mul_mat_q_id_tile(n_used, n_tokens, E);
        if (tile_d != tile) mul_mat_q_id_sort((const int32_t *) ids->data, ids->nb[0], ids->nb[1], n_used, n_tokens, E, tile_d, pl.table2, c->stream);
        mul_mat_q_id_tiles((int) ad->type, ad->data, nullptr, ad->nb[1], ad->nb[2], ad->ne[1], M, pl.y16, n_used, tile_d != tile ? pl.table2 : pl.table, tile_d, n_used, n_tokens, E, ed,
                           (float *) d_out->data, d_out->nb[1], d_out->nb[2], nullptr, c->stream);
        c->cnt.mmq_launches++; c->cnt.kernels_launched += tile_d != tile ? 2 : 1;
        c->cnt.weight_bytes += (uint64_t) E*ad->ne[1]*ggml_row_size(ad->type, M);
    }
    return consumed_to - i + 1;
}

// MUL_MAT(F32 router weights, x) [-> ADD bias] [-> SOFT_MAX] -> ARGSORT desc, one token (src/llama-graph.cpp:838-883)
// norm / normw != NULL: x (= lg->src[1]) is MUL(RMS_NORM(norm->src[0]), normw), not computed yet: the router kernel computes it and writes it
static int try_fused_moe_route(mi_backend_ctx * c, struct ggml_cgraph * g, int i, const struct ggml_tensor * norm = nullptr, const struct ggml_tensor * normw = nullptr) {
    struct ggml_tensor * lg = g->nodes[i];
    const struct ggml_tensor * w = lg->src[0]; const struct ggml_tensor * x = lg->src[1];
    if (w->type != GGML_TYPE_F32 || x->type != GGML_TYPE_F32 || lg->type != GGML_TYPE_F32) return 0;
    const int64_t K = w->ne[0], E = w->ne[1];
    if (E < 2 || E > 256 || K % 4 || K >= (1ll << 30) || w->ne[2] != 1 || w->ne[3] != 1 || w->nb[0] != 4 || w->nb[1] % 16 || ((uintptr_t) w->data % 16)) return 0;
    if (!is_row_vec_f32(x) || x->ne[0] != K || !ggml_is_contiguous(lg) || ggml_nelements(lg) != E) return 0;
    int j = next_real(g, i); if (j < 0) return 0;
    const struct ggml_tensor * cur = lg; const float * bias = nullptr; float * logits_out = nullptr;
    struct ggml_tensor * nd = g->nodes[j];
    if (nd->op == GGML_OP_ADD && (nd->src[0] == cur || nd->src[1] == cur) && ggml_is_contiguous(nd) && ggml_nelements(nd) == E && is_internal(c, cur)) {
        const struct ggml_tensor * bt = nd->src[0] == cur ? nd->src[1] : nd->src[0];
        if (bt->type != GGML_TYPE_F32 || !ggml_is_contiguous(bt) || ggml_nelements(bt) != E) return 0;
        bias = (const float *) bt->data; cur = nd;
        j = next_real(g, j); if (j < 0) return 0;
        nd = g->nodes[j];
    }
    float * probs_out = nullptr; bool softmax = false;
    if (nd->op == GGML_OP_SOFT_MAX && nd->src[0] == cur && !nd->src[1] && !nd->src[2] && op_f32(nd, 0) == 1.0f && op_f32(nd, 1) == 0.0f &&
        ggml_is_contiguous(nd) && ggml_nelements(nd) == E && is_internal(c, cur)) {
        softmax = true; probs_out = (float *) nd->data; cur = nd;
        j = next_real(g, j); if (j < 0) return 0;
        nd = g->nodes[j];
    } else {
        logits_out = (float *) cur->data;      // the logits themselves are gathered later (SOFTMAX_WEIGHT gating)
    }
    if (nd->op != GGML_OP_ARGSORT || nd->src[0] != cur || nd->op_params[0] != GGML_SORT_ORDER_DESC || nd->type != GGML_TYPE_I32 ||
        !ggml_is_contiguous(nd) || ggml_nelements(nd) != E) return 0;
    static const bool topv_on = !getenv("GGML_MI355X_MOE_TOPV") || atoi(getenv("GGML_MI355X_MOE_TOPV")) != 0;
    float * topv = topv_on && c->moe_ws ? c->moe_ws + 64 : nullptr;
    if (norm) {
        if (!moe_route_norm_supported(K, E, c->moe_ws) || ((uintptr_t) normw->data % 16) || ((uintptr_t) norm->src[0]->data % 16) || ((uintptr_t) x->data % 16)) return 0;
        moe_route((const float *) w->data, w->nb[1], (const float *) norm->src[0]->data, bias, K, E, softmax, logits_out, probs_out, (int32_t *) nd->data, c->stream, c->moe_ws,
                  (const float *) normw->data, op_f32(norm, 0), (float *) x->data, c->err_dev, topv);
    } else {
        moe_route((const float *) w->data, w->nb[1], (const float *) x->data, bias, K, E, softmax, logits_out, probs_out, (int32_t *) nd->data, c->stream, c->moe_ws, nullptr, 0.0f, nullptr, c->err_dev, topv);
    }
    c->rt.valid = topv != nullptr; c->rt.vals = softmax ? (const void *) probs_out : (const void *) logits_out; c->rt.sorted = nd->data;
    c->cnt.kernels_launched++;
    return j - i + 1;
}

// Prefill: MUL_MAT(gate) ; MUL_MAT(up) -> GLU(swiglu) on the same many-token activations (build_ffn, src/llama-graph.cpp:646-691) as one
// matrix-core kernel (mmq.hip DUAL): neither product is written, the GLU kernel is gone and the activation tile is staged once.
static int try_fused_prefill_glu(mi_backend_ctx * c, struct ggml_cgraph * g, int i) {
    static const bool on = !getenv("GGML_MI355X_PREFILL_GLU") || atoi(getenv("GGML_MI355X_PREFILL_GLU")) != 0;
    if (!on) return 0;
    struct ggml_tensor * n = g->nodes[i];
    const struct ggml_tensor * a = n->src[0]; const struct ggml_tensor * b = n->src[1];
    if (!ggml_is_quantized(a->type) || b->type != GGML_TYPE_F32 || b->ne[1] <= MMVQ_MAX_N || b->ne[2] != 1 || b->ne[3] != 1 || a->ne[2] != 1 || a->ne[3] != 1 ||
        b->nb[0] != 4 || !mul_mat_q_glu_supported(a->ne[1], b->ne[1]) || !is_internal(c, n)) return 0;
    const int j = next_real(g, i); if (j < 0) return 0;
    struct ggml_tensor * nx = g->nodes[j];
    if (nx->op != GGML_OP_MUL_MAT || nx->src[1] != b || tensor_is_split(nx->src[0]) || nx->src[0]->type != a->type || !ggml_are_same_shape(nx->src[0], a) || nx->src[0]->nb[1] != a->nb[1] ||
        !is_internal(c, nx)) return 0;
    const int j2 = next_real(g, j); if (j2 < 0) return 0;
    struct ggml_tensor * gl = g->nodes[j2];
    if (gl->op != GGML_OP_GLU || ggml_get_glu_op(gl) != GGML_GLU_OP_SWIGLU || gl->op_params[1] != 0 || gl->src[1] == NULL || gl->type != GGML_TYPE_F32 ||
        !ggml_is_contiguous(gl) || !((gl->src[0] == nx && gl->src[1] == n) || (gl->src[0] == n && gl->src[1] == nx))) return 0;
    const int64_t K = a->ne[0], M = a->ne[1], N = b->ne[1];
    if (mul_mat_q_scratch_bytes(K, N, M) > c->scratch_size) return 0;
    const bool ready = c->aq.valid && c->aq.kind == ACT_KIND_BF16 && c->aq.data == b->data && c->aq.k == K && c->aq.n_inner == N && c->aq.s_inner == b->nb[1] && c->aq.off == 0;
    // ffn_down reads the result next: the kernel writes its bf16 copy (above this kernel's own activation copy) instead of the f32 tensor
    const struct ggml_tensor * down = prefill_mm_consumer(c, g, j2, gl);
    const size_t off2 = mul_mat_q_x_bytes(K, N);
    const bool y16 = down && off2 + mul_mat_q_scratch_bytes(M, N, down->src[0]->ne[1]) <= c->scratch_size;
    prof_begin(c, (int) a->type, 2*M, K, N, (uint64_t) 2*M*ggml_row_size(a->type, K));
    mul_mat_q_glu((int) a->type, gl->src[0]->src[0]->data, gl->src[1]->src[0]->data, a->nb[1], M, K, (const float *) b->data, b->nb[1], N,
                  c->scratch, ready, y16 ? nullptr : (float *) gl->data, gl->nb[1], c->stream, y16 ? (uint16_t *) ((char *) c->scratch + off2) : nullptr);
    prof_end(c);
    if (ready) c->cnt.act_quant_reused++;
    if (y16) {
        c->aq = { gl->data, M, N, 1, gl->nb[1], 0, ACT_KIND_BF16, act_q8{}, true, (size_t)(N - 1)*gl->nb[1] + (size_t) M*4, off2 };
        c->aq_fresh = true;
    } else if (!ready) c->aq = { b->data, K, N, 1, b->nb[1], 0, ACT_KIND_BF16, act_q8{}, true, (size_t)(N - 1)*b->nb[1] + (size_t) K*4 };
    c->cnt.mmq_launches++; c->cnt.kernels_launched += ready ? 1 : 2;
    c->cnt.weight_bytes += (uint64_t) 2*M*ggml_row_size(a->type, K);
    return j2 - i + 1;
}

static int compute_node(mi_backend_ctx * c, struct ggml_cgraph * g, int i);

// Prefill: the mat-muls that directly follow each other on the same many-token activations — wq / wk / wv of build_attn
// (src/llama-model.cpp:6017-6040), each optionally followed by its (RESHAPE ->) ROPE — as ONE launch of 256-token tiles: alone, wk / wv
// (1024 rows) are 32 tiles on 256 CUs. The ROPEs run after the launch, i.e. later than in graph order relative to the following
// mat-muls: allowed only if no output of one chain overlaps an output of another (the allocator may have recycled a dead buffer).
static int try_fused_prefill_qkv(mi_backend_ctx * c, struct ggml_cgraph * g, int i) {
    static const bool on = !getenv("GGML_MI355X_PREFILL_QKV") || atoi(getenv("GGML_MI355X_PREFILL_QKV")) != 0;
    if (!on) return 0;
    const struct ggml_tensor * b = g->nodes[i]->src[1];
    // a chain: MUL_MAT [-> ADD bias (gpt-oss's bq / bk / bv, src/llama-model.cpp:17636-17645)] [-> (RESHAPE ->) ROPE]; the ADDs run as their own launches behind the
    // grouped mat-mul, like ROPEs that cannot ride on it
    struct chain { int mm, add, rope, end; } ch[3];
    int nc = 0, at = i; bool any_add = false;
    while (nc < 3 && at >= 0) {
        struct ggml_tensor * n = g->nodes[at];
        if (n->op != GGML_OP_MUL_MAT) break;
        const struct ggml_tensor * a = n->src[0];
        if (n->src[1] != b || !ggml_is_quantized(a->type) || tensor_is_split(a) || b->type != GGML_TYPE_F32 || b->ne[1] <= MMVQ_MAX_N || b->ne[2] != 1 || b->ne[3] != 1 ||
            a->ne[2] != 1 || a->ne[3] != 1 || b->nb[0] != 4 || a->ne[0] != g->nodes[i]->src[0]->ne[0] || n->nb[0] != 4) break;
        ch[nc] = { at, -1, -1, at };
        const struct ggml_tensor * last = n;
        int j = next_real(g, at);
        if (j > 0 && g->nodes[j]->op == GGML_OP_ADD && (g->nodes[j]->src[0] == n || g->nodes[j]->src[1] == n) && g->nodes[j]->src[0] != b && g->nodes[j]->src[1] != b) {
            ch[nc].add = j; ch[nc].end = j; last = g->nodes[j]; any_add = true;
            j = next_real(g, j);
        }
        if (j > 0 && g->nodes[j]->op == GGML_OP_ROPE && base_of(g->nodes[j]->src[0]) == last) { ch[nc].rope = j; ch[nc].end = j; }
        nc++;
        at = next_real(g, ch[nc - 1].end);
    }
    if (nc < 2) return 0;
    for (int q = 0; q < nc; q++) {
        const struct ggml_tensor * oq[3] = { g->nodes[ch[q].mm], ch[q].rope >= 0 ? g->nodes[ch[q].rope] : nullptr, ch[q].add >= 0 ? g->nodes[ch[q].add] : nullptr };
        for (int u = 0; u < 3; u++) {
            if (!oq[u]) continue;
            if (ranges_overlap(oq[u]->data, ggml_nbytes(oq[u]), b->data, ggml_nbytes(b))) return 0;
            for (int r = q + 1; r < nc; r++) {
                const struct ggml_tensor * orr[3] = { g->nodes[ch[r].mm], ch[r].rope >= 0 ? g->nodes[ch[r].rope] : nullptr, ch[r].add >= 0 ? g->nodes[ch[r].add] : nullptr };
                for (int v = 0; v < 3; v++) if (orr[v] && ranges_overlap(oq[u]->data, ggml_nbytes(oq[u]), orr[v]->data, ggml_nbytes(orr[v]))) return 0;
            }
        }
    }
    int types[3]; const void * W[3]; size_t wrs[3]; int64_t m[3]; float * dst[3]; size_t dstride[3];
    uint64_t wbytes = 0;
    // the ROPEs go into the launch (epilogue, or the pass that combines split-k planes) when they are NORM-mode, share one descriptor and
    // are the only readers of their mat-muls: the mat-mul then writes the ROPE node's tensor directly
    mmvq_rope rd = {}; bool have_rd = false, rope_ok = !any_add; int seg_rope[3] = { 0, 0, 0 };
    for (int q = 0; q < nc && rope_ok; q++) {
        if (ch[q].rope < 0) continue;
        const struct ggml_tensor * mm = g->nodes[ch[q].mm]; const struct ggml_tensor * rp = g->nodes[ch[q].rope];
        mmvq_rope r1 = {};
        r1.pos = (const int32_t *) rp->src[1]->data; r1.freq_factors = rp->src[2] ? (const float *) rp->src[2]->data : nullptr; r1.head_dim = (int) rp->ne[0];
        r1.p.n_dims = rp->op_params[1]; r1.p.mode = rp->op_params[2]; r1.p.n_ctx_orig = rp->op_params[4];
        r1.p.freq_base = op_f32(rp, 5); r1.p.freq_scale = op_f32(rp, 6); r1.p.ext_factor = op_f32(rp, 7);
        r1.p.attn_factor = op_f32(rp, 8); r1.p.beta_fast = op_f32(rp, 9); r1.p.beta_slow = op_f32(rp, 10);
        rope_ok = rp->op_params[2] == 0 && rp->type == GGML_TYPE_F32 && ggml_is_contiguous(rp) && rp->src[1]->type == GGML_TYPE_I32 && rp->ne[3] == 1 &&
                  rp->ne[2] == b->ne[1] && rp->ne[0]*rp->ne[1] == mm->ne[0] && rp->op_params[1] % 2 == 0 && rp->op_params[1] <= rp->ne[0] && rp->ne[0] % 4 == 0 &&
                  ggml_nelements(rp->src[1]) == b->ne[1] && is_internal(c, mm) && (rp->src[0] == mm || is_internal(c, rp->src[0])) &&
                  (!have_rd || memcmp(&rd, &r1, sizeof(rd)) == 0);
        rd = r1; have_rd = true; seg_rope[q] = 1;
    }
    const bool fuse_rope = have_rd && rope_ok;
    for (int q = 0; q < nc; q++) {
        const struct ggml_tensor * n = g->nodes[ch[q].mm]; const struct ggml_tensor * a = n->src[0];
        types[q] = (int) a->type; W[q] = a->data; wrs[q] = a->nb[1]; m[q] = a->ne[1]; dst[q] = (float *) n->data; dstride[q] = n->nb[1];
        if (fuse_rope && seg_rope[q]) { dst[q] = (float *) g->nodes[ch[q].rope]->data; dstride[q] = g->nodes[ch[q].rope]->nb[2]; }
        wbytes += (uint64_t) a->ne[1]*ggml_row_size(a->type, a->ne[0]);
    }
    const int64_t K = b->ne[0], N = b->ne[1];
    const bool ready = c->aq.valid && c->aq.kind == ACT_KIND_BF16 && c->aq.data == b->data && c->aq.k == K && c->aq.n_inner == N && c->aq.s_inner == b->nb[1] && c->aq.off == 0;
    prof_begin(c, types[0], m[0] + m[1] + (nc > 2 ? m[2] : 0), K, N, wbytes);
    // the KV-cache writes that follow (SET_ROWS of rope(k) as f16 rows, SET_ROWS of v as an element scatter into the transposed cache — the pair
    // try_fused_kv_store takes) ride on the same combine pass
    mmq_kv_store kvs = {}; bool have_kvs = false; int kv_last = -1;
    if (fuse_rope) {
        const int j1 = next_real(g, ch[nc - 1].end), j2 = j1 > 0 ? next_real(g, j1) : -1;
        if (j2 > 0 && g->nodes[j1]->op == GGML_OP_SET_ROWS && g->nodes[j2]->op == GGML_OP_SET_ROWS) {
            struct ggml_tensor * sk = g->nodes[j1]; struct ggml_tensor * sv = g->nodes[j2];
            const struct ggml_tensor * ks = sk->src[0]; const struct ggml_tensor * ki = sk->src[1];
            const struct ggml_tensor * vs = sv->src[0]; const struct ggml_tensor * vi = sv->src[1];
            int qk = -1, qv = -1;
            for (int q = 0; q < nc; q++) { if (ks->data == (void *) dst[q]) qk = q; if (vs->data == (void *) dst[q]) qv = q; }
            const bool ok = qk >= 0 && qv >= 0 && qk != qv && sk->type == GGML_TYPE_F16 && sv->type == GGML_TYPE_F16 && ks->type == GGML_TYPE_F32 && vs->type == GGML_TYPE_F32 &&
                ki->type == GGML_TYPE_I64 && vi->type == GGML_TYPE_I64 && ggml_is_contiguous(ki) && ggml_is_contiguous(vi) &&
                ks->ne[0] == m[qk] && ks->ne[1] == N && ks->ne[2] == 1 && ks->ne[3] == 1 && ks->nb[0] == 4 && ks->nb[1] == (size_t) m[qk]*4 && dstride[qk] == (size_t) m[qk]*4 &&
                sk->nb[0] == 2 && sk->nb[1] % 8 == 0 && ((uintptr_t) sk->data % 8) == 0 && ggml_nelements(ki) == N && m[qk] % 4 == 0 &&
                vs->ne[0] == 1 && vs->ne[1] == m[qv]*N && vs->ne[2] == 1 && vs->ne[3] == 1 && vs->nb[1] == 4 && dstride[qv] == (size_t) m[qv]*4 &&
                sv->ne[0] == 1 && sv->nb[1] == 2 && ggml_nelements(vi) == m[qv]*N &&
                !ranges_overlap(sk->data, ggml_nbytes(sk), b->data, ggml_nbytes(b)) && !ranges_overlap(sv->data, ggml_nbytes(sv), b->data, ggml_nbytes(b));
            if (ok) {
                kvs.st16[qk] = (uint16_t *) sk->data; kvs.st_idx[qk] = (const int64_t *) ki->data; kvs.st_row_elems[qk] = (int64_t)(sk->nb[1]/2); kvs.st_mode[qk] = 1;
                kvs.st16[qv] = (uint16_t *) sv->data; kvs.st_idx[qv] = (const int64_t *) vi->data; kvs.st_row_elems[qv] = 0; kvs.st_mode[qv] = 2;
                have_kvs = true; kv_last = j2;
            }
        }
    }
    // bias rows (ADD behind a mat-mul, the mat-mul read by nobody else): the launch's epilogue adds them and writes the ADD's tensor
    const float * seg_bias[3] = { nullptr, nullptr, nullptr }; float * bdst[3]; size_t bdstride[3]; bool biased = any_add;
    for (int q = 0; q < nc && biased; q++) {
        bdst[q] = dst[q]; bdstride[q] = dstride[q];
        if (ch[q].add < 0) continue;
        const struct ggml_tensor * mm = g->nodes[ch[q].mm]; const struct ggml_tensor * ad = g->nodes[ch[q].add];
        const struct ggml_tensor * bt = ad->src[0] == mm ? ad->src[1] : ad->src[0];
        biased = is_internal(c, mm) && ad->type == GGML_TYPE_F32 && ggml_are_same_shape(ad, mm) && ad->nb[0] == 4 && bt->type == GGML_TYPE_F32 && bt->ne[0] == m[q] &&
                 ggml_nelements(bt) == m[q] && bt->nb[0] == 4;
        seg_bias[q] = (const float *) bt->data; bdst[q] = (float *) ad->data; bdstride[q] = ad->nb[1];
    }
    bool roped = fuse_rope, stored = have_kvs;
    if (biased && mul_mat_q_multi(nc, types, W, wrs, m, bdst, bdstride, K, (const float *) b->data, b->nb[1], N, c->scratch, c->scratch_size, ready, nullptr, nullptr, nullptr, c->stream, seg_bias)) {
        prof_end(c);
        if (ready) c->cnt.act_quant_reused++;
        else c->aq = { b->data, K, N, 1, b->nb[1], 0, ACT_KIND_BF16, act_q8{}, true, (size_t)(N - 1)*b->nb[1] + (size_t) K*4 };
        c->cnt.mmq_launches++; c->cnt.kernels_launched += ready ? 1 : 2; c->cnt.weight_bytes += wbytes;
        for (int q = 0; q < nc; q++) if (ch[q].rope >= 0) compute_node(c, g, ch[q].rope);
        return ch[nc - 1].end - i + 1;
    }
    bool done = fuse_rope && mul_mat_q_multi(nc, types, W, wrs, m, dst, dstride, K, (const float *) b->data, b->nb[1], N, c->scratch, c->scratch_size, ready, &rd, seg_rope,
                                             have_kvs ? &kvs : nullptr, c->stream);
    if (!done) {
        roped = false; stored = false;
        for (int q = 0; q < nc; q++) { dst[q] = (float *) g->nodes[ch[q].mm]->data; dstride[q] = g->nodes[ch[q].mm]->nb[1]; }
        done = mul_mat_q_multi(nc, types, W, wrs, m, dst, dstride, K, (const float *) b->data, b->nb[1], N, c->scratch, c->scratch_size, ready, nullptr, nullptr, nullptr, c->stream);
    }
    prof_end(c);
    if (!done) { if (c->profiling && !c->prof_suspend) c->prof.pop_back(); return 0; }
    if (ready) c->cnt.act_quant_reused++;
    else c->aq = { b->data, K, N, 1, b->nb[1], 0, ACT_KIND_BF16, act_q8{}, true, (size_t)(N - 1)*b->nb[1] + (size_t) K*4 };
    c->cnt.mmq_launches++; c->cnt.kernels_launched += ready ? 1 : 2; c->cnt.weight_bytes += wbytes;
    for (int q = 0; q < nc; q++) {
        if (ch[q].add >= 0) compute_node(c, g, ch[q].add);
        if (!roped && ch[q].rope >= 0) compute_node(c, g, ch[q].rope);
    }
    return (stored ? kv_last : ch[nc - 1].end) - i + 1;
}

// Prefill: MUL_MAT -> ADD(residual) (build_attn's wo, build_ffn's down: src/llama-model.cpp:6057,6096) — the residual is added in the
// mat-mul's epilogue, or by the pass that combines its split-k planes
static int try_fused_prefill_add(mi_backend_ctx * c, struct ggml_cgraph * g, int i) {
    struct ggml_tensor * n = g->nodes[i];
    const struct ggml_tensor * a = n->src[0]; const struct ggml_tensor * b = n->src[1];
    if (!ggml_is_quantized(a->type) || b->type != GGML_TYPE_F32 || b->ne[1] <= MMVQ_MAX_N || b->ne[2] != 1 || b->ne[3] != 1 || a->ne[2] != 1 || a->ne[3] != 1 ||
        b->nb[0] != 4 || !is_internal(c, n)) return 0;
    const int j = next_real(g, i); if (j < 0) return 0;
    struct ggml_tensor * nx = g->nodes[j];
    if (nx->op != GGML_OP_ADD || (nx->src[0] != n && nx->src[1] != n) || nx->type != GGML_TYPE_F32 || !ggml_is_contiguous(nx) || !ggml_are_same_shape(nx, n)) return 0;
    const struct ggml_tensor * other = nx->src[0] == n ? nx->src[1] : nx->src[0];
    // the addend: the residual stream (same shape), or a bias row (gpt-oss's bo, src/llama-graph.cpp:1479-1481) — then a row stride of 0
    const bool bias_row = other->ne[0] == n->ne[0] && ggml_nelements(other) == n->ne[0] && n->ne[1] > 1;
    if (other == n || other->type != GGML_TYPE_F32 || (!ggml_are_same_shape(other, n) && !bias_row) || other->nb[0] != 4) return 0;
    // the sum is the residual stream; if RMS_NORM -> MUL(w) reads it next (ffn_norm, the next layer's attn_norm: build_norm, src/llama-graph.cpp:597-630)
    // and the mat-mul splits k, ONE pass adds the planes and the residual, writes the sum and normalises it
    static const bool norm_on = !getenv("GGML_MI355X_PREFILL_COMBINE_NORM") || atoi(getenv("GGML_MI355X_PREFILL_COMBINE_NORM")) != 0;
    struct ggml_tensor * nrm = nullptr; struct ggml_tensor * mul = nullptr; const struct ggml_tensor * w = nullptr;
    const int jn = next_real(g, j);
    if (norm_on && !bias_row && jn > 0 && jn + 1 < g->n_nodes && g->nodes[jn]->op == GGML_OP_RMS_NORM && g->nodes[jn]->src[0] == nx && g->nodes[jn + 1]->op == GGML_OP_MUL) {
        nrm = g->nodes[jn]; mul = g->nodes[jn + 1];
        w = mul->src[0] == nrm ? mul->src[1] : (mul->src[1] == nrm ? mul->src[0] : nullptr);
        const int64_t M = n->ne[0];
        const bool ok = w && is_internal(c, nrm) && w->type == GGML_TYPE_F32 && ggml_nelements(w) == M && w->ne[0] == M && w->nb[0] == 4 && ((uintptr_t) w->data % 16) == 0 &&
                        mul->type == GGML_TYPE_F32 && ggml_are_same_shape(mul, nx) && mul->nb[0] == 4 && mul->nb[1] % 16 == 0 && ((uintptr_t) mul->data % 16) == 0 &&
                        M % 4 == 0 && other->nb[1] % 16 == 0 && ((uintptr_t) other->data % 16) == 0 && nx->nb[1] % 16 == 0 && ((uintptr_t) nx->data % 16) == 0 &&
                        nrm->ne[2] == 1 && nrm->ne[3] == 1;
        if (!ok) { nrm = nullptr; mul = nullptr; }
    }
    mmq_deferred df = { 0, nullptr };
    op_mul_mat(c, n, nx, other, nrm ? &df : nullptr);
    if (!nrm || df.np == 0) return j - i + 1;          // no k split: the epilogue has added the residual; the norm runs as usual
    const int64_t M = n->ne[0], N = n->ne[1];
    // the mat-mul after the norm reads bf16: written here too (the planes sit above the activation copy's slot only if that slot is this launch's own x:
    // the copy goes to offset 0, which the finished mat-mul no longer needs — but the PLANES live right after its x region, so the new copy must not reach them)
    const int jm = next_real(g, jn + 1);
    const struct ggml_tensor * mm = jm > 0 ? g->nodes[jm] : nullptr;
    uint16_t * y16 = nullptr;
    if (mm && mm->op == GGML_OP_MUL_MAT && mm->src[1] == mul && ggml_is_quantized(mm->src[0]->type) && N > MMVQ_MAX_N && mul->ne[2] == 1 && mul->ne[3] == 1 &&
        (const char *) c->scratch + (c->aq.valid ? c->aq.off : 0) + mul_mat_q_x_bytes(M, N) <= (const char *) df.planes) {
        y16 = (uint16_t *) c->scratch;
    }
    combine_rms_norm(df.planes, df.np, M, N, (const float *) other->data, other->nb[1], (float *) nx->data, nx->nb[1], (const float *) w->data,
                     (float *) mul->data, mul->nb[1], y16, op_f32(nrm, 0), c->stream);
    c->cnt.kernels_launched++;
    if (y16) {
        c->aq = { mul->data, M, N, 1, mul->nb[1], 0, ACT_KIND_BF16, act_q8{}, true, (size_t)(N - 1)*mul->nb[1] + (size_t) M*4, 0 };
        c->aq_fresh = true;
    } else c->aq.valid = false;
    return jn + 1 - i + 1;
}

// returns the number of graph nodes consumed (>= 1)
static int compute_node(mi_backend_ctx * c, struct ggml_cgraph * g, int i) {
    struct ggml_tensor * node = g->nodes[i];
    if (ggml_is_empty(node) || is_view_op(node->op)) return 1;
    const struct ggml_tensor * s0 = node->src[0];
    const struct ggml_tensor * s1 = node->src[1];

    int consumed = 1;
    bool fresh_aq = false;   // this step produced the cached quantized activations itself
    // a MoE combine is pending: only the norm that reads its sum may go on without it (its grouped launch evaluates it)
    if (c->pp.active && !(node->op == GGML_OP_RMS_NORM && s0 && s0->data == (void *) c->pp.x_out && c->use_fusion)) pp_flush(c);
    if (c->use_fusion) {
        int f = 0;
        if (node->op == GGML_OP_MUL_MAT && !tensor_is_split(s0)) {
            const int l = try_fused_mmv(c, g, i, nullptr, nullptr); f = l >= 0 ? l - i + 1 : 0;
            if (!f) f = try_fused_attn(c, g, i);       // (one token: recorded; many tokens: flushes, then launches)
            if (!f) f = try_fused_moe_route(c, g, i);
            if (!f) f = try_fused_prefill_glu(c, g, i);
            if (!f) f = try_fused_prefill_qkv(c, g, i);
            if (!f) f = try_fused_prefill_add(c, g, i);
        } else if (node->op == GGML_OP_SET_ROWS) f = try_fused_kv_store(c, g, i);
        else if (node->op == GGML_OP_GET_ROWS) f = try_fused_moe_combine(c, g, i);
        else if (node->op == GGML_OP_MUL_MAT_ID) { f = try_fused_moe_experts(c, g, i); if (!f) f = try_fused_prefill_moe(c, g, i); }
        else if (node->op == GGML_OP_ADD && s0 && s0->view_src) f = try_fused_slot_sum(c, g, i);
        if (f) {
            consumed = f;
            fresh_aq = c->aq_fresh; c->aq_fresh = false;
            goto done;
        }
    }
    switch (node->op) {
        case GGML_OP_MUL_MAT:    if (tensor_is_split(s0)) op_mul_mat_split(c, node); else op_mul_mat(c, node); break;
        case GGML_OP_MUL_MAT_ID: op_mul_mat_id(c, node); break;
        case GGML_OP_FLASH_ATTN_EXT: {
            const struct ggml_tensor * q = s0; const struct ggml_tensor * k = node->src[1]; const struct ggml_tensor * v = node->src[2];
            const struct ggml_tensor * mask = node->src[3]; const struct ggml_tensor * sinks = node->src[4];
            const int64_t hd = k->ne[0], n_kv = k->ne[1], T = q->ne[1];
            // what supports_op cannot see (data alignment of the views handed in) fails the graph instead of the process
            MI_REQUIRE_G(node->nb[1] == (size_t) hd*4 && ((uintptr_t) q->data % 16) == 0);
            MI_REQUIRE_G(((uintptr_t) k->data % (k->type == GGML_TYPE_F16 || k->type == GGML_TYPE_BF16 ? 16 : 2)) == 0 && ((uintptr_t) v->data % (v->type == GGML_TYPE_F16 || v->type == GGML_TYPE_BF16 ? 16 : 2)) == 0);
            MI_REQUIRE_G(!mask || (mask->ne[0] == n_kv && mask->ne[1] >= T && ((uintptr_t) mask->data % 16) == 0 && mask->nb[1] % 16 == 0));
            MI_REQUIRE_G(T > 8 || attn_decode_supported(hd, n_kv) || (c->attn_part && attn_decode_supported_split(hd, n_kv) && attn_decode_part_bytes(hd, n_kv, q->ne[2], T) <= c->attn_part_bytes));
            attn_extra ex = { op_f32(node, 2), op_f32(node, 1), (int) k->type, (int) v->type };
            const void * kd = k->data; const void * vd = v->data;
            size_t k_nb1 = k->nb[1], k_nb2 = k->nb[2], v_nb1 = v->nb[1], v_nb2 = v->nb[2];
            size_t kb, vb; fa_kv16_plan(node, kb, vb);
            MI_REQUIRE_G(kb + vb <= c->kv16_size);
            bool v_trans = false;
            if (kb) {
                kv_to_f16((int) k->type, k->data, k->nb[1], k->nb[2], hd, n_kv, k->ne[2], (uint16_t *) c->kv16, false, c->stream);
                kd = c->kv16; k_nb1 = (size_t) hd*2; k_nb2 = (size_t) n_kv*hd*2; ex.k_type = T_F16; c->cnt.kernels_launched++;
            }
            if (vb) {
                v_trans = T > 8;
                kv_to_f16((int) v->type, v->data, v->nb[1], v->nb[2], hd, n_kv, v->ne[2], (uint16_t *) ((char *) c->kv16 + kb), v_trans, c->stream);
                vd = (char *) c->kv16 + kb; ex.v_type = T_F16; c->cnt.kernels_launched++;
                if (v_trans) { v_nb1 = (size_t) n_kv*2; v_nb2 = (size_t) hd*n_kv*2; } else { v_nb1 = (size_t) hd*2; v_nb2 = (size_t) n_kv*hd*2; }
            }
            if (T <= 8) attn_decode(q->data, q->nb[1], q->nb[2], kd, k_nb1, k_nb2, vd, v_nb1, v_nb2, mask ? mask->data : nullptr, mask ? mask->nb[1] : 0, true,
                                    sinks ? (const float *) sinks->data : nullptr, (float *) node->data, node->nb[2], hd, n_kv, q->ne[2], k->ne[2], T, op_f32(node, 0), c->stream, false,
                                    c->attn_part, c->attn_part_bytes, false, &ex);
            else        attn_prefill(q->data, q->nb[1], q->nb[2], kd, k_nb1, k_nb2, vd, v_nb1, v_nb2, mask ? mask->data : nullptr, mask ? mask->nb[1] : 0, true,
                                     sinks ? (const float *) sinks->data : nullptr, (float *) node->data, node->nb[2], hd, n_kv, q->ne[2], k->ne[2], T, op_f32(node, 0), c->stream, v_trans, nullptr, &ex);
            c->cnt.kernels_launched++;
        } break;
        case GGML_OP_RMS_NORM: {
            // fusion: RMS_NORM -> MUL(by weight) [-> ADD], when the intermediate has no other reader
            // (the pattern build_norm emits, src/llama-graph.cpp:597-630; pinned by tests/test-backend-ops.cpp:2856)
            if (c->use_fusion && i + 1 < g->n_nodes) {
                struct ggml_tensor * mul = g->nodes[i + 1];
                if (mul->op == GGML_OP_MUL && (mul->src[0] == node || mul->src[1] == node) && mul->type == GGML_TYPE_F32 &&
                    !(node->flags & GGML_TENSOR_FLAG_OUTPUT)) {
                    const struct ggml_tensor * w = mul->src[0] == node ? mul->src[1] : mul->src[0];
                    bool only_reader = true;
                    for (int j = i + 2; j < g->n_nodes && only_reader; j++) {
                        for (int s = 0; s < GGML_MAX_SRC; s++) if (g->nodes[j]->src[s] == node) { only_reader = false; break; }
                    }
                    if (only_reader && w->type == GGML_TYPE_F32 && w->nb[0] == sizeof(float) && ggml_are_same_shape(node, mul) &&
                        ggml_can_repeat(w, node) && mul->nb[0] == sizeof(float)) {
                        // ... and when a quantized mat-mul reads the result next, quantize it in the same kernel
                        const int jn = next_real(g, i + 1);
                        const struct ggml_tensor * mm = jn > 0 ? g->nodes[jn] : nullptr;
                        // best case: every reader of the product is a mat-vec of ONE grouped launch -> the norm runs in its prologue
                        if (mm && fusable_mmv(mm) && mm->src[1] == mul && node->ne[1] == 1 && w->ne[0] == node->ne[0] && ggml_nelements(w) == w->ne[0] &&
                            is_row_vec_f32(s0) && !(mul->flags & GGML_TENSOR_FLAG_OUTPUT)) {
                            const int l = try_fused_mmv(c, g, jn, node, w);
                            if (l >= 0) { consumed = l - i + 1; fresh_aq = c->aq_fresh; c->aq_fresh = false; break; }
                        }
                        pp_flush(c);       // (no grouped launch took the pending planes)
                        // MoE: the router's F32 mat-mul reads the product first (build_moe_ffn, src/llama-graph.cpp:838): the norm runs inside the
                        // router kernel, which also writes the product for the expert mat-vecs that follow
                        if (mm && mm->op == GGML_OP_MUL_MAT && mm->src[1] == mul && mm->src[0]->type == GGML_TYPE_F32 && node->ne[1] == 1 && w->ne[0] == node->ne[0] &&
                            ggml_nelements(w) == w->ne[0] && is_row_vec_f32(s0) && is_row_vec_f32(mul) && s0->data != mul->data) {
                            const int f = try_fused_moe_route(c, g, jn, node, w);
                            if (f) { consumed = jn + f - i; break; }
                        }
                        if (mm && (mm->op == GGML_OP_MUL_MAT) && mm->src[1] == mul && ggml_is_quantized(mm->src[0]->type) &&
                            act_kind_for((int) mm->src[0]->type) > 0 && rms_norm_mul_quant_supported(node->ne[0]) && node->ne[1] <= 8 && node->ne[2] == 1 && node->ne[3] == 1 &&
                            w->ne[0] == node->ne[0] && ggml_nelements(w) == w->ne[0] && s0->nb[0] == 4 &&
                            ((uintptr_t) s0->data % 16) == 0 && (s0->nb[1] % 16) == 0 && ((uintptr_t) mul->data % 16) == 0 && (mul->nb[1] % 16) == 0 &&
                            ((uintptr_t) w->data % 16) == 0) {
                            const int kind = act_kind_for((int) mm->src[0]->type);
                            const act_q8 q = act_q8_carve(c->scratch, kind, node->ne[0], node->ne[1]);
                            rms_norm_mul_quant((const float *) s0->data, s0->nb[1], (const float *) w->data, (float *) mul->data, mul->nb[1], q,
                                               node->ne[0], node->ne[1], op_f32(node, 0), c->stream);
                            c->aq = { mul->data, node->ne[0], node->ne[1], 1, mul->nb[1], 0, kind, q, true,
                                      (size_t)(node->ne[1] - 1)*mul->nb[1] + (size_t) node->ne[0]*4 };
                            c->cnt.kernels_launched++; c->cnt.act_quant_launches++;
                            consumed = 2;
                            fresh_aq = true;
                            break;
                        }
                        // prefill: a quantized mat-mul on many tokens reads the product next — its bf16 copy comes out of this kernel too
                        uint16_t * y16 = nullptr;
                        if (mm && mm->op == GGML_OP_MUL_MAT && mm->src[1] == mul && ggml_is_quantized(mm->src[0]->type) && mul->ne[1] > MMVQ_MAX_N &&
                            mul->ne[2] == 1 && mul->ne[3] == 1 && rms_norm_mul_bf16_supported(desc(s0), desc(w), desc(mul)) &&
                            mul_mat_q_scratch_bytes(mul->ne[0], mul->ne[1], 0) <= c->scratch_size) {
                            y16 = (uint16_t *) c->scratch;
                        }
                        rms_norm_mul(desc(s0), desc(w), nullptr, desc(mul), op_f32(node, 0), c->stream, y16);
                        if (y16) {
                            c->aq = { mul->data, mul->ne[0], mul->ne[1], 1, mul->nb[1], 0, ACT_KIND_BF16, act_q8{}, true, (size_t)(mul->ne[1] - 1)*mul->nb[1] + (size_t) mul->ne[0]*4 };
                            fresh_aq = true;
                        }
                        c->cnt.kernels_launched++;
                        consumed = 2;
                        break;
                    }
                }
            }
            pp_flush(c);
            rms_norm(desc(s0), desc(node), op_f32(node, 0), c->stream);
            c->cnt.kernels_launched++;
        } break;
        case GGML_OP_ADD: bin_bcast(BIN_ADD, desc(s0), desc(s1), desc(node), c->stream); c->cnt.kernels_launched++; break;
        case GGML_OP_MUL: bin_bcast(BIN_MUL, desc(s0), desc(s1), desc(node), c->stream); c->cnt.kernels_launched++; break;
        case GGML_OP_DIV: bin_bcast(BIN_DIV, desc(s0), desc(s1), desc(node), c->stream); c->cnt.kernels_launched++; break;
        case GGML_OP_SUB: bin_bcast(BIN_SUB, desc(s0), desc(s1), desc(node), c->stream); c->cnt.kernels_launched++; break;
        case GGML_OP_ADD_ID: add_id(desc(s0), desc(s1), desc(node->src[2]), desc(node), c->stream); c->cnt.kernels_launched++; break;
        case GGML_OP_SCALE: scale(desc(s0), desc(node), op_f32(node, 0), op_f32(node, 1), c->stream); c->cnt.kernels_launched++; break;
        case GGML_OP_CPY:  cpy(desc(s0), desc(s1), c->stream); c->cnt.kernels_launched++; break;
        case GGML_OP_CONT: case GGML_OP_DUP: cpy(desc(s0), desc(node), c->stream); c->cnt.kernels_launched++; break;
        case GGML_OP_SET_ROWS: set_rows(desc(s0), desc(s1), desc(node), c->stream); c->cnt.kernels_launched++; break;
        case GGML_OP_GET_ROWS: get_rows(desc(s0), desc(s1), desc(node), c->stream); c->cnt.kernels_launched++; break;
        case GGML_OP_SUM_ROWS: sum_rows(desc(s0), desc(node), c->stream); c->cnt.kernels_launched++; break;
        case GGML_OP_ARGSORT: argsort(desc(s0), desc(node), node->op_params[0], c->stream); c->cnt.kernels_launched++; break;
        case GGML_OP_UNARY: unary((int) ggml_get_unary_op(node), desc(s0), desc(node), c->stream); c->cnt.kernels_launched++; break;
        case GGML_OP_GLU: {
            tensor_desc b;
            if (s1) b = desc(s1);
            glu((int) ggml_get_glu_op(node), node->op_params[1] != 0, desc(s0), s1 ? &b : nullptr, desc(node), op_f32(node, 2), op_f32(node, 3), c->stream);
            c->cnt.kernels_launched++;
        } break;
        case GGML_OP_ROPE: {
            rope_params p;
            p.n_dims = node->op_params[1]; p.mode = node->op_params[2]; p.n_ctx_orig = node->op_params[4];
            p.freq_base = op_f32(node, 5); p.freq_scale = op_f32(node, 6); p.ext_factor = op_f32(node, 7);
            p.attn_factor = op_f32(node, 8); p.beta_fast = op_f32(node, 9); p.beta_slow = op_f32(node, 10);
            rope(desc(s0), (const int32_t *) s1->data, node->src[2] ? (const float *) node->src[2]->data : nullptr, desc(node), p, c->stream);
            c->cnt.kernels_launched++;
        } break;
        case GGML_OP_SOFT_MAX: {
            tensor_desc m;
            if (s1) m = desc(s1);
            soft_max(desc(s0), s1 ? &m : nullptr, node->src[2] ? (const float *) node->src[2]->data : nullptr, desc(node),
                     op_f32(node, 0), op_f32(node, 1), c->stream);
            c->cnt.kernels_launched++;
        } break;
        default:
            GGML_ABORT("MI355X backend: graph_compute met unsupported op %s (supports_op should have refused it)", ggml_op_name(node->op));
    }

done:
    // any write into the memory the cached quantized activations were made from invalidates them
    for (int j = i; j < i + consumed; j++) {
        const struct ggml_tensor * w = g->nodes[j];
        if (c->aq.valid && !fresh_aq && w->data && ranges_overlap(w->data, ggml_nbytes(w), c->aq.data, c->aq.span)) {
            c->aq.valid = false;
        }
    }
    c->cnt.nodes_computed += consumed;
    return consumed;
}

// ---- the hipGraph cache ----------------------------------------------------------------------------------------
// A decode graph is captured the second time its signature is seen and replayed afterwards. The capture is cut into SEGMENTS (a few launches, then a few
// layers, then the rest), each its own executable graph: a replay checks and launches segment 0 first, so the GPU starts on the token while the host is
// still checking and launching the rest — one 166-node hipGraphLaunch took 62 us of host time during which the GPU sat idle, on top of 21 us for the
// 1125-node signature (llama-bench synchronizes after every token, so that gap is paid per token).
// What makes launching a prefix safe: (1) a TOPOLOGY pass over the whole graph first (every node and its source tensors are the cached graph's: who reads
// what decides which intermediates a fused group may leave unwritten); (2) a segment is launched only once its nodes' full signatures (op, type, shape,
// strides, placement, op_params, the sources' placement and layout) match and every segment before it has been launched. If a later segment does not
// match (same topology, another parameter), the remaining nodes run eagerly, behind the prefix already launched.
static void end_capture_segment(mi_backend_ctx * c, graph_entry & e, int begin, int end) {
    hipGraph_t graph = nullptr;
    MI_CHECK_G(hipStreamEndCapture(c->stream, &graph));
    c->capturing = false;
    graph_seg sg = { begin, end, nullptr };
    size_t n_nodes = 0;
    if (graph && hipGraphGetNodes(graph, nullptr, &n_nodes) == hipSuccess && n_nodes > 0) {
        const hipError_t err = hipGraphInstantiate(&sg.exec, graph, nullptr, nullptr, 0);
        if (err != hipSuccess) { (void) hipGraphDestroy(graph); throw mi_graph_error{ err, __FILE__, __LINE__ }; }
    } else (void) hipGetLastError();
    if (graph) MI_CHECK_G(hipGraphDestroy(graph));
    e.segs.push_back(sg);
}

// runs nodes [start, n_nodes). cap != NULL: under stream capture into cap's segments (the caller has begun the first capture; the last one is ended here)
static void run_nodes(mi_backend_ctx * c, struct ggml_cgraph * g, int start = 0, graph_entry * cap = nullptr) {
    c->rope_tab_valid = false;      // every pass over a graph (eager or under capture) fills the token's rotation table itself
    // the grouped mat-vec module keeps its launch hooks per host thread: (re)install this backend's for the thread that computes
    mul_mat_vec_q_fused_set_hooks(c->profiling ? prof_hook_pre : nullptr, c->profiling ? prof_hook_post : nullptr, c);
    c->aq.valid = false;
    c->uses.clear();
    if (c->use_fusion) {
        for (int i = 0; i < g->n_nodes; i++) {
            for (int s = 0; s < GGML_MAX_SRC; s++) if (g->nodes[i]->src[s]) c->uses[g->nodes[i]->src[s]]++;
        }
    }
    c->pp.active = false; c->rt.valid = false;
    // segment sizes in kernel launches: the first is small (the GPU starts early), the second covers the time the host needs to launch the rest
    static const int seg_kernels[2] = { getenv("GGML_MI355X_GRAPH_SEG0") ? atoi(getenv("GGML_MI355X_GRAPH_SEG0")) : 6, getenv("GGML_MI355X_GRAPH_SEG1") ? atoi(getenv("GGML_MI355X_GRAPH_SEG1")) : 24 };
    int seg_begin = start; uint64_t k0 = c->cnt.kernels_launched;
    for (int i = start; i < g->n_nodes; ) {
        i += compute_node(c, g, i);
        if (cap && i < g->n_nodes && cap->segs.size() < 2 && seg_kernels[cap->segs.size()] > 0 && !c->pp.active &&
            c->cnt.kernels_launched - k0 >= (uint64_t) seg_kernels[cap->segs.size()]) {
            end_capture_segment(c, *cap, seg_begin, i);
            c->aq.valid = false;      // nothing cached is carried over a cut (the next segment may one day run behind an eager prefix)
            MI_CHECK_G(hipStreamBeginCapture(c->stream, hipStreamCaptureModeThreadLocal));
            c->capturing = true;
            seg_begin = i; k0 = c->cnt.kernels_launched;
        }
    }
    pp_flush(c);
    if (cap) end_capture_segment(c, *cap, seg_begin, g->n_nodes);
    c->aq.valid = false;
}

// (round 3: 64-bit lanes — one multiply per 8 bytes instead of one per 4, two independent chains for the op parameters — and only the sources that
// exist are walked)
static inline uint64_t sig_mix(uint64_t h, uint64_t v) { h = (h ^ v)*0x9E3779B97F4A7C15ull; return h ^ (h >> 32); }
static inline uint32_t hash_params(const int32_t * p) {
    uint64_t w[GGML_MAX_OP_PARAMS/8];
    memcpy(w, p, sizeof(w));
    uint64_t h0 = 0x243F6A8885A308D3ull, h1 = 0x13198A2E03707344ull;
    for (int i = 0; i < (int)(GGML_MAX_OP_PARAMS/8); i += 2) { h0 = sig_mix(h0, w[i]); h1 = sig_mix(h1, w[i + 1]); }
    const uint64_t h = sig_mix(h0, h1);
    return (uint32_t)(h ^ (h >> 32));
}

static inline void fill_sig(node_sig & s, const struct ggml_tensor * n) {
    s.node = n; s.data = n->data; s.op = (int32_t) n->op; s.type = (int32_t) n->type;
    for (int d = 0; d < 4; d++) s.ne[d] = n->ne[d];
    for (int d = 0; d < 3; d++) s.nb[d] = n->nb[d + 1];
    for (int j = 0; j < GGML_MAX_SRC; j++) {
        const struct ggml_tensor * t = n->src[j];
        s.src[j] = t;
        if (!t) { s.src_data[j] = NULL; s.src_hash[j] = 0u; continue; }
        s.src_data[j] = t->data;
        uint64_t h0 = sig_mix(0x452821E638D01377ull, (uint64_t) t->type), h1 = 0xBE5466CF34E90C6Cull;
        for (int d = 0; d < 4; d++) { h0 = sig_mix(h0, (uint64_t) t->ne[d]); h1 = sig_mix(h1, (uint64_t) t->nb[d]); }
        const uint64_t h = sig_mix(h0, h1);
        const uint32_t h32 = (uint32_t)(h ^ (h >> 32));
        s.src_hash[j] = h32 ? h32 : 1u;
    }
    s.params_hash = hash_params(n->op_params);
    s.flags = (uint32_t) n->flags;
}

static void free_entry(graph_entry & e) {
    for (auto & sg : e.segs) if (sg.exec) (void) hipGraphExecDestroy(sg.exec);
    e.segs.clear();
}
static void drop_graphs(mi_backend_ctx * c) {
    for (auto & e : c->graphs) free_entry(e);
    c->graphs.clear();
}

// ---- replay, fast path: the cached graph this submission most likely is (same first node, same length, most recently used) ----
static graph_entry * replay_candidate(mi_backend_ctx * c, const struct ggml_cgraph * g) {
    graph_entry * best = nullptr;
    for (auto & e : c->graphs) {
        if (e.segs.empty() || (int) e.sig.size() != g->n_nodes || e.sig[0].node != g->nodes[0]) continue;
        if (!best || e.last_use > best->last_use) best = &e;
    }
    return best;
}
static bool topology_matches(const graph_entry & e, const struct ggml_cgraph * g) {
    static_assert(sizeof(((struct ggml_tensor *) 0)->src) == sizeof(((node_sig *) 0)->src), "node_sig::src mirrors ggml_tensor::src");
    for (int i = 0; i < g->n_nodes; i++) {
        const struct ggml_tensor * n = g->nodes[i];
        if (n != e.sig[i].node || memcmp(n->src, e.sig[i].src, sizeof(n->src)) != 0) return false;
    }
    return true;
}
static bool signature_matches(const graph_entry & e, const struct ggml_cgraph * g, int begin, int end) {
    node_sig ns;
    for (int i = begin; i < end; i++) {
        fill_sig(ns, g->nodes[i]);
        if (memcmp(&ns, &e.sig[i], sizeof(ns)) != 0) return false;
    }
    return true;
}

// find (or create) the cache entry whose signature equals this graph's
static graph_entry & graph_lookup(mi_backend_ctx * c, const struct ggml_cgraph * g) {
    c->cur_sig.resize(g->n_nodes);
    uint64_t d0 = 0x9E3779B97F4A7C15ull, d1 = 0xC2B2AE3D27D4EB4Full;
    for (int i = 0; i < g->n_nodes; i++) {
        node_sig & ns = c->cur_sig[i];
        fill_sig(ns, g->nodes[i]);
        d0 = sig_mix(d0, (uint64_t)(uintptr_t) ns.data ^ ((uint64_t) ns.ne[1] << 20) ^ ((uint64_t) ns.ne[0] << 44) ^ ns.params_hash);
        d1 = sig_mix(d1, (uint64_t)(uintptr_t) ns.src_data[0] ^ ((uint64_t)(uintptr_t) ns.src_data[1] << 1) ^ ((uint64_t) ns.src_hash[0] << 32) ^ ns.src_hash[1] ^ ((uint64_t) ns.nb[0] << 13));
    }
    const uint64_t digest = sig_mix(d0, d1);
    c->cur_digest = digest;
    graph_entry * lru = nullptr;
    // most recently used first: consecutive decode steps re-submit the same graph (src/llama-context.cpp:728)
    graph_entry * best = nullptr;
    for (auto & e : c->graphs) {
        if (!lru || e.last_use < lru->last_use) lru = &e;
        if (e.digest == digest && (int) e.sig.size() == g->n_nodes && (!best || e.last_use > best->last_use) &&
            memcmp(e.sig.data(), c->cur_sig.data(), sizeof(node_sig)*g->n_nodes) == 0) best = &e;
    }
    if (best) { best->last_use = ++c->graph_tick; best->seen++; return *best; }
    if ((int) c->graphs.size() >= MI_MAX_GRAPHS) {
        if (!lru->segs.empty()) MI_CHECK_G(hipStreamSynchronize(c->stream));
        free_entry(*lru);
        *lru = graph_entry();
        lru->sig = c->cur_sig; lru->digest = c->cur_digest; lru->last_use = ++c->graph_tick; lru->seen = 1;
        return *lru;
    }
    c->graphs.emplace_back();
    graph_entry & e = c->graphs.back();
    e.sig = c->cur_sig; e.digest = c->cur_digest; e.last_use = ++c->graph_tick; e.seen = 1;
    return e;
}

static enum ggml_status be_graph_compute_impl(ggml_backend_t backend, struct ggml_cgraph * g);
static double g_host_ns = 0.0; static long g_host_calls = 0;     // GGML_MI355X_HOST_TIMING=1: host time spent inside graph_compute
static enum ggml_status be_graph_compute(ggml_backend_t backend, struct ggml_cgraph * g) {
    mi_backend_ctx * c = (mi_backend_ctx *) backend->context;
    (void) hipGetLastError();      // a stale thread-level error (another library's probe, a failed allocation elsewhere) is not this graph's: only what this call raises counts
    try {
        static const bool timing = getenv("GGML_MI355X_HOST_TIMING") != nullptr;
        const auto t0 = std::chrono::steady_clock::now();
        const enum ggml_status st = be_graph_compute_impl(backend, g);
        if (timing) {
            g_host_ns += std::chrono::duration<double, std::nano>(std::chrono::steady_clock::now() - t0).count();
            if (++g_host_calls % 256 == 0) fprintf(stderr, "ggml-mi355x: graph_compute host time %.1f us per call over %ld calls\n", g_host_ns/g_host_calls/1e3, g_host_calls);
        }
        // a kernel launch that failed (bad configuration, out of resources) shows up as the runtime's last error
        const hipError_t le = hipGetLastError();
        if (st == GGML_STATUS_SUCCESS && le != hipSuccess) { MI_LOG("graph_compute: %s (%s)\n", hipGetErrorName(le), hipGetErrorString(le)); return GGML_STATUS_FAILED; }
        return st;
    } catch (const mi_graph_error & e) {
        MI_LOG("graph_compute failed: HIP error %s (%s) at %s:%d\n", hipGetErrorName(e.err), hipGetErrorString(e.err), e.file, e.line);
        if (c->capturing) {       // leave the stream usable: end the capture, drop what was recorded
            hipGraph_t graph = nullptr;
            (void) hipStreamEndCapture(c->stream, &graph);
            if (graph) (void) hipGraphDestroy(graph);
            c->capturing = false;
        }
        if (c->cap_entry) { free_entry(*c->cap_entry); c->cap_entry = nullptr; }      // a partly captured graph is not replayable
        c->aq.valid = false; c->pp.active = false; c->up.n = 0;
        // a throw inside the row-split fork leaves the current device on a peer, and peer streams that were already launched may still write dst:
        // back to this backend's device, and nothing is returned before they have finished
        set_device(c->device);
        for (int i = 0; i < G().n_devices; i++) if (c->peers[i].stream) { set_device(G().devices[i].id); (void) hipStreamSynchronize(c->peers[i].stream); }
        set_device(c->device);
        (void) hipStreamSynchronize(c->stream);
        (void) hipGetLastError();
        return GGML_STATUS_FAILED;
    }
}
static enum ggml_status be_graph_compute_impl(ggml_backend_t backend, struct ggml_cgraph * g) {
    mi_backend_ctx * c = (mi_backend_ctx *) backend->context;
    set_device(c->device);
    c->cnt.graphs_computed++;
    uploads_flush(c);       // the inputs the host has set (set_tensor_async) go out first, as one launch

    static const bool timing2 = getenv("GGML_MI355X_HOST_TIMING") != nullptr;
    static double t_first = 0, t_rest = 0; static long n_rep = 0;
    const bool try_graph = c->use_graphs && (!c->profiling || c->prof_in_graph) && g->n_nodes >= 8;

    // ---- fast path: replay segment by segment ----
    if (try_graph) {
        const auto ta = std::chrono::steady_clock::now();
        graph_entry * e = replay_candidate(c, g);
        if (e && topology_matches(*e, g)) {
            size_t si = 0; auto tb = ta;
            for (; si < e->segs.size(); si++) {
                const graph_seg & sg = e->segs[si];
                if (!signature_matches(*e, g, sg.begin, sg.end)) break;
                if (sg.exec) MI_CHECK_G(hipGraphLaunch(sg.exec, c->stream));
                if (si == 0 && timing2) tb = std::chrono::steady_clock::now();
            }
            if (si == e->segs.size()) {
                e->last_use = ++c->graph_tick; e->seen++;
                c->cnt.graph_replays++;
                if (timing2) {
                    const auto tc = std::chrono::steady_clock::now();
                    t_first += std::chrono::duration<double, std::micro>(tb - ta).count(); t_rest += std::chrono::duration<double, std::micro>(tc - tb).count();
                    if (++n_rep % 128 == 0) fprintf(stderr, "ggml-mi355x: replay host us: until segment 0 is launched %.1f, the other %zu segments %.1f (n_nodes %d, %zu cached)\n", t_first/n_rep, e->segs.size() - 1, t_rest/n_rep, g->n_nodes, c->graphs.size());
                }
                return GGML_STATUS_SUCCESS;
            }
            if (si > 0) {      // same topology, a parameter changed behind an identical prefix: the prefix is on the stream, the rest runs eagerly
                c->prof_suspend = false;
                run_nodes(c, g, e->segs[si].begin);
                return GGML_STATUS_SUCCESS;
            }
        }
    }

    // ---- everything else: buffers this graph needs (allocated outside any capture; a captured graph holds their addresses) ----
    if (!c->rope_tab) { if (hipMalloc((void **) &c->rope_tab, 4096) != hipSuccess) { (void) hipGetLastError(); c->rope_tab = nullptr; } }
    if (!c->moe_ws) {
        if (hipMalloc((void **) &c->moe_ws, 4096) == hipSuccess) { MI_CHECK_G(hipMemsetAsync(c->moe_ws, 0, 4096, c->stream)); MI_CHECK_G(hipStreamSynchronize(c->stream)); }
        else { (void) hipGetLastError(); c->moe_ws = nullptr; }
    }
    if (!c->err_host) {      // the error words of the kernels with bounded waits (host-mapped)
        if (hipHostMalloc((void **) &c->err_host, 64, hipHostMallocMapped) == hipSuccess && hipHostGetDevicePointer((void **) &c->err_dev, c->err_host, 0) == hipSuccess) memset(c->err_host, 0, 64);
        else { (void) hipGetLastError(); c->err_host = nullptr; c->err_dev = nullptr; }
    }
    if (!c->fin_img) {       // allocated once, outside any capture; the counters are zero between launches (the kernels re-arm them)
        if (hipMalloc(&c->fin_img, mi_backend_ctx::FIN_IMG_BYTES) != hipSuccess) { (void) hipGetLastError(); c->fin_img = nullptr; }
        if (c->fin_img && hipMalloc((void **) &c->fin_cnt, (mi_backend_ctx::FIN_COUNTERS + 8)*4) == hipSuccess) {
            MI_CHECK_G(hipMemsetAsync(c->fin_cnt, 0, (mi_backend_ctx::FIN_COUNTERS + 8)*4, c->stream)); MI_CHECK_G(hipStreamSynchronize(c->stream));
        } else if (c->fin_img) { (void) hipGetLastError(); (void) hipFree(c->fin_img); c->fin_img = nullptr; c->fin_cnt = nullptr; }
    }
    if (!c->attn_part) {     // <= 8 tokens x 128 heads x 32 ranges x (128 + 2) floats: allocated once, outside any capture
        const size_t pb = (size_t) 8*128*32*130*4;
        if (hipMalloc((void **) &c->attn_part, pb) == hipSuccess) c->attn_part_bytes = pb; else (void) hipGetLastError();
    }
    const size_t need = graph_scratch_need(g);
    if (need > c->scratch_size) {
        MI_CHECK_G(hipStreamSynchronize(c->stream));
        if (c->scratch) MI_CHECK_G(hipFree(c->scratch));
        c->scratch = nullptr; c->scratch_size = 0;
        drop_graphs(c);   // captured graphs hold the old scratch pointer
        const size_t sz = need + (need >> 2);
        if (hipMalloc(&c->scratch, sz) != hipSuccess) { (void) hipGetLastError(); return GGML_STATUS_ALLOC_FAILED; }
        c->scratch_size = sz;
    }
    {
        size_t kv_need = 0;
        for (int i = 0; i < g->n_nodes; i++) if (g->nodes[i]->op == GGML_OP_FLASH_ATTN_EXT) { size_t kb, vb; fa_kv16_plan(g->nodes[i], kb, vb); kv_need = std::max(kv_need, kb + vb); }
        if (kv_need > c->kv16_size) {
            MI_CHECK_G(hipStreamSynchronize(c->stream));
            if (c->kv16) MI_CHECK_G(hipFree(c->kv16));
            c->kv16 = nullptr; c->kv16_size = 0;
            drop_graphs(c);   // captured graphs hold the old pointer
            const size_t sz = kv_need + (kv_need >> 1);      // (the cache view grows with the context: fewer re-allocations)
            if (hipMalloc(&c->kv16, sz) != hipSuccess) { (void) hipGetLastError(); return GGML_STATUS_ALLOC_FAILED; }
            c->kv16_size = sz;
        }
    }

    {
        size_t cv_need = 0;
        for (int i = 0; i < g->n_nodes; i++) {
            const struct ggml_tensor * nd = g->nodes[i];
            if (nd->op == GGML_OP_MUL_MAT && ggml_is_quantized(nd->src[0]->type) && nd->src[1]->type == GGML_TYPE_F16) cv_need = std::max(cv_need, (size_t) ggml_nelements(nd->src[1])*4 + 256);
        }
        if (cv_need > c->cvt_size) {
            MI_CHECK_G(hipStreamSynchronize(c->stream));
            if (c->cvt) MI_CHECK_G(hipFree(c->cvt));
            c->cvt = nullptr; c->cvt_size = 0;
            drop_graphs(c);
            if (hipMalloc(&c->cvt, cv_need) != hipSuccess) { (void) hipGetLastError(); return GGML_STATUS_ALLOC_FAILED; }
            c->cvt_size = cv_need;
        }
    }

    c->split_graph = false;
    for (int i = 0; i < g->n_nodes && !c->split_graph; i++) c->split_graph = g->nodes[i]->op == GGML_OP_MUL_MAT && tensor_is_split(g->nodes[i]->src[0]);
    if (c->split_graph) {     // several devices' streams take part: eager execution, no capture (the fusion matchers leave split weights to op_mul_mat_split)
        run_nodes(c, g);
        return GGML_STATUS_SUCCESS;
    }
    c->prof_suspend = try_graph && c->prof_in_graph;     // only what is captured gets recorded in that mode
    if (try_graph) {
        graph_entry & e = graph_lookup(c, g);
        if (!e.segs.empty()) {      // (a cached graph the fast path did not pick: e.g. several sequences' graphs taking turns)
            for (const graph_seg & sg : e.segs) if (sg.exec) MI_CHECK_G(hipGraphLaunch(sg.exec, c->stream));
            c->cnt.graph_replays++;
            return GGML_STATUS_SUCCESS;
        }
        if (e.seen >= 2) {
            c->cap_entry = &e;
            MI_CHECK_G(hipStreamBeginCapture(c->stream, hipStreamCaptureModeThreadLocal));
            c->prof_suspend = false;
            c->capturing = true;
            run_nodes(c, g, 0, &e);
            c->cap_entry = nullptr;
            c->prof_suspend = c->prof_in_graph;
            c->cnt.graph_captures++;
            for (const graph_seg & sg : e.segs) if (sg.exec) MI_CHECK_G(hipGraphLaunch(sg.exec, c->stream));
            return GGML_STATUS_SUCCESS;
        }
    }
    run_nodes(c, g);
    return GGML_STATUS_SUCCESS;
}

static void be_event_record(ggml_backend_t backend, ggml_backend_event_t event) {
    mi_backend_ctx * c = (mi_backend_ctx *) backend->context;
    set_device(c->device);
    uploads_flush(c);
    MI_CHECK(hipEventRecord((hipEvent_t) event->context, c->stream));
}
static void be_event_wait(ggml_backend_t backend, ggml_backend_event_t event) {
    mi_backend_ctx * c = (mi_backend_ctx *) backend->context;
    set_device(c->device);
    uploads_flush(c);
    MI_CHECK(hipStreamWaitEvent(c->stream, (hipEvent_t) event->context, 0));
}

static const struct ggml_backend_i mi_backend_iface = {
    /* .get_name           = */ be_get_name,
    /* .free               = */ be_free,
    /* .set_tensor_async   = */ be_set_tensor_async,
    /* .get_tensor_async   = */ be_get_tensor_async,
    /* .cpy_tensor_async   = */ be_cpy_tensor_async,
    /* .synchronize        = */ be_synchronize,
    /* .graph_plan_create  = */ NULL,
    /* .graph_plan_free    = */ NULL,
    /* .graph_plan_update  = */ NULL,
    /* .graph_plan_compute = */ NULL,
    /* .graph_compute      = */ be_graph_compute,
    /* .event_record       = */ be_event_record,
    /* .event_wait         = */ be_event_wait,
};

// ---------------------------------------------------------------------------------------------
// device
// ---------------------------------------------------------------------------------------------
static const char * dev_get_name(ggml_backend_dev_t dev) { return ((mi_device *) dev->context)->name.c_str(); }
static const char * dev_get_description(ggml_backend_dev_t dev) { return ((mi_device *) dev->context)->description.c_str(); }
static void dev_get_memory(ggml_backend_dev_t dev, size_t * free, size_t * total) {
    mi_device * d = (mi_device *) dev->context;
    set_device(d->id);
    MI_CHECK(hipMemGetInfo(free, total));
}
static enum ggml_backend_dev_type dev_get_type(ggml_backend_dev_t) { return GGML_BACKEND_DEVICE_TYPE_GPU; }
static void dev_get_props(ggml_backend_dev_t dev, struct ggml_backend_dev_props * props) {
    props->name = dev_get_name(dev);
    props->description = dev_get_description(dev);
    props->type = GGML_BACKEND_DEVICE_TYPE_GPU;
    dev_get_memory(dev, &props->memory_free, &props->memory_total);
    props->caps.async = true;
    props->caps.host_buffer = true;
    props->caps.buffer_from_host_ptr = false;
    props->caps.events = true;
}
static ggml_backend_t dev_init_backend(ggml_backend_dev_t dev, const char *) {
    return ggml_backend_mi355x_init((int)((mi_device *) dev->context - G().devices));
}
static ggml_backend_buffer_type_t dev_get_buffer_type(ggml_backend_dev_t dev) { return &((mi_device *) dev->context)->buft; }
static ggml_backend_buffer_type_t dev_get_host_buffer_type(ggml_backend_dev_t) { return ggml_backend_mi355x_host_buffer_type(); }
static bool dev_supports_op(ggml_backend_dev_t, const struct ggml_tensor * op) { return mi_supports_op(op); }
static bool dev_supports_buft(ggml_backend_dev_t dev, ggml_backend_buffer_type_t buft) {
    if (buft->iface.get_name == host_buft_get_name) return false;   // kernels read device memory only
    if (buft_is_split(buft)) return ((mi_split_buft_ctx *) buft->context)->main_device == (int)((mi_device *) dev->context - G().devices);
    if (buft->iface.get_name != buft_get_name) return false;
    return ((mi_device *) buft->context)->id == ((mi_device *) dev->context)->id;
}
static bool dev_offload_op(ggml_backend_dev_t, const struct ggml_tensor * op) {
    // worth uploading host-resident weights for large batches only (same rule of thumb as the reference's GPU backends)
    const int min_batch = 32;
    if (op->op == GGML_OP_GET_ROWS) return false;
    if (op->op == GGML_OP_MUL_MAT_ID) return op->ne[2] >= min_batch;
    return op->ne[1] >= min_batch;
}
static ggml_backend_event_t dev_event_new(ggml_backend_dev_t dev) {
    mi_device * d = (mi_device *) dev->context;
    set_device(d->id);
    hipEvent_t ev;
    MI_CHECK(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
    return new ggml_backend_event{ dev, ev };
}
static void dev_event_free(ggml_backend_dev_t, ggml_backend_event_t event) {
    MI_CHECK(hipEventDestroy((hipEvent_t) event->context));
    delete event;
}
static void dev_event_synchronize(ggml_backend_dev_t, ggml_backend_event_t event) { MI_CHECK(hipEventSynchronize((hipEvent_t) event->context)); }

static const struct ggml_backend_device_i mi_device_iface = {
    dev_get_name, dev_get_description, dev_get_memory, dev_get_type, dev_get_props, dev_init_backend,
    dev_get_buffer_type, dev_get_host_buffer_type, NULL /* buffer_from_host_ptr */,
    dev_supports_op, dev_supports_buft, dev_offload_op, dev_event_new, dev_event_free, dev_event_synchronize,
};

// ---------------------------------------------------------------------------------------------
// registry
// ---------------------------------------------------------------------------------------------
static const char * reg_get_name(ggml_backend_reg_t) { return GGML_MI355X_NAME; }
static size_t reg_get_device_count(ggml_backend_reg_t) { return (size_t) G().n_devices; }
static ggml_backend_dev_t reg_get_device(ggml_backend_reg_t, size_t index) {
    GGML_ASSERT(index < (size_t) G().n_devices);
    return &G().devices[index].dev;
}

static struct ggml_backend_feature * mi_get_features(ggml_backend_reg_t) {
    static struct ggml_backend_feature features[] = {
        { "ARCH", "gfx950" }, { "WAVE", "64" }, { "MMVQ", "sdot4+dpp" }, { "MMQ", "dequant_bf16+mfma_f32_32x32x16_bf16" }, { "GRAPHS", "1" }, { NULL, NULL },
    };
    return features;
}

static void * reg_get_proc_address(ggml_backend_reg_t, const char * name) {
    if (strcmp(name, "ggml_backend_get_features") == 0)        return (void *) mi_get_features;
    if (strcmp(name, "ggml_backend_split_buffer_type") == 0)   return (void *) ggml_backend_mi355x_split_buffer_type;
    if (strcmp(name, "ggml_backend_mi355x_get_stream") == 0)   return (void *) ggml_backend_mi355x_get_stream;
    if (strcmp(name, "ggml_backend_mi355x_get_counters") == 0) return (void *) ggml_backend_mi355x_get_counters;
    if (strcmp(name, "ggml_backend_mi355x_reset_counters") == 0) return (void *) ggml_backend_mi355x_reset_counters;
    if (strcmp(name, "ggml_backend_mi355x_set_option") == 0)   return (void *) ggml_backend_mi355x_set_option;
    if (strcmp(name, "ggml_backend_mi355x_get_profile") == 0)  return (void *) ggml_backend_mi355x_get_profile;
    if (strcmp(name, "ggml_backend_mi355x_tensor_set_from_device_async") == 0) return (void *) ggml_backend_mi355x_tensor_set_from_device_async;
    if (strcmp(name, "ggml_backend_mi355x_tensor_get_to_device_async") == 0)   return (void *) ggml_backend_mi355x_tensor_get_to_device_async;
    if (strcmp(name, "ggml_backend_mi355x_test_quantize") == 0) return (void *) ggml_backend_mi355x_test_quantize;
    if (strcmp(name, "ggml_backend_mi355x_test_hbm_read_gbps") == 0) return (void *) ggml_backend_mi355x_test_hbm_read_gbps;
    return NULL;
}

static const struct ggml_backend_reg_i mi_reg_iface = { reg_get_name, reg_get_device_count, reg_get_device, reg_get_proc_address };

static bool arch_is_gfx950(const hipDeviceProp_t & prop) { return strncmp(prop.gcnArchName, "gfx950", 6) == 0; }

static mi_globals & G() {
    static mi_globals g;
    static std::once_flag once;
    std::call_once(once, [] {
        int n = 0;
        if (hipGetDeviceCount(&n) != hipSuccess) { (void) hipGetLastError(); n = 0; }
        g.reg = { GGML_BACKEND_API_VERSION, mi_reg_iface, NULL };
        g.host_buft = { mi_host_buft_iface, NULL, NULL };
        // GGML_MI355X_VIRTUAL_DEVICES=v lists every GPU v times (same HIP device, separate backend / stream / buffer type each): lets the
        // multi-device paths (layer split in one process, row split) be exercised on a one-GPU box
        const int virt = getenv("GGML_MI355X_VIRTUAL_DEVICES") ? std::max(1, atoi(getenv("GGML_MI355X_VIRTUAL_DEVICES"))) : 1;
        for (int iv = 0; iv < n*virt && g.n_devices < GGML_MI355X_MAX_DEVICES; iv++) {
            const int i = iv/virt;
            hipDeviceProp_t prop;
            if (hipGetDeviceProperties(&prop, i) != hipSuccess) { (void) hipGetLastError(); continue; }
            if (!arch_is_gfx950(prop)) {
                MI_LOG("skipping device %d (%s, %s): this backend carries gfx950 code objects only\n", i, prop.name, prop.gcnArchName);
                continue;
            }
            mi_device & d = g.devices[g.n_devices];
            d.id = i;
            d.name = std::string(GGML_MI355X_NAME) + std::to_string(g.n_devices);
            d.description = prop.name;
            d.total_mem = prop.totalGlobalMem;
            d.dev = { mi_device_iface, &g.reg, &d };
            d.buft = { mi_buft_iface, &d.dev, &d };
            g.n_devices++;
        }
        if (g.n_devices > 0) g.host_buft.device = &g.devices[0].dev;
        // peer access for the layer-split hand-off over xGMI
        for (int i = 0; i < g.n_devices; i++) {
            for (int j = 0; j < g.n_devices; j++) {
                if (g.devices[i].id == g.devices[j].id) continue;
                int can = 0;
                if (hipDeviceCanAccessPeer(&can, g.devices[i].id, g.devices[j].id) == hipSuccess && can) {
                    (void) hipSetDevice(g.devices[i].id);
                    hipError_t e = hipDeviceEnablePeerAccess(g.devices[j].id, 0);
                    if (e != hipSuccess) (void) hipGetLastError();
                }
            }
        }
        g.initialised = true;
    });
    return g;
}

// ---------------------------------------------------------------------------------------------
// exported C-ABI
// ---------------------------------------------------------------------------------------------
extern "C" {

ggml_backend_reg_t ggml_backend_mi355x_reg(void) { return &G().reg; }

ggml_backend_reg_t ggml_backend_init(void) { return ggml_backend_mi355x_reg(); }

int ggml_backend_score(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) { (void) hipGetLastError(); return 0; }
    for (int i = 0; i < n; i++) {
        hipDeviceProp_t prop;
        if (hipGetDeviceProperties(&prop, i) == hipSuccess && arch_is_gfx950(prop)) return 100;
    }
    return 0;
}

int ggml_backend_mi355x_get_device_count(void) { return G().n_devices; }

void ggml_backend_mi355x_get_device_description(int device, char * description, size_t description_size) {
    GGML_ASSERT(device >= 0 && device < G().n_devices);
    snprintf(description, description_size, "%s", G().devices[device].description.c_str());
}
void ggml_backend_mi355x_get_device_memory(int device, size_t * free, size_t * total) {
    GGML_ASSERT(device >= 0 && device < G().n_devices);
    dev_get_memory(&G().devices[device].dev, free, total);
}

ggml_backend_buffer_type_t ggml_backend_mi355x_buffer_type(int device) {
    if (device < 0 || device >= G().n_devices) return NULL;
    return &G().devices[device].buft;
}
ggml_backend_buffer_type_t ggml_backend_mi355x_host_buffer_type(void) {
    G();
    return &G().host_buft;
}

ggml_backend_t ggml_backend_mi355x_init(int device) {
    if (device < 0 || device >= G().n_devices) {
        MI_LOG("invalid device %d (have %d gfx950 device(s))\n", device, G().n_devices);
        return NULL;
    }
    mi_device & d = G().devices[device];
    set_device(d.id);
    mi_backend_ctx * c = new mi_backend_ctx;
    c->device = d.id;
    c->dev_index = device;
    c->name = d.name;
    MI_CHECK(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
    if (const char * e = getenv("GGML_MI355X_GRAPHS")) c->use_graphs = atoi(e) != 0;
    if (const char * e = getenv("GGML_MI355X_FUSION")) c->use_fusion = atoi(e) != 0;
    ggml_backend_t backend = new ggml_backend{ mi_guid(), mi_backend_iface, &d.dev, c };
    return backend;
}

ggml_backend_buffer_type_t ggml_backend_mi355x_split_buffer_type(int main_device, const float * tensor_split) {
    static std::mutex mu;
    static std::vector<mi_split_buft_ctx *> made;          // buffer types live as long as the process (the host keeps the pointer in its buft lists)
    const int n = G().n_devices;
    if (main_device < 0 || main_device >= n) return NULL;
    float share[GGML_MI355X_MAX_DEVICES]; double sum = 0;
    for (int i = 0; i < n; i++) { share[i] = tensor_split && tensor_split[i] > 0 ? tensor_split[i] : 0.0f; sum += share[i]; }
    if (sum == 0) { for (int i = 0; i < n; i++) share[i] = 1.0f; sum = n; }       // all zero: equal shares
    mi_split_buft_ctx want; want.main_device = main_device;
    double acc = 0;
    for (int i = 0; i < n; i++) { want.cum[i] = (float)(acc/sum); acc += share[i]; }
    want.cum[n] = 1.0f;
    // every device that CAN receive rows must be mapped into the main device's address space and back: not only those with a share — the rounding of
    // split_rows hands remainder rows to the devices behind the last share too (shares [1, 0], 300 rows: 44 rows on device 1)
    // From split_rows' rounding: device i < n - 1 holds rows [round64(nrows*cum[i]), round64(nrows*cum[i + 1])) — empty exactly when its share is zero — and the LAST
    // device holds everything behind round64(nrows*cum[n - 1]), which is non-empty for some row count whatever its share (a fixed list of probe counts
    // missed row counts like 2880)
    bool gets_rows[GGML_MI355X_MAX_DEVICES] = {};
    for (int i = 0; i < n; i++) gets_rows[i] = share[i] > 0 || i == n - 1;
    for (int i = 0; i < n; i++) {
        if ((share[i] == 0 && !gets_rows[i]) || G().devices[i].id == G().devices[main_device].id) continue;
        int ab = 0, ba = 0;
        if (hipDeviceCanAccessPeer(&ab, G().devices[i].id, G().devices[main_device].id) != hipSuccess || hipDeviceCanAccessPeer(&ba, G().devices[main_device].id, G().devices[i].id) != hipSuccess || !ab || !ba) {
            (void) hipGetLastError();
            MI_LOG("row split: no peer mapping between %s and %s\n", G().devices[i].name.c_str(), G().devices[main_device].name.c_str());
            return NULL;
        }
    }
    std::lock_guard<std::mutex> lock(mu);
    for (mi_split_buft_ctx * m : made) if (m->main_device == main_device && memcmp(m->cum, want.cum, sizeof(want.cum)) == 0) return &m->buft;
    mi_split_buft_ctx * m = new mi_split_buft_ctx(want);
    if (made.empty()) MI_LOG("row split (-sm row) is EXPERIMENTAL on this backend: functional, not tuned, and so far exercised on virtual devices only\n");
    m->name = std::string(GGML_MI355X_NAME) + "_Split";
    m->buft = { mi_split_buft_iface, &G().devices[main_device].dev, m };
    made.push_back(m);
    return &m->buft;
}

bool ggml_backend_is_mi355x(ggml_backend_t backend) { return backend != NULL && ggml_guid_matches(backend->guid, mi_guid()); }

void ggml_backend_mi355x_tensor_set_from_device_async(ggml_backend_t backend, struct ggml_tensor * tensor, const void * dev_src, size_t offset, size_t size) {
    GGML_ASSERT(ggml_backend_is_mi355x(backend));
    mi_backend_ctx * c = (mi_backend_ctx *) backend->context;
    set_device(c->device);
    uploads_flush(c);
    MI_CHECK(hipMemcpyAsync((char *) tensor->data + offset, dev_src, size, hipMemcpyDeviceToDevice, c->stream));
}
void ggml_backend_mi355x_tensor_get_to_device_async(ggml_backend_t backend, const struct ggml_tensor * tensor, void * dev_dst, size_t offset, size_t size) {
    GGML_ASSERT(ggml_backend_is_mi355x(backend));
    mi_backend_ctx * c = (mi_backend_ctx *) backend->context;
    set_device(c->device);
    uploads_flush(c);
    MI_CHECK(hipMemcpyAsync(dev_dst, (const char *) tensor->data + offset, size, hipMemcpyDeviceToDevice, c->stream));
}
void * ggml_backend_mi355x_get_stream(ggml_backend_t backend) {
    GGML_ASSERT(ggml_backend_is_mi355x(backend));
    return (void *) ((mi_backend_ctx *) backend->context)->stream;
}
void ggml_backend_mi355x_get_counters(ggml_backend_t backend, struct ggml_backend_mi355x_counters * out) {
    GGML_ASSERT(ggml_backend_is_mi355x(backend));
    *out = ((mi_backend_ctx *) backend->context)->cnt;
}
void ggml_backend_mi355x_reset_counters(ggml_backend_t backend) {
    GGML_ASSERT(ggml_backend_is_mi355x(backend));
    ((mi_backend_ctx *) backend->context)->cnt = {};
}
int ggml_backend_mi355x_set_option(ggml_backend_t backend, const char * key, int value) {
    GGML_ASSERT(ggml_backend_is_mi355x(backend));
    mi_backend_ctx * c = (mi_backend_ctx *) backend->context;
    if (strcmp(key, "graphs") == 0) { c->use_graphs = value != 0; return 0; }
    if (strcmp(key, "profile") == 0) {
        // 1: eager, an event pair around every quantized mat-mul launch; 2: the same pairs captured into the hipGraphs (what the
        // replayed token really costs per launch, without the eager launch gaps); either way the graphs are rebuilt
        MI_CHECK(hipStreamSynchronize(c->stream));
        drop_graphs(c);
        for (auto & r : c->prof) { c->ev_pool.push_back(r.e0); c->ev_pool.push_back(r.e1); }
        c->prof.clear();
        c->profiling = value != 0; c->prof_in_graph = value == 2; c->prof_suspend = false;
        mul_mat_vec_q_fused_set_hooks(c->profiling ? prof_hook_pre : nullptr, c->profiling ? prof_hook_post : nullptr, c);
        return 0;
    }
    if (strcmp(key, "fusion") == 0) {
        c->use_fusion = value != 0;
        MI_CHECK(hipStreamSynchronize(c->stream));
        drop_graphs(c);
        return 0;
    }
    if (strcmp(key, "moe_dual") == 0) {       // the prompt pass's expert chain: -1 = by pairs per expert (default), 0 = gate / up as two launches, 1 = the dual launch
        c->moe_dual = value;
        MI_CHECK(hipStreamSynchronize(c->stream));
        drop_graphs(c);
        return 0;
    }
    return -1;
}

// per-kernel timing of the quantized mat-mul launches recorded while option "profile" was on (SURVEY.md §5:
// "expose per-kernel bytes/time counters"). Synchronises the stream, aggregates by (type, m, k, n), clears the log.
int ggml_backend_mi355x_get_profile(ggml_backend_t backend, struct ggml_backend_mi355x_prof_entry * out, int cap) {
    GGML_ASSERT(ggml_backend_is_mi355x(backend));
    mi_backend_ctx * c = (mi_backend_ctx *) backend->context;
    set_device(c->device);
    MI_CHECK(hipStreamSynchronize(c->stream));
    int n = 0;
    for (auto & r : c->prof) {
        float ms = 0.0f;
        if (hipEventElapsedTime(&ms, r.e0, r.e1) != hipSuccess) { (void) hipGetLastError(); continue; }     // a captured pair whose graph never ran
        if (!c->prof_in_graph) { c->ev_pool.push_back(r.e0); c->ev_pool.push_back(r.e1); }
        int j = 0;
        for (; j < n; j++) if (out[j].type == r.type && out[j].m == r.m && out[j].k == r.k && out[j].n == r.n) break;
        if (j == n) {
            if (n == cap) continue;
            out[n] = { r.type, (int32_t) r.n, r.m, r.k, 0, 0.0, r.bytes, {0} };
            if (r.kernel) snprintf(out[n].kernel, sizeof(out[n].kernel), "%s", r.kernel);
            n++;
        }
        out[j].launches++; out[j].total_ms += ms;
    }
    if (!c->prof_in_graph) c->prof.clear();      // captured pairs stay with their graphs until the option changes
    return n;
}

// ---- test hooks (reached by name through get_proc_address; not part of the ggml contract) --------------
// quantize host activations on the device and return the int8 blocks: lets tests/ compare the device
// quantizer with the oracle bit for bit.
int ggml_backend_mi355x_test_quantize(ggml_backend_t backend, const float * x, int64_t k, int64_t n, int kind,
                                      int8_t * qs, float * d, int16_t * bsums) {
    GGML_ASSERT(ggml_backend_is_mi355x(backend));
    mi_backend_ctx * c = (mi_backend_ctx *) backend->context;
    set_device(c->device);
    if (kind != T_Q8_0 && kind != T_Q8_K) return -1;
    if (k % (kind == T_Q8_0 ? 32 : 256) != 0) return -2;
    float * dx = nullptr; void * scratch = nullptr;
    const size_t sb = act_q8_bytes(kind, k, n);
    MI_CHECK(hipMalloc(&dx, (size_t) k*n*4));
    MI_CHECK(hipMalloc(&scratch, sb));
    MI_CHECK(hipMemcpy(dx, x, (size_t) k*n*4, hipMemcpyHostToDevice));
    const act_q8 q = act_q8_carve(scratch, kind, k, n);
    quantize_act(dx, n, (size_t) k*4, 0, q, c->stream);
    MI_CHECK(hipStreamSynchronize(c->stream));
    const int64_t nd = kind == T_Q8_0 ? k/32 : k/256, nbs = kind == T_Q8_0 ? k/32 : k/16;
    MI_CHECK(hipMemcpy(qs, q.qs, (size_t) k*n, hipMemcpyDeviceToHost));
    MI_CHECK(hipMemcpy(d, q.d, (size_t) nd*n*4, hipMemcpyDeviceToHost));
    MI_CHECK(hipMemcpy(bsums, q.bsums, (size_t) nbs*n*2, hipMemcpyDeviceToHost));
    MI_CHECK(hipFree(dx)); MI_CHECK(hipFree(scratch));
    return 0;
}

// achievable HBM read rate on this box (GB/s) with the same 16 B/lane nontemporal loads the mat-vec kernels use
double ggml_backend_mi355x_test_hbm_read_gbps(ggml_backend_t backend, size_t bytes, int iters) {
    GGML_ASSERT(ggml_backend_is_mi355x(backend));
    mi_backend_ctx * c = (mi_backend_ctx *) backend->context;
    set_device(c->device);
    void * p = nullptr; unsigned * sink = nullptr;
    MI_CHECK(hipMalloc(&p, bytes)); MI_CHECK(hipMalloc(&sink, 256));
    MI_CHECK(hipMemset(p, 1, bytes));
    hipEvent_t e0, e1;
    MI_CHECK(hipEventCreate(&e0)); MI_CHECK(hipEventCreate(&e1));
    hbm_read_probe(p, bytes, sink, c->stream);
    MI_CHECK(hipEventRecord(e0, c->stream));
    for (int i = 0; i < iters; i++) hbm_read_probe(p, bytes, sink, c->stream);
    MI_CHECK(hipEventRecord(e1, c->stream));
    MI_CHECK(hipEventSynchronize(e1));
    float ms = 0; MI_CHECK(hipEventElapsedTime(&ms, e0, e1));
    MI_CHECK(hipEventDestroy(e0)); MI_CHECK(hipEventDestroy(e1));
    MI_CHECK(hipFree(p)); MI_CHECK(hipFree(sink));
    return (double) bytes*iters/(ms*1e-3)/1e9;
}

} // extern "C"
