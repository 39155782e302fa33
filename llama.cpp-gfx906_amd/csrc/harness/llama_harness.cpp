// llama_harness.cpp — synthetic Llama-family model driver (HARNESS, caller side of the boundary).
//
// The reference's libllama cannot be built here (its ggml/ submodule is empty), and no GGUF weights
// exist on the GPU box, so this file restates — through the ggml API only, exactly as libllama uses it —
// the minimum of the CALLER rows of SURVEY.md §8:
//   a9  llm_build_llama               src/llama-model.cpp:5969-6123   (per-layer op sequence)
//   a6  build_attn_mha (no-FA branch) src/llama-graph.cpp:1220-1341
//   a7  build_attn / cpy_k / cpy_v    src/llama-graph.cpp:1438-1488, src/llama-kv-cache-unified.cpp:1056-1190
//   a4  build_ffn (SILU, PAR)         src/llama-graph.cpp:632-774
//   a8  build_norm (RMS)              src/llama-graph.cpp:597-630
//   a11 process_ubatch / decode       src/llama-context.cpp:714-776,946-1254 (set_inputs, graph reuse,
//                                     async compute, logits readback, synchronize)
//   a15 llama_tensor_get_type         src/llama-quant.cpp:178-434 (which tensor gets which type in a "Q4_K_M" file)
// with random VALID blocks as weights (SURVEY.md §8d "Concrete synthetic inputs") and the llama-bench
// protocol on top (tools/llama-bench/llama-bench.cpp:1762-1810). It contains no arithmetic of its own:
// every FLOP happens in the backend it is given.
#include "ggml.h"
#include "ggml-backend.h"

#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "gguf_file.h"

#include <map>
#include <memory>
#include <string>
#include <thread>
#include <tuple>
#include <vector>

extern "C" {

// llama_ftype values (include/llama.h) for the configs of BASELINE.json
enum { MI_FTYPE_Q4_0 = 2, MI_FTYPE_Q8_0 = 7, MI_FTYPE_Q4_K_M = 15, MI_FTYPE_Q5_K_M = 17, MI_FTYPE_Q6_K = 18, MI_FTYPE_MXFP4_MOE = 38 };   // llama_ftype ids (include/llama.h)

struct mi_llama_hparams {
    int32_t n_embd, n_ff, n_layer, n_head, n_head_kv, n_embd_head, n_vocab;
    int32_t n_ctx;            // KV cache size (llama-bench: n_prompt + n_gen, tools/llama-bench/llama-bench.cpp:1005)
    int32_t ftype;            // MI_FTYPE_*
    int32_t rope_type;        // 0 = NORM (llama), 2 = NEOX
    int32_t n_ctx_orig;
    int32_t has_rope_freqs;   // Llama-3.1 rope_freqs tensor (src/llama-model.cpp:2218)
    int32_t is_70b;           // the LLM_TYPE_70B attn_v bump (src/llama-quant.cpp:305-310)
    float   rope_freq_base, rope_freq_scale, f_norm_rms_eps;
    int32_t layer_begin, layer_end;   // this instance holds layers [begin, end) — layer split (src/llama-model.cpp:1949-1972)
    int32_t has_output;               // holds output_norm + output (the last device, src/llama-model.cpp:1972)
    int32_t n_seq_max;                // independent sequences, each with its own KV cache stream (llama_context_params.n_seq_max, kv_unified = false)
    int32_t n_expert, n_expert_used;  // > 0: the FFN is build_moe_ffn (src/llama-graph.cpp:811-1023); n_ff is then the expert width
    int32_t arch;                     // 0 = llm_build_llama (dense or Mixtral-style MoE), 1 = llm_build_openai_moe_iswa (gpt-oss; src/llama-model.cpp:17610-17738)
    int32_t flash_attn;               // -fa 1: ggml_flash_attn_ext, V cache not transposed, n_kv padded to 256, F16 mask (src/llama-graph.cpp:1245-1265)
    int32_t n_swa, swa_pattern;       // > 0: sliding-window attention on the layers il % swa_pattern < swa_pattern - 1 (llama_hparams::set_swa_pattern,
                                      // src/llama-hparams.cpp:5-13; gpt-oss: 128 / 2), which get their own, smaller cache (llama_kv_cache_unified_iswa)
    int32_t n_ubatch;                 // most tokens per decode call; sizes the window cache: min(n_ctx, PAD(n_swa + n_ubatch)) (src/llama-kv-cache-unified-iswa.cpp:46-60)
    int32_t type_k;                   // K cache type: 0 = F16 (default), GGML_TYPE_Q8_0 / Q4_0 = llama-bench -ctk (cpy_k: SET_ROWS quantizes the row; get_k: a quantized
                                      // src0 of the K.q mat-mul; src/llama-kv-cache-unified.cpp:114-132). V stays F16: a quantized V cache needs flash attention
    int32_t row_split;                // -sm row over this many devices (0 / 1 = off): the 2-D weight matrices go to the backend's split buffer type, equal shares
                                      // (make_gpu_buft_list, src/llama-model.cpp:368-387); everything else, the KV cache and the graph stay on `backend`'s device
    int32_t type_v;                   // V cache type (llama-bench -ctv): 0 = F16; Q8_0 / Q4_0 / BF16 need flash attention (the V cache is then rows of cells,
                                      // src/llama-context.cpp "V cache quantization requires flash_attn")
};

struct mi_llama;

} // extern "C"

namespace {

struct rng64 {
    uint64_t s;
    explicit rng64(uint64_t seed) : s(seed*0x9E3779B97F4A7C15ull + 0xD1B54A32D192ED03ull) { next(); next(); }
    uint64_t next() { s ^= s >> 12; s ^= s << 25; s ^= s >> 27; return s*0x2545F4914F6CDD1Dull; }
    float unif() { return (float)(next() >> 40)*(1.0f/16777216.0f); }   // [0,1)
};

// random valid blocks with super-scales chosen so that dequantized weights have std ~ sigma
void fill_random_blocks(enum ggml_type type, uint8_t * dst, size_t nbytes, float sigma, uint64_t seed) {
    const size_t ts = ggml_type_size(type);
    const size_t nblk = nbytes/ts;
    const unsigned nthr = std::max(1u, std::min(16u, std::thread::hardware_concurrency()));
    std::vector<std::thread> th;
    for (unsigned t = 0; t < nthr; t++) {
        th.emplace_back([=]() {
            rng64 r(seed*1315423911ull + t);
            const size_t b0 = nblk*t/nthr, b1 = nblk*(t + 1)/nthr;
            uint64_t * p = (uint64_t *) (dst + b0*ts);
            const size_t n8 = ((b1 - b0)*ts)/8;
            for (size_t i = 0; i < n8; i++) p[i] = r.next();
            for (size_t i = b0*ts + n8*8; i < b1*ts; i++) dst[i] = (uint8_t) r.next();
            for (size_t b = b0; b < b1; b++) {
                uint8_t * blk = dst + b*ts;
                const float u = 0.75f + 0.5f*r.unif();
                switch (type) {
                    case GGML_TYPE_Q4_K: case GGML_TYPE_Q5_K: {
                        const float qmean = type == GGML_TYPE_Q4_K ? 7.5f : 15.5f;
                        const float sd = type == GGML_TYPE_Q4_K ? 258.0f : 530.0f;
                        const ggml_fp16_t d = ggml_fp32_to_fp16(u*sigma/sd), dm = ggml_fp32_to_fp16(u*sigma/sd*qmean);
                        memcpy(blk, &d, 2); memcpy(blk + 2, &dm, 2);
                    } break;
                    case GGML_TYPE_Q6_K: { const ggml_fp16_t d = ggml_fp32_to_fp16(u*sigma/1367.0f); memcpy(blk + 208, &d, 2); } break;
                    case GGML_TYPE_Q8_0: { const ggml_fp16_t d = ggml_fp32_to_fp16(u*sigma/73.9f);  memcpy(blk, &d, 2); } break;
                    case GGML_TYPE_Q4_0: { const ggml_fp16_t d = ggml_fp32_to_fp16(u*sigma/4.61f);  memcpy(blk, &d, 2); } break;
                    case GGML_TYPE_MXFP4: {
                        int e = 128 + (int) lrintf(log2f(sigma/5.85f));
                        blk[0] = (uint8_t) std::max(1, std::min(254, e));
                    } break;
                    default: break;
                }
            }
        });
    }
    for (auto & t : th) t.join();
}

void fill_f32(float * p, size_t n, float lo, float hi, uint64_t seed) {
    rng64 r(seed);
    for (size_t i = 0; i < n; i++) p[i] = lo + (hi - lo)*r.unif();
}

bool use_more_bits(int i_layer, int n_layers) {   // src/llama-quant.cpp:185-187
    return i_layer < n_layers/8 || i_layer >= 7*n_layers/8 || (i_layer - n_layers/8)%3 == 2;
}

struct layer_types { enum ggml_type wq, wk, wv, wo, gate, up, down; };

// src/llama-quant.cpp:178-434 restricted to the dense-llama tensors and the ftypes of BASELINE.json's configs
layer_types types_for_layer(const mi_llama_hparams & hp, int il) {
    layer_types t;
    enum ggml_type base;
    switch (hp.ftype) {
        case MI_FTYPE_Q4_0:   base = GGML_TYPE_Q4_0; break;
        case MI_FTYPE_Q8_0:   base = GGML_TYPE_Q8_0; break;
        case MI_FTYPE_Q6_K:   base = GGML_TYPE_Q6_K; break;
        case MI_FTYPE_Q5_K_M: base = GGML_TYPE_Q5_K; break;
        default:              base = GGML_TYPE_Q4_K; break;
    }
    t.wq = t.wk = t.wv = t.wo = t.gate = t.up = t.down = base;
    if (hp.ftype == MI_FTYPE_Q4_K_M || hp.ftype == MI_FTYPE_Q5_K_M) {
        if (use_more_bits(il, hp.n_layer)) { t.wv = GGML_TYPE_Q6_K; t.down = GGML_TYPE_Q6_K; }   // :302-303, :358-364
        if (hp.is_70b && t.wv == GGML_TYPE_Q4_K) t.wv = GGML_TYPE_Q5_K;                          // :305-310
        if (hp.n_expert == 8) {                                                                   // the 8-expert bumps, :311-322 and :383-389
            t.wv = t.wk = GGML_TYPE_Q8_0;
            if (hp.ftype == MI_FTYPE_Q4_K_M) t.wo = GGML_TYPE_Q5_K;
        }
    }
    if (hp.ftype == MI_FTYPE_MXFP4_MOE) {                                                         // :229-236: 3-D tensors MXFP4, the rest Q8_0
        t.wq = t.wk = t.wv = t.wo = GGML_TYPE_Q8_0;
        t.gate = t.up = t.down = hp.n_expert > 0 ? GGML_TYPE_MXFP4 : GGML_TYPE_Q8_0;
    }
    return t;
}
enum ggml_type output_type(const mi_llama_hparams & hp) {   // :205-227
    return hp.ftype == MI_FTYPE_Q8_0 || hp.ftype == MI_FTYPE_MXFP4_MOE ? GGML_TYPE_Q8_0 : GGML_TYPE_Q6_K;
}

struct layer {
    ggml_tensor * attn_norm, * wq, * wk, * wv, * wo, * ffn_norm, * ffn_gate, * ffn_up, * ffn_down;   // ffn_*: [n_embd, n_ff(, n_expert)]
    ggml_tensor * ffn_gate_inp = nullptr;                                                            // router, F32 [n_embd, n_expert]
    // gpt-oss only (src/llama-model.cpp:5436-5462): projection biases, attention sinks, router and expert biases
    ggml_tensor * bq = nullptr, * bk = nullptr, * bv = nullptr, * bo = nullptr, * sinks = nullptr;
    ggml_tensor * gate_inp_b = nullptr, * gate_b = nullptr, * up_b = nullptr, * down_b = nullptr;
    std::vector<ggml_tensor *> k_cache, v_cache;   // one per sequence stream
    bool swa = false;                              // attends through the window cache (its k_cache / v_cache have swa_size cells)
};

struct graph_inst {
    ggml_context * ctx = nullptr;
    ggml_backend_buffer_t buf = nullptr;
    ggml_cgraph * gf = nullptr;
    ggml_tensor * inp_embd = nullptr, * inp_pos = nullptr, * kq_mask = nullptr, * k_idxs = nullptr, * v_idxs = nullptr, * out_ids = nullptr;
    ggml_tensor * kq_mask_swa = nullptr, * k_idxs_swa = nullptr, * v_idxs_swa = nullptr;     // the window cache's own inputs (llm_graph_input_attn_kv_unified_iswa)
    ggml_tensor * result = nullptr;    // logits (has_output) or the last layer's l_out
    int64_t last_use = 0;
};

} // namespace

struct mi_llama {
    mi_llama_hparams hp;
    ggml_backend_t backend;
    ggml_context * wctx = nullptr;
    ggml_context * wsplit = nullptr;                    // row-split weights (hp.row_split > 1) and their buffer
    ggml_backend_buffer_t wsplit_buf = nullptr;
    ggml_backend_buffer_type_t split_buft = nullptr;
    ggml_backend_buffer_t wbuf = nullptr;
    ggml_context * kvctx = nullptr;
    ggml_backend_buffer_t kvbuf = nullptr;
    std::vector<layer> layers;
    ggml_tensor * output_norm = nullptr, * output = nullptr, * rope_freqs = nullptr;
    std::map<std::tuple<int, int, int, int>, graph_inst> graphs;   // (seq, n_tokens, n_kv, n_kv of the window cache) -> graph: the reuse of src/llama-context.cpp:728
    int swa_size = 0;                                   // cells of the window cache (0: the model has none)
    std::vector<std::vector<int>> swa_cell_pos;         // [seq][cell] position held by a cell of the window cache, -1 = empty
    int64_t tick = 0;
    std::vector<int> n_past;                            // cells [0, n_past[s]) of sequence s are in use
    uint64_t weight_bytes = 0;                          // bytes of every dense MUL_MAT weight held here (all of them are read per token)
    uint64_t expert_bytes = 0;                          // bytes of the MUL_MAT_ID expert stacks (n_expert_used/n_expert of them are read per token)
    uint64_t seed;
    // pinned staging for the per-step inputs (set_inputs: src/llama-graph.cpp:16-58)
    ggml_backend_buffer_t hbuf = nullptr;
    uint8_t * hbase = nullptr; size_t hsize = 0; int hslot = 0;   // 4-slot ring: a slot is reused only 4 decodes later
    // the result of the last decode (logits of the last token, or l_out on a non-final layer-split rank) in PINNED host memory from the device's host buffer
    // type, as llama_context::output_reserve allocates buf_output (src/llama-context.cpp:1260-1330): the read-back is then one asynchronous copy
    ggml_backend_buffer_t lbuf = nullptr; float * logits = nullptr; size_t logits_cap = 0, logits_n = 0;   // capacity / valid count in floats
    typedef void (*dev_set_fn)(ggml_backend_t, struct ggml_tensor *, const void *, size_t, size_t);
    typedef void (*dev_get_fn)(ggml_backend_t, const struct ggml_tensor *, void *, size_t, size_t);
    dev_set_fn set_from_device = nullptr; dev_get_fn get_to_device = nullptr;      // hand-offs through raw device buffers (layer split over RCCL)
    // a model read from a GGUF file (mi_llama_create_from_gguf): the mapping stays open for the input layer's rows
    std::unique_ptr<mi355x::gguf_file> gf;
    const uint8_t * tok_embd = nullptr; int tok_embd_type = 0; size_t tok_embd_row = 0;   // token_embd.weight in the mapping (host side: see gguf_tools.cpp)
    std::vector<float> embd_tmp;
};

namespace {

const int KV_PAD = 32;   // get_padding without flash attention (src/llama-kv-cache-unified.cpp:2407-2410)

// With a GGUF file the tensor must be in it with the expected shape (llama_model_loader::check_tensor_dims, src/llama-model-loader.cpp:797-830:
// "missing tensor" / "has wrong shape" are load errors) and takes the file's type; only types the backend's mat-mul supports are accepted.
enum ggml_type file_type(mi_llama * m, enum ggml_type type, int64_t ne0, int64_t ne1, int64_t ne2, const char * name, bool f32_only) {
    if (!m->gf) return type;
    const char * lookup = name;
    if (strcmp(name, "output.weight") == 0 && !m->gf->find(name)) lookup = "token_embd.weight";   // tied embeddings (TENSOR_DUPLICATED, src/llama-model.cpp:2209-2212)
    const mi355x::gguf_tensor_info * ti = m->gf->find(lookup);
    if (!ti) throw std::runtime_error(std::string("missing tensor '") + name + "'");
    if (ti->ne[0] != ne0 || ti->ne[1] != ne1 || ti->ne[2] != ne2 || ti->ne[3] != 1) {
        char b[256];
        snprintf(b, sizeof(b), "tensor '%s' has wrong shape; expected %lld, %lld, %lld, got %lld, %lld, %lld, %lld", name, (long long) ne0, (long long) ne1,
                 (long long) ne2, (long long) ti->ne[0], (long long) ti->ne[1], (long long) ti->ne[2], (long long) ti->ne[3]);
        throw std::runtime_error(b);
    }
    const enum ggml_type ft = (enum ggml_type) ti->type;
    const bool ok = f32_only ? ft == GGML_TYPE_F32
                             : (ft == GGML_TYPE_Q4_0 || ft == GGML_TYPE_Q8_0 || ft == GGML_TYPE_Q4_K || ft == GGML_TYPE_Q5_K || ft == GGML_TYPE_Q6_K ||
                                ft == GGML_TYPE_MXFP4 || ft == GGML_TYPE_F16 || ft == GGML_TYPE_F32);
    if (!ok) throw std::runtime_error(std::string("tensor '") + name + "' has type " + std::to_string((int) ti->type) + ", which this path does not run");
    return ft;
}

// every weight tensor: the ordinary context, then the row-split one
template <typename F> void for_each_weight(mi_llama * m, F f) {
    for (ggml_context * ctx : { m->wctx, m->wsplit }) {
        if (!ctx) continue;
        for (ggml_tensor * t = ggml_get_first_tensor(ctx); t; t = ggml_get_next_tensor(ctx, t)) f(t);
    }
}

ggml_tensor * new_weight(mi_llama * m, enum ggml_type type, int64_t ne0, int64_t ne1, const char * name) {
    type = file_type(m, type, ne0, ne1, 1, name, type == GGML_TYPE_F32);
    ggml_tensor * t = ggml_new_tensor_2d(m->split_buft && ne1 > 1 && ggml_is_quantized(type) ? m->wsplit : m->wctx, type, ne0, ne1);
    ggml_set_name(t, name);
    if (ne1 > 1) m->weight_bytes += ggml_nbytes(t);
    return t;
}

ggml_tensor * new_experts(mi_llama * m, enum ggml_type type, int64_t ne0, int64_t ne1, int64_t ne2, const char * name) {
    type = file_type(m, type, ne0, ne1, ne2, name, false);
    ggml_tensor * t = ggml_new_tensor_3d(m->wctx, type, ne0, ne1, ne2);
    ggml_set_name(t, name);
    m->expert_bytes += ggml_nbytes(t);
    return t;
}
ggml_tensor * new_f32(mi_llama * m, int64_t ne0, int64_t ne1, const char * name) {   // biases, sinks, router: not part of the byte model
    (void) file_type(m, GGML_TYPE_F32, ne0, ne1, 1, name, true);
    ggml_tensor * t = ggml_new_tensor_2d(m->wctx, GGML_TYPE_F32, ne0, ne1);
    ggml_set_name(t, name);
    return t;
}

void upload_random(mi_llama * m, ggml_tensor * t, float sigma, uint64_t seed, std::vector<uint8_t> & tmp) {
    const size_t nb = ggml_nbytes(t);
    tmp.resize(nb);
    if (t->type == GGML_TYPE_F32) {
        const char * nm = t->name;
        if (strstr(nm, "ffn_gate_inp.weight"))      fill_f32((float *) tmp.data(), nb/4, -1.7320508f*sigma, 1.7320508f*sigma, seed);   // router: std sigma
        else if (strstr(nm, ".bias"))               fill_f32((float *) tmp.data(), nb/4, -0.1f, 0.1f, seed);
        else if (strstr(nm, "attn_sinks"))          fill_f32((float *) tmp.data(), nb/4, -1.0f, 1.0f, seed);
        else                                        fill_f32((float *) tmp.data(), nb/4, 0.5f, 1.5f, seed);                              // norm weights ~ 1
    } else {
        fill_random_blocks(t->type, tmp.data(), nb, sigma, seed);
    }
    ggml_backend_tensor_set(t, tmp.data(), 0, nb);
}

// build_moe_ffn (src/llama-graph.cpp:811-1023) for the two gatings the configs use: SOFTMAX + norm_w (llm_build_llama's MoE branch,
// src/llama-model.cpp:6082-6092) and SOFTMAX_WEIGHT + biases + SWIGLU_OAI (gpt-oss, src/llama-model.cpp:17700-17711)
ggml_tensor * build_moe_ffn(ggml_context * ctx0, ggml_cgraph * gf, const mi_llama_hparams & hp, const layer & L, ggml_tensor * cur, ggml_tensor ** experts_out = nullptr) {
    const int64_t n_embd = cur->ne[0], n_tokens = cur->ne[1], n_expert = hp.n_expert, n_used = hp.n_expert_used;
    const bool oai = hp.arch == 1;
    ggml_tensor * logits = ggml_mul_mat(ctx0, L.ffn_gate_inp, cur);                             // [n_expert, n_tokens]   :838
    if (L.gate_inp_b) logits = ggml_add(ctx0, logits, L.gate_inp_b);                            // :845
    ggml_tensor * probs = oai ? logits : ggml_soft_max(ctx0, logits);                           // :850-866
    ggml_tensor * selected = ggml_top_k(ctx0, probs, (int) n_used);                             // [n_used, n_tokens] I32   :883
    ggml_tensor * weights = ggml_get_rows(ctx0, ggml_reshape_3d(ctx0, probs, 1, n_expert, n_tokens), selected);   // [1, n_used, n_tokens]   :887
    if (oai) {                                                                                  // :891-896
        weights = ggml_reshape_2d(ctx0, weights, n_used, n_tokens);
        weights = ggml_soft_max(ctx0, weights);
        weights = ggml_reshape_3d(ctx0, weights, 1, n_used, n_tokens);
    } else {                                                                                    // norm_w, :898-908
        weights = ggml_reshape_2d(ctx0, weights, n_used, n_tokens);
        ggml_tensor * wsum = ggml_sum_rows(ctx0, weights);
        weights = ggml_div(ctx0, weights, wsum);
        weights = ggml_reshape_3d(ctx0, weights, 1, n_used, n_tokens);
    }
    cur = ggml_reshape_3d(ctx0, cur, n_embd, 1, n_tokens);                                      // :914
    ggml_tensor * up = ggml_mul_mat_id(ctx0, L.ffn_up, cur, selected);                          // [n_ff, n_used, n_tokens]   :923
    if (L.up_b) up = ggml_add_id(ctx0, up, L.up_b, selected);                                   // :927
    ggml_tensor * gate = ggml_mul_mat_id(ctx0, L.ffn_gate, cur, selected);                      // :933
    if (L.gate_b) gate = ggml_add_id(ctx0, gate, L.gate_b, selected);                           // :940
    ggml_tensor * act = oai ? ggml_swiglu_oai(ctx0, gate, up, 1.702f, 7.0f)                     // :961-968
                            : ggml_swiglu_split(ctx0, gate, up);                                // :947
    ggml_tensor * experts = ggml_mul_mat_id(ctx0, L.ffn_down, act, selected);                   // [n_embd, n_used, n_tokens]   :981
    if (L.down_b) experts = ggml_add_id(ctx0, experts, L.down_b, selected);                     // :985
    if (experts_out) *experts_out = experts;      // (what the combine reads: dead after it — see MI_HARNESS_ALIAS_MOE in build_graph)
    experts = ggml_mul(ctx0, experts, weights);                                                 // :990
    ggml_tensor * moe_out = nullptr;                                                            // :996-1012: views ordered before the adds
    std::vector<ggml_tensor *> views;
    for (int64_t i = 0; i < n_used; i++) {
        views.push_back(ggml_view_2d(ctx0, experts, n_embd, n_tokens, experts->nb[2], i*experts->nb[1]));
        ggml_build_forward_expand(gf, views.back());
    }
    moe_out = views[0];
    for (int64_t i = 1; i < n_used; i++) moe_out = ggml_add(ctx0, moe_out, views[i]);
    if (n_used == 1) moe_out = ggml_cont(ctx0, moe_out);                                        // :1014-1017
    return moe_out;
}

// llm_build_llama for n_tokens tokens attending to n_kv cache cells
graph_inst build_graph(mi_llama * m, int seq, int n_tokens, int n_kv, int n_kv_swa) {
    const mi_llama_hparams & hp = m->hp;
    graph_inst g;
    g.ctx = ggml_init({ 0, NULL, true });
    ggml_context * ctx0 = g.ctx;
    g.gf = ggml_new_graph_custom(ctx0, 8192, false);

    const int64_t n_embd = hp.n_embd, hd = hp.n_embd_head, n_head = hp.n_head, n_head_kv = hp.n_head_kv;
    const int64_t n_embd_k_gqa = hd*n_head_kv, n_embd_v_gqa = hd*n_head_kv;
    const float kq_scale = 1.0f/sqrtf((float) hd);
    const bool last_rank = hp.has_output != 0;

    // inputs. build_inp_embd: the token-embedding lookup runs on the CPU (input layer, src/llama-model.cpp:1963);
    // the backend receives its F32 result. On a layer-split rank > 0 this is the activation handed over by the previous rank.
    g.inp_embd = ggml_new_tensor_2d(ctx0, GGML_TYPE_F32, n_embd, n_tokens); ggml_set_input(g.inp_embd); ggml_set_name(g.inp_embd, "inp_embd");
    g.inp_pos  = ggml_new_tensor_1d(ctx0, GGML_TYPE_I32, n_tokens);         ggml_set_input(g.inp_pos);  ggml_set_name(g.inp_pos, "inp_pos");
    g.kq_mask  = ggml_new_tensor_2d(ctx0, GGML_TYPE_F32, n_kv, GGML_PAD(n_tokens, GGML_KQ_MASK_PAD)); ggml_set_input(g.kq_mask);   // src/llama-graph.cpp:1421
    g.k_idxs   = ggml_new_tensor_1d(ctx0, GGML_TYPE_I64, n_tokens);                 ggml_set_input(g.k_idxs);   // :1195
    g.v_idxs   = ggml_new_tensor_1d(ctx0, GGML_TYPE_I64, hp.flash_attn ? n_tokens : n_tokens*n_embd_v_gqa); ggml_set_input(g.v_idxs);   // :1208 (v_trans: per element)
    // with flash attention the mask is cast to F16 once per graph (src/llama-graph.cpp:1423)
    ggml_tensor * kq_mask_f16 = hp.flash_attn ? ggml_cast(ctx0, g.kq_mask, GGML_TYPE_F16) : nullptr;
    ggml_tensor * kq_mask_swa_f16 = nullptr;
    if (m->swa_size > 0) {     // build_attn_inp_kv_unified_iswa (src/llama-graph.cpp:1568-1612): a second set of inputs for the window cache
        g.kq_mask_swa = ggml_new_tensor_2d(ctx0, GGML_TYPE_F32, n_kv_swa, GGML_PAD(n_tokens, GGML_KQ_MASK_PAD)); ggml_set_input(g.kq_mask_swa);
        g.k_idxs_swa  = ggml_new_tensor_1d(ctx0, GGML_TYPE_I64, n_tokens);                                         ggml_set_input(g.k_idxs_swa);
        g.v_idxs_swa  = ggml_new_tensor_1d(ctx0, GGML_TYPE_I64, hp.flash_attn ? n_tokens : n_tokens*n_embd_v_gqa); ggml_set_input(g.v_idxs_swa);
        if (hp.flash_attn) kq_mask_swa_f16 = ggml_cast(ctx0, g.kq_mask_swa, GGML_TYPE_F16);
    }
    const int n_outputs = 1;                                                       // llama_batch_get_one: logits for the last token only
    g.out_ids  = ggml_new_tensor_1d(ctx0, GGML_TYPE_I32, n_outputs);                ggml_set_input(g.out_ids);

    ggml_tensor * inpL = g.inp_embd;
    ggml_tensor * cur = nullptr;
    std::vector<ggml_tensor *> dbg_q, dbg_experts;      // per layer: the rotated Q tensor; the experts' outputs the MoE combine reads (MI_HARNESS_ALIAS_MOE)
    const int n_local = (int) m->layers.size();
    // MI_HARNESS_TAP="<layer>:<point>" (tools/fmt_first_step.py): the graph is built and computed as always, but the tensor read back as the "result" is an
    // intermediate one — attn_norm, v, attn, wo, ffn_norm, glu, down of that layer, or result_norm. With fusions on, a tensor a fused launch never writes reads as stale memory.
    ggml_tensor * tapped = nullptr;
    const char * tap_env = getenv("MI_HARNESS_TAP");
    auto tap = [&](int li, const char * point, ggml_tensor * t) {
        if (!tap_env) return;
        char want[64]; snprintf(want, sizeof(want), "%d:%s", li, point);
        if (strcmp(want, tap_env) == 0) { tapped = t; ggml_set_output(t); }
    };
    for (int li = 0; li < n_local; li++) {
        const layer & L = m->layers[li];
        const bool last_layer = last_rank && li == n_local - 1;
        ggml_tensor * inpSA = inpL;
        // which cache / mask / indices this layer attends through (build_attn(inp_attn_kv_unified_iswa), src/llama-graph.cpp:1490-1560)
        const int64_t kv_size = L.swa ? m->swa_size : hp.n_ctx;
        const int64_t n_kv_l = L.swa ? n_kv_swa : n_kv;
        ggml_tensor * l_mask = L.swa ? g.kq_mask_swa : g.kq_mask, * l_mask_f16 = L.swa ? kq_mask_swa_f16 : kq_mask_f16;
        ggml_tensor * l_k_idxs = L.swa ? g.k_idxs_swa : g.k_idxs, * l_v_idxs = L.swa ? g.v_idxs_swa : g.v_idxs;

        // build_norm(inpL, attn_norm, NULL, LLM_NORM_RMS)
        cur = ggml_rms_norm(ctx0, inpL, hp.f_norm_rms_eps);
        cur = ggml_mul(ctx0, cur, L.attn_norm);

        tap(li, "attn_norm", cur);
        // self-attention
        ggml_tensor * Qcur = ggml_mul_mat(ctx0, L.wq, cur);
        if (L.bq) Qcur = ggml_add(ctx0, Qcur, L.bq);                  // src/llama-model.cpp:17636-17639
        ggml_tensor * Kcur = ggml_mul_mat(ctx0, L.wk, cur);
        if (L.bk) Kcur = ggml_add(ctx0, Kcur, L.bk);
        ggml_tensor * Vcur = ggml_mul_mat(ctx0, L.wv, cur);
        if (L.bv) Vcur = ggml_add(ctx0, Vcur, L.bv);
        tap(li, "v", Vcur);
        Qcur = ggml_reshape_3d(ctx0, Qcur, hd, n_head,    n_tokens);
        Kcur = ggml_reshape_3d(ctx0, Kcur, hd, n_head_kv, n_tokens);
        Vcur = ggml_reshape_3d(ctx0, Vcur, hd, n_head_kv, n_tokens);
        Qcur = ggml_rope_ext(ctx0, Qcur, g.inp_pos, m->rope_freqs, (int) hd, hp.rope_type, hp.n_ctx_orig, hp.rope_freq_base, hp.rope_freq_scale, 0.0f, 1.0f, 32.0f, 1.0f);
        Kcur = ggml_rope_ext(ctx0, Kcur, g.inp_pos, m->rope_freqs, (int) hd, hp.rope_type, hp.n_ctx_orig, hp.rope_freq_base, hp.rope_freq_scale, 0.0f, 1.0f, 32.0f, 1.0f);

        dbg_q.push_back(Qcur);
        // build_attn: q/k/v first so that they are not reordered (src/llama-graph.cpp:1449-1453)
        ggml_build_forward_expand(g.gf, Qcur);
        ggml_build_forward_expand(g.gf, Kcur);
        ggml_build_forward_expand(g.gf, Vcur);
        {   // store to KV cache: cpy_k / cpy_v with set_rows (src/llama-kv-cache-unified.cpp:1108-1190)
            ggml_tensor * k_cur2 = ggml_reshape_2d(ctx0, Kcur, n_embd_k_gqa, n_tokens);
            ggml_build_forward_expand(g.gf, ggml_set_rows(ctx0, L.k_cache[seq], k_cur2, l_k_idxs));
            ggml_tensor * v_cur2 = ggml_reshape_2d(ctx0, Vcur, n_embd_v_gqa, n_tokens);
            if (hp.flash_attn) {      // !v_trans: a row scatter like K (:1154)
                ggml_build_forward_expand(g.gf, ggml_set_rows(ctx0, L.v_cache[seq], v_cur2, l_v_idxs));
            } else {
                ggml_tensor * v_view = ggml_reshape_2d(ctx0, L.v_cache[seq], 1, n_embd_v_gqa*kv_size);     // the row becomes a single element
                v_cur2 = ggml_reshape_2d(ctx0, v_cur2, 1, n_embd_v_gqa*n_tokens);
                ggml_build_forward_expand(g.gf, ggml_set_rows(ctx0, v_view, v_cur2, l_v_idxs));
            }
        }
        // get_k / get_v (:1056-1106), v_trans layout
        const enum ggml_type tk = L.k_cache[seq]->type;
        ggml_tensor * k = ggml_view_4d(ctx0, L.k_cache[seq], hd, n_head_kv, n_kv_l, 1,
                ggml_row_size(tk, hd), ggml_row_size(tk, n_embd_k_gqa), ggml_row_size(tk, n_embd_k_gqa*kv_size), 0);
        const enum ggml_type tv = L.v_cache[seq]->type;
        ggml_tensor * v = hp.flash_attn
            ? ggml_view_4d(ctx0, L.v_cache[seq], hd, n_head_kv, n_kv_l, 1,      // !v_trans (:1087-1096)
                ggml_row_size(tv, hd), ggml_row_size(tv, n_embd_v_gqa), ggml_row_size(tv, n_embd_v_gqa*kv_size), 0)
            : ggml_view_4d(ctx0, L.v_cache[seq], n_kv_l, n_head_kv, hd, 1,
                ggml_row_size(GGML_TYPE_F16, kv_size*hd), ggml_row_size(GGML_TYPE_F16, kv_size), ggml_row_size(GGML_TYPE_F16, kv_size*n_embd_v_gqa), 0);
        if (hp.flash_attn) {   // build_attn_mha with flash attention (src/llama-graph.cpp:1245-1265, :1337): n_kv % 256 == 0 by the cache's padding
            ggml_tensor * q = ggml_reshape_4d(ctx0, Qcur, Qcur->ne[0], Qcur->ne[1], Qcur->ne[2], 1);
            q = ggml_permute(ctx0, q, 0, 2, 1, 3);
            k = ggml_permute(ctx0, k, 0, 2, 1, 3);
            v = ggml_permute(ctx0, v, 0, 2, 1, 3);
            cur = ggml_flash_attn_ext(ctx0, q, k, v, l_mask_f16, kq_scale, 0.0f, 0.0f);
            ggml_flash_attn_ext_add_sinks(cur, L.sinks);
            ggml_flash_attn_ext_set_prec(cur, GGML_PREC_F32);
            cur = ggml_reshape_2d(ctx0, cur, cur->ne[0]*cur->ne[1], cur->ne[2]*cur->ne[3]);
            ggml_build_forward_expand(g.gf, cur);
        } else {   // build_attn_mha, no flash attention (src/llama-graph.cpp:1283-1341)
            ggml_tensor * q = ggml_reshape_4d(ctx0, Qcur, Qcur->ne[0], Qcur->ne[1], Qcur->ne[2], 1);
            q = ggml_permute(ctx0, q, 0, 2, 1, 3);
            k = ggml_permute(ctx0, k, 0, 2, 1, 3);
            v = ggml_permute(ctx0, v, 0, 2, 1, 3);
            ggml_tensor * kq = ggml_mul_mat(ctx0, k, q);
            ggml_mul_mat_set_prec(kq, GGML_PREC_F32);
            kq = ggml_soft_max_ext(ctx0, kq, l_mask, kq_scale, 0.0f);
            if (L.sinks) ggml_soft_max_add_sinks(kq, L.sinks);       // build_attn_with_sinks, src/llama-graph.cpp:1313
            ggml_tensor * kqv = ggml_mul_mat(ctx0, v, kq);
            cur = ggml_permute(ctx0, kqv, 0, 2, 1, 3);
            cur = ggml_cont_2d(ctx0, cur, cur->ne[0]*cur->ne[1], cur->ne[2]*cur->ne[3]);
            ggml_build_forward_expand(g.gf, cur);
        }
        tap(li, "attn", cur);
        cur = ggml_mul_mat(ctx0, L.wo, cur);
        tap(li, "wo", cur);
        if (L.bo) cur = ggml_add(ctx0, cur, L.bo);                    // src/llama-graph.cpp:1479-1481

        if (last_layer) {   // src/llama-model.cpp:6052-6055
            cur   = ggml_get_rows(ctx0, cur,   g.out_ids);
            inpSA = ggml_get_rows(ctx0, inpSA, g.out_ids);
        }
        ggml_tensor * ffn_inp = ggml_add(ctx0, cur, inpSA);

        // feed-forward: build_norm + build_ffn(LLM_FFN_SILU, LLM_FFN_PAR)
        cur = ggml_rms_norm(ctx0, ffn_inp, hp.f_norm_rms_eps);
        cur = ggml_mul(ctx0, cur, L.ffn_norm);
        tap(li, "ffn_norm", cur);
        if (hp.n_expert > 0) {
            { ggml_tensor * ex = nullptr; cur = build_moe_ffn(ctx0, g.gf, hp, L, cur, &ex); dbg_experts.push_back(ex); }              // src/llama-model.cpp:6075-6093 / :17700-17711
        } else {
            ggml_tensor * tmp = ggml_mul_mat(ctx0, L.ffn_up, cur);
            cur = ggml_mul_mat(ctx0, L.ffn_gate, cur);
            cur = ggml_swiglu_split(ctx0, cur, tmp);
            tap(li, "glu", cur);
            cur = ggml_mul_mat(ctx0, L.ffn_down, cur);
            tap(li, "down", cur);
        }
        cur = ggml_add(ctx0, cur, ffn_inp);
        inpL = cur;
    }
    cur = inpL;
    if (last_rank) {
        cur = ggml_rms_norm(ctx0, cur, hp.f_norm_rms_eps);
        cur = ggml_mul(ctx0, cur, m->output_norm);
        ggml_set_name(cur, "result_norm");            // res->t_embd (src/llama-model.cpp:6113): read back by llama_context for embeddings, no OUTPUT flag
        tap(n_local - 1, "result_norm", cur);
        cur = ggml_mul_mat(ctx0, m->output, cur);     // lm_head
    }
    ggml_set_output(cur);
    g.result = tapped ? tapped : cur;
    ggml_build_forward_expand(g.gf, cur);

    g.buf = ggml_backend_alloc_ctx_tensors(ctx0, m->backend);
    if (!g.buf) { fprintf(stderr, "mi_llama: compute buffer allocation failed\n"); abort(); }
    // MI_HARNESS_ALIAS_MOE=1 (tests): this harness gives every tensor memory of its own, ggml-alloc does not — it hands a dead tensor's memory to the next one that
    // fits. Reproduce the case ADVICE r3 describes: layer i's rotated Q lives where layer i - 1's expert outputs were (dead after that layer's combine), so a launch
    // that still READS those outputs while it writes Q races unless the backend notices the overlap
    if (getenv("MI_HARNESS_ALIAS_MOE") && atoi(getenv("MI_HARNESS_ALIAS_MOE")) != 0) {
        for (size_t li = 1; li < dbg_q.size() && li - 1 < dbg_experts.size(); li++) {
            ggml_tensor * q = dbg_q[li]; ggml_tensor * ex = dbg_experts[li - 1];
            if (!q || !ex || q->view_src || ggml_nbytes(q) > ggml_nbytes(ex)) continue;
            q->data = ex->data;
            for (int i = 0; i < ggml_graph_n_nodes(g.gf); i++) { ggml_tensor * t = ggml_graph_node(g.gf, i); if (t->view_src == q) t->data = (char *) q->data + t->view_offs; }
        }
    }
    return g;
}

void free_graph(graph_inst & g) {
    if (g.buf) ggml_backend_buffer_free(g.buf);
    if (g.ctx) ggml_free(g.ctx);
    g = graph_inst();
}

// room for n floats of results in pinned host memory (grows; the old contents are not kept)
bool reserve_result(mi_llama * m, size_t n) {
    if (n <= m->logits_cap) return true;
    ggml_backend_synchronize(m->backend);
    if (m->lbuf) { ggml_backend_buffer_free(m->lbuf); m->lbuf = nullptr; } else free(m->logits);
    m->logits = nullptr; m->logits_cap = 0;
    ggml_backend_buffer_type_t hbt = ggml_backend_dev_host_buffer_type(ggml_backend_get_device(m->backend));
    if (hbt) { m->lbuf = ggml_backend_buft_alloc_buffer(hbt, n*4); if (m->lbuf) m->logits = (float *) ggml_backend_buffer_get_base(m->lbuf); }
    if (!m->logits) m->logits = (float *) malloc(n*4);
    if (!m->logits) return false;
    m->logits_cap = n;
    return true;
}

// deterministic synthetic embedding row for a token id (stands in for get_rows(tok_embd) on the CPU)
void synth_embedding(float * dst, int n_embd, int32_t token, uint64_t seed) {
    rng64 r(seed ^ (0x5851F42D4C957F2Dull*(uint64_t)(token + 1)));
    for (int i = 0; i < n_embd; i++) dst[i] = 2.0f*r.unif() - 1.0f;
}

} // namespace

extern "C" {

} // extern "C"

namespace {

// Weights from the file into the device tensors through a ring of pinned staging buffers with one event each, as the loader does
// (src/llama-model-loader.cpp:930-1010: 4 x 1 MiB; 4 x 4 MiB here — a chunk is one PCIe transfer and one event, and the mapping is pageable
// memory the copy engine cannot read directly): wait for the slot's event, memcpy file -> slot, tensor_set_async, record the event.
void upload_from_file(mi_llama * m) {
    const size_t CHUNK = 4u << 20; const int NBUF = 4;
    ggml_backend_dev_t dev = ggml_backend_get_device(m->backend);
    ggml_backend_buffer_type_t hbt = ggml_backend_dev_host_buffer_type(dev);
    ggml_backend_buffer_t ring = hbt ? ggml_backend_buft_alloc_buffer(hbt, CHUNK*NBUF) : nullptr;
    std::vector<ggml_backend_event_t> ev;
    if (ring) for (int i = 0; i < NBUF; i++) { ggml_backend_event_t e = ggml_backend_event_new(dev); if (!e) break; ev.push_back(e); }
    const bool async = ring && (int) ev.size() == NBUF;
    uint8_t * rbase = async ? (uint8_t *) ggml_backend_buffer_get_base(ring) : nullptr;
    int slot = 0;
    for_each_weight(m, [&](ggml_tensor * t) {
        const mi355x::gguf_tensor_info * ti = m->gf->find(t->name);
        if (!ti && strcmp(t->name, "output.weight") == 0) ti = m->gf->find("token_embd.weight");
        const size_t nb = ggml_nbytes(t);
        const uint8_t * src = m->gf->tensor_data(*ti, nb);
        // the loader's fallback: a synchronous copy per tensor (:1075); row-split tensors are written whole (their buffer spreads the rows)
        if (!async || t->buffer == m->wsplit_buf) { ggml_backend_tensor_set(t, src, 0, nb); return; }
        for (size_t off = 0; off < nb; off += CHUNK) {
            const size_t n = std::min(CHUNK, nb - off);
            ggml_backend_event_synchronize(ev[slot]);
            memcpy(rbase + slot*CHUNK, src + off, n);
            ggml_backend_tensor_set_async(m->backend, t, rbase + slot*CHUNK, off, n);
            ggml_backend_event_record(ev[slot], m->backend);
            slot = (slot + 1) % NBUF;
        }
    });
    for (ggml_backend_event_t e : ev) { ggml_backend_event_synchronize(e); ggml_backend_event_free(e); }
    if (ring) ggml_backend_buffer_free(ring);
}

mi_llama * create_body(mi_llama * m);

mi_llama * create_impl(ggml_backend_t backend, const mi_llama_hparams * hp_in, uint64_t seed, std::unique_ptr<mi355x::gguf_file> gf) {
    mi_llama * m = new mi_llama;
    m->hp = *hp_in; m->backend = backend; m->seed = seed; m->gf = std::move(gf);
    try {
        return create_body(m);
    } catch (...) {      // a tensor the file lacks or holds with another shape: release what was declared so far, the caller reports the message
        if (m->wbuf) ggml_backend_buffer_free(m->wbuf);
        if (m->kvbuf) ggml_backend_buffer_free(m->kvbuf);
        if (m->wsplit_buf) ggml_backend_buffer_free(m->wsplit_buf);
        if (m->wsplit) ggml_free(m->wsplit);
        if (m->wctx) ggml_free(m->wctx);
        if (m->kvctx) ggml_free(m->kvctx);
        delete m;
        throw;
    }
}

mi_llama * create_body(mi_llama * m) {
    ggml_backend_t backend = m->backend; const uint64_t seed = m->seed;
    const mi_llama_hparams * hp_in = &m->hp;
    m->n_past.assign(std::max(1, hp_in->n_seq_max), 0);
    const mi_llama_hparams & hp = m->hp;
    const int64_t n_embd = hp.n_embd, hd = hp.n_embd_head, n_ff = hp.n_ff;
    const int64_t n_embd_k_gqa = hd*hp.n_head_kv, n_embd_v_gqa = hd*hp.n_head_kv;
    const int kv_size = hp.n_ctx;
    if (hp.type_v && hp.type_v != GGML_TYPE_F16 && !hp.flash_attn) throw std::runtime_error("V cache quantization requires flash_attn");      // as llama_init_from_model says
    if (kv_size % (hp.flash_attn ? 256 : KV_PAD) != 0) { fprintf(stderr, "mi_llama: n_ctx must be a multiple of %d\n", hp.flash_attn ? 256 : KV_PAD); throw std::runtime_error("n_ctx is not a multiple of the KV padding"); }
    if (hp.n_swa > 0) {    // llama_kv_cache_unified_iswa (src/llama-kv-cache-unified-iswa.cpp:46-60): size_swa = min(size_base, PAD(n_swa*n_seq + n_ubatch, n_pad)); one stream per sequence here
        const int pad = hp.flash_attn ? 256 : KV_PAD;
        m->swa_size = std::min(kv_size, (int) GGML_PAD(hp.n_swa + std::max(1, hp.n_ubatch), pad));
        m->swa_cell_pos.assign(std::max(1, hp.n_seq_max), std::vector<int>(m->swa_size, -1));
    }

    m->wctx = ggml_init({ 0, NULL, true });
    m->kvctx = ggml_init({ 0, NULL, true });
    if (hp.row_split > 1) {     // make_gpu_buft_list (src/llama-model.cpp:368-387): the split buffer type through the registry's proc, if the backend has one
        ggml_backend_dev_t dev = ggml_backend_get_device(backend);
        ggml_backend_reg_t reg = ggml_backend_dev_backend_reg(dev);
        auto fn = (ggml_backend_split_buffer_type_t) ggml_backend_reg_get_proc_address(reg, "ggml_backend_split_buffer_type");
        size_t idx = 0, ndev = ggml_backend_reg_dev_count(reg);
        while (idx < ndev && ggml_backend_reg_dev_get(reg, idx) != dev) idx++;
        if (!fn || idx == ndev || (size_t) hp.row_split > ndev) throw std::runtime_error("row split over " + std::to_string(hp.row_split) + " devices: the backend has " + std::to_string(ndev));
        float ts[128] = {};
        for (int i = 0; i < hp.row_split; i++) ts[i] = 1.0f;
        m->split_buft = fn((int) idx, ts);
        if (!m->split_buft) throw std::runtime_error("the backend refused the row split (no peer mapping between the devices)");
        m->wsplit = ggml_init({ 0, NULL, true });
    }
    char name[64];
    for (int il = hp.layer_begin; il < hp.layer_end; il++) {
        const layer_types t = types_for_layer(hp, il);
        layer L;
        snprintf(name, sizeof(name), "blk.%d.attn_norm.weight", il);   L.attn_norm = new_weight(m, GGML_TYPE_F32, n_embd, 1, name);
        snprintf(name, sizeof(name), "blk.%d.attn_q.weight", il);      L.wq = new_weight(m, t.wq, n_embd, hd*hp.n_head, name);
        snprintf(name, sizeof(name), "blk.%d.attn_k.weight", il);      L.wk = new_weight(m, t.wk, n_embd, n_embd_k_gqa, name);
        snprintf(name, sizeof(name), "blk.%d.attn_v.weight", il);      L.wv = new_weight(m, t.wv, n_embd, n_embd_v_gqa, name);
        snprintf(name, sizeof(name), "blk.%d.attn_output.weight", il); L.wo = new_weight(m, t.wo, hd*hp.n_head, n_embd, name);
        snprintf(name, sizeof(name), "blk.%d.ffn_norm.weight", il);    L.ffn_norm = new_weight(m, GGML_TYPE_F32, n_embd, 1, name);
        if (hp.n_expert > 0) {      // src/llama-model.cpp:2230-2247 (llama MoE), :5450-5462 (gpt-oss)
            snprintf(name, sizeof(name), "blk.%d.ffn_gate_inp.weight", il);  L.ffn_gate_inp = new_f32(m, n_embd, hp.n_expert, name);
            snprintf(name, sizeof(name), "blk.%d.ffn_gate_exps.weight", il); L.ffn_gate = new_experts(m, t.gate, n_embd, n_ff, hp.n_expert, name);
            snprintf(name, sizeof(name), "blk.%d.ffn_up_exps.weight", il);   L.ffn_up = new_experts(m, t.up, n_embd, n_ff, hp.n_expert, name);
            snprintf(name, sizeof(name), "blk.%d.ffn_down_exps.weight", il); L.ffn_down = new_experts(m, t.down, n_ff, n_embd, hp.n_expert, name);
        } else {
            snprintf(name, sizeof(name), "blk.%d.ffn_gate.weight", il);    L.ffn_gate = new_weight(m, t.gate, n_embd, n_ff, name);
            snprintf(name, sizeof(name), "blk.%d.ffn_up.weight", il);      L.ffn_up = new_weight(m, t.up, n_embd, n_ff, name);
            snprintf(name, sizeof(name), "blk.%d.ffn_down.weight", il);    L.ffn_down = new_weight(m, t.down, n_ff, n_embd, name);
        }
        if (hp.arch == 1) {         // gpt-oss: src/llama-model.cpp:5436-5462
            snprintf(name, sizeof(name), "blk.%d.attn_q.bias", il);          L.bq = new_f32(m, hd*hp.n_head, 1, name);
            snprintf(name, sizeof(name), "blk.%d.attn_k.bias", il);          L.bk = new_f32(m, n_embd_k_gqa, 1, name);
            snprintf(name, sizeof(name), "blk.%d.attn_v.bias", il);          L.bv = new_f32(m, n_embd_v_gqa, 1, name);
            snprintf(name, sizeof(name), "blk.%d.attn_output.bias", il);     L.bo = new_f32(m, n_embd, 1, name);
            snprintf(name, sizeof(name), "blk.%d.attn_sinks.weight", il);    L.sinks = new_f32(m, hp.n_head, 1, name);
            if (hp.n_expert > 0) {
                snprintf(name, sizeof(name), "blk.%d.ffn_gate_inp.bias", il);    L.gate_inp_b = new_f32(m, hp.n_expert, 1, name);
                snprintf(name, sizeof(name), "blk.%d.ffn_gate_exps.bias", il);   L.gate_b = new_f32(m, n_ff, hp.n_expert, name);
                snprintf(name, sizeof(name), "blk.%d.ffn_up_exps.bias", il);     L.up_b = new_f32(m, n_ff, hp.n_expert, name);
                snprintf(name, sizeof(name), "blk.%d.ffn_down_exps.bias", il);   L.down_b = new_f32(m, n_embd, hp.n_expert, name);
            }
        }
        // KV cache on the layer's device, F16 (src/llama-kv-cache-unified.cpp:114-132)
        L.swa = hp.n_swa > 0 && (hp.swa_pattern <= 0 || il % hp.swa_pattern < hp.swa_pattern - 1);     // llama_hparams::set_swa_pattern
        for (int sq = 0; sq < std::max(1, hp.n_seq_max); sq++) {
            L.k_cache.push_back(ggml_new_tensor_2d(m->kvctx, hp.type_k ? (enum ggml_type) hp.type_k : GGML_TYPE_F16, n_embd_k_gqa, L.swa ? m->swa_size : kv_size));
            L.v_cache.push_back(ggml_new_tensor_2d(m->kvctx, hp.type_v ? (enum ggml_type) hp.type_v : GGML_TYPE_F16, n_embd_v_gqa, L.swa ? m->swa_size : kv_size));
        }
        m->layers.push_back(L);
    }
    if (hp.has_rope_freqs) m->rope_freqs = new_weight(m, GGML_TYPE_F32, hd/2, 1, "rope_freqs.weight");
    if (hp.has_output) {
        m->output_norm = new_weight(m, GGML_TYPE_F32, n_embd, 1, "output_norm.weight");
        m->output = new_weight(m, output_type(hp), n_embd, hp.n_vocab, "output.weight");
    }
    m->wbuf = ggml_backend_alloc_ctx_tensors(m->wctx, backend);
    if (m->wsplit) {
        m->wsplit_buf = ggml_backend_alloc_ctx_tensors_from_buft(m->wsplit, m->split_buft);
        if (!m->wsplit_buf) throw std::runtime_error("row-split weight allocation failed");
        ggml_backend_buffer_set_usage(m->wsplit_buf, GGML_BACKEND_BUFFER_USAGE_WEIGHTS);
    }
    m->kvbuf = ggml_backend_alloc_ctx_tensors(m->kvctx, backend);
    if (!m->wbuf || !m->kvbuf) { fprintf(stderr, "mi_llama: weight/KV allocation failed\n"); throw std::runtime_error("weight / KV allocation failed"); }
    ggml_backend_buffer_set_usage(m->wbuf, GGML_BACKEND_BUFFER_USAGE_WEIGHTS);   // src/llama-model.cpp:5633
    ggml_backend_buffer_clear(m->kvbuf, 0);                                      // src/llama-kv-cache-unified.cpp:175-182

    std::vector<uint8_t> tmp;
    uint64_t s = seed*7919 + 13;
    if (m->gf) upload_from_file(m);
    else for_each_weight(m, [&](ggml_tensor * t) {
        if (t == m->rope_freqs) {
            std::vector<float> ff(t->ne[0]);
            for (size_t i = 0; i < ff.size(); i++) ff[i] = i < ff.size()/2 ? 1.0f : 8.0f;   // llama-3.1-like long/short factors
            ggml_backend_tensor_set(t, ff.data(), 0, ggml_nbytes(t));
            return;
        }
        // seeded by the tensor's NAME: a rank that holds only some layers (-sm layer) generates the same bytes for them as a whole model does
        uint64_t h = 1469598103934665603ull;
        for (const char * c = ggml_get_name(t); *c; c++) h = (h ^ (uint8_t) *c)*1099511628211ull;
        upload_random(m, t, 1.0f/sqrtf((float) t->ne[0]), s ^ h, tmp);
    });

    // pinned staging for per-step inputs
    ggml_backend_buffer_type_t hbt = ggml_backend_dev_host_buffer_type(ggml_backend_get_device(backend));
    {   // 4 slots, each large enough for the inputs of one decode of up to n_ctx tokens: embeddings, mask, K / V indices, positions
        const size_t nc = (size_t) hp.n_ctx;
        const size_t slot = nc*hp.n_embd*4 + (hp.n_swa > 0 ? 2 : 1)*(nc*GGML_PAD(nc, GGML_KQ_MASK_PAD)*4 + nc*(size_t) n_embd_v_gqa*8 + nc*16) + (64u << 10);
        m->hsize = 4*((slot + 4095) & ~(size_t) 4095);
    }
    if (hbt) {
        m->hbuf = ggml_backend_buft_alloc_buffer(hbt, m->hsize);
        if (m->hbuf) m->hbase = (uint8_t *) ggml_backend_buffer_get_base(m->hbuf);
    }
    if (!m->hbase) m->hbase = (uint8_t *) malloc(m->hsize);
    {
        ggml_backend_reg_t reg = ggml_backend_dev_backend_reg(ggml_backend_get_device(backend));
        m->set_from_device = (mi_llama::dev_set_fn) ggml_backend_reg_get_proc_address(reg, "ggml_backend_mi355x_tensor_set_from_device_async");
        m->get_to_device = (mi_llama::dev_get_fn) ggml_backend_reg_get_proc_address(reg, "ggml_backend_mi355x_tensor_get_to_device_async");
    }
    m->logits_n = (size_t)(hp.has_output ? hp.n_vocab : hp.n_embd);
    if (!reserve_result(m, m->logits_n)) { fprintf(stderr, "mi_llama: cannot allocate the output buffer\n"); abort(); }
    return m;
}

} // namespace

extern "C" {

GGML_API struct mi_llama * mi_llama_create(ggml_backend_t backend, const struct mi_llama_hparams * hp_in, uint64_t seed) {
    try { return create_impl(backend, hp_in, seed, nullptr); } catch (const std::exception & e) { fprintf(stderr, "mi_llama_create: %s\n", e.what()); return nullptr; }
}

// A model from a GGUF file: hyper-parameters from its metadata as llama_model::load_hparams reads them (src/llama-model.cpp:420-560 for the
// general keys, :574-600 LLM_ARCH_LLAMA, :1838-1850 LLM_ARCH_OPENAI_MOE), tensor types and bytes from the file. What the context decides
// stays a parameter: n_ctx, n_seq_max, flash_attn, n_ubatch, and the layer range of a -sm layer rank (layer_end < 0: all layers).
// hp_out (optional) receives the hyper-parameters. On any error: NULL with the message in err (the loader's "error loading model: ...").
GGML_API struct mi_llama * mi_llama_create_from_gguf(ggml_backend_t backend, const char * path, int n_ctx, int n_seq_max, int flash_attn, int n_ubatch,
                                                      int layer_begin, int layer_end, struct mi_llama_hparams * hp_out, char * err, int err_cap) {
    try {
        std::unique_ptr<mi355x::gguf_file> gf(new mi355x::gguf_file);
        gf->open(path);
        const std::string arch = gf->get_str("general.architecture");
        if (arch != "llama" && arch != "gpt-oss") throw std::runtime_error("unknown model architecture: '" + arch + "' (this path builds llama and gpt-oss graphs)");
        auto key = [&](const char * k) { return arch + "." + k; };
        auto opt_u = [&](const char * k, uint64_t def) { return gf->has(key(k)) ? gf->get_u64(key(k)) : def; };
        auto opt_f = [&](const char * k, double def) { return gf->has(key(k)) ? gf->get_f64(key(k)) : def; };
        mi_llama_hparams hp; memset(&hp, 0, sizeof(hp));
        hp.arch    = arch == "gpt-oss";
        hp.n_embd  = (int32_t) gf->get_u64(key("embedding_length"));
        hp.n_layer = (int32_t) gf->get_u64(key("block_count"));
        hp.n_head  = (int32_t) gf->get_f64_at(key("attention.head_count"), 0);
        hp.n_head_kv = gf->has(key("attention.head_count_kv")) ? (int32_t) gf->get_f64_at(key("attention.head_count_kv"), 0) : hp.n_head;
        for (int il = 1; il < hp.n_layer; il++) {     // per-layer arrays are accepted when every layer agrees (get_key_or_arr)
            if ((int32_t) gf->get_f64_at(key("attention.head_count"), il) != hp.n_head ||
                (gf->has(key("attention.head_count_kv")) && (int32_t) gf->get_f64_at(key("attention.head_count_kv"), il) != hp.n_head_kv))
                throw std::runtime_error("per-layer head counts differ: not supported");
        }
        hp.n_expert      = (int32_t) opt_u("expert_count", 0);
        hp.n_expert_used = (int32_t) opt_u("expert_used_count", 0);
        hp.n_ff = hp.n_expert > 0 && gf->has(key("expert_feed_forward_length")) ? (int32_t) gf->get_u64(key("expert_feed_forward_length"))
                                                                                  : (int32_t) gf->get_f64_at(key("feed_forward_length"), 0);
        if (hp.n_head <= 0 || hp.n_embd <= 0 || hp.n_layer <= 0 || hp.n_head_kv <= 0 || hp.n_head % hp.n_head_kv) throw std::runtime_error("invalid head / layer counts");
        hp.n_embd_head = (int32_t) opt_u("attention.key_length", (uint64_t) hp.n_embd/hp.n_head);
        if (opt_u("attention.value_length", (uint64_t) hp.n_embd_head) != (uint64_t) hp.n_embd_head) throw std::runtime_error("key and value head sizes differ: not supported");
        if (opt_u("rope.dimension_count", (uint64_t) hp.n_embd_head) != (uint64_t) hp.n_embd_head) throw std::runtime_error("partial rotary dimension: not supported");
        hp.f_norm_rms_eps  = (float) gf->get_f64(key("attention.layer_norm_rms_epsilon"));
        hp.rope_freq_base  = (float) opt_f("rope.freq_base", 10000.0);
        const double factor = opt_f("rope.scaling.factor", 0.0);
        hp.rope_freq_scale = factor == 0.0 ? 1.0f : 1.0f/(float) factor;                       // src/llama-model.cpp:497-505
        const uint64_t n_ctx_train = opt_u("context_length", 0);
        hp.n_ctx_orig = (int32_t) opt_u("rope.scaling.original_context_length", n_ctx_train);
        hp.rope_type = hp.arch == 1 ? 2 : 0;                                                   // llama_model_rope_type: NORM for llama, NEOX for gpt-oss
        hp.ftype = gf->has("general.file_type") ? (int32_t) gf->get_u64("general.file_type") : 0;
        if (hp.arch == 1) {                                                                    // :1838-1850
            hp.n_swa = (int32_t) opt_u("attention.sliding_window", 0);
            hp.swa_pattern = hp.n_swa > 0 ? 2 : 0;
        }
        const mi355x::gguf_tensor_info * te = gf->find("token_embd.weight");
        if (!te) throw std::runtime_error("missing tensor 'token_embd.weight'");
        if (te->ne[0] != hp.n_embd) throw std::runtime_error("tensor 'token_embd.weight' has wrong shape");
        hp.n_vocab = (int32_t) te->ne[1];
        hp.has_rope_freqs = gf->find("rope_freqs.weight") != nullptr;
        hp.is_70b = 0;                                                                          // only steers the synthetic type choice
        const int pad = flash_attn ? 256 : KV_PAD;
        hp.n_ctx = (int32_t) GGML_PAD(std::max(1, n_ctx), pad);
        hp.n_seq_max = std::max(1, n_seq_max); hp.flash_attn = flash_attn != 0; hp.n_ubatch = n_ubatch > 0 ? n_ubatch : 512;
        hp.layer_begin = std::max(0, layer_begin); hp.layer_end = layer_end < 0 ? hp.n_layer : std::min(layer_end, hp.n_layer);
        hp.has_output = hp.layer_end == hp.n_layer;

        // the input layer's rows stay on the host
        const size_t row = ggml_row_size((enum ggml_type) te->type, te->ne[0]);
        const uint8_t * tdata = gf->tensor_data(*te, row*(size_t) te->ne[1]);
        std::vector<float> probe((size_t) hp.n_embd);
        if (!mi355x::dequant_row_host((int) te->type, tdata, probe.data(), hp.n_embd)) throw std::runtime_error("token_embd.weight: no host decoder for type " + std::to_string((int) te->type));
        const int tt = (int) te->type;
        mi_llama * m = create_impl(backend, &hp, 0, std::move(gf));
        if (!m) throw std::runtime_error("weight / KV allocation failed");
        m->tok_embd = tdata; m->tok_embd_type = tt; m->tok_embd_row = row;
        if (hp_out) *hp_out = m->hp;
        return m;
    } catch (const std::exception & e) {
        if (err && err_cap > 0) snprintf(err, (size_t) err_cap, "error loading model: %s", e.what());
        return nullptr;
    }
}

GGML_API void mi_llama_free(struct mi_llama * m) {
    if (!m) return;
    ggml_backend_synchronize(m->backend);
    for (auto & kv : m->graphs) free_graph(kv.second);
    if (m->hbuf) ggml_backend_buffer_free(m->hbuf); else free(m->hbase);
    if (m->lbuf) ggml_backend_buffer_free(m->lbuf); else free(m->logits);
    if (m->wbuf) ggml_backend_buffer_free(m->wbuf);
    if (m->wsplit_buf) ggml_backend_buffer_free(m->wsplit_buf);
    if (m->kvbuf) ggml_backend_buffer_free(m->kvbuf);
    if (m->wsplit) ggml_free(m->wsplit);
    ggml_free(m->wctx); ggml_free(m->kvctx);
    delete m;
}

// algorithmic weight bytes per decoded token: every dense matrix + the used fraction of the expert stacks
GGML_API uint64_t mi_llama_weight_bytes(const struct mi_llama * m) {
    return m->weight_bytes + (m->hp.n_expert > 0 ? m->expert_bytes*(uint64_t) m->hp.n_expert_used/(uint64_t) m->hp.n_expert : 0);
}
GGML_API int      mi_llama_n_past(const struct mi_llama * m, int seq) { return m->n_past[seq]; }
GGML_API void     mi_llama_kv_clear(struct mi_llama * m) { for (auto & p : m->n_past) p = 0; for (auto & c : m->swa_cell_pos) std::fill(c.begin(), c.end(), -1); }   // llama_memory_clear(mem, false): metadata only (llama-bench.cpp:1974)
GGML_API int      mi_llama_n_result(const struct mi_llama * m) { return (int) m->logits_n; }

// tensor access for graph-level parity tests
GGML_API struct ggml_tensor * mi_llama_get_tensor(struct mi_llama * m, const char * name) {
    ggml_tensor * found = nullptr;
    for_each_weight(m, [&](ggml_tensor * t) { if (!found && strcmp(t->name, name) == 0) found = t; });
    return found;
}

// One llama_decode() of n_tokens tokens of sequence 0 (src/llama-context.cpp:946-1254 restricted to one ubatch):
// find the KV slot, (re)use the graph, set_inputs, graph_compute_async, read the result back, synchronize.
//   tokens   : token ids (seed the synthetic embeddings); ignored when dev_act_in != NULL
//   dev_act_in  : optional DEVICE pointer, f32 [n_embd, n_tokens]: activations handed over from the previous layer-split rank
//   result_host : optional host buffer for the result (n_vocab logits of the last token, or [n_embd, n_tokens] l_out on non-final ranks)
//   dev_result_out : optional DEVICE pointer that receives the result instead (hand-off to the next rank)
//   do_sync : bit 0 = synchronize before returning; bit 1 = no host read-back (measurements)
GGML_API int mi_llama_decode(struct mi_llama * m, int seq, const int32_t * tokens, int n_tokens, const void * dev_act_in,
                             float * result_host, void * dev_result_out, int do_sync) {
    const mi_llama_hparams & hp = m->hp;
    if (seq < 0 || seq >= (int) m->n_past.size()) return -1;
    if (n_tokens <= 0 || m->n_past[seq] + n_tokens > hp.n_ctx) return 1;   // "could not find a KV slot" (src/llama-context.cpp:1006)
    const int head = m->n_past[seq];
    const int kv_pad = hp.flash_attn ? 256 : KV_PAD;                                                   // get_padding (:2407-2410)
    const int n_kv = std::min(hp.n_ctx, std::max(kv_pad, (int) GGML_PAD(head + n_tokens, kv_pad)));   // get_n_kv (:1040-1050)

    // the window cache: a token at position p lives in cell p % swa_size (the cell it overwrites held p - swa_size, out of every
    // window that is still open since swa_size >= n_swa + n_tokens); n_kv covers the highest cell in use (get_n_kv)
    int n_kv_swa = 0;
    if (m->swa_size > 0) {
        if (m->swa_size < hp.n_ctx && hp.n_swa + n_tokens > m->swa_size) return 1;      // a batch beyond n_ubatch would overwrite cells of its own window
        std::vector<int> & cp = m->swa_cell_pos[seq];
        for (int i = 0; i < n_tokens; i++) cp[(head + i) % m->swa_size] = head + i;
        int used_max_p1 = 0;
        for (int j = 0; j < m->swa_size; j++) if (cp[j] >= 0) used_max_p1 = j + 1;
        n_kv_swa = std::min(m->swa_size, std::max(kv_pad, (int) GGML_PAD(used_max_p1, kv_pad)));
    }
    auto key = std::make_tuple(seq, n_tokens, n_kv, n_kv_swa);
    auto it = m->graphs.find(key);
    if (it == m->graphs.end()) {
        if (m->graphs.size() >= 6*m->n_past.size()) {   // bound the number of live compute buffers
            auto victim = m->graphs.begin();
            for (auto j = m->graphs.begin(); j != m->graphs.end(); ++j) if (j->second.last_use < victim->second.last_use) victim = j;
            ggml_backend_synchronize(m->backend);
            free_graph(victim->second);
            m->graphs.erase(victim);
        }
        it = m->graphs.emplace(key, build_graph(m, seq, n_tokens, n_kv, n_kv_swa)).first;
    }
    graph_inst & g = it->second;
    g.last_use = ++m->tick;

    // ---- set_inputs (src/llama-graph.cpp:16-58, src/llama-kv-cache-unified.cpp:1219-1400) through pinned staging
    const int64_t n_embd = hp.n_embd, n_embd_v_gqa = (int64_t) hp.n_embd_head*hp.n_head_kv, kv_size = hp.n_ctx;
    const size_t slot_size = m->hsize/4;
    uint8_t * hp_ = m->hbase + (size_t) m->hslot*slot_size; size_t off = 0;
    m->hslot = (m->hslot + 1) % 4;
    auto stage = [&](size_t bytes) { uint8_t * p = hp_ + off; off += (bytes + 255) & ~(size_t) 255; if (off > slot_size) { fprintf(stderr, "mi_llama: staging overflow\n"); abort(); } return p; };

    if (dev_act_in) {
        if (!m->set_from_device) return -4;
        m->set_from_device(m->backend, g.inp_embd, dev_act_in, 0, ggml_nbytes(g.inp_embd));   // device-to-device
    } else {
        float * e = (float *) stage((size_t) n_embd*n_tokens*4);
        if (m->tok_embd) for (int i = 0; i < n_tokens; i++) { const int32_t tk = tokens ? tokens[i] : i; if (tk < 0 || tk >= hp.n_vocab) return -1; }   // llama_decode: "invalid token" (src/llama-batch.cpp:60-75)
        // GET_ROWS(token_embd, tokens) on the host (build_inp_embd, src/llama-graph.cpp:1059-1075) — the CPU backend runs it on its thread pool, so a prompt's
        // rows are spread over threads here too (one thread generated 512 x 4096 synthetic values in 3.9 ms: a quarter of a pp512 pass spent in the test harness)
        auto rows = [&](int i0, int i1) {
            for (int i = i0; i < i1; i++) {
                const int32_t tk = tokens ? tokens[i] : i;
                if (m->tok_embd) mi355x::dequant_row_host(m->tok_embd_type, m->tok_embd + (size_t) tk*m->tok_embd_row, e + (size_t) i*n_embd, n_embd);
                else synth_embedding(e + (size_t) i*n_embd, (int) n_embd, tk, m->seed);
            }
        };
        const unsigned nthr = n_tokens >= 32 ? std::max(1u, std::min(std::min(16u, std::thread::hardware_concurrency()), (unsigned) n_tokens/8)) : 1u;
        if (nthr <= 1) rows(0, n_tokens);
        else {
            std::vector<std::thread> th;
            for (unsigned t = 0; t < nthr; t++) th.emplace_back(rows, (int)((int64_t) n_tokens*t/nthr), (int)((int64_t) n_tokens*(t + 1)/nthr));
            for (auto & x : th) x.join();
        }
        ggml_backend_tensor_set_async(m->backend, g.inp_embd, e, 0, ggml_nbytes(g.inp_embd));
    }
    {
        int32_t * p = (int32_t *) stage((size_t) n_tokens*4);
        for (int i = 0; i < n_tokens; i++) p[i] = head + i;
        ggml_backend_tensor_set_async(m->backend, g.inp_pos, p, 0, ggml_nbytes(g.inp_pos));
    }
    {   // causal mask over the used cells (set_input_kq_mask)
        const int64_t ne0 = g.kq_mask->ne[0], ne1 = g.kq_mask->ne[1];
        float * p = (float *) stage((size_t) ne0*ne1*4);
        for (int64_t i = 0; i < ne0*ne1; i++) p[i] = -INFINITY;
        for (int i = 0; i < n_tokens; i++) {
            const int p1 = head + i;
            for (int j = 0; j <= p1 && j < n_kv; j++) p[(int64_t) i*ne0 + j] = 0.0f;
        }
        ggml_backend_tensor_set_async(m->backend, g.kq_mask, p, 0, ggml_nbytes(g.kq_mask));
    }
    {
        int64_t * p = (int64_t *) stage((size_t) n_tokens*8);
        for (int i = 0; i < n_tokens; i++) p[i] = head + i;
        ggml_backend_tensor_set_async(m->backend, g.k_idxs, p, 0, ggml_nbytes(g.k_idxs));
    }
    if (hp.flash_attn) {   // V rows are cells: one index per token (set_input_v_idxs :1240-1250)
        int64_t * p = (int64_t *) stage((size_t) n_tokens*8);
        for (int i = 0; i < n_tokens; i++) p[i] = head + i;
        ggml_backend_tensor_set_async(m->backend, g.v_idxs, p, 0, ggml_nbytes(g.v_idxs));
    } else {   // v_trans: one index per element (set_input_v_idxs :1252-1267)
        int64_t * p = (int64_t *) stage((size_t) n_tokens*n_embd_v_gqa*8);
        for (int i = 0; i < n_tokens; i++)
            for (int64_t j = 0; j < n_embd_v_gqa; j++) p[(int64_t) i*n_embd_v_gqa + j] = j*kv_size + head + i;
        ggml_backend_tensor_set_async(m->backend, g.v_idxs, p, 0, ggml_nbytes(g.v_idxs));
    }
    if (m->swa_size > 0) {   // the same three inputs for the window cache (llm_graph_input_attn_kv_unified_iswa::set_input, src/llama-graph.cpp:353-369)
        const std::vector<int> & cp = m->swa_cell_pos[seq];
        const int64_t ne0 = g.kq_mask_swa->ne[0], ne1 = g.kq_mask_swa->ne[1], ssz = m->swa_size;
        float * pm = (float *) stage((size_t) ne0*ne1*4);
        for (int64_t i = 0; i < ne0*ne1; i++) pm[i] = -INFINITY;
        for (int i = 0; i < n_tokens; i++) {
            const int p1 = head + i;
            // is_masked_swa, LLAMA_SWA_TYPE_STANDARD (src/llama-hparams.cpp / llama-kv-cache-unified.cpp:1337-1358): empty, future, or p1 - p0 >= n_swa
            for (int j = 0; j < n_kv_swa; j++) if (cp[j] >= 0 && cp[j] <= p1 && p1 - cp[j] < hp.n_swa) pm[(int64_t) i*ne0 + j] = 0.0f;
        }
        ggml_backend_tensor_set_async(m->backend, g.kq_mask_swa, pm, 0, ggml_nbytes(g.kq_mask_swa));
        int64_t * pk = (int64_t *) stage((size_t) n_tokens*8);
        for (int i = 0; i < n_tokens; i++) pk[i] = (head + i) % ssz;
        ggml_backend_tensor_set_async(m->backend, g.k_idxs_swa, pk, 0, ggml_nbytes(g.k_idxs_swa));
        if (hp.flash_attn) {
            int64_t * pv = (int64_t *) stage((size_t) n_tokens*8);
            for (int i = 0; i < n_tokens; i++) pv[i] = (head + i) % ssz;
            ggml_backend_tensor_set_async(m->backend, g.v_idxs_swa, pv, 0, ggml_nbytes(g.v_idxs_swa));
        } else {
            int64_t * pv = (int64_t *) stage((size_t) n_tokens*n_embd_v_gqa*8);
            for (int i = 0; i < n_tokens; i++)
                for (int64_t j = 0; j < n_embd_v_gqa; j++) pv[(int64_t) i*n_embd_v_gqa + j] = j*ssz + (head + i) % ssz;
            ggml_backend_tensor_set_async(m->backend, g.v_idxs_swa, pv, 0, ggml_nbytes(g.v_idxs_swa));
        }
    }
    {
        int32_t * p = (int32_t *) stage(4);
        p[0] = n_tokens - 1;
        ggml_backend_tensor_set_async(m->backend, g.out_ids, p, 0, 4);
    }

    // ---- graph_compute (src/llama-context.cpp:1432-1457)
    const enum ggml_status st = ggml_backend_graph_compute_async(m->backend, g.gf);
    if (st != GGML_STATUS_SUCCESS) return st == GGML_STATUS_ALLOC_FAILED ? -2 : (st == GGML_STATUS_ABORTED ? 2 : -3);   // :1101-1106

    // ---- results (src/llama-context.cpp:1132)
    const size_t rbytes = ggml_nbytes(g.result);
    if (dev_result_out) {
        // device-to-device hand-off buffer (consumed by the caller's RCCL send)
        if (!m->get_to_device) return -4;
        m->get_to_device(m->backend, g.result, dev_result_out, 0, rbytes);
    }
    if (result_host || (!dev_result_out && !(do_sync & 2))) {
        // result_host == NULL: into the model's own pinned output buffer (mi_llama_last_logits), as llama_decode leaves the logits in buf_output
        if (!result_host) { if (!reserve_result(m, rbytes/4)) return -2; m->logits_n = rbytes/4; }
        ggml_backend_tensor_get_async(m->backend, g.result, result_host ? result_host : m->logits, 0, rbytes);
    }
    m->n_past[seq] += n_tokens;
    if (do_sync & 1) ggml_backend_synchronize(m->backend);   // llama_synchronize (llama-bench.cpp:1806 does it after every token)
    return 0;
}

// the synthetic embedding row the harness feeds for `token` (so that tests can rebuild the same input)
GGML_API void mi_llama_synth_embedding(const struct mi_llama * m, int32_t token, float * out) {
    if (m->tok_embd) {
        if (token >= 0 && token < m->hp.n_vocab) mi355x::dequant_row_host(m->tok_embd_type, m->tok_embd + (size_t) token*m->tok_embd_row, out, m->hp.n_embd);
        return;
    }
    synth_embedding(out, m->hp.n_embd, token, m->seed);
}

GGML_API const float * mi_llama_last_logits(const struct mi_llama * m) { return m->logits; }

// number of graph nodes in the decode graph for (n_tokens, n_kv) — reporting only
GGML_API int mi_llama_graph_nodes(struct mi_llama * m, int n_tokens) {
    for (auto & kv : m->graphs) if (std::get<1>(kv.first) == n_tokens) return ggml_graph_n_nodes(kv.second.gf);
    return 0;
}

} // extern "C"
