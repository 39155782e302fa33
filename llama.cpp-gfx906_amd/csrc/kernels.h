// kernels.h — host-callable launchers for the gfx950 kernels. Plain pointers,
// strides in BYTES (ggml nb[] convention) and a hipStream_t; no ggml types, so the
// kernels can be driven from backend.cpp (ggml tensors) or from the flat C-ABI in
// include/ggml-mi355x.h. Every launcher is asynchronous on `stream` and performs
// no allocation or synchronisation (safe under hipGraph capture).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

namespace mi355x {

// ggml_type ids used on the path (gguf-py/gguf/constants.py:2698-2730)
enum : int {
    T_F32 = 0, T_F16 = 1, T_Q4_0 = 2, T_Q8_0 = 8, T_Q4_K = 12, T_Q5_K = 13, T_Q6_K = 14, T_Q8_K = 15,
    T_I32 = 26, T_I64 = 27, T_BF16 = 30, T_MXFP4 = 39,
};

// ---- activation quantisation (the CPU path's vec_dot_type: Q8_0 for Q4_0/Q8_0/MXFP4, Q8_K for K-quants)
// Device layout (SoA, internal): for a [k, n] activation
//   qs    : int8  [n][k]
//   d     : float [n][k/32]   (Q8_0: value of the f16-rounded scale)  | float [n][k/256] (Q8_K)
//   bsums : int16 [n][k/32]   (Q8_0: sum of the 32 quants)            | int16 [n][k/16]  (Q8_K)
struct act_q8 {
    int8_t  * qs;
    float   * d;
    int16_t * bsums;
    int       kind;   // T_Q8_0 or T_Q8_K
    int64_t   k;
    int64_t   n;
};

size_t act_q8_bytes(int kind, int64_t k, int64_t n);            // scratch bytes needed
act_q8 act_q8_carve(void * scratch, int kind, int64_t k, int64_t n);
int    act_kind_for(int type_a);                                   // T_Q8_0 / T_Q8_K / -1

// x: f32, rows of k contiguous floats; row r (< q.n) lives at
//   x + (r % n_inner)*stride_inner + (r / n_inner)*stride_outer      (bytes)
// (n_inner = q.n, stride_outer = 0 for a plain 2-D activation; the two-level form serves MUL_MAT_ID's [k, n_b, n_tokens] src1)
void quantize_act(const float * x, int64_t n_inner, size_t stride_inner, size_t stride_outer, const act_q8 & q, hipStream_t stream);

// ---- quantized mat-vec (decode, n <= MMVQ_MAX_N): dst[col*dst_stride + row] = W[row,:] . x[col,:]
constexpr int MMVQ_MAX_N = 8;
bool mul_mat_vec_q_supported(int type_a);
void mul_mat_vec_q(int type_a, const void * W, size_t w_row_stride, int64_t m, int64_t k,
                   const act_q8 & act, int64_t n, float * dst, size_t dst_col_stride_bytes, hipStream_t stream);

void mul_mat_vec_q_batched(int type_a, const void * W, size_t w_row_stride, size_t w_batch_stride, int r2, int64_t m, int64_t k,
                           const act_q8 & act, int64_t n, int64_t n_batch, float * dst, size_t dst_col_stride_bytes, size_t dst_batch_stride_bytes, hipStream_t stream);

// 2 <= n <= 8 columns of Q4_K / Q5_K / Q6_K on the streamed weight path (mmvq_stream_cols.hip: LDS-DMA ring, every unit multiplied with all columns);
// false = not taken (k % 2048, the column images + two slots must fit LDS)
bool mul_mat_vec_q_stream_cols(int type_a, const void * W, size_t w_row_stride, int64_t m, int64_t k,
                               const act_q8 & act, int64_t n, float * dst, size_t dst_col_stride_bytes, hipStream_t stream);
// 2 <= n <= 8 columns of Q4_K / Q5_K / Q6_K on the int8 matrix cores; false = not taken (the caller runs its own kernel)
bool mul_mat_vec_q_cols_mfma(int type_a, const void * W, size_t w_row_stride, int64_t m, int64_t k,
                             const act_q8 & act, int64_t n, float * dst, size_t dst_col_stride_bytes, hipStream_t stream);

// MUL_MAT_ID decode form: for pair p (< n_pairs): dst[p*dst_stride + :] = W[expert[p]] . act row (p % act rows as given by act_row[p])
void mul_mat_vec_q_id(int type_a, const void * W, size_t w_row_stride, size_t w_expert_stride, int64_t m, int64_t k,
                      const act_q8 & act, const int32_t * ids, size_t ids_nb0, size_t ids_nb1,
                      int64_t n_used, int64_t n_tokens, int64_t n_b,
                      float * dst, size_t dst_nb1, size_t dst_nb2, hipStream_t stream);

// ---- quantized mat-mat (prefill, n > MMVQ_MAX_N): dequantize-to-bf16 tiles in LDS + v_mfma_f32_32x32x16_bf16 ----
// x: f32 rows of k floats at x + i*x_row_stride; scratch: mul_mat_q_scratch_bytes(k, n, m) bytes (the bf16 copy of x + the split-K planes)
// (scratch_ready: it already holds the copy of exactly this x — the caller's cache — so the conversion pass is skipped)
// res != NULL: dst = W.x + res (f32 rows of m floats at res + i*res_row_stride; may be dst itself) — the residual ADD of build_attn / build_ffn
size_t mul_mat_q_scratch_bytes(int64_t k, int64_t n, int64_t m);
void mul_mat_q(int type_a, const void * W, size_t w_row_stride, int64_t m, int64_t k,
               const float * x, size_t x_row_stride, int64_t n, void * scratch, bool scratch_ready, float * dst, size_t dst_col_stride_bytes,
               const float * res, size_t res_row_stride, hipStream_t stream, struct mmq_deferred * defer = nullptr);
// defer != NULL: when the launch splits k, the pass that adds the planes (and res) is left to the caller (who fuses it with the norm that
// follows): np = number of planes (0: nothing deferred, dst is complete), planes = dense [np][n][m] f32 in the scratch
struct mmq_deferred { int np; const float * planes; };

struct mmvq_rope;
// KV-cache writes done by the pass that combines the launch's split-k planes: per segment, mode 0 none | 1 f16 rows st16[idx[token]*st_row_elems + col] |
// 2 element scatter st16[idx[token*m + col]] (the transposed V cache)
struct mmq_kv_store { uint16_t * st16[3]; const int64_t * st_idx[3]; int64_t st_row_elems[3]; int st_mode[3]; };
// 2 or 3 mat-muls on the same activations (wq / wk / wv) as one launch of 256-token tiles; false = not done, run them one by one
bool mul_mat_q_multi(int nseg, const int * types, const void * const * W, const size_t * w_row_stride, const int64_t * m, float * const * dst, const size_t * dst_stride,
                     int64_t k, const float * x, size_t x_row_stride, int64_t n, void * scratch, size_t scratch_size, bool scratch_ready,
                     const struct mmvq_rope * rope, const int * seg_rope, const struct mmq_kv_store * kvs, hipStream_t stream,      // rope != NULL: NORM rotary embedding on the segments flagged in seg_rope
                     const float * const * seg_bias = nullptr);     // seg_bias[s] != NULL: m[s] floats added to every row of segment s (only without a k split: else false)

// gate / up + SwiGLU for many tokens in one kernel: dst[n][m] = silu(Wg.x) * (Wu.x) (both weight tensors of one type and shape);
// supported when the 256-token tiles fill the chip
bool mul_mat_q_glu_supported(int64_t m, int64_t n);
// y16 != NULL (m % 64 == 0): the result also (dst == NULL: only) as the bf16 activation copy of the mat-mul that follows (ffn_down), rows of m elements
void mul_mat_q_glu(int type_a, const void * Wg, const void * Wu, size_t w_row_stride, int64_t m, int64_t k,
                   const float * x, size_t x_row_stride, int64_t n, void * scratch, bool scratch_ready, float * dst, size_t dst_col_stride_bytes, hipStream_t stream,
                   uint16_t * y16 = nullptr);
size_t mul_mat_q_x_bytes(int64_t k, int64_t n);      // size of the bf16 activation copy at the start of the scratch

// ---- MUL_MAT_ID for many tokens (src/llama-graph.cpp:569-595): (token, slot) pairs sorted by expert on the device, then the tiled
// MFMA kernel per (expert, 128 pairs). b: f32 [k, n_b, n_tokens] (n_b = 1 or n_used); ids: i32 [n_used, n_tokens] (strided);
// dst: f32 [m, n_used, n_tokens]. No host round trip, so the launch sequence can be captured in a hipGraph.
bool   mul_mat_q_id_supported(int64_t n_expert, int64_t n_used, int64_t n_tokens);
size_t mul_mat_q_id_scratch_bytes(int64_t k, int64_t n_b, int64_t n_tokens, int64_t n_used, int64_t n_expert);
void   mul_mat_q_id(int type_a, const void * W, size_t w_row_stride, size_t w_expert_stride, int64_t m, int64_t k,
                    const float * b, size_t b_nb1, size_t b_nb2, int64_t n_b,
                    const int32_t * ids, size_t ids_nb0, size_t ids_nb1, int64_t n_used, int64_t n_tokens, int64_t n_expert,
                    void * scratch, float * dst, size_t dst_nb1, size_t dst_nb2, hipStream_t stream);
// The pieces of the above, for the fused expert chain of a prompt pass (backend.cpp try_fused_prefill_moe): one sort and one bf16 copy of the layer input serve
// gate, up and down; gate / up run as ONE dual launch with the ADD_ID biases and the (oai) SwiGLU in its epilogue and hand the down projection bf16 rows per pair;
// the down launch adds its ADD_ID bias and multiplies with the routing weight (MUL(experts, weights), src/llama-graph.cpp:990)
struct mmq_moe_epi { const float * bias = nullptr; const float * bias2 = nullptr; size_t bias_stride = 0; const float * scale = nullptr; size_t scale_nb0 = 0, scale_nb1 = 0;
                     int oai = 0; float alpha = 0.0f, limit = 0.0f;
                     const float * glu_up = nullptr; size_t glu_up_nb1 = 0, glu_up_nb2 = 0; };     // (single-tensor launch as the gate half of a GLU: the finished up half, f32 [m, n_used, n_tokens])
struct mmq_moe_plan { uint16_t * xb; int * table; int * table2; uint16_t * y16; size_t bytes; };     // table: 256-pair tiles (dual launch); table2: the down launch's tiles
mmq_moe_plan mul_mat_q_id_plan(void * scratch, int64_t k, int64_t n_tokens, int64_t n_used, int64_t n_expert, int64_t m_glu);
int    mul_mat_q_id_tile(int64_t n_used, int64_t n_tokens, int64_t n_expert);      // pairs per tile: 128 or 256
void   mul_mat_q_id_sort(const int32_t * ids, size_t ids_nb0, size_t ids_nb1, int64_t n_used, int64_t n_tokens, int64_t n_expert, int tile, int * table, hipStream_t stream);
void   mul_mat_q_id_act16(const float * b, size_t b_nb1, size_t b_nb2, int64_t k, int64_t n_b, int64_t n_tokens, uint16_t * xb, hipStream_t stream);
void   mul_mat_q_id_tiles(int type_a, const void * W, const void * W2, size_t w_row_stride, size_t w_expert_stride, int64_t m, int64_t k,
                          const uint16_t * xb, int64_t n_b, const int * table, int tile, int64_t n_used, int64_t n_tokens, int64_t n_expert,
                          const mmq_moe_epi & epi, float * dst, size_t dst_nb1, size_t dst_nb2, uint16_t * y16, hipStream_t stream);

// ---- dense f16/f32 x f32 mat-mul with ggml broadcast (attention K.Q and V.KQ; tests/test-backend-ops.cpp:5791-5813)
struct mm_dense_args {
    const void * a; int type_a; int64_t ne00, ne01, ne02, ne03; size_t nb00, nb01, nb02, nb03;
    const void * b; int type_b; int64_t ne10, ne11, ne12, ne13; size_t nb10, nb11, nb12, nb13;
    float * dst; size_t nb1, nb2, nb3;
};
void mul_mat_dense(const mm_dense_args & p, hipStream_t stream);
// MUL_MAT_ID over an F16 / BF16 / F32 expert stack (correctness path: a wave per (row, pair))
void mul_mat_id_dense(int type_a, const void * as, size_t nb00, size_t nb01, size_t nb02, int64_t m, int64_t k, const void * b, size_t nb10, size_t nb11, size_t nb12, int64_t n_b,
                      const void * ids, size_t ids_nb0, size_t ids_nb1, int64_t n_used, int64_t n_tokens, int64_t n_expert, float * dst, size_t nb1, size_t nb2, hipStream_t stream);
// matrix-core form for more than 8 columns of f16 x f32 (prefill attention): f32 b is converted to f16 in `scratch`
bool   mul_mat_dense_mfma_supported(const mm_dense_args & p);
size_t mul_mat_dense_mfma_scratch_bytes(const mm_dense_args & p);
void   mul_mat_dense_mfma(const mm_dense_args & p, void * scratch, hipStream_t stream);

// ---- element kernels ---------------------------------------------
struct tensor_desc {           // a strided 4-D view, ggml convention (ne = elements, nb = bytes)
    void * data; int type; int64_t ne[4]; size_t nb[4];
};

void rms_norm(const tensor_desc & src, const tensor_desc & dst, float eps, hipStream_t stream);
// fused RMS_NORM * w (+ add) — tests/test-backend-ops.cpp:2856
// y16 != NULL (no add; rms_norm_mul_bf16_supported): also writes the bf16 copy of the result in the layout the prefill mat-mul reads
// (rows of ne0 rounded up to 64 elements, zero tail), so that its activation pre-pass is skipped
bool rms_norm_mul_bf16_supported(const tensor_desc & src, const tensor_desc & w, const tensor_desc & dst);
void rms_norm_mul(const tensor_desc & src, const tensor_desc & w, const tensor_desc * add, const tensor_desc & dst, float eps, hipStream_t stream, uint16_t * y16 = nullptr);
// the split-k planes of a prefill mat-mul (mul_mat_q with `defer`) + residual -> sum_out, and RMS_NORM(sum) * w -> y (+ its bf16 copy y16, or NULL);
// m % 4 == 0, all rows 16-byte aligned, np = 2 | 4 | 8
void combine_rms_norm(const float * planes, int np, int64_t m, int64_t n, const float * res, size_t res_nb1, float * sum_out, size_t sum_nb1,
                      const float * w, float * y, size_t y_nb1, uint16_t * y16, float eps, hipStream_t stream);
enum bin_op { BIN_ADD = 0, BIN_MUL = 1, BIN_DIV = 2, BIN_SUB = 3 };
void bin_bcast(int op, const tensor_desc & a, const tensor_desc & b, const tensor_desc & dst, hipStream_t stream);
void add_id(const tensor_desc & a, const tensor_desc & bias, const tensor_desc & ids, const tensor_desc & dst, hipStream_t stream);
void scale(const tensor_desc & src, const tensor_desc & dst, float s, float b, hipStream_t stream);
void cpy(const tensor_desc & src, const tensor_desc & dst, hipStream_t stream);      // f32<->f16/f32 strided copy/convert (CPY, CONT, DUP)
void set_rows(const tensor_desc & src, const tensor_desc & idx, const tensor_desc & dst, hipStream_t stream);
void get_rows(const tensor_desc & src, const tensor_desc & idx, const tensor_desc & dst, hipStream_t stream);
void sum_rows(const tensor_desc & src, const tensor_desc & dst, hipStream_t stream);
// one token: logits = w . x (+ bias) [-> soft_max -> probs] -> sorted = argsort descending (elem.hip: k_moe_route); n_expert <= 256,
// k % 4 == 0, 16-byte aligned rows; logits / probs may be NULL when the graph does not read them
bool moe_route_norm_supported(int64_t k, int64_t n_expert, const float * ws);
// norm_w != NULL: x is the raw residual stream; the router's input y = RMS_NORM(x) * norm_w is computed in the kernel and written to y_out
void moe_route(const float * w, size_t w_nb1, const float * x, const float * bias, int64_t k, int64_t n_expert, bool softmax,
               float * logits, float * probs, int32_t * sorted, hipStream_t stream, float * ws,
               const float * norm_w = nullptr, float eps = 0.0f, float * y_out = nullptr,    // ws: 257 zero-initialised words (>= 16 experts: several workgroups)
               unsigned * err = nullptr,    // host-mapped words: err[1] is set when the ranking workgroup's bounded wait for a logit gave up
               float * topv = nullptr);     // 8 floats: the best values in rank order (moe_combine / the grouped mat-vec's plane weights take them with ids == NULL)
// one token: dst[i] = sum_u experts[u][i] * w_u (+ res[i]); w from probs[ids[u]] (ids == NULL: probs[u]) normalised (mode 0) or soft_max'ed (mode 1);
// n_used <= 8, n_embd % 4 == 0, 16-byte aligned rows (elem.hip: k_moe_combine)
void moe_combine(const float * probs, const int32_t * ids, int n_used, int mode, const void * experts, size_t e_nb1, int64_t n_embd,
                 const float * res, float * dst, hipStream_t stream);
// many tokens: dst[t] = ((experts[t][0] + experts[t][1]) + ...) [+ res[t]] (m % 4 == 0, 16-byte aligned rows; elem.hip: k_slot_sum)
void moe_slot_sum(const void * experts, size_t nb1, size_t nb2, int n_used, int64_t m, int64_t n_tokens, const float * res, size_t res_nb1, float * dst, size_t dst_nb1, hipStream_t stream);
void argsort(const tensor_desc & src, const tensor_desc & dst, int order, hipStream_t stream);
void unary(int op, const tensor_desc & src, const tensor_desc & dst, hipStream_t stream);
void glu(int glu_op, bool swapped, const tensor_desc & a, const tensor_desc * b, const tensor_desc & dst, float alpha, float limit, hipStream_t stream);

struct rope_params {
    int n_dims, mode, n_ctx_orig;
    float freq_base, freq_scale, ext_factor, attn_factor, beta_fast, beta_slow;
};
void rope(const tensor_desc & src, const int32_t * pos, const float * freq_factors, const tensor_desc & dst, const rope_params & p, hipStream_t stream);

void soft_max(const tensor_desc & src, const tensor_desc * mask, const float * sinks, const tensor_desc & dst,
              float scale, float max_bias, hipStream_t stream);

// ---- fused decode kernels (decode_fused.hip): same arithmetic as the node-by-node kernels, fewer launches -----
// RMS_NORM * w -> f32 row y AND its quantized form q (rows of ne0 floats, ne0 % 256 == 0, ne0 <= 8192)
bool rms_norm_mul_quant_supported(int64_t ne0);
void rms_norm_mul_quant(const float * x, size_t x_stride, const float * w, float * y, size_t y_stride, const act_q8 & q,
                        int64_t ne0, int64_t nrows, float eps, hipStream_t stream);
// swiglu(g, u) = silu(g)*u -> f32 row y AND its quantized form q
void swiglu_quant(const float * g, size_t g_stride, const float * u, size_t u_stride, float * y, size_t y_stride, const act_q8 & q,
                  int64_t ne0, int64_t nrows, hipStream_t stream);
// SET_ROWS(k) + SET_ROWS(v, element scatter) into the f16 cache, one launch
void kv_store_f16(const float * k_src, size_t k_src_nb1, const int64_t * k_idx, void * k_dst, size_t k_dst_nb1, int64_t k_ne0, int64_t k_rows,
                  const float * v_src, const int64_t * v_idx, void * v_dst, int64_t v_n, hipStream_t stream);
// K.q -> softmax -> V^T.p -> [hd*n_head, T] for T query tokens over the f16 cache
// FLASH_ATTN_EXT's remaining parameters: logit_softcap, max_bias (ALiBi) and the cache's element types (0 = F16)
struct attn_extra { float softcap, max_bias; int k_type, v_type; };
bool attn_decode_kv_types_fused(int k_type, int v_type);      // read directly by the row-major-V decode kernel (others: kv_to_f16 first)
void attn_alibi(float max_bias, int64_t n_head, float & m0, float & m1, int & n_head_log2);
// K / V cache views [hd, n_kv, n_head_kv] of type F16 / BF16 / Q8_0 / Q4_0 -> dense f16: rows of cells [n_head_kv][n_kv][hd], or transposed
// (rows over cells, the layout the matrix-core prefill kernel stages with coalesced loads) [n_head_kv][hd][n_kv]; n_kv % 8 == 0, hd 64 / 128
void kv_to_f16(int type, const void * src, size_t nb1, size_t nb2, int64_t hd, int64_t n_kv, int64_t n_head_kv, uint16_t * dst, bool transpose, hipStream_t stream);
bool attn_decode_supported(int64_t head_dim, int64_t n_kv);
void attn_decode(const void * q, size_t q_nb1, size_t q_nb2, const void * k, size_t k_nb1, size_t k_nb2, const void * v, size_t v_nb1, size_t v_nb2,
                 const void * mask, size_t m_nb1, bool mask_f16, const float * sinks, float * dst, size_t dst_nb1,
                 int64_t head_dim, int64_t n_kv, int64_t n_head, int64_t n_head_kv, int64_t T, float scale, hipStream_t stream, bool v_trans = true,
                 float * part = nullptr, size_t part_bytes = 0, bool k_q8_0 = false,      // k_q8_0: K rows are Q8_0 blocks (transposed-V form only)
                 const attn_extra * ex = nullptr);
// long contexts (n_kv >= 384, or 256 with row-major V; a multiple of 32; GGML_MI355X_ATTN_SPLIT_MIN): with a partial buffer of attn_decode_part_bytes() the cells are split over up to 32 workgroups per
// (head, token) and merged by a second small kernel; attn_decode_supported_split: shapes that only work with the buffer (scores beyond one LDS)
size_t attn_decode_part_bytes(int64_t head_dim, int64_t n_kv, int64_t n_head, int64_t T);
bool attn_decode_supported_split(int64_t head_dim, int64_t n_kv);
// the same node group for many tokens (prefill): one wave per (head, 32 queries), scores and probabilities stay in registers
// (attn_prefill.hip); n_kv % 32 == 0
bool attn_prefill_supported(int64_t head_dim, int64_t n_kv);
void attn_prefill(const void * q, size_t q_nb1, size_t q_nb2, const void * k, size_t k_nb1, size_t k_nb2, const void * v, size_t v_nb1, size_t v_nb2,
                  const void * mask, size_t m_nb1, bool mask_f16, const float * sinks, float * dst, size_t dst_nb1,
                  int64_t head_dim, int64_t n_kv, int64_t n_head, int64_t n_head_kv, int64_t T, float scale, hipStream_t stream, bool v_trans = true,
                  uint16_t * y16 = nullptr,      // y16 != NULL: also (dst == NULL: only) the bf16 copy the following mat-mul reads, rows of hd*n_head (a multiple of 64)
                  const attn_extra * ex = nullptr);     // (K and V are f16 here: other cache types are converted by kv_to_f16 first)

// grouped mat-vec (n = 1): up to MMVQ_MAX_GROUPS weight tensors that share one activation vector, each with an epilogue
constexpr int MMVQ_MAX_GROUPS = 4;
enum mmvq_epilogue { EPI_NONE = 0, EPI_ADD = 1, EPI_ROPE = 2, EPI_GLU = 3 };
struct mmvq_group {
    const char * W; const char * W2;   // W2: the second weight tensor of EPI_GLU (dst = silu(W.x) * (W2.x))
    size_t row_stride; int m; int type;
    float * dst; int epi;
    const float * res;                 // EPI_ADD: dst[row] = W.x + res[row]
    // optional f16 store of the result (the KV-cache write that follows wk / wv in the graph):
    //   st_mode 1: row store     st16[st_idx[0]*st_row_elems + row]   (SET_ROWS of one K row)
    //   st_mode 2: element store st16[st_idx[row]]                    (SET_ROWS on the transposed-V [1, N] view)
    uint16_t * st16; const int64_t * st_idx; int64_t st_row_elems; int st_mode;
    // MUL_MAT_ID for one token (src/llama-graph.cpp:569-595): the group's matrices are expert eid[0] of a stack (W += eid[0]*estride,
    // W2 likewise); x_off: this group's activation vector starts x_off floats into the launch's x (the down projection reads one
    // activation per used expert). eid == NULL: a plain weight tensor.
    const int32_t * eid; size_t estride; int x_off;
    // EPI_GLU variants of gpt-oss's expert FFN (src/llama-graph.cpp:927-968): per-expert biases added to the two products (ADD_ID,
    // row `eid[0]` of [m, n_expert] f32 tensors) and swiglu_oai(alpha, limit) instead of swiglu when glu_alpha != 0
    const float * b_gate; const float * b_up; float glu_alpha, glu_limit;
    // EPI_ADD extras: res2 = a second addend, added after res (MUL_MAT -> ADD(bias) -> ADD(residual)); res_eid != 0: res is a [m, n_expert]
    // table and the group adds row eid[0] of it (MUL_MAT_ID -> ADD_ID)
    const float * res2; int res_eid;
};
// GLU launches only: the launch also writes its f32 output as the quantized image (act_q8 layout for n = 1) the next mat-vec reads;
// counters: >= m/256 words, zero between launches (the kernel re-arms them). m % 256 == 0, one group, no expert stack.
struct mmvq_fin { int kind; int pad; int8_t * qs; float * d; int16_t * bs; unsigned * counters; };
// EPI_ROPE. table: n_dims/2 x (cos, sin) for the token at pos[0], filled by mul_mat_vec_q_fused_rope_table on the same stream before the launch
// (one tiny launch per token and rope configuration instead of powf / cosf / sinf per row pair inside every norm+QKV launch)
struct mmvq_rope { const int32_t * pos; const float * freq_factors; int head_dim; rope_params p; const float * table; };
void mul_mat_vec_q_fused_rope_table(const mmvq_rope & rope, float * table, hipStream_t stream);

// where the activation vector comes from
enum mmvq_prologue { PRO_Q8 = 0, PRO_QUANT = 1, PRO_NORM = 2 };
struct mmvq_input {
    int mode;
    act_q8 act;            // PRO_Q8: one quantized column laid out by act_q8_carve(…, n = 1) (one contiguous image)
    const float * x;       // PRO_QUANT / PRO_NORM: the f32 vector (16-byte aligned, k % 256 == 0)
    const float * norm_w;  // PRO_NORM: y = (x * rsqrt(mean(x^2) + eps)) * norm_w, then quantized — RMS_NORM -> MUL folded in
    float eps;
    int act_kind;
    // PRO_NORM only, planes != NULL (then pl_probs != NULL too): the vector is not materialized yet and the launch also stores it to x_out (the graph's
    // ADD result)
    const float * planes; int n_planes; int plane_stride; float * x_out;
    // — the tail of build_moe_ffn (src/llama-graph.cpp:887-1012) left unevaluated: the planes are the used experts' outputs and the vector
    // is sum_u planes[u][i] * w_u (+ x[i], the residual, when x != NULL), w_u from pl_probs[pl_ids[u]] normalised (pl_mode 0) or soft_max'ed (1) — k_moe_combine's
    // arithmetic in k_moe_combine's order
    const float * pl_probs; const int32_t * pl_ids; int pl_mode;
};
bool mul_mat_vec_q_fused_supported(int64_t k, int act_kind);
bool mul_mat_vec_q_fused_prologue_supported(int64_t k, int act_kind);        // PRO_QUANT / PRO_NORM limits (k % 256, or k % 32 with Q8_0 activations)
bool mul_mat_vec_q_fused_can_group(int type_a, int type_b);         // may these two weight types share one grouped launch
bool mul_mat_vec_q_fused_can_group_mixed(int type_a, int type_b);   // pairs of different activation formats (only when the launch quantizes the activation itself)
void mul_mat_vec_q_fused(const mmvq_group * groups, int n_groups, int64_t k, const mmvq_input & in, const mmvq_rope * rope, hipStream_t stream,
                         const mmvq_fin * fin = nullptr);
// the streamed form of the grouped launch (mmvq_stream.h: loader wave + LDS slot ring + one 256-weight unit per lane): K-quant weights in
// contiguous rows, no expert stacks; mul_mat_vec_q_fused routes to it when mul_mat_vec_q_stream_takes says so (it never writes an mmvq_fin image)
bool mul_mat_vec_q_stream_enabled(void);       // GGML_MI355X_STREAM (default 1)
bool mul_mat_vec_q_stream_takes(const mmvq_group * groups, int n_groups, int64_t k, const mmvq_input & in, const mmvq_rope * rope);
void mul_mat_vec_q_stream(const mmvq_group * groups, int n_groups, int64_t k, const mmvq_input & in, const mmvq_rope * rope, hipStream_t stream,
                          hipEvent_t e0, hipEvent_t e1, const char ** kernel_name);
bool mul_mat_vec_q_fused_fin_supported(int64_t m, int64_t k_in);   // may a GLU launch with m output rows carry an mmvq_fin
int  mul_mat_vec_q_fused_share(const mmvq_group * groups, int n_groups, int fw, int * block_end);   // workgroups per group (one workgroup per CU in all)
// (round 1's chained launch held launches back; nothing is held back any more: flush is a no-op kept for its call sites)
void mul_mat_vec_q_fused_flush(hipStream_t stream);
// called right before / after every kernel this module puts on the stream (type of the first group, weight bytes, launches merged, k)
typedef void (*mmvq_launch_hook)(void * ctx, int type, uint64_t weight_bytes, int n_merged, int64_t k);
// the NEXT grouped launch carries (e0, e1) as its dispatch's start / stop events (hipExtLaunchKernelGGL); consumed by that launch
void mul_mat_vec_q_fused_set_launch_events(hipEvent_t e0, hipEvent_t e1);
const char * mul_mat_vec_q_fused_last_kernel(void);     // name of the instantiation the last grouped launch used
void mul_mat_vec_q_fused_set_hooks(mmvq_launch_hook pre, mmvq_launch_hook post, void * ctx);
int  mul_mat_vec_q_fused_pending(uint64_t * weight_bytes);

// ---- small uploads batched into one launch (backend.cpp: be_set_tensor_async / uploads_flush) ----
constexpr int UPLOAD_BATCH_MAX = 12;
struct upload_batch { int n; const void * src[UPLOAD_BATCH_MAX]; void * dst[UPLOAD_BATCH_MAX]; uint32_t bytes[UPLOAD_BATCH_MAX]; int blocks[UPLOAD_BATCH_MAX]; };   // blocks[i] = ceil(bytes[i] / 4096)
void upload_batch_launch(const upload_batch & b, hipStream_t stream);

// ---- test / bench support ------------------------------------------------------------------
// raw streaming read of `bytes` (16 B/lane, nontemporal) — measures the achievable HBM rate on the box
void hbm_read_probe(const void * p, size_t bytes, unsigned * sink, hipStream_t stream);

} // namespace mi355x
