// decode_mega.h — program format of the PERSISTENT decode kernel (decode_mega.hip): one launch runs a whole sequence of the
// single-token phases a Llama-style decode graph consists of (norm + QKV + RoPE + KV store | attention | wo + residual |
// norm + gate/up + SwiGLU | down + residual, layer after layer, then norm + lm_head), one workgroup per CU.
//
// Hand-offs between phases are DATA-TAGGED GRANULES (MI355X guide, R2: "the data is the flag"): every value a later phase needs is
// stored as ONE naturally aligned 8-byte {f32 / packed-int8 word, tag} with a write-through (`sc1`) store; a consumer reads the
// granules with `sc1` loads and re-reads until every tag equals the tag of the producing phase of THIS launch. No flag, no drain,
// no fence, no finaliser hop — the first version of this kernel (counters + last-arriver finalisers) spent ~7 dependent memory
// round trips = 7-8 us on every hand-off (tools/mega_stamps.py). A sharded arrival counter per phase is only a HINT that tells
// consumers when a sweep is worth issuing (it is added to without ordering against the stores; the tags decide).
//
// The host (backend.cpp) fills an array of mega_phase records in device memory; the kernel walks it.
#pragma once

#include "kernels.h"
#include "rope_dev.h"

namespace mi355x {

constexpr int MEGA_MAX_GROUPS = 3;
constexpr int MEGA_WG_THREADS = 512;      // 8 waves, one workgroup per CU
enum mega_kind { MEGA_END = 0, MEGA_MM = 1, MEGA_ATTN = 2 };
// where a mat-vec phase's quantized activation comes from
enum mega_in { MIN_IMAGE = 0,       // a finished image (act_q8 layout) in global memory that an earlier KERNEL wrote: plain copy
               MIN_NORM_PLAIN = 1,  // x f32 (graph input, plain memory): RMS_NORM * w, quantize — in every consumer workgroup
               MIN_NORM_GRAN = 2,   // the same from x granules written by the previous phase of this launch
               MIN_PIECES = 3,
               MIN_QUANT_GRAN = 4 };// f32 granules of the previous phase, quantized as they are by every consumer workgroup (k <= 8192)    // image pieces (one per 256-element chunk, MEGA_PIECE_WORDS granules each) written by the previous phase's chunk owners / the attention phase

constexpr int MEGA_PIECE_WORDS = 80;      // Q8_K piece of one 256-element chunk: words 0..63 packed int8, 64 the scale d (f32), 65..72 the 16 int16 bsums, rest unused

struct mega_group {
    const char * W; const char * W2;
    float * dst; const float * res;                  // res: plain f32 residual (a graph input) ...
    const unsigned long long * res_gran;             // ... or the residual's granules (written by an earlier phase of this launch)
    unsigned long long * gran;                       // this group's output rows as granules (NULL: nobody in this launch reads them)
    uint16_t * st16; const int64_t * st_idx; int64_t st_row_elems;
    uint32_t row_stride; int m; int type; int epi; int st_mode; int pad;
};

struct mega_phase {
    int kind; int n_groups; int glu; int n_active;            // n_active: workgroups that take part
    int block_end[4];                                        // MEGA_MM: cumulative workgroup counts per group (unused = INT_MAX)
    // ---- input ----
    int in_mode; int k; int act_kind; int act_chunks; int off_d; int off_bs;      // image layout in LDS (act_q8 layout for n = 1)
    const char * act;                                        // MIN_IMAGE
    const float * x; const float * norm_w; float eps; int in_tag_phase;           // MIN_NORM_*; in_tag_phase: index of the phase whose tag the input granules carry
    const unsigned long long * x_gran;                       // MIN_NORM_GRAN
    float * norm_out;                                        // the RMS_NORM*w tensor (workgroup 0 writes it; may be NULL)
    const unsigned long long * pieces; int n_pieces; int pieces_tag_phase;       // MIN_PIECES
    const unsigned * wait; unsigned wait_target; int pad1;   // hint: sharded counter (8 words, 64 bytes apart) to reach wait_target before sweeping
    // ---- output ----
    unsigned * hint;                                         // every active workgroup adds 1 to its shard when its rows are stored
    // ---- chunk owners: before this phase reads its pieces, workgroup j < n_own quantizes chunk j of the PREVIOUS phase's output
    // (granules own_src, tag of phase own_src_tag_phase, worth sweeping once own_wait reaches own_wait_target), publishes the piece
    // into `pieces` with this phase's tag and adds 1 to its shard of hint2 (= this phase's `wait`) ----
    int n_own; int own_src_tag_phase; const unsigned long long * own_src; const unsigned * own_wait; unsigned own_wait_target; int pad2;
    unsigned long long * own_pieces; unsigned * hint2;
    // ---- epilogues ----
    fused_rope rope; const int32_t * pos;
    mega_group g[MEGA_MAX_GROUPS];
    // ---- MEGA_ATTN (build_attn_mha without flash attention, one token): q / k / v of the NEW cell come as granules of the QKV phase,
    // older cells from the f16 cache: k [hd, n_kv, n_head_kv], v (transposed cache) [n_kv, hd, n_head_kv]; mask [n_kv] f32;
    // dst [hd*n_head] f32 + the piece of each 256-element chunk (two heads) ----
    const unsigned long long * q_gran; const unsigned long long * k_gran; const unsigned long long * v_gran;
    const char * kc; size_t k_nb1, k_nb2; const char * vc; size_t v_nb1, v_nb2; const int64_t * cell_idx;
    const char * mask; float * attn_dst; float scale; int n_kv, n_head, n_head_kv, head_dim, mask_f16;
};

constexpr int MEGA_SIG_WORDS = 256;       // per phase: hint shards at words 0,16,..,112; hint2 shards at 128,144,..,240 (zeroed by a memset node before every launch)

bool mega_supported_types(const int * types, int n);      // are all these weight formats served by one persistent kernel
// `epoch`: a device word that counts launches (the tag base); `err`: a host-visible word the kernel sets when a wait gives up
void mega_launch(const mega_phase * prog_dev, int n_phases, int n_wg, unsigned * epoch, unsigned * err, size_t lds_bytes, hipStream_t stream);
int  mega_max_workgroups(void);           // one per CU of the current device

} // namespace mi355x
