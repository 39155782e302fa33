// decode_mega.h — program format of the PERSISTENT decode kernel (decode_mega.hip): one launch runs a whole sequence of the
// single-token phases a Llama-style decode graph consists of (norm + QKV + RoPE + KV store | attention | wo + residual |
// norm + gate/up + SwiGLU | down + residual, layer after layer, then norm + lm_head), one workgroup per CU, hand-offs between
// phases through write-through stores, arrival counters and sc1 loads — so that what separate launches pay per phase
// (kernel boundary, cold head, redundant activation quantization in every workgroup) is replaced by one finaliser per hand-off
// whose latency the consumers spend streaming their next weights.
//
// The host (backend.cpp) fills an array of mega_phase records in device memory; the kernel walks it.
#pragma once

#include "kernels.h"
#include "rope_dev.h"

namespace mi355x {

constexpr int MEGA_MAX_GROUPS = 3;
constexpr int MEGA_WG_THREADS = 512;      // 8 waves, one workgroup per CU
enum mega_kind { MEGA_END = 0, MEGA_MM = 1, MEGA_ATTN = 2, MEGA_FIN = 3 };
// what happens to a phase's output vector before the next phase reads it
enum mega_fin { MFIN_NONE = 0,     // every active workgroup adds 1 to `signal` when its rows are stored
                MFIN_NORM = 1,     // the workgroup whose arrival (on `arrive`) is last computes RMS_NORM * w of the whole vector, quantizes it
                                   // into the image and adds 1 to `signal`
                MFIN_CHUNK = 2 };  // per 256-row chunk: the workgroup whose arrival on the chunk's counter is last quantizes the chunk and adds 1 to `signal`

struct mega_group {
    const char * W; const char * W2;
    float * dst; const float * res;
    uint16_t * st16; const int64_t * st_idx; int64_t st_row_elems;
    uint32_t row_stride; int m; int type; int epi; int st_mode; int pad;
};

struct mega_phase {
    int kind; int n_groups; int glu; int n_active;            // n_active: workgroups that take part
    int block_end[4];                                        // MEGA_MM: cumulative workgroup counts per group (unused = INT_MAX)
    // input: the quantized activation image (act_q8 layout, n = 1) in global memory and the signal that says it is complete
    const char * act; int k; int act_kind; int act_chunks; int off_d; int off_bs; int pad1;
    const unsigned * wait; unsigned wait_target; int pad2;
    // output hand-off
    int fin_mode; int fin_k; int fin_kind; float fin_eps;
    unsigned * arrive;                                       // MFIN_NORM: arrival counter; MFIN_CHUNK: one counter per 256-row chunk
    unsigned * signal;                                       // what the next phase waits on (NULL: nobody waits)
    const float * fin_x; const float * fin_norm_w; float * fin_norm_out;
    char * fin_img; int fin_off_d; int fin_off_bs;
    // epilogues
    fused_rope rope; const int32_t * pos;
    mega_group g[MEGA_MAX_GROUPS];
    // MEGA_ATTN (build_attn_mha without flash attention, one token): q [hd, n_head] f32, k [hd, n_kv, n_head_kv] f16,
    // v (transposed cache) [n_kv, hd, n_head_kv] f16, mask [n_kv] f32; dst [hd*n_head] f32 + its quantized image (fin_*)
    const char * q; size_t q_nb2; const char * kc; size_t k_nb1, k_nb2; const char * vc; size_t v_nb1, v_nb2;
    const char * mask; float * attn_dst; float scale; int n_kv, n_head, n_head_kv, head_dim, mask_f16;
};

// words of signalling state per phase (zeroed by a memset node before every launch)
constexpr int MEGA_SIG_WORDS = 16;        // [0] arrive, [4] signal (own 16-byte pieces)
constexpr int MEGA_CHUNK_WORDS = 64;      // per MFIN_CHUNK phase: counters of up to 64 chunks (k <= 16384)

bool mega_supported_types(const int * types, int n);      // are all these weight formats served by one persistent kernel
// workspace: [n_phases][MEGA_SIG_WORDS] + chunk counters; `err` is a host-visible word the kernel sets when a wait gives up
void mega_launch(const mega_phase * prog_dev, int n_phases, int n_wg, unsigned * err, size_t lds_bytes, hipStream_t stream);
int  mega_max_workgroups(void);           // one per CU of the current device

} // namespace mi355x
