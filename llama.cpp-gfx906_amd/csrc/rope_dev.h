// rope_dev.h — the NORM-mode rotary embedding of one pair (2i, 2i + 1), shared by the kernels that apply it in an epilogue
// (decode_fused.hip: grouped mat-vec; mmq.hip: prefill QKV launch and its split-k combine pass). Same formulas as elem.hip k_rope<false>
// (ggml_compute_forward_rope, YaRN: ggml/src/ggml-cpu/ops.cpp rope_yarn — cited in elem.hip).
#pragma once
#include <math.h>
#include "kernels.h"

namespace mi355x {

struct fused_rope {
    const int32_t * pos; const float * ff; int n_dims, head_dim, n_ctx_orig;
    float freq_scale, ext_factor, attn_factor, theta_scale, corr_lo, corr_hi;
    int neox;       // 0: NORM pairs (2i, 2i+1); 1: NEOX pairs (i, i + n_dims/2)
    const float * tab;   // (cos, sin)*mscale per pair index for the token at pos[0] (k_rope_table), or NULL
};

static __device__ __forceinline__ void rope_pair(const fused_rope & r, int pos, int row_in_head, float & x0, float & x1) {
    // NORM rope on the pair (2i, 2i+1) — same formulas as elem.hip k_rope<false>
    if (row_in_head >= r.n_dims) return;
    const int ip = row_in_head >> 1;
    const float theta_base = (float) pos*powf(r.theta_scale, (float) ip);
    const float theta_extrap = theta_base/(r.ff ? r.ff[ip] : 1.0f);
    float theta_interp = r.freq_scale*theta_extrap, theta = theta_interp, mscale = r.attn_factor;
    if (r.ext_factor != 0.0f) {
        const float y = ((float) ip - r.corr_lo)/fmaxf(0.001f, r.corr_hi - r.corr_lo);
        const float ramp_mix = (1.0f - fminf(1.0f, fmaxf(0.0f, y)))*r.ext_factor;
        theta = theta_interp*(1.0f - ramp_mix) + theta_extrap*ramp_mix;
        mscale *= 1.0f + 0.1f*logf(1.0f/r.freq_scale);
    }
    const float c = cosf(theta)*mscale, s = sinf(theta)*mscale;
    const float a = x0, b = x1;
    x0 = a*c - b*s;
    x1 = a*s + b*c;
}

static inline float rope_corr_dim_h(int n_dims, int n_ctx_orig, float n_rot, float base) {
    return n_dims*logf(n_ctx_orig/(n_rot*2*(float) M_PI))/(2*logf(base));
}
static inline fused_rope make_fused_rope(const mmvq_rope & rope) {
    fused_rope r;
    r.pos = rope.pos; r.ff = rope.freq_factors; r.n_dims = rope.p.n_dims; r.head_dim = rope.head_dim;
    r.n_ctx_orig = rope.p.n_ctx_orig; r.freq_scale = rope.p.freq_scale; r.ext_factor = rope.p.ext_factor;
    r.attn_factor = rope.p.attn_factor;
    r.theta_scale = powf(rope.p.freq_base, -2.0f/rope.p.n_dims);
    const float start = floorf(rope_corr_dim_h(rope.p.n_dims, rope.p.n_ctx_orig, rope.p.beta_fast, rope.p.freq_base));
    const float end   = ceilf (rope_corr_dim_h(rope.p.n_dims, rope.p.n_ctx_orig, rope.p.beta_slow, rope.p.freq_base));
    r.corr_lo = fmaxf(0.0f, start); r.corr_hi = fminf((float)(rope.p.n_dims - 1), end);
    r.neox = (rope.p.mode & 2) ? 1 : 0;
    r.tab = rope.table;
    return r;
}

} // namespace mi355x
