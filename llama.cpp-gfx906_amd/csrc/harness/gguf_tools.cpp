// gguf_tools.cpp — host-side pieces of the GGUF model path: a JSON description of a file (for tests and tooling) and the row dequantization
// the input layer needs. The reference keeps token_embd on the CPU and its CPU backend runs the GET_ROWS there (the input layer's
// device is the CPU: src/llama-model.cpp:1949-1965), so the embedding rows are decoded on the host here too.
// Block formats: csrc/blocks.h (gguf-py/gguf/quants.py).
#include "ggml.h"
#include "gguf_file.h"
#include "../blocks.h"

#include <math.h>
#include <stdio.h>

namespace mi355x {

static inline void k_scale_min(int j, const uint8_t * q, int & sc, int & m) {   // quants.py:479-501 (get_scale_min)
    if (j < 4) { sc = q[j] & 63; m = q[j + 4] & 63; }
    else       { sc = (q[j + 4] & 0xF) | ((q[j - 4] >> 6) << 4); m = (q[j + 4] >> 4) | ((q[j] >> 6) << 4); }
}

// one row of n elements of `type` -> f32; false: no host decoder for this type
bool dequant_row_host(int type, const uint8_t * src, float * dst, int64_t n) {
    switch (type) {
        case GGML_TYPE_F32: memcpy(dst, src, (size_t) n*4); return true;
        case GGML_TYPE_F16: for (int64_t i = 0; i < n; i++) { ggml_fp16_t h; memcpy(&h, src + 2*i, 2); dst[i] = ggml_fp16_to_fp32(h); } return true;
        case GGML_TYPE_BF16: for (int64_t i = 0; i < n; i++) { uint16_t h; memcpy(&h, src + 2*i, 2); const uint32_t u = (uint32_t) h << 16; memcpy(dst + i, &u, 4); } return true;
        case GGML_TYPE_Q4_0:
            for (int64_t b = 0; b < n/32; b++) {
                const block_q4_0 * x = (const block_q4_0 *) src + b;
                const float d = ggml_fp16_to_fp32(x->d);
                for (int j = 0; j < 16; j++) { dst[b*32 + j] = d*((x->qs[j] & 0xF) - 8); dst[b*32 + 16 + j] = d*((x->qs[j] >> 4) - 8); }
            }
            return true;
        case GGML_TYPE_Q8_0:
            for (int64_t b = 0; b < n/32; b++) {
                const block_q8_0 * x = (const block_q8_0 *) src + b;
                const float d = ggml_fp16_to_fp32(x->d);
                for (int j = 0; j < 32; j++) dst[b*32 + j] = d*x->qs[j];
            }
            return true;
        case GGML_TYPE_Q4_K:
            for (int64_t b = 0; b < n/256; b++) {
                const block_q4_K * x = (const block_q4_K *) src + b;
                const float d = ggml_fp16_to_fp32(x->d), dmin = ggml_fp16_to_fp32(x->dmin);
                for (int j = 0; j < 8; j++) {
                    int sc, m; k_scale_min(j, x->scales, sc, m);
                    const float dl = d*sc, ml = dmin*m;
                    const uint8_t * q = x->qs + 32*(j/2);
                    for (int e = 0; e < 32; e++) dst[b*256 + j*32 + e] = dl*((j & 1) ? q[e] >> 4 : q[e] & 0xF) - ml;
                }
            }
            return true;
        case GGML_TYPE_Q5_K:
            for (int64_t b = 0; b < n/256; b++) {
                const block_q5_K * x = (const block_q5_K *) src + b;
                const float d = ggml_fp16_to_fp32(x->d), dmin = ggml_fp16_to_fp32(x->dmin);
                for (int j = 0; j < 8; j++) {
                    int sc, m; k_scale_min(j, x->scales, sc, m);
                    const float dl = d*sc, ml = dmin*m;
                    const uint8_t * q = x->qs + 32*(j/2);
                    for (int e = 0; e < 32; e++) {
                        const int lo = (j & 1) ? q[e] >> 4 : q[e] & 0xF;
                        dst[b*256 + j*32 + e] = dl*(lo | (((x->qh[e] >> j) & 1) << 4)) - ml;
                    }
                }
            }
            return true;
        case GGML_TYPE_Q6_K:
            for (int64_t b = 0; b < n/256; b++) {
                const block_q6_K * x = (const block_q6_K *) src + b;
                const float d = ggml_fp16_to_fp32(x->d);
                for (int half = 0; half < 2; half++) {          // quants.py:554-572: 128 elements per half: ql 64 B, qh 32 B, 8 scales
                    const uint8_t * ql = x->ql + 64*half, * qh = x->qh + 32*half;
                    const int8_t * sc = x->scales + 8*half;
                    float * y = dst + b*256 + 128*half;
                    for (int l = 0; l < 32; l++) {
                        const int is = l/16;
                        const int q1 = (int)((ql[l]      & 0xF) | (((qh[l] >> 0) & 3) << 4)) - 32;
                        const int q2 = (int)((ql[l + 32] & 0xF) | (((qh[l] >> 2) & 3) << 4)) - 32;
                        const int q3 = (int)((ql[l]      >> 4)  | (((qh[l] >> 4) & 3) << 4)) - 32;
                        const int q4 = (int)((ql[l + 32] >> 4)  | (((qh[l] >> 6) & 3) << 4)) - 32;
                        y[l]      = d*sc[is + 0]*q1;
                        y[l + 32] = d*sc[is + 2]*q2;
                        y[l + 64] = d*sc[is + 4]*q3;
                        y[l + 96] = d*sc[is + 6]*q4;
                    }
                }
            }
            return true;
        default: return false;
    }
}

static void json_str(std::string & o, const std::string & s) {
    o += '"';
    for (unsigned char c : s) {
        if (c == '"' || c == '\\') { o += '\\'; o += (char) c; }
        else if (c < 0x20) { char b[8]; snprintf(b, sizeof(b), "\\u%04x", c); o += b; }
        else o += (char) c;
    }
    o += '"';
}
static void json_num(std::string & o, const gguf_value & v, uint64_t i, uint32_t t) {
    char b[40];
    if (t == GV_U64)      { uint64_t x; memcpy(&x, v.data + 8*i, 8); snprintf(b, sizeof(b), "%llu", (unsigned long long) x); }
    else if (t == GV_I64) { int64_t x;  memcpy(&x, v.data + 8*i, 8); snprintf(b, sizeof(b), "%lld", (long long) x); }
    else if (t == GV_BOOL) snprintf(b, sizeof(b), "%s", gguf_file::num(v, i, t) != 0 ? "true" : "false");
    else if (t == GV_F32 || t == GV_F64) {
        const double d = gguf_file::num(v, i, t);
        if (isfinite(d)) snprintf(b, sizeof(b), "%.17g", d); else snprintf(b, sizeof(b), "null");
    }
    else snprintf(b, sizeof(b), "%lld", (long long) gguf_file::num(v, i, t));
    o += b;
}

} // namespace mi355x

using namespace mi355x;

extern "C" {

// Describe a GGUF file as JSON into buf (NUL-terminated). Returns the length needed (excluding the NUL; call again with a larger
// buffer if it is >= cap), or -1 with the error text in buf. Arrays longer than 16 items are cut to their first 16 ("count" has the length).
// Tensors carry a 64-bit FNV-1a hash of their bytes so that a test can pin the data section without shipping a second copy.
GGML_API long long mi_gguf_describe(const char * path, char * buf, long long cap) {
    std::string o;
    try {
        gguf_file f; f.open(path);
        char b[128];
        snprintf(b, sizeof(b), "{\"version\": %u, \"alignment\": %llu, \"data_offset\": %llu, \"kv\": [", f.version,
                 (unsigned long long) f.alignment, (unsigned long long) f.data_offset);
        o += b;
        bool first = true;
        for (const std::string & k : f.keys) {
            const gguf_value & v = f.at(k);
            if (!first) o += ", ";
            first = false;
            o += "{\"key\": "; json_str(o, k);
            snprintf(b, sizeof(b), ", \"type\": %u, ", v.type); o += b;
            if (v.type == GV_STR) { o += "\"value\": "; json_str(o, v.strs[0]); }
            else if (v.type == GV_ARR) {
                snprintf(b, sizeof(b), "\"item_type\": %u, \"count\": %llu, \"value\": [", v.item_type, (unsigned long long) v.count); o += b;
                const uint64_t n = v.count < 16 ? v.count : 16;
                for (uint64_t i = 0; i < n; i++) {
                    if (i) o += ", ";
                    if (v.item_type == GV_STR) json_str(o, v.strs[(size_t) i]); else json_num(o, v, i, v.item_type);
                }
                o += "]";
            } else { o += "\"value\": "; json_num(o, v, 0, v.type); }
            o += "}";
        }
        o += "], \"tensors\": [";
        first = true;
        for (const gguf_tensor_info & t : f.tensors) {
            if (!first) o += ", ";
            first = false;
            o += "{\"name\": "; json_str(o, t.name);
            snprintf(b, sizeof(b), ", \"type\": %u, \"ne\": [", t.type); o += b;
            for (uint32_t d = 0; d < t.n_dims; d++) { snprintf(b, sizeof(b), "%s%lld", d ? ", " : "", (long long) t.ne[d]); o += b; }
            snprintf(b, sizeof(b), "], \"offset\": %llu", (unsigned long long) t.offset); o += b;
            const int64_t blck = t.type < GGML_TYPE_COUNT ? ggml_blck_size((enum ggml_type) t.type) : 0;
            if (blck > 0 && t.ne[0] % blck == 0) {
                const size_t nbytes = ggml_row_size((enum ggml_type) t.type, t.ne[0])*(size_t) t.ne[1]*(size_t) t.ne[2]*(size_t) t.ne[3];
                const uint8_t * p = f.tensor_data(t, nbytes);
                uint64_t h = 1469598103934665603ull;
                for (size_t i = 0; i < nbytes; i++) h = (h ^ p[i])*1099511628211ull;
                snprintf(b, sizeof(b), ", \"nbytes\": %llu, \"fnv1a\": \"%016llx\"", (unsigned long long) nbytes, (unsigned long long) h); o += b;
            }
            o += "}";
        }
        o += "]}";
    } catch (const std::exception & e) {
        if (buf && cap > 0) snprintf(buf, (size_t) cap, "%s", e.what());
        return -1;
    }
    if (buf && cap > 0) snprintf(buf, (size_t) cap, "%s", o.c_str());
    return (long long) o.size();
}

} // extern "C"
