// stream_prefetch_probe.hip — two PRODUCT launches of the streamed mat-vec back to back (A then B, different weights): how long does B take when A's
// loader waves prefetch B's first bytes (kernels.h: mmvq_next), and when they do not? Links the product library; eager launches; B is timed by its dispatch's
// own start/stop events. A 1 GiB default-policy read between pairs evicts L2 and the Infinity Cache.
//   hipcc --offload-arch=gfx950 -O3 -I../include -I../include/ggml-compat -I../llama.cpp-gfx906_amd/csrc stream_prefetch_probe.hip -L../llama.cpp-gfx906_amd/lib -lggml-mi355x -lggml-base-compat -Wl,-rpath,'$ORIGIN/../llama.cpp-gfx906_amd/lib' -o stream_prefetch_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>
#include <algorithm>
#include "kernels.h"
using namespace mi355x;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s at line %d\n", #x, hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
typedef int int4v_ __attribute__((ext_vector_type(4)));
__global__ void __launch_bounds__(256) k_evict(const int4v_ * p, size_t n16, unsigned * sink) {
    int acc = 0;
    for (size_t i = (size_t) blockIdx.x*256 + threadIdx.x; i < n16; i += (size_t) gridDim.x*256) { const int4v_ v = p[i]; acc ^= v.x ^ v.y ^ v.z ^ v.w; }
    if (acc == 0x12345678) *sink = 1;
}
int main() {
    const int64_t K = 4096;
    struct shape { const char * name; int64_t mA, mB; } shapes[] = { { "A = wo-like 4096 rows, B = wo-like 4096 rows", 4096, 4096 }, { "A = wo-like 4096 rows, B = 28672 rows (gate+up sized, plain epilogue)", 4096, 28672 } };
    const size_t row = K/256*144;
    char * WA; char * WB; float * x; float * dA; float * dB; char * E; unsigned * sink;
    const size_t EV = (size_t) 1 << 30;
    CK(hipMalloc(&WA, 4096*row + 4096)); CK(hipMalloc(&WB, 28672*row + 4096)); CK(hipMalloc(&x, K*4)); CK(hipMalloc(&dA, 28672*4)); CK(hipMalloc(&dB, 28672*4)); CK(hipMalloc(&E, EV)); CK(hipMalloc(&sink, 4));
    // valid-looking blocks: d = 1.0 (f16 0x3C00), dmin = 0, everything else small
    { std::vector<unsigned char> h(28672*row, 0x11); for (size_t b = 0; b < h.size(); b += 144) { h[b] = 0x00; h[b + 1] = 0x1C; h[b + 2] = 0; h[b + 3] = 0; }
      CK(hipMemcpy(WA, h.data(), 4096*row, hipMemcpyHostToDevice)); CK(hipMemcpy(WB, h.data(), 28672*row, hipMemcpyHostToDevice)); }
    { std::vector<float> hx(K, 0.5f); CK(hipMemcpy(x, hx.data(), K*4, hipMemcpyHostToDevice)); }
    CK(hipMemset(E, 3, EV));
    hipStream_t st; CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (const shape & s : shapes) {
        mmvq_group gA = {}; gA.W = WA; gA.row_stride = row; gA.m = (int) s.mA; gA.type = T_Q4_K; gA.dst = dA; gA.epi = EPI_NONE;
        mmvq_group gB = {}; gB.W = WB; gB.row_stride = row; gB.m = (int) s.mB; gB.type = T_Q4_K; gB.dst = dB; gB.epi = EPI_NONE;
        mmvq_input in = {}; in.mode = PRO_QUANT; in.x = x; in.act_kind = T_Q8_K;
        if (!mul_mat_vec_q_stream_takes(&gA, 1, K, in, nullptr) || !mul_mat_vec_q_stream_takes(&gB, 1, K, in, nullptr)) { printf("not taken\n"); return 1; }
        // B's descriptor: launch it once
        const char * kn = nullptr; mmvq_next descB;
        mul_mat_vec_q_stream(&gB, 1, K, in, nullptr, st, nullptr, nullptr, &kn); mul_mat_vec_q_stream_last_desc(&descB);
        CK(hipStreamSynchronize(st));
        printf("%s (B's descriptor: %d group(s), %d bytes per workgroup)\n", s.name, descB.n_groups, descB.bytes);
        for (int variant = 0; variant < 3; variant++) {      // 0: no prefetch; 1: A prefetches B; 2: again no prefetch
            const bool pf = variant == 1;
            std::vector<float> tb;
            for (int r = 0; r < 12; r++) {
                hipLaunchKernelGGL(k_evict, dim3(2048), dim3(256), 0, st, (const int4v_ *) E, EV/16, sink);
                mmvq_next none = {}; mul_mat_vec_q_stream_set_next(pf ? &descB : &none);
                mul_mat_vec_q_stream(&gA, 1, K, in, nullptr, st, nullptr, nullptr, &kn);
                mul_mat_vec_q_stream_set_next(&none);
                mul_mat_vec_q_stream(&gB, 1, K, in, nullptr, st, e0, e1, &kn);
                CK(hipStreamSynchronize(st));
                float ms; CK(hipEventElapsedTime(&ms, e0, e1));
                if (r >= 2) tb.push_back(ms*1e3f);
            }
            std::sort(tb.begin(), tb.end());
            printf("  %-22s B: median %.2f us, min %.2f us\n", pf ? "A prefetches B" : "no prefetch", tb[tb.size()/2], tb[0]);
        }
    }
    return 0;
}
