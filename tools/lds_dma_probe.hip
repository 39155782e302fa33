// lds_dma_probe.hip — what read rate does a Q4_K-row stream reach when the bytes travel HBM -> LDS by global_load_lds_dwordx4
// (1 KiB contiguous per wave instruction, no VGPRs) instead of per-lane fragment loads into registers?
// Each wave walks row pairs (2 x 2304 B, k = 4096) with a grid stride, keeps DEPTH KiB-pieces in flight into its own LDS ring and
// reads every landed piece back with ds_read_b128 (xor-reduced, so nothing is optimised away). Compare with
// tools/read_pattern_probe.py (register fragments: 5.6-6.2 TB/s; fully coalesced register loads: 6.9-7.1 TB/s).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
typedef int int4v __attribute__((ext_vector_type(4)));

template <int DEPTH>
__global__ void __launch_bounds__(512) k_stream(const char * p, size_t n_pieces, unsigned * sink) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    char * ring = lds + (size_t) wave*DEPTH*1024;
    const size_t stride = (size_t) gridDim.x*8;
    int acc = 0;
    size_t pc = (size_t) blockIdx.x*8 + wave;          // piece index of this wave (interleaved over the grid)
    // prologue: DEPTH pieces in flight
    size_t pf = pc;
#pragma unroll
    for (int d = 0; d < DEPTH; d++) {
        const size_t q = pf < n_pieces ? pf : pc;
        __builtin_amdgcn_global_load_lds((const void __attribute__((address_space(1))) *) (p + q*1024 + lane*16),
                                         (void __attribute__((address_space(3))) *) (ring + d*1024), 16, 0, 0);
        pf += stride;
    }
    int slot = 0;
    for (; pc < n_pieces; pc += stride) {
        // wait until the oldest piece has landed: DEPTH-1 may stay in flight
        if (DEPTH == 1) __builtin_amdgcn_s_waitcnt(0x0f70 | 0);      // vmcnt(0)
        else asm volatile("s_waitcnt vmcnt(%0)" :: "n"(DEPTH - 1) : "memory");
        const int4v v = *(const int4v *) (ring + slot*1024 + lane*16);
        acc ^= v.x ^ v.y ^ v.z ^ v.w;
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");           // the slot is read before it is refilled
        const size_t q = pf < n_pieces ? pf : pc;
        __builtin_amdgcn_global_load_lds((const void __attribute__((address_space(1))) *) (p + q*1024 + lane*16),
                                         (void __attribute__((address_space(3))) *) (ring + slot*1024), 16, 0, 0);
        pf += stride;
        slot = slot + 1 == DEPTH ? 0 : slot + 1;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (acc == 0x12345678) *sink = acc;
}

template <int DEPTH> int run(const char * p, size_t bytes, unsigned * sink, hipStream_t s) {
    const size_t n_pieces = bytes/1024;
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    CK(hipFuncSetAttribute((const void *) k_stream<DEPTH>, hipFuncAttributeMaxDynamicSharedMemorySize, 8*DEPTH*1024));
    for (int rep = 0; rep < 3; rep++) {
        CK(hipEventRecord(e0, s));
        for (int i = 0; i < 5; i++) hipLaunchKernelGGL((k_stream<DEPTH>), dim3(256), dim3(512), 8*DEPTH*1024, s, p, n_pieces, sink);
        CK(hipEventRecord(e1, s)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        printf("depth %2d KiB/wave (%3d KiB LDS/CU): %7.0f GB/s\n", DEPTH, 8*DEPTH, 5.0*bytes/ms/1e6);
    }
    return 0;
}
int main() {
    hipStream_t s; CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    const size_t bytes = (size_t) 2 << 30;
    char * p; unsigned * sink; CK(hipMalloc(&p, bytes)); CK(hipMalloc(&sink, 256)); CK(hipMemset(p, 1, bytes));
    if (run<2>(p, bytes, sink, s)) return 1;
    if (run<4>(p, bytes, sink, s)) return 1;
    if (run<8>(p, bytes, sink, s)) return 1;
    if (run<16>(p, bytes, sink, s)) return 1;
    return 0;
}
