// gran_probe.hip — 8-byte sc1 granule stores by many waves, then (same launch, behind a counter) 16-byte sc1 buffer loads vs 8-byte sc1 loads
// hipcc --offload-arch=gfx950 -O3 tools/gran_probe.hip -o tools/gran_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
typedef int int4v __attribute__((ext_vector_type(4)));
__global__ void k_both(unsigned long long * g, unsigned * ctr, int * out16, int * out8, int n, int variant) {
    const int wave = (blockIdx.x*blockDim.x + threadIdx.x)/64, lane = threadIdx.x & 63;
    const int nw = gridDim.x*blockDim.x/64;
    if (lane == 0 && wave < n) {
        const unsigned long long v = ((unsigned long long) 0xABCD0000u << 32) | (unsigned)(1000 + wave);
        __hip_atomic_store(g + wave, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __syncthreads();
    if (threadIdx.x == 0) __hip_atomic_fetch_add(ctr, blockDim.x/64, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (blockIdx.x != 0) return;
    if (threadIdx.x < 64) {
        int spins = 0;
        while (__hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < (unsigned) nw && ++spins < (1 << 22)) __builtin_amdgcn_s_sleep(1);
        const int i = threadIdx.x;
        if (i*2 < n) {
            __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void *) g, (short) 0, (int) 0x7FFFFFFF, (int) 0x00020000);
            int4v a;
            if (variant == 0) a = __builtin_bit_cast(int4v, __builtin_amdgcn_raw_buffer_load_b128(r, i*16, 0, 16));
            else              a = __builtin_bit_cast(int4v, __builtin_amdgcn_raw_buffer_load_b128(r, i*16, 0, 0));
            out16[i*4 + 0] = a.x; out16[i*4 + 1] = a.y; out16[i*4 + 2] = a.z; out16[i*4 + 3] = a.w;
            const unsigned long long b0 = __hip_atomic_load(g + 2*i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT), b1 = __hip_atomic_load(g + 2*i + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            out8[i*4 + 0] = (int) b0; out8[i*4 + 1] = (int)(b0 >> 32); out8[i*4 + 2] = (int) b1; out8[i*4 + 3] = (int)(b1 >> 32);
        }
    }
}
int main() {
    const int n = 64;
    unsigned long long * g; unsigned * ctr; int * o16; int * o8; int h16[4*n], h8[4*n];
    hipMalloc(&g, n*8); hipMalloc(&ctr, 4); hipMalloc(&o16, 4*n*4); hipMalloc(&o8, 4*n*4);
    for (int variant = 0; variant < 2; variant++) for (int rep = 0; rep < 3; rep++) {
        hipMemset(g, 0, n*8); hipMemset(ctr, 0, 4); hipMemset(o16, 0, 4*n*4);
        k_both<<<8, 512>>>(g, ctr, o16, o8, n, variant);
        hipMemcpy(h16, o16, sizeof(h16), hipMemcpyDeviceToHost); hipMemcpy(h8, o8, sizeof(h8), hipMemcpyDeviceToHost);
        int bad16 = 0, bad8 = 0;
        for (int i = 0; i < n/2; i++) { if (h16[i*4] != 1000 + 2*i || h16[i*4 + 2] != 1001 + 2*i) bad16++; if (h8[i*4] != 1000 + 2*i || h8[i*4 + 2] != 1001 + 2*i) bad8++; }
        printf("variant %d (aux %s) rep %d: 16-byte loads wrong %d/%d, 8-byte loads wrong %d/%d; first: [%d %x | %d %x]\n", variant, variant ? "0" : "sc1", rep, bad16, n/2, bad8, n/2,
               h16[0], h16[1], h16[2], h16[3]);
    }
    return 0;
}
